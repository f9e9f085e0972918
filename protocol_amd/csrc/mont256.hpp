// 256-bit Montgomery arithmetic with a run-time modulus on the host: the secp256k1 base and scalar
// fields of the signature code, and the conversion of constants for the device kernels.
#pragma once
#include <cstdint>
#include <cstring>

namespace et {

typedef unsigned __int128 u128m;

struct U256h {
    uint64_t l[4];
    bool operator==(const U256h& o) const { return !std::memcmp(l, o.l, 32); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
};
inline int cmp256(const U256h& a, const U256h& b) {
    for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] < b.l[i] ? -1 : 1;
    return 0;
}
inline uint64_t add256(U256h& r, const U256h& a, const U256h& b) {
    u128m c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128m)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
inline uint64_t sub256(U256h& r, const U256h& a, const U256h& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128m t = (u128m)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)t;
        br = (uint64_t)(t >> 64) & 1;
    }
    return br;
}

struct Mont {
    U256h m, r2, one;   // modulus, R^2 mod m, R mod m
    uint64_t n0;        // -m^-1 mod 2^64
    explicit Mont(const U256h& mod) : m(mod) {
        uint64_t inv = 1;
        for (int i = 0; i < 6; ++i) inv *= 2 - m.l[0] * inv;   // Newton: m^-1 mod 2^64
        n0 = ~inv + 1;
        // R mod m by doubling 1 two hundred fifty-six times, R^2 by another 256
        U256h x{{1, 0, 0, 0}};
        for (int i = 0; i < 512; ++i) {
            if (i == 256) one = x;
            U256h d;
            uint64_t c = add256(d, x, x);
            U256h t;
            if (c || cmp256(d, m) >= 0) { sub256(t, d, m); d = t; }
            x = d;
        }
        r2 = x;
    }
    U256h mul(const U256h& a, const U256h& b) const {   // Montgomery product (CIOS)
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128m c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128m)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t q = t[0] * n0;
            c = (u128m)q * m.l[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) { c += (u128m)q * m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        U256h r{{t[0], t[1], t[2], t[3]}}, d;
        if (t[4] || cmp256(r, m) >= 0) { sub256(d, r, m); r = d; }
        return r;
    }
    U256h to(const U256h& a) const { return mul(a, r2); }
    U256h from(const U256h& a) const { return mul(a, U256h{{1, 0, 0, 0}}); }
    U256h add(const U256h& a, const U256h& b) const {
        U256h r, d;
        uint64_t c = add256(r, a, b);
        if (c || cmp256(r, m) >= 0) { sub256(d, r, m); r = d; }
        return r;
    }
    U256h sub(const U256h& a, const U256h& b) const {
        U256h r, d;
        if (sub256(r, a, b)) { add256(d, r, m); r = d; }
        return r;
    }
    U256h neg(const U256h& a) const { return a.is_zero() ? a : sub(U256h{{0, 0, 0, 0}}, a); }
    U256h pow(const U256h& a, const U256h& e) const {
        U256h r = one;
        for (int i = 255; i >= 0; --i) {
            r = mul(r, r);
            if ((e.l[i / 64] >> (i % 64)) & 1) r = mul(r, a);
        }
        return r;
    }
    U256h inv(const U256h& a) const {   // a^(m-2), m prime
        U256h e, two{{2, 0, 0, 0}};
        sub256(e, m, two);
        return pow(a, e);
    }
};


static const U256h SECP_P{{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
static const U256h SECP_N{{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const U256h SECP_GX{{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
static const U256h SECP_GY{{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};

}  // namespace et
