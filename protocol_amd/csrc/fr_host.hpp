// Host-side BN254 scalar field (the reference's `N = bn256::Fr`, eigentrust-zk/src/lib.rs:79-83).
//
// The reference reaches this arithmetic through halo2curves; everything observable at its API is
// the canonical 32-byte little-endian integer modulo r (eigentrust-zk/src/utils.rs:327-329), so the
// representation here is free: 4 x 64-bit limbs, Montgomery form with R = 2^256.
// Used by the set API and the graph builder (row sums, batch inversion, per-source scale factors);
// the per-edge arithmetic of the hot loop runs on the GPU (fr_dev.hpp).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace et {

typedef unsigned __int128 u128;

struct FrRaw {  // canonical little-endian integer, value < r  (== et_fr at the C boundary)
    uint64_t l[4];
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const FrRaw& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const FrRaw& o) const { return !(*this == o); }
};

static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};  // 2^512 mod r
static const uint64_t FR_N0 = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64

inline bool fr_raw_is_canonical(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] < FR_MOD[i]) return true;
        if (a[i] > FR_MOD[i]) return false;
    }
    return false;
}

class Fr {  // Montgomery form
  public:
    uint64_t l[4];
    Fr() : l{0, 0, 0, 0} {}
    static Fr zero() { return Fr(); }
    static Fr one() { return from_u64(1); }
    static Fr from_raw(const FrRaw& a) {
        Fr x, r2;
        std::memcpy(x.l, a.l, 32);
        std::memcpy(r2.l, FR_R2, 32);
        return x * r2;
    }
    static Fr from_u64(uint64_t v) { FrRaw a{{v, 0, 0, 0}}; return from_raw(a); }
    static Fr from_u128(uint64_t lo, uint64_t hi) { FrRaw a{{lo, hi, 0, 0}}; return from_raw(a); }
    FrRaw to_raw() const {
        Fr one_raw;
        one_raw.l[0] = 1;
        Fr c = *this * one_raw;
        FrRaw out;
        std::memcpy(out.l, c.l, 32);
        return out;
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fr& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const Fr& o) const { return !(*this == o); }

    Fr operator+(const Fr& b) const {
        Fr c;
        u128 cy = 0;
        for (int i = 0; i < 4; ++i) {
            u128 t = (u128)l[i] + b.l[i] + cy;
            c.l[i] = (uint64_t)t;
            cy = t >> 64;
        }
        if (cy || !fr_raw_is_canonical(c.l)) c.sub_mod();
        return c;
    }
    Fr operator-(const Fr& b) const {
        Fr c;
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)l[i] - b.l[i] - br;
            c.l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
        if (br) {
            u128 cy = 0;
            for (int i = 0; i < 4; ++i) {
                u128 t = (u128)c.l[i] + FR_MOD[i] + cy;
                c.l[i] = (uint64_t)t;
                cy = t >> 64;
            }
        }
        return c;
    }
    Fr operator*(const Fr& b) const {  // CIOS
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 cy = 0;
            for (int j = 0; j < 4; ++j) {
                u128 x = (u128)l[j] * b.l[i] + t[j] + cy;
                t[j] = (uint64_t)x;
                cy = x >> 64;
            }
            u128 x = (u128)t[4] + cy;
            t[4] = (uint64_t)x;
            t[5] = (uint64_t)(x >> 64);
            uint64_t m = t[0] * FR_N0;
            cy = ((u128)m * FR_MOD[0] + t[0]) >> 64;
            for (int j = 1; j < 4; ++j) {
                u128 y = (u128)m * FR_MOD[j] + t[j] + cy;
                t[j - 1] = (uint64_t)y;
                cy = y >> 64;
            }
            x = (u128)t[4] + cy;
            t[3] = (uint64_t)x;
            t[4] = t[5] + (uint64_t)(x >> 64);
        }
        Fr c;
        std::memcpy(c.l, t, 32);
        if (t[4] || !fr_raw_is_canonical(c.l)) c.sub_mod();
        return c;
    }
    Fr& operator+=(const Fr& b) { *this = *this + b; return *this; }
    Fr& operator*=(const Fr& b) { *this = *this * b; return *this; }

    // a^(r-2).  Zero maps to zero, which is the reference's `invert().unwrap_or(ZERO)`
    // (dynamic_sets/native.rs:308).
    Fr invert_or_zero() const {
        uint64_t e[4];
        std::memcpy(e, FR_MOD, 32);
        e[0] -= 2;
        Fr acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc * acc;
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
        }
        return acc;
    }
    Fr pow2k(unsigned k) const {  // this * 2^k
        Fr c = *this;
        for (unsigned i = 0; i < k; ++i) c = c + c;
        return c;
    }

  private:
    void sub_mod() {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)l[i] - FR_MOD[i] - br;
            l[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
};

// Montgomery's trick: inverts every non-zero element with one field inversion; zeros stay zero.
inline void fr_batch_invert(std::vector<Fr>& v) {
    std::vector<Fr> prefix(v.size());
    Fr acc = Fr::one();
    for (size_t i = 0; i < v.size(); ++i) {
        prefix[i] = acc;
        if (!v[i].is_zero()) acc = acc * v[i];
    }
    Fr inv = acc.invert_or_zero();
    for (size_t i = v.size(); i-- > 0;) {
        if (v[i].is_zero()) continue;
        Fr vi = v[i];
        v[i] = inv * prefix[i];
        inv = inv * vi;
    }
}

}  // namespace et
