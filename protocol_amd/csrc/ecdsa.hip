// Batched attester recovery on the GPU: one lane per signed attestation record.
//
// What the reference does once per record while it builds the participant set
// (eigentrust/src/lib.rs:351-358 -> attestation.rs:215-239):
//   hash   = Poseidon5(about, domain, value, message, 0)[0]              (dynamic_sets/native.rs:96-104)
//   R      = the curve point with x = r and the parity rec_id            (ecdsa/native.rs:307-316)
//   Q      = r^-1 * (s * R - hash * G)                                   (ecdsa/native.rs:318-320)
//   address = last 20 bytes of Keccak-256(x(Q) || y(Q)), big-endian       (ecdsa/native.rs:90-111, eth.rs:70-74)
// The host path (crypto.cpp) does the same per record on CPU threads and is what small sets use; this
// kernel is for set building at scale.  By construction Q satisfies the verification equation
// x(hash/s * G + r/s * Q) = r whenever it is not the point at infinity, so the sanity check of
// ecdsa/native.rs:322-327 cannot fail here and is not repeated per lane.
//
// The secp256k1 scalar field n and the bn256 scalar field use one Montgomery multiplication over
// 8 x 32-bit limbs with R = 2^256, the moduli in constant memory; the secp256k1 base field p, where
// most of the work is, uses plain residues and the fold 2^256 = 2^32 + 977 (mod p).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "crypto.hpp"
#include "engine.hpp"
#include "fr_host.hpp"
#include "mont256.hpp"

namespace et {

#define HIP_TRY_E(expr)                                                               \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));             \
            return _e == hipErrorNoDevice ? ET_E_NO_DEVICE : ET_E_HIP;                \
        }                                                                             \
    } while (0)

struct F256 {
    uint32_t v[8];
};
struct ModConst {
    uint32_t m[8];     // modulus
    uint32_t r2[8];    // 2^512 mod m
    uint32_t one[8];   // 2^256 mod m
    uint32_t n0;       // -m^-1 mod 2^32
    uint32_t pad[7];
};
enum { MOD_P = 0, MOD_N = 1, MOD_R = 2 };

struct RecoverConsts {
    ModConst mod[3];
    F256 gx, gy;          // generator, plain form mod p
    F256 seven;           // 7
    // the curve's endomorphism (x, y) -> (beta * x, y) = lambda * (x, y): a scalar k splits into k1 + k2 * lambda
    // with |k1|, |k2| < 2^128 (lattice basis (a1, b1), (a2, b2) of {(x, y): x + y * lambda = 0 mod n}, nb1 = -b1,
    // g1 = round(2^384 * b2 / n), g2 = round(2^384 * nb1 / n)), which halves the doublings of a scalar multiplication
    F256 beta, g1, g2, a1, a2, nb1, b2;
    // Poseidon 5x5 (poseidon/native/mod.rs:34-97), Montgomery form mod r, with the 60 partial rounds in their
    // sparse form: the matrix of a partial round factors into a sparse one (first row, first column, identity)
    // times one that leaves element 0 alone and therefore commutes with the partial S-box; pushing those
    // factors back to the last full round before them leaves 9 instead of 25 multiplications per partial round.
    F256 rc_full[8 * 5];      // constants of the 8 full rounds; those of the first round after the partial ones include what the partial rounds carried forward
    F256 mds[25];
    F256 mds_pre[25];         // matrix of the last full round before the partial rounds
    F256 rc_first[5];         // added to all five elements before the first partial round
    F256 rc0[60];             // added to element 0 before the S-box of partial round k (k >= 1)
    F256 sp_row[60 * 5];      // first row of the sparse matrix of partial round k
    F256 sp_col[60 * 4];      // its first column below the corner
};

#define EC_DEV __host__ __device__ __forceinline__

EC_DEV bool f_is_zero(const F256& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
EC_DEV bool f_eq(const F256& a, const F256& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
EC_DEV bool f_geq(const F256& a, const uint32_t* m) {   // a >= m
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.v[i] - m[i] - br;
        br = (d >> 32) & 1;
    }
    return br == 0;
}
// a - m when `force` is set or a >= m, else a: difference and borrow in one pass, limbs chosen without a branch.
// (The form `if (a >= m) a = a - m;` on the whole struct came out wrong from the device compiler at -O2 and -O3 in
// some inlining contexts: sums off by the modulus' low limb.  Every reduction here selects limb by limb instead.)
EC_DEV F256 f_cond_sub(const F256& a, const uint32_t* m, bool force) {
    F256 d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - m[i] - br;
        d.v[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    const bool take = force || br == 0;
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = take ? d.v[i] : a.v[i];
    return r;
}
EC_DEV F256 f_add(const F256& a, const F256& b, const ModConst& M) {
    F256 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    return f_cond_sub(r, M.m, c != 0);
}
EC_DEV F256 f_sub(const F256& a, const F256& b, const ModConst& M) {
    F256 d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
        d.v[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    // a < b: add the modulus back (selected per limb, no branch)
    F256 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)d.v[i] + (br ? M.m[i] : 0u) + c;
        r.v[i] = (uint32_t)t;
        c = t >> 32;
    }
    return r;
}
// Montgomery product a * b * 2^-256 mod m (coarsely integrated operand scanning)
EC_DEV F256 f_mul(const F256& a, const F256& b, const ModConst& M) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t v = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)v;
            c = v >> 32;
        }
        uint64_t v = (uint64_t)t[8] + c;
        t[8] = (uint32_t)v;
        t[9] = (uint32_t)(v >> 32);
        const uint32_t q = t[0] * M.n0;
        v = (uint64_t)q * M.m[0] + t[0];
        c = v >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            v = (uint64_t)q * M.m[j] + t[j] + c;
            t[j - 1] = (uint32_t)v;
            c = v >> 32;
        }
        v = (uint64_t)t[8] + c;
        t[7] = (uint32_t)v;
        t[8] = t[9] + (uint32_t)(v >> 32);
    }
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return f_cond_sub(r, M.m, t[8] != 0);
}
// sum_{k<5} a[k] * b[k] * 2^-256 mod m: the five products are added unreduced (5 m^2 < 2^515 fits 17 limbs)
// and reduced once -- one Montgomery reduction per row of the Poseidon matrix instead of five
EC_DEV F256 f_dot5(const F256* a, const F256* b, const ModConst& M) {
    uint32_t t[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t v = (uint64_t)a[k].v[j] * b[k].v[i] + t[i + j] + c;
                t[i + j] = (uint32_t)v;
                c = v >> 32;
            }
#pragma unroll
            for (int j = i + 8; j < 17; ++j) {
                uint64_t v = (uint64_t)t[j] + c;
                t[j] = (uint32_t)v;
                c = v >> 32;
            }
        }
    }
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t q = t[i] * M.n0;
        uint64_t v = (uint64_t)q * M.m[0] + t[i];
        uint64_t c = v >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            v = (uint64_t)q * M.m[j] + t[i + j] + c;
            t[i + j] = (uint32_t)v;
            c = v >> 32;
        }
        v = (uint64_t)t[i + 8] + c + carry;
        t[i + 8] = (uint32_t)v;
        carry = (uint32_t)(v >> 32);
    }
    // (5 m^2 + 2^256 m) / 2^256 < 2 m < 2^256 for a modulus below 0.2 * 2^256 (the bn256 scalar field, the only
    // one this is used with): the quotient fits eight limbs and one subtraction brings it below m
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[8 + i];
    return f_cond_sub(r, M.m, false);
}
EC_DEV F256 f_load_mod(const uint32_t* p) {
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = p[i];
    return r;
}
EC_DEV F256 f_to_mont(const F256& a, const ModConst& M) { return f_mul(a, f_load_mod(M.r2), M); }
EC_DEV F256 f_from_mont(const F256& a, const ModConst& M) {
    F256 one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one.v[i] = i == 0 ? 1u : 0u;
    return f_mul(a, one, M);
}

// ---- secp256k1 base field, p = 2^256 - 2^32 - 977, plain (non-Montgomery) representation ------------------
// 2^256 = 2^32 + 977 (mod p): the high half of a 512-bit product folds into the low half with eight
// multiply-adds by 977 and a one-limb shift, instead of the 64 of a Montgomery reduction.
EC_DEV F256 fp_reduce512(const uint32_t* t, const ModConst& P) {
    uint32_t r[10];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                 // r = lo + hi * 977
        uint64_t v = (uint64_t)t[8 + j] * 977u + t[j] + c;
        r[j] = (uint32_t)v;
        c = v >> 32;
    }
    r[8] = (uint32_t)c;
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                 // r += hi << 32
        uint64_t v = (uint64_t)r[j + 1] + t[8 + j] + c;
        r[j + 1] = (uint32_t)v;
        c = v >> 32;
    }
    r[9] = (uint32_t)c;                           // 0 or 1
    // second fold: top = r[8] + r[9] * 2^32 times (2^32 + 977)
    const uint64_t a0 = (uint64_t)r[8] * 977u;
    const uint64_t a1 = (uint64_t)r[8] + (uint64_t)r[9] * 977u + (a0 >> 32);
    uint64_t v = (uint64_t)r[0] + (uint32_t)a0;
    r[0] = (uint32_t)v; c = v >> 32;
    v = (uint64_t)r[1] + (uint32_t)a1 + c;
    r[1] = (uint32_t)v; c = v >> 32;
    v = (uint64_t)r[2] + (a1 >> 32) + r[9] + c;
    r[2] = (uint32_t)v; c = v >> 32;
#pragma unroll
    for (int j = 3; j < 8; ++j) { v = (uint64_t)r[j] + c; r[j] = (uint32_t)v; c = v >> 32; }
    // a carry out of 2^256 leaves a tiny remainder: one more 2^32 + 977 cannot carry again
    const uint32_t k = (uint32_t)c;
    v = (uint64_t)r[0] + (k ? 977u : 0u);
    r[0] = (uint32_t)v; c = v >> 32;
    v = (uint64_t)r[1] + k + c;
    r[1] = (uint32_t)v; c = v >> 32;
#pragma unroll
    for (int j = 2; j < 8; ++j) { v = (uint64_t)r[j] + c; r[j] = (uint32_t)v; c = v >> 32; }
    F256 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = r[i];
    return f_cond_sub(o, P.m, false);
}
// (top:lohi) += a * b: the multiply-add takes the 64-bit column sum as its own addend and its carry-out is counted in
// the third word, two instructions per limb product
EC_DEV void ec_col_mac(uint64_t& lohi, uint32_t& top, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lohi), "+v"(top) : "v"(a), "v"(b) : "vcc");
#else       // the host build of the same functions (the checker of the device results)
    const uint64_t r = (uint64_t)a * b + lohi;
    top += r < lohi ? 1u : 0u;
    lohi = r;
#endif
}
EC_DEV F256 fp_mul(const F256& a, const F256& b, const ModConst& P) {
    uint32_t t[16];
    uint64_t lohi = 0;
    uint32_t top = 0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {                // column k of the product, complete before the next one starts
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k - j >= 0 && k - j < 8) ec_col_mac(lohi, top, a.v[j], b.v[k - j]);
        t[k] = (uint32_t)lohi;
        lohi = (lohi >> 32) | ((uint64_t)top << 32);
        top = 0;
    }
    t[15] = (uint32_t)lohi;
    return fp_reduce512(t, P);
}
EC_DEV F256 fp_sqr(const F256& a, const ModConst& P) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {                 // products of distinct limbs, once
        uint64_t c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; ++j) { uint64_t v = (uint64_t)a.v[i] * a.v[j] + t[i + j] + c; t[i + j] = (uint32_t)v; c = v >> 32; }
        t[i + 8] = (uint32_t)c;
    }
    uint32_t top = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) { const uint32_t nv = (t[k] << 1) | top; top = t[k] >> 31; t[k] = nv; }   // twice
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                 // plus the squares
        uint64_t v = (uint64_t)a.v[i] * a.v[i] + t[2 * i] + c;
        t[2 * i] = (uint32_t)v;
        v = (uint64_t)t[2 * i + 1] + (v >> 32);
        t[2 * i + 1] = (uint32_t)v;
        c = v >> 32;
    }
    return fp_reduce512(t, P);
}
EC_DEV F256 fp_sqr_n(F256 a, int n, const ModConst& P) {
    for (int i = 0; i < n; ++i) a = fp_sqr(a, P);
    return a;
}
// a^(2^223 - 1) and, on the way, a^(2^22 - 1), a^3: the exponents p - 2 and (p + 1) / 4 are 223 ones, a zero,
// 22 ones and a short tail, so both powers take 255 squarings and 15 multiplications
struct FpLadder {
    F256 x2, x22, x223;
};
EC_DEV FpLadder fp_ladder(const F256& a, const ModConst& P) {
    FpLadder L;
    L.x2 = fp_mul(fp_sqr(a, P), a, P);
    const F256 x3 = fp_mul(fp_sqr(L.x2, P), a, P);
    const F256 x6 = fp_mul(fp_sqr_n(x3, 3, P), x3, P);
    const F256 x9 = fp_mul(fp_sqr_n(x6, 3, P), x3, P);
    const F256 x11 = fp_mul(fp_sqr_n(x9, 2, P), L.x2, P);
    L.x22 = fp_mul(fp_sqr_n(x11, 11, P), x11, P);
    const F256 x44 = fp_mul(fp_sqr_n(L.x22, 22, P), L.x22, P);
    const F256 x88 = fp_mul(fp_sqr_n(x44, 44, P), x44, P);
    const F256 x176 = fp_mul(fp_sqr_n(x88, 88, P), x88, P);
    const F256 x220 = fp_mul(fp_sqr_n(x176, 44, P), x44, P);
    L.x223 = fp_mul(fp_sqr_n(x220, 3, P), x3, P);
    return L;
}
// a^(p - 2): p - 2 = [223 ones] 0 [22 ones] 0000 1 0 11 0 1
EC_DEV F256 fp_inv(const F256& a, const ModConst& P) {
    const FpLadder L = fp_ladder(a, P);
    F256 t = fp_mul(fp_sqr_n(L.x223, 23, P), L.x22, P);
    t = fp_mul(fp_sqr_n(t, 5, P), a, P);
    t = fp_mul(fp_sqr_n(t, 3, P), L.x2, P);
    return fp_mul(fp_sqr_n(t, 2, P), a, P);
}
// a^((p + 1) / 4), a square root of a when it has one: (p + 1) / 4 = [223 ones] 0 [22 ones] 0000 11 00
EC_DEV F256 fp_sqrt(const F256& a, const ModConst& P) {
    const FpLadder L = fp_ladder(a, P);
    F256 t = fp_mul(fp_sqr_n(L.x223, 23, P), L.x22, P);
    t = fp_mul(fp_sqr_n(t, 6, P), L.x2, P);
    return fp_sqr_n(t, 2, P);
}

// Jacobian points on y^2 = x^3 + 7 over F_p, plain coordinates; z == 0 is the point at infinity
struct JPt {
    F256 x, y, z;
};
EC_DEV JPt j_double(const JPt& p, const ModConst& P) {
    if (f_is_zero(p.z) || f_is_zero(p.y)) { JPt o = p; for (int i = 0; i < 8; ++i) o.z.v[i] = 0; return o; }
    F256 a = fp_sqr(p.x, P), b = fp_sqr(p.y, P), c = fp_sqr(b, P);
    F256 t = f_add(p.x, b, P);
    F256 d = f_sub(f_sub(fp_sqr(t, P), a, P), c, P);
    d = f_add(d, d, P);
    F256 e = f_add(f_add(a, a, P), a, P), f = fp_sqr(e, P);
    JPt r;
    r.x = f_sub(f, f_add(d, d, P), P);
    F256 c8 = f_add(c, c, P);
    c8 = f_add(c8, c8, P);
    c8 = f_add(c8, c8, P);
    r.y = f_sub(fp_mul(e, f_sub(d, r.x, P), P), c8, P);
    F256 yz = fp_mul(p.y, p.z, P);
    r.z = f_add(yz, yz, P);
    return r;
}
// p + (qx, qy) with the second point affine
EC_DEV JPt j_add_affine(const JPt& p, const F256& qx, const F256& qy, const ModConst& P) {
    if (f_is_zero(p.z)) {
        JPt o;
        o.x = qx; o.y = qy;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.z.v[i] = i == 0 ? 1u : 0u;
        return o;
    }
    F256 z1z1 = fp_sqr(p.z, P);
    F256 u2 = fp_mul(qx, z1z1, P), s2 = fp_mul(fp_mul(qy, p.z, P), z1z1, P);
    if (f_eq(p.x, u2)) {
        if (f_eq(p.y, s2)) return j_double(p, P);
        JPt o = p;
        for (int i = 0; i < 8; ++i) o.z.v[i] = 0;
        return o;
    }
    F256 h = f_sub(u2, p.x, P), r = f_sub(s2, p.y, P);
    F256 hh = fp_sqr(h, P), hhh = fp_mul(hh, h, P), v = fp_mul(p.x, hh, P);
    JPt o;
    o.x = f_sub(f_sub(fp_sqr(r, P), hhh, P), f_add(v, v, P), P);
    o.y = f_sub(fp_mul(r, f_sub(v, o.x, P), P), fp_mul(p.y, hhh, P), P);
    o.z = fp_mul(p.z, h, P);
    return o;
}

// a^(n - 2) mod n for the secp256k1 group order n (Montgomery form in and out).  n - 2 is 127 ones, a zero and the
// 128 bits of the order's low half minus 2: a ladder of a^(2^k - 1) for the ones (126 squarings, 12
// multiplications), then the low half two bits at a time against a, a^2, a^3.
EC_DEV F256 fn_sqr_n(F256 a, int k, const ModConst& N) {
    for (int i = 0; i < k; ++i) a = f_mul(a, a, N);
    return a;
}
EC_DEV F256 fn_inv(const F256& a, const ModConst& N) {
    const F256 a2 = f_mul(fn_sqr_n(a, 1, N), a, N);              // 2^2 - 1
    const F256 a3 = f_mul(fn_sqr_n(a2, 1, N), a, N);             // 2^3 - 1
    const F256 a6 = f_mul(fn_sqr_n(a3, 3, N), a3, N);
    const F256 a7 = f_mul(fn_sqr_n(a6, 1, N), a, N);
    const F256 a14 = f_mul(fn_sqr_n(a7, 7, N), a7, N);
    const F256 a15 = f_mul(fn_sqr_n(a14, 1, N), a, N);
    const F256 a30 = f_mul(fn_sqr_n(a15, 15, N), a15, N);
    const F256 a31 = f_mul(fn_sqr_n(a30, 1, N), a, N);
    const F256 a62 = f_mul(fn_sqr_n(a31, 31, N), a31, N);
    const F256 a63 = f_mul(fn_sqr_n(a62, 1, N), a, N);
    const F256 a126 = f_mul(fn_sqr_n(a63, 63, N), a63, N);
    F256 t = f_mul(fn_sqr_n(a126, 1, N), a, N);                  // 2^127 - 1
    t = f_mul(t, t, N);                                          // the zero bit
    const F256 sq = f_mul(a, a, N);                              // a^2; a^3 is a2
    uint32_t low[4];                                             // low half of n - 2: limb 0 of n ends in ...41
#pragma unroll
    for (int i = 0; i < 4; ++i) low[i] = N.m[i];
    low[0] -= 2;
    for (int w = 63; w >= 0; --w) {
        t = fn_sqr_n(t, 2, N);
        const uint32_t d = (low[w >> 4] >> (2 * (w & 15))) & 3u;
        if (d) {
            F256 f;
#pragma unroll
            for (int i = 0; i < 8; ++i) f.v[i] = d == 1 ? a.v[i] : (d == 2 ? sq.v[i] : a2.v[i]);
            t = f_mul(t, f, N);
        }
    }
    return t;
}

// ---- plain 256-bit helpers for the scalar split --------------------------------------------------------
EC_DEV F256 u256_mul_lo(const F256& a, const F256& b) {          // a * b mod 2^256
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j + i < 8; ++j) {
            uint64_t v = (uint64_t)a.v[j] * b.v[i] + t[i + j] + c;
            t[i + j] = (uint32_t)v;
            c = v >> 32;
        }
    }
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return r;
}
EC_DEV F256 u256_mul_shift384(const F256& k, const F256& g) {    // round(k * g / 2^384), below 2^128
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t v = (uint64_t)k.v[j] * g.v[i] + t[i + j] + c;
            t[i + j] = (uint32_t)v;
            c = v >> 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    uint64_t c = (uint64_t)t[11] + 0x80000000u;                    // + 2^383
    c >>= 32;
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint64_t v = (uint64_t)t[12 + i] + c; r.v[i] = (uint32_t)v; c = v >> 32; }
    return r;
}
EC_DEV F256 u256_sub_wrap(const F256& a, const F256& b) {        // a - b mod 2^256
    F256 r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    return r;
}
// |a| of a two's-complement value, sign in *neg
EC_DEV F256 u256_abs(const F256& a, bool* neg) {
    *neg = (a.v[7] >> 31) != 0;
    F256 zero;
#pragma unroll
    for (int i = 0; i < 8; ++i) zero.v[i] = 0;
    const F256 m = u256_sub_wrap(zero, a);
    F256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = *neg ? m.v[i] : a.v[i];
    return r;
}

EC_DEV uint64_t rol64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }
EC_DEV void keccak_f1600(uint64_t s[25]) {
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                             0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                             0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                             0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                             0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
#pragma unroll
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) {
                const int rot = ROT[x + 5 * y];
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rot ? rol64(s[x + 5 * y], rot) : s[x + 5 * y];
            }
#pragma unroll
        for (int y = 0; y < 5; ++y)
#pragma unroll
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}

EC_DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// status codes per record
enum { REC_OK = 0, REC_BAD_SCALAR = 1, REC_NOT_ON_CURVE = 2, REC_INFINITY = 3 };

EC_DEV void recover_lane(const uint8_t* rec, const RecoverConsts* K, const F256* gtab, uint8_t* out, uint32_t* ko, uint8_t* status_out) {
    const ModConst& P = K->mod[MOD_P];
    const ModConst& N = K->mod[MOD_N];
    const ModConst& R = K->mod[MOD_R];

    // ---- attestation hash: Poseidon over (about, domain, value, message, 0) ----------------------------
    F256 st[5];
    {
        F256 about, domain, value, message;
#pragma unroll
        for (int k = 0; k < 8; ++k) about.v[k] = domain.v[k] = value.v[k] = message.v[k] = 0;
        for (int b = 0; b < 20; ++b) {   // the address bytes reversed are the low 20 bytes of the scalar (eth.rs:77-95)
            about.v[b >> 2] |= (uint32_t)rec[19 - b] << (8 * (b & 3));
            domain.v[b >> 2] |= (uint32_t)rec[20 + 19 - b] << (8 * (b & 3));
        }
        value.v[0] = rec[40];
        for (int b = 0; b < 32; ++b) message.v[b >> 2] |= (uint32_t)rec[41 + 31 - b] << (8 * (b & 3));
        // the message is a 256-bit big-endian integer reduced mod r (attestation.rs:107-113): at most five subtractions
        for (int k = 0; k < 6; ++k) message = f_cond_sub(message, R.m, false);
        st[0] = f_to_mont(about, R); st[1] = f_to_mont(domain, R); st[2] = f_to_mont(value, R); st[3] = f_to_mont(message, R);
#pragma unroll
        for (int k = 0; k < 8; ++k) st[4].v[k] = 0;
    }
    for (int half = 0; half < 2; ++half) {       // 4 full rounds, 60 partial rounds, 4 full rounds
        for (int fr = 0; fr < 4; ++fr) {
            const int round = half * 4 + fr;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                st[k] = f_add(st[k], K->rc_full[round * 5 + k], R);
                F256 x2 = f_mul(st[k], st[k], R);
                st[k] = f_mul(f_mul(x2, x2, R), st[k], R);
            }
            const F256* m = round == 3 ? K->mds_pre : K->mds;
            F256 o[5];
            for (int a = 0; a < 5; ++a) o[a] = f_dot5(m + a * 5, st, R);
#pragma unroll
            for (int k = 0; k < 5; ++k) st[k] = o[k];
        }
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) st[k] = f_add(st[k], K->rc_first[k], R);
            for (int pr = 0; pr < 60; ++pr) {
                if (pr) st[0] = f_add(st[0], K->rc0[pr], R);
                F256 x2 = f_mul(st[0], st[0], R);
                st[0] = f_mul(f_mul(x2, x2, R), st[0], R);
                const F256 head = f_dot5(K->sp_row + pr * 5, st, R);
#pragma unroll
                for (int k = 1; k < 5; ++k) st[k] = f_add(st[k], f_mul(K->sp_col[pr * 4 + k - 1], st[0], R), R);
                st[0] = head;
            }
        }
    }
    F256 hash = f_from_mont(st[0], R);   // < r < n: usable as a secp256k1 scalar as it is (attestation.rs:225-233)

    // ---- signature scalars ------------------------------------------------------------------------
    F256 r, s;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r.v[k] = 0; s.v[k] = 0; }
    for (int b = 0; b < 32; ++b) {   // little-endian (ecdsa/native.rs:196-208)
        r.v[b >> 2] |= (uint32_t)rec[73 + b] << (8 * (b & 3));
        s.v[b >> 2] |= (uint32_t)rec[105 + b] << (8 * (b & 3));
    }
    const bool y_odd = rec[137] != 0;
    uint8_t code = REC_OK;
    if (f_is_zero(r) || f_is_zero(s) || f_geq(r, N.m) || f_geq(s, N.m)) code = REC_BAD_SCALAR;

    // ---- R = (r, y): y = (r^3 + 7)^((p + 1) / 4) ------------------------------------------------------
    const F256 rx = r;   // r < n < p
    F256 rhs = f_add(fp_mul(fp_sqr(rx, P), rx, P), K->seven, P);
    F256 ry = fp_sqrt(rhs, P);
    if (code == REC_OK && !f_eq(fp_sqr(ry, P), rhs)) code = REC_NOT_ON_CURVE;
    if ((ry.v[0] & 1u) != (y_odd ? 1u : 0u)) ry = f_sub(f_load_mod(P.m), ry, P);   // p - y (y != 0 on this curve)

    // ---- Q = u1 * G + u2 * R,  u1 = -hash / r,  u2 = s / r  (mod n) -----------------------------------------
    F256 ri = fn_inv(f_to_mont(r, N), N);
    F256 zero;
#pragma unroll
    for (int k = 0; k < 8; ++k) zero.v[k] = 0;
    F256 u1 = f_from_mont(f_sub(zero, f_mul(ri, f_to_mont(hash, N), N), N), N);
    F256 u2 = f_from_mont(f_mul(ri, f_to_mont(s, N), N), N);
    JPt acc;
    acc.x = zero; acc.y = zero; acc.y.v[0] = 1; acc.z = zero;
    {
        // u2 * R = k1 * R + k2 * (beta * x(R), y(R)) with |k1|, |k2| < 2^128: 129 doublings instead of 256
        const F256 c1 = u256_mul_shift384(u2, K->g1), c2 = u256_mul_shift384(u2, K->g2);
        bool neg1, neg2;
        const F256 k1 = u256_abs(u256_sub_wrap(u256_sub_wrap(u2, u256_mul_lo(c1, K->a1)), u256_mul_lo(c2, K->a2)), &neg1);
        const F256 k2 = u256_abs(u256_sub_wrap(u256_mul_lo(c1, K->nb1), u256_mul_lo(c2, K->b2)), &neg2);
        const F256 ny = f_sub(f_load_mod(P.m), ry, P);             // -y(R): y is not zero on this curve
        const F256 rx2 = fp_mul(K->beta, rx, P);
        F256 y1, y2;
#pragma unroll
        for (int k = 0; k < 8; ++k) { y1.v[k] = neg1 ? ny.v[k] : ry.v[k]; y2.v[k] = neg2 ? ny.v[k] : ry.v[k]; }
        for (int bit = 128; bit >= 0; --bit) {
            acc = j_double(acc, P);
            if ((k1.v[bit >> 5] >> (bit & 31)) & 1) acc = j_add_affine(acc, rx, y1, P);
            if ((k2.v[bit >> 5] >> (bit & 31)) & 1) acc = j_add_affine(acc, rx2, y2, P);
        }
    }
    for (int w = 0; w < 32; ++w) {                    // u1 * G: one table entry per byte of u1, no doublings
        const uint32_t d = (u1.v[w >> 2] >> (8 * (w & 3))) & 0xFFu;
        if (d) {
            const F256* e = gtab + 2 * ((size_t)w * 256 + d);
            acc = j_add_affine(acc, e[0], e[1], P);
        }
    }
    if (code == REC_OK && f_is_zero(acc.z)) code = REC_INFINITY;
    F256 zi = fp_inv(acc.z, P), zi2 = fp_sqr(zi, P);
    F256 qx = fp_mul(acc.x, zi2, P), qy = fp_mul(acc.y, fp_mul(zi2, zi, P), P);

    // ---- address = Keccak-256(x || y big-endian)[12..32] ---------------------------------------------------
    uint64_t ks[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) ks[k] = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {   // lane w of the sponge = bytes 8w..8w+7 of the message, little-endian words
        // big-endian x: byte j of the message is byte (31 - j) of the integer
        uint64_t lo = bswap32(qx.v[7 - 2 * w]), hi = bswap32(qx.v[6 - 2 * w]);
        ks[w] = lo | (hi << 32);
        lo = bswap32(qy.v[7 - 2 * w]); hi = bswap32(qy.v[6 - 2 * w]);
        ks[4 + w] = lo | (hi << 32);
    }
    ks[8] ^= 0x01ULL;                    // Keccak padding after 64 message bytes
    ks[16] ^= 0x8000000000000000ULL;     // last byte of the 136-byte rate
    keccak_f1600(ks);
    for (int b = 0; b < 20; ++b) {
        const int pos = 12 + b;
        out[b] = code == REC_OK ? (uint8_t)(ks[pos >> 3] >> (8 * (pos & 7))) : 0;
    }
    if (ko) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { ko[k] = code == REC_OK ? qx.v[k] : 0u; ko[8 + k] = code == REC_OK ? qy.v[k] : 0u; }
    }
    *status_out = code;
}

__global__ __launch_bounds__(256) void k_recover(const uint8_t* __restrict__ recs, uint32_t n, const RecoverConsts* __restrict__ K,
                                                 const F256* __restrict__ gtab, uint8_t* __restrict__ attesters,
                                                 uint8_t* __restrict__ keys, uint8_t* __restrict__ status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // record layout: about[20] domain[20] value message[32] sig_r[32] sig_s[32] rec_id
    recover_lane(recs + (size_t)i * sizeof(et_attestation), K, gtab, attesters + (size_t)i * 20,
                 keys ? reinterpret_cast<uint32_t*>(keys + (size_t)i * 64) : nullptr, status + i);
}

static void put(uint32_t* dst, const U256h& v) {
    for (int i = 0; i < 4; ++i) { dst[2 * i] = (uint32_t)v.l[i]; dst[2 * i + 1] = (uint32_t)(v.l[i] >> 32); }
}
static void fill_mod(ModConst& mc, const U256h& m) {
    Mont M(m);
    std::memset(&mc, 0, sizeof mc);
    put(mc.m, M.m);
    put(mc.r2, M.r2);
    put(mc.one, M.one);
    mc.n0 = (uint32_t)M.n0;
}


// the three moduli, the generator (plain), Poseidon's parameters in Montgomery form (R = 2^256)
static int build_consts(RecoverConsts& K) {
    U256h rmod;
    std::memcpy(rmod.l, FR_MOD, 32);
    fill_mod(K.mod[MOD_P], SECP_P);
    fill_mod(K.mod[MOD_N], SECP_N);
    fill_mod(K.mod[MOD_R], rmod);
    Mont MR(rmod);
    put(K.gx.v, SECP_GX);
    put(K.gy.v, SECP_GY);
    put(K.seven.v, U256h{{7, 0, 0, 0}});
    // secp256k1's endomorphism constants (public; beta^3 = 1 mod p, lambda^3 = 1 mod n, lambda * G = (beta * Gx, Gy))
    put(K.beta.v, U256h{{0xc1396c28719501eeULL, 0x9cf0497512f58995ULL, 0x6e64479eac3434e9ULL, 0x7ae96a2b657c0710ULL}});
    put(K.g1.v, U256h{{0xe893209a45dbb031ULL, 0x3daa8a1471e8ca7fULL, 0xe86c90e49284eb15ULL, 0x3086d221a7d46bcdULL}});
    put(K.g2.v, U256h{{0x1571b4ae8ac47f71ULL, 0x221208ac9df506c6ULL, 0x6f547fa90abfe4c4ULL, 0xe4437ed6010e8828ULL}});
    put(K.a1.v, U256h{{0xe86c90e49284eb15ULL, 0x3086d221a7d46bcdULL, 0, 0}});
    put(K.a2.v, U256h{{0x57c1108d9d44cfd8ULL, 0x14ca50f7a8e2f3f6ULL, 1, 0}});
    put(K.nb1.v, U256h{{0x6f547fa90abfe4c3ULL, 0xe4437ed6010e8828ULL, 0, 0}});
    put(K.b2.v, U256h{{0xe86c90e49284eb15ULL, 0x3086d221a7d46bcdULL, 0, 0}});
    std::vector<FrRaw> rc_raw, mds_raw;
    poseidon5_parameters(rc_raw, mds_raw);
    if (rc_raw.size() != 340 || mds_raw.size() != 25) { set_error("internal: Poseidon parameter count"); return ET_E_UNSUPPORTED; }
    auto store = [&](F256& dst, const Fr& v) { const FrRaw raw = v.to_raw(); U256h u; std::memcpy(u.l, raw.l, 32); put(dst.v, MR.to(u)); };
    typedef std::vector<std::vector<Fr>> Mat;
    Mat M(5, std::vector<Fr>(5));
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) M[i][j] = Fr::from_raw(mds_raw[i * 5 + j]);
    std::vector<std::vector<Fr>> RC(68, std::vector<Fr>(5));
    for (int k = 0; k < 68; ++k) for (int i = 0; i < 5; ++i) RC[k][i] = Fr::from_raw(rc_raw[k * 5 + i]);
    auto matmul = [](const Mat& A, const Mat& B) {
        Mat C(A.size(), std::vector<Fr>(B[0].size()));
        for (size_t i = 0; i < A.size(); ++i) for (size_t j = 0; j < B[0].size(); ++j) {
            Fr acc = Fr::zero();
            for (size_t k = 0; k < B.size(); ++k) acc += A[i][k] * B[k][j];
            C[i][j] = acc;
        }
        return C;
    };
    auto matvec = [](const Mat& A, const std::vector<Fr>& v) {
        std::vector<Fr> o(A.size());
        for (size_t i = 0; i < A.size(); ++i) { Fr acc = Fr::zero(); for (size_t j = 0; j < v.size(); ++j) acc += A[i][j] * v[j]; o[i] = acc; }
        return o;
    };
    auto invert4 = [](Mat a, bool& ok) {      // Gauss-Jordan on the 4 x 4 block
        const int n = 4;
        Mat inv(n, std::vector<Fr>(n, Fr::zero()));
        for (int i = 0; i < n; ++i) inv[i][i] = Fr::one();
        for (int col = 0; col < n; ++col) {
            int p = col;
            while (p < n && a[p][col].is_zero()) ++p;
            if (p == n) { ok = false; return inv; }
            std::swap(a[p], a[col]);
            std::swap(inv[p], inv[col]);
            const Fr iv = a[col][col].invert_or_zero();
            for (int j = 0; j < n; ++j) { a[col][j] *= iv; inv[col][j] *= iv; }
            for (int r = 0; r < n; ++r) {
                if (r == col || a[r][col].is_zero()) continue;
                const Fr f = a[r][col];
                for (int j = 0; j < n; ++j) { a[r][j] = a[r][j] - f * a[col][j]; inv[r][j] = inv[r][j] - f * inv[col][j]; }
            }
        }
        return inv;
    };
    // backward over the partial rounds: M_k = S_k * P_k with S_k sparse and P_k = diag(1, lower block of M_k);
    // P_k moves in front of the S-box of its round and joins the matrix of the round before: M_{k-1} = P_k * M
    std::vector<std::vector<Fr>> sp_row(60), sp_col(60);
    std::vector<Mat> P(60);
    Mat cur = M;
    for (int k = 59; k >= 0; --k) {
        Mat low(4, std::vector<Fr>(4));
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) low[i][j] = cur[i + 1][j + 1];
        bool ok = true;
        const Mat low_inv = invert4(low, ok);
        if (!ok) { set_error("internal: singular block in the Poseidon matrix"); return ET_E_UNSUPPORTED; }
        sp_row[k].assign(5, Fr::zero());
        sp_col[k].assign(4, Fr::zero());
        sp_row[k][0] = cur[0][0];
        for (int j = 0; j < 4; ++j) {
            Fr acc = Fr::zero();
            for (int i = 0; i < 4; ++i) acc += cur[0][i + 1] * low_inv[i][j];
            sp_row[k][j + 1] = acc;
            sp_col[k][j] = cur[j + 1][0];
        }
        Mat Pk(5, std::vector<Fr>(5, Fr::zero()));
        Pk[0][0] = Fr::one();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Pk[i + 1][j + 1] = low[i][j];
        P[k] = Pk;
        cur = matmul(Pk, M);
    }
    auto sparse_apply = [&](int k, const std::vector<Fr>& z) {
        std::vector<Fr> o(5);
        Fr acc = Fr::zero();
        for (int j = 0; j < 5; ++j) acc += sp_row[k][j] * z[j];
        o[0] = acc;
        for (int i = 1; i < 5; ++i) o[i] = sp_col[k][i - 1] * z[0] + z[i];
        return o;
    };
    // constants: P_k * c_k; only element 0 of it sits before the S-box, the rest rides through the sparse matrix
    // into the next round's constant, and what is left after the last partial round joins the next full round
    std::vector<Fr> carry(5, Fr::zero());
    const std::vector<Fr> first = matvec(P[0], RC[4]);
    for (int i = 0; i < 5; ++i) store(K.rc_first[i], first[i]);
    store(K.rc0[0], Fr::zero());
    for (int k = 1; k < 60; ++k) {
        std::vector<Fr> eff = matvec(P[k], RC[4 + k]);
        for (int i = 0; i < 5; ++i) eff[i] += carry[i];
        store(K.rc0[k], eff[0]);
        eff[0] = Fr::zero();
        carry = sparse_apply(k, eff);
    }
    for (int round = 0; round < 8; ++round)
        for (int i = 0; i < 5; ++i) {
            Fr c = RC[round < 4 ? round : 60 + round][i];
            if (round == 4) c += carry[i];
            store(K.rc_full[round * 5 + i], c);
        }
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) { store(K.mds[i * 5 + j], M[i][j]); store(K.mds_pre[i * 5 + j], cur[i][j]); }
    for (int k = 0; k < 60; ++k) {
        for (int j = 0; j < 5; ++j) store(K.sp_row[k * 5 + j], sp_row[k][j]);
        for (int j = 0; j < 4; ++j) store(K.sp_col[k * 4 + j], sp_col[k][j]);
    }
    return ET_OK;
}

// The constants are derived once per process (the sparse Poseidon matrices cost a few milliseconds).
static int recover_consts(const RecoverConsts** out) {
    static std::vector<RecoverConsts> cached(1);
    static int status = ET_OK;
    static std::once_flag once;
    std::call_once(once, [&]() { status = build_consts(cached[0]); });
    *out = &cached[0];
    return status;
}

// Fixed-base table for the generator: entry [w][d] = d * 2^(8w) * G in affine plain coordinates (x, y), d = 1..255
// (entry 0 unused), 32 windows: u1 * G is the sum of one entry per byte of u1.  Built once per process with the
// lane arithmetic itself; the Jacobian results are brought to affine form with one shared inversion.
static const std::vector<F256>& generator_table(const RecoverConsts& K) {
    static std::vector<F256> table;
    static std::once_flag once;
    std::call_once(once, [&]() {
        const ModConst& P = K.mod[MOD_P];
        std::vector<JPt> pts((size_t)32 * 256);
        F256 bx = K.gx, by = K.gy;
        F256 one;
        for (int i = 0; i < 8; ++i) one.v[i] = i == 0 ? 1u : 0u;
        for (int w = 0; w < 32; ++w) {
            JPt acc;
            acc.x = bx; acc.y = by; acc.z = one;
            pts[(size_t)w * 256 + 1] = acc;
            for (int d = 2; d < 256; ++d) { acc = j_add_affine(acc, bx, by, P); pts[(size_t)w * 256 + d] = acc; }
            acc = j_add_affine(acc, bx, by, P);            // 256 * base = the next window's base
            const F256 zi = fp_inv(acc.z, P), zi2 = fp_sqr(zi, P);
            bx = fp_mul(acc.x, zi2, P);
            by = fp_mul(acc.y, fp_mul(zi2, zi, P), P);
        }
        // batch inversion of every z (none is zero: d * 2^(8w) is never a multiple of the group order)
        std::vector<F256> prefix(pts.size());
        F256 run = one;
        for (size_t i = 0; i < pts.size(); ++i) {
            if ((i & 255) == 0) { prefix[i] = run; continue; }
            prefix[i] = run;
            run = fp_mul(run, pts[i].z, P);
        }
        F256 inv = fp_inv(run, P);
        table.assign(pts.size() * 2, F256{});
        for (size_t i = pts.size(); i-- > 0;) {
            if ((i & 255) == 0) continue;
            const F256 zi = fp_mul(inv, prefix[i], P);
            inv = fp_mul(inv, pts[i].z, P);
            const F256 zi2 = fp_sqr(zi, P);
            table[2 * i] = fp_mul(pts[i].x, zi2, P);
            table[2 * i + 1] = fp_mul(pts[i].y, fp_mul(zi2, zi, P), P);
        }
    });
    return table;
}

}  // namespace et

using namespace et;

extern "C" int et_attestations_recover_gpu(const et_attestation* atts, size_t n, uint8_t* attesters, uint8_t* public_keys, int device) {
    if ((!atts || !attesters) && n) return ET_E_INVALID_ARG;
    if (n > 0xFFFFFFFFu) return ET_E_UNSUPPORTED;
    int ndev = 0;
    HIP_TRY_E(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    if (device < 0) HIP_TRY_E(hipGetDevice(&device));
    HIP_TRY_E(hipSetDevice(device));
    if (n == 0) return ET_OK;

    const RecoverConsts* kp = nullptr;
    int rcc = recover_consts(&kp);
    if (rcc) return rcc;
    const RecoverConsts& K = *kp;

    uint8_t *d_recs = nullptr, *d_addr = nullptr, *d_keys = nullptr, *d_status = nullptr;
    RecoverConsts* d_k = nullptr;
    F256* d_tab = nullptr;
    struct FreeTab { F256** t; ~FreeTab() { if (*t) (void)hipFree(*t); } } guard_tab{&d_tab};
    const std::vector<F256>& tab = generator_table(K);
    struct Free { uint8_t** a; uint8_t** b; uint8_t** c; uint8_t** d; RecoverConsts** k;
                  ~Free() { if (*a) (void)hipFree(*a); if (*b) (void)hipFree(*b); if (*c) (void)hipFree(*c); if (*d) (void)hipFree(*d); if (*k) (void)hipFree(*k); } }
        guard{&d_recs, &d_addr, &d_keys, &d_status, &d_k};
    HIP_TRY_E(hipMalloc((void**)&d_recs, n * sizeof(et_attestation)));
    HIP_TRY_E(hipMalloc((void**)&d_addr, n * 20));
    if (public_keys) HIP_TRY_E(hipMalloc((void**)&d_keys, n * 64));
    HIP_TRY_E(hipMalloc((void**)&d_status, n));
    HIP_TRY_E(hipMalloc((void**)&d_k, sizeof(RecoverConsts)));
    HIP_TRY_E(hipMemcpy(d_recs, atts, n * sizeof(et_attestation), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMemcpy(d_k, &K, sizeof(RecoverConsts), hipMemcpyHostToDevice));
    HIP_TRY_E(hipMalloc((void**)&d_tab, tab.size() * sizeof(F256)));
    HIP_TRY_E(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(F256), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_recover, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_recs, (uint32_t)n, d_k, d_tab, d_addr, d_keys, d_status);
    HIP_TRY_E(hipGetLastError());
    std::vector<uint8_t> st(n);
    HIP_TRY_E(hipMemcpy(st.data(), d_status, n, hipMemcpyDeviceToHost));
    HIP_TRY_E(hipMemcpy(attesters, d_addr, n * 20, hipMemcpyDeviceToHost));
    if (public_keys) HIP_TRY_E(hipMemcpy(public_keys, d_keys, n * 64, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        if (st[i] == REC_OK) continue;
        const char* why = st[i] == REC_BAD_SCALAR ? "signature scalar out of range"
                        : st[i] == REC_NOT_ON_CURVE ? "signature r is not the x coordinate of a curve point"
                                                    : "recovered key is the point at infinity";
        set_error("attestation " + std::to_string(i) + ": " + why);
        return ET_E_BAD_SIGNATURE;
    }
    return ET_OK;
}

// The lane function of the kernel compiled for the host, on one record: lets the device arithmetic be
// checked where no GPU is present.  A self-test hook, not a compute path.
extern "C" int et_attestation_recover_lane_host(const et_attestation* att, uint8_t attester[20], uint8_t public_key[64]) {
    if (!att || !attester || !public_key) return ET_E_INVALID_ARG;
    const RecoverConsts* kp = nullptr;
    int rc = recover_consts(&kp);
    if (rc) return rc;
    uint8_t status = 0;
    uint32_t key[16];
    recover_lane(reinterpret_cast<const uint8_t*>(att), kp, generator_table(*kp).data(), attester, key, &status);
    std::memcpy(public_key, key, 64);
    if (status != REC_OK) { set_error("lane status " + std::to_string((int)status)); return ET_E_BAD_SIGNATURE; }
    return ET_OK;
}
