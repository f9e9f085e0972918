// Device-side BN254 scalar field arithmetic for the exact (Fr) kernels: 8 x 32-bit limbs.
//
// The reference multiplies and adds `bn256::Fr` values in its hot loop
// (dynamic_sets/native.rs:323-324).  Here a row's dot product is accumulated UNREDUCED: each
// 256 x 256-bit product is added as a plain integer to a 17-limb (544-bit) accumulator, and one
// Montgomery reduction per row brings the sum back below r.  Edge weights are stored pre-multiplied
// by 2^288 mod r, so that reduction (nine 32-bit rounds, i.e. division by 2^288) returns exactly
// sum_j c_ji * s[j] mod r in canonical form.  A 544-bit accumulator holds 2^34 products of
// canonical operands, more than any row can have.  Integer addition is associative, so the result
// does not depend on how a row's products are spread over lanes: it is bit-exact by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace et {

#define ET_DEV __device__ __forceinline__

// r, little-endian 32-bit limbs
__device__ static constexpr uint32_t FR_P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                                0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static constexpr uint32_t FR_N0_32 = 0xefffffffu;  // -r^-1 mod 2^32

struct U256 {
    uint32_t v[8];
};

ET_DEV U256 u256_load(const uint4* p) {  // 32-byte aligned field element
    uint4 lo = p[0], hi = p[1];
    U256 r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
ET_DEV void u256_store(uint4* p, const U256& a) {
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
ET_DEV U256 u256_from_planes(uint4 lo, uint4 hi) {
    U256 r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
ET_DEV U256 u256_zero() {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
ET_DEV bool u256_eq(const U256& a, const U256& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}
ET_DEV bool u256_geq_p(const U256& a) {
    // a >= r  <=>  a - r does not borrow
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.v[i] - FR_P[i] - br;
        br = (d >> 32) & 1;
    }
    return br == 0;
}
ET_DEV U256 u256_sub_p_if_geq(const U256& a) {
    U256 d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - FR_P[i] - br;
        d.v[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d.v[i];
    return r;
}
// (a + b) mod r for canonical a, b   (a + b < 2r < 2^255: no carry out of 256 bits)
ET_DEV U256 fr_add(const U256& a, const U256& b) {
    U256 s;
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + cy;
        s.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
    return u256_sub_p_if_geq(s);
}
// (a - b) mod r for canonical a, b
ET_DEV U256 fr_sub(const U256& a, const U256& b) {
    U256 d;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
        d.v[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    uint64_t cy = 0;
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)d.v[i] + (br ? FR_P[i] : 0u) + cy;
        r.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
    return r;
}

// 544-bit unreduced accumulator
struct Acc {
    uint32_t v[17];
};
ET_DEV Acc acc_zero() {
    Acc a;
#pragma unroll
    for (int i = 0; i < 17; ++i) a.v[i] = 0;
    return a;
}
ET_DEV void acc_add(Acc& a, const Acc& b) {
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + cy;
        a.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
}
ET_DEV void acc_add_u256(Acc& a, const U256& b) {
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.v[i] + b.v[i] + cy;
        a.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
#pragma unroll
    for (int i = 8; i < 17; ++i) {
        uint64_t t = (uint64_t)a.v[i] + cy;
        a.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
}

// acc += a * b  (schoolbook 8 x 8 into a fresh 16-limb product, then one 17-limb add)
ET_DEV void acc_mac(Acc& acc, const U256& a, const U256& b) {
    uint32_t p[16];
    {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t t = (uint64_t)a.v[j] * b.v[0] + c;
            p[j] = (uint32_t)t;
            c = t >> 32;
        }
        p[8] = (uint32_t)c;
    }
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t t = (uint64_t)a.v[j] * b.v[i] + p[i + j] + c;
            p[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        p[i + 8] = (uint32_t)c;
    }
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint64_t t = (uint64_t)acc.v[i] + p[i] + cy;
        acc.v[i] = (uint32_t)t;
        cy = t >> 32;
    }
    acc.v[16] += (uint32_t)cy;
}
// a (9 limbs, below 2^288) times b as a fresh 17-limb product: an unreduced row sum times a stored factor, so that
// one reduction serves both
ET_DEV Acc acc_mul_9x8(const uint32_t* a9, const U256& b) {
    Acc p;
    {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            uint64_t t = (uint64_t)a9[j] * b.v[0] + c;
            p.v[j] = (uint32_t)t;
            c = t >> 32;
        }
        p.v[9] = (uint32_t)c;
    }
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            uint64_t t = (uint64_t)a9[j] * b.v[i] + p.v[i + j] + c;
            p.v[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        p.v[i + 9] = (uint32_t)c;
    }
    return p;
}
// Column accumulator of the product-scanning forms below: 96 bits as a 64-bit pair, which is the multiply-add's own
// addend, plus the count of its carries.
ET_DEV void col_mac(uint64_t& lohi, uint32_t& top, uint32_t a, uint32_t b) {
    // the multiply-add's carry-out goes straight into the third word: two instructions per limb product
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lohi), "+v"(top) : "v"(a), "v"(b) : "vcc");
}
ET_DEV uint32_t col_shift(uint64_t& lohi, uint32_t& top) {
    const uint32_t out = (uint32_t)lohi;
    lohi = (lohi >> 32) | ((uint64_t)top << 32);
    top = 0;
    return out;
}
// a (9 limbs, below 2^288) * b * 2^-288 mod r, canonical: the product acc_mul_9x8 and the reduction acc_redc288 in
// one pass over the columns (limb i of the multiplier m is fixed when column i is complete), so that no partial
// product is written back and no 32-bit carry travels on its own.  Requires b < r.
ET_DEV U256 fr_mul9_redc288(const uint32_t* a9, const U256& b) {
    uint64_t lohi = 0;
    uint32_t top = 0;
    uint32_t m[9];
    U256 r;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = i - j;
            if (k >= 0 && k < 8) col_mac(lohi, top, a9[j], b.v[k]);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = i - j;
            if (j < i && k >= 0 && k < 8) col_mac(lohi, top, m[j], FR_P[k]);
        }
        if (i < 9) {
            m[i] = (uint32_t)lohi * FR_N0_32;
            col_mac(lohi, top, m[i], FR_P[0]);               // the column's low word becomes zero
            (void)col_shift(lohi, top);
        } else {
            r.v[i - 9] = col_shift(lohi, top);
        }
    }
    return u256_sub_p_if_geq(r);
}
// a (10 limbs, below 2^304) * b * 2^-320 mod r, canonical: the same pass with one more limb and one more reduction round
// (the quotient a * b / 2^320 stays below 2^238, so one conditional subtraction brings the result below r).  Requires b < r.
ET_DEV U256 fr_mul10_redc320(const uint32_t* a10, const U256& b) {
    uint64_t lohi = 0;
    uint32_t top = 0;
    uint32_t m[10];
    U256 r;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int k = i - j;
            if (k >= 0 && k < 8) col_mac(lohi, top, a10[j], b.v[k]);
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int k = i - j;
            if (j < i && k >= 0 && k < 8) col_mac(lohi, top, m[j], FR_P[k]);
        }
        if (i < 10) {
            m[i] = (uint32_t)lohi * FR_N0_32;
            col_mac(lohi, top, m[i], FR_P[0]);               // the column's low word becomes zero
            (void)col_shift(lohi, top);
        } else {
            r.v[i - 10] = col_shift(lohi, top);
        }
    }
    return u256_sub_p_if_geq(r);
}
// a * b * 2^-256 mod r for a below 2^256 and b < r, NOT brought below r (below 2^256 and congruent): acc_mac followed
// by acc_redc_loose<8>, as one pass over the columns
ET_DEV U256 fr_mul_redc_loose8(const U256& a, const U256& b) {
    uint64_t lohi = 0;
    uint32_t top = 0;
    uint32_t m[8];
    U256 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = i - j;
            if (k >= 0 && k < 8) col_mac(lohi, top, a.v[j], b.v[k]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = i - j;
            if (j < i && k >= 0 && k < 8) col_mac(lohi, top, m[j], FR_P[k]);
        }
        if (i < 8) {
            m[i] = (uint32_t)lohi * FR_N0_32;
            col_mac(lohi, top, m[i], FR_P[0]);
            (void)col_shift(lohi, top);
        } else {
            r.v[i - 8] = col_shift(lohi, top);
        }
    }
    return r;
}
// acc += a * v  (32-bit opinion times field element)
ET_DEV void acc_mac_u32(Acc& acc, const U256& a, uint32_t v) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t t = (uint64_t)a.v[j] * v + acc.v[j] + c;
        acc.v[j] = (uint32_t)t;
        c = t >> 32;
    }
#pragma unroll
    for (int j = 8; j < 17; ++j) {
        uint64_t t = (uint64_t)acc.v[j] + c;
        acc.v[j] = (uint32_t)t;
        c = t >> 32;
    }
}

// acc * 2^-288 mod r, canonical.  Requires acc < r * 2^288 (any sum of < 2^34 canonical products).
ET_DEV U256 acc_redc288(Acc a) {
    uint32_t carry2 = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t m = a.v[i] * FR_N0_32;
        uint64_t t = (uint64_t)m * FR_P[0] + a.v[i];
        uint64_t c = t >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            t = (uint64_t)m * FR_P[j] + a.v[i + j] + c;
            a.v[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        // limb i+8 exists for every i <= 8
        t = (uint64_t)a.v[i + 8] + c + carry2;
        a.v[i + 8] = (uint32_t)t;
        carry2 = (uint32_t)(t >> 32);
    }
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = a.v[9 + i];
    return u256_sub_p_if_geq(r);
}

// acc * 2^-(32 * ROUNDS) mod r for acc < 2^(32 * ROUNDS) * 2^255, NOT brought below r: the result is below 2^256
// and congruent, which is all a value needs that is only multiplied again.
template <int ROUNDS>
ET_DEV U256 acc_redc_loose(Acc a) {
    static_assert(ROUNDS >= 1 && ROUNDS <= 9, "the accumulator holds 17 limbs");
    uint32_t carry2 = 0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        uint32_t m = a.v[i] * FR_N0_32;
        uint64_t t = (uint64_t)m * FR_P[0] + a.v[i];
        uint64_t c = t >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            t = (uint64_t)m * FR_P[j] + a.v[i + j] + c;
            a.v[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        t = (uint64_t)a.v[i + 8] + c + carry2;
        a.v[i + 8] = (uint32_t)t;
        carry2 = (uint32_t)(t >> 32);
    }
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = a.v[ROUNDS + i];
    return r;
}

// 320-bit accumulator for sums of (32-bit opinion) x (field element): holds 2^34 such products
struct Acc10 {
    uint32_t v[10];
};
ET_DEV Acc10 acc10_zero() {
    Acc10 a;
#pragma unroll
    for (int i = 0; i < 10; ++i) a.v[i] = 0;
    return a;
}
ET_DEV void acc10_mac_u32(Acc10& acc, const U256& a, uint32_t v) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t t = (uint64_t)a.v[j] * v + acc.v[j] + c;
        acc.v[j] = (uint32_t)t;
        c = t >> 32;
    }
    uint64_t t = (uint64_t)acc.v[8] + c;
    acc.v[8] = (uint32_t)t;
    acc.v[9] += (uint32_t)(t >> 32);
}
ET_DEV Acc acc_from_acc10(const Acc10& s) {
    Acc a = acc_zero();
#pragma unroll
    for (int i = 0; i < 10; ++i) a.v[i] = s.v[i];
    return a;
}
// a * b * 2^-288 mod r: the Montgomery product the kernels use for "times a stored factor"
ET_DEV U256 fr_mul_redc288(const U256& a, const U256& b) {
    Acc t = acc_zero();
    acc_mac(t, a, b);
    return acc_redc288(t);
}

// Wavefront-wide integer sum of accumulators; every lane receives the total.
ET_DEV void acc_wave_sum(Acc& a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Acc o;
#pragma unroll
        for (int i = 0; i < 17; ++i) o.v[i] = __shfl_xor(a.v[i], off, 64);
        acc_add(a, o);
    }
}

// Integer sum of the accumulators of `lanes` adjacent lanes (a power of two, the group is aligned);
// every lane of the group receives the total.
ET_DEV void acc_group_sum(Acc& a, uint32_t lanes) {
    for (uint32_t off = lanes >> 1; off >= 1; off >>= 1) {
        Acc o;
#pragma unroll
        for (int i = 0; i < 17; ++i) o.v[i] = __shfl_xor(a.v[i], (int)off, 64);
        acc_add(a, o);
    }
}
ET_DEV void acc10_group_sum(Acc10& a, uint32_t lanes) {
    for (uint32_t off = lanes >> 1; off >= 1; off >>= 1) {
        uint64_t cy = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            uint64_t t = (uint64_t)a.v[i] + (uint32_t)__shfl_xor(a.v[i], (int)off, 64) + cy;
            a.v[i] = (uint32_t)t;
            cy = t >> 32;
        }
    }
}

}  // namespace et
