// Batches of small independent sets (ZK witness generation: thousands of 256-peer sets).
//
// One workgroup per set; the whole converge of a set (dynamic_sets/native.rs:286-337) runs in ONE
// launch.  Opinions that fit a byte (the production range, eigentrust/src/attestation.rs:304): the
// dense product runs on the matrix cores as a product of byte limbs (k_batch_fr_mfma_packed for sets
// of up to 32 slots, several per wavefront; k_batch_fr_mfma up to 256 slots; k_batch_fr_mfma_stream
// up to 1024).  Larger opinions: k_batch_fr on the vector
// ALUs with the trust vector in LDS; k_batch_fr_u8 is the byte-valued vector-ALU kernel the
// matrix-core ones are measured against.
// Sets never communicate, so a batch is split over GPUs by giving each its own sub-batch.
//
// The dense row of ones of a peer without valid opinions (native.rs:263-278) is evaluated as the
// rank-1 term (D - dangling_i * s_i) / (n_valid - 1), exactly as in the graph engine, and the
// normalisation is folded into the vector: xs_j = s_j * inv(sum_j), new_s_i = sum_j v_ji * xs_j.
// Field arithmetic is exact and distributive, so this equals the reference's
// sum_j (v_ji * inv(sum_j)) * s_j bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "engine.hpp"
#include "fr_dev.hpp"
#include "fr_host.hpp"

namespace et {

#define HIP_TRY_B(expr)                                                               \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));             \
            return _e == hipErrorNoDevice ? ET_E_NO_DEVICE : ET_E_HIP;                \
        }                                                                             \
    } while (0)

static const int BATCH_THREADS = 256;

struct BatchArgs {
    const void* vals;        // n_sets * n * n opinions, row-major (row = attester), uint8 or uint32
    const uint8_t* member;   // n_sets * n
    const uint8_t* dangling; // n_sets * n : member whose filtered row sums to zero
    const uint4* scale;      // n_sets * n : inv(row sum) * 2^576 mod r (zero for dangling / empty)
    const uint4* inv_nm1_w;  // n_sets     : inv(n_valid - 1) * 2^288 mod r
    const int32_t* pre_status; // n_sets   : ET_OK or ET_E_INSUFFICIENT_PEERS
    uint4* out;              // n_sets * n scores
    int32_t* status;         // n_sets
    uint32_t n, num_iter;
    U256 initial;
};

template <typename VT, bool VALS_IN_LDS>
__global__ __launch_bounds__(BATCH_THREADS) void k_batch_fr(BatchArgs a) {
    extern __shared__ uint4 lds4[];
    const uint32_t n = a.n;
    // LDS carve-up: s[n], next s[n], xs[n] (2 x uint4 each), reduction scratch, optional opinion matrix
    uint4* s_l = lds4;
    uint4* sn_l = lds4 + 2 * (size_t)n;
    uint4* xs_l = lds4 + 4 * (size_t)n;
    uint4* red_l = lds4 + 6 * (size_t)n;              // 2 * 4 waves
    VT* v_l = reinterpret_cast<VT*>(red_l + 2 * 4 + 2);
    __shared__ uint32_t sh_D[8];
    const uint32_t set = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const VT* vg = reinterpret_cast<const VT*>(a.vals) + (size_t)set * n * n;
    const uint8_t* mem = a.member + (size_t)set * n;
    const uint8_t* dng = a.dangling + (size_t)set * n;
    uint4* outp = a.out + 2 * (size_t)set * n;

    if (a.pre_status[set] != ET_OK) {
        for (uint32_t i = threadIdx.x; i < n; i += BATCH_THREADS) u256_store(outp + 2 * i, u256_zero());
        if (threadIdx.x == 0) a.status[set] = a.pre_status[set];
        return;
    }
    if (VALS_IN_LDS) {
        const size_t total = (size_t)n * n;
        for (size_t t = threadIdx.x; t < total; t += BATCH_THREADS) v_l[t] = vg[t];
    }
    const VT* vsrc = VALS_IN_LDS ? v_l : vg;
    const U256 inw = u256_load(a.inv_nm1_w + 2 * (size_t)set);
    uint32_t n_valid_local = 0;
    for (uint32_t i = threadIdx.x; i < n; i += BATCH_THREADS) {
        u256_store(s_l + 2 * i, mem[i] ? a.initial : u256_zero());   // native.rs:317
        n_valid_local += mem[i] != 0;
    }
    __syncthreads();

    for (uint32_t it = 0; it < a.num_iter; ++it) {
        // phase 1: xs_j = s_j * inv(sum_j) * 2^288, and the score mass of dangling peers
        U256 dpart = u256_zero();
        for (uint32_t i = threadIdx.x; i < n; i += BATCH_THREADS) {
            U256 si = u256_load(s_l + 2 * i);
            U256 k = u256_load(a.scale + 2 * ((size_t)set * n + i));
            Acc t = acc_zero();
            acc_mac(t, si, k);
            u256_store(xs_l + 2 * i, acc_redc288(t));
            if (dng[i]) dpart = fr_add(dpart, si);
        }
        // workgroup sum of dpart, fixed order
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            U256 o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(dpart.v[q], off, 64);
            dpart = fr_add(dpart, o);
        }
        if (lane == 0) u256_store(red_l + 2 * wave, dpart);
        __syncthreads();
        if (threadIdx.x == 0) {
            U256 d = u256_zero();
            for (int w = 0; w < BATCH_THREADS / 64; ++w) d = fr_add(d, u256_load(red_l + 2 * w));
#pragma unroll
            for (int q = 0; q < 8; ++q) sh_D[q] = d.v[q];
        }
        __syncthreads();
        U256 D;
#pragma unroll
        for (int q = 0; q < 8; ++q) D.v[q] = sh_D[q];
        // phase 2: new_s_i = sum_{j != i} v_ji * xs_j + (D - dangling_i * s_i) * inv(n_valid-1)
        for (uint32_t i = threadIdx.x; i < n; i += BATCH_THREADS) {
            U256 v = u256_zero();
            if (mem[i]) {
                Acc10 acc;
#pragma unroll
                for (int q = 0; q < 10; ++q) acc.v[q] = 0;
                for (uint32_t j = 0; j < n; ++j) {
                    uint32_t op = (uint32_t)vsrc[(size_t)j * n + i];
                    op = j == i ? 0u : op;                       // native.rs:254-259 (self opinion)
                    U256 xj = u256_load(xs_l + 2 * j);           // same address in every lane: LDS broadcast
                    acc10_mac_u32(acc, xj, op);
                }
                Acc full = acc_zero();
#pragma unroll
                for (int q = 0; q < 10; ++q) full.v[q] = acc.v[q];
                U256 si = u256_load(s_l + 2 * i);
                U256 t = dng[i] ? fr_sub(D, si) : D;
                acc_mac(full, inw, t);
                v = acc_redc288(full);
            }
            u256_store(sn_l + 2 * i, v);
        }
        __syncthreads();
        uint4* tmp = s_l; s_l = sn_l; sn_l = tmp;
    }
    // output + conservation check (native.rs:331-334): sum of scores == INITIAL * n_valid
    U256 tot = u256_zero();
    for (uint32_t i = threadIdx.x; i < n; i += BATCH_THREADS) {
        U256 si = u256_load(s_l + 2 * i);
        u256_store(outp + 2 * i, si);
        tot = fr_add(tot, si);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        U256 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(tot.v[q], off, 64);
        tot = fr_add(tot, o);
        n_valid_local += __shfl_xor(n_valid_local, off, 64);
    }
    __shared__ uint32_t sh_nv[BATCH_THREADS / 64];
    if (lane == 0) { u256_store(red_l + 2 * wave, tot); sh_nv[wave] = n_valid_local; }
    __syncthreads();
    if (threadIdx.x == 0) {
        U256 t = u256_zero();
        uint32_t nv = 0;
        for (int w = 0; w < BATCH_THREADS / 64; ++w) { t = fr_add(t, u256_load(red_l + 2 * w)); nv += sh_nv[w]; }
        U256 expect = u256_zero();
        for (uint32_t k = 0; k < nv; ++k) expect = fr_add(expect, a.initial);   // nv <= 1024
        a.status[set] = u256_eq(t, expect) ? ET_OK : ET_E_SUM_MISMATCH;
    }
}

// Byte-valued opinions (the production range): the accumulation runs on 16-bit limbs.
// xs_j is the same for every lane of a wavefront (the lanes differ in the column i), so its eight
// words are moved to scalar registers once per j; limb (16 bit) x opinion (8 bit) < 2^24 is a
// full-rate 24-bit multiply-add into sixteen 32-bit column sums with no carry handling, and the
// column sums are folded into the 320-bit accumulator every 256 products (256 * 2^24 = 2^32).
// A column's j range is split over `parts` threads in different wavefronts, so that each SIMD has
// several wavefronts to switch between; the partial accumulators meet in LDS.  Integer addition
// is associative, so neither the limb width nor the split can change a bit of the result.
struct BatchArgsU8 {
    BatchArgs b;
    uint32_t cols;    // n rounded up to a multiple of 64
    uint32_t parts;   // threads per column; blockDim.x == cols * parts
};

ET_DEV void acc10_fold(Acc10& acc, uint32_t (&cs)[16]) {
    uint64_t c = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        c += (uint64_t)acc.v[q] + cs[2 * q] + ((uint64_t)cs[2 * q + 1] << 16);
        acc.v[q] = (uint32_t)c;
        c >>= 32;
        cs[2 * q] = 0;
        cs[2 * q + 1] = 0;
    }
    c += acc.v[8];
    acc.v[8] = (uint32_t)c;
    acc.v[9] += (uint32_t)(c >> 32);
}

template <bool VALS_IN_LDS>
__global__ __launch_bounds__(1024) void k_batch_fr_u8(BatchArgsU8 au) {
    const BatchArgs& a = au.b;
    extern __shared__ uint4 lds4[];
    const uint32_t n = a.n, cols = au.cols, parts = au.parts;
    // LDS carve-up: s[n], next s[n] (2 x uint4 each), xs[n] as sixteen 16-bit limbs, one per dword
    // (4 x uint4 each), 16 reduction slots, partial accumulators [10][parts - 1][n], optional
    // opinion matrix
    uint4* s_l = lds4;
    uint4* sn_l = lds4 + 2 * (size_t)n;
    uint4* xs_l = lds4 + 4 * (size_t)n;
    uint4* red_l = lds4 + 8 * (size_t)n;
    uint32_t* part_l = reinterpret_cast<uint32_t*>(red_l + 2 * 16 + 2);
    uint8_t* v_l = reinterpret_cast<uint8_t*>(part_l + (size_t)10 * (parts - 1) * n);
    __shared__ uint32_t sh_D[8];
    __shared__ uint32_t sh_nv[16];
    const uint32_t set = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint8_t* vg = reinterpret_cast<const uint8_t*>(a.vals) + (size_t)set * n * n;
    const uint8_t* mem = a.member + (size_t)set * n;
    const uint8_t* dng = a.dangling + (size_t)set * n;
    uint4* outp = a.out + 2 * (size_t)set * n;

    if (a.pre_status[set] != ET_OK) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) u256_store(outp + 2 * i, u256_zero());
        if (threadIdx.x == 0) a.status[set] = a.pre_status[set];
        return;
    }
    if (VALS_IN_LDS) {
        // n * n bytes; the carve-up above keeps v_l 16-byte aligned when n is a multiple of 4
        const size_t total = (size_t)n * n;
        if ((n & 3) == 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(vg);
            uint32_t* dst = reinterpret_cast<uint32_t*>(v_l);
            for (size_t t = threadIdx.x; t < total / 4; t += blockDim.x) dst[t] = src[t];
        } else {
            for (size_t t = threadIdx.x; t < total; t += blockDim.x) v_l[t] = vg[t];
        }
    }
    const uint8_t* vsrc = VALS_IN_LDS ? v_l : vg;
    const U256 inw = u256_load(a.inv_nm1_w + 2 * (size_t)set);
    uint32_t n_valid_local = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        u256_store(s_l + 2 * i, mem[i] ? a.initial : u256_zero());   // native.rs:317
        n_valid_local += mem[i] != 0;
    }
    __syncthreads();

    const uint32_t part = threadIdx.x / cols, i = threadIdx.x - part * cols;
    const uint32_t ii = i < n ? i : n - 1;                    // lanes past the last column compute and discard
    const uint32_t j_lo = (uint32_t)((uint64_t)part * n / parts), j_hi = (uint32_t)((uint64_t)(part + 1) * n / parts);
    const bool is_member = i < n && mem[ii] != 0;
    const bool is_dangling = i < n && dng[ii] != 0;

    for (uint32_t it = 0; it < a.num_iter; ++it) {
        // phase 1: xs_j = s_j * inv(sum_j) * 2^288, and the score mass of dangling peers
        U256 dpart = u256_zero();
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
            U256 sk = u256_load(s_l + 2 * k);
            U256 f = u256_load(a.scale + 2 * ((size_t)set * n + k));
            Acc t = acc_zero();
            acc_mac(t, sk, f);
            const U256 xk = acc_redc288(t);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xs_l[4 * k + q] = make_uint4(xk.v[2 * q] & 0xFFFFu, xk.v[2 * q] >> 16, xk.v[2 * q + 1] & 0xFFFFu, xk.v[2 * q + 1] >> 16);
            if (dng[k]) dpart = fr_add(dpart, sk);
        }
        if ((uint32_t)wave * 64 < n) {                        // the other wavefronts hold no element of s
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                U256 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(dpart.v[q], off, 64);
                dpart = fr_add(dpart, o);
            }
            if (lane == 0) u256_store(red_l + 2 * wave, dpart);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            U256 d = u256_zero();
            for (uint32_t w = 0; w * 64 < n; ++w) d = fr_add(d, u256_load(red_l + 2 * w));
#pragma unroll
            for (int q = 0; q < 8; ++q) sh_D[q] = d.v[q];
        }
        __syncthreads();

        // phase 2: new_s_i = sum_j v_ji * xs_j + (D - dangling_i * s_i) * inv(n_valid-1).
        // The host zeroed the diagonal and every opinion from or about an empty slot
        // (native.rs:241-259), so the loop has no conditions.
        Acc10 acc;
#pragma unroll
        for (int q = 0; q < 10; ++q) acc.v[q] = 0;
        uint32_t cs[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) cs[q] = 0;
        for (uint32_t j0 = j_lo; j0 < j_hi; j0 += 256) {
            const uint32_t j1 = j0 + 256 < j_hi ? j0 + 256 : j_hi;
            uint32_t idx = j0 * n + ii;                       // n <= 1024: fits 32 bits
            for (uint32_t j = j0; j < j1; ++j, idx += n) {
                const uint32_t op = vsrc[idx];
                // the same address in every lane: LDS broadcast; limb (16 bit) x opinion (8 bit) < 2^24
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 x = xs_l[4 * j + q];
                    cs[4 * q] += __umul24(x.x, op);
                    cs[4 * q + 1] += __umul24(x.y, op);
                    cs[4 * q + 2] += __umul24(x.z, op);
                    cs[4 * q + 3] += __umul24(x.w, op);
                }
            }
            acc10_fold(acc, cs);
        }
        if (part > 0 && i < n) {
#pragma unroll
            for (int q = 0; q < 10; ++q) part_l[((size_t)q * (parts - 1) + (part - 1)) * n + i] = acc.v[q];
        }
        __syncthreads();
        if (part == 0 && i < n) {
            U256 v = u256_zero();
            if (is_member) {
                for (uint32_t p = 0; p + 1 < parts; ++p) {
                    uint64_t c = 0;
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        c += (uint64_t)acc.v[q] + part_l[((size_t)q * (parts - 1) + p) * n + i];
                        acc.v[q] = (uint32_t)c;
                        c >>= 32;
                    }
                }
                Acc full = acc_zero();
#pragma unroll
                for (int q = 0; q < 10; ++q) full.v[q] = acc.v[q];
                U256 D;
#pragma unroll
                for (int q = 0; q < 8; ++q) D.v[q] = sh_D[q];
                U256 t = is_dangling ? fr_sub(D, u256_load(s_l + 2 * i)) : D;
                acc_mac(full, inw, t);
                v = acc_redc288(full);
            }
            u256_store(sn_l + 2 * i, v);
        }
        __syncthreads();
        uint4* tmp = s_l; s_l = sn_l; sn_l = tmp;
    }
    // output + conservation check (native.rs:331-334): sum of scores == INITIAL * n_valid
    U256 tot = u256_zero();
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        U256 sk = u256_load(s_l + 2 * k);
        u256_store(outp + 2 * k, sk);
        tot = fr_add(tot, sk);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        U256 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(tot.v[q], off, 64);
        tot = fr_add(tot, o);
        n_valid_local += __shfl_xor(n_valid_local, off, 64);
    }
    if (lane == 0) { u256_store(red_l + 2 * wave, tot); sh_nv[wave] = n_valid_local; }
    __syncthreads();
    if (threadIdx.x == 0) {
        U256 t = u256_zero();
        uint32_t nv = 0;
        for (int w = 0; w < n_waves; ++w) { t = fr_add(t, u256_load(red_l + 2 * w)); nv += sh_nv[w]; }
        U256 expect = u256_zero();
        for (uint32_t k = 0; k < nv; ++k) expect = fr_add(expect, a.initial);   // nv <= 1024
        a.status[set] = u256_eq(t, expect) ? ET_OK : ET_E_SUM_MISMATCH;
    }
}

// Byte-valued opinions, sets of up to 256 slots: the dense product runs on the matrix cores.
// With xs_j cut into 32 byte limbs, new_s_i = sum_j v_ji * xs_j is a (32 limbs x sources) by
// (sources x columns) product of bytes whose 32-bit sums are exact (256 * 255 * 255 < 2^24), i.e.
// one i8 MFMA tile per 32 columns and 32 sources.  The MFMA multiplies SIGNED bytes, so both
// operands are stored minus 128 (byte ^ 0x80) and the cross terms are added back:
//   sum_j v_j x_j = M + 128 * sum_j v_j + 128 * M1,   M = sum_j (v_j-128)(x_j-128),  M1 = sum_j (x_j-128);
// sum_j v_j is a constant of the column and M1 comes from one more MFMA against a matrix of ones.
// A peer without valid opinions is a dense row of ones with sum n_valid - 1 here, as in the
// reference (native.rs:263-278), so there is no separate dangling term.  A lane owns one column:
// its score and factor stay in registers, the opinion bytes it multiplies stay in registers as
// MFMA operands for all iterations, and only the 32 x n bytes of xs go through LDS.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct BatchArgsM {
    BatchArgs b;             // vals: n_sets * np * np bytes, [column i][source j], each v_ji ^ 0x80
    const uint32_t* colsum;  // n_sets * np : sum_j v_ji
    uint32_t np;             // n rounded up to a multiple of 64; blockDim.x == np
    U256 initial_m;          // INITIAL_SCORE * 2^256 mod r
    uint32_t pack;           // sets per workgroup (k_batch_fr_mfma_packed), else 1
    uint32_t n_sets;
};

// The two 32 x 32 result tiles of a wavefront (limb rows x columns), the tile against the matrix of ones and the
// per-column constants, turned into ONE column sum per lane.  Register q of a lane is limb
// (q & 3) + 8 * (q >> 2) + 4 * h of its column (h = lane half): registers 4g..4g+3 are the four bytes of 32-bit
// word 2g + h.  Lane half 0 keeps tile 0, lane half 1 keeps tile 1; the words of the other tile are swapped
// with lane ^ 32.
ET_DEV Acc10 mfma_tiles_to_column(const v16i& acc0, const v16i& acc1, const v16i& acc2, int kc0, int kc1, int h) {
    uint64_t keep[4], give[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint64_t t0 = 0, t1 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m1 = 128 * acc2[4 * g + q];
            t0 += (uint64_t)(uint32_t)(acc0[4 * g + q] + m1 + kc0) << (8 * q);
            t1 += (uint64_t)(uint32_t)(acc1[4 * g + q] + m1 + kc1) << (8 * q);
        }
        keep[g] = h ? t1 : t0;
        give[g] = h ? t0 : t1;
    }
    Acc10 acc;
    uint64_t c = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t lo = __shfl_xor((uint32_t)give[g], 32, 64), hi = __shfl_xor((uint32_t)(give[g] >> 32), 32, 64);
        const uint64_t got = ((uint64_t)hi << 32) | lo;
        const uint64_t even = h ? got : keep[g], odd = h ? keep[g] : got;
        c += even;
        acc.v[2 * g] = (uint32_t)c;
        c >>= 32;
        c += odd;
        acc.v[2 * g + 1] = (uint32_t)c;
        c >>= 32;
    }
    acc.v[8] = (uint32_t)c;
    acc.v[9] = (uint32_t)(c >> 32);
    return acc;
}

static const uint32_t MFMA_ROW_PAD = 16;   // bytes between limb rows of xs in LDS: spreads the 16-byte operand reads over the banks

template <int KS>   // np / 32
__global__ __launch_bounds__(KS * 32) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_batch_fr_mfma(BatchArgsM am) {
    const BatchArgs& a = am.b;
    constexpr uint32_t NP = KS * 32, ROW = NP + MFMA_ROW_PAD;
    __shared__ __attribute__((aligned(16))) uint8_t xsb[2][32 * ROW];
    __shared__ uint4 red_l[2 * (KS / 2)];
    __shared__ uint32_t sh_nv[KS / 2];
    const uint32_t n = a.n, set = blockIdx.x, e = threadIdx.x;
    const int lane = e & 63, wave = e >> 6, r = lane & 31, h = lane >> 5;
    uint4* outp = a.out + 2 * (size_t)set * n;
    if (a.pre_status[set] != ET_OK) {
        if (e < n) u256_store(outp + 2 * e, u256_zero());
        if (e == 0) a.status[set] = a.pre_status[set];
        return;
    }
    // operands that never change: the opinion bytes of this lane's two columns (one per 32-column tile of
    // the wavefront), 16 sources per lane half and k-step
    const uint8_t* vt = reinterpret_cast<const uint8_t*>(a.vals) + (size_t)set * NP * NP;
    v4i b0[KS], b1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        b0[ks] = *reinterpret_cast<const v4i*>(vt + (size_t)(wave * 64 + r) * NP + ks * 32 + 16 * h);
        b1[ks] = *reinterpret_cast<const v4i*>(vt + (size_t)(wave * 64 + 32 + r) * NP + ks * 32 + 16 * h);
    }
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int kc0 = (int)(128u * am.colsum[(size_t)set * NP + wave * 64 + r]);
    const int kc1 = (int)(128u * am.colsum[(size_t)set * NP + wave * 64 + 32 + r]);
    const bool is_member = e < n && a.member[(size_t)set * n + e] != 0;
    const U256 scale = is_member ? u256_load(a.scale + 2 * ((size_t)set * n + e)) : u256_zero();
    U256 s = is_member ? a.initial : u256_zero();              // native.rs:317
    // Between iterations a score is carried as sm = s * 2^256 mod r, below 2^256 but not below r: the column
    // sum (s * 2^288 as an integer below 2^273) needs ONE Montgomery round to get there, the product with the
    // factor (inv(sum) * 2^288) eight rounds to come back as xs = s * inv(sum) * 2^288, and only the last
    // iteration pays the nine rounds to the canonical score.
    U256 sm = is_member ? am.initial_m : u256_zero();

    for (uint32_t it = 0; it < a.num_iter; ++it) {
        uint8_t* xb = xsb[it & 1];
        // xs_e as 32 bytes, one row of LDS per limb
        const U256 xs = fr_mul_redc_loose8(sm, scale);
#pragma unroll
        for (int l = 0; l < 32; ++l) xb[l * ROW + e] = (uint8_t)(((xs.v[l >> 2] >> (8 * (l & 3))) & 0xFFu) ^ 0x80u);
        __syncthreads();
        v16i acc0 = {0}, acc1 = {0}, acc2 = {0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4i x = *reinterpret_cast<const v4i*>(xb + r * ROW + ks * 32 + 16 * h);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b0[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b1[ks], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, ones, acc2, 0, 0, 0);
        }
        const Acc10 acc = mfma_tiles_to_column(acc0, acc1, acc2, kc0, kc1, h);
        if (it + 1 == a.num_iter) s = is_member ? acc_redc288(acc_from_acc10(acc)) : u256_zero();
        else sm = is_member ? acc_redc_loose<1>(acc_from_acc10(acc)) : u256_zero();
    }
    // output + conservation check (native.rs:331-334): sum of scores == INITIAL * n_valid
    if (e < n) u256_store(outp + 2 * e, s);
    U256 tot = s;
    uint32_t nv = is_member;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        U256 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(tot.v[q], off, 64);
        tot = fr_add(tot, o);
        nv += __shfl_xor(nv, off, 64);
    }
    if (lane == 0) { u256_store(red_l + 2 * wave, tot); sh_nv[wave] = nv; }
    __syncthreads();
    if (e == 0) {
        U256 t = u256_zero();
        uint32_t members = 0;
        for (int w = 0; w < KS / 2; ++w) { t = fr_add(t, u256_load(red_l + 2 * w)); members += sh_nv[w]; }
        U256 expect = u256_zero();
        for (uint32_t k = 0; k < members; ++k) expect = fr_add(expect, a.initial);
        a.status[set] = u256_eq(t, expect) ? ET_OK : ET_E_SUM_MISMATCH;
    }
}

// Sets of up to 32 slots (the reference's production set has 4): floor(64 / n) sets share ONE wavefront.  Their
// opinion matrices sit on the diagonal of a 64 x 64 byte matrix, so the same two MFMA tiles multiply all of them
// at once and every lane still owns one slot of one set; only the conservation check is per set.
__global__ __launch_bounds__(64) void k_batch_fr_mfma_packed(BatchArgsM am) {
    const BatchArgs& a = am.b;
    constexpr uint32_t NP = 64, ROW = NP + MFMA_ROW_PAD, KS = 2;
    __shared__ __attribute__((aligned(16))) uint8_t xsb[2][32 * ROW];
    __shared__ uint4 s_l[2 * 64];
    const uint32_t n = a.n, pack = am.pack, e = threadIdx.x;
    const int r = e & 31, h = e >> 5;
    const size_t first_set = (size_t)blockIdx.x * pack;
    const uint32_t sets_here = (uint32_t)(am.n_sets - first_set < pack ? am.n_sets - first_set : pack);
    const uint32_t n_eff = sets_here * n;
    const size_t base = first_set * n;
    const uint8_t* vt = reinterpret_cast<const uint8_t*>(a.vals) + (size_t)blockIdx.x * NP * NP;
    v4i b0[KS], b1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        b0[ks] = *reinterpret_cast<const v4i*>(vt + (size_t)r * NP + ks * 32 + 16 * h);
        b1[ks] = *reinterpret_cast<const v4i*>(vt + (size_t)(32 + r) * NP + ks * 32 + 16 * h);
    }
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int kc0 = (int)(128u * am.colsum[(size_t)blockIdx.x * NP + r]);
    const int kc1 = (int)(128u * am.colsum[(size_t)blockIdx.x * NP + 32 + r]);
    // a set with fewer than two members (native.rs:289) computes nothing and reports its status
    const bool is_member = e < n_eff && a.member[base + e] != 0 && a.pre_status[first_set + e / n] == ET_OK;
    const U256 scale = is_member ? u256_load(a.scale + 2 * (base + e)) : u256_zero();
    U256 s = is_member ? a.initial : u256_zero();
    U256 sm = is_member ? am.initial_m : u256_zero();
    for (uint32_t it = 0; it < a.num_iter; ++it) {
        uint8_t* xb = xsb[it & 1];
        const U256 xs = fr_mul_redc_loose8(sm, scale);
#pragma unroll
        for (int l = 0; l < 32; ++l) xb[l * ROW + e] = (uint8_t)(((xs.v[l >> 2] >> (8 * (l & 3))) & 0xFFu) ^ 0x80u);
        __syncthreads();
        v16i acc0 = {0}, acc1 = {0}, acc2 = {0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v4i x = *reinterpret_cast<const v4i*>(xb + r * ROW + ks * 32 + 16 * h);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b0[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b1[ks], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, ones, acc2, 0, 0, 0);
        }
        const Acc10 acc = mfma_tiles_to_column(acc0, acc1, acc2, kc0, kc1, h);
        if (it + 1 == a.num_iter) s = is_member ? acc_redc288(acc_from_acc10(acc)) : u256_zero();
        else sm = is_member ? acc_redc_loose<1>(acc_from_acc10(acc)) : u256_zero();
    }
    if (e < n_eff) u256_store(a.out + 2 * (base + e), s);
    u256_store(s_l + 2 * e, s);
    __syncthreads();
    if (e < sets_here) {                                   // conservation per set (native.rs:331-334)
        const size_t gs = first_set + e;
        int32_t status = a.pre_status[gs];
        if (status == ET_OK) {
            U256 tot = u256_zero(), expect = u256_zero();
            for (uint32_t k = 0; k < n; ++k) {
                tot = fr_add(tot, u256_load(s_l + 2 * (e * n + k)));
                if (a.member[gs * n + k]) expect = fr_add(expect, a.initial);
            }
            status = u256_eq(tot, expect) ? ET_OK : ET_E_SUM_MISMATCH;
        }
        a.status[gs] = status;
    }
}

// The same for sets of up to 1024 slots, whose opinion bytes do not fit the registers of their wavefronts:
// every k-step's two operands are read again from the transposed matrix in global memory (a set's matrix is
// at most 1 MiB and stays in the caches over its 20 iterations); LDS holds the two copies of xs only.
__global__ __launch_bounds__(1024) void k_batch_fr_mfma_stream(BatchArgsM am) {
    const BatchArgs& a = am.b;
    extern __shared__ __attribute__((aligned(16))) uint8_t xsb_dyn[];
    __shared__ uint4 red_l[2 * 16];
    __shared__ uint32_t sh_nv[16];
    const uint32_t NP = am.np, ROW = NP + MFMA_ROW_PAD, KS = NP / 32;
    const uint32_t n = a.n, set = blockIdx.x, e = threadIdx.x;
    const int lane = e & 63, wave = e >> 6, r = lane & 31, h = lane >> 5, n_waves = NP / 64;
    uint4* outp = a.out + 2 * (size_t)set * n;
    if (a.pre_status[set] != ET_OK) {
        if (e < n) u256_store(outp + 2 * e, u256_zero());
        if (e == 0) a.status[set] = a.pre_status[set];
        return;
    }
    const uint8_t* vt = reinterpret_cast<const uint8_t*>(a.vals) + (size_t)set * NP * NP;
    const uint8_t* __restrict__ p0 = vt + (size_t)(wave * 64 + r) * NP + 16 * h;
    const uint8_t* __restrict__ p1 = p0 + (size_t)32 * NP;
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int kc0 = (int)(128u * am.colsum[(size_t)set * NP + wave * 64 + r]);
    const int kc1 = (int)(128u * am.colsum[(size_t)set * NP + wave * 64 + 32 + r]);
    const bool is_member = e < n && a.member[(size_t)set * n + e] != 0;
    const U256 scale = is_member ? u256_load(a.scale + 2 * ((size_t)set * n + e)) : u256_zero();
    U256 s = is_member ? a.initial : u256_zero();
    U256 sm = is_member ? am.initial_m : u256_zero();

    for (uint32_t it = 0; it < a.num_iter; ++it) {
        uint8_t* xb = xsb_dyn + (size_t)(it & 1) * 32 * ROW;
        const U256 xs = fr_mul_redc_loose8(sm, scale);
#pragma unroll
        for (int l = 0; l < 32; ++l) xb[l * ROW + e] = (uint8_t)(((xs.v[l >> 2] >> (8 * (l & 3))) & 0xFFu) ^ 0x80u);
        __syncthreads();
        v16i acc0 = {0}, acc1 = {0}, acc2 = {0};
        const uint8_t* xrow = xb + r * ROW + 16 * h;
        for (uint32_t ks = 0; ks < KS; ++ks) {
            const v4i x = *reinterpret_cast<const v4i*>(xrow + ks * 32);
            const v4i b0 = *reinterpret_cast<const v4i*>(p0 + ks * 32);
            const v4i b1 = *reinterpret_cast<const v4i*>(p1 + ks * 32);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, b1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, ones, acc2, 0, 0, 0);
        }
        const Acc10 acc = mfma_tiles_to_column(acc0, acc1, acc2, kc0, kc1, h);
        if (it + 1 == a.num_iter) s = is_member ? acc_redc288(acc_from_acc10(acc)) : u256_zero();
        else sm = is_member ? acc_redc_loose<1>(acc_from_acc10(acc)) : u256_zero();
    }
    if (e < n) u256_store(outp + 2 * e, s);
    U256 tot = s;
    uint32_t nv = is_member;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        U256 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = __shfl_xor(tot.v[q], off, 64);
        tot = fr_add(tot, o);
        nv += __shfl_xor(nv, off, 64);
    }
    if (lane == 0) { u256_store(red_l + 2 * wave, tot); sh_nv[wave] = nv; }
    __syncthreads();
    if (e == 0) {
        U256 t = u256_zero();
        uint32_t members = 0;
        for (int w = 0; w < n_waves; ++w) { t = fr_add(t, u256_load(red_l + 2 * w)); members += sh_nv[w]; }
        U256 expect = u256_zero();
        for (uint32_t k = 0; k < members; ++k) expect = fr_add(expect, a.initial);
        a.status[set] = u256_eq(t, expect) ? ET_OK : ET_E_SUM_MISMATCH;
    }
}

static size_t batch_u8_lds_bytes(uint32_t n, uint32_t parts, bool vals_in_lds) {
    size_t b = (8 * (size_t)n + 2 * 16 + 2) * sizeof(uint4) + (size_t)40 * (parts - 1) * n;
    if (vals_in_lds) b += (size_t)n * n;
    return b;
}

struct DeviceBatch {
    int device = -1;
    uint32_t n_sets = 0, n = 0;
    bool vals_u8 = false;
    bool mfma = false;    // byte opinions and n <= 256: d_vals holds the transposed, sign-shifted matrix
    uint32_t np = 0;      // n rounded up to a multiple of 64 (mfma only)
    uint32_t pack = 1;    // sets per workgroup: floor(64 / n) for sets of up to 32 slots (mfma only)
    uint32_t* d_colsum = nullptr;
    void* d_vals = nullptr;
    uint8_t* d_member = nullptr;
    uint8_t* d_dangling = nullptr;
    uint4* d_scale = nullptr;
    uint4* d_inw = nullptr;
    int32_t* d_pre = nullptr;
    uint4* d_out = nullptr;
    int32_t* d_status = nullptr;
    uint64_t edges = 0;   // sum over sets of n_valid * (n_valid - 1): dense edge count of the reference
    bool owns = true;     // false: the arrays are carved out of the library's resident scratch array, nothing to free
    // Small batches (one set through the set API): scores and status are written by the kernel straight into pinned host
    // memory the GPU sees, so a converge is one launch and one wait, without copies (d_out / d_status then point into it).
    void* h_map = nullptr;
    ~DeviceBatch() {
        if (!owns) return;
        if (device >= 0) (void)hipSetDevice(device);
        if (h_map) { (void)hipHostFree(h_map); d_out = nullptr; d_status = nullptr; }
        void* p[] = {d_vals, d_member, d_dangling, d_scale, d_inw, d_pre, d_out, d_status, d_colsum};
        for (void* q : p) if (q) (void)hipFree(q);
    }
};

// Where a batch's device arrays come from: hipMalloc (arena == nullptr), or consecutive 256-byte aligned pieces of
// a region that already exists (the entry point that allocates nothing).
struct Arena {
    uint8_t* base = nullptr;
    size_t cap = 0, used = 0;
};
static int batch_dev_alloc(Arena* arena, void** p, size_t bytes) {
    if (!arena) {
        HIP_TRY_B(hipMalloc(p, bytes ? bytes : 1));
        return ET_OK;
    }
    const size_t need = (bytes + 255) / 256 * 256;
    if (arena->used + need > arena->cap) {
        set_error("batch needs more than the " + std::to_string(arena->cap) + " bytes of resident scratch memory");
        return ET_E_BUFFER_TOO_SMALL;
    }
    *p = arena->base + arena->used;
    arena->used += need;
    return ET_OK;
}
#define BATCH_ALLOC(ptr, bytes)                                                        \
    do {                                                                               \
        int _rc = batch_dev_alloc(arena, (void**)&(ptr), (bytes));                     \
        if (_rc) return _rc;                                                           \
    } while (0)

template <typename F>
static void parallel_sets(uint32_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(16u, hw ? hw : 1u));
    if (n < 8) nt = 1;
    std::atomic<uint32_t> next(0);
    auto work = [&]() { for (;;) { uint32_t i = next.fetch_add(1); if (i >= n) break; f(i); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

static U256 to_u256b(const FrRaw& r) {
    U256 u;
    for (int i = 0; i < 4; ++i) { u.v[2 * i] = (uint32_t)r.l[i]; u.v[2 * i + 1] = (uint32_t)(r.l[i] >> 32); }
    return u;
}

static size_t batch_lds_bytes(uint32_t n, bool vals_in_lds, size_t vsize) {
    size_t b = (6 * (size_t)n + 2 * 4 + 2) * sizeof(uint4);
    if (vals_in_lds) b += (size_t)n * n * vsize;
    return b;
}

}  // namespace et

using namespace et;
struct et_batch {
    DeviceBatch b;
};

static int batch_create(uint32_t n_sets, uint32_t n, const uint8_t* member, const uint32_t* values, int device, Arena* arena,
                        et_batch** out) {
    if (!out || !values || n_sets == 0 || n == 0) return ET_E_INVALID_ARG;
    if (n > 1024) { set_error("a batched set holds at most 1024 slots"); return ET_E_UNSUPPORTED; }
    int ndev = 0;
    HIP_TRY_B(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    if (device < 0) HIP_TRY_B(hipGetDevice(&device));
    HIP_TRY_B(hipSetDevice(device));
    const size_t rows = (size_t)n_sets * n;
    std::vector<uint8_t> mem(rows), dang(rows, 0);
    std::vector<FrRaw> scale(rows, FrRaw{{0, 0, 0, 0}}), inw(n_sets, FrRaw{{0, 0, 0, 0}});
    std::vector<int32_t> pre(n_sets, ET_OK);
    std::vector<uint64_t> edges(n_sets, 0);
    std::atomic<uint32_t> vmax(0);
    // filter_peers_ops + row sums + inversion, one set per task (native.rs:234-283, :305-309)
    parallel_sets(n_sets, [&](uint32_t s) {
        const uint32_t* v = values + (size_t)s * n * n;
        uint8_t* m = mem.data() + (size_t)s * n;
        uint32_t nv = 0, local_max = 0;
        for (uint32_t i = 0; i < n; ++i) { m[i] = member ? member[(size_t)s * n + i] != 0 : 1; nv += m[i]; }
        if (nv < 2) { pre[s] = ET_E_INSUFFICIENT_PEERS; return; }
        edges[s] = (uint64_t)nv * (nv - 1);
        std::vector<Fr> sums(n);
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t sum = 0;
            if (m[i]) for (uint32_t j = 0; j < n; ++j) if (j != i && m[j]) { sum += v[(size_t)i * n + j]; local_max = std::max(local_max, v[(size_t)i * n + j]); }
            if (m[i] && sum == 0) dang[(size_t)s * n + i] = 1;
            sums[i] = Fr::from_u64(sum);
        }
        fr_batch_invert(sums);
        // the plain inverse for now: the power of two it carries depends on the kernel chosen below
        for (uint32_t i = 0; i < n; ++i) if (m[i] && !dang[(size_t)s * n + i]) scale[(size_t)s * n + i] = sums[i].to_raw();
        inw[s] = (Fr::from_u64(nv - 1).invert_or_zero() * Fr::one().pow2k(288)).to_raw();
        uint32_t cur = vmax.load();
        while (local_max > cur && !vmax.compare_exchange_weak(cur, local_max)) {}
    });
    std::unique_ptr<et_batch> h(new et_batch());
    DeviceBatch& b = h->b;
    b.device = device; b.n_sets = n_sets; b.n = n;
    b.owns = arena == nullptr;
    for (uint64_t e : edges) b.edges += e;
    // opinions about empty slots or from empty slots never count: zero them once here so the
    // kernel only has to skip the diagonal
    const size_t cells = rows * n;
    b.vals_u8 = vmax.load() <= 255;
    const Fr two288 = Fr::one().pow2k(288), two576 = Fr::one().pow2k(576);
    const char* no_mfma = getenv("ET_BATCH_NO_MFMA");   // measurement knob: the VALU kernel for small byte sets too
    b.mfma = b.vals_u8 && !(no_mfma && no_mfma[0] == '1');   // n <= 1024 was checked above
    if (b.mfma) {
        // [column i][source j] bytes of v_ji - 128, padded to np x np with v = 0; a peer without valid
        // opinions becomes a row of ones over the other members with factor inv(n_valid - 1) (native.rs:263-278)
        const uint32_t np = (n + 63) / 64 * 64;
        const uint32_t pack = n <= 32 ? 64 / n : 1;      // small sets share a workgroup: block-diagonal matrix
        const size_t n_groups = ((size_t)n_sets + pack - 1) / pack;
        b.np = np;
        b.pack = pack;
        const size_t vt_bytes = n_groups * np * np;
        std::unique_ptr<uint8_t[]> vt(new uint8_t[vt_bytes]);     // every group's task fills its own matrix
        std::vector<uint32_t> colsum(n_groups * np, 0);
        parallel_sets((uint32_t)n_groups, [&](uint32_t grp) {
            uint8_t* gm = vt.get() + (size_t)grp * np * np;
            std::memset(gm, 0x80, (size_t)np * np);
            for (uint32_t s = grp * pack; s < std::min<uint64_t>((uint64_t)(grp + 1) * pack, n_sets); ++s) {
                if (pre[s] != ET_OK) continue;
                const uint32_t* v = values + (size_t)s * n * n;
                const uint8_t* m = mem.data() + (size_t)s * n;
                const uint8_t* dg = dang.data() + (size_t)s * n;
                const size_t corner = (size_t)(s % pack) * n;   // this set's corner on the diagonal of its group's matrix
                uint8_t* o = gm + corner * np + corner;
                uint32_t* cs = colsum.data() + (size_t)grp * np + corner;
                uint32_t nv = 0;
                for (uint32_t i = 0; i < n; ++i) nv += m[i];
                // factors carry 2^288: the kernel multiplies them with scores that carry 2^256 and divides by 2^256
                const FrRaw uniform = (Fr::from_u64(nv - 1).invert_or_zero() * two288).to_raw();
                for (uint32_t j = 0; j < n; ++j) {
                    if (!m[j]) continue;
                    FrRaw& f = scale[(size_t)s * n + j];
                    f = dg[j] ? uniform : (Fr::from_raw(f) * two288).to_raw();
                }
                // transpose in 32 x 32 tiles so that reads and writes both stay in the cache
                for (uint32_t jb = 0; jb < n; jb += 32)
                    for (uint32_t ib = 0; ib < n; ib += 32)
                        for (uint32_t j = jb; j < std::min(n, jb + 32); ++j) {
                            if (!m[j]) continue;
                            for (uint32_t i = ib; i < std::min(n, ib + 32); ++i) {
                                if (!m[i] || i == j) continue;
                                const uint32_t x = dg[j] ? 1u : v[(size_t)j * n + i];
                                o[(size_t)i * np + j] = (uint8_t)(x ^ 0x80u);
                                cs[i] += x;
                            }
                        }
            }
        });
        BATCH_ALLOC(b.d_vals, vt_bytes);
        HIP_TRY_B(hipMemcpy(b.d_vals, vt.get(), vt_bytes, hipMemcpyHostToDevice));
        BATCH_ALLOC(b.d_colsum, colsum.size() * 4);
        HIP_TRY_B(hipMemcpy(b.d_colsum, colsum.data(), colsum.size() * 4, hipMemcpyHostToDevice));
    } else {
        // the vector kernels multiply the factor with a canonical score and divide by 2^288 twice per iteration
        parallel_sets(n_sets, [&](uint32_t s) {
            for (uint32_t i = 0; i < n; ++i) {
                FrRaw& f = scale[(size_t)s * n + i];
                f = (Fr::from_raw(f) * two576).to_raw();
            }
        });
    }
    if (!b.mfma && b.vals_u8) {
        std::vector<uint8_t> v8(cells);
        parallel_sets(n_sets, [&](uint32_t s) {
            const uint32_t* v = values + (size_t)s * n * n;
            const uint8_t* m = mem.data() + (size_t)s * n;
            uint8_t* o = v8.data() + (size_t)s * n * n;
            for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < n; ++j) o[(size_t)i * n + j] = (m[i] && m[j] && i != j) ? (uint8_t)v[(size_t)i * n + j] : 0;
        });
        BATCH_ALLOC(b.d_vals, cells);
        HIP_TRY_B(hipMemcpy(b.d_vals, v8.data(), cells, hipMemcpyHostToDevice));
    } else if (!b.mfma) {
        std::vector<uint32_t> v32(cells);
        parallel_sets(n_sets, [&](uint32_t s) {
            const uint32_t* v = values + (size_t)s * n * n;
            const uint8_t* m = mem.data() + (size_t)s * n;
            uint32_t* o = v32.data() + (size_t)s * n * n;
            for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < n; ++j) o[(size_t)i * n + j] = (m[i] && m[j] && i != j) ? v[(size_t)i * n + j] : 0;
        });
        BATCH_ALLOC(b.d_vals, cells * 4);
        HIP_TRY_B(hipMemcpy(b.d_vals, v32.data(), cells * 4, hipMemcpyHostToDevice));
    }
    BATCH_ALLOC(b.d_member, rows);
    BATCH_ALLOC(b.d_dangling, rows);
    BATCH_ALLOC(b.d_scale, rows * 32);
    BATCH_ALLOC(b.d_inw, (size_t)n_sets * 32);
    BATCH_ALLOC(b.d_pre, (size_t)n_sets * 4);
    const size_t out_bytes = (rows * 32 + 255) / 256 * 256;
    if (!arena && out_bytes + (size_t)n_sets * 4 <= (256u << 10) && !getenv("ET_BATCH_NO_MAPPED_OUTPUT")) {
        HIP_TRY_B(hipHostMalloc(&b.h_map, out_bytes + (size_t)n_sets * 4, hipHostMallocMapped));
        void* dev = nullptr;
        HIP_TRY_B(hipHostGetDevicePointer(&dev, b.h_map, 0));
        b.d_out = static_cast<uint4*>(dev);
        b.d_status = reinterpret_cast<int32_t*>(static_cast<char*>(dev) + out_bytes);
    } else {
        BATCH_ALLOC(b.d_out, rows * 32);
        BATCH_ALLOC(b.d_status, (size_t)n_sets * 4);
    }
    HIP_TRY_B(hipMemcpy(b.d_member, mem.data(), rows, hipMemcpyHostToDevice));
    HIP_TRY_B(hipMemcpy(b.d_dangling, dang.data(), rows, hipMemcpyHostToDevice));
    HIP_TRY_B(hipMemcpy(b.d_scale, scale.data(), rows * 32, hipMemcpyHostToDevice));
    HIP_TRY_B(hipMemcpy(b.d_inw, inw.data(), (size_t)n_sets * 32, hipMemcpyHostToDevice));
    HIP_TRY_B(hipMemcpy(b.d_pre, pre.data(), (size_t)n_sets * 4, hipMemcpyHostToDevice));
    *out = h.release();
    return ET_OK;
}

extern "C" {

int et_batch_create(uint32_t n_sets, uint32_t n, const uint8_t* member, const uint32_t* values, int device, et_batch** out) {
    return batch_create(n_sets, n, member, values, device, nullptr, out);
}

void et_batch_free(et_batch* b) { delete b; }

int et_batch_run_fr_async(et_batch* h, uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi, void* stream) {
    if (!h) return ET_E_INVALID_ARG;
    DeviceBatch& b = h->b;
    HIP_TRY_B(hipSetDevice(b.device));
    BatchArgs a;
    a.vals = b.d_vals; a.member = b.d_member; a.dangling = b.d_dangling; a.scale = b.d_scale; a.inv_nm1_w = b.d_inw;
    a.pre_status = b.d_pre; a.out = b.d_out; a.status = b.d_status; a.n = b.n; a.num_iter = num_iterations;
    a.initial = to_u256b(Fr::from_u128(initial_lo, initial_hi).to_raw());
    const size_t budget = 150 * 1024;
    hipStream_t st = (hipStream_t)stream;
    if (b.mfma) {
        BatchArgsM am;
        am.b = a; am.colsum = b.d_colsum; am.np = b.np;
        am.initial_m = to_u256b(Fr::from_u128(initial_lo, initial_hi).pow2k(256).to_raw());
        am.pack = b.pack; am.n_sets = b.n_sets;
        if (b.pack > 1) {
            hipLaunchKernelGGL(k_batch_fr_mfma_packed, dim3((b.n_sets + b.pack - 1) / b.pack), dim3(64), 0, st, am);
        } else switch (b.np / 64) {
            case 1: hipLaunchKernelGGL((k_batch_fr_mfma<2>), dim3(b.n_sets), dim3(64), 0, st, am); break;
            case 2: hipLaunchKernelGGL((k_batch_fr_mfma<4>), dim3(b.n_sets), dim3(128), 0, st, am); break;
            case 3: hipLaunchKernelGGL((k_batch_fr_mfma<6>), dim3(b.n_sets), dim3(192), 0, st, am); break;
            case 4: hipLaunchKernelGGL((k_batch_fr_mfma<8>), dim3(b.n_sets), dim3(256), 0, st, am); break;
            default: {
                const size_t lds = (size_t)2 * 32 * (b.np + MFMA_ROW_PAD);
                HIP_TRY_B(hipFuncSetAttribute((const void*)k_batch_fr_mfma_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_batch_fr_mfma_stream, dim3(b.n_sets), dim3(b.np), lds, st, am);
            }
        }
    } else if (b.vals_u8) {
        BatchArgsU8 au;
        au.b = a;
        au.cols = (b.n + 63) / 64 * 64;
        au.parts = std::max(1u, std::min(1024u / au.cols, b.n / 32));
        bool in_lds = true;
        if (batch_u8_lds_bytes(b.n, au.parts, true) > budget) {
            // prefer the resident opinion matrix over the split when both do not fit
            if (batch_u8_lds_bytes(b.n, 1, true) <= budget) au.parts = 1;
            else in_lds = false;
        }
        const size_t lds = batch_u8_lds_bytes(b.n, au.parts, in_lds);
        const dim3 block(au.cols * au.parts);
        if (in_lds) {
            HIP_TRY_B(hipFuncSetAttribute((const void*)k_batch_fr_u8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
            hipLaunchKernelGGL((k_batch_fr_u8<true>), dim3(b.n_sets), block, lds, st, au);
        } else {
            HIP_TRY_B(hipFuncSetAttribute((const void*)k_batch_fr_u8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
            hipLaunchKernelGGL((k_batch_fr_u8<false>), dim3(b.n_sets), block, lds, st, au);
        }
    } else {
        bool in_lds = batch_lds_bytes(b.n, true, 4) <= budget;
        size_t lds = batch_lds_bytes(b.n, in_lds, 4);
        if (in_lds) {
            HIP_TRY_B(hipFuncSetAttribute((const void*)k_batch_fr<uint32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
            hipLaunchKernelGGL((k_batch_fr<uint32_t, true>), dim3(b.n_sets), dim3(BATCH_THREADS), lds, st, a);
        } else {
            hipLaunchKernelGGL((k_batch_fr<uint32_t, false>), dim3(b.n_sets), dim3(BATCH_THREADS), lds, st, a);
        }
    }
    HIP_TRY_B(hipGetLastError());
    return ET_OK;
}

int et_batch_result_fr(et_batch* h, et_fr* out, int32_t* status, void* stream) {
    if (!h || !out) return ET_E_INVALID_ARG;
    DeviceBatch& b = h->b;
    HIP_TRY_B(hipSetDevice(b.device));
    hipStream_t st = (hipStream_t)stream;
    if (b.h_map) {                                         // the kernel wrote into host memory: wait for it, nothing to copy
        HIP_TRY_B(hipStreamSynchronize(st));
        const size_t out_bytes = ((size_t)b.n_sets * b.n * 32 + 255) / 256 * 256;
        std::memcpy(out, b.h_map, (size_t)b.n_sets * b.n * 32);
        if (status) std::memcpy(status, static_cast<const char*>(b.h_map) + out_bytes, (size_t)b.n_sets * 4);
        return ET_OK;
    }
    HIP_TRY_B(hipMemcpyAsync(out, b.d_out, (size_t)b.n_sets * b.n * 32, hipMemcpyDeviceToHost, st));
    if (status) HIP_TRY_B(hipMemcpyAsync(status, b.d_status, (size_t)b.n_sets * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY_B(hipStreamSynchronize(st));
    return ET_OK;
}

int et_batch_edges(const et_batch* h, uint64_t* edges) {
    if (!h || !edges) return ET_E_INVALID_ARG;
    *edges = h->b.edges;
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The variant of et_batch_converge_fr that allocates no device memory.  Everything the batch kernels need beyond
// the caller's host arrays (the filtered, transposed opinion bytes, column sums, factors, member flags, per-set
// status, and the staging of the scores) lives in ONE array that is part of the library's code object: every GPU
// the library runs on holds its own copy, at its own address, from the first call on it until the process ends.
// Host code never uses the array's name as an address (on the host it is only a handle): the device address is
// asked for once per device, with that device current, and kept.
// ---------------------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[ET_NOALLOC_SCRATCH_BYTES];

static std::mutex g_scratch_lock;                   // the array is shared by every call on a GPU: one at a time
static void* g_scratch_ptr[ET_NOALLOC_MAX_DEVICES]; // per device: address of that GPU's copy, null until first use

// Address of the current device's copy of the array (device `dev` must be current).
static int scratch_address(int dev, void** out) {
    if (dev < 0 || dev >= ET_NOALLOC_MAX_DEVICES) { set_error("device ordinal out of range"); return ET_E_OUT_OF_RANGE; }
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY_B(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { set_error("hipGetSymbolAddress returned a null address"); return ET_E_HIP; }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return ET_OK;
}

int et_batch_converge_fr_noalloc(uint32_t n_sets, uint32_t n, const uint8_t* member, const uint32_t* values,
                                 uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi, et_fr* out,
                                 int32_t* status) {
    if (!values || !out || n_sets == 0 || n == 0) return ET_E_INVALID_ARG;
    int ndev = 0, dev = -1;
    HIP_TRY_B(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    HIP_TRY_B(hipGetDevice(&dev));
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    void* base = nullptr;
    int rc = scratch_address(dev, &base);
    if (rc) return rc;
    Arena arena;
    arena.base = static_cast<uint8_t*>(base);
    arena.cap = ET_NOALLOC_SCRATCH_BYTES;
    et_batch* b = nullptr;
    // carves the batch's arrays out of the scratch array and fills all of them except scores and status; a batch
    // that needs more than the array holds is refused there (ET_E_BUFFER_TOO_SMALL)
    rc = batch_create(n_sets, n, member, values, dev, &arena, &b);
    if (rc) return rc;
    std::unique_ptr<et_batch> own(b);                // host-side object only: its device arrays are not freed
    // the array keeps the previous call's contents: scores and status start from zero
    HIP_TRY_B(hipMemsetAsync(b->b.d_out, 0, (size_t)n_sets * n * 32, nullptr));
    HIP_TRY_B(hipMemsetAsync(b->b.d_status, 0, (size_t)n_sets * 4, nullptr));
    rc = et_batch_run_fr_async(b, num_iterations, initial_lo, initial_hi, nullptr);
    if (!rc) rc = et_batch_result_fr(b, out, status, nullptr);   // copies back and waits for the stream
    return rc;
}

int et_batch_converge_fr(uint32_t n_sets, uint32_t n, const uint8_t* member, const uint32_t* values, uint32_t num_iterations,
                         uint64_t initial_lo, uint64_t initial_hi, et_fr* out, int32_t* status) {
    et_batch* b = nullptr;
    int rc = et_batch_create(n_sets, n, member, values, -1, &b);
    if (rc) return rc;
    rc = et_batch_run_fr_async(b, num_iterations, initial_lo, initial_hi, nullptr);
    if (!rc) rc = et_batch_result_fr(b, out, status, nullptr);
    et_batch_free(b);
    return rc;
}

}  // extern "C"
