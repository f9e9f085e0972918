// GPU engine: upload of a prepared graph, on-device normalisation, the per-iteration kernels.
//
// One launch per iteration (dynamic_sets/native.rs:319-329).  Each launch fuses, per destination
// row i:   new_s[i] = sum_j c_ji * s[j]                    (native.rs:320-326)
//                   + (D - dangling_i * s[i]) / (n_valid-1)  (the rows of ones of native.rs:263-278)
//          [f64 only] new_s[i] = (1-alpha) * new_s[i] + alpha * p[i]
// and the reductions the next launch / the caller need: the score mass D' of dangling peers, the
// sum of scores (conservation check, native.rs:331-334), and the residual (f64: L1 distance,
// Fr: number of changed rows).  Reductions are per-workgroup partial results combined in a fixed
// order.  The exact variant produces identical bits on every run and every GPU count; the f64
// variant accumulates a row with LDS atomics whose order is not fixed, so its last bits may vary.
#include <hip/hip_runtime.h>
#include <array>
#include <map>
#include <mutex>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "builder.hpp"
#include "engine.hpp"
#include "fr_dev.hpp"

namespace et {

thread_local std::string g_last_error;

void set_error(const std::string& s) { g_last_error = s; }
const char* last_error() { return g_last_error.c_str(); }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                     \
            return _e == hipErrorNoDevice ? ET_E_NO_DEVICE : ET_E_HIP;                        \
        }                                                                                     \
    } while (0)

static const uint32_t STATS_RING = 64;

// ---------------------------------------------------------------------------------------------
// Two forms of the same iteration.
//
// "Raw" form, used whenever the opinions are integers (the production case: an attestation value
// is a u8, eigentrust/src/attestation.rs:304): the normalised weight c_ji = v_ji * inv(sum_j)
// (native.rs:305-314) is never stored.  The vector exchanged between iterations holds
// xs_j = s_j * inv(sum_j), an edge record holds the raw opinion, and
//     new_s_i = sum_j v_ji * xs_j
// which equals sum_j c_ji * s_j exactly in the field (multiplication is associative and
// distributive) and up to rounding in f64.  An edge then costs 5 (Fr) or 7 (f64) streamed bytes
// instead of 36 or 14.  The plain scores of a share's own rows stay in a private vector (`own`);
// the step flagged ET_STEP_FINAL writes plain scores to the exchanged vector as well.
//
// "Weighted" form, for opinions that are arbitrary field elements: weights are normalised once
// per graph (k_normalise_*) and the exchanged vector holds the plain scores.
// ---------------------------------------------------------------------------------------------
__global__ void k_normalise_f64_fr(const uint32_t* __restrict__ col, const uint4* __restrict__ val,
                                   const double* __restrict__ scale, double* __restrict__ w, size_t slots, Geometry g) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    U256 v = u256_load(val + 2 * i);
    double d = 0.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) d = d * 4294967296.0 + (double)v.v[k];
    w[i] = d * scale[g.pos_of_col(col[i])];
}
__global__ void k_normalise_fr_fr(const uint32_t* __restrict__ col, const uint4* __restrict__ val,
                                  const uint4* __restrict__ scale, uint4* __restrict__ wlo,
                                  uint4* __restrict__ whi, size_t slots, Geometry g) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    U256 k = u256_load(scale + 2 * (size_t)g.pos_of_col(col[i]));
    U256 v = u256_load(val + 2 * i);
    U256 w = fr_mul_redc288(k, v);  // v * inv(sum) * 2^576 * 2^-288 = weight * 2^288
    wlo[i] = make_uint4(w.v[0], w.v[1], w.v[2], w.v[3]);
    whi[i] = make_uint4(w.v[4], w.v[5], w.v[6], w.v[7]);
}

// ---------------------------------------------------------------------------------------------
// kernels: initial vector (native.rs:184-185, :317).  Writes this segment only.
// `scale` (raw form) or null: the exchanged entry is initial * scale[row]; `own` (raw form) or null
// receives the plain score.
// ---------------------------------------------------------------------------------------------
ET_DEV bool row_is_member(const Geometry& g, uint32_t local) {
    uint64_t q = ((uint64_t)(local >> 6) * g.n_seg + g.seg) * 64 + (local & 63);
    return q < g.n_active || (q >= g.n_active_pad && q < (uint64_t)g.n_active_pad + g.n_dangling);
}
ET_DEV bool row_is_dangling(const Geometry& g, uint32_t local) {
    uint64_t q = ((uint64_t)(local >> 6) * g.n_seg + g.seg) * 64 + (local & 63);
    return q >= g.n_active_pad && q < (uint64_t)g.n_active_pad + g.n_dangling;
}
__global__ void k_init_f64(Geometry g, double initial, double d0, double* __restrict__ x,
                           double* __restrict__ own, const double* __restrict__ scale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = g.seg_stride();
    if (i >= stride) return;
    const size_t p = (size_t)g.own_off + i;
    double v = 0.0;
    if (i < g.rows_cap) {
        v = row_is_member(g, i) ? initial : 0.0;
        if (own) own[i] = v;
        if (scale) v *= scale[i];
    } else if (i == g.rows_cap) v = d0;
    x[p] = v;
}
__global__ void k_init_fr(Geometry g, U256 initial, U256 d0, uint4* __restrict__ x,
                          uint4* __restrict__ own, const uint4* __restrict__ scale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = g.seg_stride();
    if (i >= stride) return;
    const size_t p = (size_t)g.own_off + i;
    U256 v = u256_zero();
    if (i < g.rows_cap) {
        if (row_is_member(g, i)) v = initial;
        if (own) u256_store(own + 2 * (size_t)i, v);
        if (scale) v = fr_mul_redc288(v, u256_load(scale + 2 * (size_t)i));
    } else if (i == g.rows_cap) v = d0;
    u256_store(x + 2 * p, v);
}

// ---------------------------------------------------------------------------------------------
// kernels: one iteration
// ---------------------------------------------------------------------------------------------
static const int WPB_FR = 8;    // wavefronts per workgroup of the Fr kernels

ET_DEV double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// f64 variant, "blocked sweep".  One 1024-thread workgroup per CU owns up to geo.f64_rb rows (row
// groups handed out so that every workgroup gets the same number of edges) with their accumulators in
// LDS, and walks its edges in ascending SOURCE order: the edge records stream in coalesced, each
// edge costs one 8-byte gather and one LDS atomic add.  Because every workgroup sweeps the source
// range at the same pace, the part of the trust vector being gathered at any moment is a narrow
// window that is shared through the 4 MiB per-XCD L2 although the whole vector does not fit, and
// the many consecutive edges of a popular source read one address.
// The LDS adds of one row arrive in no fixed order, so the last bits of an f64 score can differ
// from run to run; the Fr variant is exact.
// VT = uint8_t / uint32_t: raw form (per edge: source, accumulator slot, opinion);
// VT = double: weighted form.
// PACKED (byte opinions): 5 bytes per edge.  A workgroup's list is sorted by source, so the 64
// consecutive edges one wavefront loads together span a narrow range: the record holds the
// accumulator slot (its low rec_shift bits: as many as the workgroup's slots need) and the source
// position minus the position of the chunk's first source (the bits above), and one base position
// per 64 edges sits in a side array.
// ---------------------------------------------------------------------------------------------
struct StepArgsF64B {
    const uint32_t* rec;      // PACKED: [edge] slot | (source - chunk base) << rec_shift
    uint32_t rec_shift;
    const uint32_t* cbase;    // PACKED: base position per chunk of 64 edges, workgroup b's chunks start at cptr[b]
    const uint64_t* cptr;
    const uint64_t* bk_ptr;
    const uint32_t* grp;      // [f64_blocks][f64_rb / BLOCK_GROUP] row group behind each run of accumulator slots, ~0 = none
    const uint32_t* col;
    const void* w;            // VT per edge
    const uint16_t* row;
    const double* x;
    double* y;
    double* own;              // raw form: plain scores of this segment's dangling rows (factor zero), updated in place
    const double* scale;      // raw form: 1 / row sum by position
    const double* pretrust;
    double* res_part;   // [f64_blocks]
    double* sum_part;
    Geometry geo;
    double inv_nm1;
    double alpha;
    uint32_t final_step;
    // chunked exchange: this launch is run `phase` of `n_phases` over the workgroup's list (phase_ptr[b][n_phases + 1]);
    // the rows' partial sums travel between the launches of one iteration in acc_g[b][f64_rb].  n_phases == 1: one launch.
    const uint64_t* phase_ptr;
    double* acc_g;
    uint32_t phase, n_phases;
    uint32_t bar_period;      // batches between two barriers of the sweep
    uint32_t bar_first_256;   // the first one after this many 256ths of the workgroup's batches
    // A whole converge on one GPU (run_f64): the first step reads a vector that is kept from converge to converge, so
    // the plain scores of the dangling rows (`own`) have not been reset: their old score is `initial`.  The last step
    // writes the scores where the caller reads them, out[peer[row]] (slot order), in place of its segment of y.
    uint32_t first_step;
    double initial;
    double* out;
    const uint32_t* peer;     // [rows_cap] slot of the peer at a position, ~0 = none
    // residual and score sum of this iteration wanted (a converge asks on its last step only): without them a row with
    // opinions needs neither its old entry nor the division that turns it back into a score
    uint32_t stats;
};

// PHASED: one of several launches of an iteration (chunked exchange); otherwise the launch is the whole iteration and
// none of the phase bookkeeping is compiled in.
// EDGE: the first or the last launch of a whole converge on one GPU (first_step / out of the arguments); the launches in
// between compile neither.
template <typename VT, bool HAS_ALPHA, bool PACKED = false, bool PHASED = false, bool EDGE = false>
__global__ __launch_bounds__(1024) void k_step_f64_blocked(StepArgsF64B a) {
    static_assert(!(PACKED && PHASED), "runs of a chunked list do not start on a 64-edge boundary");
    static_assert(!(EDGE && PHASED), "a whole converge runs one launch per iteration");
    const bool first_step = EDGE && a.first_step != 0;
    double* const out = EDGE ? a.out : nullptr;
    constexpr bool RAW = sizeof(VT) != sizeof(double);
    static_assert(!PACKED || RAW, "packed records carry integer opinions (one byte or four)");
    extern __shared__ double acc_l[];
    __shared__ double red[3][16];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t RB = a.geo.f64_rb;
    const uint32_t stride = a.geo.seg_stride();
    const size_t my_base = a.geo.own_off;
    const double* __restrict__ xr = a.x;
    const VT* __restrict__ wr = static_cast<const VT*>(a.w);
    const bool first_phase = !PHASED || a.phase == 0, last_phase = !PHASED || a.phase + 1 == a.n_phases;
    for (uint32_t i = threadIdx.x; i < RB; i += 1024) acc_l[i] = first_phase ? 0.0 : a.acc_g[(size_t)b * RB + i];
    // the step owns the whole partial-sum tail of its segment of y: slots no workgroup writes are set to zero, so the
    // next step's sum over all nd_cap slots never reads what the caller's allocation happened to hold
    if (b == 0 && last_phase)
        for (uint32_t k = a.geo.f64_dblocks + threadIdx.x; k < a.geo.nd_cap; k += 1024) a.y[my_base + a.geo.rows_cap + k] = 0.0;
    // score mass of dangling peers: fixed-order sum of every segment's partial sums.  Only the rows' epilogue needs
    // it: the first four entries per lane are requested here and added after the sweep, so their latency is hidden
    const uint32_t d_total = a.geo.n_dangling != 0 && last_phase ? a.geo.n_seg * a.geo.nd_cap : 0u;
    double dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t t = lane + 64 * u;
        dv[u] = 0.0;
        if (t < d_total) {
            const uint32_t sg = t / a.geo.nd_cap, k = t - sg * a.geo.nd_cap;
            dv[u] = xr[(size_t)a.geo.dtail[sg] + k];
        }
    }
    // The rows' epilogue needs, per row, its group entry and then the old score and the factor: two dependent trips to
    // memory with nothing else left to do.  The first four rows of every thread (all of them when the workgroup holds
    // at most 4096) are asked for here, before the sweep, and wait in registers.
    uint32_t pgrp[4], ppeer[4];
    double pold[4], psc[4], ppt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t lr = threadIdx.x + u * 1024;
        pgrp[u] = lr < RB && last_phase ? a.grp[(size_t)b * (RB / BLOCK_GROUP) + lr / BLOCK_GROUP] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t lr = threadIdx.x + u * 1024;
        const uint32_t row = pgrp[u] * BLOCK_GROUP + lr % BLOCK_GROUP;
        pold[u] = psc[u] = ppt[u] = 0.0;
        ppeer[u] = 0xFFFFFFFFu;
        if (pgrp[u] != 0xFFFFFFFFu) {                      // rows beyond the members read a valid address and are dropped later
            // raw form: a peer with opinions keeps no plain score between iterations (its old one is the exchanged entry
            // divided by the factor); a peer without has factor zero and keeps its plain score in `own`
            if (RAW && row_is_dangling(a.geo, row)) pold[u] = first_step ? a.initial : a.own[row];
            else if (!RAW || a.stats) pold[u] = xr[my_base + row];
            if (RAW) psc[u] = a.scale[row];
            if (HAS_ALPHA) ppt[u] = a.pretrust[my_base + row];
            if (out) ppeer[u] = a.peer[row];
        }
    }
    __syncthreads();

    const uint64_t e0 = PHASED ? a.phase_ptr[(size_t)b * (a.n_phases + 1) + a.phase] : a.bk_ptr[b];
    const uint64_t e1 = PHASED ? a.phase_ptr[(size_t)b * (a.n_phases + 1) + a.phase + 1] : a.bk_ptr[b + 1];
#ifndef ET_F64_U
#define ET_F64_U 4
#endif
    constexpr int U = ET_F64_U;
    // Full batches of U * 1024 edges, the same number for every thread.  The hardware favours the
    // oldest wavefront of a workgroup, which lets it run far ahead of the youngest; a barrier every
    // second batch keeps all sixteen within one window of the source range.
    const uint32_t n_batches = (uint32_t)((e1 - e0) / (U * 1024));
    uint64_t e = e0 + threadIdx.x;
    int since_bar = (int)a.bar_period - (int)max(a.bar_period, (n_batches * a.bar_first_256) >> 8);
    if (PACKED) {
        const uint32_t* __restrict__ cb = a.cbase + a.cptr[b];
        const uint32_t rec_shift = a.rec_shift, rec_mask = (1u << rec_shift) - 1u;
        uint32_t chunk = wave;                             // (e - e0) / 64
        for (uint32_t k = 0; k < n_batches; ++k, e += U * 1024, chunk += U * 16) {
            uint32_t rc[U], base[U];
            double w[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { rc[u] = a.rec[e + u * 1024]; w[u] = (double)wr[e + u * 1024]; base[u] = cb[chunk + u * 16]; }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = xr[(size_t)base[u] + (rc[u] >> rec_shift)];
#pragma unroll
            for (int u = 0; u < U; ++u) atomicAdd(&acc_l[rc[u] & rec_mask], w[u] * xv[u]);
            if (++since_bar == (int)a.bar_period) { since_bar = 0; __syncthreads(); }
        }
        {   // the rest of the list (fewer than U * 1024 edges) as one more batch, its loads in flight together like the
            // others': taken 1024 at a time it was up to three more trips to memory, one after the other
            uint32_t rc[U], base[U];
            double w[U], xv[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                in[u] = e + u * 1024 < e1;
                rc[u] = in[u] ? a.rec[e + u * 1024] : 0u;
                w[u] = in[u] ? (double)wr[e + u * 1024] : 0.0;
                base[u] = in[u] ? cb[chunk + u * 16] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = in[u] ? xr[(size_t)base[u] + (rc[u] >> rec_shift)] : 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (in[u]) atomicAdd(&acc_l[rc[u] & rec_mask], w[u] * xv[u]);
        }
    } else {
        for (uint32_t k = 0; k < n_batches; ++k, e += U * 1024) {
            uint32_t c[U];
            double w[U], xv[U];
            uint32_t r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { c[u] = a.col[e + u * 1024]; w[u] = (double)wr[e + u * 1024]; r[u] = a.row[e + u * 1024]; }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = xr[(size_t)(c[u] >> COL_SEG_SHIFT) * stride + (c[u] & COL_LOCAL_MASK)];
#pragma unroll
            for (int u = 0; u < U; ++u) atomicAdd(&acc_l[r[u]], w[u] * xv[u]);
            if (++since_bar == (int)a.bar_period) { since_bar = 0; __syncthreads(); }
        }
        {   // the rest of the list as one more batch (see the packed form above)
            uint32_t c[U], r[U];
            double w[U], xv[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                in[u] = e + u * 1024 < e1;
                c[u] = in[u] ? a.col[e + u * 1024] : 0u;
                w[u] = in[u] ? (double)wr[e + u * 1024] : 0.0;
                r[u] = in[u] ? (uint32_t)a.row[e + u * 1024] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = in[u] ? xr[(size_t)(c[u] >> COL_SEG_SHIFT) * stride + (c[u] & COL_LOCAL_MASK)] : 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (in[u]) atomicAdd(&acc_l[r[u]], w[u] * xv[u]);
        }
    }
    __syncthreads();
    if (!last_phase) {                                     // the partial sums wait for the next launch of this iteration
        for (uint32_t i = threadIdx.x; i < RB; i += 1024) a.acc_g[(size_t)b * RB + i] = acc_l[i];
        return;
    }

    double D = (dv[0] + dv[1]) + (dv[2] + dv[3]);
    for (uint32_t t = lane + 256; t < d_total; t += 64) {
        const uint32_t sg = t / a.geo.nd_cap, k = t - sg * a.geo.nd_cap;
        D += xr[(size_t)a.geo.dtail[sg] + k];
    }
    D = wave_sum_f64(D);
    const double dterm = D * a.inv_nm1;
    double res = 0.0, tot = 0.0, dacc = 0.0;
    // four rows per thread at a time: their group entries, then their old scores and factors are requested together
    for (uint32_t lr0 = threadIdx.x; lr0 < RB; lr0 += 4 * 1024) {
        uint32_t grp[4], row[4], peer[4];
        bool live[4], dang[4];
        double old[4], sc[4], pt[4];
        const bool first = lr0 == threadIdx.x;             // this pass was asked for before the sweep
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t lr = lr0 + u * 1024;
            grp[u] = first ? pgrp[u] : (lr < RB ? a.grp[(size_t)b * (RB / BLOCK_GROUP) + lr / BLOCK_GROUP] : 0xFFFFFFFFu);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t lr = lr0 + u * 1024;
            row[u] = grp[u] * BLOCK_GROUP + lr % BLOCK_GROUP;
            const uint64_t q = ((uint64_t)(row[u] >> 6) * a.geo.n_seg + a.geo.seg) * 64 + (row[u] & 63);
            const bool active = q < a.geo.n_active;
            dang[u] = q >= a.geo.n_active_pad && q < (uint64_t)a.geo.n_active_pad + a.geo.n_dangling;
            live[u] = grp[u] != 0xFFFFFFFFu && (active || dang[u]);
            old[u] = sc[u] = pt[u] = 0.0;
            peer[u] = 0xFFFFFFFFu;
            if (first) {
                if (live[u]) { old[u] = pold[u]; sc[u] = psc[u]; pt[u] = ppt[u]; peer[u] = ppeer[u]; }
            } else if (live[u]) {
                if (RAW && dang[u]) old[u] = first_step ? a.initial : a.own[row[u]];
                else if (!RAW || a.stats) old[u] = xr[my_base + row[u]];
                if (RAW) sc[u] = a.scale[row[u]];
                if (HAS_ALPHA) pt[u] = a.pretrust[my_base + row[u]];
                if (out) peer[u] = a.peer[row[u]];
            }
            if (RAW && a.stats && live[u] && !dang[u]) old[u] = old[u] / sc[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!live[u]) continue;
            const double s = acc_l[lr0 + u * 1024];
            double v = dang[u] ? s + (D - old[u]) * a.inv_nm1 : s + dterm;
            if (HAS_ALPHA) v = (1.0 - a.alpha) * v + a.alpha * pt[u];
            if (RAW && dang[u]) a.own[row[u]] = v;
            if (out) {
                if (peer[u] != 0xFFFFFFFFu) out[peer[u]] = v;
            } else {
                a.y[my_base + row[u]] = RAW && !a.final_step ? v * sc[u] : v;
            }
            if (a.stats) { res += fabs(v - old[u]); tot += v; }
            if (dang[u]) dacc += v;
        }
    }
    // without statistics only the workgroups that hold dangling rows have anything to add up
    if (!a.stats && b >= a.geo.f64_dblocks) {
        if (threadIdx.x == 0) { a.res_part[b] = 0.0; a.sum_part[b] = 0.0; }
        return;
    }
    if (a.stats) { res = wave_sum_f64(res); tot = wave_sum_f64(tot); }
    dacc = wave_sum_f64(dacc);
    if (lane == 0) { red[0][wave] = res; red[1][wave] = tot; red[2][wave] = dacc; }
    __syncthreads();
    if (wave == 0) {                                       // sixteen partial results, one per lane
        double r = lane < 16 ? red[0][lane] : 0.0, t = lane < 16 ? red[1][lane] : 0.0, d = lane < 16 ? red[2][lane] : 0.0;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            r += __shfl_xor(r, off, 64);
            t += __shfl_xor(t, off, 64);
            d += __shfl_xor(d, off, 64);
        }
        if (lane == 0) {
            a.res_part[b] = r;
            a.sum_part[b] = t;
            if (b < a.geo.f64_dblocks) a.y[my_base + a.geo.rows_cap + b] = d;   // partial sum slot b of this segment
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fr variant: one lane per row over the 64-row slices (row-per-wavefront for long rows).
// ---------------------------------------------------------------------------------------------
struct StepArgsFr {
    const WorkDesc* work;
    const SliceDesc* slices;
    const uint32_t* col;
    const uint4* wlo;      // weighted form: weight * 2^288, two planes
    const uint4* whi;
    const void* val;       // raw form: opinion per slot (VT)
    const uint4* x;
    uint4* y;
    uint4* own;            // raw form: plain scores of this segment's rows, updated in place
    const uint4* scale;    // raw form: inv(row sum) * 2^576 by position
    uint4* sum_part;       // [n_work][2]
    uint32_t* chg_part;    // [n_work]
    Geometry geo;
    uint32_t n_work;
    uint32_t n_dslots;     // partial-sum slots of this segment's tail that the launch writes (dangling work items)
    uint32_t final_step;
    U256 inv_nm1_w;        // inv(n_valid-1) * 2^288 mod r
};

// A step owns the whole partial-sum tail of its segment of y: the slots no workgroup of this launch writes are
// set to zero by the first workgroup, so that the next step's sum over all nd_cap slots never reads what a
// caller's allocation happened to hold.
ET_DEV void fr_zero_unused_tail(uint4* __restrict__ y, const Geometry& geo, uint32_t written, uint32_t threads) {
    if (blockIdx.x != 0) return;
    const size_t tail = (size_t)geo.own_off + geo.rows_cap;
    for (uint32_t k = written + threadIdx.x; k < geo.nd_cap; k += threads) u256_store(y + 2 * (tail + k), u256_zero());
}

ET_DEV U256 fr_wave_sum(U256 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        U256 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = __shfl_xor(v.v[i], off, 64);
        v = fr_add(v, o);
    }
    return v;
}

// Score mass D of the dangling peers (fixed-order integer sum of every segment's partial sums) as
// the two values the rows need: sh_d = D + r (9 limbs), sh_pd = inv_nm1_w * D (17 limbs,
// unreduced).  Called by the first wavefront of a workgroup; followed by __syncthreads().
ET_DEV void fr_dangling_mass(const uint4* __restrict__ x, const Geometry& geo, const U256& inv_nm1_w, int lane,
                             uint32_t* sh_d, uint32_t* sh_pd) {
    if (geo.n_dangling == 0) {           // no peer without opinions anywhere: D = 0, nothing to read
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sh_d[i] = i < 8 ? FR_P[i] : 0u;
#pragma unroll
            for (int i = 0; i < 17; ++i) sh_pd[i] = 0u;
        }
        return;
    }
    Acc d = acc_zero();
    const uint32_t total = geo.n_seg * geo.nd_cap;
    for (uint32_t t0 = lane; t0 < total; t0 += 256) {     // four entries per lane in flight
        U256 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t t = t0 + 64 * u;
            p[u] = u256_zero();
            if (t < total) {
                const uint32_t sg = t / geo.nd_cap, k = t - sg * geo.nd_cap;
                p[u] = u256_load(x + 2 * ((size_t)geo.dtail[sg] + k));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc_add_u256(d, p[u]);
    }
    acc_wave_sum(d);  // < 2^(254+log2(total)) : limbs 0..8
    Acc pd = acc_zero();
    U256 dlo;
#pragma unroll
    for (int i = 0; i < 8; ++i) dlo.v[i] = d.v[i];
    acc_mac(pd, inv_nm1_w, dlo);
    // + inv_nm1_w * d.v[8] * 2^256
    {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t t = (uint64_t)inv_nm1_w.v[j] * d.v[8] + pd.v[8 + j] + c;
            pd.v[8 + j] = (uint32_t)t;
            c = t >> 32;
        }
        pd.v[16] += (uint32_t)c;
    }
    if (lane == 0) {
        // D + r: the partial sums are reduced mod r, so D may be smaller than one of its own
        // summands as an integer; (D + r - old) is the non-negative representative a dangling
        // row needs
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint64_t t = (uint64_t)d.v[i] + (i < 8 ? FR_P[i] : 0u) + c;
            sh_d[i] = (uint32_t)t;
            c = t >> 32;
        }
#pragma unroll
        for (int i = 0; i < 17; ++i) sh_pd[i] = pd.v[i];
    }
}
// acc += inv_nm1_w * (D + r - old): the share a dangling row receives from the other dangling peers
ET_DEV void fr_add_dangling_term(Acc& acc, const uint32_t* sh_d, const U256& old, const U256& inv_nm1_w) {
    // (D + r - old) as a 9-limb integer: old < r, so no borrow out
    uint32_t diff[9];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint64_t t = (uint64_t)sh_d[i] - (i < 8 ? old.v[i] : 0u) - br;
        diff[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    U256 dlo;
#pragma unroll
    for (int i = 0; i < 8; ++i) dlo.v[i] = diff[i];
    acc_mac(acc, inv_nm1_w, dlo);
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t t = (uint64_t)inv_nm1_w.v[j] * diff[8] + acc.v[8 + j] + c;
        acc.v[8 + j] = (uint32_t)t;
        c = t >> 32;
    }
    acc.v[16] += (uint32_t)c;
}

// weighted form
__global__ __launch_bounds__(WPB_FR * 64) void k_step_fr(StepArgsFr a) {
    __shared__ uint32_t sh_d[9];
    __shared__ uint32_t sh_pd[17];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t stride = a.geo.seg_stride();
    const size_t my_base = a.geo.own_off;
    fr_zero_unused_tail(a.y, a.geo, a.n_dslots, WPB_FR * 64);
    // The first wavefront prepares the dangling mass while the others already walk their first slice.  Every
    // wavefront of the workgroup stays alive (one without a work item walks no slice) and all of them meet at
    // ONE barrier, after the first slice's edges and before its rows are finished.
    if (wave == 0) fr_dangling_mass(a.x, a.geo, a.inv_nm1_w, lane, sh_d, sh_pd);
    const uint32_t gw = blockIdx.x * WPB_FR + wave;
    const bool has_work = gw < a.n_work;
    const WorkDesc wd = has_work ? a.work[gw] : WorkDesc{0u, 0u, 0u, 0u};
    const bool dang = (wd.flags & WORK_DANGLING) != 0;
    U256 tot = u256_zero(), dacc = u256_zero();
    uint32_t changed = 0;
    bool met = false;
    for (uint32_t s = wd.slice_begin;; ++s) {
        const bool have = s < wd.slice_end;
        SliceDesc sd{0u, 0u, 0u, 0u};
        Acc acc = acc_zero();
        if (have) {
            sd = a.slices[s];
            const uint32_t* __restrict__ cp = a.col + (size_t)sd.off * 64 + lane;
            const uint4* __restrict__ lop = a.wlo + (size_t)sd.off * 64 + lane;
            const uint4* __restrict__ hip_ = a.whi + (size_t)sd.off * 64 + lane;
#pragma unroll 2
            for (uint32_t k = 0; k < sd.len; ++k) {
                const uint32_t c = cp[(size_t)k * 64];
                U256 w = u256_from_planes(lop[(size_t)k * 64], hip_[(size_t)k * 64]);
                U256 xv = u256_load(a.x + 2 * ((size_t)(c >> COL_SEG_SHIFT) * stride + (c & COL_LOCAL_MASK)));
                acc_mac(acc, w, xv);
            }
        }
        if (!met) { __syncthreads(); met = true; }     // first pass of every wavefront, with or without a slice
        if (!have) break;
        const uint32_t sh = slice_log2_lanes(sd.nrows);
        acc_group_sum(acc, 1u << sh);
        Acc PD;
#pragma unroll
        for (int i = 0; i < 17; ++i) PD.v[i] = sh_pd[i];
        const uint32_t row = sd.row0 + ((uint32_t)lane >> sh);
        const bool live = ((uint32_t)lane & ((1u << sh) - 1)) == 0 && ((uint32_t)lane >> sh) < slice_rows(sd.nrows);
        if (live) {
            U256 old = u256_load(a.x + 2 * (my_base + row));
            if (dang) fr_add_dangling_term(acc, sh_d, old, a.inv_nm1_w);
            else acc_add(acc, PD);
            U256 v = acc_redc288(acc);
            u256_store(a.y + 2 * (my_base + row), v);
            changed += u256_eq(v, old) ? 0u : 1u;
            tot = fr_add(tot, v);
            if (dang) dacc = fr_add(dacc, v);
        }
    }
    if (!has_work) return;                              // past the only barrier
    tot = fr_wave_sum(tot);
    if (dang) dacc = fr_wave_sum(dacc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) changed += __shfl_xor(changed, off, 64);
    if (lane == 0) {
        u256_store(a.sum_part + 2 * (size_t)gw, tot);
        a.chg_part[gw] = changed;
        if (dang) u256_store(a.y + 2 * (my_base + a.geo.rows_cap + wd.dslot), dacc);
    }
}

// raw form: per slot a source index and the opinion (VT = uint8_t or uint32_t); the gathered
// vector holds xs_j = s_j * inv(sum_j) * 2^288, the 320-bit sum of opinion * xs_j is reduced once.
template <typename VT>
__global__ __launch_bounds__(WPB_FR * 64) void k_step_fr_raw(StepArgsFr a) {
    __shared__ uint32_t sh_d[9];
    __shared__ uint32_t sh_pd[17];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t stride = a.geo.seg_stride();
    const size_t my_base = a.geo.own_off;
    fr_zero_unused_tail(a.y, a.geo, a.n_dslots, WPB_FR * 64);
    // The first wavefront prepares the dangling mass while the others already walk their first slice.  Every
    // wavefront of the workgroup stays alive (one without a work item walks no slice) and all of them meet at
    // ONE barrier, after the first slice's edges and before its rows are finished.
    if (wave == 0) fr_dangling_mass(a.x, a.geo, a.inv_nm1_w, lane, sh_d, sh_pd);
    const uint32_t gw = blockIdx.x * WPB_FR + wave;
    const bool has_work = gw < a.n_work;
    const WorkDesc wd = has_work ? a.work[gw] : WorkDesc{0u, 0u, 0u, 0u};
    const bool dang = (wd.flags & WORK_DANGLING) != 0;

    U256 tot = u256_zero(), dacc = u256_zero();
    uint32_t changed = 0;
    bool met = false;
    for (uint32_t s = wd.slice_begin;; ++s) {
        const bool have = s < wd.slice_end;
        SliceDesc sd{0u, 0u, 0u, 0u};
        Acc10 acc = acc10_zero();
        if (have) {
            sd = a.slices[s];
            const uint32_t* __restrict__ cp = a.col + (size_t)sd.off * 64 + lane;
            const VT* __restrict__ vp = static_cast<const VT*>(a.val) + (size_t)sd.off * 64 + lane;
            // Batches of U steps with all U gathers in flight together; the last batch may be partial:
            // its surplus steps re-read the slice's last step with the opinion forced to zero.
            constexpr int U = 4;
            const uint32_t len = sd.len;
            if (len > 0) {
                const uint32_t last = len - 1;
                // the records of the next batch are asked for before this batch's gathers, so the two waits overlap
                uint32_t c[U], v[U];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const size_t o = (size_t)min((uint32_t)i, last) * 64;
                    c[i] = cp[o];
                    v[i] = (uint32_t)i < len ? (uint32_t)vp[o] : 0u;
                }
                for (uint32_t k = 0; k < len; k += U) {
                    uint32_t cn[U], vn[U];
#pragma unroll
                    for (int i = 0; i < U; ++i) {
                        const size_t o = (size_t)min(k + U + i, last) * 64;
                        cn[i] = cp[o];
                        vn[i] = k + U + i < len ? (uint32_t)vp[o] : 0u;
                    }
                    U256 xv[U];
#pragma unroll
                    for (int i = 0; i < U; ++i) xv[i] = u256_load(a.x + 2 * ((size_t)(c[i] >> COL_SEG_SHIFT) * stride + (c[i] & COL_LOCAL_MASK)));
#pragma unroll
                    for (int i = 0; i < U; ++i) acc10_mac_u32(acc, xv[i], v[i]);
#pragma unroll
                    for (int i = 0; i < U; ++i) { c[i] = cn[i]; v[i] = vn[i]; }
                }
            }
        }
        if (!met) { __syncthreads(); met = true; }     // first pass of every wavefront, with or without a slice
        if (!have) break;
        const uint32_t sh = slice_log2_lanes(sd.nrows);
        acc10_group_sum(acc, 1u << sh);
        Acc full = acc_from_acc10(acc);
        const uint32_t row = sd.row0 + ((uint32_t)lane >> sh);
        const bool live = ((uint32_t)lane & ((1u << sh) - 1)) == 0 && ((uint32_t)lane >> sh) < slice_rows(sd.nrows);
        if (live) {
            U256 old = u256_load(a.own + 2 * (size_t)row);
            if (dang) {
                fr_add_dangling_term(full, sh_d, old, a.inv_nm1_w);
            } else {
                Acc PD;
#pragma unroll
                for (int i = 0; i < 17; ++i) PD.v[i] = sh_pd[i];
                acc_add(full, PD);
            }
            U256 v = acc_redc288(full);
            u256_store(a.own + 2 * (size_t)row, v);
            // a dangling peer has no out-edges: its exchanged entry is never gathered
            U256 out = a.final_step ? v : (dang ? u256_zero() : fr_mul_redc288(v, u256_load(a.scale + 2 * (size_t)row)));
            u256_store(a.y + 2 * (my_base + row), out);
            changed += u256_eq(v, old) ? 0u : 1u;
            tot = fr_add(tot, v);
            if (dang) dacc = fr_add(dacc, v);
        }
    }
    if (!has_work) return;                              // past the only barrier
    tot = fr_wave_sum(tot);
    if (dang) dacc = fr_wave_sum(dacc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) changed += __shfl_xor(changed, off, 64);
    if (lane == 0) {
        u256_store(a.sum_part + 2 * (size_t)gw, tot);
        a.chg_part[gw] = changed;
        if (dang) u256_store(a.y + 2 * (my_base + a.geo.rows_cap + wd.dslot), dacc);
    }
}

// ---------------------------------------------------------------------------------------------
// Exact variant as a blocked sweep (byte opinions, one share, 512K to 2M rows).  The workgroup keeps eight
// 64-bit column sums per row in LDS (limb of xs times opinion byte, added with 64-bit LDS atomics: exact
// integers, any order), walks its edges in ascending source order like the f64 sweep -- so the gathered window of
// the vector is shared through the L2 and every gather is an L2 hit -- and then folds, reduces and scales its
// rows.  Uniform-source microbenchmark: 143 us against 271 us for the row-per-lane access pattern.
//
// Row arithmetic.  A row's folded sum T = new score * 2^288 (plus the dangling term, reduced once per workgroup)
// stays below 2^288, so the exchanged entry new score * inv(sum) * 2^288 is ONE reduction of the 9 x 8-limb
// product T * (inv(sum) * 2^288): 144 instead of 208 limb multiply-adds per row.  The plain score is only formed
// on the final step; the score sum is the reduction of the integer sum of the T (reduction is linear), and a score
// changed exactly when its exchanged entry did (the factor is invertible).  Rows of dangling peers keep their
// plain scores in `own` as the row-per-lane kernel does.
// ---------------------------------------------------------------------------------------------
struct StepArgsFrB {
    const uint64_t* ptr;      // [fr_blocks + 1]
    const uint32_t* grp;      // [fr_blocks][fr_rb / BLOCK_GROUP]
    const uint32_t* col;
    const uint16_t* row;
    const void* val;          // opinion per edge: a byte, or 32 bits (WIDE)
    // packed records (see k_step_f64_blocked): accumulator row | (source position - base of the 64-edge chunk) << rec_shift
    uint32_t rec_shift;
    const uint32_t* rec;
    const uint32_t* cbase;
    const uint64_t* cptr;
    const uint4* x;
    uint4* y;
    uint4* own;
    const uint4* scale288;    // per row: inv(sum) * 2^288, or inv(sum) * 2^320 (WIDE)
    uint4* sum_part;          // [fr_blocks][2]
    uint32_t* chg_part;       // [fr_blocks]
    Geometry geo;
    uint32_t final_step;
    // chunked exchange: run `phase` of `n_phases` over the workgroup's list (phase_ptr[b][n_phases + 1]); the rows'
    // unreduced sums travel between the launches of one iteration in acc_g[b][LDS image].  n_phases == 1: one launch.
    const uint64_t* phase_ptr;
    unsigned long long* acc_g;
    uint32_t phase, n_phases;
    U256 inv_nm1_w;
    U256 inv_nm1_w576;        // inv(n_valid - 1) * 2^576
    uint32_t bar_period;      // passes between two barriers of the sweep
    uint32_t bar_first_256;   // the first one after this many 256ths of the workgroup's passes
    // statistics of this iteration wanted: the rows whose entry changed (one more 32-byte load per row) and the score sum
    // (ten-limb additions per row and the reductions at the end); a converge asks for them on its last step only
    uint32_t count_changes;
    // a whole converge on one GPU (run_fr), as for the f64 sweep: dangling rows' old score on the first step, and the
    // last step's scores written in slot order
    uint32_t first_step;
    U256 initial;
    uint4* out;
    const uint32_t* peer;
};

__global__ void k_scale_288(const uint4* __restrict__ scale576, uint4* __restrict__ scale288, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const U256 s = u256_load(scale576 + 2 * i);
    Acc t = acc_zero();
#pragma unroll
    for (int k = 0; k < 8; ++k) t.v[k] = s.v[k];
    u256_store(scale288 + 2 * i, acc_redc288(t));
}

// inv(sum) * 2^320 from inv(sum) * 2^576: what the rows of 32-bit opinions are multiplied by
__global__ void k_scale_320(const uint4* __restrict__ scale576, uint4* __restrict__ scale320, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    U256 two32 = u256_zero();
    two32.v[1] = 1u;
    u256_store(scale320 + 2 * i, fr_mul_redc288(u256_load(scale576 + 2 * i), two32));
}

// COMPACT: a row's sum as nine 32-bit limbs, stored as four 64-bit words [4][RB] followed by the top limbs [RB]
// (36 bytes per row, so one workgroup per CU covers 1M rows in ONE round and the vector crosses every L2 once).
// An edge adds its 264-bit product word by word with LDS atomics that return the old word: the add that wraps a word
// is the one that sees it, and sends the carry into the next word (a carry that wraps a word of all ones travels
// on).  Each wrap is compensated exactly once, every addend is non-negative and the final sum is below 2^288, so once
// all adds have landed the limbs hold the exact integer whatever the order.
// !COMPACT: eight 64-bit column sums per row [8][RB] (limb times opinion byte, no carries until the row is folded).
// PHASED: as for the f64 sweep.
// PACKED: 5-byte edge records (one 32-bit record and the opinion byte) in place of 7 bytes (source, row, opinion).
// WIDE: 32-bit opinions (the reference's opinion is any field element, native.rs:128; integers above a byte take this
// form).  The product has 288 bits; a row with fewer than 2^17 in-edges (the planner checks) sums to less than 2^304, so a
// row carries sixteen more bits beside its nine limbs (38 bytes: the tenth limbs of two rows share a 32-bit word), the
// add that wraps the ninth limb sending its carry there.  The row's ten limbs are multiplied by inv(sum) * 2^320 and
// reduced in one pass of ten rounds (160 limb multiply-adds per row).
// EDGE: as for the f64 sweep.
template <bool COMPACT, bool PHASED = false, bool PACKED = false, bool WIDE = false, bool EDGE = false>
__global__ __launch_bounds__(1024) void k_step_fr_blocked(StepArgsFrB a) {
    static_assert(!(PACKED && PHASED), "runs of a chunked list do not start on a 64-edge boundary");
    static_assert(!(EDGE && PHASED), "a whole converge runs one launch per iteration");
    const bool first_step = EDGE && a.first_step != 0;
    uint4* const out = EDGE ? a.out : nullptr;
    static_assert(!WIDE || (COMPACT && !PHASED), "32-bit opinions: compact rows, one launch per iteration");
    typedef typename std::conditional<WIDE, uint32_t, uint8_t>::type VT;
    const VT* __restrict__ valp = static_cast<const VT*>(a.val);
    extern __shared__ unsigned long long colsum_l[];
    __shared__ uint32_t sh_d[9];
    __shared__ uint32_t sh_pd[17];
    __shared__ uint32_t sh_pdr[8];
    __shared__ uint4 red_l[2 * 16];
    __shared__ uint32_t red_c[16];
    __shared__ uint32_t red_raw[16][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x, RB = a.geo.fr_rb;
    const uint32_t stride = a.geo.seg_stride();
    const size_t my_base = a.geo.own_off;
    uint32_t* const top_l = reinterpret_cast<uint32_t*>(colsum_l + 4 * (size_t)RB);   // COMPACT: limb 8 of every row
    uint32_t* const ext_l = top_l + RB;                                               // WIDE: limb 9, sixteen bits per row
    const bool first_phase = !PHASED || a.phase == 0, last_phase = !PHASED || a.phase + 1 == a.n_phases;
    const uint32_t lds_words = WIDE ? 4 * RB + RB / 2 + RB / 4 : (COMPACT ? 4 * RB + RB / 2 : 8 * RB);   // RB is a multiple of 16
    for (uint32_t i = threadIdx.x; i < lds_words; i += 1024) colsum_l[i] = first_phase ? 0ull : a.acc_g[(size_t)b * lds_words + i];
    // partial-sum slots no workgroup writes: zero, whatever the caller's buffer held (see fr_zero_unused_tail)
    if (last_phase) fr_zero_unused_tail(a.y, a.geo, a.geo.fr_dblocks, 1024);
    // the rows' epilogue starts from their group entries: those of a thread's first four rows are asked for here and
    // wait in registers through the sweep, one trip to memory less after it
    uint32_t pgrp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t lr = threadIdx.x + u * 1024;
        pgrp[u] = lr < RB && last_phase ? a.grp[(size_t)b * (RB / BLOCK_GROUP) + lr / BLOCK_GROUP] : 0xFFFFFFFFu;
    }
    __syncthreads();
    // the first wavefront prepares the dangling term while the others already sweep; the workgroup next meets at
    // the sweep's own barriers, and the term is read after the barrier that ends the sweep
    if (wave == 0 && last_phase) {
        // the product with inv(n_valid - 1) * 2^576, reduced once: inv(n_valid - 1) * D * 2^288 below r
        fr_dangling_mass(a.x, a.geo, a.inv_nm1_w576, lane, sh_d, sh_pd);
        if (lane == 0) {
            Acc pd;
#pragma unroll
            for (int i = 0; i < 17; ++i) pd.v[i] = sh_pd[i];
            const U256 pr = acc_redc288(pd);
#pragma unroll
            for (int i = 0; i < 8; ++i) sh_pdr[i] = pr.v[i];
        }
    }

    const uint64_t e0 = PHASED ? a.phase_ptr[(size_t)b * (a.n_phases + 1) + a.phase] : a.ptr[b];
    const uint64_t e1 = PHASED ? a.phase_ptr[(size_t)b * (a.n_phases + 1) + a.phase + 1] : a.ptr[b + 1];
    // ONE edge per thread in flight and a barrier every eighth pass.  The two 16-byte halves of a gathered element share a
    // cache line: with one element per thread the second half's request meets the line the first half's miss is already
    // bringing; with more elements in flight per thread (16 wavefronts x 64 lanes x U lines against a 256-line L1) the line
    // is often gone again and fetched twice.  1M-peer benchmark graph: U = 1 / 2 / 3 / 4 at their best barrier periods
    // 112 / 119 / 119 / 121 us; U = 1 with a barrier every 4 / 8 / 16 / 32 passes 112.7 / 112.3 / 115.0 / 117.4, none 126.
    // Asking for the next element's halves before this element's adds (two in flight again) costs 6 us, non-temporal
    // record loads 17 us.
    constexpr int U = 1;
    const uint32_t n_batches = (uint32_t)((e1 - e0) / (U * 1024));
    uint64_t e = e0 + threadIdx.x;
    int since_bar = (int)a.bar_period - (int)max(a.bar_period, (n_batches * a.bar_first_256) >> 8);
    // a carry into word k of a row (k = 1..3), or into the top limb (k = 4)
    auto carry_into = [&](uint32_t slot, uint32_t k) {
        for (; k < 4; ++k)
            if (atomicAdd(colsum_l + (size_t)k * RB + slot, 1ull) != ~0ull) return;
        if (!WIDE) atomicAdd(top_l + slot, 1u);
        else if (atomicAdd(top_l + slot, 1u) == 0xFFFFFFFFu) atomicAdd(ext_l + (slot >> 1), 1u << (16 * (slot & 1)));
    };
    auto add_edge = [&](uint32_t slot, uint32_t v, const uint4& lo, const uint4& hi) {
        unsigned long long* p = colsum_l + slot;
        if (COMPACT) {
            // the 264-bit product opinion * xs as four 64-bit words and a top byte
            uint64_t t = (uint64_t)lo.x * v;
            uint64_t w0 = (uint32_t)t;
            t = (uint64_t)lo.y * v + (t >> 32); w0 |= t << 32;
            t = (uint64_t)lo.z * v + (t >> 32);
            uint64_t w1 = (uint32_t)t;
            t = (uint64_t)lo.w * v + (t >> 32); w1 |= t << 32;
            t = (uint64_t)hi.x * v + (t >> 32);
            uint64_t w2 = (uint32_t)t;
            t = (uint64_t)hi.y * v + (t >> 32); w2 |= t << 32;
            t = (uint64_t)hi.z * v + (t >> 32);
            uint64_t w3 = (uint32_t)t;
            t = (uint64_t)hi.w * v + (t >> 32); w3 |= t << 32;
            uint64_t w4 = t >> 32;                          // a byte opinion leaves 8 bits here, a 32-bit one 32
            const uint64_t o0 = atomicAdd(p + 0 * (size_t)RB, (unsigned long long)w0);
            const uint64_t o1 = atomicAdd(p + 1 * (size_t)RB, (unsigned long long)w1);
            const uint64_t o2 = atomicAdd(p + 2 * (size_t)RB, (unsigned long long)w2);
            const uint64_t o3 = atomicAdd(p + 3 * (size_t)RB, (unsigned long long)w3);
            // old + addend wrapped  <=>  old > ~addend
            const bool c1 = o0 > ~w0, c2 = o1 > ~w1, c3 = o2 > ~w2;
            w4 += o3 > ~w3 ? 1u : 0u;
            if (WIDE) {                                     // w4 <= 2^32: its low word into limb 8, what wraps into limb 9
                const uint32_t low = (uint32_t)w4;
                const uint32_t old = atomicAdd(top_l + slot, low);
                const uint32_t up = (uint32_t)(w4 >> 32) + (old > ~low ? 1u : 0u);
                if (up) atomicAdd(ext_l + (slot >> 1), up << (16 * (slot & 1)));
            } else if (w4) atomicAdd(top_l + slot, (uint32_t)w4);
            if (c1) carry_into(slot, 1);
            if (c2) carry_into(slot, 2);
            if (c3) carry_into(slot, 3);
            return;
        }
        atomicAdd(p + 0 * RB, (unsigned long long)lo.x * v);
        atomicAdd(p + 1 * RB, (unsigned long long)lo.y * v);
        atomicAdd(p + 2 * RB, (unsigned long long)lo.z * v);
        atomicAdd(p + 3 * RB, (unsigned long long)lo.w * v);
        atomicAdd(p + 4 * RB, (unsigned long long)hi.x * v);
        atomicAdd(p + 5 * RB, (unsigned long long)hi.y * v);
        atomicAdd(p + 6 * RB, (unsigned long long)hi.z * v);
        atomicAdd(p + 7 * RB, (unsigned long long)hi.w * v);
    };
    {
        // the records of batch k+1 are asked for before the gathers of batch k are used: a record comes from memory
        // and the gather's address from the record, so the two trips of a batch would otherwise follow each other
        // c: the source's position in the vector (PACKED) or its (segment, local row) code; r: the accumulator row
        uint32_t c[U], v[U], r[U];
        const uint32_t* __restrict__ cb = PACKED ? a.cbase + a.cptr[b] : nullptr;
        auto fetch = [&](uint64_t i, uint32_t& cc, uint32_t& vv, uint32_t& rr) {
            vv = valp[i];
            if (PACKED) {
                const uint32_t rc = a.rec[i];
                cc = cb[(i - e0) >> 6] + (rc >> a.rec_shift);
                rr = rc & ((1u << a.rec_shift) - 1u);
            } else {
                cc = a.col[i];
                rr = a.row[i];
            }
        };
        auto entry = [&](uint32_t cc) -> const uint4* {
            return a.x + 2 * (PACKED ? (size_t)cc : (size_t)(cc >> COL_SEG_SHIFT) * stride + (cc & COL_LOCAL_MASK));
        };
        if (n_batches) {
#pragma unroll
            for (int u = 0; u < U; ++u) fetch(e + u * 1024, c[u], v[u], r[u]);
        }
        for (uint32_t k = 0; k < n_batches; ++k, e += U * 1024) {
            uint4 lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint4* p = entry(c[u]);
                lo[u] = p[0];
                hi[u] = p[1];
            }
            uint32_t cn[U], vn[U], rn[U];
            const uint64_t en = k + 1 < n_batches ? e + U * 1024 : e;      // the last pass reads its own records again
#pragma unroll
            for (int u = 0; u < U; ++u) fetch(en + u * 1024, cn[u], vn[u], rn[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) add_edge(r[u], v[u], lo[u], hi[u]);
            if (++since_bar == (int)a.bar_period) { since_bar = 0; __syncthreads(); }
#pragma unroll
            for (int u = 0; u < U; ++u) { c[u] = cn[u]; v[u] = vn[u]; r[u] = rn[u]; }
        }
    }
    {   // the rest of the list as one more batch, its loads in flight together like the others'
        uint32_t c[U], v[U], r[U];
        uint4 lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = e + u * 1024;
            const bool in = i < e1;
            c[u] = v[u] = r[u] = 0u;                        // position / code 0 is a valid address
            if (in) {
                v[u] = valp[i];
                if (PACKED) {
                    const uint32_t rc = a.rec[i];
                    c[u] = a.cbase[a.cptr[b] + ((i - e0) >> 6)] + (rc >> a.rec_shift);
                    r[u] = rc & ((1u << a.rec_shift) - 1u);
                } else {
                    c[u] = a.col[i];
                    r[u] = a.row[i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4* p = a.x + 2 * (PACKED ? (size_t)c[u] : (size_t)(c[u] >> COL_SEG_SHIFT) * stride + (c[u] & COL_LOCAL_MASK));
            lo[u] = p[0];
            hi[u] = p[1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e + u * 1024 < e1) add_edge(r[u], v[u], lo[u], hi[u]);
    }
    __syncthreads();
    if (!last_phase) {                                    // the unreduced sums wait for the next launch of this iteration
        for (uint32_t i = threadIdx.x; i < lds_words; i += 1024) a.acc_g[(size_t)b * lds_words + i] = colsum_l[i];
        return;
    }

    U256 dacc = u256_zero();                              // field sum of this thread's dangling rows' scores
    Acc10 traw = acc10_zero();                            // integer sum of this thread's T
    uint32_t changed = 0;
    // eight 64-bit column sums, 32 bits apart, as ten limbs
    auto fold_row = [&](uint32_t lr, uint32_t* t) {
        if (COMPACT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long w = colsum_l[lr + (size_t)j * RB];
                t[2 * j] = (uint32_t)w;
                t[2 * j + 1] = (uint32_t)(w >> 32);
            }
            t[8] = top_l[lr];
            t[9] = WIDE ? (ext_l[lr >> 1] >> (16 * (lr & 1))) & 0xFFFFu : 0u;
            return;
        }
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long s = colsum_l[lr + (size_t)j * RB];
            c += s & 0xFFFFFFFFull;
            t[j] = (uint32_t)c;
            c = (c >> 32) + (s >> 32);
        }
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
    };
    // rows with opinions first (the common case, no dangling arithmetic in the loop), then the dangling rows, which
    // only the leading workgroups hold.  Two rows per thread at a time: their factors are requested together.
    constexpr int EP = 2;
    for (uint32_t k = 0; threadIdx.x + k * 1024 < RB; k += EP) {
        const uint32_t lr0 = threadIdx.x + k * 1024;
        uint32_t row[EP], peer[EP];
        bool act[EP];
        U256 factor[EP];
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            const uint32_t lr = lr0 + u * 1024;
            // group entries of the first four rows: asked for before the sweep
            uint32_t grp = lr < RB ? 0u : 0xFFFFFFFFu;
            if (k == 0) grp = pgrp[u];
            else if (k == 2) grp = pgrp[2 + u];
            else if (lr < RB) grp = a.grp[(size_t)b * (RB / BLOCK_GROUP) + lr / BLOCK_GROUP];
            row[u] = grp * BLOCK_GROUP + lr % BLOCK_GROUP;
            const uint64_t q = ((uint64_t)(row[u] >> 6) * a.geo.n_seg + a.geo.seg) * 64 + (row[u] & 63);
            act[u] = grp != 0xFFFFFFFFu && q < a.geo.n_active;
        }
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            factor[u] = u256_zero();
            peer[u] = 0xFFFFFFFFu;
            if (act[u]) {
                factor[u] = u256_load(a.scale288 + 2 * (size_t)row[u]);
                if (out) peer[u] = a.peer[row[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < EP; ++u) {
            if (!act[u]) continue;
            // T = sum of opinion * xs (below 2^24 * 2^8 * r) + dangling term (below r): limb 9 stays empty
            uint32_t t[10];
            fold_row(lr0 + u * 1024, t);
            {
                uint64_t c = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    c += (uint64_t)t[i] + sh_pdr[i];
                    t[i] = (uint32_t)c;
                    c >>= 32;
                }
                if (WIDE) { c += t[8]; t[8] = (uint32_t)c; t[9] += (uint32_t)(c >> 32); }
                else t[8] += (uint32_t)c;
            }
            if (a.count_changes) {       // statistics wanted: the integer sum of the T is the score sum before reduction
                uint64_t c = 0;
#pragma unroll
                for (int i = 0; i < (WIDE ? 10 : 9); ++i) {
                    c += (uint64_t)traw.v[i] + t[i];
                    traw.v[i] = (uint32_t)c;
                    c >>= 32;
                }
                if (!WIDE) traw.v[9] += (uint32_t)c;
            }
            // the exchanged entry: T * inv(sum); the resident factor carries 2^288 (nine limbs, nine rounds) or 2^320 (ten)
            if (!a.final_step || a.count_changes) {
                const U256 xs = WIDE ? fr_mul10_redc320(t, factor[u]) : fr_mul9_redc288(t, factor[u]);
                if (a.count_changes) {
                    const U256 old = u256_load(a.x + 2 * (my_base + row[u]));
                    changed += u256_eq(xs, old) ? 0u : 1u;
                }
                if (!a.final_step) u256_store(a.y + 2 * (my_base + row[u]), xs);
            }
            if (a.final_step) {
                Acc full = acc_zero();
#pragma unroll
                for (int i = 0; i < (WIDE ? 10 : 9); ++i) full.v[i] = t[i];
                const U256 plain = acc_redc288(full);
                if (!out) u256_store(a.y + 2 * (my_base + row[u]), plain);
                else if (peer[u] != 0xFFFFFFFFu) u256_store(out + 2 * (size_t)peer[u], plain);
            }
        }
    }
    if (b < a.geo.fr_dblocks) {
        for (uint32_t lr = threadIdx.x; lr < RB; lr += 1024) {
            const uint32_t grp = a.grp[(size_t)b * (RB / BLOCK_GROUP) + lr / BLOCK_GROUP];
            if (grp == 0xFFFFFFFFu) continue;
            const uint32_t row = grp * BLOCK_GROUP + lr % BLOCK_GROUP;
            const uint64_t q = ((uint64_t)(row >> 6) * a.geo.n_seg + a.geo.seg) * 64 + (row & 63);
            if (q < a.geo.n_active_pad || q >= (uint64_t)a.geo.n_active_pad + a.geo.n_dangling) continue;
            Acc full = acc_zero();
            fold_row(lr, full.v);
            const U256 old = first_step ? a.initial : u256_load(a.own + 2 * (size_t)row);
            fr_add_dangling_term(full, sh_d, old, a.inv_nm1_w);
            const U256 v = acc_redc288(full);
            u256_store(a.own + 2 * (size_t)row, v);
            // a dangling peer has no out-edges: its exchanged entry is never gathered
            if (!out) u256_store(a.y + 2 * (my_base + row), a.final_step ? v : u256_zero());
            else {
                const uint32_t pe = a.peer[row];
                if (pe != 0xFFFFFFFFu) u256_store(out + 2 * (size_t)pe, v);
            }
            if (a.count_changes) changed += u256_eq(v, old) ? 0u : 1u;
            dacc = fr_add(dacc, v);
        }
    }
    // without statistics only the workgroups that hold dangling rows have anything to add up
    if (!a.count_changes && b >= a.geo.fr_dblocks) {
        if (threadIdx.x == 0) { u256_store(a.sum_part + 2 * (size_t)b, u256_zero()); a.chg_part[b] = 0u; }
        return;
    }
    // per wavefront: integer sum of the T, field sum of the dangling rows' scores (leading workgroups only), count
    acc10_group_sum(traw, 64);
    if (b < a.geo.fr_dblocks) dacc = fr_wave_sum(dacc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) changed += __shfl_xor(changed, off, 64);
    if (lane == 0) {
        u256_store(red_l + 2 * wave, dacc);
        red_c[wave] = changed;
#pragma unroll
        for (int i = 0; i < 10; ++i) red_raw[wave][i] = traw.v[i];
    }
    __syncthreads();
    // the first wavefront adds the sixteen partial results, one per lane
    if (wave == 0) {
        Acc10 r10 = acc10_zero();
        U256 d = u256_zero();
        uint32_t ch = 0;
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 10; ++i) r10.v[i] = red_raw[lane][i];
            d = u256_load(red_l + 2 * lane);
            ch = red_c[lane];
        }
        acc10_group_sum(r10, 16);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) ch += __shfl_xor(ch, off, 64);
        if (b < a.geo.fr_dblocks) d = fr_wave_sum(d);
        if (lane == 0) {
            // score sum of the workgroup's rows: the dangling rows' plain scores plus the reduced integer sum of the T
            // (at most 4400 rows below 2^287 each, or 4240 below 2^304: ten limbs)
            u256_store(a.sum_part + 2 * (size_t)b, a.count_changes ? fr_add(d, acc_redc288(acc_from_acc10(r10))) : u256_zero());
            a.chg_part[b] = ch;
            if (b < a.geo.fr_dblocks) u256_store(a.y + 2 * (my_base + a.geo.rows_cap + b), d);   // partial sum slot b
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernels: statistics, un-permutation, self-test
// ---------------------------------------------------------------------------------------------
__global__ void k_stats_f64(const double* __restrict__ res_part, const double* __restrict__ sum_part,
                            uint32_t n_work, double* __restrict__ out2) {
    // one wavefront, fixed order
    double r = 0.0, t = 0.0;
    for (uint32_t i = threadIdx.x; i < n_work; i += 64) { r += res_part[i]; t += sum_part[i]; }
    r = wave_sum_f64(r);
    t = wave_sum_f64(t);
    if (threadIdx.x == 0) { out2[0] = r; out2[1] = t; }
}
__global__ void k_stats_fr(const uint4* __restrict__ sum_part, const uint32_t* __restrict__ chg_part,
                           uint32_t n_work, uint4* __restrict__ out_sum, unsigned long long* __restrict__ out_chg) {
    U256 t = u256_zero();
    unsigned long long c = 0;
    for (uint32_t i = threadIdx.x; i < n_work; i += 64) {
        t = fr_add(t, u256_load(sum_part + 2 * (size_t)i));
        c += chg_part[i];
    }
    t = fr_wave_sum(t);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if (threadIdx.x == 0) { u256_store(out_sum, t); *out_chg = c; }
}
__global__ void k_unpermute_f64(const uint32_t* __restrict__ pos, const double* __restrict__ x, double* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = pos[i];
    out[i] = p == 0xFFFFFFFFu ? 0.0 : x[p];
}
__global__ void k_unpermute_fr(const uint32_t* __restrict__ pos, const uint4* __restrict__ x, uint4* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = pos[i];
    U256 v = p == 0xFFFFFFFFu ? u256_zero() : u256_load(x + 2 * (size_t)p);
    u256_store(out + 2 * (size_t)i, v);
}
// Gather vector: the entries of this share's own block that another share cites, in the order it asked for (n entries
// picked by idx into the send buffer), and the entries the share's own rows cite (n_self entries picked by idx_self into
// the share's own block of cited entries, which lies behind the own block in the same vector).
// One thread per 8 bytes (f64) or per 16-byte half of a field element.
__global__ void k_pack_f64(const double* own, const uint32_t* __restrict__ idx, double* __restrict__ out, uint32_t n,
                           const uint32_t* __restrict__ idx_self, double* out_self, uint32_t n_self) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = own[idx[i]];
    else if (i - n < n_self) out_self[i - n] = own[idx_self[i - n]];
}
__global__ void k_pack_fr(const uint4* own, const uint32_t* __restrict__ idx, uint4* __restrict__ out, uint32_t n,
                          const uint32_t* __restrict__ idx_self, uint4* out_self, uint32_t n_self) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * n) out[i] = own[2 * (size_t)idx[i >> 1] + (i & 1)];
    else if (i - 2 * n < 2 * n_self) { const uint32_t j = i - 2 * n; out_self[j] = own[2 * (size_t)idx_self[j >> 1] + (j & 1)]; }
}
// a * b mod r through the same multiply-accumulate and reduction the step kernel uses:
// (a * 2^576 * 2^-288) * b * 2^-288.  `c576` = 2^576 mod r.
__global__ void k_fr_mul_test(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, U256 c576, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    U256 av = u256_load(a + 2 * (size_t)i), bv = u256_load(b + 2 * (size_t)i);
    Acc t = acc_zero();
    acc_mac(t, av, c576);
    U256 aw = acc_redc288(t);  // a * 2^288
    Acc u = acc_zero();
    acc_mac(u, aw, bv);
    // exercise the wavefront sum as well: every lane adds its product, lane totals are compared
    // on the host against per-element results, so only add zero from the other lanes
    U256 r = acc_redc288(u);
    u256_store(out + 2 * (size_t)i, r);
}

// ---------------------------------------------------------------------------------------------
// DeviceGraph
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(T** p, size_t count, uint64_t& total) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    total += count * sizeof(T);
    return ET_OK;
}
template <typename T>
static int dev_upload(T** p, const std::vector<T>& v, uint64_t& total) {
    int rc = dev_alloc(p, v.size(), total);
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return ET_OK;
}

DeviceGraph::~DeviceGraph() {
    if (device >= 0) (void)hipSetDevice(device);
    void* ptrs[] = {d_send_idx, d_fb_rec, d_fb_cbase, d_fb_cptr, d_bk_ptr, d_bk_grp, d_bk_col, d_bk_row, d_bk_w, d_bk_rec, d_bk_cbase, d_bk_cptr, d_fb_ptr, d_fb_grp, d_fb_col, d_fb_row, d_fb_val, d_work, d_slices, d_col, d_val, d_wlo, d_whi, d_pos, d_res_part, d_sum_part, d_frsum_part,
                    d_chg_part, d_stat_out, d_bk_phase, d_fb_phase, d_f64_acc, d_fr_acc, vx[0], vx[1], fx[0], fx[1], out_f64, out_fr, d_pretrust, d_scale64, d_scale_fr, d_fb_scale, own64, own_fr, d_peer, x0_64, x0_fr, d_self_idx};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (void* e : ev_begin) (void)hipEventDestroy((hipEvent_t)e);
    for (void* e : ev_end) (void)hipEventDestroy((hipEvent_t)e);
}

static U256 to_u256(const FrRaw& r) {
    U256 u;
    for (int i = 0; i < 4; ++i) { u.v[2 * i] = (uint32_t)r.l[i]; u.v[2 * i + 1] = (uint32_t)(r.l[i] >> 32); }
    return u;
}

// Packs a blocked sweep's (source, row) pairs into one 32-bit record each: row | (position - base) << shift, with one base
// position per 64 consecutive edges of a workgroup's list (the lists ascend by position) and `shift` = the bits the
// workgroup's accumulator slots need.  False when some chunk spans more positions than the remaining bits hold.
static uint32_t slot_bits(uint32_t slots) { uint32_t b = 1; while ((1u << b) < slots) ++b; return b; }
static bool pack_records(const Geometry& geo, uint32_t B, uint32_t shift, const std::vector<uint64_t>& ptr, const std::vector<uint32_t>& col,
                         const std::vector<uint16_t>& row, std::vector<uint64_t>& cptr, std::vector<uint32_t>& cbase,
                         std::vector<uint32_t>& rec) {
    const uint32_t span_max = (uint32_t)((1ull << (32 - shift)) - 1);
    cptr.assign((size_t)B + 1, 0);
    for (uint32_t b = 0; b < B; ++b) cptr[b + 1] = cptr[b] + (ptr[b + 1] - ptr[b] + 63) / 64;
    cbase.assign(cptr[B], 0);
    rec.assign(col.size(), 0);
    std::atomic<bool> fits(true);
    const unsigned T = host_threads();
    run_threads(T, [&](unsigned t) {
        for (uint32_t b = (uint32_t)((uint64_t)B * t / T); b < (uint32_t)((uint64_t)B * (t + 1) / T); ++b) {
            uint32_t base = 0;
            for (uint64_t q = ptr[b]; q < ptr[b + 1]; ++q) {
                const uint32_t pos = geo.pos_of_col(col[q]);
                const uint64_t rel = q - ptr[b];
                if ((rel & 63) == 0) { base = pos; cbase[cptr[b] + rel / 64] = base; }
                if (pos - base > span_max || ((uint32_t)row[q] >> shift) != 0) fits = false;
                rec[q] = (uint32_t)row[q] | ((pos - base) << shift);
            }
        }
    });
    return fits;
}

// ---------------------------------------------------------------------------------------------
// XCD probe: what a workgroup's gathers cost on each XCD
// ---------------------------------------------------------------------------------------------
// One 1024-thread workgroup per CU, as the sweeps launch; every workgroup walks a 32 MB table of 32-byte entries from
// one end to the other the way a sweep walks the vector -- in each pass its 1024 threads gather one entry out of each of
// 1024 consecutive groups of `n_entries / (passes * 1024)` entries, a different one per workgroup -- and reports how
// long it took.
__global__ __launch_bounds__(1024) void k_xcd_probe(const uint4* __restrict__ tab, uint32_t n_entries, uint32_t passes,
                                                    unsigned long long* __restrict__ ticks, uint32_t* __restrict__ sink) {
    __syncthreads();
    const uint64_t t0 = wall_clock64();
    const uint32_t group = n_entries / (passes * 1024u);
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t k = 0; k < passes; ++k) {
        x = x * 1664525u + 1013904223u;
        const uint4* p = tab + 2 * ((size_t)(k * 1024u + threadIdx.x) * group + (x >> 8) % group);
        const uint4 lo = p[0], hi = p[1];
        acc ^= lo.x ^ hi.w;
        if ((k & 7) == 7) __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) ticks[blockIdx.x] = wall_clock64() - t0;
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

void xcd_cost_factors(int device, uint32_t out[8]) {
    static std::mutex lock;
    static std::map<int, std::array<uint32_t, 8>> cache;
    for (int x = 0; x < 8; ++x) out[x] = 1024;
    if (const char* e = getenv("ET_XCD_BALANCE")) if (atoi(e) == 0) return;
    std::lock_guard<std::mutex> guard(lock);
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) { (void)hipGetLastError(); return; }
    if (device < 0) device = previous;
    auto hit = cache.find(device);
    if (hit != cache.end()) { for (int x = 0; x < 8; ++x) out[x] = hit->second[x]; return; }
    std::array<uint32_t, 8> f;
    f.fill(1024);
    // The probe assumes what it measures: one workgroup per compute unit, workgroup b on XCD b mod 8.  A card that is
    // partitioned or is another part (other CU or XCD counts) keeps the uniform factors.
    int cus = 0, xccs = 0;
    bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
              hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && cus == 256 && xccs == 8;
    const uint32_t n_entries = 1u << 20, blocks = 256, passes = 64, reps = 7;
    uint4* tab = nullptr;
    unsigned long long* d_ticks = nullptr;
    uint32_t* d_sink = nullptr;
    ok = ok && hipSetDevice(device) == hipSuccess && hipMalloc(&tab, (size_t)n_entries * 32) == hipSuccess &&
         hipMalloc(&d_ticks, blocks * sizeof(unsigned long long)) == hipSuccess && hipMalloc(&d_sink, 64) == hipSuccess &&
         hipMemset(tab, 1, (size_t)n_entries * 32) == hipSuccess;
    // two halves of the launches, each ranked on its own: factors are applied only when both halves rank the XCDs alike
    // (a card shared with another job, or clocks settling, give rankings that do not repeat)
    double half[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    for (uint32_t r = 0; ok && r < reps; ++r) {
        hipLaunchKernelGGL(k_xcd_probe, dim3(blocks), dim3(1024), 0, 0, tab, n_entries, passes, d_ticks, d_sink);
        std::vector<unsigned long long> t(blocks);
        ok = hipMemcpy(t.data(), d_ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && r > 0)                                   // the first launch warms the table into the caches
            for (uint32_t b = 0; b < blocks; ++b) half[(r - 1) & 1][b % 8] += (double)t[b];
    }
    if (tab) (void)hipFree(tab);
    if (d_ticks) (void)hipFree(d_ticks);
    if (d_sink) (void)hipFree(d_sink);
    if (ok) {
        double per_xcd[8], lo = 0, hi = 0, mean = 0;
        for (int x = 0; x < 8; ++x) { per_xcd[x] = half[0][x] + half[1][x]; mean += per_xcd[x] / 8; }
        lo = hi = per_xcd[0];
        for (int x = 1; x < 8; ++x) { lo = std::min(lo, per_xcd[x]); hi = std::max(hi, per_xcd[x]); }
        // the same XCDs above and below the mean in both halves
        bool repeats = true;
        for (int h = 0; h < 2; ++h) {
            double m = 0;
            for (int x = 0; x < 8; ++x) m += half[h][x] / 8;
            for (int x = 0; x < 8; ++x)
                if (std::fabs(per_xcd[x] - mean) > 0.005 * mean && (half[h][x] > m) != (per_xcd[x] > mean)) repeats = false;
        }
        if (repeats && lo > 0 && hi / lo >= 1.01 && hi / lo <= 1.25)
            for (int x = 0; x < 8; ++x) f[x] = (uint32_t)(1024.0 * per_xcd[x] / lo + 0.5);
    }
    (void)hipGetLastError();
    if (previous >= 0) (void)hipSetDevice(previous);       // the caller's current device stays what it was
    cache[device] = f;
    for (int x = 0; x < 8; ++x) out[x] = f[x];
    if (getenv("ET_XCD_VERBOSE") || getenv("ET_BUILD_STATS"))
        fprintf(stderr, "[et build] xcd cost factors (device %d, %d CUs, %d XCDs): %u %u %u %u %u %u %u %u\n", device, cus, xccs, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
}

// Where a sweep's barriers start.  The barriers hold the sixteen wavefronts of a workgroup inside one window of the vector
// so that the window is shared through the L2; over the leading, densely gathered positions (the most-attesting peers, a
// megabyte or so that stays cached whatever the wavefronts do) they only cost: every barrier waits for the slowest
// wavefront.  So on one share the first barrier comes where the workgroups' lists leave that prefix, if the prefix carries a
// quarter of the edges or more; the exact sweep then takes a barrier every fourth pass instead of every eighth.  1M peers /
// 16M edges, power-law out-degrees: f64 45.5 -> 43 us, exact 106.4 -> 102 us; a uniform graph of the same size has no such
// prefix and loses 2 us (f64) and 20 us (exact) if made to wait, a share of a 16M-peer graph 18 us.
// ET_F64_BARRIER / ET_FR_BARRIER (period) and ET_F64_BARRIER_FROM / ET_FR_BARRIER_FROM (percent of the list) override.
static void sweep_barriers(const HostGraph& h, DeviceGraph& g) {
    auto dense_share_256 = [&](const std::vector<uint64_t>& ptr, const std::vector<uint32_t>& col, uint32_t B) -> uint32_t {
        if (h.geo.n_seg != 1 || h.geo.n_chunks != 1 || h.dense_prefix == 0 || B == 0 || ptr.size() != (size_t)B + 1 || ptr[B] == 0) return 0;
        uint64_t dense = 0;
        for (uint32_t b = 0; b < B; ++b)      // a list ascends by position; one share: the source code is the position
            dense += std::partition_point(col.begin() + ptr[b], col.begin() + ptr[b + 1], [&](uint32_t c) { return c < h.dense_prefix; }) - (col.begin() + ptr[b]);
        return (uint32_t)(dense * 256 / ptr[B]);
    };
    auto env_int = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; };
    uint32_t share = dense_share_256(h.bk_ptr, h.bk_col, h.geo.f64_blocks);
    g.f64_bar_period = 2; g.f64_bar_first_256 = share >= 64 ? std::min(share, 230u) : 0u;
    share = dense_share_256(h.fb_ptr, h.fb_col, h.geo.fr_blocks);
    g.fr_bar_first_256 = share >= 64 ? std::min(share, 230u) : 0u;
    g.fr_bar_period = g.fr_bar_first_256 ? 4 : 8;
    if (env_int("ET_F64_BARRIER") > 0) g.f64_bar_period = (uint32_t)env_int("ET_F64_BARRIER");
    if (env_int("ET_F64_BARRIER_FROM") >= 0) g.f64_bar_first_256 = (uint32_t)env_int("ET_F64_BARRIER_FROM") * 256 / 100;
    if (env_int("ET_FR_BARRIER") > 0) g.fr_bar_period = (uint32_t)env_int("ET_FR_BARRIER");
    if (env_int("ET_FR_BARRIER_FROM") >= 0) g.fr_bar_first_256 = (uint32_t)env_int("ET_FR_BARRIER_FROM") * 256 / 100;
    if (getenv("ET_BUILD_STATS")) fprintf(stderr, "[et upload] barriers: f64 every %u from %u/256, exact every %u from %u/256\n", g.f64_bar_period, g.f64_bar_first_256, g.fr_bar_period, g.fr_bar_first_256);
}

int DeviceGraph::create(const HostGraph& h, int variants_, int device_, DeviceGraph** out) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    if (device_ < 0) HIP_TRY(hipGetDevice(&device_));
    HIP_TRY(hipSetDevice(device_));
    std::unique_ptr<DeviceGraph> g(new DeviceGraph());
    g->device = device_;
    g->variants = variants_ ? variants_ : (ET_VARIANT_FR | ET_VARIANT_F64);
    g->n = h.n; g->n_valid = h.n_valid; g->nnz = h.nnz; g->nnz_local = h.nnz_local;
    g->geo = h.geo; g->n_steps = h.n_steps; g->n_slices = (uint32_t)h.slices.size();
    g->n_work = (uint32_t)h.work.size(); g->n_dangling_local = h.n_dangling_local;
    for (const WorkDesc& w : h.work) if (w.flags & WORK_DANGLING) g->n_dslots = std::max(g->n_dslots, w.dslot + 1);
    g->fr_blocked = h.geo.fr_blocks != 0 && !h.fb_ptr.empty();
    g->n_fr_parts = g->fr_blocked ? h.geo.fr_blocks : g->n_work;
    g->pos_of_ext = h.pos_of_ext;
    for (int x = 0; x < 8; ++x) g->xcd_cost[x] = h.xcd_cost[x];
    g->inv_nm1_w = h.inv_nm1_w;
    g->inv_nm1 = 1.0 / (double)(h.n_valid - 1);
    uint64_t& tot = g->device_bytes;
    int rc;
    if ((rc = dev_upload(&g->d_work, h.work, tot))) return rc;
    if ((rc = dev_upload(&g->d_slices, h.slices, tot))) return rc;
    if ((rc = dev_upload(&g->d_col, h.col, tot))) return rc;
    if ((rc = dev_upload(&g->d_pos, h.pos_of_ext, tot))) return rc;
    if (h.geo.n_seg == 1) {             // the whole graph: the last launch of a converge writes slot order itself
        std::vector<uint32_t> peer(h.geo.rows_cap, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < h.n; ++i) {
            const uint32_t p = h.pos_of_ext[i];
            if (p == 0xFFFFFFFFu) continue;
            if (p >= h.geo.rows_cap) { set_error("internal: position beyond the rows"); return ET_E_OUT_OF_RANGE; }
            peer[p] = i;
        }
        if ((rc = dev_upload(&g->d_peer, peer, tot))) return rc;
        g->no_fused_run = getenv("ET_NO_FUSED_RUN") != nullptr;
    }
    const size_t slots = (size_t)h.n_steps * 64;
    const size_t vec_len = h.geo.vec_len;                                  // entries of the vectors the steps take
    const size_t stride = h.geo.seg_stride(), full_len = (size_t)h.geo.n_seg * stride;
    g->vec_len = vec_len;
    g->full_len = full_len;
    sweep_barriers(h, *g);
    // raw form: opinions stay as they are (one byte each when they fit), the source factors stay
    // resident; weighted form: opinions and factors are only needed until the weights are computed
    g->raw = h.values_are_u32;
    uint64_t tmp_bytes = 0;
    uint4* d_valfr = nullptr;
    auto free_tmp = [&]() { if (d_valfr) (void)hipFree(d_valfr); d_valfr = nullptr; };
    if (g->raw) {
        // how wide the edge records' opinions are: from the plan's figure over ALL surviving opinions (the per-row
        // stream below is empty where both variants run the blocked sweep)
        uint32_t vmax = h.value_max;
        for (uint32_t v : h.val_u32) vmax = std::max(vmax, v);
        for (uint32_t v : h.bk_val_u32) vmax = std::max(vmax, v);
        g->val_bytes = vmax <= 255 ? 1 : 4;
        if ((h.geo.fr_compact == 2) != (g->val_bytes == 4) && h.geo.fr_blocks != 0) {
            set_error("internal: the exact sweep's rows do not match the width of the opinions");
            return ET_E_INVALID_ARG;
        }
    }
    // narrows 32-bit opinions to bytes when they fit
    auto upload_values = [&](void** dst, const std::vector<uint32_t>& v) -> int {
        if (g->val_bytes == 4) { uint32_t* p = nullptr; int r = dev_upload(&p, v, tot); *dst = p; return r; }
        std::vector<uint8_t> v8(v.size());
        for (size_t i = 0; i < v.size(); ++i) v8[i] = (uint8_t)v[i];
        uint8_t* p = nullptr;
        int r = dev_upload(&p, v8, tot);
        *dst = p;
        return r;
    };
    // the factors of this share's own rows stay resident (a step scales the rows it owns), indexed by local row
    if (h.scale_f64.size() != full_len || h.scale_fr.size() != full_len) { set_error("internal: factor arrays"); return ET_E_INVALID_ARG; }
    const size_t own_first = (size_t)h.geo.seg * stride;
    if ((rc = dev_alloc(&g->d_scale64, stride, tot)) || (rc = dev_alloc(&g->d_scale_fr, stride * 2, tot))) return rc;
    HIP_TRY(hipMemcpy(g->d_scale64, h.scale_f64.data() + own_first, stride * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g->d_scale_fr, h.scale_fr.data() + own_first, stride * 32, hipMemcpyHostToDevice));
    // weighted form only: every source's factor, by position, until the weights are computed
    double* d_scale64_all = nullptr;
    uint4* d_scale_fr_all = nullptr;
    auto free_all = [&]() { if (d_scale64_all) (void)hipFree(d_scale64_all); if (d_scale_fr_all) (void)hipFree(d_scale_fr_all); d_scale64_all = nullptr; d_scale_fr_all = nullptr; };
    if (!g->raw) {
        if ((rc = dev_upload(&d_scale64_all, h.scale_f64, tmp_bytes)) || (rc = dev_alloc(&d_scale_fr_all, full_len * 2, tmp_bytes))) { free_all(); return rc; }
        HIP_TRY(hipMemcpy(d_scale_fr_all, h.scale_fr.data(), full_len * 32, hipMemcpyHostToDevice));
    }
    const unsigned nb = (unsigned)((slots + 255) / 256);
    const bool can_pack = g->raw && vec_len <= 0xFFFFFFFFull && h.geo.n_chunks == 1;
    if (g->variants & ET_VARIANT_F64) {
        g->n_f64_parts = h.geo.f64_blocks;
        const size_t ne = h.bk_col.size();
        if ((rc = dev_upload(&g->d_bk_ptr, h.bk_ptr, tot)) || (rc = dev_upload(&g->d_bk_grp, h.bk_grp, tot))) return rc;
        // byte opinions: 5-byte edge records (see k_step_f64_blocked) unless a chunk of 64 consecutive edges of a
        // workgroup spans more than 65535 positions (a few edges over very many peers) or positions need 33 bits
        // (a chunked exchange cuts a list into runs that do not start on a 64-edge boundary: 7-byte records there)
        if (can_pack && !getenv("ET_F64_UNPACKED")) {
            std::vector<uint64_t> cptr;
            std::vector<uint32_t> cbase, rec;
            g->f64_rec_shift = slot_bits(h.geo.f64_rb);
            if (pack_records(h.geo, h.geo.f64_blocks, g->f64_rec_shift, h.bk_ptr, h.bk_col, h.bk_row, cptr, cbase, rec)) {
                g->packed = true;
                if ((rc = dev_upload(&g->d_bk_rec, rec, tot)) || (rc = dev_upload(&g->d_bk_cbase, cbase, tot)) ||
                    (rc = dev_upload(&g->d_bk_cptr, cptr, tot))) return rc;
            }
        }
        if (!g->packed && ((rc = dev_upload(&g->d_bk_col, h.bk_col, tot)) || (rc = dev_upload(&g->d_bk_row, h.bk_row, tot)))) return rc;
        if (g->raw) {
            if ((rc = upload_values(&g->d_bk_w, h.bk_val_u32))) return rc;
            if ((rc = dev_alloc(&g->own64, h.geo.rows_cap, tot))) return rc;
            HIP_TRY(hipMemset(g->own64, 0, (size_t)h.geo.rows_cap * sizeof(double)));
        } else {
            double* w = nullptr;
            if ((rc = dev_alloc(&w, ne, tot))) return rc;
            g->d_bk_w = w;
            if (ne) {
                uint4* d_bv = nullptr;
                if ((rc = dev_alloc(&d_bv, ne * 2, tmp_bytes))) return rc;
                HIP_TRY(hipMemcpy(d_bv, h.bk_val_fr.data(), ne * 32, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_normalise_f64_fr, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, 0, g->d_bk_col, d_bv, d_scale64_all, w, ne, h.geo);
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(d_bv);
            }
        }
        if (h.geo.n_chunks > 1) {
            if ((rc = dev_upload(&g->d_bk_phase, h.bk_phase, tot)) || (rc = dev_alloc(&g->d_f64_acc, (size_t)h.geo.f64_blocks * h.geo.f64_rb, tot))) return rc;
        }
        // the graph's own vectors serve the whole-graph runs (run_f64); a share of several steps on its driver's vectors
        for (int i = 0; i < 2 && h.geo.n_seg == 1; ++i) {
            if ((rc = dev_alloc(&g->vx[i], vec_len, tot))) return rc;
            HIP_TRY(hipMemset(g->vx[i], 0, vec_len * sizeof(double)));
        }
        if (h.geo.n_seg == 1) {         // slots without a member read zero and are never written again
            if ((rc = dev_alloc(&g->out_f64, h.n, tot))) return rc;
            HIP_TRY(hipMemset(g->out_f64, 0, (size_t)h.n * sizeof(double)));
        }
        if ((rc = dev_alloc(&g->d_res_part, (size_t)STATS_RING * g->n_f64_parts, tot)) ||
            (rc = dev_alloc(&g->d_sum_part, (size_t)STATS_RING * g->n_f64_parts, tot))) return rc;
    }
    if (g->variants & ET_VARIANT_FR) {
        if (g->raw) {
            if ((rc = upload_values(&g->d_val, h.val_u32))) return rc;
            if ((rc = dev_alloc(&g->own_fr, (size_t)h.geo.rows_cap * 2, tot))) return rc;
            HIP_TRY(hipMemset(g->own_fr, 0, (size_t)h.geo.rows_cap * 32));
        } else {
            if ((rc = dev_alloc(&d_valfr, slots * 2, tmp_bytes))) return rc;
            if (slots) HIP_TRY(hipMemcpy(d_valfr, h.val_fr.data(), slots * 32, hipMemcpyHostToDevice));
            if ((rc = dev_alloc(&g->d_wlo, slots, tot)) || (rc = dev_alloc(&g->d_whi, slots, tot))) { free_tmp(); return rc; }
            if (slots) hipLaunchKernelGGL(k_normalise_fr_fr, dim3(nb), dim3(256), 0, 0, g->d_col, d_valfr, d_scale_fr_all, g->d_wlo, g->d_whi, slots, h.geo);
            HIP_TRY(hipDeviceSynchronize());
            free_tmp();
        }
        for (int i = 0; i < 2 && h.geo.n_seg == 1; ++i) {
            if ((rc = dev_alloc(&g->fx[i], vec_len * 2, tot))) return rc;
            HIP_TRY(hipMemset(g->fx[i], 0, vec_len * 32));
        }
        if (g->fr_blocked && h.geo.n_chunks > 1) {
            const size_t words = h.geo.fr_compact ? 4 * (size_t)h.geo.fr_rb + h.geo.fr_rb / 2 : 8 * (size_t)h.geo.fr_rb;
            if ((rc = dev_upload(&g->d_fb_phase, h.fb_phase, tot)) || (rc = dev_alloc(&g->d_fr_acc, (size_t)h.geo.fr_blocks * words, tot))) return rc;
        }
        if (g->fr_blocked) {
            if ((rc = dev_upload(&g->d_fb_ptr, h.fb_ptr, tot)) || (rc = dev_upload(&g->d_fb_grp, h.fb_grp, tot))) return rc;
            if (h.geo.fr_compact == 2) { uint32_t* p = nullptr; if ((rc = dev_upload(&p, h.fb_val_u32, tot))) return rc; g->d_fb_val = p; }
            else { uint8_t* p = nullptr; if ((rc = dev_upload(&p, h.fb_val, tot))) return rc; g->d_fb_val = p; }
            if (can_pack && h.geo.fr_compact && !getenv("ET_FR_UNPACKED")) {
                std::vector<uint64_t> cptr;
                std::vector<uint32_t> cbase, rec;
                g->fr_rec_shift = slot_bits(h.geo.fr_rb);
                if (pack_records(h.geo, h.geo.fr_blocks, g->fr_rec_shift, h.fb_ptr, h.fb_col, h.fb_row, cptr, cbase, rec)) {
                    g->fr_packed = true;
                    if ((rc = dev_upload(&g->d_fb_rec, rec, tot)) || (rc = dev_upload(&g->d_fb_cbase, cbase, tot)) ||
                        (rc = dev_upload(&g->d_fb_cptr, cptr, tot))) return rc;
                }
            }
            if (!g->fr_packed && ((rc = dev_upload(&g->d_fb_col, h.fb_col, tot)) || (rc = dev_upload(&g->d_fb_row, h.fb_row, tot)))) return rc;
            const void* sweeps[] = {(const void*)k_step_fr_blocked<true, false, true, true>, (const void*)k_step_fr_blocked<true, false, false, true>,
                                    (const void*)k_step_fr_blocked<true, false, true>,
                                    (const void*)k_step_fr_blocked<true, false>, (const void*)k_step_fr_blocked<false, false>,
                                    (const void*)k_step_fr_blocked<true, false, true, true, true>, (const void*)k_step_fr_blocked<true, false, false, true, true>,
                                    (const void*)k_step_fr_blocked<true, false, true, false, true>,
                                    (const void*)k_step_fr_blocked<true, false, false, false, true>, (const void*)k_step_fr_blocked<false, false, false, false, true>,
                                    (const void*)k_step_fr_blocked<true, true>, (const void*)k_step_fr_blocked<false, true>};
            for (const void* f : sweeps) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            // the sweep multiplies unreduced row sums by inv(sum) * 2^288 (the resident factors carry 2^576)
            if ((rc = dev_alloc(&g->d_fb_scale, stride * 2, tot))) return rc;
            if (h.geo.fr_compact == 2) hipLaunchKernelGGL(k_scale_320, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, 0, g->d_scale_fr, g->d_fb_scale, stride);
            else hipLaunchKernelGGL(k_scale_288, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, 0, g->d_scale_fr, g->d_fb_scale, stride);
            HIP_TRY(hipDeviceSynchronize());
        }
        if (h.geo.n_seg == 1) {
            if ((rc = dev_alloc(&g->out_fr, (size_t)h.n * 2, tot))) return rc;
            HIP_TRY(hipMemset(g->out_fr, 0, (size_t)h.n * 32));
        }
        if ((rc = dev_alloc(&g->d_frsum_part, (size_t)STATS_RING * g->n_fr_parts * 2, tot)) ||
            (rc = dev_alloc(&g->d_chg_part, (size_t)STATS_RING * g->n_fr_parts, tot))) return rc;
    }
    if ((rc = dev_alloc(&g->d_stat_out, 64, tot))) return rc;
    // the exchange: what this share sends to and receives from every other share
    {
        const Geometry& geo = h.geo;
        const uint32_t own_len = geo.seg_stride();
        if (geo.compact) {
            for (uint32_t c = 0; c < geo.n_seg; ++c) {
                if (c == geo.seg) continue;
                g->recv_off[c] = h.blk_off[c];
                g->recv_cnt[c] = (uint32_t)h.need[c].size() + geo.nd_cap;
            }
            // the rows of this share that its own rows cite: copied by pack into their block, in the sweeps' order
            for (uint32_t l : h.need[geo.seg]) if (l >= geo.rows_cap) { set_error("internal: cited row outside the share's rows"); return ET_E_OUT_OF_RANGE; }
            g->self_off = h.blk_off[geo.seg];
            g->self_cnt = (uint32_t)h.need[geo.seg].size();
            if ((size_t)g->self_off + g->self_cnt > vec_len) { set_error("internal: block of own cited entries outside the vector"); return ET_E_OUT_OF_RANGE; }
            if ((rc = dev_upload(&g->d_self_idx, h.need[geo.seg], tot))) return rc;
            // what the other shares cite here: known now, or told later (set_send_lists) by whoever lays out all shares
            if (h.send_known) {
                const std::vector<uint32_t>* lists[8];
                for (int c = 0; c < 8; ++c) lists[c] = &h.send[c];
                if ((rc = g->set_send_lists(lists))) return rc;
            }
        } else {
            for (uint32_t c = 0; c < geo.n_seg; ++c) {
                if (c == geo.seg) continue;
                g->send_off[c] = 0; g->send_cnt[c] = own_len;
                g->recv_off[c] = c * own_len; g->recv_cnt[c] = own_len;
            }
            g->send_total = geo.n_seg > 1 ? own_len : 0;
        }
    }
    // the LDS accumulators may exceed the 64 KiB a kernel may use without opting in
    {
        const int lds_max = 160 * 1024 - 1024;
        const void* fns[] = {(const void*)k_step_f64_blocked<uint8_t, false, true>, (const void*)k_step_f64_blocked<uint8_t, true, true>,
                             (const void*)k_step_f64_blocked<uint32_t, false, true>, (const void*)k_step_f64_blocked<uint32_t, true, true>,
                             (const void*)k_step_f64_blocked<uint8_t, false, false, true>, (const void*)k_step_f64_blocked<uint8_t, true, false, true>,
                             (const void*)k_step_f64_blocked<uint8_t, false>, (const void*)k_step_f64_blocked<uint8_t, true>,
                             (const void*)k_step_f64_blocked<uint32_t, false>, (const void*)k_step_f64_blocked<uint32_t, true>,
                             (const void*)k_step_f64_blocked<double, false>, (const void*)k_step_f64_blocked<double, true>,
                             (const void*)k_step_f64_blocked<uint8_t, false, true, false, true>, (const void*)k_step_f64_blocked<uint8_t, true, true, false, true>,
                             (const void*)k_step_f64_blocked<uint32_t, false, true, false, true>, (const void*)k_step_f64_blocked<uint32_t, true, true, false, true>,
                             (const void*)k_step_f64_blocked<uint8_t, false, false, false, true>, (const void*)k_step_f64_blocked<uint8_t, true, false, false, true>,
                             (const void*)k_step_f64_blocked<uint32_t, false, false, false, true>, (const void*)k_step_f64_blocked<uint32_t, true, false, false, true>,
                             (const void*)k_step_f64_blocked<double, false, false, false, true>, (const void*)k_step_f64_blocked<double, true, false, false, true>};
        for (const void* f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    free_tmp();
    free_all();
    if (getenv("ET_BUILD_STATS"))
        fprintf(stderr, "[et upload] edge records: f64 %s, exact %s; vectors of %zu entries (%zu in position order), %.2f GB on the device\n",
                g->packed ? "packed (5 bytes)" : "source, slot, opinion apart", !g->fr_blocked ? "per-row stream" : (g->fr_packed ? "packed (5 bytes)" : "source, slot, opinion apart"),
                g->vec_len, g->full_len, (double)g->device_bytes / 1e9);
    *out = g.release();
    return ET_OK;
}

int DeviceGraph::init_f64(double initial, double* x, hipStream_t st, bool plain) {
    if (!(variants & ET_VARIANT_F64)) { set_error("graph was built without the f64 variant"); return ET_E_UNSUPPORTED; }
    uint32_t stride = geo.seg_stride();
    double d0 = (double)n_dangling_local * initial;
    hipLaunchKernelGGL(k_init_f64, dim3((stride + 255) / 256), dim3(256), 0, st, geo, initial, d0, x,
                       raw && !plain ? own64 : nullptr, raw && !plain ? d_scale64 : nullptr);
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
int DeviceGraph::init_fr(uint64_t lo, uint64_t hi, uint4* x, hipStream_t st, bool plain) {
    if (!(variants & ET_VARIANT_FR)) { set_error("graph was built without the Fr variant"); return ET_E_UNSUPPORTED; }
    uint32_t stride = geo.seg_stride();
    Fr init = Fr::from_u128(lo, hi);
    Fr d0 = init * Fr::from_u64(n_dangling_local);
    hipLaunchKernelGGL(k_init_fr, dim3((stride + 255) / 256), dim3(256), 0, st, geo, to_u256(init.to_raw()), to_u256(d0.to_raw()), x,
                       raw && !plain ? own_fr : nullptr, raw && !plain ? d_scale_fr : nullptr);
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
template <typename VT, bool PACKED = false, bool PHASED = false>
static void launch_f64(const StepArgsF64B& a, unsigned blocks, size_t lds, hipStream_t st) {
    if (!PHASED && (a.first_step || a.out)) {          // the first or last launch of a whole converge
        if (a.alpha != 0.0) hipLaunchKernelGGL((k_step_f64_blocked<VT, true, PACKED, false, !PHASED>), dim3(blocks), dim3(1024), lds, st, a);
        else hipLaunchKernelGGL((k_step_f64_blocked<VT, false, PACKED, false, !PHASED>), dim3(blocks), dim3(1024), lds, st, a);
        return;
    }
    if (a.alpha != 0.0) hipLaunchKernelGGL((k_step_f64_blocked<VT, true, PACKED, PHASED>), dim3(blocks), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((k_step_f64_blocked<VT, false, PACKED, PHASED>), dim3(blocks), dim3(1024), lds, st, a);
}
int DeviceGraph::step_f64(const double* x, double* y, uint32_t iteration, double alpha, const double* pretrust, uint32_t flags, hipStream_t st, int phase,
                          const StepExtra* ex) {
    if (!(variants & ET_VARIANT_F64)) { set_error("graph was built without the f64 variant"); return ET_E_UNSUPPORTED; }
    if (x == y) { set_error("step needs distinct input and output vectors"); return ET_E_INVALID_ARG; }
    if (alpha != 0.0 && !pretrust) { set_error("alpha != 0 needs a pre-trust vector"); return ET_E_INVALID_ARG; }
    const size_t ring = (size_t)(iteration % STATS_RING) * n_f64_parts;
    StepArgsF64B a;
    a.bk_ptr = d_bk_ptr; a.grp = d_bk_grp; a.col = d_bk_col; a.w = d_bk_w; a.row = d_bk_row; a.x = x; a.y = y; a.pretrust = pretrust;
    a.rec = d_bk_rec; a.cbase = d_bk_cbase; a.cptr = d_bk_cptr; a.rec_shift = f64_rec_shift;
    a.own = own64; a.scale = d_scale64;
    a.res_part = d_res_part + ring; a.sum_part = d_sum_part + ring;
    a.geo = geo; a.inv_nm1 = inv_nm1; a.alpha = alpha; a.final_step = (flags & ET_STEP_FINAL) ? 1u : 0u;
    a.phase_ptr = d_bk_phase; a.acc_g = d_f64_acc; a.n_phases = geo.n_chunks;
    a.bar_period = f64_bar_period; a.bar_first_256 = f64_bar_first_256;
    a.stats = (flags & ET_STEP_NO_STATS) ? 0u : 1u;
    a.first_step = ex && ex->first_step ? 1u : 0u; a.initial = ex ? ex->initial64 : 0.0;
    a.out = ex && a.final_step ? static_cast<double*>(ex->out) : nullptr; a.peer = d_peer;
    if (a.out && (!d_peer || geo.n_chunks != 1)) { set_error("internal: scores in slot order need the whole graph in one launch"); return ET_E_UNSUPPORTED; }
    if (phase >= (int)geo.n_chunks) { set_error("no such phase"); return ET_E_OUT_OF_RANGE; }
    const size_t lds = (size_t)geo.f64_rb * sizeof(double);
    // phase < 0: the whole iteration, as n_chunks launches one after the other
    for (uint32_t q = phase < 0 ? 0u : (uint32_t)phase; q < (phase < 0 ? geo.n_chunks : (uint32_t)phase + 1); ++q) {
        a.phase = q;
        if (geo.n_chunks > 1) {   // several launches per iteration: byte opinions, unpacked records (see create)
            if (!raw || val_bytes != 1 || packed) { set_error("internal: chunked graph without byte records"); return ET_E_UNSUPPORTED; }
            launch_f64<uint8_t, false, true>(a, geo.f64_blocks, lds, st);
        }
        else if (!raw) launch_f64<double>(a, geo.f64_blocks, lds, st);
        else if (packed && val_bytes == 1) launch_f64<uint8_t, true>(a, geo.f64_blocks, lds, st);
        else if (packed) launch_f64<uint32_t, true>(a, geo.f64_blocks, lds, st);
        else if (val_bytes == 1) launch_f64<uint8_t>(a, geo.f64_blocks, lds, st);
        else launch_f64<uint32_t>(a, geo.f64_blocks, lds, st);
    }
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
int DeviceGraph::step_fr(const uint4* x, uint4* y, uint32_t iteration, uint32_t flags, hipStream_t st, int phase, const StepExtra* ex) {
    if (!(variants & ET_VARIANT_FR)) { set_error("graph was built without the Fr variant"); return ET_E_UNSUPPORTED; }
    if (x == y) { set_error("step needs distinct input and output vectors"); return ET_E_INVALID_ARG; }
    if (fr_blocked) {
        StepArgsFrB fb;
        const size_t ringb = (size_t)(iteration % STATS_RING) * n_fr_parts;
        fb.ptr = d_fb_ptr; fb.grp = d_fb_grp; fb.col = d_fb_col; fb.row = d_fb_row; fb.val = d_fb_val; fb.x = x; fb.y = y;
        fb.rec = d_fb_rec; fb.cbase = d_fb_cbase; fb.cptr = d_fb_cptr; fb.rec_shift = fr_rec_shift;
        fb.own = own_fr; fb.scale288 = d_fb_scale; fb.sum_part = d_frsum_part + 2 * ringb; fb.chg_part = d_chg_part + ringb;
        fb.geo = geo; fb.final_step = (flags & ET_STEP_FINAL) ? 1u : 0u; fb.inv_nm1_w = to_u256(inv_nm1_w);
        fb.inv_nm1_w576 = to_u256((Fr::from_raw(inv_nm1_w) * Fr::one().pow2k(288)).to_raw());
        fb.phase_ptr = d_fb_phase; fb.acc_g = d_fr_acc; fb.n_phases = geo.n_chunks;
        fb.bar_period = fr_bar_period; fb.bar_first_256 = fr_bar_first_256;
        fb.count_changes = (flags & ET_STEP_NO_STATS) ? 0u : 1u;
        fb.first_step = ex && ex->first_step ? 1u : 0u; fb.initial = to_u256(ex ? ex->initial_fr : FrRaw{});
        fb.out = ex && fb.final_step ? static_cast<uint4*>(ex->out) : nullptr; fb.peer = d_peer;
        if (fb.out && (!d_peer || geo.n_chunks != 1)) { set_error("internal: scores in slot order need the whole graph in one launch"); return ET_E_UNSUPPORTED; }
        if (phase >= (int)geo.n_chunks) { set_error("no such phase"); return ET_E_OUT_OF_RANGE; }
        for (uint32_t q = phase < 0 ? 0u : (uint32_t)phase; q < (phase < 0 ? geo.n_chunks : (uint32_t)phase + 1); ++q) {
            fb.phase = q;
            const dim3 grid(geo.fr_blocks), wg(1024);
            const size_t lds = (size_t)geo.fr_rb * (geo.fr_compact == 2 ? 38 : (geo.fr_compact ? 36 : 64));
            if (geo.n_chunks > 1) {
                if (geo.fr_compact) hipLaunchKernelGGL((k_step_fr_blocked<true, true>), grid, wg, lds, st, fb);
                else hipLaunchKernelGGL((k_step_fr_blocked<false, true>), grid, wg, lds, st, fb);
            } else if (fb.first_step || fb.out) {          // the first or last launch of a whole converge
                if (geo.fr_compact == 2 && fr_packed) hipLaunchKernelGGL((k_step_fr_blocked<true, false, true, true, true>), grid, wg, lds, st, fb);
                else if (geo.fr_compact == 2) hipLaunchKernelGGL((k_step_fr_blocked<true, false, false, true, true>), grid, wg, lds, st, fb);
                else if (fr_packed) hipLaunchKernelGGL((k_step_fr_blocked<true, false, true, false, true>), grid, wg, lds, st, fb);
                else if (geo.fr_compact) hipLaunchKernelGGL((k_step_fr_blocked<true, false, false, false, true>), grid, wg, lds, st, fb);
                else hipLaunchKernelGGL((k_step_fr_blocked<false, false, false, false, true>), grid, wg, lds, st, fb);
            } else {
                if (geo.fr_compact == 2 && fr_packed) hipLaunchKernelGGL((k_step_fr_blocked<true, false, true, true>), grid, wg, lds, st, fb);
                else if (geo.fr_compact == 2) hipLaunchKernelGGL((k_step_fr_blocked<true, false, false, true>), grid, wg, lds, st, fb);
                else if (fr_packed) hipLaunchKernelGGL((k_step_fr_blocked<true, false, true>), grid, wg, lds, st, fb);
                else if (geo.fr_compact) hipLaunchKernelGGL((k_step_fr_blocked<true, false>), grid, wg, lds, st, fb);
                else hipLaunchKernelGGL((k_step_fr_blocked<false, false>), grid, wg, lds, st, fb);
            }
        }
        HIP_TRY(hipGetLastError());
        return ET_OK;
    }
    if (phase > 0) { set_error("no such phase"); return ET_E_OUT_OF_RANGE; }   // n_chunks is 1 without the blocked sweep
    if (ex) { set_error("internal: the row-per-lane kernels take a vector written by init"); return ET_E_UNSUPPORTED; }
    StepArgsFr a;
    a.work = d_work; a.slices = d_slices; a.col = d_col; a.wlo = d_wlo; a.whi = d_whi; a.val = d_val; a.x = x; a.y = y;
    a.own = own_fr; a.scale = d_scale_fr;
    size_t ring = (size_t)(iteration % STATS_RING) * n_work;
    a.sum_part = d_frsum_part + 2 * ring; a.chg_part = d_chg_part + ring;
    a.geo = geo; a.n_work = n_work; a.inv_nm1_w = to_u256(inv_nm1_w); a.final_step = (flags & ET_STEP_FINAL) ? 1u : 0u;
    a.n_dslots = n_dslots;
    // a share without rows still launches one workgroup: it owns the partial-sum tail of its segment
    const unsigned grid = std::max(1u, (n_work + WPB_FR - 1) / WPB_FR);
    if (!raw) hipLaunchKernelGGL(k_step_fr, dim3(grid), dim3(WPB_FR * 64), 0, st, a);
    else if (val_bytes == 1) hipLaunchKernelGGL(k_step_fr_raw<uint8_t>, dim3(grid), dim3(WPB_FR * 64), 0, st, a);
    else hipLaunchKernelGGL(k_step_fr_raw<uint32_t>, dim3(grid), dim3(WPB_FR * 64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
int DeviceGraph::unpermute_f64(const double* x, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_unpermute_f64, dim3((n + 255) / 256), dim3(256), 0, st, d_pos, x, out, n);
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
int DeviceGraph::unpermute_fr(const uint4* x, uint4* out, hipStream_t st) {
    hipLaunchKernelGGL(k_unpermute_fr, dim3((n + 255) / 256), dim3(256), 0, st, d_pos, x, out, n);
    HIP_TRY(hipGetLastError());
    return ET_OK;
}
int DeviceGraph::stats_f64(uint32_t iteration, hipStream_t st, et_iter_stats* out) {
    if (!(variants & ET_VARIANT_F64)) return ET_E_UNSUPPORTED;
    size_t ring = (size_t)(iteration % STATS_RING) * n_f64_parts;
    double* o = (double*)d_stat_out;
    hipLaunchKernelGGL(k_stats_f64, dim3(1), dim3(64), 0, st, d_res_part + ring, d_sum_part + ring, n_f64_parts, o);
    HIP_TRY(hipGetLastError());
    double h[2];
    HIP_TRY(hipMemcpyAsync(h, o, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memset(out, 0, sizeof(*out));
    out->residual_l1 = h[0];
    out->sum_f64 = h[1];
    return ET_OK;
}
int DeviceGraph::stats_fr(uint32_t iteration, hipStream_t st, et_iter_stats* out) {
    if (!(variants & ET_VARIANT_FR)) return ET_E_UNSUPPORTED;
    size_t ring = (size_t)(iteration % STATS_RING) * n_fr_parts;
    uint4* o = (uint4*)d_stat_out;
    unsigned long long* oc = (unsigned long long*)(o + 2);
    hipLaunchKernelGGL(k_stats_fr, dim3(1), dim3(64), 0, st, d_frsum_part + 2 * ring, d_chg_part + ring, n_fr_parts, o, oc);
    HIP_TRY(hipGetLastError());
    uint8_t h[40];
    HIP_TRY(hipMemcpyAsync(h, o, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->sum.b, h, 32);
    std::memcpy(&out->changed, h + 32, 8);
    return ET_OK;
}

int DeviceGraph::set_send_lists(const std::vector<uint32_t>* const lists[8]) {
    if (!geo.compact) { set_error("only a share with a gather vector packs"); return ET_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    const uint32_t own_len = geo.seg_stride();
    std::vector<uint32_t> idx;
    for (uint32_t c = 0; c < geo.n_seg; ++c) {
        send_off[c] = send_cnt[c] = 0;
        if (c == geo.seg) continue;
        send_off[c] = (uint32_t)idx.size();
        for (uint32_t v : *lists[c]) {
            if (v >= geo.rows_cap) { set_error("cited row outside this share's rows"); return ET_E_OUT_OF_RANGE; }
            idx.push_back(v);
        }
        for (uint32_t k = 0; k < geo.nd_cap; ++k) idx.push_back(geo.rows_cap + k);   // the partial sums travel with every piece
        send_cnt[c] = (uint32_t)idx.size() - send_off[c];
    }
    (void)own_len;
    if (d_send_idx) { (void)hipFree(d_send_idx); d_send_idx = nullptr; }
    int rc = dev_upload(&d_send_idx, idx, device_bytes);
    if (rc) return rc;
    send_total = idx.size();
    send_known = true;
    return ET_OK;
}

int DeviceGraph::pack(void* y, void* send, size_t elem, hipStream_t st) {
    if (!y || (!send && send_total) || (elem != 8 && elem != 32)) return ET_E_INVALID_ARG;
    if (geo.compact && !send_known) { set_error("this share was not told which of its rows the other shares cite"); return ET_E_UNSUPPORTED; }
    if (!geo.compact) {       // position order: a share's segment is what it sends
        if (send_total) HIP_TRY(hipMemcpyAsync(send, static_cast<const char*>(y) + (size_t)geo.own_off * elem, send_total * elem, hipMemcpyDeviceToDevice, st));
        return ET_OK;
    }
    const uint32_t n = (uint32_t)send_total;
    if (n + self_cnt == 0) return ET_OK;
    if (elem == 8) {
        double* v = static_cast<double*>(y);
        hipLaunchKernelGGL(k_pack_f64, dim3((n + self_cnt + 255) / 256), dim3(256), 0, st, v + geo.own_off, d_send_idx, static_cast<double*>(send), n,
                           d_self_idx, v + self_off, self_cnt);
    } else {
        uint4* v = static_cast<uint4*>(y);
        hipLaunchKernelGGL(k_pack_fr, dim3((2 * (n + self_cnt) + 255) / 256), dim3(256), 0, st, v + 2 * (size_t)geo.own_off, d_send_idx, static_cast<uint4*>(send), n,
                           d_self_idx, v + 2 * (size_t)self_off, self_cnt);
    }
    HIP_TRY(hipGetLastError());
    return ET_OK;
}

int DeviceGraph::record_begin(hipStream_t st) {
    if (ev_count >= EVENT_RING) { set_error("event ring full: call et_graph_event_times"); return ET_E_UNSUPPORTED; }
    if (ev_begin.size() <= ev_count) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        ev_begin.push_back(e0);
        ev_end.push_back(e1);
    }
    HIP_TRY(hipEventRecord((hipEvent_t)ev_begin[ev_count], st));
    return ET_OK;
}
int DeviceGraph::record_end(hipStream_t st) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev_end[ev_count], st));
    ++ev_count;
    return ET_OK;
}
int DeviceGraph::event_times(float* total_ms, uint32_t* pairs) {
    float total = 0.f;
    for (uint32_t i = 0; i < ev_count; ++i) {
        HIP_TRY(hipEventSynchronize((hipEvent_t)ev_end[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)ev_begin[i], (hipEvent_t)ev_end[i]));
        total += ms;
    }
    if (total_ms) *total_ms = total;
    if (pairs) *pairs = ev_count;
    ev_count = 0;
    return ET_OK;
}

// A whole converge.  The blocked sweeps take the first iteration's input from a vector that is written once and kept
// while the initial score stays the same (no init launch per converge), and their last launch writes the scores in slot
// order (no un-permutation launch); slots without a member are zero from the allocation on and never written.
int DeviceGraph::run_f64(uint32_t iters, double initial, double alpha, hipStream_t st, bool record) {
    if (geo.n_seg != 1) { set_error("run needs the whole graph on one GPU"); return ET_E_UNSUPPORTED; }
    const bool fused = iters > 0 && geo.n_chunks == 1 && d_peer && !no_fused_run;
    int rc;
    if (!fused) {
        if ((rc = init_f64(initial, vx[0], st, iters == 0))) return rc;
    } else if (!x0_64 || x0_64_initial != initial) {
        if (!x0_64) {
            uint64_t got = 0;
            if ((rc = dev_alloc(&x0_64, vec_len, got))) return rc;
            device_bytes += got;
        }
        x0_64_initial = std::nan("");
        const double d0 = (double)n_dangling_local * initial;
        hipLaunchKernelGGL(k_init_f64, dim3((geo.seg_stride() + 255) / 256), dim3(256), 0, st, geo, initial, d0, x0_64,
                           (double*)nullptr, raw ? d_scale64 : nullptr);
        HIP_TRY(hipGetLastError());
        x0_64_initial = initial;
    }
    const double* pre = nullptr;
    if (alpha != 0.0) {
        // default pre-trust: the initial vector (plain scores, position order)
        if (!d_pretrust) {
            uint64_t dummy = 0;
            if ((rc = dev_alloc(&d_pretrust, vec_len, dummy))) return rc;
            device_bytes += dummy;
        }
        if ((rc = init_f64(initial, d_pretrust, st, true))) return rc;
        pre = d_pretrust;
    }
    StepExtra ex;
    ex.initial64 = initial;
    ex.out = out_f64;
    if (record && (rc = record_begin(st))) return rc;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t flags = it + 1 == iters ? ET_STEP_FINAL : ET_STEP_NO_STATS;      // a converge reports its last step's figures
        if (!fused) rc = step_f64(vx[it & 1], vx[(it + 1) & 1], it, alpha, pre, flags, st);
        else {
            ex.first_step = it == 0;
            rc = step_f64(it == 0 ? x0_64 : vx[it & 1], vx[(it + 1) & 1], it, alpha, pre, flags, st, -1, &ex);
        }
        if (rc) return rc;
    }
    if (record && (rc = record_end(st))) return rc;
    if (!fused && (rc = unpermute_f64(vx[iters & 1], out_f64, st))) return rc;
    last_iters = iters;
    return ET_OK;
}
int DeviceGraph::run_fr(uint32_t iters, uint64_t lo, uint64_t hi, hipStream_t st, bool record) {
    if (geo.n_seg != 1) { set_error("run needs the whole graph on one GPU"); return ET_E_UNSUPPORTED; }
    const bool fused = iters > 0 && geo.n_chunks == 1 && fr_blocked && raw && d_peer && !no_fused_run;
    int rc;
    const Fr init = Fr::from_u128(lo, hi);
    if (!fused) {
        if ((rc = init_fr(lo, hi, fx[0], st, iters == 0))) return rc;
    } else if (!x0_fr || !x0_fr_valid || x0_fr_lo != lo || x0_fr_hi != hi) {
        if (!x0_fr) {
            uint64_t got = 0;
            if ((rc = dev_alloc(&x0_fr, vec_len * 2, got))) return rc;
            device_bytes += got;
        }
        x0_fr_valid = false;
        const Fr d0 = init * Fr::from_u64(n_dangling_local);
        hipLaunchKernelGGL(k_init_fr, dim3((geo.seg_stride() + 255) / 256), dim3(256), 0, st, geo, to_u256(init.to_raw()), to_u256(d0.to_raw()),
                           x0_fr, (uint4*)nullptr, d_scale_fr);
        HIP_TRY(hipGetLastError());
        x0_fr_lo = lo; x0_fr_hi = hi; x0_fr_valid = true;
    }
    StepExtra ex;
    ex.initial_fr = init.to_raw();
    ex.out = out_fr;
    if (record && (rc = record_begin(st))) return rc;
    for (uint32_t it = 0; it < iters; ++it) {
        // the number of changed rows is asked of the last step only (what a converge reports)
        const uint32_t flags = it + 1 == iters ? ET_STEP_FINAL : ET_STEP_NO_STATS;
        if (!fused) rc = step_fr(fx[it & 1], fx[(it + 1) & 1], it, flags, st);
        else {
            ex.first_step = it == 0;
            rc = step_fr(it == 0 ? x0_fr : fx[it & 1], fx[(it + 1) & 1], it, flags, st, -1, &ex);
        }
        if (rc) return rc;
    }
    if (record && (rc = record_end(st))) return rc;
    if (!fused && (rc = unpermute_fr(fx[iters & 1], out_fr, st))) return rc;
    last_iters = iters;
    return ET_OK;
}

int dev_fr_mul(uint32_t count, const et_fr* a, const et_fr* b, et_fr* out) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    uint4 *da = nullptr, *db = nullptr, *dc = nullptr;
    size_t bytes = (size_t)count * 32;
    HIP_TRY(hipMalloc((void**)&da, bytes ? bytes : 32));
    HIP_TRY(hipMalloc((void**)&db, bytes ? bytes : 32));
    HIP_TRY(hipMalloc((void**)&dc, bytes ? bytes : 32));
    HIP_TRY(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    U256 c576 = to_u256(Fr::one().pow2k(576).to_raw());
    if (count) hipLaunchKernelGGL(k_fr_mul_test, dim3((count + 255) / 256), dim3(256), 0, 0, da, db, dc, c576, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dc, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    return ET_OK;
}

}  // namespace et
