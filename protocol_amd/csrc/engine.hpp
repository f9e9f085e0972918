// Device-resident prepared graph and the launches that iterate it (implementation: engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "builder.hpp"

namespace et {

void set_error(const std::string& s);
const char* last_error();

// Per-XCD cost factors for BuildOpts::xcd_cost, measured once per device and process by a short gather probe (the XCDs
// of a card do not serve L2 misses equally fast; workgroup b of a launch runs on XCD b mod 8).  All 1024 when the
// spread is below 1 %, when ET_XCD_BALANCE=0, or when the probe fails.
void xcd_cost_factors(int device, uint32_t out[8]);

struct DeviceGraph {
    int device = -1;
    int variants = 0;
    uint32_t n = 0, n_valid = 0;
    uint64_t nnz = 0, nnz_local = 0, n_steps = 0;
    uint32_t n_slices = 0, n_work = 0, n_dangling_local = 0;
    uint32_t n_dslots = 0;            // partial-sum slots the row-per-lane exact kernels write (their dangling work items)
    Geometry geo{};
    size_t vec_len = 0;               // entries of the vectors the steps read and write (geo.vec_len)
    size_t full_len = 0;              // n_seg * seg_stride: the position-order vector the un-permutation reads
    uint64_t device_bytes = 0;
    std::vector<uint32_t> pos_of_ext;
    FrRaw inv_nm1_w{};
    double inv_nm1 = 0.0;
    uint32_t last_iters = 0;
    uint32_t xcd_cost[8] = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};   // what the layout was levelled with

    WorkDesc* d_work = nullptr;
    SliceDesc* d_slices = nullptr;
    uint32_t* d_col = nullptr;
    // raw form (integer opinions): edge records keep the opinion (val_bytes = 1 or 4 each), the
    // source factors stay resident and the plain scores of this share's rows live in own64/own_fr;
    // weighted form (field-valued opinions): normalised weights in d_bk_w (double) / d_wlo, d_whi
    bool raw = true;
    int val_bytes = 4;
    void* d_val = nullptr;
    double* d_scale64 = nullptr;
    uint4* d_scale_fr = nullptr;
    double* own64 = nullptr;
    uint4* own_fr = nullptr;
    uint32_t n_f64_parts = 0;
    uint64_t* d_bk_ptr = nullptr;
    uint32_t* d_bk_grp = nullptr;
    uint32_t* d_bk_col = nullptr;
    uint16_t* d_bk_row = nullptr;
    void* d_bk_w = nullptr;
    // byte opinions: d_bk_col / d_bk_row replaced by packed records (slot | source offset << 16) plus one base
    // position per 64 edges
    // where the sweeps' barriers start and how often they come (see sweep_barriers in engine.hip)
    uint32_t f64_bar_period = 2, f64_bar_first_256 = 0, fr_bar_period = 8, fr_bar_first_256 = 0;
    bool packed = false;
    uint32_t f64_rec_shift = 16, fr_rec_shift = 16;   // bits of a packed record that hold the accumulator slot
    uint32_t* d_bk_rec = nullptr;
    uint32_t* d_bk_cbase = nullptr;
    uint64_t* d_bk_cptr = nullptr;
    // the same packing for the exact blocked sweep's lists
    bool fr_packed = false;
    uint32_t* d_fb_rec = nullptr;
    uint32_t* d_fb_cbase = nullptr;
    uint64_t* d_fb_cptr = nullptr;
    // exact blocked sweep (geo.fr_blocks != 0): per-workgroup edge lists like the f64 ones
    bool fr_blocked = false;
    uint32_t n_fr_parts = 0;          // partial statistics per launch: workgroups of the blocked sweep, else wavefront work items
    uint64_t* d_fb_ptr = nullptr;
    uint32_t* d_fb_grp = nullptr;
    uint32_t* d_fb_col = nullptr;
    uint16_t* d_fb_row = nullptr;
    void* d_fb_val = nullptr;         // a byte per edge, or 32 bits (geo.fr_compact == 2)
    uint4* d_fb_scale = nullptr;      // inv(sum) * 2^288 per position
    // chunked exchange (geo.n_chunks > 1): runs of the workgroups' lists per piece, partial sums between an iteration's launches
    uint64_t* d_bk_phase = nullptr;
    uint64_t* d_fb_phase = nullptr;
    double* d_f64_acc = nullptr;
    unsigned long long* d_fr_acc = nullptr;
    uint4* d_wlo = nullptr;
    uint4* d_whi = nullptr;
    uint32_t* d_pos = nullptr;
    double* d_res_part = nullptr;
    double* d_sum_part = nullptr;
    uint4* d_frsum_part = nullptr;
    uint32_t* d_chg_part = nullptr;
    uint4* d_stat_out = nullptr;
    double* vx[2] = {nullptr, nullptr};
    uint4* fx[2] = {nullptr, nullptr};
    double* out_f64 = nullptr;
    uint4* out_fr = nullptr;
    double* d_pretrust = nullptr;
    // a whole converge on one GPU (run_*): slot of the peer at every position, and the first iteration's input vector,
    // kept from converge to converge while the initial score is the same
    uint32_t* d_peer = nullptr;
    double* x0_64 = nullptr;
    double x0_64_initial = 0.0;
    uint4* x0_fr = nullptr;
    uint64_t x0_fr_lo = 0, x0_fr_hi = 0;
    bool x0_fr_valid = false;
    bool no_fused_run = false;        // ET_NO_FUSED_RUN=1: init and un-permutation as launches of their own
    // Gather vector (geo.compact): after a step this share packs, for every other share c, the entries of its own block that
    // c cites followed by its partial sums -- send_cnt[c] entries at send_off[c] of the pack buffer, picked by d_send_idx --
    // and receives recv_cnt[p] entries from share p at recv_off[p] of the vector.  Position order: every share sends its whole
    // segment (send_off 0) and receives segment p at p * seg_stride.
    uint32_t* d_send_idx = nullptr;
    // ... and copies the self_cnt entries its own rows cite (picked by d_self_idx) to self_off of the same vector
    uint32_t* d_self_idx = nullptr;
    uint32_t self_off = 0, self_cnt = 0;
    uint32_t send_off[8] = {}, send_cnt[8] = {}, recv_off[8] = {}, recv_cnt[8] = {};
    uint64_t send_total = 0;
    bool send_known = false;          // gather vector: the rows the other shares cite have been set
    // ring of HIP event pairs bracketing the iteration launches of timed runs
    static const uint32_t EVENT_RING = 1024;
    std::vector<void*> ev_begin, ev_end;
    uint32_t ev_count = 0;

    ~DeviceGraph();
    static int create(const HostGraph& h, int variants, int device, DeviceGraph** out);

    // plain = true writes plain scores whatever the form (a vector that is read back, not iterated)
    int init_f64(double initial, double* x, hipStream_t st, bool plain = false);
    int init_fr(uint64_t lo, uint64_t hi, uint4* x, hipStream_t st, bool plain = false);
    // What run_* tells the blocked sweeps beyond the public flags: the step reads the kept initial vector (the dangling
    // rows' plain scores are then `initial`, not what the last converge left), and the final step writes slot order to out.
    struct StepExtra {
        bool first_step = false;
        double initial64 = 0.0;
        FrRaw initial_fr{};
        void* out = nullptr;
    };
    // phase: which of the geo.n_chunks launches of the iteration to enqueue; -1 = all of them, one after the other
    int step_f64(const double* x, double* y, uint32_t iteration, double alpha, const double* pretrust, uint32_t flags, hipStream_t st, int phase = -1,
                 const StepExtra* ex = nullptr);
    int step_fr(const uint4* x, uint4* y, uint32_t iteration, uint32_t flags, hipStream_t st, int phase = -1, const StepExtra* ex = nullptr);
    int unpermute_f64(const double* x, double* out, hipStream_t st);
    int unpermute_fr(const uint4* x, uint4* out, hipStream_t st);
    int stats_f64(uint32_t iteration, hipStream_t st, et_iter_stats* out);
    int stats_fr(uint32_t iteration, hipStream_t st, et_iter_stats* out);
    int run_f64(uint32_t iters, double initial, double alpha, hipStream_t st, bool record);
    int run_fr(uint32_t iters, uint64_t lo, uint64_t hi, hipStream_t st, bool record);
    // copies the entries the other shares cite out of the own block of y into `send` (elem = 8 or 32 bytes per entry), and
    // the entries the share's own rows cite into their block of y
    int pack(void* y, void* send, size_t elem, hipStream_t st);
    // gather vector: lists[c] = the local rows of this share that share c cites, in c's order (replaces earlier lists)
    int set_send_lists(const std::vector<uint32_t>* const lists[8]);
    int record_begin(hipStream_t st);
    int record_end(hipStream_t st);
    int event_times(float* total_ms, uint32_t* pairs);
};

int dev_fr_mul(uint32_t count, const et_fr* a, const et_fr* b, et_fr* out);

}  // namespace et
