// extern "C" surface of the graph engine (declarations and contracts: include/eigentrust_hip.h).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "builder.hpp"
#include "engine.hpp"
#include "fr_host.hpp"

using namespace et;

static inline DeviceGraph* G(et_graph* g) { return reinterpret_cast<DeviceGraph*>(g); }
static inline const DeviceGraph* G(const et_graph* g) { return reinterpret_cast<const DeviceGraph*>(g); }

#define HIP_TRY_C(expr)                                                              \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));            \
            return _e == hipErrorNoDevice ? ET_E_NO_DEVICE : ET_E_HIP;               \
        }                                                                            \
    } while (0)

extern "C" {

const char* et_strerror(int code) {
    switch (code) {
        case ET_OK: return "ok";
        case ET_E_DUP_MEMBER: return "member already in the set";
        case ET_E_SET_FULL: return "no empty slot left in the set";
        case ET_E_NOT_MEMBER: return "address is not a member of the set";
        case ET_E_INSUFFICIENT_PEERS: return "Insufficient peers for calculation!";
        case ET_E_SUM_MISMATCH: return "sum of scores changed during convergence";
        case ET_E_NONCANONICAL_FR: return "field element is not canonical (>= r)";
        case ET_E_INVALID_ARG: return "invalid argument";
        case ET_E_HIP: return "HIP runtime error";
        case ET_E_NO_DEVICE: return "no HIP device available";
        case ET_E_BUFFER_TOO_SMALL: return "output buffer too small";
        case ET_E_OUT_OF_RANGE: return "index out of range";
        case ET_E_UNSUPPORTED: return "unsupported configuration";
        case ET_E_THRESHOLD_RANGE: return "threshold or decimal limb bound out of range";
        case ET_E_RATIO_DIGITS: return "rational score has more decimal digits than the limbs hold";
        case ET_E_RATIO_MISMATCH: return "rational score does not match the field score";
        case ET_E_ZERO_LIMB: return "denominator (or its top decimal limb) is zero";
        case ET_E_BAD_DOMAIN: return "attestation belongs to another domain";
        case ET_E_RATIONAL_TOO_LARGE: return "rational score does not fit 32 bytes";
        case ET_E_PARSE: return "malformed input";
        case ET_E_BAD_SIGNATURE: return "signature does not recover or verify";
        case ET_E_BAD_ABOUT: return "attestation is about another peer than its slot";
        case ET_E_RCCL: return "RCCL error";
        default: return "unknown error";
    }
}
const char* et_last_error_message(void) { return last_error(); }
uint32_t et_abi_version(void) { return (0u << 16) | 2u; }
int et_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ET_E_HIP; }
    return n;
}

// the caller's per-XCD cost factors, if it gave any (all zero = measure on the device, or uniform without one)
static bool given_xcd_cost(const et_build_opts* opts, BuildOpts& bo) {
    bool any = false;
    for (int x = 0; opts && x < 8; ++x) any = any || opts->xcd_cost[x] != 0;
    for (int x = 0; any && x < 8; ++x) bo.xcd_cost[x] = opts->xcd_cost[x] ? opts->xcd_cost[x] : 1024u;
    return any;
}

static void read_opts(const et_build_opts* opts, BuildOpts& bo, int& variants, int& device) {
    variants = 0;
    device = -1;
    if (!opts) return;
    if (opts->n_seg) bo.n_seg = opts->n_seg;
    bo.seg = opts->seg;
    if (opts->long_row) bo.long_row = opts->long_row;
    if (opts->target_work) bo.target_work = opts->target_work;
    variants = (int)opts->variants;
    device = opts->device;
    if (opts->n_chunks) bo.n_chunks = opts->n_chunks;
    bo.full_vector = (opts->flags & ET_BUILD_FULL_VECTOR) != 0;
    bo.reused = (opts->flags & ET_BUILD_REUSED) != 0;
}

// Tuning knobs for experiments (not part of the ABI): ET_TARGET_WORK, ET_LONG_ROW, ET_F64_BLOCKS, ET_HOT_PREFIX.
static void read_env(BuildOpts& bo) {
    if (const char* e = getenv("ET_TARGET_WORK")) bo.target_work = (uint32_t)atoi(e);
    if (const char* e = getenv("ET_LONG_ROW")) bo.long_row = (uint32_t)atoi(e);
    if (const char* e = getenv("ET_HOT_PREFIX")) bo.hot_prefix = (uint32_t)atoi(e);
    if (const char* e = getenv("ET_F64_BLOCKS")) bo.f64_blocks = (uint32_t)atoi(e);
}

static int plan_build(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                      const uint32_t* value_u32, const et_fr* value_fr, const et_build_opts* opts, bool for_device, et_plan** out);

int et_plan_build(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                  const uint32_t* value_u32, const et_fr* value_fr, const et_build_opts* opts, et_plan** out) {
    return plan_build(n, member, n_edges, from, to, value_u32, value_fr, opts, false, out);
}

static int plan_build(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                      const uint32_t* value_u32, const et_fr* value_fr, const et_build_opts* opts, bool for_device, et_plan** out) {
    if (!out) return ET_E_INVALID_ARG;
    BuildInput in;
    in.n = n;
    in.member = member;
    in.n_edges = n_edges;
    in.from = from;
    in.to = to;
    in.val_u32 = value_u32;
    in.val_fr = reinterpret_cast<const FrRaw*>(value_fr);
    static const uint32_t dummy32 = 0;
    if (n_edges == 0 && !value_u32 && !value_fr) in.val_u32 = &dummy32;
    BuildOpts bo;
    int variants, device;
    read_opts(opts, bo, variants, device);
    read_env(bo);
    if (!given_xcd_cost(opts, bo) && for_device) xcd_cost_factors(device, bo.xcd_cost);
    std::unique_ptr<et_plan> p(new et_plan());
    std::string err;
    int rc = build_graph(in, bo, p->h, err);
    if (rc) { set_error(err); return rc; }
    *out = p.release();
    return ET_OK;
}
void et_plan_free(et_plan* p) { delete p; }

static void fill_info(const HostGraph& h, et_graph_info* info) {
    std::memset(info, 0, sizeof(*info));
    info->n = h.n; info->n_valid = h.n_valid; info->n_dangling = h.geo.n_dangling;
    info->n_seg = h.geo.n_seg; info->seg = h.geo.seg;
    info->nnz = h.nnz; info->nnz_local = h.nnz_local; info->padded_local = h.n_steps * 64;
    info->n_slices = (uint32_t)h.slices.size(); info->n_work = (uint32_t)h.work.size();
    info->rows_cap = h.geo.rows_cap; info->nd_cap = h.geo.nd_cap;
    info->vec_len = h.geo.vec_len;
    info->full_len = (uint64_t)h.geo.n_seg * h.geo.seg_stride();
    info->gather = h.geo.compact;
    info->n_chunks = h.geo.n_chunks; info->chunk_rows = h.geo.chunk_rows;
    for (int x = 0; x < 8; ++x) info->xcd_cost[x] = h.xcd_cost[x];
}
int et_plan_info_get(const et_plan* p, et_graph_info* info) {
    if (!p || !info) return ET_E_INVALID_ARG;
    fill_info(p->h, info);
    return ET_OK;
}
int et_plan_arrays_get(const et_plan* p, et_plan_arrays* out) {
    if (!p || !out) return ET_E_INVALID_ARG;
    const HostGraph& h = p->h;
    static_assert(sizeof(et_slice_desc) == sizeof(SliceDesc) && sizeof(et_work_desc) == sizeof(WorkDesc), "layout");
    out->pos_of_slot = h.pos_of_ext.data();
    out->slices = reinterpret_cast<const et_slice_desc*>(h.slices.data());
    out->work = reinterpret_cast<const et_work_desc*>(h.work.data());
    out->col = h.col.data();
    out->value_u32 = h.values_are_u32 ? h.val_u32.data() : nullptr;
    out->value_fr = h.values_are_u32 ? nullptr : reinterpret_cast<const et_fr*>(h.val_fr.data());
    out->scale_fr = reinterpret_cast<const et_fr*>(h.scale_fr.data());
    out->scale_f64 = h.scale_f64.data();
    std::memcpy(out->inv_nm1_w.b, h.inv_nm1_w.l, 32);
    out->n_active = h.geo.n_active; out->n_active_pad = h.geo.n_active_pad;
    out->n_dangling_pad = h.geo.n_dangling_pad; out->n_dangling_local = h.n_dangling_local;
    return ET_OK;
}
int et_plan_sweep_stats(const et_plan* p, int variant, et_sweep_stats* out) {
    if (!p || !out || (variant != ET_VARIANT_FR && variant != ET_VARIANT_F64)) return ET_E_INVALID_ARG;
    const SweepStats ss = sweep_stats(p->h, variant == ET_VARIANT_FR);
    out->blocks = ss.blocks; out->clustered = ss.clustered;
    out->edges = ss.edges; out->edges_most = ss.edges_most;
    out->line_fills = ss.line_fills; out->line_fills_most = ss.line_fills_most;
    out->chunk_span_most = ss.chunk_span_most;
    out->xcd_line_fills = ss.xcd_line_fills;
    return ET_OK;
}
int et_graph_from_plan(const et_plan* p, const et_build_opts* opts, et_graph** out) {
    if (!p || !out) return ET_E_INVALID_ARG;
    DeviceGraph* dg = nullptr;
    int rc = DeviceGraph::create(p->h, opts ? (int)opts->variants : 0, opts ? opts->device : -1, &dg);
    if (rc) return rc;
    *out = reinterpret_cast<et_graph*>(dg);
    return ET_OK;
}

int et_graph_build(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                   const uint32_t* value_u32, const et_fr* value_fr, const et_build_opts* opts, et_graph** out) {
    if (!out) return ET_E_INVALID_ARG;
    et_plan* p = nullptr;
    int rc = plan_build(n, member, n_edges, from, to, value_u32, value_fr, opts, true, &p);   // levelled for this card's XCDs
    if (rc) return rc;
    rc = et_graph_from_plan(p, opts, out);
    et_plan_free(p);
    return rc;
}

void et_graph_free(et_graph* g) { delete G(g); }

int et_graph_info_get(const et_graph* g, et_graph_info* info) {
    if (!g || !info) return ET_E_INVALID_ARG;
    const DeviceGraph* d = G(g);
    std::memset(info, 0, sizeof(*info));
    info->n = d->n; info->n_valid = d->n_valid; info->n_dangling = d->geo.n_dangling;
    info->n_seg = d->geo.n_seg; info->seg = d->geo.seg;
    info->nnz = d->nnz; info->nnz_local = d->nnz_local; info->padded_local = d->n_steps * 64;
    info->n_slices = d->n_slices; info->n_work = d->n_work;
    info->rows_cap = d->geo.rows_cap; info->nd_cap = d->geo.nd_cap;
    info->vec_len = d->vec_len; info->device_bytes = d->device_bytes;
    info->f64_sweep_blocks = d->n_f64_parts;
    info->exact_sweep_blocks = d->fr_blocked ? d->geo.fr_blocks : 0;
    info->n_chunks = d->geo.n_chunks; info->chunk_rows = d->geo.chunk_rows;
    info->gather = d->geo.compact; info->full_len = d->full_len;
    for (int x = 0; x < 8; ++x) info->xcd_cost[x] = d->xcd_cost[x];
    return ET_OK;
}

static void exchange_common(const Geometry& geo, et_exchange_info* out) {
    std::memset(out, 0, sizeof(*out));
    out->gather = geo.compact; out->n_seg = geo.n_seg; out->seg = geo.seg;
    out->own_off = geo.own_off; out->own_len = geo.seg_stride();
    out->vec_len = geo.vec_len; out->full_len = (uint64_t)geo.n_seg * geo.seg_stride();
}
int et_graph_exchange_get(const et_graph* g, et_exchange_info* out) {
    if (!g || !out) return ET_E_INVALID_ARG;
    const DeviceGraph* d = G(g);
    exchange_common(d->geo, out);
    for (int c = 0; c < 8; ++c) {
        out->recv_off[c] = d->recv_off[c]; out->recv_count[c] = d->recv_cnt[c];
        out->send_off[c] = d->send_off[c]; out->send_count[c] = d->send_cnt[c];
    }
    out->send_total = d->send_total;
    out->self_off = d->self_off; out->self_count = d->self_cnt;
    return ET_OK;
}
int et_plan_exchange_get(const et_plan* p, et_exchange_info* out, const uint32_t** need, const uint32_t** send) {
    if (!p || !out) return ET_E_INVALID_ARG;
    const HostGraph& h = p->h;
    const Geometry& geo = h.geo;
    exchange_common(geo, out);
    uint32_t at = 0;
    for (uint32_t c = 0; c < 8; ++c) {
        if (need) need[c] = nullptr;
        if (send) send[c] = nullptr;
        if (c >= geo.n_seg) continue;
        if (c == geo.seg) {           // the share's own cited rows: nothing travels, the block is filled by the pack
            if (geo.compact) {
                out->self_off = h.blk_off[c]; out->self_count = (uint32_t)h.need[c].size();
                if (need) need[c] = h.need[c].data();
            }
            continue;
        }
        if (geo.compact) {
            out->recv_off[c] = h.blk_off[c]; out->recv_count[c] = (uint32_t)h.need[c].size() + geo.nd_cap;
            out->send_off[c] = at; out->send_count[c] = (uint32_t)h.send[c].size() + geo.nd_cap;
            at += out->send_count[c];
            if (need) need[c] = h.need[c].data();
            if (send) send[c] = h.send[c].data();
        } else {
            out->recv_off[c] = c * geo.seg_stride(); out->recv_count[c] = geo.seg_stride();
            out->send_off[c] = 0; out->send_count[c] = geo.seg_stride();
            at = geo.seg_stride();
        }
    }
    out->send_total = at;
    return ET_OK;
}
int et_graph_pack_fr_async(et_graph* g, void* y, void* send, void* stream) {
    if (!g || !y || !send) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->pack(y, send, 32, (hipStream_t)stream);
}
int et_graph_pack_f64_async(et_graph* g, void* y, void* send, void* stream) {
    if (!g || !y || !send) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->pack(y, send, 8, (hipStream_t)stream);
}

int et_graph_converge_fr(et_graph* g, uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi, et_fr* out) {
    if (!g || !out) return ET_E_INVALID_ARG;
    DeviceGraph* d = G(g);
    HIP_TRY_C(hipSetDevice(d->device));
    int rc = d->run_fr(num_iterations, initial_lo, initial_hi, nullptr, false);
    if (rc) return rc;
    if ((rc = et_graph_result_fr(g, out, nullptr))) return rc;
    // conservation (native.rs:331-334): sum of the initial scores == sum of the final scores
    Fr expect = Fr::from_u128(initial_lo, initial_hi) * Fr::from_u64(d->n_valid);
    if (num_iterations > 0) {
        et_iter_stats st;
        if ((rc = d->stats_fr(num_iterations - 1, nullptr, &st))) return rc;
        FrRaw got;
        std::memcpy(got.l, st.sum.b, 32);
        if (got != expect.to_raw()) { set_error("reputation leak: field sum of scores changed"); return ET_E_SUM_MISMATCH; }
    }
    return ET_OK;
}

int et_graph_converge_f64(et_graph* g, uint32_t num_iterations, double initial, double alpha, const double* pretrust,
                          double* out, double* residual_l1) {
    if (!g || !out) return ET_E_INVALID_ARG;
    DeviceGraph* d = G(g);
    HIP_TRY_C(hipSetDevice(d->device));
    if (d->geo.n_seg != 1) { set_error("converge needs the whole graph on one GPU"); return ET_E_UNSUPPORTED; }
    int rc;
    if (pretrust && alpha != 0.0) {
        std::vector<double> pos(d->vec_len);
        if ((rc = et_graph_permute_f64_host(g, pretrust, pos.data()))) return rc;
        if (!d->d_pretrust) {
            HIP_TRY_C(hipMalloc((void**)&d->d_pretrust, d->vec_len * sizeof(double)));
            d->device_bytes += d->vec_len * sizeof(double);
        }
        HIP_TRY_C(hipMemcpy(d->d_pretrust, pos.data(), d->vec_len * sizeof(double), hipMemcpyHostToDevice));
        if ((rc = d->init_f64(initial, d->vx[0], nullptr, num_iterations == 0))) return rc;
        for (uint32_t it = 0; it < num_iterations; ++it)
            if ((rc = d->step_f64(d->vx[it & 1], d->vx[(it + 1) & 1], it, alpha, d->d_pretrust,
                                  it + 1 == num_iterations ? ET_STEP_FINAL : ET_STEP_NO_STATS, nullptr))) return rc;
        if ((rc = d->unpermute_f64(d->vx[num_iterations & 1], d->out_f64, nullptr))) return rc;
    } else {
        if ((rc = d->run_f64(num_iterations, initial, alpha, nullptr, false))) return rc;
    }
    if ((rc = et_graph_result_f64(g, out, nullptr))) return rc;
    if (residual_l1) {
        *residual_l1 = 0.0;
        if (num_iterations > 0) {
            et_iter_stats st;
            if ((rc = d->stats_f64(num_iterations - 1, nullptr, &st))) return rc;
            *residual_l1 = st.residual_l1;
        }
    }
    return ET_OK;
}

int et_graph_init_fr_async(et_graph* g, uint64_t lo, uint64_t hi, void* x, uint32_t flags, void* stream) {
    if (!g || !x) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->init_fr(lo, hi, (uint4*)x, (hipStream_t)stream, (flags & ET_INIT_PLAIN) != 0);
}
int et_graph_init_f64_async(et_graph* g, double initial, void* x, uint32_t flags, void* stream) {
    if (!g || !x) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->init_f64(initial, (double*)x, (hipStream_t)stream, (flags & ET_INIT_PLAIN) != 0);
}
int et_graph_step_fr_async(et_graph* g, const void* x, void* y, uint32_t iteration, uint32_t flags, void* stream) {
    if (!g || !x || !y) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->step_fr((const uint4*)x, (uint4*)y, iteration, flags, (hipStream_t)stream);
}
int et_graph_step_f64_async(et_graph* g, const void* x, void* y, uint32_t iteration, double alpha,
                            const void* pretrust_pos, uint32_t flags, void* stream) {
    if (!g || !x || !y) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->step_f64((const double*)x, (double*)y, iteration, alpha, (const double*)pretrust_pos, flags, (hipStream_t)stream);
}
int et_graph_step_fr_phase_async(et_graph* g, const void* x, void* y, uint32_t iteration, uint32_t flags, uint32_t phase, void* stream) {
    if (!g || !x || !y) return ET_E_INVALID_ARG;
    if (phase >= G(g)->geo.n_chunks) { set_error("no such phase"); return ET_E_OUT_OF_RANGE; }
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->step_fr((const uint4*)x, (uint4*)y, iteration, flags, (hipStream_t)stream, (int)phase);
}
int et_graph_step_f64_phase_async(et_graph* g, const void* x, void* y, uint32_t iteration, double alpha,
                                  const void* pretrust_pos, uint32_t flags, uint32_t phase, void* stream) {
    if (!g || !x || !y) return ET_E_INVALID_ARG;
    if (phase >= G(g)->geo.n_chunks) { set_error("no such phase"); return ET_E_OUT_OF_RANGE; }
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->step_f64((const double*)x, (double*)y, iteration, alpha, (const double*)pretrust_pos, flags, (hipStream_t)stream, (int)phase);
}
int et_graph_unpermute_fr_async(et_graph* g, const void* x, void* out_slots, void* stream) {
    if (!g || !x || !out_slots) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->unpermute_fr((const uint4*)x, (uint4*)out_slots, (hipStream_t)stream);
}
int et_graph_unpermute_f64_async(et_graph* g, const void* x, void* out_slots, void* stream) {
    if (!g || !x || !out_slots) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->unpermute_f64((const double*)x, (double*)out_slots, (hipStream_t)stream);
}
int et_graph_permute_f64_host(const et_graph* g, const double* slots, double* pos) {
    if (!g || !slots || !pos) return ET_E_INVALID_ARG;
    const DeviceGraph* d = G(g);
    std::memset(pos, 0, d->vec_len * sizeof(double));
    const uint32_t stride = d->geo.seg_stride();
    for (uint32_t i = 0; i < d->n; ++i) {
        const uint32_t p = d->pos_of_ext[i];
        if (p == 0xFFFFFFFFu) continue;
        if (!d->geo.compact) pos[p] = slots[i];
        else if (p / stride == d->geo.seg) pos[p % stride] = slots[i];     // gather vector: the own block comes first
    }
    return ET_OK;
}
int et_graph_stats_fr(et_graph* g, uint32_t iteration, void* stream, et_iter_stats* out) {
    if (!g || !out) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->stats_fr(iteration, (hipStream_t)stream, out);
}
int et_graph_stats_f64(et_graph* g, uint32_t iteration, void* stream, et_iter_stats* out) {
    if (!g || !out) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->stats_f64(iteration, (hipStream_t)stream, out);
}
int et_graph_run_fr(et_graph* g, uint32_t num_iterations, uint64_t lo, uint64_t hi, void* stream, int record_events) {
    if (!g) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->run_fr(num_iterations, lo, hi, (hipStream_t)stream, record_events != 0);
}
int et_graph_run_f64(et_graph* g, uint32_t num_iterations, double initial, double alpha, void* stream, int record_events) {
    if (!g) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->run_f64(num_iterations, initial, alpha, (hipStream_t)stream, record_events != 0);
}
int et_graph_record_begin(et_graph* g, void* stream) {
    if (!g) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->record_begin((hipStream_t)stream);
}
int et_graph_record_end(et_graph* g, void* stream) {
    if (!g) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->record_end((hipStream_t)stream);
}
int et_graph_event_times(et_graph* g, float* total_ms, uint32_t* pairs) {
    if (!g) return ET_E_INVALID_ARG;
    HIP_TRY_C(hipSetDevice(G(g)->device));
    return G(g)->event_times(total_ms, pairs);
}
int et_graph_result_fr(et_graph* g, et_fr* out, void* stream) {
    if (!g || !out) return ET_E_INVALID_ARG;
    DeviceGraph* d = G(g);
    HIP_TRY_C(hipSetDevice(d->device));
    if (!d->out_fr) return ET_E_UNSUPPORTED;
    HIP_TRY_C(hipMemcpyAsync(out, d->out_fr, (size_t)d->n * 32, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY_C(hipStreamSynchronize((hipStream_t)stream));
    return ET_OK;
}
int et_graph_result_f64(et_graph* g, double* out, void* stream) {
    if (!g || !out) return ET_E_INVALID_ARG;
    DeviceGraph* d = G(g);
    HIP_TRY_C(hipSetDevice(d->device));
    if (!d->out_f64) return ET_E_UNSUPPORTED;
    HIP_TRY_C(hipMemcpyAsync(out, d->out_f64, (size_t)d->n * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY_C(hipStreamSynchronize((hipStream_t)stream));
    return ET_OK;
}

int et_dev_fr_mul(uint32_t count, const et_fr* a, const et_fr* b, et_fr* out) {
    if (count && (!a || !b || !out)) return ET_E_INVALID_ARG;
    for (uint32_t i = 0; i < count; ++i) {
        FrRaw x, y;
        std::memcpy(x.l, a[i].b, 32);
        std::memcpy(y.l, b[i].b, 32);
        if (!fr_raw_is_canonical(x.l) || !fr_raw_is_canonical(y.l)) return ET_E_NONCANONICAL_FR;
    }
    return dev_fr_mul(count, a, b, out);
}

}  // extern "C"
