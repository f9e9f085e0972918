// A share laid out by a process that holds only the attestations of its own attesters (et_dplan_* in
// include/eigentrust_hip.h): several processes ingest disjoint sets of attesters, agree on per-peer statistics by summing
// them, derive the same order and geometry, send every surviving edge to the share that owns its destination row, and lay
// out their own share from what they received.  No process ever holds the whole attestation list.
//
// The filter, the last-write-wins rule, the row sums and the dangling test are the ones of the one-process plan
// (builder.cpp; dynamic_sets/native.rs:234-283, :305-314, eigentrust/src/lib.rs:398-416): they only need an attester's own
// list, which is why all attestations of an attester must be given to the same process.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "builder.hpp"
#include "engine.hpp"

using namespace et;

struct et_dplan {
    uint32_t n = 0;
    std::vector<uint8_t> member;         // empty = every slot holds a peer
    PlanCommon pc;                       // the local filter's results (stage 1); pc.g becomes the common part (stage 2)
    RowSums sums;
    PeerStats local;
    std::vector<uint32_t> from, to, val; // copies of the local attestations (the caller's arrays need not outlive stage 1)
    BuildOpts opt;
    int variants = 0, device = -1;
    bool xcd_given = false;              // the caller named the per-XCD cost factors: no probe
    int stage = 1;
    // stage 2: records by destination share
    std::vector<uint64_t> out_begin;     // n_seg + 1
    std::vector<uint32_t> out_src, out_row, out_val;
    // stage 2: records received
    std::vector<uint32_t> in_src, in_row, in_val;
    // stage 3 (shares with a gather vector): the records sorted, g.need known, g.send being collected
    SortedRecords* sorted = nullptr;
    std::vector<uint8_t> told;           // which shares have said what they cite here
    ~et_dplan() { free_sorted_records(sorted); }
};

// surviving attestations per attester, whatever their values (a peer whose opinions here sum to zero still wrote them here)
static void count_attested(const et_dplan* p, uint32_t* out) {
    const std::vector<uint64_t>& out_ptr = p->pc.out_ptr;
    for (uint32_t f = 0; f < p->n; ++f) {
        uint32_t c = 0;
        for (uint64_t k = out_ptr[f]; k < out_ptr[f + 1]; ++k) c += p->pc.drop[k] ? 0u : 1u;
        out[f] = c;
    }
}

extern "C" {

int et_dplan_create(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                    const uint32_t* value_u32, et_dplan** out) {
    if (!out || n == 0 || (n_edges && (!from || !to || !value_u32))) return ET_E_INVALID_ARG;
    std::unique_ptr<et_dplan> p(new et_dplan());
    p->n = n;
    if (member) p->member.assign(member, member + n);
    p->from.assign(from, from + n_edges);
    p->to.assign(to, to + n_edges);
    p->val.assign(value_u32, value_u32 + n_edges);
    static const uint32_t dummy32 = 0;
    BuildInput in;
    in.n = n;
    in.member = member ? p->member.data() : nullptr;
    in.n_edges = n_edges;
    in.from = p->from.data();
    in.to = p->to.data();
    in.val_u32 = n_edges ? p->val.data() : &dummy32;
    std::string err;
    int rc = filter_edges(in, p->pc, p->sums, p->local, err);
    if (rc) { set_error(err); return rc; }
    *out = p.release();
    return ET_OK;
}

int et_dplan_local_stats(const et_dplan* p, uint64_t* row_sum, uint32_t* in_degree, uint32_t* out_degree, uint32_t* attested,
                         uint32_t* value_max) {
    if (!p || !row_sum || !in_degree || !out_degree || !attested || !value_max) return ET_E_INVALID_ARG;
    if (p->stage != 1) { set_error("local statistics are gone once the global ones were set"); return ET_E_INVALID_ARG; }
    std::memcpy(row_sum, p->sums.u64.data(), (size_t)p->n * sizeof(uint64_t));
    std::memcpy(in_degree, p->local.in_deg.data(), (size_t)p->n * sizeof(uint32_t));
    std::memcpy(out_degree, p->local.out_deg.data(), (size_t)p->n * sizeof(uint32_t));
    count_attested(p, attested);
    *value_max = p->local.value_max;
    return ET_OK;
}

int et_dplan_set_global_stats(et_dplan* p, const uint64_t* row_sum, const uint32_t* in_degree, const uint32_t* out_degree,
                              const uint32_t* attested, uint32_t value_max, const et_build_opts* opts) {
    if (!p || !row_sum || !in_degree || !out_degree || !attested || !opts) return ET_E_INVALID_ARG;
    if (p->stage != 1) { set_error("global statistics were already set"); return ET_E_INVALID_ARG; }
    const uint32_t n = p->n;
    BuildOpts bo;
    if (opts->n_seg) bo.n_seg = opts->n_seg;
    bo.seg = opts->seg;
    if (opts->long_row) bo.long_row = opts->long_row;
    if (opts->target_work) bo.target_work = opts->target_work;
    if (opts->n_chunks) bo.n_chunks = opts->n_chunks;
    bo.full_vector = (opts->flags & ET_BUILD_FULL_VECTOR) != 0;
    for (int x = 0; x < 8; ++x) p->xcd_given = p->xcd_given || opts->xcd_cost[x] != 0;
    for (int x = 0; p->xcd_given && x < 8; ++x) bo.xcd_cost[x] = opts->xcd_cost[x] ? opts->xcd_cost[x] : 1024u;
    if (bo.seg >= bo.n_seg) { set_error("bad segment"); return ET_E_INVALID_ARG; }
    p->opt = bo;
    p->variants = (int)opts->variants;
    p->device = opts->device;
    if (value_max < p->local.value_max) {
        // a maximum that is too small would pick byte-wide edge records for opinions that do not fit them
        set_error("value_max is smaller than this process's own largest opinion: take the maximum over all processes");
        return ET_E_INVALID_ARG;
    }
    // every attestation of an attester must have come through one process: then the summed count of this process's
    // attesters is its own (zero-valued attestations count too: an attester whose opinions here sum to zero is dangling here
    // but must not have other opinions elsewhere)
    {
        std::vector<uint32_t> mine(n);
        count_attested(p, mine.data());
        for (uint32_t i = 0; i < n; ++i)
            if (mine[i] != 0 && (attested[i] != mine[i] || out_degree[i] != p->local.out_deg[i] || row_sum[i] != p->sums.u64[i])) {
                set_error("the attestations of one attester were given to more than one process");
                return ET_E_INVALID_ARG;
            }
    }
    HostGraph& g = p->pc.g;
    const uint8_t* member = p->member.empty() ? nullptr : p->member.data();
    auto is_member = [&](uint32_t i) { return member ? member[i] != 0 : true; };
    std::vector<uint8_t> dangling(n, 0);
    uint64_t nnz = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (is_member(i) && row_sum[i] == 0) dangling[i] = 1;
        nnz += out_degree[i];
    }
    g.nnz = nnz;
    PeerStats st;
    st.in_deg.assign(in_degree, in_degree + n);
    st.out_deg.assign(out_degree, out_degree + n);
    st.value_max = value_max;
    std::string err;
    int rc = layout_common(n, member, true, dangling, st, bo, g, p->pc.order, err);
    if (rc) { set_error(err); return rc; }
    const Geometry geo = g.geo;
    const uint32_t stride = geo.seg_stride();
    // factors of this share's own rows only: a step scales the rows it owns
    {
        std::vector<uint32_t> own;
        own.reserve(geo.n_active / geo.n_seg + 64);
        for (uint32_t r = 0; r < geo.n_active; ++r)
            if (g.pos_of_ext[p->pc.order[r]] / stride == bo.seg) own.push_back(p->pc.order[r]);
        fill_factors_u64(g, own.data(), (uint32_t)own.size(), row_sum);
    }
    // the local surviving edges as records, grouped by the share that owns the destination row
    const std::vector<uint64_t>& out_ptr = p->pc.out_ptr;
    const std::vector<uint64_t>& kept = p->pc.kept;
    const std::vector<uint8_t>& drop = p->pc.drop;
    std::vector<uint64_t> cnt(geo.n_seg + 1, 0);
    for (uint32_t f = 0; f < n; ++f) {
        if (p->pc.dangling[f]) continue;                 // locally dangling: no opinions here, or opinions that sum to zero
        for (uint64_t k = out_ptr[f]; k < out_ptr[f + 1]; ++k)
            if (!drop[k]) ++cnt[g.pos_of_ext[p->to[kept[k]]] / stride + 1];
    }
    for (uint32_t s = 0; s < geo.n_seg; ++s) cnt[s + 1] += cnt[s];
    p->out_begin = cnt;
    p->out_src.resize(cnt[geo.n_seg]);
    p->out_row.resize(cnt[geo.n_seg]);
    p->out_val.resize(cnt[geo.n_seg]);
    std::vector<uint64_t> cur(cnt.begin(), cnt.end() - 1);
    for (uint32_t f = 0; f < n; ++f) {
        if (p->pc.dangling[f]) continue;
        const uint32_t pf = g.pos_of_ext[f];
        const uint32_t code = ((pf / stride) << COL_SEG_SHIFT) | (pf % stride);
        for (uint64_t k = out_ptr[f]; k < out_ptr[f + 1]; ++k) {
            if (drop[k]) continue;
            const uint64_t e = kept[k];
            const uint32_t pt = g.pos_of_ext[p->to[e]];
            const uint64_t q = cur[pt / stride]++;
            p->out_src[q] = code;
            p->out_row[q] = pt % stride;
            p->out_val[q] = p->val[e];
        }
    }
    // the attestation list and the filter's tables are not needed any more
    p->from = std::vector<uint32_t>(); p->to = std::vector<uint32_t>(); p->val = std::vector<uint32_t>();
    p->pc.out_ptr = std::vector<uint64_t>(); p->pc.kept = std::vector<uint64_t>(); p->pc.drop = std::vector<uint8_t>();
    p->pc.order = std::vector<uint32_t>();
    p->sums = RowSums();
    p->local = PeerStats();
    p->stage = 2;
    return ET_OK;
}

int et_dplan_outgoing(const et_dplan* p, uint32_t share, uint64_t* count, const uint32_t** src, const uint32_t** row, const uint32_t** value) {
    if (!p || !count || !src || !row || !value) return ET_E_INVALID_ARG;
    if (p->stage != 2) { set_error("records exist between set_global_stats and finish"); return ET_E_INVALID_ARG; }
    if (share >= p->pc.g.geo.n_seg) return ET_E_OUT_OF_RANGE;
    const uint64_t b = p->out_begin[share];
    *count = p->out_begin[share + 1] - b;
    *src = p->out_src.data() + b;
    *row = p->out_row.data() + b;
    *value = p->out_val.data() + b;
    return ET_OK;
}

int et_dplan_add_incoming(et_dplan* p, uint64_t count, const uint32_t* src, const uint32_t* row, const uint32_t* value) {
    if (!p || (count && (!src || !row || !value))) return ET_E_INVALID_ARG;
    if (p->stage != 2) { set_error("records are taken between set_global_stats and finish"); return ET_E_INVALID_ARG; }
    p->in_src.insert(p->in_src.end(), src, src + count);
    p->in_row.insert(p->in_row.end(), row, row + count);
    p->in_val.insert(p->in_val.end(), value, value + count);
    return ET_OK;
}

int et_dplan_info_get(const et_dplan* p, et_graph_info* info) {
    if (!p || !info) return ET_E_INVALID_ARG;
    if (p->stage < 2) { set_error("the geometry exists once the global statistics were set"); return ET_E_INVALID_ARG; }
    const HostGraph& h = p->pc.g;
    std::memset(info, 0, sizeof(*info));
    info->n = h.n; info->n_valid = h.n_valid; info->n_dangling = h.geo.n_dangling;
    info->n_seg = h.geo.n_seg; info->seg = h.geo.seg; info->nnz = h.nnz;
    info->rows_cap = h.geo.rows_cap; info->nd_cap = h.geo.nd_cap;
    info->vec_len = h.geo.vec_len;            // with a gather vector: the own block only until et_dplan_needs has run
    info->full_len = (uint64_t)h.geo.n_seg * h.geo.seg_stride();
    info->gather = h.geo.compact;
    info->n_chunks = h.geo.n_chunks; info->chunk_rows = h.geo.chunk_rows;
    for (int x = 0; x < 8; ++x) info->xcd_cost[x] = p->opt.xcd_cost[x];
    return ET_OK;
}

// stage 2 -> 3: no more records; sorts them and finds the rows of the other shares this share cites
static int seal_records(et_dplan* p) {
    if (p->stage == 3) return ET_OK;
    if (p->stage != 2) { set_error("the records are complete between set_global_stats and finish"); return ET_E_INVALID_ARG; }
    p->out_src = std::vector<uint32_t>(); p->out_row = std::vector<uint32_t>(); p->out_val = std::vector<uint32_t>();
    // the deal levels edges x the XCDs' cost: measured now where the caller named the device the share will run on
    if (!p->xcd_given && p->device >= 0) xcd_cost_factors(p->device, p->opt.xcd_cost);
    p->opt.cluster_cited = true;         // the order this share asks in reaches the others through et_dplan_needs
    std::string err;
    int rc = records_to_needs(p->pc.g, p->opt, p->opt.seg, p->in_src, p->in_row, p->in_val, &p->sorted, err);
    if (rc) { set_error(err); return rc; }
    p->in_src = std::vector<uint32_t>(); p->in_row = std::vector<uint32_t>(); p->in_val = std::vector<uint32_t>();
    p->told.assign(p->pc.g.geo.n_seg, 0);
    p->told[p->opt.seg] = 1;
    p->stage = 3;
    return ET_OK;
}

int et_dplan_needs(et_dplan* p, uint32_t share, uint64_t* count, const uint32_t** rows) {
    if (!p || !count || !rows) return ET_E_INVALID_ARG;
    int rc = seal_records(p);
    if (rc) return rc;
    if (share >= p->pc.g.geo.n_seg) return ET_E_OUT_OF_RANGE;
    *count = p->pc.g.need[share].size();
    *rows = p->pc.g.need[share].data();
    return ET_OK;
}

int et_dplan_add_wanted(et_dplan* p, uint32_t share, uint64_t count, const uint32_t* rows) {
    if (!p || (count && !rows)) return ET_E_INVALID_ARG;
    int rc = seal_records(p);
    if (rc) return rc;
    HostGraph& g = p->pc.g;
    if (share >= g.geo.n_seg) return ET_E_OUT_OF_RANGE;
    if (share == p->opt.seg || !g.geo.compact) {
        if (count == 0) return ET_OK;
        set_error("this share packs nothing for that share");
        return ET_E_INVALID_ARG;
    }
    for (uint64_t k = 0; k < count; ++k) if (rows[k] >= g.geo.rows_cap) { set_error("cited row out of range"); return ET_E_OUT_OF_RANGE; }
    g.send[share].assign(rows, rows + count);
    p->told[share] = 1;
    return ET_OK;
}

static int lay_out(et_dplan* p, bool for_device) {
    int rc = seal_records(p);
    if (rc) return rc;
    HostGraph& g = p->pc.g;
    if (g.geo.compact) {
        for (uint32_t c = 0; c < g.geo.n_seg; ++c)
            if (!p->told[c]) { set_error("et_dplan_add_wanted has not been called for every other share"); return ET_E_INVALID_ARG; }
        g.send_known = true;
    }
    (void)for_device;
    std::string err;
    rc = plan_segment_from_sorted(g, p->opt, p->sorted, err);
    if (rc) { set_error(err); return rc; }
    free_sorted_records(p->sorted);
    p->sorted = nullptr;
    p->stage = 4;
    return ET_OK;
}

int et_dplan_finish(et_dplan* p, et_graph** out) {
    if (!p || !out) return ET_E_INVALID_ARG;
    int rc = lay_out(p, true);
    if (rc) return rc;
    DeviceGraph* dg = nullptr;
    rc = DeviceGraph::create(p->pc.g, p->variants, p->device, &dg);
    if (rc) return rc;
    *out = reinterpret_cast<et_graph*>(dg);
    return ET_OK;
}

int et_dplan_finish_host(et_dplan* p, et_plan** out) {
    if (!p || !out) return ET_E_INVALID_ARG;
    int rc = lay_out(p, false);
    if (rc) return rc;
    std::unique_ptr<et_plan> plan(new et_plan());
    plan->h = std::move(p->pc.g);
    *out = plan.release();
    return ET_OK;
}

void et_dplan_free(et_dplan* p) { delete p; }

}  // extern "C"
