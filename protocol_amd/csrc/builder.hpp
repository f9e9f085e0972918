// Graph builder: attestation list -> filtered, transposed, sliced edge stream for one GPU.
//
// Implements, in sparse form, what the reference does densely before its iteration loop:
//   * filter_peers_ops (dynamic_sets/native.rs:234-283): opinions about empty slots and about
//     oneself are dropped; a peer whose remaining opinions sum to zero IN THE FIELD gives every
//     other valid peer the opinion 1 ("dangling" here; evaluated by the kernels as a rank-1 term
//     instead of materialising the row of ones);
//   * the row sums and their inverses of native.rs:305-314 (the per-edge multiply is done on the GPU
//     from the per-source factors computed here);
//   * for the f64 variant the integer row sums of converge_rational (native.rs:367-377).
// A later attestation for the same (from, to) pair replaces an earlier one, as
// attestation_matrix[origin][dest] = Some(att) does in eigentrust/src/lib.rs:398-416.
#pragma once
#include <cstdint>
#include <string>
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

#include "fr_host.hpp"
#include "layout.hpp"

namespace et {

struct BuildInput {
    uint32_t n = 0;                    // slots (peers incl. empty slots)
    const uint8_t* member = nullptr;   // n flags, non-zero = slot holds a peer; nullptr = all members
    uint64_t n_edges = 0;
    const uint32_t* from = nullptr;    // attester slot
    const uint32_t* to = nullptr;      // rated slot
    const uint32_t* val_u32 = nullptr; // either small integer opinions ...
    const FrRaw* val_fr = nullptr;     // ... or canonical field elements
};

struct BuildOpts {
    uint32_t n_seg = 1;          // GPUs sharing the rows
    uint32_t seg = 0;            // this GPU
    uint32_t long_row = 128;     // in-degree above which a row is spread over a whole wavefront
    uint32_t target_work = 16384; // wavefront work items per iteration launch (several rounds of 256 CUs x 16 waves)
    uint32_t hot_prefix = 16384; // the most-attesting (most gathered) peers take the first positions: one contiguous prefix of the vector
    uint32_t f64_blocks = 256;   // workgroups of the f64 variant (one per CU)
    uint32_t n_chunks = 1;       // pieces a segment is exchanged in (see Geometry::n_chunks); honoured where both variants run the blocked sweep
    // gather vectors: the other shares' entries ordered by the workgroup that cites them (order_cited_sources in builder.cpp)
    // instead of ascending.  The shares that pack for this one must then be told the order (HostGraph::need): set by the
    // callers that deliver it (the one-thread driver, the rank-local plan).
    bool cluster_cited = false;
    bool reused = false;         // one share: worth the planner's time to place peers beside the peers their workgroup gathers
    bool full_vector = false;    // several shares: keep every share's whole segment in the vectors (all-gather exchange) instead of the cited entries only
    // Relative time a workgroup on XCD x (workgroup b runs on XCD b mod 8) needs per edge of a blocked sweep, in 1/1024
    // (1024 = the same everywhere): the deal levels edges x cost instead of edges.  Measured per device by
    // xcd_cost_factors (engine.hpp); a plan made without a device keeps the default.
    uint32_t xcd_cost[8] = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
};

struct HostGraph {
    uint32_t n = 0;
    uint32_t n_valid = 0;
    uint64_t nnz = 0;            // surviving edges of non-dangling peers, all segments
    uint64_t nnz_local = 0;      // ... whose destination row is in this segment
    bool values_are_u32 = true;
    uint32_t value_max = 0;      // bit-or of the surviving integer opinions (how wide the edge records' opinions are)
    Geometry geo{};
    std::vector<uint32_t> pos_of_ext;   // n entries: position of each peer, 0xFFFFFFFF for empty slots
    std::vector<SliceDesc> slices;
    std::vector<WorkDesc> work;
    uint64_t n_steps = 0;
    std::vector<uint32_t> col;          // n_steps*64 sources, (segment << 29) | local row
    std::vector<uint32_t> val_u32;      // n_steps*64 raw opinions (values_are_u32)
    std::vector<FrRaw> val_fr;          // n_steps*64 raw opinions (!values_are_u32)
    // f64 variant ("blocked sweep"): per workgroup b the edges of its rows, ascending by source
    // position; entries bk_ptr[b] .. bk_ptr[b+1].
    std::vector<uint64_t> bk_ptr;
    // bk_grp[b * (f64_rb / BLOCK_GROUP) + j]: the row group whose accumulators are slot j of workgroup b, or ~0
    std::vector<uint32_t> bk_grp;
    std::vector<uint32_t> bk_col;       // source, (segment << 29) | local row
    std::vector<uint16_t> bk_row;       // row inside the workgroup's accumulator array
    std::vector<uint32_t> bk_val_u32;   // raw opinions (values_are_u32)
    // the same arrangement for the exact blocked sweep (geo.fr_blocks workgroups of geo.fr_rb accumulator rows)
    std::vector<uint64_t> fb_ptr;
    std::vector<uint32_t> fb_grp, fb_col;
    std::vector<uint16_t> fb_row;
    std::vector<uint8_t> fb_val;        // byte opinions (geo.fr_compact != 2)
    std::vector<uint32_t> fb_val_u32;   // 32-bit opinions (geo.fr_compact == 2)
    // chunked exchange: [workgroup][n_chunks + 1] absolute indices into the lists above, run q = edges whose source
    // lies in piece q (empty when n_chunks == 1)
    std::vector<uint64_t> bk_phase, fb_phase;
    std::vector<FrRaw> bk_val_fr;
    // Per-position source factors (zero for dangling / non-member / padding positions):
    std::vector<FrRaw> scale_fr;        // inv(row sum) * 2^576 mod r
    std::vector<double> scale_f64;      // 1 / integer row sum
    FrRaw inv_nm1_w{};                  // inv(n_valid - 1) * 2^288 mod r
    uint32_t n_dangling_local = 0;      // real dangling rows in this segment
    // Sources clustered by consumer (cluster_cold_sources): the workgroup each group of active rows is meant for when a
    // blocked sweep runs hint_blocks workgroups, ~0 = none; empty = positions are in plain degree order
    std::vector<uint32_t> grp_hint;
    uint32_t hint_blocks = 0;
    // One share: the leading positions held by the most-attesting peers in order of out-degree (the hot prefix; with clustered
    // positions every peer above the clustering threshold).  A sweep's edges from these sources come first in every
    // workgroup's list and read a dense, cache-resident part of the vector.  0 = not known (several shares).
    uint32_t dense_prefix = 0;
    // Several shares with a gather vector (geo.compact, see layout.hpp).  need[p]: the local rows of share p that this share's
    // rows cite, in the order their entries take in block p of the gather vector (ascending); send[c]: the local rows of THIS
    // share that share c cites, in c's order -- what this share packs for c after every step.  Both are empty for the share
    // itself.  send_known: the send lists are filled (a share laid out from the whole list derives them itself; one laid out
    // from edge records is told by the other shares).
    std::vector<uint32_t> need[8], send[8];
    bool send_known = false;
    uint32_t blk_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // first entry of block p (the share's own block is [0, seg_stride()))
    uint32_t xcd_cost[8] = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};   // what the deals were levelled with
};

// Returns 0 or a negative ET_E_* code; `err` receives a human-readable reason.
int build_graph(const BuildInput& in, const BuildOpts& opt, HostGraph& out, std::string& err);

// The same in two stages, for a caller that lays out every share of one graph: what does not depend on the share
// (filter, dangling peers, order, positions, factors: steps 1-6) is computed once, then each share's edge lists.
// The input arrays must stay alive until the last plan_segment call.
struct PlanCommon {
    BuildInput in;
    HostGraph g;                        // geometry (seg = 0), positions, factors; no edge lists
    std::vector<uint64_t> out_ptr, kept;
    std::vector<uint8_t> drop, dangling;
    std::vector<uint32_t> order;
};
int plan_common(const BuildInput& in, const BuildOpts& opt, PlanCommon& out, std::string& err);
// last_share: no further share will be cut from `pc` (its common part is handed over rather than copied; pc.g is empty afterwards)
int plan_segment(const PlanCommon& pc, const BuildOpts& opt, uint32_t seg, HostGraph& out, std::string& err, bool last_share = false);

// A share laid out by a process that never sees the whole attestation list (several processes, each holding the
// attestations of its own attesters).  The per-peer statistics are summed over the processes; every process then derives the
// same order, geometry and positions (layout_common fills g.geo, g.pos_of_ext and `order`; g.n, g.n_valid, g.nnz and
// g.values_are_u32 are set by the caller), sends each surviving edge to the share that owns its destination row as a record
// (source code = (segment << 29) | local row, destination local row, integer opinion), and lays out its own share from the
// records it received.
struct PeerStats {
    std::vector<uint32_t> in_deg, out_deg;   // over the surviving edges of non-dangling attesters
    uint32_t value_max = 0;                  // bit-or of the surviving integer opinions (0 for field-valued ones)
};
int layout_common(uint32_t n, const uint8_t* member, bool values_are_u32, const std::vector<uint8_t>& dangling, const PeerStats& st,
                  const BuildOpts& opt, HostGraph& g, std::vector<uint32_t>& order, std::string& err);
// factors of the active peers listed (integer row sums); positions outside the list keep zero factors
void fill_factors_u64(HostGraph& g, const uint32_t* peers, uint32_t count, const uint64_t* row_sum);
// `g` holds the common part on entry (layout_common + fill_factors_u64) and the share's plan on return
int plan_segment_from_records(HostGraph& g, const BuildOpts& opt, uint32_t seg, const std::vector<uint32_t>& src_code,
                              const std::vector<uint32_t>& row, const std::vector<uint32_t>& val, std::string& err);
// With a gather vector the other shares must learn which of their rows this share cites before they can pack them: the same
// in two steps.  records_to_needs sorts the records, deals the row groups to the workgroups (opt.xcd_cost must be final) and
// fills g.need (and the geometry of the gather vector); the caller
// delivers need[p] to share p and stores what the others want from this share in g.send (g.send_known = true);
// plan_segment_from_sorted then lays out the share.  (Position-order layouts: need stays empty, nothing to deliver.)
struct SortedRecords;
void free_sorted_records(SortedRecords* r);
int records_to_needs(HostGraph& g, const BuildOpts& opt, uint32_t seg, const std::vector<uint32_t>& src_code, const std::vector<uint32_t>& row,
                     const std::vector<uint32_t>& val, SortedRecords** out, std::string& err);
int plan_segment_from_sorted(HostGraph& g, const BuildOpts& opt, SortedRecords* rec, std::string& err);
// steps 1-4 alone (filter, last write wins, row sums, degrees) over the attestations given
struct RowSums {
    std::vector<uint64_t> u64;          // integer opinions
    std::vector<Fr> fr;                 // field-valued opinions: in the field ...
    std::vector<long double> ld;        // ... and as real numbers, for the f64 variant
};
int filter_edges(const BuildInput& in, PlanCommon& pc, RowSums& sums, PeerStats& st, std::string& err);

// Lines of the vector one launch of a blocked sweep pulls into the L1s: distinct (workgroup, 128-byte line) pairs of the
// sorted per-workgroup lists (f64: 16 entries per line; exact: 4).  blocks = 0 when the plan holds no such lists.
struct SweepStats { uint32_t blocks = 0, clustered = 0; uint64_t edges = 0, edges_most = 0, line_fills = 0, line_fills_most = 0, chunk_span_most = 0, xcd_line_fills = 0; };
SweepStats sweep_stats(const HostGraph& g, bool exact);

// host threads for the scatter passes over the edges (the GPU boxes give a process 16 CPUs)
inline unsigned host_threads() {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned t = std::max(1u, std::min(16u, hw ? hw : 1u));
    if (const char* e = getenv("ET_BUILD_THREADS")) t = (unsigned)std::max(1, atoi(e));
    return t;
}
template <typename F>
inline void run_threads(unsigned count, F f) {
    if (count <= 1) { f(0); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < count; ++t) th.emplace_back([&f, t]() { f(t); });
    f(0);
    for (auto& x : th) x.join();
}

// std::sort on the host threads: the range in `count` pieces sorted side by side, then merged pairwise (the same result as
// std::sort for a strict weak order without equal elements, which is what the planner's orders are: every one ends in
// the peer's index)
template <typename It, typename Cmp>
inline void parallel_sort(It first, It last, Cmp cmp, unsigned count) {
    const size_t len = (size_t)(last - first);
    if (count <= 1 || len < 65536) { std::sort(first, last, cmp); return; }
    unsigned pieces = 1;
    while (pieces * 2 <= count) pieces *= 2;
    auto cut = [&](unsigned i) { return first + (ptrdiff_t)(len * i / pieces); };
    run_threads(pieces, [&](unsigned t) { std::sort(cut(t), cut(t + 1), cmp); });
    for (unsigned width = 1; width < pieces; width *= 2)
        run_threads(pieces / (2 * width), [&](unsigned t) {
            const unsigned i = t * 2 * width;
            std::inplace_merge(cut(i), cut(i + width), cut(i + 2 * width), cmp);
        });
}

}  // namespace et

// the C ABI's host-side plan handle (et_plan_* and et_dplan_finish_host produce it)
struct et_plan {
    et::HostGraph h;
};
