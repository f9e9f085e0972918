#include "builder.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <queue>
#include <thread>
#include <cstdlib>

#include "../../include/eigentrust_hip.h"

namespace et {

namespace {

inline uint32_t round_up(uint32_t v, uint32_t m) { return m == 0 ? v : (v + m - 1) / m * m; }

struct Pos {  // global sorted row -> (segment, local row, position)
    uint32_t n_seg, stride;
    inline uint32_t seg_of(uint32_t q) const { return (q >> 6) % n_seg; }
    inline uint32_t local_of(uint32_t q) const { return ((q >> 6) / n_seg) * 64 + (q & 63); }
    inline uint32_t pos_of(uint32_t q) const { return seg_of(q) * stride + local_of(q); }
};

}  // namespace

// Step 5 of the plan: the global row order, the geometry every share derives alike, and each peer's position.
int layout_common(uint32_t n, const uint8_t* member, bool values_are_u32, const std::vector<uint8_t>& dangling, const PeerStats& st,
                  const BuildOpts& opt, HostGraph& g, std::vector<uint32_t>& order, std::string& err) {
    auto is_member = [&](uint32_t i) { return member ? member[i] != 0 : true; };
    const uint32_t n_valid = g.n_valid;
    g.value_max = values_are_u32 ? st.value_max : 0u;
    uint32_t n_dangling = 0;
    for (uint32_t i = 0; i < n; ++i) n_dangling += is_member(i) && dangling[i];
    // ---- 5. global row order ------------------------------------------------------------------
    // Active peers: the opt.hot_prefix most-gathered (highest out-degree), then the rest; inside each
    // group by in-degree (descending) so that the 64 rows of a slice have similar lengths.  Dangling
    // peers follow.
    const uint32_t n_active = n_valid - n_dangling;
    order.clear();
    order.reserve(n_valid);
    for (uint32_t i = 0; i < n; ++i) if (is_member(i) && !dangling[i]) order.push_back(i);
    for (uint32_t i = 0; i < n; ++i) if (is_member(i) && dangling[i]) order.push_back(i);
    auto by_deg = [&](uint32_t a, uint32_t b) {
        if (st.in_deg[a] != st.in_deg[b]) return st.in_deg[a] > st.in_deg[b];
        if (st.out_deg[a] != st.out_deg[b]) return st.out_deg[a] > st.out_deg[b];
        return a < b;
    };
    auto by_out = [&](uint32_t a, uint32_t b) {
        if (st.out_deg[a] != st.out_deg[b]) return st.out_deg[a] > st.out_deg[b];
        return a < b;
    };
    {
        const uint32_t h1 = std::min<uint32_t>(n_active, opt.hot_prefix);
        auto b = order.begin();
        if (h1 < n_active) std::nth_element(b, b + h1, b + n_active, by_out);
        std::sort(b, b + h1, by_deg);
        parallel_sort(b + h1, b + n_active, by_deg, host_threads());
    }
    parallel_sort(order.begin() + n_active, order.end(), by_deg, host_threads());

    const uint32_t G = opt.n_seg;
    Geometry geo{};
    geo.n_seg = G;
    geo.seg = opt.seg;
    geo.n_active = n_active;
    geo.n_active_pad = round_up(n_active, 64 * G);
    geo.n_dangling = n_dangling;
    geo.n_dangling_pad = round_up(n_dangling, 64 * G);
    geo.rows_cap = (geo.n_active_pad + geo.n_dangling_pad) / G;
    // one partial-sum slot per wavefront that owns dangling rows; the count depends only on global
    // quantities so that every segment derives the same geometry
    const uint32_t target_work = opt.target_work ? opt.target_work : 16384;
    {
        uint64_t share = ((uint64_t)target_work * n_dangling + n_valid - 1) / n_valid;
        geo.nd_cap = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(16, (share + 15) / 16 * 16));
    }
    {
        // f64 workgroups: one per CU unless the rows of a workgroup would not fit its LDS accumulators
        // (12288 rows plus the slack below stay under 160 KiB of f64 accumulators)
        const uint32_t n_groups = geo.rows_cap / BLOCK_GROUP, rb_max = 12288;
        const uint32_t per_round = opt.f64_blocks ? opt.f64_blocks : 256;
        uint32_t B = std::max<uint32_t>(1, std::min<uint32_t>(per_round, n_groups));
        const uint32_t need = (geo.rows_cap + rb_max - 1) / rb_max;
        // more than one workgroup per CU: whole rounds, so that the last round is as full as the first
        if (need > B) B = (need + per_round - 1) / per_round * per_round;
        geo.f64_blocks = B;
        // a quarter more accumulator slots than an even deal needs: workgroups holding light rows
        // take more 64-row groups than those holding heavy ones
        const uint32_t even = (n_groups + B - 1) / B;
        geo.f64_rb = (even + even / 4 + 2) * BLOCK_GROUP;
        geo.f64_dblocks = dangling_blocks(B, geo.f64_rb / BLOCK_GROUP, geo.n_dangling_pad / BLOCK_GROUP / G);
        geo.nd_cap = std::max<uint32_t>(geo.nd_cap, (geo.f64_dblocks + 15) / 16 * 16);
        // exact blocked sweep: whole rounds of 256 workgroups of about 2048 rows (64 bytes of LDS per row); only
        // where it was measured to pay
        geo.fr_blocks = geo.fr_rb = geo.fr_dblocks = geo.fr_compact = 0;
        bool bytes_only = values_are_u32 && !getenv("ET_FR_ROW_PER_LANE");
        // from 16K rows per share upwards the sweep wins at every size and share count measured (16K peers: 19 against
        // 30 us per iteration; one of eight shares of 16M peers: 421 against 588 us); the knobs move the bounds for tests
        const uint32_t win_hi = getenv("ET_FR_SWEEP_MAX_ROWS") ? (uint32_t)atoll(getenv("ET_FR_SWEEP_MAX_ROWS")) : 0xFFFFFFFFu;
        const uint32_t win_lo = getenv("ET_FR_SWEEP_MIN_ROWS") ? (uint32_t)atoll(getenv("ET_FR_SWEEP_MIN_ROWS")) : (1u << 14);
        const bool window = geo.rows_cap >= win_lo && geo.rows_cap <= win_hi;
        // 32-bit opinions: 38-byte rows, which hold the sum of fewer than 2^17 products of 288 bits
        uint32_t in_deg_max = 0;
        if (st.value_max > 255) for (uint32_t d : st.in_deg) in_deg_max = std::max(in_deg_max, d);
        const bool wide = st.value_max > 255 && in_deg_max < FR_U32_MAX_IN_DEGREE && !getenv("ET_FR_SWEEP_BYTES_ONLY");
        bytes_only = bytes_only && (st.value_max <= 255 || wide);
        if (bytes_only && window) {
            // whole rounds of 256 workgroups (one per CU), as few as the rows' LDS footprint allows: every round
            // pulls the whole vector through every L2 again
            const bool compact = wide || !getenv("ET_FR_WIDE_ROWS");  // measurement knob: the 64-byte rows
            const uint32_t rb_max = wide ? FR_RB_MAX_U32 : (compact ? FR_RB_MAX_COMPACT : FR_RB_MAX_WIDE);
            uint32_t Bf = 256;
            while ((n_groups + Bf - 1) / Bf * BLOCK_GROUP > rb_max) Bf += 256;
            const uint32_t even_f = (n_groups + Bf - 1) / Bf;
            // no spare accumulator slots: every workgroup holds exactly its share of the row groups (the deal still
            // levels the edges to about 2 %), so that its 1024 threads finish the rows in full passes
            geo.fr_blocks = Bf;
            geo.fr_rb = even_f * BLOCK_GROUP;
            geo.fr_compact = wide ? 2u : (compact ? 1u : 0u);
            geo.fr_dblocks = dangling_blocks(Bf, geo.fr_rb / BLOCK_GROUP, geo.n_dangling_pad / BLOCK_GROUP / G);
            geo.nd_cap = std::max<uint32_t>(geo.nd_cap, (geo.fr_dblocks + 15) / 16 * 16);
        }
    }
    geo.n_chunks = 1;
    geo.chunk_rows = geo.rows_cap;
    if (opt.n_chunks > 1 && geo.fr_blocks != 0 && geo.fr_compact != 2) {
        const uint32_t K = std::min<uint32_t>(opt.n_chunks, 8);
        const uint32_t cr = round_up((geo.rows_cap + K - 1) / K, 64);
        if ((uint64_t)(K - 1) * cr < geo.rows_cap) { geo.n_chunks = K; geo.chunk_rows = cr; }
    }
    // several shares, integer opinions, one exchange per iteration: a share's vectors hold the entries its rows cite, not
    // every share's whole segment
    geo.compact = (G > 1 && values_are_u32 && geo.n_chunks == 1 && !opt.full_vector && !getenv("ET_FULL_VECTOR")) ? 1u : 0u;
    geo.set_share(opt.seg);
    g.geo = geo;
    if (G > 8 || geo.seg_stride() >= COL_LOCAL_MASK) {
        err = "more than 8 segments or more than 2^29 rows per segment";
        return ET_E_UNSUPPORTED;
    }
    Pos P{G, geo.seg_stride()};
    // q (global sorted row incl. class padding) of the i-th entry of `order`
    auto q_of_rank = [&](uint32_t r) { return r < n_active ? r : geo.n_active_pad + (r - n_active); };
    g.pos_of_ext.assign(n, 0xFFFFFFFFu);
    for (uint32_t r = 0; r < n_valid; ++r) g.pos_of_ext[order[r]] = P.pos_of(q_of_rank(r));
    g.dense_prefix = G == 1 ? std::min<uint32_t>(n_active, opt.hot_prefix) : 0u;

    return ET_OK;
}

// Step 6: inv(row sum) * 2^576 and 1 / row sum at the positions of the given active peers (native.rs:308), one batch inversion
// per host thread; inv(n_valid - 1) * 2^288.
template <typename SumFr, typename InvF64>
static void fill_factors(HostGraph& g, const uint32_t* peers, uint32_t count, SumFr sum_of, InvF64 inv_f64_of) {
    const size_t vec_len = (size_t)g.geo.n_seg * g.geo.seg_stride();
    g.scale_fr.assign(vec_len, FrRaw{{0, 0, 0, 0}});
    g.scale_f64.assign(vec_len, 0.0);
    const Fr two576 = Fr::one().pow2k(576);
    const unsigned T6 = host_threads();
    run_threads(T6, [&](unsigned t) {
        const uint32_t r0 = (uint32_t)((uint64_t)count * t / T6), r1 = (uint32_t)((uint64_t)count * (t + 1) / T6);
        std::vector<Fr> inv(r1 - r0);
        for (uint32_t r = r0; r < r1; ++r) inv[r - r0] = sum_of(peers[r]);
        fr_batch_invert(inv);
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t i = peers[r], p = g.pos_of_ext[i];
            g.scale_fr[p] = (inv[r - r0] * two576).to_raw();
            g.scale_f64[p] = inv_f64_of(i);
        }
    });
    g.inv_nm1_w = (Fr::from_u64(g.n_valid - 1).invert_or_zero() * Fr::one().pow2k(288)).to_raw();
}

void fill_factors_u64(HostGraph& g, const uint32_t* peers, uint32_t count, const uint64_t* row_sum) {
    fill_factors(g, peers, count, [&](uint32_t i) { return Fr::from_u64(row_sum[i]); }, [&](uint32_t i) { return 1.0 / (double)row_sum[i]; });
}

// Steps 1-4 of the plan over the attestations given: filter, last write wins, row sums, degrees.  `locally_dangling` marks
// the members whose opinions here sum to zero (all of them, for a peer without attestations in this list).
int filter_edges(const BuildInput& in, PlanCommon& pc, RowSums& sums, PeerStats& st, std::string& err) {
    HostGraph& g = pc.g;
    const uint32_t n = in.n;
    if (n == 0 || (in.n_edges && (!in.from || !in.to)) || ((in.val_u32 != nullptr) == (in.val_fr != nullptr) && in.n_edges)) {
        err = "bad build input";
        return ET_E_INVALID_ARG;
    }
    g = HostGraph();
    g.n = n;
    g.values_are_u32 = in.val_fr == nullptr;
    auto is_member = [&](uint32_t i) { return in.member ? in.member[i] != 0 : true; };
    uint32_t n_valid = 0;
    for (uint32_t i = 0; i < n; ++i) n_valid += is_member(i);
    g.n_valid = n_valid;
    if (n_valid < 2) {  // native.rs:288-289
        err = "Insufficient peers for calculation!";
        return ET_E_INSUFFICIENT_PEERS;
    }

    // ---- 1. filter (native.rs:248-260) and group by attester --------------------------------
    // The passes over the attestation list run on the host threads.  Lists that arrive grouped by attester (ascending
    // `from`: what a caller that walks its attesters produces) are filtered in place, chunk by chunk; any other order takes
    // the counting sort below on one thread (the order inside an attester's group must stay the list's: a later attestation
    // for the same pair replaces an earlier one).
    const uint64_t ne = in.n_edges;
    const unsigned T = host_threads();
    std::vector<uint64_t>& out_ptr = pc.out_ptr;
    out_ptr.assign((size_t)n + 1, 0);
    auto keeps = [&](uint64_t e) { const uint32_t f = in.from[e], t = in.to[e]; return f != t && is_member(f) && is_member(t); };
    std::vector<uint64_t> chunk_kept(T + 1, 0);
    std::vector<uint8_t> chunk_bad(T, 0), chunk_unsorted(T, 0);
    run_threads(T, [&](unsigned t) {
        uint64_t c = 0;
        const uint64_t e0 = ne * t / T, e1 = ne * (t + 1) / T;
        for (uint64_t e = e0; e < e1; ++e) {
            const uint32_t f = in.from[e], to = in.to[e];
            if (f >= n || to >= n) { chunk_bad[t] = 1; return; }
            if (e > 0 && in.from[e - 1] > f) chunk_unsorted[t] = 1;
            if (f == to || !is_member(f) || !is_member(to)) continue;
            ++c;
            __atomic_fetch_add(&out_ptr[f + 1], 1, __ATOMIC_RELAXED);
        }
        chunk_kept[t + 1] = c;
    });
    for (unsigned t = 0; t < T; ++t) if (chunk_bad[t]) { err = "edge endpoint out of range"; return ET_E_OUT_OF_RANGE; }
    bool grouped = true;
    for (unsigned t = 0; t < T; ++t) grouped = grouped && !chunk_unsorted[t];
    for (uint32_t i = 0; i < n; ++i) out_ptr[i + 1] += out_ptr[i];
    const uint64_t m0 = out_ptr[n];
    // kept[k] = original edge index, grouped by attester, original order inside a group
    std::vector<uint64_t>& kept = pc.kept;
    kept.assign(m0, 0);
    if (grouped) {
        for (unsigned t = 0; t < T; ++t) chunk_kept[t + 1] += chunk_kept[t];
        run_threads(T, [&](unsigned t) {
            uint64_t k = chunk_kept[t];
            for (uint64_t e = ne * t / T; e < ne * (t + 1) / T; ++e) if (keeps(e)) kept[k++] = e;
        });
    } else {
        std::vector<uint64_t> cur(out_ptr.begin(), out_ptr.end() - 1);
        for (uint64_t e = 0; e < ne; ++e) if (keeps(e)) kept[cur[in.from[e]]++] = e;
    }
    // every later pass walks attesters: T ranges of them with about equal numbers of attestations
    std::vector<uint32_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) cut[t] = (uint32_t)(std::lower_bound(out_ptr.begin(), out_ptr.end(), m0 * t / T) - out_ptr.begin());
    for (unsigned t = 1; t <= T; ++t) cut[t] = std::min<uint32_t>(n, std::max(cut[t], cut[t - 1]));
    cut[T] = n;
    // ---- 2. inside each attester: order by rated peer, last write wins (lib.rs:415) ---------
    std::vector<uint8_t>& drop = pc.drop;
    drop.assign(m0, 0);
    run_threads(T, [&](unsigned t) {
        for (uint32_t f = cut[t]; f < cut[t + 1]; ++f) {
            uint64_t b = out_ptr[f], e = out_ptr[f + 1];
            if (e - b < 2) continue;
            bool sorted_unique = true;
            for (uint64_t k = b + 1; k < e; ++k)
                if (in.to[kept[k - 1]] >= in.to[kept[k]]) { sorted_unique = false; break; }
            if (sorted_unique) continue;
            std::stable_sort(kept.begin() + b, kept.begin() + e, [&](uint64_t x, uint64_t y) { return in.to[x] < in.to[y]; });
            for (uint64_t k = b; k + 1 < e; ++k)
                if (in.to[kept[k]] == in.to[kept[k + 1]]) drop[k] = 1;  // an equal later one follows
        }
    });
    // ---- 3. row sums: in the field (dangling test, native.rs:263-264) and as integers -------
    std::vector<Fr>& sum_fr = sums.fr;
    std::vector<uint64_t>& sum_u64 = sums.u64;
    std::vector<long double>& sum_ld = sums.ld;
    if (g.values_are_u32) sum_u64.assign(n, 0);
    else { sum_fr.assign(n, Fr::zero()); sum_ld.assign(n, 0.0L); }
    std::vector<uint32_t> vmax_t(T, 0);                 // largest surviving integer opinion, per thread
    std::vector<uint8_t> noncanonical(T, 0);
    run_threads(T, [&](unsigned t) {
        uint32_t vmax = 0;
        for (uint32_t f = cut[t]; f < cut[t + 1]; ++f) {
            for (uint64_t k = out_ptr[f]; k < out_ptr[f + 1]; ++k) {
                if (drop[k]) continue;
                uint64_t e = kept[k];
                if (g.values_are_u32) { sum_u64[f] += in.val_u32[e]; vmax = std::max(vmax, in.val_u32[e]); }
                else {
                    const FrRaw& v = in.val_fr[e];
                    if (!fr_raw_is_canonical(v.l)) { noncanonical[t] = 1; return; }
                    sum_fr[f] += Fr::from_raw(v);
                    sum_ld[f] += (long double)v.l[0] + 18446744073709551616.0L * ((long double)v.l[1] + 18446744073709551616.0L * ((long double)v.l[2] + 18446744073709551616.0L * (long double)v.l[3]));
                }
            }
        }
        vmax_t[t] = vmax;
    });
    uint32_t vmax = 0;
    for (unsigned t = 0; t < T; ++t) {
        if (noncanonical[t]) { err = "opinion value is not a canonical field element"; return ET_E_NONCANONICAL_FR; }
        vmax = std::max(vmax, vmax_t[t]);
    }
    std::vector<uint8_t>& dangling = pc.dangling;
    dangling.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (!is_member(i)) continue;
        bool zero = g.values_are_u32 ? sum_u64[i] == 0 : sum_fr[i].is_zero();
        if (zero) dangling[i] = 1;
    }
    // ---- 4. degrees over surviving edges -----------------------------------------------------
    st.in_deg.assign(n, 0);
    st.out_deg.assign(n, 0);
    st.value_max = vmax;
    std::vector<uint64_t> nnz_t(T, 0);
    run_threads(T, [&](unsigned t) {
        uint64_t c = 0;
        for (uint32_t f = cut[t]; f < cut[t + 1]; ++f) {
            if (dangling[f]) continue;
            for (uint64_t k = out_ptr[f]; k < out_ptr[f + 1]; ++k) {
                if (drop[k]) continue;
                __atomic_fetch_add(&st.in_deg[in.to[kept[k]]], 1u, __ATOMIC_RELAXED);
                st.out_deg[f]++;
                ++c;
            }
        }
        nnz_t[t] = c;
    });
    uint64_t nnz = 0;
    for (unsigned t = 0; t < T; ++t) nnz += nnz_t[t];
    g.nnz = nnz;
    pc.in = in;
    return ET_OK;
}

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// Step 5b (one share, both variants on the blocked sweep): positions that put a source next to the sources the same
// workgroup gathers.
//
// A blocked sweep is bound by the 128-byte lines a workgroup's gathers pull into its L1.  In degree order a source with
// few out-edges costs one line per edge: its line-mates are gathered by other workgroups.  The rows of a workgroup are
// 16-row groups of consecutive positions, so a peer's position fixes both where its score lives and which workgroup sums
// its row.  Peers with at most `max_deg` out-edges are therefore placed by label propagation under capacity: a peer joins
// the workgroup that holds most of the peers it rates or is rated by (each such edge then reads from the workgroup's own
// 16-row groups, every line of which it uses in full), and inside a workgroup's peers the order is by the workgroup of
// another rated peer, so that one more edge per source finds its line-mates gathered together.  The most-attesting peers
// keep their degree order in front.  Row groups are dealt here (`grp_hint`); lay_out_segment follows the hints where the
// workgroup has room and levels the rest.  Positions are internal: results are un-permuted by `pos_of_ext` as before.
static void cluster_cold_sources(PlanCommon& pc, const PeerStats& st, const BuildOpts& opt) {
    const double t_begin = now_s();
    HostGraph& g = pc.g;
    const Geometry& geo = g.geo;
    const BuildInput& in = pc.in;
    std::vector<uint32_t>& order = pc.order;
    // one cluster per workgroup of the sweep with the most workgroups (the exact sweep's rows take more LDS, so it may run a
    // multiple of the f64 sweep's count; the f64 sweep then gives each workgroup several whole clusters)
    const uint32_t n = in.n, n_active = geo.n_active, B = std::max(geo.f64_blocks, geo.fr_blocks);
    // Who is clustered: a line of the f64 vector holds 16 sources; when they have more out-edges together than the sweep has
    // workgroups, nearly every workgroup gathers from the line anyway and the sources are better off among their like, in order
    // of out-degree.  Hence: out-degree up to workgroups / 16 (16 for one workgroup per CU; measured optimum 12-16 for the f64
    // sweep, 16-32 for the exact one, whose lines hold four sources).
    const uint32_t max_deg = getenv("ET_AFFINITY_MAX_DEG") ? (uint32_t)atoi(getenv("ET_AFFINITY_MAX_DEG")) : std::max<uint32_t>(4, geo.f64_blocks / 16);
    const uint32_t rounds = getenv("ET_AFFINITY_ROUNDS") ? (uint32_t)std::max(1, atoi(getenv("ET_AFFINITY_ROUNDS"))) : 3;
    const uint32_t n_agroups = geo.n_active_pad / BLOCK_GROUP;
    // below about a thousand rows per workgroup the lines are dense anyway (every workgroup gathers from most of them): measured
    // gain under 5 % of the fills at 512 rows, for a plan that takes three times as long
    if (B < 2 || n_agroups / B < 64) return;
    const uint32_t H = n_agroups / B - 3;          // hinted groups per workgroup; the rest is dealt for balance
    const uint32_t NONE = 0xFFFFFFFFu;
    auto by_out = [&](uint32_t a, uint32_t b) {
        if (st.out_deg[a] != st.out_deg[b]) return st.out_deg[a] > st.out_deg[b];
        return a < b;
    };
    // ---- classes: front (hot prefix, then the other peers above max_deg by out-degree), cold ----
    const uint32_t h1 = std::min<uint32_t>(n_active, opt.hot_prefix);
    std::vector<uint32_t> front(order.begin(), order.begin() + h1), mid, cold;
    for (uint32_t r = h1; r < n_active; ++r) (st.out_deg[order[r]] > max_deg ? mid : cold).push_back(order[r]);
    const uint32_t pad = (uint32_t)((BLOCK_GROUP - (front.size() + mid.size()) % BLOCK_GROUP) % BLOCK_GROUP);
    if (cold.size() < (size_t)pad + (size_t)B * BLOCK_GROUP * 8) return;
    if (pad) {
        std::nth_element(cold.begin(), cold.begin() + pad, cold.end(), by_out);
        mid.insert(mid.end(), cold.begin(), cold.begin() + pad);
        cold.erase(cold.begin(), cold.begin() + pad);
    }
    std::sort(cold.begin(), cold.end());               // visit order = slot order: the attestation list is walked front to back
    std::sort(mid.begin(), mid.end(), by_out);
    front.insert(front.end(), mid.begin(), mid.end());
    const uint32_t n_front = (uint32_t)front.size(), n_fg = n_front / BLOCK_GROUP, n_cold = (uint32_t)cold.size();
    if ((n_fg + B - 1) / B + 8 > H) return;
    // ---- the front's groups: heaviest first to the lightest workgroup, equal counts ----
    std::vector<uint32_t> hint(n_agroups, NONE), used(B, 0), label(n, NONE);
    std::vector<uint64_t> load(B, 0);
    uint64_t front_edges = 0, cold_edges = 0;
    {
        std::vector<uint64_t> w(n_fg, 0);
        for (uint32_t r = 0; r < n_front; ++r) w[r / BLOCK_GROUP] += st.in_deg[front[r]];
        std::vector<uint32_t> by_w(n_fg);
        std::iota(by_w.begin(), by_w.end(), 0u);
        std::stable_sort(by_w.begin(), by_w.end(), [&](uint32_t x, uint32_t y) { return w[x] > w[y]; });
        typedef std::pair<uint64_t, uint32_t> Load;
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        const uint32_t cap = (n_fg + B - 1) / B;
        for (uint32_t b = 0; b < B; ++b) heap.push(Load(0, b));
        for (uint32_t gi : by_w) {
            const uint32_t b = heap.top().second;
            heap.pop();
            hint[gi] = b;
            load[b] += w[gi] * opt.xcd_cost[b % 8];
            front_edges += w[gi];
            if (++used[b] < cap) heap.push(Load(load[b], b));
        }
        for (uint32_t r = 0; r < n_front; ++r) label[front[r]] = hint[r / BLOCK_GROUP];
    }
    // ---- cold peers: who rates them (cold raters only: a front source's lines are gathered by every workgroup anyway) ----
    std::vector<uint32_t> cidx(n, NONE);
    for (uint32_t k = 0; k < n_cold; ++k) { cidx[cold[k]] = k; cold_edges += st.in_deg[cold[k]]; }
    // (on all host threads: the raters of a peer are only counted per workgroup, so their order in its list is free)
    const unsigned T = host_threads();
    std::vector<uint64_t> rin_ptr((size_t)n_cold + 1, 0);
    run_threads(T, [&](unsigned t) {
        for (uint32_t k = (uint32_t)((uint64_t)n_cold * t / T); k < (uint32_t)((uint64_t)n_cold * (t + 1) / T); ++k) {
            const uint32_t f = cold[k];
            for (uint64_t e = pc.out_ptr[f]; e < pc.out_ptr[f + 1]; ++e) {
                if (pc.drop[e]) continue;
                const uint32_t c = cidx[in.to[pc.kept[e]]];
                if (c != NONE) __atomic_fetch_add(&rin_ptr[c + 1], 1, __ATOMIC_RELAXED);
            }
        }
    });
    for (uint32_t k = 0; k < n_cold; ++k) rin_ptr[k + 1] += rin_ptr[k];
    std::vector<uint32_t> rin(rin_ptr[n_cold]);
    {
        std::vector<uint64_t> cur(rin_ptr.begin(), rin_ptr.end() - 1);
        run_threads(T, [&](unsigned t) {
            for (uint32_t k = (uint32_t)((uint64_t)n_cold * t / T); k < (uint32_t)((uint64_t)n_cold * (t + 1) / T); ++k) {
                const uint32_t f = cold[k];
                for (uint64_t e = pc.out_ptr[f]; e < pc.out_ptr[f + 1]; ++e) {
                    if (pc.drop[e]) continue;
                    const uint32_t c = cidx[in.to[pc.kept[e]]];
                    if (c != NONE) rin[__atomic_fetch_add(&cur[c], 1, __ATOMIC_RELAXED)] = f;
                }
            }
        });
    }
    const double t_lp = now_s();
    // ---- label propagation under capacity ----
    // A workgroup's rows and its in-edges must both come out level.  Rows: every workgroup has room for the same number of
    // hinted groups.  Both fill at the pace of the round: after a fraction of the peers has been visited, a workgroup may
    // hold that fraction (plus a twentieth) of its room and of the edges between what the front's groups gave it and the
    // common target.  A popular workgroup therefore stays open to the end of the round, and a peer none of its candidates
    // takes goes to the emptiest workgroup that does.
    std::vector<uint32_t> rows_cap(B);
    uint64_t cold_room = 0;
    for (uint32_t b = 0; b < B; ++b) { rows_cap[b] = (H - used[b]) * BLOCK_GROUP; cold_room += rows_cap[b]; }
    if (cold_room > n_cold) return;
    uint64_t cost_sum = 0;
    for (uint32_t b = 0; b < B; ++b) cost_sum += opt.xcd_cost[b % 8];
    const long double hinted_edges = (long double)front_edges + (long double)cold_edges * cold_room / n_cold;
    const double target = (double)(hinted_edges * ((long double)cost_sum / B) / B), allowance = target * 0.002;
    std::vector<uint32_t> rows_left(B);
    std::vector<double> load_r(B);
    const bool stats = getenv("ET_BUILD_STATS") != nullptr;
    // A round visits the cold peers in SLICES interleaved slices.  Inside a slice the votes are counted on all host threads
    // from the labels as they stand (the earlier slices of this round included), the three best labels per peer are kept,
    // and the peers are then placed one after the other under the capacities: the outcome does not depend on the thread count.
    const uint32_t SLICES = 8, KEEP = 3;
    std::vector<uint32_t> cand((size_t)n_cold * KEEP, NONE);
    auto mix = [](uint32_t f, uint32_t l) { uint64_t z = ((uint64_t)f << 20 ^ l) * 0x9E3779B97F4A7C15ull; return z ^ (z >> 29); };
    for (uint32_t round = 0; round < rounds; ++round) {
        rows_left = rows_cap;
        for (uint32_t b = 0; b < B; ++b) load_r[b] = (double)load[b];
        uint64_t by_vote = 0, moved = 0;
        uint32_t visited = 0;
        for (uint32_t sl = 0; sl < SLICES; ++sl) {
            const uint32_t members = n_cold > sl ? (n_cold - sl + SLICES - 1) / SLICES : 0;
            run_threads(T, [&](unsigned t) {
                std::vector<uint32_t> votes(B, 0), touched;
                for (uint32_t j = (uint32_t)((uint64_t)members * t / T); j < (uint32_t)((uint64_t)members * (t + 1) / T); ++j) {
                    const uint32_t k = sl + SLICES * j, f = cold[k];
                    touched.clear();
                    auto vote = [&](uint32_t ext) {
                        const uint32_t l = label[ext];
                        if (l == NONE) return;
                        if (votes[l]++ == 0) touched.push_back(l);
                    };
                    for (uint64_t e = pc.out_ptr[f]; e < pc.out_ptr[f + 1]; ++e) if (!pc.drop[e]) vote(in.to[pc.kept[e]]);
                    for (uint64_t e = rin_ptr[k]; e < rin_ptr[k + 1]; ++e) vote(rin[e]);
                    for (uint32_t c = 0; c < KEEP; ++c) {
                        uint32_t best = NONE, at = 0;
                        for (uint32_t i = 0; i < touched.size(); ++i) {
                            const uint32_t l = touched[i];
                            if (votes[l] && (best == NONE || votes[l] > votes[best] || (votes[l] == votes[best] && mix(f, l) < mix(f, best)))) { best = l; at = i; }
                        }
                        cand[(size_t)k * KEEP + c] = best;
                        if (best != NONE) { votes[best] = 0; touched[at] = touched.back(); touched.pop_back(); }
                    }
                    for (uint32_t l : touched) votes[l] = 0;
                }
            });
            for (uint32_t j = 0; j < members; ++j, ++visited) {
                const uint32_t k = sl + SLICES * j, f = cold[k];
                // how far a workgroup may have filled by now: the round's progress plus a twentieth, in rows and in edges
                const double pace = std::min(1.0, (double)(visited + 1) / n_cold * 1.05);
                auto takes = [&](uint32_t b) {
                    return rows_left[b] && (double)(rows_cap[b] - rows_left[b]) < pace * rows_cap[b] + 16.0 &&
                           load_r[b] + (double)st.in_deg[f] * opt.xcd_cost[b % 8] <= (double)load[b] + (target - (double)load[b]) * pace + allowance;
                };
                uint32_t best = NONE;
                for (uint32_t c = 0; c < KEEP && best == NONE; ++c) {
                    const uint32_t l = cand[(size_t)k * KEEP + c];
                    if (l == NONE) break;
                    if (takes(l)) best = l;
                }
                by_vote += best != NONE;
                if (best == NONE) {
                    // the emptiest workgroup that takes the peer; if none does the peer stays without one and is dealt for balance
                    for (uint32_t b = 0; b < B; ++b)
                        if (takes(b) && (best == NONE || (uint64_t)rows_left[b] * rows_cap[best] > (uint64_t)rows_left[best] * rows_cap[b])) best = b;
                }
                moved += label[f] != best;
                label[f] = best;
                if (best != NONE) { --rows_left[best]; load_r[best] += (double)st.in_deg[f] * opt.xcd_cost[best % 8]; }
            }
        }
        if (stats) {
            uint64_t local = 0, all = 0;
            for (uint32_t k = 0; k < n_cold; ++k) {
                const uint32_t f = cold[k];
                for (uint64_t e = pc.out_ptr[f]; e < pc.out_ptr[f + 1]; ++e) if (!pc.drop[e]) { ++all; local += label[f] != NONE && label[in.to[pc.kept[e]]] == label[f]; }
            }
            double lo = 1e300, hi = 0;
            for (uint32_t b = 0; b < B; ++b) { lo = std::min(lo, load_r[b]); hi = std::max(hi, load_r[b]); }
            fprintf(stderr, "[et build] clustering round %u: %u cold peers of %u active, %llu placed by vote, %llu changed, %llu of their %llu out-edges stay in the workgroup; load %.0f..%.0f of %.0f\n",
                    round, n_cold, n_active, (unsigned long long)by_vote, (unsigned long long)moved, (unsigned long long)local, (unsigned long long)all, lo, hi, target);
        }
    }
    const double t_order = now_s();
    // ---- order: front, then workgroup by workgroup, inside by the workgroup of another rated peer; then the rest ----
    std::vector<uint64_t> key(n_cold);
    run_threads(T, [&](unsigned t) {
      std::vector<uint32_t> votes(B, 0), touched;
      for (uint32_t k = (uint32_t)((uint64_t)n_cold * t / T); k < (uint32_t)((uint64_t)n_cold * (t + 1) / T); ++k) {
        const uint32_t f = cold[k], own = label[f];
        touched.clear();
        for (uint64_t e = pc.out_ptr[f]; e < pc.out_ptr[f + 1]; ++e) {
            if (pc.drop[e]) continue;
            const uint32_t l = label[in.to[pc.kept[e]]];
            if (l == NONE || l == own) continue;
            if (votes[l]++ == 0) touched.push_back(l);
        }
        uint32_t second = B;
        // ties are broken by a hash of (peer, workgroup): every workgroup gets its share of the second-level runs
        for (uint32_t l : touched) if (second == B || votes[l] > votes[second] || (votes[l] == votes[second] && mix(f, l) < mix(f, second))) second = l;
        for (uint32_t l : touched) votes[l] = 0;
        key[k] = ((uint64_t)(own == NONE ? B : own) << 32) | second;
      }
    });
    std::vector<uint32_t> by_key(n_cold);
    std::iota(by_key.begin(), by_key.end(), 0u);
    parallel_sort(by_key.begin(), by_key.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : cold[a] < cold[b]; }, T);
    for (uint32_t r = 0; r < n_front; ++r) order[r] = front[r];
    {
        // a workgroup's peers fill whole groups; the few beyond its last whole group join the peers without a workgroup, which
        // follow in order of in-degree: groups of like weight, the light ones left for the workgroups that are heavy already
        std::vector<uint32_t> rest;
        uint32_t r = n_front, k = 0;
        while (k < n_cold) {
            const uint32_t own = (uint32_t)(key[by_key[k]] >> 32);
            uint32_t k1 = k;
            while (k1 < n_cold && (uint32_t)(key[by_key[k1]] >> 32) == own) ++k1;
            const uint32_t whole = own < B ? (k1 - k) / BLOCK_GROUP * BLOCK_GROUP : 0;
            for (uint32_t i = k; i < k + whole; ++i, ++r) { order[r] = cold[by_key[i]]; hint[r / BLOCK_GROUP] = own; }
            for (uint32_t i = k + whole; i < k1; ++i) rest.push_back(cold[by_key[i]]);
            k = k1;
        }
        std::sort(rest.begin(), rest.end(), [&](uint32_t a, uint32_t b) { return st.in_deg[a] != st.in_deg[b] ? st.in_deg[a] > st.in_deg[b] : a < b; });
        for (uint32_t i : rest) order[r++] = i;
    }
    for (uint32_t r = 0; r < n_active; ++r) g.pos_of_ext[order[r]] = r;     // one share: position = rank
    g.grp_hint.swap(hint);
    g.hint_blocks = B;
    g.dense_prefix = n_front;
    if (stats) fprintf(stderr, "[et build] clustering: classes and raters %.2f s, label propagation %.2f s, order %.2f s\n", t_lp - t_begin, t_order - t_lp, now_s() - t_order);
}

int plan_common(const BuildInput& in, const BuildOpts& opt, PlanCommon& pc, std::string& err) {
    if (opt.n_seg == 0 || opt.seg >= opt.n_seg) {
        err = "bad segment";
        return ET_E_INVALID_ARG;
    }
    const bool stats = getenv("ET_BUILD_STATS") != nullptr;
    const double t0 = now_s();
    RowSums sums;
    PeerStats st;
    int rc = filter_edges(in, pc, sums, st, err);
    if (rc) return rc;
    const double t1 = now_s();
    HostGraph& g = pc.g;
    if ((rc = layout_common(in.n, in.member, g.values_are_u32, pc.dangling, st, opt, g, pc.order, err))) return rc;
    if (stats) fprintf(stderr, "[et build] filter %.3f s, order and geometry %.3f s\n", t1 - t0, now_s() - t1);
    {
        const char* e = getenv("ET_AFFINITY");
        const bool wanted = e ? atoi(e) != 0 : opt.reused;      // pays off after thousands of converges of the same graph
        if (opt.n_seg == 1 && g.geo.n_chunks == 1 && g.geo.fr_blocks != 0 && g.geo.fr_blocks % g.geo.f64_blocks == 0 && wanted)
            cluster_cold_sources(pc, st, opt);
    }
    // ---- 6. per-position source factors ------------------------------------------------------
    const double t2 = now_s();
    fill_factors(g, pc.order.data(), g.geo.n_active, [&](uint32_t i) { return g.values_are_u32 ? Fr::from_u64(sums.u64[i]) : sums.fr[i]; },
                 [&](uint32_t i) { return g.values_are_u32 ? 1.0 / (double)sums.u64[i] : (double)(1.0L / sums.ld[i]); });
    if (stats) fprintf(stderr, "[et build] factors %.3f s\n", now_s() - t2);
    return ET_OK;
}

namespace {

// The surviving in-edges of one share's rows as the layout below consumes them: counted per row, then visited in ascending
// order of the source's visit key (position; piece by piece with a chunked exchange) in T contiguous chunks of the order.
// visit(local row, source code (segment << 29 | local row), handle); the handle fetches the edge's value.

// Gather vectors: the order of a share's blocks of cited entries (the other shares' rows, and its own rows' entries copied
// out of the own block by the pack).
//
// A blocked sweep is bound by the 128-byte lines a workgroup's gathers pull into its L1, and a source cited a few times
// costs one line per edge when its line-mates are gathered by other workgroups.  The cited entries have no place of their
// own in this share's vector (the own block, in row order, is only written and packed from), so the share puts them where
// its workgroups want them: the
// sources its rows cite `hot` times or more keep the front of their block, in position order (most-attesting first:
// nearly every workgroup gathers them anyway), and the others follow grouped by the workgroup that cites them most,
// then by the workgroup that cites them next, so that a line of sixteen (four) entries serves one workgroup sixteen
// (four) times.  `src` lists one block's cited sources: local row, citing edges, the two workgroups citing most.
struct CitedSource { uint32_t local, edges, wg1, wg2; };
static void order_cited_sources(std::vector<CitedSource>& src, uint32_t hot, uint32_t fold) {
    // workgroups b and b + fold of the sweep with more workgroups are ONE workgroup of the other sweep: side by side
    auto wg_key = [&](uint32_t w) -> uint64_t { return w == 0xFFFFFFFFu ? 0xFFFFFFFFull : (uint64_t)(w % fold) * 64 + w / fold; };
    // (Tried on the share of eight of the 16M-peer graph: the cold sources grouped first by the SET of XCDs whose workgroups
    // cite them halves the lines the eight L2s pull, 2.3M -> 1.1M, for 4 % more lines into the L1s, and the launch gets 5 %
    // slower: the L1 fills are what the sweeps wait for.  Sources two workgroups cite kept apart from those three or more
    // cite: no difference.)
    std::sort(src.begin(), src.end(), [&](const CitedSource& a, const CitedSource& b) {
        const bool ha = a.edges >= hot, hb = b.edges >= hot;
        if (ha != hb) return ha;
        if (ha) return a.local < b.local;
        const uint64_t ka = wg_key(a.wg1), kb = wg_key(b.wg1);
        if (ka != kb) return ka < kb;
        const uint64_t sa = wg_key(a.wg2), sb = wg_key(b.wg2);
        if (sa != sb) return sa < sb;
        return a.local < b.local;
    });
}
// the two workgroups with the most votes (ties: the smaller hash of (source, workgroup), so that no workgroup is favoured)
struct VoteBox {
    std::vector<uint32_t> votes, touched;
    explicit VoteBox(uint32_t B) : votes(B, 0) {}
    void add(uint32_t w) { if (votes[w]++ == 0) touched.push_back(w); }
    void take(uint32_t who, uint32_t& w1, uint32_t& w2) {
        auto mix = [&](uint32_t l) { uint64_t z = ((uint64_t)who << 20 ^ l) * 0x9E3779B97F4A7C15ull; return z ^ (z >> 29); };
        w1 = w2 = 0xFFFFFFFFu;
        for (uint32_t l : touched) if (w1 == 0xFFFFFFFFu || votes[l] > votes[w1] || (votes[l] == votes[w1] && mix(l) < mix(w1))) w1 = l;
        for (uint32_t l : touched) if (l != w1 && (w2 == 0xFFFFFFFFu || votes[l] > votes[w2] || (votes[l] == votes[w2] && mix(l) < mix(w2)))) w2 = l;
        for (uint32_t l : touched) votes[l] = 0;
        touched.clear();
    }
};

// ... from the attestation list itself (one process holds all of it): attesters are walked in visit order.
struct AttesterWalk {
    const PlanCommon& pc;
    HostGraph& g;
    unsigned T;
    std::vector<uint32_t> lr_of_edge;   // local row of edge k's destination, ~0 = dropped or another segment's
    std::vector<uint32_t> by_pos;       // attesters with edges into this share's rows, in visit order
    std::vector<uint32_t> chunk_cut;    // T + 1 cuts of by_pos with about equal edge counts
    std::vector<uint32_t> code_of_ext;  // the source index the share's edge records carry for each attester

    // Fills g.need / g.send and the geometry of the gather vector (geo.compact), then the visit order.
    AttesterWalk(const PlanCommon& pc_, HostGraph& g_, unsigned T_) : pc(pc_), g(g_), T(T_) {
        Geometry& geo = g.geo;
        const BuildInput& in = pc.in;
        const uint32_t stride = geo.seg_stride(), n = in.n;
        const uint32_t my_base = geo.seg * stride;
        const uint64_t n_kept = pc.out_ptr[n];
        lr_of_edge.resize(n_kept);
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = n_kept * t / T; k < n_kept * (t + 1) / T; ++k) {
                uint32_t lr = 0xFFFFFFFFu;
                if (!pc.drop[k]) {
                    const uint32_t p = g.pos_of_ext[in.to[pc.kept[k]]];
                    if (p >= my_base && p < my_base + geo.rows_cap) lr = p - my_base;
                }
                lr_of_edge[k] = lr;
            }
        });
        const uint32_t n_active = geo.n_active;
        code_of_ext.assign(n, 0xFFFFFFFFu);
        if (geo.compact) {
            // who is cited here, and which shares cite this share's own rows (bit c of `wanted`)
            std::vector<uint8_t> cited(n, 0), wanted(n, 0);
            run_threads(T, [&](unsigned t) {
                for (uint32_t f = (uint32_t)((uint64_t)n * t / T); f < (uint32_t)((uint64_t)n * (t + 1) / T); ++f) {
                    if (pc.dangling[f] || g.pos_of_ext[f] == 0xFFFFFFFFu) continue;
                    const bool mine = g.pos_of_ext[f] / stride == geo.seg;
                    uint8_t c = 0, w = 0;
                    for (uint64_t k = pc.out_ptr[f]; k < pc.out_ptr[f + 1]; ++k) {
                        if (pc.drop[k]) continue;
                        if (lr_of_edge[k] != 0xFFFFFFFFu) c = 1;
                        else if (mine) w |= (uint8_t)(1u << (g.pos_of_ext[in.to[pc.kept[k]]] / stride));
                    }
                    cited[f] = c;
                    wanted[f] = w;
                }
            });
            for (uint32_t p = 0; p < 8; ++p) { g.need[p].clear(); g.send[p].clear(); }
            // ranks ascend with the local row inside every share, so the lists come out ascending
            for (uint32_t r = 0; r < n_active; ++r) {
                const uint32_t f = pc.order[r], pos = g.pos_of_ext[f], p = pos / stride, local = pos % stride;
                if (p == geo.seg)
                    for (uint32_t c = 0; c < geo.n_seg; ++c) if (wanted[f] >> c & 1) g.send[c].push_back(local);
                if (cited[f]) g.need[p].push_back(local);      // the share's own rows too: their block of cited entries
            }
            g.send_known = true;
            uint32_t counts[8];
            for (uint32_t p = 0; p < 8; ++p) counts[p] = (uint32_t)g.need[p].size();
            geo.set_gather_blocks(counts, g.blk_off);
            std::vector<uint32_t> at(8, 0);
            for (uint32_t r = 0; r < n_active; ++r) {
                const uint32_t f = pc.order[r], p = g.pos_of_ext[f] / stride;
                if (cited[f]) code_of_ext[f] = g.blk_off[p] + at[p]++;
            }
            for (uint32_t r = 0; r < n_active; ++r) if (cited[pc.order[r]]) by_pos.push_back(pc.order[r]);
        } else {
            for (uint32_t r = 0; r < n_active; ++r) {
                const uint32_t f = pc.order[r], pos = g.pos_of_ext[f];
                code_of_ext[f] = ((pos / stride) << COL_SEG_SHIFT) | (pos % stride);
            }
            by_pos.assign(pc.order.begin(), pc.order.begin() + n_active);
        }
        sort_walk();
    }
    void sort_walk() {
        const Geometry& geo = g.geo;
        const uint32_t stride = geo.seg_stride();
        auto visit_key = [&](uint32_t ext) -> uint64_t {
            if (geo.compact) return code_of_ext[ext];
            const uint32_t p = g.pos_of_ext[ext];
            return ((uint64_t)(geo.n_chunks > 1 ? geo.chunk_of_local(p % stride) : 0u) << 32) | p;
        };
        // one share, one exchange per iteration: position = rank in `order`, which is how by_pos was filled
        const bool in_order = geo.n_seg == 1 && geo.n_chunks == 1 && std::is_sorted(by_pos.begin(), by_pos.end(), [&](uint32_t a, uint32_t b) { return visit_key(a) < visit_key(b); });
        if (!in_order) parallel_sort(by_pos.begin(), by_pos.end(), [&](uint32_t a, uint32_t b) { return visit_key(a) < visit_key(b); }, T);
        const uint32_t n_walk = (uint32_t)by_pos.size();
        std::vector<uint64_t> pos_edges((size_t)n_walk + 1, 0);
        for (uint32_t i = 0; i < n_walk; ++i) pos_edges[i + 1] = pos_edges[i] + (pc.out_ptr[by_pos[i] + 1] - pc.out_ptr[by_pos[i]]);
        chunk_cut.assign(T + 1, n_walk);
        chunk_cut[0] = 0;
        for (unsigned t = 1; t < T; ++t)
            chunk_cut[t] = (uint32_t)(std::lower_bound(pos_edges.begin(), pos_edges.end(), pos_edges[n_walk] * t / T) - pos_edges.begin());
        for (unsigned t = 1; t <= T; ++t) chunk_cut[t] = std::min<uint32_t>(n_walk, std::max(chunk_cut[t], chunk_cut[t - 1]));
        chunk_cut[T] = n_walk;
    }
    // Gather vectors: the other shares' entries in the order this share's workgroups want them (order_cited_sources);
    // wg_of_group = the workgroup that sums each 16-row group.  The shares that pack for this one must be told (g.need).
    void cluster_cited(HostGraph&, const std::vector<uint32_t>& wg_of_group, uint32_t n_wg, uint32_t hot, uint32_t fold) {
        const Geometry& geo = g.geo;
        const uint32_t stride = geo.seg_stride(), n_walk = (uint32_t)by_pos.size();
        std::vector<CitedSource> cs(n_walk);
        run_threads(T, [&](unsigned t) {
            VoteBox box(n_wg);
            for (uint32_t i = (uint32_t)((uint64_t)n_walk * t / T); i < (uint32_t)((uint64_t)n_walk * (t + 1) / T); ++i) {
                const uint32_t f = by_pos[i];
                uint32_t edges = 0;
                for (uint64_t k = pc.out_ptr[f]; k < pc.out_ptr[f + 1]; ++k)
                    if (lr_of_edge[k] != 0xFFFFFFFFu) { box.add(wg_of_group[lr_of_edge[k] / BLOCK_GROUP]); ++edges; }
                cs[i].local = g.pos_of_ext[f] % stride;
                cs[i].edges = edges;
                box.take(f, cs[i].wg1, cs[i].wg2);
            }
        });
        std::vector<std::vector<CitedSource>> per(geo.n_seg);
        for (uint32_t i = 0; i < n_walk; ++i) per[g.pos_of_ext[by_pos[i]] / stride].push_back(cs[i]);
        std::vector<uint32_t> ext_of_local;
        for (uint32_t p = 0; p < geo.n_seg; ++p) {
            ext_of_local.assign(stride, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < n_walk; ++i) {
                const uint32_t pos = g.pos_of_ext[by_pos[i]];
                if (pos / stride == p) ext_of_local[pos % stride] = by_pos[i];
            }
            order_cited_sources(per[p], hot, fold);
            g.need[p].resize(per[p].size());
            for (uint32_t j = 0; j < per[p].size(); ++j) {
                g.need[p][j] = per[p][j].local;
                code_of_ext[ext_of_local[per[p][j].local]] = g.blk_off[p] + j;
            }
        }
        g.send_known = false;        // the lists derived above assumed every share asks in ascending order: no longer true of the others
        sort_walk();
    }
    void count_rows(std::vector<uint64_t>& in_ptr) const {
        const uint32_t n = pc.in.n;
        run_threads(T, [&](unsigned t) {
            for (uint32_t f = (uint32_t)((uint64_t)n * t / T); f < (uint32_t)((uint64_t)n * (t + 1) / T); ++f) {
                if (pc.dangling[f]) continue;
                for (uint64_t k = pc.out_ptr[f]; k < pc.out_ptr[f + 1]; ++k) {
                    const uint32_t lr = lr_of_edge[k];
                    if (lr != 0xFFFFFFFFu) __atomic_fetch_add(&in_ptr[lr + 1], 1, __ATOMIC_RELAXED);
                }
            }
        });
    }
    template <typename V>
    void walk(unsigned t, V visit) const {
        for (uint32_t i = chunk_cut[t]; i < chunk_cut[t + 1]; ++i) {
            const uint32_t f = by_pos[i];
            const uint32_t pf = code_of_ext[f];
            for (uint64_t k = pc.out_ptr[f]; k < pc.out_ptr[f + 1]; ++k) {
                const uint32_t lr = lr_of_edge[k];
                if (lr != 0xFFFFFFFFu) visit(lr, pf, pc.kept[k]);
            }
        }
    }
    uint32_t value_u32(uint64_t h) const { return pc.in.val_u32[h]; }
    const FrRaw& value_fr(uint64_t h) const { return pc.in.val_fr[h]; }
};

// ... from edge records that other processes sent: (source code = (segment << 29) | local row, local row of the
// destination, integer opinion), any order on arrival
struct RecordWalk {
    Geometry geo;
    unsigned T;
    uint64_t count;
    std::vector<uint32_t> src, row, val;   // sorted by the source's visit key; src = the source index the edge records carry

    // Sorts the records; with a gather vector also fills g.need and the geometry of the gather vector.
    RecordWalk(HostGraph& g, unsigned T_, const std::vector<uint32_t>& src_in, const std::vector<uint32_t>& row_in,
               const std::vector<uint32_t>& val_in) : geo(g.geo), T(T_), count(src_in.size()) {
        const uint32_t stride = geo.seg_stride();
        // visit key: position (piece by piece with a chunked exchange); gather vector: the block of the share's own cited
        // rows first, then the other shares' in ascending order, local rows ascending inside a block
        auto block_rank = [&](uint32_t p) -> uint32_t { return !geo.compact ? p : (p == geo.seg ? 0u : (p < geo.seg ? p + 1 : p)); };
        auto key_of = [&](uint32_t code) -> uint64_t {
            const uint32_t local = code & COL_LOCAL_MASK;
            return ((uint64_t)(geo.n_chunks > 1 ? geo.chunk_of_local(local) : 0u) << 32) | ((uint64_t)block_rank(code >> COL_SEG_SHIFT) * stride + local);
        };
        // sort by visit key: T buckets of equal key ranges (sources are spread evenly over the positions), each sorted by its owner
        // bucket = position of the key in [0, key_end) taken over (piece, position / vector length)
        const uint64_t vec_len = (uint64_t)geo.n_seg * stride;
        auto bucket_of = [&](uint64_t key) -> unsigned {
            const long double frac = ((long double)(key >> 32) + (long double)(key & 0xFFFFFFFFull) / (long double)vec_len) / (long double)geo.n_chunks;
            return (unsigned)std::min<long double>(T - 1, frac * T);
        };
        std::vector<uint64_t> cnt((size_t)T * T, 0);
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) ++cnt[(size_t)t * T + bucket_of(key_of(src_in[k]))];
        });
        std::vector<uint64_t> begin(T + 1, 0);
        {
            uint64_t run = 0;
            for (unsigned d = 0; d < T; ++d) {
                begin[d] = run;
                for (unsigned t = 0; t < T; ++t) { const uint64_t c = cnt[(size_t)t * T + d]; cnt[(size_t)t * T + d] = run; run += c; }
            }
            begin[T] = run;
        }
        struct Rec { uint64_t key; uint32_t row, val; };
        std::vector<Rec> buf(count);
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) {
                const uint64_t key = key_of(src_in[k]);
                buf[cnt[(size_t)t * T + bucket_of(key)]++] = Rec{key, row_in[k], val_in[k]};
            }
        });
        run_threads(T, [&](unsigned d) {
            std::sort(buf.begin() + begin[d], buf.begin() + begin[d + 1], [](const Rec& a, const Rec& b) {
                if (a.key != b.key) return a.key < b.key;
                return a.row < b.row;
            });
        });
        src.resize(count); row.resize(count); val.resize(count);
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) { row[k] = buf[k].row; val[k] = buf[k].val; }
        });
        if (!geo.compact) {
            run_threads(T, [&](unsigned t) {
                for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) {
                    const uint32_t pos = (uint32_t)(buf[k].key & 0xFFFFFFFFull);
                    src[k] = ((pos / stride) << COL_SEG_SHIFT) | (pos % stride);
                }
            });
            return;
        }
        // gather vector: the distinct sources of every share (this one included), ascending, are what this share needs from it
        auto share_of_rank = [&](uint32_t br) -> uint32_t { return br == 0 ? geo.seg : (br <= geo.seg ? br - 1 : br); };
        for (uint32_t p = 0; p < 8; ++p) { g.need[p].clear(); }
        for (uint64_t k = 0; k < count; ++k) {
            const uint32_t v = (uint32_t)(buf[k].key & 0xFFFFFFFFull);
            if (k && buf[k - 1].key == buf[k].key) continue;
            g.need[share_of_rank(v / stride)].push_back(v % stride);
        }
        uint32_t counts[8];
        for (uint32_t p = 0; p < 8; ++p) counts[p] = (uint32_t)g.need[p].size();
        g.geo.set_gather_blocks(counts, g.blk_off);
        geo = g.geo;
        run_threads(T, [&](unsigned t) {
            uint64_t k = count * t / T;
            const uint64_t k1 = count * (t + 1) / T;
            while (k < k1) {
                const uint32_t v = (uint32_t)(buf[k].key & 0xFFFFFFFFull), p = share_of_rank(v / stride), local = v % stride;
                const uint32_t code = g.blk_off[p] + (uint32_t)(std::lower_bound(g.need[p].begin(), g.need[p].end(), local) - g.need[p].begin());
                const uint64_t key = buf[k].key;
                for (; k < k1 && buf[k].key == key; ++k) src[k] = code;
            }
        });
    }
    // see AttesterWalk::cluster_cited
    void cluster_cited(HostGraph& g, const std::vector<uint32_t>& wg_of_group, uint32_t n_wg, uint32_t hot, uint32_t fold) {
        const uint32_t stride = geo.seg_stride();
        std::vector<std::vector<CitedSource>> per(geo.n_seg);
        std::vector<std::vector<uint32_t>> old_code(geo.n_seg);
        {
            VoteBox box(n_wg);
            uint64_t k = 0;
            while (k < count) {
                const uint32_t code = src[k];
                uint32_t edges = 0;
                for (; k < count && src[k] == code; ++k) { box.add(wg_of_group[row[k] / BLOCK_GROUP]); ++edges; }
                CitedSource c{0, edges, 0, 0};
                box.take(code, c.wg1, c.wg2);
                uint32_t p = geo.n_seg;
                for (uint32_t q = 0; q < geo.n_seg; ++q) if (code >= g.blk_off[q] && code - g.blk_off[q] < g.need[q].size()) p = q;
                if (p == geo.n_seg) continue;                                   // cannot happen: every code lies in one block
                c.local = g.need[p][code - g.blk_off[p]];
                per[p].push_back(c);
            }
        }
        std::vector<uint32_t> remap(geo.vec_len);
        for (uint32_t i = 0; i < geo.vec_len; ++i) remap[i] = i;
        std::vector<uint32_t> code_of_local;
        for (uint32_t p = 0; p < geo.n_seg; ++p) {
            code_of_local.assign(stride, 0xFFFFFFFFu);
            for (uint32_t j = 0; j < g.need[p].size(); ++j) code_of_local[g.need[p][j]] = g.blk_off[p] + j;
            order_cited_sources(per[p], hot, fold);
            for (uint32_t j = 0; j < per[p].size(); ++j) {
                remap[code_of_local[per[p][j].local]] = g.blk_off[p] + j;
                g.need[p][j] = per[p][j].local;
            }
        }
        // the records again in ascending order of their (new) source index
        struct Rec { uint32_t src, row, val; };
        std::vector<uint64_t> cnt((size_t)T * T, 0);
        auto bucket_of = [&](uint32_t c) -> unsigned { return (unsigned)std::min<uint64_t>(T - 1, (uint64_t)c * T / geo.vec_len); };
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) { src[k] = remap[src[k]]; ++cnt[(size_t)t * T + bucket_of(src[k])]; }
        });
        std::vector<uint64_t> begin(T + 1, 0);
        {
            uint64_t run = 0;
            for (unsigned d = 0; d < T; ++d) {
                begin[d] = run;
                for (unsigned t = 0; t < T; ++t) { const uint64_t c = cnt[(size_t)t * T + d]; cnt[(size_t)t * T + d] = run; run += c; }
            }
            begin[T] = run;
        }
        std::vector<Rec> buf(count);
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) buf[cnt[(size_t)t * T + bucket_of(src[k])]++] = Rec{src[k], row[k], val[k]};
        });
        run_threads(T, [&](unsigned d) {
            std::sort(buf.begin() + begin[d], buf.begin() + begin[d + 1], [](const Rec& a, const Rec& b) { return a.src != b.src ? a.src < b.src : a.row < b.row; });
        });
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) { src[k] = buf[k].src; row[k] = buf[k].row; val[k] = buf[k].val; }
        });
    }
    void count_rows(std::vector<uint64_t>& in_ptr) const {
        run_threads(T, [&](unsigned t) {
            for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) __atomic_fetch_add(&in_ptr[row[k] + 1], 1, __ATOMIC_RELAXED);
        });
    }
    template <typename V>
    void walk(unsigned t, V visit) const {
        for (uint64_t k = count * t / T; k < count * (t + 1) / T; ++k) visit(row[k], src[k], k);
    }
    uint32_t value_u32(uint64_t h) const { return val[h]; }
    const FrRaw& value_fr(uint64_t) const { static const FrRaw zero{{0, 0, 0, 0}}; return zero; }   // records carry integer opinions only
};

// Steps 7-10: one share's edge lists.  `g` holds the common part (geometry with geo.seg set, positions, factors).
// In two parts: deal_segment counts the rows' in-edges, hands the row groups to the sweeps' workgroups and -- gather
// vectors with cluster_cited -- settles the order of the other shares' entries (g.need; the shares that pack for this one
// are told between the two parts); fill_segment writes the edge lists.
struct Deal { std::vector<uint32_t> block, slot; };
struct DealtSegment {
    std::vector<uint64_t> in_ptr;
    Deal d64, dfr;
};

template <class Src>
int deal_segment(HostGraph& g, const BuildOpts& opt, Src& src, DealtSegment& ds) {
    const Geometry geo = g.geo;
    for (int x = 0; x < 8; ++x) g.xcd_cost[x] = opt.xcd_cost[x];
    const uint32_t G = geo.n_seg;
    std::vector<uint64_t>& in_ptr = ds.in_ptr;
    // ---- 7. in-edges of this segment's rows, sources ascending by position -------------------
    // The passes below scatter every surviving edge by destination row; they run on several host
    // threads, each owning a contiguous range of rows (or of workgroups), so that no two threads
    // write the same counter and every thread emits its edges in the same, source-ascending order.
    in_ptr.assign((size_t)geo.rows_cap + 1, 0);
    src.count_rows(in_ptr);
    for (uint32_t i = 0; i < geo.rows_cap; ++i) in_ptr[i + 1] += in_ptr[i];
    g.nnz_local = in_ptr[geo.rows_cap];
    // The same edges grouped by workgroup of the f64 blocked sweep, ascending by source.  The row
    // groups are handed out heaviest first, each to the workgroup with the fewest edges so far that
    // still has a free accumulator slot: a launch ends with its slowest workgroup, so edge counts
    // must be level.  Groups of dangling rows go first, to the leading `dblocks` workgroups.
    const uint32_t n_groups = geo.rows_cap / BLOCK_GROUP, first_dangling_group = geo.n_active_pad / BLOCK_GROUP / G;
    std::vector<uint64_t> grp_edges(n_groups);
    for (uint32_t gi = 0; gi < n_groups; ++gi) grp_edges[gi] = in_ptr[(size_t)(gi + 1) * BLOCK_GROUP] - in_ptr[(size_t)gi * BLOCK_GROUP];
    auto deal = [&](uint32_t B, uint32_t gpb, uint32_t dblocks, std::vector<uint32_t>& grp_table, std::vector<uint64_t>& ptr) {
        Deal d;
        d.block.assign(n_groups, 0);
        d.slot.assign(n_groups, 0);
        grp_table.assign((size_t)B * gpb, 0xFFFFFFFFu);
        ptr.assign((size_t)B + 1, 0);
        std::vector<uint32_t> used(B, 0);
        std::vector<uint64_t> load(B, 0), edges(B, 0);   // load: edges weighted by the cost of the workgroup's XCD
        typedef std::pair<uint64_t, uint32_t> Load;   // (load so far, workgroup)
        auto place = [&](uint32_t gi, uint32_t bb) {
            d.block[gi] = bb;
            d.slot[gi] = used[bb];
            grp_table[(size_t)bb * gpb + used[bb]] = gi;
            load[bb] += grp_edges[gi] * opt.xcd_cost[bb % 8];
            edges[bb] += grp_edges[gi];
            ++used[bb];
        };
        auto pass = [&](uint32_t g0, uint32_t g1, uint32_t n_blocks, bool hinted) {
            std::vector<uint32_t> by_weight;
            by_weight.reserve(g1 - g0);
            // groups placed with the sources their workgroup gathers (cluster_cold_sources) go where they were meant to
            // (the planner clusters for the sweep with the most workgroups; a sweep with a divisor of that count gives each of its
            // workgroups several whole clusters)
            for (uint32_t gi = g0; gi < g1; ++gi) {
                uint32_t h = hinted && gi < g.grp_hint.size() ? g.grp_hint[gi] : 0xFFFFFFFFu;
                if (h != 0xFFFFFFFFu) h %= n_blocks;
                if (h < n_blocks && used[h] < gpb) place(gi, h);
                else by_weight.push_back(gi);
            }
            std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t x, uint32_t y) { return grp_edges[x] > grp_edges[y]; });
            std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
            for (uint32_t bb = 0; bb < n_blocks; ++bb) if (used[bb] < gpb) heap.push(Load(load[bb], bb));
            for (uint32_t gi : by_weight) {
                const uint32_t bb = heap.top().second;
                heap.pop();
                place(gi, bb);
                if (used[bb] < gpb) heap.push(Load(load[bb], bb));   // a full workgroup leaves the heap
            }
        };
        pass(first_dangling_group, n_groups, dblocks, false);
        pass(0, first_dangling_group, B, g.hint_blocks != 0 && g.hint_blocks % B == 0);
        for (uint32_t bb = 0; bb < B; ++bb) ptr[bb + 1] = ptr[bb] + edges[bb];
        return d;
    };
    // The exact sweep's deal first.  Gather vectors whose entries are ordered by citing workgroup (cluster_cited): the f64
    // sweep's workgroup b then takes the row groups of the exact sweep's workgroups b, b + f64_blocks, ..., so that both
    // sweeps find a workgroup's sources side by side.
    Deal& dfr = ds.dfr;
    if (geo.fr_blocks) dfr = deal(geo.fr_blocks, geo.fr_rb / BLOCK_GROUP, geo.fr_dblocks, g.fb_grp, g.fb_ptr);
    const bool cluster = opt.cluster_cited && geo.compact && geo.fr_blocks != 0 && geo.fr_blocks % geo.f64_blocks == 0 &&
                         geo.fr_rb >= 1024 && !getenv("ET_NO_CLUSTER_SHARES");
    if (cluster) {
        g.grp_hint.assign(dfr.block.begin(), dfr.block.begin() + first_dangling_group);
        g.hint_blocks = geo.fr_blocks;
    }
    // Equally many workgroups in both sweeps and room for the exact sweep's rows in the f64 sweep's workgroups: one deal
    // serves both (it is levelled by the same edge counts), and the edge lists are then written once (fill_segment)
    const uint32_t gpb64 = geo.f64_rb / BLOCK_GROUP;
    if (geo.fr_blocks == geo.f64_blocks && geo.f64_rb >= geo.fr_rb && geo.f64_dblocks >= geo.fr_dblocks && !getenv("ET_SEPARATE_DEALS")) {
        ds.d64 = dfr;
        g.bk_ptr = g.fb_ptr;
        g.bk_grp.assign((size_t)geo.f64_blocks * gpb64, 0xFFFFFFFFu);
        for (uint32_t gi = 0; gi < n_groups; ++gi) g.bk_grp[(size_t)dfr.block[gi] * gpb64 + dfr.slot[gi]] = gi;
    } else
    ds.d64 = deal(geo.f64_blocks, gpb64, geo.f64_dblocks, g.bk_grp, g.bk_ptr);
    if (cluster) {
        // sources cited this often by the share's rows stay in front, in position order: nearly every workgroup gathers them
        const uint32_t hot = getenv("ET_CLUSTER_HOT") ? (uint32_t)atoi(getenv("ET_CLUSTER_HOT")) : 16u;   // 8 / 16 / 32 measured: 140.5 / 131.3 / 134.6 us (f64 share of eight)
        src.cluster_cited(g, dfr.block, geo.fr_blocks, hot, geo.f64_blocks);
    }
    return ET_OK;
}

template <class Src>
int fill_segment(HostGraph& g, const BuildOpts& opt, const Src& src, unsigned T, const DealtSegment& ds, std::string& err) {
    const Geometry geo = g.geo;
    const uint32_t seg_index = geo.seg;
    const uint32_t G = geo.n_seg, n_active = geo.n_active, n_dangling = geo.n_dangling;
    const uint32_t target_work = opt.target_work ? opt.target_work : 16384;
    const std::vector<uint64_t>& in_ptr = ds.in_ptr;
    const Deal& d64 = ds.d64;
    const Deal& dfr = ds.dfr;
    const bool per_row_stream = geo.fr_blocks == 0;      // else the exact variant runs the blocked sweep too
    std::vector<uint32_t> e_src(per_row_stream ? g.nnz_local : 0);   // source index as the edge records carry it
    std::vector<uint64_t> e_idx(per_row_stream ? g.nnz_local : 0);   // the edge's value handle
    g.bk_col.resize(g.nnz_local);
    g.bk_row.resize(g.nnz_local);
    if (g.values_are_u32) g.bk_val_u32.resize(g.nnz_local);
    else g.bk_val_fr.resize(g.nnz_local);
    {
        std::vector<uint64_t> cur(in_ptr.begin(), in_ptr.end() - 1);
        std::vector<uint64_t> bcur(g.bk_ptr.begin(), g.bk_ptr.end() - 1);
        // Every thread walks ONE contiguous chunk of the visit order and drops each edge into the bucket of the thread that
        // owns its destination; a bucket therefore holds its edges in ascending source order, and its owner appends them to
        // the rows (or workgroups) it owns.
        struct Entry { uint32_t lr, pf; uint64_t h; };
        auto scatter = [&](auto bucket_of, auto consume) {
            std::vector<uint64_t> cnt((size_t)T * T, 0);
            run_threads(T, [&](unsigned t) { src.walk(t, [&](uint32_t lr, uint32_t, uint64_t) { ++cnt[(size_t)t * T + bucket_of(lr)]; }); });
            std::vector<uint64_t> bucket_begin(T + 1, 0);
            {
                uint64_t run = 0;
                for (unsigned d = 0; d < T; ++d) {
                    bucket_begin[d] = run;
                    for (unsigned t = 0; t < T; ++t) { const uint64_t c = cnt[(size_t)t * T + d]; cnt[(size_t)t * T + d] = run; run += c; }
                }
                bucket_begin[T] = run;
            }
            std::vector<Entry> buf(bucket_begin[T]);
            run_threads(T, [&](unsigned t) { src.walk(t, [&](uint32_t lr, uint32_t pf, uint64_t h) { buf[cnt[(size_t)t * T + bucket_of(lr)]++] = Entry{lr, pf, h}; }); });
            run_threads(T, [&](unsigned d) { for (uint64_t i = bucket_begin[d]; i < bucket_begin[d + 1]; ++i) consume(buf[i]); });
        };
        // (a) per-row lists: thread t owns the rows whose edges fall in its share of the edge count
        std::vector<uint32_t> row_cut(T + 1, geo.rows_cap);
        row_cut[0] = 0;
        for (unsigned t = 1; t < T; ++t)
            row_cut[t] = (uint32_t)(std::lower_bound(in_ptr.begin(), in_ptr.end(), g.nnz_local * t / T) - in_ptr.begin());
        for (unsigned t = 1; t <= T; ++t) row_cut[t] = std::max(row_cut[t], row_cut[t - 1]);
        row_cut[T] = geo.rows_cap;
        if (per_row_stream)
            scatter([&](uint32_t lr) { return (unsigned)(std::upper_bound(row_cut.begin() + 1, row_cut.end() - 1, lr) - (row_cut.begin() + 1)); },
                    [&](const Entry& e) {
                        const uint64_t p = cur[e.lr]++;
                        e_src[p] = e.pf;
                        e_idx[p] = e.h;
                    });
        const double t_b = now_s();
        // (b) per-workgroup lists: thread t owns a contiguous range of workgroups
        const uint32_t B64 = geo.f64_blocks;
        std::vector<uint32_t> owner_of_block(B64, 0);
        for (unsigned t = 0; t < T; ++t)
            for (uint32_t bb = (uint32_t)((uint64_t)B64 * t / T); bb < (uint32_t)((uint64_t)B64 * (t + 1) / T); ++bb) owner_of_block[bb] = t;
        scatter([&](uint32_t lr) { return (unsigned)owner_of_block[d64.block[lr / BLOCK_GROUP]]; },
                [&](const Entry& e) {
                    const uint32_t grp = e.lr / BLOCK_GROUP;
                    const uint64_t q = bcur[d64.block[grp]]++;
                    g.bk_col[q] = e.pf;
                    g.bk_row[q] = (uint16_t)(d64.slot[grp] * BLOCK_GROUP + e.lr % BLOCK_GROUP);
                    if (g.values_are_u32) g.bk_val_u32[q] = src.value_u32(e.h);
                    else g.bk_val_fr[q] = src.value_fr(e.h);
                });
        const double t_c = now_s();
        // (c) the workgroups of the exact blocked sweep, when the geometry asks for them
        if (getenv("ET_BUILD_STATS")) fprintf(stderr, "[et build] f64 lists %.3f s (from %.3f)\n", t_c - t_b, t_b);
        if (geo.fr_blocks) {
            const bool wide = geo.fr_compact == 2;
            g.fb_col.resize(g.nnz_local);
            g.fb_row.resize(g.nnz_local);
            if (wide) g.fb_val_u32.resize(g.nnz_local);
            else g.fb_val.resize(g.nnz_local);
            std::vector<uint64_t> fcur(g.fb_ptr.begin(), g.fb_ptr.end() - 1);
            const uint32_t BF = geo.fr_blocks;
            std::vector<uint32_t> owner_fr(BF, 0);
            for (unsigned t = 0; t < T; ++t)
                for (uint32_t bb = (uint32_t)((uint64_t)BF * t / T); bb < (uint32_t)((uint64_t)BF * (t + 1) / T); ++bb) owner_fr[bb] = t;
            // Both sweeps gave every row group to the same workgroup (the usual case when they run equally many): the exact
            // sweep's lists are the f64 sweep's with the rows' accumulator slots renamed, no second pass over the edges
            if (g.values_are_u32 && BF == B64 && dfr.block == d64.block && g.fb_ptr == g.bk_ptr) {
                const uint32_t gpb64 = geo.f64_rb / BLOCK_GROUP;
                run_threads(T, [&](unsigned t) {
                    for (uint32_t bb = (uint32_t)((uint64_t)BF * t / T); bb < (uint32_t)((uint64_t)BF * (t + 1) / T); ++bb)
                        for (uint64_t q = g.bk_ptr[bb]; q < g.bk_ptr[bb + 1]; ++q) {
                            const uint32_t slot64 = g.bk_row[q], grp = g.bk_grp[(size_t)bb * gpb64 + slot64 / BLOCK_GROUP];
                            g.fb_col[q] = g.bk_col[q];
                            g.fb_row[q] = (uint16_t)(dfr.slot[grp] * BLOCK_GROUP + slot64 % BLOCK_GROUP);
                            if (wide) g.fb_val_u32[q] = g.bk_val_u32[q];
                            else g.fb_val[q] = (uint8_t)g.bk_val_u32[q];
                        }
                });
            } else
            scatter([&](uint32_t lr) { return (unsigned)owner_fr[dfr.block[lr / BLOCK_GROUP]]; },
                    [&](const Entry& e) {
                        const uint32_t grp = e.lr / BLOCK_GROUP;
                        const uint64_t q = fcur[dfr.block[grp]]++;
                        g.fb_col[q] = e.pf;
                        g.fb_row[q] = (uint16_t)(dfr.slot[grp] * BLOCK_GROUP + e.lr % BLOCK_GROUP);
                        if (wide) g.fb_val_u32[q] = src.value_u32(e.h);
                        else g.fb_val[q] = (uint8_t)src.value_u32(e.h);
                    });
        }
    }
    if (getenv("ET_BUILD_STATS")) {
        fprintf(stderr, "[et build] edge lists done at %.3f s\n", now_s());
        for (int exact = 0; exact < 2; ++exact) {
            const SweepStats ss = sweep_stats(g, exact != 0);
            if (ss.blocks)
                fprintf(stderr, "[et build] %s sweep: %u workgroups, %llu edges (most %llu, mean %.1f), %llu line fills (most %llu, mean %.1f), clustered %u, widest 64-edge chunk %llu positions, %llu lines over the eight L2s\n",
                        exact ? "exact" : "f64", ss.blocks, (unsigned long long)ss.edges, (unsigned long long)ss.edges_most, (double)ss.edges / ss.blocks,
                        (unsigned long long)ss.line_fills, (unsigned long long)ss.line_fills_most, (double)ss.line_fills / ss.blocks, ss.clustered,
                        (unsigned long long)ss.chunk_span_most, (unsigned long long)ss.xcd_line_fills);
        }
    }
    if (geo.n_chunks > 1) {
        // a list is sorted by piece: the runs' boundaries by binary search on the piece of an entry's source
        auto runs = [&](const std::vector<uint64_t>& ptr, const std::vector<uint32_t>& col, uint32_t B, std::vector<uint64_t>& out) {
            const uint32_t K = geo.n_chunks;
            out.assign((size_t)B * (K + 1), 0);
            for (uint32_t b = 0; b < B; ++b) {
                const uint32_t* lo = col.data() + ptr[b];
                const uint32_t* hi = col.data() + ptr[b + 1];
                for (uint32_t q = 0; q <= K; ++q)
                    out[(size_t)b * (K + 1) + q] = ptr[b] + (std::partition_point(lo, hi, [&](uint32_t c) { return geo.chunk_of_local(c & COL_LOCAL_MASK) < q; }) - lo);
            }
        };
        runs(g.bk_ptr, g.bk_col, geo.f64_blocks, g.bk_phase);
        runs(g.fb_ptr, g.fb_col, geo.fr_blocks, g.fb_phase);
    }
    // ---- 8. slices -----------------------------------------------------------------------------
    g.n_dangling_local = 0;
    {
        const uint32_t active_local = geo.n_active_pad / G;       // local rows before the dangling class
        for (uint32_t lr = active_local; lr < geo.rows_cap; ++lr) {
            const uint64_t q = ((uint64_t)(lr >> 6) * G + seg_index) * 64 + (lr & 63);
            if (q < (uint64_t)geo.n_active_pad + n_dangling) ++g.n_dangling_local;
        }
    }
    if (geo.fr_blocks != 0) return ET_OK;   // the exact variant runs the blocked sweep: no per-row stream is needed
    const uint32_t long_row = opt.long_row ? opt.long_row : 128;
    const uint32_t n_local_slices = geo.rows_cap / 64;
    const uint32_t active_local_slices = geo.n_active_pad / 64 / G;
    std::vector<uint8_t> slice_dangling;
    uint64_t steps = 0;
    auto real_rows_in_slice = [&](uint32_t ls) -> uint32_t {
        uint64_t q0 = ((uint64_t)ls * G + seg_index) * 64;  // global row of lane 0
        uint64_t end = ls < active_local_slices ? n_active : (uint64_t)geo.n_active_pad + n_dangling;
        if (q0 >= end) return 0;
        return (uint32_t)std::min<uint64_t>(64, end - q0);
    };
    // lanes per ordinary row: one for large graphs; small graphs spread a row over 2^s lanes until the
    // segment yields about 2048 wavefronts (two per SIMD), but never more lanes than the average row has
    // edges.  More lanes shorten a row's walk but multiply the per-row field arithmetic, which only one
    // lane of a group performs: on a 64K-peer / 1M-edge graph s = 0, 1, 2, 3 take 36.7, 25.9, 33.1, 47.1 us.
    uint32_t s_base = 0;
    {
        const uint64_t avg = geo.rows_cap ? g.nnz_local / std::max<uint64_t>(1, (uint64_t)n_active / G) : 0;
        while (s_base < 5 && ((uint64_t)n_local_slices << (s_base + 1)) <= 2048 && (2ull << s_base) <= std::max<uint64_t>(1, avg)) ++s_base;
        if (const char* e = getenv("ET_LANES_LOG2")) s_base = (uint32_t)std::min(5, std::max(0, atoi(e)));   // experiment knob
    }
    const uint32_t L = 1u << s_base, rows_per_slice = 64 / L;
    for (uint32_t ls = 0; ls < n_local_slices; ++ls) {
        uint32_t nr = real_rows_in_slice(ls);
        if (nr == 0) continue;
        bool dang = ls >= active_local_slices;
        for (uint32_t r0 = 0; r0 < nr; r0 += rows_per_slice) {
            const uint32_t cnt = std::min(rows_per_slice, nr - r0);
            uint32_t maxd = 0;
            for (uint32_t l = 0; l < cnt; ++l) maxd = std::max<uint32_t>(maxd, (uint32_t)(in_ptr[ls * 64 + r0 + l + 1] - in_ptr[ls * 64 + r0 + l]));
            if (maxd > long_row * L && s_base < 6) {
                for (uint32_t l = 0; l < cnt; ++l) {
                    uint32_t d = (uint32_t)(in_ptr[ls * 64 + r0 + l + 1] - in_ptr[ls * 64 + r0 + l]);
                    SliceDesc sd{(uint32_t)steps, (d + 63) / 64, ls * 64 + r0 + l, (6u << 8) | 1u};
                    steps += sd.len;
                    g.slices.push_back(sd);
                    slice_dangling.push_back(dang);
                }
            } else {
                SliceDesc sd{(uint32_t)steps, (maxd + L - 1) / L, ls * 64 + r0, (s_base << 8) | cnt};
                steps += sd.len;
                g.slices.push_back(sd);
                slice_dangling.push_back(dang);
            }
            if (steps >= 0xFFFFFFFFull) {
                err = "edge stream exceeds 2^32 steps";
                return ET_E_UNSUPPORTED;
            }
        }
    }
    g.n_steps = steps;
    {
        uint32_t check = 0;
        for (uint32_t ls = active_local_slices; ls < n_local_slices; ++ls) check += real_rows_in_slice(ls);
        if (check != g.n_dangling_local) { err = "internal: dangling row count"; return ET_E_UNSUPPORTED; }
    }
    // ---- 9. fill the edge stream -----------------------------------------------------------------
    const size_t slots = (size_t)steps * 64;
    g.col.assign(slots, 0);
    if (g.values_are_u32) g.val_u32.assign(slots, 0);
    else g.val_fr.assign(slots, FrRaw{{0, 0, 0, 0}});
    run_threads(T, [&](unsigned t) {
      const size_t s0 = g.slices.size() * t / T, s1 = g.slices.size() * (t + 1) / T;
      for (size_t si = s0; si < s1; ++si) {   // slices own disjoint slots
        const SliceDesc& sd = g.slices[si];
        // entry (step k, lane l): edge k * 2^s + (l & (2^s - 1)) of row row0 + (l >> s)
        const uint32_t sh = slice_log2_lanes(sd.nrows), lanes = 1u << sh;
        for (uint32_t l = 0; l < slice_rows(sd.nrows); ++l) {
            uint64_t b = in_ptr[sd.row0 + l], e = in_ptr[sd.row0 + l + 1];
            for (uint64_t k = b; k < e; ++k) {
                const uint64_t j = k - b;
                size_t slot = ((size_t)sd.off + j / lanes) * 64 + ((size_t)l << sh) + j % lanes;
                g.col[slot] = e_src[k];
                if (g.values_are_u32) g.val_u32[slot] = src.value_u32(e_idx[k]);
                else g.val_fr[slot] = src.value_fr(e_idx[k]);
            }
        }
      }
    });
    // ---- 10. wavefront work items: contiguous, class-pure, equal estimated cost ------------------
    const uint32_t n_slices = (uint32_t)g.slices.size();
    uint32_t first_dang = n_slices;
    for (uint32_t s = 0; s < n_slices; ++s) if (slice_dangling[s]) { first_dang = s; break; }
    auto cost = [&](uint32_t s) -> uint64_t { return (uint64_t)g.slices[s].len + 6; };
    uint64_t cost_a = 0, cost_d = 0;
    for (uint32_t s = 0; s < first_dang; ++s) cost_a += cost(s);
    for (uint32_t s = first_dang; s < n_slices; ++s) cost_d += cost(s);
    auto partition = [&](uint32_t sb, uint32_t se, uint32_t max_items, bool dang) {
        if (sb >= se) return;
        uint64_t total = 0;
        for (uint32_t s = sb; s < se; ++s) total += cost(s);
        uint32_t nb = std::max<uint32_t>(1, std::min<uint32_t>(max_items, se - sb));
        uint64_t per = (total + nb - 1) / nb;
        uint32_t s = sb, dslot = 0;
        while (s < se) {
            uint64_t acc = 0;
            uint32_t b = s;
            while (s < se && (acc < per || s == b)) acc += cost(s++);
            g.work.push_back(WorkDesc{b, s, dang ? WORK_DANGLING : 0u, dang ? dslot++ : 0u});
        }
    };
    uint32_t items_d = 0;
    if (cost_d) items_d = (uint32_t)std::min<uint64_t>(geo.nd_cap, std::max<uint64_t>(1, target_work * cost_d / (cost_a + cost_d)));
    partition(0, first_dang, std::max<uint32_t>(1, target_work - items_d), false);
    size_t n_active_items = g.work.size();
    partition(first_dang, n_slices, items_d, true);
    if (g.work.size() - n_active_items > geo.nd_cap) {
        err = "internal: more dangling work items than partial-sum slots";
        return ET_E_UNSUPPORTED;
    }
    return ET_OK;
}

template <class Src>
int lay_out_segment(HostGraph& g, const BuildOpts& opt, Src& src, unsigned T, std::string& err) {
    const double t0 = now_s();
    DealtSegment ds;
    int rc = deal_segment(g, opt, src, ds);
    if (rc) return rc;
    const double t1 = now_s();
    rc = fill_segment(g, opt, src, T, ds, err);
    if (getenv("ET_BUILD_STATS")) fprintf(stderr, "[et build] deal %.3f s, edge lists %.3f s\n", t1 - t0, now_s() - t1);
    return rc;
}

}  // namespace

SweepStats sweep_stats(const HostGraph& g, bool exact) {
    SweepStats ss;
    const std::vector<uint64_t>& ptr = exact ? g.fb_ptr : g.bk_ptr;
    const std::vector<uint32_t>& col = exact ? g.fb_col : g.bk_col;
    const uint32_t B = exact ? g.geo.fr_blocks : g.geo.f64_blocks, shift = exact ? 2 : 4;
    if (B == 0 || ptr.size() != (size_t)B + 1 || col.size() < ptr[B]) return ss;
    ss.blocks = B;
    ss.clustered = g.hint_blocks != 0 && g.hint_blocks % B == 0 && !g.grp_hint.empty();
    ss.edges = ptr[B];
    // which of the eight L2s pull each line of the vector (workgroup b runs on XCD b mod 8)
    std::vector<uint8_t> xcds_of_line((((size_t)g.geo.n_seg * g.geo.seg_stride() + g.geo.vec_len) >> shift) + 1, 0);
    for (uint32_t b = 0; b < B; ++b) {
        uint64_t c = 0;
        for (uint64_t k = ptr[b]; k < ptr[b + 1]; ++k) {
            c += k == ptr[b] || (col[k] >> shift) != (col[k - 1] >> shift);
            xcds_of_line[g.geo.pos_of_col(col[k]) >> shift] |= (uint8_t)(1u << (b % 8));
        }
        // widest range of positions that 64 consecutive edges of a list span (the packed edge records hold a 16-bit offset)
        for (uint64_t k = ptr[b]; k < ptr[b + 1]; k += 64) {
            const uint64_t last = std::min<uint64_t>(k + 63, ptr[b + 1] - 1);
            ss.chunk_span_most = std::max<uint64_t>(ss.chunk_span_most, g.geo.pos_of_col(col[last]) - g.geo.pos_of_col(col[k]));
        }
        ss.line_fills += c;
        ss.line_fills_most = std::max(ss.line_fills_most, c);
        ss.edges_most = std::max(ss.edges_most, ptr[b + 1] - ptr[b]);
    }
    for (uint8_t m : xcds_of_line) ss.xcd_line_fills += (uint64_t)__builtin_popcount(m);
    return ss;
}

int plan_segment(const PlanCommon& pc, const BuildOpts& opt, uint32_t seg_index, HostGraph& g, std::string& err, bool last_share) {
    if (seg_index >= pc.g.geo.n_seg) { err = "bad segment"; return ET_E_INVALID_ARG; }
    // everything that does not depend on the segment: filter results, order, positions, factors (taken over instead of
    // copied when no other share will be cut from this common part)
    const double t0 = now_s();
    if (last_share) g = std::move(const_cast<PlanCommon&>(pc).g);
    else g = pc.g;
    g.geo.set_share(seg_index);
    const unsigned T = host_threads();
    AttesterWalk src(pc, g, T);
    if (getenv("ET_BUILD_STATS")) fprintf(stderr, "[et build] copy of the common part and visit order %.3f s\n", now_s() - t0);
    if (g.geo.vec_len >= COL_LOCAL_MASK) { err = "more than 2^29 entries in a share's gather vector"; return ET_E_UNSUPPORTED; }
    BuildOpts o = opt;
    if (getenv("ET_CLUSTER_SHARES")) o.cluster_cited = true;     // measurement knob: such a share steps but cannot pack
    return lay_out_segment(g, o, src, T, err);
}

struct SortedRecords {
    RecordWalk walk;
    DealtSegment dealt;
    SortedRecords(HostGraph& g, unsigned T, const std::vector<uint32_t>& s, const std::vector<uint32_t>& r, const std::vector<uint32_t>& v)
        : walk(g, T, s, r, v) {}
};
void free_sorted_records(SortedRecords* r) { delete r; }

int records_to_needs(HostGraph& g, const BuildOpts& opt, uint32_t seg_index, const std::vector<uint32_t>& src_code, const std::vector<uint32_t>& row,
                     const std::vector<uint32_t>& val, SortedRecords** out, std::string& err) {
    if (seg_index >= g.geo.n_seg) { err = "bad segment"; return ET_E_INVALID_ARG; }
    if (!g.values_are_u32) { err = "edge records carry integer opinions only"; return ET_E_UNSUPPORTED; }
    if (src_code.size() != row.size() || row.size() != val.size()) { err = "edge record arrays differ in length"; return ET_E_INVALID_ARG; }
    g.geo.set_share(seg_index);
    for (size_t k = 0; k < row.size(); ++k) {
        const uint32_t s = src_code[k] >> COL_SEG_SHIFT, l = src_code[k] & COL_LOCAL_MASK;
        if (row[k] >= g.geo.rows_cap || s >= g.geo.n_seg || l >= g.geo.rows_cap) { err = "edge record out of range"; return ET_E_OUT_OF_RANGE; }
    }
    *out = new SortedRecords(g, host_threads(), src_code, row, val);
    if (g.geo.vec_len >= COL_LOCAL_MASK) { delete *out; *out = nullptr; err = "more than 2^29 entries in a share's gather vector"; return ET_E_UNSUPPORTED; }
    // the row groups dealt to the workgroups, and with them the order this share wants the other shares' entries in
    return deal_segment(g, opt, (*out)->walk, (*out)->dealt);
}

int plan_segment_from_sorted(HostGraph& g, const BuildOpts& opt, SortedRecords* rec, std::string& err) {
    if (!rec) { err = "no records"; return ET_E_INVALID_ARG; }
    if (g.geo.compact && !g.send_known) { err = "the rows the other shares cite are not known yet"; return ET_E_INVALID_ARG; }
    for (uint32_t c = 0; c < 8; ++c)
        for (uint32_t l : g.send[c]) if (l >= g.geo.rows_cap) { err = "cited row out of range"; return ET_E_OUT_OF_RANGE; }
    return fill_segment(g, opt, rec->walk, rec->walk.T, rec->dealt, err);
}

int plan_segment_from_records(HostGraph& g, const BuildOpts& opt, uint32_t seg_index, const std::vector<uint32_t>& src_code,
                              const std::vector<uint32_t>& row, const std::vector<uint32_t>& val, std::string& err) {
    if (g.geo.compact) { err = "a share with a gather vector is laid out in two steps (records_to_needs, plan_segment_from_sorted)"; return ET_E_INVALID_ARG; }
    SortedRecords* rec = nullptr;
    int rc = records_to_needs(g, opt, seg_index, src_code, row, val, &rec, err);
    if (rc) return rc;
    rc = plan_segment_from_sorted(g, opt, rec, err);
    delete rec;
    return rc;
}

int build_graph(const BuildInput& in, const BuildOpts& opt, HostGraph& g, std::string& err) {
    PlanCommon pc;
    int rc = plan_common(in, opt, pc, err);
    if (rc) return rc;
    return plan_segment(pc, opt, opt.seg, g, err, true);
}

}  // namespace et
