// Host side of the set API: the state of the reference's EigenTrustSet
// (dynamic_sets/native.rs:109-131) with run-time sizes, and converge_rational.
// converge / converge_f64 hand the stored opinions to the graph builder and the GPU engine.
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "bigrat.hpp"
#include "builder.hpp"
#include "engine.hpp"
#include "fr_host.hpp"

using namespace et;

struct et_set {
    uint32_t n = 0;
    uint32_t num_iter = 0;
    uint64_t init_lo = 0, init_hi = 0;
    FrRaw domain{};
    std::vector<FrRaw> addr;    // native.rs:127 `set`, first tuple field; zero = empty slot
    std::vector<FrRaw> score;   // second tuple field (INITIAL_SCORE for members, native.rs:184-185)
    // native.rs:128 `ops`: keyed by the sender's address there; a sender is always a member when
    // its row is stored (opinion/native.rs:66-67) and the row is dropped with the member
    // (native.rs:197), so keying by slot is equivalent.  Only non-zero entries are kept.
    std::vector<uint8_t> has_row;
    std::vector<std::vector<std::pair<uint32_t, FrRaw>>> rows;
    bool all_u32 = true;        // every stored opinion fits 32 bits
    // The reference's converge takes &self and is called again and again between membership changes
    // (native.rs:873-878): the device-resident form of the set is kept until the set changes, so that a repeated
    // converge is a launch and a copy, with no allocation and no upload.
    uint64_t version = 1;       // bumped by every call that changes members or opinions
    et_batch* dev_batch = nullptr;
    DeviceGraph* dev_graph[2] = {nullptr, nullptr};   // exact variant, f64 variant
    uint64_t batch_version = 0, graph_version[2] = {0, 0};
    ~et_set() {
        if (dev_batch) et_batch_free(dev_batch);
        for (DeviceGraph* g : dev_graph) delete g;
    }

    int find(const FrRaw& a) const {
        for (uint32_t i = 0; i < n; ++i) if (addr[i] == a) return (int)i;
        return -1;
    }
};

static bool read_fr(const et_fr* p, FrRaw& out) {
    std::memcpy(out.l, p->b, 32);
    return fr_raw_is_canonical(out.l);
}

extern "C" {

int et_set_new(uint32_t num_neighbours, uint32_t num_iterations, uint64_t initial_score_lo,
               uint64_t initial_score_hi, const et_fr* domain, et_set** out) {
    if (!out || num_neighbours == 0) return ET_E_INVALID_ARG;
    std::unique_ptr<et_set> s(new et_set());
    s->n = num_neighbours;
    s->num_iter = num_iterations;
    s->init_lo = initial_score_lo;
    s->init_hi = initial_score_hi;
    if (domain && !read_fr(domain, s->domain)) return ET_E_NONCANONICAL_FR;
    s->addr.assign(num_neighbours, FrRaw{{0, 0, 0, 0}});   // native.rs:167
    s->score.assign(num_neighbours, FrRaw{{0, 0, 0, 0}});
    s->has_row.assign(num_neighbours, 0);
    s->rows.resize(num_neighbours);
    *out = s.release();
    return ET_OK;
}

void et_set_free(et_set* set) { delete set; }

int et_set_add_member(et_set* s, const et_fr* addr) {
    if (!s || !addr) return ET_E_INVALID_ARG;
    FrRaw a;
    if (!read_fr(addr, a)) return ET_E_NONCANONICAL_FR;
    if (a.is_zero()) { set_error("the zero address marks an empty slot"); return ET_E_INVALID_ARG; }
    if (s->find(a) >= 0) return ET_E_DUP_MEMBER;                       // native.rs:176-178
    int idx = s->find(FrRaw{{0, 0, 0, 0}});                             // native.rs:180
    if (idx < 0) return ET_E_SET_FULL;                                  // native.rs:181
    s->addr[idx] = a;
    s->score[idx] = Fr::from_u128(s->init_lo, s->init_hi).to_raw();    // native.rs:184-185
    s->has_row[idx] = 0;
    s->rows[idx].clear();
    ++s->version;
    return ET_OK;
}

int et_set_remove_member(et_set* s, const et_fr* addr) {
    if (!s || !addr) return ET_E_INVALID_ARG;
    FrRaw a;
    if (!read_fr(addr, a)) return ET_E_NONCANONICAL_FR;
    if (a.is_zero()) return ET_E_NOT_MEMBER;
    int idx = s->find(a);
    if (idx < 0) return ET_E_NOT_MEMBER;                                // native.rs:192
    s->addr[idx] = FrRaw{{0, 0, 0, 0}};                                 // native.rs:195
    s->score[idx] = FrRaw{{0, 0, 0, 0}};
    s->has_row[idx] = 0;                                                // native.rs:197
    s->rows[idx].clear();
    ++s->version;
    return ET_OK;
}

static int store_row(et_set* s, int from_slot, uint32_t count, const uint32_t* about, const et_fr* values_fr,
                     const uint32_t* values_u32) {
    std::map<uint32_t, FrRaw> row;  // a later entry for the same slot replaces an earlier one
    for (uint32_t k = 0; k < count; ++k) {
        uint32_t j = about ? about[k] : k;
        if (j >= s->n) return ET_E_OUT_OF_RANGE;
        FrRaw v{{0, 0, 0, 0}};
        if (values_fr) { if (!read_fr(values_fr + k, v)) return ET_E_NONCANONICAL_FR; }
        else v.l[0] = values_u32[k];
        if (s->addr[j].is_zero()) v = FrRaw{{0, 0, 0, 0}};              // opinion/native.rs:92-98
        row[j] = v;
    }
    auto& dst = s->rows[from_slot];
    dst.clear();
    for (auto& kv : row) {
        if (kv.second.is_zero()) continue;
        if (kv.second.l[1] | kv.second.l[2] | kv.second.l[3] | (kv.second.l[0] >> 32)) s->all_u32 = false;
        dst.push_back(kv);
    }
    s->has_row[from_slot] = 1;                                          // native.rs:228
    ++s->version;
    return ET_OK;
}

int et_set_update_op(et_set* s, const et_fr* from, const et_fr* scores) {
    if (!s || !from || !scores) return ET_E_INVALID_ARG;
    FrRaw a;
    if (!read_fr(from, a)) return ET_E_NONCANONICAL_FR;
    int idx = a.is_zero() ? -1 : s->find(a);
    if (idx < 0) return ET_E_NOT_MEMBER;                                // opinion/native.rs:66-67
    return store_row(s, idx, s->n, nullptr, scores, nullptr);
}

int et_set_update_op_sparse(et_set* s, const et_fr* from, uint32_t count, const uint32_t* about_slot,
                            const et_fr* values) {
    if (!s || !from || (count && (!about_slot || !values))) return ET_E_INVALID_ARG;
    FrRaw a;
    if (!read_fr(from, a)) return ET_E_NONCANONICAL_FR;
    int idx = a.is_zero() ? -1 : s->find(a);
    if (idx < 0) return ET_E_NOT_MEMBER;
    return store_row(s, idx, count, about_slot, values, nullptr);
}

int et_set_update_op_u32(et_set* s, uint32_t from_slot, uint32_t count, const uint32_t* about_slot,
                         const uint32_t* values) {
    if (!s || (count && (!about_slot || !values))) return ET_E_INVALID_ARG;
    if (from_slot >= s->n) return ET_E_OUT_OF_RANGE;
    if (s->addr[from_slot].is_zero()) return ET_E_NOT_MEMBER;
    return store_row(s, (int)from_slot, count, about_slot, nullptr, values);
}

int et_set_members(const et_set* s, et_fr* addrs, et_fr* scores) {
    if (!s) return ET_E_INVALID_ARG;
    for (uint32_t i = 0; i < s->n; ++i) {
        if (addrs) std::memcpy(addrs[i].b, s->addr[i].l, 32);
        if (scores) std::memcpy(scores[i].b, s->score[i].l, 32);
    }
    return ET_OK;
}

// Flattens the stored rows into an attestation list for the builder.
static void flatten(const et_set* s, std::vector<uint8_t>& member, std::vector<uint32_t>& from,
                    std::vector<uint32_t>& to, std::vector<uint32_t>& v32, std::vector<FrRaw>& vfr, bool as_u32) {
    member.assign(s->n, 0);
    for (uint32_t i = 0; i < s->n; ++i) member[i] = !s->addr[i].is_zero();
    for (uint32_t i = 0; i < s->n; ++i) {
        if (!member[i] || !s->has_row[i]) continue;
        for (auto& kv : s->rows[i]) {
            from.push_back(i);
            to.push_back(kv.first);
            if (as_u32) v32.push_back((uint32_t)kv.second.l[0]);
            else vfr.push_back(kv.second);
        }
    }
}

// The device-resident graph of the set for one variant (0 = exact, 1 = f64), rebuilt only after the set changed.
static int build_for_set(const et_set* s, int variants, DeviceGraph** out);
static int graph_for_set(et_set* s, int which, DeviceGraph** out) {
    if (!s->dev_graph[which] || s->graph_version[which] != s->version) {
        delete s->dev_graph[which];
        s->dev_graph[which] = nullptr;
        int rc = build_for_set(s, which == 0 ? ET_VARIANT_FR : ET_VARIANT_F64, &s->dev_graph[which]);
        if (rc) return rc;
        s->graph_version[which] = s->version;
    }
    *out = s->dev_graph[which];
    return ET_OK;
}

static int build_for_set(const et_set* s, int variants, DeviceGraph** out) {
    std::vector<uint8_t> member;
    std::vector<uint32_t> from, to, v32;
    std::vector<FrRaw> vfr;
    bool as_u32 = s->all_u32;
    flatten(s, member, from, to, v32, vfr, as_u32);
    BuildInput in;
    in.n = s->n;
    in.member = member.data();
    in.n_edges = from.size();
    in.from = from.data();
    in.to = to.data();
    static const uint32_t dummy32 = 0;
    if (as_u32) in.val_u32 = v32.empty() ? &dummy32 : v32.data();
    else in.val_fr = vfr.data();
    BuildOpts opt;
    HostGraph h;
    std::string err;
    int rc = build_graph(in, opt, h, err);
    if (rc) { set_error(err); return rc; }
    return DeviceGraph::create(h, variants, -1, out);
}

int et_set_converge(et_set* s, et_fr* out) {
    if (!s || !out) return ET_E_INVALID_ARG;
    {
        uint32_t valid = 0;                                             // native.rs:288-289
        for (uint32_t i = 0; i < s->n; ++i) valid += !s->addr[i].is_zero();
        if (valid < 2) { set_error("Insufficient peers for calculation!"); return ET_E_INSUFFICIENT_PEERS; }
    }
    if (s->n <= 1024 && s->all_u32) {
        // small set: one workgroup runs the whole converge in a single launch (batch of one)
        if (!s->dev_batch || s->batch_version != s->version) {
            if (s->dev_batch) et_batch_free(s->dev_batch);
            s->dev_batch = nullptr;
            std::vector<uint8_t> member(s->n);
            std::vector<uint32_t> dense((size_t)s->n * s->n, 0);
            for (uint32_t i = 0; i < s->n; ++i) {
                member[i] = !s->addr[i].is_zero();
                if (member[i] && s->has_row[i])
                    for (auto& kv : s->rows[i]) dense[(size_t)i * s->n + kv.first] = (uint32_t)kv.second.l[0];
            }
            int rc = et_batch_create(1, s->n, member.data(), dense.data(), -1, &s->dev_batch);
            if (rc) return rc;
            s->batch_version = s->version;
        }
        int32_t status = ET_OK;
        int rc = et_batch_run_fr_async(s->dev_batch, s->num_iter, s->init_lo, s->init_hi, nullptr);
        if (!rc) rc = et_batch_result_fr(s->dev_batch, out, &status, nullptr);
        if (rc) return rc;
        if (status == ET_E_INSUFFICIENT_PEERS) set_error("Insufficient peers for calculation!");
        if (status == ET_E_SUM_MISMATCH) set_error("reputation leak: field sum of scores changed");
        return status;
    }
    DeviceGraph* g = nullptr;
    int rc = graph_for_set(s, 0, &g);
    if (rc) return rc;
    return et_graph_converge_fr(reinterpret_cast<et_graph*>(g), s->num_iter, s->init_lo, s->init_hi, out);
}

int et_set_converge_f64(et_set* s, double alpha, const double* pretrust, double* out, double* residual_l1) {
    if (!s || !out) return ET_E_INVALID_ARG;
    DeviceGraph* g = nullptr;
    int rc = graph_for_set(s, 1, &g);
    if (rc) return rc;
    double initial = (double)s->init_lo + 18446744073709551616.0 * (double)s->init_hi;
    return et_graph_converge_f64(reinterpret_cast<et_graph*>(g), s->num_iter, initial, alpha, pretrust, out, residual_l1);
}

// converge_rational, native.rs:340-392
int et_set_converge_rational(const et_set* s, uint8_t* num_le, uint8_t* den_le, size_t stride, size_t* needed) {
    if (!s) return ET_E_INVALID_ARG;
    const uint32_t n = s->n;
    std::vector<uint8_t> member(n);
    uint32_t n_valid = 0;
    for (uint32_t i = 0; i < n; ++i) { member[i] = !s->addr[i].is_zero(); n_valid += member[i]; }
    // filter_peers_ops (native.rs:234-283) in sparse form
    struct Row { std::vector<std::pair<uint32_t, BigUint>> e; bool ones = false; BigUint sum; };
    std::vector<Row> rows(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!member[i]) continue;                       // empty slot: zero row (:346-347)
        Fr fsum = Fr::zero();
        if (s->has_row[i]) {
            for (auto& kv : s->rows[i]) {
                uint32_t j = kv.first;
                if (j == i || !member[j]) continue;     // :254-259
                fsum += Fr::from_raw(kv.second);
                rows[i].e.push_back({j, BigUint::from_le_u64(kv.second.l, 4)});
            }
        }
        if (fsum.is_zero()) {                           // :263-278
            rows[i].e.clear();
            rows[i].ones = true;
            rows[i].sum = BigUint(n_valid > 0 ? n_valid - 1 : 0);
        } else {
            for (auto& kv : rows[i].e) rows[i].sum = BigUint::add(rows[i].sum, kv.second);
        }
        if (rows[i].sum.is_zero()) rows[i].sum = BigUint(1);  // :368-371
    }
    BigUint init = BigUint::from_le_u64(&s->init_lo, 1);
    if (s->init_hi) { uint64_t l[2] = {s->init_lo, s->init_hi}; init = BigUint::from_le_u64(l, 2); }
    std::vector<BigFrac> sc(n, BigFrac(init, BigUint(1)));    // :362-364 (every slot)
    for (uint32_t it = 0; it < s->num_iter; ++it) {           // :380-390
        std::vector<BigFrac> ns(n);
        BigFrac ones_mass;                                    // sum over rows of ones of s[j]/sum_j
        std::vector<BigFrac> share(n);
        for (uint32_t j = 0; j < n; ++j) {
            if (!member[j] || sc[j].num.is_zero()) continue;
            share[j] = BigFrac(sc[j].num, BigUint::mul(sc[j].den, rows[j].sum));
            if (rows[j].ones) ones_mass = BigFrac::add(ones_mass, share[j]);
            else for (auto& kv : rows[j].e) ns[kv.first] = BigFrac::add(ns[kv.first], BigFrac(BigUint::mul(kv.second, share[j].num), share[j].den));
        }
        for (uint32_t i = 0; i < n; ++i) {
            if (!member[i]) continue;
            BigFrac m = ones_mass;
            if (rows[i].ones && !share[i].num.is_zero()) {
                // the row of ones skips its own column: subtract this peer's share
                BigUint a = BigUint::mul(m.num, share[i].den), b = BigUint::mul(share[i].num, m.den);
                m = BigFrac(BigUint::sub(a, b), BigUint::mul(m.den, share[i].den));
            }
            ns[i] = BigFrac::add(ns[i], m);
        }
        sc.swap(ns);
    }
    size_t need = 1;
    for (uint32_t i = 0; i < n; ++i) need = std::max(need, std::max(sc[i].num.byte_len(), sc[i].den.byte_len()));
    if (needed) *needed = need;
    if (!num_le || !den_le || stride < need) return ET_E_BUFFER_TOO_SMALL;
    for (uint32_t i = 0; i < n; ++i) {
        sc[i].num.to_le_bytes(num_le + i * stride, stride);
        sc[i].den.to_le_bytes(den_le + i * stride, stride);
    }
    return ET_OK;
}

}  // extern "C"
