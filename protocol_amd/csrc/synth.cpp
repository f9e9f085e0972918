// Deterministic synthetic attestation lists.  The reference never runs above a dozen peers
// (its sizes are compile-time constants, circuits/mod.rs:39-43), so the large benchmark inputs are
// generated here; the same lists feed the GPU path and the CPU checker.
//
// Every peer draws from its own xoshiro256** stream seeded by splitmix64(seed, peer), so the list
// does not depend on thread count or generation order.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/eigentrust_hip.h"

namespace {

inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Xoshiro {
    uint64_t s[4];
    Xoshiro(uint64_t seed, uint64_t stream) {
        uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL);
        for (int i = 0; i < 4; ++i) s[i] = splitmix64(x);
    }
    static inline uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    inline uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    inline double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    inline uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};

template <typename F>
void parallel_for(uint32_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(16u, hw ? hw : 1u));
    if (n < 4096) nt = 1;
    std::atomic<uint32_t> next(0);
    const uint32_t chunk = 1024;
    auto work = [&]() {
        for (;;) {
            uint32_t b = next.fetch_add(chunk);
            if (b >= n) break;
            uint32_t e = std::min(n, b + chunk);
            for (uint32_t i = b; i < e; ++i) f(i);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

// Walker alias table over weights w[i]
struct Alias {
    std::vector<double> prob;
    std::vector<uint32_t> alias;
    void build(const std::vector<double>& w) {
        size_t n = w.size();
        prob.assign(n, 0.0);
        alias.assign(n, 0);
        double sum = 0;
        for (double x : w) sum += x;
        std::vector<double> p(n);
        std::vector<uint32_t> small, large;
        for (size_t i = 0; i < n; ++i) { p[i] = w[i] * n / sum; (p[i] < 1.0 ? small : large).push_back((uint32_t)i); }
        while (!small.empty() && !large.empty()) {
            uint32_t s = small.back(), l = large.back();
            small.pop_back();
            prob[s] = p[s]; alias[s] = l;
            p[l] = (p[l] + p[s]) - 1.0;
            if (p[l] < 1.0) { large.pop_back(); small.push_back(l); }
        }
        for (uint32_t i : large) { prob[i] = 1.0; alias[i] = i; }
        for (uint32_t i : small) { prob[i] = 1.0; alias[i] = i; }
    }
    inline uint32_t sample(Xoshiro& r) const {
        uint32_t i = (uint32_t)r.below(prob.size());
        return r.unit() < prob[i] ? i : alias[i];
    }
};

}  // namespace

extern "C" int et_synth_graph_range(uint32_t kind, uint32_t n, uint64_t target_edges, uint64_t seed, uint32_t value_max,
                                    uint32_t attester_begin, uint32_t attester_end, uint64_t* n_edges, uint32_t* from, uint32_t* to,
                                    uint32_t* value) {
    if (!n_edges || n < 2 || value_max == 0 || (kind != 2 && kind != 3) || attester_begin > attester_end || attester_end > n) return ET_E_INVALID_ARG;
    std::vector<uint32_t> deg(n, 0);
    std::vector<double> popularity;
    if (kind == 2) {
        uint32_t d = (uint32_t)std::min<uint64_t>(n - 1, (target_edges + n - 1) / n);
        std::fill(deg.begin(), deg.end(), d);
    } else {
        // discrete power law, gamma = 2.2, d_min = 1, d_max = min(65536, n-1); 1 % of the peers
        // hold no opinion at all; degrees are rescaled so that they sum to target_edges
        const double gamma = 2.2;
        const uint32_t dmax = std::min<uint32_t>(65536, n - 1);
        std::vector<double> raw(n, 0.0);
        parallel_for(n, [&](uint32_t i) {
            Xoshiro r(seed, (uint64_t)i * 2 + 1);
            if (r.below(100) == 0) return;
            double u = 1.0 - r.unit();  // (0,1]
            raw[i] = std::min<double>(dmax, std::floor(std::pow(u, -1.0 / (gamma - 1.0))));
        });
        double sum = 0;
        for (double x : raw) sum += x;
        double scale = sum > 0 ? (double)target_edges / sum : 0.0;
        uint64_t total = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (raw[i] == 0.0) continue;
            deg[i] = (uint32_t)std::min<double>(dmax, std::max(1.0, std::floor(raw[i] * scale + 0.5)));
            total += deg[i];
        }
        // spread the rounding remainder deterministically
        for (uint32_t i = 0; total != target_edges && i < 64u * n; ++i) {
            uint32_t k = (uint32_t)(((uint64_t)i * 2654435761u) % n);
            if (deg[k] == 0) continue;
            if (total < target_edges && deg[k] < dmax) { ++deg[k]; ++total; }
            else if (total > target_edges && deg[k] > 1) { --deg[k]; --total; }
        }
        // a peer is rated with probability proportional to the square root of its own activity
        popularity.resize(n);
        for (uint32_t i = 0; i < n; ++i) popularity[i] = std::sqrt((double)deg[i] + 1.0);
    }
    // the degrees and the target distribution depend on all peers; only the attesters asked for emit their lists
    std::vector<uint64_t> off((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + ((i >= attester_begin && i < attester_end) ? deg[i] : 0u);
    *n_edges = off[n];
    if (!from || !to || !value) return ET_OK;
    Alias alias;
    if (kind == 3) alias.build(popularity);
    parallel_for(n, [&](uint32_t i) {
        uint32_t d = deg[i];
        if (d == 0 || i < attester_begin || i >= attester_end) return;
        Xoshiro r(seed, (uint64_t)i * 2 + 2);
        std::vector<uint32_t> t;
        t.reserve((size_t)d * 2);
        // draw until d distinct targets other than i are collected
        size_t have = 0;
        for (int round = 0; have < d && round < 64; ++round) {
            size_t want = (d - have) + (d - have) / 8 + 4;
            for (size_t k = 0; k < want; ++k) {
                uint32_t x = kind == 3 ? alias.sample(r) : (uint32_t)r.below(n);
                if (x != i) t.push_back(x);
            }
            std::sort(t.begin(), t.end());
            t.erase(std::unique(t.begin(), t.end()), t.end());
            have = t.size();
        }
        if (have > d) {
            // drop the surplus at evenly spread ranks so no part of the id range is favoured
            std::vector<uint32_t> keep;
            keep.reserve(d);
            for (uint32_t k = 0; k < d; ++k) keep.push_back(t[(size_t)k * have / d]);
            t.swap(keep);
        }
        // (have < d only if the distribution cannot supply d distinct targets; fill linearly)
        for (uint32_t x = 0; t.size() < d && x < n; ++x)
            if (x != i && !std::binary_search(t.begin(), t.end(), x)) { t.insert(std::upper_bound(t.begin(), t.end(), x), x); }
        uint64_t base = off[i];
        for (uint32_t k = 0; k < d; ++k) {
            from[base + k] = i;
            to[base + k] = t[k];
            value[base + k] = 1 + (uint32_t)r.below(value_max);
        }
    });
    return ET_OK;
}

extern "C" int et_synth_graph(uint32_t kind, uint32_t n, uint64_t target_edges, uint64_t seed, uint32_t value_max,
                              uint64_t* n_edges, uint32_t* from, uint32_t* to, uint32_t* value) {
    return et_synth_graph_range(kind, n, target_edges, seed, value_max, 0, n, n_edges, from, to, value);
}
