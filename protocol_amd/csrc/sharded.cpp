// Row-sharded converge over several GPUs of one node, driven from ONE host thread (et_sharded_* in
// include/eigentrust_hip.h).
//
// The reference iterates one dense matrix on one core (dynamic_sets/native.rs:319-329).  Here the rows
// (destination peers) of the filtered graph are dealt to the GPUs; every GPU keeps a full-length trust
// vector and, per iteration, runs the step kernel on its rows and then receives the other GPUs' segments:
// one all-gather per iteration, the only exchange on the path.  A segment carries its rows' entries and,
// in its tail, the GPU's partial sums of the dangling peers' score mass, so no second collective is needed.
//
// Integer opinions (the production case): a GPU does not keep the other GPUs' whole segments but a gather vector --
// its own block, then only the entries of the other GPUs' rows that its own rows cite (layout.hpp).  After a step
// every GPU packs what each other GPU cites into a send buffer (DeviceGraph::pack) and the exchange is an
// all-to-all of those pieces: grouped ncclSend / ncclRecv, or pulls from the producers' send buffers.  The result is
// assembled once, at the end, from the GPUs' own blocks.
//
// Exchange.  RCCL's all-gather (in place, one group call over the GPUs' communicators) when the devices are
// distinct; librccl is loaded at run time, so the library itself has no link-time dependency on it.  When
// the same device is listed more than once (a rehearsal of the N-share layout on fewer GPUs: RCCL refuses two
// ranks of one communicator on one device) or on request, every share PULLS the other shares' segments with
// device-to-device copies ordered by events; on a fully connected xGMI node that puts one segment on each
// link, which is also what the best all-gather does there.
//
// Chunked exchange (n_chunks = K > 1).  The exchange moves more bytes per iteration than the kernels take time for,
// so a converge that runs kernel, exchange, kernel, ... leaves the GPUs idle most of the time.  With K pieces per
// segment an iteration runs as K launches, launch q over the edges whose sources lie in piece q of any segment, and
// the pieces travel on a second stream per share: launch q waits for piece q only, and piece q + 1 arrives while it
// runs.  Pieces are pulled piece by piece (all segments' piece 0 first), over RCCL as an all-gather into a staging
// buffer followed by a strided copy into place.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "builder.hpp"
#include "engine.hpp"
#include "fr_host.hpp"

using namespace et;

#define HIP_TRY_S(expr)                                                              \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));            \
            return _e == hipErrorNoDevice ? ET_E_NO_DEVICE : ET_E_HIP;               \
        }                                                                            \
    } while (0)

namespace {

// The handful of RCCL entry points the exchange needs, resolved from librccl at first use.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) { r.why = std::string("librccl not loadable: ") + dlerror(); return r; }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.Send = (decltype(r.Send))dlsym(r.handle, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.handle, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.GetErrorString;
    if (!r.ok) r.why = "librccl lacks an entry point";
    return r;
}

}  // namespace

#define RCCL_TRY(expr)                                                               \
    do {                                                                             \
        ncclResult_t _r = (expr);                                                    \
        if (_r != ncclSuccess) {                                                     \
            set_error(std::string(#expr) + ": " + rccl().GetErrorString(_r));        \
            return ET_E_RCCL;                                                        \
        }                                                                            \
    } while (0)

struct et_sharded {
    uint32_t n = 0, n_valid = 0, world = 0;
    int variants = 0;
    uint32_t exchange = ET_EXCHANGE_COPY;
    std::vector<int> device;
    std::vector<DeviceGraph*> g;
    std::vector<hipStream_t> stream;
    std::vector<hipStream_t> cstream;         // chunked exchange: the stream the pieces travel on
    std::vector<hipEvent_t> ready;            // chunked exchange: [share * n_chunks + q] piece q of every segment is here
    std::vector<void*> staging;               // chunked exchange over RCCL: world x (longest piece) per share
    uint32_t n_chunks = 1;
    std::vector<hipEvent_t> done;             // share r's segment of the vector just written is complete
    std::vector<void*> vec[2];                // per share: two vectors (g[r]->vec_len entries), sized for the widest variant built
    bool gather = false;                      // the shares' vectors are gather vectors
    std::vector<void*> sendbuf[2];            // gather vectors: what share r packed for the others after writing vec[b][r]
    void* full = nullptr;                     // gather vectors: the result in position order, on the first share's device
    std::vector<double*> pretrust;            // per share, f64 with alpha != 0 only
    std::vector<ncclComm_t> comm;
    size_t vec_len = 0, seg_stride = 0;
    void* out_dev = nullptr;                  // n entries in slot order, on the first share's device

    ~et_sharded() {
        for (uint32_t r = 0; r < world; ++r) {
            if (r < device.size()) (void)hipSetDevice(device[r]);
            if (r < stream.size() && stream[r]) (void)hipStreamSynchronize(stream[r]);
        }
        if (!comm.empty() && rccl().ok)
            for (ncclComm_t c : comm) if (c) (void)rccl().CommDestroy(c);
        for (uint32_t r = 0; r < world; ++r) {
            if (r < device.size()) (void)hipSetDevice(device[r]);
            for (int b = 0; b < 2; ++b) if (r < vec[b].size() && vec[b][r]) (void)hipFree(vec[b][r]);
            for (int b = 0; b < 2; ++b) if (r < sendbuf[b].size() && sendbuf[b][r]) (void)hipFree(sendbuf[b][r]);
            if (r == 0 && full) (void)hipFree(full);
            if (r < pretrust.size() && pretrust[r]) (void)hipFree(pretrust[r]);
            if (r < done.size() && done[r]) (void)hipEventDestroy(done[r]);
            for (uint32_t q = 0; q < n_chunks; ++q)
                if ((size_t)r * n_chunks + q < ready.size() && ready[(size_t)r * n_chunks + q]) (void)hipEventDestroy(ready[(size_t)r * n_chunks + q]);
            if (r < staging.size() && staging[r]) (void)hipFree(staging[r]);
            if (r < cstream.size() && cstream[r]) { (void)hipStreamSynchronize(cstream[r]); (void)hipStreamDestroy(cstream[r]); }
            if (r < stream.size() && stream[r]) (void)hipStreamDestroy(stream[r]);
            if (r < g.size()) delete g[r];
            if (r == 0 && out_dev) (void)hipFree(out_dev);
        }
    }

    // Every share receives the other shares' segments of buffer `b` (elements of `elem` bytes).
    int exchange_segments(int b, size_t elem) {
        if (world == 1 && exchange != ET_EXCHANGE_RCCL) return ET_OK;   // (one rank over RCCL: the call is made all the same)
        if (gather) return exchange_cited(b, elem);
        const size_t seg_bytes = seg_stride * elem;
        if (exchange == ET_EXCHANGE_RCCL) {
            RCCL_TRY(rccl().GroupStart());
            for (uint32_t r = 0; r < world; ++r) {
                char* base = static_cast<char*>(vec[b][r]);
                ncclResult_t rc = rccl().AllGather(base + r * seg_bytes, base, seg_bytes, ncclChar, comm[r], stream[r]);
                if (rc != ncclSuccess) {
                    (void)rccl().GroupEnd();
                    set_error(std::string("ncclAllGather: ") + rccl().GetErrorString(rc));
                    return ET_E_RCCL;
                }
            }
            RCCL_TRY(rccl().GroupEnd());
            return ET_OK;
        }
        for (uint32_t r = 0; r < world; ++r) {
            HIP_TRY_S(hipSetDevice(device[r]));
            HIP_TRY_S(hipEventRecord(done[r], stream[r]));
        }
        for (uint32_t q = 0; q < world; ++q) {
            HIP_TRY_S(hipSetDevice(device[q]));
            for (uint32_t k = 1; k < world; ++k) {
                const uint32_t r = (q + k) % world;          // every share starts with another source: one segment per link
                HIP_TRY_S(hipStreamWaitEvent(stream[q], done[r], 0));
                HIP_TRY_S(hipMemcpyAsync(static_cast<char*>(vec[b][q]) + r * seg_bytes, static_cast<char*>(vec[b][r]) + r * seg_bytes,
                                         seg_bytes, hipMemcpyDeviceToDevice, stream[q]));
            }
        }
        // a share may overwrite its segment of the OTHER buffer in its next step only after every share has pulled
        // that buffer's previous contents; the pulls above are ordered before each puller's own next step, and a share's
        // step after next needs the pullers' next segments first, so the chain of events already enforces it
        return ET_OK;
    }

    // Gather vectors: every share packs what the others cite of the block it has just written into vec[b], and every share
    // receives its pieces at the blocks of its own vec[b].
    int exchange_cited(int b, size_t elem) {
        int rc;
        for (uint32_t r = 0; r < world; ++r) {
            HIP_TRY_S(hipSetDevice(device[r]));
            if ((rc = g[r]->pack(vec[b][r], sendbuf[b][r], elem, stream[r]))) return rc;
        }
        if (exchange == ET_EXCHANGE_RCCL) {
            RCCL_TRY(rccl().GroupStart());
            for (uint32_t r = 0; r < world; ++r)
                for (uint32_t c = 0; c < world; ++c) {
                    if (c == r) continue;
                    ncclResult_t e = rccl().Send(static_cast<char*>(sendbuf[b][r]) + (size_t)g[r]->send_off[c] * elem, (size_t)g[r]->send_cnt[c] * elem,
                                                 ncclChar, (int)c, comm[r], stream[r]);
                    if (e == ncclSuccess)
                        e = rccl().Recv(static_cast<char*>(vec[b][r]) + (size_t)g[r]->recv_off[c] * elem, (size_t)g[r]->recv_cnt[c] * elem,
                                        ncclChar, (int)c, comm[r], stream[r]);
                    if (e != ncclSuccess) {
                        (void)rccl().GroupEnd();
                        set_error(std::string("ncclSend/ncclRecv: ") + rccl().GetErrorString(e));
                        return ET_E_RCCL;
                    }
                }
            RCCL_TRY(rccl().GroupEnd());
            return ET_OK;
        }
        for (uint32_t r = 0; r < world; ++r) {
            HIP_TRY_S(hipSetDevice(device[r]));
            HIP_TRY_S(hipEventRecord(done[r], stream[r]));
        }
        for (uint32_t q = 0; q < world; ++q) {
            HIP_TRY_S(hipSetDevice(device[q]));
            for (uint32_t k = 1; k < world; ++k) {
                const uint32_t r = (q + k) % world;          // every share starts with another producer: one piece per link
                if (g[q]->recv_cnt[r] != g[r]->send_cnt[q]) { set_error("internal: shares disagree on the cited rows"); return ET_E_INVALID_ARG; }
                HIP_TRY_S(hipStreamWaitEvent(stream[q], done[r], 0));
                HIP_TRY_S(hipMemcpyAsync(static_cast<char*>(vec[b][q]) + (size_t)g[q]->recv_off[r] * elem,
                                         static_cast<char*>(sendbuf[b][r]) + (size_t)g[r]->send_off[q] * elem,
                                         (size_t)g[q]->recv_cnt[r] * elem, hipMemcpyDeviceToDevice, stream[q]));
            }
        }
        // a producer packs into sendbuf[b] again two steps later; by then every puller has run the step in between, which its
        // pulls above precede on its stream, and the producer's step after next waited for those pullers' next pieces
        return ET_OK;
    }

    // The vector in position order on the first share's device, from the own blocks of vec[b] (the last step's output):
    // returns the buffer et_graph_unpermute_* reads, ordered on stream[0].
    int collect(int b, size_t elem, void** out) {
        if (!gather) {
            int rc = exchange_segments(b, elem);
            *out = vec[b][0];
            return rc;
        }
        const size_t seg_bytes = seg_stride * elem;
        for (uint32_t r = 1; r < world; ++r) {
            HIP_TRY_S(hipSetDevice(device[r]));
            HIP_TRY_S(hipEventRecord(done[r], stream[r]));
        }
        HIP_TRY_S(hipSetDevice(device[0]));
        for (uint32_t r = 0; r < world; ++r) {
            if (r) HIP_TRY_S(hipStreamWaitEvent(stream[0], done[r], 0));
            HIP_TRY_S(hipMemcpyAsync(static_cast<char*>(full) + r * seg_bytes, vec[b][r], seg_bytes, hipMemcpyDeviceToDevice, stream[0]));
        }
        *out = full;
        return ET_OK;
    }
};

// Chunked exchange of buffer `b`, just written by every share (init, or the last launch of an iteration): the pieces
// travel on the shares' second streams in order 0 .. K-1, and ready[share][q] fires when piece q of every segment has
// reached `share`.  The next iteration's launch q waits for that event only.
static int start_chunked_exchange(et_sharded* s, int b, size_t elem) {
    const uint32_t W = s->world, K = s->n_chunks;
    const Geometry& geo = s->g[0]->geo;
    for (uint32_t r = 0; r < W; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        HIP_TRY_S(hipEventRecord(s->done[r], s->stream[r]));
    }
    for (uint32_t r = 0; r < W; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        for (uint32_t q = 0; q < W; ++q) HIP_TRY_S(hipStreamWaitEvent(s->cstream[r], s->done[q], 0));   // every producer, the share itself included
    }
    for (uint32_t q = 0; q < K; ++q) {
        const size_t off = (size_t)geo.chunk_begin(q) * elem, len = (size_t)(geo.chunk_begin(q + 1) - geo.chunk_begin(q)) * elem;
        const size_t seg_bytes = s->seg_stride * elem;
        if (s->exchange == ET_EXCHANGE_RCCL) {
            RCCL_TRY(rccl().GroupStart());
            for (uint32_t r = 0; r < W; ++r) {
                char* base = static_cast<char*>(s->vec[b][r]);
                ncclResult_t rc = rccl().AllGather(base + r * seg_bytes + off, s->staging[r], len, ncclChar, s->comm[r], s->cstream[r]);
                if (rc != ncclSuccess) {
                    (void)rccl().GroupEnd();
                    set_error(std::string("ncclAllGather: ") + rccl().GetErrorString(rc));
                    return ET_E_RCCL;
                }
            }
            RCCL_TRY(rccl().GroupEnd());
            for (uint32_t r = 0; r < W; ++r) {          // staging [segment][piece] -> piece q of every segment of the vector
                HIP_TRY_S(hipSetDevice(s->device[r]));
                HIP_TRY_S(hipMemcpy2DAsync(static_cast<char*>(s->vec[b][r]) + off, seg_bytes, s->staging[r], len, len, W,
                                           hipMemcpyDeviceToDevice, s->cstream[r]));
            }
        } else {
            for (uint32_t dst = 0; dst < W; ++dst) {
                HIP_TRY_S(hipSetDevice(s->device[dst]));
                for (uint32_t k = 1; k < W; ++k) {
                    const uint32_t src = (dst + k) % W;
                    HIP_TRY_S(hipMemcpyAsync(static_cast<char*>(s->vec[b][dst]) + src * seg_bytes + off,
                                             static_cast<char*>(s->vec[b][src]) + src * seg_bytes + off, len, hipMemcpyDeviceToDevice,
                                             s->cstream[dst]));
                }
            }
        }
        for (uint32_t r = 0; r < W; ++r) {
            HIP_TRY_S(hipSetDevice(s->device[r]));
            HIP_TRY_S(hipEventRecord(s->ready[(size_t)r * K + q], s->cstream[r]));
        }
    }
    return ET_OK;
}

extern "C" {

int et_sharded_create(uint32_t n, const uint8_t* member, uint64_t n_edges, const uint32_t* from, const uint32_t* to,
                      const uint32_t* value_u32, const et_fr* value_fr, uint32_t n_devices, const int32_t* devices,
                      uint32_t variants, uint32_t exchange, uint32_t n_chunks, et_sharded** out) {
    if (!out || n_devices == 0 || !devices) return ET_E_INVALID_ARG;
    if (n_devices > 8) { set_error("at most 8 shares"); return ET_E_UNSUPPORTED; }
    int ndev = 0;
    HIP_TRY_S(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return ET_E_NO_DEVICE; }
    bool distinct = true;
    for (uint32_t r = 0; r < n_devices; ++r) {
        if (devices[r] < 0 || devices[r] >= ndev) { set_error("device ordinal out of range"); return ET_E_OUT_OF_RANGE; }
        for (uint32_t q = 0; q < r; ++q) distinct = distinct && devices[q] != devices[r];
    }
    std::unique_ptr<et_sharded> s(new et_sharded());
    s->world = n_devices;
    s->variants = variants ? (int)variants : (ET_VARIANT_FR | ET_VARIANT_F64);
    s->device.assign(devices, devices + n_devices);
    if (exchange == ET_EXCHANGE_AUTO) exchange = (distinct && n_devices > 1 && rccl().ok) ? ET_EXCHANGE_RCCL : ET_EXCHANGE_COPY;
    if (exchange == ET_EXCHANGE_RCCL) {
        if (!rccl().ok) { set_error(rccl().why); return ET_E_RCCL; }
        if (!distinct) { set_error("RCCL needs pairwise distinct devices"); return ET_E_RCCL; }
    } else if (exchange != ET_EXCHANGE_COPY) {
        return ET_E_INVALID_ARG;
    }
    s->exchange = exchange;

    // plan once what every share has in common, then each share's own edge lists, uploaded and dropped in turn
    BuildInput in;
    in.n = n; in.member = member; in.n_edges = n_edges; in.from = from; in.to = to;
    in.val_u32 = value_u32; in.val_fr = reinterpret_cast<const FrRaw*>(value_fr);
    static const uint32_t dummy32 = 0;
    if (n_edges == 0 && !value_u32 && !value_fr) in.val_u32 = &dummy32;
    BuildOpts bo;
    bo.n_seg = n_devices;
    // (one rank over RCCL takes the chunked exchange too: the staged all-gather, the strided copy and the phased kernels
    // then run as several ranks would run them)
    if (n_chunks > 1 && (n_devices > 1 || exchange == ET_EXCHANGE_RCCL)) bo.n_chunks = n_chunks;
    PlanCommon pc;
    std::string err;
    int rc = plan_common(in, bo, pc, err);
    if (rc) { set_error(err); return rc; }
    s->n = n;
    s->n_valid = pc.g.n_valid;
    s->g.assign(n_devices, nullptr);
    // gather vectors: need[r][p] = the rows of share p that share r cites, in the order r asked for; share p packs them
    std::vector<std::vector<std::vector<uint32_t>>> need(n_devices);
    bo.cluster_cited = true;                             // every share is laid out here: each is told the others' orders below
    for (uint32_t r = 0; r < n_devices; ++r) {
        HostGraph h;
        BuildOpts bs = bo;
        xcd_cost_factors(devices[r], bs.xcd_cost);      // each share levelled for the XCDs of the card it runs on
        if ((rc = plan_segment(pc, bs, r, h, err))) { set_error(err); return rc; }
        h.send_known = false;                            // set below, from the other shares' plans
        if ((rc = DeviceGraph::create(h, s->variants, devices[r], &s->g[r]))) return rc;
        need[r].resize(n_devices);
        for (uint32_t p = 0; p < n_devices; ++p) need[r][p].swap(h.need[p]);
    }
    if (s->g[0]->geo.compact)
        for (uint32_t p = 0; p < n_devices; ++p) {
            const std::vector<uint32_t>* lists[8];
            static const std::vector<uint32_t> none;
            for (uint32_t c = 0; c < 8; ++c) lists[c] = c < n_devices && c != p ? &need[c][p] : &none;
            if ((rc = s->g[p]->set_send_lists(lists))) return rc;
        }
    s->vec_len = s->g[0]->vec_len;              // position order: the same on every share; gather vectors: see g[r]->vec_len
    s->gather = s->g[0]->geo.compact != 0;
    s->seg_stride = s->g[0]->geo.seg_stride();
    s->n_chunks = s->g[0]->geo.n_chunks;        // 1 where the graph does not take the blocked sweeps
    const size_t elem = (s->variants & ET_VARIANT_FR) ? 32 : 8;
    s->stream.assign(n_devices, nullptr);
    s->done.assign(n_devices, nullptr);
    s->pretrust.assign(n_devices, nullptr);
    for (int b = 0; b < 2; ++b) { s->vec[b].assign(n_devices, nullptr); s->sendbuf[b].assign(n_devices, nullptr); }
    for (uint32_t r = 0; r < n_devices; ++r) {
        HIP_TRY_S(hipSetDevice(devices[r]));
        HIP_TRY_S(hipStreamCreateWithFlags(&s->stream[r], hipStreamNonBlocking));
        HIP_TRY_S(hipEventCreateWithFlags(&s->done[r], hipEventDisableTiming));
        for (int b = 0; b < 2; ++b) HIP_TRY_S(hipMalloc(&s->vec[b][r], s->g[r]->vec_len * elem));
        if (s->gather)
            for (int b = 0; b < 2; ++b) HIP_TRY_S(hipMalloc(&s->sendbuf[b][r], std::max<size_t>(1, s->g[r]->send_total) * elem));
        if (distinct)
            for (uint32_t q = 0; q < n_devices; ++q) {
                if (q == r) continue;
                int can = 0;
                HIP_TRY_S(hipDeviceCanAccessPeer(&can, devices[r], devices[q]));
                if (can) {
                    hipError_t e = hipDeviceEnablePeerAccess(devices[q], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error(hipGetErrorString(e)); return ET_E_HIP; }
                    (void)hipGetLastError();
                }
            }
    }
    if (s->n_chunks > 1) {
        const Geometry& geo = s->g[0]->geo;
        uint32_t longest = 0;
        for (uint32_t q = 0; q < s->n_chunks; ++q) longest = std::max(longest, geo.chunk_begin(q + 1) - geo.chunk_begin(q));
        s->cstream.assign(n_devices, nullptr);
        s->ready.assign((size_t)n_devices * s->n_chunks, nullptr);
        s->staging.assign(n_devices, nullptr);
        for (uint32_t r = 0; r < n_devices; ++r) {
            HIP_TRY_S(hipSetDevice(devices[r]));
            HIP_TRY_S(hipStreamCreateWithFlags(&s->cstream[r], hipStreamNonBlocking));
            for (uint32_t q = 0; q < s->n_chunks; ++q) HIP_TRY_S(hipEventCreateWithFlags(&s->ready[(size_t)r * s->n_chunks + q], hipEventDisableTiming));
            if (exchange == ET_EXCHANGE_RCCL) HIP_TRY_S(hipMalloc(&s->staging[r], (size_t)n_devices * longest * elem));
        }
    }
    HIP_TRY_S(hipSetDevice(devices[0]));
    HIP_TRY_S(hipMalloc(&s->out_dev, (size_t)n * elem));
    if (s->gather) HIP_TRY_S(hipMalloc(&s->full, s->g[0]->full_len * elem));
    if (exchange == ET_EXCHANGE_RCCL) {
        s->comm.assign(n_devices, nullptr);
        RCCL_TRY(rccl().CommInitAll(s->comm.data(), (int)n_devices, s->device.data()));
    }
    *out = s.release();
    return ET_OK;
}

void et_sharded_free(et_sharded* s) { delete s; }

int et_sharded_exchange_kind(const et_sharded* s) { return s ? (int)s->exchange : ET_E_INVALID_ARG; }

int et_sharded_info_get(const et_sharded* s, uint32_t share, et_graph_info* info) {
    if (!s || !info || share >= s->world) return ET_E_INVALID_ARG;
    return et_graph_info_get(reinterpret_cast<const et_graph*>(s->g[share]), info);
}

int et_sharded_converge_fr(et_sharded* s, uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi, et_fr* out) {
    if (!s || !out) return ET_E_INVALID_ARG;
    if (!(s->variants & ET_VARIANT_FR)) { set_error("built without the Fr variant"); return ET_E_UNSUPPORTED; }
    int rc;
    for (uint32_t r = 0; r < s->world; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        if ((rc = s->g[r]->init_fr(initial_lo, initial_hi, (uint4*)s->vec[0][r], s->stream[r], num_iterations == 0))) return rc;
    }
    const bool chunked = s->n_chunks > 1;
    if (chunked) { if ((rc = start_chunked_exchange(s, 0, 32))) return rc; }
    else if (num_iterations && (rc = s->exchange_segments(0, 32))) return rc;
    for (uint32_t it = 0; it < num_iterations; ++it) {
        const int a = it & 1, b = (it + 1) & 1;
        const uint32_t flags = it + 1 == num_iterations ? ET_STEP_FINAL : ET_STEP_NO_STATS;
        for (uint32_t q = 0; q < s->n_chunks; ++q)
            for (uint32_t r = 0; r < s->world; ++r) {
                HIP_TRY_S(hipSetDevice(s->device[r]));
                if (chunked) HIP_TRY_S(hipStreamWaitEvent(s->stream[r], s->ready[(size_t)r * s->n_chunks + q], 0));
                if ((rc = s->g[r]->step_fr((const uint4*)s->vec[a][r], (uint4*)s->vec[b][r], it, flags, s->stream[r], chunked ? (int)q : -1))) return rc;
            }
        if (chunked) { if ((rc = start_chunked_exchange(s, b, 32))) return rc; }
        else if (it + 1 < num_iterations && (rc = s->exchange_segments(b, 32))) return rc;
    }
    if (chunked)   // the result is read from the first share's vector: all of its pieces must be in
        for (uint32_t q = 0; q < s->n_chunks; ++q) { HIP_TRY_S(hipSetDevice(s->device[0])); HIP_TRY_S(hipStreamWaitEvent(s->stream[0], s->ready[q], 0)); }
    void* result = s->vec[num_iterations & 1][0];
    if (!chunked && (rc = s->collect(num_iterations & 1, 32, &result))) return rc;
    HIP_TRY_S(hipSetDevice(s->device[0]));
    if ((rc = s->g[0]->unpermute_fr((const uint4*)result, (uint4*)s->out_dev, s->stream[0]))) return rc;
    HIP_TRY_S(hipMemcpyAsync(out, s->out_dev, (size_t)s->n * 32, hipMemcpyDeviceToHost, s->stream[0]));
    for (uint32_t r = 0; r < s->world; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        HIP_TRY_S(hipStreamSynchronize(s->stream[r]));
        if (chunked) HIP_TRY_S(hipStreamSynchronize(s->cstream[r]));
    }
    // conservation (native.rs:331-334) over the shares' partial sums
    if (num_iterations > 0) {
        Fr total = Fr::zero();
        for (uint32_t r = 0; r < s->world; ++r) {
            HIP_TRY_S(hipSetDevice(s->device[r]));
            et_iter_stats st;
            if ((rc = s->g[r]->stats_fr(num_iterations - 1, s->stream[r], &st))) return rc;
            FrRaw part;
            std::memcpy(part.l, st.sum.b, 32);
            total += Fr::from_raw(part);
        }
        const Fr expect = Fr::from_u128(initial_lo, initial_hi) * Fr::from_u64(s->n_valid);
        if (total.to_raw() != expect.to_raw()) { set_error("reputation leak: field sum of scores changed"); return ET_E_SUM_MISMATCH; }
    }
    return ET_OK;
}

int et_sharded_converge_f64(et_sharded* s, uint32_t num_iterations, double initial, double alpha, const double* pretrust,
                            double* out, double* residual_l1) {
    if (!s || !out) return ET_E_INVALID_ARG;
    if (!(s->variants & ET_VARIANT_F64)) { set_error("built without the f64 variant"); return ET_E_UNSUPPORTED; }
    int rc;
    std::vector<std::vector<double>> pos(alpha != 0.0 && pretrust ? s->world : 0);   // a share reads its own block of its own order
    for (uint32_t r = 0; r < s->world; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        if (alpha != 0.0) {
            const size_t len = s->g[r]->vec_len;
            if (!s->pretrust[r]) HIP_TRY_S(hipMalloc((void**)&s->pretrust[r], len * sizeof(double)));
            if (pretrust) {
                pos[r].resize(len);
                if ((rc = et_graph_permute_f64_host(reinterpret_cast<const et_graph*>(s->g[r]), pretrust, pos[r].data()))) return rc;
                HIP_TRY_S(hipMemcpyAsync(s->pretrust[r], pos[r].data(), len * sizeof(double), hipMemcpyHostToDevice, s->stream[r]));
            }
            else if ((rc = s->g[r]->init_f64(initial, s->pretrust[r], s->stream[r], true))) return rc;   // default: the initial scores
        }
        if ((rc = s->g[r]->init_f64(initial, (double*)s->vec[0][r], s->stream[r], num_iterations == 0))) return rc;
    }
    if (alpha != 0.0 && pretrust)   // the host staging vector must outlive the uploads
        for (uint32_t r = 0; r < s->world; ++r) { HIP_TRY_S(hipSetDevice(s->device[r])); HIP_TRY_S(hipStreamSynchronize(s->stream[r])); }
    const bool chunked = s->n_chunks > 1;
    if (chunked) { if ((rc = start_chunked_exchange(s, 0, 8))) return rc; }
    else if (num_iterations && (rc = s->exchange_segments(0, 8))) return rc;
    for (uint32_t it = 0; it < num_iterations; ++it) {
        const int a = it & 1, b = (it + 1) & 1;
        const uint32_t flags = it + 1 == num_iterations ? ET_STEP_FINAL : ET_STEP_NO_STATS;
        for (uint32_t q = 0; q < s->n_chunks; ++q)
            for (uint32_t r = 0; r < s->world; ++r) {
                HIP_TRY_S(hipSetDevice(s->device[r]));
                if (chunked) HIP_TRY_S(hipStreamWaitEvent(s->stream[r], s->ready[(size_t)r * s->n_chunks + q], 0));
                if ((rc = s->g[r]->step_f64((const double*)s->vec[a][r], (double*)s->vec[b][r], it, alpha,
                                            alpha != 0.0 ? s->pretrust[r] : nullptr, flags, s->stream[r], chunked ? (int)q : -1))) return rc;
            }
        if (chunked) { if ((rc = start_chunked_exchange(s, b, 8))) return rc; }
        else if (it + 1 < num_iterations && (rc = s->exchange_segments(b, 8))) return rc;
    }
    if (chunked)
        for (uint32_t q = 0; q < s->n_chunks; ++q) { HIP_TRY_S(hipSetDevice(s->device[0])); HIP_TRY_S(hipStreamWaitEvent(s->stream[0], s->ready[q], 0)); }
    void* result = s->vec[num_iterations & 1][0];
    if (!chunked && (rc = s->collect(num_iterations & 1, 8, &result))) return rc;
    HIP_TRY_S(hipSetDevice(s->device[0]));
    if ((rc = s->g[0]->unpermute_f64((const double*)result, (double*)s->out_dev, s->stream[0]))) return rc;
    HIP_TRY_S(hipMemcpyAsync(out, s->out_dev, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, s->stream[0]));
    for (uint32_t r = 0; r < s->world; ++r) {
        HIP_TRY_S(hipSetDevice(s->device[r]));
        HIP_TRY_S(hipStreamSynchronize(s->stream[r]));
        if (chunked) HIP_TRY_S(hipStreamSynchronize(s->cstream[r]));
    }
    if (residual_l1) {
        *residual_l1 = 0.0;
        if (num_iterations > 0)
            for (uint32_t r = 0; r < s->world; ++r) {
                HIP_TRY_S(hipSetDevice(s->device[r]));
                et_iter_stats st;
                if ((rc = s->g[r]->stats_f64(num_iterations - 1, s->stream[r], &st))) return rc;
                *residual_l1 += st.residual_l1;
            }
    }
    return ET_OK;
}

}  // extern "C"
