/*
 * eigentrust_hip.h -- C ABI of the MI355X-native EigenTrust convergence engine.
 *
 * The reference (kumavis/protocol, Rust) has no FFI for this path: it is the generic struct
 * `EigenTrustSet` in eigentrust-zk/src/circuits/dynamic_sets/native.rs with the methods
 * new / add_member / remove_member / update_op / converge / converge_rational, called from
 * eigentrust/src/lib.rs:422-440.  eigentrust-zk forbids unsafe code (src/lib.rs:9), so a thin
 * `-sys` crate binding the functions below is where the reference would link this library.
 *
 * Conventions
 *   - et_fr is the canonical 32-byte little-endian integer < r, i.e. `Fr::to_repr()`
 *     (eigentrust-zk/src/utils.rs:327-329).  Montgomery form never crosses this boundary.
 *   - Every function returns 0 (ET_OK) or a negative ET_E_* code; nothing aborts or throws.
 *     The reference panics instead (assert!/unwrap); each code names the panic it replaces.
 *   - The caller owns every host buffer passed in or out; the library owns the device memory
 *     behind et_graph / et_set and keeps no host pointer after a call returns.
 *   - Calls are synchronous unless the name ends in _async; an et_set / et_graph must not be used
 *     from two threads at once (the reference's `&mut self`).
 *   - The hot path runs on the GPU only.  Without a usable HIP device the compute entry points
 *     return ET_E_NO_DEVICE / ET_E_HIP; there is no CPU fallback.
 */
#ifndef EIGENTRUST_HIP_H
#define EIGENTRUST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint8_t b[32]; } et_fr;

enum {
    ET_OK = 0,
    ET_E_DUP_MEMBER = -1,         /* native.rs:178  assert!(pos.is_none())                  */
    ET_E_SET_FULL = -2,           /* native.rs:181  first_available.unwrap()                */
    ET_E_NOT_MEMBER = -3,         /* native.rs:192, opinion/native.rs:67                    */
    ET_E_INSUFFICIENT_PEERS = -4, /* native.rs:289  "Insufficient peers for calculation!"   */
    ET_E_SUM_MISMATCH = -5,       /* native.rs:334  assert!(sum_initial == sum_final)       */
    ET_E_NONCANONICAL_FR = -6,    /* a 32-byte value >= r                                   */
    ET_E_INVALID_ARG = -7,
    ET_E_HIP = -8,                /* a HIP runtime call failed; see et_last_error_message   */
    ET_E_NO_DEVICE = -9,
    ET_E_BUFFER_TOO_SMALL = -10,
    ET_E_OUT_OF_RANGE = -11,      /* slot / peer index >= number of slots                   */
    ET_E_UNSUPPORTED = -12,
    ET_E_THRESHOLD_RANGE = -13,   /* threshold/native.rs:37, :63  limb or threshold bound     */
    ET_E_RATIO_DIGITS = -14,      /* utils.rs:340  more decimal digits than the limbs hold    */
    ET_E_RATIO_MISMATCH = -15,    /* threshold/native.rs:81  num * den^-1 != score            */
    ET_E_ZERO_LIMB = -16,         /* threshold/native.rs:79, :90  zero denominator (limb)     */
    ET_E_BAD_DOMAIN = -17,        /* opinion/native.rs:76  attestation of another domain      */
    ET_E_RATIONAL_TOO_LARGE = -18,/* eigentrust/src/lib.rs:222-229  magnitude over 32 bytes   */
    ET_E_PARSE = -19,             /* storage.rs:286-331, circuit.rs:128-132  malformed input  */
    ET_E_BAD_SIGNATURE = -20,     /* ecdsa/native.rs:202-203, :316, :322-327  unwrap / assert */
    ET_E_BAD_ABOUT = -21,         /* opinion/native.rs:75  attestation about another peer     */
    ET_E_RCCL = -22               /* an RCCL call failed or librccl could not be loaded       */
};

const char *et_strerror(int code);
/* Detail of the last failure on the calling thread (e.g. the HIP error string). */
const char *et_last_error_message(void);
/* ABI version: major << 16 | minor. */
uint32_t et_abi_version(void);
/* Number of HIP devices visible, or a negative code. */
int et_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Set API: mirrors EigenTrustSet (native.rs:109-131) with run-time sizes.
 * ---------------------------------------------------------------------------------------- */
typedef struct et_set et_set;

/* EigenTrustSet::new (native.rs:165-172).  num_neighbours / num_iterations / initial_score are
 * the reference's const generics NUM_NEIGHBOURS, NUM_ITERATIONS, INITIAL_SCORE (u128, given as
 * two 64-bit halves); production values 4 / 20 / 1000 (circuits/mod.rs:39-43). */
int et_set_new(uint32_t num_neighbours, uint32_t num_iterations, uint64_t initial_score_lo,
               uint64_t initial_score_hi, const et_fr *domain, et_set **out);
void et_set_free(et_set *set);

/* add_member (native.rs:175-186): first empty slot, score = INITIAL_SCORE.
 * ET_E_DUP_MEMBER / ET_E_SET_FULL; the zero address is ET_E_INVALID_ARG (it marks empty slots). */
int et_set_add_member(et_set *set, const et_fr *addr);
/* remove_member (native.rs:189-198): slot -> (0, 0), the member's own opinion row is dropped;
 * other members' opinions about it stay stored and are nullified at filter time. */
int et_set_remove_member(et_set *set, const et_fr *addr);

/* update_op (native.rs:201-231) with the part of Opinion::validate (opinion/native.rs:63-109)
 * that is not signature cryptography: `scores` is the vector validate returns, one field element
 * per slot (zero where the attestation was absent or its signature invalid).  Scores about
 * empty slots are stored as zero (opinion/native.rs:92-98).  Replaces any earlier row of `from`.
 * ET_E_NOT_MEMBER when `from` is not in the set (opinion/native.rs:66-67). */
int et_set_update_op(et_set *set, const et_fr *from, const et_fr *scores /* num_neighbours */);
/* The same with only the non-zero entries listed (slots must be < num_neighbours). */
int et_set_update_op_sparse(et_set *set, const et_fr *from, uint32_t count,
                            const uint32_t *about_slot, const et_fr *values);
/* Bulk form for large sets: sender given by slot index, values as small integers (the production
 * opinion is a u8, eigentrust/src/attestation.rs:304). */
int et_set_update_op_u32(et_set *set, uint32_t from_slot, uint32_t count,
                         const uint32_t *about_slot, const uint32_t *values);

/* Slot layout: addresses (zero = empty) and current initial scores, num_neighbours each. */
int et_set_members(const et_set *set, et_fr *addrs, et_fr *scores);

/* converge (native.rs:286-337) on the GPU: filter, normalise, num_iterations field-exact
 * iterations, conservation check.  out: num_neighbours scores in slot order, empty slots 0.
 * ET_E_INSUFFICIENT_PEERS, ET_E_SUM_MISMATCH. */
int et_set_converge(et_set *set, et_fr *out);
/* The f64 variant of the same iteration, new = (1-alpha)*(C^T s + dangling term) + alpha*p.
 * alpha = 0 and pretrust = NULL reproduce converge's update; pretrust (num_neighbours doubles, slot
 * order) defaults to the initial scores.  residual_l1 (optional) = sum |s_k - s_{k-1}| of the last
 * iteration.  Scores are on the reference's scale (sum = INITIAL_SCORE * members). */
int et_set_converge_f64(et_set *set, double alpha, const double *pretrust, double *out,
                        double *residual_l1);
/* converge_rational (native.rs:340-392): exact fractions, computed on the host with the library's
 * own big integers (arbitrary precision has no GPU formulation; usable for small sets only, as in
 * the reference).  Slot i's reduced numerator / denominator are written as little-endian
 * magnitudes, zero padded, at num_le + i*stride and den_le + i*stride.  If a value needs more
 * than `stride` bytes nothing is written, *needed (optional) receives the stride required and
 * ET_E_BUFFER_TOO_SMALL is returned. */
int et_set_converge_rational(const et_set *set, uint8_t *num_le, uint8_t *den_le, size_t stride,
                             size_t *needed);

/* ------------------------------------------------------------------------------------------
 * Graph engine: what et_set_converge runs on, exposed for large graphs, sharded runs and benches.
 * ---------------------------------------------------------------------------------------- */
typedef struct et_graph et_graph;

enum { ET_VARIANT_FR = 1, ET_VARIANT_F64 = 2 };

typedef struct {
    uint32_t n_seg;         /* GPUs sharing the rows of this graph (1 = whole graph here)        */
    uint32_t seg;           /* which share this et_graph holds                                   */
    uint32_t long_row;      /* 0 = default; in-degree above which a row gets a whole wavefront   */
    uint32_t target_work; /* 0 = default; wavefront work items per launch                      */
    uint32_t variants;      /* ET_VARIANT_* bit mask; 0 = both                                   */
    int32_t device;         /* HIP device ordinal; -1 = current device                           */
    uint32_t n_chunks;      /* 0 / 1 = off; K > 1: a segment is exchanged in K pieces and an iteration
                               runs as K launches (et_graph_step_*_phase_async), so that the exchange of
                               piece q + 1 can run beside the launch over piece q.  Honoured where both
                               variants run the blocked sweep (see et_graph_info.n_chunks)               */
    uint32_t flags;         /* ET_BUILD_* bit mask                                                    */
    uint32_t xcd_cost[8];   /* relative time per edge of a workgroup on XCD 0..7, in 1/1024 (workgroup b of a
                               sweep runs on XCD b mod 8): the deal levels edges x cost.  All zero = measured on
                               the device by a short probe (et_graph_build, et_dplan_finish, et_sharded_create) or
                               uniform (et_plan_build).  et_graph_info.xcd_cost reports what a layout was levelled
                               with; passing those values back reproduces the layout                          */
} et_build_opts;
/* Several shares: keep every share's whole segment in the trust vectors (position order, all-gather exchange) even
 * where the gather vector (see et_exchange_info) would be used. */
#define ET_BUILD_FULL_VECTOR 1u
/* The graph will be converged many times (thousands) before it changes: the planner may then spend about 60 % more time
 * (label propagation over the attestations, 0.4 s at a million peers) on positions that put a peer beside the peers the
 * same workgroup gathers, which saves about 13 % per iteration (see et_plan_sweep_stats).  Without the flag a single
 * share keeps the plain degree order.  ET_AFFINITY=1 / 0 in the environment forces either. */
#define ET_BUILD_REUSED 2u

typedef struct {
    uint32_t n;             /* slots                                                             */
    uint32_t n_valid;       /* members                                                           */
    uint32_t n_dangling;    /* members whose filtered opinion row is empty                       */
    uint32_t n_seg, seg;
    uint64_t nnz;           /* surviving attestations of non-dangling members (all shares)       */
    uint64_t nnz_local;     /* ... whose rated peer belongs to this share                        */
    uint64_t padded_local;  /* edge-stream entries incl. padding in this share                   */
    uint32_t n_slices, n_work;
    uint32_t rows_cap, nd_cap; /* position-space geometry: segment = rows_cap + nd_cap entries   */
    uint64_t vec_len;       /* entries of the trust vectors the steps of this share read and write: all
                               segments in position order, or this share's gather vector (gather != 0)   */
    uint64_t device_bytes;  /* device memory held by this graph                                  */
    uint32_t f64_sweep_blocks;   /* workgroups of the f64 blocked sweep (0: variant not prepared)     */
    uint32_t exact_sweep_blocks; /* workgroups of the exact blocked sweep; 0: the exact variant runs
                                    its row-per-lane kernel on this graph                             */
    uint32_t n_chunks;      /* pieces a segment is exchanged in (1 = whole)                            */
    uint32_t chunk_rows;    /* piece q of a segment = its entries [q*chunk_rows, (q+1)*chunk_rows); the
                               last piece runs to the end of the segment (rows_cap + nd_cap)            */
    uint32_t gather;        /* 1: the share's vectors are gather vectors (see et_exchange_info)          */
    uint64_t full_len;      /* n_seg * (rows_cap + nd_cap): entries of a vector in position order, which is
                               what et_graph_unpermute_* reads                                            */
    uint32_t xcd_cost[8];   /* the per-XCD cost factors the layout was levelled with (see et_build_opts)  */
} et_graph_info;

/* Filter + normalise + lay out an attestation list (see et_set_update_op for the meaning of a
 * row; a later entry for the same (from, to) replaces an earlier one, eigentrust/src/lib.rs:415).
 * member: n flags or NULL (= every slot holds a peer).  Exactly one of value_u32 / value_fr. */
int et_graph_build(uint32_t n, const uint8_t *member, uint64_t n_edges, const uint32_t *from,
                   const uint32_t *to, const uint32_t *value_u32, const et_fr *value_fr,
                   const et_build_opts *opts, et_graph **out);
void et_graph_free(et_graph *g);
int et_graph_info_get(const et_graph *g, et_graph_info *info);

/* ---- host-side plan (no GPU needed): the layout et_graph_build uploads ----------------------
 * Lets a caller (or a test) inspect how an attestation list was filtered, ordered, sliced and
 * split over segments.  Arrays stay owned by the plan. */
typedef struct et_plan et_plan;
typedef struct { uint32_t off, len, row0, nrows; } et_slice_desc;   /* nrows bits 0-7: rows, bits 8-11: log2(lanes per row) */
typedef struct { uint32_t slice_begin, slice_end, flags, dslot; } et_work_desc;  /* one wavefront; flags bit 0: dangling rows */
typedef struct {
    const uint32_t *pos_of_slot;      /* n: position of each slot's peer, 0xFFFFFFFF for empty slots  */
    const et_slice_desc *slices;      /* n_slices                                                      */
    const et_work_desc *work;        /* n_work                                                      */
    const uint32_t *col;              /* padded_local sources, (segment << 29) | local row; step-major */
    const uint32_t *value_u32;        /* padded_local raw opinions, or NULL                            */
    const et_fr *value_fr;            /* padded_local raw opinions, or NULL                            */
    const et_fr *scale_fr;            /* vec_len: inv(row sum) * 2^576 mod r per source position       */
    const double *scale_f64;          /* vec_len: 1 / integer row sum per source position              */
    et_fr inv_nm1_w;                  /* inv(n_valid - 1) * 2^288 mod r                                */
    uint32_t n_active, n_active_pad, n_dangling_pad, n_dangling_local;
} et_plan_arrays;
int et_plan_build(uint32_t n, const uint8_t *member, uint64_t n_edges, const uint32_t *from,
                  const uint32_t *to, const uint32_t *value_u32, const et_fr *value_fr,
                  const et_build_opts *opts, et_plan **out);
void et_plan_free(et_plan *p);
int et_plan_info_get(const et_plan *p, et_graph_info *info);   /* device_bytes = 0 */
int et_plan_arrays_get(const et_plan *p, et_plan_arrays *out);
/* How a plan's blocked sweep is laid out for one variant (ET_VARIANT_FR or ET_VARIANT_F64).  No counterpart in the
 * reference (it multiplies one dense matrix, dynamic_sets/native.rs:319-329); for callers that want to see what a launch
 * will cost before uploading.  line_fills = distinct (workgroup, 128-byte line of the trust vector) pairs of one launch:
 * the number of lines the compute units pull into their L1s, which is what bounds a sweep.  clustered = 1 when the
 * positions of peers with few opinions were placed beside the peers the same workgroup gathers (one share, byte
 * opinions, both variants on the sweep, one's workgroup count a multiple of the other's; ET_AFFINITY=0 switches it off). */
typedef struct {
    uint32_t blocks;          /* workgroups; 0 = this variant does not run the blocked sweep on this plan */
    uint32_t clustered;
    uint64_t edges, edges_most;            /* all workgroups, the busiest one */
    uint64_t line_fills, line_fills_most;
    uint64_t chunk_span_most;              /* widest range of positions 64 consecutive edges of a list span */
    uint64_t xcd_line_fills;               /* distinct (line, XCD) pairs, workgroup b on XCD b mod 8: the lines the eight L2s pull */
} et_sweep_stats;
int et_plan_sweep_stats(const et_plan *p, int variant, et_sweep_stats *out);
/* Uploads a plan (what et_graph_build does after planning). */
int et_graph_from_plan(const et_plan *p, const et_build_opts *opts, et_graph **out);

/* Whole converge on this graph (n_seg must be 1): scores for all n slots in slot order. */
int et_graph_converge_fr(et_graph *g, uint32_t num_iterations, uint64_t initial_lo,
                         uint64_t initial_hi, et_fr *out);
int et_graph_converge_f64(et_graph *g, uint32_t num_iterations, double initial, double alpha,
                          const double *pretrust, double *out, double *residual_l1);

/* ---- device-resident stepping (benches, the sharded multi-GPU driver) ------------------------
 * Trust vectors are device buffers of vec_len entries in position order (et_fr = 32 bytes, or
 * double), allocated by the caller; they need no zero fill (init and every step write their whole
 * segment, partial-sum tail included).  `stream` is a hipStream_t (NULL = default stream).  These
 * calls make the graph's device current, only enqueue work and allocate nothing. */

/* Writes this share's segment of the initial vector (INITIAL_SCORE for members, native.rs:184,317)
 * plus its partial sums into x; other segments are left untouched (all-gather them).  Without
 * ET_INIT_PLAIN the vector is in the form the steps exchange (see et_graph_step_*). */
#define ET_INIT_PLAIN 1u   /* write plain scores: a pre-trust vector, or a converge of zero iterations */
int et_graph_init_fr_async(et_graph *g, uint64_t initial_lo, uint64_t initial_hi, void *x,
                           uint32_t flags, void *stream);
int et_graph_init_f64_async(et_graph *g, double initial, void *x, uint32_t flags, void *stream);
/* One iteration (native.rs:320-328 for every row of this share): reads all of x, writes this
 * share's segment of y.  `iteration` selects the slot of the per-iteration statistics.
 * pretrust_pos: device doubles in position order, or NULL.
 *
 * When the opinions are integers the weights c_ji = v_ji * inv(sum_j) (native.rs:305-314) are
 * never stored: between iterations the vectors hold s_j * inv(sum_j), the edge records hold v_ji,
 * and the plain scores of the share's own rows are kept inside the graph object (so one graph
 * serves one converge at a time).  The LAST step of a converge must carry ET_STEP_FINAL: its
 * output holds plain scores and is the only one et_graph_unpermute_* may be given.  A vector
 * written by et_graph_init_* must go through at least one step before it is read back. */
#define ET_STEP_FINAL 1u
/* No statistics for this iteration: the exact step does not read the old entries to count the rows that changed (32
 * bytes per row less to move) and forms no score sum, the f64 step reads no old entry and forms neither residual nor
 * score sum (a division per row less).  et_iter_stats of that iteration is unspecified (zeros from the blocked sweeps;
 * the row-per-lane kernels form their statistics regardless).  A converge sets it on every step but the last. */
#define ET_STEP_NO_STATS 2u
int et_graph_step_fr_async(et_graph *g, const void *x, void *y, uint32_t iteration, uint32_t flags,
                           void *stream);
int et_graph_step_f64_async(et_graph *g, const void *x, void *y, uint32_t iteration, double alpha,
                            const void *pretrust_pos, uint32_t flags, void *stream);
/* One of the n_chunks launches of an iteration (graphs built with et_build_opts.n_chunks > 1): launch
 * `phase` reads only the entries of x in piece `phase` of every segment, so it may be enqueued as soon as those
 * have arrived; the launches of one iteration run in order 0 .. n_chunks-1 on one stream, the last one writes
 * this share's segment of y.  (et_graph_step_*_async on such a graph enqueues all of them.) */
int et_graph_step_fr_phase_async(et_graph *g, const void *x, void *y, uint32_t iteration,
                                 uint32_t flags, uint32_t phase, void *stream);
int et_graph_step_f64_phase_async(et_graph *g, const void *x, void *y, uint32_t iteration, double alpha,
                                  const void *pretrust_pos, uint32_t flags, uint32_t phase,
                                  void *stream);
/* ---- the exchange between the shares of one graph ------------------------------------------------
 * The reference iterates one matrix in one address space (native.rs:319-329) and has no counterpart.  Here a
 * step writes this share's own block of y (its rows' entries, then its partial sums of the dangling peers' score
 * mass: own_len entries at own_off) and the shares exchange before the next step.
 *   gather == 0 (one share; field-valued opinions; a chunked exchange; ET_BUILD_FULL_VECTOR): vectors hold every
 *     share's block in position order, own_off = seg * own_len, and the exchange is an in-place all-gather of blocks.
 *   gather == 1 (several shares, integer opinions): a share's vectors hold its own block first (own_off = 0), then the
 *     entries of its OWN rows that its rows cite (self_count entries at self_off, in the order its sweeps want them), and
 *     then, for every other share p in ascending order, only the entries of p's rows that this share's rows cite, followed
 *     by p's partial sums: recv_count[p] entries at recv_off[p].  The sweeps gather from these blocks; the own block is what
 *     a step writes.  After a step, et_graph_pack_*_async copies what every other share c cites out of the own block of y
 *     into a send buffer of send_total entries (send_count[c] entries at send_off[c]) and the share's own cited entries
 *     into their block of y; an all-to-all (send block c to share c, receive block p at recv_off[p] of y: one contiguous
 *     range in share order) completes y.  The pack also follows et_graph_init_*.
 *     The 16M-peer benchmark graph on 8 shares: 8.0M of 14.7M entries received per share and iteration, i.e. 0.54 of the
 *     all-gather's bytes.
 *     The LAST step's y holds plain scores of the own block only; the result in slot order needs the shares' own blocks
 *     all-gathered into a vector in position order (full_len entries, block p at p * own_len) for et_graph_unpermute_*. */
typedef struct {
    uint32_t gather;
    uint32_t n_seg, seg;
    uint32_t own_off, own_len;
    uint64_t vec_len, full_len;
    uint32_t recv_off[8], recv_count[8];   /* entries; zero for the share itself                         */
    uint32_t send_off[8], send_count[8];   /* gather == 0: every share is sent the whole own block        */
    uint64_t send_total;                   /* entries of the pack buffer                                   */
    uint32_t self_off, self_count;         /* gather == 1: the block of the share's own cited entries      */
} et_exchange_info;
int et_graph_exchange_get(const et_graph *g, et_exchange_info *out);
/* Fills `send` (send_total entries of 32 / 8 bytes) from the own block of y and, with a gather vector, the block of the
 * share's own cited entries inside y.  gather == 0: a copy of the own block. */
int et_graph_pack_fr_async(et_graph *g, void *y, void *send, void *stream);
int et_graph_pack_f64_async(et_graph *g, void *y, void *send, void *stream);
/* The exchange tables of a host-side plan and, with a gather vector, the lists behind them: need[p] = the local rows of
 * share p this share cites (recv_count[p] - nd_cap entries; p == seg: self_count entries), in the order of their block,
 * send[c] = the local rows of this share that share c cites (send_count[c] - nd_cap entries).  need / send: arrays of 8
 * pointers to fill, or NULL. */
int et_plan_exchange_get(const et_plan *p, et_exchange_info *out, const uint32_t **need, const uint32_t **send);

/* Position order -> slot order (device to device): x has full_len entries in position order, out has n entries. */
int et_graph_unpermute_fr_async(et_graph *g, const void *x, void *out_slots, void *stream);
int et_graph_unpermute_f64_async(et_graph *g, const void *x, void *out_slots, void *stream);
/* Slot order -> the order of this share's vectors for a host vector of doubles (pretrust); fills vec_len entries
 * (a step reads the entries of the share's own block only; with a gather vector only those are filled). */
int et_graph_permute_f64_host(const et_graph *g, const double *slots, double *pos);

typedef struct {
    et_fr sum;              /* field sum of this share's new scores (conservation, native.rs:333) */
    uint64_t changed;       /* rows of this share whose score changed in that iteration           */
    double residual_l1;     /* f64: sum |new - old| over this share                               */
    double sum_f64;         /* f64: sum of this share's new scores                                */
} et_iter_stats;
/* Reduces the per-workgroup statistics of `iteration` (synchronises the stream). */
int et_graph_stats_fr(et_graph *g, uint32_t iteration, void *stream, et_iter_stats *out);
int et_graph_stats_f64(et_graph *g, uint32_t iteration, void *stream, et_iter_stats *out);

/* Runs init + num_iterations steps + unpermute back to back on `stream` with the graph's own
 * double-buffered vectors (n_seg must be 1); leaves the slot-order result on the device and does
 * not synchronise.  With record_events != 0 the iteration launches are bracketed by a pair of HIP
 * events taken from a ring of 1024 pairs (see et_graph_event_times). */
int et_graph_run_fr(et_graph *g, uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi,
                    void *stream, int record_events);
int et_graph_run_f64(et_graph *g, uint32_t num_iterations, double initial, double alpha,
                     void *stream, int record_events);
/* The same bracketing for callers that issue et_graph_step_* themselves. */
int et_graph_record_begin(et_graph *g, void *stream);
int et_graph_record_end(et_graph *g, void *stream);
/* Waits for every recorded pair, returns the sum of their elapsed times and their number, and
 * empties the ring. */
int et_graph_event_times(et_graph *g, float *total_ms, uint32_t *pairs);
/* Copies the result of the last et_graph_run_* to the host (n entries, slot order). */
int et_graph_result_fr(et_graph *g, et_fr *out, void *stream);
int et_graph_result_f64(et_graph *g, double *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Row-sharded converge over the GPUs of one node, driven by ONE host thread (a Rust caller has no
 * torch.distributed).  The reference has no counterpart: its converge runs one dense matrix on one core
 * (native.rs:286-337); results here are the same bits as et_graph_converge_fr on one GPU.
 * Rows are dealt to n_devices shares (<= 8); per iteration every GPU runs the step on its rows and the
 * segments are exchanged once: RCCL all-gather over xGMI, or device-to-device pulls ordered by events.
 *   devices  : n_devices HIP ordinals.  Listing a device more than once rehearses the N-share layout on
 *              fewer GPUs (exchange by copies: RCCL refuses two ranks of a communicator on one device).
 *   exchange : ET_EXCHANGE_AUTO = RCCL when the devices are distinct and librccl loads, else copies.
 *   n_chunks : 0 / 1 = one exchange after each iteration's launch; K > 1 = the segments travel in K pieces on
 *              a second stream per share while the iteration's K launches run, launch q waiting only for
 *              piece q (see et_build_opts.n_chunks).
 * Errors: those of et_graph_build, ET_E_RCCL, ET_E_SUM_MISMATCH (native.rs:334).
 * ---------------------------------------------------------------------------------------- */
typedef struct et_sharded et_sharded;
enum { ET_EXCHANGE_AUTO = 0, ET_EXCHANGE_RCCL = 1, ET_EXCHANGE_COPY = 2 };
int et_sharded_create(uint32_t n, const uint8_t *member, uint64_t n_edges, const uint32_t *from,
                      const uint32_t *to, const uint32_t *value_u32, const et_fr *value_fr,
                      uint32_t n_devices, const int32_t *devices, uint32_t variants, uint32_t exchange,
                      uint32_t n_chunks, et_sharded **out);
void et_sharded_free(et_sharded *s);
/* ET_EXCHANGE_RCCL or ET_EXCHANGE_COPY: what et_sharded_create settled on. */
int et_sharded_exchange_kind(const et_sharded *s);
int et_sharded_info_get(const et_sharded *s, uint32_t share, et_graph_info *info);
/* Whole converge: scores for all n slots in slot order (host buffers); synchronous. */
int et_sharded_converge_fr(et_sharded *s, uint32_t num_iterations, uint64_t initial_lo,
                           uint64_t initial_hi, et_fr *out);
int et_sharded_converge_f64(et_sharded *s, uint32_t num_iterations, double initial, double alpha,
                            const double *pretrust, double *out, double *residual_l1);

/* ------------------------------------------------------------------------------------------
 * One share of a row-sharded graph, laid out by a process that holds only the attestations of ITS OWN
 * attesters (one process per GPU, each ingesting a disjoint set of attesters): no process ever holds the
 * whole attestation list.  The reference builds its matrix from the whole list in one process
 * (eigentrust/src/lib.rs:398-440); the filter, the last-write-wins rule and the row sums applied here are
 * the same ones (native.rs:234-283, :305-314), per attester, which is why EVERY attestation of an attester
 * must be given to the same process, in its original order.  Integer opinions only.
 *   1. et_dplan_create        : this process's attestations; filter, row sums, degrees.
 *   2. et_dplan_local_stats   : per-peer figures (n entries each).  The caller sums row_sum, in_degree, out_degree
 *                               and attested (the number of surviving attestations an attester wrote, whatever
 *                               their values) over the processes and takes the maximum of value_max (one
 *                               all-reduce each; the maximum as an unsigned or 64-bit number), then
 *      et_dplan_set_global_stats: the summed figures and the build options (n_seg = number of shares, seg =
 *                               this process's share; variants, device, n_chunks, flags as for et_graph_build).
 *                               Every process derives the same order, geometry and positions.
 *                               ET_E_INVALID_ARG when an attester's attestations reached two processes, or when
 *                               value_max is below this process's own maximum.
 *   3. et_dplan_outgoing      : the surviving local edges whose destination row share `share` owns, as three
 *                               parallel arrays (source position code, destination local row, opinion); the
 *                               caller delivers them (one all-to-all), own share included, and hands what it
 *                               received to et_dplan_add_incoming, in any order, in any number of calls.
 *   4. et_dplan_needs         : (ends step 3) the local rows of share `share` that this share's rows cite,
 *                               ascending; the caller delivers each list to its share (one all-to-all) and hands
 *                               what share c sent it to et_dplan_add_wanted(c): the rows this share packs for c
 *                               after every step (see et_exchange_info).  Lists are empty for the share itself and
 *                               in layouts without a gather vector (et_dplan_info_get: gather == 0), where step 4
 *                               may be skipped.
 *   5. et_dplan_finish        : lays out the share from the records received and uploads it; the graph steps
 *                               and exchanges like one built by et_graph_build with the same n_seg / seg, and
 *                               its results are the same bits.
 * The pointers et_dplan_outgoing returns stay valid until et_dplan_needs, et_dplan_finish or et_dplan_free; those of
 * et_dplan_needs until et_dplan_finish or et_dplan_free.
 * ---------------------------------------------------------------------------------------- */
typedef struct et_dplan et_dplan;
int et_dplan_create(uint32_t n, const uint8_t *member, uint64_t n_edges, const uint32_t *from,
                    const uint32_t *to, const uint32_t *value_u32, et_dplan **out);
int et_dplan_local_stats(const et_dplan *p, uint64_t *row_sum, uint32_t *in_degree,
                         uint32_t *out_degree, uint32_t *attested, uint32_t *value_max);
int et_dplan_set_global_stats(et_dplan *p, const uint64_t *row_sum, const uint32_t *in_degree,
                              const uint32_t *out_degree, const uint32_t *attested, uint32_t value_max,
                              const et_build_opts *opts);
int et_dplan_outgoing(const et_dplan *p, uint32_t share, uint64_t *count, const uint32_t **src,
                      const uint32_t **row, const uint32_t **value);
int et_dplan_add_incoming(et_dplan *p, uint64_t count, const uint32_t *src, const uint32_t *row,
                          const uint32_t *value);
int et_dplan_needs(et_dplan *p, uint32_t share, uint64_t *count, const uint32_t **rows);
int et_dplan_add_wanted(et_dplan *p, uint32_t share, uint64_t count, const uint32_t *rows);
/* Geometry of the share (n_valid, n_dangling, nnz, rows_cap, nd_cap, full_len, gather, n_chunks ...): valid after
 * step 2; vec_len of a gather vector once et_dplan_needs has run. */
int et_dplan_info_get(const et_dplan *p, et_graph_info *info);
int et_dplan_finish(et_dplan *p, et_graph **out);
/* Step 4 without a GPU: the share's host-side plan (et_plan_info_get, et_plan_arrays_get, et_graph_from_plan). */
int et_dplan_finish_host(et_dplan *p, et_plan **out);
void et_dplan_free(et_dplan *p);

/* ------------------------------------------------------------------------------------------
 * Batches of small independent sets (ZK witness generation): n_sets sets of n <= 1024 slots.
 * One workgroup per set runs the whole converge in a single launch; sets never communicate, so a
 * batch is split over GPUs by creating one sub-batch per GPU.  Sets whose opinions fit a byte (the
 * production range, eigentrust/src/attestation.rs:304) run their dense product on the matrix cores;
 * larger opinions run on the vector ALUs.  Results are bit-exact either way.
 *   member : n_sets*n flags or NULL (every slot holds a peer)
 *   values : n_sets*n*n opinions, values[s][i][j] = what Opinion::validate returns for attester i
 *            about slot j (opinion/native.rs:63-109), as small integers
 *   out    : n_sets*n scores in slot order; status: per set ET_OK / ET_E_INSUFFICIENT_PEERS
 *            (native.rs:289) / ET_E_SUM_MISMATCH (native.rs:334)
 * ---------------------------------------------------------------------------------------- */
typedef struct et_batch et_batch;
int et_batch_create(uint32_t n_sets, uint32_t n, const uint8_t *member, const uint32_t *values,
                    int device, et_batch **out);
void et_batch_free(et_batch *b);
/* Enqueues the converge of every set on `stream`; allocates nothing, does not synchronise. */
int et_batch_run_fr_async(et_batch *b, uint32_t num_iterations, uint64_t initial_lo,
                          uint64_t initial_hi, void *stream);
int et_batch_result_fr(et_batch *b, et_fr *out, int32_t *status, void *stream);
/* Sum over sets of n_valid*(n_valid-1): the dense edge count the reference iterates over. */
int et_batch_edges(const et_batch *b, uint64_t *edges);
/* create + run + result + free. */
int et_batch_converge_fr(uint32_t n_sets, uint32_t n, const uint8_t *member, const uint32_t *values,
                         uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi,
                         et_fr *out, int32_t *status);

/* et_batch_converge_fr for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc): same arguments, same results bit for bit, but no hipMalloc / hipFree -- the
 * batch's device arrays are carved out of one array of ET_NOALLOC_SCRATCH_BYTES that is part of the
 * library's code object (each GPU the library runs on holds its copy until the process ends).  This
 * entry point has no counterpart in the reference.  It runs on the current device, is synchronous,
 * and returns ET_E_BUFFER_TOO_SMALL when the batch needs more than the array holds (byte opinions:
 * about n_sets * (roundup(n, 64)^2 + 72 n) bytes; larger opinions 4 n^2 + 72 n per set).  Every call
 * on a GPU uses that GPU's one array: calls from several host threads are serialised by a lock inside
 * the library, and nothing else may be using the array's memory (there is no other way to reach it). */
#define ET_NOALLOC_SCRATCH_BYTES (4u << 20)
#define ET_NOALLOC_MAX_DEVICES 64
int et_batch_converge_fr_noalloc(uint32_t n_sets, uint32_t n, const uint8_t *member,
                                 const uint32_t *values, uint32_t num_iterations, uint64_t initial_lo,
                                 uint64_t initial_hi, et_fr *out, int32_t *status);

/* ------------------------------------------------------------------------------------------
 * Deterministic synthetic attestation lists (benchmarks and parity tests use the same inputs).
 * kind: 2 = uniform out-degree, 3 = power-law out-degree with preferential targets and a share of
 * peers without opinions.  Two-call protocol: with from == NULL only *n_edges is written.
 * ---------------------------------------------------------------------------------------- */
int et_synth_graph(uint32_t kind, uint32_t n, uint64_t target_edges, uint64_t seed,
                   uint32_t value_max, uint64_t *n_edges, uint32_t *from, uint32_t *to,
                   uint32_t *value);
/* The same list restricted to the attesters [attester_begin, attester_end): what one of several ingesting
 * processes generates.  The concatenation over a partition of [0, n) is et_synth_graph's list. */
int et_synth_graph_range(uint32_t kind, uint32_t n, uint64_t target_edges, uint64_t seed,
                         uint32_t value_max, uint32_t attester_begin, uint32_t attester_end,
                         uint64_t *n_edges, uint32_t *from, uint32_t *to, uint32_t *value);

/* ------------------------------------------------------------------------------------------
 * Around the converge path (host side): threshold check, score packaging, the client's CSV
 * formats, the ET public-input layout.  No GPU work except inside et_local_scores.
 * Rational numbers are little-endian magnitudes, as et_set_converge_rational writes them. */

/* big_to_fe_rat (eigentrust-zk/src/utils.rs:332-354): both magnitudes scaled by the same power
 * of ten so that the larger has num_limbs*power_of_ten digits, then cut into num_limbs limbs of
 * power_of_ten decimal digits, least significant first (reference: 2 limbs of 72 digits,
 * circuits/mod.rs:53-55).  ET_E_RATIO_DIGITS where the reference's usize subtraction panics. */
int et_ratio_decompose(const uint8_t *num_le, size_t num_len, const uint8_t *den_le, size_t den_len,
                       uint32_t num_limbs, uint32_t power_of_ten, et_fr *num_limbs_out,
                       et_fr *den_limbs_out);
/* Threshold::new + check_threshold (threshold/native.rs:33-96): *passed = 1 when the top limb of
 * the numerator >= top limb of the denominator * threshold.  Every assert of the reference is an
 * error code: ET_E_THRESHOLD_RANGE, ET_E_RATIO_DIGITS, ET_E_RATIO_MISMATCH, ET_E_ZERO_LIMB. */
int et_threshold_check(const et_fr *score_fr, const uint8_t *num_le, size_t num_len,
                       const uint8_t *den_le, size_t den_len, const et_fr *threshold,
                       uint32_t num_limbs, uint32_t power_of_ten, uint32_t num_neighbours,
                       uint64_t initial_lo, uint64_t initial_hi, int *passed);

/* scalar_from_address (eigentrust/src/eth.rs:77-95): a 20-byte Ethereum address as a scalar. */
int et_scalar_from_address(const uint8_t address[20], et_fr *out);

/* Score (eigentrust/src/lib.rs:207-233): all magnitudes big-endian, right-aligned in 32 bytes. */
typedef struct {
    uint8_t address[20];
    uint8_t score_fr[32];
    uint8_t numerator[32];
    uint8_t denominator[32];
    uint8_t score[32];          /* floor(numerator / denominator) */
} et_score;
/* ET_E_RATIONAL_TOO_LARGE where the reference's `32 - len` index panics (lib.rs:222-229). */
int et_score_pack(const uint8_t address[20], const et_fr *score_fr, const uint8_t *num_le,
                  size_t num_len, const uint8_t *den_le, size_t den_len, et_score *out);

/* SignedAttestationRaw (eigentrust/src/attestation.rs:283-330), the record of attestations.csv. */
typedef struct {
    uint8_t about[20];
    uint8_t domain[20];
    uint8_t value;
    uint8_t message[32];
    uint8_t sig_r[32];
    uint8_t sig_s[32];
    uint8_t rec_id;
} et_attestation;
/* attestations.csv (storage.rs:247-307; header about,domain,value,message,sig_r,sig_s,rec_id).
 * parse: out may be NULL to count records; format/scores: *needed = bytes incl. the final NUL. */
int et_attestations_csv_parse(const char *text, size_t len, et_attestation *out, size_t cap,
                              size_t *count);
int et_attestations_csv_format(const et_attestation *recs, size_t n, char *buf, size_t cap,
                               size_t *needed);
/* scores.csv (storage.rs:184-217; header peer_address,score_fr,numerator,denominator,score). */
int et_scores_csv_format(const et_score *scores, size_t n, char *buf, size_t cap, size_t *needed);

/* ETPublicInputs::to_bytes / from_bytes (eigentrust/src/circuit.rs:84-151):
 * participants || scores || domain || opinions_hash as 32-byte little-endian scalars. */
int et_public_inputs_pack(const et_fr *participants, const et_fr *scores, uint32_t n,
                          const et_fr *domain, const et_fr *opinions_hash, uint8_t *out, size_t cap);
int et_public_inputs_unpack(const uint8_t *bytes, size_t len, uint32_t n, et_fr *participants,
                            et_fr *scores, et_fr *domain, et_fr *opinions_hash);

/* calculate_scores (eigentrust/src/lib.rs:201-236 over et_circuit_setup :339-440) for attestations
 * whose attester is already known: attesters[k] (20 bytes each) is the address recovered from
 * attestation k's signature -- ECDSA recovery and verification (opinion/native.rs:84-87) are NOT
 * done here.  Participants = all attesters and all `about` addresses in ascending order; a later
 * attestation of the same (attester, about) pair replaces an earlier one; converge runs on the GPU,
 * converge_rational on the host.  out may be NULL to query *n_participants.
 * Errors: ET_E_SET_FULL (lib.rs:366-369), ET_E_INSUFFICIENT_PEERS (:371-374), ET_E_BAD_DOMAIN,
 * ET_E_RATIONAL_TOO_LARGE, and those of et_set_converge. */
int et_local_scores(const et_attestation *atts, const uint8_t *attesters, size_t n_att,
                    const uint8_t domain[20], uint32_t num_neighbours, uint32_t num_iterations,
                    uint64_t initial_lo, uint64_t initial_hi, et_score *out, size_t cap,
                    size_t *n_participants);

/* ------------------------------------------------------------------------------------------
 * Opinion validation crypto (host side): what turns signed attestation records into the score
 * rows update_op stores.  Public keys are 64 bytes: x then y, each little-endian (the layout of
 * PublicKey::to_bytes, ecdsa/native.rs:124-131). */

/* Poseidon 5x5 permutation (poseidon/native/mod.rs:34-97; known answer :122-146) and the sponge
 * over it (poseidon/native/sponge.rs:46-68). */
int et_poseidon5_permute(const et_fr *in /* 5 */, et_fr *out /* 5 */);
int et_poseidon5_sponge(const et_fr *in, size_t n, et_fr *out);
int et_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);
/* Attestation::hash over (about, domain, value, message, 0) with the scalar conversions of
 * AttestationEth::to_attestation_fr (dynamic_sets/native.rs:96-104, attestation.rs:81-115). */
int et_attestation_hash(const et_attestation *att, et_fr *out);
/* SignedAttestationEth::recover_public_key + address_from_ecdsa_key for n records
 * (attestation.rs:215-239, ecdsa/native.rs:298-331, eth.rs:70-74): attesters = n*20 bytes,
 * public_keys = n*64 bytes or NULL.  ET_E_BAD_SIGNATURE where the reference unwraps or asserts. */
int et_attestations_recover(const et_attestation *atts, size_t n, uint8_t *attesters,
                            uint8_t *public_keys);
/* The same on the GPU, one lane per record (Poseidon, the curve arithmetic and Keccak in one kernel): for set
 * building at scale.  Results are identical to et_attestations_recover.  device < 0: the current device. */
int et_attestations_recover_gpu(const et_attestation *atts, size_t n, uint8_t *attesters,
                                uint8_t *public_keys, int device);
/* Self-test hook: the lane function of that kernel compiled for the host, on one record, so that the device
 * arithmetic can be checked where no GPU is present.  Not a compute path. */
int et_attestation_recover_lane_host(const et_attestation *att, uint8_t attester[20],
                                     uint8_t public_key[64]);
/* secret * G and its Ethereum address (ecdsa/native.rs:254-262, :90-111); for tests and tools. */
int et_secp256k1_public_key(const uint8_t secret_le[32], uint8_t public_key[64], uint8_t address[20]);
/* EcdsaKeypair::sign with the caller's nonce (ecdsa/native.rs:264-296): r, s little-endian, rec_id = parity of
 * the nonce point's y.  For tests and tools that need signed records. */
int et_secp256k1_sign(const uint8_t secret_le[32], const et_fr *msg_hash, const uint8_t nonce_le[32],
                      uint8_t sig_r[32], uint8_t sig_s[32], uint8_t *rec_id);
/* Opinion::validate (opinion/native.rs:63-109) for one attester: atts has num_neighbours records,
 * present[i] == 0 (or a NULL `present` = all present) marks an entry the reference fills with
 * empty_with_about (dynamic_sets/native.rs:209-219); set_addresses = num_neighbours * 20 bytes,
 * zero = empty slot.  scores[i] = value when the signature verifies, the slot is occupied and
 * the key is not the default key, else 0; op_hash = sponge of the per-entry hashes.
 * ET_E_NOT_MEMBER (:67), ET_E_BAD_ABOUT (:75), ET_E_BAD_DOMAIN (:76). */
int et_opinion_validate(const uint8_t public_key[64], const et_attestation *atts,
                        const uint8_t *present, const uint8_t *set_addresses, uint32_t num_neighbours,
                        const uint8_t domain[20], et_fr *scores, et_fr *op_hash);
/* et_circuit_setup without the circuit (eigentrust/src/lib.rs:339-466): recovers the attesters,
 * builds the participant set and the attestation matrix, validates every opinion, runs converge
 * (GPU) and converge_rational (host), packs the scores and, when public_inputs is not NULL, the
 * ETPublicInputs bytes ((2*num_neighbours + 2) * 32).  out may be NULL to query *n_participants. */
int et_et_setup(const et_attestation *atts, size_t n_att, const uint8_t domain[20],
                uint32_t num_neighbours, uint32_t num_iterations, uint64_t initial_lo,
                uint64_t initial_hi, et_score *out, size_t cap, size_t *n_participants,
                uint8_t *public_inputs, size_t public_inputs_cap);

/* calculate_scores for attestation lists of any size: the attesters are recovered on the GPU
 * (et_attestations_recover_gpu), the participants are all attesters and all `about` addresses in
 * ascending order (every one of them a member: no slot limit), one edge per record (a later record of
 * the same pair replaces an earlier one) goes through et_graph_build, and converge runs on the GPU.
 * Outputs (each may be NULL): addresses = participants * 20 bytes, scores_fr / scores_f64 in that
 * order.  All NULL: only *n_participants is set.  No rational scores (no arbitrary precision at
 * this size).  Errors: ET_E_BAD_DOMAIN, ET_E_BAD_SIGNATURE, ET_E_INSUFFICIENT_PEERS, and those of
 * et_graph_build / et_graph_converge_*. */
int et_scores_from_attestations(const et_attestation *atts, size_t n_att, const uint8_t domain[20],
                                uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi,
                                uint8_t *addresses, et_fr *scores_fr, double *scores_f64, size_t cap,
                                size_t *n_participants);

/* Field arithmetic self-test hook: runs the device multiply-accumulate + reduction used by the
 * Fr kernel on `count` (a, b) pairs and returns a[i]*b[i] mod r (canonical in, canonical out). */
int et_dev_fr_mul(uint32_t count, const et_fr *a, const et_fr *b, et_fr *out);

#ifdef __cplusplus
}
#endif
#endif /* EIGENTRUST_HIP_H */
