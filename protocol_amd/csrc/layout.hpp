// Device data layout of one prepared trust graph (shared by the host builder and the kernels).
//
// The local-trust matrix C (c_ij = filtered opinion of peer i about peer j divided by i's row sum,
// dynamic_sets/native.rs:305-314) is stored transposed: one row per DESTINATION peer holding its
// in-edges, because the iteration new_s[i] = sum_j c_ji * s[j] (native.rs:319-329) is then a
// row-wise dot product with no write conflicts.
//
// Two arrangements of the same edges are kept.
// Exact variant: rows are grouped in slices, one wavefront each, 2^s lanes per row (s = 0 for large
// graphs: 64 rows per slice).  A slice stores its edges step-major: entry (step k, lane l) lives at
// (off + k) * 64 + l, so every load of the edge stream is one contiguous 64-lane access.  A row with
// many in-edges gets a slice of its own (s = 6) whose 64 lanes each hold every 64th edge of that
// one row and are summed across the wavefront.
// f64 variant: one edge list per workgroup, sorted by source, for rows whose accumulators the
// workgroup keeps in LDS ("blocked sweep", see engine.hip).  One share of 512K to 2M rows of byte
// opinions keeps such lists for the exact variant too (eight 64-bit column sums per row in LDS).
//
// The peers gathered most often (highest out-degree) are placed first: the busiest part of the
// trust vector is one contiguous prefix.
//
// Trust vectors live in "position" order: G segments (one per GPU that owns rows; G = 1 on a single
// GPU), each `rows_cap` row entries followed by `nd_cap` entries holding that GPU's partial sums of
// the score mass of dangling peers (peers whose filtered row is empty, native.rs:263-278).  An
// in-place all-gather of the segments therefore carries vector and partial sums in one message.
//
// Several GPUs, integer opinions: a share does not keep the other shares' whole segments.  Its vectors
// hold its own segment first and then, per other share p, only the entries of p's rows that its own
// rows cite (in the order it asked for) followed by p's partial sums: the "gather vector".  The exchange
// sends every share exactly the entries it cites, and the gathered vector shrinks from all peers to
// the cited ones.  A source index in an edge record is then the entry's index in the gather vector.
#pragma once
#include <cstdint>

#ifdef __HIPCC__
#define ET_HD __host__ __device__
#else
#define ET_HD
#endif

namespace et {

struct SliceDesc {
    uint32_t off;    // first step of the slice in the edge stream (units of 64 entries)
    uint32_t len;    // number of steps
    uint32_t row0;   // local row of lane 0
    uint32_t nrows;  // bits 0..7: rows in the slice (1..64 >> s); bits 8..11: s = log2(lanes per row)
};
// A row of a slice is spread over 2^s adjacent lanes (lane l serves row l >> s and takes every 2^s-th
// edge of it, starting with edge l & (2^s - 1)); the lanes' sums are added across the group.  s = 0 is
// one lane per row, s = 6 one wavefront per row (rows with very many in-edges).  Small graphs use
// s > 0 for all rows so that there are enough wavefronts to fill the GPU.
ET_HD inline uint32_t slice_rows(uint32_t nrows) { return nrows & 0xFFu; }
ET_HD inline uint32_t slice_log2_lanes(uint32_t nrows) { return (nrows >> 8) & 0xFu; }

// Work of ONE wavefront: a contiguous run of slices of one class (active or dangling rows).
struct WorkDesc {
    uint32_t slice_begin, slice_end;
    uint32_t flags;                   // bit 0: rows are dangling peers
    uint32_t dslot;                   // slot in the segment's partial-sum tail (flag bit 0 only)
};
static const uint32_t WORK_DANGLING = 1u;

// Rows are handed to the workgroups of a blocked sweep in groups of BLOCK_GROUP consecutive rows (position
// order is degree order, so the heaviest rows sit together: small groups keep the heaviest group well
// below a workgroup's share of the edges).
static const uint32_t BLOCK_GROUP = 16;
// workgroups of a blocked sweep that may hold dangling rows: the first 256, or as many as it takes for the
// dangling groups to fill at most half of their accumulator slots
inline uint32_t dangling_blocks(uint32_t blocks, uint32_t groups_per_block, uint32_t dangling_groups) {
    uint32_t need = (uint32_t)(((uint64_t)dangling_groups * 2 + groups_per_block - 1) / groups_per_block);
    uint32_t d = need > 256 ? need : 256;
    return d < blocks ? d : blocks;
}
static const uint32_t FR_RB_MAX_COMPACT = 4400, FR_RB_MAX_WIDE = 2432;
static const uint32_t FR_RB_MAX_U32 = 4240;   // 32-bit opinions: 38 bytes of LDS per row
static const uint32_t FR_U32_MAX_IN_DEGREE = 1u << 17;   // ... which hold a row's sum while it has fewer in-edges than this
// A source index in an edge record: segment in the top 3 bits, local row below.
static const uint32_t COL_SEG_SHIFT = 29;
static const uint32_t COL_LOCAL_MASK = (1u << COL_SEG_SHIFT) - 1;

// Geometry of the position space, identical on every GPU of a sharded run.
struct Geometry {
    uint32_t n_seg;        // G
    uint32_t seg;          // this GPU's segment
    uint32_t rows_cap;     // row entries per segment (multiple of 64)
    uint32_t nd_cap;       // partial-sum entries per segment
    uint32_t n_active;     // peers with a non-empty filtered row (global count)
    uint32_t n_active_pad; // rounded up to 64 * G
    uint32_t n_dangling;   // member peers whose filtered row is empty
    uint32_t n_dangling_pad;
    // f64 blocked sweep: the segment's BLOCK_GROUP-row groups are handed to f64_blocks workgroups, each
    // of which keeps up to f64_rb row accumulators in LDS and walks its edges in ascending source order.
    uint32_t f64_blocks, f64_rb;
    // groups of dangling rows go to the first f64_dblocks workgroups only; workgroup b of those writes
    // partial-sum slot b of its segment, so nd_cap >= f64_dblocks
    uint32_t f64_dblocks;
    // exact blocked sweep (integer opinions, 16K rows per share and more; 0 workgroups = not used): the same deal with a
    // row's unreduced sum in LDS.  fr_compact == 1: byte opinions, nine 32-bit limbs per row (36 bytes, carries propagated
    // as the edges arrive), up to FR_RB_MAX_COMPACT rows per workgroup; 2: 32-bit opinions, nine limbs and sixteen bits
    // (38 bytes), up to FR_RB_MAX_U32 rows, graphs whose rows have fewer than FR_U32_MAX_IN_DEGREE in-edges; 0 (a measurement knob, byte opinions): eight 64-bit column sums (64 bytes, no carries until the
    // row is folded), up to FR_RB_MAX_WIDE rows.
    uint32_t fr_blocks, fr_rb, fr_dblocks, fr_compact;
    // Chunked exchange (several GPUs): a segment travels in n_chunks pieces of chunk_rows rows (the last piece takes
    // the rest and the partial-sum tail), and an iteration runs as n_chunks launches, launch q over the edges whose
    // source lies in piece q of ANY segment; a workgroup's edge list is sorted by (piece, position).  1 = off.
    uint32_t n_chunks, chunk_rows;
    // Where this share's own segment and every share's partial sums sit in the vectors the steps read and write.
    // Position order (compact == 0): own_off = seg * seg_stride(), dtail[p] = p * seg_stride() + rows_cap,
    // vec_len = n_seg * seg_stride().  Gather vector (compact == 1): the share's own block first (own_off = 0: its rows'
    // entries in row order, then its partial sums at dtail[seg]), then one block per share p -- the share's own first, the
    // others in ascending order: the cited entries of p's rows in the order this share's sweeps want them, followed (p != seg)
    // by p's partial sums at dtail[p].  The sweeps gather from these blocks only; the own block is what a step writes,
    // what its rows' old entries are read from, and what is packed for the others and for the share's own block of
    // cited entries.  vec_len = entries of the gather vector.
    uint32_t compact, own_off, vec_len;
    uint32_t dtail[8];
    ET_HD uint32_t seg_stride() const { return rows_cap + nd_cap; }
    // this share's view of the layout; with a gather vector the blocks of cited entries follow once the cited rows are
    // known (set_gather_blocks)
    ET_HD void set_share(uint32_t s) {
        seg = s;
        for (uint32_t p = 0; p < 8; ++p) dtail[p] = 0u;
        if (compact) {
            own_off = 0;
            vec_len = seg_stride();
            dtail[s] = rows_cap;
            return;
        }
        own_off = s * seg_stride();
        vec_len = n_seg * seg_stride();
        for (uint32_t p = 0; p < n_seg; ++p) dtail[p] = p * seg_stride() + rows_cap;
    }
    // gather vector: block p = cited[p] entries of share p's rows, then (p != seg) its nd_cap partial sums; the share's
    // own block of cited entries comes first (what is received from the others is then one contiguous range, in share
    // order); blk_off[p] receives the block's first entry
    ET_HD void set_gather_blocks(const uint32_t* cited, uint32_t* blk_off) {
        own_off = 0;
        for (uint32_t p = 0; p < 8; ++p) { dtail[p] = 0u; blk_off[p] = 0u; }
        dtail[seg] = rows_cap;
        blk_off[seg] = seg_stride();
        uint32_t at = seg_stride() + cited[seg];
        for (uint32_t p = 0; p < n_seg; ++p) {
            if (p == seg) continue;
            blk_off[p] = at;
            dtail[p] = at + cited[p];
            at = dtail[p] + nd_cap;
        }
        vec_len = at;
    }
    ET_HD uint32_t chunk_of_local(uint32_t local) const {
        const uint32_t q = local / chunk_rows;
        return q < n_chunks ? q : n_chunks - 1;
    }
    // piece q of a segment: entries [chunk_begin(q), chunk_begin(q + 1)) of its rows_cap + nd_cap
    ET_HD uint32_t chunk_begin(uint32_t q) const { return q >= n_chunks ? seg_stride() : q * chunk_rows; }
    ET_HD uint32_t pos_of_col(uint32_t c) const { return (c >> COL_SEG_SHIFT) * seg_stride() + (c & COL_LOCAL_MASK); }
};

}  // namespace et
