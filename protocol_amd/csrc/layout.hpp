// Device data layout of one prepared trust graph (shared by the host builder and the kernels).
//
// The local-trust matrix C (c_ij = filtered opinion of peer i about peer j divided by i's row sum,
// dynamic_sets/native.rs:305-314) is stored transposed: one row per DESTINATION peer holding its
// in-edges, because the iteration new_s[i] = sum_j c_ji * s[j] (native.rs:319-329) is then a
// row-wise dot product with no write conflicts.
//
// Rows are grouped in slices of 64 (one wavefront lane per row).  A slice stores its edges
// step-major: entry (step k, lane l) lives at (off + k) * 64 + l, so every load of the edge
// stream is one contiguous 64-lane access.  A row with many in-edges gets a slice of its own whose
// 64 lanes each hold every 64th edge of that one row and are summed across the wavefront.
//
// Trust vectors live in "position" order: G segments (one per GPU that owns rows; G = 1 on a single
// GPU), each `rows_cap` row entries followed by `nd_cap` entries holding that GPU's partial sums of
// the score mass of dangling peers (peers whose filtered row is empty, native.rs:263-278).  An
// in-place all-gather of the segments therefore carries vector and partial sums in one message.
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
    uint32_t nrows;  // bit 31: all lanes belong to row0 (wavefront-reduced row); low bits: rows (1..64)
};
static const uint32_t SLICE_REDUCE = 0x80000000u;

struct BlockDesc {
    uint32_t slice_begin, slice_end;  // slices of this workgroup; wave w takes begin+w, begin+w+4, ...
    uint32_t flags;                   // bit 0: rows are dangling peers
    uint32_t dslot;                   // slot in the segment's partial-sum tail (flag bit 0 only)
};
static const uint32_t BLOCK_DANGLING = 1u;

static const int WAVES_PER_BLOCK = 4;
static const int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;

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
    ET_HD uint32_t seg_stride() const { return rows_cap + nd_cap; }
};

}  // namespace et
