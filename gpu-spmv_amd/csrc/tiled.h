// tiled.h — the LDS-tiled SpMV engine ("x staged into LDS tiles"): a two-phase,
// propagation-blocking execution of y = A x for matrices whose x does not fit on chip.
// See tiled.hip for the algorithm; this is the host-side plan object.
#ifndef SPMV_AMD_TILED_H
#define SPMV_AMD_TILED_H

#include "internal.h"

#include <cstdint>
#include <mutex>
#include <vector>

namespace spmv {
namespace detail {

struct PrState;
struct PushTargets;
struct RowMap;

// Per-matrix bucketed copy of the entries (built once on the device, cached in the
// side table, dropped by csr_free_gpu).
struct TiledPlan {
    int num_rows = 0, num_cols = 0;
    long long nnz = 0;              // SLOTS held in cells: the short rows' entries + row-skip markers + padding
    long long entries = 0;          // of which real matrix entries
    int strip_cols = 0;             // W: x columns per LDS strip
    int tile_rows = 0;              // R: y rows per LDS tile
    int num_strips = 0, num_tiles = 0;

    // slots sorted by cell (strip-major, tile inside a strip); inside a cell by (row, column).
    // Every cell's length is a multiple of 4 slots.
    float*    a_val = nullptr;      // [nnz]; null when the values are folded into col_weight
    float*    col_weight = nullptr; // [num_cols] the one value every entry of a column carries, or null
    uint16_t* a_lcol = nullptr;     // [nnz] column - strip * W
    uint8_t*  a_drow = nullptr;     // [nnz] row - (row of the cell's previous slot), 0..254; 255 = advance 255 rows, no entry
    float*    prod = nullptr;       // [nnz] phase-1 output / phase-2 input, same order
    int*      cells_t = nullptr;    // [2 * num_tiles * num_strips]: (begin, length) pairs, tile-major
    // phase 2's work, laid out at build time (tiled.hip, pass_layout_kernel): the slots of wavefront w of tile t's workgroup
    // are the passes [pass_first[16 t + w], pass_first[16 t + w + 1]), one 32-byte descriptor each
    int*      pass_first = nullptr; // [16 * num_tiles + 1]
    void*     pass_desc = nullptr;  // [num_passes] PassDesc
    long long num_passes = 0;

    // phase-1 work items: (strip, begin, end), at most kItemEntries slots each
    int* items = nullptr;           // [3 * num_items]
    int  num_items = 0;
    int* strip_first_item = nullptr; // HOST [num_strips + 1]: the items are sorted by strip
    // rows longer than long_row: summed by one wavefront per 512-entry chunk from the CSR arrays
    int*   long_rows = nullptr;     // [num_long] ascending
    int    num_long = 0;
    int    long_row = 1024;         // rows with more entries than this are "long"
    int*   long_chunks = nullptr;   // [3 * num_long_chunks] (row, begin, end) over the CSR arrays
    int    num_long_chunks = 0;
    int*   long_first = nullptr;    // [num_long + 1] first chunk of every long row
    float* long_sums = nullptr;     // [num_long_chunks] per-chunk partial sums of the current SpMV
    int*   tile_long = nullptr;     // [num_tiles + 1] first long row of every tile (index into long_rows)
    const int*   csr_row_ptrs = nullptr;   // borrowed from the matrix
    const int*   csr_cols = nullptr;
    const float* csr_vals = nullptr;
    long long    csr_nnz = 0;

    // Scratch per stream.  `prod` / `long_sums` above are written by every call, so they belong to ONE stream: the
    // first that runs the plan.  A call on another stream gets its own pair (allocated on first use, kept with
    // the plan), so asynchronous calls on one matrix from different streams do not share a product stream — the
    // reference's kernels are stateless and its callers may rely on that.
    struct StreamScratch {
        hipStream_t stream;
        float* prod;
        float* long_sums;
    };
    mutable std::mutex scratch_lock;
    mutable std::mutex launch_lock;     // held across the two launches of one SpMV
    mutable bool primary_taken = false;
    mutable hipStream_t primary_stream = nullptr;
    mutable std::vector<StreamScratch> extra_scratch;

    // what the build cost (reported by bench.py)
    double    build_ms = 0.0;       // host wall clock of the build, allocations and syncs included
    long long plan_bytes = 0;       // device memory the plan holds
};

// the (strip columns, tile rows) the engine would pick for a matrix of this shape, and whether it
// would take the matrix at all (pure host logic; exposed for tests and reports)
bool tiled_shape_for(long long rows, long long cols, long long nnz, int* strip_cols, int* tile_rows);

// true when the matrix is worth (and able) to run through the tiled engine
bool tiled_eligible(const CSRMatrix* A);
bool tiled_eligible(const ELLMatrix* A);

// builds the plan for A's device arrays (synchronises the stream a few times)
hipError_t tiled_build(const CSRMatrix* A, TiledPlan** out, hipStream_t s);
hipError_t tiled_build(const ELLMatrix* A, TiledPlan** out, hipStream_t s);   // from the ELL slabs (no long-row path)
void tiled_free(TiledPlan* plan);

// position-weighted checksums of a_val / a_lcol / a_drow / cells_t (test aid: equal plans, equal numbers)
hipError_t tiled_checksum(const TiledPlan& plan, unsigned long long out[4], hipStream_t s);

// y = A x.  hipErrorOutOfMemory: no scratch for this (additional) stream — the caller may use another kernel.
hipError_t tiled_spmv(const TiledPlan& plan, const float* d_x, float* d_y, hipStream_t s);

// PageRank step on the same plan, in two parts.
// Phase 1 for the strips [strip_begin, strip_end) — products of the entries whose columns lie there — and,
// with_long, the long rows (which read all of r_old).  A step needs every strip and the long rows exactly
// once, in any number of calls, before its finish.  No-op when state->done.
hipError_t tiled_pagerank_expand(const TiledPlan& plan, int strip_begin, int strip_end, bool with_long,
                                 const float* d_r_old, const PrState* d_state, hipStream_t s);
// Phase 2: r_new[map.at(i)] = d * (A r_old)_i + d*s/n + (1-d)/n, block partial sums of (r_new - r_old)^2
// and of r_new over dangling nodes -> block_partials [2 * plan.num_tiles]; no-op when state->done.
hipError_t tiled_pagerank_finish(const TiledPlan& plan, const RowMap& map, int n_global,
                                 const float* d_r_old, float* d_r_new,
                                 const unsigned char* d_dangling, float damping,
                                 const PrState* d_state, double* d_block_partials,
                                 const PushTargets& push, hipStream_t s);

} // namespace detail
} // namespace spmv

#endif
