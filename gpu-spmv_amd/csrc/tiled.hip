// tiled.hip — the LDS-tiled SpMV engine for matrices whose x does not fit on chip
// (the MI355X replacement of the reference's texture-cache read of x,
// src/spmv_kernels.cu:7-39; selected by SpMVConfig::use_texture).
//
// Why: on gfx950 a 4-byte gather through the vector-memory path costs one 64-byte
// fabric request and runs at <= 0.3 lane/clk/CU even from L1 (tools/gather_bench.hip:
// 295 / 185 / 61 G gathers/s from L1 / L2 / Infinity Cache), while an LDS gather runs
// at ~7 lanes/clk/CU (tools/lds_bench.hip).  So x must be gathered from LDS — but with
// e.g. 10 M columns and 16 entries per row no (row block x column strip) tile is dense
// enough to amortise loading its strip.  The engine therefore runs y = A x in two
// streaming phases over a bucketed copy of the entries (propagation blocking):
//
//   layout : entries sorted by cell = (column strip, row tile), strip-major; per entry
//        value f32, local column u16, local row u16 (8 B, as CSR's 8 B) + a product slot.
//   phase 1 "expand" : a workgroup loads one x strip (W = 4 K .. 32 K columns, chosen per
//        matrix = 16 .. 128 KiB) into LDS, streams its share of the strip's entries (value, local column) with
//        16-byte loads, gathers x from LDS and stores the products — same order, so
//        loads and stores are all contiguous.
//   phase 2 "reduce" : a workgroup owns one row tile (R rows in dynamic LDS, R a multiple of 64
//        up to 9984 = 39 KiB, stretched so that the tiles fill whole rounds of resident workgroups).
//        The tile's entries are one contiguous run per strip (cell table); the waves
//        walk the runs, add each product into the LDS tile and finally write the tile
//        out with coalesced stores (optionally through the fused PageRank update).
//        gfx950's ds_add_f32 is ~30x slower than its integer LDS atomics (0.38 vs 11.7
//        lanes/clk/CU measured), so the add is a compare-and-swap on the word's integer
//        image (3.5 lanes/clk/CU measured; race-free for any row multiplicity).
//   folding : when every stored entry of a column has the same bits, the value stream is dropped
//        and phase 1 gathers w_j * x_j from LDS (column_weight_probe_kernel, FOLD instantiation).
//   both phases walk their work lists in per-XCD contiguous slices (xcd_contiguous).
//   long rows (more than min(2048, 2 or 4 entries per strip)) would make many lanes fight over one
//        LDS word; they are left out of the cells and summed in 512-entry chunks by extra
//        wavefronts of the phase-1 grid (direct gather) into a side vector that seeds the tiles.
//
// HBM traffic per entry: 6 B read + 4 B written in phase 1, 6 B read in phase 2
// (16 B vs CSR's 8 B) — but all of it is streamed, which beats one 64-byte random
// fetch per entry by a wide margin once x leaves L2.
// The order in which a row's products are added depends on scheduling, so the low
// bits of y may differ from run to run (as with any atomic accumulation; the
// reference's merge-path kernel has the same property).
#include "tiled.h"
#include "device_common.h"
#include "pagerank_engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace spmv {
namespace detail {

namespace {

using namespace dev;

// W (x columns per LDS strip) and R (y rows per LDS tile) are chosen per matrix
// (choose_shape below):
//   W in {4096, 8192, 16384, 32768} = 16 .. 128 KiB of static LDS in phase 1 (template instantiations)
//   R = any multiple of 64 in [64, kMaxTileRows]: dynamic LDS in phase 2
constexpr int kMaxItemEntries = 65536;   // phase-1 work item size bounds (entries)
constexpr int kMinItemEntries = 4096;
constexpr int kMaxLongRow = 2048;     // rows longer than min(this, 2 or 4 entries per strip) bypass the cells
constexpr int kLongChunk = 512;       // entries per wavefront of the long-row path
constexpr long long kMaxCells = 1LL << 26;
constexpr long long kTargetRun = 128;        // wanted mean entries per cell (run length seen by phase 2)
constexpr long long kResidentTiles = 1024;   // phase-2 workgroups resident at once: 256 CUs x 4 (32 wavefronts / 8)
constexpr int kMaxTileRows = 9984;           // 4 tiles of this height (+ the reduction scratch) fit one CU's 160 KiB

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ plan building ----
// LANES lanes walk one row; every entry of a short row is assigned to cell (strip, tile).
// PASS 0 counts (and lists the long rows), PASS 1 scatters.
template <int LANES, int PASS>
__global__ __launch_bounds__(kBlock)
void bucket_kernel(int num_rows, int num_tiles, int strip_cols, int tile_rows, int long_row,
                   const int* __restrict__ row_ptrs, const int* __restrict__ cols,
                   const float* __restrict__ vals,
                   int* __restrict__ cell_counter,            // [num_strips * num_tiles]
                   const int* __restrict__ offs,              // strip-major exclusive scan
                   float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                   unsigned short* __restrict__ a_lrow,
                   int* __restrict__ long_rows, int* __restrict__ num_long) {
    constexpr int kRowsPerBlock = kBlock / LANES;
    const int lane = threadIdx.x % LANES;
    const long long row = static_cast<long long>(blockIdx.x) * kRowsPerBlock + threadIdx.x / LANES;
    if (row >= num_rows) return;
    const int begin = row_ptrs[row], end = row_ptrs[row + 1];
    if (end - begin > long_row) {
        if (PASS == 0 && lane == 0) long_rows[atomicAdd(num_long, 1)] = static_cast<int>(row);
        return;
    }
    const int tile = static_cast<int>(row / tile_rows);
    const unsigned short lrow = static_cast<unsigned short>(row % tile_rows);
    for (int j = begin + lane; j < end; j += LANES) {
        const int c = cols[j];
        const int strip = c / strip_cols;
        const long long cell = static_cast<long long>(strip) * num_tiles + tile;
        if (PASS == 0) {
            atomicAdd(&cell_counter[cell], 1);
        } else {
            const int at = offs[cell] + atomicAdd(&cell_counter[cell], 1);
            if (a_val) a_val[at] = vals[j];
            a_lcol[at] = static_cast<unsigned short>(c - strip * strip_cols);
            a_lrow[at] = lrow;
        }
    }
}

// The same two passes with the cell counters aggregated in LDS: a workgroup takes a block of
// rows inside ONE tile, so its entries differ only in the strip; it counts them per strip in LDS
// and touches each global cell counter once (pass 0: add the count; pass 1: reserve a range and
// hand out its slots from an LDS cursor).  Global atomics are device-scope round trips (~21 G/s
// chip-wide, 160 M of them = 7.7 ms on C5); this way there are num_strips per workgroup instead
// of one per entry.  Dynamic LDS: num_strips ints in pass 0, twice that in pass 1.
template <int LANES, int PASS>
__global__ __launch_bounds__(kBlock)
void bucket_lds_kernel(int num_rows, int num_tiles, int num_strips, int strip_cols, int tile_rows, int long_row,
                       int rows_per_block, int blocks_per_tile,
                       const int* __restrict__ row_ptrs, const int* __restrict__ cols,
                       const float* __restrict__ vals,
                       int* __restrict__ cell_counter, const int* __restrict__ offs,
                       float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                       unsigned short* __restrict__ a_lrow,
                       int* __restrict__ long_rows, int* __restrict__ num_long) {
    extern __shared__ int bucket_lds[];
    int* hist = bucket_lds;
    int* cursor = bucket_lds + num_strips;                 // pass 1 only
    const int tile = blockIdx.x / blocks_per_tile;
    const long long tile_first = static_cast<long long>(tile) * tile_rows;
    const long long row0 = tile_first + static_cast<long long>(blockIdx.x % blocks_per_tile) * rows_per_block;
    const long long row1 = min(min(row0 + rows_per_block, tile_first + tile_rows), static_cast<long long>(num_rows));
    for (int i = threadIdx.x; i < num_strips; i += kBlock) hist[i] = 0;
    __syncthreads();

    constexpr int kRowsPerSweep = kBlock / LANES;
    const int lane = threadIdx.x % LANES;
    for (long long row = row0 + threadIdx.x / LANES; row < row1; row += kRowsPerSweep) {
        const int begin = row_ptrs[row], end = row_ptrs[row + 1];
        if (end - begin > long_row) {
            if (PASS == 0 && lane == 0) long_rows[atomicAdd(num_long, 1)] = static_cast<int>(row);
            continue;
        }
        for (int j = begin + lane; j < end; j += LANES) atomicAdd(&hist[cols[j] / strip_cols], 1);
    }
    __syncthreads();
    if (PASS == 0) {
        for (int i = threadIdx.x; i < num_strips; i += kBlock) {
            if (hist[i]) atomicAdd(&cell_counter[static_cast<long long>(i) * num_tiles + tile], hist[i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < num_strips; i += kBlock) {
        const long long cell = static_cast<long long>(i) * num_tiles + tile;
        cursor[i] = hist[i] ? offs[cell] + atomicAdd(&cell_counter[cell], hist[i]) : 0;
    }
    __syncthreads();
    for (long long row = row0 + threadIdx.x / LANES; row < row1; row += kRowsPerSweep) {
        const int begin = row_ptrs[row], end = row_ptrs[row + 1];
        if (end - begin > long_row) continue;
        const unsigned short lrow = static_cast<unsigned short>(row - tile_first);
        for (int j = begin + lane; j < end; j += LANES) {
            const int c = cols[j];
            const int strip = c / strip_cols;
            const int at = atomicAdd(&cursor[strip], 1);
            if (a_val) a_val[at] = vals[j];
            a_lcol[at] = static_cast<unsigned short>(c - strip * strip_cols);
            a_lrow[at] = lrow;
        }
    }
}

// ELL source: one thread per row walks the K column-major slabs (padding: col < 0).
template <int PASS>
__global__ __launch_bounds__(kBlock)
void bucket_ell_kernel(int num_rows, int width, int num_tiles, int strip_cols, int tile_rows,
                       const int* __restrict__ cols, const float* __restrict__ vals,
                       int* __restrict__ cell_counter, const int* __restrict__ offs,
                       float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                       unsigned short* __restrict__ a_lrow) {
    const long long row = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
    if (row >= num_rows) return;
    const int tile = static_cast<int>(row / tile_rows);
    const unsigned short lrow = static_cast<unsigned short>(row % tile_rows);
    for (int k = 0; k < width; ++k) {
        const long long slot = static_cast<long long>(k) * num_rows + row;
        const int c = cols[slot];
        if (c < 0) continue;
        const int strip = c / strip_cols;
        const long long cell = static_cast<long long>(strip) * num_tiles + tile;
        if (PASS == 0) {
            atomicAdd(&cell_counter[cell], 1);
        } else {
            const int at = offs[cell] + atomicAdd(&cell_counter[cell], 1);
            if (a_val) a_val[at] = vals[slot];
            a_lcol[at] = static_cast<unsigned short>(c - strip * strip_cols);
            a_lrow[at] = lrow;
        }
    }
}

// out[i] = sum of in[0..i), out[n] = total.  One workgroup of 1024; each thread owns a
// contiguous chunk (one-time cost, n <= 2^26).
__global__ __launch_bounds__(1024)
void exclusive_scan_kernel(const int* __restrict__ in, long long n, int* __restrict__ out) {
    __shared__ long long s_part[1024];
    const long long chunk = (n + 1023) / 1024;
    const long long lo = min(n, chunk * threadIdx.x);
    const long long hi = min(n, lo + chunk);
    long long sum = 0;
    for (long long i = lo; i < hi; ++i) sum += in[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele over the partials
        const long long add = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (long long i = lo; i < hi; ++i) {
        out[i] = static_cast<int>(run);
        run += in[i];
    }
    if (threadIdx.x == 1023) out[n] = static_cast<int>(s_part[1023]);
}

// Column-weight folding: when every stored entry of a column carries the same value
// (adjacency matrices, the column-stochastic matrices of PageRank: a_ij = 1 / outdeg(j)),
// a_ij * x_j = (w_j * x_j) is one product per column instead of one per entry, and phase 1 no
// longer needs the value stream.  PASS 0 records a value per column, PASS 1 compares every entry
// with it bit for bit; the plan folds only if none differs.  Padding (col < 0) is skipped.
template <int PASS>
__global__ __launch_bounds__(kBlock)
void column_weight_probe_kernel(const int* __restrict__ cols, const float* __restrict__ vals, long long count,
                                float* __restrict__ weight, int* __restrict__ differs) {
    bool bad = false;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < count;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        const int c = cols[i];
        if (c < 0) continue;
        if (PASS == 0) {
            weight[c] = vals[i];
        } else {
            bad |= __float_as_uint(weight[c]) != __float_as_uint(vals[i]);
        }
    }
    if (PASS == 1 && bad) *differs = 1;
}

// cells_t[tile * num_strips + strip] = (begin, length) of the cell's run;
// strip_begin[s] = first entry of strip s (s <= num_strips)
__global__ __launch_bounds__(kBlock)
void cell_table_kernel(const int* __restrict__ offs, int num_strips, int num_tiles,
                       int2* __restrict__ cells_t, int* __restrict__ strip_begin) {
    const long long cells = static_cast<long long>(num_strips) * num_tiles;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i <= cells;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        if (i < cells) {
            const long long tile = i / num_strips, strip = i % num_strips;
            const long long cell = strip * num_tiles + tile;
            cells_t[i] = make_int2(offs[cell], offs[cell + 1] - offs[cell]);
        }
        if (i <= num_strips) strip_begin[i] = offs[i * num_tiles];
    }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its
// own L2).  Both phases hand every XCD a CONTIGUOUS range of the work list, walked in order:
// neighbours in the list then run on one XCD at about the same time and share what they both
// touch through its L2 — the x strip of consecutive phase-1 items, the 128-byte lines that
// adjacent runs of neighbouring tiles straddle in phase 2.  Returns -1 for the padding blocks of a
// grid rounded up to a multiple of 8.  (Speed only: correctness never depends on placement.
// Measured against the plain order on one box: C2 59.1 -> 55.0 us, C5 535.5 -> 530.1 us, 1/8 shard 84.5 -> 85.2 us.)
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_contiguous(int block, int count) {
    const int per_xcd = (count + kXcds - 1) / kXcds;
    const int which = (block % kXcds) * per_xcd + block / kXcds;
    return block / kXcds < per_xcd && which < count ? which : -1;
}
__host__ inline int xcd_grid(int count) { return (count + kXcds - 1) / kXcds * kXcds; }

// ------------------------------------------------------------------------ phase 1 ----
// Rows too long for the cells are cut into chunks of kLongChunk entries; one wavefront per
// chunk sums it by direct gather and adds the sum atomically into seed[row].  seed is zero on
// entry (zeroed at build; phase 2 re-zeroes what it consumes).  These wavefronts ride in extra
// workgroups at the head of the phase-1 grid, so they overlap the expansion at no launch cost.
struct LongRows {
    const int* chunks;        // (row, begin, end) triples over the CSR arrays
    int num_chunks;
    long long nnz;
    const int* cols;
    const float* vals;
    float* seed;
};

__device__ __forceinline__ void long_row_chunk(const LongRows& lr, int which, const float* __restrict__ x) {
    if (which >= lr.num_chunks) return;
    const int row = lr.chunks[3 * which];
    float acc = row_partial_dot<64>(lr.chunks[3 * which + 1], lr.chunks[3 * which + 2], threadIdx.x & 63, lr.nnz,
                                    lr.cols, lr.vals, x);
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&lr.seed[row], acc);
}

// FOLD: the plan holds one weight per column instead of a value per entry; the strip is staged
// as w_j * x_j and an entry's product is a plain LDS read (the same rounded product as a_ij * x_j).
template <int W, int kExpandBlock, bool FOLD>
__global__ __launch_bounds__(kExpandBlock)
void tiled_expand_kernel(const int* __restrict__ items, int num_items, int long_blocks,
                         const float* __restrict__ a_val,
                         const unsigned short* __restrict__ a_lcol,
                         const float* __restrict__ col_weight,
                         const float* __restrict__ x, int num_cols,
                         float* __restrict__ prod, LongRows long_rows,
                         const PrState* __restrict__ state) {
    // PageRank steps enqueued past convergence must leave the seed vector alone: phase 2 returns
    // before consuming it, and the long-row wavefronts ADD into it (a later call on the same
    // matrix would start from stale sums)
    if (state && state->done) return;
    if (static_cast<int>(blockIdx.x) < long_blocks) {     // the long-row workgroups go first (latency-bound)
        constexpr int kPerBlock = kExpandBlock / 64;
        long_row_chunk(long_rows, blockIdx.x * kPerBlock + (threadIdx.x >> 6), x);
        return;
    }
    __shared__ float xs[W];
    const int item = xcd_contiguous(blockIdx.x - long_blocks, num_items);     // long_blocks is a multiple of 8
    if (item < 0) return;
    const int strip = items[3 * item];
    const int begin = items[3 * item + 1];
    const int end = items[3 * item + 2];

    const long long base = static_cast<long long>(strip) * W;
    const int width = static_cast<int>(min(static_cast<long long>(W), num_cols - base));
    const float* src = x + base;
    if (FOLD) {
        const float* wsrc = col_weight + base;            // hipMalloc'd and base % 4 == 0: always aligned
        const bool aligned = (reinterpret_cast<unsigned long long>(src) & 15) == 0;
        for (int i = threadIdx.x * 4; i < width; i += kExpandBlock * 4) {
            if (aligned && i + 3 < width) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(src + i);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wsrc + i);
                f32x4 z;
                z[0] = __fmul_rn(wv[0], xv[0]);
                z[1] = __fmul_rn(wv[1], xv[1]);
                z[2] = __fmul_rn(wv[2], xv[2]);
                z[3] = __fmul_rn(wv[3], xv[3]);
                *reinterpret_cast<f32x4*>(xs + i) = z;
            } else {
                for (int k = i; k < min(i + 4, width); ++k) xs[k] = __fmul_rn(wsrc[k], src[k]);
            }
        }
    } else if ((reinterpret_cast<unsigned long long>(src) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < width; i += kExpandBlock * 4) {
            if (i + 3 < width) {
                *reinterpret_cast<f32x4*>(xs + i) = *reinterpret_cast<const f32x4*>(src + i);
            } else {
                for (int k = i; k < width; ++k) xs[k] = src[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < width; i += kExpandBlock) xs[i] = src[i];
    }
    __syncthreads();

    if (FOLD) {
        // eight entries per lane per step: one 16-byte index load, two 16-byte product stores
        for (int q = (begin & ~7) + threadIdx.x * 8; q < end; q += kExpandBlock * 8) {
            if (q >= begin && q + 7 < end) {
                const u16x8 c = *reinterpret_cast<const u16x8*>(a_lcol + q);
                f32x4 lo, hi;
                lo[0] = xs[c[0]]; lo[1] = xs[c[1]]; lo[2] = xs[c[2]]; lo[3] = xs[c[3]];
                hi[0] = xs[c[4]]; hi[1] = xs[c[5]]; hi[2] = xs[c[6]]; hi[3] = xs[c[7]];
                *reinterpret_cast<f32x4*>(prod + q) = lo;
                *reinterpret_cast<f32x4*>(prod + q + 4) = hi;
            } else {
                for (int k = max(q, begin); k < min(q + 8, end); ++k) prod[k] = xs[a_lcol[k]];
            }
        }
        return;
    }
    // four entries per lane per step, groups aligned to 4 entries (16-byte loads and stores)
    for (int q = (begin & ~3) + threadIdx.x * 4; q < end; q += kExpandBlock * 4) {
        if (q >= begin && q + 3 < end) {
            const u16x4 c = *reinterpret_cast<const u16x4*>(a_lcol + q);
            f32x4 p;
            if (FOLD) {
                p[0] = xs[c[0]];
                p[1] = xs[c[1]];
                p[2] = xs[c[2]];
                p[3] = xs[c[3]];
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a_val + q);
                p[0] = v[0] * xs[c[0]];
                p[1] = v[1] * xs[c[1]];
                p[2] = v[2] * xs[c[2]];
                p[3] = v[3] * xs[c[3]];
            }
            *reinterpret_cast<f32x4*>(prod + q) = p;
        } else {
            for (int k = max(q, begin); k < min(q + 4, end); ++k) {
                prod[k] = FOLD ? xs[a_lcol[k]] : a_val[k] * xs[a_lcol[k]];
            }
        }
    }
}

// ------------------------------------------------------------------------ phase 2 ----
// float add on an LDS word by compare-and-swap on its integer image
__device__ __forceinline__ void lds_add(float* slot, float v) {
    unsigned int* word = reinterpret_cast<unsigned int*>(slot);
    unsigned int seen = *word;
    for (;;) {
        const unsigned int want = __float_as_uint(__uint_as_float(seen) + v);
        const unsigned int got = atomicCAS(word, seen, want);
        if (got == seen) break;
        seen = got;
    }
}

// Fills the LDS tile with the sums of this tile's rows.  `seed` (may be null) holds the
// long rows' sums and zeros elsewhere.
template <int kReduceBlock, int U>
__device__ __forceinline__ void tile_accumulate(float* tile, int R, int tile_index, int num_strips, int num_rows,
                                                const int2* __restrict__ cells_t,
                                                const float* __restrict__ prod,
                                                const unsigned short* __restrict__ a_lrow,
                                                float* __restrict__ seed) {
    const long long first = static_cast<long long>(tile_index) * R;
    for (int i = threadIdx.x; i < R; i += kReduceBlock) {
        float v = 0.0f;
        if (seed && first + i < num_rows) {
            v = seed[first + i];
            if (v != 0.0f) seed[first + i] = 0.0f;      // leave the seed vector clean for the next call
        }
        tile[i] = v;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kWaves = kReduceBlock / 64;
    // A group of kRuns runs is processed together, U chunks of 64 entries from each per pass:
    // all 8 loads of a pass are issued before its first add, and a pass is repeated only while
    // some run of the group still has entries left (U is picked from the mean run length).
    constexpr int kSlots = 8;             // (run, chunk) loads in flight per lane (16 measured slower)
    constexpr int kRuns = kSlots / U;
    const int2* mine = cells_t + static_cast<long long>(tile_index) * num_strips;
    int2 meta_next = lane < num_strips ? mine[lane] : make_int2(0, 0);
    for (int s0 = 0; s0 < num_strips; s0 += 64) {
        // the tile's next 64 runs: every wavefront loads their (begin, length) (one coalesced
        // 512-byte load, L1-shared; fetched one batch ahead so its latency hides behind the
        // current batch) and takes every kWaves-th group of kRuns runs
        const int2 meta = meta_next;
        meta_next = s0 + 64 + lane < num_strips ? mine[s0 + 64 + lane] : make_int2(0, 0);
        const int runs = min(64, num_strips - s0);
        for (int k0 = wave * kRuns; k0 < runs; k0 += kWaves * kRuns) {
            int begin[kRuns], len[kRuns], lead[kRuns];
            int longest = 0;
#pragma unroll
            for (int j = 0; j < kRuns; ++j) {
                const int k = min(k0 + j, 63);
                begin[j] = __shfl(meta.x, k, 64);
                len[j] = k0 + j < runs ? __shfl(meta.y, k, 64) : 0;
                longest = max(longest, len[j]);
            }
            if (U >= 2) {
                // pairs of entries per lane (8-byte product / 4-byte row loads): runs re-based to an
                // even entry, the odd leading entry masked off
#pragma unroll
                for (int j = 0; j < kRuns; ++j) {
                    lead[j] = begin[j] & 1;
                    len[j] += lead[j];
                    begin[j] &= ~1;
                }
                longest += 1;
            }
            for (int done = 0; done < longest; done += 64 * U) {
                float p[kSlots];
                int r[kSlots];
                if (U >= 2) {
#pragma unroll
                    for (int j = 0; j < kRuns; ++j) {
#pragma unroll
                        for (int u = 0; u < U; u += 2) {
                            const int i = done + u * 64 + 2 * lane;
                            const bool ok = i < len[j];          // the pair is inside the allocation whenever its first entry is
                            f32x2 pv = {0.0f, 0.0f};
                            u16x2 rv = {0, 0};
                            if (ok) {
                                pv = *reinterpret_cast<const f32x2*>(prod + begin[j] + i);
                                rv = *reinterpret_cast<const u16x2*>(a_lrow + begin[j] + i);
                            }
                            const bool first = ok && i >= lead[j];
                            p[j * U + u] = pv[0];
                            r[j * U + u] = first ? rv[0] : -1;
                            p[j * U + u + 1] = pv[1];
                            r[j * U + u + 1] = (ok && i + 1 < len[j]) ? rv[1] : -1;
                            // two neighbouring entries of one row (the build places a row's entries of a
                            // cell side by side): one LDS add instead of two colliding ones
                            if (r[j * U + u] >= 0 && r[j * U + u] == r[j * U + u + 1]) {
                                p[j * U + u] = __fadd_rn(p[j * U + u], p[j * U + u + 1]);
                                r[j * U + u + 1] = -1;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kRuns; ++j) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int i = done + u * 64 + lane;
                            const bool ok = i < len[j];
                            p[j * U + u] = ok ? prod[begin[j] + i] : 0.0f;
                            r[j * U + u] = ok ? a_lrow[begin[j] + i] : -1;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < kSlots; ++q) {
                    if (r[q] >= 0) lds_add(&tile[r[q]], p[q]);
                }
            }
        }
    }
    __syncthreads();
}

// The tile (R floats, R = plan.tile_rows: any multiple of 64) lives in dynamic LDS.
template <int kReduceBlock, int U>
__global__ __launch_bounds__(kReduceBlock)
void tiled_reduce_kernel(int R, int num_tiles, const int2* __restrict__ cells_t, int num_strips,
                         const float* __restrict__ prod,
                         const unsigned short* __restrict__ a_lrow,
                         float* __restrict__ seed,
                         int num_rows, float* __restrict__ y) {
    extern __shared__ float tile[];
    const int tile_index = xcd_contiguous(blockIdx.x, num_tiles);
    if (tile_index < 0) return;
    tile_accumulate<kReduceBlock, U>(tile, R, tile_index, num_strips, num_rows, cells_t, prod, a_lrow, seed);
    const long long first = static_cast<long long>(tile_index) * R;
    for (int i = threadIdx.x; i < R && first + i < num_rows; i += kReduceBlock) y[first + i] = tile[i];
}

// phase 2 with the PageRank update fused into the tile write-out (cf. pr_step_kernel)
template <int kReduceBlock, int U>
__global__ __launch_bounds__(kReduceBlock)
void tiled_pagerank_reduce_kernel(int R, int num_tiles, const int2* __restrict__ cells_t, int num_strips,
                                  const float* __restrict__ prod,
                                  const unsigned short* __restrict__ a_lrow,
                                  float* __restrict__ seed,
                                  int local_rows, int row_offset, int n_global,
                                  const float* __restrict__ r_old, float* __restrict__ r_new,
                                  const unsigned char* __restrict__ dangling, float damping,
                                  const PrState* __restrict__ state,
                                  double* __restrict__ block_partials, PushTargets push) {
    if (state->done) return;
    extern __shared__ float tile[];
    const int tile_index = xcd_contiguous(blockIdx.x, num_tiles);
    if (tile_index < 0) return;
    tile_accumulate<kReduceBlock, U>(tile, R, tile_index, num_strips, local_rows, cells_t, prod, a_lrow, seed);

    const float teleport = __fdiv_rn(1.0f - damping, static_cast<float>(n_global));
    const float dangling_term = __fdiv_rn(__fmul_rn(damping, state->dangling_sum),
                                          static_cast<float>(n_global));
    double res2 = 0.0, mass = 0.0;
    const long long first = static_cast<long long>(tile_index) * R;
    for (int i = threadIdx.x; i < R && first + i < local_rows; i += kReduceBlock) {
        const long long node = row_offset + first + i;
        const float fresh = __fadd_rn(__fadd_rn(__fmul_rn(damping, tile[i]), dangling_term), teleport);
        r_new[node] = fresh;
        for (int p = 0; p < push.count; ++p) push.ptr[p][node] = fresh;     // straight into the peers' vectors
        const float diff = __fsub_rn(fresh, r_old[node]);
        res2 += static_cast<double>(__fmul_rn(diff, diff));
        if (dangling[node]) mass += static_cast<double>(fresh);
    }
    block_sum2<kReduceBlock>(res2, mass);
    if (threadIdx.x == 0) {
        block_partials[2 * tile_index] = res2;
        block_partials[2 * tile_index + 1] = mass;
    }
}

template <typename T>
hipError_t dev_alloc(T** p, long long count) {
    return hipMalloc(reinterpret_cast<void**>(p), static_cast<size_t>(std::max<long long>(count, 1)) * sizeof(T));
}

constexpr int kBucketLdsMaxStrips = 7680;     // 2 x 4 B x strips of dynamic LDS stays under the 64 KiB default limit

template <int LANES, int PASS>
hipError_t launch_bucket(const CSRMatrix* A, const TiledPlan& plan, int* counter, const int* offs,
                         int* num_long, hipStream_t s) {
    if (plan.num_strips <= kBucketLdsMaxStrips) {
        // ~16 K entries per workgroup, whole sweeps of rows, never across a tile boundary
        constexpr int kRowsPerSweep = kBlock / LANES;
        const double mean = std::max(1.0, static_cast<double>(A->nnz) / std::max(A->num_rows, 1));
        long long rows_per_block = static_cast<long long>(16384.0 / mean);
        rows_per_block = std::max<long long>(kRowsPerSweep, rows_per_block / kRowsPerSweep * kRowsPerSweep);
        rows_per_block = std::min<long long>(rows_per_block, (plan.tile_rows + kRowsPerSweep - 1) / kRowsPerSweep * kRowsPerSweep);
        const int blocks_per_tile = static_cast<int>((plan.tile_rows + rows_per_block - 1) / rows_per_block);
        const long long grid = static_cast<long long>(plan.num_tiles) * blocks_per_tile;
        if (grid <= 0x7fffffffLL) {
            const size_t lds = static_cast<size_t>(plan.num_strips) * sizeof(int) * (PASS == 0 ? 1 : 2);
            bucket_lds_kernel<LANES, PASS><<<static_cast<int>(grid), kBlock, lds, s>>>(
                A->num_rows, plan.num_tiles, plan.num_strips, plan.strip_cols, plan.tile_rows, plan.long_row,
                static_cast<int>(rows_per_block), blocks_per_tile, A->d_row_ptrs, A->d_col_indices, A->d_values,
                counter, offs, plan.a_val, plan.a_lcol, plan.a_lrow, plan.long_rows, num_long);
            return hipGetLastError();
        }
    }
    const int rows_per_block = kBlock / LANES;
    const int grid = (A->num_rows + rows_per_block - 1) / rows_per_block;
    bucket_kernel<LANES, PASS><<<grid, kBlock, 0, s>>>(
        A->num_rows, plan.num_tiles, plan.strip_cols, plan.tile_rows, plan.long_row, A->d_row_ptrs,
        A->d_col_indices, A->d_values,
        counter, offs, plan.a_val, plan.a_lcol, plan.a_lrow, plan.long_rows, num_long);
    return hipGetLastError();
}

template <int PASS>
hipError_t launch_bucket_lanes(int lanes, const CSRMatrix* A, const TiledPlan& plan, int* counter,
                               const int* offs, int* num_long, hipStream_t s) {
    switch (lanes) {
        case 1:  return launch_bucket<1, PASS>(A, plan, counter, offs, num_long, s);
        case 2:  return launch_bucket<2, PASS>(A, plan, counter, offs, num_long, s);
        case 4:  return launch_bucket<4, PASS>(A, plan, counter, offs, num_long, s);
        case 8:  return launch_bucket<8, PASS>(A, plan, counter, offs, num_long, s);
        case 16: return launch_bucket<16, PASS>(A, plan, counter, offs, num_long, s);
        case 32: return launch_bucket<32, PASS>(A, plan, counter, offs, num_long, s);
        default: return launch_bucket<64, PASS>(A, plan, counter, offs, num_long, s);
    }
}

// W / R for a matrix: as many row tiles as it takes to fill the chip several times over
// (phase 2 parallelism), strips wide enough that a cell's run averages >= ~128 entries
// (phase 2 reads one run per cell); when even the widest strip cannot give that (wide
// shards of a row-partitioned matrix), trade tiles for run length.
void choose_shape(long long num_rows, long long num_cols, long long nnz, int* strip_cols, int* tile_rows) {
    auto tiles_for = [&](int r) { return (num_rows + r - 1) / r; };
    auto strips_for = [&](int w) { return (num_cols + w - 1) / w; };
    int r = 8192;
    while (r > 1024 && tiles_for(r) < 1024) r >>= 1;
    int w = 4096;
    for (;;) {
        w = 4096;
        while (w < 32768 && nnz / (strips_for(w) * tiles_for(r)) < kTargetRun) w <<= 1;
        const bool long_enough = nnz / (strips_for(w) * tiles_for(r)) >= kTargetRun;
        // taller tiles lengthen the runs but cost phase-2 parallelism: keep >= ~600 tiles
        // (measured on a 1.25 M x 10 M shard: 611 tiles / 107-entry runs 88 us, 306 / 213 105 us)
        if (long_enough || r >= 8192 || tiles_for(2 * r) < 600) break;
        r <<= 1;
    }
    // Phase 2 keeps kResidentTiles workgroups on the chip at once (4 per CU while a tile is <= ~39 KiB);
    // tiles all cost the same, so a count just above a multiple of that leaves the chip nearly idle
    // for a whole extra round (C5 at R = 8192: 1221 tiles = 1.19 rounds).  Stretch the tiles so they
    // fill whole rounds (R need not be a power of two), or shrink them if stretching would not fit.
    {
        const long long tiles = tiles_for(r);
        const long long rounds = tiles / kResidentTiles;
        if (rounds >= 1 && tiles % kResidentTiles != 0) {
            auto snapped = [&](long long rounds_wanted) {
                const long long per_tile = (num_rows + rounds_wanted * kResidentTiles - 1) / (rounds_wanted * kResidentTiles);
                return static_cast<int>((per_tile + 63) / 64 * 64);
            };
            int stretched = snapped(rounds);
            r = stretched <= kMaxTileRows ? stretched : snapped(rounds + 1);
        }
    }
    if (const char* env = std::getenv("SPMV_TILED_STRIP")) {
        const int v = std::atoi(env);
        if (v == 4096 || v == 8192 || v == 16384 || v == 32768) w = v;
    }
    if (const char* env = std::getenv("SPMV_TILED_TILE")) {
        const int v = std::atoi(env);
        if (v >= 64 && v <= 15360 && v % 64 == 0) r = v;
    }
    *strip_cols = w;
    *tile_rows = r;
}

template <int W, int BLOCK>
hipError_t launch_expand_as(const TiledPlan& plan, const float* d_x, const PrState* d_state, hipStream_t s) {
    const LongRows lr{plan.long_chunks, plan.num_long_chunks, plan.csr_nnz, plan.csr_cols, plan.csr_vals, plan.seed};
    const int long_blocks = xcd_grid((plan.num_long_chunks + BLOCK / 64 - 1) / (BLOCK / 64));
    const int grid = long_blocks + xcd_grid(plan.num_items);
    if (plan.col_weight) {
        tiled_expand_kernel<W, BLOCK, true><<<grid, BLOCK, 0, s>>>(
            plan.items, plan.num_items, long_blocks, nullptr, plan.a_lcol, plan.col_weight, d_x, plan.num_cols, plan.prod, lr, d_state);
    } else {
        tiled_expand_kernel<W, BLOCK, false><<<grid, BLOCK, 0, s>>>(
            plan.items, plan.num_items, long_blocks, plan.a_val, plan.a_lcol, nullptr, d_x, plan.num_cols, plan.prod, lr, d_state);
    }
    return hipGetLastError();
}

hipError_t launch_expand(const TiledPlan& plan, const float* d_x, const PrState* d_state, hipStream_t s) {
    if (plan.num_items == 0 && plan.num_long_chunks == 0) return hipSuccess;
    switch (plan.strip_cols) {
        case 4096:  return launch_expand_as<4096, 512>(plan, d_x, d_state, s);
        case 8192:  return launch_expand_as<8192, 512>(plan, d_x, d_state, s);
        case 16384: return launch_expand_as<16384, 512>(plan, d_x, d_state, s);
        default:    return launch_expand_as<32768, 1024>(plan, d_x, d_state, s);   // 128 KiB of LDS: one workgroup per CU
    }
}

template <int U>
hipError_t launch_reduce_as(const TiledPlan& plan, float* d_y, hipStream_t s) {
    tiled_reduce_kernel<512, U><<<xcd_grid(plan.num_tiles), 512, plan.tile_rows * sizeof(float), s>>>(
        plan.tile_rows, plan.num_tiles, reinterpret_cast<const int2*>(plan.cells_t), plan.num_strips, plan.prod, plan.a_lrow,
        plan.seed, plan.num_rows, d_y);
    return hipGetLastError();
}

hipError_t launch_reduce(const TiledPlan& plan, float* d_y, hipStream_t s) {
    switch (plan.run_chunks) {
        case 1:  return launch_reduce_as<1>(plan, d_y, s);
        case 2:  return launch_reduce_as<2>(plan, d_y, s);
        default: return launch_reduce_as<4>(plan, d_y, s);
    }
}

template <int U>
hipError_t launch_pagerank_reduce_as(const TiledPlan& plan, int row_offset, int n_global, const float* d_r_old,
                                     float* d_r_new, const unsigned char* d_dangling, float damping,
                                     const PrState* d_state, double* d_block_partials,
                                     const PushTargets& push, hipStream_t s) {
    tiled_pagerank_reduce_kernel<512, U><<<xcd_grid(plan.num_tiles), 512, plan.tile_rows * sizeof(float), s>>>(
        plan.tile_rows, plan.num_tiles, reinterpret_cast<const int2*>(plan.cells_t), plan.num_strips, plan.prod, plan.a_lrow,
        plan.seed, plan.num_rows, row_offset, n_global, d_r_old, d_r_new, d_dangling, damping, d_state,
        d_block_partials, push);
    return hipGetLastError();
}

hipError_t launch_pagerank_reduce(const TiledPlan& plan, int row_offset, int n_global, const float* d_r_old,
                                  float* d_r_new, const unsigned char* d_dangling, float damping,
                                  const PrState* d_state, double* d_block_partials,
                                  const PushTargets& push, hipStream_t s) {
    switch (plan.run_chunks) {
        case 1:  return launch_pagerank_reduce_as<1>(plan, row_offset, n_global, d_r_old, d_r_new, d_dangling,
                                                     damping, d_state, d_block_partials, push, s);
        case 2:  return launch_pagerank_reduce_as<2>(plan, row_offset, n_global, d_r_old, d_r_new, d_dangling,
                                                     damping, d_state, d_block_partials, push, s);
        default: return launch_pagerank_reduce_as<4>(plan, row_offset, n_global, d_r_old, d_r_new, d_dangling,
                                                     damping, d_state, d_block_partials, push, s);
    }
}

} // namespace

namespace {

bool eligible_dims(long long rows, long long cols, long long nnz) {
    static const bool enabled = [] {
        const char* env = std::getenv("SPMV_TILED");
        return !(env && env[0] == '0');
    }();
    static const long long min_cols = [] {
        const char* env = std::getenv("SPMV_TILED_MIN_COLS");
        // measured crossover (tools/quick_bench.py crossover, 1 M rows x 16): 65536 columns tie
        // (69 vs 71 us), 131072 columns 60 vs 74 us, 262144 columns 57 vs 76 us
        return env ? std::atoll(env) : 65536LL;
    }();
    static const long long min_nnz = [] {
        const char* env = std::getenv("SPMV_TILED_MIN_NNZ");      // tests force small matrices through the engine
        return env ? std::atoll(env) : 1LL << 20;
    }();
    if (!enabled || rows <= 0 || nnz < min_nnz || cols < min_cols) return false;
    int w = 0, r = 0;
    choose_shape(rows, cols, nnz, &w, &r);
    return ((cols + w - 1) / w) * ((rows + r - 1) / r) <= kMaxCells;
}

// where the entries come from: exactly one of csr / ell is set
struct Source {
    const CSRMatrix* csr = nullptr;
    const ELLMatrix* ell = nullptr;
    int rows = 0, cols = 0;
    long long nnz = 0;        // CSR: exact; ELL: slots (upper bound, used for shape / capacity only)
};

hipError_t build_plan(const Source& src, TiledPlan** out, hipStream_t s);

} // namespace

bool tiled_shape_for(long long rows, long long cols, long long nnz, int* strip_cols, int* tile_rows) {
    int w = 0, r = 0;
    if (rows > 0 && cols > 0 && nnz > 0) choose_shape(rows, cols, nnz, &w, &r);
    if (strip_cols) *strip_cols = w;
    if (tile_rows) *tile_rows = r;
    return eligible_dims(rows, cols, nnz);
}

bool tiled_eligible(const CSRMatrix* A) {
    return A && eligible_dims(A->num_rows, A->num_cols, A->nnz);
}

bool tiled_eligible(const ELLMatrix* A) {
    return A && eligible_dims(A->num_rows, A->num_cols, static_cast<long long>(A->num_rows) * A->max_nnz_per_row);
}

hipError_t tiled_build(const CSRMatrix* A, TiledPlan** out, hipStream_t s) {
    Source src;
    src.csr = A;
    src.rows = A->num_rows;
    src.cols = A->num_cols;
    src.nnz = A->nnz;
    return build_plan(src, out, s);
}

hipError_t tiled_build(const ELLMatrix* A, TiledPlan** out, hipStream_t s) {
    Source src;
    src.ell = A;
    src.rows = A->num_rows;
    src.cols = A->num_cols;
    src.nnz = static_cast<long long>(A->num_rows) * A->max_nnz_per_row;
    return build_plan(src, out, s);
}

void tiled_free(TiledPlan* p) {
    if (!p) return;
    void* owned[] = {p->a_val, p->a_lcol, p->a_lrow, p->prod, p->cells_t, p->items, p->long_rows, p->long_chunks,
                     p->seed, p->col_weight};
    for (void* q : owned) if (q) (void)hipFree(q);
    delete p;
}

namespace {

hipError_t build_plan(const Source& src, TiledPlan** out, hipStream_t s) {
    const CSRMatrix* A = src.csr;          // null for an ELL source (then no long-row path)
    *out = nullptr;
    TiledPlan* plan = new TiledPlan();
    plan->num_rows = src.rows;
    plan->num_cols = src.cols;
    plan->csr_nnz = src.nnz;
    if (A) {
        plan->csr_row_ptrs = A->d_row_ptrs;
        plan->csr_cols = A->d_col_indices;
        plan->csr_vals = A->d_values;
    }
    choose_shape(src.rows, src.cols, src.nnz, &plan->strip_cols, &plan->tile_rows);
    plan->run_chunks = 2;
    plan->num_strips = (src.cols + plan->strip_cols - 1) / plan->strip_cols;
    plan->num_tiles = (src.rows + plan->tile_rows - 1) / plan->tile_rows;
    const long long cells = static_cast<long long>(plan->num_strips) * plan->num_tiles;
    // A row spreads over the strips; once it averages more than ~2 entries per cell its lanes
    // start to collide on one LDS word in phase 2, so such rows take the direct path instead.
    // Where the line sits depends on what the direct path's gathers cost: with x inside the L2s
    // (<= 8 MB) they are cheap and 2 entries per cell is the limit (C4, 62 strips: 124 entries 53 us,
    // 248 entries 55 us); with a large x every gather is a fabric request and rows stay in the cells
    // longer (10 M x 10 M power-law matrix, 611 strips: limit 1024 475 us, 2048 449 us, 4096 454 us).
    int long_factor = static_cast<long long>(src.cols) * 4 > (8LL << 20) ? 4 : 2;
    if (const char* env = std::getenv("SPMV_TILED_LONG_FACTOR")) long_factor = std::max(1, std::atoi(env));
    int long_cap = kMaxLongRow;
    if (const char* env = std::getenv("SPMV_TILED_LONG_CAP")) long_cap = std::max(64, std::atoi(env));
    plan->long_row = A ? std::max(64, std::min(long_cap, long_factor * plan->num_strips)) : 0x7fffffff;
    const long long long_capacity = src.nnz / plan->long_row + 1;

    int *cnt = nullptr, *offs = nullptr, *strip_begin = nullptr, *num_long = nullptr;
    auto cleanup = [&](hipError_t e) {
        for (int* q : {cnt, offs, strip_begin, num_long}) if (q) (void)hipFree(q);
        if (e != hipSuccess) tiled_free(plan);
        return e;
    };

    hipError_t e = dev_alloc(&cnt, cells);
    if (e == hipSuccess) e = dev_alloc(&offs, cells + 1);
    if (e == hipSuccess) e = dev_alloc(&strip_begin, plan->num_strips + 1);
    if (e == hipSuccess) e = dev_alloc(&num_long, 1);
    if (e == hipSuccess) e = dev_alloc(&plan->long_rows, long_capacity);
    if (e == hipSuccess) e = dev_alloc(&plan->cells_t, 2 * cells);
    if (e != hipSuccess) return cleanup(e);

    const int lanes = std::min(pick_lanes_per_row(static_cast<float>(src.nnz) / src.rows) * 4, 64);
    auto bucket = [&](int pass) -> hipError_t {
        if (A) {
            return pass == 0 ? launch_bucket_lanes<0>(lanes, A, *plan, cnt, nullptr, num_long, s)
                             : launch_bucket_lanes<1>(lanes, A, *plan, cnt, offs, num_long, s);
        }
        const ELLMatrix* E = src.ell;
        const int grid = (E->num_rows + kBlock - 1) / kBlock;
        if (pass == 0) {
            bucket_ell_kernel<0><<<grid, kBlock, 0, s>>>(E->num_rows, E->max_nnz_per_row, plan->num_tiles,
                                                        plan->strip_cols, plan->tile_rows, E->d_col_indices,
                                                        E->d_values, cnt, nullptr, nullptr, nullptr, nullptr);
        } else {
            bucket_ell_kernel<1><<<grid, kBlock, 0, s>>>(E->num_rows, E->max_nnz_per_row, plan->num_tiles,
                                                        plan->strip_cols, plan->tile_rows, E->d_col_indices,
                                                        E->d_values, cnt, offs, plan->a_val, plan->a_lcol,
                                                        plan->a_lrow);
        }
        return hipGetLastError();
    };

    // pass 0: cell sizes + the list of long rows
    e = hipMemsetAsync(cnt, 0, cells * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(num_long, 0, sizeof(int), s);
    if (e == hipSuccess) e = bucket(0);
    if (e == hipSuccess) {
        exclusive_scan_kernel<<<1, 1024, 0, s>>>(cnt, cells, offs);
        e = hipGetLastError();
    }
    int totals[2] = {0, 0};   // entries in cells, long rows
    if (e == hipSuccess) e = hipMemcpyAsync(&totals[0], offs + cells, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&totals[1], num_long, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);
    plan->nnz = totals[0];
    plan->num_long = totals[1];
    {   // 64-entry chunks taken from a run per pass of phase 2: cover the mean run with some slack
        const long long mean_run = plan->nnz / std::max<long long>(cells, 1);
        // same-box A/B with the paired loads: 4 chunks win from ~130-entry runs on (C2 133: 54.7 -> 54.2 us,
        // C4 168: 55.8 -> 54.4, C5 256: 532 -> 521), 2 chunks below (1/8 shard, 107: 84 vs 86.5)
        plan->run_chunks = mean_run <= 48 ? 1 : (mean_run <= 120 ? 2 : 4);
        if (const char* env = std::getenv("SPMV_TILED_CHUNKS")) {
            const int v = std::atoi(env);
            if (v == 1 || v == 2 || v == 4) plan->run_chunks = v;
        }
    }

    if (A && plan->num_long > 0) {
        // cut the long rows into wavefront-sized chunks (the list is short: <= nnz / long_row rows)
        std::vector<int> rows(plan->num_long);
        e = hipMemcpy(rows.data(), plan->long_rows, rows.size() * sizeof(int), hipMemcpyDeviceToHost);
        std::vector<int> chunks;
        std::vector<int> all_ptrs;                       // many long rows: one bulk copy instead
        const int* host_ptrs = A->row_ptrs;
        if (!host_ptrs && plan->num_long > 256 && e == hipSuccess) {
            all_ptrs.resize(static_cast<size_t>(A->num_rows) + 1);
            e = hipMemcpy(all_ptrs.data(), A->d_row_ptrs, all_ptrs.size() * sizeof(int), hipMemcpyDeviceToHost);
            host_ptrs = all_ptrs.data();
        }
        for (int row : rows) {
            int span[2] = {0, 0};
            if (host_ptrs) {
                span[0] = host_ptrs[row];
                span[1] = host_ptrs[row + 1];
            } else if (e == hipSuccess) {
                e = hipMemcpy(span, A->d_row_ptrs + row, sizeof(span), hipMemcpyDeviceToHost);
            }
            for (int b = span[0]; e == hipSuccess && b < span[1]; b += kLongChunk) {
                chunks.push_back(row);
                chunks.push_back(b);
                chunks.push_back(std::min(b + kLongChunk, span[1]));
            }
        }
        plan->num_long_chunks = static_cast<int>(chunks.size() / 3);
        if (e == hipSuccess) e = dev_alloc(&plan->long_chunks, static_cast<long long>(chunks.size()));
        if (e == hipSuccess) e = hipMemcpy(plan->long_chunks, chunks.data(), chunks.size() * sizeof(int),
                                           hipMemcpyHostToDevice);
        if (e != hipSuccess) return cleanup(e);
    }

    // column-weight folding (see column_weight_probe_kernel): on unless SPMV_TILED_FOLD=0
    bool fold = true;
    if (const char* env = std::getenv("SPMV_TILED_FOLD")) fold = env[0] != '0';
    if (fold && plan->nnz > 0) {
        const int* src_cols = A ? A->d_col_indices : src.ell->d_col_indices;
        const float* src_vals = A ? A->d_values : src.ell->d_values;
        int* differs = num_long;                       // its count is on the host already: reuse the word
        const int grid = static_cast<int>(std::min<long long>((src.nnz + kBlock - 1) / kBlock, 16384));
        e = dev_alloc(&plan->col_weight, plan->num_cols);
        if (e == hipSuccess) e = hipMemsetAsync(plan->col_weight, 0, static_cast<size_t>(plan->num_cols) * sizeof(float), s);
        if (e == hipSuccess) e = hipMemsetAsync(differs, 0, sizeof(int), s);
        // first the leading 1 M entries only: with arbitrary values a column that occurs twice there
        // already differs, and the two full passes (6 ms on C5) are skipped
        int host_differs = 1;
        const long long sample = std::min<long long>(src.nnz, 1LL << 20);
        for (long long count : {sample, static_cast<long long>(src.nnz)}) {
            const int launch = static_cast<int>(std::min<long long>((count + kBlock - 1) / kBlock, grid));
            if (e == hipSuccess) {
                column_weight_probe_kernel<0><<<launch, kBlock, 0, s>>>(src_cols, src_vals, count, plan->col_weight, differs);
                column_weight_probe_kernel<1><<<launch, kBlock, 0, s>>>(src_cols, src_vals, count, plan->col_weight, differs);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&host_differs, differs, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return cleanup(e);
            if (host_differs || count == src.nnz) break;
        }
        if (host_differs) {
            (void)hipFree(plan->col_weight);
            plan->col_weight = nullptr;
        }
    }

    if (!plan->col_weight) e = dev_alloc(&plan->a_val, plan->nnz);
    // + 8: the paired / 16-byte loads of a run's last group may touch a few entries past the end
    if (e == hipSuccess) e = dev_alloc(&plan->a_lcol, plan->nnz + 8);
    if (e == hipSuccess) e = dev_alloc(&plan->a_lrow, plan->nnz + 8);
    if (e == hipSuccess) e = dev_alloc(&plan->prod, plan->nnz + 8);
    if (e == hipSuccess && plan->num_long > 0) {
        e = dev_alloc(&plan->seed, plan->num_rows);
        if (e == hipSuccess) e = hipMemsetAsync(plan->seed, 0, static_cast<size_t>(plan->num_rows) * sizeof(float), s);
    }
    if (e != hipSuccess) return cleanup(e);

    // pass 1: scatter the short rows' entries into their cells
    e = hipMemsetAsync(cnt, 0, cells * sizeof(int), s);
    if (e == hipSuccess) e = bucket(1);
    if (e == hipSuccess) {
        const int grid = static_cast<int>(std::min<long long>((cells + kBlock) / kBlock, 4096));
        cell_table_kernel<<<grid, kBlock, 0, s>>>(offs, plan->num_strips, plan->num_tiles,
                                                reinterpret_cast<int2*>(plan->cells_t), strip_begin);
        e = hipGetLastError();
    }
    std::vector<int> host_strip(plan->num_strips + 1);
    if (e == hipSuccess) e = hipMemcpyAsync(host_strip.data(), strip_begin, host_strip.size() * sizeof(int),
                                            hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);

    // phase-1 work items: every strip's range cut into EQUAL pieces of <= item_entries (enough
    // pieces to fill the chip several times), piece boundaries on multiples of 4 entries
    const long long floor_entries = std::max<long long>(kMinItemEntries, plan->strip_cols);   // strip load <= 40 % of the stream
    int item_entries = static_cast<int>(std::min<long long>(
        kMaxItemEntries, std::max<long long>(floor_entries, (plan->nnz / 2048 + 3) / 4 * 4)));
    if (const char* env = std::getenv("SPMV_TILED_ITEM")) item_entries = std::max(1024, std::atoi(env));
    std::vector<int> items;
    for (int strip = 0; strip < plan->num_strips; ++strip) {
        const int begin = host_strip[strip], stop = host_strip[strip + 1];
        const int parts = (stop - begin + item_entries - 1) / item_entries;
        int b = begin;
        for (int part = 1; part <= parts; ++part) {
            int next = part == parts ? stop
                                     : static_cast<int>(begin + static_cast<long long>(stop - begin) * part / parts) / 4 * 4;
            next = std::max(next, b);
            if (next == b && part != parts) continue;
            items.push_back(strip);
            items.push_back(b);
            items.push_back(next);
            b = next;
        }
    }
    plan->num_items = static_cast<int>(items.size() / 3);
    e = dev_alloc(&plan->items, static_cast<long long>(items.size()));
    if (e == hipSuccess && !items.empty()) {
        e = hipMemcpy(plan->items, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return cleanup(e);
    (void)cleanup(hipSuccess);
    *out = plan;
    return hipSuccess;
}

} // namespace

hipError_t tiled_spmv(const TiledPlan& plan, const float* d_x, float* d_y, hipStream_t s) {
    const hipError_t e = launch_expand(plan, d_x, nullptr, s);       // phase 1 + the long rows
    if (e != hipSuccess) return e;
    return launch_reduce(plan, d_y, s);
}

hipError_t tiled_pagerank_step(const TiledPlan& plan, int row_offset, int n_global,
                               const float* d_r_old, float* d_r_new,
                               const unsigned char* d_dangling, float damping,
                               const PrState* d_state, double* d_block_partials,
                               const PushTargets& push, hipStream_t s) {
    // After convergence both kernels return at once: r_new, the product stream and the seed vector
    // stay as the last committed step left them.
    const hipError_t e = launch_expand(plan, d_r_old, d_state, s);   // phase 1 + the long rows (no-op once done)
    if (e != hipSuccess) return e;
    return launch_pagerank_reduce(plan, row_offset, n_global, d_r_old, d_r_new, d_dangling, damping, d_state,
                                  d_block_partials, push, s);
}

} // namespace detail
} // namespace spmv
