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
//   layout : slots sorted by cell = (column strip, row tile), strip-major, and by (row, column)
//        inside a cell.  Per slot: value f32, local column u16, ROW DELTA u8 (row minus the row of the
//        cell's previous slot; 255 = "advance 255 rows, no entry" for the rare larger gaps) — 7 B
//        against CSR's 8 — plus a 4-byte product slot.  Cells are padded to a multiple of 4 slots.
//   phase 1 "expand" : a workgroup loads one x strip (W = 4 K .. 32 K columns, chosen per
//        matrix = 16 .. 128 KiB) into LDS, streams its share of the strip's slots (value, local column) with
//        16-byte loads, gathers x from LDS and stores the products — same order, so
//        loads and stores are all contiguous.
//   phase 2 "reduce" : a workgroup (1024 threads) owns one row tile: R rows of DOUBLES in dynamic LDS, R a
//        multiple of 64 up to 9984 = 78 KiB, two tiles per CU, R stretched so that the tiles fill whole rounds
//        of the 512 resident workgroups.  The tile's slots are one contiguous run per strip (cell table); a
//        wavefront walks the runs of its share of the strips as one stream of 256-slot PASSES whose geometry was
//        laid out when the plan was built (pass descriptors, below): it loads 4 products (16 B) + 4 row deltas
//        (4 B) per lane, rebuilds the rows with an in-lane prefix and one DPP wavefront scan, adds each product
//        into the tile with the hardware ds_add_f64 and finally the tile is rounded to fp32 and written out with
//        coalesced stores (optionally through the fused PageRank update).  Why doubles: gfx950's ds_add_f32 runs at 0.38 lanes/clk/CU, a
//        compare-and-swap add at 3.4 but with retry storms when the tile's wavefronts meet on hot rows (round
//        1: phase 2 VALU-bound at ~225 us whatever was changed), ds_add_f64 at 3.5 (8.6 on consecutive rows)
//        with no retries (tools/lds_bench.hip, profiles/r02_lds_bench.txt, r02_phase2_counters.txt).
//   folding : when every stored entry of a column has the same bits, the value stream is dropped
//        and phase 1 gathers w_j * x_j from LDS (strip_weight_kernel, FOLD instantiation).
//   both phases walk their work lists in per-XCD contiguous slices (xcd_contiguous).
//   long rows (more than min(4096, 8 entries per strip)) would make many lanes meet on one LDS word; they are
//        left out of the cells and summed in 512-entry chunks by extra wavefronts of the phase-1 grid (direct
//        gather) into one slot per chunk; phase 2's tile start folds a long row's chunk sums in chunk order.
//   build : batches of <= ~5 K entries (consecutive rows of ONE tile).  Ranking pass: the batch is binned by
//        strip in LDS and every entry ranked inside its bin by (row, column, source index); it leaves a 4-byte
//        record per entry (position inside its group, row delta, skip markers) and the group sizes.  One thread
//        per cell then places the groups of its batches, a scan over (strip, tile) gives the cell offsets, and
//        the placing pass writes every entry straight to its slot.  No global atomics anywhere, so the layout
//        is a pure function of the matrix.
//
// HBM traffic per slot: 6 B read + 4 B written in phase 1, 5 B read in phase 2 (15 B vs CSR's 8 B per entry) —
// but all of it is streamed, which beats one 64-byte random fetch per entry by a wide margin once x leaves L2.
// Reproducibility: layout and long-row sums are order-free; inside a tile the products of a row meet in
// scheduling order, but they are added as doubles (every fp32 product is exact there, the sum of a row's
// products carries ~29 spare bits) and rounded to fp32 once, so two runs give the same bits
// (tests/test_gpu_spmv.py holds this on C4 and C5 shapes).
#include "tiled.h"
#include "device_common.h"
#include "pagerank_engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace spmv {
namespace detail {

namespace {

using namespace dev;

// W (x columns per LDS strip) and R (y rows per LDS tile) are chosen per matrix
// (choose_shape below):
//   W in {4096, 8192, 16384, 32768} = 16 .. 128 KiB of static LDS in phase 1 (template instantiations)
//   R = any multiple of 64 in [64, kMaxTileRows]: dynamic LDS in phase 2
constexpr int kMaxItemEntries = 16384;   // phase-1 work item size bounds (entries).  Round 4: 65536 -> 16384 — with the strip staging
                                         // pipelined, many short workgroups balance better than few long ones (C5 phase 1: 321.7 us with
                                         // ~53 K-slot items, 332.6 with 44 K (7.16 rounds of 512), 315.9 with 29 K, 315.2 with 22 K, 310.8-311.4 with
                                         // 15.5 K; on a faster box 296-300 / 290-295 (16 K) / 292-294 (12 K) / 308-311 (8 K))
constexpr int kMinItemEntries = 4096;
constexpr int kMaxLongRow = 4096;     // rows longer than min(this, 8 entries per strip) bypass the cells
constexpr int kLongChunk = 512;       // entries per wavefront of the long-row path
constexpr long long kMaxCells = 1LL << 26;
constexpr long long kTargetRun = 128;        // wanted mean entries per cell (run length seen by phase 2)
constexpr long long kResidentTiles = 512;    // phase-2 workgroups resident at once: 256 CUs x 2 (a tile of doubles is <= 78 KiB)
constexpr int kMaxTileRows = 9984;           // 2 tiles of this many doubles (+ the reduction scratch) fit one CU's 160 KiB

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// The products are written once and read once, by other CUs, a whole kernel later: stored NON-TEMPORAL they do not linger
// as dirty lines in the Infinity Cache, whose write-back would otherwise compete with phase 2's reads (phase 1 takes
// ~13 us longer on C5, phase 2 ~29 us less: 518 -> 509 us per SpMV same box, 512 -> 489 on another;
// profiles/r04_store_flavours.txt — sc1 / sc0 sc1 write-through stores and nt LOADS on either phase lose).
template <typename T>
__device__ __forceinline__ void store_product(T* p, T v) { __builtin_nontemporal_store(v, p); }

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
__host__ __device__ inline int xcd_grid(int count) { return (count + kXcds - 1) / kXcds * kXcds; }

// ------------------------------------------------------------------ plan building ----
constexpr int kSkip = 255;                  // row-delta byte: advance 255 rows, no entry
constexpr int kBuildBlock = 1024;           // threads of a builder workgroup
constexpr int kBuildRowCache = 1024;        // row offsets of the batch kept in LDS for the entry -> row search
constexpr int kBuildLdsSmall = 70 * 1024;   // dynamic LDS of a builder workgroup (two per CU, next to 8 KiB static) ...
constexpr int kBuildLdsLarge = 148 * 1024;  // ... or one per CU when the strips are many
constexpr int kBuildBinWords = 4;           // LDS ints per strip: start, cursor, markers, first|last
constexpr int kBuildEntryBytes = 11;        // LDS bytes per entry: key 4, source index 4 (the row marks, 2, live there first), bin 2, markers 1
// per strip, next to the four bin words: one byte per wavefront of the ranking workgroup (sixteen): the stable binning's counters
constexpr int kBuildWaveCountBytes = kBuildBlock / 64;
constexpr int kMaxBuildStrips = 3072;

// where the entries come from.  offset(row) = index of the row's first entry in a virtual row-major
// numbering; col() < 0 marks ELL padding.
struct CsrSource {
    const int* row_ptrs;
    const int* cols;
    const float* vals;
    __device__ __forceinline__ long long offset(int row) const { return row_ptrs[row]; }
    __device__ __forceinline__ int col(long long j) const { return cols[j]; }
    __device__ __forceinline__ float val(long long j) const { return vals[j]; }
    static constexpr bool kSearchRows = true;      // an entry's row comes from a search over the row offsets
    __device__ __forceinline__ int direct_row(long long) const { return 0; }
};
struct EllSource {
    int rows, width;
    const int* cols;
    const float* vals;
    __device__ __forceinline__ long long offset(int row) const { return static_cast<long long>(row) * width; }
    __device__ __forceinline__ long long slot(long long j, int row) const {
        return (j - static_cast<long long>(row) * width) * rows + row;
    }
    static constexpr bool kSearchRows = false;     // rows have a fixed width
    __device__ __forceinline__ int direct_row(long long j) const { return static_cast<int>(j / width); }
    __device__ __forceinline__ int col(long long j) const { return cols[slot(j, direct_row(j))]; }
    __device__ __forceinline__ float val(long long j) const { return vals[slot(j, direct_row(j))]; }
};

struct BuildShape {
    int num_rows, num_tiles, num_strips, strip_shift, tile_rows, long_row;
    int any_long;               // some row is longer than long_row (then every entry's row length is checked)
    int stable_bins;            // the ranking pass may bin stably (no ranking loop); 0: always rank by comparison (SPMV_DEBUG=rank=plain)
    long long quota;            // entries per batch before the next one starts
};

// longest row (capped by the caller): sizes the batches
// one atomicMax per WORKGROUP: every wavefront adding its own to one address serialises 8 K atomics at the memory side (~80 us)
__device__ __forceinline__ void publish_block_max(int best, int* __restrict__ out) {
    __shared__ int s_best[kBlock / 64];
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        int all = s_best[0];
        for (int w = 1; w < kBlock / 64; ++w) all = max(all, s_best[w]);
        if (all > 0) atomicMax(out, all);
    }
}

template <typename Src>
__global__ __launch_bounds__(kBlock)
void max_row_kernel(Src src, int num_rows, int* __restrict__ out) {
    int best = 0;
    if constexpr (Src::kSearchRows) {
        // CSR: four rows per thread and step — one 16-byte load of the row pointers + the one behind them (the pointer array
        // is hipMalloc'd or a whole torch tensor in every caller; an unaligned base takes the plain loop below)
        const int* rp = src.row_ptrs;
        if ((reinterpret_cast<unsigned long long>(rp) & 15) == 0) {
            const long long groups = num_rows / 4;
            for (long long g = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; g < groups;
                 g += static_cast<long long>(gridDim.x) * kBlock) {
                const i32x4 v = *reinterpret_cast<const i32x4*>(rp + 4 * g);
                const int next = rp[4 * g + 4];
                best = max(max(best, v[1] - v[0]), max(max(v[2] - v[1], v[3] - v[2]), next - v[3]));
            }
            if (blockIdx.x == 0 && threadIdx.x < num_rows % 4) {
                const int r = num_rows / 4 * 4 + threadIdx.x;
                best = max(best, rp[r + 1] - rp[r]);
            }
            publish_block_max(best, out);
            return;
        }
    }
    for (long long r = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; r < num_rows;
         r += static_cast<long long>(gridDim.x) * kBlock) {
        best = max(best, static_cast<int>(src.offset(static_cast<int>(r) + 1) - src.offset(static_cast<int>(r))));
    }
    publish_block_max(best, out);
}

// batches per tile: a tile's rows are cut wherever the running entry count passes a multiple of quota
template <typename Src>
__global__ __launch_bounds__(kBlock)
void tile_batches_kernel(Src src, BuildShape sh, int* __restrict__ count) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= sh.num_tiles) return;
    const int r0 = static_cast<int>(min(static_cast<long long>(t) * sh.tile_rows, static_cast<long long>(sh.num_rows)));
    const int r1 = static_cast<int>(min(static_cast<long long>(r0) + sh.tile_rows, static_cast<long long>(sh.num_rows)));
    const long long entries = src.offset(r1) - src.offset(r0);
    count[t] = static_cast<int>(max(1LL, (entries + sh.quota - 1) / sh.quota));
}

// first row of every batch (binary search for the batch's entry offset inside its tile)
template <typename Src>
__global__ __launch_bounds__(kBlock)
void batch_rows_kernel(Src src, BuildShape sh, const int* __restrict__ tile_batch /*[tiles + 1]*/,
                       int* __restrict__ batch_row /*[batches + 1]*/, int* __restrict__ batch_tile) {
    const int t = blockIdx.x;
    const int r0 = static_cast<int>(min(static_cast<long long>(t) * sh.tile_rows, static_cast<long long>(sh.num_rows)));
    const int r1 = static_cast<int>(min(static_cast<long long>(r0) + sh.tile_rows, static_cast<long long>(sh.num_rows)));
    const long long origin = src.offset(r0);
    const int first = tile_batch[t], n = tile_batch[t + 1] - first;
    for (int b = threadIdx.x; b < n; b += kBlock) {
        const long long target = origin + static_cast<long long>(b) * sh.quota;
        int lo = r0, hi = r1;                         // first row whose offset >= target
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (src.offset(mid) >= target) hi = mid; else lo = mid + 1;
        }
        batch_row[first + b] = lo;
        batch_tile[first + b] = t;
    }
    if (t == sh.num_tiles - 1 && threadIdx.x == 0) batch_row[tile_batch[sh.num_tiles]] = sh.num_rows;
}

// per (batch, strip) group: what the ranking pass learns / what the placing pass needs (same 8-byte slot)
struct GroupCount { unsigned short count, first, last, escapes; };      // escapes: markers in front of the non-first slots
struct GroupPlace { unsigned int rel; unsigned short prev_last, lead; };   // rel: offset inside the cell; lead: markers
                                                                         // in front of the group's first slot
static_assert(sizeof(GroupCount) == 8 && sizeof(GroupPlace) == 8, "group records share storage");

// Per-entry record the ranking pass leaves for the placing pass (indexed like the source entries):
//   the first slot of its group : 1 << 31 | row inside the tile   (its delta and markers depend on the cell's
//                                                                   earlier batches: cell_place_kernel settles them)
//   any other slot              : p << 16 | markers << 8 | delta   (p = slots of the group in front of it, its own
//                                                                   markers included, the group's lead excluded)
//   an entry of a long row      : kMetaSkip
constexpr unsigned int kMetaFirst = 1u << 31;
constexpr unsigned int kMetaSkip = 0xFFFFFFFFu;
constexpr int kBuildPerThread = 8;                     // entries a builder thread keeps in registers
constexpr int kBuildMaxCapacity = kBuildBlock * kBuildPerThread;

// Exclusive scan over the threads of a builder workgroup (1024 = 16 wavefronts) of one non-negative int each —
// a sum, or a running maximum.  Shuffles inside the wavefronts and 16 wavefront totals through LDS: two barriers
// where a Hillis-Steele ladder over an LDS array takes twenty (three such scans per batch were about half of a
// batch's time).  `scratch`: 16 ints of LDS; `*all` receives the total of the whole workgroup.
template <bool kMax>
__device__ __forceinline__ int block_exclusive_scan(int own, int* scratch, int* all) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int other = __shfl_up(incl, off, 64);
        if (lane >= off) incl = kMax ? max(incl, other) : incl + other;
    }
    int before = __shfl_up(incl, 1, 64);
    if (lane == 0) before = 0;
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBuildBlock / 64; ++w) {
        const int t = scratch[w];
        if (w < wave) prefix = kMax ? max(prefix, t) : prefix + t;
        total = kMax ? max(total, t) : total + t;
    }
    __syncthreads();
    *all = total;
    return kMax ? max(prefix, before) : prefix + before;
}

// One batch (consecutive rows of one tile, at most `capacity` short-row entries): bin the entries by strip in
// LDS, rank every entry inside its bin by (row, column), derive the row deltas and the skip markers they need;
// report every bin's size and leave the per-entry records.  Dynamic LDS: kBuildBinWords ints per strip, then
// per entry key u32, source index u32, bin u16, row mark u16, markers u8.
template <typename Src>
__global__ __launch_bounds__(kBuildBlock, 8)          // two workgroups per CU: at most 64 registers
void batch_rank_kernel(Src src, BuildShape sh, int num_batches, int capacity,
                       const int* __restrict__ batch_row, const int* __restrict__ batch_tile,
                       uint2* __restrict__ groups,                 // [batches * strips] GroupCount
                       unsigned int* __restrict__ meta,            // [source entries]
                       int* __restrict__ long_rows, int* __restrict__ num_long) {
    extern __shared__ int build_lds[];
    __shared__ int s_partial[kBuildBlock / 64];
    __shared__ int s_row_cache[kBuildRowCache + 1];
    __shared__ int s_overflow;
    const int batch = xcd_contiguous(blockIdx.x, num_batches);
    if (batch < 0) return;
    const int S = sh.num_strips;
    int* bin_start = build_lds;                 // [S] first slot of the bin (after the scan)
    int* bin_cursor = build_lds + S;            // [S] histogram, then fill cursor (= bin end once filled)
    int* bin_escapes = build_lds + 2 * S;       // [S] skip markers needed in front of the bin's non-first slots
    int* bin_ends = build_lds + 3 * S;          // [S] first lrow << 16 | last lrow
    // per wavefront and strip one BYTE (four strips to a word): how many of the wavefront's entries fall into the strip,
    // then where in the bin its next one goes (stable binning, below)
    const int S4 = (S + 3) / 4;                 // words per wavefront
    unsigned int* wave_count = reinterpret_cast<unsigned int*>(build_lds + kBuildBinWords * S);
    unsigned int* keys = wave_count + (kBuildBlock / 64) * S4;    // lrow << 16 | lcol
    unsigned int* source = keys + capacity;     // index of the slot's entry, relative to the batch's first entry
    unsigned short* bin_of = reinterpret_cast<unsigned short*>(source + capacity);
    unsigned short* row_mark = reinterpret_cast<unsigned short*>(source);   // entry index -> row (relative), after a max-scan; read
                                                                            // for the last time before `source` is first written
    unsigned char* markers = reinterpret_cast<unsigned char*>(bin_of + capacity);
    __shared__ int s_plain;                     // this batch ranks its bins by comparison (the stable binning does not apply)

    const int tile = batch_tile[batch];
    const int row0 = batch_row[batch];
    // the next batch starts where this one ends — unless it belongs to the next tile
    const long long tile_end = min(static_cast<long long>(tile + 1) * sh.tile_rows, static_cast<long long>(sh.num_rows));
    const int row1 = batch + 1 < num_batches && batch_tile[batch + 1] == tile ? batch_row[batch + 1]
                                                                                : static_cast<int>(tile_end);
    const int tile_first = tile * sh.tile_rows;
    const long long entry0 = src.offset(row0), entry1 = src.offset(row1);
    // FAST: the batch's whole entry range fits the LDS arrays (always, unless long rows sit inside it): every
    // thread keeps its entries' columns and rows in registers between the phases, rows come from a scan
    const bool fast = entry1 - entry0 <= capacity;
    const int span = fast ? static_cast<int>(entry1 - entry0) : 0;
    // the fast path's column loads are issued first: they travel while the rows are being marked and scanned
    // Every wavefront takes a CONTIGUOUS share of the batch's entries, 64 at a time in source order (so that the stable
    // binning below can rely on "earlier wavefront, earlier instruction, lower lane = earlier entry").
    const int wave_id = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    const int per_wave = ((span + kBuildBlock / 64 - 1) / (kBuildBlock / 64) + 63) / 64 * 64;      // <= 64 * kBuildPerThread
    auto entry_of = [&](int u) {                       // index of this thread's u-th entry, or `span` (none)
        const int within = u * 64 + lane_id;
        return within < per_wave ? min(wave_id * per_wave + within, span) : span;
    };
    int my_col[kBuildPerThread];
#pragma unroll
    for (int u = 0; u < kBuildPerThread; ++u) {
        const int idx = entry_of(u);
        my_col[u] = idx < span ? src.col(entry0 + idx) : -1;
    }

    for (int i = threadIdx.x; i < S; i += kBuildBlock) {
        bin_cursor[i] = 0;
        bin_escapes[i] = 0;
        bin_ends[i] = 0;
    }
    for (int i = threadIdx.x; i < (kBuildBlock / 64) * S4; i += kBuildBlock) wave_count[i] = 0;
    if (threadIdx.x == 0) {
        s_overflow = 0;
        s_plain = fast && sh.stable_bins ? 0 : 1;
    }
    const bool rows_cached = Src::kSearchRows && !fast && row1 - row0 <= kBuildRowCache;
    if (rows_cached) {
        for (int r = row0 + threadIdx.x; r <= row1; r += kBuildBlock) s_row_cache[r - row0] = static_cast<int>(src.offset(r));
    }
    if (fast && Src::kSearchRows) {
        for (int i = threadIdx.x; i < span; i += kBuildBlock) row_mark[i] = 0;
    }
    __syncthreads();
    if (fast && Src::kSearchRows) {
        // every non-empty row marks its first entry; an inclusive max-scan then gives every entry its row
        for (int r = row0 + threadIdx.x; r < row1; r += kBuildBlock) {
            const long long b = src.offset(r);
            if (src.offset(r + 1) > b) row_mark[b - entry0] = static_cast<unsigned short>(r - row0);
        }
        __syncthreads();
        const int per = (span + kBuildBlock - 1) / kBuildBlock;
        const int lo = min(span, per * static_cast<int>(threadIdx.x)), hi = min(span, lo + per);
        int best = 0;
        for (int i = lo; i < hi; ++i) best = max(best, static_cast<int>(row_mark[i]));
        int unused;
        int run = block_exclusive_scan<true>(best, s_partial, &unused);
        for (int i = lo; i < hi; ++i) {
            run = max(run, static_cast<int>(row_mark[i]));
            row_mark[i] = static_cast<unsigned short>(run);
        }
        __syncthreads();
    }

    // SLOW path helpers (a batch whose entry range holds long rows): entries re-read per phase, rows searched
    auto row_of = [&](long long j) -> int {
        if (!Src::kSearchRows) return src.direct_row(j);
        int lo = row0, hi = row1;                  // offset(lo) <= j < offset(hi)
        if (rows_cached) {
            while (hi - lo > 1) {
                const int mid = lo + (hi - lo) / 2;
                if (s_row_cache[mid - row0] <= j) lo = mid; else hi = mid;
            }
        } else {
            while (hi - lo > 1) {
                const int mid = lo + (hi - lo) / 2;
                if (src.offset(mid) <= j) lo = mid; else hi = mid;
            }
        }
        return lo;
    };
    auto for_each_entry = [&](auto&& body) {
        for (long long j0 = entry0 + threadIdx.x; j0 < entry1; j0 += 4 * kBuildBlock) {
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long j = j0 + static_cast<long long>(u) * kBuildBlock;
                c[u] = j < entry1 ? src.col(j) : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c[u] >= 0) body(j0 + static_cast<long long>(u) * kBuildBlock, c[u]);
            }
        }
    };
    // the row of entry j, or -1 when that row is long (then listed once, at its first entry, and its entries
    // are marked for the placing pass)
    auto short_row = [&](long long j, int row) -> int {
        if (!sh.any_long) return row;
        const long long begin = src.offset(row);
        if (src.offset(row + 1) - begin <= sh.long_row) return row;
        if (j == begin) long_rows[atomicAdd(num_long, 1)] = row;
        meta[j] = kMetaSkip;
        return -1;
    };

    // ---- histogram of the batch's short-row entries over the strips
    int my_lrow[kBuildPerThread];
    if (fast) {
#pragma unroll
        for (int u = 0; u < kBuildPerThread; ++u) {
            const int idx = entry_of(u);
            my_lrow[u] = 0;
            if (my_col[u] >= 0) {
                const long long j = entry0 + idx;
                const int row = short_row(j, Src::kSearchRows ? row0 + row_mark[idx] : src.direct_row(j));
                if (row < 0) {
                    my_col[u] = -1;
                } else {
                    my_lrow[u] = row - tile_first;
                    const int strip = my_col[u] >> sh.strip_shift;
                    atomicAdd(&bin_cursor[strip], 1);
                    // (bytes may run over into their neighbours when a bin holds more than 255: such a batch ranks by comparison)
                    atomicAdd(&wave_count[wave_id * S4 + (strip >> 2)], 1u << (8 * (strip & 3)));
                }
            }
        }
    } else {
        for_each_entry([&](long long j, int c) {
            if (sh.any_long && short_row(j, row_of(j)) < 0) return;
            atomicAdd(&bin_cursor[c >> sh.strip_shift], 1);
        });
    }
    __syncthreads();

    // ---- exclusive scan of the histogram: every thread owns a contiguous piece of the strips
    int total = 0;
    {
        const int per = (S + kBuildBlock - 1) / kBuildBlock;
        const int lo = min(S, per * static_cast<int>(threadIdx.x)), hi = min(S, lo + per);
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += bin_cursor[i];
        int run = block_exclusive_scan<false>(sum, s_partial, &total);
        bool big = false;
        for (int i = lo; i < hi; ++i) {
            const int n = bin_cursor[i];
            bin_start[i] = run;
            bin_cursor[i] = run;
            run += n;
            big = big || n > 255;
        }
        if (big) s_plain = 1;           // a byte counter per wavefront and strip cannot hold this bin
        if (threadIdx.x == 0 && total > capacity) s_overflow = 1;
    }
    __syncthreads();
    if (s_overflow) return;          // cannot happen (the batch quota bounds the count); never write past LDS

    // STABLE BINNING (the usual case).  The entries of a batch arrive in (row, column) order — the order a bin must end up
    // in.  So instead of filling the bins in whatever order the atomics land and ranking every entry against its whole
    // bin afterwards (a loop as long as the longest bin of the wavefront: 2.4 of the kernel's 4.3 vector instructions per
    // entry, profiles/r03_build_counters.txt), every entry is sent straight to its final place: the count of its strip
    // in earlier wavefronts (the byte counters, turned into running offsets here) + its turn among its own wavefront's
    // entries (a returning LDS add; a wavefront issues its entries in source order).  The order in which ONE instruction's
    // lanes get their turn at the same counter is the hardware's; the bins are therefore checked afterwards (each slot
    // against its predecessor) and a batch that is out of order — also: rows whose columns are not ascending — falls back to
    // ranking by comparison.  Either way the layout is the same function of the matrix.
    const bool try_stable = s_plain == 0;
    if (try_stable) {
        // four strips at a time: the bytes of a word never carry into each other (every strip's total is <= 255 here)
        for (int i = threadIdx.x; i < S4; i += kBuildBlock) {
            unsigned int running = 0;
#pragma unroll
            for (int w = 0; w < kBuildBlock / 64; ++w) {
                const unsigned int mine = wave_count[w * S4 + i];
                wave_count[w * S4 + i] = running;
                running += mine;
            }
        }
        __syncthreads();
    }

    // ---- fill the bins (order inside a bin is arbitrary here; the ranking below fixes it)
    auto put = [&](int c, int lrow, unsigned int from) {
        const int strip = c >> sh.strip_shift;
        const int u = atomicAdd(&bin_cursor[strip], 1);
        keys[u] = (static_cast<unsigned int>(lrow) << 16) | static_cast<unsigned int>(c - (strip << sh.strip_shift));
        source[u] = from;
        bin_of[u] = static_cast<unsigned short>(strip);
    };
    if (try_stable) {
#pragma unroll
        for (int u = 0; u < kBuildPerThread; ++u) {
            if (my_col[u] >= 0) {
                const int strip = my_col[u] >> sh.strip_shift;
                const int shift = 8 * (strip & 3);
                const unsigned int before = atomicAdd(&wave_count[wave_id * S4 + (strip >> 2)], 1u << shift);
                const int slot = bin_start[strip] + static_cast<int>((before >> shift) & 0xFF);
                atomicAdd(&bin_cursor[strip], 1);          // (ends as the bin's end, like the unordered fill leaves it)
                keys[slot] = (static_cast<unsigned int>(my_lrow[u]) << 16) | static_cast<unsigned int>(my_col[u] - (strip << sh.strip_shift));
                source[slot] = static_cast<unsigned int>(entry_of(u));
                bin_of[slot] = static_cast<unsigned short>(strip);
            }
        }
    } else if (fast) {
#pragma unroll
        for (int u = 0; u < kBuildPerThread; ++u) {
            if (my_col[u] >= 0) put(my_col[u], my_lrow[u], entry_of(u));
        }
    } else {
        for_each_entry([&](long long j, int c) {
            const int row = row_of(j);
            if (sh.any_long && src.offset(row + 1) - src.offset(row) > sh.long_row) return;    // (listed above)
            put(c, row - tile_first, static_cast<unsigned int>(j - entry0));
        });
    }
    __syncthreads();

    // ---- rank inside the bin = number of slots ordered before this one; the largest key among them is the
    //      predecessor's.  Order: (row, column).  A row that stores one column twice (legal CSR) ties: such
    //      slots are ordered by their source index (the CSR order), found in a second, rare, loop.
    if (try_stable) {                     // is every slot behind its bin's previous one?  (row, column), twins by source index
        bool ordered = true;
#pragma unroll
        for (int k = 0; k < kBuildPerThread; ++k) {
            const int u = threadIdx.x + k * kBuildBlock;
            if (u < total && u > bin_start[bin_of[u]]) {
                const unsigned int pred = keys[u - 1], mine = keys[u];
                ordered = ordered && (pred < mine || (pred == mine && source[u - 1] < source[u]));
            }
        }
        if (!ordered) s_plain = 1;
        __syncthreads();
    }
    const bool stable = s_plain == 0;     // (the same for every thread: read after a barrier)
    int my_rank[kBuildPerThread], my_need[kBuildPerThread], my_delta[kBuildPerThread];
#pragma unroll
    for (int k = 0; k < kBuildPerThread; ++k) {
        const int u = threadIdx.x + k * kBuildBlock;
        my_rank[k] = -1;
        my_need[k] = 0;
        my_delta[k] = 0;
        if (u < total) {
            const int bin = bin_of[u];
            const int lo = bin_start[bin], hi = bin_cursor[bin];
            const unsigned int mine = keys[u];
            int rank = 0;
            unsigned int pred = 0;
            if (stable) {                 // the slot IS the rank
                rank = u - lo;
                pred = rank > 0 ? keys[u - 1] : 0u;
            } else {
                int ties = 0;
                for (int v = lo; v < hi; ++v) {
                    const unsigned int key = keys[v];
                    const bool less = key < mine;
                    rank += less;
                    pred = less ? max(pred, key) : pred;
                    ties += key == mine;
                }
                if (ties > 1) {           // duplicate (row, column): order the twins by source index
                    const unsigned int me = source[u];
                    for (int v = lo; v < hi; ++v) {
                        if (keys[v] == mine && source[v] < me) {
                            ++rank;
                            pred = mine;
                        }
                    }
                }
            }
            const int lrow = static_cast<int>(mine >> 16);
            my_rank[k] = rank;
            if (rank > 0) {
                const int gap = lrow - static_cast<int>(pred >> 16);
                my_need[k] = gap / kSkip;                 // skip markers in front of this slot
                my_delta[k] = gap - my_need[k] * kSkip;
                if (my_need[k]) atomicAdd(&bin_escapes[bin], my_need[k]);
            } else {
                atomicOr(&bin_ends[bin], lrow << 16);     // exactly one slot per bin comes first ...
            }
            if (rank == hi - lo - 1) atomicOr(&bin_ends[bin], lrow);     // ... and exactly one last
            markers[u] = static_cast<unsigned char>(my_need[k]);
        }
    }
    __syncthreads();

    // ---- the per-entry records for the placing pass
#pragma unroll
    for (int k = 0; k < kBuildPerThread; ++k) {
        const int u = threadIdx.x + k * kBuildBlock;
        if (u < total) {
            const int bin = bin_of[u];
            const unsigned int mine = keys[u];
            unsigned int record;
            if (my_rank[k] == 0) {
                record = kMetaFirst | (mine >> 16);
            } else {
                int in_front = my_rank[k] + my_need[k];
                if (bin_escapes[bin] != my_need[k]) {       // rare: other slots of this bin need markers too
                    const unsigned int me = source[u];
                    for (int v = bin_start[bin]; v < bin_cursor[bin]; ++v) {
                        const unsigned int key = keys[v];
                        if (key < mine || (key == mine && source[v] < me)) in_front += markers[v];
                    }
                }
                record = (static_cast<unsigned int>(in_front) << 16) | (static_cast<unsigned int>(my_need[k]) << 8) |
                         static_cast<unsigned int>(my_delta[k]);
            }
            meta[entry0 + source[u]] = record;
        }
    }
    for (int i = threadIdx.x; i < S; i += kBuildBlock) {
        GroupCount g;
        g.count = static_cast<unsigned short>(bin_cursor[i] - bin_start[i]);
        g.first = static_cast<unsigned short>(static_cast<unsigned int>(bin_ends[i]) >> 16);
        g.last = static_cast<unsigned short>(bin_ends[i] & 0xFFFF);
        g.escapes = static_cast<unsigned short>(bin_escapes[i]);
        uint2 packed;
        __builtin_memcpy(&packed, &g, sizeof(g));
        groups[static_cast<long long>(batch) * S + i] = packed;
    }
}

// The placing pass: no sorting any more — every entry of the batch goes to cell begin + group offset + the
// position the ranking pass recorded, preceded by its skip markers.  Dynamic LDS: 12 bytes per strip (the
// batch's group records and its tile's cell begins).
template <typename Src>
__global__ __launch_bounds__(kBuildBlock)
void batch_place_kernel(Src src, BuildShape sh, int num_batches,
                        const int* __restrict__ batch_row, const int* __restrict__ batch_tile,
                        const uint2* __restrict__ groups,              // [batches * strips] GroupPlace
                        const unsigned int* __restrict__ meta, const int2* __restrict__ cells_t,
                        float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                        unsigned char* __restrict__ a_drow,
                        const int* __restrict__ todo /*null: every batch; else [0] = how many, [1 ...] = which (left by the staged pass)*/) {
    extern __shared__ int place_lds[];
    const int S = sh.num_strips;
    uint2* place = reinterpret_cast<uint2*>(place_lds);
    int* cell_begin = place_lds + 2 * S;
    // behind the staged pass only the batches it listed are left (usually none): a fixed grid walks the list — a workgroup
    // per batch just to find out that there is nothing to do cost ~40 us of launches on C5
    const int work = todo ? todo[0] : xcd_grid(num_batches);
    for (int position = blockIdx.x; position < work; position += gridDim.x) {
    const int batch = todo ? todo[1 + position] : xcd_contiguous(position, num_batches);
    if (batch < 0) continue;
    __syncthreads();                           // the previous batch's LDS records are no longer read
    const int tile = batch_tile[batch];
    const int row0 = batch_row[batch];
    const long long tile_end = min(static_cast<long long>(tile + 1) * sh.tile_rows, static_cast<long long>(sh.num_rows));
    const int row1 = batch + 1 < num_batches && batch_tile[batch + 1] == tile ? batch_row[batch + 1]
                                                                                : static_cast<int>(tile_end);
    const long long entry0 = src.offset(row0), entry1 = src.offset(row1);
    for (int i = threadIdx.x; i < S; i += kBuildBlock) {
        place[i] = groups[static_cast<long long>(batch) * S + i];
        cell_begin[i] = cells_t[static_cast<long long>(tile) * S + i].x;     // (tile-major table: one contiguous read; the strip-major
                                                                          //  offsets sit num_tiles ints apart: a line per strip)
    }
    __syncthreads();
    for (long long j0 = entry0 + threadIdx.x; j0 < entry1; j0 += 4 * kBuildBlock) {
        int c[4];
        unsigned int m[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = j0 + static_cast<long long>(u) * kBuildBlock;
            c[u] = -1;
            if (j < entry1) {
                c[u] = src.col(j);
                m[u] = meta[j];
                v[u] = a_val ? src.val(j) : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c[u] < 0 || m[u] == kMetaSkip) continue;
            const int strip = c[u] >> sh.strip_shift;
            GroupPlace p;
            __builtin_memcpy(&p, &place[strip], sizeof(p));
            long long at = static_cast<long long>(cell_begin[strip]) + p.rel;
            int need, delta;
            if (m[u] & kMetaFirst) {
                need = p.lead;
                delta = static_cast<int>(m[u] & 0xFFFF) - p.prev_last - need * kSkip;
            } else {
                need = (m[u] >> 8) & 0xFF;
                delta = m[u] & 0xFF;
                at += p.lead + (m[u] >> 16) - need;
            }
            for (int k = 0; k < need; ++k) {
                if (a_val) a_val[at + k] = 0.0f;
                a_lcol[at + k] = 0;
                a_drow[at + k] = kSkip;
            }
            if (a_val) a_val[at + need] = v[u];
            a_lcol[at + need] = static_cast<unsigned short>(c[u] - (strip << sh.strip_shift));
            a_drow[at + need] = static_cast<unsigned char>(delta);
        }
    }
    }
}

// The same placing pass with the batch's slots assembled in LDS first, in destination order, so that the global
// stores leave the workgroup as contiguous segments (one per group and array) instead of one element per lane:
// the scattered form above issues ~3 partial-line writes per entry (C5: 480 M of them, 3.8 ms), this one ~3 per
// GROUP.  Takes the batches whose entries a workgroup can hold in registers (entry range <= capacity) and whose
// slots (markers included) fit the staging area; every other batch is left to batch_place_kernel (`todo` flag).
// Dynamic LDS: per strip place (8 B), cell begin, slot count, local offset (4 B each); per staged slot value f32,
// local column u16, strip u16, row delta u8.
constexpr int kStageBytesPerStrip = 20, kStageBytesPerSlot = 9;
template <typename Src>
__global__ __launch_bounds__(kBuildBlock, 8)
void batch_place_staged_kernel(Src src, BuildShape sh, int num_batches, int capacity, int stage_slots,
                               const int* __restrict__ batch_row, const int* __restrict__ batch_tile,
                               const uint2* __restrict__ groups,              // [batches * strips] GroupPlace
                               const unsigned int* __restrict__ meta, const int2* __restrict__ cells_t,
                               float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                               unsigned char* __restrict__ a_drow, int* __restrict__ todo /*[0] count, [1 ...] batches left over*/) {
    extern __shared__ int stage_lds[];
    __shared__ int s_partial[kBuildBlock / 64];
    const int batch = xcd_contiguous(blockIdx.x, num_batches);
    if (batch < 0) return;
    const int S = sh.num_strips;
    uint2* place = reinterpret_cast<uint2*>(stage_lds);                 // [S]
    int* cell_begin = stage_lds + 2 * S;                                 // [S]
    int* count = stage_lds + 3 * S;                                      // [S] slots of the batch per strip
    int* local = stage_lds + 4 * S;                                      // [S] first staged slot of the strip
    float* st_val = reinterpret_cast<float*>(stage_lds + 5 * S);         // [stage_slots]
    unsigned short* st_lcol = reinterpret_cast<unsigned short*>(st_val + stage_slots);
    unsigned short* st_strip = st_lcol + stage_slots;
    unsigned char* st_drow = reinterpret_cast<unsigned char*>(st_strip + stage_slots);

    const int tile = batch_tile[batch];
    const int row0 = batch_row[batch];
    const long long tile_end = min(static_cast<long long>(tile + 1) * sh.tile_rows, static_cast<long long>(sh.num_rows));
    const int row1 = batch + 1 < num_batches && batch_tile[batch + 1] == tile ? batch_row[batch + 1]
                                                                                : static_cast<int>(tile_end);
    const long long entry0 = src.offset(row0), entry1 = src.offset(row1);
    if (entry1 - entry0 > capacity) {          // (a batch with long rows inside: the scattered kernel takes it)
        if (threadIdx.x == 0) todo[1 + atomicAdd(&todo[0], 1)] = batch;
        return;
    }
    const int span = static_cast<int>(entry1 - entry0);
    for (int i = threadIdx.x; i < S; i += kBuildBlock) {
        place[i] = groups[static_cast<long long>(batch) * S + i];
        cell_begin[i] = cells_t[static_cast<long long>(tile) * S + i].x;     // (tile-major table: one contiguous read; the strip-major
                                                                          //  offsets sit num_tiles ints apart: a line per strip)
        count[i] = 0;
    }
    __syncthreads();

    // every thread keeps its entries: strip, local column, slots it needs (itself + the markers in front of it),
    // position inside its group
    int my_strip[kBuildPerThread], my_front[kBuildPerThread], my_need[kBuildPerThread], my_delta[kBuildPerThread];
    float my_val[kBuildPerThread];
    unsigned short my_lcol[kBuildPerThread];
#pragma unroll
    for (int u = 0; u < kBuildPerThread; ++u) {
        const int idx = threadIdx.x + u * kBuildBlock;
        my_strip[u] = -1;
        if (idx < span) {
            const long long j = entry0 + idx;
            const int c = src.col(j);
            const unsigned int m = meta[j];
            if (c >= 0 && m != kMetaSkip) {
                const int strip = c >> sh.strip_shift;
                GroupPlace p;
                __builtin_memcpy(&p, &place[strip], sizeof(p));
                my_strip[u] = strip;
                my_lcol[u] = static_cast<unsigned short>(c - (strip << sh.strip_shift));
                my_val[u] = a_val ? src.val(j) : 0.0f;
                if (m & kMetaFirst) {
                    my_need[u] = p.lead;
                    my_delta[u] = static_cast<int>(m & 0xFFFF) - p.prev_last - my_need[u] * kSkip;
                    my_front[u] = 0;
                } else {
                    my_need[u] = (m >> 8) & 0xFF;
                    my_delta[u] = m & 0xFF;
                    my_front[u] = p.lead + static_cast<int>(m >> 16) - my_need[u];
                }
                atomicAdd(&count[strip], 1 + my_need[u]);
            }
        }
    }
    __syncthreads();

    // exclusive scan of the per-strip slot counts (every thread owns a contiguous piece of the strips)
    int total;
    {
        const int per = (S + kBuildBlock - 1) / kBuildBlock;
        const int lo = min(S, per * static_cast<int>(threadIdx.x)), hi = min(S, lo + per);
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += count[i];
        int run = block_exclusive_scan<false>(sum, s_partial, &total);
        for (int i = lo; i < hi; ++i) {
            local[i] = run;
            run += count[i];
        }
    }
    __syncthreads();
    if (total > stage_slots) {                 // (markers galore: more slots than the staging area holds)
        if (threadIdx.x == 0) todo[1 + atomicAdd(&todo[0], 1)] = batch;
        return;
    }

    // assemble the batch's slots in destination order
#pragma unroll
    for (int u = 0; u < kBuildPerThread; ++u) {
        if (my_strip[u] < 0) continue;
        const int at = local[my_strip[u]] + my_front[u];
        for (int k = 0; k < my_need[u]; ++k) {
            st_val[at + k] = 0.0f;
            st_lcol[at + k] = 0;
            st_strip[at + k] = static_cast<unsigned short>(my_strip[u]);
            st_drow[at + k] = kSkip;
        }
        st_val[at + my_need[u]] = my_val[u];
        st_lcol[at + my_need[u]] = my_lcol[u];
        st_strip[at + my_need[u]] = static_cast<unsigned short>(my_strip[u]);
        st_drow[at + my_need[u]] = static_cast<unsigned char>(my_delta[u]);
    }
    __syncthreads();

    // ... and write them out: consecutive lanes, consecutive slots of a group, consecutive addresses
    for (int u = threadIdx.x; u < total; u += kBuildBlock) {
        const int strip = st_strip[u];
        GroupPlace p;
        __builtin_memcpy(&p, &place[strip], sizeof(p));
        const long long at = static_cast<long long>(cell_begin[strip]) + p.rel + (u - local[strip]);
        if (a_val) a_val[at] = st_val[u];
        a_lcol[at] = st_lcol[u];
        a_drow[at] = st_drow[u];
    }
}

// One thread per cell (tile, strip): walks the tile's batches in row order, places every group inside the
// cell (markers between groups included) and records the cell's slot count.
__global__ __launch_bounds__(kBlock)
void cell_place_kernel(int num_tiles, int num_strips, const int* __restrict__ tile_batch,
                       uint2* __restrict__ groups, int* __restrict__ cell_slots /*strip-major*/,
                       unsigned long long* __restrict__ entry_total) {
    const long long id = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
    unsigned long long mine = 0;
    if (id < static_cast<long long>(num_tiles) * num_strips) {
        const int tile = static_cast<int>(id / num_strips), strip = static_cast<int>(id % num_strips);
        int last = 0;
        unsigned int total = 0;
        const int b_end = tile_batch[tile + 1];
        uint2 ahead = make_uint2(0, 0);       // the next batch's record is fetched before this one's is rewritten (other addresses)
        if (tile_batch[tile] < b_end) ahead = groups[static_cast<long long>(tile_batch[tile]) * num_strips + strip];
        for (int b = tile_batch[tile]; b < b_end; ++b) {
            uint2* slot = groups + static_cast<long long>(b) * num_strips + strip;
            const uint2 raw = ahead;
            if (b + 1 < b_end) ahead = groups[static_cast<long long>(b + 1) * num_strips + strip];
            GroupCount g;
            __builtin_memcpy(&g, &raw, sizeof(g));
            GroupPlace p;
            p.rel = total;
            p.prev_last = static_cast<unsigned short>(last);
            p.lead = 0;
            if (g.count) {
                p.lead = static_cast<unsigned short>((g.first - last) / kSkip);
                total += g.count + g.escapes + p.lead;
                last = g.last;
                mine += g.count;
            }
            uint2 packed;
            __builtin_memcpy(&packed, &p, sizeof(p));
            *slot = packed;
        }
        cell_slots[static_cast<long long>(strip) * num_tiles + tile] = static_cast<int>(total);
    }
    // one atomic per workgroup (ten thousand wavefronts adding to one address serialise at the memory side)
    __shared__ unsigned long long s_mine[kBlock / 64];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_mine[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int w = 0; w < kBlock / 64; ++w) all += s_mine[w];
        if (all) atomicAdd(entry_total, all);
    }
}

// exclusive scan of round_up_4(in[i]) in three launches: block sums, scan of the sums, block scans
constexpr int kScanBlock = 1024, kScanPerThread = 4, kScanTile = kScanBlock * kScanPerThread;
__device__ __forceinline__ int padded4(int v) { return (v + 3) & ~3; }

__global__ __launch_bounds__(kScanBlock)
void scan_sums_kernel(const int* __restrict__ in, long long n, long long* __restrict__ block_sum) {
    __shared__ long long s_wave[kScanBlock / 64];
    const long long first = static_cast<long long>(blockIdx.x) * kScanTile + threadIdx.x * kScanPerThread;
    long long sum = 0;
    for (int k = 0; k < kScanPerThread; ++k) if (first + k < n) sum += padded4(in[first + k]);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long total = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) total += s_wave[w];
        block_sum[blockIdx.x] = total;
    }
}

__global__ __launch_bounds__(kScanBlock)
void scan_top_kernel(long long* __restrict__ block_sum, int blocks, long long* __restrict__ grand_total) {
    __shared__ long long s_part[kScanBlock];
    const int per = (blocks + kScanBlock - 1) / kScanBlock;
    const int lo = min(blocks, per * static_cast<int>(threadIdx.x)), hi = min(blocks, lo + per);
    long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_sum[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const long long add = static_cast<int>(threadIdx.x) >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        const long long v = block_sum[i];
        block_sum[i] = run;
        run += v;
    }
    if (threadIdx.x == kScanBlock - 1) *grand_total = s_part[kScanBlock - 1];
}

__global__ __launch_bounds__(kScanBlock)
void scan_apply_kernel(const int* __restrict__ in, long long n, const long long* __restrict__ block_sum,
                       int* __restrict__ out /*[n + 1]*/) {
    __shared__ int s_part[kScanBlock];
    const long long first = static_cast<long long>(blockIdx.x) * kScanTile + threadIdx.x * kScanPerThread;
    int v[kScanPerThread], sum = 0;
    for (int k = 0; k < kScanPerThread; ++k) {
        v[k] = first + k < n ? padded4(in[first + k]) : 0;
        sum += v[k];
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const int add = static_cast<int>(threadIdx.x) >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = block_sum[blockIdx.x] + (threadIdx.x ? s_part[threadIdx.x - 1] : 0);
    for (int k = 0; k < kScanPerThread; ++k) {
        if (first + k < n) out[first + k] = static_cast<int>(run);
        run += v[k];
        if (first + k == n - 1) out[n] = static_cast<int>(run);
    }
}

// counts[0 .. n) -> exclusive prefix sums in place, counts[n] = total.  One workgroup, each thread a contiguous piece.
__global__ __launch_bounds__(kScanBlock)
void exclusive_scan_small_kernel(int* __restrict__ counts, int n) {
    __shared__ long long s_part[kScanBlock];
    const int per = (n + kScanBlock - 1) / kScanBlock;
    const int lo = min(n, per * static_cast<int>(threadIdx.x)), hi = min(n, lo + per);
    long long sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const long long add = static_cast<int>(threadIdx.x) >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = threadIdx.x ? s_part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        const int v = counts[i];
        counts[i] = static_cast<int>(run);
        run += v;
    }
    if (threadIdx.x == kScanBlock - 1) counts[n] = static_cast<int>(s_part[kScanBlock - 1]);
}

// the padding slots at the end of every cell (and nothing else): skip markers
__global__ __launch_bounds__(kBlock)
void cell_padding_kernel(const int* __restrict__ cell_slots, const int* __restrict__ offs, long long cells,
                         float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                         unsigned char* __restrict__ a_drow) {
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < cells;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        const int used = cell_slots[i];
        for (int k = used; k < padded4(used); ++k) {
            const long long at = static_cast<long long>(offs[i]) + k;
            if (a_val) a_val[at] = 0.0f;
            a_lcol[at] = 0;
            a_drow[at] = kSkip;
        }
    }
}

// Column-weight folding: when every stored entry of a column carries the same value
// (adjacency matrices, the column-stochastic matrices of PageRank: a_ij = 1 / outdeg(j)),
// a_ij * x_j = (w_j * x_j) is one product per column instead of one per entry, and phase 1 no
// longer needs the value stream.  One workgroup per strip, the strip's weights in LDS: first every
// entry stores its value at its column, then every entry compares its bits with what stayed there.
// Skip markers / padding (row delta 255) carry no value.  Columns without an entry in the cells keep the
// kNoWeight bit pattern until the long rows have had their say (weight_finish_kernel turns what is left into 0).
constexpr unsigned int kNoWeight = 0x7FC0BEEFu;       // a NaN payload no arithmetic produces
template <int W>
__global__ __launch_bounds__(1024)
void strip_weight_kernel(int first_strip, int slot_limit, const int* __restrict__ strip_begin, int num_cols,
                         const float* __restrict__ a_val, const unsigned short* __restrict__ a_lcol,
                         const unsigned char* __restrict__ a_drow,
                         float* __restrict__ weight, int* __restrict__ differs) {
    __shared__ float ws[W];
    const int strip = first_strip + blockIdx.x;
    const int begin = strip_begin[strip];
    const int end = static_cast<int>(min(static_cast<long long>(strip_begin[strip + 1]), static_cast<long long>(begin) + slot_limit));
    for (int i = threadIdx.x; i < W; i += 1024) ws[i] = __uint_as_float(kNoWeight);
    __syncthreads();
    for (int q = begin + threadIdx.x; q < end; q += 1024) {
        if (a_drow[q] != kSkip) ws[a_lcol[q]] = a_val[q];
    }
    __syncthreads();
    bool bad = false;
    for (int q = begin + threadIdx.x; q < end; q += 1024) {
        if (a_drow[q] != kSkip) bad |= __float_as_uint(ws[a_lcol[q]]) != __float_as_uint(a_val[q]);
    }
    if (bad) *differs = 1;
    if (!weight) return;                 // sampling round: only the verdict is wanted
    const long long base = static_cast<long long>(strip) * W;
    for (int i = threadIdx.x; i < W && base + i < num_cols; i += 1024) weight[base + i] = ws[i];
}

// the long rows' entries are not in the cells: PASS 0 gives columns that only they touch a weight,
// PASS 1 checks that every long-row entry carries its column's weight
template <int PASS>
__global__ __launch_bounds__(kBlock)
void long_row_weight_kernel(const int* __restrict__ chunks, int num_chunks, const int* __restrict__ cols,
                            const float* __restrict__ vals, float* __restrict__ weight,
                            int* __restrict__ differs) {
    const int which = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (which >= num_chunks) return;
    bool bad = false;
    for (int j = chunks[3 * which + 1] + (threadIdx.x & 63); j < chunks[3 * which + 2]; j += 64) {
        unsigned int* slot = reinterpret_cast<unsigned int*>(weight + cols[j]);
        if (PASS == 0) {
            if (*slot == kNoWeight) atomicCAS(slot, kNoWeight, __float_as_uint(vals[j]));
        } else {
            bad |= *slot != __float_as_uint(vals[j]);
        }
    }
    if (PASS == 1 && bad) *differs = 1;
}

__global__ __launch_bounds__(kBlock)
void weight_finish_kernel(float* __restrict__ weight, int n) {
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        if (__float_as_uint(weight[i]) == kNoWeight) weight[i] = 0.0f;
    }
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

// ------------------------------------------------------------------------ phase 1 ----
// Rows too long for the cells are cut into chunks of kLongChunk entries; one wavefront per
// chunk sums it by direct gather into its own slot; phase 2's tile start then adds a long row's chunk sums in
// chunk order (no atomics, no state between calls: the same bits on every run).  The chunk wavefronts ride in
// extra workgroups at the head of the phase-1 grid, so they overlap the expansion at no launch cost.
struct LongRows {
    const int* chunks;        // (row, begin, end) triples over the CSR arrays
    int num_chunks;
    long long nnz;
    const int* cols;
    const float* vals;
    float* chunk_sum;         // [num_chunks] one partial sum per chunk (no atomics: long_rows_finish_kernel adds them in order)
};

__device__ __forceinline__ void long_row_chunk(const LongRows& lr, int which, const float* __restrict__ x) {
    if (which >= lr.num_chunks) return;
    float acc = row_partial_dot<64>(lr.chunks[3 * which + 1], lr.chunks[3 * which + 2], threadIdx.x & 63, lr.nnz,
                                    lr.cols, lr.vals, x);
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) lr.chunk_sum[which] = acc;
}

// Stages x strip `strip` (W columns from `base`) into xs.  FOLD: as w_j * x_j (one rounded product per column).
// Whole rounds of the workgroup first, several 16-byte loads per lane in flight.  One load per iteration behind its own wait
// made a W = 16384 strip eight L2 latencies per item, ~20 % of a workgroup's life during which it streams nothing: C5
// phase 1 322-328 -> 310-312 us with four in flight (two: 313-317; eight, or the entry loop software-pipelined on top:
// no further change — profiles/r04_kernel_ab_descriptors.txt).
template <int W, int kExpandBlock, bool FOLD>
__device__ __forceinline__ void stage_strip(float* xs, const float* __restrict__ x, const float* __restrict__ col_weight,
                                            long long base, int num_cols) {
    const int width = static_cast<int>(min(static_cast<long long>(W), num_cols - base));
    const float* src = x + base;
    constexpr int kRound = kExpandBlock * 4;
    constexpr int kInFlight = W / kRound >= 4 ? 4 : (W / kRound >= 2 ? 2 : 1);
    if (FOLD) {
        const float* wsrc = col_weight + base;            // hipMalloc'd and base % 4 == 0: always aligned
        const bool aligned = (reinterpret_cast<unsigned long long>(src) & 15) == 0;
        int i_block = 0;                                   // wave-uniform
        if (aligned) {
#pragma unroll 1
            for (; i_block + kInFlight * kRound <= width; i_block += kInFlight * kRound) {
                const int i = i_block + threadIdx.x * 4;
                f32x4 xv[kInFlight], wv[kInFlight];
#pragma unroll
                for (int u = 0; u < kInFlight; ++u) {
                    xv[u] = *reinterpret_cast<const f32x4*>(src + i + u * kRound);
                    wv[u] = *reinterpret_cast<const f32x4*>(wsrc + i + u * kRound);
                }
#pragma unroll
                for (int u = 0; u < kInFlight; ++u) {
                    f32x4 z;
                    z[0] = __fmul_rn(wv[u][0], xv[u][0]);
                    z[1] = __fmul_rn(wv[u][1], xv[u][1]);
                    z[2] = __fmul_rn(wv[u][2], xv[u][2]);
                    z[3] = __fmul_rn(wv[u][3], xv[u][3]);
                    *reinterpret_cast<f32x4*>(xs + i + u * kRound) = z;
                }
            }
        }
        for (int i = i_block + threadIdx.x * 4; i < width; i += kRound) {
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
        int i_block = 0;                                   // wave-uniform
#pragma unroll 1
        for (; i_block + kInFlight * kRound <= width; i_block += kInFlight * kRound) {
            const int i = i_block + threadIdx.x * 4;
            f32x4 t[kInFlight];
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + i + u * kRound);
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) *reinterpret_cast<f32x4*>(xs + i + u * kRound) = t[u];
        }
        for (int i = i_block + threadIdx.x * 4; i < width; i += kRound) {
            if (i + 3 < width) {
                *reinterpret_cast<f32x4*>(xs + i) = *reinterpret_cast<const f32x4*>(src + i);
            } else {
                for (int k = i; k < width; ++k) xs[k] = src[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < width; i += kExpandBlock) xs[i] = src[i];
    }
}

// The products of the slots [begin, end) of the staged strip: value, local column -> value * xs[column], same order.
template <int kExpandBlock, bool FOLD>
__device__ __forceinline__ void expand_slots(const float* xs, int begin, int end, const float* __restrict__ a_val,
                                             const unsigned short* __restrict__ a_lcol, float* __restrict__ prod) {
    constexpr int kStride = kExpandBlock * 4;
    if (FOLD) {
        // four entries per lane per group, two groups a workgroup-stride apart per step: every store instruction writes
        // one contiguous KB per wavefront (round 3's eight consecutive entries per lane made each instruction write every
        // other 16 bytes; harmless with plain stores, which meet in L2, but 31 % more write traffic with the non-temporal
        // ones: WRITE_SIZE 826 against 631 MB on C5)
        for (int q = (begin & ~3) + threadIdx.x * 4; q < end; q += 2 * kStride) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int g = q + u * kStride;
                if (g >= begin && g + 3 < end) {
                    const u16x4 c = *reinterpret_cast<const u16x4*>(a_lcol + g);
                    f32x4 p;
                    p[0] = xs[c[0]]; p[1] = xs[c[1]]; p[2] = xs[c[2]]; p[3] = xs[c[3]];
                    store_product(reinterpret_cast<f32x4*>(prod + g), p);
                } else {
                    for (int k = max(g, begin); k < min(g + 4, end); ++k) prod[k] = xs[a_lcol[k]];
                }
            }
        }
        return;
    }
    // four entries per lane per step, groups aligned to 4 entries (16-byte loads and stores).  (Round 4 tried 2 / 4 / 8 groups
    // per lane with all loads issued up front, and the next group's loads in flight while this one is multiplied: no gain,
    // the bunched forms lose 1-2 % — profiles/r04_kernel_ab_descriptors.txt; once a workgroup streams, phase 1 runs at the
    // rate its 10 bytes per slot allow.  What paid was the start of a workgroup's life: the staging.)
    for (int q = (begin & ~3) + threadIdx.x * 4; q < end; q += kStride) {
        if (q >= begin && q + 3 < end) {
            const u16x4 c = *reinterpret_cast<const u16x4*>(a_lcol + q);
            const f32x4 v = *reinterpret_cast<const f32x4*>(a_val + q);
            f32x4 p;
            p[0] = v[0] * xs[c[0]];
            p[1] = v[1] * xs[c[1]];
            p[2] = v[2] * xs[c[2]];
            p[3] = v[3] * xs[c[3]];
            store_product(reinterpret_cast<f32x4*>(prod + q), p);
        } else {
            for (int k = max(q, begin); k < min(q + 4, end); ++k) prod[k] = a_val[k] * xs[a_lcol[k]];
        }
    }
}

// FOLD: the plan holds one weight per column instead of a value per entry; the strip is staged
// as w_j * x_j and an entry's product is a plain LDS read (the same rounded product as a_ij * x_j).
// One workgroup per work item.  (Round 4 tried 2 / 3 / 5 CONSECUTIVE items per workgroup, staging a strip only when it changes:
// 347 / 359 / 433 against 320 us on C5 — fewer, longer workgroups balance worse than their saved strip loads are worth.)
template <int W, int kExpandBlock, bool FOLD>
__global__ __launch_bounds__(kExpandBlock)
void tiled_expand_kernel(const int* __restrict__ items, int first_item, int num_items, int long_blocks,
                         const float* __restrict__ a_val,
                         const unsigned short* __restrict__ a_lcol,
                         const float* __restrict__ col_weight,
                         const float* __restrict__ x, int num_cols,
                         float* __restrict__ prod, LongRows long_rows,
                         const PrState* __restrict__ state) {
    // PageRank steps enqueued past convergence are no-ops
    if (state && state->done) return;
    if (static_cast<int>(blockIdx.x) < long_blocks) {     // the long-row workgroups go first (latency-bound)
        constexpr int kPerBlock = kExpandBlock / 64;
        long_row_chunk(long_rows, blockIdx.x * kPerBlock + (threadIdx.x >> 6), x);
        return;
    }
    __shared__ float xs[W];
    const int window = xcd_contiguous(blockIdx.x - long_blocks, num_items);   // long_blocks is a multiple of 8
    if (window < 0) return;
    const int item = first_item + window;
    const int strip = items[3 * item];
    const int begin = items[3 * item + 1];
    const int end = items[3 * item + 2];
    stage_strip<W, kExpandBlock, FOLD>(xs, x, col_weight, static_cast<long long>(strip) * W, num_cols);
    __syncthreads();
    expand_slots<kExpandBlock, FOLD>(xs, begin, end, a_val, a_lcol, prod);
}

// ------------------------------------------------------------------------ phase 2 ----
// inclusive prefix sum over the 64 lanes of a wavefront: Hillis-Steele inside each 16-lane DPP row
// (row_shr 1, 2, 4, 8; lanes shifted in from outside the row contribute 0), then the classic wave64 tail:
// row_bcast:15 adds lane 15 of the previous row into rows 1 and 3, row_bcast:31 adds lane 31 into rows 2, 3
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// the long rows (direct path): which of them fall into which tile, and where their chunk sums are
struct LongSeeds {
    const int* rows;          // [num_long] ascending
    const int* first_chunk;   // [num_long + 1]
    const int* tile_first;    // [num_tiles + 1] first long row of every tile (null: no long rows)
    const float* chunk_sum;   // [num_chunks] written by phase 1 of this SpMV
};

// ---- phase 2 as a list of PASSES laid out when the plan is built (round 4) ----
// A tile's slots are one run per strip (cell table); a wavefront of the tile's workgroup owns a contiguous share of the
// strips and walks its runs as ONE stream of slots, 256 per PASS whatever the run boundaries: a pass is up to kPassSegs
// SEGMENTS (the end of one run, whole short runs, the start of the next), 4 slots per lane.  Round 3 derived the segments
// inside the kernel, from the (begin, length) records of the runs — ~100 scalar instructions and ~18 branches per pass,
// and with 8 wavefronts per SIMD sharing one scalar issue slot every 4 cycles that was the kernel's bound: a phase 2 that
// only LOADS (no row rebuild, no LDS adds) took 192 us of the full kernel's 200, whatever the access pattern, the loads in
// flight or the source array (profiles/r04_phase2_bound.txt).  Everything the scalar code computed is a function of the
// plan alone, so it is computed ONCE, by pass_layout_kernel below, into one 32-byte descriptor per pass:
//     base[k]  slot index lane 0 WOULD read if it belonged to segment k (a lane reads base + 4 * lane)
//     adj[k]   row of the slot in front of segment k's first slot (0 at a run's start) minus the sum of the delta bytes
//              of all lanes in front of the segment: row of a slot = adj[segment] + (wave-wide exclusive prefix of the lanes'
//              delta sums) + the in-lane prefix — no carry from pass to pass, nothing read back from the scan
//     geom     first lane of segments 1 and 2, lanes in use
// The hot loop is then: 8 v_readlane per pass, two loads, one wave scan, four ds_add_f64 — and passes are independent of
// each other, so the next ones' loads are always in flight.  Same slots, same rows, same fp64 adds as the round-3 form:
// bit-identical results (tests/tiled_small_shapes_worker.py: ~150 awkward shapes and the hand-picked run-length patterns —
// every boundary case of a pass — against the oracle).
constexpr int kPassSegs = 3;            // segments per pass: 2 / 3 / 4 measured 492 / 482 / 483 us on C5 in round 3
constexpr int kPassSlots = 256;         // slots per pass: four per lane
constexpr int kPassDepth = 3;           // passes in flight per wavefront (2 / 3 / 4: 158.7 / 157.0 / 156.9 us on C5)
constexpr int kReduceThreads = 1024;
constexpr int kReduceWaves = kReduceThreads / 64;

struct PassDesc {                       // 32 bytes = two 16-byte loads
    int base[kPassSegs];
    int adj[kPassSegs];
    unsigned int geom;                  // start[1] | start[2] << 8 | lanes in use << 16
    unsigned int reserved;
};
static_assert(sizeof(PassDesc) == 32, "a lane loads its pass descriptor as two 16-byte words");

// the strips (= runs of a tile) wavefront `wave` of a tile's workgroup owns
__device__ __forceinline__ void wave_runs(int num_strips, int wave, int* lo, int* hi) {
    const int per_wave = (num_strips + kReduceWaves - 1) / kReduceWaves;
    *lo = min(num_strips, wave * per_wave);
    *hi = min(num_strips, *lo + per_wave);
}

// Lays out the passes of every (tile, wavefront): FILL = false counts them (-> pass_count[tile * 16 + wave]), FILL = true
// writes their descriptors at pass_first[tile * 16 + wave].  One 1024-thread workgroup per tile, wavefront w does the share
// of wavefront w of the phase-2 workgroup.  The (begin, length) records of a wavefront's runs come 64 at a time (lane l holds
// run window_first + l, read back with v_readlane); a pass never straddles two such windows.
template <bool FILL>
__global__ __launch_bounds__(kReduceThreads)
void pass_layout_kernel(int num_tiles, int num_strips, const int2* __restrict__ cells_t,
                        const unsigned char* __restrict__ a_drow,
                        int* __restrict__ pass_count, const int* __restrict__ pass_first, PassDesc* __restrict__ desc) {
    const int tile_index = blockIdx.x;
    if (tile_index >= num_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    int run_lo, run_hi;
    wave_runs(num_strips, wave, &run_lo, &run_hi);
    const int2* mine = cells_t + static_cast<long long>(tile_index) * num_strips;
    int passes = 0;
    long long out = FILL ? pass_first[tile_index * kReduceWaves + wave] : 0;
    for (int window_first = run_lo; window_first < run_hi; window_first += 64) {
        const int2 window = window_first + lane < run_hi ? mine[window_first + lane] : make_int2(0, 0);
        const int window_runs = min(64, run_hi - window_first);
        int next_run = 0, cur_begin = 0, cur_len = 0, off = 0;     // the stream cursor (wave-uniform)
        int row_carry = 0;                                         // the row the open run has reached
        for (;;) {
            int base[kPassSegs], start[kPassSegs];
            int filled = 0, last = 0;
            bool fresh0 = true;
#pragma unroll
            for (int k = 0; k < kPassSegs; ++k) {
                while (off >= cur_len && next_run < window_runs) {       // the next run that holds slots (banded matrices: most are empty)
                    cur_begin = __builtin_amdgcn_readlane(window.x, next_run);
                    cur_len = __builtin_amdgcn_readlane(window.y, next_run);
                    off = 0;
                    ++next_run;
                }
                const int take = max(min(cur_len - off, kPassSlots - filled), 0);
                if (k == 0) fresh0 = off == 0;
                base[k] = cur_begin + off - filled;
                start[k] = filled >> 2;
                last = take > 0 ? k : last;
                filled += take;
                off += take;
            }
            if (filled == 0) break;
            ++passes;
            if (!FILL) continue;
            const bool open_end = off < cur_len;
            const int groups = filled >> 2;
            // the delta bytes of the pass: what phase 2 will read, lane by lane
            const int at = min(lane, groups - 1);
            int mine_base = base[0];
#pragma unroll
            for (int k = 1; k < kPassSegs; ++k) mine_base = at >= start[k] ? base[k] : mine_base;
            const unsigned int word = lane < groups ? *reinterpret_cast<const unsigned int*>(a_drow + mine_base + 4 * at) : 0u;
            const int sum = static_cast<int>((word & 0xFF) + ((word >> 8) & 0xFF) + ((word >> 16) & 0xFF) + (word >> 24));
            const int incl = wave_inclusive_scan(sum);
            // row in front of each segment, and the delta sums in front of it
            int adj[kPassSegs];
            int origin = fresh0 ? 0 : row_carry;
            adj[0] = origin;
#pragma unroll
            for (int k = 1; k < kPassSegs; ++k) {
                const int before = start[k] > 0 ? __builtin_amdgcn_readlane(incl, max(start[k] - 1, 0)) : 0;
                adj[k] = -before;                          // a segment behind the first starts a run: its rows count from 0
                origin = k <= last ? adj[k] : origin;
            }
            row_carry = open_end ? origin + __builtin_amdgcn_readlane(incl, groups - 1) : 0;
            if (lane == 0) {
                PassDesc d;
#pragma unroll
                for (int k = 0; k < kPassSegs; ++k) {
                    d.base[k] = base[k];
                    d.adj[k] = adj[k];
                }
                d.geom = static_cast<unsigned int>(start[1]) | static_cast<unsigned int>(start[2]) << 8 | static_cast<unsigned int>(groups) << 16;
                d.reserved = 0;
                desc[out] = d;
            }
            ++out;
        }
    }
    if (!FILL && lane == 0) pass_count[tile_index * kReduceWaves + wave] = passes;
}

// Fills the LDS tile with the sums of this tile's rows; the long rows' sums come from their chunk sums.  The tile
// accumulates in DOUBLE: every fp32 product is added exactly as often as fp64 allows (products of one row rarely span
// more than 29 binades), so the row sums do not depend on the order in which the wavefronts' adds meet, and they are
// rounded to fp32 once, on the way out.
__device__ __forceinline__ void tile_accumulate(double* tile, int R, int tile_index,
                                                const int* __restrict__ pass_first, const PassDesc* __restrict__ desc,
                                                const float* __restrict__ prod,
                                                const unsigned char* __restrict__ a_drow,
                                                const LongSeeds seeds) {
    __shared__ double spare[64];           // where a lane's slots without an entry "add" (never read)
    const long long first = static_cast<long long>(tile_index) * R;
    for (int i = threadIdx.x; i < R; i += kReduceThreads) tile[i] = 0.0;
    if (seeds.tile_first) {
        __syncthreads();
        for (int k = seeds.tile_first[tile_index] + threadIdx.x; k < seeds.tile_first[tile_index + 1]; k += kReduceThreads) {
            double total = 0.0;                         // a long row's chunk sums, in chunk order
            for (int c = seeds.first_chunk[k]; c < seeds.first_chunk[k + 1]; ++c) total += static_cast<double>(seeds.chunk_sum[c]);
            tile[seeds.rows[k] - first] = total;
        }
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int pass_lo = pass_first[tile_index * kReduceWaves + wave], pass_hi = pass_first[tile_index * kReduceWaves + wave + 1];

    struct Pass {
        int lane_adj;                      // adj of the lane's segment (selected when the pass is opened: a select over struct
        int groups;                        //  members kept for later turns into an indexed load from scratch memory)
        f32x4 p;
        unsigned int d;
    };
    // descriptors 64 at a time: lane l holds pass window_first + l
    for (int window_first = pass_lo; window_first < pass_hi; window_first += 64) {
        const uint4* mine = reinterpret_cast<const uint4*>(desc + window_first + min(lane, pass_hi - window_first - 1));
        uint4 lo = mine[0], hi = mine[1];
        // (waited for HERE, once: read for the first time inside the loop below, the compiler would wait vmcnt(0) — the
        // passes' own loads included — at every pass it opens)
        asm volatile("; pass descriptors settled" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(lo.w), "+v"(hi.x), "+v"(hi.y), "+v"(hi.z));
        const int window_passes = min(64, pass_hi - window_first);

        auto open = [&](Pass& ps, int k) {            // pass k of the window (k < window_passes): geometry + its two loads
            const int idx = __builtin_amdgcn_readfirstlane(k);
            const int base0 = __builtin_amdgcn_readlane(static_cast<int>(lo.x), idx);
            const int base1 = __builtin_amdgcn_readlane(static_cast<int>(lo.y), idx);
            const int base2 = __builtin_amdgcn_readlane(static_cast<int>(lo.z), idx);
            const int adj0 = __builtin_amdgcn_readlane(static_cast<int>(lo.w), idx);
            const int adj1 = __builtin_amdgcn_readlane(static_cast<int>(hi.x), idx);
            const int adj2 = __builtin_amdgcn_readlane(static_cast<int>(hi.y), idx);
            const unsigned int geom = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(hi.z), idx));
            const int start1 = static_cast<int>(geom & 0xFF), start2 = static_cast<int>((geom >> 8) & 0xFF);
            ps.groups = static_cast<int>(geom >> 16);
            // lanes past the pass's end re-read its last group (same cache line) and are masked in add()
            const int at = min(lane, ps.groups - 1);
            int base = base0, adj = adj0;
            base = at >= start1 ? base1 : base;
            adj = at >= start1 ? adj1 : adj;
            base = at >= start2 ? base2 : base;
            adj = at >= start2 ? adj2 : adj;
            ps.lane_adj = adj;
            const unsigned int slot = static_cast<unsigned int>(base + 4 * at);
            ps.p = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(prod) + (static_cast<size_t>(slot) << 2));
            ps.d = *reinterpret_cast<const unsigned int*>(a_drow + slot);
        };
        auto add = [&](const Pass& ps) {
            const unsigned int word = lane < ps.groups ? ps.d : 0xFFFFFFFFu;
            int delta[4], upto[4], sum = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                delta[e] = (word >> (8 * e)) & 0xFF;
                sum += delta[e];
                upto[e] = sum;
            }
            const int incl = wave_inclusive_scan(sum);
            const int lane_base = ps.lane_adj + incl - sum;
            // One LDS atomic per slot: ds_add_f64 (no return value, no retry loop, equal rows in one instruction are the
            // hardware's business; gfx950 runs it at 3.5 lanes/clk/CU on random rows, the fp32 form at 0.38 —
            // tools/lds_bench.hip).  Skip markers aim at a per-lane spare word, so nothing here needs an execution mask.
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double* target = delta[e] != kSkip ? &tile[lane_base + upto[e]] : &spare[lane];
                atomicAdd(target, static_cast<double>(ps.p[e]));
            }
        };

        // kPassDepth passes in flight, pass i of the window in slot i mod kPassDepth.  The steady-state loop holds nothing
        // but the pipeline (every add is followed by an open: no branch around a load, so the compiler's s_waitcnt in front
        // of an add counts exactly the newer passes' loads); what is left at the end is drained by the two short tails.
        Pass ps[kPassDepth];
#pragma unroll
        for (int u = 0; u < kPassDepth; ++u) {
            // (unconditional — a window shorter than the pipeline re-opens its last pass and never adds it —: with branches
            // here the loads in flight differ from path to path and the first wait of the loop below becomes vmcnt(0))
            open(ps[u], min(u, window_passes - 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        int k = 0;
        for (; k + 2 * kPassDepth <= window_passes; k += kPassDepth) {
#pragma unroll
            for (int u = 0; u < kPassDepth; ++u) {
                // (scheduling barriers: left alone, the compiler gathers the three adds behind ONE s_waitcnt vmcnt(0) and issues
                // all six loads at the end of the iteration — nothing in flight while a pass is added)
                add(ps[u]);
                __builtin_amdgcn_sched_barrier(0);
                open(ps[u], k + u + kPassDepth);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < kPassDepth; ++u) {
            if (k + u < window_passes) {
                add(ps[u]);
                if (k + u + kPassDepth < window_passes) open(ps[u], k + u + kPassDepth);
            }
        }
#pragma unroll
        for (int u = 0; u < kPassDepth; ++u) {
            if (k + kPassDepth + u < window_passes) add(ps[u]);
        }
    }
    __syncthreads();
}

// The tile (R doubles, R = plan.tile_rows: any multiple of 64) lives in dynamic LDS.
template <int kReduceBlock>
__global__ __launch_bounds__(kReduceBlock, kReduceBlock / 128)     // two tiles per CU: 8 wavefronts per SIMD
void tiled_reduce_kernel(int R, int num_tiles, const int* __restrict__ pass_first, const PassDesc* __restrict__ pass_desc,
                         const float* __restrict__ prod,
                         const unsigned char* __restrict__ a_drow,
                         const LongSeeds seeds,
                         int num_rows, float* __restrict__ y) {
    static_assert(kReduceBlock == kReduceThreads, "the pass layout is made for this workgroup size");
    extern __shared__ double tile[];
    const int tile_index = xcd_contiguous(blockIdx.x, num_tiles);
    if (tile_index < 0) return;
    tile_accumulate(tile, R, tile_index, pass_first, pass_desc, prod, a_drow, seeds);
    const long long first = static_cast<long long>(tile_index) * R;
    for (int i = threadIdx.x; i < R && first + i < num_rows; i += kReduceBlock) y[first + i] = static_cast<float>(tile[i]);
}

// phase 2 with the PageRank update fused into the tile write-out (cf. pr_step_kernel)
template <int kReduceBlock>
__global__ __launch_bounds__(kReduceBlock, kReduceBlock / 128)
void tiled_pagerank_reduce_kernel(int R, int num_tiles, const int* __restrict__ pass_first, const PassDesc* __restrict__ pass_desc,
                                  const float* __restrict__ prod,
                                  const unsigned char* __restrict__ a_drow,
                                  const LongSeeds seeds,
                                  int local_rows, RowMap map, int n_global,
                                  const float* __restrict__ r_old, float* __restrict__ r_new,
                                  const unsigned char* __restrict__ dangling, float damping,
                                  const PrState* __restrict__ state,
                                  double* __restrict__ block_partials, PushTargets push) {
    if (state->done) return;
    extern __shared__ double tile[];
    const int tile_index = xcd_contiguous(blockIdx.x, num_tiles);
    if (tile_index < 0) return;
    tile_accumulate(tile, R, tile_index, pass_first, pass_desc, prod, a_drow, seeds);

    const float teleport = __fdiv_rn(1.0f - damping, static_cast<float>(n_global));
    const float dangling_term = __fdiv_rn(__fmul_rn(damping, state->dangling_sum),
                                          static_cast<float>(n_global));
    double res2 = 0.0, mass = 0.0;
    const long long first = static_cast<long long>(tile_index) * R;
    if (map.piece == 0x7fffffff && push.count == 0) {
        // The usual case (one contiguous slice, no peer stores).  The update needs r_old and the dangling flag of the tile's
        // rows: read inside the loop they were one full memory latency per row and thread, ten in a row (the loop below waits
        // vmcnt(0) in every iteration: +22 us per step on C5, 182 against 160 us for the plain reduce); here all of a
        // thread's loads go out together, in front of the arithmetic.  Same operations in the same order: same bits.
        constexpr int kPerThread = (kMaxTileRows + kReduceBlock - 1) / kReduceBlock;
        const long long node_first = map.base + first;
        const int rows_here = static_cast<int>(min(static_cast<long long>(R), local_rows - first));     // >= 1: the tile exists
        float old_rank[kPerThread];
        unsigned char is_dangling[kPerThread];
        // (unconditional loads at clamped rows: a guarded load is a branch, and the compiler waits for each behind its join)
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = min(static_cast<int>(threadIdx.x) + u * kReduceBlock, rows_here - 1);
            old_rank[u] = r_old[node_first + i];
            is_dangling[u] = dangling[node_first + i];
        }
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = threadIdx.x + u * kReduceBlock;
            if (i < rows_here) {
                const float fresh = __fadd_rn(__fadd_rn(__fmul_rn(damping, static_cast<float>(tile[i])), dangling_term), teleport);
                r_new[node_first + i] = fresh;
                const float diff = __fsub_rn(fresh, old_rank[u]);
                res2 += static_cast<double>(__fmul_rn(diff, diff));
                if (is_dangling[u]) mass += static_cast<double>(fresh);
            }
        }
    } else {
        for (int i = threadIdx.x; i < R && first + i < local_rows; i += kReduceBlock) {
            const long long node = map.at(first + i);
            const float fresh = __fadd_rn(__fadd_rn(__fmul_rn(damping, static_cast<float>(tile[i])), dangling_term), teleport);
            r_new[node] = fresh;
            for (int p = 0; p < push.count; ++p) push.ptr[p][node] = fresh;     // straight into the peers' vectors
            const float diff = __fsub_rn(fresh, r_old[node]);
            res2 += static_cast<double>(__fmul_rn(diff, diff));
            if (dangling[node]) mass += static_cast<double>(fresh);
        }
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
    // The strip width was chosen for the tile count before the snap; the snap usually halves the tiles (doubles
    // the runs), so a narrower strip may do now — half the LDS per phase-1 workgroup, twice the wavefronts per CU.
    // Narrow only while the runs stay comfortably long: C4 (1 M power-law rows) 16384 -> 8192 columns, runs
    // 361 -> 182: 44.0 -> 41.9 us; C5 at 8192 would have 128-slot runs: 503 -> 547 us (profiles/r02_shape_sweep.txt,
    // r02_c4_sweep.txt), hence the margin over kTargetRun.
    {
        constexpr long long kComfortableRun = 160;
        int narrower = 4096;
        while (narrower < w && nnz / (strips_for(narrower) * tiles_for(r)) < kComfortableRun) narrower <<= 1;
        w = narrower;
    }
    {   // SPMV_DEBUG=strip=W,tile=R: shape overrides for experiments and boundary-case tests
        const long long v = debug_number("strip", 0);
        if (v == 4096 || v == 8192 || v == 16384 || v == 32768) w = static_cast<int>(v);
        const long long t = debug_number("tile", 0);
        if (t >= 64 && t <= kMaxTileRows && t % 64 == 0) r = static_cast<int>(t);
    }
    *strip_cols = w;
    *tile_rows = r;
}

// the product stream and the long-row chunk sums a call on stream `s` writes (see TiledPlan::StreamScratch)
struct Scratch {
    float* prod;
    float* long_sums;
};
constexpr size_t kMaxExtraScratch = 7;

hipError_t scratch_for(const TiledPlan& plan, hipStream_t s, Scratch* out) {
    std::lock_guard<std::mutex> guard(plan.scratch_lock);
    if (!plan.primary_taken) {
        plan.primary_taken = true;
        plan.primary_stream = s;
    }
    if (plan.primary_stream == s) {
        *out = Scratch{plan.prod, plan.long_sums};
        return hipSuccess;
    }
    for (const TiledPlan::StreamScratch& e : plan.extra_scratch) {
        if (e.stream == s) {
            *out = Scratch{e.prod, e.long_sums};
            return hipSuccess;
        }
    }
    if (plan.extra_scratch.size() >= kMaxExtraScratch) return hipErrorOutOfMemory;
    TiledPlan::StreamScratch fresh{s, nullptr, nullptr};
    if (malloc_any_time(reinterpret_cast<void**>(&fresh.prod), static_cast<size_t>(plan.nnz + 8) * sizeof(float)) != hipSuccess ||
        malloc_any_time(reinterpret_cast<void**>(&fresh.long_sums),
                        static_cast<size_t>(std::max(plan.num_long_chunks, 1)) * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        if (fresh.prod) (void)hipFree(fresh.prod);
        return hipErrorOutOfMemory;
    }
    plan.extra_scratch.push_back(fresh);
    *out = Scratch{fresh.prod, fresh.long_sums};
    return hipSuccess;
}

// phase 1 for the items [first_item, first_item + num_items) and, with_long, the long-row chunks
template <int W, int BLOCK>
hipError_t launch_expand_as(const TiledPlan& plan, const Scratch& sc, int first_item, int num_items, bool with_long,
                            const float* d_x, const PrState* d_state, hipStream_t s) {
    const int chunks = with_long ? plan.num_long_chunks : 0;
    const LongRows lr{plan.long_chunks, chunks, plan.csr_nnz, plan.csr_cols, plan.csr_vals, sc.long_sums};
    const int long_blocks = xcd_grid((chunks + BLOCK / 64 - 1) / (BLOCK / 64));
    const int grid = long_blocks + xcd_grid(num_items);
    if (grid == 0) return hipSuccess;
    if (plan.col_weight) {
        tiled_expand_kernel<W, BLOCK, true><<<grid, BLOCK, 0, s>>>(
            plan.items, first_item, num_items, long_blocks, nullptr, plan.a_lcol, plan.col_weight, d_x, plan.num_cols, sc.prod, lr, d_state);
    } else {
        tiled_expand_kernel<W, BLOCK, false><<<grid, BLOCK, 0, s>>>(
            plan.items, first_item, num_items, long_blocks, plan.a_val, plan.a_lcol, nullptr, d_x, plan.num_cols, sc.prod, lr, d_state);
    }
    return hipGetLastError();
}

hipError_t launch_expand(const TiledPlan& plan, const Scratch& sc, int first_item, int num_items, bool with_long,
                         const float* d_x, const PrState* d_state, hipStream_t s) {
    switch (plan.strip_cols) {
        case 4096:  return launch_expand_as<4096, 512>(plan, sc, first_item, num_items, with_long, d_x, d_state, s);
        case 8192:  return launch_expand_as<8192, 512>(plan, sc, first_item, num_items, with_long, d_x, d_state, s);
        case 16384: return launch_expand_as<16384, 512>(plan, sc, first_item, num_items, with_long, d_x, d_state, s);
        default:    return launch_expand_as<32768, 1024>(plan, sc, first_item, num_items, with_long, d_x, d_state, s);   // 128 KiB of LDS: one workgroup per CU
    }
}

LongSeeds long_seeds(const TiledPlan& plan, const Scratch& sc) {
    return LongSeeds{plan.long_rows, plan.long_first, plan.num_long > 0 ? plan.tile_long : nullptr, sc.long_sums};
}

hipError_t launch_reduce(const TiledPlan& plan, const Scratch& sc, float* d_y, hipStream_t s) {
    const size_t lds = static_cast<size_t>(plan.tile_rows) * sizeof(double);
    const void* kernel = reinterpret_cast<const void*>(&tiled_reduce_kernel<kReduceThreads>);
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    tiled_reduce_kernel<kReduceThreads><<<xcd_grid(plan.num_tiles), kReduceThreads, lds, s>>>(
        plan.tile_rows, plan.num_tiles, plan.pass_first, reinterpret_cast<const PassDesc*>(plan.pass_desc), sc.prod, plan.a_drow,
        long_seeds(plan, sc), plan.num_rows, d_y);
    return hipGetLastError();
}

hipError_t launch_pagerank_reduce(const TiledPlan& plan, const Scratch& sc, const RowMap& map, int n_global, const float* d_r_old,
                                  float* d_r_new, const unsigned char* d_dangling, float damping,
                                  const PrState* d_state, double* d_block_partials,
                                  const PushTargets& push, hipStream_t s) {
    const size_t lds = static_cast<size_t>(plan.tile_rows) * sizeof(double);
    const void* kernel = reinterpret_cast<const void*>(&tiled_pagerank_reduce_kernel<kReduceThreads>);
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    tiled_pagerank_reduce_kernel<kReduceThreads><<<xcd_grid(plan.num_tiles), kReduceThreads, lds, s>>>(
        plan.tile_rows, plan.num_tiles, plan.pass_first, reinterpret_cast<const PassDesc*>(plan.pass_desc), sc.prod, plan.a_drow,
        long_seeds(plan, sc), plan.num_rows, map, n_global, d_r_old, d_r_new, d_dangling, damping, d_state,
        d_block_partials, push);
    return hipGetLastError();
}

} // namespace

namespace {

bool eligible_dims(long long rows, long long cols, long long nnz) {
    static const bool enabled = [] {
        const char* env = std::getenv("SPMV_TILED");
        return !(env && env[0] == '0');
    }();
    // Everything wider than what one CU's LDS holds of x (<= 32768 columns: the x-in-LDS vector kernel).  Rounds 1-3 drew the line
    // at 65536 columns (a tie then: 69 vs 71 us on 1 M rows x 16); with round 4's engine it wins from the first column past the
    // LDS kernel's reach — 34000 columns: 43.7 against 65.5 us (1 M x 16), 92 against 150 us (4 M x 8); tools/crossover_probe.py,
    // profiles/r04_crossover.txt.  (SPMV_DEBUG=min_cols=1,min_nnz=1: tests force small matrices through the engine)
    const long long min_cols = debug_number("min_cols", 32769LL);
    const long long min_nnz = debug_number("min_nnz", 1LL << 20);
    if (!enabled || rows <= 0 || nnz < min_nnz || cols < min_cols) return false;
    int w = 0, r = 0;
    choose_shape(rows, cols, nnz, &w, &r);
    const long long strips = (cols + w - 1) / w;
    return strips <= kMaxBuildStrips && strips * ((rows + r - 1) / r) <= kMaxCells;
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
    void* owned[] = {p->a_val, p->a_lcol, p->a_drow, p->prod, p->cells_t, p->items, p->long_rows, p->long_chunks,
                     p->long_first, p->long_sums, p->tile_long, p->col_weight, p->pass_first, p->pass_desc};
    for (void* q : owned) if (q) (void)hipFree(q);
    for (const TiledPlan::StreamScratch& e : p->extra_scratch) {
        if (e.prod) (void)hipFree(e.prod);
        if (e.long_sums) (void)hipFree(e.long_sums);
    }
    delete[] p->strip_first_item;
    delete p;
}

namespace {

// the device passes of the build for one entry source
// The build's temporaries are carved out of two allocations: every hipFree synchronises the device, and eleven
// of them were ~0.8 ms of a 5.5 ms build.
struct BuildArena {
    char* base = nullptr;
    size_t size = 0, used = 0;
    static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }
    template <typename T>
    T* take(long long count) {
        T* p = reinterpret_cast<T*>(base + used);
        used += padded(static_cast<size_t>(std::max<long long>(count, 1)) * sizeof(T));
        return p;
    }
};

// SPMV_TRACE=1: wall-clock time of each host phase of a plan build, on stderr
struct BuildTrace {
    bool on = std::getenv("SPMV_TRACE") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char* phase) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[spmv trace] plan build: %-28s %8.3f ms\n", phase,
                     std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

template <typename Src>
hipError_t build_cells(const Src& dev_src, bool has_long_path, TiledPlan* plan, bool want_values,
                       std::vector<int>* host_strip, hipStream_t s) {
    BuildTrace trace;
    const int S = plan->num_strips, T = plan->num_tiles;
    const long long cells = static_cast<long long>(S) * T;

    int *d_small = nullptr;            // [0] longest row, [1] long-row count
    int *tile_batch = nullptr, *batch_row = nullptr, *batch_tile = nullptr, *cell_slots = nullptr, *offs = nullptr;
    int *strip_begin = nullptr;
    long long* block_sum = nullptr;    // scan scratch; [blocks] sums, then [blocks] grand total, [blocks + 1] entry count
    uint2* groups = nullptr;
    unsigned int* meta = nullptr;      // per-entry records between the ranking and the placing pass
    int* place_todo = nullptr;             // [0] how many, [1 ...] which batches the staged placing pass left to the scattered one
    BuildArena first, second;     // what is known up front; what depends on the batch count
    auto cleanup = [&](hipError_t e) {
        for (void* q : {static_cast<void*>(first.base), static_cast<void*>(second.base), static_cast<void*>(strip_begin)}) {
            if (q) (void)hipFree(q);
        }
        return e;
    };
    const int scan_blocks = static_cast<int>((cells + kScanTile - 1) / kScanTile);
    const int tile_scan_blocks = (T + kScanTile - 1) / kScanTile;

    const long long scan_slots = std::max(scan_blocks, tile_scan_blocks) + 2;
    first.size = BuildArena::padded(2 * sizeof(int)) + BuildArena::padded((static_cast<size_t>(T) + 1) * sizeof(int)) +
                 BuildArena::padded(static_cast<size_t>(cells) * sizeof(int)) + BuildArena::padded((static_cast<size_t>(cells) + 1) * sizeof(int)) +
                 BuildArena::padded(static_cast<size_t>(scan_slots) * sizeof(long long));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&first.base), first.size);
    if (e == hipSuccess) {
        d_small = first.take<int>(2);
        tile_batch = first.take<int>(static_cast<long long>(T) + 1);
        cell_slots = first.take<int>(cells);
        offs = first.take<int>(cells + 1);
        block_sum = first.take<long long>(scan_slots);
        e = dev_alloc(&strip_begin, S + 1);          // (outlives this function: the fold probe reads it)
    }
    if (e == hipSuccess) e = hipMemsetAsync(d_small, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return cleanup(e);

    // ---- batch geometry: how many entries a builder workgroup can hold in LDS
    const int lds_bytes = S <= 1024 ? kBuildLdsSmall : kBuildLdsLarge;
    int capacity = (lds_bytes - kBuildBinWords * 4 * S - kBuildWaveCountBytes * ((S + 3) / 4 * 4)) / kBuildEntryBytes / 64 * 64;
    if (capacity < 512) return cleanup(hipErrorInvalidValue);
    capacity = std::min(capacity, kBuildMaxCapacity);        // what a workgroup's threads keep in registers
    max_row_kernel<<<std::min(1024, (plan->num_rows + kBlock - 1) / kBlock), kBlock, 0, s>>>(dev_src, plan->num_rows, d_small);
    int longest = 0;
    e = hipMemcpyAsync(&longest, d_small, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);
    trace.mark("allocations + longest row");
    const int longest_short = std::min(longest, plan->long_row);
    if (longest_short >= capacity) plan->long_row = capacity / 2;      // (rows that long go the direct way)
    BuildShape sh;
    sh.num_rows = plan->num_rows;
    sh.num_tiles = T;
    sh.num_strips = S;
    sh.strip_shift = __builtin_ctz(static_cast<unsigned>(plan->strip_cols));
    sh.tile_rows = plan->tile_rows;
    sh.long_row = plan->long_row;
    sh.quota = std::max(64, capacity - std::min(longest, plan->long_row));
    sh.any_long = longest > plan->long_row ? 1 : 0;
    sh.stable_bins = 1;
    if (debug_is("rank", "plain")) sh.stable_bins = 0;

    tile_batches_kernel<<<(T + kBlock - 1) / kBlock, kBlock, 0, s>>>(dev_src, sh, tile_batch);
    // exclusive scan of the per-tile batch counts (one workgroup: T is at most a few hundred thousand)
    exclusive_scan_small_kernel<<<1, kScanBlock, 0, s>>>(tile_batch, T);
    int num_batches = 0;
    e = hipMemcpyAsync(&num_batches, tile_batch + T, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);

    trace.mark("batch counts");
    const long long group_count = static_cast<long long>(num_batches) * S;
    second.size = BuildArena::padded((static_cast<size_t>(num_batches) + 1) * sizeof(int)) +
                  BuildArena::padded(static_cast<size_t>(std::max(num_batches, 1)) * sizeof(int)) +
                  BuildArena::padded(static_cast<size_t>(std::max<long long>(group_count, 1)) * sizeof(uint2)) +
                  BuildArena::padded(static_cast<size_t>(std::max<long long>(plan->csr_nnz, 1)) * sizeof(unsigned int)) +
                  BuildArena::padded((static_cast<size_t>(std::max(num_batches, 1)) + 1) * sizeof(int));
    e = hipMalloc(reinterpret_cast<void**>(&second.base), second.size);
    if (e == hipSuccess) {
        batch_row = second.take<int>(static_cast<long long>(num_batches) + 1);
        batch_tile = second.take<int>(num_batches);
        groups = second.take<uint2>(group_count);
        meta = second.take<unsigned int>(plan->csr_nnz);
        place_todo = second.take<int>(static_cast<long long>(num_batches) + 1);
    }
    if (e == hipSuccess && has_long_path) {
        e = dev_alloc(&plan->long_rows, plan->csr_nnz / std::max(plan->long_row, 1) + 1);
    }
    if (e != hipSuccess) return cleanup(e);
    batch_rows_kernel<<<T, kBlock, 0, s>>>(dev_src, sh, tile_batch, batch_row, batch_tile);

    trace.mark("allocations (batches)");
    // ---- ranking pass: group sizes + per-entry records; cell placement; scan
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&batch_rank_kernel<Src>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return cleanup(e);
    batch_rank_kernel<Src><<<xcd_grid(num_batches), kBuildBlock, lds_bytes, s>>>(
        dev_src, sh, num_batches, capacity, batch_row, batch_tile, groups, meta, plan->long_rows, d_small + 1);
    unsigned long long* entry_total = reinterpret_cast<unsigned long long*>(block_sum + scan_blocks + 1);
    e = hipMemsetAsync(entry_total, 0, sizeof(unsigned long long), s);
    cell_place_kernel<<<static_cast<int>((cells + kBlock - 1) / kBlock), kBlock, 0, s>>>(T, S, tile_batch, groups, cell_slots,
                                                                                       entry_total);
    scan_sums_kernel<<<scan_blocks, kScanBlock, 0, s>>>(cell_slots, cells, block_sum);
    scan_top_kernel<<<1, kScanBlock, 0, s>>>(block_sum, scan_blocks, block_sum + scan_blocks);
    scan_apply_kernel<<<scan_blocks, kScanBlock, 0, s>>>(cell_slots, cells, block_sum, offs);
    if (e == hipSuccess) e = hipGetLastError();
    long long totals[2] = {0, 0};          // slots, entries
    int num_long = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(totals, block_sum + scan_blocks, 2 * sizeof(long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&num_long, d_small + 1, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);
    trace.mark("ranking + scans (sync)");
    if (totals[0] >= 0x7fffffffLL - 64) return cleanup(hipErrorInvalidValue);       // slot indices are 32-bit
    plan->nnz = totals[0];
    plan->entries = totals[1];
    plan->num_long = num_long;

    // ---- placing pass: write the slots
    if (want_values) e = dev_alloc(&plan->a_val, plan->nnz + 8);
    // + 8: the 16-byte loads of a run's last group stay inside the allocation whatever its alignment
    if (e == hipSuccess) e = dev_alloc(&plan->a_lcol, plan->nnz + 8);
    if (e == hipSuccess) e = dev_alloc(&plan->a_drow, plan->nnz + 8);
    if (e == hipSuccess) e = dev_alloc(&plan->cells_t, 2 * cells);
    const int pass_waves = T * kReduceWaves;
    if (e == hipSuccess) e = dev_alloc(&plan->pass_first, static_cast<long long>(pass_waves) + 1);
    if (e != hipSuccess) return cleanup(e);
    {   // the tile-major cell table first: the placing kernels read their tile's cell begins from it (one contiguous read)
        const int grid = static_cast<int>(std::min<long long>((cells + kBlock) / kBlock, 4096));
        cell_table_kernel<<<grid, kBlock, 0, s>>>(offs, S, T, reinterpret_cast<int2*>(plan->cells_t), strip_begin);
        // phase 2's passes are a function of the cell table alone: counted and scanned here, beside the placing pass, so that
        // their total arrives with this function's last synchronisation (the descriptors are written by build_plan)
        pass_layout_kernel<false><<<T, kReduceThreads, 0, s>>>(T, S, reinterpret_cast<const int2*>(plan->cells_t), nullptr, plan->pass_first,
                                                              nullptr, nullptr);
        exclusive_scan_small_kernel<<<1, kScanBlock, 0, s>>>(plan->pass_first, pass_waves);
    }
    if (plan->nnz > 0) {
        // staged placing pass (contiguous segments); the batches it cannot hold are flagged for the scattered one
        bool staged = true;
        if (debug_is("place", "scattered")) staged = false;
        int* todo = place_todo;
        if (staged) {
            const int stage_slots = (capacity + 1024 + 63) / 64 * 64;
            const size_t stage_lds = static_cast<size_t>(kStageBytesPerStrip) * S + static_cast<size_t>(kStageBytesPerSlot) * stage_slots;
            staged = stage_lds + sizeof(int) * kBuildBlock + 64 <= 160 * 1024;
            if (staged) e = hipMemsetAsync(todo, 0, sizeof(int), s);
            if (staged && e == hipSuccess) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&batch_place_staged_kernel<Src>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(stage_lds));
            }
            if (e != hipSuccess) return cleanup(e);
            if (staged) {
                batch_place_staged_kernel<Src><<<xcd_grid(num_batches), kBuildBlock, stage_lds, s>>>(
                    dev_src, sh, num_batches, capacity, stage_slots, batch_row, batch_tile, groups, meta, reinterpret_cast<const int2*>(plan->cells_t),
                    plan->a_val, plan->a_lcol, plan->a_drow, todo);
            }
        }
        batch_place_kernel<Src><<<staged ? std::min(xcd_grid(num_batches), 512) : xcd_grid(num_batches), kBuildBlock, 12 * static_cast<size_t>(S), s>>>(
            dev_src, sh, num_batches, batch_row, batch_tile, groups, meta, reinterpret_cast<const int2*>(plan->cells_t), plan->a_val, plan->a_lcol, plan->a_drow,
            staged ? todo : nullptr);
        cell_padding_kernel<<<static_cast<int>(std::min<long long>((cells + kBlock - 1) / kBlock, 4096)), kBlock, 0, s>>>(
            cell_slots, offs, cells, plan->a_val, plan->a_lcol, plan->a_drow);
    }
    e = hipGetLastError();
    host_strip->assign(S + 1, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(host_strip->data(), strip_begin, host_strip->size() * sizeof(int),
                                            hipMemcpyDeviceToHost, s);
    int pass_total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&pass_total, plan->pass_first + pass_waves, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);
    plan->num_passes = pass_total;
    trace.mark("allocations + placing (sync)");
    // strip_begin is needed again by the fold probe: hand it to the plan's scratch (freed by the caller)
    plan->items = strip_begin;          // temporarily; build_plan replaces it
    strip_begin = nullptr;
    return cleanup(hipSuccess);
}

hipError_t build_plan(const Source& src, TiledPlan** out, hipStream_t s) {
    const auto t_begin = std::chrono::steady_clock::now();
    BuildTrace trace;
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
    } else {
        plan->csr_vals = src.ell->d_values;       // identity of the slabs the plan was built from (aux_table.cpp)
    }
    choose_shape(src.rows, src.cols, src.nnz, &plan->strip_cols, &plan->tile_rows);
    plan->num_strips = (src.cols + plan->strip_cols - 1) / plan->strip_cols;
    plan->num_tiles = (src.rows + plan->tile_rows - 1) / plan->tile_rows;
    const long long cells = static_cast<long long>(plan->num_strips) * plan->num_tiles;
    // A row spreads over the strips; rows with many entries per cell make lanes meet on one LDS word in phase 2,
    // so the longest rows take the direct path instead (512-entry chunks, direct gather, seeds).  With the
    // compare-and-swap add of round 1 the line sat at 2-4 entries per strip; the hardware ds_add_f64 takes
    // collisions far better (profiles/r02_long_row_sweep.txt, C4 = 1 M power-law rows, 62 strips: limit 124
    // entries 49.5 us, 248: 47.2, 496: 44.2, unlimited 47.8; 10 M x 10 M power-law: 426 / 435 / 416 / 402 us).
    int long_factor = 8;
    long_factor = static_cast<int>(std::max(1LL, debug_number("long_factor", long_factor)));
    int long_cap = kMaxLongRow;
    long_cap = static_cast<int>(std::max(64LL, debug_number("long_cap", long_cap)));
    plan->long_row = A ? std::max(64, std::min(long_cap, long_factor * plan->num_strips)) : 0x3fffffff;

    auto fail = [&](hipError_t e) {
        tiled_free(plan);
        return e;
    };

    std::vector<int> host_strip;
    bool fold = true;
    if (const char* env = std::getenv("SPMV_TILED_FOLD")) fold = env[0] != '0';
    hipError_t e;
    if (A) {
        const CsrSource dev_src{A->d_row_ptrs, A->d_col_indices, A->d_values};
        e = build_cells(dev_src, true, plan, true, &host_strip, s);
    } else {
        const ELLMatrix* E = src.ell;
        const EllSource dev_src{E->num_rows, E->max_nnz_per_row, E->d_col_indices, E->d_values};
        e = build_cells(dev_src, false, plan, true, &host_strip, s);
    }
    trace.mark("cells built, temporaries freed");
    int* strip_begin = plan->items;         // parked there by build_cells
    plan->items = nullptr;
    auto fail_with_strip = [&](hipError_t err) {
        if (strip_begin) (void)hipFree(strip_begin);
        return fail(err);
    };
    if (e != hipSuccess) return fail_with_strip(e);
    {   // phase 2's pass descriptors (pass_layout_kernel; counted and scanned beside the placing pass in build_cells)
        e = hipMalloc(&plan->pass_desc, static_cast<size_t>(std::max<long long>(plan->num_passes, 1)) * sizeof(PassDesc));
        if (e != hipSuccess) return fail_with_strip(e);
        pass_layout_kernel<true><<<plan->num_tiles, kReduceThreads, 0, s>>>(plan->num_tiles, plan->num_strips,
                                                                           reinterpret_cast<const int2*>(plan->cells_t), plan->a_drow, nullptr,
                                                                           plan->pass_first, static_cast<PassDesc*>(plan->pass_desc));
        e = hipGetLastError();
        if (e != hipSuccess) return fail_with_strip(e);
        trace.mark("phase-2 pass descriptors");
    }

    if (A && plan->num_long > 0) {
        // cut the long rows into wavefront-sized chunks (the list is short: <= nnz / long_row rows)
        std::vector<int> rows(plan->num_long);
        e = hipMemcpy(rows.data(), plan->long_rows, rows.size() * sizeof(int), hipMemcpyDeviceToHost);
        std::sort(rows.begin(), rows.end());           // the device listed them in arrival order
        if (e == hipSuccess) e = hipMemcpy(plan->long_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice);
        std::vector<int> chunks;
        std::vector<int> all_ptrs;                       // many long rows: one bulk copy instead
        const int* host_ptrs = A->row_ptrs;
        if (!host_ptrs && plan->num_long > 256 && e == hipSuccess) {
            all_ptrs.resize(static_cast<size_t>(A->num_rows) + 1);
            e = hipMemcpy(all_ptrs.data(), A->d_row_ptrs, all_ptrs.size() * sizeof(int), hipMemcpyDeviceToHost);
            host_ptrs = all_ptrs.data();
        }
        std::vector<int> first_chunk;
        for (int row : rows) {
            first_chunk.push_back(static_cast<int>(chunks.size() / 3));
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
        first_chunk.push_back(plan->num_long_chunks);
        if (e == hipSuccess) e = dev_alloc(&plan->long_first, static_cast<long long>(first_chunk.size()));
        if (e == hipSuccess) e = hipMemcpy(plan->long_first, first_chunk.data(), first_chunk.size() * sizeof(int),
                                           hipMemcpyHostToDevice);
        if (e == hipSuccess) e = dev_alloc(&plan->long_sums, plan->num_long_chunks);
        if (e == hipSuccess) e = dev_alloc(&plan->long_chunks, static_cast<long long>(chunks.size()));
        if (e == hipSuccess) e = hipMemcpy(plan->long_chunks, chunks.data(), chunks.size() * sizeof(int),
                                           hipMemcpyHostToDevice);
        if (e == hipSuccess) {          // first long row of every tile (the list is ascending)
            std::vector<int> tile_long(static_cast<size_t>(plan->num_tiles) + 1);
            size_t at = 0;
            for (int t = 0; t <= plan->num_tiles; ++t) {
                const long long bound = static_cast<long long>(t) * plan->tile_rows;
                while (at < rows.size() && rows[at] < bound) ++at;
                tile_long[t] = static_cast<int>(at);
            }
            e = dev_alloc(&plan->tile_long, static_cast<long long>(tile_long.size()));
            if (e == hipSuccess) e = hipMemcpy(plan->tile_long, tile_long.data(), tile_long.size() * sizeof(int),
                                               hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) return fail_with_strip(e);
    }

    // column-weight folding (see strip_weight_kernel): on unless SPMV_TILED_FOLD=0.  The first few strips
    // alone settle it for arbitrary values (a column that occurs twice there already differs).
    if (fold && plan->nnz > 0) {
        int* differs = nullptr;
        e = dev_alloc(&differs, 1);
        if (e == hipSuccess) e = hipMemsetAsync(differs, 0, sizeof(int), s);
        int host_differs = 1;
        auto probe = [&](int first, int count, int limit, float* weight) {
            switch (plan->strip_cols) {
                case 4096:  strip_weight_kernel<4096><<<count, 1024, 0, s>>>(first, limit, strip_begin, plan->num_cols, plan->a_val, plan->a_lcol, plan->a_drow, weight, differs); break;
                case 8192:  strip_weight_kernel<8192><<<count, 1024, 0, s>>>(first, limit, strip_begin, plan->num_cols, plan->a_val, plan->a_lcol, plan->a_drow, weight, differs); break;
                case 16384: strip_weight_kernel<16384><<<count, 1024, 0, s>>>(first, limit, strip_begin, plan->num_cols, plan->a_val, plan->a_lcol, plan->a_drow, weight, differs); break;
                default:    strip_weight_kernel<32768><<<count, 1024, 0, s>>>(first, limit, strip_begin, plan->num_cols, plan->a_val, plan->a_lcol, plan->a_drow, weight, differs); break;
            }
        };
        // round 0: the first 32 K slots of up to 64 strips, verdict only (with arbitrary values some column
        // repeats there and the matter is settled before anything is allocated); round 1: everything
        const int sample = std::min(plan->num_strips, 64);
        for (int round = 0; round < 2 && e == hipSuccess; ++round) {
            if (round == 0) {
                probe(0, sample, 32768, nullptr);
            } else {
                e = dev_alloc(&plan->col_weight, plan->num_cols);
                if (e != hipSuccess) break;
                probe(0, plan->num_strips, 0x7fffffff, plan->col_weight);
                if (plan->num_long_chunks > 0) {
                    const int grid = (plan->num_long_chunks + kBlock / 64 - 1) / (kBlock / 64);
                    long_row_weight_kernel<0><<<grid, kBlock, 0, s>>>(plan->long_chunks, plan->num_long_chunks, plan->csr_cols,
                                                                   plan->csr_vals, plan->col_weight, differs);
                    long_row_weight_kernel<1><<<grid, kBlock, 0, s>>>(plan->long_chunks, plan->num_long_chunks, plan->csr_cols,
                                                                   plan->csr_vals, plan->col_weight, differs);
                }
                weight_finish_kernel<<<std::min(2048, (plan->num_cols + kBlock - 1) / kBlock), kBlock, 0, s>>>(
                    plan->col_weight, plan->num_cols);
            }
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&host_differs, differs, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (host_differs) break;
        }
        if (differs) (void)hipFree(differs);
        if (e != hipSuccess) return fail_with_strip(e);
        if (host_differs) {
            if (plan->col_weight) (void)hipFree(plan->col_weight);
            plan->col_weight = nullptr;
        } else {
            (void)hipFree(plan->a_val);           // folded: phase 1 reads weights, not values
            plan->a_val = nullptr;
        }
    }
    (void)hipFree(strip_begin);
    strip_begin = nullptr;

    e = dev_alloc(&plan->prod, plan->nnz + 8);
    if (e != hipSuccess) return fail(e);

    // phase-1 work items: every strip's range cut into EQUAL pieces of <= item_entries (enough
    // pieces to fill the chip several times), piece boundaries on multiples of 8 slots
    const long long floor_entries = std::max<long long>(kMinItemEntries, plan->strip_cols);   // strip load <= 40 % of the stream
    // (a folded plan stages TWO arrays per item — x and the column weights — and streams 6 bytes per slot instead of 10:
    // 16 K-slot items cost it 12 %, 0.460 against 0.41 ms per PageRank step on C5; it keeps the larger items)
    const long long item_cap = plan->col_weight ? 4 * kMaxItemEntries : kMaxItemEntries;
    int item_entries = static_cast<int>(std::max<long long>(
        floor_entries, std::min<long long>(item_cap, (plan->nnz / 2048 + 7) / 8 * 8)));
    item_entries = static_cast<int>(std::max(1024LL, debug_number("item", item_entries)));
    std::vector<int> items;
    plan->strip_first_item = new int[static_cast<size_t>(plan->num_strips) + 1];
    for (int strip = 0; strip < plan->num_strips; ++strip) {
        plan->strip_first_item[strip] = static_cast<int>(items.size() / 3);
        const int begin = host_strip[strip], stop = host_strip[strip + 1];
        const int parts = (stop - begin + item_entries - 1) / item_entries;
        int b = begin;
        for (int part = 1; part <= parts; ++part) {
            int next = part == parts ? stop
                                     : static_cast<int>(begin + static_cast<long long>(stop - begin) * part / parts) / 8 * 8;
            next = std::max(next, b);
            if (next == b && part != parts) continue;
            items.push_back(strip);
            items.push_back(b);
            items.push_back(next);
            b = next;
        }
    }
    plan->num_items = static_cast<int>(items.size() / 3);
    plan->strip_first_item[plan->num_strips] = plan->num_items;
    e = dev_alloc(&plan->items, static_cast<long long>(items.size()));
    if (e == hipSuccess && !items.empty()) {
        e = hipMemcpy(plan->items, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return fail(e);

    trace.mark("fold probe, long rows, items");
    plan->plan_bytes = plan->nnz * (4 /*prod*/ + 2 + 1 + (plan->a_val ? 4 : 0)) + cells * 8 +
                       plan->num_passes * static_cast<long long>(sizeof(PassDesc)) + 4LL * (plan->num_tiles * kReduceWaves + 1) +
                       (plan->col_weight ? 4LL * plan->num_cols : 0) +
                       12LL * plan->num_items + 12LL * plan->num_long_chunks;
    plan->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = plan;
    return hipSuccess;
}

} // namespace

namespace {
// position-weighted checksums of the three slot arrays and of the cell table (a debugging / test aid: two
// builds of one matrix must give the same four numbers whatever path the builder took)
__global__ __launch_bounds__(kBlock)
void plan_checksum_kernel(long long slots, const float* __restrict__ a_val, const unsigned short* __restrict__ a_lcol,
                          const unsigned char* __restrict__ a_drow, long long table_ints, const int* __restrict__ cells_t,
                          unsigned long long* __restrict__ out /*[4]*/) {
    unsigned long long v = 0, c = 0, d = 0, t = 0;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < slots;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        const unsigned long long w = 2 * static_cast<unsigned long long>(i) + 1;
        if (a_val) v += w * __float_as_uint(a_val[i]);
        c += w * a_lcol[i];
        d += w * a_drow[i];
    }
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < table_ints;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        t += (2 * static_cast<unsigned long long>(i) + 1) * static_cast<unsigned int>(cells_t[i]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
        c += __shfl_xor(c, off, 64);
        d += __shfl_xor(d, off, 64);
        t += __shfl_xor(t, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], v);
        atomicAdd(&out[1], c);
        atomicAdd(&out[2], d);
        atomicAdd(&out[3], t);
    }
}
} // namespace

hipError_t tiled_checksum(const TiledPlan& plan, unsigned long long out[4], hipStream_t s) {
    unsigned long long* d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), 4 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        plan_checksum_kernel<<<1024, kBlock, 0, s>>>(plan.nnz, plan.a_val, plan.a_lcol, plan.a_drow,
                                                     2LL * plan.num_strips * plan.num_tiles, plan.cells_t, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_out);
    return e;
}

hipError_t tiled_spmv(const TiledPlan& plan, const float* d_x, float* d_y, hipStream_t s) {
    Scratch sc;
    hipError_t e = scratch_for(plan, s, &sc);
    if (e != hipSuccess) return e;
    // two host threads may call on the same stream: the pair of launches must not interleave with another pair
    std::lock_guard<std::mutex> pair(plan.launch_lock);
    e = launch_expand(plan, sc, 0, plan.num_items, true, d_x, nullptr, s);       // phase 1 + the long rows
    if (e != hipSuccess) return e;
    return launch_reduce(plan, sc, d_y, s);
}

// After convergence the kernels of both parts return at once: r_new and the product stream stay as the last
// committed step left them.
hipError_t tiled_pagerank_expand(const TiledPlan& plan, int strip_begin, int strip_end, bool with_long,
                                 const float* d_r_old, const PrState* d_state, hipStream_t s) {
    Scratch sc;
    const hipError_t e = scratch_for(plan, s, &sc);
    if (e != hipSuccess) return e;
    strip_begin = std::max(0, std::min(strip_begin, plan.num_strips));
    strip_end = std::max(strip_begin, std::min(strip_end, plan.num_strips));
    const int first = plan.strip_first_item[strip_begin];
    return launch_expand(plan, sc, first, plan.strip_first_item[strip_end] - first, with_long, d_r_old, d_state, s);
}

hipError_t tiled_pagerank_finish(const TiledPlan& plan, const RowMap& map, int n_global,
                                 const float* d_r_old, float* d_r_new,
                                 const unsigned char* d_dangling, float damping,
                                 const PrState* d_state, double* d_block_partials,
                                 const PushTargets& push, hipStream_t s) {
    Scratch sc;
    const hipError_t e = scratch_for(plan, s, &sc);
    if (e != hipSuccess) return e;
    return launch_pagerank_reduce(plan, sc, map, n_global, d_r_old, d_r_new, d_dangling, damping, d_state,
                                  d_block_partials, push, s);
}

} // namespace detail
} // namespace spmv
