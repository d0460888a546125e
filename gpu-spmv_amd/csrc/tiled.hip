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
//   phase 1 "expand" : entries grouped by COLUMN STRIP (W = 8192 columns = 32 KiB of
//        LDS).  A workgroup loads its x strip into LDS once, streams (value, local
//        column, destination) with coalesced loads, gathers x from LDS and stores the
//        product to the entry's slot in layout B.
//   phase 2 "reduce" : products grouped by ROW TILE (R = 2048 rows = 8 KiB of LDS),
//        strips in order inside a tile, so phase 1's stores land in contiguous runs.
//        A workgroup zeroes its y tile in LDS, streams (product, local row), adds into
//        the tile and writes the tile out with coalesced stores.  gfx950's ds_add_f32
//        is ~30x slower than its integer LDS atomics (0.38 vs 11.7 lanes/clk/CU
//        measured), so the add is a compare-and-swap on the word's integer image
//        (3.5 lanes/clk/CU measured, race-free for any row multiplicity).
//
// HBM traffic per entry: 10 B read + 4 B written in phase 1, 6 B read in phase 2
// (20 B vs CSR's 8 B) — but every access is a coalesced stream, which beats one
// 64-byte random fetch per entry by ~4x at 10 M columns.
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

constexpr int kStripCols = 8192;     // W
constexpr int kTileRows = 2048;      // R
constexpr int kItemEntries = 16384;  // phase-1 work item size
constexpr long long kMaxCells = 1LL << 26;

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ plan building ----
// LANES lanes walk one row; every entry is assigned to cell (strip, tile).
template <int LANES, bool SCATTER>
__global__ __launch_bounds__(kBlock)
void bucket_kernel(int num_rows, int num_tiles, int num_strips,
                   const int* __restrict__ row_ptrs, const int* __restrict__ cols,
                   const float* __restrict__ vals,
                   int* __restrict__ cell_counter,            // [num_strips * num_tiles]
                   const int* __restrict__ offs_a,            // strip-major exclusive scan
                   const int* __restrict__ offs_b,            // tile-major exclusive scan
                   float* __restrict__ a_val, unsigned short* __restrict__ a_lcol,
                   int* __restrict__ a_dst, unsigned short* __restrict__ b_lrow) {
    constexpr int kRowsPerBlock = kBlock / LANES;
    const int lane = threadIdx.x % LANES;
    const long long row = static_cast<long long>(blockIdx.x) * kRowsPerBlock + threadIdx.x / LANES;
    if (row >= num_rows) return;
    const int tile = static_cast<int>(row / kTileRows);
    const unsigned short lrow = static_cast<unsigned short>(row % kTileRows);
    for (int j = row_ptrs[row] + lane, end = row_ptrs[row + 1]; j < end; j += LANES) {
        const int c = cols[j];
        const int strip = c / kStripCols;
        const long long cell = static_cast<long long>(strip) * num_tiles + tile;
        if (!SCATTER) {
            atomicAdd(&cell_counter[cell], 1);
        } else {
            const int k = atomicAdd(&cell_counter[cell], 1);
            const int pos_a = offs_a[cell] + k;
            const int pos_b = offs_b[static_cast<long long>(tile) * num_strips + strip] + k;
            a_val[pos_a] = vals[j];
            a_lcol[pos_a] = static_cast<unsigned short>(c - strip * kStripCols);
            a_dst[pos_a] = pos_b;
            b_lrow[pos_b] = lrow;
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
    // Hillis-Steele over the 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
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

__global__ __launch_bounds__(kBlock)
void transpose_counts_kernel(const int* __restrict__ cnt, int num_strips, int num_tiles,
                             int* __restrict__ cnt_t) {
    const long long cells = static_cast<long long>(num_strips) * num_tiles;
    for (long long i = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; i < cells;
         i += static_cast<long long>(gridDim.x) * kBlock) {
        const long long tile = i / num_strips, strip = i % num_strips;
        cnt_t[i] = cnt[strip * num_tiles + tile];
    }
}

// tile_begin[t] = offs_b[t * num_strips]; strip_begin[s] = offs_a[s * num_tiles]
__global__ __launch_bounds__(kBlock)
void boundaries_kernel(const int* __restrict__ offs_a, const int* __restrict__ offs_b,
                       int num_strips, int num_tiles, int* __restrict__ strip_begin,
                       int* __restrict__ tile_begin) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i <= num_strips) strip_begin[i] = offs_a[static_cast<long long>(i) * num_tiles];
    if (i <= num_tiles) tile_begin[i] = offs_b[static_cast<long long>(i) * num_strips];
}

// ------------------------------------------------------------------------ phase 1 ----
__global__ __launch_bounds__(kBlock)
void tiled_expand_kernel(const int* __restrict__ items,
                         const float* __restrict__ a_val,
                         const unsigned short* __restrict__ a_lcol,
                         const int* __restrict__ a_dst,
                         const float* __restrict__ x, int num_cols,
                         float* __restrict__ prod) {
    __shared__ float xs[kStripCols];
    const int strip = items[3 * blockIdx.x];
    const int begin = items[3 * blockIdx.x + 1];
    const int end = items[3 * blockIdx.x + 2];

    const long long base = static_cast<long long>(strip) * kStripCols;
    const int width = static_cast<int>(min(static_cast<long long>(kStripCols), num_cols - base));
    const float* src = x + base;
    if ((reinterpret_cast<unsigned long long>(src) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < width; i += kBlock * 4) {
            if (i + 3 < width) {
                *reinterpret_cast<f32x4*>(xs + i) = *reinterpret_cast<const f32x4*>(src + i);
            } else {
                for (int k = i; k < width; ++k) xs[k] = src[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < width; i += kBlock) xs[i] = src[i];
    }
    __syncthreads();

    // four entries per lane per step; groups aligned to 4 entries (16-byte loads)
    for (int q = (begin & ~3) + threadIdx.x * 4; q < end; q += kBlock * 4) {
        if (q >= begin && q + 3 < end) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a_val + q);
            const u16x4 c = *reinterpret_cast<const u16x4*>(a_lcol + q);
            const i32x4 d = *reinterpret_cast<const i32x4*>(a_dst + q);
            const float p0 = v[0] * xs[c[0]], p1 = v[1] * xs[c[1]];
            const float p2 = v[2] * xs[c[2]], p3 = v[3] * xs[c[3]];
            prod[d[0]] = p0;
            prod[d[1]] = p1;
            prod[d[2]] = p2;
            prod[d[3]] = p3;
        } else {
            for (int k = max(q, begin); k < min(q + 4, end); ++k) {
                prod[a_dst[k]] = a_val[k] * xs[a_lcol[k]];
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

__device__ __forceinline__ void tile_accumulate(float* tile, int begin, int end,
                                                const float* __restrict__ prod,
                                                const unsigned short* __restrict__ b_lrow) {
    for (int q = (begin & ~3) + threadIdx.x * 4; q < end; q += kBlock * 4) {
        if (q >= begin && q + 3 < end) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(prod + q);
            const u16x4 r = *reinterpret_cast<const u16x4*>(b_lrow + q);
            lds_add(&tile[r[0]], p[0]);
            lds_add(&tile[r[1]], p[1]);
            lds_add(&tile[r[2]], p[2]);
            lds_add(&tile[r[3]], p[3]);
        } else {
            for (int k = max(q, begin); k < min(q + 4, end); ++k) lds_add(&tile[b_lrow[k]], prod[k]);
        }
    }
}

__global__ __launch_bounds__(kBlock)
void tiled_reduce_kernel(const int* __restrict__ tile_begin,
                         const float* __restrict__ prod,
                         const unsigned short* __restrict__ b_lrow,
                         int num_rows, float* __restrict__ y) {
    __shared__ float tile[kTileRows];
    for (int i = threadIdx.x; i < kTileRows; i += kBlock) tile[i] = 0.0f;
    __syncthreads();
    tile_accumulate(tile, tile_begin[blockIdx.x], tile_begin[blockIdx.x + 1], prod, b_lrow);
    __syncthreads();
    const long long first = static_cast<long long>(blockIdx.x) * kTileRows;
    for (int i = threadIdx.x; i < kTileRows && first + i < num_rows; i += kBlock) y[first + i] = tile[i];
}

// phase 2 with the PageRank update fused into the tile write-out (cf. pr_step_kernel)
__global__ __launch_bounds__(kBlock)
void tiled_pagerank_reduce_kernel(const int* __restrict__ tile_begin,
                                  const float* __restrict__ prod,
                                  const unsigned short* __restrict__ b_lrow,
                                  int local_rows, int row_offset, int n_global,
                                  const float* __restrict__ r_old, float* __restrict__ r_new,
                                  const unsigned char* __restrict__ dangling, float damping,
                                  const PrState* __restrict__ state,
                                  double* __restrict__ block_partials) {
    if (state->done) return;
    __shared__ float tile[kTileRows];
    for (int i = threadIdx.x; i < kTileRows; i += kBlock) tile[i] = 0.0f;
    __syncthreads();
    tile_accumulate(tile, tile_begin[blockIdx.x], tile_begin[blockIdx.x + 1], prod, b_lrow);
    __syncthreads();

    const float teleport = __fdiv_rn(1.0f - damping, static_cast<float>(n_global));
    const float dangling_term = __fdiv_rn(__fmul_rn(damping, state->dangling_sum),
                                          static_cast<float>(n_global));
    double res2 = 0.0, mass = 0.0;
    const long long first = static_cast<long long>(blockIdx.x) * kTileRows;
    for (int i = threadIdx.x; i < kTileRows && first + i < local_rows; i += kBlock) {
        const long long node = row_offset + first + i;
        const float fresh = __fadd_rn(__fadd_rn(__fmul_rn(damping, tile[i]), dangling_term), teleport);
        r_new[node] = fresh;
        const float diff = __fsub_rn(fresh, r_old[node]);
        res2 += static_cast<double>(__fmul_rn(diff, diff));
        if (dangling[node]) mass += static_cast<double>(fresh);
    }
    block_sum2(res2, mass);
    if (threadIdx.x == 0) {
        block_partials[2 * blockIdx.x] = res2;
        block_partials[2 * blockIdx.x + 1] = mass;
    }
}

template <typename T>
hipError_t dev_alloc(T** p, long long count) {
    return hipMalloc(reinterpret_cast<void**>(p), static_cast<size_t>(std::max<long long>(count, 1)) * sizeof(T));
}

template <int LANES, bool SCATTER>
hipError_t launch_bucket(const CSRMatrix* A, const TiledPlan& plan, int* counter, const int* offs_a,
                         const int* offs_b, hipStream_t s) {
    const int rows_per_block = kBlock / LANES;
    const int grid = (A->num_rows + rows_per_block - 1) / rows_per_block;
    bucket_kernel<LANES, SCATTER><<<grid, kBlock, 0, s>>>(
        A->num_rows, plan.num_tiles, plan.num_strips, A->d_row_ptrs, A->d_col_indices, A->d_values,
        counter, offs_a, offs_b, plan.a_val, plan.a_lcol, plan.a_dst, plan.b_lrow);
    return hipGetLastError();
}

template <bool SCATTER>
hipError_t launch_bucket_lanes(int lanes, const CSRMatrix* A, const TiledPlan& plan, int* counter,
                               const int* offs_a, const int* offs_b, hipStream_t s) {
    switch (lanes) {
        case 1:  return launch_bucket<1, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        case 2:  return launch_bucket<2, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        case 4:  return launch_bucket<4, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        case 8:  return launch_bucket<8, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        case 16: return launch_bucket<16, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        case 32: return launch_bucket<32, SCATTER>(A, plan, counter, offs_a, offs_b, s);
        default: return launch_bucket<64, SCATTER>(A, plan, counter, offs_a, offs_b, s);
    }
}

} // namespace

bool tiled_eligible(const CSRMatrix* A) {
    static const bool enabled = [] {
        const char* env = std::getenv("SPMV_TILED");
        return !(env && env[0] == '0');
    }();
    static const long long min_cols = [] {
        const char* env = std::getenv("SPMV_TILED_MIN_COLS");
        return env ? std::atoll(env) : 262144LL;    // below this x sits in L2 and the direct gather wins
    }();
    if (!enabled || !A || A->num_rows <= 0 || A->nnz < (1 << 20) || A->num_cols < min_cols) return false;
    const long long strips = (static_cast<long long>(A->num_cols) + kStripCols - 1) / kStripCols;
    const long long tiles = (static_cast<long long>(A->num_rows) + kTileRows - 1) / kTileRows;
    return strips * tiles <= kMaxCells;
}

void tiled_free(TiledPlan* p) {
    if (!p) return;
    void* owned[] = {p->a_val, p->a_lcol, p->a_dst, p->b_lrow, p->prod, p->tile_begin, p->items};
    for (void* q : owned) if (q) (void)hipFree(q);
    delete p;
}

hipError_t tiled_build(const CSRMatrix* A, TiledPlan** out, hipStream_t s) {
    *out = nullptr;
    TiledPlan* plan = new TiledPlan();
    plan->num_rows = A->num_rows;
    plan->num_cols = A->num_cols;
    plan->nnz = A->nnz;
    plan->strip_cols = kStripCols;
    plan->tile_rows = kTileRows;
    plan->num_strips = (A->num_cols + kStripCols - 1) / kStripCols;
    plan->num_tiles = (A->num_rows + kTileRows - 1) / kTileRows;
    const long long cells = static_cast<long long>(plan->num_strips) * plan->num_tiles;

    int *cnt = nullptr, *cnt_t = nullptr, *offs_a = nullptr, *offs_b = nullptr, *strip_begin = nullptr;
    auto cleanup = [&](hipError_t e) {
        for (int* q : {cnt, cnt_t, offs_a, offs_b, strip_begin}) if (q) (void)hipFree(q);
        if (e != hipSuccess) tiled_free(plan);
        return e;
    };

    hipError_t e = dev_alloc(&cnt, cells);
    if (e == hipSuccess) e = dev_alloc(&cnt_t, cells);
    if (e == hipSuccess) e = dev_alloc(&offs_a, cells + 1);
    if (e == hipSuccess) e = dev_alloc(&offs_b, cells + 1);
    if (e == hipSuccess) e = dev_alloc(&strip_begin, plan->num_strips + 1);
    if (e == hipSuccess) e = dev_alloc(&plan->a_val, plan->nnz);
    if (e == hipSuccess) e = dev_alloc(&plan->a_lcol, plan->nnz + 4);
    if (e == hipSuccess) e = dev_alloc(&plan->a_dst, plan->nnz);
    if (e == hipSuccess) e = dev_alloc(&plan->b_lrow, plan->nnz + 4);
    if (e == hipSuccess) e = dev_alloc(&plan->prod, plan->nnz);
    if (e == hipSuccess) e = dev_alloc(&plan->tile_begin, plan->num_tiles + 1);
    if (e != hipSuccess) return cleanup(e);

    const int lanes = std::min(pick_lanes_per_row(static_cast<float>(A->nnz) / A->num_rows) * 4, 64);
    const int small_grid = static_cast<int>(std::min<long long>((cells + kBlock - 1) / kBlock, 4096));

    e = hipMemsetAsync(cnt, 0, cells * sizeof(int), s);
    if (e == hipSuccess) e = launch_bucket_lanes<false>(lanes, A, *plan, cnt, nullptr, nullptr, s);
    if (e == hipSuccess) {
        exclusive_scan_kernel<<<1, 1024, 0, s>>>(cnt, cells, offs_a);
        transpose_counts_kernel<<<small_grid, kBlock, 0, s>>>(cnt, plan->num_strips, plan->num_tiles, cnt_t);
        exclusive_scan_kernel<<<1, 1024, 0, s>>>(cnt_t, cells, offs_b);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, cells * sizeof(int), s);
    if (e == hipSuccess) e = launch_bucket_lanes<true>(lanes, A, *plan, cnt, offs_a, offs_b, s);
    if (e == hipSuccess) {
        const int n = std::max(plan->num_strips, plan->num_tiles) + 1;
        boundaries_kernel<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(
            offs_a, offs_b, plan->num_strips, plan->num_tiles, strip_begin, plan->tile_begin);
        e = hipGetLastError();
    }
    std::vector<int> host_strip(plan->num_strips + 1);
    if (e == hipSuccess) e = hipMemcpyAsync(host_strip.data(), strip_begin, host_strip.size() * sizeof(int),
                                            hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(e);

    // phase-1 work items: every strip's range cut into pieces of <= kItemEntries
    std::vector<int> items;
    for (int strip = 0; strip < plan->num_strips; ++strip) {
        for (int b = host_strip[strip]; b < host_strip[strip + 1]; b += kItemEntries) {
            items.push_back(strip);
            items.push_back(b);
            items.push_back(std::min(b + kItemEntries, host_strip[strip + 1]));
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

hipError_t tiled_spmv(const TiledPlan& plan, const float* d_x, float* d_y, hipStream_t s) {
    if (plan.num_items > 0) {
        tiled_expand_kernel<<<plan.num_items, kBlock, 0, s>>>(plan.items, plan.a_val, plan.a_lcol, plan.a_dst,
                                                              d_x, plan.num_cols, plan.prod);
    }
    tiled_reduce_kernel<<<plan.num_tiles, kBlock, 0, s>>>(plan.tile_begin, plan.prod, plan.b_lrow,
                                                          plan.num_rows, d_y);
    return hipGetLastError();
}

hipError_t tiled_pagerank_step(const TiledPlan& plan, int row_offset, int n_global,
                               const float* d_r_old, float* d_r_new,
                               const unsigned char* d_dangling, float damping,
                               const PrState* d_state, double* d_block_partials, hipStream_t s) {
    // After convergence the reduce kernel returns before touching r_new; the expand kernel
    // then only rewrites the scratch product stream, which nothing reads.
    if (plan.num_items > 0) {
        tiled_expand_kernel<<<plan.num_items, kBlock, 0, s>>>(plan.items, plan.a_val, plan.a_lcol, plan.a_dst,
                                                              d_r_old, plan.num_cols, plan.prod);
    }
    tiled_pagerank_reduce_kernel<<<plan.num_tiles, kBlock, 0, s>>>(
        plan.tile_begin, plan.prod, plan.b_lrow, plan.num_rows, row_offset, n_global, d_r_old, d_r_new,
        d_dangling, damping, d_state, d_block_partials);
    return hipGetLastError();
}

} // namespace detail
} // namespace spmv
