// kernels.hip — hand-written SpMV kernels for gfx950 (CDNA4, wave64) and their
// launch wrappers.  No MFMA anywhere: SpMV is ~0.23 flop/byte and bound by HBM
// (matrix stream) and by the gather of x through L2 / Infinity Cache.
//
// What each kernel replaces in the reference (behaviour, not code):
//   csr_vector_kernel   <- spmv_csr_vector_kernel   src/spmv_kernels.cu:133-165
//   csr_stream_kernel   <- spmv_csr_scalar_kernel   src/spmv_kernels.cu:168-188
//   merge_*             <- spmv_csr_merge_path_kernel + merge_path_search
//                          src/spmv_kernels.cu:48-130 (whose results are wrong,
//                          built from the published algorithm)
//   ell_kernel*         <- spmv_ell_kernel          src/spmv_kernels.cu:191-213
// Layouts in HBM are the reference's: CSR (row_ptrs / col_indices / values,
// int32 / fp32) and column-major ELL with -1 padding.
#include "internal.h"
#include "device_common.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

namespace spmv {
namespace detail {

namespace {

using namespace dev;

// ---------------------------------------------------------------------------
// VECTOR_CSR: LANES lanes per row, four entries per lane per step (dwordx4
// loads of col_indices / values, 16-byte aligned), butterfly reduction.
// LANES = 4 covers a 16-entry row in one step with 16 rows per wavefront.
// ---------------------------------------------------------------------------
template <int LANES>
__global__ __launch_bounds__(kBlock)
void csr_vector_kernel(int num_rows, long long nnz,
                       const int* __restrict__ row_ptrs,
                       const int* __restrict__ cols,
                       const float* __restrict__ vals,
                       const float* __restrict__ x,
                       float* __restrict__ y) {
    constexpr int kRowsPerBlock = kBlock / LANES;
    const int lane = threadIdx.x % LANES;
    const int slot = threadIdx.x / LANES;

    for (long long first = static_cast<long long>(blockIdx.x) * kRowsPerBlock; first < num_rows;
         first += static_cast<long long>(gridDim.x) * kRowsPerBlock) {
        const long long row = first + slot;
        float acc = 0.0f;
        if (row < num_rows) {
            const int begin = row_ptrs[row];
            const int end = row_ptrs[row + 1];
            acc = row_partial_dot<LANES>(begin, end, lane, nnz, cols, vals, x);
        }
        acc = group_sum<LANES>(acc);
        if (lane == 0 && row < num_rows) y[row] = acc;
    }
}

// VECTOR_CSR with the whole of x resident in LDS (use_texture on matrices with at most
// 32 K columns): every workgroup (1024 threads, one per CU) copies x into LDS once and then
// walks its rows as csr_vector_kernel does, gathering from LDS (~7 lanes/clk/CU) instead of
// through the vector-memory path (<= 0.3 lane/clk/CU).
template <int LANES>
__global__ __launch_bounds__(1024)
void csr_vector_ldsx_kernel(int num_rows, int num_cols, long long nnz,
                            const int* __restrict__ row_ptrs,
                            const int* __restrict__ cols,
                            const float* __restrict__ vals,
                            const float* __restrict__ x,
                            float* __restrict__ y) {
    extern __shared__ float xs[];
    if ((reinterpret_cast<unsigned long long>(x) & 15) == 0) {
        for (int i = threadIdx.x * 4; i < num_cols; i += 1024 * 4) {
            if (i + 3 < num_cols) {
                *reinterpret_cast<f32x4*>(xs + i) = *reinterpret_cast<const f32x4*>(x + i);
            } else {
                for (int k = i; k < num_cols; ++k) xs[k] = x[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < num_cols; i += 1024) xs[i] = x[i];
    }
    __syncthreads();

    constexpr int kRowsPerBlock = 1024 / LANES;
    const int lane = threadIdx.x % LANES;
    const int slot = threadIdx.x / LANES;
    for (long long first = static_cast<long long>(blockIdx.x) * kRowsPerBlock; first < num_rows;
         first += static_cast<long long>(gridDim.x) * kRowsPerBlock) {
        const long long row = first + slot;
        float acc = 0.0f;
        if (row < num_rows) {
            acc = row_partial_dot<LANES>(row_ptrs[row], row_ptrs[row + 1], lane, nnz, cols, vals, xs);
        }
        acc = group_sum<LANES>(acc);
        if (lane == 0 && row < num_rows) y[row] = acc;
    }
}

// ---------------------------------------------------------------------------
// SCALAR_CSR ("stream"): the workgroup streams its rows' entries with
// coalesced loads, parks the rounded products in LDS, then one thread per row
// adds them left to right — the CPU's summation order, with separate
// multiply and add roundings (no FMA), so results are bit-identical to
// spmv_cpu_csr.  LDS indices are skewed by idx/32 to spread row strides over
// the banks.
// ---------------------------------------------------------------------------
constexpr int kStreamChunk = 2048;
__device__ __forceinline__ int skew(int i) { return i + (i >> 5); }

__global__ __launch_bounds__(kBlock)
void csr_stream_kernel(int num_rows,
                       const int* __restrict__ row_ptrs,
                       const int* __restrict__ cols,
                       const float* __restrict__ vals,
                       const float* __restrict__ x,
                       float* __restrict__ y) {
    __shared__ float products[kStreamChunk + kStreamChunk / 32 + 1];

    for (long long first = static_cast<long long>(blockIdx.x) * kBlock; first < num_rows;
         first += static_cast<long long>(gridDim.x) * kBlock) {
        const long long row = first + threadIdx.x;
        const long long last = min(first + kBlock, static_cast<long long>(num_rows));
        const int tile_begin = row_ptrs[first];
        const int tile_end = row_ptrs[last];

        int cursor = 0, my_end = 0;
        if (row < num_rows) {
            cursor = row_ptrs[row];
            my_end = row_ptrs[row + 1];
        }
        float acc = 0.0f;

        for (int base = tile_begin; base < tile_end; base += kStreamChunk) {
            const int count = min(kStreamChunk, tile_end - base);
            for (int i = threadIdx.x; i < count; i += kBlock) {
                products[skew(i)] = __fmul_rn(vals[base + i], x[cols[base + i]]);
            }
            __syncthreads();
            const int stop = min(my_end, base + count);
            while (cursor < stop) {
                acc = __fadd_rn(acc, products[skew(cursor - base)]);
                ++cursor;
            }
            __syncthreads();
        }
        if (row < num_rows) y[row] = acc;
    }
}

// ---------------------------------------------------------------------------
// ELL: one thread per row walks the K column-major slabs in order (CPU order,
// unfused multiply/add => bit-identical to spmv_cpu_ell).  When num_rows is a
// multiple of 4 a thread owns four consecutive rows and loads each slab with
// one 16-byte access; 64-bit slot arithmetic (rows*K may exceed 2^31).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void ell_kernel_x4(int num_rows, int width,
                   const int* __restrict__ cols,
                   const float* __restrict__ vals,
                   const float* __restrict__ x,
                   float* __restrict__ y) {
    const long long groups = num_rows / 4;
    for (long long g = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; g < groups;
         g += static_cast<long long>(gridDim.x) * kBlock) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const int* cp = cols + g * 4;
        const float* vp = vals + g * 4;
        int k = 0;
        for (; k + 4 <= width; k += 4) {
            i32x4 c[4];
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = *reinterpret_cast<const i32x4*>(cp + static_cast<long long>(k + u) * num_rows);
                v[u] = *reinterpret_cast<const f32x4*>(vp + static_cast<long long>(k + u) * num_rows);
            }
            float xv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[u][r] = x[c[u][r] >= 0 ? c[u][r] : 0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c[u][r] >= 0) acc[r] = __fadd_rn(acc[r], __fmul_rn(v[u][r], xv[u][r]));
        }
        for (; k < width; ++k) {
            const i32x4 c = *reinterpret_cast<const i32x4*>(cp + static_cast<long long>(k) * num_rows);
            const f32x4 v = *reinterpret_cast<const f32x4*>(vp + static_cast<long long>(k) * num_rows);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c[r] >= 0) acc[r] = __fadd_rn(acc[r], __fmul_rn(v[r], x[c[r]]));
        }
        *reinterpret_cast<f32x4*>(y + g * 4) = acc;
    }
}

__global__ __launch_bounds__(kBlock)
void ell_kernel_x1(int num_rows, int width,
                   const int* __restrict__ cols,
                   const float* __restrict__ vals,
                   const float* __restrict__ x,
                   float* __restrict__ y) {
    for (long long row = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; row < num_rows;
         row += static_cast<long long>(gridDim.x) * kBlock) {
        float acc = 0.0f;
        const int* cp = cols + row;
        const float* vp = vals + row;
        int k = 0;
        for (; k + 4 <= width; k += 4) {
            int c[4];
            float v[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[u] = cp[static_cast<long long>(k + u) * num_rows];
                v[u] = vp[static_cast<long long>(k + u) * num_rows];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = x[c[u] >= 0 ? c[u] : 0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c[u] >= 0) acc = __fadd_rn(acc, __fmul_rn(v[u], xv[u]));
        }
        for (; k < width; ++k) {
            const int c = cp[static_cast<long long>(k) * num_rows];
            if (c >= 0) acc = __fadd_rn(acc, __fmul_rn(vp[static_cast<long long>(k) * num_rows], x[c]));
        }
        y[row] = acc;
    }
}

// ---------------------------------------------------------------------------
// MERGE_PATH (Merrill & Garland): the (rows + nnz) merge items of "row-end
// offsets" vs "entry indices" are cut into tiles of kMergeTile items; every
// workgroup owns one tile, every thread kMergeItems consecutive items.  No
// atomics and no memset: every y[row] is stored exactly once by the thread
// that consumes the row's end item; partial sums of rows cut by thread
// boundaries travel through an LDS segmented scan, partial sums of rows cut by
// tile boundaries through (row, value) carry slots added by merge_fixup in a
// fixed order => deterministic.
// ---------------------------------------------------------------------------
constexpr int kMergeItems = 7;                       // odd: conflict-free LDS walk
constexpr int kMergeTile = kBlock * kMergeItems;     // 1792 items per workgroup

// number of row-end items among the first `diag` merge items
__device__ __forceinline__ int merge_split(const int* __restrict__ row_end, int num_rows, int nnz,
                                           long long diag) {
    long long lo = diag > nnz ? diag - nnz : 0;
    long long hi = diag < num_rows ? diag : num_rows;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (row_end[mid] <= diag - mid - 1) lo = mid + 1; else hi = mid;
    }
    return static_cast<int>(lo);
}

__global__ __launch_bounds__(kBlock)
void merge_partition_kernel(int num_rows, int nnz, const int* __restrict__ row_ptrs,
                            int num_tiles, int* __restrict__ tile_rows) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > num_tiles) return;
    const long long total = static_cast<long long>(num_rows) + nnz;
    const long long diag = min(static_cast<long long>(t) * kMergeTile, total);
    tile_rows[t] = merge_split(row_ptrs + 1, num_rows, nnz, diag);
}

__global__ __launch_bounds__(kBlock)
void merge_tile_kernel(int num_rows, int nnz,
                       const int* __restrict__ row_ptrs,
                       const int* __restrict__ cols,
                       const float* __restrict__ vals,
                       const float* __restrict__ x,
                       const int* __restrict__ tile_rows,
                       float* __restrict__ y,
                       int* __restrict__ carry_row,
                       float* __restrict__ carry_val) {
    __shared__ int   s_row_end[kMergeTile + 1];
    __shared__ float s_prod[kMergeTile];
    __shared__ int   s_key[2][kBlock];
    __shared__ float s_val[2][kBlock];

    const int tile = blockIdx.x;
    const long long total = static_cast<long long>(num_rows) + nnz;
    const long long diag0 = static_cast<long long>(tile) * kMergeTile;
    const long long diag1 = min(diag0 + kMergeTile, total);

    const int row0 = tile_rows[tile];
    const int row1 = tile_rows[tile + 1];
    const int nz0 = static_cast<int>(diag0 - row0);
    const int nz1 = static_cast<int>(diag1 - row1);
    const int n_rows = row1 - row0;     // row-end items in this tile
    const int n_nz = nz1 - nz0;         // entries in this tile

    // stage row ends (one extra: the row left open at the tile's end) and products
    for (int i = threadIdx.x; i <= n_rows; i += kBlock) {
        const int r = row0 + i;
        s_row_end[i] = r < num_rows ? row_ptrs[r + 1] : INT_MAX;
    }
    for (int i = threadIdx.x; i < n_nz; i += kBlock) {
        s_prod[i] = vals[nz0 + i] * x[cols[nz0 + i]];
    }
    __syncthreads();

    // this thread's slice of the tile's merge path
    const int items = static_cast<int>(diag1 - diag0);
    const int d_begin = min(static_cast<int>(threadIdx.x) * kMergeItems, items);
    const int d_end = min(d_begin + kMergeItems, items);

    int i;   // local row index
    {
        int lo = d_begin > n_nz ? d_begin - n_nz : 0;
        int hi = d_begin < n_rows ? d_begin : n_rows;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_row_end[mid] <= nz0 + (d_begin - mid - 1)) lo = mid + 1; else hi = mid;
        }
        i = lo;
    }
    int j = d_begin - i;   // local entry index

    float running = 0.0f;
    bool have_first = false;
    int first_row = 0;
    float first_sum = 0.0f;

    for (int step = d_begin; step < d_end; ++step) {
        if (nz0 + j < s_row_end[i]) {
            running += s_prod[j];
            ++j;
        } else {
            if (!have_first) {
                have_first = true;
                first_row = row0 + i;
                first_sum = running;
            } else {
                y[row0 + i] = running;
            }
            running = 0.0f;
            ++i;
        }
    }

    // segmented inclusive scan of the per-thread carries (key = open row)
    int cur = 0;
    s_key[0][threadIdx.x] = row0 + i;
    s_val[0][threadIdx.x] = running;
    __syncthreads();
#pragma unroll
    for (int off = 1; off < kBlock; off <<= 1) {
        const int k = s_key[cur][threadIdx.x];
        float v = s_val[cur][threadIdx.x];
        if (static_cast<int>(threadIdx.x) >= off && s_key[cur][threadIdx.x - off] == k) {
            v = s_val[cur][threadIdx.x - off] + v;
        }
        s_key[cur ^ 1][threadIdx.x] = k;
        s_val[cur ^ 1][threadIdx.x] = v;
        cur ^= 1;
        __syncthreads();
    }

    if (have_first) {
        // everything the threads to the left accumulated for my first row
        float carry_in = 0.0f;
        if (threadIdx.x > 0 && s_key[cur][threadIdx.x - 1] == first_row) {
            carry_in = s_val[cur][threadIdx.x - 1];
        }
        y[first_row] = carry_in + first_sum;
    }
    if (threadIdx.x == kBlock - 1) {
        carry_row[tile] = s_key[cur][kBlock - 1];
        carry_val[tile] = s_val[cur][kBlock - 1];
    }
}

// Adds the tile carries: the first tile of every run of equal carry rows sums
// the run left to right and updates y once.
__global__ __launch_bounds__(kBlock)
void merge_fixup_kernel(int num_rows, int num_tiles,
                        const int* __restrict__ carry_row,
                        const float* __restrict__ carry_val,
                        float* __restrict__ y) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= num_tiles) return;
    const int row = carry_row[t];
    if (row >= num_rows) return;
    if (t > 0 && carry_row[t - 1] == row) return;
    float sum = carry_val[t];
    for (int u = t + 1; u < num_tiles && carry_row[u] == row; ++u) sum += carry_val[u];
    y[row] += sum;
}

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void fill_zero_kernel(float* __restrict__ y, size_t n) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        y[i] = 0.0f;
    }
}

__global__ __launch_bounds__(kBlock)
void ell_count_kernel(const int* __restrict__ cols, size_t slots, unsigned long long* __restrict__ out) {
    unsigned long long local = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < slots;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        local += cols[i] >= 0;
    }
    // wave reduce then one atomic per wavefront
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

__global__ __launch_bounds__(kBlock)
void row_stats_kernel(const int* __restrict__ row_ptrs, int num_rows, int* __restrict__ out /*[max, min]*/) {
    int longest = 0, shortest = INT_MAX;
    for (long long r = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; r < num_rows;
         r += static_cast<long long>(gridDim.x) * kBlock) {
        const int len = row_ptrs[r + 1] - row_ptrs[r];
        longest = max(longest, len);
        shortest = min(shortest, len);
    }
    for (int off = 32; off > 0; off >>= 1) {
        longest = max(longest, __shfl_xor(longest, off, 64));
        shortest = min(shortest, __shfl_xor(shortest, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], longest);
        atomicMin(&out[1], shortest);
    }
}

// CSR -> column-major ELL on the device (reference src/ell_matrix.cpp:139-156 is a host loop):
// LANES lanes per row copy its entries into slabs k = 0.. and pad the rest with (-1, 0.0f).
__global__ __launch_bounds__(kBlock)
void ell_from_csr_kernel(int num_rows, int width, const int* __restrict__ row_ptrs,
                         const int* __restrict__ cols, const float* __restrict__ vals,
                         int* __restrict__ ell_cols, float* __restrict__ ell_vals) {
    for (long long row = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; row < num_rows;
         row += static_cast<long long>(gridDim.x) * kBlock) {
        const int begin = row_ptrs[row];
        const int len = row_ptrs[row + 1] - begin;
        for (int k = 0; k < width; ++k) {
            const long long slot = static_cast<long long>(k) * num_rows + row;    // coalesced over rows
            ell_cols[slot] = k < len ? cols[begin + k] : -1;
            ell_vals[slot] = k < len ? vals[begin + k] : 0.0f;
        }
    }
}

inline int capped_grid(long long work_items, int per_block) {
    const long long blocks = (work_items + per_block - 1) / per_block;
    return static_cast<int>(std::max(1LL, std::min<long long>(blocks, kMaxResidentBlocks)));
}

template <int LANES>
hipError_t launch_vector(const CSRMatrix* A, const float* d_x, float* d_y, hipStream_t s) {
    const int grid = capped_grid(A->num_rows, kBlock / LANES);
    csr_vector_kernel<LANES><<<grid, kBlock, 0, s>>>(A->num_rows, A->nnz, A->d_row_ptrs,
                                                     A->d_col_indices, A->d_values, d_x, d_y);
    return hipGetLastError();
}

template <int LANES>
hipError_t launch_vector_ldsx(const CSRMatrix* A, const float* d_x, float* d_y, int grid, hipStream_t s) {
    const size_t lds = (static_cast<size_t>(A->num_cols) * sizeof(float) + 15) & ~size_t(15);
    // > 64 KiB of dynamic LDS needs the limit raised; the attribute is per DEVICE and cheap to set, so it is
    // set on every launch (no process-wide flag to go stale after hipSetDevice, nothing shared between threads)
    const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_vector_ldsx_kernel<LANES>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (raised != hipSuccess) return raised;
    csr_vector_ldsx_kernel<LANES><<<grid, 1024, lds, s>>>(A->num_rows, A->num_cols, A->nnz, A->d_row_ptrs,
                                                          A->d_col_indices, A->d_values, d_x, d_y);
    return hipGetLastError();
}

} // namespace

// x fits one CU's LDS and there are enough entries that 64+ workgroups each copying x is
// noise next to the matrix stream: returns the grid to use, or 0 when not worthwhile.
int vector_ldsx_grid(const CSRMatrix* A) {
    if (A->num_cols <= 0 || A->num_cols > 32768 || A->num_rows < 4096) return 0;
    const long long matrix_bytes = static_cast<long long>(A->nnz) * 8;
    const long long x_bytes = static_cast<long long>(A->num_cols) * 4;
    const long long grid = std::min<long long>(256, matrix_bytes / (8 * x_bytes));
    return grid >= 64 ? static_cast<int>(grid) : 0;
}

hipError_t launch_csr_vector_ldsx(const CSRMatrix* A, const float* d_x, float* d_y, int lanes, int grid,
                                  hipStream_t s) {
    switch (lanes) {
        case 1:  return launch_vector_ldsx<1>(A, d_x, d_y, grid, s);
        case 2:  return launch_vector_ldsx<2>(A, d_x, d_y, grid, s);
        case 4:  return launch_vector_ldsx<4>(A, d_x, d_y, grid, s);
        case 8:  return launch_vector_ldsx<8>(A, d_x, d_y, grid, s);
        case 16: return launch_vector_ldsx<16>(A, d_x, d_y, grid, s);
        case 32: return launch_vector_ldsx<32>(A, d_x, d_y, grid, s);
        default: return launch_vector_ldsx<64>(A, d_x, d_y, grid, s);
    }
}

int pick_lanes_per_row(float avg) {
    // each lane takes four entries per step: aim for one step per row
    int lanes = 1;
    while (lanes < 64 && lanes * 4 < avg) lanes <<= 1;
    return lanes;
}

hipError_t launch_fill_zero(float* d_y, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    fill_zero_kernel<<<capped_grid(static_cast<long long>(n), kBlock), kBlock, 0, s>>>(d_y, n);
    return hipGetLastError();
}

hipError_t launch_csr_scalar(const CSRMatrix* A, const float* d_x, float* d_y, hipStream_t s) {
    const int grid = capped_grid(A->num_rows, kBlock);
    csr_stream_kernel<<<grid, kBlock, 0, s>>>(A->num_rows, A->d_row_ptrs, A->d_col_indices,
                                              A->d_values, d_x, d_y);
    return hipGetLastError();
}

hipError_t launch_csr_vector(const CSRMatrix* A, const float* d_x, float* d_y,
                             int lanes, hipStream_t s) {
    switch (lanes) {
        case 1:  return launch_vector<1>(A, d_x, d_y, s);
        case 2:  return launch_vector<2>(A, d_x, d_y, s);
        case 4:  return launch_vector<4>(A, d_x, d_y, s);
        case 8:  return launch_vector<8>(A, d_x, d_y, s);
        case 16: return launch_vector<16>(A, d_x, d_y, s);
        case 32: return launch_vector<32>(A, d_x, d_y, s);
        default: return launch_vector<64>(A, d_x, d_y, s);
    }
}

// the tile table of the merge-path kernels: built on first use (or ahead of a timed call), cached with the matrix
static hipError_t prepare_csr_merge_locked(const CSRMatrix* A, CsrAux* aux, hipStream_t s) {
    const long long total = static_cast<long long>(A->num_rows) + A->nnz;
    const int num_tiles = static_cast<int>((total + kMergeTile - 1) / kMergeTile);
    if (num_tiles == 0 || !aux) return hipSuccess;
    if (aux->num_tiles == num_tiles && aux->tile_items == kMergeTile && aux->d_tile_rows) return hipSuccess;
    if (aux->d_tile_rows) (void)hipFree(aux->d_tile_rows);
    if (aux->d_carry_row) (void)hipFree(aux->d_carry_row);
    if (aux->d_carry_val) (void)hipFree(aux->d_carry_val);
    for (const CsrAux::MergeCarry& c : aux->extra_carry) {
        if (c.row) (void)hipFree(c.row);
        if (c.val) (void)hipFree(c.val);
    }
    aux->extra_carry.clear();
    aux->carry_taken = false;
    aux->d_tile_rows = nullptr;
    aux->d_carry_row = nullptr;
    aux->d_carry_val = nullptr;
    aux->num_tiles = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&aux->d_tile_rows), (num_tiles + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&aux->d_carry_row), num_tiles * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&aux->d_carry_val), num_tiles * sizeof(float));
    if (e != hipSuccess) return e;
    merge_partition_kernel<<<(num_tiles + 1 + kBlock - 1) / kBlock, kBlock, 0, s>>>(
        A->num_rows, A->nnz, A->d_row_ptrs, num_tiles, aux->d_tile_rows);
    e = hipGetLastError();
    // the table is read by calls on ANY stream from now on: finish it before it is announced (once per matrix)
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    aux->num_tiles = num_tiles;
    aux->tile_items = kMergeTile;
    return hipSuccess;
}

hipError_t prepare_csr_merge(const CSRMatrix* A, CsrAux* aux, hipStream_t s) {
    if (!aux) return hipSuccess;
    std::lock_guard<std::mutex> guard(aux->merge_lock);
    return prepare_csr_merge_locked(A, aux, s);
}

constexpr size_t kMaxExtraCarry = 7;      // streams (beyond the first) that keep a carry pair with the matrix
hipError_t launch_csr_merge(const CSRMatrix* A, CsrAux* aux, const float* d_x, float* d_y,
                            hipStream_t s) {
    const long long total = static_cast<long long>(A->num_rows) + A->nnz;
    const int num_tiles = static_cast<int>((total + kMergeTile - 1) / kMergeTile);
    if (num_tiles == 0) return hipSuccess;
    // (the lock also keeps the tile + fix-up pair of one call together when two host threads share a stream)
    std::lock_guard<std::mutex> guard(aux->merge_lock);
    hipError_t prepared = prepare_csr_merge_locked(A, aux, s);
    if (prepared != hipSuccess) return prepared;

    // the carry-out slots of THIS stream
    int* carry_row = nullptr;
    float* carry_val = nullptr;
    if (!aux->carry_taken) {
        aux->carry_taken = true;
        aux->carry_stream = s;
    }
    if (aux->carry_stream == s) {
        carry_row = aux->d_carry_row;
        carry_val = aux->d_carry_val;
    } else {
        for (const CsrAux::MergeCarry& c : aux->extra_carry) {
            if (c.stream == s) {
                carry_row = c.row;
                carry_val = c.val;
            }
        }
        if (!carry_row && aux->extra_carry.size() < kMaxExtraCarry) {
            CsrAux::MergeCarry fresh{s, nullptr, nullptr};
            if (malloc_any_time(reinterpret_cast<void**>(&fresh.row), num_tiles * sizeof(int)) != hipSuccess ||
                malloc_any_time(reinterpret_cast<void**>(&fresh.val), num_tiles * sizeof(float)) != hipSuccess) {
                if (fresh.row) (void)hipFree(fresh.row);
                return hipErrorOutOfMemory;
            }
            aux->extra_carry.push_back(fresh);
            carry_row = fresh.row;
            carry_val = fresh.val;
        }
    }
    // Past kMaxExtraCarry streams (a caller that makes a new stream per call would otherwise grow the list for ever —
    // and a kept pair cannot be handed to another stream while its owner may still be running): this call borrows a
    // pair from the stream-ordered allocator and gives it back behind its own kernels.
    bool borrowed = false;
    if (!carry_row) {
        if (hipMallocAsync(reinterpret_cast<void**>(&carry_row), num_tiles * sizeof(int), s) != hipSuccess ||
            hipMallocAsync(reinterpret_cast<void**>(&carry_val), num_tiles * sizeof(float), s) != hipSuccess) {
            (void)hipGetLastError();
            if (carry_row) (void)hipFreeAsync(carry_row, s);
            return hipErrorOutOfMemory;
        }
        borrowed = true;
    }

    merge_tile_kernel<<<num_tiles, kBlock, 0, s>>>(A->num_rows, A->nnz, A->d_row_ptrs,
                                                   A->d_col_indices, A->d_values, d_x,
                                                   aux->d_tile_rows, d_y, carry_row, carry_val);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (borrowed) { (void)hipFreeAsync(carry_row, s); (void)hipFreeAsync(carry_val, s); }
        return e;
    }
    merge_fixup_kernel<<<(num_tiles + kBlock - 1) / kBlock, kBlock, 0, s>>>(
        A->num_rows, num_tiles, carry_row, carry_val, d_y);
    e = hipGetLastError();
    if (borrowed) {
        (void)hipFreeAsync(carry_row, s);
        (void)hipFreeAsync(carry_val, s);
    }
    return e;
}

hipError_t launch_ell(const ELLMatrix* A, const float* d_x, float* d_y, hipStream_t s) {
    if (A->num_rows % 4 == 0) {
        const int grid = capped_grid(A->num_rows / 4, kBlock);
        ell_kernel_x4<<<grid, kBlock, 0, s>>>(A->num_rows, A->max_nnz_per_row, A->d_col_indices,
                                              A->d_values, d_x, d_y);
    } else {
        const int grid = capped_grid(A->num_rows, kBlock);
        ell_kernel_x1<<<grid, kBlock, 0, s>>>(A->num_rows, A->max_nnz_per_row, A->d_col_indices,
                                              A->d_values, d_x, d_y);
    }
    return hipGetLastError();
}

hipError_t launch_ell_from_csr(const CSRMatrix* csr, int width, int* d_ell_cols, float* d_ell_vals,
                               hipStream_t s) {
    if (csr->num_rows == 0 || width == 0) return hipSuccess;
    ell_from_csr_kernel<<<capped_grid(csr->num_rows, kBlock), kBlock, 0, s>>>(
        csr->num_rows, width, csr->d_row_ptrs, csr->d_col_indices, csr->d_values, d_ell_cols, d_ell_vals);
    return hipGetLastError();
}

hipError_t device_count_ell_nnz(const ELLMatrix* A, long long* out, hipStream_t s) {
    const size_t slots = static_cast<size_t>(A->num_rows) * A->max_nnz_per_row;
    *out = 0;
    if (slots == 0 || !A->d_col_indices) return hipSuccess;
    unsigned long long* d_count = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_count), sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        ell_count_kernel<<<capped_grid(static_cast<long long>(slots), kBlock * 8), kBlock, 0, s>>>(
            A->d_col_indices, slots, d_count);
        e = hipGetLastError();
    }
    unsigned long long host = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&host, d_count, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_count);
    *out = static_cast<long long>(host);
    return e;
}

hipError_t device_row_stats(const int* d_row_ptrs, int num_rows, int* max_out, int* min_out,
                            hipStream_t s) {
    int* d_pair = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_pair), 2 * sizeof(int));
    if (e != hipSuccess) return e;
    const int init[2] = {0, INT_MAX};
    e = hipMemcpyAsync(d_pair, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        row_stats_kernel<<<capped_grid(num_rows, kBlock * 4), kBlock, 0, s>>>(d_row_ptrs, num_rows, d_pair);
        e = hipGetLastError();
    }
    int host[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, d_pair, sizeof(host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_pair);
    *max_out = host[0];
    *min_out = host[1];
    return e;
}

} // namespace detail
} // namespace spmv
