// generators.hip — counter-based synthetic matrix / vector generators that run
// on the device, so the benchmark configs (up to 10 M rows,
// 160 M entries) are built directly in HBM instead of crossing PCIe.
//
// Everything is a pure function of (seed, row, slot): the numpy twin in
// gpu-spmv_amd/synth.py produces bit-identical arrays on the host, which is how
// the parity tests regenerate the same inputs for the CPU oracle.
// (The reference has no generator of its own beyond the dense test helper in
// include/spmv/test_utils.h:35-46, which cannot reach these sizes.)
#include "internal.h"
#include "generators.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace spmv {
namespace detail {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxUniformK = 64;

__host__ __device__ inline unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one 64-bit draw per (seed, stream, a, b)
__host__ __device__ inline unsigned long long draw(unsigned long long seed, unsigned long long stream,
                                                   unsigned long long a, unsigned long long b) {
    unsigned long long h = mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull));
    h = mix64(h ^ (a * 0x9E3779B97F4A7C15ull));
    h = mix64(h ^ (b * 0xC2B2AE3D27D4EB4Full));
    return h;
}

// uniform in [-1, 1) with 24 random bits
__host__ __device__ inline float to_unit(unsigned long long h) {
    return static_cast<float>(h >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

// uniform integer in [0, m), m < 2^32
__host__ __device__ inline unsigned int to_range(unsigned long long h, unsigned int m) {
    return static_cast<unsigned int>(((h >> 32) * static_cast<unsigned long long>(m)) >> 32);
}

constexpr unsigned long long kStreamCols = 1, kStreamVals = 2, kStreamVec = 3;

// Exactly k entries per row; the column set is a uniform random k-subset of
// [0, n_cols): draw k values in [0, n_cols - k], sort, add the rank.
// ELL = true stores the same entries column-major (slot (r, s) at s * local_rows + r).
template <bool ELL>
__global__ __launch_bounds__(kBlock)
void uniform_rows_kernel(unsigned long long seed, int row_begin, int local_rows, int n_cols, int k,
                         int* __restrict__ row_ptrs, int* __restrict__ cols,
                         float* __restrict__ vals) {
    const long long r = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x;
    if (r > local_rows) return;
    if (!ELL) row_ptrs[r] = static_cast<int>(r * k);
    if (r == local_rows) return;

    const unsigned long long row = static_cast<unsigned long long>(row_begin) + r;
    const unsigned int span = static_cast<unsigned int>(n_cols - k + 1);
    unsigned int picks[kMaxUniformK];
    for (int s = 0; s < k; ++s) {
        const unsigned int v = to_range(draw(seed, kStreamCols, row, s), span);
        int pos = s;   // insertion sort, stable for equal keys
        while (pos > 0 && picks[pos - 1] > v) {
            picks[pos] = picks[pos - 1];
            --pos;
        }
        picks[pos] = v;
    }
    for (int s = 0; s < k; ++s) {
        const long long at = ELL ? static_cast<long long>(s) * local_rows + r : r * k + s;
        cols[at] = static_cast<int>(picks[s]) + s;
        vals[at] = to_unit(draw(seed, kStreamVals, row, s));
    }
}

// Arbitrary row lengths (row_ptrs given): slot s of a row of length L picks a
// column uniformly inside stratum [s*C/L, (s+1)*C/L) => unique and ascending.
__global__ __launch_bounds__(kBlock)
void stratified_rows_kernel(unsigned long long seed, int row_begin, int local_rows, int n_cols,
                            const int* __restrict__ row_ptrs, int* __restrict__ cols,
                            float* __restrict__ vals) {
    // one wavefront per row so long rows are filled with coalesced stores
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int waves = (gridDim.x * kBlock) >> 6;
    for (long long r = wave; r < local_rows; r += waves) {
        const int begin = row_ptrs[r], end = row_ptrs[r + 1];
        const unsigned long long len = end - begin;
        const unsigned long long row = static_cast<unsigned long long>(row_begin) + r;
        for (unsigned long long s = lane; s < len; s += 64) {
            const unsigned long long lo = s * n_cols / len;
            const unsigned long long hi = (s + 1) * n_cols / len;
            const unsigned int width = static_cast<unsigned int>(hi - lo);
            cols[begin + s] = static_cast<int>(lo + to_range(draw(seed, kStreamCols, row, s), width));
            vals[begin + s] = to_unit(draw(seed, kStreamVals, row, s));
        }
    }
}

__global__ __launch_bounds__(kBlock)
void vector_kernel(unsigned long long seed, unsigned long long tag, size_t n, float* __restrict__ x) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        x[i] = to_unit(draw(seed, kStreamVec, tag, i));
    }
}

__global__ __launch_bounds__(kBlock)
void count_columns_kernel(long long nnz, const int* __restrict__ cols, int n_cols,
                          int* __restrict__ counts) {
    for (long long j = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; j < nnz;
         j += static_cast<long long>(gridDim.x) * kBlock) {
        const int c = cols[j];
        if (c >= 0 && c < n_cols) atomicAdd(&counts[c], 1);
    }
}

__global__ __launch_bounds__(kBlock)
void reciprocal_values_kernel(long long nnz, const int* __restrict__ cols,
                              const int* __restrict__ counts, float* __restrict__ vals) {
    for (long long j = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; j < nnz;
         j += static_cast<long long>(gridDim.x) * kBlock) {
        vals[j] = __fdiv_rn(1.0f, static_cast<float>(counts[cols[j]]));
    }
}

int blocks_for(long long items) {
    return static_cast<int>(std::max(1LL, std::min<long long>((items + kBlock - 1) / kBlock, 256 * 16)));
}

} // namespace

int gen_uniform_rows(unsigned long long seed, int row_begin, int local_rows, int n_cols, int k,
                     int* d_row_ptrs, int* d_cols, float* d_vals, hipStream_t s) {
    if (local_rows < 0 || k < 0 || k > kMaxUniformK || k > n_cols || !d_row_ptrs ||
        (static_cast<long long>(local_rows) * k > 0 && (!d_cols || !d_vals)) ||
        static_cast<long long>(local_rows) * k > 2147483647LL) {
        return code(SpMVError::INVALID_ARGUMENT);
    }
    const int grid = static_cast<int>((static_cast<long long>(local_rows) + 1 + kBlock - 1) / kBlock);
    uniform_rows_kernel<false><<<grid, kBlock, 0, s>>>(seed, row_begin, local_rows, n_cols, k,
                                                       d_row_ptrs, d_cols, d_vals);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

int gen_uniform_ell(unsigned long long seed, int rows, int n_cols, int k, int* d_cols, float* d_vals,
                    hipStream_t s) {
    if (rows < 0 || k < 0 || k > kMaxUniformK || k > n_cols ||
        (static_cast<long long>(rows) * k > 0 && (!d_cols || !d_vals))) {
        return code(SpMVError::INVALID_ARGUMENT);
    }
    if (rows == 0 || k == 0) return 0;
    const int grid = static_cast<int>((static_cast<long long>(rows) + kBlock - 1) / kBlock);
    uniform_rows_kernel<true><<<grid, kBlock, 0, s>>>(seed, 0, rows, n_cols, k, nullptr, d_cols, d_vals);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

int gen_stratified_rows(unsigned long long seed, int row_begin, int local_rows, int n_cols,
                        const int* d_row_ptrs, int* d_cols, float* d_vals, hipStream_t s) {
    if (local_rows < 0 || n_cols <= 0 || !d_row_ptrs) return code(SpMVError::INVALID_ARGUMENT);
    if (local_rows == 0) return 0;
    stratified_rows_kernel<<<blocks_for(static_cast<long long>(local_rows) * 64), kBlock, 0, s>>>(
        seed, row_begin, local_rows, n_cols, d_row_ptrs, d_cols, d_vals);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

int gen_vector(unsigned long long seed, unsigned long long tag, size_t n, float* d_x, hipStream_t s) {
    if (n == 0) return 0;
    if (!d_x) return code(SpMVError::INVALID_ARGUMENT);
    vector_kernel<<<blocks_for(static_cast<long long>(n)), kBlock, 0, s>>>(seed, tag, n, d_x);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

int count_columns(long long nnz, const int* d_cols, int n_cols, int* d_counts, hipStream_t s) {
    if (nnz == 0) return 0;
    if (!d_cols || !d_counts) return code(SpMVError::INVALID_ARGUMENT);
    count_columns_kernel<<<blocks_for(nnz), kBlock, 0, s>>>(nnz, d_cols, n_cols, d_counts);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

int reciprocal_values(long long nnz, const int* d_cols, const int* d_counts, float* d_vals,
                      hipStream_t s) {
    if (nnz == 0) return 0;
    if (!d_cols || !d_counts || !d_vals) return code(SpMVError::INVALID_ARGUMENT);
    reciprocal_values_kernel<<<blocks_for(nnz), kBlock, 0, s>>>(nnz, d_cols, d_counts, d_vals);
    return hipGetLastError() == hipSuccess ? 0 : code(SpMVError::KERNEL_LAUNCH);
}

} // namespace detail
} // namespace spmv
