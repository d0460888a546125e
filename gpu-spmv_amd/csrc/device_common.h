// device_common.h — device-side building blocks shared by kernels.hip and
// pagerank.hip (gfx950, wave64).
#ifndef SPMV_AMD_DEVICE_COMMON_H
#define SPMV_AMD_DEVICE_COMMON_H

#include <hip/hip_runtime.h>

namespace spmv {
namespace detail {
namespace dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int   i32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;          // 4 wavefronts per workgroup
constexpr int kMaxResidentBlocks = 256 * 8;   // 256 CUs x 8 workgroups of 256 threads

// ---------------------------------------------------------------------------
// cross-lane helpers (DPP inside 16-lane rows, ds_bpermute across rows)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float,
        __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over an aligned group of LANES consecutive lanes; every lane of the
// group ends up with the total (butterfly: quad perms, half-mirror, mirror,
// then xor-16 / xor-32 across DPP rows).
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LANES >= 2)  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (LANES >= 4)  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (LANES >= 8)  v += dpp<0x141>(v);   // row_half_mirror
    if constexpr (LANES >= 16) v += dpp<0x140>(v);   // row_mirror
    if constexpr (LANES >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LANES >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// 16-byte loads of four consecutive entries; the tail of the arrays (fewer
// than four entries left) falls back to guarded scalar loads so nothing is
// read past nnz.
__device__ __forceinline__ void load4(const int* __restrict__ cols, const float* __restrict__ vals,
                                      long long j, long long nnz, i32x4& c, f32x4& v) {
    if (j + 3 < nnz) {
        c = *reinterpret_cast<const i32x4*>(cols + j);
        v = *reinterpret_cast<const f32x4*>(vals + j);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = j + k < nnz;
            c[k] = ok ? cols[j + k] : 0;
            v[k] = ok ? vals[j + k] : 0.0f;
        }
    }
}

// Dot product of CSR row [begin, end) with x, spread over an aligned group of
// LANES lanes: four entries per lane per step through 16-byte aligned loads.
// Returns this lane's partial sum (reduce with group_sum<LANES>).
template <int LANES>
__device__ __forceinline__ float row_partial_dot(int begin, int end, int lane, long long nnz,
                                                 const int* __restrict__ cols,
                                                 const float* __restrict__ vals,
                                                 const float* __restrict__ x) {
    float acc = 0.0f;
    // start at the 16-byte boundary at or below `begin`
    for (long long j = (begin & ~3) + lane * 4; j < end; j += LANES * 4) {
        i32x4 c;
        f32x4 v;
        load4(cols, vals, j, nnz, c, v);
        // entries outside [begin, end) belong to neighbouring rows: masked by
        // select (not multiply-by-zero: x may hold inf / nan elsewhere)
        bool mine[4];
        float xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mine[k] = j + k >= begin && j + k < end;
            xv[k] = x[mine[k] ? c[k] : 0];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc = mine[k] ? __builtin_fmaf(v[k], xv[k], acc) : acc;
        }
    }
    return acc;
}

// block-wide sum of two doubles; result valid in thread 0
template <int BLOCK = kBlock>
__device__ __forceinline__ void block_sum2(double& a, double& b) {
    __shared__ double s_a[BLOCK / 64];
    __shared__ double s_b[BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_a[wave] = a;
        s_b[wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = s_a[0];
        b = s_b[0];
        for (int w = 1; w < BLOCK / 64; ++w) {
            a += s_a[w];
            b += s_b[w];
        }
    }
    __syncthreads();
}

} // namespace dev
} // namespace detail
} // namespace spmv

#endif
