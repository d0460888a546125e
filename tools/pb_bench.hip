// pb_bench.hip — upper-bound microbenchmark for the two-phase LDS-tiled SpMV:
//   phase 1: stream (value f32, local column u16), gather x from an LDS-resident strip,
//            write the product stream;
//   phase 2: stream (product f32, local row u16), accumulate into an LDS-resident y tile
//            (LDS float atomics), write the tile.
// Entries are synthetic (random local indices); only rates matter here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

template <int W>   // strip width in floats
__global__ __launch_bounds__(256) void phase1(const float* __restrict__ vals, const unsigned short* __restrict__ lcol,
                                              const float* __restrict__ x, float* __restrict__ prod,
                                              long long per_wg) {
    extern __shared__ float xs[];
    const long long begin = (long long)blockIdx.x * per_wg;
    const float* strip = x + ((long long)blockIdx.x * 977 % 600) * W;
    for (int i = threadIdx.x * 4; i < W; i += 256 * 4)
        *reinterpret_cast<f32x4*>(xs + i) = *reinterpret_cast<const f32x4*>(strip + i);
    __syncthreads();
    for (long long i = begin + threadIdx.x * 4; i < begin + per_wg; i += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(vals + i);
        const u16x4 c = *reinterpret_cast<const u16x4*>(lcol + i);
        f32x4 p;
        p[0] = v[0] * xs[c[0] & (W - 1)]; p[1] = v[1] * xs[c[1] & (W - 1)];
        p[2] = v[2] * xs[c[2] & (W - 1)]; p[3] = v[3] * xs[c[3] & (W - 1)];
        *reinterpret_cast<f32x4*>(prod + i) = p;
    }
}

// float add on an LDS word by compare-and-swap on its integer image (what csrc/tiled.hip uses)
__device__ __forceinline__ void cas_add(float* slot, float v) {
    unsigned int* word = reinterpret_cast<unsigned int*>(slot);
    unsigned int seen = *word;
    for (;;) {
        const unsigned int got = atomicCAS(word, seen, __float_as_uint(__uint_as_float(seen) + v));
        if (got == seen) break;
        seen = got;
    }
}

template <int R>   // tile height in floats
__global__ __launch_bounds__(256) void phase2(const float* __restrict__ prod, const unsigned short* __restrict__ lrow,
                                              float* __restrict__ y, long long per_wg) {
    extern __shared__ float ys[];
    for (int i = threadIdx.x; i < R; i += 256) ys[i] = 0.f;
    __syncthreads();
    const long long begin = (long long)blockIdx.x * per_wg;
    for (long long i = begin + threadIdx.x * 4; i < begin + per_wg; i += 256 * 4) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(prod + i);
        const u16x4 r = *reinterpret_cast<const u16x4*>(lrow + i);
        cas_add(&ys[r[0] & (R - 1)], p[0]); cas_add(&ys[r[1] & (R - 1)], p[1]);
        cas_add(&ys[r[2] & (R - 1)], p[2]); cas_add(&ys[r[3] & (R - 1)], p[3]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += 256) y[(long long)blockIdx.x * R + i] = ys[i];
}

__global__ void fill(unsigned short* a, float* v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        a[i] = (unsigned short)z; v[i] = 1.0f;
    }
}

template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    const long long N = 160LL * 1000 * 1000;
    float *vals, *prod, *x, *y; unsigned short *lcol, *lrow;
    hipMalloc(&vals, N * 4); hipMalloc(&prod, N * 4); hipMalloc(&lcol, N * 2); hipMalloc(&lrow, N * 2);
    hipMalloc(&x, 40u << 20); hipMalloc(&y, 9760ull * 8192 * 4);   // the largest wgs x R below
    fill<<<4096, 256>>>(lcol, vals, N); fill<<<4096, 256>>>(lrow, prod, N); hipMemset(x, 0, 40u << 20);
    hipDeviceSynchronize();
    printf("N = %lld entries\n", N);
    for (int wgs : {610, 1220, 2440, 4880, 9760}) {
        const long long per = (N / wgs) & ~1023LL;
        hipFuncSetAttribute((const void*)phase1<16384>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        float t = timeit([&] { phase1<16384><<<wgs, 256, 65536>>>(vals, lcol, x, prod, per); });
        printf("phase1 W=16K wgs=%5d  %8.1f us  %.2f TB/s (10 B/entry)\n", wgs, t * 1e3, per * wgs * 10.0 / t / 1e9);
        float t8 = timeit([&] { phase1<8192><<<wgs, 256, 32768>>>(vals, lcol, x, prod, per); });
        printf("phase1 W= 8K wgs=%5d  %8.1f us  %.2f TB/s\n", wgs, t8 * 1e3, per * wgs * 10.0 / t8 / 1e9);
        float t2 = timeit([&] { phase2<8192><<<wgs, 256, 32768>>>(prod, lrow, y, per); });
        printf("phase2 R= 8K wgs=%5d  %8.1f us  %.2f TB/s (6 B/entry)\n", wgs, t2 * 1e3, per * wgs * 6.0 / t2 / 1e9);
        float t3 = timeit([&] { phase2<4096><<<wgs, 256, 16384>>>(prod, lrow, y, per); });
        printf("phase2 R= 4K wgs=%5d  %8.1f us  %.2f TB/s\n", wgs, t3 * 1e3, per * wgs * 6.0 / t3 / 1e9);
        fflush(stdout);
    }
    return 0;
}
