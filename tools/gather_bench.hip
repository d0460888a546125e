// gather_bench.hip — microbenchmark of the x-gather path on gfx950: every lane
// loads 4-byte words at random indices of a table of `n` floats (the access
// pattern of SpMV's x[col]), for table sizes from L2-resident to HBM-resident,
// with the index stream read as dwordx4 like the real kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
    if (MODE == 0) return *p;
    if (MODE == 1) return __builtin_nontemporal_load(p);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const int* __restrict__ idx, long long count,
                                              const float* __restrict__ table, float* __restrict__ out) {
    float acc = 0.f;
    const long long quads = count / 4;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const i32x4 c = *reinterpret_cast<const i32x4*>(idx + q * 4);
        acc += ld<MODE>(table + c[0]) + ld<MODE>(table + c[1]) + ld<MODE>(table + c[2]) + ld<MODE>(table + c[3]);
    }
    if (acc == 123.456f) out[0] = acc;   // keep the loads alive
}

__global__ void fill_idx(int* idx, long long count, unsigned n, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        idx[i] = (int)(((z >> 32) * n) >> 32);
    }
}

template <int MODE>
float run(const int* idx, long long count, const float* table, float* out, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) gather<MODE><<<grid, 256>>>(idx, count, table, out);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) gather<MODE><<<grid, 256>>>(idx, count, table, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const long long count = 160LL * 1000 * 1000;   // gathers per launch (C5: 160 M)
    int* idx; float* out;
    hipMalloc(&idx, count * 4); hipMalloc(&out, 4);
    const unsigned sizes[] = {16u << 10, 256u << 10, 1000000u, 4000000u, 10000000u, 40000000u, 100000000u};
    printf("%12s %10s %6s %10s %12s\n", "table_floats", "table_MB", "mode", "us", "Ggather/s");
    for (unsigned n : sizes) {
        float* table; hipMalloc(&table, (size_t)n * 4); hipMemset(table, 0, (size_t)n * 4);
        fill_idx<<<4096, 256>>>(idx, count, n, 42);
        hipDeviceSynchronize();
        for (int grid : {2048, 8192}) {
            float t0 = run<0>(idx, count, table, out, grid);
            float t1 = run<1>(idx, count, table, out, grid);
            float t2 = run<2>(idx, count, table, out, grid);
            printf("%12u %10.1f plain %10.1f %12.2f  (grid %d)\n", n, n * 4.0 / 1e6, t0 * 1e3, count / t0 / 1e6, grid);
            printf("%12u %10.1f nt    %10.1f %12.2f\n", n, n * 4.0 / 1e6, t1 * 1e3, count / t1 / 1e6);
            printf("%12u %10.1f sc1   %10.1f %12.2f\n", n, n * 4.0 / 1e6, t2 * 1e3, count / t2 / 1e6);
            fflush(stdout);
        }
        hipFree(table);
    }
    // streaming reference: the index stream alone
    return 0;
}
