// lds_bench.hip — LDS scatter-accumulate rates on gfx950: float atomic add, integer atomic add,
// plain read-modify-write, random vs sequential addresses.  One 256-thread workgroup per CU x 8.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int R = 8192;

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float t[R];
    double* td = reinterpret_cast<double*>(t);
    for (int i = threadIdx.x; i < R; i += 256) t[i] = 0.f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int a = RANDOM ? (s >> 8) & (R - 1) : (threadIdx.x + it * 256) & (R - 1);
        if (MODE == 0) atomicAdd(&t[a], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<int*>(&t[a]), 1);
        else if (MODE == 2) { float v = t[a]; t[a] = v + 1.0f; }
        else if (MODE == 4) {   // float add by compare-and-swap on the integer image
            unsigned* w = reinterpret_cast<unsigned*>(&t[a]);
            unsigned seen = *w;
            for (;;) {
                const unsigned want = __float_as_uint(__uint_as_float(seen) + 1.0f);
                const unsigned got = atomicCAS(w, seen, want);
                if (got == seen) break;
                seen = got;
            }
        }
        else if (MODE == 5) atomicAdd(&td[a & (R / 2 - 1)], 1.0);     // ds_add_f64 (R / 2 doubles = the same LDS bytes)
        else if (MODE == 6) { float old = atomicAdd(&t[a], 1.0f); if (old == 0.123f) out[0] = old; }   // ds_add_rtn_f32
        else { float v = t[a]; if (v == 123.f) out[0] = v; }   // read only
    }
    __syncthreads();
    if (t[threadIdx.x] == 0.12345f) out[0] = 1;
}

template <int MODE, bool RANDOM> void run(const char* name, float* out) {
    const int iters = 4096, grid = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, RANDOM><<<grid, 256>>>(out, iters);
    hipEventRecord(a);
    k<MODE, RANDOM><<<grid, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)grid * 256 * iters;
    printf("%-28s %8.1f us  %7.1f Gop/s  %.2f lanes/clk/CU (2.1 GHz)\n", name, ms * 1e3, ops / ms / 1e6,
           ops / ms / 1e6 / 256 / 2.1);
}

int main() {
    float* out; hipMalloc(&out, 4);
    run<0, true>("ds_add_f32 random", out);
    run<0, false>("ds_add_f32 sequential", out);
    run<1, true>("ds_add_u32 random", out);
    run<1, false>("ds_add_u32 sequential", out);
    run<5, true>("ds_add_f64 random", out);
    run<5, false>("ds_add_f64 sequential", out);
    run<6, true>("ds_add_rtn_f32 random", out);
    run<4, true>("cas float add random", out);
    run<4, false>("cas float add sequential", out);
    run<2, true>("read+write random", out);
    run<2, false>("read+write sequential", out);
    run<3, true>("read random", out);
    run<3, false>("read sequential", out);
    return 0;
}
