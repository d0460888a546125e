// ic_bench.hip — does a stream written by one kernel get re-read from the Infinity Cache by the next?
// For each size: kernel W writes `bytes`, kernel R reads them back (both 16 B/lane streams).
// Also a mixed pattern: between W and R, stream-read `other` bytes of a different buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(f32x4* p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void rd(const f32x4* p, long long n, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0.12345f) out[0] = 1;
}
float t(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    const long long maxb = 1LL << 30;
    f32x4 *buf, *other; float* out;
    hipMalloc(&buf, maxb); hipMalloc(&other, maxb); hipMalloc(&out, 4);
    hipMemset(other, 0, maxb);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    printf("%8s %8s | %10s %10s | %10s\n", "MB", "otherMB", "write GB/s", "read GB/s", "read us");
    for (long long mb : {32, 64, 128, 192, 256, 512, 1024}) {
        for (long long omb : {0LL, mb, 2 * mb}) {
            if (omb > 1024) continue;
            const long long n = mb * (1 << 20) / 16, on = omb * (1 << 20) / 16;
            float wbest = 1e9, rbest = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e[0]);
                wr<<<2048, 256>>>(buf, n);
                hipEventRecord(e[1]);
                if (on) rd<<<2048, 256>>>(other, on, out);
                hipEventRecord(e[2]);
                rd<<<2048, 256>>>(buf, n, out);
                hipEventRecord(e[3]);
                hipEventSynchronize(e[3]);
                if (rep) { wbest = fminf(wbest, t(e[0], e[1])); rbest = fminf(rbest, t(e[2], e[3])); }
            }
            printf("%8lld %8lld | %10.0f %10.0f | %10.1f\n", mb, omb, mb * 1.048576 / wbest, mb * 1.048576 / rbest, rbest * 1e3);
        }
    }
    return 0;
}
