// mall_bench.hip — can the product round trip of the two-phase SpMV live in the Infinity Cache?
//
// One kernel carries BOTH phases' traffic at fine grain (same instruction stream in every variant):
//   "phase 1" part: stream value f32 + local column u16 (6 B/entry, HBM, read once),
//                   write a product f32 into a RING (4 B/entry);
//   "phase 2" part: read the product written one lap of the ring ago (4 B/entry) + local row u16
//                   (2 B/entry, HBM, read once).
// Only the ring's size changes between variants: a ring far larger than the 256 MiB Infinity Cache
// makes every product travel to HBM and back (what csrc/tiled.hip does today); a ring of ~100-200 MB
// keeps the round trip on the die.  Every workgroup owns a private segment of the ring (reads what
// it wrote itself one lap earlier, so no cross-workgroup visibility question arises).
// Also: pure read / write / copy rates with the same launch shape, for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// entries handled per workgroup iteration: BLOCK threads x 4 entries
template <int BLOCK, int UNROLL, bool P1, bool P2, bool NT, bool CROSS = false>
__global__ __launch_bounds__(BLOCK) void mix(const float* __restrict__ vals, const unsigned short* __restrict__ lcol,
                                             const unsigned short* __restrict__ lrow, float* __restrict__ ring,
                                             long long per_wg, long long seg_entries, float* __restrict__ sink) {
    const long long base = (long long)blockIdx.x * per_wg;
    float* seg = ring + (long long)blockIdx.x * seg_entries;
    // CROSS: read the segment a workgroup on ANOTHER XCD writes (blocks b and b + 1 sit on different XCDs), so that no L2 can serve it
    const float* rseg = CROSS ? ring + (long long)((blockIdx.x + 1) % gridDim.x) * seg_entries : seg;
    float acc = 0.f;
    constexpr int STEP = BLOCK * 4;
    long long pos = 0;       // position inside the segment (multiple of STEP; seg_entries is a multiple of STEP * UNROLL)
    for (long long i = 0; i < per_wg; i += STEP * UNROLL) {
        f32x4 v[UNROLL]; u16x4 c[UNROLL]; f32x4 p[UNROLL]; u16x4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long e = base + i + u * STEP + threadIdx.x * 4;
            if (P1) {
                if (NT) { v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vals + e));
                          c[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(lcol + e)); }
                else    { v[u] = *reinterpret_cast<const f32x4*>(vals + e); c[u] = *reinterpret_cast<const u16x4*>(lcol + e); }
            }
            if (P2) { p[u] = *reinterpret_cast<const f32x4*>(rseg + pos + u * STEP + threadIdx.x * 4);
                      if (NT) r[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x4*>(lrow + e));
                      else    r[u] = *reinterpret_cast<const u16x4*>(lrow + e); }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (P2) acc += p[u][0] * r[u][0] + p[u][1] * r[u][1] + p[u][2] * r[u][2] + p[u][3] * r[u][3];
            if (P1) {
                f32x4 q;
                q[0] = v[u][0] * c[u][0]; q[1] = v[u][1] * c[u][1]; q[2] = v[u][2] * c[u][2]; q[3] = v[u][3] * c[u][3];
                *reinterpret_cast<f32x4*>(seg + pos + u * STEP + threadIdx.x * 4) = q;
            }
        }
        pos += STEP * UNROLL;
        if (pos >= seg_entries) pos = 0;
    }
    if (acc == 0.123456f) sink[0] = acc;
}

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void rd(const f32x4* __restrict__ p, long long n, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += t[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0.12345f) sink[0] = 1;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void wr(f32x4* __restrict__ p, long long n) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void cp(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long n) {
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 t[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) t[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) d[i + u * stride] = t[u];
    }
}
// read a window of `win` float4 repeatedly (MALL-resident when small) while also streaming `n` float4 of HBM:
// every iteration takes H loads from the stream and M loads from the window
template <int BLOCK, int H, int M>
__global__ __launch_bounds__(BLOCK) void rd_mix(const f32x4* __restrict__ stream, long long n,
                                                const f32x4* __restrict__ window, long long win, long long iters, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * BLOCK;
    const long long t0 = (long long)blockIdx.x * BLOCK + threadIdx.x;
    long long hs = t0, ws = t0 % win;
    for (long long it = 0; it < iters; ++it) {
        f32x4 a[H > 0 ? H : 1], b[M > 0 ? M : 1];
#pragma unroll
        for (int u = 0; u < H; ++u) { a[u] = stream[hs]; hs += stride; if (hs >= n) hs -= n; }
#pragma unroll
        for (int u = 0; u < M; ++u) { b[u] = window[ws]; ws += stride; if (ws >= win) ws -= win; }
#pragma unroll
        for (int u = 0; u < H; ++u) acc += a[u];
#pragma unroll
        for (int u = 0; u < M; ++u) acc += b[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0.12345f) sink[0] = 1;
}

__global__ void fill(unsigned short* a, unsigned short* b, float* v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        a[i] = (unsigned short)z; b[i] = (unsigned short)(z >> 16); v[i] = 1.0f;
    }
}

template <class F> float timeit(F f, int reps = 6) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

template <int BLOCK, int UNROLL>
void run_mix(int wgs, long long N, long long ring_mb, const float* vals, const unsigned short* lcol,
             const unsigned short* lrow, float* ring, float* sink) {
    constexpr long long quantum = (long long)BLOCK * 4 * UNROLL;
    const long long per = N / wgs / quantum * quantum;
    long long seg = ring_mb * (1LL << 20) / 4 / wgs / quantum * quantum;
    if (seg < quantum) seg = quantum;
    if (seg > per) seg = per;
    const double total = (double)per * wgs;
    float t12 = timeit([&] { mix<BLOCK, UNROLL, true, true, false><<<wgs, BLOCK>>>(vals, lcol, lrow, ring, per, seg, sink); });
    float tnt = timeit([&] { mix<BLOCK, UNROLL, true, true, true><<<wgs, BLOCK>>>(vals, lcol, lrow, ring, per, seg, sink); });
    float tx = timeit([&] { mix<BLOCK, UNROLL, true, true, true, true><<<wgs, BLOCK>>>(vals, lcol, lrow, ring, per, seg, sink); });
    printf("mix  block %4d unroll %d wgs %5d ring %5lld MB (seg %7.1f KB): %7.1f us  %5.2f TB/s | nt streams %7.1f us %5.2f TB/s | nt + cross-XCD read %7.1f us %5.2f TB/s\n",
           BLOCK, UNROLL, wgs, ring_mb, seg * 4 / 1024.0, t12 * 1e3, total * 16 / t12 / 1e9, tnt * 1e3, total * 16 / tnt / 1e9, tx * 1e3, total * 16 / tx / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long N = 160LL * 1000 * 1000;
    const long long ring_max = 1400LL << 20;    // bytes
    float *vals, *ring, *sink; unsigned short *lcol, *lrow;
    CHECK(hipMalloc(&vals, N * 4)); CHECK(hipMalloc(&lcol, N * 2)); CHECK(hipMalloc(&lrow, N * 2));
    CHECK(hipMalloc(&ring, ring_max)); CHECK(hipMalloc(&sink, 4));
    fill<<<4096, 256>>>(lcol, lrow, vals, N);
    CHECK(hipMemset(ring, 0, ring_max));
    CHECK(hipDeviceSynchronize());

    // ---- reference rates: pure read / write / copy over buffers of different size (back-to-back launches) ----
    const f32x4* rbuf = reinterpret_cast<const f32x4*>(ring);
    for (long long mb : {64, 128, 192, 1024}) {
        const long long n = mb * (1LL << 20) / 16;
        const int reps = mb <= 192 ? 40 : 6;
        float t1 = timeit([&] { rd<256, 4><<<2048, 256>>>(rbuf, n, sink); }, reps);
        float t2 = timeit([&] { rd<256, 8><<<2048, 256>>>(rbuf, n, sink); }, reps);
        float t3 = timeit([&] { rd<512, 4><<<1024, 512>>>(rbuf, n, sink); }, reps);
        float t4 = timeit([&] { wr<256><<<2048, 256>>>(reinterpret_cast<f32x4*>(ring), n); }, reps);
        printf("pure  %5lld MB: read u4 %6.2f  u8 %6.2f  b512 %6.2f TB/s | write %6.2f TB/s  (read %.1f us)\n", mb,
               mb * 1.048576e-3 / t1, mb * 1.048576e-3 / t2, mb * 1.048576e-3 / t3, mb * 1.048576e-3 / t4, t1 * 1e3);
        fflush(stdout);
    }
    {
        const long long n = 640LL * (1 << 20) / 16;
        f32x4* dst = reinterpret_cast<f32x4*>(ring) + n;
        float t = timeit([&] { cp<256, 4><<<2048, 256>>>(rbuf, dst, n); });
        printf("copy 640 MB -> 640 MB: %6.2f TB/s (read + written)\n", 2 * 640 * 1.048576e-3 / t);
    }
    // ---- HBM stream + MALL window read mixes (iters chosen for ~1.5 GB total) ----
    {
        const long long n = (1024LL << 20) / 16;                 // 1 GB stream (beyond the cache)
        const f32x4* window = rbuf + n;                          // window after the stream
        for (long long wmb : {48, 96, 320}) {
            const long long win = wmb * (1LL << 20) / 16;
            const long long threads = 2048LL * 256;
            auto report = [&](const char* name, int h, int m, float t, long long iters) {
                const double bytes = (double)iters * threads * 16 * (h + m);
                printf("rdmix window %4lld MB  %s  %6.2f TB/s  (%.0f us)\n", wmb, name, bytes / t / 1e9, t * 1e3);
                fflush(stdout);
            };
            { const long long it = 48;  float t = timeit([&] { rd_mix<256, 4, 0><<<2048, 256>>>(rbuf, n, window, win, it, sink); }); report("H4 M0", 4, 0, t, it); }
            { const long long it = 48;  float t = timeit([&] { rd_mix<256, 0, 4><<<2048, 256>>>(rbuf, n, window, win, it, sink); }); report("H0 M4", 0, 4, t, it); }
            { const long long it = 24;  float t = timeit([&] { rd_mix<256, 4, 4><<<2048, 256>>>(rbuf, n, window, win, it, sink); }); report("H4 M4", 4, 4, t, it); }
            { const long long it = 32;  float t = timeit([&] { rd_mix<256, 4, 2><<<2048, 256>>>(rbuf, n, window, win, it, sink); }); report("H4 M2", 4, 2, t, it); }
            { const long long it = 32;  float t = timeit([&] { rd_mix<256, 2, 4><<<2048, 256>>>(rbuf, n, window, win, it, sink); }); report("H2 M4", 2, 4, t, it); }
        }
    }
    // ---- the two phases' traffic in one kernel; only the ring size differs ----
    for (long long ring_mb : {1280, 320, 160, 96, 64, 32, 16}) {
        run_mix<256, 2>(2048, N, ring_mb, vals, lcol, lrow, ring, sink);
        run_mix<256, 4>(2048, N, ring_mb, vals, lcol, lrow, ring, sink);
        run_mix<512, 2>(1024, N, ring_mb, vals, lcol, lrow, ring, sink);
    }
    // phase-1-only and phase-2-only traffic with the same shape (ring 1280 MB = no reuse)
    {
        constexpr long long q = 256 * 4 * 4;
        const int wgs = 2048;
        const long long per = N / wgs / q * q, seg = per;
        float t1 = timeit([&] { mix<256, 4, true, false, false><<<wgs, 256>>>(vals, lcol, lrow, ring, per, seg, sink); });
        float t2 = timeit([&] { mix<256, 4, false, true, false><<<wgs, 256>>>(vals, lcol, lrow, ring, per, seg, sink); });
        printf("phase-1 traffic alone %7.1f us %5.2f TB/s (10 B/entry) | phase-2 traffic alone %7.1f us %5.2f TB/s (6 B/entry)\n",
               t1 * 1e3, (double)per * wgs * 10 / t1 / 1e9, t2 * 1e3, (double)per * wgs * 6 / t2 / 1e9);
    }
    return 0;
}
