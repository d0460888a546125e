// dropin_smoke.cpp — a C++ caller written the way the reference's own tests are
// (tests/test_spmv.cu, test_common.cpp, test_pagerank.cu, test_benchmark.cu,
// test_bandwidth.cu): `#include "spmv/*.h"`, namespace spmv, CudaBuffer, direct
// struct-field access.  Built with plain g++ against include/ and libspmv_amd.so
// to show the C++ boundary is a source-level drop-in.  Needs a GPU to run.
#include "spmv/spmv.h"
#include "spmv/bandwidth.h"
#include "spmv/benchmark.h"
#include "spmv/cuda_buffer.h"
#include "spmv/pagerank.h"
#include "spmv/test_utils.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace spmv;
using namespace spmv::test;

static int g_failures = 0;
#define CHECK(cond) do { if (!(cond)) { ++g_failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)

static bool close_enough(const float* a, const float* b, int n, float rel) {
    for (int i = 0; i < n; ++i) {
        const float diff = std::fabs(a[i] - b[i]);
        const float scale = std::fmax(std::fabs(a[i]), std::fabs(b[i]));
        if (scale < 1e-10f ? diff > 1e-6f : diff / scale > rel) return false;
    }
    return true;
}

// reordering kernels: |got - want| <= 1e-5 * max(|want|, sum_j |a_ij x_j|)  
static bool reordered_ok(const CSRMatrix* A, const float* x, const float* want, const float* got) {
    for (int i = 0; i < A->num_rows; ++i) {
        double scale = std::fabs(want[i]);
        double abs_sum = 0.0;
        for (int j = A->row_ptrs[i]; j < A->row_ptrs[i + 1]; ++j) {
            abs_sum += std::fabs(static_cast<double>(A->values[j]) * x[A->col_indices[j]]);
        }
        scale = std::fmax(scale, abs_sum);
        if (std::fabs(static_cast<double>(want[i]) - got[i]) > 1e-5 * std::fmax(scale, 1e-30)) return false;
    }
    return true;
}

static void buffers() {   // reference tests/test_common.cpp:21-98
    CudaBuffer<float> empty;
    CHECK(empty.get() == nullptr && empty.size() == 0 && empty.empty());
    CudaBuffer<float> buf(100);
    CHECK(buf.get() != nullptr && buf.size() == 100);
    std::vector<float> host(100), back(100);
    for (int i = 0; i < 100; ++i) host[i] = static_cast<float>(i);
    buf.copyFromHost(host.data(), 100);
    buf.copyToHost(back.data(), 100);
    CHECK(host == back);
    CudaBuffer<float> moved(std::move(buf));
    CHECK(buf.get() == nullptr && moved.size() == 100);
    moved.resize(200);
    CHECK(moved.size() == 200);
    moved.release();
    CHECK(moved.get() == nullptr && moved.size() == 0);
    bool threw = false;
    try { CudaBuffer<float> small(4); small.copyFromHost(host.data(), 5); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    CHECK(std::strcmp(spmv_error_string(SpMVError::INVALID_FORMAT), "Invalid sparse matrix format") == 0);
}

static void csr_kernels_vs_cpu() {   // reference tests/test_spmv.cu:40-78, all three kernels
    RandomGenerator rng(42);
    for (int iter = 0; iter < 30; ++iter) {
        const int rows = rng.randInt(1, 200), cols = rng.randInt(1, 200);
        auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.01f, 0.3f), rng);
        auto x = generateRandomVector(cols, rng);
        CSRMatrix* csr = csr_create(0, 0, 0);
        csr_from_dense(csr, dense.data(), rows, cols);
        csr_to_gpu(csr);
        std::vector<float> y_cpu(rows), y_gpu(rows);
        spmv_cpu_csr(csr, x.data(), y_cpu.data());
        CudaBuffer<float> d_x(cols), d_y(rows);
        d_x.copyFromHost(x.data(), cols);
        for (auto kt : {SpMVConfig::SCALAR_CSR, SpMVConfig::VECTOR_CSR, SpMVConfig::MERGE_PATH}) {
            SpMVConfig config;
            config.kernel_type = kt;
            SpMVResult r = spmv_csr(csr, d_x.get(), d_y.get(), &config, cols);
            if (csr->nnz == 0) { CHECK(r.error_code == 0); continue; }
            CHECK(r.error_code == static_cast<int>(SpMVError::SUCCESS));
            d_y.copyToHost(y_gpu.data(), rows);
            if (kt == SpMVConfig::SCALAR_CSR) CHECK(std::memcmp(y_cpu.data(), y_gpu.data(), rows * sizeof(float)) == 0);
            else CHECK(reordered_ok(csr, x.data(), y_cpu.data(), y_gpu.data()));
        }
        // ELL path, reference tests/test_spmv.cu:82-118
        ELLMatrix* ell = ell_create(0, 0, 0);
        ell_from_dense(ell, dense.data(), rows, cols);
        ell_to_gpu(ell);
        std::vector<float> y_ell(rows);
        spmv_cpu_ell(ell, x.data(), y_cpu.data());
        SpMVResult r = spmv_ell(ell, d_x.get(), d_y.get(), nullptr, cols);
        CHECK(r.error_code == 0);
        if (ell->max_nnz_per_row > 0) {
            d_y.copyToHost(y_ell.data(), rows);
            CHECK(std::memcmp(y_cpu.data(), y_ell.data(), rows * sizeof(float)) == 0);
        }
        ell_destroy(ell);
        csr_destroy(csr);
    }
}

static void unit_cases() {   // reference tests/test_spmv.cu:148-237
    {
        CSRMatrix* csr = csr_create(0, 0, 0);
        csr_to_gpu(csr);
        CudaBuffer<float> d_x(1), d_y(1);
        CHECK(spmv_csr(csr, d_x.get(), d_y.get(), nullptr, 1).error_code == 0);   // EmptyMatrix
        csr_destroy(csr);
    }
    {
        std::vector<float> dense = {1, 2, 0, 0, 0, 0, 3, 0, 4}, x = {1, 1, 1}, y(3);
        CSRMatrix* csr = csr_create(0, 0, 0);
        csr_from_dense(csr, dense.data(), 3, 3);
        csr_to_gpu(csr);
        CHECK(csr->d_values != nullptr && csr->owns_device_memory);
        CudaBuffer<float> d_x(3), d_y(3);
        d_x.copyFromHost(x.data(), 3);
        spmv_csr(csr, d_x.get(), d_y.get(), nullptr, 3);
        d_y.copyToHost(y.data(), 3);
        CHECK(y[0] == 3.0f && y[1] == 0.0f && y[2] == 7.0f);                        // ZeroRows
        SpMVConfig cfg = spmv_auto_config(csr);
        CHECK(cfg.block_size >= 32 && cfg.block_size <= 1024 && cfg.block_size % 32 == 0);
        BandwidthMetrics m = compute_bandwidth_csr(csr, 1.0f);
        CHECK(m.achieved_bandwidth_gb_s >= 0 && m.theoretical_bandwidth_gb_s > 0 && m.efficiency <= 1.0f);
        CHECK(get_gpu_peak_bandwidth() > 0 && get_gpu_peak_bandwidth() < 10000);
        csr_destroy(csr);
    }
}

static void pagerank_cases() {   // reference tests/test_pagerank.cu:140-189
    std::vector<float> adj = {0, 0, 1, 1, 0, 0, 0, 1, 0};
    CSRMatrix* csr = csr_create(0, 0, 0);
    csr_from_dense(csr, adj.data(), 3, 3);
    csr_to_gpu(csr);
    PageRankResult r = pagerank(csr);
    CHECK(r.ranks != nullptr && r.converged);
    for (int i = 0; i < 3; ++i) CHECK(std::fabs(r.ranks[i] - 1.0f / 3.0f) < 1e-4f);
    TopKNode top[2];
    pagerank_top_k(&r, 3, 2, top);
    CHECK(top[0].rank >= top[1].rank);
    pagerank_free(&r);
    CHECK(r.ranks == nullptr);
    csr_destroy(csr);
}

static void benchmark_cases() {   // reference tests/test_benchmark.cu:17-103
    RandomGenerator rng(42);
    auto dense = generateRandomDenseMatrix(300, 300, 0.05f, rng);
    auto x = generateRandomVector(300, rng);
    CSRMatrix* csr = csr_create(0, 0, 0);
    csr_from_dense(csr, dense.data(), 300, 300);
    csr_to_gpu(csr);
    BenchmarkConfig bc;
    bc.num_warmup_runs = 2;
    bc.num_runs = 5;
    SpMVConfig cfg;
    cfg.kernel_type = SpMVConfig::VECTOR_CSR;
    BenchmarkResult b = benchmark_csr(csr, x.data(), &cfg, &bc);
    CHECK(b.num_runs == 5 && b.min_time_ms <= b.avg_time_ms && b.avg_time_ms <= b.max_time_ms && b.gflops > 0);
    BenchmarkResult back = benchmark_from_json(benchmark_to_json(b));
    CHECK(back.avg_time_ms == b.avg_time_ms && back.gflops == b.gflops && back.num_runs == b.num_runs);
    ComparisonResult cmp = compare_gpu_cpu_csr(csr, x.data(), &cfg, &bc);
    CHECK(cmp.speedup > 0 && comparison_to_json(cmp).find("\"speedup\"") != std::string::npos);
    csr_destroy(csr);
}

int main() {
    buffers();
    csr_kernels_vs_cpu();
    unit_cases();
    pagerank_cases();
    benchmark_cases();
    std::printf(g_failures ? "dropin_smoke: %d failure(s)\n" : "dropin_smoke: all checks passed\n", g_failures);
    return g_failures ? 1 : 0;
}
