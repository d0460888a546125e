// spmv_benchmark.cpp — the benchmark executable of the library (a working counterpart of the
// reference's benchmarks/main.cu, which does not compile as shipped: SURVEY.md §0 D4).
// Same programme: device info, the three CSR kernels + ELL on a 1000 x 1000 matrix of 5 % density,
// GPU vs CPU, PageRank on a 100-node graph — plus, per line, the fraction of the device's peak
// bandwidth (the roofline figure) and an optional larger uniform matrix.
// Plain host C++ against include/spmv/*.h; built by __graft_entry__.build().
//   usage: spmv_benchmark [rows cols nnz_per_row]
#include "spmv/bandwidth.h"
#include "spmv/benchmark.h"
#include "spmv/csr_matrix.h"
#include "spmv/cuda_buffer.h"
#include "spmv/ell_matrix.h"
#include "spmv/pagerank.h"
#include "spmv/spmv.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace spmv;

static void rule(const char* title) { std::printf("\n==== %s ====\n", title); }

static void report(const char* name, const BenchmarkResult& r) {
    const float peak = get_gpu_peak_bandwidth();
    std::printf("%-28s avg %9.4f ms  min %9.4f  stddev %8.4f  %8.2f GFLOPS  %9.2f GB/s  %5.1f %% of peak (%d runs)\n",
                name, r.avg_time_ms, r.min_time_ms, r.stddev_time_ms, r.gflops, r.bandwidth_gb_s,
                peak > 0 ? 100.0f * r.bandwidth_gb_s / peak : 0.0f, r.num_runs);
}

static void device_info() {
    rule("device");
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        std::printf("no HIP device visible\n");
        std::exit(2);
    }
    std::printf("%s (%s), %d CUs, %.1f GiB, wavefront %d, LDS/workgroup %zu KiB, peak bandwidth %.0f GB/s\n",
                prop.name, prop.gcnArchName, prop.multiProcessorCount,
                prop.totalGlobalMem / 1073741824.0, prop.warpSize, prop.sharedMemPerBlock / 1024,
                get_gpu_peak_bandwidth());
}

static void small_dense_case() {
    rule("SpMV, 1000 x 1000, 5 % dense (reference benchmark case)");
    const int rows = 1000, cols = 1000;
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> unit(0.0f, 1.0f);
    std::vector<float> dense(static_cast<size_t>(rows) * cols, 0.0f);
    for (float& v : dense) {
        if (unit(rng) < 0.05f) v = unit(rng) * 10.0f;
    }
    std::vector<float> x(cols, 1.0f);

    CSRMatrix* csr = csr_create(0, 0, 0);
    csr_from_dense(csr, dense.data(), rows, cols);
    csr_to_gpu(csr);
    std::printf("nnz %d, density %.4f\n", csr->nnz, static_cast<float>(csr->nnz) / (rows * cols));

    BenchmarkConfig bc;          // 5 warm-up + 20 timed runs
    const struct { SpMVConfig::KernelType type; const char* name; } kernels[] = {
        {SpMVConfig::SCALAR_CSR, "scalar CSR"}, {SpMVConfig::VECTOR_CSR, "vector CSR"}, {SpMVConfig::MERGE_PATH, "merge path"}};
    for (const auto& k : kernels) {
        SpMVConfig cfg;
        cfg.kernel_type = k.type;
        report(k.name, benchmark_csr(csr, x.data(), &cfg, &bc));
    }
    ELLMatrix* ell = ell_create(0, 0, 0);
    ell_from_csr(ell, csr);
    ell_to_gpu(ell);
    report("ELL", benchmark_ell(ell, x.data(), &bc));

    SpMVConfig chosen = spmv_auto_config(csr);
    std::printf("auto config: kernel %d, block %d, x-on-chip hint %d\n", static_cast<int>(chosen.kernel_type),
                chosen.block_size, static_cast<int>(chosen.use_texture));
    const ComparisonResult cmp = compare_gpu_cpu_csr(csr, x.data(), &chosen, &bc);
    report("GPU (auto config)", cmp.gpu_result);
    std::printf("%-28s avg %9.4f ms   speed-up GPU/CPU %.1fx\n", "CPU spmv_cpu_csr", cmp.cpu_result.avg_time_ms, cmp.speedup);
    std::printf("json: %s\n", benchmark_to_json(cmp.gpu_result).c_str());
    ell_destroy(ell);
    csr_destroy(csr);
}

static void uniform_case(int rows, int cols, int per_row) {
    char title[128];
    std::snprintf(title, sizeof(title), "SpMV, %d x %d, %d per row (uniform random)", rows, cols, per_row);
    rule(title);
    CSRMatrix* csr = csr_create(rows, cols, rows * per_row);
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> val(-1.0f, 1.0f);
    std::vector<int> picks(per_row);
    for (int r = 0; r < rows; ++r) {
        csr->row_ptrs[r] = r * per_row;
        for (int s = 0; s < per_row; ++s) {          // one column per stratum: unique and ascending
            const long long lo = static_cast<long long>(s) * cols / per_row, hi = static_cast<long long>(s + 1) * cols / per_row;
            picks[s] = static_cast<int>(lo + rng() % std::max<long long>(hi - lo, 1));
        }
        for (int s = 0; s < per_row; ++s) {
            csr->col_indices[r * per_row + s] = picks[s];
            csr->values[r * per_row + s] = val(rng);
        }
    }
    csr->row_ptrs[rows] = rows * per_row;
    csr_to_gpu(csr);
    std::vector<float> x(cols);
    for (float& v : x) v = val(rng);
    BenchmarkConfig bc;
    SpMVConfig direct;
    direct.kernel_type = SpMVConfig::VECTOR_CSR;
    report("vector CSR (direct gather)", benchmark_csr(csr, x.data(), &direct, &bc));
    SpMVConfig chosen = spmv_auto_config(csr);
    report("auto config", benchmark_csr(csr, x.data(), &chosen, &bc));
    csr_destroy(csr);
}

static void pagerank_case() {
    rule("PageRank, 100-node random graph");
    const int n = 100;
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> unit(0.0f, 1.0f);
    std::vector<float> adj(static_cast<size_t>(n) * n, 0.0f);
    for (int c = 0; c < n; ++c) {
        int out = 0;
        for (int r = 0; r < n; ++r) {
            if (r != c && unit(rng) < 0.1f) { adj[static_cast<size_t>(r) * n + c] = 1.0f; ++out; }
        }
        for (int r = 0; r < n && out > 0; ++r) adj[static_cast<size_t>(r) * n + c] /= out;   // column-normalised
    }
    CSRMatrix* csr = csr_create(0, 0, 0);
    csr_from_dense(csr, adj.data(), n, n);
    csr_to_gpu(csr);
    PageRankResult r = pagerank(csr);
    std::printf("iterations %d, converged %d, residual %.3g\n", r.iterations, static_cast<int>(r.converged), r.final_residual);
    TopKNode top[5];
    pagerank_top_k(&r, n, 5, top);
    for (const TopKNode& t : top) std::printf("  node %3d  rank %.6f\n", t.node_id, t.rank);
    pagerank_free(&r);
    csr_destroy(csr);
}

int main(int argc, char** argv) {
    device_info();
    small_dense_case();
    if (argc == 4) uniform_case(std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]));
    pagerank_case();
    return 0;
}
