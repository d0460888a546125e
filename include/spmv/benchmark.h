// spmv/benchmark.h — timing harness around spmv_csr / spmv_ell.
//
// API as in the reference (include/spmv/benchmark.h:14-78).  JSON numbers are
// printed round-trip-safe (%.9g) so benchmark_from_json(benchmark_to_json(r))
// reproduces r exactly.
#ifndef SPMV_BENCHMARK_H
#define SPMV_BENCHMARK_H

#include "spmv.h"
#include "csr_matrix.h"
#include "ell_matrix.h"
#include <string>
#include <vector>

namespace spmv {

struct BenchmarkResult {
    std::string name;
    float execution_time_ms;
    float gflops;
    float bandwidth_gb_s;

    float avg_time_ms;
    float min_time_ms;
    float max_time_ms;
    float stddev_time_ms;   // sample standard deviation

    int num_runs;

    BenchmarkResult() : execution_time_ms(0.0f), gflops(0.0f),
                        bandwidth_gb_s(0.0f), avg_time_ms(0.0f),
                        min_time_ms(0.0f), max_time_ms(0.0f),
                        stddev_time_ms(0.0f), num_runs(0) {}
};

struct BenchmarkConfig {
    int  num_warmup_runs;
    int  num_runs;
    bool compare_cpu;

    BenchmarkConfig() : num_warmup_runs(5), num_runs(20), compare_cpu(true) {}
};

BenchmarkResult benchmark_csr(const CSRMatrix* A, const float* x,
                              const SpMVConfig* config,
                              const BenchmarkConfig* bench_config = nullptr);

BenchmarkResult benchmark_ell(const ELLMatrix* A, const float* x,
                              const BenchmarkConfig* bench_config = nullptr);

struct ComparisonResult {
    BenchmarkResult gpu_result;
    BenchmarkResult cpu_result;
    float speedup;   // cpu avg time / gpu avg time

    ComparisonResult() : speedup(0.0f) {}
};

ComparisonResult compare_gpu_cpu_csr(const CSRMatrix* A, const float* x,
                                     const SpMVConfig* config,
                                     const BenchmarkConfig* bench_config = nullptr);

std::string benchmark_to_json(const BenchmarkResult& result);
std::string comparison_to_json(const ComparisonResult& result);
BenchmarkResult benchmark_from_json(const std::string& json);

} // namespace spmv

#endif // SPMV_BENCHMARK_H
