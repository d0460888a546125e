// benchmark.cpp — warm-up + repeated-run timing harness and its JSON form.
//
// Protocol as the reference (src/benchmark.cu:21-185): num_warmup_runs untimed
// calls, num_runs timed calls, per-call time = the kernel-only event time that
// spmv_csr / spmv_ell report, statistics = min / max / mean / sample stddev.
// Differences: the CPU leg of compare_gpu_cpu_csr is timed with a host clock
// (the reference brackets a host function with device events)
// and JSON numbers are printed with 9 significant digits so they round-trip.
#include "internal.h"
#include "spmv/benchmark.h"
#include "spmv/cuda_buffer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

namespace spmv {

namespace {

void summarise(const std::vector<float>& times, BenchmarkResult* out) {
    if (times.empty()) return;
    out->num_runs = static_cast<int>(times.size());
    out->min_time_ms = *std::min_element(times.begin(), times.end());
    out->max_time_ms = *std::max_element(times.begin(), times.end());
    float total = 0.0f;
    for (float t : times) total += t;
    out->avg_time_ms = total / times.size();
    out->execution_time_ms = out->avg_time_ms;
    float spread = 0.0f;
    if (times.size() > 1) {
        for (float t : times) spread += (t - out->avg_time_ms) * (t - out->avg_time_ms);
        spread = std::sqrt(spread / (times.size() - 1));
    }
    out->stddev_time_ms = spread;
}

BenchmarkResult run_device(const char* name, const BenchmarkConfig* cfg,
                           const std::function<SpMVResult()>& call) {
    BenchmarkResult result;
    result.name = name;
    const BenchmarkConfig fallback;
    if (!cfg) cfg = &fallback;

    for (int i = 0; i < cfg->num_warmup_runs; ++i) call();

    std::vector<float> times;
    times.reserve(std::max(cfg->num_runs, 0));
    for (int i = 0; i < cfg->num_runs; ++i) {
        const SpMVResult r = call();
        if (r.error_code != 0) continue;
        times.push_back(r.elapsed_ms);
        result.gflops = r.gflops;
        result.bandwidth_gb_s = r.bandwidth_gb_s;
    }
    summarise(times, &result);
    return result;
}

void append_number(std::string* out, const char* key, double value, bool last = false) {
    char buf[96];
    snprintf(buf, sizeof(buf), "  \"%s\": %.9g%s\n", key, value, last ? "" : ",");
    *out += buf;
}

} // namespace

BenchmarkResult benchmark_csr(const CSRMatrix* A, const float* x, const SpMVConfig* config,
                              const BenchmarkConfig* bench_config) {
    if (!A || !x) {
        BenchmarkResult r;
        r.name = "CSR SpMV";
        return r;
    }
    CudaBuffer<float> d_x(A->num_cols);
    CudaBuffer<float> d_y(A->num_rows);
    d_x.copyFromHost(x, A->num_cols);
    return run_device("CSR SpMV", bench_config, [&] {
        return spmv_csr(A, d_x.get(), d_y.get(), config, A->num_cols);
    });
}

BenchmarkResult benchmark_ell(const ELLMatrix* A, const float* x,
                              const BenchmarkConfig* bench_config) {
    if (!A || !x) {
        BenchmarkResult r;
        r.name = "ELL SpMV";
        return r;
    }
    CudaBuffer<float> d_x(A->num_cols);
    CudaBuffer<float> d_y(A->num_rows);
    d_x.copyFromHost(x, A->num_cols);
    return run_device("ELL SpMV", bench_config, [&] {
        return spmv_ell(A, d_x.get(), d_y.get(), nullptr, A->num_cols);
    });
}

ComparisonResult compare_gpu_cpu_csr(const CSRMatrix* A, const float* x, const SpMVConfig* config,
                                     const BenchmarkConfig* bench_config) {
    ComparisonResult comp;
    comp.gpu_result = benchmark_csr(A, x, config, bench_config);
    comp.cpu_result.name = "CPU CSR SpMV";
    if (!A || !x) return comp;

    const BenchmarkConfig fallback;
    if (!bench_config) bench_config = &fallback;

    std::vector<float> y(std::max(A->num_rows, 0));
    std::vector<float> times;
    for (int i = 0; i < bench_config->num_runs; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        spmv_cpu_csr(A, x, y.data());
        const auto t1 = std::chrono::steady_clock::now();
        times.push_back(std::chrono::duration<float, std::milli>(t1 - t0).count());
    }
    summarise(times, &comp.cpu_result);
    if (comp.cpu_result.avg_time_ms > 0.0f) {
        comp.cpu_result.gflops = (2.0f * A->nnz) / (comp.cpu_result.avg_time_ms * 1e6f);
    }
    if (comp.gpu_result.avg_time_ms > 0.0f) {
        comp.speedup = comp.cpu_result.avg_time_ms / comp.gpu_result.avg_time_ms;
    }
    return comp;
}

std::string benchmark_to_json(const BenchmarkResult& r) {
    std::string out = "{\n";
    out += "  \"name\": \"" + r.name + "\",\n";
    append_number(&out, "execution_time_ms", r.execution_time_ms);
    append_number(&out, "gflops", r.gflops);
    append_number(&out, "bandwidth_gb_s", r.bandwidth_gb_s);
    append_number(&out, "avg_time_ms", r.avg_time_ms);
    append_number(&out, "min_time_ms", r.min_time_ms);
    append_number(&out, "max_time_ms", r.max_time_ms);
    append_number(&out, "stddev_time_ms", r.stddev_time_ms);
    append_number(&out, "num_runs", r.num_runs, true);
    out += "}";
    return out;
}

std::string comparison_to_json(const ComparisonResult& c) {
    std::string out = "{\n";
    out += "  \"gpu\": " + benchmark_to_json(c.gpu_result) + ",\n";
    out += "  \"cpu\": " + benchmark_to_json(c.cpu_result) + ",\n";
    char buf[64];
    snprintf(buf, sizeof(buf), "  \"speedup\": %.9g\n", static_cast<double>(c.speedup));
    out += buf;
    out += "}";
    return out;
}

BenchmarkResult benchmark_from_json(const std::string& json) {
    BenchmarkResult r;
    auto number = [&json](const char* key) -> float {
        const std::string needle = std::string("\"") + key + "\":";
        const size_t at = json.find(needle);
        if (at == std::string::npos) return 0.0f;
        return std::strtof(json.c_str() + at + needle.size(), nullptr);
    };
    const std::string name_key = "\"name\": \"";
    const size_t at = json.find(name_key);
    if (at != std::string::npos) {
        const size_t from = at + name_key.size();
        const size_t to = json.find('"', from);
        if (to != std::string::npos) r.name = json.substr(from, to - from);
    }
    r.execution_time_ms = number("execution_time_ms");
    r.gflops = number("gflops");
    r.bandwidth_gb_s = number("bandwidth_gb_s");
    r.avg_time_ms = number("avg_time_ms");
    r.min_time_ms = number("min_time_ms");
    r.max_time_ms = number("max_time_ms");
    r.stddev_time_ms = number("stddev_time_ms");
    r.num_runs = static_cast<int>(number("num_runs"));
    return r;
}

} // namespace spmv
