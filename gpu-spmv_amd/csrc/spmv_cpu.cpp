// spmv_cpu.cpp — host SpMV (the reference's CPU path, part of its public API)
// and the kernel selector.
//
// spmv_cpu_csr / spmv_cpu_ell: sequential fp32 row sums, as reference
// src/spmv_cpu.cpp:6-32.  spmv_auto_config: reference src/spmv_cpu.cpp:34-50
// with the skew threshold re-measured for 64-wide wavefronts.
// None of the device entry points ever routes through these functions.
#include "internal.h"

#include <cstdlib>

namespace spmv {

void spmv_cpu_csr(const CSRMatrix* A, const float* x, float* y) {
    if (!A || !x || !y) return;
    const int* ptr = A->row_ptrs;
    for (int r = 0; r < A->num_rows; ++r) {
        float acc = 0.0f;
        for (int j = ptr[r], end = ptr[r + 1]; j < end; ++j) {
            acc += A->values[j] * x[A->col_indices[j]];
        }
        y[r] = acc;
    }
}

void spmv_cpu_ell(const ELLMatrix* A, const float* x, float* y) {
    if (!A || !x || !y) return;
    const size_t rows = A->num_rows;
    for (size_t r = 0; r < rows; ++r) {
        float acc = 0.0f;
        size_t slot = r;
        for (int k = 0; k < A->max_nnz_per_row; ++k, slot += rows) {
            const int c = A->col_indices[slot];
            if (c >= 0) acc += A->values[slot] * x[c];
        }
        y[r] = acc;
    }
}

namespace detail {

// Row-length skew (max / (min + 1)) from which the merge-path kernel is chosen.
// The reference uses 10 for 32-lane warps; see DESIGN.md §6 for the wave64 sweep.
float skew_threshold() {
    static const float value = [] {
        if (const char* env = std::getenv("SPMV_SKEW_THRESHOLD")) {
            const float v = static_cast<float>(std::atof(env));
            if (v > 0.0f) return v;
        }
        return 10.0f;
    }();
    return value;
}

} // namespace detail

SpMVConfig spmv_auto_config(const CSRMatrix* A) {
    SpMVConfig config;
    config.block_size = 256;
    if (!A) return config;

    config.use_texture = A->num_cols > 10000;

    // Host arrays present: scan them on every call, as the reference does (they
    // may have been refilled).  Device-only matrices: reduce on the device once
    // and remember the answer in the side table.
    CSRStats stats;
    detail::CsrAux* aux = (!A->row_ptrs && A->d_row_ptrs)
                        ? detail::aux_lookup(A->d_row_ptrs, true) : nullptr;
    if (aux && aux->have_stats) {
        stats = aux->stats;
    } else {
        stats = csr_compute_stats(A);
        if (aux) {
            aux->stats = stats;
            aux->have_stats = true;
        }
    }

    if (stats.avg_nnz_per_row < 4.0f) {
        config.kernel_type = SpMVConfig::SCALAR_CSR;
    } else if (stats.skewness < detail::skew_threshold()) {
        config.kernel_type = SpMVConfig::VECTOR_CSR;
    } else {
        config.kernel_type = SpMVConfig::MERGE_PATH;
    }
    return config;
}

} // namespace spmv
