/* spmv_oracle.c — CPU restatement of the reference's SpMV / PageRank path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpu-spmv_amd/ links, imports or
 * calls this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker (never as the thing shipped).
 *
 * Each function restates one reference routine in plain C and cites it
 * (paths relative to the reference repository LessUp/gpu-spmv).  Build with
 * `-ffp-contract=off` on x86-64 so that `sum += v * x` is a rounded multiply
 * followed by a rounded add, exactly as the reference's g++ build computes it.
 *
 * Pinning: tests/test_oracle_golden.py checks these functions against
 *   (1) tests/golden/ref_*.npz — outputs of the reference's own
 *       spmv_cpu.cpp / csr_matrix.cpp / ell_matrix.cpp compiled from
 *       /root/reference by oracle/Makefile (oracle/_ref/ref_cpu), and
 *   (2) the known-answer vectors in the reference's README.md:75-99,
 *       .kiro/specs/spmv-gpu/design.md:372-385 and tests/test_spmv.cu:161-218.
 * PageRank (pagerank.cu) cannot be compiled here (it needs the CUDA runtime),
 * so oracle_pagerank is pinned only by the reference tests' known answers
 * (tests/test_pagerank.cu:140-164 three-cycle; :18-77 invariants).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y = A x, CSR, sequential fp32 per row — src/spmv_cpu.cpp:6-16 */
void oracle_spmv_csr(int num_rows, const int* row_ptrs, const int* col_indices,
                     const float* values, const float* x, float* y) {
    for (int i = 0; i < num_rows; i++) {
        float sum = 0.0f;
        for (int j = row_ptrs[i]; j < row_ptrs[i + 1]; j++) {
            sum += values[j] * x[col_indices[j]];
        }
        y[i] = sum;
    }
}

/* The same row loop with OpenMP static row blocks — the "all host cores" CPU baseline that
 * BASELINE.md §4 asks for next to the single-thread one (the reference itself has no OpenMP).
 * Per-row arithmetic is unchanged, so y is bit-identical to oracle_spmv_csr. */
void oracle_spmv_csr_parallel(int num_rows, const int* row_ptrs, const int* col_indices,
                              const float* values, const float* x, float* y, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int i = 0; i < num_rows; i++) {
        float sum = 0.0f;
        for (int j = row_ptrs[i]; j < row_ptrs[i + 1]; j++) {
            sum += values[j] * x[col_indices[j]];
        }
        y[i] = sum;
    }
}

/* y = A x, column-major ELL, padding (col < 0) skipped — src/spmv_cpu.cpp:18-32 */
void oracle_spmv_ell(int num_rows, int max_nnz_per_row, const int* col_indices,
                     const float* values, const float* x, float* y) {
    for (int i = 0; i < num_rows; i++) {
        float sum = 0.0f;
        for (int k = 0; k < max_nnz_per_row; k++) {
            size_t idx = (size_t)k * (size_t)num_rows + (size_t)i;   /* ell_index, ell_matrix.h:64-66 */
            int col = col_indices[idx];
            if (col >= 0) {
                sum += values[idx] * x[col];
            }
        }
        y[i] = sum;
    }
}

/* number of entries != 0.0f — first pass of csr_from_dense, src/csr_matrix.cpp:55-61 */
int oracle_count_nonzeros(const float* dense, int rows, int cols) {
    int nnz = 0;
    size_t total = (size_t)rows * (size_t)cols;
    for (size_t i = 0; i < total; i++) {
        if (dense[i] != 0.0f) nnz++;
    }
    return nnz;
}

/* dense (row-major) -> CSR, ascending columns — src/csr_matrix.cpp:80-93.
 * Arrays sized by oracle_count_nonzeros; returns nnz. */
int oracle_csr_from_dense(const float* dense, int rows, int cols,
                          int* row_ptrs, int* col_indices, float* values) {
    int idx = 0;
    for (int i = 0; i < rows; i++) {
        row_ptrs[i] = idx;
        for (int j = 0; j < cols; j++) {
            float v = dense[(size_t)i * cols + j];
            if (v != 0.0f) {
                values[idx] = v;
                col_indices[idx] = j;
                idx++;
            }
        }
    }
    row_ptrs[rows] = idx;
    return idx;
}

/* widest row — src/ell_matrix.cpp:116-121 */
int oracle_max_row_nnz(int num_rows, const int* row_ptrs) {
    int widest = 0;
    for (int i = 0; i < num_rows; i++) {
        int len = row_ptrs[i + 1] - row_ptrs[i];
        if (len > widest) widest = len;
    }
    return widest;
}

/* CSR -> column-major ELL with (-1, 0.0f) padding — src/ell_matrix.cpp:139-156.
 * ell arrays hold num_rows * K slots, K = oracle_max_row_nnz. */
void oracle_ell_from_csr(int num_rows, int K, const int* row_ptrs, const int* col_indices,
                         const float* values, int* ell_cols, float* ell_vals) {
    size_t slots = (size_t)num_rows * (size_t)K;
    for (size_t s = 0; s < slots; s++) {
        ell_cols[s] = -1;
        ell_vals[s] = 0.0f;
    }
    for (int i = 0; i < num_rows; i++) {
        int k = 0;
        for (int j = row_ptrs[i]; j < row_ptrs[i + 1]; j++, k++) {
            size_t idx = (size_t)k * (size_t)num_rows + (size_t)i;
            ell_vals[idx] = values[j];
            ell_cols[idx] = col_indices[j];
        }
    }
}

/* row-length statistics — src/csr_matrix.cpp:281-300.
 * out = {avg, max, min, skewness = max / (min + 1)} with max/min stored as floats. */
void oracle_csr_stats(int num_rows, int nnz, const int* row_ptrs, float* out4) {
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0f;
    if (num_rows == 0) return;
    int longest = 0, shortest = INT_MAX;
    for (int i = 0; i < num_rows; i++) {
        int len = row_ptrs[i + 1] - row_ptrs[i];
        if (len > longest) longest = len;
        if (len < shortest) shortest = len;
    }
    out4[0] = (float)nnz / (float)num_rows;
    out4[1] = (float)longest;
    out4[2] = (float)shortest;
    out4[3] = (float)longest / (float)(shortest + 1);
}

/* kernel selector with the reference's thresholds — src/spmv_cpu.cpp:34-50.
 * returns kernel_type (0 scalar, 1 vector, 2 merge-path); *use_texture = cols > 10000 */
int oracle_auto_config(int num_rows, int num_cols, int nnz, const int* row_ptrs, int* use_texture) {
    float st[4];
    oracle_csr_stats(num_rows, nnz, row_ptrs, st);
    *use_texture = num_cols > 10000;
    if (st[0] < 4.0f) return 0;
    if (st[3] < 10.0f) return 1;
    return 2;
}

/* algorithmic bytes of one SpMV — src/bandwidth.cpp:34-42 (CSR), :66-75 (ELL) */
double oracle_bytes_csr(int num_rows, int num_cols, int nnz) {
    return (double)nnz * 8.0 + ((double)num_rows + 1.0) * 4.0 + (double)num_cols * 4.0 + (double)num_rows * 4.0;
}
double oracle_bytes_ell(int num_rows, int num_cols, int K) {
    return (double)num_rows * (double)K * 8.0 + (double)num_cols * 4.0 + (double)num_rows * 4.0;
}

/* dangling columns: sequential fp32 column sums, dangling <=> sum == 0.0f —
 * src/pagerank.cu:20-48.  mask has num_cols bytes. */
void oracle_dangling_mask(int num_rows, int num_cols, const int* row_ptrs, const int* col_indices,
                          const float* values, unsigned char* mask) {
    float* sums = (float*)calloc((size_t)(num_cols > 0 ? num_cols : 1), sizeof(float));
    for (int r = 0; r < num_rows; r++) {
        for (int j = row_ptrs[r]; j < row_ptrs[r + 1]; j++) {
            int c = col_indices[j];
            if (c >= 0 && c < num_cols) sums[c] += values[j];
        }
    }
    for (int c = 0; c < num_cols; c++) mask[c] = sums[c] == 0.0f;
    free(sums);
}

/* PageRank power iteration — src/pagerank.cu:50-153 (host loop) with the SpMV
 * of src/spmv_cpu.cpp:6-16 in place of the device call.
 *   wide_sums == 0: every reduction is a sequential fp32 sum, as the reference;
 *   wide_sums != 0: reductions (dangling mass, residual, final sum) accumulate
 *                   in double — the conditioning fix used for n ~ 1e7.
 * ranks[n] receives the normalised result; returns iterations. */
int oracle_pagerank(int n, int num_cols, const int* row_ptrs, const int* col_indices,
                    const float* values, float damping, float tolerance, int max_iterations,
                    int wide_sums, float* ranks, float* final_residual, int* converged) {
    *final_residual = 0.0f;
    *converged = 0;
    if (n <= 0) return 0;

    float* r_old = (float*)malloc((size_t)n * sizeof(float));
    float* r_new = (float*)malloc((size_t)n * sizeof(float));
    unsigned char* dangling = (unsigned char*)malloc((size_t)(num_cols > 0 ? num_cols : 1));
    oracle_dangling_mask(n, num_cols, row_ptrs, col_indices, values, dangling);

    float init = 1.0f / n;
    for (int i = 0; i < n; i++) r_old[i] = init;
    float teleport = (1.0f - damping) / n;

    int iterations = 0;
    int final_from_new = 0;
    for (int iter = 0; iter < max_iterations; iter++) {
        float dangling_sum;
        if (wide_sums) {
            double s = 0.0;
            for (int c = 0; c < num_cols && c < n; c++) if (dangling[c]) s += r_old[c];
            dangling_sum = (float)s;
        } else {
            dangling_sum = 0.0f;
            for (int c = 0; c < num_cols && c < n; c++) if (dangling[c]) dangling_sum += r_old[c];
        }

        oracle_spmv_csr(n, row_ptrs, col_indices, values, r_old, r_new);

        float dangling_contrib = damping * dangling_sum / n;
        for (int i = 0; i < n; i++) {
            r_new[i] = damping * r_new[i] + dangling_contrib + teleport;
        }

        float residual;
        if (wide_sums) {
            double s = 0.0;
            for (int i = 0; i < n; i++) {
                float diff = r_new[i] - r_old[i];
                s += (double)(diff * diff);
            }
            residual = (float)sqrt(s);
        } else {
            float s = 0.0f;
            for (int i = 0; i < n; i++) {
                float diff = r_new[i] - r_old[i];
                s += diff * diff;
            }
            residual = sqrtf(s);
        }

        iterations = iter + 1;
        *final_residual = residual;
        if (residual < tolerance) {
            *converged = 1;
            final_from_new = 1;
            break;
        }
        float* t = r_old; r_old = r_new; r_new = t;
    }

    const float* last = final_from_new ? r_new : r_old;
    memcpy(ranks, last, (size_t)n * sizeof(float));

    if (wide_sums) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += ranks[i];
        float sf = (float)s;
        if (sf > 0.0f) for (int i = 0; i < n; i++) ranks[i] /= sf;
    } else {
        float s = 0.0f;
        for (int i = 0; i < n; i++) s += ranks[i];
        if (s > 0.0f) for (int i = 0; i < n; i++) ranks[i] /= s;
    }

    free(r_old);
    free(r_new);
    free(dangling);
    return iterations;
}
