// spmv/csr_matrix.h — CSR container (host arrays + device mirrors in HBM).
//
// Field order, types and function signatures follow the reference
// (include/spmv/csr_matrix.h:11-71) because callers read and write the fields
// directly (reference tests/test_csr.cpp:178-185); sizeof(CSRMatrix) == 72.
#ifndef SPMV_CSR_MATRIX_H
#define SPMV_CSR_MATRIX_H

#include "common.h"
#include <cstddef>
#include <vector>

namespace spmv {

struct CSRMatrix {
    int num_rows;
    int num_cols;
    int nnz;

    // host arrays (new[]-owned when owns_host_memory)
    float* values;        // [nnz]
    int*   col_indices;   // [nnz], ascending inside a row
    int*   row_ptrs;      // [num_rows + 1]

    // device arrays (hipMalloc-owned when owns_device_memory)
    float* d_values;
    int*   d_col_indices;
    int*   d_row_ptrs;

    bool owns_host_memory;
    bool owns_device_memory;
};

CSRMatrix* csr_create(int rows, int cols, int nnz);   // nullptr on negative sizes
void csr_destroy(CSRMatrix* mat);

// dense is row-major [rows * cols]; entries != 0.0f are kept, columns ascending.
int csr_from_dense(CSRMatrix* csr, const float* dense, int rows, int cols);
int csr_to_dense(const CSRMatrix* csr, float* dense);
float csr_get_element(const CSRMatrix* mat, int row, int col);   // 0.0f when absent / out of range

int csr_to_gpu(CSRMatrix* mat);     // (re)uploads host arrays to HBM
int csr_from_gpu(CSRMatrix* mat);   // downloads device arrays into the host arrays
void csr_free_gpu(CSRMatrix* mat);
// extension: per-matrix auxiliary data (row statistics, merge-path tables, the LDS-tiled plan with its
// copy of the entries) is derived from the device arrays at first use and cached until they are
// freed or re-uploaded.  After writing into d_values / d_col_indices / d_row_ptrs IN PLACE, call this
// so that the next call rebuilds it.
void csr_invalidate_gpu_cache(const CSRMatrix* mat);

// File layout: int32 rows, cols, nnz; float[nnz]; int32[nnz]; int32[rows+1] (native endian).
int csr_serialize(const CSRMatrix* mat, const char* filename);
int csr_deserialize(CSRMatrix* mat, const char* filename);

struct CSRStats {
    float avg_nnz_per_row;
    int   max_nnz_per_row;
    int   min_nnz_per_row;
    float skewness;   // max / (min + 1)
};

CSRStats csr_compute_stats(const CSRMatrix* mat);

} // namespace spmv

#endif // SPMV_CSR_MATRIX_H
