// spmv/ell_matrix.h — ELLPACK container, column-major slabs so that
// consecutive rows sit in consecutive addresses (one coalesced wavefront
// load per slab on gfx950).
//
// Layout and signatures follow the reference (include/spmv/ell_matrix.h:12-66);
// sizeof(ELLMatrix) == 56; padding slots hold (col = -1, value = 0.0f).
#ifndef SPMV_ELL_MATRIX_H
#define SPMV_ELL_MATRIX_H

#include "common.h"
#include "csr_matrix.h"
#include <cstddef>

namespace spmv {

struct ELLMatrix {
    int num_rows;
    int num_cols;
    int max_nnz_per_row;   // K

    // slot (row, k) lives at k * num_rows + row
    float* values;         // [num_rows * K]
    int*   col_indices;    // [num_rows * K], -1 marks padding

    float* d_values;
    int*   d_col_indices;

    bool owns_host_memory;
    bool owns_device_memory;
};

ELLMatrix* ell_create(int rows, int cols, int max_nnz_per_row);
void ell_destroy(ELLMatrix* mat);

int ell_from_dense(ELLMatrix* ell, const float* dense, int rows, int cols);
int ell_from_csr(ELLMatrix* ell, const CSRMatrix* csr);
// extension: the conversion on the device (csr must be uploaded); fills ell's device slabs
// only — call ell_from_gpu to mirror them into the host slabs
int ell_from_csr_gpu(ELLMatrix* ell, const CSRMatrix* csr);
int ell_to_dense(const ELLMatrix* ell, float* dense);
float ell_get_element(const ELLMatrix* mat, int row, int col);

int ell_to_gpu(ELLMatrix* mat);
int ell_from_gpu(ELLMatrix* mat);
void ell_free_gpu(ELLMatrix* mat);
// extension: drops the cached auxiliary data of the device arrays (see csr_invalidate_gpu_cache)
void ell_invalidate_gpu_cache(const ELLMatrix* mat);

// File layout: int32 rows, cols, K; float[rows*K]; int32[rows*K] (column-major).
int ell_serialize(const ELLMatrix* mat, const char* filename);
int ell_deserialize(ELLMatrix* mat, const char* filename);

inline int ell_index(int row, int k, int num_rows) {
    return k * num_rows + row;
}

} // namespace spmv

#endif // SPMV_ELL_MATRIX_H
