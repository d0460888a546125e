// generators.h — device-side synthetic input generators (see generators.hip).
#ifndef SPMV_AMD_GENERATORS_H
#define SPMV_AMD_GENERATORS_H

#include "internal.h"

namespace spmv {
namespace detail {

int gen_uniform_rows(unsigned long long seed, int row_begin, int local_rows, int n_cols, int k,
                     int* d_row_ptrs, int* d_cols, float* d_vals, hipStream_t s);
int gen_uniform_ell(unsigned long long seed, int rows, int n_cols, int k, int* d_cols, float* d_vals,
                    hipStream_t s);
int gen_stratified_rows(unsigned long long seed, int row_begin, int local_rows, int n_cols,
                        const int* d_row_ptrs, int* d_cols, float* d_vals, hipStream_t s);
int gen_vector(unsigned long long seed, unsigned long long tag, size_t n, float* d_x, hipStream_t s);
int count_columns(long long nnz, const int* d_cols, int n_cols, int* d_counts, hipStream_t s);
int reciprocal_values(long long nnz, const int* d_cols, const int* d_counts, float* d_vals,
                      hipStream_t s);

} // namespace detail
} // namespace spmv

#endif
