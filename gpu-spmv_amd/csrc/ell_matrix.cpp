// ell_matrix.cpp — ELLPACK container (column-major slabs, -1 / 0.0f padding).
//
// Behaviour follows the reference's src/ell_matrix.cpp (create :8-36,
// from_dense :53-109, from_csr :111-159, to_dense :162-182, get_element
// :184-200, to_gpu :202-222, from_gpu :224-238, free_gpu :240-252,
// serialize :254-279, deserialize :281-324); the code is written for this library.
#include "internal.h"

#include <algorithm>
#include <cstdio>
#include <memory>

namespace spmv {

namespace {

size_t slots(const ELLMatrix* m) {
    return static_cast<size_t>(m->num_rows) * m->max_nnz_per_row;
}

void release_host(ELLMatrix* m) {
    if (m->owns_host_memory) {
        delete[] m->values;
        delete[] m->col_indices;
    }
    m->values = nullptr;
    m->col_indices = nullptr;
}

// (re)allocates host slabs filled with padding
void adopt_shape(ELLMatrix* m, int rows, int cols, int k) {
    release_host(m);
    m->num_rows = rows;
    m->num_cols = cols;
    m->max_nnz_per_row = k;
    const size_t n = slots(m);
    if (n > 0) {
        m->values = new float[n];
        m->col_indices = new int[n];
        std::fill_n(m->values, n, 0.0f);
        std::fill_n(m->col_indices, n, -1);
    }
    m->owns_host_memory = true;
}

struct FileCloser { void operator()(FILE* f) const { if (f) fclose(f); } };
using File = std::unique_ptr<FILE, FileCloser>;

} // namespace

ELLMatrix* ell_create(int rows, int cols, int max_nnz_per_row) {
    if (rows < 0 || cols < 0 || max_nnz_per_row < 0) return nullptr;
    ELLMatrix* m = new ELLMatrix{};
    m->owns_host_memory = true;
    adopt_shape(m, rows, cols, max_nnz_per_row);
    m->owns_device_memory = false;
    return m;
}

void ell_destroy(ELLMatrix* mat) {
    if (!mat) return;
    release_host(mat);
    if (mat->owns_device_memory) {
        ell_free_gpu(mat);
    } else if (mat->d_col_indices) {
        detail::ell_aux_drop(mat->d_col_indices);
    }
    delete mat;
}

int ell_from_dense(ELLMatrix* ell, const float* dense, int rows, int cols) {
    if (!ell || !dense || rows <= 0 || cols <= 0) {
        return detail::code(SpMVError::INVALID_ARGUMENT);
    }

    int widest = 0;
    for (int r = 0; r < rows; ++r) {
        const float* line = dense + static_cast<size_t>(r) * cols;
        const int len = static_cast<int>(std::count_if(line, line + cols,
                                                       [](float v) { return v != 0.0f; }));
        widest = std::max(widest, len);
    }
    adopt_shape(ell, rows, cols, widest);

    for (int r = 0; r < rows; ++r) {
        const float* line = dense + static_cast<size_t>(r) * cols;
        size_t slot = r;   // k = 0
        for (int c = 0; c < cols; ++c) {
            if (line[c] != 0.0f) {
                ell->values[slot] = line[c];
                ell->col_indices[slot] = c;
                slot += rows;   // next slab
            }
        }
    }
    return detail::code(SpMVError::SUCCESS);
}

int ell_from_csr(ELLMatrix* ell, const CSRMatrix* csr) {
    if (!ell || !csr) return detail::code(SpMVError::INVALID_ARGUMENT);

    int widest = 0;
    for (int r = 0; r < csr->num_rows; ++r) {
        widest = std::max(widest, csr->row_ptrs[r + 1] - csr->row_ptrs[r]);
    }
    adopt_shape(ell, csr->num_rows, csr->num_cols, widest);

    const size_t rows = csr->num_rows;
    for (size_t r = 0; r < rows; ++r) {
        size_t slot = r;
        for (int j = csr->row_ptrs[r]; j < csr->row_ptrs[r + 1]; ++j, slot += rows) {
            ell->values[slot] = csr->values[j];
            ell->col_indices[slot] = csr->col_indices[j];
        }
    }
    return detail::code(SpMVError::SUCCESS);
}

// Extension (SURVEY.md §8f "next" #1): the same conversion on the device, from the CSR's
// device arrays straight into freshly allocated device slabs — no host pass, no PCIe.
// The ELL matrix ends up device-only (host slabs released); ell_from_gpu can fetch them.
int ell_from_csr_gpu(ELLMatrix* ell, const CSRMatrix* csr) {
    if (!ell || !csr) return detail::code(SpMVError::INVALID_ARGUMENT);
    if (csr->num_rows > 0 && (!csr->d_row_ptrs || (csr->nnz > 0 && (!csr->d_col_indices || !csr->d_values)))) {
        return detail::code(SpMVError::INVALID_FORMAT);
    }
    hipStream_t stream = detail::current_stream();
    int widest = 0, shortest = 0;
    if (csr->num_rows > 0 &&
        detail::device_row_stats(csr->d_row_ptrs, csr->num_rows, &widest, &shortest, stream) != hipSuccess) {
        return detail::code(SpMVError::KERNEL_LAUNCH);
    }
    ell_free_gpu(ell);
    adopt_shape(ell, csr->num_rows, csr->num_cols, 0);      // drops the old host slabs
    ell->max_nnz_per_row = widest;
    ell->owns_device_memory = true;
    const size_t n = slots(ell);
    if (n > 0) {
        if (hipMalloc(reinterpret_cast<void**>(&ell->d_values), n * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ell->d_col_indices), n * sizeof(int)) != hipSuccess) {
            ell_free_gpu(ell);
            return detail::code(SpMVError::CUDA_MALLOC);
        }
        if (detail::launch_ell_from_csr(csr, widest, ell->d_col_indices, ell->d_values, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            ell_free_gpu(ell);
            return detail::code(SpMVError::KERNEL_LAUNCH);
        }
        // host slabs of the right size, filled on demand by ell_from_gpu
        ell->values = new float[n];
        ell->col_indices = new int[n];
        std::fill_n(ell->values, n, 0.0f);
        std::fill_n(ell->col_indices, n, -1);
    }
    ell->owns_device_memory = true;
    return detail::code(SpMVError::SUCCESS);
}

int ell_to_dense(const ELLMatrix* ell, float* dense) {
    if (!ell || !dense) return detail::code(SpMVError::INVALID_ARGUMENT);

    const size_t rows = ell->num_rows, cols = ell->num_cols;
    std::fill_n(dense, rows * cols, 0.0f);
    for (int k = 0; k < ell->max_nnz_per_row; ++k) {
        const size_t slab = static_cast<size_t>(k) * rows;
        for (size_t r = 0; r < rows; ++r) {
            const int c = ell->col_indices[slab + r];
            if (c >= 0) dense[r * cols + c] = ell->values[slab + r];
        }
    }
    return detail::code(SpMVError::SUCCESS);
}

float ell_get_element(const ELLMatrix* mat, int row, int col) {
    if (!mat || row < 0 || row >= mat->num_rows || col < 0 || col >= mat->num_cols) {
        return 0.0f;
    }
    size_t slot = row;
    for (int k = 0; k < mat->max_nnz_per_row; ++k, slot += mat->num_rows) {
        const int c = mat->col_indices[slot];
        if (c == col) return mat->values[slot];
        if (c < 0) break;   // reached the padding
    }
    return 0.0f;
}

int ell_to_gpu(ELLMatrix* mat) {
    if (!mat) return detail::code(SpMVError::INVALID_ARGUMENT);

    ell_free_gpu(mat);
    mat->owns_device_memory = true;

    const size_t n = slots(mat);
    if (n > 0) {
        if (hipMalloc(reinterpret_cast<void**>(&mat->d_values), n * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&mat->d_col_indices), n * sizeof(int)) != hipSuccess) {
            ell_free_gpu(mat);
            return detail::code(SpMVError::CUDA_MALLOC);
        }
        if (hipMemcpy(mat->d_values, mat->values, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(mat->d_col_indices, mat->col_indices, n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            ell_free_gpu(mat);
            return detail::code(SpMVError::CUDA_MEMCPY);
        }
    }
    mat->owns_device_memory = true;
    return detail::code(SpMVError::SUCCESS);
}

int ell_from_gpu(ELLMatrix* mat) {
    if (!mat) return detail::code(SpMVError::INVALID_ARGUMENT);

    const size_t n = slots(mat);
    if (n > 0 && mat->d_values && mat->d_col_indices) {
        SPMV_HIP_CHECK_AS(hipMemcpy(mat->values, mat->d_values, n * sizeof(float),
                                    hipMemcpyDeviceToHost), SpMVError::CUDA_MEMCPY);
        SPMV_HIP_CHECK_AS(hipMemcpy(mat->col_indices, mat->d_col_indices, n * sizeof(int),
                                    hipMemcpyDeviceToHost), SpMVError::CUDA_MEMCPY);
    }
    return detail::code(SpMVError::SUCCESS);
}

void ell_free_gpu(ELLMatrix* mat) {
    if (!mat) return;
    if (mat->d_col_indices) detail::ell_aux_drop(mat->d_col_indices);
    if (mat->owns_device_memory) {
        if (mat->d_values)      (void)hipFree(mat->d_values);
        if (mat->d_col_indices) (void)hipFree(mat->d_col_indices);
    }
    mat->d_values = nullptr;
    mat->d_col_indices = nullptr;
    mat->owns_device_memory = false;
}

void ell_invalidate_gpu_cache(const ELLMatrix* mat) {
    if (mat && mat->d_col_indices) detail::ell_aux_drop(mat->d_col_indices);
}

int ell_serialize(const ELLMatrix* mat, const char* filename) {
    if (!mat || !filename) return detail::code(SpMVError::INVALID_ARGUMENT);

    File f(fopen(filename, "wb"));
    if (!f) return detail::code(SpMVError::FILE_IO);

    const int header[3] = {mat->num_rows, mat->num_cols, mat->max_nnz_per_row};
    const size_t n = slots(mat);
    bool ok = fwrite(header, sizeof(int), 3, f.get()) == 3;
    if (ok && n > 0) {
        ok = fwrite(mat->values, sizeof(float), n, f.get()) == n
          && fwrite(mat->col_indices, sizeof(int), n, f.get()) == n;
    }
    ok = ok && fflush(f.get()) == 0;
    return detail::code(ok ? SpMVError::SUCCESS : SpMVError::FILE_IO);
}

int ell_deserialize(ELLMatrix* mat, const char* filename) {
    if (!mat || !filename) return detail::code(SpMVError::INVALID_ARGUMENT);

    File f(fopen(filename, "rb"));
    if (!f) return detail::code(SpMVError::FILE_IO);

    int header[3];
    if (fread(header, sizeof(int), 3, f.get()) != 3 ||
        header[0] < 0 || header[1] < 0 || header[2] < 0) {
        return detail::code(SpMVError::FILE_IO);
    }
    // the header is not trusted: the slabs it promises must be in the file before anything is allocated
    const unsigned long long promised = 8ULL * static_cast<unsigned long long>(header[0]) * static_cast<unsigned long long>(header[2]);
    const long here = ftell(f.get());
    if (here < 0 || fseek(f.get(), 0, SEEK_END) != 0) return detail::code(SpMVError::FILE_IO);
    const long end = ftell(f.get());
    if (end < here || static_cast<unsigned long long>(end - here) < promised || promised > 0x7fffffffULL * 8ULL ||
        fseek(f.get(), here, SEEK_SET) != 0) {
        return detail::code(SpMVError::FILE_IO);
    }
    adopt_shape(mat, header[0], header[1], header[2]);

    const size_t n = slots(mat);
    bool ok = true;
    if (n > 0) {
        ok = fread(mat->values, sizeof(float), n, f.get()) == n
          && fread(mat->col_indices, sizeof(int), n, f.get()) == n;
    }
    // column indices are used unchecked by every kernel: -1 marks padding, anything else must be a column
    for (size_t j = 0; ok && j < n; ++j) ok = mat->col_indices[j] >= -1 && mat->col_indices[j] < mat->num_cols;
    if (!ok) adopt_shape(mat, 0, 0, 0);
    return detail::code(ok ? SpMVError::SUCCESS : SpMVError::FILE_IO);
}

} // namespace spmv
