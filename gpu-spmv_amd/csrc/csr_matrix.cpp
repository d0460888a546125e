// csr_matrix.cpp — CSR container: host construction/conversion, HBM upload,
// on-disk format, row statistics.
//
// Behaviour follows the reference's src/csr_matrix.cpp (create :10-32,
// destroy :34-48, from_dense :50-95, to_dense :97-114, get_element :116-135,
// to_gpu :138-165, from_gpu :167-182, free_gpu :184-200, serialize :202-230,
// deserialize :232-279, stats :281-300); the code is written for this library.
#include "internal.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <memory>

namespace spmv {

namespace {

void release_host(CSRMatrix* m) {
    if (m->owns_host_memory) {
        delete[] m->values;
        delete[] m->col_indices;
        delete[] m->row_ptrs;
    }
    m->values = nullptr;
    m->col_indices = nullptr;
    m->row_ptrs = nullptr;
}

// (re)allocates uninitialised host arrays for the given shape
void adopt_shape(CSRMatrix* m, int rows, int cols, int nnz) {
    release_host(m);
    m->num_rows = rows;
    m->num_cols = cols;
    m->nnz = nnz;
    m->values = nnz > 0 ? new float[nnz] : nullptr;
    m->col_indices = nnz > 0 ? new int[nnz] : nullptr;
    m->row_ptrs = new int[static_cast<size_t>(rows) + 1];
    m->owns_host_memory = true;
}

struct FileCloser { void operator()(FILE* f) const { if (f) fclose(f); } };
using File = std::unique_ptr<FILE, FileCloser>;

template <typename T>
bool put(FILE* f, const T* data, size_t count) {
    return count == 0 || fwrite(data, sizeof(T), count, f) == count;
}
template <typename T>
bool get(FILE* f, T* data, size_t count) {
    return count == 0 || fread(data, sizeof(T), count, f) == count;
}

} // namespace

CSRMatrix* csr_create(int rows, int cols, int nnz) {
    if (rows < 0 || cols < 0 || nnz < 0) return nullptr;

    CSRMatrix* m = new CSRMatrix{};
    m->owns_host_memory = true;   // so adopt_shape's release is a no-op on null arrays
    adopt_shape(m, rows, cols, nnz);
    if (nnz > 0) {
        std::fill_n(m->values, nnz, 0.0f);
        std::fill_n(m->col_indices, nnz, 0);
    }
    std::fill_n(m->row_ptrs, static_cast<size_t>(rows) + 1, 0);
    m->owns_device_memory = false;
    return m;
}

void csr_destroy(CSRMatrix* mat) {
    if (!mat) return;
    release_host(mat);
    if (mat->owns_device_memory) {
        csr_free_gpu(mat);
    } else if (mat->d_row_ptrs) {
        detail::aux_drop(mat->d_row_ptrs);   // wrapped device arrays: drop only our side table
    }
    delete mat;
}

int csr_from_dense(CSRMatrix* csr, const float* dense, int rows, int cols) {
    if (!csr || !dense || rows <= 0 || cols <= 0) {
        return detail::code(SpMVError::INVALID_ARGUMENT);
    }

    const size_t total = static_cast<size_t>(rows) * cols;
    const int nnz = static_cast<int>(std::count_if(dense, dense + total,
                                                   [](float v) { return v != 0.0f; }));
    adopt_shape(csr, rows, cols, nnz);

    int cursor = 0;
    for (int r = 0; r < rows; ++r) {
        csr->row_ptrs[r] = cursor;
        const float* line = dense + static_cast<size_t>(r) * cols;
        for (int c = 0; c < cols; ++c) {
            if (line[c] != 0.0f) {
                csr->values[cursor] = line[c];
                csr->col_indices[cursor] = c;
                ++cursor;
            }
        }
    }
    csr->row_ptrs[rows] = nnz;
    return detail::code(SpMVError::SUCCESS);
}

int csr_to_dense(const CSRMatrix* csr, float* dense) {
    if (!csr || !dense) return detail::code(SpMVError::INVALID_ARGUMENT);

    const size_t total = static_cast<size_t>(csr->num_rows) * csr->num_cols;
    std::fill_n(dense, total, 0.0f);
    for (int r = 0; r < csr->num_rows; ++r) {
        float* line = dense + static_cast<size_t>(r) * csr->num_cols;
        for (int j = csr->row_ptrs[r]; j < csr->row_ptrs[r + 1]; ++j) {
            line[csr->col_indices[j]] = csr->values[j];
        }
    }
    return detail::code(SpMVError::SUCCESS);
}

float csr_get_element(const CSRMatrix* mat, int row, int col) {
    if (!mat || row < 0 || row >= mat->num_rows || col < 0 || col >= mat->num_cols) {
        return 0.0f;
    }
    // Linear walk with an early exit once the stored column passes `col`
    // (same answers as the reference also for rows a caller left unsorted).
    for (int j = mat->row_ptrs[row]; j < mat->row_ptrs[row + 1]; ++j) {
        const int c = mat->col_indices[j];
        if (c == col) return mat->values[j];
        if (c > col) break;
    }
    return 0.0f;
}

int csr_to_gpu(CSRMatrix* mat) {
    if (!mat) return detail::code(SpMVError::INVALID_ARGUMENT);

    csr_free_gpu(mat);
    mat->owns_device_memory = true;   // set first so a partial failure is still freed

    const size_t nnz = static_cast<size_t>(mat->nnz);
    const size_t ptrs = static_cast<size_t>(mat->num_rows) + 1;

    struct Upload { void** dst; const void* src; size_t bytes; };
    const Upload plan[] = {
        {reinterpret_cast<void**>(&mat->d_values),      mat->values,      nnz * sizeof(float)},
        {reinterpret_cast<void**>(&mat->d_col_indices), mat->col_indices, nnz * sizeof(int)},
        {reinterpret_cast<void**>(&mat->d_row_ptrs),    mat->row_ptrs,    ptrs * sizeof(int)},
    };
    for (const Upload& u : plan) {
        if (u.bytes == 0) continue;
        if (hipMalloc(u.dst, u.bytes) != hipSuccess) {
            csr_free_gpu(mat);
            return detail::code(SpMVError::CUDA_MALLOC);
        }
        if (hipMemcpy(*u.dst, u.src, u.bytes, hipMemcpyHostToDevice) != hipSuccess) {
            csr_free_gpu(mat);
            return detail::code(SpMVError::CUDA_MEMCPY);
        }
    }
    mat->owns_device_memory = true;
    return detail::code(SpMVError::SUCCESS);
}

int csr_from_gpu(CSRMatrix* mat) {
    if (!mat || !mat->d_row_ptrs) return detail::code(SpMVError::INVALID_ARGUMENT);

    const size_t nnz = static_cast<size_t>(mat->nnz);
    if (nnz > 0 && mat->d_values && mat->d_col_indices) {
        SPMV_HIP_CHECK_AS(hipMemcpy(mat->values, mat->d_values, nnz * sizeof(float),
                                    hipMemcpyDeviceToHost), SpMVError::CUDA_MEMCPY);
        SPMV_HIP_CHECK_AS(hipMemcpy(mat->col_indices, mat->d_col_indices, nnz * sizeof(int),
                                    hipMemcpyDeviceToHost), SpMVError::CUDA_MEMCPY);
    }
    SPMV_HIP_CHECK_AS(hipMemcpy(mat->row_ptrs, mat->d_row_ptrs,
                                (static_cast<size_t>(mat->num_rows) + 1) * sizeof(int),
                                hipMemcpyDeviceToHost), SpMVError::CUDA_MEMCPY);
    return detail::code(SpMVError::SUCCESS);
}

void csr_free_gpu(CSRMatrix* mat) {
    if (!mat) return;
    if (mat->d_row_ptrs) detail::aux_drop(mat->d_row_ptrs);
    if (mat->owns_device_memory) {
        if (mat->d_values)      (void)hipFree(mat->d_values);
        if (mat->d_col_indices) (void)hipFree(mat->d_col_indices);
        if (mat->d_row_ptrs)    (void)hipFree(mat->d_row_ptrs);
    }
    mat->d_values = nullptr;
    mat->d_col_indices = nullptr;
    mat->d_row_ptrs = nullptr;
    mat->owns_device_memory = false;
}

void csr_invalidate_gpu_cache(const CSRMatrix* mat) {
    if (mat && mat->d_row_ptrs) detail::aux_drop(mat->d_row_ptrs);
}

int csr_serialize(const CSRMatrix* mat, const char* filename) {
    if (!mat || !filename) return detail::code(SpMVError::INVALID_ARGUMENT);

    File f(fopen(filename, "wb"));
    if (!f) return detail::code(SpMVError::FILE_IO);

    const int header[3] = {mat->num_rows, mat->num_cols, mat->nnz};
    const size_t nnz = static_cast<size_t>(mat->nnz);
    bool ok = put(f.get(), header, 3)
           && put(f.get(), mat->values, nnz)
           && put(f.get(), mat->col_indices, nnz)
           && put(f.get(), mat->row_ptrs, static_cast<size_t>(mat->num_rows) + 1);
    ok = ok && fflush(f.get()) == 0;
    return detail::code(ok ? SpMVError::SUCCESS : SpMVError::FILE_IO);
}

int csr_deserialize(CSRMatrix* mat, const char* filename) {
    if (!mat || !filename) return detail::code(SpMVError::INVALID_ARGUMENT);

    File f(fopen(filename, "rb"));
    if (!f) return detail::code(SpMVError::FILE_IO);

    int header[3];
    if (!get(f.get(), header, 3) || header[0] < 0 || header[1] < 0 || header[2] < 0) {
        return detail::code(SpMVError::FILE_IO);
    }
    // The header is not trusted: the payload it promises must be in the file before anything is allocated
    // (a corrupt nnz field must not turn into a multi-gigabyte allocation), ...
    const unsigned long long promised = 8ULL * static_cast<unsigned long long>(header[2]) +
                                        4ULL * (static_cast<unsigned long long>(header[0]) + 1);
    const long here = ftell(f.get());
    if (here < 0 || fseek(f.get(), 0, SEEK_END) != 0) return detail::code(SpMVError::FILE_IO);
    const long end = ftell(f.get());
    if (end < here || static_cast<unsigned long long>(end - here) < promised || fseek(f.get(), here, SEEK_SET) != 0) {
        return detail::code(SpMVError::FILE_IO);
    }
    adopt_shape(mat, header[0], header[1], header[2]);

    const size_t nnz = static_cast<size_t>(mat->nnz);
    bool ok = get(f.get(), mat->values, nnz)
           && get(f.get(), mat->col_indices, nnz)
           && get(f.get(), mat->row_ptrs, static_cast<size_t>(mat->num_rows) + 1);
    // ... and the arrays must describe a CSR matrix: every later loop (CPU and GPU) indexes with them unchecked
    if (ok) {
        ok = mat->row_ptrs[0] == 0 && mat->row_ptrs[mat->num_rows] == mat->nnz;
        for (int r = 0; ok && r < mat->num_rows; ++r) ok = mat->row_ptrs[r] <= mat->row_ptrs[r + 1];
        for (size_t j = 0; ok && j < nnz; ++j) ok = mat->col_indices[j] >= 0 && mat->col_indices[j] < mat->num_cols;
    }
    if (!ok) {                                   // never leave half-read arrays behind a plausible header
        adopt_shape(mat, 0, 0, 0);
        mat->row_ptrs[0] = 0;
    }
    return detail::code(ok ? SpMVError::SUCCESS : SpMVError::FILE_IO);
}

CSRStats csr_compute_stats(const CSRMatrix* mat) {
    CSRStats s{0.0f, 0, 0, 0.0f};
    if (!mat || mat->num_rows == 0) return s;

    s.avg_nnz_per_row = static_cast<float>(mat->nnz) / mat->num_rows;

    int longest = 0, shortest = INT_MAX;
    if (mat->row_ptrs) {
        for (int r = 0; r < mat->num_rows; ++r) {
            const int len = mat->row_ptrs[r + 1] - mat->row_ptrs[r];
            longest = std::max(longest, len);
            shortest = std::min(shortest, len);
        }
    } else if (mat->d_row_ptrs) {
        // device-only matrix (wrapped arrays): reduce on the device
        if (detail::device_row_stats(mat->d_row_ptrs, mat->num_rows, &longest, &shortest,
                                     detail::current_stream()) != hipSuccess) {
            longest = 0;
            shortest = 0;
        }
    } else {
        shortest = 0;
    }
    s.max_nnz_per_row = longest;
    s.min_nnz_per_row = shortest;
    s.skewness = static_cast<float>(longest) / (shortest + 1);
    return s;
}

} // namespace spmv
