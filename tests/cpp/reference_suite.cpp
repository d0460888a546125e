// reference_suite.cpp — the 48 cases of the reference's gtest files, restated against the drop-in
// headers with a 30-line harness of our own (the image has no gtest).  Same case names, so a
// maintainer can put the two side by side; each case cites the reference test it stands for.
// The checks are at least as strict as the originals (bit-exact where the library promises it).
// Plain host C++: g++ -Iinclude ... -lspmv_amd.  Needs a GPU to run.
#include "spmv/spmv.h"
#include "spmv/bandwidth.h"
#include "spmv/benchmark.h"
#include "spmv/cuda_buffer.h"
#include "spmv/pagerank.h"
#include "spmv/test_utils.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

using namespace spmv;
using namespace spmv::test;

// ---------------------------------------------------------------- harness ----
namespace {

struct Case {
    const char* name;
    std::function<void()> body;
};
std::vector<Case>& registry() {
    static std::vector<Case> cases;
    return cases;
}
struct Registrar {
    Registrar(const char* name, std::function<void()> body) { registry().push_back({name, std::move(body)}); }
};
int g_failed_checks = 0;

#define CASE(suite, name)                                                        \
    static void suite##_##name();                                               \
    static Registrar reg_##suite##_##name(#suite "." #name, suite##_##name);    \
    static void suite##_##name()
#define EXPECT(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            ++g_failed_checks;                                                  \
            std::printf("    FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
        }                                                                       \
    } while (0)

constexpr int kRounds = 25;     // iterations of the randomised ("property") cases
const int kOk = static_cast<int>(SpMVError::SUCCESS);

struct Csr {                    // owning handle: destroyed on scope exit
    CSRMatrix* m;
    Csr() : m(csr_create(0, 0, 0)) {}
    ~Csr() { csr_destroy(m); }
    CSRMatrix* operator->() const { return m; }
};
struct Ell {
    ELLMatrix* m;
    Ell() : m(ell_create(0, 0, 0)) {}
    ~Ell() { ell_destroy(m); }
    ELLMatrix* operator->() const { return m; }
};

// |got - want| <= 1e-5 * max(|want|, sum_j |a_ij x_j|): the criterion for kernels that reorder a row's sum
bool reordered_ok(const CSRMatrix* A, const float* x, const float* want, const float* got) {
    for (int i = 0; i < A->num_rows; ++i) {
        double bound = std::fabs(want[i]);
        double abs_sum = 0.0;
        for (int j = A->row_ptrs[i]; j < A->row_ptrs[i + 1]; ++j) {
            abs_sum += std::fabs(static_cast<double>(A->values[j]) * x[A->col_indices[j]]);
        }
        bound = std::max(bound, abs_sum);
        if (std::fabs(static_cast<double>(want[i]) - got[i]) > 1e-5 * std::max(bound, 1e-30)) return false;
    }
    return true;
}

std::vector<float> run_csr(const CSRMatrix* A, const std::vector<float>& x, SpMVConfig::KernelType kernel,
                           bool keep_x_on_chip = false) {
    CudaBuffer<float> d_x(x.size()), d_y(A->num_rows);
    d_x.copyFromHost(x.data(), x.size());
    SpMVConfig config;
    config.kernel_type = kernel;
    config.use_texture = keep_x_on_chip;
    const SpMVResult r = spmv_csr(A, d_x.get(), d_y.get(), &config, static_cast<int>(x.size()));
    EXPECT(r.error_code == kOk);
    std::vector<float> y(A->num_rows);
    d_y.copyToHost(y.data(), y.size());
    return y;
}

} // namespace

// ------------------------------------------------- tests/test_common.cpp ----
CASE(CommonTest, ErrorStringConversion) {          // reference tests/test_common.cpp:8-18
    EXPECT(std::strcmp(spmv_error_string(SpMVError::SUCCESS), "Success") == 0);
    EXPECT(std::strcmp(spmv_error_string(SpMVError::INVALID_DIMENSION), "Invalid matrix/vector dimension") == 0);
    EXPECT(std::strcmp(spmv_error_string(SpMVError::CUDA_MALLOC), "CUDA memory allocation failed") == 0);
    EXPECT(std::strcmp(spmv_error_string(SpMVError::FILE_IO), "File I/O error") == 0);
    for (int code = 0; code >= -8; --code) EXPECT(spmv_error_string(static_cast<SpMVError>(code))[0] != '\0');
}
CASE(CudaBufferTest, DefaultConstruction) {        // :21-26
    CudaBuffer<float> b;
    EXPECT(b.get() == nullptr && b.size() == 0 && b.empty());
}
CASE(CudaBufferTest, SizedConstruction) {          // :28-33
    CudaBuffer<float> b(100);
    EXPECT(b.get() != nullptr && b.size() == 100 && !b.empty());
}
CASE(CudaBufferTest, ZeroSizeConstruction) {       // :35-40
    CudaBuffer<int> b(0);
    EXPECT(b.get() == nullptr && b.size() == 0 && b.empty());
}
CASE(CudaBufferTest, MoveConstruction) {           // :42-51
    CudaBuffer<float> a(64);
    float* p = a.get();
    CudaBuffer<float> b(std::move(a));
    EXPECT(b.get() == p && b.size() == 64 && a.get() == nullptr && a.size() == 0);
}
CASE(CudaBufferTest, MoveAssignment) {             // :53-62
    CudaBuffer<float> a(64), b(8);
    float* p = a.get();
    b = std::move(a);
    EXPECT(b.get() == p && b.size() == 64 && a.get() == nullptr && a.size() == 0);
}
CASE(CudaBufferTest, CopyFromHost) {               // :64-76
    std::vector<float> host(257), back(257, -1.0f);
    std::iota(host.begin(), host.end(), 0.5f);
    CudaBuffer<float> b(host.size());
    b.copyFromHost(host.data(), host.size());
    b.copyToHost(back.data(), back.size());
    EXPECT(host == back);
    bool threw = false;
    try { b.copyFromHost(host.data(), host.size() + 1); } catch (const std::runtime_error&) { threw = true; }
    EXPECT(threw);
}
CASE(CudaBufferTest, Resize) {                     // :78-89
    CudaBuffer<float> b(10);
    b.resize(200);
    EXPECT(b.size() == 200 && b.get() != nullptr);
    b.resize(0);
    EXPECT(b.size() == 0 && b.get() == nullptr);
}
CASE(CudaBufferTest, Release) {                    // :91-98
    CudaBuffer<float> b(10);
    b.release();
    EXPECT(b.get() == nullptr && b.size() == 0 && b.empty());
}

// ---------------------------------------------------- tests/test_csr.cpp ----
CASE(CSRPropertyTest, DenseToSparseRoundTrip) {    // reference tests/test_csr.cpp:18-43
    RandomGenerator rng(101);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 60), cols = rng.randInt(1, 60);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.05f, 0.6f), rng);
        Csr csr;
        EXPECT(csr_from_dense(csr.m, dense.data(), rows, cols) == kOk);
        EXPECT(csr->nnz == static_cast<int>(std::count_if(dense.begin(), dense.end(), [](float v) { return v != 0.0f; })));
        std::vector<float> back(dense.size(), -1.0f);
        EXPECT(csr_to_dense(csr.m, back.data()) == kOk);
        EXPECT(std::memcmp(back.data(), dense.data(), dense.size() * sizeof(float)) == 0);
        for (int i = 0; i < rows; ++i) {           // ascending columns inside a row, monotone row_ptrs
            EXPECT(csr->row_ptrs[i] <= csr->row_ptrs[i + 1]);
            for (int j = csr->row_ptrs[i] + 1; j < csr->row_ptrs[i + 1]; ++j) {
                EXPECT(csr->col_indices[j - 1] < csr->col_indices[j]);
            }
        }
        EXPECT(csr->row_ptrs[0] == 0 && csr->row_ptrs[rows] == csr->nnz);
    }
}
CASE(CSRPropertyTest, ElementLookupCorrectness) {  // :47-76
    RandomGenerator rng(102);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 40), cols = rng.randInt(1, 40);
        const auto dense = generateRandomDenseMatrix(rows, cols, 0.3f, rng);
        Csr csr;
        csr_from_dense(csr.m, dense.data(), rows, cols);
        bool same = true;
        for (int i = 0; i < rows; ++i) {
            for (int j = 0; j < cols; ++j) same = same && csr_get_element(csr.m, i, j) == dense[static_cast<size_t>(i) * cols + j];
        }
        EXPECT(same);
        EXPECT(csr_get_element(csr.m, -1, 0) == 0.0f && csr_get_element(csr.m, rows, 0) == 0.0f);
        EXPECT(csr_get_element(csr.m, 0, -1) == 0.0f && csr_get_element(csr.m, 0, cols) == 0.0f);
    }
}
CASE(CSRPropertyTest, SerializationRoundTrip) {    // :80-126
    RandomGenerator rng(103);
    const std::string path = "/tmp/reference_suite_csr.bin";
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 50), cols = rng.randInt(1, 50);
        const auto dense = generateRandomDenseMatrix(rows, cols, 0.25f, rng);
        Csr a, b;
        csr_from_dense(a.m, dense.data(), rows, cols);
        EXPECT(csr_serialize(a.m, path.c_str()) == kOk);
        EXPECT(csr_deserialize(b.m, path.c_str()) == kOk);
        EXPECT(b->num_rows == rows && b->num_cols == cols && b->nnz == a->nnz);
        EXPECT(intArraysEqual(a->row_ptrs, b->row_ptrs, rows + 1));
        EXPECT(intArraysEqual(a->col_indices, b->col_indices, a->nnz));
        EXPECT(std::memcmp(a->values, b->values, a->nnz * sizeof(float)) == 0);
    }
    std::remove(path.c_str());
    Csr c;
    EXPECT(csr_deserialize(c.m, "/tmp/reference_suite_no_such_file.bin") == static_cast<int>(SpMVError::FILE_IO));
}
CASE(CSRUnitTest, EmptyMatrix) {                   // :130-137
    Csr csr;
    EXPECT(csr.m != nullptr && csr->num_rows == 0 && csr->num_cols == 0 && csr->nnz == 0);
    EXPECT(csr_create(-1, 2, 0) == nullptr);
}
CASE(CSRUnitTest, AllZeroMatrix) {                 // :139-151
    const std::vector<float> dense(12, 0.0f);
    Csr csr;
    EXPECT(csr_from_dense(csr.m, dense.data(), 3, 4) == kOk);
    EXPECT(csr->nnz == 0 && csr->num_rows == 3 && csr->num_cols == 4);
    for (int i = 0; i <= 3; ++i) EXPECT(csr->row_ptrs[i] == 0);
}
CASE(CSRUnitTest, SingleElementMatrix) {           // :153-166
    const std::vector<float> dense = {0, 0, 0, 0, 7.5f, 0, 0, 0, 0};
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 3, 3);
    EXPECT(csr->nnz == 1 && csr->values[0] == 7.5f && csr->col_indices[0] == 1);
    EXPECT(csr->row_ptrs[0] == 0 && csr->row_ptrs[1] == 0 && csr->row_ptrs[2] == 1 && csr->row_ptrs[3] == 1);
    EXPECT(csr_get_element(csr.m, 1, 1) == 7.5f && csr_get_element(csr.m, 0, 0) == 0.0f);
}
CASE(CSRUnitTest, GPUTransfer) {                   // :168-200
    const std::vector<float> dense = {1, 0, 2, 0, 3, 4, 0, 0, 5};
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 3, 3);
    EXPECT(csr->d_values == nullptr && !csr->owns_device_memory);
    EXPECT(csr_to_gpu(csr.m) == kOk);
    EXPECT(csr->d_values && csr->d_col_indices && csr->d_row_ptrs && csr->owns_device_memory);
    std::fill(csr->values, csr->values + csr->nnz, 0.0f);          // wipe the host side, fetch it back
    EXPECT(csr_from_gpu(csr.m) == kOk);
    const float want[5] = {1, 2, 3, 4, 5};
    EXPECT(std::memcmp(csr->values, want, sizeof(want)) == 0);
    csr_free_gpu(csr.m);
    EXPECT(csr->d_values == nullptr && csr->d_row_ptrs == nullptr && !csr->owns_device_memory);
}

// ---------------------------------------------------- tests/test_ell.cpp ----
CASE(ELLPropertyTest, DenseToSparseRoundTrip) {    // reference tests/test_ell.cpp:19-44
    RandomGenerator rng(201);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 50), cols = rng.randInt(1, 50);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.05f, 0.5f), rng);
        Ell ell;
        EXPECT(ell_from_dense(ell.m, dense.data(), rows, cols) == kOk);
        std::vector<float> back(dense.size(), -1.0f);
        EXPECT(ell_to_dense(ell.m, back.data()) == kOk);
        EXPECT(std::memcmp(back.data(), dense.data(), dense.size() * sizeof(float)) == 0);
    }
}
CASE(ELLPropertyTest, PaddingCorrectness) {        // :48-80
    RandomGenerator rng(202);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(2, 40), cols = rng.randInt(2, 40);
        const auto dense = generateRandomDenseMatrix(rows, cols, 0.2f, rng);
        Ell ell;
        ell_from_dense(ell.m, dense.data(), rows, cols);
        int widest = 0;
        for (int i = 0; i < rows; ++i) {
            int in_row = 0;
            for (int j = 0; j < cols; ++j) in_row += dense[static_cast<size_t>(i) * cols + j] != 0.0f;
            widest = std::max(widest, in_row);
            for (int k = 0; k < ell->max_nnz_per_row; ++k) {
                const size_t slot = ell_index(i, k, rows);
                if (k < in_row) EXPECT(ell->col_indices[slot] >= 0 && ell->values[slot] != 0.0f);
                else            EXPECT(ell->col_indices[slot] == -1 && ell->values[slot] == 0.0f);
            }
        }
        EXPECT(ell->max_nnz_per_row == widest);
    }
}
CASE(ELLPropertyTest, ColumnMajorLayout) {         // :84-108
    const std::vector<float> dense = {1, 2, 0, 0,  0, 3, 4, 0,  0, 0, 0, 5};      // design.md's 3 x 4 example
    Ell ell;
    ell_from_dense(ell.m, dense.data(), 3, 4);
    EXPECT(ell->max_nnz_per_row == 2);
    const float want_values[6] = {1, 3, 5, 2, 4, 0};
    const int want_cols[6] = {0, 1, 3, 1, 2, -1};
    EXPECT(std::memcmp(ell->values, want_values, sizeof(want_values)) == 0);
    EXPECT(std::memcmp(ell->col_indices, want_cols, sizeof(want_cols)) == 0);
    for (int row = 0; row < 3; ++row) {
        for (int k = 0; k < 2; ++k) EXPECT(ell_index(row, k, 3) == k * 3 + row);
    }
}
CASE(ELLPropertyTest, SerializationRoundTrip) {    // :112-149
    RandomGenerator rng(204);
    const std::string path = "/tmp/reference_suite_ell.bin";
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 40), cols = rng.randInt(1, 40);
        const auto dense = generateRandomDenseMatrix(rows, cols, 0.3f, rng);
        Ell a, b;
        ell_from_dense(a.m, dense.data(), rows, cols);
        EXPECT(ell_serialize(a.m, path.c_str()) == kOk && ell_deserialize(b.m, path.c_str()) == kOk);
        EXPECT(b->num_rows == rows && b->num_cols == cols && b->max_nnz_per_row == a->max_nnz_per_row);
        const size_t slots = static_cast<size_t>(rows) * a->max_nnz_per_row;
        EXPECT(std::memcmp(a->values, b->values, slots * sizeof(float)) == 0);
        EXPECT(std::memcmp(a->col_indices, b->col_indices, slots * sizeof(int)) == 0);
    }
    std::remove(path.c_str());
}
CASE(ELLUnitTest, FromCSR) {                       // :153-172
    RandomGenerator rng(205);
    const auto dense = generateRandomDenseMatrix(37, 29, 0.3f, rng);
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 37, 29);
    Ell from_csr, from_dense;
    EXPECT(ell_from_csr(from_csr.m, csr.m) == kOk);
    ell_from_dense(from_dense.m, dense.data(), 37, 29);
    EXPECT(from_csr->max_nnz_per_row == from_dense->max_nnz_per_row);
    const size_t slots = static_cast<size_t>(37) * from_csr->max_nnz_per_row;
    EXPECT(std::memcmp(from_csr->values, from_dense->values, slots * sizeof(float)) == 0);
    EXPECT(std::memcmp(from_csr->col_indices, from_dense->col_indices, slots * sizeof(int)) == 0);
    // and the device-side conversion (extension) produces the same slabs
    csr_to_gpu(csr.m);
    Ell on_device;
    EXPECT(ell_from_csr_gpu(on_device.m, csr.m) == kOk && ell_from_gpu(on_device.m) == kOk);
    EXPECT(std::memcmp(on_device->values, from_dense->values, slots * sizeof(float)) == 0);
    EXPECT(std::memcmp(on_device->col_indices, from_dense->col_indices, slots * sizeof(int)) == 0);
}
CASE(ELLUnitTest, GPUTransfer) {                   // :174-200
    const std::vector<float> dense = {1, 0, 2, 0, 3, 4, 0, 0, 5};
    Ell ell;
    ell_from_dense(ell.m, dense.data(), 3, 3);
    EXPECT(ell->d_values == nullptr);
    EXPECT(ell_to_gpu(ell.m) == kOk);
    EXPECT(ell->d_values && ell->d_col_indices && ell->owns_device_memory);
    const std::vector<float> saved(ell->values, ell->values + 3 * ell->max_nnz_per_row);
    std::fill(ell->values, ell->values + saved.size(), -9.0f);
    EXPECT(ell_from_gpu(ell.m) == kOk);
    EXPECT(std::memcmp(ell->values, saved.data(), saved.size() * sizeof(float)) == 0);
    ell_free_gpu(ell.m);
    EXPECT(ell->d_values == nullptr && !ell->owns_device_memory);
}

// ---------------------------------------- tests/test_kernel_selector.cpp ----
namespace {
// rows x cols matrix whose row i holds lens[i] entries (columns 0 .. lens[i]-1)
void csr_with_row_lengths(CSRMatrix* csr, const std::vector<int>& lens, int cols) {
    std::vector<float> dense(lens.size() * static_cast<size_t>(cols), 0.0f);
    for (size_t i = 0; i < lens.size(); ++i) {
        for (int j = 0; j < lens[i]; ++j) dense[i * cols + j] = 1.0f + j;
    }
    csr_from_dense(csr, dense.data(), static_cast<int>(lens.size()), cols);
}
} // namespace
CASE(KernelSelectorPropertyTest, SelectorValidity) {   // reference tests/test_kernel_selector.cpp:17-49
    RandomGenerator rng(301);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 80), cols = rng.randInt(1, 80);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.02f, 0.7f), rng);
        Csr csr;
        csr_from_dense(csr.m, dense.data(), rows, cols);
        const SpMVConfig c = spmv_auto_config(csr.m);
        EXPECT(c.kernel_type == SpMVConfig::SCALAR_CSR || c.kernel_type == SpMVConfig::VECTOR_CSR ||
               c.kernel_type == SpMVConfig::MERGE_PATH);
        EXPECT(c.block_size >= 32 && c.block_size <= 1024 && c.block_size % 32 == 0);
        EXPECT(c.use_texture == (cols > 10000));
        // and the rule itself: avg < 4 -> scalar; skew = max / (min + 1) < 10 -> vector; else merge-path
        const CSRStats s = csr_compute_stats(csr.m);
        const SpMVConfig::KernelType want = s.avg_nnz_per_row < 4.0f ? SpMVConfig::SCALAR_CSR
                                          : s.skewness < 10.0f       ? SpMVConfig::VECTOR_CSR
                                                                     : SpMVConfig::MERGE_PATH;
        EXPECT(c.kernel_type == want);
    }
}
CASE(KernelSelectorUnitTest, ShortRowsSelectScalar) {  // :53-71
    Csr csr;
    csr_with_row_lengths(csr.m, std::vector<int>(20, 2), 20);
    EXPECT(spmv_auto_config(csr.m).kernel_type == SpMVConfig::SCALAR_CSR);
}
CASE(KernelSelectorUnitTest, UniformRowsSelectVector) {    // :73-93
    Csr csr;
    csr_with_row_lengths(csr.m, std::vector<int>(20, 12), 20);
    EXPECT(spmv_auto_config(csr.m).kernel_type == SpMVConfig::VECTOR_CSR);
}
CASE(KernelSelectorUnitTest, SkewedRowsSelectMergePath) {  // :95-118
    std::vector<int> lens(40, 4);
    lens[7] = 100;                                  // skew = 100 / (4 + 1) = 20
    Csr csr;
    csr_with_row_lengths(csr.m, lens, 100);
    EXPECT(spmv_auto_config(csr.m).kernel_type == SpMVConfig::MERGE_PATH);
}
CASE(KernelSelectorUnitTest, LargeVectorUsesTexture) {     // :120-137
    Csr narrow, wide;
    csr_with_row_lengths(narrow.m, std::vector<int>(4, 3), 10000);
    csr_with_row_lengths(wide.m, std::vector<int>(4, 3), 10001);
    EXPECT(!spmv_auto_config(narrow.m).use_texture && spmv_auto_config(wide.m).use_texture);
}

// --------------------------------------------------- tests/test_spmv.cu ----
CASE(SpMVPropertyTest, CSRCorrectness) {           // reference tests/test_spmv.cu:40-78 (all three kernels)
    RandomGenerator rng(401);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 120), cols = rng.randInt(1, 120);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.02f, 0.5f), rng);
        const auto x = generateRandomVector(cols, rng);
        Csr csr;
        csr_from_dense(csr.m, dense.data(), rows, cols);
        csr_to_gpu(csr.m);
        std::vector<float> want(rows);
        spmv_cpu_csr(csr.m, x.data(), want.data());
        const auto scalar = run_csr(csr.m, x, SpMVConfig::SCALAR_CSR);
        EXPECT(std::memcmp(scalar.data(), want.data(), rows * sizeof(float)) == 0);   // CPU summation order: bit-exact
        EXPECT(reordered_ok(csr.m, x.data(), want.data(), run_csr(csr.m, x, SpMVConfig::VECTOR_CSR).data()));
        EXPECT(reordered_ok(csr.m, x.data(), want.data(), run_csr(csr.m, x, SpMVConfig::MERGE_PATH).data()));
        EXPECT(reordered_ok(csr.m, x.data(), want.data(), run_csr(csr.m, x, SpMVConfig::VECTOR_CSR, true).data()));
    }
}
CASE(SpMVPropertyTest, ELLCorrectness) {           // :82-118
    RandomGenerator rng(402);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(1, 120), cols = rng.randInt(1, 120);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.02f, 0.4f), rng);
        const auto x = generateRandomVector(cols, rng);
        Ell ell;
        ell_from_dense(ell.m, dense.data(), rows, cols);
        ell_to_gpu(ell.m);
        std::vector<float> want(rows), got(rows);
        spmv_cpu_ell(ell.m, x.data(), want.data());
        CudaBuffer<float> d_x(cols), d_y(rows);
        d_x.copyFromHost(x.data(), cols);
        EXPECT(spmv_ell(ell.m, d_x.get(), d_y.get(), nullptr, cols).error_code == kOk);
        d_y.copyToHost(got.data(), rows);
        EXPECT(std::memcmp(got.data(), want.data(), rows * sizeof(float)) == 0);      // bit-exact
    }
}
CASE(SpMVPropertyTest, DimensionValidation) {      // :122-144
    RandomGenerator rng(403);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(2, 40), cols = rng.randInt(2, 40);
        const auto dense = generateRandomDenseMatrix(rows, cols, 0.3f, rng);
        Csr csr;
        csr_from_dense(csr.m, dense.data(), rows, cols);
        csr_to_gpu(csr.m);
        CudaBuffer<float> d_x(cols + 5), d_y(rows);
        const int wrong = cols + rng.randInt(1, 5);
        EXPECT(spmv_csr(csr.m, d_x.get(), d_y.get(), nullptr, wrong).error_code == static_cast<int>(SpMVError::INVALID_DIMENSION));
        EXPECT(spmv_csr(csr.m, d_x.get(), d_y.get(), nullptr, cols).error_code == kOk);
        EXPECT(spmv_csr(csr.m, d_x.get(), d_y.get(), nullptr).error_code == kOk);     // -1: no check
        EXPECT(spmv_validate_dimensions(cols, cols) && !spmv_validate_dimensions(cols, wrong));
    }
    EXPECT(spmv_csr(nullptr, nullptr, nullptr, nullptr).error_code == static_cast<int>(SpMVError::INVALID_ARGUMENT));
}
CASE(SpMVUnitTest, EmptyMatrix) {                  // :148-159
    Csr csr;
    CudaBuffer<float> d_x(1), d_y(1);
    EXPECT(spmv_csr(csr.m, d_x.get(), d_y.get(), nullptr, 1).error_code == kOk);      // a 0-row matrix is a no-op (D2)
}
CASE(SpMVUnitTest, SingleElement) {                // :161-186
    const float dense[1] = {5.0f};
    Csr csr;
    csr_from_dense(csr.m, dense, 1, 1);
    csr_to_gpu(csr.m);
    for (auto kernel : {SpMVConfig::SCALAR_CSR, SpMVConfig::VECTOR_CSR, SpMVConfig::MERGE_PATH}) {
        EXPECT(run_csr(csr.m, {2.0f}, kernel)[0] == 10.0f);
    }
}
CASE(SpMVUnitTest, ZeroRows) {                     // :188-218
    const std::vector<float> dense = {1, 2, 0, 0, 0, 0, 3, 0, 4};
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 3, 3);
    csr_to_gpu(csr.m);
    for (auto kernel : {SpMVConfig::SCALAR_CSR, SpMVConfig::VECTOR_CSR, SpMVConfig::MERGE_PATH}) {
        const auto y = run_csr(csr.m, {1.0f, 1.0f, 1.0f}, kernel);
        EXPECT(y[0] == 3.0f && y[1] == 0.0f && y[2] == 7.0f);
    }
}
CASE(SpMVUnitTest, KernelSelector) {               // :220-237
    const std::vector<float> dense = {1, 0, 2, 0, 3, 4, 0, 0, 5};                     // the README example
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 3, 3);
    csr_to_gpu(csr.m);
    const SpMVConfig c = spmv_auto_config(csr.m);
    EXPECT(c.kernel_type == SpMVConfig::SCALAR_CSR && c.block_size == 256 && !c.use_texture);
    const auto y = run_csr(csr.m, {1.0f, 2.0f, 3.0f}, c.kernel_type);
    EXPECT(y[0] == 7.0f && y[1] == 18.0f && y[2] == 15.0f);
}

// ----------------------------------------------- tests/test_bandwidth.cu ----
CASE(BandwidthPropertyTest, MetricsValidity) {     // reference tests/test_bandwidth.cu:19-56
    RandomGenerator rng(501);
    for (int round = 0; round < kRounds; ++round) {
        const int rows = rng.randInt(10, 200), cols = rng.randInt(10, 200);
        const auto dense = generateRandomDenseMatrix(rows, cols, rng.randFloat(0.05f, 0.3f), rng);
        Csr csr;
        csr_from_dense(csr.m, dense.data(), rows, cols);
        csr_to_gpu(csr.m);
        CudaBuffer<float> d_x(cols), d_y(rows);
        const SpMVResult r = spmv_csr(csr.m, d_x.get(), d_y.get(), nullptr, cols);
        EXPECT(r.error_code == kOk && r.bandwidth_gb_s >= 0.0f && r.gflops >= 0.0f && r.y == d_y.get());
        const BandwidthMetrics bw = compute_bandwidth_csr(csr.m, r.elapsed_ms);
        EXPECT(bw.theoretical_bandwidth_gb_s > 0.0f && bw.efficiency >= 0.0f && bw.efficiency <= 1.0f);
    }
}
CASE(BandwidthUnitTest, PeakBandwidth) {           // :60-64
    EXPECT(get_gpu_peak_bandwidth() > 0.0f && get_gpu_peak_bandwidth() < 10000.0f);
}
CASE(BandwidthUnitTest, CSRBandwidthCalculation) { // :66-81
    const std::vector<float> dense = {1, 0, 2, 0, 3, 4, 0, 0, 5};
    Csr csr;
    csr_from_dense(csr.m, dense.data(), 3, 3);
    const BandwidthMetrics bw = compute_bandwidth_csr(csr.m, 1.0f);
    EXPECT(bw.achieved_bandwidth_gb_s > 0.0f && bw.theoretical_bandwidth_gb_s > 0.0f);
    EXPECT(bw.efficiency >= 0.0f && bw.efficiency <= 1.0f);
    // the byte model (src/bandwidth.cpp:34-42): nnz * 8 + (rows + 1) * 4 + cols * 4 + rows * 4, in 1 ms
    const double bytes = 5 * 8 + 4 * 4 + 3 * 4 + 3 * 4;
    EXPECT(std::fabs(bw.achieved_bandwidth_gb_s - bytes / 1e9 / 1e-3) <= 1e-9);
}
CASE(BandwidthUnitTest, ELLBandwidthCalculation) { // :83-98
    const std::vector<float> dense = {1, 0, 2, 0, 3, 4, 0, 0, 5};
    Ell ell;
    ell_from_dense(ell.m, dense.data(), 3, 3);
    const BandwidthMetrics bw = compute_bandwidth_ell(ell.m, 1.0f);
    EXPECT(bw.achieved_bandwidth_gb_s > 0.0f && bw.theoretical_bandwidth_gb_s > 0.0f);
    EXPECT(bw.efficiency >= 0.0f && bw.efficiency <= 1.0f);
    const double bytes = 3 * 2 * 8 + 3 * 4 + 3 * 4;                                   // rows * K * 8 + cols * 4 + rows * 4
    EXPECT(std::fabs(bw.achieved_bandwidth_gb_s - bytes / 1e9 / 1e-3) <= 1e-9);
}
CASE(BandwidthUnitTest, ZeroElapsedTime) {         // :100-113
    const float dense[3] = {1, 0, 2};
    Csr csr;
    csr_from_dense(csr.m, dense, 1, 3);
    const BandwidthMetrics bw = compute_bandwidth_csr(csr.m, 0.0f);
    EXPECT(bw.achieved_bandwidth_gb_s == 0.0f && bw.efficiency == 0.0f);
}

// ----------------------------------------------- tests/test_pagerank.cu ----
namespace {
// column-stochastic adjacency of a random directed graph: a_ij = 1 / outdeg(j) for an edge j -> i
void random_graph(CSRMatrix* csr, int n, float density, RandomGenerator& rng) {
    std::vector<float> dense(static_cast<size_t>(n) * n, 0.0f);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) dense[static_cast<size_t>(i) * n + j] = (i != j && rng.randBool(density)) ? 1.0f : 0.0f;
    }
    for (int j = 0; j < n; ++j) {
        int out = 0;
        for (int i = 0; i < n; ++i) out += dense[static_cast<size_t>(i) * n + j] != 0.0f;
        for (int i = 0; i < n && out > 0; ++i) dense[static_cast<size_t>(i) * n + j] /= static_cast<float>(out);
    }
    csr_from_dense(csr, dense.data(), n, n);
    csr_to_gpu(csr);
}
} // namespace
CASE(PageRankPropertyTest, ScoreInvariants) {      // reference tests/test_pagerank.cu:18-77
    RandomGenerator rng(601);
    for (int round = 0; round < kRounds; ++round) {
        const int n = rng.randInt(5, 80);
        Csr graph;
        random_graph(graph.m, n, rng.randFloat(0.05f, 0.4f), rng);
        PageRankResult r = pagerank(graph.m);
        EXPECT(r.ranks != nullptr && r.iterations >= 1 && r.iterations <= 100);
        double total = 0.0;
        bool non_negative = true;
        for (int i = 0; i < n; ++i) {
            total += r.ranks[i];
            non_negative = non_negative && r.ranks[i] >= 0.0f;
        }
        EXPECT(non_negative && std::fabs(total - 1.0) < 1e-4);
        if (r.converged) EXPECT(r.final_residual < 1e-6f);
        pagerank_free(&r);
        EXPECT(r.ranks == nullptr);
    }
}
CASE(PageRankPropertyTest, TopKOrdering) {         // :81-136
    RandomGenerator rng(602);
    for (int round = 0; round < kRounds; ++round) {
        const int n = rng.randInt(10, 80), k = rng.randInt(1, 10);
        Csr graph;
        random_graph(graph.m, n, 0.2f, rng);
        PageRankResult r = pagerank(graph.m);
        std::vector<TopKNode> top(k);
        pagerank_top_k(&r, n, k, top.data());
        for (int i = 1; i < k; ++i) EXPECT(top[i - 1].rank >= top[i].rank);
        EXPECT(top[0].rank == *std::max_element(r.ranks, r.ranks + n));
        for (int i = 0; i < k; ++i) EXPECT(top[i].node_id >= 0 && top[i].node_id < n && r.ranks[top[i].node_id] == top[i].rank);
        pagerank_free(&r);
    }
}
CASE(PageRankUnitTest, SimpleGraph) {              // :140-164 — 0 -> 1 -> 2 -> 0
    const std::vector<float> dense = {0, 0, 1, 1, 0, 0, 0, 1, 0};
    Csr graph;
    csr_from_dense(graph.m, dense.data(), 3, 3);
    csr_to_gpu(graph.m);
    PageRankResult r = pagerank(graph.m);
    EXPECT(r.converged);
    for (int i = 0; i < 3; ++i) EXPECT(std::fabs(r.ranks[i] - 1.0f / 3.0f) < 1e-4f);
    pagerank_free(&r);
    const PageRankResult none = pagerank(nullptr);
    EXPECT(none.ranks == nullptr && none.iterations == 0 && !none.converged);
}
CASE(PageRankUnitTest, TopKExtraction) {           // :166-189
    float ranks[5] = {0.1f, 0.4f, 0.05f, 0.3f, 0.15f};
    PageRankResult r;
    r.ranks = ranks;
    TopKNode top[3];
    pagerank_top_k(&r, 5, 3, top);
    EXPECT(top[0].node_id == 1 && top[1].node_id == 3 && top[2].node_id == 4);
    EXPECT(top[0].rank == 0.4f && top[1].rank == 0.3f && top[2].rank == 0.15f);
    r.ranks = nullptr;                              // not ours to free
}

// ---------------------------------------------- tests/test_benchmark.cu ----
namespace {
struct BenchFixture {
    Csr csr;
    CudaBuffer<float> d_x;
    BenchFixture(int rows, int cols, float density, unsigned seed) : d_x(cols) {
        RandomGenerator rng(seed);
        const auto dense = generateRandomDenseMatrix(rows, cols, density, rng);
        const auto x = generateRandomVector(cols, rng);
        csr_from_dense(csr.m, dense.data(), rows, cols);
        csr_to_gpu(csr.m);
        d_x.copyFromHost(x.data(), cols);
    }
};
BenchmarkConfig quick(int warmup, int runs) {
    BenchmarkConfig c;
    c.num_warmup_runs = warmup;
    c.num_runs = runs;
    return c;
}
} // namespace
CASE(BenchmarkPropertyTest, MetricsCompleteness) { // reference tests/test_benchmark.cu:17-61
    for (unsigned seed = 1; seed <= 5; ++seed) {
        BenchFixture f(150 + 10 * seed, 170, 0.1f, 700 + seed);
        const BenchmarkConfig bc = quick(1, 4);
        for (auto kernel : {SpMVConfig::SCALAR_CSR, SpMVConfig::VECTOR_CSR, SpMVConfig::MERGE_PATH}) {
            SpMVConfig config;
            config.kernel_type = kernel;
            const BenchmarkResult b = benchmark_csr(f.csr.m, f.d_x.get(), &config, &bc);
            EXPECT(b.num_runs == 4 && !b.name.empty());
            EXPECT(b.avg_time_ms > 0.0f && b.min_time_ms > 0.0f && b.min_time_ms <= b.avg_time_ms && b.avg_time_ms <= b.max_time_ms);
            EXPECT(b.stddev_time_ms >= 0.0f && b.gflops > 0.0f && b.bandwidth_gb_s > 0.0f);
        }
    }
}
CASE(BenchmarkPropertyTest, JSONRoundTrip) {       // :65-102
    BenchFixture f(120, 140, 0.15f, 710);
    const BenchmarkConfig bc = quick(1, 3);
    SpMVConfig config;
    config.kernel_type = SpMVConfig::VECTOR_CSR;
    const BenchmarkResult b = benchmark_csr(f.csr.m, f.d_x.get(), &config, &bc);
    const BenchmarkResult back = benchmark_from_json(benchmark_to_json(b));
    EXPECT(back.name == b.name && back.num_runs == b.num_runs);
    EXPECT(back.avg_time_ms == b.avg_time_ms && back.min_time_ms == b.min_time_ms && back.max_time_ms == b.max_time_ms);
    EXPECT(back.stddev_time_ms == b.stddev_time_ms && back.gflops == b.gflops && back.bandwidth_gb_s == b.bandwidth_gb_s);
}
CASE(BenchmarkUnitTest, BasicBenchmark) {          // :106-125
    BenchFixture f(100, 100, 0.1f, 720);
    const BenchmarkResult b = benchmark_csr(f.csr.m, f.d_x.get(), nullptr);                 // defaults: 5 + 20 runs
    EXPECT(b.num_runs == 20 && b.avg_time_ms > 0.0f && b.gflops > 0.0f);
    Ell ell;
    ell_from_csr(ell.m, f.csr.m);
    ell_to_gpu(ell.m);
    const BenchmarkConfig bc = quick(1, 3);
    const BenchmarkResult e = benchmark_ell(ell.m, f.d_x.get(), &bc);
    EXPECT(e.num_runs == 3 && e.avg_time_ms > 0.0f && e.gflops > 0.0f);
}
CASE(BenchmarkUnitTest, GPUvsCPUComparison) {      // :127-149
    BenchFixture f(200, 200, 0.1f, 730);
    const BenchmarkConfig bc = quick(1, 3);
    SpMVConfig config;
    config.kernel_type = SpMVConfig::VECTOR_CSR;
    const ComparisonResult c = compare_gpu_cpu_csr(f.csr.m, f.d_x.get(), &config, &bc);
    EXPECT(c.gpu_result.avg_time_ms > 0.0f && c.cpu_result.avg_time_ms > 0.0f && c.speedup > 0.0f);
    EXPECT(std::fabs(c.speedup - c.cpu_result.avg_time_ms / c.gpu_result.avg_time_ms) <= 1e-3f * c.speedup);
}
CASE(BenchmarkUnitTest, JSONFormat) {              // :151-170
    BenchmarkResult b;
    b.name = "unit";
    b.avg_time_ms = 1.5f;
    b.num_runs = 3;
    const std::string json = benchmark_to_json(b);
    for (const char* key : {"\"name\"", "\"avg_time_ms\"", "\"min_time_ms\"", "\"max_time_ms\"", "\"stddev_time_ms\"",
                            "\"gflops\"", "\"bandwidth_gb_s\"", "\"num_runs\""}) {
        EXPECT(json.find(key) != std::string::npos);
    }
    EXPECT(json.front() == '{' && json.back() == '}');
    ComparisonResult c;
    c.speedup = 2.0f;
    const std::string cj = comparison_to_json(c);
    EXPECT(cj.find("\"speedup\"") != std::string::npos && cj.find("\"gpu\"") != std::string::npos && cj.find("\"cpu\"") != std::string::npos);
}

// ------------------------------------------------------------------ main ----
int main(int argc, char** argv) {
    const std::string only = argc > 1 ? argv[1] : "";
    int ran = 0, failed_cases = 0;
    for (const Case& c : registry()) {
        if (!only.empty() && std::string(c.name).find(only) == std::string::npos) continue;
        const int before = g_failed_checks;
        std::printf("[ RUN  ] %s\n", c.name);
        try {
            c.body();
        } catch (const std::exception& e) {
            ++g_failed_checks;
            std::printf("    EXCEPTION %s\n", e.what());
        }
        const bool ok = g_failed_checks == before;
        std::printf("[ %s ] %s\n", ok ? " OK " : "FAIL", c.name);
        ++ran;
        failed_cases += !ok;
    }
    std::printf("%d cases run, %d failed\n", ran, failed_cases);
    if (failed_cases == 0 && ran > 0) std::printf("all reference cases passed\n");
    return failed_cases == 0 && ran > 0 ? 0 : 1;
}
