// ref_driver.cpp — drives the REFERENCE's own CPU path (spmv_cpu.cpp,
// csr_matrix.cpp, ell_matrix.cpp compiled where they lie under /root/reference;
// see oracle/Makefile) on inputs handed over in a file, and dumps every output
// as named records.  Test infrastructure: it validates oracle/spmv_oracle.c and
// produces tests/golden/ref_*.npz (tests/golden/make_golden.py).  This file is
// our code; it only *calls* the reference through its public headers.
//
// usage:
//   ref_cpu case  <in.bin> <out.bin>   in : int32 rows, cols; float dense[rows*cols]; float x[cols]
//   ref_cpu probe                       prints the SURVEY.md §8(c) probe (1000x1000, density 0.008, seed 42)
//   ref_cpu time  <in.bin> <reps>       in : int32 rows, cols, nnz; int32 row_ptrs[rows+1]; int32 cols[nnz];
//                                            float vals[nnz]; float x[cols]  -> prints best seconds of spmv_cpu_csr
#include "spmv/spmv.h"
#include "spmv/test_utils.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace spmv;

namespace {

struct Writer {
    FILE* f;
    void record(const char* name, char kind, const void* data, long long count, size_t width) {
        int len = static_cast<int>(strlen(name));
        fwrite(&len, sizeof(int), 1, f);
        fwrite(name, 1, len, f);
        fwrite(&kind, 1, 1, f);
        fwrite(&count, sizeof(long long), 1, f);
        if (count > 0) fwrite(data, width, count, f);
    }
    void ints(const char* name, const int* d, long long n) { record(name, 'i', d, n, 4); }
    void floats(const char* name, const float* d, long long n) { record(name, 'f', d, n, 4); }
    void bytes(const char* name, const unsigned char* d, long long n) { record(name, 'b', d, n, 1); }
};

std::vector<unsigned char> slurp(const char* path) {
    std::vector<unsigned char> out;
    FILE* f = fopen(path, "rb");
    if (!f) return out;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n);
    if (n > 0 && fread(out.data(), 1, n, f) != static_cast<size_t>(n)) out.clear();
    fclose(f);
    return out;
}

int run_case(const char* in_path, const char* out_path) {
    FILE* in = fopen(in_path, "rb");
    if (!in) return 2;
    int shape[2];
    if (fread(shape, sizeof(int), 2, in) != 2) return 2;
    const int rows = shape[0], cols = shape[1];
    std::vector<float> dense(static_cast<size_t>(rows) * cols), x(cols);
    if (fread(dense.data(), sizeof(float), dense.size(), in) != dense.size()) return 2;
    if (fread(x.data(), sizeof(float), x.size(), in) != x.size()) return 2;
    fclose(in);

    FILE* out = fopen(out_path, "wb");
    if (!out) return 2;
    Writer w{out};

    // CSR from dense + host SpMV
    CSRMatrix* csr = csr_create(0, 0, 0);
    int status = csr_from_dense(csr, dense.data(), rows, cols);
    w.ints("csr_from_dense_status", &status, 1);
    const int shape3[3] = {csr->num_rows, csr->num_cols, csr->nnz};
    w.ints("csr_shape", shape3, 3);
    w.ints("csr_row_ptrs", csr->row_ptrs, csr->num_rows + 1);
    w.ints("csr_col_indices", csr->col_indices, csr->nnz);
    w.floats("csr_values", csr->values, csr->nnz);

    std::vector<float> y(rows, -777.0f);
    spmv_cpu_csr(csr, x.data(), y.data());
    w.floats("y_csr", y.data(), rows);

    // round trip + element queries
    std::vector<float> back(dense.size(), -1.0f);
    csr_to_dense(csr, back.data());
    w.floats("csr_to_dense", back.data(), static_cast<long long>(back.size()));
    std::vector<float> diag;
    for (int i = 0; i < rows && i < cols; ++i) diag.push_back(csr_get_element(csr, i, i));
    w.floats("csr_get_diag", diag.data(), static_cast<long long>(diag.size()));

    // statistics + selector
    const CSRStats st = csr_compute_stats(csr);
    const float stats4[4] = {st.avg_nnz_per_row, static_cast<float>(st.max_nnz_per_row),
                             static_cast<float>(st.min_nnz_per_row), st.skewness};
    w.floats("csr_stats", stats4, 4);
    const SpMVConfig cfg = spmv_auto_config(csr);
    const int cfg3[3] = {static_cast<int>(cfg.kernel_type), cfg.block_size, cfg.use_texture ? 1 : 0};
    w.ints("auto_config", cfg3, 3);

    // on-disk format
    const std::string tmp = std::string(out_path) + ".csr";
    status = csr_serialize(csr, tmp.c_str());
    w.ints("csr_serialize_status", &status, 1);
    const std::vector<unsigned char> file = slurp(tmp.c_str());
    w.bytes("csr_file", file.data(), static_cast<long long>(file.size()));
    remove(tmp.c_str());

    // ELL both ways + host SpMV
    ELLMatrix* ell = ell_create(0, 0, 0);
    status = ell_from_csr(ell, csr);
    w.ints("ell_from_csr_status", &status, 1);
    const int eshape[3] = {ell->num_rows, ell->num_cols, ell->max_nnz_per_row};
    w.ints("ell_shape", eshape, 3);
    const long long slots = static_cast<long long>(ell->num_rows) * ell->max_nnz_per_row;
    w.ints("ell_col_indices", ell->col_indices, slots);
    w.floats("ell_values", ell->values, slots);
    std::vector<float> y2(rows, -777.0f);
    spmv_cpu_ell(ell, x.data(), y2.data());
    w.floats("y_ell", y2.data(), rows);

    ELLMatrix* ell2 = ell_create(0, 0, 0);
    ell_from_dense(ell2, dense.data(), rows, cols);
    const long long slots2 = static_cast<long long>(ell2->num_rows) * ell2->max_nnz_per_row;
    w.ints("ell_dense_col_indices", ell2->col_indices, slots2);
    w.floats("ell_dense_values", ell2->values, slots2);

    const std::string tmp2 = std::string(out_path) + ".ell";
    status = ell_serialize(ell, tmp2.c_str());
    const std::vector<unsigned char> efile = slurp(tmp2.c_str());
    w.bytes("ell_file", efile.data(), static_cast<long long>(efile.size()));
    remove(tmp2.c_str());

    fclose(out);
    return 0;   // matrices leak on purpose: *_destroy would pull in the device-free path
}

int run_probe() {
    test::RandomGenerator rng(42);
    auto dense = test::generateRandomDenseMatrix(1000, 1000, 0.008f, rng);
    auto x = test::generateRandomVector(1000, rng);
    CSRMatrix* csr = csr_create(0, 0, 0);
    csr_from_dense(csr, dense.data(), 1000, 1000);
    std::vector<float> y(1000);
    spmv_cpu_csr(csr, x.data(), y.data());
    printf("nnz %d y0 %.9g y999 %.9g\n", csr->nnz, y[0], y[999]);
    return 0;
}

int run_time(const char* in_path, int reps) {
    FILE* in = fopen(in_path, "rb");
    if (!in) return 2;
    int shape[3];
    if (fread(shape, sizeof(int), 3, in) != 3) return 2;
    CSRMatrix* csr = csr_create(shape[0], shape[1], shape[2]);
    std::vector<float> x(shape[1]), y(shape[0]);
    bool ok = fread(csr->row_ptrs, sizeof(int), shape[0] + 1, in) == static_cast<size_t>(shape[0] + 1)
           && fread(csr->col_indices, sizeof(int), shape[2], in) == static_cast<size_t>(shape[2])
           && fread(csr->values, sizeof(float), shape[2], in) == static_cast<size_t>(shape[2])
           && fread(x.data(), sizeof(float), x.size(), in) == x.size();
    fclose(in);
    if (!ok) return 2;
    double best = 1e30;
    for (int r = 0; r < reps + 1; ++r) {   // first pass is the warm-up
        const auto t0 = std::chrono::steady_clock::now();
        spmv_cpu_csr(csr, x.data(), y.data());
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        if (r > 0 && s < best) best = s;
    }
    double checksum = 0.0;
    for (float v : y) checksum += v;
    printf("best_seconds %.9g checksum %.9g\n", best, checksum);
    return 0;
}

} // namespace

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "case")) return run_case(argv[2], argv[3]);
    if (argc >= 2 && !strcmp(argv[1], "probe")) return run_probe();
    if (argc >= 4 && !strcmp(argv[1], "time")) return run_time(argv[2], atoi(argv[3]));
    fprintf(stderr, "usage: ref_cpu case <in> <out> | probe | time <in> <reps>\n");
    return 1;
}
