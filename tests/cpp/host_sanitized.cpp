// host_sanitized.cpp — the library's host-side code under AddressSanitizer + UBSan ("sanitizer
// build for the CPU tier").  The host-only sources (containers, file formats, CPU SpMV, byte model, selector)
// are compiled INTO this executable with -fsanitize=address,undefined and take precedence over the copies in
// libspmv_amd.so, which only supplies what they call into (side tables, launch layer).  No GPU is touched:
// nothing here uploads a matrix or launches a kernel.  Run by tests/test_host_sanitized.py.
#include "spmv/spmv.h"
#include "spmv/bandwidth.h"
#include "spmv/pagerank.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace spmv;

static int g_failed = 0;
#define EXPECT(cond)                                                           \
    do {                                                                      \
        if (!(cond)) {                                                        \
            ++g_failed;                                                       \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);     \
        }                                                                     \
    } while (0)

static std::vector<float> random_dense(int rows, int cols, double density, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> value(-2.0f, 2.0f);
    std::uniform_real_distribution<double> coin(0.0, 1.0);
    std::vector<float> dense(static_cast<size_t>(rows) * cols, 0.0f);
    for (float& v : dense) if (coin(rng) < density) v = value(rng);
    return dense;
}

static void dense_product(const std::vector<float>& dense, int rows, int cols, const std::vector<float>& x,
                          std::vector<float>* y) {
    y->assign(rows, 0.0f);
    for (int r = 0; r < rows; ++r) {
        float acc = 0.0f;
        for (int c = 0; c < cols; ++c) {
            const float a = dense[static_cast<size_t>(r) * cols + c];
            if (a != 0.0f) acc += a * x[c];
        }
        (*y)[r] = acc;
    }
}

static void containers_and_cpu_spmv(const std::string& dir) {
    const int shapes[][2] = {{1, 1}, {7, 13}, {64, 64}, {33, 5}, {5, 200}};
    unsigned seed = 1;
    for (const auto& shape : shapes) {
        const int rows = shape[0], cols = shape[1];
        const std::vector<float> dense = random_dense(rows, cols, 0.3, seed++);
        CSRMatrix* csr = csr_create(0, 0, 0);
        EXPECT(csr != nullptr);
        EXPECT(csr_from_dense(csr, dense.data(), rows, cols) == 0);
        std::vector<float> back(dense.size(), -1.0f);
        EXPECT(csr_to_dense(csr, back.data()) == 0);
        EXPECT(back == dense);
        for (int r = 0; r < rows; ++r) {
            for (int c = 0; c < cols; ++c) EXPECT(csr_get_element(csr, r, c) == dense[static_cast<size_t>(r) * cols + c]);
        }
        EXPECT(csr_get_element(csr, -1, 0) == 0.0f && csr_get_element(csr, rows, 0) == 0.0f && csr_get_element(csr, 0, cols) == 0.0f);
        const CSRStats stats = csr_compute_stats(csr);
        EXPECT(stats.max_nnz_per_row >= stats.min_nnz_per_row || rows == 0);
        const SpMVConfig config = spmv_auto_config(csr);
        EXPECT(config.block_size == 256);
        EXPECT(spmv_validate_dimensions(cols, cols) && (cols == 0 || !spmv_validate_dimensions(cols, cols - 1)));

        std::vector<float> x(cols), want, got(rows, -1.0f);
        for (int c = 0; c < cols; ++c) x[c] = 0.25f * static_cast<float>((c * 7) % 11) - 1.0f;
        dense_product(dense, rows, cols, x, &want);
        spmv_cpu_csr(csr, x.data(), got.data());
        EXPECT(got == want);                                    // same order, same rounding

        ELLMatrix* ell = ell_create(0, 0, 0);
        EXPECT(ell != nullptr);
        EXPECT(ell_from_csr(ell, csr) == 0);
        std::vector<float> ell_back(dense.size(), -1.0f);
        EXPECT(ell_to_dense(ell, ell_back.data()) == 0);
        EXPECT(ell_back == dense);
        std::vector<float> ell_y(rows, -1.0f);
        spmv_cpu_ell(ell, x.data(), ell_y.data());
        EXPECT(ell_y == want);
        ELLMatrix* ell2 = ell_create(0, 0, 0);
        EXPECT(ell_from_dense(ell2, dense.data(), rows, cols) == 0);
        EXPECT(ell2->max_nnz_per_row == ell->max_nnz_per_row);
        for (int r = 0; r < rows && r < 4; ++r) {
            for (int c = 0; c < cols; ++c) EXPECT(ell_get_element(ell2, r, c) == dense[static_cast<size_t>(r) * cols + c]);
        }

        // file formats: round trip, then every truncation of the file must be refused without reading past it
        const std::string csr_file = dir + "/m.csr", ell_file = dir + "/m.ell";
        EXPECT(csr_serialize(csr, csr_file.c_str()) == 0);
        CSRMatrix* loaded = csr_create(0, 0, 0);
        EXPECT(csr_deserialize(loaded, csr_file.c_str()) == 0);
        EXPECT(loaded->num_rows == rows && loaded->num_cols == cols && loaded->nnz == csr->nnz);
        std::vector<float> loaded_dense(dense.size(), -1.0f);
        EXPECT(csr_to_dense(loaded, loaded_dense.data()) == 0 && loaded_dense == dense);
        EXPECT(ell_serialize(ell, ell_file.c_str()) == 0);
        ELLMatrix* loaded_ell = ell_create(0, 0, 0);
        EXPECT(ell_deserialize(loaded_ell, ell_file.c_str()) == 0);
        std::vector<float> loaded_ell_dense(dense.size(), -1.0f);
        EXPECT(ell_to_dense(loaded_ell, loaded_ell_dense.data()) == 0 && loaded_ell_dense == dense);
        for (const std::string& file : {csr_file, ell_file}) {
            std::FILE* f = std::fopen(file.c_str(), "rb");
            EXPECT(f != nullptr);
            if (!f) continue;
            std::vector<unsigned char> bytes(1 << 20);
            bytes.resize(std::fread(bytes.data(), 1, bytes.size(), f));
            std::fclose(f);
            const std::string cut_file = file + ".cut";
            for (size_t keep = 0; keep < bytes.size(); keep += (bytes.size() > 256 ? 37 : 1)) {
                std::FILE* out = std::fopen(cut_file.c_str(), "wb");
                std::fwrite(bytes.data(), 1, keep, out);
                std::fclose(out);
                if (file == csr_file) {
                    CSRMatrix* victim = csr_create(0, 0, 0);
                    EXPECT(csr_deserialize(victim, cut_file.c_str()) != 0);
                    csr_destroy(victim);
                } else {
                    ELLMatrix* victim = ell_create(0, 0, 0);
                    EXPECT(ell_deserialize(victim, cut_file.c_str()) != 0);
                    ell_destroy(victim);
                }
            }
            // a header that promises more than the file holds / nonsense sizes
            if (bytes.size() >= 16) {
                std::vector<unsigned char> lying = bytes;
                const int huge = 0x7fffff00;
                std::memcpy(lying.data() + 8, &huge, sizeof(huge));
                std::FILE* out = std::fopen(cut_file.c_str(), "wb");
                std::fwrite(lying.data(), 1, lying.size(), out);
                std::fclose(out);
                if (file == csr_file) {
                    CSRMatrix* victim = csr_create(0, 0, 0);
                    EXPECT(csr_deserialize(victim, cut_file.c_str()) != 0);   // refused before anything is allocated
                    csr_destroy(victim);
                } else {
                    ELLMatrix* victim = ell_create(0, 0, 0);
                    EXPECT(ell_deserialize(victim, cut_file.c_str()) != 0);
                    ell_destroy(victim);
                }
            }
            // fuzz: a few random bytes changed.  Either the loader refuses the file, or what it hands out is
            // a matrix every later loop can index blindly — proven by running the CPU SpMV on it under ASan.
            {
                std::mt19937 fuzz(1234u + static_cast<unsigned>(bytes.size()));
                for (int round = 0; round < 400 && !bytes.empty(); ++round) {
                    std::vector<unsigned char> mutated = bytes;
                    const int flips = 1 + static_cast<int>(fuzz() % 4);
                    for (int k = 0; k < flips; ++k) mutated[fuzz() % mutated.size()] = static_cast<unsigned char>(fuzz());
                    std::FILE* out = std::fopen(cut_file.c_str(), "wb");
                    std::fwrite(mutated.data(), 1, mutated.size(), out);
                    std::fclose(out);
                    if (file == csr_file) {
                        CSRMatrix* victim = csr_create(0, 0, 0);
                        if (csr_deserialize(victim, cut_file.c_str()) == 0 && victim->num_cols <= (1 << 13) && victim->num_rows <= (1 << 13)) {   // (huge but empty is a valid matrix)
                            std::vector<float> vx(static_cast<size_t>(victim->num_cols) + 1, 1.0f), vy(static_cast<size_t>(victim->num_rows) + 1);
                            spmv_cpu_csr(victim, vx.data(), vy.data());
                            (void)csr_compute_stats(victim);
                        }
                        csr_destroy(victim);
                    } else {
                        ELLMatrix* victim = ell_create(0, 0, 0);
                        if (ell_deserialize(victim, cut_file.c_str()) == 0 && victim->num_cols <= (1 << 13) && victim->num_rows <= (1 << 13)) {
                            std::vector<float> vx(static_cast<size_t>(victim->num_cols) + 1, 1.0f), vy(static_cast<size_t>(victim->num_rows) + 1);
                            spmv_cpu_ell(victim, vx.data(), vy.data());
                        }
                        ell_destroy(victim);
                    }
                }
            }
            // arrays that do not describe a matrix: a column index out of range, a row pointer running backwards
            if (file == csr_file && csr->nnz > 0 && rows > 1) {
                const size_t col0 = 12 + 4 * static_cast<size_t>(csr->nnz), ptr1 = 12 + 8 * static_cast<size_t>(csr->nnz) + 4;
                for (int which = 0; which < 2; ++which) {
                    std::vector<unsigned char> broken = bytes;
                    const int bad = which == 0 ? cols : -5;
                    std::memcpy(broken.data() + (which == 0 ? col0 : ptr1), &bad, sizeof(bad));
                    std::FILE* out = std::fopen(cut_file.c_str(), "wb");
                    std::fwrite(broken.data(), 1, broken.size(), out);
                    std::fclose(out);
                    CSRMatrix* victim = csr_create(0, 0, 0);
                    EXPECT(csr_deserialize(victim, cut_file.c_str()) != 0);
                    EXPECT(victim->nnz == 0 && victim->num_rows == 0);
                    csr_destroy(victim);
                }
            }
            std::remove(cut_file.c_str());
        }
        EXPECT(csr_deserialize(loaded, (dir + "/does-not-exist").c_str()) != 0);

        const BandwidthMetrics bw = compute_bandwidth_csr(csr, 1.0f);
        EXPECT(bw.achieved_bandwidth_gb_s >= 0.0f && bw.efficiency <= 1.0f);
        const BandwidthMetrics bw_ell = compute_bandwidth_ell(ell, 1.0f);
        EXPECT(bw_ell.achieved_bandwidth_gb_s >= 0.0f && bw_ell.efficiency <= 1.0f);

        csr_destroy(loaded);
        ell_destroy(loaded_ell);
        ell_destroy(ell2);
        ell_destroy(ell);
        csr_destroy(csr);
    }
    // argument errors come back as codes
    EXPECT(csr_create(-1, 2, 3) == nullptr);
    EXPECT(csr_from_dense(nullptr, nullptr, 1, 1) != 0);
    EXPECT(csr_to_dense(nullptr, nullptr) != 0);
    EXPECT(ell_from_csr(nullptr, nullptr) != 0);
    EXPECT(std::strlen(spmv_error_string(SpMVError::INVALID_DIMENSION)) > 0);
}

static void pagerank_host_helpers() {
    const int ptrs[] = {0, 10, 10, 11, 12, 40, 41};
    for (int shards : {1, 2, 3, 6, 9}) {
        const std::vector<int> b = pagerank_shard_bounds(ptrs, 6, shards);
        EXPECT(static_cast<int>(b.size()) == shards + 1 && b.front() == 0 && b.back() == 6);
        for (int p = 0; p < shards; ++p) EXPECT(b[p] <= b[p + 1]);
    }
    PageRankResult r;
    r.ranks = new float[5]{0.1f, 0.4f, 0.2f, 0.25f, 0.05f};
    TopKNode top[8];
    pagerank_top_k(&r, 5, 3, top);
    EXPECT(top[0].node_id == 1 && top[1].node_id == 3 && top[2].node_id == 2);
    pagerank_top_k(&r, 5, 8, top);                               // k > n: clamps
    EXPECT(top[4].node_id == 4);
    pagerank_free(&r);
    EXPECT(r.ranks == nullptr);
    pagerank_free(&r);                                           // idempotent
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    containers_and_cpu_spmv(dir);
    pagerank_host_helpers();
    if (g_failed) {
        std::printf("%d checks FAILED\n", g_failed);
        return 1;
    }
    std::printf("host code under ASan + UBSan: all checks passed\n");
    return 0;
}
