// multi_gpu_pagerank.cpp — C++ caller of the native multi-GPU entry point
// PageRankResult pagerank_multi_gpu(const CSRMatrix*, const PageRankConfig*, int num_gpus)
// (extension of the reference's include/spmv/pagerank.h:29-43; the reference itself is single-GPU).
//
//   multi_gpu_pagerank bounds         CPU only: the equal-nnz row boundaries
//   multi_gpu_pagerank run [N]        needs N (default 1) GPUs: pagerank_multi_gpu(N) == pagerank() on a uniform
//                                     and on a power-law graph, plain and through SPMV_NUM_GPUS; with N == 1 the
//                                     collective path is exercised too (SPMV_MULTI_GPU=force_rccl), and the P > 1
//                                     partition / layout / commit with 2, 3 and 8 shards sharing the device
//                                     (SPMV_MULTI_GPU=share_devices: slices exchanged by device copies), also with
//                                     the overlapped exchange (SPMV_MULTI_GPU=blocks=C) on a graph large enough for
//                                     the tiled engine, whose phase 1 then follows the blocks
// Plain host C++: g++ -Iinclude ... -lspmv_amd.
#include "spmv/pagerank.h"
#include "spmv/spmv.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace spmv;

static int g_failed = 0;
#define EXPECT(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            ++g_failed;                                                          \
            std::printf("    FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);    \
        }                                                                        \
    } while (0)

// column-stochastic graph in CSR: row lengths from `lens`, distinct random columns, a few columns left empty
static CSRMatrix* make_graph(const std::vector<int>& lens, unsigned seed) {
    const int n = static_cast<int>(lens.size());
    std::mt19937 rng(seed);
    long long nnz = 0;
    for (int l : lens) nnz += l;
    CSRMatrix* m = csr_create(n, n, static_cast<int>(nnz));
    std::vector<int> count(n, 0);
    int at = 0;
    for (int r = 0; r < n; ++r) {
        m->row_ptrs[r] = at;
        std::vector<int> cols;
        while (static_cast<int>(cols.size()) < lens[r]) {
            for (int missing = lens[r] - static_cast<int>(cols.size()); missing > 0; --missing) {
                const int c = static_cast<int>(rng() % n);
                if (c % 97 != 5) cols.push_back(c);         // dangling nodes: nobody links to them
            }
            std::sort(cols.begin(), cols.end());
            cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        }
        for (int c : cols) {
            m->col_indices[at++] = c;
            ++count[c];
        }
    }
    m->row_ptrs[n] = at;
    m->nnz = at;
    for (int j = 0; j < at; ++j) m->values[j] = 1.0f / static_cast<float>(count[m->col_indices[j]]);
    return m;
}

static double worst_relative(const PageRankResult& a, const PageRankResult& b, int n) {
    double worst = 0.0;
    for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(double(a.ranks[i]) - b.ranks[i]) / b.ranks[i]);
    return worst;
}

static void test_bounds() {
    // 6 rows with 10, 0, 1, 1, 28, 1 entries: two shards of ~20 -> the cut after row 3 (12 | 29) beats 40 | 1
    const int ptrs[] = {0, 10, 10, 11, 12, 40, 41};
    const std::vector<int> two = pagerank_shard_bounds(ptrs, 6, 2);
    EXPECT(two.size() == 3 && two[0] == 0 && two[1] == 4 && two[2] == 6);
    const std::vector<int> one = pagerank_shard_bounds(ptrs, 6, 1);
    EXPECT(one.size() == 2 && one[0] == 0 && one[1] == 6);
    const std::vector<int> many = pagerank_shard_bounds(ptrs, 6, 8);      // more shards than rows: empty ones allowed
    EXPECT(many.size() == 9 && many.front() == 0 && many.back() == 6 && std::is_sorted(many.begin(), many.end()));
    std::vector<int> uniform(1001);
    for (int i = 0; i <= 1000; ++i) uniform[i] = 16 * i;
    const std::vector<int> eight = pagerank_shard_bounds(uniform.data(), 1000, 8);
    for (int p = 0; p < 8; ++p) EXPECT(eight[p + 1] - eight[p] == 125);
}

// the overlapped exchange: blocks of the chunk-major vector, on the side stream, head starts of phase 1
static void test_blocks() {
    std::vector<int> lens(700000, 6);                       // 4.2 M entries: both halves run on the tiled engine
    for (int i = 0; i < 3000; ++i) lens[i * 200] = 3000;    // and some rows long enough for the direct path
    CSRMatrix* g = make_graph(lens, 99u);
    const int n = g->num_rows;
    EXPECT(csr_to_gpu(g) == 0);
    PageRankConfig cfg;
    cfg.max_iterations = 12;                                // fixed work: equal iteration counts by construction
    cfg.tolerance = 0.0f;
    PageRankResult single = pagerank(g, &cfg);
    EXPECT(single.ranks && single.iterations == 12);
    for (int blocks : {1, 3, 4}) {
        char text[48];
        std::snprintf(text, sizeof(text), "share_devices,blocks=%d", blocks);
        setenv("SPMV_MULTI_GPU", text, 1);
        for (int shards : {2, 3}) {
            PageRankResult r = pagerank_multi_gpu(g, &cfg, shards);
            EXPECT(r.ranks != nullptr && r.iterations == 12);
            if (r.ranks) EXPECT(worst_relative(r, single, n) <= 4e-6);
            pagerank_free(&r);
        }
    }
    setenv("SPMV_MULTI_GPU", "force_rccl,blocks=2", 1);     // RCCL with one rank, two blocks, side stream
    PageRankResult forced = pagerank_multi_gpu(g, &cfg, 1);
    EXPECT(forced.ranks != nullptr && forced.iterations == 12);
    if (forced.ranks) EXPECT(worst_relative(forced, single, n) <= 4e-6);
    pagerank_free(&forced);
    unsetenv("SPMV_MULTI_GPU");
    pagerank_free(&single);
    csr_destroy(g);
}

static void test_run(int gpus) {
    std::vector<int> uniform(20000, 8);
    std::vector<int> skewed(20000);
    for (int i = 0; i < 20000; ++i) skewed[i] = 2 + 4000 / (i + 1);       // power-law-ish: the first rows are long
    int which = 0;
    if (gpus == 1) test_blocks();
    for (const std::vector<int>& lens : {uniform, skewed}) {
        CSRMatrix* g = make_graph(lens, 7u + which++);
        const int n = g->num_rows;
        EXPECT(csr_to_gpu(g) == 0);
        const PageRankConfig cfg;
        PageRankResult single = pagerank(g, &cfg);
        EXPECT(single.ranks && single.converged);
        PageRankResult multi = pagerank_multi_gpu(g, &cfg, gpus);
        EXPECT(multi.ranks != nullptr);
        if (multi.ranks) {
            EXPECT(multi.converged == single.converged && std::abs(multi.iterations - single.iterations) <= 1);
            if (multi.iterations == single.iterations) EXPECT(worst_relative(multi, single, n) <= 2e-6);
            double sum = 0.0;
            for (int i = 0; i < n; ++i) sum += multi.ranks[i];
            EXPECT(std::fabs(sum - 1.0) < 1e-4);
        }
        if (gpus == 1) {                                    // the collective path on one device
            setenv("SPMV_MULTI_GPU", "force_rccl", 1);
            PageRankResult forced = pagerank_multi_gpu(g, &cfg, 1);
            unsetenv("SPMV_MULTI_GPU");
            EXPECT(forced.ranks != nullptr);
            if (forced.ranks && forced.iterations == single.iterations) EXPECT(worst_relative(forced, single, n) <= 2e-6);
            pagerank_free(&forced);
        } else {                                            // SPMV_NUM_GPUS routes pagerank() itself
            char text[16];
            std::snprintf(text, sizeof(text), "%d", gpus);
            setenv("SPMV_NUM_GPUS", text, 1);
            PageRankResult routed = pagerank(g, &cfg);
            unsetenv("SPMV_NUM_GPUS");
            EXPECT(routed.ranks && multi.ranks && routed.iterations == multi.iterations);
            if (routed.ranks && multi.ranks) EXPECT(worst_relative(routed, multi, n) <= 2e-6);
            pagerank_free(&routed);
        }
        if (gpus == 1) {                                    // P > 1 shards on the one device, exchange by copies
            setenv("SPMV_MULTI_GPU", "share_devices", 1);
            for (int shards : {2, 3, 8}) {
                PageRankResult shared = pagerank_multi_gpu(g, &cfg, shards);
                EXPECT(shared.ranks != nullptr);
                if (shared.ranks) {
                    EXPECT(shared.converged == single.converged && std::abs(shared.iterations - single.iterations) <= 1);
                    if (shared.iterations == single.iterations) EXPECT(worst_relative(shared, single, n) <= 2e-6);
                    double sum = 0.0;
                    for (int i = 0; i < n; ++i) sum += shared.ranks[i];
                    EXPECT(std::fabs(sum - 1.0) < 1e-4);
                }
                pagerank_free(&shared);
            }
            unsetenv("SPMV_MULTI_GPU");
        }
        // more devices than the machine has: an empty result, not a crash
        PageRankResult none = pagerank_multi_gpu(g, &cfg, 1024);
        EXPECT(none.ranks == nullptr && none.iterations == 0);
        pagerank_free(&multi);
        pagerank_free(&single);
        csr_destroy(g);
    }
}

int main(int argc, char** argv) {
    const bool bounds_only = argc > 1 && std::strcmp(argv[1], "bounds") == 0;
    test_bounds();
    if (!bounds_only) test_run(argc > 2 ? std::atoi(argv[2]) : 1);
    if (g_failed) {
        std::printf("%d checks FAILED\n", g_failed);
        return 1;
    }
    std::printf("multi-gpu pagerank: all checks passed (%s)\n", bounds_only ? "bounds only" : "bounds + run");
    return 0;
}
