// spmv/pagerank.h — PageRank power iteration on top of the CSR SpMV.
//
// Same API as the reference (include/spmv/pagerank.h:9-43).  The loop itself
// is device-resident here: one fused kernel per iteration (SpMV + damping +
// teleport + residual and dangling-mass partial sums), no PCIe copies inside
// the loop.  Row-sharded multi-GPU operation goes through the shard API below.
#ifndef SPMV_PAGERANK_H
#define SPMV_PAGERANK_H

#include "csr_matrix.h"

#include <vector>

namespace spmv {

struct PageRankConfig {
    float damping_factor;
    float tolerance;
    int   max_iterations;

    PageRankConfig() : damping_factor(0.85f), tolerance(1e-6f), max_iterations(100) {}
};

struct PageRankResult {
    float* ranks;            // new float[num_nodes]; release with pagerank_free
    int    iterations;
    float  final_residual;   // L2 norm of the last update
    bool   converged;

    PageRankResult() : ranks(nullptr), iterations(0), final_residual(0.0f), converged(false) {}
};

// adj_matrix: column-normalised adjacency in CSR (row i = in-links of node i).
// Extension: with SPMV_NUM_GPUS=N (N > 1) in the environment and host arrays present, the call runs
// pagerank_multi_gpu(adj_matrix, config, N) instead; unset, behaviour is the reference's single-device one.
PageRankResult pagerank(const CSRMatrix* adj_matrix, const PageRankConfig* config = nullptr);

// Extension (the reference is single-GPU): the same iteration with the CSR rows sharded
// over `num_gpus` devices of this process — contiguous row blocks cut for equal nnz (binary search on
// row_ptrs), every device a full-length rank vector, ONE RCCL all-gather of the new slices per iteration
// (single-process ncclCommInitAll, one stream per device).  Shards are cut from the HOST arrays; the matrix
// need not be resident on any device.  Same result contract as pagerank(): ranks released by pagerank_free;
// an empty result (ranks == nullptr) when fewer than num_gpus devices or no RCCL are available.
PageRankResult pagerank_multi_gpu(const CSRMatrix* adj_matrix, const PageRankConfig* config, int num_gpus);
// the row boundaries it cuts (num_shards + 1 ascending row indices, bounds[0] = 0, bounds[num_shards] = num_rows)
std::vector<int> pagerank_shard_bounds(const int* row_ptrs, int num_rows, int num_shards);

void pagerank_free(PageRankResult* result);

struct TopKNode {
    int   node_id;
    float rank;
};

void pagerank_top_k(const PageRankResult* result, int num_nodes, int k, TopKNode* top_k);

} // namespace spmv

#endif // SPMV_PAGERANK_H
