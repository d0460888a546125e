// spmv/pagerank.h — PageRank power iteration on top of the CSR SpMV.
//
// Same API as the reference (include/spmv/pagerank.h:9-43).  The loop itself
// is device-resident here: one fused kernel per iteration (SpMV + damping +
// teleport + residual and dangling-mass partial sums), no PCIe copies inside
// the loop.  Row-sharded multi-GPU operation goes through the shard API below.
#ifndef SPMV_PAGERANK_H
#define SPMV_PAGERANK_H

#include "csr_matrix.h"

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
PageRankResult pagerank(const CSRMatrix* adj_matrix, const PageRankConfig* config = nullptr);

void pagerank_free(PageRankResult* result);

struct TopKNode {
    int   node_id;
    float rank;
};

void pagerank_top_k(const PageRankResult* result, int num_nodes, int k, TopKNode* top_k);

} // namespace spmv

#endif // SPMV_PAGERANK_H
