// pagerank_engine.h — the per-shard PageRank step engine shared by the
// single-GPU pagerank() and the row-sharded multi-GPU host loop (C-ABI pr_*).
#ifndef SPMV_AMD_PAGERANK_ENGINE_H
#define SPMV_AMD_PAGERANK_ENGINE_H

#include "internal.h"

namespace spmv {
namespace detail {

// Lives in device memory; every engine kernel reads `done` first.
struct PrState {
    float dangling_sum;     // dangling mass of the current r_old
    float final_residual;   // ||r_new - r_old||_2 of the last committed step
    int   iterations;       // committed steps
    int   converged;        // residual < tolerance was observed
    int   done;             // steps after this are no-ops
    int   reserved;
};

// Peers' rank vectors (IPC-mapped device pointers) that a step also writes its new slice
// into, so that no all-gather is needed afterwards.  Passed to the kernels by value.
constexpr int kMaxPushPeers = 15;
struct PushTargets {
    float* ptr[kMaxPushPeers];
    int count = 0;
};

// Where a shard's local row i sits in the (padded) rank vector:
//     base + (i / piece) * block + i % piece
// One contiguous slice per rank: piece = everything (the default), position = base + i.  The chunked layout
// of the overlapped exchange (pagerank_dist.py): the vector is a sequence of blocks, block c holds piece c of
// EVERY rank back to back (so that one in-place all-gather per block delivers it), base = rank * piece,
// block = world * piece.
struct RowMap {
    int base = 0;
    int piece = 0x7fffffff;
    int block = 0;
    __host__ __device__ long long at(long long row) const {
        return piece == 0x7fffffff ? base + row : base + (row / piece) * block + row % piece;
    }
};

// One rank's slice of the problem: local_rows consecutive rows of the n_global x n_global matrix (CSR with
// row_ptrs rebased to 0), whose nodes sit at map.at(0 .. local_rows - 1) of the rank vectors.
struct PrShard {
    int local_rows = 0;
    RowMap map;
    int n_global = 0;
    long long nnz = 0;
    const int* d_row_ptrs = nullptr;
    const int* d_cols = nullptr;
    const float* d_vals = nullptr;
    const unsigned char* d_dangling = nullptr;   // [>= n_global] 1 = dangling node
    PrState* d_state = nullptr;
    double* d_block_partials = nullptr;          // [2 * pr_max_blocks()]
    int lanes = 4;                               // lanes per row, from the mean row length
    int grid = 1;                                // workgroups that write block partials
    PlanRef tiled;                               // set => steps run through the LDS-tiled engine (kept alive by the shard)
    mutable int expanded_strips = 0;             // tiled engine: strips of the coming step already expanded (pr_expand)
    mutable bool expanded_long = false;          // ... and its long rows
};

int pr_max_blocks();
// chooses lanes / grid; `tiled` (may be null) switches the step to the tiled engine.
// Returns the number of block-partial pairs the shard needs (size of d_block_partials / 2).
int pr_shard_prepare(PrShard* shard, PlanRef tiled);
hipError_t pr_step(const PrShard& shard, const float* d_r_old, float* d_r_new, float damping,
                   const PushTargets& push, hipStream_t s);
// Optional head start on the coming step: the caller promises that columns [0, cols_ready) of d_r_old are
// final (the rest may still be arriving); with the tiled engine, phase 1 of every strip inside that range
// that has not run yet is enqueued now.  pr_step then runs only what is left.  Without a tiled plan: no-op.
hipError_t pr_expand(const PrShard& shard, const float* d_r_old, long long cols_ready, hipStream_t s);
hipError_t pr_reduce(const PrShard& shard, double* d_sums /*[2]*/, hipStream_t s);
hipError_t pr_commit(const PrShard& shard, const double* d_sums, float tolerance, hipStream_t s);
hipError_t pr_reduce_commit(const PrShard& shard, float tolerance, hipStream_t s);   // single rank: both in one launch
hipError_t pr_commit_gathered(const PrShard& shard, const float* d_gathered, int world, long long stride,
                              long long shard_len, float tolerance, hipStream_t s);
hipError_t pr_fill(float* d_r, size_t n, float value, hipStream_t s);
hipError_t pr_column_sums(long long nnz, const int* d_cols, const float* d_vals, int n_cols,
                          float* d_col_sums, hipStream_t s);
hipError_t pr_mask_from_column_sums(const float* d_col_sums, int n, unsigned char* d_mask,
                                    unsigned long long* d_count, hipStream_t s);

} // namespace detail
} // namespace spmv

#endif
