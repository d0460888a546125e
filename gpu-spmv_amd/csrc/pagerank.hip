// pagerank.hip — device-resident PageRank power iteration.
//
// Same mathematics and stop rule as the reference's host loop
// (src/pagerank.cu:50-153):
//     s      = sum of r_old over dangling nodes (columns whose stored values sum to 0)
//     r_new  = d * (A r_old) + d * s / n + (1 - d) / n
//     stop when ||r_new - r_old||_2 < tol or after max_iterations
//     result = last computed vector, divided by its sum
// but nothing crosses PCIe inside the loop: one fused kernel per iteration does
// the vector-CSR SpMV, the damping/teleport update, and the partial sums for the
// residual and the next dangling mass; a one-workgroup kernel folds the partials
// in a fixed order (double accumulators) and raises a `done` flag on the device.
// Kernels return immediately once `done` is set, so the host may run ahead of the
// convergence check without changing the result.
//
// The same kernels serve the row-sharded multi-GPU loop (pr_* entry points):
// each rank owns local_rows consecutive rows of A (their nodes placed by a RowMap) and a
// full-length rank vector; the host all-reduces the two partial sums and
// all-gathers the new slice (RCCL) between `pr_reduce` and `pr_commit`.
#include "internal.h"
#include "device_common.h"
#include "pagerank_engine.h"
#include "tiled.h"
#include "spmv/pagerank.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace spmv {
namespace detail {

namespace {

using namespace dev;

// One power-iteration step over this shard's rows.
template <int LANES>
__global__ __launch_bounds__(kBlock)
void pr_step_kernel(int local_rows, RowMap map, int n_global, long long nnz,
                    const int* __restrict__ row_ptrs,
                    const int* __restrict__ cols,
                    const float* __restrict__ vals,
                    const float* __restrict__ r_old,            // full length
                    float* __restrict__ r_new,                  // full length; slice written
                    const unsigned char* __restrict__ dangling, // full length mask
                    float damping,
                    const PrState* __restrict__ state,
                    double* __restrict__ block_partials,        // [2 * gridDim.x]
                    PushTargets push) {                         // peers' vectors that also get the new slice
    if (state->done) return;

    constexpr int kRowsPerBlock = kBlock / LANES;
    const int lane = threadIdx.x % LANES;
    const int slot = threadIdx.x / LANES;

    // same expression order as the host loop: d * y + (d * s / n) + (1 - d) / n
    const float teleport = (1.0f - damping) / n_global;
    const float dangling_term = __fdiv_rn(__fmul_rn(damping, state->dangling_sum),
                                          static_cast<float>(n_global));

    double res2 = 0.0, mass = 0.0;
    for (long long first = static_cast<long long>(blockIdx.x) * kRowsPerBlock; first < local_rows;
         first += static_cast<long long>(gridDim.x) * kRowsPerBlock) {
        const long long row = first + slot;
        float acc = 0.0f;
        if (row < local_rows) {
            acc = row_partial_dot<LANES>(row_ptrs[row], row_ptrs[row + 1], lane, nnz, cols, vals, r_old);
        }
        acc = group_sum<LANES>(acc);
        if (lane == 0 && row < local_rows) {
            const long long node = map.at(row);
            const float fresh = __fadd_rn(__fadd_rn(__fmul_rn(damping, acc), dangling_term), teleport);
            r_new[node] = fresh;
            for (int p = 0; p < push.count; ++p) push.ptr[p][node] = fresh;
            const float diff = __fsub_rn(fresh, r_old[node]);
            res2 += static_cast<double>(__fmul_rn(diff, diff));
            if (dangling[node]) mass += static_cast<double>(fresh);
        }
    }
    block_sum2(res2, mass);
    if (threadIdx.x == 0) {
        block_partials[2 * blockIdx.x] = res2;
        block_partials[2 * blockIdx.x + 1] = mass;
    }
}

// Folds the block partials left to right into sums[0] (residual^2) and sums[1]
// (dangling mass of r_new).  One workgroup; fixed order => reproducible.
__global__ __launch_bounds__(kBlock)
void pr_reduce_kernel(const double* __restrict__ block_partials, int num_blocks,
                      const PrState* __restrict__ state, double* __restrict__ sums) {
    if (state->done) return;
    double res2 = 0.0, mass = 0.0;
    for (int b = threadIdx.x; b < num_blocks; b += kBlock) {
        res2 += block_partials[2 * b];
        mass += block_partials[2 * b + 1];
    }
    block_sum2(res2, mass);
    if (threadIdx.x == 0) {
        sums[0] = res2;
        sums[1] = mass;
    }
}

// Single-rank form: fold the block partials and apply them in one launch.
__global__ __launch_bounds__(kBlock)
void pr_reduce_commit_kernel(const double* __restrict__ block_partials, int num_blocks, float tolerance,
                             PrState* __restrict__ state) {
    if (state->done) return;
    double res2 = 0.0, mass = 0.0;
    for (int b = threadIdx.x; b < num_blocks; b += kBlock) {
        res2 += block_partials[2 * b];
        mass += block_partials[2 * b + 1];
    }
    block_sum2(res2, mass);
    if (threadIdx.x == 0) {
        const float residual = static_cast<float>(sqrt(res2));
        state->iterations += 1;
        state->final_residual = residual;
        state->dangling_sum = static_cast<float>(mass);
        if (residual < tolerance) {
            state->converged = 1;
            state->done = 1;
        }
    }
}

// Applies the (already globally reduced) sums: residual, iteration count,
// convergence flag, dangling mass for the next step.
__global__ void pr_commit_kernel(const double* __restrict__ sums, float tolerance,
                                 PrState* __restrict__ state) {
    if (state->done) return;
    const float residual = static_cast<float>(sqrt(sums[0]));
    state->iterations += 1;
    state->final_residual = residual;
    state->dangling_sum = static_cast<float>(sums[1]);
    if (residual < tolerance) {
        state->converged = 1;
        state->done = 1;
    }
}

// Multi-rank form: every rank's (residual^2, dangling mass) pair travels in the 16-byte tail
// of its slice of the all-gathered vector; all ranks fold the pairs in rank order (same
// result everywhere, no separate all-reduce).
__global__ void pr_commit_gathered_kernel(const float* __restrict__ gathered, int world,
                                          long long stride, long long shard_len, float tolerance,
                                          PrState* __restrict__ state) {
    if (state->done) return;
    double res2 = 0.0, mass = 0.0;
    for (int p = 0; p < world; ++p) {
        const double* tail = reinterpret_cast<const double*>(gathered + p * stride + shard_len);
        res2 += tail[0];
        mass += tail[1];
    }
    const float residual = static_cast<float>(sqrt(res2));
    state->iterations += 1;
    state->final_residual = residual;
    state->dangling_sum = static_cast<float>(mass);
    if (residual < tolerance) {
        state->converged = 1;
        state->done = 1;
    }
}

__global__ __launch_bounds__(kBlock)
void pr_fill_kernel(float* __restrict__ r, size_t n, float value) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        r[i] = value;
    }
}

// Column sums of the stored values (atomic fp32 adds; only `== 0` is consumed).
__global__ __launch_bounds__(kBlock)
void pr_colsum_kernel(long long nnz, const int* __restrict__ cols, const float* __restrict__ vals,
                      int n_cols, float* __restrict__ col_sums) {
    for (long long j = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; j < nnz;
         j += static_cast<long long>(gridDim.x) * kBlock) {
        const int c = cols[j];
        if (c >= 0 && c < n_cols) atomicAdd(&col_sums[c], vals[j]);
    }
}

// mask[c] = (col_sums[c] == 0); also the dangling mass of a constant vector.
__global__ __launch_bounds__(kBlock)
void pr_mask_kernel(const float* __restrict__ col_sums, int n, unsigned char* __restrict__ mask,
                    unsigned long long* __restrict__ count) {
    unsigned long long local = 0;
    for (long long c = static_cast<long long>(blockIdx.x) * kBlock + threadIdx.x; c < n;
         c += static_cast<long long>(gridDim.x) * kBlock) {
        const bool d = col_sums[c] == 0.0f;
        mask[c] = d;
        local += d;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

// ---- final renormalisation r /= sum(r) on the device: block partial sums in double, then every
// block folds the partials in the same fixed order and scales its share (IEEE division) ----
__global__ __launch_bounds__(kBlock)
void pr_vector_sum_kernel(const float* __restrict__ v, size_t n, double* __restrict__ block_out) {
    double acc = 0.0, unused = 0.0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        acc += static_cast<double>(v[i]);
    }
    block_sum2<kBlock>(acc, unused);
    if (threadIdx.x == 0) block_out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kBlock)
void pr_scale_kernel(float* __restrict__ v, size_t n, const double* __restrict__ block_sums, int blocks) {
    __shared__ float s_total;
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int b = 0; b < blocks; ++b) total += block_sums[b];
        s_total = static_cast<float>(total);
    }
    __syncthreads();
    const float total = s_total;
    if (!(total > 0.0f)) return;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        v[i] = __fdiv_rn(v[i], total);
    }
}

// ---- top-k by radix select on the float bit patterns (non-negative ranks order like uints) ----
// pass A: histogram of the high 16 bits; pass B: histogram of the low 16 bits of the values
// whose high half equals `prefix`; pass C: gather everything above the threshold plus as many
// equal-to-threshold values as are still needed.
__global__ __launch_bounds__(kBlock)
void topk_hist_hi_kernel(const unsigned int* __restrict__ bits, size_t n, unsigned int* __restrict__ hist,
                         unsigned int* __restrict__ bad) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const unsigned int b = bits[i];
        if (b > 0x7F800000u) atomicOr(bad, 1u);           // negative or NaN: not orderable this way
        atomicAdd(&hist[b >> 16], 1u);
    }
}

__global__ __launch_bounds__(kBlock)
void topk_hist_lo_kernel(const unsigned int* __restrict__ bits, size_t n, unsigned int prefix,
                         unsigned int* __restrict__ hist) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const unsigned int b = bits[i];
        if ((b >> 16) == prefix) atomicAdd(&hist[b & 0xFFFFu], 1u);
    }
}

__global__ __launch_bounds__(kBlock)
void topk_gather_kernel(const unsigned int* __restrict__ bits, size_t n, unsigned int threshold,
                        unsigned int take_equal, unsigned int* __restrict__ cursor /*[2]: out, equal*/,
                        TopKNode* __restrict__ out) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const unsigned int b = bits[i];
        bool keep = b > threshold;
        if (b == threshold) keep = atomicAdd(&cursor[1], 1u) < take_equal;
        if (keep) {
            const unsigned int at = atomicAdd(&cursor[0], 1u);
            out[at].node_id = static_cast<int>(i);
            out[at].rank = __uint_as_float(b);
        }
    }
}

int grid_for(long long rows, int rows_per_block) {
    const long long blocks = (rows + rows_per_block - 1) / rows_per_block;
    return static_cast<int>(std::max(1LL, std::min<long long>(blocks, kMaxResidentBlocks)));
}

template <int LANES>
hipError_t launch_step(const PrShard& sh, const float* r_old, float* r_new, float damping,
                       const PushTargets& push, hipStream_t s) {
    pr_step_kernel<LANES><<<sh.grid, kBlock, 0, s>>>(
        sh.local_rows, sh.map, sh.n_global, sh.nnz, sh.d_row_ptrs, sh.d_cols, sh.d_vals,
        r_old, r_new, sh.d_dangling, damping, sh.d_state, sh.d_block_partials, push);
    return hipGetLastError();
}

} // namespace

int pr_max_blocks() { return kMaxResidentBlocks; }

int pr_shard_prepare(PrShard* sh, PlanRef tiled) {
    const float avg = sh->local_rows > 0 ? static_cast<float>(sh->nnz) / sh->local_rows : 0.0f;
    sh->lanes = pick_lanes_per_row(avg);
    sh->tiled = std::move(tiled);
    sh->grid = sh->tiled ? sh->tiled->num_tiles : grid_for(sh->local_rows, kBlock / sh->lanes);
    return std::max(sh->grid, kMaxResidentBlocks);
}

hipError_t pr_step(const PrShard& sh, const float* r_old, float* r_new, float damping,
                   const PushTargets& push, hipStream_t s) {
    if (sh.local_rows <= 0) return hipSuccess;
    if (sh.tiled) {
        // phase 1 of whatever pr_expand has not run yet (all of it, normally), then phase 2
        const int done = std::min(sh.expanded_strips, sh.tiled->num_strips);
        const bool long_done = sh.expanded_long;
        sh.expanded_strips = 0;
        sh.expanded_long = false;
        // Host threads may share a matrix and a stream (a pagerank() loop beside an spmv_csr call): the two launches
        // of a step go out together, like tiled_spmv's pair, or another pair's expand could land between them and
        // overwrite the stream's product scratch.  (After a head start by pr_expand the caller orders the launches.)
        std::unique_lock<std::mutex> pair(sh.tiled->launch_lock, std::defer_lock);
        if (done == 0 && !long_done) pair.lock();
        if (done < sh.tiled->num_strips || !long_done) {
            const hipError_t e = tiled_pagerank_expand(*sh.tiled, done, sh.tiled->num_strips, !long_done, r_old, sh.d_state, s);
            if (e != hipSuccess) return e;
        }
        return tiled_pagerank_finish(*sh.tiled, sh.map, sh.n_global, r_old, r_new, sh.d_dangling,
                                     damping, sh.d_state, sh.d_block_partials, push, s);
    }
    switch (sh.lanes) {
        case 1:  return launch_step<1>(sh, r_old, r_new, damping, push, s);
        case 2:  return launch_step<2>(sh, r_old, r_new, damping, push, s);
        case 4:  return launch_step<4>(sh, r_old, r_new, damping, push, s);
        case 8:  return launch_step<8>(sh, r_old, r_new, damping, push, s);
        case 16: return launch_step<16>(sh, r_old, r_new, damping, push, s);
        case 32: return launch_step<32>(sh, r_old, r_new, damping, push, s);
        default: return launch_step<64>(sh, r_old, r_new, damping, push, s);
    }
}

hipError_t pr_expand(const PrShard& sh, const float* r_old, long long cols_ready, hipStream_t s) {
    if (sh.local_rows <= 0 || !sh.tiled) return hipSuccess;
    const TiledPlan& plan = *sh.tiled;
    const bool all = cols_ready >= plan.num_cols;
    const int ready = all ? plan.num_strips : static_cast<int>(std::max(0LL, cols_ready) / plan.strip_cols);
    const int done = std::min(sh.expanded_strips, plan.num_strips);
    const bool want_long = all && !sh.expanded_long;       // the long rows read all of r_old
    if (ready <= done && !want_long) return hipSuccess;
    const hipError_t e = tiled_pagerank_expand(plan, done, std::max(ready, done), want_long, r_old, sh.d_state, s);
    if (e == hipSuccess) {
        sh.expanded_strips = std::max(ready, done);
        sh.expanded_long = sh.expanded_long || want_long;
    }
    return e;
}

hipError_t pr_reduce(const PrShard& sh, double* d_sums, hipStream_t s) {
    pr_reduce_kernel<<<1, kBlock, 0, s>>>(sh.d_block_partials, sh.local_rows > 0 ? sh.grid : 0,
                                          sh.d_state, d_sums);
    return hipGetLastError();
}

hipError_t pr_reduce_commit(const PrShard& sh, float tolerance, hipStream_t s) {
    pr_reduce_commit_kernel<<<1, kBlock, 0, s>>>(sh.d_block_partials, sh.local_rows > 0 ? sh.grid : 0, tolerance,
                                                 sh.d_state);
    return hipGetLastError();
}

hipError_t pr_commit(const PrShard& sh, const double* d_sums, float tolerance, hipStream_t s) {
    pr_commit_kernel<<<1, 1, 0, s>>>(d_sums, tolerance, sh.d_state);
    return hipGetLastError();
}

hipError_t pr_commit_gathered(const PrShard& sh, const float* d_gathered, int world, long long stride,
                              long long shard_len, float tolerance, hipStream_t s) {
    pr_commit_gathered_kernel<<<1, 1, 0, s>>>(d_gathered, world, stride, shard_len, tolerance, sh.d_state);
    return hipGetLastError();
}

hipError_t pr_fill(float* d_r, size_t n, float value, hipStream_t s) {
    if (n == 0) return hipSuccess;
    pr_fill_kernel<<<grid_for(static_cast<long long>(n), kBlock * 4), kBlock, 0, s>>>(d_r, n, value);
    return hipGetLastError();
}

hipError_t pr_column_sums(long long nnz, const int* d_cols, const float* d_vals, int n_cols,
                          float* d_col_sums, hipStream_t s) {
    if (nnz == 0) return hipSuccess;
    pr_colsum_kernel<<<grid_for(nnz, kBlock * 4), kBlock, 0, s>>>(nnz, d_cols, d_vals, n_cols, d_col_sums);
    return hipGetLastError();
}

hipError_t pr_mask_from_column_sums(const float* d_col_sums, int n, unsigned char* d_mask,
                                    unsigned long long* d_count, hipStream_t s) {
    if (n == 0) return hipSuccess;
    pr_mask_kernel<<<grid_for(n, kBlock * 4), kBlock, 0, s>>>(d_col_sums, n, d_mask, d_count);
    return hipGetLastError();
}

// v /= sum(v), the sum accumulated in double; d_scratch holds >= kNormaliseBlocks doubles
constexpr int kNormaliseBlocks = 1024;
hipError_t pr_normalise(float* d_v, size_t n, double* d_scratch, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = static_cast<int>(std::min<size_t>(kNormaliseBlocks, (n + kBlock - 1) / kBlock));
    pr_vector_sum_kernel<<<blocks, kBlock, 0, s>>>(d_v, n, d_scratch);
    pr_scale_kernel<<<blocks, kBlock, 0, s>>>(d_v, n, d_scratch, blocks);
    return hipGetLastError();
}

} // namespace detail

// ---------------------------------------------------------------------------
// public single-GPU entry point
// ---------------------------------------------------------------------------
namespace {

// Dangling columns exactly as the reference's host scan (src/pagerank.cu:20-48):
// sequential fp32 column sums in storage order, dangling <=> sum == 0.0f.
std::vector<unsigned char> dangling_mask_host(const CSRMatrix* A) {
    const int n = A->num_cols;
    std::vector<float> sums(std::max(n, 0), 0.0f);
    for (int r = 0; r < A->num_rows; ++r) {
        for (int j = A->row_ptrs[r]; j < A->row_ptrs[r + 1]; ++j) {
            const int c = A->col_indices[j];
            if (c >= 0 && c < n) sums[c] += A->values[j];
        }
    }
    std::vector<unsigned char> mask(sums.size());
    for (size_t c = 0; c < sums.size(); ++c) mask[c] = sums[c] == 0.0f;
    return mask;
}

template <typename T>
struct DeviceArray {
    T* ptr = nullptr;
    ~DeviceArray() { if (ptr) (void)hipFree(ptr); }
    hipError_t alloc(size_t count) {
        return hipMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(count, 1) * sizeof(T));
    }
};

// SPMV_TRACE=1: wall-clock time of each host phase of a call, on stderr
struct Trace {
    const char* what;
    bool on;
    mutable std::chrono::steady_clock::time_point last;
    explicit Trace(const char* name) : what(name), on(std::getenv("SPMV_TRACE") != nullptr),
                                       last(std::chrono::steady_clock::now()) {}
    void mark(const char* phase) const {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[spmv trace] %s: %-22s %9.3f ms\n", what, phase,
                     std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

} // namespace

namespace {

// PageRankResult::ranks of large graphs: pinned host arrays owned by the library and recycled by
// pagerank_free(), so that the final device -> host copy runs at PCIe rate straight into the caller's array
// (a fresh pageable 40 MB array costs ~7 ms in page faults and staging; the reference's contract is that
// pagerank_free() releases the array, include/spmv/pagerank.h:36).  Small results stay plain new[] arrays.
class ResultPool {
public:
    static constexpr size_t kMinPooled = 1u << 18;       // floats
    float* take(size_t n) {
        {
            std::lock_guard<std::mutex> guard(lock_);
            for (size_t i = 0; i < idle_.size(); ++i) {
                if (idle_[i].second == n) {
                    float* p = idle_[i].first;
                    idle_.erase(idle_.begin() + i);
                    lent_.push_back({p, n});
                    return p;
                }
            }
        }
        float* p = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::lock_guard<std::mutex> guard(lock_);
        lent_.push_back({p, n});
        return p;
    }
    // true when `p` was one of ours (now back in the pool, or released when the pool is full)
    bool give_back(float* p) {
        std::pair<float*, size_t> entry{nullptr, 0};
        {
            std::lock_guard<std::mutex> guard(lock_);
            for (size_t i = 0; i < lent_.size(); ++i) {
                if (lent_[i].first == p) {
                    entry = lent_[i];
                    lent_.erase(lent_.begin() + i);
                    break;
                }
            }
            if (!entry.first) return false;
            if (idle_.size() < 2) {
                idle_.push_back(entry);
                return true;
            }
        }
        (void)hipHostFree(entry.first);
        return true;
    }
    bool owns(const float* p) {
        std::lock_guard<std::mutex> guard(lock_);
        for (const auto& e : lent_) if (e.first == p) return true;
        return false;
    }

private:
    std::mutex lock_;
    std::vector<std::pair<float*, size_t>> idle_, lent_;
};

ResultPool& result_pool() {
    static ResultPool* pool = new ResultPool();      // never destroyed: results may outlive static teardown
    return *pool;
}

// Claims the matrix's cached workspace (or, when another call holds it, a private one) and makes sure
// it holds vectors of `len` elements and `partial_pairs` pairs of partial sums.
class WorkspaceLease {
public:
    WorkspaceLease(const CSRMatrix* adj) {
        detail::CsrAux* aux = detail::aux_lookup(adj->d_row_ptrs, true);
        static std::mutex claim;
        std::lock_guard<std::mutex> guard(claim);
        if (aux && !aux->pagerank.busy) {
            aux->pagerank.busy = true;
            ws_ = &aux->pagerank;
        } else {
            ws_ = &own_;
            private_ = true;
        }
    }
    ~WorkspaceLease() {
        if (private_) own_.release();
        else ws_->busy = false;
    }
    detail::PrWorkspace* operator->() { return ws_; }
    bool is_private() const { return private_; }

    bool ensure(size_t len, size_t partial_count) {
        detail::PrWorkspace& w = *ws_;
        if (w.len != len) {
            const bool keep_busy = w.busy;
            w.release();
            w.busy = keep_busy;
            bool ok = hipMalloc(reinterpret_cast<void**>(&w.r[0]), len * sizeof(float)) == hipSuccess
                   && hipMalloc(reinterpret_cast<void**>(&w.r[1]), len * sizeof(float)) == hipSuccess
                   && hipMalloc(reinterpret_cast<void**>(&w.mask), len) == hipSuccess
                   && hipMalloc(reinterpret_cast<void**>(&w.sums), 2 * sizeof(double)) == hipSuccess
                   && hipMalloc(&w.state, sizeof(detail::PrState)) == hipSuccess
                   && hipMalloc(reinterpret_cast<void**>(&w.dangling_count), sizeof(unsigned long long)) == hipSuccess
                   && hipHostMalloc(&w.pinned_state, 2 * sizeof(detail::PrState)) == hipSuccess
                   && hipEventCreateWithFlags(&w.seen[0], hipEventDisableTiming) == hipSuccess
                   && hipEventCreateWithFlags(&w.seen[1], hipEventDisableTiming) == hipSuccess;
            if (!ok) return false;
            w.len = len;
        }
        if (w.partial_count < partial_count) {
            if (w.partials) (void)hipFree(w.partials);
            w.partials = nullptr;
            w.partial_count = 0;
            if (hipMalloc(reinterpret_cast<void**>(&w.partials), partial_count * sizeof(double)) != hipSuccess) return false;
            w.partial_count = partial_count;
        }
        return true;
    }

private:
    detail::PrWorkspace* ws_ = nullptr;
    detail::PrWorkspace own_;
    bool private_ = false;
};

} // namespace

PageRankResult pagerank(const CSRMatrix* adj, const PageRankConfig* config) {
    PageRankResult result;
    if (!adj) return result;

    const PageRankConfig fallback;
    if (!config) config = &fallback;

    // extension: SPMV_NUM_GPUS=N shards the call over N devices (include/spmv/pagerank.h)
    if (const char* env = std::getenv("SPMV_NUM_GPUS")) {
        const int gpus = std::atoi(env);
        if (gpus > 1 && adj->num_rows > 0 && adj->row_ptrs && (adj->nnz == 0 || (adj->col_indices && adj->values))) {
            // pagerank() always hands back an allocated ranks array (the reference's contract: callers index it
            // unchecked); the empty result of a multi-device run that could not start or did not finish — fewer
            // devices than asked for, no librccl, a device error — is therefore not passed on: one device does it
            PageRankResult sharded = pagerank_multi_gpu(adj, config, gpus);
            if (sharded.ranks) return sharded;
            std::fprintf(stderr, "[spmv] SPMV_NUM_GPUS=%d: the multi-device run returned nothing; using one device\n", gpus);
        }
    }

    const int n = adj->num_rows;
    if (static_cast<size_t>(std::max(n, 0)) >= ResultPool::kMinPooled) result.ranks = result_pool().take(static_cast<size_t>(n));
    const bool pinned_result = result.ranks != nullptr;
    if (!result.ranks) result.ranks = new float[std::max(n, 0)];
    const float start = n > 0 ? 1.0f / n : 0.0f;
    if (n <= 0) return result;
    // every exit that does not deliver computed ranks hands back the start vector (filled on the way
    // out, so the successful path does not pay a 4n-byte host pass)
    struct StartVector {
        float* ranks; int n; float value; bool armed;
        ~StartVector() { if (armed) std::fill_n(ranks, n, value); }
    } start_vector{result.ranks, n, start, true};

    // The matrix must be resident on the device (csr_to_gpu), like the reference's
    // spmv_csr call requires; otherwise the loop ends at once with the start vector.
    if (!adj->d_row_ptrs || (adj->nnz > 0 && (!adj->d_col_indices || !adj->d_values))) {
        return result;
    }

    const detail::TraceRange range("spmv:pagerank");
    hipStream_t stream = detail::current_stream();
    using detail::PrState;
    const Trace trace("pagerank");

    // Steps through the LDS-tiled engine pay once x has left the L2s — but its plan costs about as much
    // as a few direct-gather steps, and max_iterations says nothing about how soon the loop converges
    // (the 10 M-node uniform graph: 3 iterations).  So: a plan the matrix already holds is used from the
    // first step; otherwise the loop starts on the direct kernel and builds the plan only once it has
    // spent about one build's worth of time on direct steps (ski rental: never more than ~2x the better
    // choice).  Estimates per stored entry, measured on C5: build 34 ps, direct step 17 ps, tiled step 3.3 ps:
    // break-even after 2.5 direct steps.  One more than that: the loop enqueues one step past convergence, and
    // that step of a run that converges in exactly three (the 10 M-node uniform graph) should not start a build.
    // The matrix's workspace goes to one call at a time; a second call running concurrently on the same matrix
    // gets a private one — and stays on the direct kernels, because a step through the tiled engine is two
    // launches around the plan's product stream and two loops interleaving on one stream would mix them up.
    WorkspaceLease ws(adj);
    detail::PlanRef plan = ws.is_private() ? detail::PlanRef() : detail::tiled_plan_if_cached(adj);
    int build_plan_at = -1;
    if (!plan && !ws.is_private() && detail::tiled_eligible(adj)) {
        build_plan_at = static_cast<int>(std::ceil(34.0 / (17.0 - 3.3))) + 1;        // = 4 direct steps
        build_plan_at = static_cast<int>(std::max(0LL, detail::debug_number("pr_plan_after", build_plan_at)));
        if (build_plan_at == 0) plan = detail::tiled_plan_for(adj, stream);
    }

    // vectors are indexed by column during the SpMV and by row during the update
    const size_t len = static_cast<size_t>(std::max(n, adj->num_cols));
    detail::PrShard shard;
    shard.local_rows = n;
    shard.n_global = n;
    shard.nnz = adj->nnz;
    shard.d_row_ptrs = adj->d_row_ptrs;
    shard.d_cols = adj->d_col_indices;
    shard.d_vals = adj->d_values;
    int partial_pairs = detail::pr_shard_prepare(&shard, plan);
    if (build_plan_at > 0) {       // room for the tiled engine's partial sums too, should the plan arrive later
        int w = 0, r = 0;
        detail::tiled_shape_for(n, adj->num_cols, adj->nnz, &w, &r);
        if (r > 0) partial_pairs = std::max(partial_pairs, (n + r - 1) / r);
    }
    if (!ws.ensure(len, std::max<size_t>(2 * static_cast<size_t>(partial_pairs), detail::kNormaliseBlocks))) return result;
    shard.d_dangling = ws->mask;
    shard.d_state = static_cast<PrState*>(ws->state);
    shard.d_block_partials = ws->partials;

    bool ok = detail::pr_fill(ws->r[0], len, start, stream) == hipSuccess
           && detail::pr_fill(ws->r[1], len, start, stream) == hipSuccess;
    trace.mark("workspace + plan lookup");

    // Dangling mask (kept with the matrix: it depends on the matrix only).  A plan with folded values knows
    // every column's one stored value w (0 where the column has no entry): the reference's sequential fp32
    // column sum of k copies of w is 0 exactly when w == 0, so the mask is read off the weights.  Otherwise:
    // host scan when host arrays exist (reference semantics), else atomic column sums on the device.
    if (ws->mask_valid && (ws->mask_cols != adj->d_col_indices || ws->mask_vals != adj->d_values || ws->mask_nnz != adj->nnz ||
                           ws->mask_rows != adj->num_rows || ws->mask_num_cols != adj->num_cols)) {
        ws->mask_valid = false;          // another matrix over the same row pointers, or arrays swapped on the handle
    }
    if (ok && !ws->mask_valid) {
        unsigned long long num_dangling = 0;
        ok = hipMemsetAsync(ws->mask, 0, len, stream) == hipSuccess
          && hipMemsetAsync(ws->dangling_count, 0, sizeof(unsigned long long), stream) == hipSuccess;
        if (ok && plan && plan->col_weight) {
            ok = detail::pr_mask_from_column_sums(plan->col_weight, std::min(n, adj->num_cols), ws->mask,
                                                  ws->dangling_count, stream) == hipSuccess
              && hipMemcpyAsync(&num_dangling, ws->dangling_count, sizeof(num_dangling),
                                hipMemcpyDeviceToHost, stream) == hipSuccess
              && hipStreamSynchronize(stream) == hipSuccess;
        } else if (ok && adj->values && adj->col_indices && adj->row_ptrs) {
            const std::vector<unsigned char> host_mask = dangling_mask_host(adj);
            const size_t m = std::min(host_mask.size(), static_cast<size_t>(n));
            for (size_t c = 0; c < m; ++c) num_dangling += host_mask[c];
            ok = hipMemcpyAsync(ws->mask, host_mask.data(), m, hipMemcpyHostToDevice, stream) == hipSuccess
              && hipStreamSynchronize(stream) == hipSuccess;
        } else if (ok) {
            DeviceArray<float> col_sums;
            ok = col_sums.alloc(len) == hipSuccess
              && hipMemsetAsync(col_sums.ptr, 0, len * sizeof(float), stream) == hipSuccess
              && detail::pr_column_sums(adj->nnz, adj->d_col_indices, adj->d_values, adj->num_cols,
                                        col_sums.ptr, stream) == hipSuccess
              && detail::pr_mask_from_column_sums(col_sums.ptr, std::min(n, adj->num_cols), ws->mask,
                                                  ws->dangling_count, stream) == hipSuccess
              && hipMemcpyAsync(&num_dangling, ws->dangling_count, sizeof(num_dangling),
                                hipMemcpyDeviceToHost, stream) == hipSuccess
              && hipStreamSynchronize(stream) == hipSuccess;
        }
        if (ok) {
            ws->num_dangling = num_dangling;
            ws->mask_valid = true;
            ws->mask_cols = adj->d_col_indices;
            ws->mask_vals = adj->d_values;
            ws->mask_nnz = adj->nnz;
            ws->mask_rows = adj->num_rows;
            ws->mask_num_cols = adj->num_cols;
        }
    }
    if (!ok) return result;
    trace.mark("dangling mask");

    // dangling mass of the start vector: the same left-to-right fp32 sum as the host loop
    PrState host_state{};
    for (unsigned long long k = 0; k < ws->num_dangling; ++k) host_state.dangling_sum += start;
    ok = hipMemcpyAsync(ws->state, &host_state, sizeof(PrState), hipMemcpyHostToDevice, stream) == hipSuccess;

    // Pinned mirrors of the device state, two deep: the host enqueues step k+1
    // before it looks at the outcome of step k.
    PrState* pinned = static_cast<PrState*>(ws->pinned_state);
    float* bufs[2] = {ws->r[0], ws->r[1]};
    for (int iter = 0; ok && iter < config->max_iterations; ++iter) {
        if (!plan && iter == build_plan_at) {
            // enough direct steps paid: switch to the tiled engine for the rest (the queue is drained first:
            // the outcome of every enqueued step is known, so nothing is built for a loop that has converged)
            ok = hipStreamSynchronize(stream) == hipSuccess;
            if (ok && iter >= 1 && pinned[(iter - 1) & 1].done) break;
            plan = ok ? detail::tiled_plan_for(adj, stream) : nullptr;
            if (plan) (void)detail::pr_shard_prepare(&shard, plan);
            trace.mark("plan build (after direct steps)");
        }
        const detail::TraceRange step_range("spmv:pagerank_step");
        const float* r_old = bufs[iter & 1];
        float* r_new = bufs[(iter + 1) & 1];
        ok = ok && detail::pr_step(shard, r_old, r_new, config->damping_factor, detail::PushTargets{}, stream) == hipSuccess
          && detail::pr_reduce_commit(shard, config->tolerance, stream) == hipSuccess
          && hipMemcpyAsync(&pinned[iter & 1], ws->state, sizeof(PrState),
                            hipMemcpyDeviceToHost, stream) == hipSuccess
          && hipEventRecord(ws->seen[iter & 1], stream) == hipSuccess;
        if (ok && iter >= 1) {
            ok = hipEventSynchronize(ws->seen[(iter - 1) & 1]) == hipSuccess;
            if (ok && pinned[(iter - 1) & 1].done) break;
        }
    }

    if (ok) {
        ok = hipMemcpyAsync(&host_state, ws->state, sizeof(PrState), hipMemcpyDeviceToHost, stream) == hipSuccess
          && hipStreamSynchronize(stream) == hipSuccess;
    }
    trace.mark("iterations");
    if (ok) {
        result.iterations = host_state.iterations;
        result.final_residual = host_state.final_residual;
        result.converged = host_state.converged != 0;
        // the last written vector: step k (0-based) writes bufs[(k + 1) & 1].
        // Final renormalisation on the device before the copy: r /= sum(r), the sum accumulated in
        // double (the reference's fp32 running sum loses digits at n ~ 1e7).
        float* last = bufs[host_state.iterations & 1];
        ok = detail::pr_normalise(last, static_cast<size_t>(n), ws->partials, stream) == hipSuccess;
        // copy out: device -> pinned staging at PCIe rate, then into the caller's (pageable, freshly
        // allocated) array; SPMV_DEBUG=pr_copy=direct hands the pageable array to the runtime instead
        const bool direct_copy = detail::debug_is("pr_copy", "direct");
        const size_t bytes = static_cast<size_t>(n) * sizeof(float);
        if (ok && (direct_copy || pinned_result)) {
            ok = hipMemcpyAsync(result.ranks, last, bytes, hipMemcpyDeviceToHost, stream) == hipSuccess
              && hipStreamSynchronize(stream) == hipSuccess;
        } else if (ok) {
            // in pieces, so that the host copy of piece k runs while piece k + 1 crosses PCIe; the staging array is
            // allocated by the first call that comes this way (large graphs get the pooled pinned result instead)
            if (!ws->pinned_ranks) {
                ok = hipHostMalloc(reinterpret_cast<void**>(&ws->pinned_ranks), ws->len * sizeof(float)) == hipSuccess;
                if (!ok) { (void)hipGetLastError(); ws->pinned_ranks = nullptr; }
            }
            const size_t piece = 4u << 20;          // floats
            size_t done = 0;
            hipEvent_t landed[2] = {ws->seen[0], ws->seen[1]};
            size_t issued = 0;
            int turn = 0;
            while (ok && done < static_cast<size_t>(n)) {
                while (ok && issued < static_cast<size_t>(n) && issued < done + 2 * piece) {
                    const size_t count = std::min(piece, static_cast<size_t>(n) - issued);
                    ok = hipMemcpyAsync(ws->pinned_ranks + issued, last + issued, count * sizeof(float),
                                        hipMemcpyDeviceToHost, stream) == hipSuccess
                      && hipEventRecord(landed[turn & 1], stream) == hipSuccess;
                    issued += count;
                    ++turn;
                }
                const size_t count = std::min(piece, static_cast<size_t>(n) - done);
                const int which = static_cast<int>((done / piece) & 1);
                ok = ok && hipEventSynchronize(landed[which]) == hipSuccess;
                if (ok) std::memcpy(result.ranks + done, ws->pinned_ranks + done, count * sizeof(float));
                done += count;
            }
        }
    }
    trace.mark("normalise + copy out");
    start_vector.armed = !ok;
    return result;
}

void pagerank_free(PageRankResult* result) {
    if (result && result->ranks) {
        if (!result_pool().give_back(result->ranks)) delete[] result->ranks;
        result->ranks = nullptr;
    }
}

namespace {

// Device selection for large vectors: uploads the ranks, finds the k-th
// largest value by a two-level radix histogram, gathers the k winners and sorts only those on the
// host.  Returns false (caller falls back to the host partial sort) on any failure or when the
// ranks hold negative values / NaNs.
bool top_k_on_device(const float* ranks, int n, int keep, TopKNode* top_k) {
    using namespace detail;
    hipStream_t s = current_stream();
    DeviceArray<unsigned int> bits, hist, scratch;
    DeviceArray<TopKNode> winners;
    if (bits.alloc(n) != hipSuccess || hist.alloc(65536) != hipSuccess || scratch.alloc(4) != hipSuccess ||
        winners.alloc(keep) != hipSuccess) {
        return false;
    }
    std::vector<unsigned int> host_hist(65536);
    unsigned int host_scratch[4] = {0, 0, 0, 0};
    const int grid = 2048;
    bool ok = hipMemcpyAsync(bits.ptr, ranks, static_cast<size_t>(n) * sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess
           && hipMemsetAsync(hist.ptr, 0, 65536 * sizeof(unsigned int), s) == hipSuccess
           && hipMemsetAsync(scratch.ptr, 0, sizeof(host_scratch), s) == hipSuccess;
    if (!ok) return false;
    topk_hist_hi_kernel<<<grid, dev::kBlock, 0, s>>>(bits.ptr, n, hist.ptr, scratch.ptr + 2);
    ok = hipMemcpyAsync(host_hist.data(), hist.ptr, 65536 * sizeof(unsigned int), hipMemcpyDeviceToHost, s) == hipSuccess
      && hipMemcpyAsync(host_scratch, scratch.ptr, sizeof(host_scratch), hipMemcpyDeviceToHost, s) == hipSuccess
      && hipStreamSynchronize(s) == hipSuccess;
    if (!ok || host_scratch[2] != 0) return false;

    // walk the histogram from the top: `prefix` is the bin holding the k-th largest value
    unsigned long long above = 0;
    int prefix = 65535;
    for (; prefix >= 0; --prefix) {
        if (above + host_hist[prefix] >= static_cast<unsigned long long>(keep)) break;
        above += host_hist[prefix];
    }
    if (prefix < 0) return false;
    ok = hipMemsetAsync(hist.ptr, 0, 65536 * sizeof(unsigned int), s) == hipSuccess;
    topk_hist_lo_kernel<<<grid, dev::kBlock, 0, s>>>(bits.ptr, n, static_cast<unsigned int>(prefix), hist.ptr);
    ok = ok && hipMemcpyAsync(host_hist.data(), hist.ptr, 65536 * sizeof(unsigned int), hipMemcpyDeviceToHost, s) == hipSuccess
            && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return false;
    int low = 65535;
    for (; low >= 0; --low) {
        if (above + host_hist[low] >= static_cast<unsigned long long>(keep)) break;
        above += host_hist[low];
    }
    if (low < 0) return false;
    const unsigned int threshold = (static_cast<unsigned int>(prefix) << 16) | static_cast<unsigned int>(low);
    const unsigned int take_equal = static_cast<unsigned int>(keep - above);   // ties: any of the equals
    topk_gather_kernel<<<grid, dev::kBlock, 0, s>>>(bits.ptr, n, threshold, take_equal, scratch.ptr, winners.ptr);
    std::vector<TopKNode> host_winners(keep);
    ok = hipGetLastError() == hipSuccess
      && hipMemcpyAsync(host_winners.data(), winners.ptr, static_cast<size_t>(keep) * sizeof(TopKNode),
                        hipMemcpyDeviceToHost, s) == hipSuccess
      && hipMemcpyAsync(host_scratch, scratch.ptr, sizeof(host_scratch), hipMemcpyDeviceToHost, s) == hipSuccess
      && hipStreamSynchronize(s) == hipSuccess;
    if (!ok || host_scratch[0] != static_cast<unsigned int>(keep)) return false;
    std::sort(host_winners.begin(), host_winners.end(), [](const TopKNode& a, const TopKNode& b) {
        return a.rank > b.rank || (a.rank == b.rank && a.node_id < b.node_id);
    });
    std::copy(host_winners.begin(), host_winners.end(), top_k);
    return true;
}

} // namespace

void pagerank_top_k(const PageRankResult* result, int num_nodes, int k, TopKNode* top_k) {
    if (!result || !result->ranks || !top_k || k <= 0 || num_nodes <= 0) return;

    // large vectors, small k: select on the device (the host partial sort is O(n log k))
    if (num_nodes >= (1 << 20) && k <= 65536) {
        int devices = 0;
        if (hipGetDeviceCount(&devices) == hipSuccess && devices > 0 &&
            top_k_on_device(result->ranks, num_nodes, std::min(k, num_nodes), top_k)) {
            return;
        }
        (void)hipGetLastError();
    }

    std::vector<int> order(num_nodes);
    for (int i = 0; i < num_nodes; ++i) order[i] = i;
    const int keep = std::min(k, num_nodes);
    const float* ranks = result->ranks;
    std::partial_sort(order.begin(), order.begin() + keep, order.end(),
                      [ranks](int a, int b) { return ranks[a] > ranks[b]; });
    for (int i = 0; i < keep; ++i) {
        top_k[i].node_id = order[i];
        top_k[i].rank = ranks[order[i]];
    }
}

} // namespace spmv
