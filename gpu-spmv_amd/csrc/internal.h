// internal.h — helpers shared by the library's translation units (not installed).
#ifndef SPMV_AMD_INTERNAL_H
#define SPMV_AMD_INTERNAL_H

#include "spmv/common.h"
#include "spmv/csr_matrix.h"
#include "spmv/ell_matrix.h"
#include "spmv/spmv.h"

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

namespace spmv {
namespace detail {

inline int code(SpMVError e) { return static_cast<int>(e); }

// ---- per-matrix auxiliary data (side table keyed by the device row_ptrs) ----
// The public structs cannot grow (callers poke the fields), so anything the
// kernels precompute for a matrix lives here and is dropped by csr_free_gpu.
struct TiledPlan;
// A plan is shared between the side table and whoever is using it (a dispatch in flight, a PageRank shard
// engine): replacing or dropping the table's entry never pulls it from under a user.
using PlanRef = std::shared_ptr<const TiledPlan>;

// Buffers of pagerank() kept with the matrix between calls (seven device allocations, a pinned mirror and
// two events per call cost more than the iterations on a large graph), plus the dangling mask, which
// depends on the matrix only.
struct PrWorkspace {
    size_t len = 0;                       // vector length everything below is sized for
    float* r[2] = {nullptr, nullptr};     // the two rank vectors
    unsigned char* mask = nullptr;        // dangling mask
    bool mask_valid = false;
    // the matrix the mask was computed for: two matrices may share one row-pointer array (every k-per-row
    // graph has the same one), and a handle's column / value arrays may be swapped under it
    const void* mask_cols = nullptr;
    const void* mask_vals = nullptr;
    long long mask_nnz = 0;
    int mask_rows = 0, mask_num_cols = 0;
    unsigned long long num_dangling = 0;
    double* partials = nullptr;           // block partial sums / normalisation scratch
    size_t partial_count = 0;
    double* sums = nullptr;               // [2]
    void* state = nullptr;                // PrState on the device
    unsigned long long* dangling_count = nullptr;
    void* pinned_state = nullptr;         // PrState[2], pinned host
    float* pinned_ranks = nullptr;        // [len] pinned host staging for the copy out: allocated by the first call that needs it
    hipEvent_t seen[2] = {nullptr, nullptr};
    bool busy = false;                    // a call is using it (a concurrent call on the same matrix allocates its own)
    void release();
};

struct CsrAux {
    PrWorkspace pagerank;
    // row-length statistics computed once (host scan or device reduction)
    bool   have_stats = false;
    CSRStats stats{};
    // merge-path partition: first row of every tile, built on first use
    int*   d_tile_rows = nullptr;    // [num_tiles + 1]
    int    num_tiles = 0;
    int    tile_items = 0;
    // merge-path carry-out slots: written by every call, so they belong to the first stream that uses them;
    // a call on another stream gets its own pair (extra_carry), like the tiled plan's product stream
    int*   d_carry_row = nullptr;    // [num_tiles]
    float* d_carry_val = nullptr;    // [num_tiles]
    struct MergeCarry {
        hipStream_t stream;
        int* row;
        float* val;
    };
    std::mutex merge_lock;           // guards the fields below and the tile + fix-up pair of launches
    bool carry_taken = false;
    hipStream_t carry_stream = nullptr;
    std::vector<MergeCarry> extra_carry;
    // LDS-tiled engine: bucketed copy of the entries, built on first use (tiled.h)
    std::shared_ptr<TiledPlan> tiled;
    bool tiled_failed = false;       // build failed once (e.g. out of memory): do not retry
    // spmv_csr() calls on this matrix that named a REORDERING kernel (VECTOR_CSR / MERGE_PATH) without use_texture:
    // from the call after tiled_promotion() of them on, the plan is built and they take the tiled engine too
    std::atomic<int> reorder_calls{0};
    // plans thrown away because another matrix over the same row-pointer array (the side table's key) wanted one: past
    // two, promotion stops building for this key — callers that alternate such matrices would rebuild on every call
    int plan_replacements = 0;
};

// Ski-rental promotion: the reference's own callers spell the kernel and never set use_texture
// (benchmarks/main.cu:52-56 {VECTOR_CSR, 256, false}; src/pagerank.cu:89-90), which on a large matrix costs 5x the tiled
// engine's time per call.  After this many direct calls on one eligible matrix — about one plan build's worth of time,
// the rule pagerank() applies to itself — the next call builds the plan (outside its timed region) and every later
// VECTOR_CSR / MERGE_PATH call on the matrix takes the tiled route.  0 = never (SPMV_TILED_PROMOTE=0 sets that at start).
// SCALAR_CSR and spmv_ell without use_texture keep their CPU-order contract and never promote.
int tiled_promotion();
void set_tiled_promotion(int calls);

// the matrix's tiled plan (built on first call), or nullptr when not eligible / not buildable
PlanRef tiled_plan_for(const CSRMatrix* A, hipStream_t s);
// the plan only if the matrix already holds a valid one (never builds)
PlanRef tiled_plan_if_cached(const CSRMatrix* A);

CsrAux* aux_lookup(const void* key, bool create);
void    aux_drop(const void* key);

struct EllAux {
    bool have_nnz = false;
    long long actual_nnz = 0;   // non-padding slots, counted once on the device
    std::shared_ptr<TiledPlan> tiled; // LDS-tiled engine plan built from the slabs (use_texture)
    bool tiled_failed = false;
};
PlanRef tiled_plan_for(const ELLMatrix* A, hipStream_t s);
EllAux* ell_aux_lookup(const void* key, bool create);
void    ell_aux_drop(const void* key);

// ---- debugging / tuning overrides: ONE environment variable ----
// SPMV_DEBUG="key=value,key,..." (read at every use: tests change it between plan builds).  Keys:
//   min_cols=N, min_nnz=N      size thresholds of the LDS-tiled engine (tests push small matrices through it)
//   strip=W, tile=R, item=N    plan shape overrides: strip columns (4096..32768), tile rows (multiple of 64), slots per phase-1 item
//   long_factor=N, long_cap=N  long-row limit = min(long_cap, long_factor * strips)
//   rank=plain                 ranking pass of the plan build by comparison (no stable binning)
//   place=scattered            placing pass of the plan build entry by entry (no LDS staging)
//   pr_plan_after=N            direct steps pagerank() takes before it builds a plan (default 4)
//   pr_copy=pageable           pagerank() copies its result into a pageable array (no pinned pool)
// Everything else the library reads from the environment is listed in INTEGRATION.md.
bool debug_option(const char* key, long long* value = nullptr, char* text = nullptr, size_t text_size = 0);
// the same "key=value,key" syntax for another variable's value (SPMV_MULTI_GPU); `list` may be null
bool list_option(const char* list, const char* key, long long* value = nullptr, char* text = nullptr, size_t text_size = 0);
inline long long debug_number(const char* key, long long fallback) {
    long long v = 0;
    return debug_option(key, &v) ? v : fallback;
}
bool debug_is(const char* key, const char* expected);

// ---- launch layer (kernels.hip) ----
// All return hipError_t from the launch; none synchronise.
hipError_t launch_csr_scalar(const CSRMatrix* A, const float* d_x, float* d_y, hipStream_t s);
hipError_t launch_csr_vector(const CSRMatrix* A, const float* d_x, float* d_y,
                             int lanes_per_row, hipStream_t s);
int vector_ldsx_grid(const CSRMatrix* A);      // 0 = x-in-LDS variant not worthwhile for this matrix
hipError_t launch_csr_vector_ldsx(const CSRMatrix* A, const float* d_x, float* d_y, int lanes_per_row,
                                  int grid, hipStream_t s);
hipError_t launch_csr_merge(const CSRMatrix* A, CsrAux* aux, const float* d_x, float* d_y,
                            hipStream_t s);
hipError_t prepare_csr_merge(const CSRMatrix* A, CsrAux* aux, hipStream_t s);   // the merge tile table, ahead of a timed call
hipError_t launch_ell(const ELLMatrix* A, const float* d_x, float* d_y, hipStream_t s);
hipError_t launch_fill_zero(float* d_y, size_t n, hipStream_t s);
hipError_t launch_ell_from_csr(const CSRMatrix* csr, int width, int* d_ell_cols, float* d_ell_vals,
                               hipStream_t s);
hipError_t device_count_ell_nnz(const ELLMatrix* A, long long* out, hipStream_t s);
hipError_t device_row_stats(const int* d_row_ptrs, int num_rows, int* max_out, int* min_out,
                            hipStream_t s);

// lanes-per-row choice for VECTOR_CSR from the mean row length (wave64 tuning)
int pick_lanes_per_row(float avg_nnz_per_row);

// timing helper: a cached event pair per thread
struct EventPair {
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
};
EventPair& thread_events();

hipStream_t current_stream();

// hipMalloc that is also legal while some stream of this thread is being captured into a hipGraph (the
// thread's capture mode is switched to "relaxed" around the call): per-stream scratch is allocated at a
// stream's first call on a matrix, and that first call may well be the one a graph capture records.
hipError_t malloc_any_time(void** ptr, size_t bytes);

// roctx range around a host-side phase (plan build, dispatch, PageRank step): shows up under
// `rocprofv3 --marker-trace`, costs two indirect calls when a profiler has loaded the roctx library and
// nothing measurable otherwise (the library is looked up once, lazily; no link-time dependency).
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
private:
    bool open_;
};

} // namespace detail
} // namespace spmv

#endif
