// capi.cpp — the extern "C" boundary declared in include/spmv_c.h: thin shims
// over the C++ API in namespace spmv (same struct layouts, checked below).
#include "internal.h"
#include "generators.h"
#include "pagerank_engine.h"
#include "tiled.h"
#include "spmv/bandwidth.h"
#include "spmv/pagerank.h"
#include "spmv_c.h"

#include <cstddef>
#include <cstring>
#include <vector>
#include <new>

using namespace spmv;

// ---- layout contracts: a C struct and its C++ twin must be byte-identical ----
static_assert(sizeof(spmv_c_csr) == sizeof(CSRMatrix) && sizeof(CSRMatrix) == 72, "CSRMatrix layout");
static_assert(offsetof(spmv_c_csr, d_values) == offsetof(CSRMatrix, d_values), "CSRMatrix layout");
static_assert(offsetof(spmv_c_csr, owns_host_memory) == offsetof(CSRMatrix, owns_host_memory), "CSRMatrix layout");
static_assert(offsetof(spmv_c_csr, owns_device_memory) == offsetof(CSRMatrix, owns_device_memory), "CSRMatrix layout");
static_assert(sizeof(spmv_c_ell) == sizeof(ELLMatrix) && sizeof(ELLMatrix) == 56, "ELLMatrix layout");
static_assert(offsetof(spmv_c_ell, d_col_indices) == offsetof(ELLMatrix, d_col_indices), "ELLMatrix layout");
static_assert(offsetof(spmv_c_ell, owns_host_memory) == offsetof(ELLMatrix, owns_host_memory), "ELLMatrix layout");
static_assert(sizeof(spmv_c_config) == sizeof(SpMVConfig) && sizeof(SpMVConfig) == 12, "SpMVConfig layout");
static_assert(offsetof(spmv_c_config, use_texture) == offsetof(SpMVConfig, use_texture), "SpMVConfig layout");
static_assert(sizeof(spmv_c_result) == sizeof(SpMVResult) && sizeof(SpMVResult) == 24, "SpMVResult layout");
static_assert(sizeof(spmv_c_csr_stats) == sizeof(CSRStats) && sizeof(CSRStats) == 16, "CSRStats layout");
static_assert(sizeof(spmv_c_bandwidth) == sizeof(BandwidthMetrics) && sizeof(BandwidthMetrics) == 12, "BandwidthMetrics layout");
static_assert(sizeof(spmv_c_pagerank_config) == sizeof(PageRankConfig) && sizeof(PageRankConfig) == 12, "PageRankConfig layout");
static_assert(sizeof(spmv_c_pagerank_result) == sizeof(PageRankResult) && sizeof(PageRankResult) == 24, "PageRankResult layout");
static_assert(offsetof(spmv_c_pagerank_result, converged) == offsetof(PageRankResult, converged), "PageRankResult layout");
static_assert(sizeof(spmv_c_topk_node) == sizeof(TopKNode) && sizeof(TopKNode) == 8, "TopKNode layout");
static_assert(sizeof(spmv_c_pr_status) == sizeof(detail::PrState), "PrState layout");

namespace {

inline CSRMatrix* cxx(spmv_c_csr* m) { return reinterpret_cast<CSRMatrix*>(m); }
inline const CSRMatrix* cxx(const spmv_c_csr* m) { return reinterpret_cast<const CSRMatrix*>(m); }
inline ELLMatrix* cxx(spmv_c_ell* m) { return reinterpret_cast<ELLMatrix*>(m); }
inline const ELLMatrix* cxx(const spmv_c_ell* m) { return reinterpret_cast<const ELLMatrix*>(m); }
inline const SpMVConfig* cxx(const spmv_c_config* c) { return reinterpret_cast<const SpMVConfig*>(c); }
inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

constexpr int kInvalidArgument = static_cast<int>(SpMVError::INVALID_ARGUMENT);
constexpr int kLaunch = static_cast<int>(SpMVError::KERNEL_LAUNCH);

inline int launch_code(hipError_t e) { return e == hipSuccess ? 0 : kLaunch; }

} // namespace

struct spmv_c_pr_shard {
    detail::PrShard shard;
};

extern "C" {

const char* spmv_c_error_string(int code) {
    return spmv_error_string(static_cast<SpMVError>(code));
}

const char* spmv_c_version(void) { return "spmv-amd 0.1 (gfx950)"; }

int spmv_c_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int spmv_c_device_name(char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return kInvalidArgument;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        buf[0] = '\0';
        return kLaunch;
    }
    std::strncpy(buf, prop.gcnArchName, buf_len - 1);
    buf[buf_len - 1] = '\0';
    return 0;
}

int spmv_c_set_device(int ordinal) { return launch_code(hipSetDevice(ordinal)); }

int spmv_c_enable_peer_access(int peer_ordinal) {
    int current = 0;
    if (hipGetDevice(&current) != hipSuccess) return kLaunch;
    if (current == peer_ordinal) return 0;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, current, peer_ordinal) != hipSuccess || !can) {
        (void)hipGetLastError();
        return kInvalidArgument;
    }
    const hipError_t e = hipDeviceEnablePeerAccess(peer_ordinal, 0);
    if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        return 0;
    }
    return kLaunch;
}

void spmv_c_set_stream(void* hip_stream) { spmv_set_stream(as_stream(hip_stream)); }

int spmv_c_device_malloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return kInvalidArgument;
    *d_ptr = nullptr;
    if (bytes == 0) return 0;
    return hipMalloc(d_ptr, bytes) == hipSuccess ? 0 : static_cast<int>(SpMVError::CUDA_MALLOC);
}

int spmv_c_device_free(void* d_ptr) {
    if (!d_ptr) return 0;
    return hipFree(d_ptr) == hipSuccess ? 0 : static_cast<int>(SpMVError::CUDA_MALLOC);
}

int spmv_c_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return 0;
    if (!d_dst || !src) return kInvalidArgument;
    return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess
         ? 0 : static_cast<int>(SpMVError::CUDA_MEMCPY);
}

int spmv_c_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return 0;
    if (!dst || !d_src) return kInvalidArgument;
    return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess
         ? 0 : static_cast<int>(SpMVError::CUDA_MEMCPY);
}

int spmv_c_device_synchronize(void) { return launch_code(hipDeviceSynchronize()); }

static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");

int spmv_c_ipc_get_handle(void* d_ptr, unsigned char handle_out[64]) {
    if (!d_ptr || !handle_out) return kInvalidArgument;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, d_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return kLaunch;
    }
    std::memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int spmv_c_ipc_open_handle(const unsigned char handle[64], void** d_ptr_out) {
    if (!handle || !d_ptr_out) return kInvalidArgument;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    *d_ptr_out = nullptr;
    if (hipIpcOpenMemHandle(d_ptr_out, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        *d_ptr_out = nullptr;
        return kLaunch;
    }
    return 0;
}

int spmv_c_ipc_close(void* d_ptr) {
    if (!d_ptr) return 0;
    return launch_code(hipIpcCloseMemHandle(d_ptr));
}

// ---- CSR ----
spmv_c_csr* spmv_c_csr_create(int rows, int cols, int nnz) {
    return reinterpret_cast<spmv_c_csr*>(csr_create(rows, cols, nnz));
}
void spmv_c_csr_destroy(spmv_c_csr* mat) { csr_destroy(cxx(mat)); }
int spmv_c_csr_from_dense(spmv_c_csr* csr, const float* dense, int rows, int cols) {
    return csr_from_dense(cxx(csr), dense, rows, cols);
}
int spmv_c_csr_to_dense(const spmv_c_csr* csr, float* dense) { return csr_to_dense(cxx(csr), dense); }
float spmv_c_csr_get_element(const spmv_c_csr* mat, int row, int col) {
    return csr_get_element(cxx(mat), row, col);
}
int spmv_c_csr_to_gpu(spmv_c_csr* mat) { return csr_to_gpu(cxx(mat)); }
int spmv_c_csr_from_gpu(spmv_c_csr* mat) { return csr_from_gpu(cxx(mat)); }
void spmv_c_csr_free_gpu(spmv_c_csr* mat) { csr_free_gpu(cxx(mat)); }
void spmv_c_csr_invalidate_gpu_cache(const spmv_c_csr* mat) { csr_invalidate_gpu_cache(cxx(mat)); }
int spmv_c_csr_serialize(const spmv_c_csr* mat, const char* filename) {
    return csr_serialize(cxx(mat), filename);
}
int spmv_c_csr_deserialize(spmv_c_csr* mat, const char* filename) {
    return csr_deserialize(cxx(mat), filename);
}
int spmv_c_csr_compute_stats(const spmv_c_csr* mat, spmv_c_csr_stats* out) {
    if (!out) return kInvalidArgument;
    const CSRStats s = csr_compute_stats(cxx(mat));
    std::memcpy(out, &s, sizeof(s));
    return 0;
}

spmv_c_csr* spmv_c_csr_wrap_device(int rows, int cols, int nnz, const int32_t* d_row_ptrs,
                                   const int32_t* d_col_indices, const float* d_values) {
    if (rows < 0 || cols < 0 || nnz < 0) return nullptr;
    CSRMatrix* m = new (std::nothrow) CSRMatrix{};
    if (!m) return nullptr;
    m->num_rows = rows;
    m->num_cols = cols;
    m->nnz = nnz;
    m->d_row_ptrs = const_cast<int*>(d_row_ptrs);
    m->d_col_indices = const_cast<int*>(d_col_indices);
    m->d_values = const_cast<float*>(d_values);
    m->owns_host_memory = false;
    m->owns_device_memory = false;
    return reinterpret_cast<spmv_c_csr*>(m);
}

// ---- ELL ----
spmv_c_ell* spmv_c_ell_create(int rows, int cols, int k) {
    return reinterpret_cast<spmv_c_ell*>(ell_create(rows, cols, k));
}
void spmv_c_ell_destroy(spmv_c_ell* mat) { ell_destroy(cxx(mat)); }
int spmv_c_ell_from_dense(spmv_c_ell* ell, const float* dense, int rows, int cols) {
    return ell_from_dense(cxx(ell), dense, rows, cols);
}
int spmv_c_ell_from_csr(spmv_c_ell* ell, const spmv_c_csr* csr) { return ell_from_csr(cxx(ell), cxx(csr)); }
int spmv_c_ell_from_csr_gpu(spmv_c_ell* ell, const spmv_c_csr* csr) { return ell_from_csr_gpu(cxx(ell), cxx(csr)); }
int spmv_c_ell_to_dense(const spmv_c_ell* ell, float* dense) { return ell_to_dense(cxx(ell), dense); }
float spmv_c_ell_get_element(const spmv_c_ell* mat, int row, int col) {
    return ell_get_element(cxx(mat), row, col);
}
int spmv_c_ell_to_gpu(spmv_c_ell* mat) { return ell_to_gpu(cxx(mat)); }
int spmv_c_ell_from_gpu(spmv_c_ell* mat) { return ell_from_gpu(cxx(mat)); }
void spmv_c_ell_free_gpu(spmv_c_ell* mat) { ell_free_gpu(cxx(mat)); }
void spmv_c_ell_invalidate_gpu_cache(const spmv_c_ell* mat) { ell_invalidate_gpu_cache(cxx(mat)); }
int spmv_c_ell_serialize(const spmv_c_ell* mat, const char* filename) {
    return ell_serialize(cxx(mat), filename);
}
int spmv_c_ell_deserialize(spmv_c_ell* mat, const char* filename) {
    return ell_deserialize(cxx(mat), filename);
}
int spmv_c_ell_index(int row, int k, int num_rows) { return ell_index(row, k, num_rows); }

spmv_c_ell* spmv_c_ell_wrap_device(int rows, int cols, int k, const int32_t* d_col_indices,
                                   const float* d_values) {
    if (rows < 0 || cols < 0 || k < 0) return nullptr;
    ELLMatrix* m = new (std::nothrow) ELLMatrix{};
    if (!m) return nullptr;
    m->num_rows = rows;
    m->num_cols = cols;
    m->max_nnz_per_row = k;
    m->d_col_indices = const_cast<int*>(d_col_indices);
    m->d_values = const_cast<float*>(d_values);
    m->owns_host_memory = false;
    m->owns_device_memory = false;
    return reinterpret_cast<spmv_c_ell*>(m);
}

// ---- SpMV ----
void spmv_c_cpu_csr(const spmv_c_csr* A, const float* x, float* y) { spmv_cpu_csr(cxx(A), x, y); }
void spmv_c_cpu_ell(const spmv_c_ell* A, const float* x, float* y) { spmv_cpu_ell(cxx(A), x, y); }

int spmv_c_spmv_csr(const spmv_c_csr* A, const float* d_x, float* d_y,
                    const spmv_c_config* config, int vec_size, spmv_c_result* out) {
    const SpMVResult r = spmv_csr(cxx(A), d_x, d_y, cxx(config), vec_size);
    if (out) std::memcpy(out, &r, sizeof(r));
    return r.error_code;
}

int spmv_c_spmv_ell(const spmv_c_ell* A, const float* d_x, float* d_y,
                    const spmv_c_config* config, int vec_size, spmv_c_result* out) {
    const SpMVResult r = spmv_ell(cxx(A), d_x, d_y, cxx(config), vec_size);
    if (out) std::memcpy(out, &r, sizeof(r));
    return r.error_code;
}

int spmv_c_auto_config(const spmv_c_csr* A, spmv_c_config* out) {
    if (!A || !out) return kInvalidArgument;
    const SpMVConfig c = spmv_auto_config(cxx(A));
    std::memset(out, 0, sizeof(*out));
    out->kernel_type = static_cast<int32_t>(c.kernel_type);
    out->block_size = c.block_size;
    out->use_texture = c.use_texture ? 1 : 0;
    return 0;
}

int spmv_c_validate_dimensions(int num_cols, int vec_size) {
    return spmv_validate_dimensions(num_cols, vec_size) ? 1 : 0;
}

void spmv_c_set_tiled_promotion(int calls) { spmv_set_tiled_promotion(calls); }
int spmv_c_get_tiled_promotion(void) { return spmv_get_tiled_promotion(); }

int spmv_c_csr_has_tiled_plan(const spmv_c_csr* A_c) {
    const CSRMatrix* A = cxx(A_c);
    if (!A || !A->d_row_ptrs) return 0;
    detail::CsrAux* aux = detail::aux_lookup(A->d_row_ptrs, false);
    return aux && aux->tiled ? 1 : 0;
}

int spmv_c_tiled_shape(int64_t rows, int64_t cols, int64_t nnz, int32_t* strip_cols, int32_t* tile_rows) {
    int w = 0, r = 0;
    const bool takes = detail::tiled_shape_for(rows, cols, nnz, &w, &r);
    if (strip_cols) *strip_cols = w;
    if (tile_rows) *tile_rows = r;
    return takes ? 1 : 0;
}

int spmv_c_csr_tiled_info(const spmv_c_csr* A_c, int64_t out[8]) {
    const CSRMatrix* A = cxx(A_c);
    if (!A || !A->d_row_ptrs || !out) return 0;
    detail::CsrAux* aux = detail::aux_lookup(A->d_row_ptrs, false);
    if (!aux || !aux->tiled) return 0;
    const detail::TiledPlan& p = *aux->tiled;
    const int64_t v[8] = {p.strip_cols, p.tile_rows, p.num_strips, p.num_tiles, p.nnz, p.num_long, 4 /* slots per lane per phase-2 pass */,
                          p.long_row};
    std::memcpy(out, v, sizeof(v));
    return 1;
}

int spmv_c_csr_tiled_stats(const spmv_c_csr* A_c, double out[4]) {
    const CSRMatrix* A = cxx(A_c);
    if (!A || !A->d_row_ptrs || !out) return 0;
    detail::CsrAux* aux = detail::aux_lookup(A->d_row_ptrs, false);
    if (!aux || !aux->tiled) return 0;
    const detail::TiledPlan& p = *aux->tiled;
    out[0] = p.build_ms;
    out[1] = static_cast<double>(p.plan_bytes);
    out[2] = static_cast<double>(p.nnz);
    out[3] = static_cast<double>(p.entries);
    return 1;
}

int spmv_c_csr_tiled_checksum(const spmv_c_csr* A_c, uint64_t out[4]) {
    const CSRMatrix* A = cxx(A_c);
    if (!A || !A->d_row_ptrs || !out) return 0;
    const detail::PlanRef plan = detail::tiled_plan_if_cached(A);
    if (!plan) return 0;
    unsigned long long sums[4] = {0, 0, 0, 0};
    if (detail::tiled_checksum(*plan, sums, detail::current_stream()) != hipSuccess) return 0;
    for (int i = 0; i < 4; ++i) out[i] = sums[i];
    return 1;
}

int spmv_c_csr_tiled_folded(const spmv_c_csr* A_c) {
    const CSRMatrix* A = cxx(A_c);
    if (!A || !A->d_row_ptrs) return 0;
    detail::CsrAux* aux = detail::aux_lookup(A->d_row_ptrs, false);
    return aux && aux->tiled && aux->tiled->col_weight ? 1 : 0;
}

int spmv_c_spmv_csr_async(const spmv_c_csr* A, const float* d_x, float* d_y,
                          const spmv_c_config* config, int vec_size, void* hip_stream) {
    return spmv_csr_async(cxx(A), d_x, d_y, cxx(config), vec_size, as_stream(hip_stream));
}

int spmv_c_spmv_ell_async(const spmv_c_ell* A, const float* d_x, float* d_y,
                          const spmv_c_config* config, int vec_size, void* hip_stream) {
    return spmv_ell_async(cxx(A), d_x, d_y, cxx(config), vec_size, as_stream(hip_stream));
}

// ---- bandwidth ----
int spmv_c_compute_bandwidth_csr(const spmv_c_csr* A, float elapsed_ms, spmv_c_bandwidth* out) {
    if (!out) return kInvalidArgument;
    const BandwidthMetrics m = compute_bandwidth_csr(cxx(A), elapsed_ms);
    std::memcpy(out, &m, sizeof(m));
    return 0;
}
int spmv_c_compute_bandwidth_ell(const spmv_c_ell* A, float elapsed_ms, spmv_c_bandwidth* out) {
    if (!out) return kInvalidArgument;
    const BandwidthMetrics m = compute_bandwidth_ell(cxx(A), elapsed_ms);
    std::memcpy(out, &m, sizeof(m));
    return 0;
}
float spmv_c_get_gpu_peak_bandwidth(void) { return get_gpu_peak_bandwidth(); }

// ---- PageRank ----
int spmv_c_pagerank(const spmv_c_csr* adj, const spmv_c_pagerank_config* config,
                    spmv_c_pagerank_result* out) {
    if (!out) return kInvalidArgument;
    const PageRankResult r = pagerank(cxx(adj), reinterpret_cast<const PageRankConfig*>(config));
    std::memset(out, 0, sizeof(*out));
    out->ranks = r.ranks;
    out->iterations = r.iterations;
    out->final_residual = r.final_residual;
    out->converged = r.converged ? 1 : 0;
    return adj ? 0 : kInvalidArgument;
}

int spmv_c_pagerank_multi_gpu(const spmv_c_csr* adj, const spmv_c_pagerank_config* config, int num_gpus,
                              spmv_c_pagerank_result* out) {
    if (!out) return kInvalidArgument;
    const PageRankResult r = pagerank_multi_gpu(cxx(adj), reinterpret_cast<const PageRankConfig*>(config), num_gpus);
    std::memset(out, 0, sizeof(*out));
    out->ranks = r.ranks;
    out->iterations = r.iterations;
    out->final_residual = r.final_residual;
    out->converged = r.converged ? 1 : 0;
    return adj ? 0 : kInvalidArgument;
}

int spmv_c_pagerank_shard_bounds(const int32_t* row_ptrs, int num_rows, int num_shards, int32_t* bounds) {
    if (!row_ptrs || !bounds || num_rows < 0 || num_shards < 1) return kInvalidArgument;
    const std::vector<int> b = pagerank_shard_bounds(row_ptrs, num_rows, num_shards);
    std::copy(b.begin(), b.end(), bounds);
    return 0;
}

void spmv_c_pagerank_free(spmv_c_pagerank_result* result) {
    pagerank_free(reinterpret_cast<PageRankResult*>(result));
}

void spmv_c_pagerank_top_k(const spmv_c_pagerank_result* result, int num_nodes, int k,
                           spmv_c_topk_node* top_k) {
    pagerank_top_k(reinterpret_cast<const PageRankResult*>(result), num_nodes, k,
                   reinterpret_cast<TopKNode*>(top_k));
}

// ---- PageRank shard engine ----
spmv_c_pr_shard* spmv_c_pr_shard_create(const spmv_c_csr* A_local, int row_offset, int n_global,
                                        const uint8_t* d_dangling_mask) {
    return spmv_c_pr_shard_create_chunked(A_local, row_offset, 0x7fffffff, 0, n_global, d_dangling_mask);
}

spmv_c_pr_shard* spmv_c_pr_shard_create_chunked(const spmv_c_csr* A_local, int base, int piece, int block,
                                                int n_global, const uint8_t* d_dangling_mask) {
    const CSRMatrix* A = cxx(A_local);
    // the rank vector is indexed by column (num_cols long, possibly padded); n_global is the true node count
    if (!A || base < 0 || piece <= 0 || block < 0 || n_global <= 0 || !d_dangling_mask || n_global > A->num_cols ||
        (A->num_rows > 0 && !A->d_row_ptrs) ||
        (A->nnz > 0 && (!A->d_col_indices || !A->d_values))) {
        return nullptr;
    }
    detail::RowMap map;
    map.base = base;
    map.piece = piece;
    map.block = block;
    // every local row must land inside the vector, pieces must not overlap
    if (A->num_rows > 0) {
        const bool chunked = piece != 0x7fffffff;
        if (chunked && block < piece) return nullptr;
        const long long pieces = chunked ? (static_cast<long long>(A->num_rows) + piece - 1) / piece : 1;
        const long long last = map.at(static_cast<long long>(A->num_rows) - 1);
        if (last >= A->num_cols || (pieces > 1 && static_cast<long long>(base) + piece > block)) return nullptr;
    }
    spmv_c_pr_shard* h = new (std::nothrow) spmv_c_pr_shard();
    if (!h) return nullptr;
    detail::PrShard& sh = h->shard;
    sh.local_rows = A->num_rows;
    sh.map = map;
    sh.n_global = n_global;
    sh.nnz = A->nnz;
    sh.d_row_ptrs = A->d_row_ptrs;
    sh.d_cols = A->d_col_indices;
    sh.d_vals = A->d_values;
    sh.d_dangling = d_dangling_mask;
    const int partial_pairs = detail::pr_shard_prepare(&sh, detail::tiled_plan_for(A, nullptr));
    if (hipMalloc(reinterpret_cast<void**>(&sh.d_state), sizeof(detail::PrState)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&sh.d_block_partials),
                  2 * sizeof(double) * static_cast<size_t>(partial_pairs)) != hipSuccess) {
        spmv_c_pr_shard_destroy(h);
        return nullptr;
    }
    return h;
}

void spmv_c_pr_shard_destroy(spmv_c_pr_shard* h) {
    if (!h) return;
    if (h->shard.d_state) (void)hipFree(h->shard.d_state);
    if (h->shard.d_block_partials) (void)hipFree(h->shard.d_block_partials);
    delete h;
}

int spmv_c_pr_reset(spmv_c_pr_shard* h, float dangling_sum, void* hip_stream) {
    if (!h) return kInvalidArgument;
    detail::PrState fresh{};
    fresh.dangling_sum = dangling_sum;
    h->shard.expanded_strips = 0;          // a head start taken for a step that never ran is void
    h->shard.expanded_long = false;
    // pageable source: the copy is staged before the call returns
    return hipMemcpyAsync(h->shard.d_state, &fresh, sizeof(fresh), hipMemcpyHostToDevice,
                          as_stream(hip_stream)) == hipSuccess
         ? 0 : static_cast<int>(SpMVError::CUDA_MEMCPY);
}

int spmv_c_pr_step(spmv_c_pr_shard* h, const float* d_r_old, float* d_r_new, float damping,
                   void* hip_stream) {
    if (!h || !d_r_old || !d_r_new) return kInvalidArgument;
    return launch_code(detail::pr_step(h->shard, d_r_old, d_r_new, damping, detail::PushTargets{},
                                       as_stream(hip_stream)));
}

int spmv_c_pr_expand(spmv_c_pr_shard* h, const float* d_r_old, int64_t cols_ready, void* hip_stream) {
    if (!h || !d_r_old) return kInvalidArgument;
    return launch_code(detail::pr_expand(h->shard, d_r_old, cols_ready, as_stream(hip_stream)));
}

int spmv_c_pr_step_push(spmv_c_pr_shard* h, const float* d_r_old, float* d_r_new, float damping,
                        float* const* peer_r_new, int num_peers, void* hip_stream) {
    if (!h || !d_r_old || !d_r_new || num_peers < 0 || num_peers > detail::kMaxPushPeers ||
        (num_peers > 0 && !peer_r_new)) {
        return kInvalidArgument;
    }
    detail::PushTargets push;
    for (int p = 0; p < num_peers; ++p) {
        if (!peer_r_new[p]) return kInvalidArgument;
        push.ptr[push.count++] = peer_r_new[p];
    }
    return launch_code(detail::pr_step(h->shard, d_r_old, d_r_new, damping, push, as_stream(hip_stream)));
}

int spmv_c_pr_reduce(spmv_c_pr_shard* h, double* d_sums, void* hip_stream) {
    if (!h || !d_sums) return kInvalidArgument;
    return launch_code(detail::pr_reduce(h->shard, d_sums, as_stream(hip_stream)));
}

int spmv_c_pr_commit(spmv_c_pr_shard* h, const double* d_sums, float tolerance, void* hip_stream) {
    if (!h || !d_sums) return kInvalidArgument;
    return launch_code(detail::pr_commit(h->shard, d_sums, tolerance, as_stream(hip_stream)));
}

int spmv_c_pr_reduce_commit(spmv_c_pr_shard* h, float tolerance, void* hip_stream) {
    if (!h) return kInvalidArgument;
    return launch_code(detail::pr_reduce_commit(h->shard, tolerance, as_stream(hip_stream)));
}

int spmv_c_pr_commit_gathered(spmv_c_pr_shard* h, const float* d_gathered, int world, int64_t stride,
                              int64_t shard_len, float tolerance, void* hip_stream) {
    if (!h || !d_gathered || world < 1 || stride < shard_len + 4 || (stride & 1) || (shard_len & 1)) {
        return kInvalidArgument;
    }
    return launch_code(detail::pr_commit_gathered(h->shard, d_gathered, world, stride, shard_len, tolerance,
                                                  as_stream(hip_stream)));
}

int spmv_c_pr_status_get(spmv_c_pr_shard* h, spmv_c_pr_status* out, void* hip_stream) {
    if (!h || !out) return kInvalidArgument;
    hipStream_t s = as_stream(hip_stream);
    if (hipMemcpyAsync(out, h->shard.d_state, sizeof(*out), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        return static_cast<int>(SpMVError::CUDA_MEMCPY);
    }
    return 0;
}

int spmv_c_pr_column_sums(const spmv_c_csr* A_local, float* d_col_sums, void* hip_stream) {
    const CSRMatrix* A = cxx(A_local);
    if (!A || !d_col_sums) return kInvalidArgument;
    return launch_code(detail::pr_column_sums(A->nnz, A->d_col_indices, A->d_values, A->num_cols,
                                              d_col_sums, as_stream(hip_stream)));
}

int spmv_c_pr_mask_from_column_sums(const float* d_col_sums, int n, uint8_t* d_mask,
                                    uint64_t* d_count, void* hip_stream) {
    if (!d_col_sums || !d_mask || !d_count || n < 0) return kInvalidArgument;
    return launch_code(detail::pr_mask_from_column_sums(
        d_col_sums, n, d_mask, reinterpret_cast<unsigned long long*>(d_count), as_stream(hip_stream)));
}

int spmv_c_fill(float* d_r, size_t n, float value, void* hip_stream) {
    if (n > 0 && !d_r) return kInvalidArgument;
    return launch_code(detail::pr_fill(d_r, n, value, as_stream(hip_stream)));
}

// ---- generators ----
int spmv_c_gen_uniform_rows(uint64_t seed, int row_begin, int local_rows, int n_cols, int k,
                            int32_t* d_row_ptrs, int32_t* d_cols, float* d_vals, void* hip_stream) {
    return detail::gen_uniform_rows(seed, row_begin, local_rows, n_cols, k, d_row_ptrs, d_cols,
                                    d_vals, as_stream(hip_stream));
}

int spmv_c_gen_uniform_ell(uint64_t seed, int rows, int n_cols, int k, int32_t* d_cols, float* d_vals,
                           void* hip_stream) {
    return detail::gen_uniform_ell(seed, rows, n_cols, k, d_cols, d_vals, as_stream(hip_stream));
}

int spmv_c_gen_stratified_rows(uint64_t seed, int row_begin, int local_rows, int n_cols,
                               const int32_t* d_row_ptrs, int32_t* d_cols, float* d_vals,
                               void* hip_stream) {
    return detail::gen_stratified_rows(seed, row_begin, local_rows, n_cols, d_row_ptrs, d_cols,
                                       d_vals, as_stream(hip_stream));
}

int spmv_c_gen_vector(uint64_t seed, uint64_t tag, size_t n, float* d_x, void* hip_stream) {
    return detail::gen_vector(seed, tag, n, d_x, as_stream(hip_stream));
}

int spmv_c_count_columns(int64_t nnz, const int32_t* d_cols, int n_cols, int32_t* d_counts,
                         void* hip_stream) {
    return detail::count_columns(nnz, d_cols, n_cols, d_counts, as_stream(hip_stream));
}

int spmv_c_reciprocal_values(int64_t nnz, const int32_t* d_cols, const int32_t* d_counts,
                             float* d_vals, void* hip_stream) {
    return detail::reciprocal_values(nnz, d_cols, d_counts, d_vals, as_stream(hip_stream));
}

} // extern "C"
