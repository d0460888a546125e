/* spmv_c.h — C-ABI boundary of the MI355X-native SpMV library (libspmv_amd.so).
 *
 * The reference (LessUp/gpu-spmv) is a C++ library: free functions in
 * `namespace spmv` over plain structs (its include/spmv headers).  This header exports
 * the same entry points with C linkage — plain pointers, sizes and POD structs,
 * no C++ or torch types — so any FFI (ctypes, cgo, JNI, N-API) can bind them.
 * Every struct below has the byte layout of the reference's C++ struct of the
 * same name (x86-64 SysV), so a `spmv::CSRMatrix*` and a `spmv_c_csr*` are
 * interchangeable.  Each declaration cites the reference interface it replaces.
 *
 * Functions that return `int` return a spmv error code (0 = success, negative
 * values as reference include/spmv/common.h:13-23).  Results that the C++ API
 * returns by value come back through an `out` pointer.
 * Device pointers (`d_*`) are addresses in the current HIP device's memory.
 */
#ifndef SPMV_C_H
#define SPMV_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: reference include/spmv/common.h:13-23 ---- */
enum {
    SPMV_C_SUCCESS = 0,
    SPMV_C_INVALID_DIMENSION = -1,
    SPMV_C_DEVICE_MALLOC = -2,
    SPMV_C_DEVICE_MEMCPY = -3,
    SPMV_C_KERNEL_LAUNCH = -4,
    SPMV_C_INVALID_FORMAT = -5,
    SPMV_C_FILE_IO = -6,
    SPMV_C_OUT_OF_MEMORY = -7,
    SPMV_C_INVALID_ARGUMENT = -8
};
/* reference include/spmv/common.h:26-39 (spmv_error_string) */
const char* spmv_c_error_string(int code);

/* ---- structs ---- */

/* reference include/spmv/csr_matrix.h:11-28 (72 bytes) */
typedef struct spmv_c_csr {
    int32_t num_rows, num_cols, nnz;
    float*   values;
    int32_t* col_indices;
    int32_t* row_ptrs;
    float*   d_values;
    int32_t* d_col_indices;
    int32_t* d_row_ptrs;
    uint8_t  owns_host_memory;
    uint8_t  owns_device_memory;
} spmv_c_csr;

/* reference include/spmv/ell_matrix.h:12-28 (56 bytes) */
typedef struct spmv_c_ell {
    int32_t num_rows, num_cols, max_nnz_per_row;
    float*   values;
    int32_t* col_indices;
    float*   d_values;
    int32_t* d_col_indices;
    uint8_t  owns_host_memory;
    uint8_t  owns_device_memory;
} spmv_c_ell;

/* reference include/spmv/csr_matrix.h:64-69 */
typedef struct spmv_c_csr_stats {
    float   avg_nnz_per_row;
    int32_t max_nnz_per_row;
    int32_t min_nnz_per_row;
    float   skewness;
} spmv_c_csr_stats;

/* reference include/spmv/spmv.h:11-24; kernel_type: 0 SCALAR_CSR, 1 VECTOR_CSR,
 * 2 MERGE_PATH, 3 ELL_KERNEL (12 bytes) */
typedef struct spmv_c_config {
    int32_t kernel_type;
    int32_t block_size;
    uint8_t use_texture;
} spmv_c_config;

/* reference include/spmv/spmv.h:27-36 (24 bytes) */
typedef struct spmv_c_result {
    float*  y;
    float   elapsed_ms;
    float   gflops;
    float   bandwidth_gb_s;
    int32_t error_code;
} spmv_c_result;

/* reference include/spmv/bandwidth.h:10-18 */
typedef struct spmv_c_bandwidth {
    float theoretical_bandwidth_gb_s;
    float achieved_bandwidth_gb_s;
    float efficiency;
} spmv_c_bandwidth;

/* reference include/spmv/pagerank.h:9-15 */
typedef struct spmv_c_pagerank_config {
    float   damping_factor;
    float   tolerance;
    int32_t max_iterations;
} spmv_c_pagerank_config;

/* reference include/spmv/pagerank.h:18-25 (24 bytes) */
typedef struct spmv_c_pagerank_result {
    float*  ranks;
    int32_t iterations;
    float   final_residual;
    uint8_t converged;
} spmv_c_pagerank_result;

/* reference include/spmv/pagerank.h:38-41 */
typedef struct spmv_c_topk_node {
    int32_t node_id;
    float   rank;
} spmv_c_topk_node;

/* ---- library / device ---- */
const char* spmv_c_version(void);
int spmv_c_device_count(void);                       /* 0 when no HIP device is visible */
int spmv_c_device_name(char* buf, size_t buf_len);   /* gcnArchName of the current device */
int spmv_c_set_device(int ordinal);
/* lets kernels of the current device store into memory of `peer_ordinal` (xGMI / PCIe peer access) */
int spmv_c_enable_peer_access(int peer_ordinal);
/* stream used by the synchronous entry points of the calling thread (NULL = null stream) */
void spmv_c_set_stream(void* hip_stream);

/* ---- device buffers: reference include/spmv/cuda_buffer.h:12-101 (CudaBuffer<T>) ---- */
int spmv_c_device_malloc(void** d_ptr, size_t bytes);            /* ctor / resize */
int spmv_c_device_free(void* d_ptr);                             /* dtor / release */
int spmv_c_memcpy_h2d(void* d_dst, const void* src, size_t bytes);   /* copyFromHost */
int spmv_c_memcpy_d2h(void* dst, const void* d_src, size_t bytes);   /* copyToHost */
int spmv_c_device_synchronize(void);
/* extension: share a spmv_c_device_malloc'ed buffer with another process on the same node
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle); handles are 64 opaque bytes.  The opener maps the
 * buffer for ITS current device, so its kernels may load/store it directly (over xGMI when the
 * buffer lives on another GPU). */
int spmv_c_ipc_get_handle(void* d_ptr, unsigned char handle_out[64]);
int spmv_c_ipc_open_handle(const unsigned char handle[64], void** d_ptr_out);
int spmv_c_ipc_close(void* d_ptr);

/* ---- CSR container: reference include/spmv/csr_matrix.h:31-71 ---- */
spmv_c_csr* spmv_c_csr_create(int rows, int cols, int nnz);
void spmv_c_csr_destroy(spmv_c_csr* mat);
int spmv_c_csr_from_dense(spmv_c_csr* csr, const float* dense, int rows, int cols);
int spmv_c_csr_to_dense(const spmv_c_csr* csr, float* dense);
float spmv_c_csr_get_element(const spmv_c_csr* mat, int row, int col);
int spmv_c_csr_to_gpu(spmv_c_csr* mat);
int spmv_c_csr_from_gpu(spmv_c_csr* mat);
void spmv_c_csr_free_gpu(spmv_c_csr* mat);
/* extension: forget the cached auxiliary data after the device arrays were modified in place */
void spmv_c_csr_invalidate_gpu_cache(const spmv_c_csr* mat);
int spmv_c_csr_serialize(const spmv_c_csr* mat, const char* filename);
int spmv_c_csr_deserialize(spmv_c_csr* mat, const char* filename);
int spmv_c_csr_compute_stats(const spmv_c_csr* mat, spmv_c_csr_stats* out);
/* extension: a matrix header over device arrays the caller owns (no host arrays) */
spmv_c_csr* spmv_c_csr_wrap_device(int rows, int cols, int nnz, const int32_t* d_row_ptrs,
                                   const int32_t* d_col_indices, const float* d_values);

/* ---- ELL container: reference include/spmv/ell_matrix.h:31-66 ---- */
spmv_c_ell* spmv_c_ell_create(int rows, int cols, int max_nnz_per_row);
void spmv_c_ell_destroy(spmv_c_ell* mat);
int spmv_c_ell_from_dense(spmv_c_ell* ell, const float* dense, int rows, int cols);
int spmv_c_ell_from_csr(spmv_c_ell* ell, const spmv_c_csr* csr);
/* extension (device-side conversion, no host pass): see ell_from_csr_gpu in spmv/ell_matrix.h */
int spmv_c_ell_from_csr_gpu(spmv_c_ell* ell, const spmv_c_csr* csr);
int spmv_c_ell_to_dense(const spmv_c_ell* ell, float* dense);
float spmv_c_ell_get_element(const spmv_c_ell* mat, int row, int col);
int spmv_c_ell_to_gpu(spmv_c_ell* mat);
int spmv_c_ell_from_gpu(spmv_c_ell* mat);
void spmv_c_ell_free_gpu(spmv_c_ell* mat);
void spmv_c_ell_invalidate_gpu_cache(const spmv_c_ell* mat);
int spmv_c_ell_serialize(const spmv_c_ell* mat, const char* filename);
int spmv_c_ell_deserialize(spmv_c_ell* mat, const char* filename);
int spmv_c_ell_index(int row, int k, int num_rows);
spmv_c_ell* spmv_c_ell_wrap_device(int rows, int cols, int max_nnz_per_row,
                                   const int32_t* d_col_indices, const float* d_values);

/* ---- SpMV: reference include/spmv/spmv.h:39-54 ---- */
void spmv_c_cpu_csr(const spmv_c_csr* A, const float* x, float* y);      /* spmv_cpu_csr */
void spmv_c_cpu_ell(const spmv_c_ell* A, const float* x, float* y);      /* spmv_cpu_ell */
/* spmv_csr / spmv_ell: config may be NULL (defaults), vec_size < 0 skips the size check;
 * the return value equals out->error_code */
int spmv_c_spmv_csr(const spmv_c_csr* A, const float* d_x, float* d_y,
                    const spmv_c_config* config, int vec_size, spmv_c_result* out);
int spmv_c_spmv_ell(const spmv_c_ell* A, const float* d_x, float* d_y,
                    const spmv_c_config* config, int vec_size, spmv_c_result* out);
int spmv_c_auto_config(const spmv_c_csr* A, spmv_c_config* out);          /* spmv_auto_config */
int spmv_c_validate_dimensions(int num_cols, int vec_size);               /* 1 = match */
/* extension (spmv::spmv_set_tiled_promotion, include/spmv/spmv.h): after `calls` spmv_csr() calls that name
 * VECTOR_CSR / MERGE_PATH without use_texture on one large matrix, later ones run on the LDS-tiled engine
 * (default 4, 0 = never; the reference's callers never set use_texture: benchmarks/main.cu:52-56) */
void spmv_c_set_tiled_promotion(int calls);
int spmv_c_get_tiled_promotion(void);
/* extension: 1 when the matrix currently holds an LDS-tiled plan (built by the first
 * use_texture call / PageRank on a large matrix), 0 otherwise */
int spmv_c_csr_has_tiled_plan(const spmv_c_csr* A);
/* extension: what the LDS-tiled engine would do with a rows x cols matrix of nnz entries: returns 1
 * when it would take it (use_texture), and the strip width / tile height it would use (host logic) */
int spmv_c_tiled_shape(int64_t rows, int64_t cols, int64_t nnz, int32_t* strip_cols, int32_t* tile_rows);
/* extension: the plan a matrix currently holds — out[8] = strip_cols, tile_rows, num_strips, num_tiles,
 * slots in cells (entries + row-skip markers + padding), long rows, slots per lane per phase-2 load,
 * long-row limit; returns 0 if none */
int spmv_c_csr_tiled_info(const spmv_c_csr* A, int64_t out[8]);
/* extension: what the plan cost — out[4] = build time in ms (host wall clock, allocations included),
 * device bytes held, slots in cells, matrix entries in cells; returns 0 if the matrix has no plan */
int spmv_c_csr_tiled_stats(const spmv_c_csr* A, double out[4]);
/* position-weighted checksums of the plan's slot arrays (values, local columns, row deltas) and its cell table:
 * two builds of one matrix give the same four numbers (the layout is a pure function of the matrix).  1 = filled. */
int spmv_c_csr_tiled_checksum(const spmv_c_csr* A, uint64_t out[4]);
/* extension: 1 when the matrix's plan folded its values into one weight per column (every stored
 * entry of a column bit-identical: adjacency / column-stochastic matrices), so that the tiled engine
 * streams no values; 0 otherwise or without a plan.  SPMV_TILED_FOLD=0 at build time disables it. */
int spmv_c_csr_tiled_folded(const spmv_c_csr* A);
/* extension: enqueue on a caller stream without timing or synchronisation */
int spmv_c_spmv_csr_async(const spmv_c_csr* A, const float* d_x, float* d_y,
                          const spmv_c_config* config, int vec_size, void* hip_stream);
int spmv_c_spmv_ell_async(const spmv_c_ell* A, const float* d_x, float* d_y,
                          const spmv_c_config* config, int vec_size, void* hip_stream);

/* ---- bandwidth model: reference include/spmv/bandwidth.h:21-27 ---- */
int spmv_c_compute_bandwidth_csr(const spmv_c_csr* A, float elapsed_ms, spmv_c_bandwidth* out);
int spmv_c_compute_bandwidth_ell(const spmv_c_ell* A, float elapsed_ms, spmv_c_bandwidth* out);
float spmv_c_get_gpu_peak_bandwidth(void);

/* ---- PageRank: reference include/spmv/pagerank.h:29-43 ---- */
int spmv_c_pagerank(const spmv_c_csr* adj, const spmv_c_pagerank_config* config,
                    spmv_c_pagerank_result* out);
void spmv_c_pagerank_free(spmv_c_pagerank_result* result);
/* extension (the reference's pagerank(), include/spmv/pagerank.h:29-31, is single-device): the same call
 * with the CSR rows sharded over num_gpus devices of this process, equal-nnz row blocks, one RCCL all-gather
 * of the rank slices per iteration.  Needs the matrix's HOST arrays.  out->ranks == NULL when fewer than
 * num_gpus devices or no librccl are available.  Returns 0, or -8 for null arguments. */
int spmv_c_pagerank_multi_gpu(const spmv_c_csr* adj, const spmv_c_pagerank_config* config, int num_gpus,
                              spmv_c_pagerank_result* out);
/* extension: the row boundaries pagerank_multi_gpu uses — bounds[num_shards + 1], binary search on the host
 * row_ptrs for equal nnz */
int spmv_c_pagerank_shard_bounds(const int32_t* row_ptrs, int num_rows, int num_shards, int32_t* bounds);
void spmv_c_pagerank_top_k(const spmv_c_pagerank_result* result, int num_nodes, int k,
                           spmv_c_topk_node* top_k);

/* ---- PageRank shard engine (extension: the row-sharded multi-GPU loop) ----
 * One rank owns A_local->num_rows consecutive rows of the n_global-node matrix (A_local: device
 * CSR, row_ptrs rebased to 0) and full-length device vectors of A_local->num_cols floats, which
 * the column indices address; its nodes sit at [row_offset, row_offset + num_rows) of those
 * vectors (num_cols may exceed n_global when slices carry padding, see pagerank_dist.py).
 * Per iteration: step -> reduce (partials into the slice tail) -> [all-gather] ->
 * commit_gathered; or with one rank: step -> reduce -> commit.  All calls enqueue on
 * `hip_stream` and return. */
typedef struct spmv_c_pr_shard spmv_c_pr_shard;   /* opaque */
typedef struct spmv_c_pr_status {                 /* device-side state, copied out */
    float   dangling_sum;
    float   final_residual;
    int32_t iterations;
    int32_t converged;
    int32_t done;
    int32_t reserved;
} spmv_c_pr_status;

spmv_c_pr_shard* spmv_c_pr_shard_create(const spmv_c_csr* A_local, int row_offset, int n_global,
                                        const uint8_t* d_dangling_mask);
/* the same for a CHUNKED vector layout (the overlapped exchange of pagerank_dist.py): local row i sits at
 * base + (i / piece) * block + i % piece — the vector is a sequence of blocks, block c holding piece c of every
 * rank back to back (base = rank * piece, block = world * piece), so that one in-place all-gather per block
 * delivers it while the blocks that have arrived are already being multiplied (spmv_c_pr_expand) */
spmv_c_pr_shard* spmv_c_pr_shard_create_chunked(const spmv_c_csr* A_local, int base, int piece, int block,
                                                int n_global, const uint8_t* d_dangling_mask);
void spmv_c_pr_shard_destroy(spmv_c_pr_shard* shard);
/* resets the iteration state; dangling_sum = dangling mass of the start vector */
int spmv_c_pr_reset(spmv_c_pr_shard* shard, float dangling_sum, void* hip_stream);
int spmv_c_pr_step(spmv_c_pr_shard* shard, const float* d_r_old, float* d_r_new, float damping,
                   void* hip_stream);
/* optional head start on the NEXT spmv_c_pr_step (same d_r_old): columns [0, cols_ready) of d_r_old are final,
 * the rest may still be arriving.  With the LDS-tiled engine the products of the entries in the strips inside
 * that range are computed now (each strip once; the following step does only what is left); otherwise a no-op.
 * spmv_c_pr_reset voids a head start. */
int spmv_c_pr_expand(spmv_c_pr_shard* shard, const float* d_r_old, int64_t cols_ready, void* hip_stream);
/* the same step, additionally storing every new value at the same offset of `num_peers` other
 * vectors (host array of device pointers: the peers' r_new buffers, IPC-mapped) — a push-style
 * all-gather over xGMI fused into the step's epilogue */
int spmv_c_pr_step_push(spmv_c_pr_shard* shard, const float* d_r_old, float* d_r_new, float damping,
                        float* const* peer_r_new, int num_peers, void* hip_stream);
int spmv_c_pr_reduce(spmv_c_pr_shard* shard, double* d_sums /*[2]*/, void* hip_stream);
int spmv_c_pr_commit(spmv_c_pr_shard* shard, const double* d_sums, float tolerance, void* hip_stream);
/* single rank: reduce + commit in one launch (no sums buffer leaves the engine) */
int spmv_c_pr_reduce_commit(spmv_c_pr_shard* shard, float tolerance, void* hip_stream);
/* multi-rank commit without an all-reduce: rank p's two partial sums (as doubles) sit in the 16-byte
 * tail of its slice, d_gathered[p * stride + shard_len ...]; stride >= shard_len + 4, both even */
int spmv_c_pr_commit_gathered(spmv_c_pr_shard* shard, const float* d_gathered, int world, int64_t stride,
                              int64_t shard_len, float tolerance, void* hip_stream);
int spmv_c_pr_status_get(spmv_c_pr_shard* shard, spmv_c_pr_status* out, void* hip_stream); /* syncs */
/* dangling-node detection on the device: accumulate this shard's column sums
 * (atomic adds into d_col_sums[n_global]); after the sums of all shards are
 * combined, mask[c] = (sum == 0). */
int spmv_c_pr_column_sums(const spmv_c_csr* A_local, float* d_col_sums, void* hip_stream);
int spmv_c_pr_mask_from_column_sums(const float* d_col_sums, int n, uint8_t* d_mask,
                                    uint64_t* d_count, void* hip_stream);
int spmv_c_fill(float* d_r, size_t n, float value, void* hip_stream);

/* ---- synthetic inputs generated in HBM (extension; numpy twin: gpu-spmv_amd/synth.py) ---- */
int spmv_c_gen_uniform_rows(uint64_t seed, int row_begin, int local_rows, int n_cols, int k,
                            int32_t* d_row_ptrs, int32_t* d_cols, float* d_vals, void* hip_stream);
/* the same entries as gen_uniform_rows(row_begin = 0), stored column-major (ELL, K = k, no padding) */
int spmv_c_gen_uniform_ell(uint64_t seed, int rows, int n_cols, int k, int32_t* d_cols, float* d_vals,
                           void* hip_stream);
int spmv_c_gen_stratified_rows(uint64_t seed, int row_begin, int local_rows, int n_cols,
                               const int32_t* d_row_ptrs, int32_t* d_cols, float* d_vals,
                               void* hip_stream);
int spmv_c_gen_vector(uint64_t seed, uint64_t tag, size_t n, float* d_x, void* hip_stream);
int spmv_c_count_columns(int64_t nnz, const int32_t* d_cols, int n_cols, int32_t* d_counts,
                         void* hip_stream);
int spmv_c_reciprocal_values(int64_t nnz, const int32_t* d_cols, const int32_t* d_counts,
                             float* d_vals, void* hip_stream);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_C_H */
