// rocsparse_compare.cpp — developer probe: the vendor library's CSR SpMV (rocSPARSE, generic API, every CSR
// algorithm it offers) on the bench matrices, next to this library's spmv_csr on the same device arrays.
// Not part of the product and not linked by it.  Build (see tools/README.md):
//   hipcc -O2 -std=c++17 -Iinclude tools/rocsparse_compare.cpp -Lgpu-spmv_amd/lib -lspmv_amd -lrocsparse \
//         -Wl,-rpath,$PWD/gpu-spmv_amd/lib -o tools/rocsparse_compare
#include "spmv_c.h"

#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#pragma clang diagnostic ignored "-Wdeprecated-declarations"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
#define RS_OK(call) do { rocsparse_status s_ = (call); if (s_ != rocsparse_status_success) { std::printf("rocSPARSE status %d at line %d\n", static_cast<int>(s_), __LINE__); std::exit(1); } } while (0)

static std::string g_only;
static bool wanted(const char* what) { return g_only.empty() || g_only == what; }

static double csr_bytes(long long rows, long long cols, long long nnz) {
    return nnz * 8.0 + (rows + 1) * 4.0 + cols * 4.0 + rows * 4.0;
}

static void run_case(const char* label, int rows, int cols, int k) {
    const long long nnz = static_cast<long long>(rows) * k;
    int *d_ptr = nullptr, *d_col = nullptr;
    float *d_val = nullptr, *d_x = nullptr, *d_y = nullptr, *d_ref = nullptr;
    HIP_OK(hipMalloc(&d_ptr, (rows + 1) * sizeof(int)));
    HIP_OK(hipMalloc(&d_col, nnz * sizeof(int)));
    HIP_OK(hipMalloc(&d_val, nnz * sizeof(float)));
    HIP_OK(hipMalloc(&d_x, cols * sizeof(float)));
    HIP_OK(hipMalloc(&d_y, rows * sizeof(float)));
    HIP_OK(hipMalloc(&d_ref, rows * sizeof(float)));
    if (spmv_c_gen_uniform_rows(42, 0, rows, cols, k, d_ptr, d_col, d_val, nullptr) != 0 ||
        spmv_c_gen_vector(42, 1, cols, d_x, nullptr) != 0) {
        std::printf("generator failed\n");
        std::exit(1);
    }
    HIP_OK(hipDeviceSynchronize());
    const double bytes = csr_bytes(rows, cols, nnz);
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0));
    HIP_OK(hipEventCreate(&t1));

    // ---- this library (spmv_auto_config's choice for these shapes: VECTOR_CSR + use_texture) ----
    spmv_c_csr* A = spmv_c_csr_wrap_device(rows, cols, static_cast<int>(nnz), d_ptr, d_col, d_val);
    spmv_c_config cfg{};
    cfg.kernel_type = 1;
    cfg.block_size = 256;
    cfg.use_texture = 1;
    spmv_c_result res{};
    double ours_ms = 0.0;
    for (int i = 0; i < 25; ++i) {
        spmv_c_spmv_csr(A, d_x, d_ref, &cfg, cols, &res);
        if (res.error_code != 0) { std::printf("spmv_csr error %d\n", res.error_code); std::exit(1); }
        if (i >= 5) ours_ms += res.elapsed_ms;
    }
    ours_ms /= 20.0;
    std::printf("%-14s %-28s %9.1f us  %7.1f GB/s\n", label, "this library (LDS tiles)", ours_ms * 1e3, bytes / ours_ms / 1e6);
    std::vector<float> ref(rows), got(rows);
    HIP_OK(hipMemcpy(ref.data(), d_ref, rows * sizeof(float), hipMemcpyDeviceToHost));

    // ---- rocSPARSE ----
    rocsparse_handle handle;
    RS_OK(rocsparse_create_handle(&handle));
    rocsparse_spmat_descr mat;
    rocsparse_dnvec_descr vx, vy;
    RS_OK(rocsparse_create_csr_descr(&mat, rows, cols, nnz, d_ptr, d_col, d_val, rocsparse_indextype_i32,
                                     rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RS_OK(rocsparse_create_dnvec_descr(&vx, cols, d_x, rocsparse_datatype_f32_r));
    RS_OK(rocsparse_create_dnvec_descr(&vy, rows, d_y, rocsparse_datatype_f32_r));
    const float alpha = 1.0f, beta = 0.0f;
    const struct { rocsparse_spmv_alg alg; const char* name; } algs[] = {
        {rocsparse_spmv_alg_csr_adaptive, "rocsparse csr_adaptive"},
        {rocsparse_spmv_alg_csr_rowsplit, "rocsparse csr_rowsplit"},
        // csr_lrb / csr_nnzsplit are left out: one of them took a GPU memory fault (address 0) through this
        // (deprecated) entry point on ROCm 7.2 and was not pursued
    };
    for (const auto& a : algs) {
        if (!wanted(a.name + 14)) continue;            // the part after "rocsparse csr_"
        std::printf("%-14s %-28s ...\n", label, a.name);
        size_t buffer_size = 0;
        void* buffer = nullptr;
        if (rocsparse_spmv(handle, rocsparse_operation_none, &alpha, mat, vx, &beta, vy, rocsparse_datatype_f32_r, a.alg,
                           rocsparse_spmv_stage_buffer_size, &buffer_size, nullptr) != rocsparse_status_success) {
            std::printf("%-14s %-28s not available\n", label, a.name);
            continue;
        }
        HIP_OK(hipMalloc(&buffer, std::max<size_t>(buffer_size, 16)));
        HIP_OK(hipEventRecord(t0));
        const rocsparse_status pre = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, mat, vx, &beta, vy,
                                                    rocsparse_datatype_f32_r, a.alg, rocsparse_spmv_stage_preprocess,
                                                    &buffer_size, buffer);
        HIP_OK(hipEventRecord(t1));
        HIP_OK(hipEventSynchronize(t1));
        float pre_ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&pre_ms, t0, t1));
        if (pre != rocsparse_status_success) {
            std::printf("%-14s %-28s preprocess failed (%d)\n", label, a.name, static_cast<int>(pre));
            HIP_OK(hipFree(buffer));
            continue;
        }
        double total = 0.0;
        bool ok = true;
        for (int i = 0; i < 25 && ok; ++i) {
            HIP_OK(hipEventRecord(t0));
            ok = rocsparse_spmv(handle, rocsparse_operation_none, &alpha, mat, vx, &beta, vy, rocsparse_datatype_f32_r, a.alg,
                                rocsparse_spmv_stage_compute, &buffer_size, buffer) == rocsparse_status_success;
            HIP_OK(hipEventRecord(t1));
            HIP_OK(hipEventSynchronize(t1));
            float ms = 0.0f;
            HIP_OK(hipEventElapsedTime(&ms, t0, t1));
            if (i >= 5) total += ms;
        }
        if (!ok) {
            std::printf("%-14s %-28s compute failed\n", label, a.name);
        } else {
            const double ms = total / 20.0;
            HIP_OK(hipMemcpy(got.data(), d_y, rows * sizeof(float), hipMemcpyDeviceToHost));
            double worst = 0.0;
            for (int i = 0; i < rows; i += 97) worst = std::max(worst, std::fabs(static_cast<double>(got[i]) - ref[i]));
            std::printf("%-14s %-28s %9.1f us  %7.1f GB/s   (preprocess %.2f ms, max |diff| vs ours %.2e)\n", label, a.name,
                        ms * 1e3, bytes / ms / 1e6, pre_ms, worst);
        }
        HIP_OK(hipFree(buffer));
    }
    rocsparse_destroy_dnvec_descr(vx);
    rocsparse_destroy_dnvec_descr(vy);
    rocsparse_destroy_spmat_descr(mat);
    rocsparse_destroy_handle(handle);
    spmv_c_csr_destroy(A);
    for (void* p : {static_cast<void*>(d_ptr), static_cast<void*>(d_col), static_cast<void*>(d_val),
                    static_cast<void*>(d_x), static_cast<void*>(d_y), static_cast<void*>(d_ref)}) HIP_OK(hipFree(p));
}

int main(int argc, char** argv) {
    std::setvbuf(stdout, nullptr, _IONBF, 0);          // nothing is lost if a library call aborts the process
    g_only = argc > 1 ? argv[1] : "";                  // "ours", "adaptive", "rowsplit" or nothing = all
    run_case("C2 1Mx16", 1000000, 1000000, 16);
    run_case("C5 10Mx16", 10000000, 10000000, 16);
    return 0;
}
