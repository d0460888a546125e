// spmv/common.h — error codes, exception type and check macros for the
// MI355X-native SpMV library.
//
// Drop-in for the reference's include/spmv/common.h (reference
// include/spmv/common.h:13-67): same enum values, same message strings
// (pinned by reference tests/test_common.cpp:8-18), same macro and class
// names.  The runtime underneath is HIP (gfx950); the CUDA-flavoured names
// are kept only because callers spell them that way.
#ifndef SPMV_COMMON_H
#define SPMV_COMMON_H

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1   // lets plain g++ translation units include the HIP host API
#endif
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace spmv {

enum class SpMVError {
    SUCCESS           =  0,
    INVALID_DIMENSION = -1,
    CUDA_MALLOC       = -2,   // device allocation failed (hipMalloc)
    CUDA_MEMCPY       = -3,   // device copy failed (hipMemcpy)
    KERNEL_LAUNCH     = -4,
    INVALID_FORMAT    = -5,
    FILE_IO           = -6,
    OUT_OF_MEMORY     = -7,
    INVALID_ARGUMENT  = -8
};

inline const char* spmv_error_string(SpMVError err) {
    static const struct { SpMVError code; const char* text; } table[] = {
        {SpMVError::SUCCESS,           "Success"},
        {SpMVError::INVALID_DIMENSION, "Invalid matrix/vector dimension"},
        {SpMVError::CUDA_MALLOC,       "CUDA memory allocation failed"},
        {SpMVError::CUDA_MEMCPY,       "CUDA memory copy failed"},
        {SpMVError::KERNEL_LAUNCH,     "CUDA kernel launch failed"},
        {SpMVError::INVALID_FORMAT,    "Invalid sparse matrix format"},
        {SpMVError::FILE_IO,           "File I/O error"},
        {SpMVError::OUT_OF_MEMORY,     "Out of memory"},
        {SpMVError::INVALID_ARGUMENT,  "Invalid argument"},
    };
    for (const auto& e : table) {
        if (e.code == err) return e.text;
    }
    return "Unknown error";
}

// Thrown by CudaBuffer only; carries the HIP status.
class CudaException : public std::runtime_error {
public:
    explicit CudaException(hipError_t status)
        : std::runtime_error(std::string("CUDA error: ") + hipGetErrorString(status)),
          status_(status) {}
    hipError_t error() const { return status_; }
private:
    hipError_t status_;
};

// Allocation failures map to CUDA_MALLOC, copy failures to CUDA_MEMCPY
// (the reference folds both into CUDA_MALLOC — SURVEY.md §0 D7).
#define SPMV_HIP_CHECK_AS(call, code) do {                                   \
    hipError_t spmv_status_ = (call);                                        \
    if (spmv_status_ != hipSuccess) {                                        \
        fprintf(stderr, "HIP error at %s:%d: %s\n", __FILE__, __LINE__,      \
                hipGetErrorString(spmv_status_));                            \
        return static_cast<int>(code);                                       \
    }                                                                        \
} while (0)

#define CUDA_CHECK(call) SPMV_HIP_CHECK_AS(call, spmv::SpMVError::CUDA_MALLOC)

#define CUDA_CHECK_THROW(call) do {                                          \
    hipError_t spmv_status_ = (call);                                        \
    if (spmv_status_ != hipSuccess) {                                        \
        throw spmv::CudaException(spmv_status_);                             \
    }                                                                        \
} while (0)

} // namespace spmv

#endif // SPMV_COMMON_H
