// spmv/cuda_buffer.h — move-only RAII owner of a device allocation in HBM.
//
// Same shape and semantics as the reference's CudaBuffer<T>
// (reference include/spmv/cuda_buffer.h:12-101, pinned by
// tests/test_common.cpp:21-98): sized ctor throws CudaException on
// allocation failure, copies throw std::runtime_error when count > size,
// resize discards contents, zero-size buffers hold nullptr.  Backed by
// hipMalloc/hipMemcpy/hipFree.
#ifndef SPMV_CUDA_BUFFER_H
#define SPMV_CUDA_BUFFER_H

#include "common.h"
#include <cstddef>
#include <utility>

namespace spmv {

template <typename T>
class CudaBuffer {
public:
    CudaBuffer() = default;

    explicit CudaBuffer(size_t count) : size_(count) { allocate(); }

    ~CudaBuffer() { drop(); }

    CudaBuffer(const CudaBuffer&) = delete;
    CudaBuffer& operator=(const CudaBuffer&) = delete;

    CudaBuffer(CudaBuffer&& other) noexcept
        : ptr_(std::exchange(other.ptr_, nullptr)),
          size_(std::exchange(other.size_, 0)) {}

    CudaBuffer& operator=(CudaBuffer&& other) noexcept {
        if (this != &other) {
            drop();
            ptr_  = std::exchange(other.ptr_, nullptr);
            size_ = std::exchange(other.size_, 0);
        }
        return *this;
    }

    T* get() { return ptr_; }
    const T* get() const { return ptr_; }
    size_t size() const { return size_; }
    bool empty() const { return ptr_ == nullptr || size_ == 0; }

    void copyFromHost(const T* host_data, size_t count) {
        check_count(count);
        CUDA_CHECK_THROW(hipMemcpy(ptr_, host_data, count * sizeof(T), hipMemcpyHostToDevice));
    }

    void copyToHost(T* host_data, size_t count) const {
        check_count(count);
        CUDA_CHECK_THROW(hipMemcpy(host_data, ptr_, count * sizeof(T), hipMemcpyDeviceToHost));
    }

    void resize(size_t new_count) {
        if (new_count == size_) return;
        drop();
        size_ = new_count;
        allocate();
    }

    void release() {
        drop();
        size_ = 0;
    }

private:
    void allocate() {
        if (size_ == 0) return;
        void* raw = nullptr;
        hipError_t status = hipMalloc(&raw, size_ * sizeof(T));
        if (status != hipSuccess) {
            size_ = 0;
            throw CudaException(status);
        }
        ptr_ = static_cast<T*>(raw);
    }

    void drop() {
        if (ptr_) {
            (void)hipFree(ptr_);
            ptr_ = nullptr;
        }
    }

    void check_count(size_t count) const {
        if (count > size_) {
            throw std::runtime_error("Copy size exceeds buffer size");
        }
    }

    T* ptr_ = nullptr;
    size_t size_ = 0;
};

} // namespace spmv

#endif // SPMV_CUDA_BUFFER_H
