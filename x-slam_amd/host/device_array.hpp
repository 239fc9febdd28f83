// device_array.hpp — reference-counted device memory over the HIP runtime, with the public
// methods of the reference's DeviceMemory / DeviceMemory2D / DeviceArray<T> / DeviceArray2D<T>
// (DeviceArray/include/device_memory.h:18-277, device_array.hpp:25-442,
// src/device_memory.cpp:74-286): create (a no-op when the size is unchanged), release, copyTo,
// upload, download, swap, ptr, implicit conversion to the kernel views, and a user-pointer
// constructor that disables counting.
//
// Differences by design (MI355X): 2-D allocations use a pitch rounded up to 256 B — one
// coalesced wave-row of floats — from plain hipMalloc instead of cudaMallocPitch, and copies
// are ordered on the current stream (hipMemcpy*Async + one stream synchronise) instead of a
// device-wide cudaDeviceSynchronize after every copy (device_memory.cpp:128-129 etc.).
#pragma once
#include "kernel_containers.hpp"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <utility>
#include <vector>

namespace xs_host {
// Common/include/cx.h:124-130: print and exit(-1)
inline void safe_call(hipError_t err, const char *msg = nullptr) {
    if (hipSuccess != err) {
        printf("HIP error(%s): %s\n", msg ? msg : "", hipGetErrorString(err));
        exit(-1);
    }
}
inline hipStream_t &current_stream() { static thread_local hipStream_t s = nullptr; return s; }  // per host thread
}  // namespace xs_host
#define hipSafeCall(e) xs_host::safe_call((e), #e)

class DeviceMemory {
public:
    DeviceMemory() : data_(nullptr), sizeBytes_(0), refcount_(nullptr) {}
    ~DeviceMemory() { release(); }
    DeviceMemory(size_t sizeBytes_arg) : data_(nullptr), sizeBytes_(0), refcount_(nullptr) { create(sizeBytes_arg); }
    DeviceMemory(void *ptr_arg, size_t sizeBytes_arg) : data_(ptr_arg), sizeBytes_(sizeBytes_arg), refcount_(nullptr) {}
    DeviceMemory(const DeviceMemory &o) : data_(o.data_), sizeBytes_(o.sizeBytes_), refcount_(o.refcount_) { if (refcount_) refcount_->fetch_add(1); }
    DeviceMemory &operator=(const DeviceMemory &o) {
        if (this != &o) {
            if (o.refcount_) o.refcount_->fetch_add(1);
            release();
            data_ = o.data_; sizeBytes_ = o.sizeBytes_; refcount_ = o.refcount_;
        }
        return *this;
    }
    void create(size_t sizeBytes_arg) {
        if (sizeBytes_arg == sizeBytes_) return;
        if (sizeBytes_arg > 0) {
            if (data_) release();
            sizeBytes_ = sizeBytes_arg;
            hipSafeCall(hipMalloc(&data_, sizeBytes_));
            refcount_ = new std::atomic<int>(1);
        }
    }
    void release() {
        if (refcount_ && refcount_->fetch_sub(1) == 1) {
            delete refcount_;
            hipSafeCall(hipFree(data_));
        }
        data_ = nullptr; sizeBytes_ = 0; refcount_ = nullptr;
    }
    void copyTo(DeviceMemory &other) const {
        if (empty()) other.release();
        else {
            other.create(sizeBytes_);
            hipSafeCall(hipMemcpyAsync(other.data_, data_, sizeBytes_, hipMemcpyDeviceToDevice, xs_host::current_stream()));
            hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
        }
    }
    void upload(const void *host_ptr_arg, size_t sizeBytes_arg) {
        create(sizeBytes_arg);
        hipSafeCall(hipMemcpyAsync(data_, host_ptr_arg, sizeBytes_, hipMemcpyHostToDevice, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
    }
    bool upload(const void *host_ptr_arg, std::size_t device_begin_byte_offset, std::size_t num_bytes) {
        if (device_begin_byte_offset + num_bytes > sizeBytes_) return false;
        hipSafeCall(hipMemcpyAsync((char *)data_ + device_begin_byte_offset, host_ptr_arg, num_bytes, hipMemcpyHostToDevice, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
        return true;
    }
    void download(void *host_ptr_arg) const {
        hipSafeCall(hipMemcpyAsync(host_ptr_arg, data_, sizeBytes_, hipMemcpyDeviceToHost, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
    }
    bool download(void *host_ptr_arg, std::size_t device_begin_byte_offset, std::size_t num_bytes) const {
        if (device_begin_byte_offset + num_bytes > sizeBytes_) return false;
        hipSafeCall(hipMemcpyAsync(host_ptr_arg, (const char *)data_ + device_begin_byte_offset, num_bytes, hipMemcpyDeviceToHost, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
        return true;
    }
    void swap(DeviceMemory &o) { std::swap(data_, o.data_); std::swap(sizeBytes_, o.sizeBytes_); std::swap(refcount_, o.refcount_); }
    template <class T> T *ptr() { return (T *)data_; }
    template <class T> const T *ptr() const { return (const T *)data_; }
    template <class U> operator PtrSz<U>() const { PtrSz<U> r; r.data = (U *)ptr<U>(); r.size = sizeBytes_ / sizeof(U); return r; }
    bool empty() const { return !data_; }
    size_t sizeBytes() const { return sizeBytes_; }

private:
    void *data_;
    std::size_t sizeBytes_;
    std::atomic<int> *refcount_;
};

class DeviceMemory2D {
public:
    DeviceMemory2D() : data_(nullptr), step_(0), colsBytes_(0), rows_(0), refcount_(nullptr) {}
    ~DeviceMemory2D() { release(); }
    DeviceMemory2D(int rows_arg, int colsBytes_arg) : data_(nullptr), step_(0), colsBytes_(0), rows_(0), refcount_(nullptr) { create(rows_arg, colsBytes_arg); }
    DeviceMemory2D(int rows_arg, int colsBytes_arg, void *data_arg, size_t step_arg)
        : data_(data_arg), step_(step_arg), colsBytes_(colsBytes_arg), rows_(rows_arg), refcount_(nullptr) {}
    DeviceMemory2D(const DeviceMemory2D &o) : data_(o.data_), step_(o.step_), colsBytes_(o.colsBytes_), rows_(o.rows_), refcount_(o.refcount_) {
        if (refcount_) refcount_->fetch_add(1);
    }
    DeviceMemory2D &operator=(const DeviceMemory2D &o) {
        if (this != &o) {
            if (o.refcount_) o.refcount_->fetch_add(1);
            release();
            colsBytes_ = o.colsBytes_; rows_ = o.rows_; data_ = o.data_; step_ = o.step_; refcount_ = o.refcount_;
        }
        return *this;
    }
    void create(int rows_arg, int colsBytes_arg) {
        if (colsBytes_ == colsBytes_arg && rows_ == rows_arg) return;
        if (rows_arg > 0 && colsBytes_arg > 0) {
            if (data_) release();
            colsBytes_ = colsBytes_arg; rows_ = rows_arg;
            step_ = ((size_t)colsBytes_ + 255) / 256 * 256;  // one 64-lane wave-row of floats
            hipSafeCall(hipMalloc(&data_, step_ * (size_t)rows_));
            refcount_ = new std::atomic<int>(1);
        }
    }
    void release() {
        if (refcount_ && refcount_->fetch_sub(1) == 1) {
            delete refcount_;
            hipSafeCall(hipFree(data_));
        }
        colsBytes_ = 0; rows_ = 0; data_ = nullptr; step_ = 0; refcount_ = nullptr;
    }
    void copyTo(DeviceMemory2D &other) const {
        if (empty()) other.release();
        else {
            other.create(rows_, colsBytes_);
            hipSafeCall(hipMemcpy2DAsync(other.data_, other.step_, data_, step_, colsBytes_, rows_, hipMemcpyDeviceToDevice, xs_host::current_stream()));
            hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
        }
    }
    void upload(const void *host_ptr_arg, std::size_t host_step_arg, int rows_arg, int colsBytes_arg) {
        create(rows_arg, colsBytes_arg);
        hipSafeCall(hipMemcpy2DAsync(data_, step_, host_ptr_arg, host_step_arg, colsBytes_, rows_, hipMemcpyHostToDevice, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
    }
    void download(void *host_ptr_arg, std::size_t host_step_arg) const {
        hipSafeCall(hipMemcpy2DAsync(host_ptr_arg, host_step_arg, data_, step_, colsBytes_, rows_, hipMemcpyDeviceToHost, xs_host::current_stream()));
        hipSafeCall(hipStreamSynchronize(xs_host::current_stream()));
    }
    void swap(DeviceMemory2D &o) {
        std::swap(data_, o.data_); std::swap(step_, o.step_); std::swap(colsBytes_, o.colsBytes_); std::swap(rows_, o.rows_); std::swap(refcount_, o.refcount_);
    }
    template <class T> T *ptr(int y_arg = 0) { return (T *)((char *)data_ + y_arg * step_); }
    template <class T> const T *ptr(int y_arg = 0) const { return (const T *)((const char *)data_ + y_arg * step_); }
    template <class U> operator PtrStep<U>() const { PtrStep<U> r; r.data = (U *)ptr<U>(); r.step = step_; return r; }
    template <class U> operator PtrStepSz<U>() const {
        PtrStepSz<U> r; r.data = (U *)ptr<U>(); r.step = step_; r.cols = colsBytes_ / sizeof(U); r.rows = rows_; return r;
    }
    bool empty() const { return !data_; }
    int colsBytes() const { return colsBytes_; }
    int rows() const { return rows_; }
    size_t step() const { return step_; }

private:
    void *data_;
    std::size_t step_;
    int colsBytes_;
    int rows_;
    std::atomic<int> *refcount_;
};

template <class T>
class DeviceArray : public DeviceMemory {
public:
    using type = T;
    enum { elem_size = sizeof(T) };
    DeviceArray() {}
    DeviceArray(std::size_t size) : DeviceMemory(size * elem_size) {}
    DeviceArray(T *ptr, std::size_t size) : DeviceMemory(ptr, size * elem_size) {}
    DeviceArray(const DeviceArray &other) : DeviceMemory(other) {}
    DeviceArray &operator=(const DeviceArray &other) { DeviceMemory::operator=(other); return *this; }
    void create(std::size_t size) { DeviceMemory::create(size * elem_size); }
    void release() { DeviceMemory::release(); }
    void copyTo(DeviceArray &other) const { DeviceMemory::copyTo(other); }
    void upload(const T *host_ptr, std::size_t size) { DeviceMemory::upload(host_ptr, size * elem_size); }
    bool upload(const T *host_ptr, std::size_t device_begin_offset, std::size_t num_elements) {
        return DeviceMemory::upload(host_ptr, device_begin_offset * elem_size, num_elements * elem_size);
    }
    void download(T *host_ptr) const { DeviceMemory::download(host_ptr); }
    bool download(T *host_ptr, std::size_t device_begin_offset, std::size_t num_elements) const {
        return DeviceMemory::download(host_ptr, device_begin_offset * elem_size, num_elements * elem_size);
    }
    template <class A> void upload(const std::vector<T, A> &data) { upload(&data[0], data.size()); }
    template <typename A> void download(std::vector<T, A> &data) const { data.resize(size()); if (!data.empty()) download(&data[0]); }
    void swap(DeviceArray &other_arg) { DeviceMemory::swap(other_arg); }
    T *ptr() { return DeviceMemory::ptr<T>(); }
    const T *ptr() const { return DeviceMemory::ptr<T>(); }
    operator T *() { return ptr(); }
    operator const T *() const { return ptr(); }
    std::size_t size() const { return sizeBytes() / elem_size; }
};

template <class T>
class DeviceArray2D : public DeviceMemory2D {
public:
    using type = T;
    enum { elem_size = sizeof(T) };
    DeviceArray2D() {}
    DeviceArray2D(int rows, int cols) : DeviceMemory2D(rows, cols * elem_size) {}
    DeviceArray2D(int rows, int cols, void *data, std::size_t stepBytes) : DeviceMemory2D(rows, cols * elem_size, data, stepBytes) {}
    DeviceArray2D(const DeviceArray2D &other) : DeviceMemory2D(other) {}
    DeviceArray2D &operator=(const DeviceArray2D &other) { DeviceMemory2D::operator=(other); return *this; }
    void create(int rows, int cols) { DeviceMemory2D::create(rows, cols * elem_size); }
    void release() { DeviceMemory2D::release(); }
    void copyTo(DeviceArray2D &other) const { DeviceMemory2D::copyTo(other); }
    void upload(const void *host_ptr, std::size_t host_step, int rows, int cols) { DeviceMemory2D::upload(host_ptr, host_step, rows, cols * elem_size); }
    void download(void *host_ptr, std::size_t host_step) const { DeviceMemory2D::download(host_ptr, host_step); }
    void swap(DeviceArray2D &other_arg) { DeviceMemory2D::swap(other_arg); }
    template <class A> void upload(const std::vector<T, A> &data, int cols) { upload(&data[0], cols * elem_size, data.size() / cols, cols); }
    template <class A> void download(std::vector<T, A> &data, int &elem_step) const {
        elem_step = cols();
        data.resize(cols() * rows());
        if (!data.empty()) download(&data[0], colsBytes());
    }
    T *ptr(int y = 0) { return DeviceMemory2D::ptr<T>(y); }
    const T *ptr(int y = 0) const { return DeviceMemory2D::ptr<T>(y); }
    operator T *() { return ptr(); }
    operator const T *() const { return ptr(); }
    int cols() const { return DeviceMemory2D::colsBytes() / elem_size; }
    int rows() const { return DeviceMemory2D::rows(); }
    std::size_t elem_step() const { return DeviceMemory2D::step() / elem_size; }
};
