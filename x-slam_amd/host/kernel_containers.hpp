// kernel_containers.hpp — borrowed views of device memory: pointer (+ size / pitch in bytes).
// Same public surface as the reference's DevPtr / PtrSz / PtrStep / PtrStepSz
// (DeviceArray/include/kernel_containers.hpp:16-81), so host code written against the
// reference's launchers compiles against xs_launchers.hpp unchanged.
#pragma once
#include <cstddef>

template <typename T>
struct DevPtr {
    typedef T elem_type;
    const static size_t elem_size = sizeof(elem_type);
    T *data;
    DevPtr() : data(nullptr) {}
    DevPtr(T *data_arg) : data(data_arg) {}
    std::size_t elemSize() const { return elem_size; }
    operator T *() { return data; }
    operator const T *() const { return data; }
};

template <typename T>
struct PtrSz : public DevPtr<T> {
    PtrSz() : size(0) {}
    PtrSz(T *data_arg, std::size_t size_arg) : DevPtr<T>(data_arg), size(size_arg) {}
    std::size_t size;
};

template <typename T>
struct PtrStep : public DevPtr<T> {
    PtrStep() : step(0) {}
    PtrStep(T *data_arg, size_t step_arg) : DevPtr<T>(data_arg), step(step_arg) {}
    size_t step;  // row pitch in BYTES
    T *ptr(int y = 0) { return (T *)((char *)DevPtr<T>::data + y * step); }
    const T *ptr(int y = 0) const { return (const T *)((const char *)DevPtr<T>::data + y * step); }
};

template <typename T>
struct PtrStepSz : public PtrStep<T> {
    PtrStepSz() : cols(0), rows(0) {}
    PtrStepSz(int rows_arg, int cols_arg, T *data_arg, size_t step_arg) : PtrStep<T>(data_arg, step_arg), cols(cols_arg), rows(rows_arg) {}
    int cols;
    int rows;
};
