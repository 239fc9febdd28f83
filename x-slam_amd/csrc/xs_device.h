// xs_device.h — device-side vector helpers, kernel argument PODs and small utilities shared
// by the HIP kernels.  Mirrors XKinectFusion/include/Internal.h:42-154 (Intr, devComplex3,
// MatS33 and their operators) and Common/include/cx.h:131,158 (divUp, quiet_NaN bits).
#pragma once
#include "xs_complex.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace xs {

struct Intr { float fx, fy, cx, cy; };  // Internal.h:49-59

struct cfloat3 { cfloat x, y, z; };          // devComplex3, 24 B
struct MatS33 { cfloat3 data[3]; };          // 72 B, row-major
struct dcfloat3 { dcfloat x, y, z; };        // devDComplex3, 48 B
struct MatD33 { dcfloat3 data[3]; };         // 144 B

__device__ __forceinline__ cfloat3 mk3(cfloat x, cfloat y, cfloat z) { cfloat3 t; t.x = x; t.y = y; t.z = z; return t; }
__device__ __forceinline__ cfloat dot(const cfloat3 &a, const cfloat3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ cfloat3 operator+(const cfloat3 &a, const cfloat3 &b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ cfloat3 operator-(const cfloat3 &a, const cfloat3 &b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ cfloat3 operator*(const cfloat3 &a, float v) { return mk3(a.x * v, a.y * v, a.z * v); }
__device__ __forceinline__ cfloat3 operator*(const cfloat3 &a, cfloat v) { return mk3(a.x * v, a.y * v, a.z * v); }
__device__ __forceinline__ cfloat norm(const cfloat3 &v) { return sqrt(dot(v, v)); }
__device__ __forceinline__ cfloat squarednorm(const cfloat3 &v) { return dot(v, v); }
// Internal.h:134-137 evaluates norm(v) three times; the value is the same each time
__device__ __forceinline__ cfloat3 normalized(const cfloat3 &v) { cfloat n = norm(v); return mk3(v.x / n, v.y / n, v.z / n); }
__device__ __forceinline__ cfloat3 cross(const cfloat3 &a, const cfloat3 &b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ cfloat3 operator*(const MatS33 &m, const cfloat3 &v) {
    return mk3(dot(m.data[0], v), dot(m.data[1], v), dot(m.data[2], v));
}

__device__ __forceinline__ dcfloat dot(const dcfloat3 &a, const dcfloat3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ dcfloat norm(const dcfloat3 &v) { return sqrt(dot(v, v)); }

// TsdfFusion.h:7-25: a voxel's complex TSDF is stored as value (real part) and grad (imaginary part)
// in two float arrays, its weight in a third
__device__ __forceinline__ void pack_tsdf(cfloat tsdf, int weight, float &save_value, int &save_weight, float &save_grad) {
    save_value = tsdf.re; save_weight = weight; save_grad = tsdf.im;
}
__device__ __forceinline__ void unpack_tsdf(float save_value, int save_weight, float save_grad, cfloat &tsdf, int &weight) {
    weight = save_weight; tsdf = cfloat(save_value, save_grad);
}
__device__ __forceinline__ cfloat unpack_tsdf(float save_value, float save_grad) { return cfloat(save_value, save_grad); }

__host__ __device__ __forceinline__ float qnan_f() {  // cx.h:158: __int_as_float(0x7fffffff)
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(0x7fffffff);
#else
    union { uint32_t u; float f; } v; v.u = 0x7fffffffu; return v.f;
#endif
}

// floor(x / c) for a launch constant c > 0 without a divide: q = x * r with r = RN(1 / c), one fused residual e = q * c - x
// (exact), q - e * r.  For r correctly rounded this is the correctly rounded quotient (Markstein) wherever nothing underflows — and
// it is not taken on trust: xs_const_div_prepare(c) runs both forms over all 2^32 operands on the device (2 ms) and the short form is
// used for a constant only after that.  `ok` bit 1: floor(short form) == floor(x / c) for every |x| <= 2^60, denormals and zeros
// included (the voxel-index use); bit 0: the quotients themselves are bit-identical for 2^-60 <= |x| <= 2^60 and +-0 (written this
// way the residual keeps the sign of a zero) — recorded, but no kernel uses it: with a launch-constant divisor the compiler hoists half
// of the IEEE sequence out of the loops and the short quotient with its domain guard measured slower (xs_raycast.hip, voxel_index).
// Explicit fma builtins: the library is built with -ffp-contract=off.
struct ConstDiv { float c, rc; unsigned ok; };
__device__ __forceinline__ float div_short(float x, float c, float rc) {
    const float q = x * rc;
    const float e = __builtin_fmaf(q, c, -x);
    return __builtin_fmaf(e, -rc, q);
}
__device__ __forceinline__ int cvt_floor(float v) {  // (int)floorf(v) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
template <bool SHORT> __device__ __forceinline__ int floor_div_by(float x, const ConstDiv &d) {   // (int)floorf(x / c)
    if (!SHORT) return __float2int_rd(x / d.c);
    // branch-free: beyond 2^60 — where the check does not reach — the plain product x * r stands in: for 2^-20 <= c <= 2^20
    // (xs_const_div_prepare refuses others) it and the quotient both exceed 2^39 in magnitude, or are both the same infinity or
    // both NaN, and the conversion saturates alike
    const float q0 = x * d.rc;
    const float e = __builtin_fmaf(q0, d.c, -x);
    const float q1 = __builtin_fmaf(e, -d.rc, q0);
    return cvt_floor(fabsf(x) <= 0x1p60f ? q1 : q0);
}

template <class T> __device__ __forceinline__ T *row_ptr(T *base, size_t step, int y) { return (T *)((char *)base + (size_t)y * step); }
template <class T> __device__ __forceinline__ const T *row_ptr(const T *base, size_t step, int y) {
    return (const T *)((const char *)base + (size_t)y * step);
}

static inline int div_up(int total, int grain) { return (total + grain - 1) / grain; }

// wave64 sum of a 32-bit count via cross-lane shuffles
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace xs

// status plumbing for the C ABI: 0 = ok, otherwise the hipError_t value
#define XS_CHECK(expr)                                   \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return xs_set_error(_e, #expr); \
    } while (0)
extern "C" int xs_set_error(hipError_t e, const char *what);
