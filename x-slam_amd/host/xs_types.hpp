// xs_types.hpp — the POD argument types of the reference's launcher API (XKinectFusion/include/Internal.h:19-59,
// 63-65, 146-148, 159-161, 190-192) over this repository's containers: what a translation unit needs to declare
// or define functions with the reference's signatures (xs_launchers.hpp, and the shim shown in INTEGRATION.md,
// which tests/test_abi_cpu.py compiles against this header).
#pragma once
#include "device_array.hpp"
#include <complex>

// --- POD argument types (Internal.h:19-59, 63-65, 146-148, 159-161, 190-192) -----------------
using ushort = unsigned short;
using floatType = float;
using floatTypeICP = double;
using hostComplex = std::complex<floatType>;
using hostComplexICP = std::complex<floatTypeICP>;
struct devComplex { float re, im; };                 // (re, im); cuda::std::complex<float> in the reference
struct devComplexICP { double re, im; };
struct devDComplex { float re_re, re_im, im_re, im_im; };
using MapArr = DeviceArray2D<devComplex>;
#define H_ 1e-7
#define invH_ 1e7
// int3 / float2 / float3 / float4 come from <hip/hip_runtime.h>
struct Intr {
    float fx, fy, cx, cy;
    Intr() {}
    Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
    Intr operator()(int level_index) const {
        int div = 1 << level_index;
        return (Intr(fx / div, fy / div, cx / div, cy / div));
    }
};
struct devComplex3 { devComplex x, y, z; };
struct MatS33 { devComplex3 data[3]; };
struct devDComplex3 { devDComplex x, y, z; };
struct MatD33 { devDComplex3 data[3]; };
struct Mat33 { float3 data[3]; };
static_assert(sizeof(MatS33) == 72 && sizeof(devComplex3) == 24 && sizeof(MatD33) == 144 && sizeof(devDComplex3) == 48, "POD layout");

template <class D, class Matx>
D &device_cast(Matx &matx) { return (*reinterpret_cast<D *>(const_cast<float *>(matx.data()))); }
