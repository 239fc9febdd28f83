// xs_csfd.hip — elementwise CSFD / DCSFD kernels over DeviceArray<complex> storage.  The GPU
// form of the reference's CPU DeviceArray demo (Experiments/test_CSFD/main.cpp): the five
// "standard" (raw) and "accelerated" (our, O(h^2) terms dropped) scalar kernels of :18-86
// applied to arrays instead of one scalar pair, and f1(x, y) = (x + y)^2 of :8-11 in
// dual-complex arithmetic (DeviceArray/src/DoubleComplex.cpp:155-162).
// Pure streaming: 16 B in + 8 B out per element for the binary ops; two complex per lane per
// access (16 B) so a wave moves 1 KiB per instruction.
#include "xs_device.h"
#include "xs_env.h"
#include "../../include/xslam_amd.h"

using namespace xs;

namespace {
__device__ __forceinline__ cfloat mul_our(cfloat a, cfloat b) { return cfloat(a.re * b.re, a.im * b.re + a.re * b.im); }
__device__ __forceinline__ cfloat mul_raw(cfloat a, cfloat b) { return cfloat(a.re * b.re - a.im * b.im, a.im * b.re + a.re * b.im); }
__device__ __forceinline__ cfloat div_our(cfloat a, cfloat b) {
    return cfloat(a.re / b.re, (a.im * b.re - a.re * b.im) / (b.re * b.re + b.im * b.im));
}
__device__ __forceinline__ cfloat div_raw(cfloat a, cfloat b) {
    return cfloat((a.re * b.re + a.im * b.im) / (b.re * b.re + b.im * b.im), (a.im * b.re - a.re * b.im) / (b.re * b.re + b.im * b.im));
}
// (sine and cosine of one argument through sincosf: one range reduction, the same two values)
__device__ __forceinline__ cfloat exp_our(cfloat a) { return cfloat(expf(a.re), expf(a.re) * sinf(a.im)); }
__device__ __forceinline__ cfloat exp_raw(cfloat a) { float sn, cs; sincosf(a.im, &sn, &cs); return cfloat(expf(a.re) * cs, expf(a.re) * sn); }
__device__ __forceinline__ cfloat sin_our(cfloat a) { float sn, cs; sincosf(a.re, &sn, &cs); return cfloat(sn, -sinhf(-a.im) * cs); }
__device__ __forceinline__ cfloat sin_raw(cfloat a) { float sn, cs; sincosf(a.re, &sn, &cs); return cfloat(sn * coshf(-a.im), -sinhf(-a.im) * cs); }
// std::pow(float, int) promotes to double in the reference's host code (main.cpp:74-86): pow(double(x), double(n)).  For the small integer
// exponents of the demo (3) the double power is formed by multiplication — at most |n| - 1 roundings of 2^-53 each, i.e. closer to the exact
// power than a general pow() implementation promises (glibc's and the device library's agree with each other only to their own last bits), and a
// float result that differs from a correctly rounded pow's with probability ~1e-8 per element; the general double pow() of the device library
// is two hundred instructions per call and made this op VALU-bound at 1.8 TB/s.  Larger exponents take pow().
__device__ __forceinline__ double pow_int(double x, int n) {
    const int m = n < 0 ? -n : n;
    if (m > 8) return pow(x, (double)n);
    double r = 1.0, b = x;
    for (int k = m; k; k >>= 1) { if (k & 1) r = r * b; b = b * b; }   // (n = 3: x * x^2, two roundings)
    return n < 0 ? 1.0 / r : r;
}
__device__ __forceinline__ cfloat pow_our(cfloat a, int n) {
    const float nr = a.re * a.re + a.im * a.im, ar = atan2f(a.im, a.re);
    return cfloat((float)pow_int((double)a.re, n), (float)(pow_int((double)nr, n) * (double)sinf(n * ar)));
}
__device__ __forceinline__ cfloat pow_raw(cfloat a, int n) {
    const float nr = a.re * a.re + a.im * a.im, ar = atan2f(a.im, a.re);
    return cfloat((float)(pow_int((double)nr, n) * (double)cosf(n * ar)), (float)(pow_int((double)nr, n) * (double)sinf(n * ar)));
}
template <int WHICH, int OUR> __device__ __forceinline__ cfloat apply(cfloat x, cfloat y) {
    if (WHICH == 0) return OUR ? mul_our(x, y) : mul_raw(x, y);
    if (WHICH == 1) return OUR ? div_our(x, y) : div_raw(x, y);
    const cfloat s = x + y;
    if (WHICH == 2) return OUR ? exp_our(s) : exp_raw(s);
    if (WHICH == 3) return OUR ? sin_our(s) : sin_raw(s);
    return OUR ? pow_our(s, 3) : pow_raw(s, 3);
}
}  // namespace

// NT: arrays that cannot stay in the 256 MiB Infinity Cache (launcher: more than 128 MB moved) are read and written with nontemporal
// accesses — each element is touched once, and not allocating its lines spares the eviction of what the previous launch left dirty
// (64 M elements, 1.5 GB per launch: mul 5.1 -> see profiles/r06_csfd_nt.txt; the same lever as the residual kernels' scan)
typedef float xs_v4 __attribute__((ext_vector_type(4)));
template <int WHICH, int OUR, bool NT, int U>
__global__ void __launch_bounds__(256) k_csfd(const cfloat *a, const cfloat *b, cfloat *out, long n) {
    // a workgroup takes U consecutive 4 KiB pieces of each array per round (a lane's U requests per array lie 4 KiB apart and are issued before
    // the first result is needed); the rounds of a workgroup lie gridDim.x * U pieces apart
    const long n2 = n / 2;
    const long round = (long)gridDim.x * 256 * U;
    const xs_v4 *a4 = reinterpret_cast<const xs_v4 *>(a), *b4 = reinterpret_cast<const xs_v4 *>(b);
    xs_v4 *o4 = reinterpret_cast<xs_v4 *>(out);
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n2; base += round) {
        xs_v4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < n2) {
                x[u] = NT ? __builtin_nontemporal_load(a4 + i) : a4[i];
                y[u] = NT ? __builtin_nontemporal_load(b4 + i) : b4[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < n2) {
                const cfloat r0 = apply<WHICH, OUR>(cfloat(x[u].x, x[u].y), cfloat(y[u].x, y[u].y));
                const cfloat r1 = apply<WHICH, OUR>(cfloat(x[u].z, x[u].w), cfloat(y[u].z, y[u].w));
                const xs_v4 r = {r0.re, r0.im, r1.re, r1.im};
                if (NT) __builtin_nontemporal_store(r, o4 + i); else o4[i] = r;
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = apply<WHICH, OUR>(a[n - 1], b[n - 1]);
}

__global__ void __launch_bounds__(256) k_dcsfd_f1(const dcfloat *x, const dcfloat *y, dcfloat *out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const dcfloat s = x[i] + y[i];
        out[i] = s * s;
    }
}

template <int W, int O> static void launch(const float *a, const float *b, float *out, long n, hipStream_t s) {
    static const int env_nt = exp_env_int("XS_CSFD_NT", -1);   // A/B aid: 0 never, 1 always
    static const int env_u = exp_env_int("XS_CSFD_UNROLL", 0);  // A/B aid: pieces per round (1, 2, 4, 8)
    static const int env_blocks = exp_env_int("XS_CSFD_BLOCKS", 0);
    const bool nt = env_nt < 0 ? n * 24 > (128L << 20) : env_nt != 0;
    // pieces per round and workgroups (profiles/r06_ab_csfd_unroll.txt, 64 M elements: one piece per round and 2048 workgroups 5.4-5.6 TB/s for the
    // product, 4.5 for exp, 3.5-3.6 for sin; four pieces and 4096 workgroups 6.2 / 6.2 / 5.2; at 1e6 elements — cache-resident, launches of 4 us —
    // two pieces help sin and exp and leave the rest alone)
    const int U = env_u == 1 || env_u == 2 || env_u == 4 || env_u == 8 ? env_u : (nt ? 4 : 2);
    long blocks = (n / 2 + 256 * U - 1) / (256 * U);
    const long cap = env_blocks > 0 ? env_blocks : (nt ? 4096 : 2048);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(256);
#define XS_CSFD_LAUNCH(NTV, UV) hipLaunchKernelGGL((k_csfd<W, O, NTV, UV>), grid, block, 0, s, (const cfloat *)a, (const cfloat *)b, (cfloat *)out, n)
    if (nt) { if (U == 8) XS_CSFD_LAUNCH(true, 8); else if (U == 4) XS_CSFD_LAUNCH(true, 4); else if (U == 2) XS_CSFD_LAUNCH(true, 2); else XS_CSFD_LAUNCH(true, 1); }
    else    { if (U == 8) XS_CSFD_LAUNCH(false, 8); else if (U == 4) XS_CSFD_LAUNCH(false, 4); else if (U == 2) XS_CSFD_LAUNCH(false, 2); else XS_CSFD_LAUNCH(false, 1); }
#undef XS_CSFD_LAUNCH
}

/* test_CSFD part 1 over arrays.  which: 0 multiplication, 1 division, 2 exp(a+b), 3 sin(a+b),
 * 4 pow(a+b, 3); our: 0 = *_raw, 1 = *_our (Experiments/test_CSFD/main.cpp:18-86).
 * a, b, out: n complex<float> as (re, im) pairs, 16-byte aligned. */
extern "C" int xs_csfd_array_op(int which, int our, const float *a, const float *b, float *out, long n, void *stream) {
    if (!a || !b || !out) return xs_set_error(hipErrorInvalidValue, "xs_csfd_array_op: null pointer");
    if (which < 0 || which > 4) return xs_set_error(hipErrorInvalidValue, "xs_csfd_array_op: bad op");
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (which * 2 + (our ? 1 : 0)) {
        case 0: launch<0, 0>(a, b, out, n, s); break;
        case 1: launch<0, 1>(a, b, out, n, s); break;
        case 2: launch<1, 0>(a, b, out, n, s); break;
        case 3: launch<1, 1>(a, b, out, n, s); break;
        case 4: launch<2, 0>(a, b, out, n, s); break;
        case 5: launch<2, 1>(a, b, out, n, s); break;
        case 6: launch<3, 0>(a, b, out, n, s); break;
        case 7: launch<3, 1>(a, b, out, n, s); break;
        case 8: launch<4, 0>(a, b, out, n, s); break;
        case 9: launch<4, 1>(a, b, out, n, s); break;
    }
    XS_CHECK(hipGetLastError());
    return 0;
}

/* f1(x, y) = (x + y) * (x + y) in dual-complex arithmetic (test_CSFD/main.cpp:8-11); x, y, out:
 * n groups of (re.re, re.im, im.re, im.im). */
extern "C" int xs_dcsfd_f1(const float *x, const float *y, float *out, long n, void *stream) {
    if (!x || !y || !out) return xs_set_error(hipErrorInvalidValue, "xs_dcsfd_f1: null pointer");
    if (n <= 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_dcsfd_f1, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const dcfloat *)x, (const dcfloat *)y, (dcfloat *)out, n);
    XS_CHECK(hipGetLastError());
    return 0;
}

// scalar op tables through the device math header (parity tests of xs_complex.h on the GPU)
template <class F> __global__ void k_table(const cfloat *a, const cfloat *b, cfloat *out, long n, F f) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f(a[i], b[i]);
}
__global__ void k_ctable(int op, const cfloat *a, const cfloat *b, cfloat *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cfloat x = a[i], y = b[i];
    cfloat r(0.f, 0.f);
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        case 4: r = sqrt(x); break;
        case 5: r = cfloat(abs(x), 0.f); break;
        case 6: r = exp(x); break;
        case 7: r = log(x); break;
        case 8: r = pow(x, y); break;
        case 9: r = sin(x); break;
        case 10: r = cos(x); break;
        case 11: r = sinh(x); break;
        case 12: r = cosh(x); break;
        case 13: r = sin_new(x); break;
        case 14: r = sinh_new(x); break;
        case 15: r = cfloat(norm(x), 0.f); break;
        case 16: r = cfloat(arg(x), 0.f); break;
        case 17: r = conj(x); break;
        case 18: r = polar(x.re, y.re); break;
        case 19: r = x / y.re; break;
        case 20: r = y.re / x; break;
        case 21: r = x * y.re; break;
        case 22: r = y.re - x; break;
        case 23: r = proj(x); break;
        case 24: r = log10(x); break;
        case 25: r = tanh(x); break;
        case 26: r = tan(x); break;
        case 27: r = asinh(x); break;
        case 28: r = acosh(x); break;
        case 29: r = atanh(x); break;
        case 30: r = asin(x); break;
        case 31: r = acos(x); break;
        case 32: r = atan(x); break;
    }
    out[i] = r;
}
__global__ void k_dtable(int op, const dcfloat *a, const dcfloat *b, dcfloat *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dcfloat x = a[i], y = b[i];
    dcfloat r(0.f);
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        case 4: r = sqrt(x); break;
        case 5: r = dcfloat(abs(x), cfloat(0.f, 0.f)); break;
        case 6: r = x * y.value(); break;
        case 7: r = x / y.value(); break;
        case 8: r = x + y.value(); break;
        case 9: r = y.value() - x; break;
    }
    out[i] = r;
}
/* Elementwise tables of the complex<float> (width 2) / d_complex<float> (width 4) operators in
 * csrc/xs_complex.h; op codes as in oracle/oc_capi.cpp (cop / dop).  Exposes the DeviceArray
 * operator API (cuda_complex.hpp:100-881, cuda_double_complex.hpp:137-260) over arrays. */
extern "C" int xs_complex_table(int dual, int op, const float *a, const float *b, float *out, long n, void *stream) {
    if (!a || !b || !out) return xs_set_error(hipErrorInvalidValue, "xs_complex_table: null pointer");
    if (op < 0 || op > (dual ? 9 : 32)) return xs_set_error(hipErrorInvalidValue, "xs_complex_table: bad op");
    if (n <= 0) return 0;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dual)
        hipLaunchKernelGGL(k_dtable, grid, block, 0, (hipStream_t)stream, op, (const dcfloat *)a, (const dcfloat *)b, (dcfloat *)out, n);
    else
        hipLaunchKernelGGL(k_ctable, grid, block, 0, (hipStream_t)stream, op, (const cfloat *)a, (const cfloat *)b, (cfloat *)out, n);
    XS_CHECK(hipGetLastError());
    return 0;
}
