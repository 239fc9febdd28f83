// xs_complex.h — complex (CSFD) and dual-complex (DCSFD) scalars for gfx950 kernels and
// the C++ host.  Replaces the reference's DeviceArray math:
//   complex<T>    DeviceArray/include/cuda_complex.hpp:20-881   (and libcu++'s
//                 cuda::std::complex<float>, the kernels' actual type, Internal.h:24)
//   d_complex<T>  DeviceArray/include/cuda_double_complex.hpp:16-260
// Same formulas, same evaluation order, so results carry the same bits wherever the
// underlying libm calls agree; built with -ffp-contract=off (no FMA contraction), like
// the parity oracle.
//
// CDNA4 notes.  The hot kernels live in the CSFD regime: real part O(1), imaginary part
// ~1e-7 x real.  Two reference operations are expensive as written and are given
// bit-equivalent short forms that a wave takes uniformly:
//   * sqrt(z) = polar(sqrt(hypot(a,b)), atan2(b,a)/2).  For a > 0 and |b| <= 2^-13 a,
//     hypot rounds to a, atan2 rounds to b/a, cos(theta) rounds to 1 and sin(theta) to
//     theta, so the result is (sqrt a, sqrt a * (b/a)/2) — one v_sqrt, one divide.
//     Outside that cone the general libm path runs.
//   * z / w scales w by 2^-ilogb(max|c|,|d|) before dividing and scales back.  Power-of-two
//     scaling is exact, so for operands away from the exponent limits the unscaled
//     quotient has the same bits; the scaled path runs only near the limits.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define XS_HD __host__ __device__ __forceinline__
#else
#define XS_HD inline
#endif

namespace xs {

template <class T>
struct cplx {
    T re, im;
    cplx() = default;
    XS_HD cplx(T r) : re(r), im(T(0)) {}
    XS_HD cplx(T r, T i) : re(r), im(i) {}
    XS_HD T real() const { return re; }
    XS_HD T imag() const { return im; }
    XS_HD cplx &operator+=(T r) { re += r; return *this; }
    XS_HD cplx &operator-=(T r) { re -= r; return *this; }
    XS_HD cplx &operator*=(T r) { re *= r; im *= r; return *this; }
    XS_HD cplx &operator/=(T r) { re /= r; im /= r; return *this; }
    XS_HD cplx &operator+=(const cplx &c) { re += c.re; im += c.im; return *this; }
    XS_HD cplx &operator-=(const cplx &c) { re -= c.re; im -= c.im; return *this; }
};
typedef cplx<float> cfloat;
typedef cplx<double> cdouble;

template <class T> XS_HD cplx<T> operator+(cplx<T> x, cplx<T> y) { return cplx<T>(x.re + y.re, x.im + y.im); }
template <class T> XS_HD cplx<T> operator+(cplx<T> x, T y) { return cplx<T>(x.re + y, x.im); }
template <class T> XS_HD cplx<T> operator+(T x, cplx<T> y) { return cplx<T>(y.re + x, y.im); }
template <class T> XS_HD cplx<T> operator-(cplx<T> x, cplx<T> y) { return cplx<T>(x.re - y.re, x.im - y.im); }
template <class T> XS_HD cplx<T> operator-(cplx<T> x, T y) { return cplx<T>(x.re - y, x.im); }
template <class T> XS_HD cplx<T> operator-(cplx<T> x) { return cplx<T>(-x.re, -x.im); }
// scalar - complex is (-y) += x in the reference (cuda_complex.hpp:148-154)
template <class T> XS_HD cplx<T> operator-(T x, cplx<T> y) { return cplx<T>(-y.re + x, -y.im); }

template <class T> XS_HD cplx<T> operator*(cplx<T> z, cplx<T> w) {
    T ac = z.re * w.re, bd = z.im * w.im, ad = z.re * w.im, bc = z.im * w.re;
    return cplx<T>(ac - bd, ad + bc);
}
template <class T> XS_HD cplx<T> operator*(cplx<T> x, T y) { return cplx<T>(x.re * y, x.im * y); }
template <class T> XS_HD cplx<T> operator*(T x, cplx<T> y) { return cplx<T>(y.re * x, y.im * x); }
template <class T> XS_HD cplx<T> operator/(cplx<T> x, T y) { return cplx<T>(x.re / y, x.im / y); }

namespace detail {
XS_HD float xlogb(float x) { return logbf(x); }
XS_HD double xlogb(double x) { return logb(x); }
XS_HD float xscalbn(float x, int n) { return scalbnf(x, n); }
XS_HD double xscalbn(double x, int n) { return scalbn(x, n); }
XS_HD float xhypot(float a, float b) { return hypotf(a, b); }
XS_HD double xhypot(double a, double b) { return hypot(a, b); }
XS_HD float xatan2(float a, float b) { return atan2f(a, b); }
XS_HD double xatan2(double a, double b) { return atan2(a, b); }
XS_HD float xsqrt(float a) { return sqrtf(a); }
XS_HD double xsqrt(double a) { return ::sqrt(a); }
XS_HD float xsin(float a) { return sinf(a); }
XS_HD double xsin(double a) { return ::sin(a); }
XS_HD float xcos(float a) { return cosf(a); }
XS_HD double xcos(double a) { return ::cos(a); }
XS_HD float xexp(float a) { return expf(a); }
XS_HD double xexp(double a) { return ::exp(a); }
XS_HD float xlog(float a) { return logf(a); }
XS_HD double xlog(double a) { return ::log(a); }
XS_HD float xsinh(float a) { return sinhf(a); }
XS_HD double xsinh(double a) { return ::sinh(a); }
XS_HD float xcosh(float a) { return coshf(a); }
XS_HD double xcosh(double a) { return ::cosh(a); }
template <class T> struct lim;
template <> struct lim<float> {
    XS_HD static float lo() { return 0x1p-40f; }
    XS_HD static float hi() { return 0x1p40f; }
    XS_HD static float cone() { return 0x1p-13f; }
};
template <> struct lim<double> {
    XS_HD static double lo() { return 0x1p-300; }
    XS_HD static double hi() { return 0x1p300; }
    XS_HD static double cone() { return 0x1p-28; }
};
}  // namespace detail

// scaled quotient, cuda_complex.hpp:285-326 (the float form drops the NaN recovery; the
// recovery never triggers for finite operands, so one body serves both)
template <class T> XS_HD cplx<T> div_scaled(cplx<T> z, cplx<T> w) {
    int ilogbw = 0;
    T a = z.re, b = z.im, c = w.re, d = w.im;
    T logbw = detail::xlogb(fmax(fabs(c), fabs(d)));
    if (isfinite(logbw)) {
        ilogbw = (int)logbw;
        c = detail::xscalbn(c, -ilogbw);
        d = detail::xscalbn(d, -ilogbw);
    }
    T denom = c * c + d * d;
    T x = detail::xscalbn((a * c + b * d) / denom, -ilogbw);
    T y = detail::xscalbn((b * c - a * d) / denom, -ilogbw);
    return cplx<T>(x, y);
}
template <class T> XS_HD cplx<T> operator/(cplx<T> z, cplx<T> w) {
    T mw = fmax(fabs(w.re), fabs(w.im));
    T mz = fmax(fabs(z.re), fabs(z.im));
    typedef detail::lim<T> L;
    if (__builtin_expect(mw >= L::lo() && mw <= L::hi() && mz >= L::lo() && mz <= L::hi(), 1)) {   // (likely: the scaled path is laid out of line)
        T denom = w.re * w.re + w.im * w.im;
        return cplx<T>((z.re * w.re + z.im * w.im) / denom, (z.im * w.re - z.re * w.im) / denom);
    }
    return div_scaled(z, w);
}
template <class T> XS_HD cplx<T> operator/(T x, cplx<T> y) { return cplx<T>(x) / y; }

template <class T> XS_HD bool operator==(cplx<T> x, cplx<T> y) { return x.re == y.re && x.im == y.im; }
template <class T> XS_HD bool operator==(cplx<T> x, T y) { return x.re == y && x.im == 0; }

template <class T> XS_HD T abs(cplx<T> c) { return detail::xhypot(c.re, c.im); }
template <class T> XS_HD T arg(cplx<T> c) { return detail::xatan2(c.im, c.re); }
template <class T> XS_HD T norm(cplx<T> c) {
    if (isinf(c.re)) return fabs(c.re);
    if (isinf(c.im)) return fabs(c.im);
    return c.re * c.re + c.im * c.im;
}
template <class T> XS_HD cplx<T> conj(cplx<T> c) { return cplx<T>(c.re, -c.im); }

template <class T> XS_HD cplx<T> polar(T rho, T theta = T(0)) {
    if (isnan(rho) || signbit(rho)) return cplx<T>(T(NAN), T(NAN));
    if (isnan(theta)) {
        if (isinf(rho)) return cplx<T>(rho, theta);
        return cplx<T>(theta, theta);
    }
    if (isinf(theta)) {
        if (isinf(rho)) return cplx<T>(rho, T(NAN));
        return cplx<T>(T(NAN), T(NAN));
    }
    T x = rho * detail::xcos(theta);
    if (isnan(x)) x = 0;
    T y = rho * detail::xsin(theta);
    if (isnan(y)) y = 0;
    return cplx<T>(x, y);
}

template <class T> XS_HD cplx<T> sqrt_general(cplx<T> x) {
    if (isinf(x.im)) return cplx<T>(T(INFINITY), x.im);
    if (isinf(x.re)) {
        if (x.re > T(0)) return cplx<T>(x.re, isnan(x.im) ? x.im : copysign(T(0), x.im));
        return cplx<T>(isnan(x.im) ? x.im : T(0), copysign(x.re, x.im));
    }
    return polar(detail::xsqrt(abs(x)), arg(x) / T(2));
}
template <class T> XS_HD cplx<T> sqrt(cplx<T> x) {
    typedef detail::lim<T> L;
    T a = x.re, b = x.im;
    if (__builtin_expect(a >= L::lo() && a <= L::hi() && fabs(b) <= a * L::cone(), 1)) {
        T s = detail::xsqrt(a);
        return cplx<T>(s, s * ((b / a) / T(2)));
    }
    return sqrt_general(x);
}

template <class T> XS_HD cplx<T> log(cplx<T> x) { return cplx<T>(detail::xlog(abs(x)), arg(x)); }
template <class T> XS_HD cplx<T> exp(cplx<T> x) {
    T i = x.im;
    if (isinf(x.re)) {
        if (x.re < T(0)) {
            if (!isfinite(i)) i = T(1);
        } else if (i == 0 || !isfinite(i)) {
            if (isinf(i)) i = T(NAN);
            return cplx<T>(x.re, i);
        }
    } else if (isnan(x.re) && x.im == 0)
        return x;
    T e = detail::xexp(x.re);
    return cplx<T>(e * detail::xcos(i), e * detail::xsin(i));
}
template <class T> XS_HD cplx<T> pow(cplx<T> x, cplx<T> y) { return exp(y * log(x)); }
template <class T> XS_HD cplx<T> pow(cplx<T> x, T y) { return pow(x, cplx<T>(y)); }
template <class T> XS_HD cplx<T> sinh(cplx<T> x) {
    if (isinf(x.re) && !isfinite(x.im)) return cplx<T>(x.re, T(NAN));
    if (x.re == 0 && !isfinite(x.im)) return cplx<T>(x.re, T(NAN));
    if (x.im == 0 && !isfinite(x.re)) return x;
    return cplx<T>(detail::xsinh(x.re) * detail::xcos(x.im), detail::xcosh(x.re) * detail::xsin(x.im));
}
template <class T> XS_HD cplx<T> sinh_new(cplx<T> x) {  // cuda_complex.hpp:740-751
    if (isinf(x.re) && !isfinite(x.im)) return cplx<T>(x.re, T(NAN));
    if (x.re == 0 && !isfinite(x.im)) return cplx<T>(x.re, T(NAN));
    if (x.im == 0 && !isfinite(x.re)) return x;
    return cplx<T>(detail::xsinh(x.re), detail::xcosh(x.re) * detail::xsin(x.im));
}
template <class T> XS_HD cplx<T> cosh(cplx<T> x) {
    if (isinf(x.re) && !isfinite(x.im)) return cplx<T>(fabs(x.re), T(NAN));
    if (x.re == 0 && !isfinite(x.im)) return cplx<T>(T(NAN), x.re);
    if (x.re == 0 && x.im == 0) return cplx<T>(T(1), x.im);
    if (x.im == 0 && !isfinite(x.re)) return cplx<T>(fabs(x.re), x.im);
    return cplx<T>(detail::xcosh(x.re) * detail::xcos(x.im), detail::xsinh(x.re) * detail::xsin(x.im));
}
template <class T> XS_HD cplx<T> sin(cplx<T> x) { cplx<T> z = sinh(cplx<T>(-x.im, x.re)); return cplx<T>(z.im, -z.re); }
template <class T> XS_HD cplx<T> sin_new(cplx<T> x) {  // cuda_complex.hpp:855-862
    return cplx<T>(detail::xsin(x.re), detail::xsinh(x.im) * detail::xcos(x.re));
}
template <class T> XS_HD cplx<T> cos(cplx<T> x) { return cosh(cplx<T>(-x.im, x.re)); }

// ---- the rest of the reference's surface (none of it on the kernels' path) ---------------------
template <class T> XS_HD cplx<T> proj(cplx<T> c) {  // cuda_complex.hpp:508-516
    if (isinf(c.re) || isinf(c.im)) return cplx<T>(T(INFINITY), copysign(T(0), c.im));
    return c;
}
template <class T> XS_HD cplx<T> log10(cplx<T> x) { return log(x) / detail::xlog(T(10)); }  // :572-577
template <class T> XS_HD cplx<T> tanh(cplx<T> x) {  // :772-787
    if (isinf(x.re)) {
        if (!isfinite(x.im)) return cplx<T>(T(1), T(0));
        return cplx<T>(T(1), copysign(T(0), detail::xsin(T(2) * x.im)));
    }
    if (isnan(x.re) && x.im == 0) return x;
    T r2 = T(2) * x.re, i2 = T(2) * x.im;
    T d = detail::xcosh(r2) + detail::xcos(i2);
    return cplx<T>(detail::xsinh(r2) / d, detail::xsin(i2) / d);
}
template <class T> XS_HD cplx<T> tan(cplx<T> x) { cplx<T> z = tanh(cplx<T>(-x.im, x.re)); return cplx<T>(z.im, -z.re); }  // :875-881
namespace detail {
template <class T> XS_HD T pi() { return T(3.14159265358979323846); }  // the reference's atan2(+0., -0.) rounded to T
}
template <class T> XS_HD cplx<T> asinh(cplx<T> x) {  // :642-665
    const T pi = detail::pi<T>();
    if (isinf(x.re)) {
        if (isnan(x.im)) return x;
        if (isinf(x.im)) return cplx<T>(x.re, copysign(pi * T(0.25), x.im));
        return cplx<T>(x.re, copysign(T(0), x.im));
    }
    if (isnan(x.re)) {
        if (isinf(x.im)) return cplx<T>(x.im, x.re);
        if (x.im == 0) return x;
        return cplx<T>(x.re, x.re);
    }
    if (isinf(x.im)) return cplx<T>(copysign(x.im, x.re), copysign(pi / T(2), x.im));
    cplx<T> z = log(x + sqrt(pow(x, T(2)) + T(1)));
    return cplx<T>(copysign(z.re, x.re), copysign(z.im, x.im));
}
template <class T> XS_HD cplx<T> acosh(cplx<T> x) {  // :669-695
    const T pi = detail::pi<T>();
    if (isinf(x.re)) {
        if (isnan(x.im)) return cplx<T>(fabs(x.re), x.im);
        if (isinf(x.im)) {
            if (x.re > 0) return cplx<T>(x.re, copysign(pi * T(0.25), x.im));
            return cplx<T>(-x.re, copysign(pi * T(0.75), x.im));
        }
        if (x.re < 0) return cplx<T>(-x.re, copysign(pi, x.im));
        return cplx<T>(x.re, copysign(T(0), x.im));
    }
    if (isnan(x.re)) {
        if (isinf(x.im)) return cplx<T>(fabs(x.im), x.re);
        return cplx<T>(x.re, x.re);
    }
    if (isinf(x.im)) return cplx<T>(fabs(x.im), copysign(pi / T(2), x.im));
    cplx<T> z = log(x + sqrt(pow(x, T(2)) - T(1)));
    return cplx<T>(copysign(z.re, T(0)), copysign(z.im, x.im));
}
template <class T> XS_HD cplx<T> atanh(cplx<T> x) {  // :699-723
    const T pi = detail::pi<T>();
    if (isinf(x.im)) return cplx<T>(copysign(T(0), x.re), copysign(pi / T(2), x.im));
    if (isnan(x.im)) {
        if (isinf(x.re) || x.re == 0) return cplx<T>(copysign(T(0), x.re), x.im);
        return cplx<T>(x.im, x.im);
    }
    if (isnan(x.re)) return cplx<T>(x.re, x.re);
    if (isinf(x.re)) return cplx<T>(copysign(T(0), x.re), copysign(pi / T(2), x.im));
    if (fabs(x.re) == T(1) && x.im == T(0)) return cplx<T>(copysign(T(INFINITY), x.re), copysign(T(0), x.im));
    cplx<T> z = log((T(1) + x) / (T(1) - x)) / T(2);
    return cplx<T>(copysign(z.re, x.re), copysign(z.im, x.im));
}
template <class T> XS_HD cplx<T> asin(cplx<T> x) { cplx<T> z = asinh(cplx<T>(-x.im, x.re)); return cplx<T>(z.im, -z.re); }  // :791-797
template <class T> XS_HD cplx<T> acos(cplx<T> x) {  // :801-831
    const T pi = detail::pi<T>();
    if (isinf(x.re)) {
        if (isnan(x.im)) return cplx<T>(x.im, x.re);
        if (isinf(x.im)) {
            if (x.re < T(0)) return cplx<T>(T(0.75) * pi, -x.im);
            return cplx<T>(T(0.25) * pi, -x.im);
        }
        if (x.re < T(0)) return cplx<T>(pi, signbit(x.im) ? -x.re : x.re);
        return cplx<T>(T(0), signbit(x.im) ? x.re : -x.re);
    }
    if (isnan(x.re)) {
        if (isinf(x.im)) return cplx<T>(x.re, -x.im);
        return cplx<T>(x.re, x.re);
    }
    if (isinf(x.im)) return cplx<T>(pi / T(2), -x.im);
    if (x.re == 0) return cplx<T>(pi / T(2), -x.im);
    cplx<T> z = log(x + sqrt(pow(x, T(2)) - T(1)));
    if (signbit(x.im)) return cplx<T>(fabs(z.im), fabs(z.re));
    return cplx<T>(fabs(z.im), -fabs(z.re));
}
template <class T> XS_HD cplx<T> atan(cplx<T> x) { cplx<T> z = atanh(cplx<T>(-x.im, x.re)); return cplx<T>(z.im, -z.re); }  // :835-841

// ---------------------------------------------------------------------------------------
// dual complex a + b*j, a and b complex; 16 B for T = float, member order re.re re.im
// im.re im.im (cuda_double_complex.hpp:24-31)
template <class T>
struct dcplx {
    cplx<T> re, im;
    dcplx() = default;
    XS_HD explicit dcplx(T rr, T ri = 0, T ir = 0, T ii = 0) : re(rr, ri), im(ir, ii) {}
    XS_HD explicit dcplx(cplx<T> r, cplx<T> i) : re(r), im(i) {}
    XS_HD cplx<T> norm() const { return re * re + im * im; }
    XS_HD T value() const { return re.re; }
    XS_HD T grad() const { return re.im; }
    XS_HD T hessian() const { return im.im; }
};
typedef dcplx<float> dcfloat;
template <class T> XS_HD dcplx<T> operator-(dcplx<T> x) { return dcplx<T>(-x.re, -x.im); }
template <class T> XS_HD dcplx<T> operator+(dcplx<T> l, T r) { return dcplx<T>(l.re + r, l.im); }
template <class T> XS_HD dcplx<T> operator-(dcplx<T> l, T r) { return dcplx<T>(l.re - r, l.im); }
template <class T> XS_HD dcplx<T> operator*(dcplx<T> l, T r) { return dcplx<T>(l.re * r, l.im * r); }
template <class T> XS_HD dcplx<T> operator/(dcplx<T> l, T r) { return dcplx<T>(l.re / r, l.im / r); }
template <class T> XS_HD dcplx<T> operator-(T l, dcplx<T> r) { return dcplx<T>(-r.re + l, -r.im); }
template <class T> XS_HD dcplx<T> operator*(dcplx<T> l, cplx<T> r) { return dcplx<T>(l.re * r, l.im * r); }
template <class T> XS_HD dcplx<T> operator+(dcplx<T> l, dcplx<T> r) { return dcplx<T>(l.re + r.re, l.im + r.im); }
template <class T> XS_HD dcplx<T> operator-(dcplx<T> l, dcplx<T> r) { return dcplx<T>(l.re - r.re, l.im - r.im); }
template <class T> XS_HD dcplx<T> operator*(dcplx<T> l, dcplx<T> r) {  // :119-125
    return dcplx<T>(l.re * r.re - l.im * r.im, l.im * r.re + l.re * r.im);
}
template <class T> XS_HD dcplx<T> operator/(dcplx<T> l, dcplx<T> o) {  // :126-133
    const cplx<T> r = l.re * o.re + l.im * o.im;
    const cplx<T> n = o.norm();
    return dcplx<T>(r / n, (l.im * o.re - l.re * o.im) / n);
}
template <class T> XS_HD cplx<T> abs(dcplx<T> x) { return sqrt(x.re * x.re + x.im * x.im); }  // :233-239
template <class T> XS_HD dcplx<T> sqrt(dcplx<T> x) {  // :242-260
    dcplx<T> result = x;
    cplx<T> r = abs(x);
    cplx<T> sqrt_r = sqrt(r);
    result.re = result.re + r;
    cplx<T> zrnorm = abs(result);
    if (fabs(zrnorm.re) < (T)1e-20 && fabs(zrnorm.im) < (T)1e-20) return result * sqrt_r;
    cplx<T> scale = sqrt_r / zrnorm;
    return result * scale;
}

}  // namespace xs
