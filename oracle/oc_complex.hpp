// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference's scalar complex / dual-complex arithmetic.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, load or call anything under oracle/.  The product path never does.
//
// Follows (formulas and evaluation order, not text):
//   complex<T>   : DeviceArray/include/cuda_complex.hpp:20-96 (members),
//                  :100-342 (+ - * /), :436-473 (abs arg norm), :536-593
//                  (polar log sqrt), :596-640 (exp pow), :705-751 (sinh
//                  sinh_new cosh), :842-870 (sin sin_new cos), and the rest of
//                  the header's functions (:506-516, :570-577, :640-723, :770-841,
//                  :873-881: proj log10 asinh acosh atanh tanh asin acos atan tan)
//   d_complex<T> : DeviceArray/include/cuda_double_complex.hpp:16-134
//                  (members, compound ops), :137-231 (free ops), :233-260
//                  (abs, sqrt)
// The kernels' real scalar type is cuda::std::complex<float> (libcu++,
// CUDA 11.8; Internal.h:24), which is absent from /root/reference; the in-tree
// header above is of the same libc++ lineage and is what this file restates.
//
// Pinning: checked against oracle/_ref (the reference's own cuda_complex.hpp
// compiled by g++ from where it lies) on seeded operand tables, and against
// the known answers the reference's test_CSFD demo prints (tests/golden/).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oc {

template <class T>
struct cplx {
    T re_, im_;
    cplx(T r = T(), T i = T()) : re_(r), im_(i) {}
    T real() const { return re_; }
    T imag() const { return im_; }
    void real(T r) { re_ = r; }
    void imag(T i) { im_ = i; }
    // scalar compound ops touch both parts only for * and /  (hpp:62-79)
    cplx &operator+=(const T &r) { re_ += r; return *this; }
    cplx &operator-=(const T &r) { re_ -= r; return *this; }
    cplx &operator*=(const T &r) { re_ *= r; im_ *= r; return *this; }
    cplx &operator/=(const T &r) { re_ /= r; im_ /= r; return *this; }
    cplx &operator+=(const cplx &c) { re_ += c.re_; im_ += c.im_; return *this; }
    cplx &operator-=(const cplx &c) { re_ -= c.re_; im_ -= c.im_; return *this; }
    cplx &operator*=(const cplx &c);
    cplx &operator/=(const cplx &c);
};

template <class T> inline cplx<T> operator+(const cplx<T> &x, const cplx<T> &y) { cplx<T> t(x); t += y; return t; }
template <class T> inline cplx<T> operator+(const cplx<T> &x, const T &y) { cplx<T> t(x); t += y; return t; }
template <class T> inline cplx<T> operator+(const T &x, const cplx<T> &y) { cplx<T> t(y); t += x; return t; }
template <class T> inline cplx<T> operator-(const cplx<T> &x, const cplx<T> &y) { cplx<T> t(x); t -= y; return t; }
template <class T> inline cplx<T> operator-(const cplx<T> &x, const T &y) { cplx<T> t(x); t -= y; return t; }
template <class T> inline cplx<T> operator-(const cplx<T> &x) { return cplx<T>(-x.real(), -x.imag()); }
// scalar - complex is built as (-y) += x  (hpp:148-154)
template <class T> inline cplx<T> operator-(const T &x, const cplx<T> &y) { cplx<T> t(-y); t += x; return t; }

// product: four products, one subtraction, one addition; the libc++ NaN/Inf
// recovery is disabled in the reference (hpp:168-228)
template <class T> inline cplx<T> operator*(const cplx<T> &z, const cplx<T> &w) {
    T a = z.real(), b = z.imag(), c = w.real(), d = w.imag();
    T ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    return cplx<T>(ac - bd, ad + bc);
}
template <class T> inline cplx<T> operator*(const cplx<T> &x, const T &y) { cplx<T> t(x); t *= y; return t; }
template <class T> inline cplx<T> operator*(const T &x, const cplx<T> &y) { cplx<T> t(y); t *= x; return t; }

// quotient: scale the divisor by 2^-ilogb(max|c|,|d|), divide, scale back
// (hpp:248-283 generic, :285-326 float specialisation without the recovery)
template <class T> inline cplx<T> operator/(const cplx<T> &z, const cplx<T> &w) {
    int ilogbw = 0;
    T a = z.real(), b = z.imag(), c = w.real(), d = w.imag();
    T logbw = std::logb(std::fmax(std::fabs(c), std::fabs(d)));
    if (std::isfinite(logbw)) {
        ilogbw = static_cast<int>(logbw);
        c = std::scalbn(c, -ilogbw);
        d = std::scalbn(d, -ilogbw);
    }
    T denom = c * c + d * d;
    T x = std::scalbn((a * c + b * d) / denom, -ilogbw);
    T y = std::scalbn((b * c - a * d) / denom, -ilogbw);
    return cplx<T>(x, y);
}
template <class T> inline cplx<T> operator/(const cplx<T> &x, const T &y) { return cplx<T>(x.real() / y, x.imag() / y); }
template <class T> inline cplx<T> operator/(const T &x, const cplx<T> &y) { cplx<T> t(x); t /= y; return t; }

template <class T> inline cplx<T> &cplx<T>::operator*=(const cplx<T> &c) { *this = *this * c; return *this; }
template <class T> inline cplx<T> &cplx<T>::operator/=(const cplx<T> &c) { *this = *this / c; return *this; }

template <class T> inline bool operator==(const cplx<T> &x, const cplx<T> &y) { return x.real() == y.real() && x.imag() == y.imag(); }
template <class T> inline bool operator==(const cplx<T> &x, const T &y) { return x.real() == y && x.imag() == 0; }

template <class T> inline T abs(const cplx<T> &c) { return std::hypot(c.real(), c.imag()); }
template <class T> inline T arg(const cplx<T> &c) { return std::atan2(c.imag(), c.real()); }
template <class T> inline T norm(const cplx<T> &c) {
    if (std::isinf(c.real())) return std::fabs(c.real());
    if (std::isinf(c.imag())) return std::fabs(c.imag());
    return c.real() * c.real() + c.imag() * c.imag();
}
template <class T> inline cplx<T> conj(const cplx<T> &c) { return cplx<T>(c.real(), -c.imag()); }

template <class T> inline cplx<T> polar(const T &rho, const T &theta = T(0)) {
    if (std::isnan(rho) || std::signbit(rho)) return cplx<T>(T(NAN), T(NAN));
    if (std::isnan(theta)) {
        if (std::isinf(rho)) return cplx<T>(rho, theta);
        return cplx<T>(theta, theta);
    }
    if (std::isinf(theta)) {
        if (std::isinf(rho)) return cplx<T>(rho, T(NAN));
        return cplx<T>(T(NAN), T(NAN));
    }
    T x = rho * std::cos(theta);
    if (std::isnan(x)) x = 0;
    T y = rho * std::sin(theta);
    if (std::isnan(y)) y = 0;
    return cplx<T>(x, y);
}

template <class T> inline cplx<T> log(const cplx<T> &x) { return cplx<T>(std::log(abs(x)), arg(x)); }

// sqrt(z) = polar(sqrt|z|, arg z / 2)   (hpp:581-593)
template <class T> inline cplx<T> sqrt(const cplx<T> &x) {
    if (std::isinf(x.imag())) return cplx<T>(T(INFINITY), x.imag());
    if (std::isinf(x.real())) {
        if (x.real() > T(0))
            return cplx<T>(x.real(), std::isnan(x.imag()) ? x.imag() : std::copysign(T(0), x.imag()));
        return cplx<T>(std::isnan(x.imag()) ? x.imag() : T(0), std::copysign(x.real(), x.imag()));
    }
    return polar(std::sqrt(abs(x)), arg(x) / T(2));
}

template <class T> inline cplx<T> exp(const cplx<T> &x) {
    T i = x.imag();
    if (std::isinf(x.real())) {
        if (x.real() < T(0)) {
            if (!std::isfinite(i)) i = T(1);
        } else if (i == 0 || !std::isfinite(i)) {
            if (std::isinf(i)) i = T(NAN);
            return cplx<T>(x.real(), i);
        }
    } else if (std::isnan(x.real()) && x.imag() == 0)
        return x;
    T e = std::exp(x.real());
    return cplx<T>(e * std::cos(i), e * std::sin(i));
}

template <class T> inline cplx<T> pow(const cplx<T> &x, const cplx<T> &y) { return exp(y * log(x)); }
template <class T> inline cplx<T> pow(const cplx<T> &x, const T &y) { return pow(x, cplx<T>(y)); }

template <class T> inline cplx<T> sinh(const cplx<T> &x) {
    if (std::isinf(x.real()) && !std::isfinite(x.imag())) return cplx<T>(x.real(), T(NAN));
    if (x.real() == 0 && !std::isfinite(x.imag())) return cplx<T>(x.real(), T(NAN));
    if (x.imag() == 0 && !std::isfinite(x.real())) return x;
    return cplx<T>(std::sinh(x.real()) * std::cos(x.imag()), std::cosh(x.real()) * std::sin(x.imag()));
}
// the reference's reduced variant: real part keeps sinh only (hpp:740-751)
template <class T> inline cplx<T> sinh_new(const cplx<T> &x) {
    if (std::isinf(x.real()) && !std::isfinite(x.imag())) return cplx<T>(x.real(), T(NAN));
    if (x.real() == 0 && !std::isfinite(x.imag())) return cplx<T>(x.real(), T(NAN));
    if (x.imag() == 0 && !std::isfinite(x.real())) return x;
    return cplx<T>(std::sinh(x.real()), std::cosh(x.real()) * std::sin(x.imag()));
}
template <class T> inline cplx<T> cosh(const cplx<T> &x) {
    if (std::isinf(x.real()) && !std::isfinite(x.imag())) return cplx<T>(std::fabs(x.real()), T(NAN));
    if (x.real() == 0 && !std::isfinite(x.imag())) return cplx<T>(T(NAN), x.real());
    if (x.real() == 0 && x.imag() == 0) return cplx<T>(T(1), x.imag());
    if (x.imag() == 0 && !std::isfinite(x.real())) return cplx<T>(std::fabs(x.real()), x.imag());
    return cplx<T>(std::cosh(x.real()) * std::cos(x.imag()), std::sinh(x.real()) * std::sin(x.imag()));
}
template <class T> inline cplx<T> sin(const cplx<T> &x) {
    cplx<T> z = sinh(cplx<T>(-x.imag(), x.real()));
    return cplx<T>(z.imag(), -z.real());
}
template <class T> inline cplx<T> sin_new(const cplx<T> &x) {
    return cplx<T>(std::sin(x.real()), std::sinh(x.imag()) * std::cos(x.real()));
}
template <class T> inline cplx<T> cos(const cplx<T> &x) { return cosh(cplx<T>(-x.imag(), x.real())); }

// ---- the remaining functions of the header (hpp:506-516 proj, :570-577 log10, :640-723
// asinh acosh atanh, :770-787 tanh, :789-841 asin acos atan, :873-881 tan) -------------
template <class T> inline cplx<T> proj(const cplx<T> &c) {
    cplx<T> r = c;
    if (std::isinf(c.real()) || std::isinf(c.imag())) r = cplx<T>(T(INFINITY), std::copysign(T(0), c.imag()));
    return r;
}
template <class T> inline cplx<T> log10(const cplx<T> &x) { return log(x) / std::log(T(10)); }
template <class T> inline cplx<T> asinh(const cplx<T> &x) {
    const T pi(std::atan2(+0., -0.));
    if (std::isinf(x.real())) {
        if (std::isnan(x.imag())) return x;
        if (std::isinf(x.imag())) return cplx<T>(x.real(), std::copysign(pi * T(0.25), x.imag()));
        return cplx<T>(x.real(), std::copysign(T(0), x.imag()));
    }
    if (std::isnan(x.real())) {
        if (std::isinf(x.imag())) return cplx<T>(x.imag(), x.real());
        if (x.imag() == 0) return x;
        return cplx<T>(x.real(), x.real());
    }
    if (std::isinf(x.imag())) return cplx<T>(std::copysign(x.imag(), x.real()), std::copysign(pi / T(2), x.imag()));
    cplx<T> z = log(x + sqrt(pow(x, T(2)) + T(1)));
    return cplx<T>(std::copysign(z.real(), x.real()), std::copysign(z.imag(), x.imag()));
}
template <class T> inline cplx<T> acosh(const cplx<T> &x) {
    const T pi(std::atan2(+0., -0.));
    if (std::isinf(x.real())) {
        if (std::isnan(x.imag())) return cplx<T>(std::fabs(x.real()), x.imag());
        if (std::isinf(x.imag())) {
            if (x.real() > 0) return cplx<T>(x.real(), std::copysign(pi * T(0.25), x.imag()));
            return cplx<T>(-x.real(), std::copysign(pi * T(0.75), x.imag()));
        }
        if (x.real() < 0) return cplx<T>(-x.real(), std::copysign(pi, x.imag()));
        return cplx<T>(x.real(), std::copysign(T(0), x.imag()));
    }
    if (std::isnan(x.real())) {
        if (std::isinf(x.imag())) return cplx<T>(std::fabs(x.imag()), x.real());
        return cplx<T>(x.real(), x.real());
    }
    if (std::isinf(x.imag())) return cplx<T>(std::fabs(x.imag()), std::copysign(pi / T(2), x.imag()));
    cplx<T> z = log(x + sqrt(pow(x, T(2)) - T(1)));
    return cplx<T>(std::copysign(z.real(), T(0)), std::copysign(z.imag(), x.imag()));
}
template <class T> inline cplx<T> atanh(const cplx<T> &x) {
    const T pi(std::atan2(+0., -0.));
    if (std::isinf(x.imag())) return cplx<T>(std::copysign(T(0), x.real()), std::copysign(pi / T(2), x.imag()));
    if (std::isnan(x.imag())) {
        if (std::isinf(x.real()) || x.real() == 0) return cplx<T>(std::copysign(T(0), x.real()), x.imag());
        return cplx<T>(x.imag(), x.imag());
    }
    if (std::isnan(x.real())) return cplx<T>(x.real(), x.real());
    if (std::isinf(x.real())) return cplx<T>(std::copysign(T(0), x.real()), std::copysign(pi / T(2), x.imag()));
    if (std::fabs(x.real()) == T(1) && x.imag() == T(0))
        return cplx<T>(std::copysign(T(INFINITY), x.real()), std::copysign(T(0), x.imag()));
    cplx<T> z = log((T(1) + x) / (T(1) - x)) / T(2);
    return cplx<T>(std::copysign(z.real(), x.real()), std::copysign(z.imag(), x.imag()));
}
template <class T> inline cplx<T> tanh(const cplx<T> &x) {
    if (std::isinf(x.real())) {
        if (!std::isfinite(x.imag())) return cplx<T>(T(1), T(0));
        return cplx<T>(T(1), std::copysign(T(0), std::sin(T(2) * x.imag())));
    }
    if (std::isnan(x.real()) && x.imag() == 0) return x;
    T r2(T(2) * x.real());
    T i2(T(2) * x.imag());
    T d(std::cosh(r2) + std::cos(i2));
    return cplx<T>(std::sinh(r2) / d, std::sin(i2) / d);
}
template <class T> inline cplx<T> asin(const cplx<T> &x) {
    cplx<T> z = asinh(cplx<T>(-x.imag(), x.real()));
    return cplx<T>(z.imag(), -z.real());
}
template <class T> inline cplx<T> acos(const cplx<T> &x) {
    const T pi(std::atan2(+0., -0.));
    if (std::isinf(x.real())) {
        if (std::isnan(x.imag())) return cplx<T>(x.imag(), x.real());
        if (std::isinf(x.imag())) {
            if (x.real() < T(0)) return cplx<T>(T(0.75) * pi, -x.imag());
            return cplx<T>(T(0.25) * pi, -x.imag());
        }
        if (x.real() < T(0)) return cplx<T>(pi, std::signbit(x.imag()) ? -x.real() : x.real());
        return cplx<T>(T(0), std::signbit(x.imag()) ? x.real() : -x.real());
    }
    if (std::isnan(x.real())) {
        if (std::isinf(x.imag())) return cplx<T>(x.real(), -x.imag());
        return cplx<T>(x.real(), x.real());
    }
    if (std::isinf(x.imag())) return cplx<T>(pi / T(2), -x.imag());
    if (x.real() == 0) return cplx<T>(pi / T(2), -x.imag());
    cplx<T> z = log(x + sqrt(pow(x, T(2)) - T(1)));
    if (std::signbit(x.imag())) return cplx<T>(std::fabs(z.imag()), std::fabs(z.real()));
    return cplx<T>(std::fabs(z.imag()), -std::fabs(z.real()));
}
template <class T> inline cplx<T> atan(const cplx<T> &x) {
    cplx<T> z = atanh(cplx<T>(-x.imag(), x.real()));
    return cplx<T>(z.imag(), -z.real());
}
template <class T> inline cplx<T> tan(const cplx<T> &x) {
    cplx<T> z = tanh(cplx<T>(-x.imag(), x.real()));
    return cplx<T>(z.imag(), -z.real());
}

// ---------------------------------------------------------------------------
// dual complex a + b*j with a, b complex  (cuda_double_complex.hpp)
// value = re.re, gradient seed lives in re.im, Hessian falls out of im.im.
// C is the single-complex type, so the same text runs over oc::cplx<T> and,
// in oracle/_ref, over the reference's own ::complex<T>.
template <class C>
struct dcplx {
    typedef decltype(C().real()) T;
    C re_, im_;
    explicit dcplx(T rr = 0, T ri = 0, T ir = 0, T ii = 0) : re_(rr, ri), im_(ir, ii) {}
    explicit dcplx(C re, C im) : re_(re), im_(im) {}
    explicit dcplx(C re) : re_(re), im_(T(0)) {}
    C real() const { return re_; }
    C imag() const { return im_; }
    void real(C r) { re_ = r; }
    void imag(C i) { im_ = i; }
    C norm() const { return re_ * re_ + im_ * im_; }
    T value() const { return re_.real(); }
    T grad() const { return re_.imag(); }
    T hessian() const { return im_.imag(); }
    dcplx &operator+=(const T &r) { re_ += r; return *this; }
    dcplx &operator-=(const T &r) { re_ -= r; return *this; }
    dcplx &operator*=(const T &r) { re_ *= r; im_ *= r; return *this; }
    dcplx &operator/=(const T &r) { re_ /= r; im_ /= r; return *this; }
    dcplx &operator+=(const C &r) { re_ += r; return *this; }
    dcplx &operator-=(const C &r) { re_ -= r; return *this; }
    dcplx &operator*=(const C &r) { re_ *= r; im_ *= r; return *this; }
    dcplx &operator/=(const C &r) { re_ /= r; im_ /= r; return *this; }
    dcplx &operator+=(const dcplx &o) { re_ += o.re_; im_ += o.im_; return *this; }
    dcplx &operator-=(const dcplx &o) { re_ -= o.re_; im_ -= o.im_; return *this; }
    dcplx &operator*=(const dcplx &o) {  // hpp:119-125
        C real = re_ * o.re_ - im_ * o.im_;
        C imag = im_ * o.re_ + re_ * o.im_;
        re_ = real; im_ = imag;
        return *this;
    }
    dcplx &operator/=(const dcplx &o) {  // hpp:126-133
        const C r = re_ * o.re_ + im_ * o.im_;
        const C n = o.norm();
        im_ = (im_ * o.re_ - re_ * o.im_) / n;
        re_ = r / n;
        return *this;
    }
};
template <class C> inline dcplx<C> operator-(const dcplx<C> &x) { return dcplx<C>(-x.real(), -x.imag()); }
template <class C> inline dcplx<C> operator+(const dcplx<C> &l, const typename dcplx<C>::T &r) { dcplx<C> t(l); t += r; return t; }
template <class C> inline dcplx<C> operator-(const dcplx<C> &l, const typename dcplx<C>::T &r) { dcplx<C> t(l); t -= r; return t; }
template <class C> inline dcplx<C> operator*(const dcplx<C> &l, const typename dcplx<C>::T &r) { dcplx<C> t(l); t *= r; return t; }
template <class C> inline dcplx<C> operator/(const dcplx<C> &l, const typename dcplx<C>::T &r) { dcplx<C> t(l); t /= r; return t; }
template <class C> inline dcplx<C> operator-(const typename dcplx<C>::T &l, const dcplx<C> &r) { dcplx<C> t(-r); t += l; return t; }
template <class C> inline dcplx<C> operator+(const dcplx<C> &l, const dcplx<C> &r) { dcplx<C> t(l); t += r; return t; }
template <class C> inline dcplx<C> operator-(const dcplx<C> &l, const dcplx<C> &r) { dcplx<C> t(l); t -= r; return t; }
template <class C> inline dcplx<C> operator*(const dcplx<C> &l, const dcplx<C> &r) { dcplx<C> t(l); t *= r; return t; }
template <class C> inline dcplx<C> operator/(const dcplx<C> &l, const dcplx<C> &r) { dcplx<C> t(l); t /= r; return t; }

// |z| as a single complex: sqrt(re^2 + im^2)  (hpp:233-239)
template <class C> inline C dabs(const dcplx<C> &x) {
    C temp = x.real() * x.real() + x.imag() * x.imag();
    return sqrt(temp);
}
// sqrt(z) = (z + |z|) * sqrt|z| / |z + |z||   (hpp:242-260)
template <class C> inline dcplx<C> dsqrt(const dcplx<C> &x) {
    typedef typename dcplx<C>::T T;
    dcplx<C> result = x;
    C r = dabs(x);
    C sqrt_r = sqrt(r);
    result.real(result.real() + r);
    C zrnorm = dabs(result);
    if (std::fabs(zrnorm.real()) < static_cast<T>(1e-20) && std::fabs(zrnorm.imag()) < static_cast<T>(1e-20)) {
        result *= sqrt_r;
        return result;
    }
    C scale = sqrt_r / zrnorm;
    result *= scale;
    return result;
}

inline float qnan_f() {  // cx.h:158: __int_as_float(0x7fffffff)
    uint32_t b = 0x7fffffffu; float f; std::memcpy(&f, &b, 4); return f;
}

}  // namespace oc
