// ORACLE — TEST INFRASTRUCTURE ONLY (see oc_complex.hpp for the rule).
// CPU restatement of the reference's CPU DeviceArray path:
//   * host DoubleComplex over std::complex<float>
//       DeviceArray/include/DoubleComplex.h:15-95, src/DoubleComplex.cpp:6-436
//   * the test_CSFD demo's scalar kernels and its chain-rule check
//       Experiments/test_CSFD/main.cpp:8-86 (f1, *_raw / *_our), :194-219
// DoubleComplex.cpp cannot be compiled here (it includes <Eigen/Dense>, absent
// from this image), so this restatement is pinned by the demo's printed known
// answers only (tests/golden/test_csfd_known_answers.json).
// Not restated: atanh/atan/atan2 on DoubleComplex — the reference's atanh
// evaluates log(a - a) (DoubleComplex.cpp:373) and returns non-finite values.
#pragma once
#include <cmath>
#include <complex>

namespace oc {

typedef std::complex<float> SC;

struct HDC {  // host DoubleComplex
    SC real_, imag_;
    HDC() : real_(0), imag_(0) {}
    HDC(SC r, SC i) : real_(r), imag_(i) {}
    HDC(SC r) : real_(r), imag_(0) {}
    HDC(float r) : real_(r), imag_(0) {}
    HDC(float rr, float ri, float ir, float ii) : real_(rr, ri), imag_(ir, ii) {}
    SC real() const { return real_; }
    SC imag() const { return imag_; }
    void addPerturbation() { float h = 1e-6; real_ = SC(real_.real(), h); imag_ = SC(h, 0); }  // .cpp:61-66
    HDC operator-() const { return HDC(-real_, -imag_); }
    HDC &operator+=(const float &o) { real_ += o; return *this; }
    HDC &operator-=(const float &o) { real_ -= o; return *this; }
    HDC &operator*=(const float &o) { real_ *= o; imag_ *= o; return *this; }
    HDC &operator/=(const float &o) { real_ /= o; imag_ /= o; return *this; }
    HDC &operator+=(const SC &o) { real_ += o; return *this; }
    HDC &operator*=(const SC &o) { real_ *= o; imag_ *= o; return *this; }
    HDC &operator/=(const SC &o) { real_ /= o; imag_ /= o; return *this; }
    HDC &operator+=(const HDC &o) { real_ += o.real_; imag_ += o.imag_; return *this; }
    HDC &operator-=(const HDC &o) { real_ -= o.real_; imag_ -= o.imag_; return *this; }
    HDC &operator*=(const HDC &o) {  // .cpp:155-162
        SC r = real_ * o.real_ - imag_ * o.imag_;
        SC i = imag_ * o.real_ + real_ * o.imag_;
        real_ = r; imag_ = i;
        return *this;
    }
    HDC &operator/=(const HDC &o);
};
inline SC hnorm(const HDC &x) { return x.real() * x.real() + x.imag() * x.imag(); }  // .cpp:305-308
inline HDC &HDC::operator/=(const HDC &o) {  // .cpp:164-171
    const SC r = real_ * o.real_ + imag_ * o.imag_;
    const SC n = hnorm(o);
    imag_ = (imag_ * o.real() - real_ * o.imag()) / n;
    real_ = r / n;
    return *this;
}
inline HDC operator+(const HDC &l, const HDC &r) { HDC t(l); t += r; return t; }
inline HDC operator-(const HDC &l, const HDC &r) { HDC t(l); t -= r; return t; }
inline HDC operator*(const HDC &l, const HDC &r) { HDC t(l); t *= r; return t; }
inline HDC operator/(const HDC &l, const HDC &r) { HDC t(l); t /= r; return t; }
inline HDC operator*(const HDC &l, const float &r) { HDC t(l); t *= r; return t; }
inline SC habs(const HDC &x) { SC t = x.real() * x.real() + x.imag() * x.imag(); return std::sqrt(t); }  // .cpp:287-291
inline HDC hpolar(const SC &rho, const SC &theta) { return HDC(rho * std::cos(theta), rho * std::sin(theta)); }
inline HDC hsqrt(const HDC &x) {  // .cpp:325-342
    HDC result = x;
    SC r = habs(x);
    SC sqrt_r = std::sqrt(r);
    result += r;
    SC zrnorm = habs(result);
    if (std::fabs(zrnorm.real()) < 1e-20f && std::fabs(zrnorm.imag()) < 1e-20f) { result *= sqrt_r; return result; }
    SC scale = sqrt_r / zrnorm;
    result *= scale;
    return result;
}
inline HDC hexp(const HDC &x) { return HDC(std::exp(x.real()) * std::cos(x.imag()), std::exp(x.real()) * std::sin(x.imag())); }
// SingleComplex atan2(y, x) of DoubleComplex.cpp:384-399 (used by log)
inline SC hatan2(const SC &y, const SC &x) {
    SC r = x * x + y * y;
    r = std::sqrt(r);
    // "r > 0" on std::complex does not exist; the reference file resolves it
    // through operator>(DoubleComplex, float) via implicit conversion, which
    // compares the real part (.cpp:258-261).
    if (r.real() > 0.0f) { r += x; r = y / r; }
    else { r -= x; r = r / y; }
    r = std::atan(r);
    r *= 2.0f;
    return r;
}
inline HDC hlog(const HDC &x) {  // .cpp:352-360
    SC r = habs(x);
    SC imag = hatan2(x.imag(), x.real());
    SC real = std::log(r);
    return HDC(real, imag);
}
inline HDC hsin(const HDC &x) {  // .cpp:418-423
    return HDC(std::cosh(-x.imag()) * std::sin(x.real()), -std::sinh(-x.imag()) * std::cos(x.real()));
}
inline HDC hcos(const HDC &x) {  // .cpp:425-430
    return HDC(std::cosh(-x.imag()) * std::cos(x.real()), std::sinh(-x.imag()) * std::sin(x.real()));
}
inline HDC hpow(const HDC &x, const float y) {  // .cpp:432-437
    HDC r = hlog(x);
    return hpolar(std::exp(y * r.real()), y * r.imag());
}

// test_CSFD/main.cpp:8-11
inline HDC f1(HDC x, HDC y) { return (x + y) * (x + y); }

// test_CSFD/main.cpp:18-86 — the "raw" (full) and "our" (O(h^2) dropped) forms
inline SC multiplication_our(const SC &a, const SC &b) { return SC(a.real() * b.real(), a.imag() * b.real() + a.real() * b.imag()); }
inline SC multiplication_raw(const SC &a, const SC &b) { return SC(a.real() * b.real() - a.imag() * b.imag(), a.imag() * b.real() + a.real() * b.imag()); }
inline SC division_our(const SC &a, const SC &b) {
    return SC(a.real() / b.real(), (a.imag() * b.real() - a.real() * b.imag()) / (b.real() * b.real() + b.imag() * b.imag()));
}
inline SC division_raw(const SC &a, const SC &b) {
    return SC((a.real() * b.real() + a.imag() * b.imag()) / (b.real() * b.real() + b.imag() * b.imag()),
              (a.imag() * b.real() - a.real() * b.imag()) / (b.real() * b.real() + b.imag() * b.imag()));
}
inline SC exp_our(const SC &a) { return SC(std::exp(a.real()), std::exp(a.real()) * std::sin(a.imag())); }
inline SC exp_raw(const SC &a) { return SC(std::exp(a.real()) * std::cos(a.imag()), std::exp(a.real()) * std::sin(a.imag())); }
inline SC sin_our(const SC &a) { return SC(std::sin(a.real()), -std::sinh(-a.imag()) * std::cos(a.real())); }
inline SC sin_raw(const SC &a) { return SC(std::sin(a.real()) * std::cosh(-a.imag()), -std::sinh(-a.imag()) * std::cos(a.real())); }
// pow(float, int), norm(), arg() resolve to std:: overloads in the reference;
// std::pow(float,int) computes in double and the result is narrowed to float.
inline SC pow_our(const SC &a, const int &n) {
    float re = std::pow(a.real(), n);
    float im = std::pow(std::norm(a), n) * std::sin(n * std::arg(a));
    return SC(re, im);
}
inline SC pow_raw(const SC &a, const int &n) {
    float re = std::pow(std::norm(a), n) * std::cos(n * std::arg(a));
    float im = std::pow(std::norm(a), n) * std::sin(n * std::arg(a));
    return SC(re, im);
}

}  // namespace oc
