// DoubleComplex.h — host-side dual-complex number for second-order complex-step differentiation
// (DCSFD), with the public surface of the reference's class
// (DeviceArray/include/DoubleComplex.h:15-95, src/DoubleComplex.cpp:6-436): value = real().real(),
// gradient seed in real().imag(), second seed in imag().real(), the second derivative falls out
// of imag().imag() / h^2.  Built on std::complex<float> like the reference (SingleComplex).
//
// Kept as in the reference: comparisons look at real().real() only (.cpp:248-276);
// addPerturbation() uses h = 1e-6 (.cpp:61-66); sqrt is (z + |z|) * sqrt|z| / |z + |z|| (.cpp:325-342);
// log takes its imaginary part from the half-angle atan2 of .cpp:384-399; pow = polar(exp(y ln|z|), y arg z).
// Not provided: atanh / atan / atan2 on DoubleComplex — the reference's atanh evaluates
// log(a - a) (.cpp:373) and returns non-finite values.
// Eigen's NumTraits specialisation (EigenSupport.h) is not provided: Eigen is not a dependency.
#pragma once
#include <cmath>
#include <complex>
#include <ostream>

typedef float MyFloat;
typedef std::complex<MyFloat> SingleComplex;

class DoubleComplex {
    SingleComplex real_, imag_;

public:
    DoubleComplex() : real_(0), imag_(0) {}
    DoubleComplex(SingleComplex real, SingleComplex imag) : real_(real), imag_(imag) {}
    DoubleComplex(SingleComplex real) : real_(real), imag_(0) {}
    DoubleComplex(MyFloat real) : real_(real), imag_(0) {}
    DoubleComplex(MyFloat real_real, MyFloat real_imag, MyFloat imag_real, MyFloat imag_imag)
        : real_(real_real, real_imag), imag_(imag_real, imag_imag) {}

    SingleComplex real() const { return real_; }
    SingleComplex imag() const { return imag_; }
    void real(SingleComplex r) { real_ = r; }
    void imag(SingleComplex i) { imag_ = i; }
    void addPerturbation() { const float h = 1e-6; real_ = SingleComplex(real_.real(), h); imag_ = SingleComplex(h, 0); }
    void clearPerturbation() { real_ = SingleComplex(real_.real(), 0); imag_ = SingleComplex(0, 0); }

    DoubleComplex operator-() const { return DoubleComplex(-real_, -imag_); }
    DoubleComplex &operator=(const SingleComplex &o) { real_ = o; imag_ = 0; return *this; }
    DoubleComplex &operator=(const MyFloat &o) { real_ = o; imag_ = 0; return *this; }
    DoubleComplex &operator+=(const MyFloat &o) { real_ += o; return *this; }
    DoubleComplex &operator-=(const MyFloat &o) { real_ -= o; return *this; }
    DoubleComplex &operator*=(const MyFloat &o) { real_ *= o; imag_ *= o; return *this; }
    DoubleComplex &operator/=(const MyFloat &o) { real_ /= o; imag_ /= o; return *this; }
    DoubleComplex &operator+=(const SingleComplex &o) { real_ += o; return *this; }
    DoubleComplex &operator-=(const SingleComplex &o) { real_ -= o; return *this; }
    DoubleComplex &operator*=(const SingleComplex &o) { real_ *= o; imag_ *= o; return *this; }
    DoubleComplex &operator/=(const SingleComplex &o) { real_ /= o; imag_ /= o; return *this; }
    DoubleComplex &operator+=(const DoubleComplex &o) { real_ += o.real_; imag_ += o.imag_; return *this; }
    DoubleComplex &operator-=(const DoubleComplex &o) { real_ -= o.real_; imag_ -= o.imag_; return *this; }
    DoubleComplex &operator*=(const DoubleComplex &o) {
        const SingleComplex r = real_ * o.real_ - imag_ * o.imag_;
        const SingleComplex i = imag_ * o.real_ + real_ * o.imag_;
        real_ = r; imag_ = i;
        return *this;
    }
    DoubleComplex &operator/=(const DoubleComplex &o) {
        const SingleComplex r = real_ * o.real_ + imag_ * o.imag_;
        const SingleComplex n = o.real_ * o.real_ + o.imag_ * o.imag_;
        imag_ = (imag_ * o.real_ - real_ * o.imag_) / n;
        real_ = r / n;
        return *this;
    }
};

inline std::ostream &operator<<(std::ostream &os, const DoubleComplex &x) { return os << '(' << x.real() << ',' << x.imag() << ')'; }

inline bool operator>(const DoubleComplex &l, const DoubleComplex &r) { return l.real().real() > r.real().real(); }
inline bool operator>(const DoubleComplex &l, const SingleComplex &r) { return l.real().real() > r.real(); }
inline bool operator>(const DoubleComplex &l, const MyFloat &r) { return l.real().real() > r; }
inline bool operator<(const DoubleComplex &l, const DoubleComplex &r) { return l.real().real() < r.real().real(); }
inline bool operator<(const DoubleComplex &l, const SingleComplex &r) { return l.real().real() < r.real(); }
inline bool operator<(const DoubleComplex &l, const MyFloat &r) { return l.real().real() < r; }

inline DoubleComplex operator+(const DoubleComplex &l, const MyFloat &r) { DoubleComplex t(l); t += r; return t; }
inline DoubleComplex operator-(const DoubleComplex &l, const MyFloat &r) { DoubleComplex t(l); t -= r; return t; }
inline DoubleComplex operator*(const DoubleComplex &l, const MyFloat &r) { DoubleComplex t(l); t *= r; return t; }
inline DoubleComplex operator/(const DoubleComplex &l, const MyFloat &r) { DoubleComplex t(l); t /= r; return t; }
inline DoubleComplex operator+(const DoubleComplex &l, const DoubleComplex &r) { DoubleComplex t(l); t += r; return t; }
inline DoubleComplex operator-(const DoubleComplex &l, const DoubleComplex &r) { DoubleComplex t(l); t -= r; return t; }
inline DoubleComplex operator*(const DoubleComplex &l, const DoubleComplex &r) { DoubleComplex t(l); t *= r; return t; }
inline DoubleComplex operator/(const DoubleComplex &l, const DoubleComplex &r) { DoubleComplex t(l); t /= r; return t; }

inline SingleComplex real(const DoubleComplex &x) { return x.real(); }
inline SingleComplex imag(const DoubleComplex &x) { return x.imag(); }
inline MyFloat fabs(const DoubleComplex &x) { return std::fabs(x.real().real()); }
inline SingleComplex norm(const DoubleComplex &x) { return x.real() * x.real() + x.imag() * x.imag(); }
inline SingleComplex abs(const DoubleComplex &x) { return std::sqrt(norm(x)); }
inline DoubleComplex abs2(const DoubleComplex &x) { return x * x; }
inline DoubleComplex conj(const DoubleComplex &x) { return DoubleComplex(x.real(), -x.imag()); }
inline DoubleComplex polar(const SingleComplex &rho, const SingleComplex &theta) { return DoubleComplex(rho * std::cos(theta), rho * std::sin(theta)); }
inline DoubleComplex sqrt(const DoubleComplex &x) {
    DoubleComplex result = x;
    const SingleComplex r = abs(x), sqrt_r = std::sqrt(r);
    result += r;
    const SingleComplex zrnorm = abs(result);
    if (std::fabs(zrnorm.real()) < 1e-20f && std::fabs(zrnorm.imag()) < 1e-20f) { result *= sqrt_r; return result; }
    result *= sqrt_r / zrnorm;
    return result;
}
inline DoubleComplex abs_d(const DoubleComplex &x) { return sqrt(x * x); }
inline DoubleComplex exp(const DoubleComplex &x) { return DoubleComplex(std::exp(x.real()) * std::cos(x.imag()), std::exp(x.real()) * std::sin(x.imag())); }
// half-angle form the reference uses for arg (.cpp:384-399); "r > 0" compares the real part
inline SingleComplex atan2(const SingleComplex &y, const SingleComplex &x) {
    SingleComplex r = std::sqrt(x * x + y * y);
    if (r.real() > 0.0f) { r += x; r = y / r; }
    else { r -= x; r = r / y; }
    r = std::atan(r);
    r *= 2.0f;
    return r;
}
inline SingleComplex arg(const DoubleComplex &x) { return atan2(x.imag(), x.real()); }
inline DoubleComplex log(const DoubleComplex &x) { return DoubleComplex(std::log(abs(x)), atan2(x.imag(), x.real())); }
inline DoubleComplex sin(const DoubleComplex &x) { return DoubleComplex(std::cosh(-x.imag()) * std::sin(x.real()), -std::sinh(-x.imag()) * std::cos(x.real())); }
inline DoubleComplex cos(const DoubleComplex &x) { return DoubleComplex(std::cosh(-x.imag()) * std::cos(x.real()), std::sinh(-x.imag()) * std::sin(x.real())); }
inline DoubleComplex pow(const DoubleComplex &x, const MyFloat y) {
    const DoubleComplex r = log(x);
    return polar(std::exp(y * r.real()), y * r.imag());
}
