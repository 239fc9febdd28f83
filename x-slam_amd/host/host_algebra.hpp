// host_algebra.hpp — the small dense complex algebra the orchestrator performs between
// kernel launches.  The reference takes it from Eigen (KinectFusionReconstruction.cpp:167,
// 170, 182, 203, 211, 215-221, 231, 248-250, 305-307); Eigen is not a dependency here, so the
// fixed-size algorithms Eigen uses for these shapes are written out: cofactor inverses for
// 3x3 / 4x4, partial-pivot LU determinant, the unblocked lower Cholesky L L^H that reads only
// real(A(k,k)) — Eigen's complex LLT is Hermitian, so the imaginary part of the pose increment
// is Eigen's, not the analytic continuation — and AngleAxis' Rodrigues form.
// Scalars are std::complex, as in the reference (Internal.h:22-23).
#pragma once
#include <cmath>
#include <complex>

namespace xs_host {

typedef std::complex<float> hostComplex;       // Internal.h:22
typedef std::complex<double> hostComplexICP;   // Internal.h:23

template <int N>
struct MatC {  // row-major N x N complex float: the storage device_cast<> hands to the kernels
    hostComplex m[N][N];
    hostComplex &operator()(int r, int c) { return m[r][c]; }
    const hostComplex &operator()(int r, int c) const { return m[r][c]; }
    static MatC Identity() {
        MatC r;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) r.m[i][j] = hostComplex(i == j ? 1.f : 0.f, 0.f);
        return r;
    }
    const float *data() const { return reinterpret_cast<const float *>(&m[0][0]); }
};
typedef MatC<4> Matrix4cf;
typedef MatC<3> Matrix3cf;
struct Vector3cf {
    hostComplex v[3];
    hostComplex &operator[](int i) { return v[i]; }
    const hostComplex &operator[](int i) const { return v[i]; }
    const float *data() const { return reinterpret_cast<const float *>(&v[0]); }
};

template <int N>
inline MatC<N> operator*(const MatC<N> &a, const MatC<N> &b) {
    MatC<N> r;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            hostComplex s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < N; ++k) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline Vector3cf operator*(const Matrix3cf &a, const Vector3cf &x) {
    Vector3cf r;
    for (int i = 0; i < 3; ++i) r.v[i] = (a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1]) + a.m[i][2] * x.v[2];
    return r;
}

namespace detail {
inline hostComplex minor3(const Matrix4cf &a, int i1, int i2, int i3, int j1, int j2, int j3) {
    return a.m[i1][j1] * (a.m[i2][j2] * a.m[i3][j3] - a.m[i2][j3] * a.m[i3][j2]);
}
inline hostComplex cof4(const Matrix4cf &a, int i, int j) {
    const int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4, j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
    return minor3(a, i1, i2, i3, j1, j2, j3) + minor3(a, i2, i3, i1, j1, j2, j3) + minor3(a, i3, i1, i2, j1, j2, j3);
}
inline hostComplex cof3(const Matrix3cf &a, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
}  // namespace detail

inline Matrix4cf inverse(const Matrix4cf &a) {
    Matrix4cf r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const hostComplex c = detail::cof4(a, j, i);
            r.m[i][j] = ((i + j) & 1) ? -c : c;
        }
    hostComplex det = a.m[0][0] * r.m[0][0];
    for (int k = 1; k < 4; ++k) det = det + a.m[k][0] * r.m[0][k];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = r.m[i][j] / det;
    return r;
}
inline Matrix3cf inverse(const Matrix3cf &a) {
    const hostComplex c0 = detail::cof3(a, 0, 0), c1 = detail::cof3(a, 1, 0), c2 = detail::cof3(a, 2, 0);
    const hostComplex det = (c0 * a.m[0][0] + c1 * a.m[1][0]) + c2 * a.m[2][0];
    const hostComplex invdet = hostComplex(1.f, 0.f) / det;
    Matrix3cf r;
    r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
    for (int i = 1; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = detail::cof3(a, j, i) * invdet;
    return r;
}

// A.real().determinant(), A = 6x6 complex<double> stored (re, im) interleaved, [i*6+j]
inline double real_determinant6(const hostComplexICP *A) {
    double a[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i][j] = A[i * 6 + j].real();
    double det = 1.0;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = std::fabs(a[k][k]);
        for (int i = k + 1; i < 6; ++i) if (std::fabs(a[i][k]) > best) { best = std::fabs(a[i][k]); p = i; }
        if (best == 0.0) return 0.0;
        if (p != k) { for (int j = 0; j < 6; ++j) std::swap(a[k][j], a[p][j]); det = -det; }
        det *= a[k][k];
        for (int i = k + 1; i < 6; ++i) {
            const double f = a[i][k] / a[k][k];
            for (int j = k + 1; j < 6; ++j) a[i][j] -= f * a[k][j];
        }
    }
    return det;
}

// A.llt().solve(b) for Matrix<complex<double>,6,6>: lower, unblocked, Hermitian
inline void llt_solve6(const hostComplexICP *A, const hostComplexICP *b, hostComplexICP *x) {
    hostComplexICP L[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) L[i][j] = A[i * 6 + j];
    for (int k = 0; k < 6; ++k) {
        double d = L[k][k].real();
        for (int j = 0; j < k; ++j) d -= std::norm(L[k][j]);
        if (d <= 0.0) break;
        d = std::sqrt(d);
        L[k][k] = hostComplexICP(d, 0.0);
        for (int i = k + 1; i < 6; ++i) {
            hostComplexICP s = L[i][k];
            for (int j = 0; j < k; ++j) s -= L[i][j] * std::conj(L[k][j]);
            L[i][k] = s / d;
        }
    }
    hostComplexICP y[6];
    for (int i = 0; i < 6; ++i) y[i] = b[i];
    for (int i = 0; i < 6; ++i) {
        y[i] = y[i] / L[i][i];
        for (int r = i + 1; r < 6; ++r) y[r] -= y[i] * L[r][i];
    }
    for (int i = 5; i >= 0; --i) {
        hostComplexICP s = y[i];
        for (int r = i + 1; r < 6; ++r) s -= std::conj(L[r][i]) * y[r];
        y[i] = s / std::conj(L[i][i]);
    }
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// Eigen::AngleAxis<complex<float>>(angle, unit axis).toRotationMatrix()
inline Matrix3cf angle_axis(hostComplex angle, int axis) {
    hostComplex ax[3] = {hostComplex(0, 0), hostComplex(0, 0), hostComplex(0, 0)};
    ax[axis] = hostComplex(1.f, 0.f);
    const hostComplex s = std::sin(angle), c = std::cos(angle);
    const hostComplex sin_axis[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const hostComplex omc = hostComplex(1.f, 0.f) - c;
    const hostComplex cos1_axis[3] = {omc * ax[0], omc * ax[1], omc * ax[2]};
    Matrix3cf r;
    hostComplex tmp;
    tmp = cos1_axis[0] * ax[1]; r.m[0][1] = tmp - sin_axis[2]; r.m[1][0] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; r.m[0][2] = tmp + sin_axis[1]; r.m[2][0] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; r.m[1][2] = tmp - sin_axis[0]; r.m[2][1] = tmp + sin_axis[0];
    for (int i = 0; i < 3; ++i) r.m[i][i] = cos1_axis[i] * ax[i] + c;
    return r;
}

// se3Exp(const Eigen::VectorXcf& xi), KinectFusionReconstruction.h:176-219: xi = (v, omega); Rodrigues
// with complex scalars; below |omega| = 1e-6 the first-order form R = I + omega^, V = I + omega^
// (|omega| is the Euclidean norm of the complex vector, as Eigen's norm()).
inline Matrix4cf se3Exp(const hostComplex xi[6]) {
    const hostComplex *v = xi, *omega = xi + 3;
    Matrix3cf omegaHat;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) omegaHat.m[i][j] = hostComplex(0.f, 0.f);
    omegaHat.m[0][1] = -omega[2]; omegaHat.m[0][2] = omega[1]; omegaHat.m[1][2] = -omega[0];
    omegaHat.m[1][0] = omega[2]; omegaHat.m[2][0] = -omega[1]; omegaHat.m[2][1] = omega[0];
    Matrix3cf R = Matrix3cf::Identity(), V = Matrix3cf::Identity();
    const float nrm = std::sqrt(std::norm(omega[0]) + std::norm(omega[1]) + std::norm(omega[2]));
    if (nrm < 1e-6f) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { R.m[i][j] += omegaHat.m[i][j]; V.m[i][j] += omegaHat.m[i][j]; }
    } else {
        const hostComplex sum = (omega[0] * omega[0] + omega[1] * omega[1]) + omega[2] * omega[2];
        const hostComplex theta = std::sqrt(sum);
        const hostComplex s = std::sin(theta), c = std::cos(theta);
        const Matrix3cf sq = omegaHat * omegaHat;
        const hostComplex A = s / theta;
        const hostComplex B = (1.0f - c) / std::pow(theta, 2.0f);
        const hostComplex C = (theta - s) / std::pow(theta, 3.0f);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                R.m[i][j] = (R.m[i][j] + A * omegaHat.m[i][j]) + B * sq.m[i][j];
                V.m[i][j] = (V.m[i][j] + B * omegaHat.m[i][j]) + C * sq.m[i][j];
            }
    }
    Vector3cf vv; vv.v[0] = v[0]; vv.v[1] = v[1]; vv.v[2] = v[2];
    const Vector3cf t = V * vv;
    Matrix4cf out;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[i][j] = hostComplex(0.f, 0.f);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) out.m[i][j] = R.m[i][j]; out.m[i][3] = t.v[i]; }
    out.m[3][3] = hostComplex(1.f, 0.f);
    return out;
}

// x = A^-1 b for a symmetric positive definite real 6x6 (row-major), plain Cholesky; false if a pivot fails
inline bool solve_spd6(const double *A, const double *b, double *x) {
    double L[6][6] = {};
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / L[j][j];
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
    return true;
}

inline Matrix3cf GetRotation(const Matrix4cf &t) { Matrix3cf r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = t.m[i][j]; return r; }
inline Vector3cf GetTranslation(const Matrix4cf &t) { Vector3cf v; for (int i = 0; i < 3; ++i) v.v[i] = t.m[i][3]; return v; }

}  // namespace xs_host
