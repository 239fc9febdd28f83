// xs_icp_solve.h — the pose update the reference performs on the host between two ICP launches
// (KinectFusionReconstruction.cpp:203-224), as device code for a one-workgroup kernel that follows the
// reduction on the stream: det(A.real()) gate, complex<double> Cholesky solve, cast to complex<float>,
// AngleAxis about Z, Y, X, pose composition.  Same operations in the same order as
// x-slam_amd/host/host_algebra.hpp (which the parity suite pins against the oracle): the
// double-precision part uses only + - * / sqrt, which are correctly rounded on both sides, so it
// carries the same bits; the float sin/cos/sinh/cosh of the three angles are evaluated in double
// and rounded, which agrees with the host libm except in rare last-bit cases.
#pragma once
#include "xs_device.h"

namespace xs {

// Device-resident pose of the ICP loop (128 bytes).  The same layout is mirrored to a host-visible
// copy when the caller gives one.
struct IcpPoseState {
    float R[18];   // Rcurr, 3x3 complex row-major
    float t[6];    // tcurr
    int status;    // 0 = ok, 1 = |det| < 1e-15, 2 = det is NaN; sticky until a launch reloads the pose
    int iters;     // iterations applied since the pose was loaded
    double det;    // determinant of the last iteration
    double pad[2];
};
static_assert(sizeof(IcpPoseState) == 128, "IcpPoseState layout");

namespace icp_solve {

// ICP.cu:419-428: 27 (re, im) sums -> the entries of the symmetric A and b
__device__ __forceinline__ int tri_index(int i, int j) {  // position of (i, j), j >= i, j == 6 means b[i]
    return i * 7 - (i * (i - 1)) / 2 + (j - i);
}
__device__ __forceinline__ cdouble a_entry(const double *sums, int i, int j) {
    const int s = (j >= i) ? tri_index(i, j) : tri_index(j, i);
    return cdouble(sums[2 * s], sums[2 * s + 1]);
}

__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ int bcast(int v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ cdouble bcast(cdouble v, int lane) { return cdouble(__shfl(v.re, lane, 64), __shfl(v.im, lane, 64)); }

// A.real().determinant(): partial-pivot LU (host_algebra.hpp real_determinant6), one wave, lane i
// holding row i.  Rows are not moved: lane_at[q] (wave-uniform) names the lane whose row stands at
// position q of the host's array, the pivot search scans positions upwards with the host's strict
// comparison, the pivot row is broadcast, and the rows at later positions eliminate side by side —
// one division per step instead of five in a row.  Every entry sees the host's operations in the
// host's order.
__device__ inline double real_determinant6_wave(const double *sums, int lane) {
    const int row = lane < 6 ? lane : 5;
    double a[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) a[j] = a_entry(sums, row, j).re;
    int lane_at[6] = {0, 1, 2, 3, 4, 5};
    double det = 1.0;
    bool zero = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double v = fabs(a[k]);
        int p = k;
        double best = bcast(v, lane_at[k]);
#pragma unroll
        for (int q = k + 1; q < 6; ++q) {
            const double vq = bcast(v, lane_at[q]);
            if (vq > best) { best = vq; p = q; }
        }
        if (best == 0.0) zero = true;
        if (p != k) {
#pragma unroll
            for (int q = k + 1; q < 6; ++q)
                if (q == p) { const int tmp = lane_at[k]; lane_at[k] = lane_at[q]; lane_at[q] = tmp; }
            det = -det;
        }
        double rk[6];
#pragma unroll
        for (int j = k; j < 6; ++j) rk[j] = bcast(a[j], lane_at[k]);
        det *= rk[k];
        bool later = false;
#pragma unroll
        for (int q = k + 1; q < 6; ++q) later = later || (lane_at[q] == lane);
        const double f = a[k] / rk[k];
        if (later) {
#pragma unroll
            for (int j = k + 1; j < 6; ++j) a[j] -= f * rk[j];
        }
    }
    return zero ? 0.0 : det;
}

// A.llt().solve(b): lower, unblocked, Hermitian (host_algebra.hpp llt_solve6), one wave, lane i
// owning row i of L and y(i); every entry goes through the host's operations in the host's order.
// The forward substitution rides along with the factorisation — step k of it needs only column k
// of L — so its six divisions overlap the column's instead of following them.  Called by a whole
// wave; lane i returns x(i).  A failed pivot stops the factorisation where it is, as on the host;
// the substitutions still run (on the unfactored entries, as there).
__device__ inline cdouble llt_solve6_wave(const double *sums, int lane) {
    const int row = lane < 6 ? lane : 5;  // idle lanes shadow row 5
    cdouble Lr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) Lr[j] = a_entry(sums, row, j);
    cdouble y = a_entry(sums, row, 6);
    // libgcc's __divdc3 (Smith's form); a divisor with zero imaginary part divides component-wise
    auto cdiv = [](cdouble z, cdouble w) -> cdouble {
        if (w.im == 0.0) return cdouble(z.re / w.re, z.im / w.re);
        if (fabs(w.re) < fabs(w.im)) {
            const double r = w.re / w.im, den = w.re * r + w.im;
            return cdouble((z.re * r + z.im) / den, (z.im * r - z.re) / den);
        }
        const double r = w.im / w.re, den = w.im * r + w.re;
        return cdouble((z.im * r + z.re) / den, (z.im - z.re * r) / den);
    };
    int kfail = 6;  // first step whose pivot failed
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double d = Lr[k].re;
#pragma unroll
        for (int j = 0; j < k; ++j) d -= Lr[j].re * Lr[j].re + Lr[j].im * Lr[j].im;
        double dk = bcast(d, k);  // row k's value
        if (kfail == 6) {
            if (dk <= 0.0) kfail = k;
            else {
                dk = ::sqrt(dk);
                cdouble s = Lr[k];
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    const cdouble lkj = bcast(Lr[j], k);
                    s -= Lr[j] * cdouble(lkj.re, -lkj.im);
                }
                const cdouble q(s.re / dk, s.im / dk);
                if (row > k) Lr[k] = q;
                else if (row == k) Lr[k] = cdouble(dk, 0.0);
                // forward step k: y(k) /= L(k,k) (real), then y(r) -= y(k) L(r,k) below it
                const cdouble yk = bcast(cdouble(y.re / dk, y.im / dk), k);
                if (row == k) y = yk;
                else if (row > k) y -= yk * q;
            }
        }
    }
    if (kfail < 6) {  // the forward steps the factorisation did not reach
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i >= kfail) {
                const cdouble yi = bcast(cdiv(y, Lr[i]), i);
                if (row == i) y = yi;
                else if (row > i) y -= yi * Lr[i];
            }
        }
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        cdouble s = bcast(y, i);
#pragma unroll
        for (int r = i + 1; r < 6; ++r) {
            const cdouble lri = bcast(Lr[i], r);
            s -= cdouble(lri.re, -lri.im) * bcast(y, r);
        }
        const cdouble lii = bcast(Lr[i], i);
        const cdouble yi = cdiv(s, cdouble(lii.re, -lii.im));
        if (row == i) y = yi;
    }
    return y;
}

// sin, cos, sinh, cosh in double for the float rounding below.  ICP increments are small angles with
// ~1e-7 derivative parts: inside |x| <= 1/8 a short Taylor polynomial (first dropped term < 1e-13
// relative, far below the float rounding it feeds) replaces the general library routines, which stay
// for the rest.  (These are approximations of transcendental functions, not the reference's
// arithmetic: fused multiply-adds are fine here.)
__device__ __forceinline__ void sincos_d(double x, double &s, double &c) {
    if (fabs(x) <= 0.125) {
        const double x2 = x * x;
        const double ps = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, 1.0 / 362880, -1.0 / 5040), 1.0 / 120), -1.0 / 6), 1.0);
        s = x * ps;
        c = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, -1.0 / 3628800, 1.0 / 40320), -1.0 / 720), 1.0 / 24), -0.5), 1.0);
    } else {
        s = ::sin(x); c = ::cos(x);
    }
}
__device__ __forceinline__ void sinhcosh_d(double x, double &s, double &c) {
    if (fabs(x) <= 0.125) {
        const double x2 = x * x;
        const double ps = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, 1.0 / 362880, 1.0 / 5040), 1.0 / 120), 1.0 / 6), 1.0);
        s = x * ps;
        c = __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, __builtin_fma(x2, 1.0 / 3628800, 1.0 / 40320), 1.0 / 720), 1.0 / 24), 0.5), 1.0);
    } else {
        s = ::sinh(x); c = ::cosh(x);
    }
}
// std::sin / std::cos of a complex<float> the way glibc's csinf / ccosf build them: real functions of
// the two parts, each rounded to float, then one float product
__device__ inline void csincos(cfloat z, cfloat &s, cfloat &c) {
    double sd, cd, shd, chd;
    sincos_d((double)z.re, sd, cd);
    sinhcosh_d((double)z.im, shd, chd);
    const float sx = (float)sd, cx = (float)cd, chy = (float)chd, shy = (float)shd;
    s = cfloat(chy * sx, shy * cx);
    c = cfloat(chy * cx, -(shy * sx));
}

struct M3 { cfloat m[3][3]; };
__device__ __forceinline__ M3 mul(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cfloat s = a.m[i][0] * b.m[0][j];
#pragma unroll
            for (int k = 1; k < 3; ++k) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
// Eigen::AngleAxis<complex<float>>(angle, unit axis).toRotationMatrix() (host_algebra.hpp angle_axis)
template <int AXIS>
__device__ __forceinline__ M3 angle_axis(cfloat s, cfloat c) {
    cfloat ax[3] = {cfloat(0.f, 0.f), cfloat(0.f, 0.f), cfloat(0.f, 0.f)};
    ax[AXIS] = cfloat(1.f, 0.f);
    const cfloat sin_axis[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const cfloat omc = cfloat(1.f, 0.f) - c;
    const cfloat cos1_axis[3] = {omc * ax[0], omc * ax[1], omc * ax[2]};
    M3 r;
    cfloat tmp;
    tmp = cos1_axis[0] * ax[1]; r.m[0][1] = tmp - sin_axis[2]; r.m[1][0] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; r.m[0][2] = tmp + sin_axis[1]; r.m[2][0] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; r.m[1][2] = tmp - sin_axis[0]; r.m[2][1] = tmp + sin_axis[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) r.m[i][i] = cos1_axis[i] * ax[i] + c;
    return r;
}

// KinectFusionReconstruction.cpp:211-221: Rinc = (Rz(gamma) Ry(beta)) Rx(alpha); tcurr = Rinc tcurr + tinc;
// Rcurr = Rinc Rcurr.  Three lanes, lane i forming row i of every product (each entry with the host's
// operation order: a(i,0) b(0,j), + a(i,1) b(1,j), + a(i,2) b(2,j)); the right-hand factors are
// wave-uniform.  sn / cs: sin and cos of result[0..2].  Lane i < 3 returns row i of the new Rcurr and
// tcurr(i).
struct Row3 { cfloat v[3]; };
__device__ __forceinline__ Row3 pick_row(const M3 &m, int i) {
    Row3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r.v[j].re = i == 0 ? m.m[0][j].re : (i == 1 ? m.m[1][j].re : m.m[2][j].re);
        r.v[j].im = i == 0 ? m.m[0][j].im : (i == 1 ? m.m[1][j].im : m.m[2][j].im);
    }
    return r;
}
__device__ __forceinline__ Row3 row_times(const Row3 &a, const M3 &b) {
    Row3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        cfloat s = a.v[0] * b.m[0][j];
        s = s + a.v[1] * b.m[1][j];
        s = s + a.v[2] * b.m[2][j];
        r.v[j] = s;
    }
    return r;
}
__device__ inline void compose_pose_rows(int i, const cfloat *result, const cfloat *sn, const cfloat *cs, const MatS33 &Rcurr, const cfloat3 &tcurr,
                                         Row3 &Rn, cfloat &tn) {
    const M3 Rz = angle_axis<2>(sn[2], cs[2]), Ry = angle_axis<1>(sn[1], cs[1]), Rx = angle_axis<0>(sn[0], cs[0]);
    const Row3 rinc = row_times(row_times(pick_row(Rz, i), Ry), Rx);
    M3 Rc;
#pragma unroll
    for (int r = 0; r < 3; ++r) { Rc.m[r][0] = Rcurr.data[r].x; Rc.m[r][1] = Rcurr.data[r].y; Rc.m[r][2] = Rcurr.data[r].z; }
    tn = ((rinc.v[0] * tcurr.x + rinc.v[1] * tcurr.y) + rinc.v[2] * tcurr.z) + result[3 + i];
    Rn = row_times(rinc, Rc);
}

}  // namespace icp_solve
}  // namespace xs
