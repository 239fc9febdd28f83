// ORACLE — TEST INFRASTRUCTURE ONLY (see oc_complex.hpp for the rule).
// C entry points over the templated restatement, for ctypes (tests/, smoke(),
// bench.py's cpu_baseline).  Built twice by oracle/Makefile:
//   liboracle.so            C = oc::cplx<float>            prefix orc_
//   _ref/liboracle_ref.so   C = the reference's ::complex<float>, included
//                           from /root/reference/DeviceArray/include/
//                           cuda_complex.hpp where it lies   prefix orcref_
// The _ref build is the "real reference arithmetic under restated kernel
// control flow" checker; it exists only in the build container.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#ifdef ORC_REF
#include "cuda_complex.hpp"  // the reference's header, -I/root/reference/DeviceArray/include
typedef ::complex<float> CF;
typedef ::complex<double> CD;
#define ORC(name) orcref_##name
#define CPOLAR ::polar<float>
#else
#include "oc_complex.hpp"
typedef oc::cplx<float> CF;
typedef oc::cplx<double> CD;
#define ORC(name) orc_##name
#define CPOLAR oc::polar<float>
#endif
#include "oc_csfd.hpp"
#include "oc_host.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oc;
typedef dcplx<CF> DCF;

extern "C" {

int ORC(num_threads)() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ORC(set_num_threads)(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// ---- scalar op tables ------------------------------------------------------
// a, b, out: n interleaved (re, im) pairs.  Returns 0, or -1 for a bad op.
int ORC(cop)(int op, long n, const float *a, const float *b, float *out) {
    for (long i = 0; i < n; ++i) {
        CF x(a[2 * i], a[2 * i + 1]), y(b[2 * i], b[2 * i + 1]), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = sqrt(x); break;
            case 5: r = CF(abs(x), 0.f); break;
            case 6: r = exp(x); break;
            case 7: r = log(x); break;
            case 8: r = pow(x, y); break;
            case 9: r = sin(x); break;
            case 10: r = cos(x); break;
            case 11: r = sinh(x); break;
            case 12: r = cosh(x); break;
            case 13: r = sin_new(x); break;
            case 14: r = sinh_new(x); break;
            case 15: r = CF(norm(x), 0.f); break;
            case 16: r = CF(arg(x), 0.f); break;
            case 17: r = conj(x); break;
            case 18: r = CPOLAR(x.real(), y.real()); break;
            case 19: r = x / y.real(); break;        // complex / scalar
            case 20: r = y.real() / x; break;        // scalar / complex
            case 21: r = x * y.real(); break;        // complex * scalar
            case 22: r = y.real() - x; break;        // scalar - complex
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
            default: return -1;
        }
        out[2 * i] = r.real(); out[2 * i + 1] = r.imag();
    }
    return 0;
}
int ORC(cop_f64)(int op, long n, const double *a, const double *b, double *out) {
    for (long i = 0; i < n; ++i) {
        CD x(a[2 * i], a[2 * i + 1]), y(b[2 * i], b[2 * i + 1]), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = sqrt(x); break;
            default: return -1;
        }
        out[2 * i] = r.real(); out[2 * i + 1] = r.imag();
    }
    return 0;
}
// dual complex: a, b, out are n groups of (re.re, re.im, im.re, im.im)
int ORC(dop)(int op, long n, const float *a, const float *b, float *out) {
    for (long i = 0; i < n; ++i) {
        DCF x(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]), y(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = dsqrt(x); break;
            case 5: { CF t = dabs(x); r = DCF(t); break; }
            case 6: r = x * y.value(); break;
            case 7: r = x / y.value(); break;
            case 8: r = x + y.value(); break;
            case 9: r = y.value() - x; break;
            default: return -1;
        }
        out[4 * i] = r.real().real(); out[4 * i + 1] = r.real().imag();
        out[4 * i + 2] = r.imag().real(); out[4 * i + 3] = r.imag().imag();
    }
    return 0;
}

#ifndef ORC_REF
// host DoubleComplex (std::complex based), same 4-float groups
int ORC(hdop)(int op, long n, const float *a, const float *b, float *out) {
    for (long i = 0; i < n; ++i) {
        HDC x(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]), y(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = hsqrt(x); break;
            case 5: r = HDC(habs(x)); break;
            case 6: r = hexp(x); break;
            case 7: r = hlog(x); break;
            case 8: r = hsin(x); break;
            case 9: r = hcos(x); break;
            case 10: r = hpow(x, y.real().real()); break;
            case 11: r = f1(x, y); break;
            default: return -1;
        }
        out[4 * i] = r.real().real(); out[4 * i + 1] = r.real().imag();
        out[4 * i + 2] = r.imag().real(); out[4 * i + 3] = r.imag().imag();
    }
    return 0;
}
// test_CSFD scalar kernels over arrays: which 0 mul 1 div 2 exp 3 sin 4 pow(n=3);
// variant 0 = raw, 1 = our.  exp/sin/pow take a[i] + b[i] like the demo.
int ORC(csfd_op)(int which, int variant, long n, const float *a, const float *b, float *out) {
#pragma omp parallel for
    for (long i = 0; i < n; ++i) {
        SC x(a[2 * i], a[2 * i + 1]), y(b[2 * i], b[2 * i + 1]), r;
        switch (which) {
            case 0: r = variant ? multiplication_our(x, y) : multiplication_raw(x, y); break;
            case 1: r = variant ? division_our(x, y) : division_raw(x, y); break;
            case 2: r = variant ? exp_our(x + y) : exp_raw(x + y); break;
            case 3: r = variant ? sin_our(x + y) : sin_raw(x + y); break;
            case 4: r = variant ? pow_our(x + y, 3) : pow_raw(x + y, 3); break;
            default: r = SC(0, 0);
        }
        out[2 * i] = r.real(); out[2 * i + 1] = r.imag();
    }
    return (which >= 0 && which <= 4) ? 0 : -1;
}
// the demo's part 2 (main.cpp:194-219): out = {dcsfd grad, dcsfd second,
// chain-rule grad, chain-rule second}
void ORC(csfd_chain_rule)(float t0, float h, float out[4]) {
    HDC t(SC(t0, h), SC(h, 0));
    HDC x = t * t;
    HDC y = hsin(t);
    float part_x_part_t = x.real().imag() / h;
    float part_xx_part_tt = x.imag().imag() / h / h;
    float part_y_part_t = y.real().imag() / h;
    float part_yy_part_tt = y.imag().imag() / h / h;
    HDC loss = f1(x, y);
    out[0] = loss.real().imag() / h;
    out[1] = loss.imag().imag() / h / h;
    float x_ = x.real().real(), y_ = y.real().real();
    float part_f_part_x = f1(HDC(x_, h, h, 0), HDC(y_, 0, 0, 0)).real().imag() / h;
    float part_f_part_y = f1(HDC(x_, 0, 0, 0), HDC(y_, h, h, 0)).real().imag() / h;
    float part_ff_part_xx = f1(HDC(x_, h, h, 0), HDC(y_, 0, 0, 0)).imag().imag() / h / h;
    float part_ff_part_yy = f1(HDC(x_, 0, 0, 0), HDC(y_, h, h, 0)).imag().imag() / h / h;
    float part_ff_part_xy = f1(HDC(x_, h, 0, 0), HDC(y_, 0, h, 0)).imag().imag() / h / h;
    out[2] = part_f_part_x * part_x_part_t + part_f_part_y * part_y_part_t;
    out[3] = part_f_part_x * part_xx_part_tt + part_f_part_y * part_yy_part_tt +
             part_x_part_t * part_x_part_t * part_ff_part_xx + part_y_part_t * part_y_part_t * part_ff_part_yy +
             part_x_part_t * part_y_part_t * (part_ff_part_xy + part_ff_part_xy);
}
#endif

// ---- kernels ---------------------------------------------------------------
void ORC(init_volume)(float *value, int *weight, float *grad, size_t step, const int *res) { init_volume(value, weight, grad, step, res); }

void ORC(scale_depth)(const uint16_t *depth, size_t dstep, int rows, int cols, float *scaled, size_t sstep) {
    scale_depth(depth, dstep, rows, cols, scaled, sstep);
}

long long ORC(integrate)(const float *depthScaled, size_t dstep, int drows, int dcols, float *value, int *weight, float *grad,
                         size_t vstep, const int *res, float tranc_dist, int max_weight, const float *Rv2c18, const float *tv2c6,
                         const float *intr4, float voxel_size, float threshold, int z0, int z1) {
    Intr k{intr4[0], intr4[1], intr4[2], intr4[3]};
    return integrate<CF>(depthScaled, dstep, drows, dcols, value, weight, grad, vstep, res, tranc_dist, max_weight,
                         load_mat33<CF>(Rv2c18), load_vec3<CF>(tv2c6), k, voxel_size, threshold, z0, z1);
}

long long ORC(raycast)(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                       float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vstep,
                       float *vmap, float *nmap, size_t mstep, int rows, int cols) {
    Intr k{intr4[0], intr4[1], intr4[2], intr4[3]};
    return raycast<CF>(k, load_mat33<CF>(Rc2v18), load_vec3<CF>(tc2v6), load_mat33<CF>(Rv2w18), load_vec3<CF>(tv2w6), tranc_dist, res,
                       voxel_size, value, grad, vstep, vmap, nmap, mstep, rows, cols);
}

long long ORC(raycast_slab)(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                            float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vstep, int zs0,
                            int zs1, int z0, int z1, float *vmap, float *nmap, size_t mstep, int rows, int cols, int *keys) {
    Intr k{intr4[0], intr4[1], intr4[2], intr4[3]};
    return raycast_slab<CF>(k, load_mat33<CF>(Rc2v18), load_vec3<CF>(tc2v6), load_mat33<CF>(Rv2w18), load_vec3<CF>(tv2w6), tranc_dist, res,
                            voxel_size, value, grad, vstep, zs0, zs1, z0, z1, vmap, nmap, mstep, rows, cols, keys);
}

void ORC(bilateral)(const uint16_t *src, size_t sstep, int rows, int cols, float *dst, size_t dstep) { bilateral<CF>(src, sstep, rows, cols, dst, dstep); }
void ORC(pyr_down)(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep) { pyr_down<CF>(src, sstep, srows, scols, dst, dstep); }
void ORC(create_vmap)(const float *intr4, const float *depth, size_t dstep, int rows, int cols, float *vmap, size_t mstep) {
    create_vmap<CF>(Intr{intr4[0], intr4[1], intr4[2], intr4[3]}, depth, dstep, rows, cols, vmap, mstep);
}
void ORC(create_nmap)(int rows, int cols, const float *vmap, float *nmap, size_t mstep) { create_nmap<CF>(rows, cols, vmap, nmap, mstep); }
void ORC(resize_map)(int normalize, int srows, int scols, const float *in, size_t istep, float *out, size_t ostep) {
    resize_map<CF>(normalize != 0, srows, scols, in, istep, out, ostep);
}

void ORC(tsdf_gn_terms)(const float *depthScaled, size_t dstep, int drows, int dcols, const int *res, float voxel_size, const float *Rv2c108,
                        const float *tv2c36, float tranc_dist, const float *intr4, const float *gt, int z0, int z1, double *out29) {
    tsdf_gn_terms<CF>(depthScaled, dstep, drows, dcols, res, voxel_size, Rv2c108, tv2c36, tranc_dist, Intr{intr4[0], intr4[1], intr4[2], intr4[3]},
                      gt, z0, z1, out29);
}
long long ORC(extract_points)(const float *value, size_t vstep, const int *res, float voxel_size, int zs0, int z0, int z1, float *out,
                              long long capacity) {
    return (long long)extract_points(value, vstep, res[0], res[1], res[2], voxel_size, zs0, z0, z1, out, (size_t)capacity);
}
void ORC(extract_normals)(const float *value, size_t vstep, const int *res, float voxel_size, const float *points, long long n, float *normals) {
    extract_normals(value, vstep, res[0], res[1], res[2], voxel_size, points, (size_t)n, normals);
}

// sums54: 27 x (re, im) doubles in the reference's gbuf row order; A72/b12 the
// symmetric unpack of ICP.cu:419-428 (either may be null)
long long ORC(icp_combined)(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                            const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                            const float *nmap_g_prev, size_t mstep, int rows, int cols, float distThres, float angleThres, int y0,
                            int y1, double *sums54, double *A72, double *b12) {
    double sums[54];
    long long inl = icp_combined<CF>(load_mat33<CF>(Rcurr18), load_vec3<CF>(tcurr6), vmap_curr, nmap_curr, load_mat33<CF>(Rprev_inv18),
                                     load_vec3<CF>(tprev6), Intr{intr4[0], intr4[1], intr4[2], intr4[3]}, vmap_g_prev, nmap_g_prev, mstep,
                                     rows, cols, distThres, angleThres, y0, y1, sums);
    if (sums54) std::memcpy(sums54, sums, sizeof(sums));
    if (A72 && b12) icp_unpack(sums, A72, b12);
    return inl;
}

void ORC(tsdf_hessian)(const float *depthScaled, size_t dstep, int drows, int dcols, const int *res, float voxel_size,
                       const float *Rv2c36, const float *tv2c12, float tranc_dist, const float *intr4, const float *gt, float *real_out,
                       float *grad_out, float *hess_out, int *count_out, int z0, int z1, double *out4) {
    tsdf_hessian<CF>(depthScaled, dstep, drows, dcols, res, voxel_size, Rv2c36, tv2c12, tranc_dist,
                     Intr{intr4[0], intr4[1], intr4[2], intr4[3]}, gt, real_out, grad_out, hess_out, count_out, z0, z1, out4);
}
void ORC(tsdf_loss)(const float *depthScaled, size_t dstep, int drows, int dcols, const int *res, float voxel_size, const float *Rv2c9,
                    const float *tv2c3, float tranc_dist, const float *intr4, const float *gt, float *real_out, int *count_out, int z0,
                    int z1, double *out2) {
    tsdf_loss(depthScaled, dstep, drows, dcols, res, voxel_size, Rv2c9, tv2c3, tranc_dist, Intr{intr4[0], intr4[1], intr4[2], intr4[3]}, gt,
              real_out, count_out, z0, z1, out2);
}

// ---- host algebra ----------------------------------------------------------
static M4 ld4(const float *p) { M4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = hc(p[(i * 4 + j) * 2], p[(i * 4 + j) * 2 + 1]); return m; }
static void st4(const M4 &m, float *p) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { p[(i * 4 + j) * 2] = m.m[i][j].real(); p[(i * 4 + j) * 2 + 1] = m.m[i][j].imag(); } }
static M3 ld3(const float *p) { M3 m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[i][j] = hc(p[(i * 3 + j) * 2], p[(i * 3 + j) * 2 + 1]); return m; }

void ORC(m4_inverse)(const float *in32, float *out32) { st4(m4_inverse(ld4(in32)), out32); }
void ORC(m4_mul)(const float *a32, const float *b32, float *out32) { st4(m4_mul(ld4(a32), ld4(b32)), out32); }
void ORC(m3_inverse)(const float *in18, float *out18) { m3_to_floats(m3_inverse(ld3(in18)), out18); }
double ORC(det6_real)(const double *A72) { return det6_real(A72); }
void ORC(llt_solve6)(const double *A72, const double *b12, double *x12) {
    hcd x[6];
    llt_solve6(A72, b12, x);
    for (int i = 0; i < 6; ++i) { x12[2 * i] = x[i].real(); x12[2 * i + 1] = x[i].imag(); }
}
// Rinc = Rz(gamma) Ry(beta) Rx(alpha), each angle a complex float (re, im)
void ORC(rinc)(const float *alpha2, const float *beta2, const float *gamma2, float *out18) {
    M3 r = m3_mul(m3_mul(angle_axis(hc(gamma2[0], gamma2[1]), 2), angle_axis(hc(beta2[0], beta2[1]), 1)), angle_axis(hc(alpha2[0], alpha2[1]), 0));
    m3_to_floats(r, out18);
}

// ---- pipeline --------------------------------------------------------------
typedef KinFu<CF> KF;
void *ORC(kf_create)(const KfParams *p) { KF *k = new KF(); k->set_parameters(*p); return k; }
void ORC(kf_destroy)(void *h) { delete (KF *)h; }
// gt poses: n camera-to-world 4x4 complex matrices (32 floats each, row-major)
void ORC(kf_set_gt_poses)(void *h, int n, const float *c2w32) { KF *k = (KF *)h; k->gt_poses.clear(); for (int i = 0; i < n; ++i) k->gt_poses.push_back(ld4(c2w32 + 32 * i)); }
int ORC(kf_process_frame)(void *h, const uint16_t *depth) { return ((KF *)h)->process_frame(depth); }
int ORC(kf_frame_id)(void *h) { return ((KF *)h)->frame_id; }
int ORC(kf_num_poses)(void *h) { return (int)((KF *)h)->record.size(); }
void ORC(kf_get_world2camera)(void *h, int idx, float *out32) { KF *k = (KF *)h; if (idx < 0) idx += (int)k->record.size(); st4(k->record[idx], out32); }
float ORC(kf_tranc_dist)(void *h) { return ((KF *)h)->tranc_dist; }
long long ORC(kf_last_U)(void *h) { return ((KF *)h)->last_U; }
long long ORC(kf_last_hits)(void *h) { return ((KF *)h)->last_hits; }
int ORC(kf_icp_log_size)(void *h) { return (int)((KF *)h)->icp_log.size(); }
void ORC(kf_icp_log)(void *h, double *out) { KF *k = (KF *)h; std::memcpy(out, k->icp_log.data(), k->icp_log.size() * sizeof(double)); }
const float *ORC(kf_value)(void *h) { return ((KF *)h)->value.data(); }
const float *ORC(kf_grad)(void *h) { return ((KF *)h)->grad.data(); }
const int *ORC(kf_weight)(void *h) { return ((KF *)h)->weight.data(); }
// which: 0 depths_curr 1 vmaps_curr 2 nmaps_curr 3 vmaps_g_prev 4 nmaps_g_prev
const float *ORC(kf_map)(void *h, int which, int level) {
    KF *k = (KF *)h;
    switch (which) {
        case 0: return k->depths_curr[level].data();
        case 1: return k->vmaps_curr[level].data();
        case 2: return k->nmaps_curr[level].data();
        case 3: return k->vmaps_g_prev[level].data();
        case 4: return k->nmaps_g_prev[level].data();
    }
    return nullptr;
}

}  // extern "C"
