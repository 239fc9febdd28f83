// ORACLE — TEST INFRASTRUCTURE ONLY (see oc_complex.hpp for the rule).
// CPU restatement of the host side of XKinectFusion: the Eigen algebra the
// orchestrator performs between kernel launches, and the per-frame pipeline
// (KinectFusionReconstruction.cpp:9-332).  Host scalars are std::complex, as
// in the reference (Internal.h:22-23).
//
// parity unpinned: Eigen (version unpinned, system package in the reference
// build; KinectFusionReconstruction.h:10-12) is absent from /root/reference
// and from this image.  The routines below restate Eigen's published
// fixed-size algorithms (cofactor inverses, unblocked lower LLT, AngleAxis
// Rodrigues form, coefficient-wise small products); their operation order
// could not be checked against an Eigen build.
#pragma once
#include "oc_kernels.hpp"
#include <complex>
#include <cstdio>
#include <vector>

namespace oc {

typedef std::complex<float> hc;    // hostComplex
typedef std::complex<double> hcd;  // hostComplexICP

struct M4 { hc m[4][4]; };  // Eigen::Matrix4cf, indexed (row, col)
struct M3 { hc m[3][3]; };

inline M4 m4_identity() { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = hc(i == j ? 1.f : 0.f, 0.f); return r; }

// coefficient-wise product, inner index ascending (Eigen lazy product for
// small fixed sizes)
inline M4 m4_mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            hc s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            hc s = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 3; ++k) s = s + a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

// Eigen general 4x4 inverse (Inverse_impl: cofactor_4x4 / general_det3_helper)
inline hc det3_helper(const M4 &a, int i1, int i2, int i3, int j1, int j2, int j3) {
    return a.m[i1][j1] * (a.m[i2][j2] * a.m[i3][j3] - a.m[i2][j3] * a.m[i3][j2]);
}
inline hc cofactor4(const M4 &a, int i, int j) {
    int i1 = (i + 1) % 4, i2 = (i + 2) % 4, i3 = (i + 3) % 4;
    int j1 = (j + 1) % 4, j2 = (j + 2) % 4, j3 = (j + 3) % 4;
    return det3_helper(a, i1, i2, i3, j1, j2, j3) + det3_helper(a, i2, i3, i1, j1, j2, j3) +
           det3_helper(a, i3, i1, i2, j1, j2, j3);
}
inline M4 m4_inverse(const M4 &a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            hc c = cofactor4(a, j, i);
            r.m[i][j] = ((i + j) & 1) ? -c : c;
        }
    // det = sum_k a(k,0) * r(0,k)
    hc det = a.m[0][0] * r.m[0][0];
    for (int k = 1; k < 4; ++k) det = det + a.m[k][0] * r.m[0][k];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = r.m[i][j] / det;
    return r;
}
// Eigen 3x3 inverse: cofactors, det from column 0, multiply by 1/det
inline hc cofactor3(const M3 &a, int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];
}
inline M3 m3_inverse(const M3 &a) {
    hc c0 = cofactor3(a, 0, 0), c1 = cofactor3(a, 1, 0), c2 = cofactor3(a, 2, 0);
    hc det = (c0 * a.m[0][0] + c1 * a.m[1][0]) + c2 * a.m[2][0];
    hc invdet = hc(1.f, 0.f) / det;
    M3 r;
    r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
    r.m[1][0] = cofactor3(a, 0, 1) * invdet; r.m[1][1] = cofactor3(a, 1, 1) * invdet; r.m[1][2] = cofactor3(a, 2, 1) * invdet;
    r.m[2][0] = cofactor3(a, 0, 2) * invdet; r.m[2][1] = cofactor3(a, 1, 2) * invdet; r.m[2][2] = cofactor3(a, 2, 2) * invdet;
    return r;
}

// A.real().determinant() for 6x6 double: partial-pivot LU
inline double det6_real(const double A[72]) {
    double a[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i][j] = A[2 * (i * 6 + j)];
    double det = 1.0;
    for (int k = 0; k < 6; ++k) {
        int p = k; double best = std::fabs(a[k][k]);
        for (int i = k + 1; i < 6; ++i) if (std::fabs(a[i][k]) > best) { best = std::fabs(a[i][k]); p = i; }
        if (best == 0.0) return 0.0;
        if (p != k) { for (int j = 0; j < 6; ++j) std::swap(a[k][j], a[p][j]); det = -det; }
        det *= a[k][k];
        for (int i = k + 1; i < 6; ++i) {
            double f = a[i][k] / a[k][k];
            for (int j = k + 1; j < 6; ++j) a[i][j] -= f * a[k][j];
        }
    }
    return det;
}

// Eigen LLT<Matrix<complex<double>,6,6>, Lower>::solve: unblocked lower
// Cholesky that reads real(A(k,k)) and uses L L^H (KinectFusionReconstruction.cpp:211;
// SURVEY Appendix B "Complex LLT is Hermitian").  A is (row i, col j) at [i*6+j]
// and symmetric, so storage order does not matter.
inline void llt_solve6(const double A[72], const double b[12], hcd x[6]) {
    hcd L[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) L[i][j] = hcd(A[2 * (i * 6 + j)], A[2 * (i * 6 + j) + 1]);
    for (int k = 0; k < 6; ++k) {
        double xk = L[k][k].real();
        for (int j = 0; j < k; ++j) xk -= std::norm(L[k][j]);
        if (xk <= 0.0) break;  // Eigen stops factorising (info = NumericalIssue)
        xk = std::sqrt(xk);
        L[k][k] = hcd(xk, 0.0);
        for (int i = k + 1; i < 6; ++i) {
            hcd s = L[i][k];
            for (int j = 0; j < k; ++j) s -= L[i][j] * std::conj(L[k][j]);
            L[i][k] = s / xk;
        }
    }
    hcd y[6];
    for (int i = 0; i < 6; ++i) y[i] = hcd(b[2 * i], b[2 * i + 1]);
    for (int i = 0; i < 6; ++i) {  // L y = b, column-oriented
        y[i] = y[i] / L[i][i];
        for (int r = i + 1; r < 6; ++r) y[r] -= y[i] * L[r][i];
    }
    for (int i = 5; i >= 0; --i) {  // L^H x = y
        hcd s = y[i];
        for (int r = i + 1; r < 6; ++r) s -= std::conj(L[r][i]) * y[r];
        y[i] = s / std::conj(L[i][i]);
    }
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// Eigen::AngleAxis<complex<float>>::toRotationMatrix() about a unit axis
inline M3 angle_axis(hc angle, int axis) {
    hc ax[3] = {hc(0, 0), hc(0, 0), hc(0, 0)};
    ax[axis] = hc(1.f, 0.f);
    hc s = std::sin(angle), c = std::cos(angle);
    hc sin_axis[3] = {s * ax[0], s * ax[1], s * ax[2]};
    hc one_c = hc(1.f, 0.f) - c;
    hc cos1_axis[3] = {one_c * ax[0], one_c * ax[1], one_c * ax[2]};
    M3 r;
    hc tmp;
    tmp = cos1_axis[0] * ax[1]; r.m[0][1] = tmp - sin_axis[2]; r.m[1][0] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; r.m[0][2] = tmp + sin_axis[1]; r.m[2][0] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; r.m[1][2] = tmp - sin_axis[0]; r.m[2][1] = tmp + sin_axis[0];
    for (int i = 0; i < 3; ++i) r.m[i][i] = cos1_axis[i] * ax[i] + c;
    return r;
}

inline void m3_to_floats(const M3 &a, float out[18]) {  // row-major (Matrix3frm / MatS33)
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { out[(i * 3 + j) * 2] = a.m[i][j].real(); out[(i * 3 + j) * 2 + 1] = a.m[i][j].imag(); }
}
inline M3 m4_rot(const M4 &a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j]; return r; }
inline void m4_trans(const M4 &a, hc t[3]) { for (int i = 0; i < 3; ++i) t[i] = a.m[i][3]; }

// ---------------------------------------------------------------------------
struct KfParams {  // the 25 YAML keys of SetYamlParameters (:12-72)
    int tsdf_size[3];
    float tsdf_voxel_size;
    int max_integration_weight;
    float thres_range;
    float init[3];
    float r_deg[3];
    int depth_width, depth_height;
    float fx, fy, cx, cy;
    int num_levels;
    float distThres;
    float angleThres_deg;
    float biInterpolate_threshold;
    float trunc_logistic_k;
    int flag_use_gtPose;
    int frame_step;
    // CSFD seed: i*seed_h on world2camera(seed_row, seed_col); row < 0 = none
    // (the seed line is commented out in the reference, :22)
    int seed_row, seed_col;
    float seed_h;
};

template <class C>
struct KinFu {
    KfParams p;
    Intr intr;
    int res[3];
    float voxel_size, tranc_dist, angleThres;
    int icp_iterations[3];
    M4 world2camera, world2volume;
    std::vector<M4> record;
    int frame_id;
    size_t vstep;
    std::vector<float> value, grad; std::vector<int> weight;
    std::vector<float> depthScaled;
    std::vector<std::vector<float>> depths_curr, vmaps_curr, nmaps_curr, vmaps_g_prev, nmaps_g_prev;
    std::vector<M4> gt_poses;  // c2w, real
    // diagnostics of the last frame
    long long last_U, last_hits; std::vector<double> icp_log;  // per iteration: 54 sums + inliers

    int lrows(int l) const { return p.depth_height >> l; }
    int lcols(int l) const { return p.depth_width >> l; }
    size_t lstep(int l) const { return (size_t)lcols(l) * 2 * sizeof(float); }

    void set_parameters(const KfParams &prm) {  // SetYamlParameters :9-73, AllocateBuffers :75-106, TsdfVolume.cpp:11-38
        p = prm;
        for (int i = 0; i < 3; ++i) res[i] = p.tsdf_size[i];
        voxel_size = p.tsdf_voxel_size;
        world2camera = m4_identity();
        if (p.seed_row >= 0) world2camera.m[p.seed_row][p.seed_col].imag(p.seed_h);
        record.clear();
        record.push_back(world2camera);
        world2volume = m4_identity();
        float rx = p.r_deg[0] / 180.0f * float(M_PI), ry = p.r_deg[1] / 180.0f * float(M_PI), rz = p.r_deg[2] / 180.0f * float(M_PI);
        // real AngleAxisf product Rx*Ry*Rz (as rotation matrices)
        auto R = [](float a, int axis) { return angle_axis(hc(a, 0.f), axis); };
        M3 rot = m3_mul(m3_mul(R(rx, 0), R(ry, 1)), R(rz, 2));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) world2volume.m[i][j] = hc(rot.m[i][j].real(), 0.f);
        for (int i = 0; i < 3; ++i) world2volume.m[i][3] = hc(p.init[i], 0.f);
        intr = Intr{p.fx, p.fy, p.cx, p.cy};
        int iters[] = {5, 4, 3};
        for (int i = 0; i < p.num_levels && i < 3; ++i) icp_iterations[i] = iters[i];
        angleThres = float(std::sin(p.angleThres_deg / 180.f * M_PI));
        depths_curr.resize(p.num_levels); vmaps_curr.resize(p.num_levels); nmaps_curr.resize(p.num_levels);
        vmaps_g_prev.resize(p.num_levels); nmaps_g_prev.resize(p.num_levels);
        for (int l = 0; l < p.num_levels; ++l) {
            size_t n = (size_t)lrows(l) * lcols(l) * 2;
            depths_curr[l].assign(n, 0.f);
            vmaps_curr[l].assign(n * 3, 0.f); nmaps_curr[l].assign(n * 3, 0.f);
            vmaps_g_prev[l].assign(n * 3, 0.f); nmaps_g_prev[l].assign(n * 3, 0.f);
        }
        size_t nv = (size_t)res[0] * res[1] * res[2];
        vstep = (size_t)res[0] * sizeof(float);
        value.assign(nv, 0.f); grad.assign(nv, 0.f); weight.assign(nv, 0);
        float default_tranc = voxel_size * p.thres_range;
        tranc_dist = std::max(default_tranc, 2.1f * voxel_size);
        depthScaled.assign((size_t)p.depth_width * p.depth_height, 0.f);
        frame_id = 0;
        last_U = last_hits = 0;
    }

    void surface_measure(const uint16_t *depth) {  // :280-299
        bilateral<C>(depth, (size_t)p.depth_width * 2, p.depth_height, p.depth_width, depths_curr[0].data(), lstep(0));
        for (int l = 1; l < p.num_levels; ++l)
            pyr_down<C>(depths_curr[l - 1].data(), lstep(l - 1), lrows(l - 1), lcols(l - 1), depths_curr[l].data(), lstep(l));
        for (int l = 0; l < p.num_levels; ++l) {
            create_vmap<C>(intr_level(intr, l), depths_curr[l].data(), lstep(l), lrows(l), lcols(l), vmaps_curr[l].data(), lstep(l));
            create_nmap<C>(lrows(l), lcols(l), vmaps_curr[l].data(), nmaps_curr[l].data(), lstep(l));
        }
    }

    int pose_estimate() {  // AlignDepthToReconstruction :167-173 + PoseEstimate :177-235
        icp_log.clear();
        if (frame_id == 0) return 0;
        M4 c2w_prev = m4_inverse(record.back());
        M3 Rprev = m4_rot(c2w_prev);
        hc tprev[3]; m4_trans(c2w_prev, tprev);
        M3 Rprev_inv = m3_inverse(Rprev);
        M3 Rcurr = Rprev;
        hc tcurr[3] = {tprev[0], tprev[1], tprev[2]};
        M4 c2w_curr = c2w_prev;
        float fRprev_inv[18], ftprev[6];
        m3_to_floats(Rprev_inv, fRprev_inv);
        for (int i = 0; i < 3; ++i) { ftprev[2 * i] = tprev[i].real(); ftprev[2 * i + 1] = tprev[i].imag(); }
        for (int level = p.num_levels - 1; level >= 0; --level) {
            for (int iter = 0; iter < icp_iterations[level]; ++iter) {
                float fRcurr[18], ftcurr[6];
                m3_to_floats(Rcurr, fRcurr);
                for (int i = 0; i < 3; ++i) { ftcurr[2 * i] = tcurr[i].real(); ftcurr[2 * i + 1] = tcurr[i].imag(); }
                double sums[54], A[72], b[12];
                long long inl = icp_combined<C>(load_mat33<C>(fRcurr), load_vec3<C>(ftcurr), vmaps_curr[level].data(),
                                                nmaps_curr[level].data(), load_mat33<C>(fRprev_inv), load_vec3<C>(ftprev),
                                                intr_level(intr, level), vmaps_g_prev[level].data(), nmaps_g_prev[level].data(),
                                                lstep(level), lrows(level), lcols(level), p.distThres, angleThres, 0, lrows(level), sums);
                for (int k = 0; k < 54; ++k) icp_log.push_back(sums[k]);
                icp_log.push_back((double)inl);
                icp_unpack(sums, A, b);
                double det = det6_real(A);
                if (std::fabs(det) < 1e-15 || std::isnan(det)) return 0;
                hcd sol[6];
                llt_solve6(A, b, sol);
                hc result[6];
                for (int i = 0; i < 6; ++i) result[i] = hc((float)sol[i].real(), (float)sol[i].imag());
                hc alpha = result[0], beta = result[1], gamma = result[2];
                M3 Rinc = m3_mul(m3_mul(angle_axis(gamma, 2), angle_axis(beta, 1)), angle_axis(alpha, 0));
                hc tn[3];
                for (int i = 0; i < 3; ++i) {
                    hc s = Rinc.m[i][0] * tcurr[0];
                    s = s + Rinc.m[i][1] * tcurr[1];
                    s = s + Rinc.m[i][2] * tcurr[2];
                    tn[i] = s + result[3 + i];
                }
                for (int i = 0; i < 3; ++i) tcurr[i] = tn[i];
                Rcurr = m3_mul(Rinc, Rcurr);
                for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c2w_curr.m[i][j] = Rcurr.m[i][j]; c2w_curr.m[i][3] = tcurr[i]; }
                c2w_curr.m[3][3] = hc(1.f, 0.f);
            }
        }
        world2camera = m4_inverse(c2w_curr);
        record.push_back(world2camera);
        return 1;
    }

    void integrate_frame(const uint16_t *depth) {  // :237-278 + CalculatePointCloud :302-332
        if (p.flag_use_gtPose) {
            M4 c2w = gt_poses[frame_id];
            world2camera = m4_inverse(c2w);
            record.back() = world2camera;
        }
        M4 c2w = m4_inverse(record.back());
        M4 c2v = m4_mul(world2volume, c2w);
        M4 v2c = m4_inverse(c2v);
        float fRv2c[18], ftv2c[6];
        m3_to_floats(m4_rot(v2c), fRv2c);
        for (int i = 0; i < 3; ++i) { ftv2c[2 * i] = v2c.m[i][3].real(); ftv2c[2 * i + 1] = v2c.m[i][3].imag(); }
        scale_depth(depth, (size_t)p.depth_width * 2, p.depth_height, p.depth_width, depthScaled.data(), (size_t)p.depth_width * 4);
        last_U = integrate<C>(depthScaled.data(), (size_t)p.depth_width * 4, p.depth_height, p.depth_width, value.data(),
                              weight.data(), grad.data(), vstep, res, tranc_dist, p.max_integration_weight, load_mat33<C>(fRv2c),
                              load_vec3<C>(ftv2c), intr, voxel_size, p.biInterpolate_threshold, 0, res[2]);
        // raycast from world2camera (== record.back() on every path that reaches here)
        M4 c2w2 = m4_inverse(world2camera);
        M4 c2v2 = m4_mul(world2volume, c2w2);
        M4 v2w = m4_inverse(world2volume);
        float fRc2v[18], ftc2v[6], fRv2w[18], ftv2w[6];
        m3_to_floats(m4_rot(c2v2), fRc2v); m3_to_floats(m4_rot(v2w), fRv2w);
        for (int i = 0; i < 3; ++i) {
            ftc2v[2 * i] = c2v2.m[i][3].real(); ftc2v[2 * i + 1] = c2v2.m[i][3].imag();
            ftv2w[2 * i] = v2w.m[i][3].real(); ftv2w[2 * i + 1] = v2w.m[i][3].imag();
        }
        last_hits = raycast<C>(intr, load_mat33<C>(fRc2v), load_vec3<C>(ftc2v), load_mat33<C>(fRv2w), load_vec3<C>(ftv2w), tranc_dist,
                               res, voxel_size, value.data(), grad.data(), vstep, vmaps_g_prev[0].data(), nmaps_g_prev[0].data(),
                               lstep(0), lrows(0), lcols(0));
        for (int l = 1; l < p.num_levels; ++l) {
            resize_map<C>(false, lrows(l - 1), lcols(l - 1), vmaps_g_prev[l - 1].data(), lstep(l - 1), vmaps_g_prev[l].data(), lstep(l));
            resize_map<C>(true, lrows(l - 1), lcols(l - 1), nmaps_g_prev[l - 1].data(), lstep(l - 1), nmaps_g_prev[l].data(), lstep(l));
        }
    }

    int process_frame(const uint16_t *depth) {  // :147-159
        surface_measure(depth);
        int align_return = p.flag_use_gtPose ? 1 : pose_estimate();
        if (frame_id > 0 && !align_return) return 0;
        integrate_frame(depth);
        frame_id += p.frame_step;
        return 1;
    }
};

}  // namespace oc
