// ORACLE — TEST INFRASTRUCTURE ONLY (see oc_complex.hpp for the rule).
// CPU restatement of the XKinectFusion device kernels, one plain loop nest per
// CUDA kernel, templated over the single-complex type C so that the same text
// runs over oc::cplx<float> (the oracle) and over the reference's own
// ::complex<float> (oracle/_ref).  Each function cites the reference lines it
// follows.  Built -O2 -ffp-contract=off: no FMA contraction, so every float
// operation rounds exactly where the source expression says it does.
//
// parity unpinned at kernel level: the reference ships no tests, fixtures or
// golden vectors for these kernels and its .cu files cannot be compiled here
// (they need the CUDA toolkit headers and libcu++).  What IS pinned: the
// scalar arithmetic underneath (oc_complex.hpp, against oracle/_ref and the
// test_CSFD known answers).
#pragma once
#include "oc_complex.hpp"
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oc {

struct Intr { float fx, fy, cx, cy; };  // Internal.h:49-59
inline Intr intr_level(const Intr &k, int level) {
    int div = 1 << level;
    return Intr{k.fx / div, k.fy / div, k.cx / div, k.cy / div};
}

// ---- 3-vector / 3x3 helpers (Internal.h:63-154) ---------------------------
template <class C> struct vec3 { C x, y, z; };
template <class C> inline vec3<C> mk3(C x, C y, C z) { vec3<C> t; t.x = x; t.y = y; t.z = z; return t; }
template <class C> inline C dot(const vec3<C> &a, const vec3<C> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class C> inline vec3<C> operator+(const vec3<C> &a, const vec3<C> &b) { return mk3<C>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class C> inline vec3<C> operator-(const vec3<C> &a, const vec3<C> &b) { return mk3<C>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class C> inline vec3<C> operator*(const vec3<C> &a, const float &v) { return mk3<C>(a.x * v, a.y * v, a.z * v); }
template <class C> inline vec3<C> operator*(const vec3<C> &a, const C &v) { return mk3<C>(a.x * v, a.y * v, a.z * v); }
template <class C> inline C norm3(const vec3<C> &v) { return sqrt(dot(v, v)); }
template <class C> inline C squarednorm3(const vec3<C> &v) { return dot(v, v); }
// normalized() evaluates norm(v) once per component (Internal.h:134-137); the
// three evaluations give the same value, so one is kept here.
template <class C> inline vec3<C> normalized3(const vec3<C> &v) { C n = norm3(v); return mk3<C>(v.x / n, v.y / n, v.z / n); }
template <class C> inline vec3<C> cross3(const vec3<C> &a, const vec3<C> &b) {
    return mk3<C>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class C> struct mat33 { vec3<C> data[3]; };
template <class C> inline vec3<C> operator*(const mat33<C> &m, const vec3<C> &v) {
    return mk3<C>(dot(m.data[0], v), dot(m.data[1], v), dot(m.data[2], v));
}
// load a row-major 3x3 / 3-vector from interleaved (re,im) floats — the layout
// device_cast<> reinterprets (Internal.h:42-45)
template <class C> inline mat33<C> load_mat33(const float *p) {
    mat33<C> m;
    for (int r = 0; r < 3; ++r) {
        m.data[r].x = C(p[r * 6 + 0], p[r * 6 + 1]);
        m.data[r].y = C(p[r * 6 + 2], p[r * 6 + 3]);
        m.data[r].z = C(p[r * 6 + 4], p[r * 6 + 5]);
    }
    return m;
}
template <class C> inline vec3<C> load_vec3(const float *p) {
    return mk3<C>(C(p[0], p[1]), C(p[2], p[3]), C(p[4], p[5]));
}

// CUDA conversion intrinsics
inline int f2i_rd(float x) { return (int)std::floor(x); }       // __float2int_rd
inline int f2i_rn(float x) { return (int)std::nearbyintf(x); }  // __float2int_rn (ties to even)

template <class T> inline T *row_ptr(T *base, size_t step_bytes, int y) {
    return (T *)((char *)base + (size_t)y * step_bytes);
}
template <class T> inline const T *row_ptr(const T *base, size_t step_bytes, int y) {
    return (const T *)((const char *)base + (size_t)y * step_bytes);
}

// ---- TsdfFusion.cu:4-30 initializeVolume ----------------------------------
inline void init_volume(float *value, int *weight, float *grad, size_t step, const int res[3]) {
    for (int z = 0; z < res[2]; ++z)
        for (int y = 0; y < res[1]; ++y)
            for (int x = 0; x < res[0]; ++x) {
                row_ptr(value, step, res[1] * z + y)[x] = 0.f;
                row_ptr(weight, step, res[1] * z + y)[x] = 0;
                row_ptr(grad, step, res[1] * z + y)[x] = 0.f;
            }
}

// ---- TsdfFusion.cu:68-82 scaleDepthKernal ---------------------------------
inline void scale_depth(const uint16_t *depth, size_t dstep, int rows, int cols, float *scaled, size_t sstep) {
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int Dp = row_ptr(depth, dstep, y)[x];
            if (Dp > 5000 || Dp < 200) { row_ptr(scaled, sstep, y)[x] = 0; continue; }
            row_ptr(scaled, sstep, y)[x] = float(Dp) / 1000.f;
        }
}

// ---- TsdfFusion.cu:85-171 tsdfFusionKernal --------------------------------
// returns U = number of voxels whose weight is written.
// z range [z0, z1) lets a z-slab be integrated on its own (multi-GPU tests).
template <class C>
long long integrate(const float *depthScaled, size_t dstep, int drows, int dcols,
                    float *value, int *weight, float *grad, size_t vstep, const int res[3],
                    float tranc_dist, int max_weight, const mat33<C> &Rv2c, const vec3<C> &tv2c,
                    Intr intr, float voxel_size, float threshold, int z0, int z1) {
    long long updated = 0;
    float tranc_dist_inv = 1.0f / tranc_dist;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : updated)
    for (int y = 0; y < res[1]; ++y)
        for (int x = 0; x < res[0]; ++x)
            for (int z = z0; z < z1; ++z) {
                float *pos = row_ptr(value, vstep, res[1] * z + y) + x;
                int *weight_pos = row_ptr(weight, vstep, res[1] * z + y) + x;
                float *grad_pos = row_ptr(grad, vstep, res[1] * z + y) + x;
                C v_g_x = (x + 0.5f) * voxel_size;
                C v_g_y = (y + 0.5f) * voxel_size;
                C v_g_z = (z + 0.5f) * voxel_size;
                vec3<C> v_g = mk3<C>(v_g_x, v_g_y, v_g_z);
                vec3<C> v_c = Rv2c * v_g + tv2c;
                C inv_z = 1.0f / (v_c.z);
                if (inv_z.real() < 0) continue;
                C image_x = v_c.x * intr.fx * inv_z + intr.cx;
                C image_y = v_c.y * intr.fy * inv_z + intr.cy;
                int coo_x = f2i_rd(image_x.real() - 0.5f);
                int coo_y = f2i_rd(image_y.real() - 0.5f);
                if (coo_x > 1 && coo_y > 1 && coo_x < dcols - 1 && coo_y < drows - 1) {
                    int near_x = f2i_rn(image_x.real());
                    int near_y = f2i_rn(image_y.real());
                    C Dp_near(row_ptr(depthScaled, dstep, near_y)[near_x], 0.0f);
                    C Dp;
                    float d00 = row_ptr(depthScaled, dstep, coo_y)[coo_x];
                    float d10 = row_ptr(depthScaled, dstep, coo_y)[coo_x + 1];
                    float d01 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x];
                    float d11 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x + 1];
                    float gird_max = std::fmax(d00, std::fmax(d01, std::fmax(d10, d11)));
                    float gird_min = std::fmin(d00, std::fmin(d01, std::fmin(d10, d11)));
                    if (gird_max - gird_min < threshold && d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
                        C one(1.0f, 0.0f);
                        C a = image_x - C(coo_x + 0.5f, 0.0f);
                        C b = image_y - C(coo_y + 0.5f, 0.0f);
                        C Dp_inter = d00 * (one - a) * (one - b) + d10 * a * (one - b) + d01 * (one - a) * b + d11 * a * b;
                        Dp = Dp_inter;
                    } else {
                        Dp = Dp_near;
                    }
                    C xl = (image_x - intr.cx) / intr.fx;
                    C yl = (image_y - intr.cy) / intr.fy;
                    vec3<C> v_c_1 = mk3<C>(Dp * xl, Dp * yl, Dp);
                    C sdf = norm3(v_c_1) - norm3(v_c);
                    if (Dp.real() > 0 && sdf.real() >= -tranc_dist) {
                        C tsdf;
                        if (sdf.real() > tranc_dist)
                            tsdf = C(1.0f, 0.0f);
                        else
                            tsdf = sdf * tranc_dist_inv;
                        C tsdf_prev(*pos, *grad_pos);
                        int weight_prev = *weight_pos;
                        int Wrk = 1;
                        C tsdf_new = (tsdf_prev * float(weight_prev) + float(Wrk) * tsdf) / float(weight_prev + Wrk);
                        int weight_new = std::min(weight_prev + Wrk, max_weight);
                        *pos = tsdf_new.real();
                        *weight_pos = weight_new;
                        *grad_pos = tsdf_new.imag();
                        ++updated;
                    }
                }
            }
    return updated;
}

// ---- RayCaster.cu:26-141, 197-310 -----------------------------------------
template <class C>
struct RayCasterO {
    mat33<C> Rc2v; vec3<C> tc2v; mat33<C> Rv2w; vec3<C> tv2w;
    int res[3]; float voxel_size; float time_step; int cols, rows;
    const float *value; const float *grad; size_t vstep;
    Intr intr;
    float *vmap; float *nmap; size_t mstep;  // complex maps as (re,im) float pairs, 3 planes

    static int sgn(float val) { return (0.0f < val) - (val < 0.0f); }
    C *map_at(float *m, int row, int x) const { return (C *)((char *)m + (size_t)row * mstep) + x; }
    bool checkInds(const int g[3]) const {
        return g[0] >= 0 && g[1] >= 0 && g[2] >= 0 && g[0] < res[0] && g[1] < res[1] && g[2] < res[2];
    }
    C readTsdf(int x, int y, int z) const {  // :69-78, adds 1e-5 to the real part
        x = x % res[0]; y = y % res[1]; z = z % res[2];
        C r(row_ptr(value, vstep, res[1] * z + y)[x], row_ptr(grad, vstep, res[1] * z + y)[x]);
        r += 1e-5f;
        return r;
    }
    void getVoxel(float px, float py, float pz, int g[3]) const {
        g[0] = f2i_rd(px / voxel_size); g[1] = f2i_rd(py / voxel_size); g[2] = f2i_rd(pz / voxel_size);
    }
    C interp(const vec3<C> &point) const {  // :99-141
        int g[3];
        getVoxel(point.x.real(), point.y.real(), point.z.real(), g);
        float qnan = qnan_f();
        if (g[0] <= 0 || g[0] >= res[0] - 1) return C(qnan, 0);
        if (g[1] <= 0 || g[1] >= res[1] - 1) return C(qnan, 0);
        if (g[2] <= 0 || g[2] >= res[2] - 1) return C(qnan, 0);
        float vx = (g[0] + 0.5f) * voxel_size;
        float vy = (g[1] + 0.5f) * voxel_size;
        float vz = (g[2] + 0.5f) * voxel_size;
        g[0] += -(sgn(vx - point.x.real()) + 1) >> 1;
        g[1] += -(sgn(vy - point.y.real()) + 1) >> 1;
        g[2] += -(sgn(vz - point.z.real()) + 1) >> 1;
        C a0 = (point.x - (g[0] + 0.5f) * voxel_size) / voxel_size;
        C b0 = (point.y - (g[1] + 0.5f) * voxel_size) / voxel_size;
        C c0 = (point.z - (g[2] + 0.5f) * voxel_size) / voxel_size;
        C one(1.0f, 0.0f);
        C a1 = one - a0, b1 = one - b0, c1 = one - c0;
        C r = readTsdf(g[0] + 0, g[1] + 0, g[2] + 0) * a1 * b1 * c1 +
              readTsdf(g[0] + 0, g[1] + 0, g[2] + 1) * a1 * b1 * c0 +
              readTsdf(g[0] + 0, g[1] + 1, g[2] + 0) * a1 * b0 * c1 +
              readTsdf(g[0] + 0, g[1] + 1, g[2] + 1) * a1 * b0 * c0 +
              readTsdf(g[0] + 1, g[1] + 0, g[2] + 0) * a0 * b1 * c1 +
              readTsdf(g[0] + 1, g[1] + 0, g[2] + 1) * a0 * b1 * c0 +
              readTsdf(g[0] + 1, g[1] + 1, g[2] + 0) * a0 * b0 * c1 +
              readTsdf(g[0] + 1, g[1] + 1, g[2] + 1) * a0 * b0 * c0;
        return r;
    }
    // returns 1 if a vertex was written
    int pixel(int x, int y) const {  // :197-310
        *map_at(vmap, y, x) = C(qnan_f(), 0);
        *map_at(nmap, y, x) = C(qnan_f(), 0);
        vec3<C> ray_start = tc2v;
        vec3<C> rn;
        rn.x = (x - intr.cx) / intr.fx;
        rn.y = (y - intr.cy) / intr.fy;
        rn.z = 1;
        vec3<C> ray_next = Rc2v * rn + tc2v;
        vec3<C> ray_dir = normalized3(ray_next - ray_start);
        ray_dir.x = (ray_dir.x == 0.f) ? C(1e-15f) : ray_dir.x;
        ray_dir.y = (ray_dir.y == 0.f) ? C(1e-15f) : ray_dir.y;
        ray_dir.z = (ray_dir.z == 0.f) ? C(1e-15f) : ray_dir.z;
        float time_start_volume = 0.2f;
        float time_exit_volume = 5.0f;
        float time_curr = time_start_volume;
        int g[3];
        {
            vec3<C> p = ray_start + ray_dir * time_curr;
            getVoxel(p.x.real(), p.y.real(), p.z.real(), g);
        }
        g[0] = std::max(0, std::min(g[0], res[0] - 1));
        g[1] = std::max(0, std::min(g[1], res[1] - 1));
        g[2] = std::max(0, std::min(g[2], res[2] - 1));
        C tsdf = readTsdf(g[0], g[1], g[2]);
        const float max_time = time_exit_volume;
        for (; time_curr < max_time; time_curr += time_step) {
            C tsdf_prev = tsdf;
            vec3<C> cp = ray_start + ray_dir * (time_curr + time_step);
            getVoxel(cp.x.real(), cp.y.real(), cp.z.real(), g);
            if (!checkInds(g)) break;
            tsdf = readTsdf(g[0], g[1], g[2]);
            if (tsdf_prev.real() < 0.f && tsdf.real() > 0.f) break;
            if (tsdf_prev.real() > 0.f && tsdf.real() < 0.f) {
                C Ftdt = interp(ray_start + ray_dir * (time_curr + time_step));
                if (std::isnan(Ftdt.real())) break;
                C Ft = interp(ray_start + ray_dir * time_curr);
                if (std::isnan(Ft.real())) break;
                C coef = Ft / (Ftdt - Ft);
                if (Ft.real() < 0.0f || Ftdt.real() > 0.0f) break;
                C Ts = time_curr - time_step * coef;
                vec3<C> vertex_found = ray_start + ray_dir * Ts;
                vec3<C> vertex_found_w = Rv2w * vertex_found + tv2w;
                *map_at(vmap, y, x) = vertex_found_w.x;
                *map_at(vmap, y + rows, x) = vertex_found_w.y;
                *map_at(vmap, y + 2 * rows, x) = vertex_found_w.z;
                getVoxel(vertex_found.x.real(), vertex_found.y.real(), vertex_found.z.real(), g);
                if (g[0] > 1 && g[1] > 1 && g[2] > 1 && g[0] < res[0] - 2 && g[1] < res[1] - 2 && g[2] < res[2] - 2) {
                    vec3<C> t, n;
                    float half_voxel_size = voxel_size * 0.5f;
                    t = vertex_found; t.x += half_voxel_size; C Fx1 = interp(t);
                    t = vertex_found; t.x -= half_voxel_size; C Fx2 = interp(t);
                    n.x = (Fx1 - Fx2);
                    t = vertex_found; t.y += half_voxel_size; C Fy1 = interp(t);
                    t = vertex_found; t.y -= half_voxel_size; C Fy2 = interp(t);
                    n.y = (Fy1 - Fy2);
                    t = vertex_found; t.z += half_voxel_size; C Fz1 = interp(t);
                    t = vertex_found; t.z -= half_voxel_size; C Fz2 = interp(t);
                    n.z = (Fz1 - Fz2);
                    if (squarednorm3(n).real() == 0) return 1;
                    vec3<C> n_g = Rv2w * normalized3(n);
                    *map_at(nmap, y, x) = n_g.x;
                    *map_at(nmap, y + rows, x) = n_g.y;
                    *map_at(nmap, y + 2 * rows, x) = n_g.z;
                }
                return 1;
            }
        }
        return 0;
    }
};

// Slab form used by the multi-GPU protocol tests (no reference counterpart): the same march,
// but only the steps whose sample voxel lies in the owned planes [z0, z1) are evaluated, and
// every volume read asserts that it stays inside the stored planes [zs0, zs1).  Writes the
// rank's first event key (step << 1 | no_vertex, INT_MAX if none) and its vertex / normal
// (zeros unless it produced them).  Returns the number of reads outside the stored planes.
template <class C>
struct RaySlabO : RayCasterO<C> {
    typedef RayCasterO<C> B;
    int zs0, zs1, z0, z1;
    mutable long long violations = 0;
    float rv(int x, int y, int z) const {
        if (z < zs0 || z >= zs1) { ++violations; return 0.f; }
        return row_ptr(B::value, B::vstep, B::res[1] * z + y)[x] + 1e-5f;
    }
    C rc(int x, int y, int z) const {
        x = x % B::res[0]; y = y % B::res[1]; z = z % B::res[2];
        if (z < zs0 || z >= zs1) { ++violations; return C(0.f, 0.f); }
        C r(row_ptr(B::value, B::vstep, B::res[1] * z + y)[x], row_ptr(B::grad, B::vstep, B::res[1] * z + y)[x]);
        r += 1e-5f;
        return r;
    }
    C interp(const vec3<C> &point) const {
        int g[3];
        B::getVoxel(point.x.real(), point.y.real(), point.z.real(), g);
        float qnan = qnan_f();
        if (g[0] <= 0 || g[0] >= B::res[0] - 1) return C(qnan, 0);
        if (g[1] <= 0 || g[1] >= B::res[1] - 1) return C(qnan, 0);
        if (g[2] <= 0 || g[2] >= B::res[2] - 1) return C(qnan, 0);
        float vs = B::voxel_size;
        float vx = (g[0] + 0.5f) * vs, vy = (g[1] + 0.5f) * vs, vz = (g[2] + 0.5f) * vs;
        g[0] += -(B::sgn(vx - point.x.real()) + 1) >> 1;
        g[1] += -(B::sgn(vy - point.y.real()) + 1) >> 1;
        g[2] += -(B::sgn(vz - point.z.real()) + 1) >> 1;
        C a0 = (point.x - (g[0] + 0.5f) * vs) / vs, b0 = (point.y - (g[1] + 0.5f) * vs) / vs, c0 = (point.z - (g[2] + 0.5f) * vs) / vs;
        C one(1.0f, 0.0f);
        C a1 = one - a0, b1 = one - b0, c1 = one - c0;
        return rc(g[0] + 0, g[1] + 0, g[2] + 0) * a1 * b1 * c1 + rc(g[0] + 0, g[1] + 0, g[2] + 1) * a1 * b1 * c0 +
               rc(g[0] + 0, g[1] + 1, g[2] + 0) * a1 * b0 * c1 + rc(g[0] + 0, g[1] + 1, g[2] + 1) * a1 * b0 * c0 +
               rc(g[0] + 1, g[1] + 0, g[2] + 0) * a0 * b1 * c1 + rc(g[0] + 1, g[1] + 0, g[2] + 1) * a0 * b1 * c0 +
               rc(g[0] + 1, g[1] + 1, g[2] + 0) * a0 * b0 * c1 + rc(g[0] + 1, g[1] + 1, g[2] + 1) * a0 * b0 * c0;
    }
    int pixel_slab(int x, int y) const {
        const int rows = B::rows;
        for (int p = 0; p < 3; ++p) { *B::map_at(B::vmap, y + p * rows, x) = C(0.f, 0.f); *B::map_at(B::nmap, y + p * rows, x) = C(0.f, 0.f); }
        vec3<C> ray_start = B::tc2v, rn;
        rn.x = (x - B::intr.cx) / B::intr.fx; rn.y = (y - B::intr.cy) / B::intr.fy; rn.z = 1;
        vec3<C> ray_dir = normalized3(B::Rc2v * rn + B::tc2v - ray_start);
        ray_dir.x = (ray_dir.x == 0.f) ? C(1e-15f) : ray_dir.x;
        ray_dir.y = (ray_dir.y == 0.f) ? C(1e-15f) : ray_dir.y;
        ray_dir.z = (ray_dir.z == 0.f) ? C(1e-15f) : ray_dir.z;
        float time_curr = 0.2f;
        const float max_time = 5.0f, time_step = B::time_step;
        int g[3], q[3];
        { vec3<C> p = ray_start + ray_dir * time_curr; B::getVoxel(p.x.real(), p.y.real(), p.z.real(), q); }
        for (int i = 0; i < 3; ++i) q[i] = std::max(0, std::min(q[i], B::res[i] - 1));
        int step = 0;
        for (; time_curr < max_time; time_curr += time_step, ++step) {
            vec3<C> cp = ray_start + ray_dir * (time_curr + time_step);
            B::getVoxel(cp.x.real(), cp.y.real(), cp.z.real(), g);
            if (!B::checkInds(g)) break;
            int pq[3] = {q[0], q[1], q[2]};
            q[0] = g[0]; q[1] = g[1]; q[2] = g[2];
            if (!(g[2] >= z0 && g[2] < z1)) continue;
            float tsdf_prev = rv(pq[0], pq[1], pq[2]);
            float tsdf = rv(g[0], g[1], g[2]);
            if (tsdf_prev < 0.f && tsdf > 0.f) return (step << 1) | 1;
            if (tsdf_prev > 0.f && tsdf < 0.f) {
                C Ftdt = interp(ray_start + ray_dir * (time_curr + time_step));
                if (std::isnan(Ftdt.real())) return (step << 1) | 1;
                C Ft = interp(ray_start + ray_dir * time_curr);
                if (std::isnan(Ft.real())) return (step << 1) | 1;
                C coef = Ft / (Ftdt - Ft);
                if (Ft.real() < 0.0f || Ftdt.real() > 0.0f) return (step << 1) | 1;
                C Ts = time_curr - time_step * coef;
                vec3<C> vf = ray_start + ray_dir * Ts;
                vec3<C> vw = B::Rv2w * vf + B::tv2w;
                *B::map_at(B::vmap, y, x) = vw.x; *B::map_at(B::vmap, y + rows, x) = vw.y; *B::map_at(B::vmap, y + 2 * rows, x) = vw.z;
                *B::map_at(B::nmap, y, x) = C(qnan_f(), 0);
                B::getVoxel(vf.x.real(), vf.y.real(), vf.z.real(), g);
                if (g[0] > 1 && g[1] > 1 && g[2] > 1 && g[0] < B::res[0] - 2 && g[1] < B::res[1] - 2 && g[2] < B::res[2] - 2) {
                    vec3<C> t, n;
                    float h = B::voxel_size * 0.5f;
                    t = vf; t.x += h; C Fx1 = interp(t); t = vf; t.x -= h; C Fx2 = interp(t); n.x = Fx1 - Fx2;
                    t = vf; t.y += h; C Fy1 = interp(t); t = vf; t.y -= h; C Fy2 = interp(t); n.y = Fy1 - Fy2;
                    t = vf; t.z += h; C Fz1 = interp(t); t = vf; t.z -= h; C Fz2 = interp(t); n.z = Fz1 - Fz2;
                    if (squarednorm3(n).real() == 0) return step << 1;
                    vec3<C> ng = B::Rv2w * normalized3(n);
                    *B::map_at(B::nmap, y, x) = ng.x; *B::map_at(B::nmap, y + rows, x) = ng.y; *B::map_at(B::nmap, y + 2 * rows, x) = ng.z;
                }
                return step << 1;
            }
        }
        return 0x7fffffff;
    }
};

template <class C>
long long raycast_slab(Intr intr, const mat33<C> &Rc2v, const vec3<C> &tc2v, const mat33<C> &Rv2w, const vec3<C> &tv2w, float tranc_dist,
                       const int res[3], float voxel_size, const float *value, const float *grad, size_t vstep, int zs0, int zs1, int z0,
                       int z1, float *vmap, float *nmap, size_t mstep, int rows, int cols, int *keys) {
    RaySlabO<C> rc;
    rc.Rc2v = Rc2v; rc.tc2v = tc2v; rc.Rv2w = Rv2w; rc.tv2w = tv2w;
    rc.res[0] = res[0]; rc.res[1] = res[1]; rc.res[2] = res[2];
    rc.voxel_size = voxel_size; rc.time_step = tranc_dist * 0.8f; rc.cols = cols; rc.rows = rows;
    rc.intr = intr; rc.value = value; rc.grad = grad; rc.vstep = vstep; rc.vmap = vmap; rc.nmap = nmap; rc.mstep = mstep;
    rc.zs0 = zs0; rc.zs1 = zs1; rc.z0 = z0; rc.z1 = z1;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) keys[y * cols + x] = rc.pixel_slab(x, y);
    return rc.violations;
}

template <class C>
long long raycast(Intr intr, const mat33<C> &Rc2v, const vec3<C> &tc2v, const mat33<C> &Rv2w, const vec3<C> &tv2w,
                  float tranc_dist, const int res[3], float voxel_size, const float *value, const float *grad,
                  size_t vstep, float *vmap, float *nmap, size_t mstep, int rows, int cols) {
    RayCasterO<C> rc;  // RayCaster.cu:327-363
    rc.Rc2v = Rc2v; rc.tc2v = tc2v; rc.Rv2w = Rv2w; rc.tv2w = tv2w;
    rc.res[0] = res[0]; rc.res[1] = res[1]; rc.res[2] = res[2];
    rc.voxel_size = voxel_size;
    rc.time_step = tranc_dist * 0.8f;
    rc.cols = cols; rc.rows = rows;
    rc.intr = intr; rc.value = value; rc.grad = grad; rc.vstep = vstep;
    rc.vmap = vmap; rc.nmap = nmap; rc.mstep = mstep;
    long long hits = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : hits)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) hits += rc.pixel(x, y);
    return hits;
}

// ---- Map.cu:155-199 bilateralKernel ---------------------------------------
// __expf is restated as expf (the fast-math intrinsic has no CPU twin).
template <class C>
void bilateral(const uint16_t *src, size_t sstep, int rows, int cols, float *dst, size_t dstep) {
    const float sigma_color = 30, sigma_space = 4.5f;  // Map.cu:4-5
    float sigma_space2_inv_half = 0.5f / (sigma_space * sigma_space);
    float sigma_color2_inv_half = 0.5f / (sigma_color * sigma_color);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int value = row_ptr(src, sstep, y)[x];
            const int R = 6, D = R * 2 + 1;
            int tx = std::min(x - D / 2 + D, cols - 1);
            int ty = std::min(y - D / 2 + D, rows - 1);
            float sum1 = 0, sum2 = 0;
            for (int cy = std::max(y - D / 2, 0); cy < ty; ++cy)
                for (int cx = std::max(x - D / 2, 0); cx < tx; ++cx) {
                    int tmp = row_ptr(src, sstep, cy)[cx];
                    float space2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
                    float color2 = (value - tmp) * (value - tmp);
                    float w = expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
                    sum1 += tmp * w;
                    sum2 += w;
                }
            int round = f2i_rn(sum1 / sum2);
            if (round > 5000 || round < 200) round = 0;
            round = std::max(0, std::min(round, 32767));
            *((C *)((char *)dst + (size_t)y * dstep) + x) = C(float(round), 0);
        }
}

// ---- Map.cu:202-230 pyrDownKernel -----------------------------------------
template <class C>
void pyr_down(const float *src, size_t sstep, int srows, int scols, float *dst, size_t dstep) {
    const float sigma_color = 30;
    int drows = srows / 2, dcols = scols / 2;
#pragma omp parallel for
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int D = 5;
            auto S = [&](int r, int c) { return ((const C *)((const char *)src + (size_t)r * sstep) + c)->real(); };
            int center = f2i_rn(S(2 * y, 2 * x));
            int tx = std::min(2 * x - D / 2 + D, scols - 1);
            int ty = std::min(2 * y - D / 2 + D, srows - 1);
            int cy = std::max(0, 2 * y - D / 2);
            int sum = 0, count = 0;
            for (; cy < ty; ++cy)
                for (int cx = std::max(0, 2 * x - D / 2); cx < tx; ++cx) {
                    int val = f2i_rn(S(cy, cx));
                    if (std::abs(val - center) < 3 * sigma_color) { sum += val; ++count; }
                }
            float r = float(sum / count);
            *((C *)((char *)dst + (size_t)y * dstep) + x) = C(r, 0);
        }
}

// ---- Map.cu:8-29 computeVmapKernel ----------------------------------------
template <class C>
void create_vmap(Intr intr, const float *depth, size_t dstep, int rows, int cols, float *vmap, size_t mstep) {
    float fx_inv = 1.f / intr.fx, fy_inv = 1.f / intr.fy, cx = intr.cx, cy = intr.cy;
    auto M = [&](int r, int c) { return (C *)((char *)vmap + (size_t)r * mstep) + c; };
#pragma omp parallel for
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            C z = *((const C *)((const char *)depth + (size_t)v * dstep) + u);
            z /= 1000.f;
            if (z.real() != 0) {
                C vx = z * (float(u) - cx) * fx_inv;
                C vy = z * (float(v) - cy) * fy_inv;
                C vz = z;
                *M(v, u) = C(vx.real(), vx.imag());
                *M(v + rows, u) = C(vy.real(), vy.imag());
                *M(v + rows * 2, u) = C(vz.real(), vz.imag());
            } else
                *M(v, u) = C(qnan_f(), 0);
        }
}

// ---- Map.cu:32-70 computeNmapKernel ---------------------------------------
template <class C>
void create_nmap(int rows, int cols, const float *vmap, float *nmap, size_t mstep) {
    auto V = [&](int r, int c) { return *((const C *)((const char *)vmap + (size_t)r * mstep) + c); };
    auto N = [&](int r, int c) { return (C *)((char *)nmap + (size_t)r * mstep) + c; };
#pragma omp parallel for
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            if (u == cols - 1 || v == rows - 1) { *N(v, u) = C(qnan_f()); continue; }
            vec3<C> v00, v01, v10;
            v00.x = V(v, u); v01.x = V(v, u + 1); v10.x = V(v + 1, u);
            if (!std::isnan(v00.x.real()) && !std::isnan(v01.x.real()) && !std::isnan(v10.x.real())) {
                v00.y = V(v + rows, u); v01.y = V(v + rows, u + 1); v10.y = V(v + 1 + rows, u);
                v00.z = V(v + 2 * rows, u); v01.z = V(v + 2 * rows, u + 1); v10.z = V(v + 1 + 2 * rows, u);
                vec3<C> r = normalized3(cross3(v01 - v00, v10 - v00));
                *N(v, u) = r.x; *N(v + rows, u) = r.y; *N(v + 2 * rows, u) = r.z;
            } else
                *N(v, u) = C(qnan_f());
        }
}

// ---- Map.cu:105-152 resizeMapKernel<normalize> ----------------------------
template <class C>
void resize_map(bool normalize, int srows, int scols, const float *in, size_t istep, float *out, size_t ostep) {
    int drows = srows / 2, dcols = scols / 2;
    auto I = [&](int r, int c) { return *((const C *)((const char *)in + (size_t)r * istep) + c); };
    auto O = [&](int r, int c) { return (C *)((char *)out + (size_t)r * ostep) + c; };
#pragma omp parallel for
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            int xs = x * 2, ys = y * 2;
            C x00 = I(ys, xs), x01 = I(ys, xs + 1), x10 = I(ys + 1, xs), x11 = I(ys + 1, xs + 1);
            if (std::isnan(x00.real()) || std::isnan(x01.real()) || std::isnan(x10.real()) || std::isnan(x11.real())) {
                *O(y, x) = C(qnan_f());
                continue;
            }
            vec3<C> n;
            n.x = (x00 + x01 + x10 + x11) / 4.0f;
            C y00 = I(ys + srows, xs), y01 = I(ys + srows, xs + 1), y10 = I(ys + srows + 1, xs), y11 = I(ys + srows + 1, xs + 1);
            n.y = (y00 + y01 + y10 + y11) / 4.0f;
            C z00 = I(ys + 2 * srows, xs), z01 = I(ys + 2 * srows, xs + 1), z10 = I(ys + 2 * srows + 1, xs), z11 = I(ys + 2 * srows + 1, xs + 1);
            n.z = (z00 + z01 + z10 + z11) / 4.0f;
            if (normalize) n = normalized3(n);
            *O(y, x) = n.x; *O(y + drows, x) = n.y; *O(y + 2 * drows, x) = n.z;
        }
}

// ---- ICP.cu:166-281 Combined + :120-161 TranformReduction + :419-428 ------
// Products are formed in C (complex float) and accumulated in double
// (ICP.cu:273-274).  The GPU sums per block in an LDS tree and then across
// blocks; here each image row is summed left to right and the rows in order —
// a different association of the same double additions (differences ~1e-16).
// [y0, y1) restricts the pixel rows (multi-GPU tests shard pixels by rows).
template <class C>
long long icp_combined(const mat33<C> &Rcurr, const vec3<C> &tcurr, const float *vmap_curr, const float *nmap_curr,
                       const mat33<C> &Rprev_inv, const vec3<C> &tprev, Intr intr, const float *vmap_g_prev,
                       const float *nmap_g_prev, size_t mstep, int rows, int cols, float distThres, float angleThres,
                       int y0, int y1, double sums[54]) {
    auto M = [&](const float *m, int r, int c) { return *((const C *)((const char *)m + (size_t)r * mstep) + c); };
    std::vector<double> rowsum((size_t)rows * 54, 0.0);
    std::vector<long long> rowcnt(rows, 0);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; ++y) {
        double *acc = &rowsum[(size_t)y * 54];
        for (int x = 0; x < cols; ++x) {
            // search_newton, ICP.cu:196-244
            vec3<C> ncurr;
            ncurr.x = M(nmap_curr, y, x);
            if (std::isnan(ncurr.x.real())) continue;
            ncurr.y = M(nmap_curr, y + rows, x);
            ncurr.z = M(nmap_curr, y + 2 * rows, x);
            vec3<C> vcurr;
            vcurr.x = M(vmap_curr, y, x); vcurr.y = M(vmap_curr, y + rows, x); vcurr.z = M(vmap_curr, y + 2 * rows, x);
            vec3<C> vcurr_g = Rcurr * vcurr + tcurr;
            vec3<C> vcp = Rprev_inv * (vcurr_g - tprev);
            float cpx = vcp.x.real(), cpy = vcp.y.real(), cpz = vcp.z.real();
            int ux = f2i_rn(cpx * intr.fx / cpz + intr.cx);
            int uy = f2i_rn(cpy * intr.fy / cpz + intr.cy);
            if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || cpz < 0) continue;
            vec3<C> nprev_g;
            nprev_g.x = M(nmap_g_prev, uy, ux);
            if (std::isnan(nprev_g.x.real())) continue;
            nprev_g.y = M(nmap_g_prev, uy + rows, ux);
            nprev_g.z = M(nmap_g_prev, uy + 2 * rows, ux);
            vec3<C> vprev_g;
            vprev_g.x = M(vmap_g_prev, uy, ux); vprev_g.y = M(vmap_g_prev, uy + rows, ux); vprev_g.z = M(vmap_g_prev, uy + 2 * rows, ux);
            C dist = norm3(vprev_g - vcurr_g);
            if (dist.real() > distThres) continue;
            vec3<C> ncurr_g = Rcurr * ncurr;
            C sine = norm3(cross3(ncurr_g, nprev_g));
            if (sine.real() >= angleThres) continue;
            // operator(), ICP.cu:254-280: row = [s x n, n], b = n.(d - s)
            const vec3<C> &n = nprev_g, &d = vprev_g, &s = vcurr_g;
            C row[7];
            vec3<C> cr = cross3(s, n);
            row[0] = cr.x; row[1] = cr.y; row[2] = cr.z;
            row[3] = n.x; row[4] = n.y; row[5] = n.z;
            row[6] = dot(n, d - s);
            int shift = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 7; ++j) {
                    C p = row[i] * row[j];
                    acc[2 * shift] += (double)p.real();
                    acc[2 * shift + 1] += (double)p.imag();
                    ++shift;
                }
            rowcnt[y]++;
        }
    }
    long long inliers = 0;
    for (int k = 0; k < 54; ++k) sums[k] = 0.0;
    for (int y = y0; y < y1; ++y) {
        for (int k = 0; k < 54; ++k) sums[k] += rowsum[(size_t)y * 54 + k];
        inliers += rowcnt[y];
    }
    return inliers;
}

// ICP.cu:419-428: unpack 27 complex sums into symmetric A (6x6) and b (6)
inline void icp_unpack(const double sums[54], double A[72], double b[12]) {
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            double re = sums[2 * shift], im = sums[2 * shift + 1];
            ++shift;
            if (j == 6) { b[2 * i] = re; b[2 * i + 1] = im; }
            else {
                A[2 * (j * 6 + i)] = re; A[2 * (j * 6 + i) + 1] = im;
                A[2 * (i * 6 + j)] = re; A[2 * (i * 6 + j) + 1] = im;
            }
        }
}

// ---- TsdfFusion.cu:204-283 ComputeLocalTsdfHessianKernel + :286-331 -------
// DC = dcplx<C>.  gt is dense and unpitched (flat index, :220).  Per-voxel
// outputs are optional (null to skip); out4 = {sum loss, sum grad, sum
// hessian, count}.  thrust::reduce sums floats in an unspecified tree order;
// here the per-voxel floats are summed in double and rounded once.
template <class C>
void tsdf_hessian(const float *depthScaled, size_t dstep, int drows, int dcols, const int res[3], float voxel_size,
                  const float *Rv2c144, const float *tv2c48, float tranc_dist, Intr intr, const float *gt,
                  float *real_out, float *grad_out, float *hess_out, int *count_out, int z0, int z1, double out4[4]) {
    typedef dcplx<C> DC;
    auto LD = [](const float *p) { return DC(p[0], p[1], p[2], p[3]); };
    DC R[3][3], t[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r][c] = LD(Rv2c144 + (r * 3 + c) * 4);
        t[r] = LD(tv2c48 + r * 4);
    }
    float tranc_dist_inv = 1.0f / tranc_dist;
    double sl = 0, sg = 0, sh = 0; long long sc = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : sl, sg, sh, sc)
    for (int y = 0; y < res[1]; ++y)
        for (int x = 0; x < res[0]; ++x)
            for (int z = z0; z < z1; ++z) {
                size_t index = (size_t)z * res[1] * res[0] + (size_t)y * res[0] + x;
                DC gt_tsdf(gt[index], 0, 0, 0);
                if (gt_tsdf.value() == 0 || std::fabs(gt_tsdf.value()) > 0.95) continue;
                DC vgx((float(x) + 0.5f) * voxel_size, 0, 0, 0);
                DC vgy((float(y) + 0.5f) * voxel_size, 0, 0, 0);
                DC vgz((float(z) + 0.5f) * voxel_size, 0, 0, 0);
                DC vc[3];
                for (int r = 0; r < 3; ++r) vc[r] = (R[r][0] * vgx + R[r][1] * vgy + R[r][2] * vgz) + t[r];
                DC inv_z = DC(1.0f) / vc[2];
                if (inv_z.value() < 0) continue;
                DC image_x = vc[0] * inv_z * intr.fx + intr.cx;
                DC image_y = vc[1] * inv_z * intr.fy + intr.cy;
                int coo_x = f2i_rd(image_x.value() - 0.5f), coo_y = f2i_rd(image_y.value() - 0.5f);
                if (!(coo_x > 1 && coo_y > 1 && coo_x < dcols - 1 && coo_y < drows - 1)) continue;
                int near_x = f2i_rn(image_x.value()), near_y = f2i_rn(image_y.value());
                DC Dp;
                DC Dp_near(row_ptr(depthScaled, dstep, near_y)[near_x]);
                DC d00(row_ptr(depthScaled, dstep, coo_y)[coo_x]);
                DC d10(row_ptr(depthScaled, dstep, coo_y)[coo_x + 1]);
                DC d01(row_ptr(depthScaled, dstep, coo_y + 1)[coo_x]);
                DC d11(row_ptr(depthScaled, dstep, coo_y + 1)[coo_x + 1]);
                if (d00.value() != 0.0f && d01.value() != 0.0f && d10.value() != 0.0f && d11.value() != 0.0f) {
                    DC one(1.0f, 0.0f, 0.0f, 0.0f);
                    DC a = image_x - DC(float(coo_x) + 0.5f, 0, 0, 0);
                    DC b = image_y - DC(float(coo_y) + 0.5f, 0, 0, 0);
                    Dp = d00 * (one - a) * (one - b) + d10 * a * (one - b) + d01 * (one - a) * b + d11 * a * b;
                } else
                    Dp = Dp_near;
                if (Dp.value() > 5 || Dp.value() < 0.2) continue;
                DC xl = (image_x - intr.cx) / intr.fx;
                DC yl = (image_y - intr.cy) / intr.fy;
                DC v1x = Dp * xl, v1y = Dp * yl, v1z = Dp;
                DC n1 = dsqrt(v1x * v1x + v1y * v1y + v1z * v1z);
                DC n0 = dsqrt(vc[0] * vc[0] + vc[1] * vc[1] + vc[2] * vc[2]);
                DC distance = n1 - n0;
                DC gt_distance = gt_tsdf * tranc_dist;
                DC error = (distance - gt_distance) * tranc_dist_inv;
                if (std::fabs(error.value()) > 1) continue;
                DC loss = error * error;
                if (real_out) real_out[index] = loss.value();
                if (grad_out) grad_out[index] = loss.grad();
                if (hess_out) hess_out[index] = loss.hessian();
                if (count_out) count_out[index] = 1;
                sl += loss.value(); sg += loss.grad(); sh += loss.hessian(); sc += 1;
            }
    out4[0] = sl; out4[1] = sg; out4[2] = sh; out4[3] = (double)sc;
}

// ---- Gauss-Newton terms of the same residual in first-order CSFD (BASELINE config 5) -----------
// The residual of the kernel above in complex<float> for six poses (pose k seeded with i*h on degree of
// freedom k): out29 = sum d_j d_k (j <= k, 21), sum d_k r (6), sum r^2, count, with d_k = Im(error_k),
// r = Re(error_0), over the voxels every seeded evaluation keeps.  gt indexed from plane z0 (slab-relative).
template <class C>
bool tsdf_error_c(const float *depthScaled, size_t dstep, int drows, int dcols, float tranc_dist, float tranc_dist_inv, Intr intr,
                  const mat33<C> &R, const vec3<C> &t, float vgx, float vgy, float vgz, float gt, C &error) {
    vec3<C> v_g = mk3<C>(C(vgx), C(vgy), C(vgz));
    vec3<C> v_c = mk3<C>(dot(R.data[0], v_g) + t.x, dot(R.data[1], v_g) + t.y, dot(R.data[2], v_g) + t.z);
    C inv_z = C(1.0f) / v_c.z;
    if (inv_z.real() < 0) return false;
    C image_x = v_c.x * inv_z * intr.fx + intr.cx;
    C image_y = v_c.y * inv_z * intr.fy + intr.cy;
    int coo_x = f2i_rd(image_x.real() - 0.5f), coo_y = f2i_rd(image_y.real() - 0.5f);
    if (!(coo_x > 1 && coo_y > 1 && coo_x < dcols - 1 && coo_y < drows - 1)) return false;
    int near_x = f2i_rn(image_x.real()), near_y = f2i_rn(image_y.real());
    C Dp(row_ptr(depthScaled, dstep, near_y)[near_x]);
    float d00 = row_ptr(depthScaled, dstep, coo_y)[coo_x], d10 = row_ptr(depthScaled, dstep, coo_y)[coo_x + 1];
    float d01 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x], d11 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x + 1];
    if (d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
        C one(1.0f);
        C fa = image_x - C(float(coo_x) + 0.5f);
        C fb = image_y - C(float(coo_y) + 0.5f);
        Dp = d00 * (one - fa) * (one - fb) + d10 * fa * (one - fb) + d01 * (one - fa) * fb + d11 * fa * fb;
    }
    if (Dp.real() > 5 || Dp.real() < 0.2) return false;
    C xl = (image_x - intr.cx) / intr.fx;
    C yl = (image_y - intr.cy) / intr.fy;
    vec3<C> v_c_1 = mk3<C>(Dp * xl, Dp * yl, Dp);
    C distance = norm3(v_c_1) - norm3(v_c);
    C gt_distance = C(gt) * tranc_dist;
    error = (distance - gt_distance) * tranc_dist_inv;
    return !(std::fabs(error.real()) > 1);
}
template <class C>
void tsdf_gn_terms(const float *depthScaled, size_t dstep, int drows, int dcols, const int res[3], float voxel_size, const float *Rv2c108,
                   const float *tv2c36, float tranc_dist, Intr intr, const float *gt, int z0, int z1, double out29[29]) {
    mat33<C> R[6]; vec3<C> t[6];
    for (int k = 0; k < 6; ++k) { R[k] = load_mat33<C>(Rv2c108 + 18 * k); t[k] = load_vec3<C>(tv2c36 + 6 * k); }
    const float tranc_dist_inv = 1.0f / tranc_dist;
    double acc[29];
    for (int k = 0; k < 29; ++k) acc[k] = 0.0;
#pragma omp parallel
    {
        double loc[29];
        for (int k = 0; k < 29; ++k) loc[k] = 0.0;
#pragma omp for schedule(dynamic, 4) nowait
        for (int y = 0; y < res[1]; ++y)
            for (int x = 0; x < res[0]; ++x)
                for (int z = z0; z < z1; ++z) {
                    const size_t index = (size_t)(z - z0) * res[1] * res[0] + (size_t)y * res[0] + x;
                    const float g = gt[index];
                    if (g == 0 || std::fabs(g) > 0.95) continue;
                    const float vgx = (float(x) + 0.5f) * voxel_size, vgy = (float(y) + 0.5f) * voxel_size, vgz = (float(z) + 0.5f) * voxel_size;
                    C e[6];
                    bool ok = true;
                    for (int k = 0; k < 6; ++k)
                        ok = ok && tsdf_error_c<C>(depthScaled, dstep, drows, dcols, tranc_dist, tranc_dist_inv, intr, R[k], t[k], vgx, vgy, vgz, g, e[k]);
                    if (!ok) continue;
                    const double r = (double)e[0].real();
                    int s = 0;
                    for (int j = 0; j < 6; ++j)
                        for (int k = j; k < 6; ++k) loc[s++] += (double)e[j].imag() * (double)e[k].imag();
                    for (int k = 0; k < 6; ++k) loc[21 + k] += (double)e[k].imag() * r;
                    loc[27] += r * r;
                    loc[28] += 1.0;
                }
#pragma omp critical
        for (int k = 0; k < 29; ++k) acc[k] += loc[k];
    }
    for (int k = 0; k < 29; ++k) out29[k] = acc[k];
}

// ---- TsdfFusion.cu:335-410 ComputeLocalTsdfLossKernel + :412-447 ----------
inline void tsdf_loss(const float *depthScaled, size_t dstep, int drows, int dcols, const int res[3], float voxel_size,
                      const float *Rv2c9, const float *tv2c3, float tranc_dist, Intr intr, const float *gt,
                      float *real_out, int *count_out, int z0, int z1, double out2[2]) {
    float tranc_dist_inv = 1.0f / tranc_dist;
    double sl = 0; long long sc = 0;
    auto dot3 = [](const float *r, float a, float b, float c) { return r[0] * a + r[1] * b + r[2] * c; };
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : sl, sc)
    for (int y = 0; y < res[1]; ++y)
        for (int x = 0; x < res[0]; ++x)
            for (int z = z0; z < z1; ++z) {
                size_t index = (size_t)z * res[1] * res[0] + (size_t)y * res[0] + x;
                float gt_tsdf = gt[index];
                if (gt_tsdf == 0 || std::fabs(gt_tsdf) > 0.95) continue;
                float vgx = (float(x) + 0.5f) * voxel_size, vgy = (float(y) + 0.5f) * voxel_size, vgz = (float(z) + 0.5f) * voxel_size;
                float vcx = dot3(Rv2c9 + 0, vgx, vgy, vgz) + tv2c3[0];
                float vcy = dot3(Rv2c9 + 3, vgx, vgy, vgz) + tv2c3[1];
                float vcz = dot3(Rv2c9 + 6, vgx, vgy, vgz) + tv2c3[2];
                float inv_z = 1.0f / vcz;
                if (inv_z < 0) continue;
                float image_x = vcx * inv_z * intr.fx + intr.cx;
                float image_y = vcy * inv_z * intr.fy + intr.cy;
                int coo_x = f2i_rd(image_x - 0.5f), coo_y = f2i_rd(image_y - 0.5f);
                if (!(coo_x > 1 && coo_y > 1 && coo_x < dcols - 1 && coo_y < drows - 1)) continue;
                int near_x = f2i_rn(image_x), near_y = f2i_rn(image_y);
                float Dp;
                float Dp_near = row_ptr(depthScaled, dstep, near_y)[near_x];
                float d00 = row_ptr(depthScaled, dstep, coo_y)[coo_x];
                float d10 = row_ptr(depthScaled, dstep, coo_y)[coo_x + 1];
                float d01 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x];
                float d11 = row_ptr(depthScaled, dstep, coo_y + 1)[coo_x + 1];
                if (d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
                    float one = 1.0f;
                    float a = image_x - (float(coo_x) + 0.5f);
                    float b = image_y - (float(coo_y) + 0.5f);
                    Dp = d00 * (one - a) * (one - b) + d10 * a * (one - b) + d01 * (one - a) * b + d11 * a * b;
                } else
                    Dp = Dp_near;
                if (Dp > 5 || Dp < 0.2) continue;
                float xl = (image_x - intr.cx) / intr.fx;
                float yl = (image_y - intr.cy) / intr.fy;
                float v1x = Dp * xl, v1y = Dp * yl, v1z = Dp;
                float distance = std::sqrt(v1x * v1x + v1y * v1y + v1z * v1z) - std::sqrt(vcx * vcx + vcy * vcy + vcz * vcz);
                float gt_distance = gt_tsdf * tranc_dist;
                float error = (distance - gt_distance) * tranc_dist_inv;
                if (std::fabs(error) > 1) continue;
                float loss = error * error;
                if (real_out) real_out[index] = loss;
                if (count_out) count_out[index] = 1;
                sl += loss; sc += 1;
            }
    out2[0] = sl; out2[1] = (double)sc;
}

// ---- ExtractPointCloud.cu:25-185 Scanner (points), :214-341 ExtractNormals -----------------
// Real-valued export of the zero level set.  Points are appended in (z, y, x, direction) order; the
// reference's order depends on atomics, so comparisons sort both sides.  Returns the number found;
// at most `capacity` are stored.
inline size_t extract_points(const float *value, size_t vstep, int X, int Y, int Z, float voxel_size, int zs0, int z0, int z1, float *out,
                             size_t capacity) {
    auto F_ = [&](int x, int y, int z) { return row_ptr(value, vstep, Y * (z - zs0) + y)[x]; };
    size_t n = 0;
    auto put = [&](float px, float py, float pz) {
        if (n < capacity) { out[3 * n] = px; out[3 * n + 1] = py; out[3 * n + 2] = pz; }
        ++n;
    };
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < Y - 1; ++y)
            for (int x = 0; x < X - 1; ++x) {
                const float F = F_(x, y, z);
                if (!(F < 0.99f)) continue;
                const float Vx = (x + 0.5f) * voxel_size, Vy = (y + 0.5f) * voxel_size, Vz = (z + 0.5f) * voxel_size;
                float Fn = F_(x + 1, y, z);
                if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) put(Vx - (F / (Fn - F)) * voxel_size, Vy, Vz);
                Fn = F_(x, y + 1, z);
                if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) put(Vx, Vy - (F / (Fn - F)) * voxel_size, Vz);
                Fn = F_(x, y, z + 1);
                if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) put(Vx, Vy, Vz - (F / (Fn - F)) * voxel_size);
            }
    return n;
}
inline void extract_normals(const float *value, size_t vstep, int X, int Y, int Z, float vs, const float *points, size_t n, float *normals) {
    auto rd = [&](int x, int y, int z) { return row_ptr(value, vstep, Y * z + y)[x]; };
    auto interp = [&](float px, float py, float pz) {
        int gx = (int)std::floor(px / vs), gy = (int)std::floor(py / vs), gz = (int)std::floor(pz / vs);
        const float vx = (gx + 0.5f) * vs, vy = (gy + 0.5f) * vs, vz = (gz + 0.5f) * vs;
        gx = (px < vx) ? (gx - 1) : gx; gy = (py < vy) ? (gy - 1) : gy; gz = (pz < vz) ? (gz - 1) : gz;
        const float a = (px - (gx + 0.5f) * vs) / vs, b = (py - (gy + 0.5f) * vs) / vs, c = (pz - (gz + 0.5f) * vs) / vs;
        return rd(gx + 0, gy + 0, gz + 0) * (1 - a) * (1 - b) * (1 - c) + rd(gx + 0, gy + 0, gz + 1) * (1 - a) * (1 - b) * c +
               rd(gx + 0, gy + 1, gz + 0) * (1 - a) * b * (1 - c) + rd(gx + 0, gy + 1, gz + 1) * (1 - a) * b * c +
               rd(gx + 1, gy + 0, gz + 0) * a * (1 - b) * (1 - c) + rd(gx + 1, gy + 0, gz + 1) * a * (1 - b) * c +
               rd(gx + 1, gy + 1, gz + 0) * a * b * (1 - c) + rd(gx + 1, gy + 1, gz + 1) * a * b * c;
    };
#pragma omp parallel for
    for (long long i = 0; i < (long long)n; ++i) {
        const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        float nx = 0.f, ny = 0.f, nz = 0.f;
        const int gx = (int)std::floor(px / vs), gy = (int)std::floor(py / vs), gz = (int)std::floor(pz / vs);
        if (gx > 1 && gy > 1 && gz > 1 && gx < X - 2 && gy < Y - 2 && gz < Z - 2) {
            nx = interp(px + vs, py, pz) - interp(px - vs, py, pz);
            ny = interp(px, py + vs, pz) - interp(px, py - vs, pz);
            nz = interp(px, py, pz + vs) - interp(px, py, pz - vs);
            const float norm = nx * nx + ny * ny + nz * nz;
            nx = nx / norm; ny = ny / norm; nz = nz / norm;
        }
        normals[3 * i] = nx; normals[3 * i + 1] = ny; normals[3 * i + 2] = nz;
    }
}

}  // namespace oc
