// xs_raycast.hip — volume raycast for gfx950.  Replaces XKinectFusion/src/RayCaster.cu:26-141
// (RayCaster fields, get_ray_next, checkInds, readTsdf, getVoxel, interpolateTrilineary),
// :197-310 (operator()), :324 (rayCastKernel), :327-368 (raycast launcher).
//
// One ray per lane; a wave covers an 8x8 pixel tile and a workgroup a 16x16 tile, so the
// voxels a wave touches while marching stay within a few cache lines of each other (the march
// is a chain of dependent gathers: L2/latency-bound, not HBM-bound).  Only the real part of
// the position feeds the fixed-step march (RayCaster.cu:236-247), so the march reads the
// value volume alone and in real arithmetic — the same float operations the reference's
// real(ray_start + ray_dir * t) performs; the gradient volume is read only at the crossing,
// where the 2 + 6 trilinear samples run in complex arithmetic.
#include "xs_device.h"
#include "../../include/xslam_amd.h"

using namespace xs;

struct RaycastArgs {
    MatS33 Rc2v; cfloat3 tc2v; MatS33 Rv2w; cfloat3 tv2w;
    int X, Y, Z;
    float voxel_size, time_step;
    int cols, rows;
    const float *value; const float *grad; size_t vstep;
    Intr intr;
    cfloat *vmap; cfloat *nmap; size_t mstep;
    int z0, z1;  // z-slab resident behind value/grad (whole volume: 0, Z)
    unsigned long long *hits;
};

namespace {
__device__ __forceinline__ int sgn(float v) { return (0.0f < v) - (v < 0.0f); }

struct Vol {
    const float *value; const float *grad; size_t vstep; int X, Y, Z; float vs;
    __device__ __forceinline__ float read_value(int x, int y, int z) const {
        return row_ptr(value, vstep, Y * z + y)[x] + 1e-5f;  // RayCaster.cu:76
    }
    __device__ __forceinline__ cfloat read(int x, int y, int z) const {  // readTsdf, :69-78
        x = x % X; y = y % Y; z = z % Z;
        cfloat r(row_ptr(value, vstep, Y * z + y)[x], row_ptr(grad, vstep, Y * z + y)[x]);
        r += 1e-5f;
        return r;
    }
    __device__ __forceinline__ cfloat interp(const cfloat3 &p) const {  // :99-141
        int gx = __float2int_rd(p.x.re / vs), gy = __float2int_rd(p.y.re / vs), gz = __float2int_rd(p.z.re / vs);
        const float qn = qnan_f();
        if (gx <= 0 || gx >= X - 1) return cfloat(qn, 0.f);
        if (gy <= 0 || gy >= Y - 1) return cfloat(qn, 0.f);
        if (gz <= 0 || gz >= Z - 1) return cfloat(qn, 0.f);
        const float vx = (gx + 0.5f) * vs, vy = (gy + 0.5f) * vs, vz = (gz + 0.5f) * vs;
        gx += -(sgn(vx - p.x.re) + 1) >> 1;
        gy += -(sgn(vy - p.y.re) + 1) >> 1;
        gz += -(sgn(vz - p.z.re) + 1) >> 1;
        const cfloat a0 = (p.x - (gx + 0.5f) * vs) / vs;
        const cfloat b0 = (p.y - (gy + 0.5f) * vs) / vs;
        const cfloat c0 = (p.z - (gz + 0.5f) * vs) / vs;
        const cfloat one(1.0f, 0.0f);
        const cfloat a1 = one - a0, b1 = one - b0, c1 = one - c0;
        return read(gx + 0, gy + 0, gz + 0) * a1 * b1 * c1 + read(gx + 0, gy + 0, gz + 1) * a1 * b1 * c0 +
               read(gx + 0, gy + 1, gz + 0) * a1 * b0 * c1 + read(gx + 0, gy + 1, gz + 1) * a1 * b0 * c0 +
               read(gx + 1, gy + 0, gz + 0) * a0 * b1 * c1 + read(gx + 1, gy + 0, gz + 1) * a0 * b1 * c0 +
               read(gx + 1, gy + 1, gz + 0) * a0 * b0 * c1 + read(gx + 1, gy + 1, gz + 1) * a0 * b0 * c0;
    }
};
}  // namespace

__global__ void __launch_bounds__(256) k_raycast(const RaycastArgs a) {
    // lane -> pixel inside an 8x8 tile; 4 waves -> 16x16 tile per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    unsigned hit = 0;
    if (x < a.cols && y < a.rows) {
        row_ptr(a.vmap, a.mstep, y)[x] = cfloat(qnan_f(), 0.f);
        row_ptr(a.nmap, a.mstep, y)[x] = cfloat(qnan_f(), 0.f);
        Vol vol{a.value, a.grad, a.vstep, a.X, a.Y, a.Z, a.voxel_size};
        const cfloat3 ray_start = a.tc2v;
        cfloat3 rn;
        rn.x = cfloat((x - a.intr.cx) / a.intr.fx);
        rn.y = cfloat((y - a.intr.cy) / a.intr.fy);
        rn.z = cfloat(1.f);
        const cfloat3 ray_next = a.Rc2v * rn + a.tc2v;
        cfloat3 ray_dir = normalized(ray_next - ray_start);
        ray_dir.x = (ray_dir.x == 0.f) ? cfloat(1e-15f) : ray_dir.x;
        ray_dir.y = (ray_dir.y == 0.f) ? cfloat(1e-15f) : ray_dir.y;
        ray_dir.z = (ray_dir.z == 0.f) ? cfloat(1e-15f) : ray_dir.z;
        const float sx = ray_start.x.re, sy = ray_start.y.re, sz = ray_start.z.re;
        const float dx = ray_dir.x.re, dy = ray_dir.y.re, dz = ray_dir.z.re;
        const float vs = a.voxel_size, time_step = a.time_step;
        float time_curr = 0.2f;
        const float max_time = 5.0f;
        int gx = __float2int_rd((sx + dx * time_curr) / vs);
        int gy = __float2int_rd((sy + dy * time_curr) / vs);
        int gz = __float2int_rd((sz + dz * time_curr) / vs);
        gx = max(0, min(gx, a.X - 1)); gy = max(0, min(gy, a.Y - 1)); gz = max(0, min(gz, a.Z - 1));
        float tsdf = vol.read_value(gx, gy, gz);
        for (; time_curr < max_time; time_curr += time_step) {
            const float tsdf_prev = tsdf;
            const float tn = time_curr + time_step;
            gx = __float2int_rd((sx + dx * tn) / vs);
            gy = __float2int_rd((sy + dy * tn) / vs);
            gz = __float2int_rd((sz + dz * tn) / vs);
            if (!(gx >= 0 && gy >= 0 && gz >= 0 && gx < a.X && gy < a.Y && gz < a.Z)) break;
            tsdf = vol.read_value(gx, gy, gz);
            if (tsdf_prev < 0.f && tsdf > 0.f) break;
            if (tsdf_prev > 0.f && tsdf < 0.f) {  // zero crossing
                const cfloat Ftdt = vol.interp(ray_start + ray_dir * tn);
                if (isnan(Ftdt.re)) break;
                const cfloat Ft = vol.interp(ray_start + ray_dir * time_curr);
                if (isnan(Ft.re)) break;
                const cfloat coef = Ft / (Ftdt - Ft);
                if (Ft.re < 0.0f || Ftdt.re > 0.0f) break;
                const cfloat Ts = time_curr - time_step * coef;
                const cfloat3 vertex_found = ray_start + ray_dir * Ts;
                const cfloat3 vw = a.Rv2w * vertex_found + a.tv2w;
                row_ptr(a.vmap, a.mstep, y)[x] = vw.x;
                row_ptr(a.vmap, a.mstep, y + a.rows)[x] = vw.y;
                row_ptr(a.vmap, a.mstep, y + 2 * a.rows)[x] = vw.z;
                hit = 1;
                gx = __float2int_rd(vertex_found.x.re / vs);
                gy = __float2int_rd(vertex_found.y.re / vs);
                gz = __float2int_rd(vertex_found.z.re / vs);
                if (gx > 1 && gy > 1 && gz > 1 && gx < a.X - 2 && gy < a.Y - 2 && gz < a.Z - 2) {
                    cfloat3 t, n;
                    const float half = vs * 0.5f;
                    t = vertex_found; t.x += half; const cfloat Fx1 = vol.interp(t);
                    t = vertex_found; t.x -= half; const cfloat Fx2 = vol.interp(t);
                    n.x = Fx1 - Fx2;
                    t = vertex_found; t.y += half; const cfloat Fy1 = vol.interp(t);
                    t = vertex_found; t.y -= half; const cfloat Fy2 = vol.interp(t);
                    n.y = Fy1 - Fy2;
                    t = vertex_found; t.z += half; const cfloat Fz1 = vol.interp(t);
                    t = vertex_found; t.z -= half; const cfloat Fz2 = vol.interp(t);
                    n.z = Fz1 - Fz2;
                    if (squarednorm(n).re == 0) break;
                    const cfloat3 n_g = a.Rv2w * normalized(n);
                    row_ptr(a.nmap, a.mstep, y)[x] = n_g.x;
                    row_ptr(a.nmap, a.mstep, y + a.rows)[x] = n_g.y;
                    row_ptr(a.nmap, a.mstep, y + 2 * a.rows)[x] = n_g.z;
                }
                break;
            }
        }
    }
    if (a.hits) {
        unsigned s = wave_sum_u32(hit);
        if (lane == 0 && s) atomicAdd(a.hits, (unsigned long long)s);
    }
}

static void ld_mat(const float *p, MatS33 &m) {
    for (int r = 0; r < 3; ++r) {
        m.data[r].x = cfloat(p[r * 6 + 0], p[r * 6 + 1]);
        m.data[r].y = cfloat(p[r * 6 + 2], p[r * 6 + 3]);
        m.data[r].z = cfloat(p[r * 6 + 4], p[r * 6 + 5]);
    }
}
static void ld_vec(const float *p, cfloat3 &v) { v.x = cfloat(p[0], p[1]); v.y = cfloat(p[2], p[3]); v.z = cfloat(p[4], p[5]); }

/* raycast(const Intr&, const MatS33& Rc2v, const devComplex3& tc2v, const MatS33& Rv2w,
 *         const devComplex3& tv2w, float tranc_dist, const int3& res, float voxel_size,
 *         const PtrStep<float>& value, const PtrStep<float>& grad, MapArr& vmap, MapArr& nmap)
 *                                                   RayCaster.h:21-25, RayCaster.cu:327-368
 * rows/cols: size of one map plane.  hits_dev: optional device counter of pixels that got a
 * vertex.  No synchronisation (neither does the reference, :367). */
extern "C" int xs_raycast(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                          float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                          float *vmap, float *nmap, size_t map_step, int rows, int cols, unsigned long long *hits_dev, void *stream) {
    if (!intr4 || !Rc2v18 || !tc2v6 || !Rv2w18 || !tv2w6 || !res || !value || !grad || !vmap || !nmap)
        return xs_set_error(hipErrorInvalidValue, "xs_raycast: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    RaycastArgs a;
    ld_mat(Rc2v18, a.Rc2v); ld_vec(tc2v6, a.tc2v); ld_mat(Rv2w18, a.Rv2w); ld_vec(tv2w6, a.tv2w);
    a.X = res[0]; a.Y = res[1]; a.Z = res[2];
    a.voxel_size = voxel_size;
    a.time_step = tranc_dist * 0.8f;  // RayCaster.cu:350
    a.cols = cols; a.rows = rows;
    a.value = value; a.grad = grad; a.vstep = vol_step;
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.vmap = (cfloat *)vmap; a.nmap = (cfloat *)nmap; a.mstep = map_step;
    a.z0 = 0; a.z1 = res[2];
    a.hits = hits_dev;
    dim3 block(256), grid(div_up(cols, 16), div_up(rows, 16));
    hipLaunchKernelGGL(k_raycast, grid, block, 0, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}
