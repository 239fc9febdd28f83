// xs_signmap.hip — owner-side operations of the sign map (xs_signmap.h): size, reset, rebuild from a volume.
// The integrate kernels (xs_tsdf.hip) set its bytes, the single-GPU ray march (xs_raycast.hip) reads them.
#include "xs_device.h"
#include "xs_signmap.h"
#include "../../include/xslam_amd.h"

using namespace xs;

namespace {
__global__ void __launch_bounds__(256) k_signmap_reset(SignMap m, int *head, float *t, float time_step, int X, int Y, int Z) {
    const int nb = m.nx * m.ny * m.nz;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) {
        const int bx = i % m.nx, by = (i / m.nx) % m.ny, bz = i / (m.nx * m.ny);
        // a clear byte promises that the brick's whole 3x3x3 neighbourhood lies inside the volume: set wherever a neighbour is missing
        // or overhangs the volume's end (sizes that are not a multiple of the brick edge)
        const bool shell = bx == 0 || by == 0 || bz == 0 || ((bx + 2) << m.shift) > X || ((by + 2) << m.shift) > Y || ((bz + 2) << m.shift) > Z;
        m.raw[i] = 0;
        m.dil[i] = shell ? 1 : 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        head[0] = m.shift; head[1] = m.nx; head[2] = m.ny; head[3] = m.nz; head[4] = m.nt; head[5] = __float_as_int(time_step);
        head[6] = (int)(m.dil - m.raw);
        // the reference's running sum (RayCaster.cu:222, 245: time_curr += time_step), one float addition per iteration
        float tc = 0.2f;
        for (int j = 0; j < m.nt; ++j) { t[j] = tc; tc = tc + time_step; }
    }
}
// one workgroup per brick row (all bricks of one (by, bz)): each thread scans voxels of the row's bricks and marks negative ones
__global__ void __launch_bounds__(256) k_signmap_scan(SignMap m, const float *value, size_t vstep, int X, int Y, int Z, int zs0, int zs1) {
    const int by = blockIdx.x % m.ny, bz = blockIdx.x / m.ny;
    const int e = 1 << m.shift;
    const int y0 = by << m.shift, z0 = bz << m.shift;
    const int y1 = min(Y, y0 + e), z1 = min(min(Z, z0 + e), zs1);
    for (int z = max(z0, zs0); z < z1; ++z)     // (value holds planes zs0 .. zs1 - 1, the first at offset 0)
        for (int y = y0; y < y1; ++y) {
            const float *row = reinterpret_cast<const float *>(reinterpret_cast<const char *>(value) + ((size_t)(z - zs0) * Y + y) * vstep);
            for (int x = threadIdx.x; x < X; x += 256)
                if (row[x] < 0.0f) signmap_mark(m, x, y, z);
        }
}
static bool good(const int *res, int shift) {
    return res && res[0] > 0 && res[1] > 0 && res[2] > 0 && shift >= 2 && shift <= 6;
}
}  // namespace

extern "C" size_t xs_signmap_bytes(const int *res, int shift) {
    if (!good(res, shift)) return 0;
    return SIGNMAP_HEAD_BYTES + SIGNMAP_MAX_STEPS * sizeof(float) + 2 * signmap_bricks_padded(res, shift);
}

extern "C" int xs_signmap_reset(void *signmap, const int *res, int shift, float tranc_dist, void *stream) {
    if (!signmap || !good(res, shift)) return xs_set_error(hipErrorInvalidValue, "xs_signmap_reset: bad argument");
    const float time_step = tranc_dist * 0.8f;   // RayCaster.cu:350, as xs_raycast forms it
    const int nt = signmap_steps(time_step);
    if (nt == 0) return xs_set_error(hipErrorInvalidValue, "xs_signmap_reset: the march has more steps than the time table holds");
    SignMap m = signmap_view(signmap, res, shift, nt);
    const int nb = m.nx * m.ny * m.nz;
    hipLaunchKernelGGL(k_signmap_reset, dim3((nb + 255) / 256 < 1024 ? (nb + 255) / 256 : 1024), dim3(256), 0, (hipStream_t)stream, m,
                       static_cast<int *>(signmap), const_cast<float *>(m.t), time_step, res[0], res[1], res[2]);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : xs_set_error(e, "xs_signmap_reset: launch failed");
}

extern "C" int xs_signmap_rebuild(void *signmap, const int *res, int shift, float tranc_dist, const float *value, size_t vol_step, void *stream) {
    return xs_signmap_rebuild_slab(signmap, res, shift, tranc_dist, value, vol_step, 0, res ? res[2] : 0, stream);
}
/* the same for one rank's storage of a z-sharded volume: value holds planes [zs0, zs1) */
extern "C" int xs_signmap_rebuild_slab(void *signmap, const int *res, int shift, float tranc_dist, const float *value, size_t vol_step, int zs0, int zs1,
                                       void *stream) {
    if (!value) return xs_set_error(hipErrorInvalidValue, "xs_signmap_rebuild: null volume");
    if (!res || zs0 < 0 || zs1 > res[2] || zs1 < zs0) return xs_set_error(hipErrorInvalidValue, "xs_signmap_rebuild_slab: bad slab");
    const int rc = xs_signmap_reset(signmap, res, shift, tranc_dist, stream);
    if (rc) return rc;
    SignMap m = signmap_view(signmap, res, shift, signmap_steps(tranc_dist * 0.8f));
    hipLaunchKernelGGL(k_signmap_scan, dim3(m.ny * m.nz), dim3(256), 0, (hipStream_t)stream, m, value, vol_step, res[0], res[1], res[2], zs0, zs1);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : xs_set_error(e, "xs_signmap_rebuild: launch failed");
}
