// xs_signmap.h — the sign map: one byte per brick of 2^shift voxels a side saying "a voxel of this brick may read negative".
// No counterpart in the reference: its ray march (RayCaster.cu:222-247) evaluates every step from t = 0.2 m.  Every event that can end
// that march — a step whose sample leaves the volume, a + to - crossing, a - to + crossing — needs either a sample outside the volume
// or a NEGATIVE voxel on one side of the step, so steps whose samples are known to lie in bricks without negative voxels cannot end it
// and need not be read: the march evaluates only the steps that may matter, each from the same float time the reference's running sum
// has there, and produces the same bits.  The map is only ever a superset: the integrate kernels set a brick's byte when they write a
// negative value into it (written values are never looked at again: bytes are never cleared except by reset / rebuild).
//
// Device layout of the buffer (xs_signmap_bytes):
//   int   head[16]              shift, nx, ny, nz, nt, time_step bits, byte offset of dil from raw
//   float t[SIGNMAP_MAX_STEPS]  t[0] = 0.2, t[j+1] = t[j] + time_step in float: the reference's time_curr at iteration j
//   u8    raw[nb]               brick holds a voxel that was written with a negative value
//   u8    dil[nb]               some brick of the 3x3x3 neighbourhood is raw, is missing, or overhangs the volume's end
// The march (k_raycast, MAP) samples `dil` once per wave, along the ray through the centre of the wave's pixel tile, every dt with
// dt + (5.2 + dt) * delta <= 0.9 brick edges (delta: the tile's half diagonal in normalised image coordinates): every point any of the
// tile's rays reaches between two samples lies in the sample's brick or one of its 26 neighbours, so a clear `dil` byte clears that
// stretch for the whole tile, and keeps it inside the volume; a ballot over the table t[] turns the cleared stretches into a bit mask of
// the iterations that still have to be evaluated.
#pragma once
#include <hip/hip_runtime.h>

namespace xs {
enum { SIGNMAP_MAX_STEPS = 320, SIGNMAP_HEAD_BYTES = 64 };   // (the march kernel keeps one bit per iteration in scalar registers: 5 words)
struct SignMap {            // host-built view, passed in kernel arguments
    unsigned char *raw, *dil;
    const float *t;
    int shift, nx, ny, nz, nt;
};
static inline size_t signmap_bricks_padded(const int *res, int shift) {
    const size_t e = (size_t)1 << shift;
    const size_t nb = ((res[0] + e - 1) >> shift) * ((res[1] + e - 1) >> shift) * ((res[2] + e - 1) >> shift);
    return (nb + 255) & ~(size_t)255;
}
static inline SignMap signmap_view(void *buf, const int *res, int shift, int nt) {
    SignMap m;
    char *p = static_cast<char *>(buf);
    m.t = reinterpret_cast<const float *>(p + SIGNMAP_HEAD_BYTES);
    m.raw = reinterpret_cast<unsigned char *>(p + SIGNMAP_HEAD_BYTES + SIGNMAP_MAX_STEPS * sizeof(float));
    m.dil = m.raw + signmap_bricks_padded(res, shift);
    m.shift = shift; m.nt = nt;
    const int e = 1 << shift;
    m.nx = (res[0] + e - 1) >> shift; m.ny = (res[1] + e - 1) >> shift; m.nz = (res[2] + e - 1) >> shift;
    return m;
}
// entries of the time table for a time step: t[nt-1] is the first time >= 5.0 (the march's max_time), so the reference's loop
// runs nt - 1 iterations on a ray nothing ends; 0 if the table would not fit
static inline int signmap_steps(float time_step) {
    if (!(time_step > 0.0f)) return 0;
    volatile float t = 0.2f;
    for (int n = 1; n <= SIGNMAP_MAX_STEPS; ++n) {
        if (!(t < 5.0f)) return n;
        t = t + time_step;
    }
    return 0;
}
#if defined(__HIPCC__)
// a voxel of brick-space cell (x >> shift, ...) was written with a negative value
__device__ __forceinline__ void signmap_mark(const SignMap &m, int x, int y, int z) {
    const int bx = x >> m.shift, by = y >> m.shift, bz = z >> m.shift;
    const int i = (bz * m.ny + by) * m.nx + bx;
    if (m.raw[i]) return;   // (a stale 0 from another XCD's L2 only repeats the stores below)
    m.raw[i] = 1;
    // (rare — once per brick and launch at most — and inlined into the integrate walk: kept as loops)
#pragma unroll 1
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll 1
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll 1
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = bx + dx, cy = by + dy, cz = bz + dz;
                if ((unsigned)cx < (unsigned)m.nx && (unsigned)cy < (unsigned)m.ny && (unsigned)cz < (unsigned)m.nz)
                    m.dil[(cz * m.ny + cy) * m.nx + cx] = 1;
            }
}
// the same from the buffer alone (the integrate kernels carry one pointer): voxels (x, y, zb..ze-1) of one column
__device__ __forceinline__ void signmap_mark_span(unsigned char *buf, int x, int y, int zb, int ze) {
    const int *h = reinterpret_cast<const int *>(buf);
    SignMap m;
    m.shift = h[0]; m.nx = h[1]; m.ny = h[2]; m.nz = h[3]; m.nt = 0; m.t = nullptr;
    m.raw = buf + SIGNMAP_HEAD_BYTES + SIGNMAP_MAX_STEPS * sizeof(float);
    m.dil = m.raw + h[6];
#pragma unroll 1
    for (int bz = zb >> m.shift; bz <= (ze - 1) >> m.shift; ++bz) signmap_mark(m, x, y, bz << m.shift);
}
#endif
}  // namespace xs
