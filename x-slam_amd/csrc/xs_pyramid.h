// xs_pyramid.h — the model-map pyramid's per-pixel work (resizeVMap / resizeNMap twice: Map.cu:105-152, Map.h:46-54), shared by
// k_resize_pyramid (xs_map.hip) and the raycast kernel's epilogue (xs_raycast.hip): a level-2 pixel needs a 4 x 4 block of level-0 pixels,
// which lies inside the raycast workgroup's own pixel tile.
#pragma once
#include "xs_device.h"

namespace xs {
struct PyramidArgs {
    const cfloat *in[2]; cfloat *mid[2]; cfloat *out[2];
    size_t istep, mstep, ostep;
    int rows0, cols0;
};
// one level-1 pixel (x, y) of map m from its 2 x 2 level-0 pixels (as k_resize does): stored, and returned for the level above; false = no value (NaN sentinel stored)
template <bool NORMALIZE>
__device__ __forceinline__ bool pyramid_level1_pixel(const PyramidArgs &a, int m, int x, int y, cfloat3 &n) {
    const int rows1 = a.rows0 / 2;
    const int xs_ = x * 2, ys = y * 2;
    const cfloat *in = a.in[m];
    const cfloat x00 = row_ptr(in, a.istep, ys)[xs_], x01 = row_ptr(in, a.istep, ys)[xs_ + 1];
    const cfloat x10 = row_ptr(in, a.istep, ys + 1)[xs_], x11 = row_ptr(in, a.istep, ys + 1)[xs_ + 1];
    if (isnan(x00.re) || isnan(x01.re) || isnan(x10.re) || isnan(x11.re)) {
        row_ptr(a.mid[m], a.mstep, y)[x] = cfloat(qnan_f(), 0.f);
        return false;
    }
    n.x = (x00 + x01 + x10 + x11) / 4.0f;
    const cfloat y00 = row_ptr(in, a.istep, ys + a.rows0)[xs_], y01 = row_ptr(in, a.istep, ys + a.rows0)[xs_ + 1];
    const cfloat y10 = row_ptr(in, a.istep, ys + a.rows0 + 1)[xs_], y11 = row_ptr(in, a.istep, ys + a.rows0 + 1)[xs_ + 1];
    n.y = (y00 + y01 + y10 + y11) / 4.0f;
    const cfloat z00 = row_ptr(in, a.istep, ys + 2 * a.rows0)[xs_], z01 = row_ptr(in, a.istep, ys + 2 * a.rows0)[xs_ + 1];
    const cfloat z10 = row_ptr(in, a.istep, ys + 2 * a.rows0 + 1)[xs_], z11 = row_ptr(in, a.istep, ys + 2 * a.rows0 + 1)[xs_ + 1];
    n.z = (z00 + z01 + z10 + z11) / 4.0f;
    if (NORMALIZE) n = normalized(n);
    row_ptr(a.mid[m], a.mstep, y)[x] = n.x;
    row_ptr(a.mid[m], a.mstep, y + rows1)[x] = n.y;
    row_ptr(a.mid[m], a.mstep, y + 2 * rows1)[x] = n.z;
    return true;
}
// one level-2 pixel (x2, y2) of map m from the four level-1 pixels under it (values in hand: what it would read back)
template <bool NORMALIZE>
__device__ __forceinline__ void pyramid_level2_pixel(const PyramidArgs &a, int m, int x2, int y2, const cfloat3 (&l1)[4], const bool (&ok)[4]) {
    const int rows2 = (a.rows0 / 2) / 2;
    if (!(ok[0] && ok[1] && ok[2] && ok[3])) {
        row_ptr(a.out[m], a.ostep, y2)[x2] = cfloat(qnan_f(), 0.f);
        return;
    }
    cfloat3 n;
    n.x = (l1[0].x + l1[1].x + l1[2].x + l1[3].x) / 4.0f;
    n.y = (l1[0].y + l1[1].y + l1[2].y + l1[3].y) / 4.0f;
    n.z = (l1[0].z + l1[1].z + l1[2].z + l1[3].z) / 4.0f;
    if (NORMALIZE) n = normalized(n);
    row_ptr(a.out[m], a.ostep, y2)[x2] = n.x;
    row_ptr(a.out[m], a.ostep, y2 + rows2)[x2] = n.y;
    row_ptr(a.out[m], a.ostep, y2 + 2 * rows2)[x2] = n.z;
}
template <bool NORMALIZE>
__device__ __forceinline__ void resize_two_levels(const PyramidArgs &a, int m, int x2, int y2) {
    const int rows1 = a.rows0 / 2, cols1 = a.cols0 / 2, rows2 = rows1 / 2, cols2 = cols1 / 2;
    cfloat3 l1[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = 2 * x2 + (q & 1), y = 2 * y2 + (q >> 1);
        ok[q] = (x < cols1 && y < rows1) && pyramid_level1_pixel<NORMALIZE>(a, m, x, y, l1[q]);
    }
    if (x2 >= cols2 || y2 >= rows2) return;
    pyramid_level2_pixel<NORMALIZE>(a, m, x2, y2, l1, ok);
}
}  // namespace xs
