// xs_map.hip — depth / vertex / normal map preparation for gfx950.  Replaces
// XKinectFusion/src/Map.cu: bilateralKernel (:155-199, launcher :262-271), pyrDownKernel
// (:202-230, :274-283), computeVmapKernel (:8-29, :73-86), computeNmapKernel (:32-70, :89-102),
// resizeMapKernel<normalize> (:105-152, :233-259).
//
// Maps are complex (re, im) float pairs; a vertex / normal map is three stacked planes of
// rows x cols (x rows, then y, then z).  Invalid pixels carry the NaN sentinel in the x plane
// only (Map.cu:27,42,69,126).  All kernels put 64 consecutive columns on a wave so every row
// access is one coalesced segment (512 B of complex per wave-row).
#include <hip/hip_ext.h>
#include "xs_device.h"
#include "../../include/xslam_amd.h"

using namespace xs;

namespace {
constexpr float kSigmaColor = 30.f;   // mm, Map.cu:4
constexpr float kSigmaSpace = 4.5f;   // px, Map.cu:5
constexpr int BR = 6, BD = 2 * BR + 1;  // 13x13 window (Map.cu:169-170)
constexpr int BTX = 64, BTY = 4;        // block = 4 waves, one image row each
}  // namespace

// The 13x13 window of every pixel of a 64x4 block is staged once in LDS (76 x 16 tile, 4.9 KB),
// so the 169 taps per pixel are LDS reads instead of 169 global loads.  The taps are
// accumulated in the reference's order (rows outer, columns inner): float sums are order
// dependent and the result is rounded to an integer millimetre.
//
// The tile holds floats (the taps are needed as floats anyway) and every position the reference's
// clipped loops never visit — outside the image, and the last column and row (Map.cu:172-179) —
// holds kSkip: its colour distance is so large that the weight is exactly 0, and adding 0 leaves
// both sums bit-identical to skipping the tap.  With that the 169 taps unroll with no bounds
// logic: the spatial term is a literal per tap, taps go through the pipeline eight at a time
// (v_pk_* on the element-wise part with independent chains interleaved; the two running sums
// stay in tap order), and LDS latency hides behind independent reads.  (float)(d*d) of the reference equals fl(d)*fl(d): d is exact in
// float and both round the exact product once.
namespace {
constexpr int TP = 4;  // tap pairs per group: eight independent exp chains in flight per lane
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kSkip = -1048576.f;
constexpr float kSpaceInvHalf = 0.5f / (kSigmaSpace * kSigmaSpace);
constexpr float kColorInvHalf = 0.5f / (kSigmaColor * kSigmaColor);
constexpr int TW = BTX + 2 * BR, TH = BTY + 2 * BR;
constexpr int NTAPS = BD * BD;
constexpr int clamp_tap(int tap) { return tap < NTAPS ? tap : 0; }  // the last group is padded; its extras are not summed
constexpr int tap_offset(int tap) { return (clamp_tap(tap) / BD) * TW + clamp_tap(tap) % BD; }
constexpr float space_term(int tap) {
    const int dx = clamp_tap(tap) % BD - BR, dy = clamp_tap(tap) / BD - BR;
    return (float)(dx * dx + dy * dy) * kSpaceInvHalf;
}
}  // namespace

__global__ void __launch_bounds__(BTX *BTY) k_bilateral(const uint16_t *src, size_t sstep, int rows, int cols, cfloat *dst, size_t dstep) {
    __shared__ float tile[TH][TW];
    const int bx = blockIdx.x * BTX, by = blockIdx.y * BTY;
    const int tid = threadIdx.y * BTX + threadIdx.x;
    for (int i = tid; i < TH * TW; i += BTX * BTY) {
        int ty = i / TW, tx = i % TW;
        int gy = by + ty - BR, gx = bx + tx - BR;
        float v = kSkip;
        if (gy >= 0 && gy < rows - 1 && gx >= 0 && gx < cols - 1) v = (float)row_ptr(src, sstep, gy)[gx];
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const f32x2 value = f32x2((float)row_ptr(src, sstep, y)[x]);
    const float *win = &tile[threadIdx.y][threadIdx.x];
    float sum1 = 0, sum2 = 0;
#pragma unroll
    for (int g = 0; g < NTAPS; g += 2 * TP) {
        // Each statement runs over the group's pairs before the next one starts, so the
        // dependent steps of one chain are separated by the same step of the other chains.
        // exp(-p), p >= 0, is the device libm's expf sequence (hi/lo product with log2(e),
        // v_exp_f32 on the fraction, ldexp) without its overflow branch; underflow falls out
        // of ldexp.
        f32x2 tmp[TP], p[TP], ph[TP], pl[TP], e[TP], w[TP], tw[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) tmp[k] = f32x2{win[tap_offset(g + 2 * k)], win[tap_offset(g + 2 * k + 1)]};
#pragma unroll
        for (int k = 0; k < TP; ++k) p[k] = value - tmp[k];
#pragma unroll
        for (int k = 0; k < TP; ++k) p[k] = p[k] * p[k];
#pragma unroll
        for (int k = 0; k < TP; ++k) p[k] = f32x2{space_term(g + 2 * k), space_term(g + 2 * k + 1)} + p[k] * kColorInvHalf;
#pragma unroll
        for (int k = 0; k < TP; ++k) ph[k] = p[k] * -0x1.715476p+0f;
#pragma unroll
        for (int k = 0; k < TP; ++k) pl[k] = __builtin_elementwise_fma(p[k], f32x2(-0x1.715476p+0f), -ph[k]);
#pragma unroll
        for (int k = 0; k < TP; ++k) pl[k] = __builtin_elementwise_fma(p[k], f32x2(-0x1.4ae0bep-26f), pl[k]);
#pragma unroll
        for (int k = 0; k < TP; ++k) e[k] = __builtin_elementwise_rint(ph[k]);
#pragma unroll
        for (int k = 0; k < TP; ++k) ph[k] = (ph[k] - e[k]) + pl[k];
#pragma unroll
        for (int k = 0; k < TP; ++k)
            w[k] = f32x2{__builtin_ldexpf(__builtin_amdgcn_exp2f(ph[k].x), (int)e[k].x), __builtin_ldexpf(__builtin_amdgcn_exp2f(ph[k].y), (int)e[k].y)};
#pragma unroll
        for (int k = 0; k < TP; ++k) tw[k] = tmp[k] * w[k];
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            // (the empty asm keeps the two running sums out of the SLP vectoriser: packing them
            // costs more in register shuffles than the packed add saves)
            if (g + 2 * k < NTAPS) { sum1 += tw[k].x; sum2 += w[k].x; asm("" : "+v"(sum1)); }
            if (g + 2 * k + 1 < NTAPS) { sum1 += tw[k].y; sum2 += w[k].y; asm("" : "+v"(sum1)); }
        }
    }
    int round = __float2int_rn(sum1 / sum2);
    if (round > 5000 || round < 200) round = 0;
    round = max(0, min(round, 32767));
    row_ptr(dst, dstep, y)[x] = cfloat(__int2float_rd(round), 0.f);
}

/* bilateralFilter(const DeviceArray2D<ushort>& src, MapArr& dst)   Map.h:16-22, Map.cu:262-271 */
extern "C" int xs_bilateral_filter(const uint16_t *src, size_t src_step, int rows, int cols, float *dst, size_t dst_step, void *stream) {
    if (!src || !dst) return xs_set_error(hipErrorInvalidValue, "xs_bilateral_filter: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    dim3 block(BTX, BTY), grid(div_up(cols, BTX), div_up(rows, BTY));
    hipLaunchKernelGGL(k_bilateral, grid, block, 0, (hipStream_t)stream, src, src_step, rows, cols, (cfloat *)dst, dst_step);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyr_down(const cfloat *src, size_t sstep, int srows, int scols, cfloat *dst, size_t dstep,
                                                  int drows, int dcols, float sigma_color) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dcols || y >= drows) return;
    const int D = 5;
    const int center = __float2int_rn(row_ptr(src, sstep, 2 * y)[2 * x].re);
    const int tx = min(2 * x - D / 2 + D, scols - 1);
    const int ty = min(2 * y - D / 2 + D, srows - 1);
    int sum = 0, count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const int val = __float2int_rn(row_ptr(src, sstep, cy)[cx].re);
            if (abs(val - center) < 3 * sigma_color) { sum += val; ++count; }
        }
    row_ptr(dst, dstep, y)[x] = cfloat(__int2float_rd(sum / count), 0.f);  // integer division (Map.cu:228)
}

/* pyrDown(const MapArr& src, MapArr& dst)   Map.h:24-29, Map.cu:274-283; dst is (rows/2) x (cols/2) */
extern "C" int xs_pyr_down(const float *src, size_t src_step, int src_rows, int src_cols, float *dst, size_t dst_step, void *stream) {
    if (!src || !dst) return xs_set_error(hipErrorInvalidValue, "xs_pyr_down: null pointer");
    int drows = src_rows / 2, dcols = src_cols / 2;
    if (drows <= 0 || dcols <= 0) return 0;
    dim3 block(64, 4), grid(div_up(dcols, 64), div_up(drows, 4));
    hipLaunchKernelGGL(k_pyr_down, grid, block, 0, (hipStream_t)stream, (const cfloat *)src, src_step, src_rows, src_cols, (cfloat *)dst,
                       dst_step, drows, dcols, kSigmaColor);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_vmap(const cfloat *depth, size_t dstep, int rows, int cols, cfloat *vmap, size_t mstep, float fx_inv,
                                              float fy_inv, float cx, float cy) {
    const int u = threadIdx.x + blockIdx.x * blockDim.x;
    const int v = threadIdx.y + blockIdx.y * blockDim.y;
    if (u >= cols || v >= rows) return;
    cfloat z = row_ptr(depth, dstep, v)[u];
    z /= 1000.f;
    if (z.re != 0) {
        const cfloat vx = z * (float(u) - cx) * fx_inv;
        const cfloat vy = z * (float(v) - cy) * fy_inv;
        row_ptr(vmap, mstep, v)[u] = vx;
        row_ptr(vmap, mstep, v + rows)[u] = vy;
        row_ptr(vmap, mstep, v + rows * 2)[u] = z;
    } else
        row_ptr(vmap, mstep, v)[u] = cfloat(qnan_f(), 0.f);
}

/* createVMap(const Intr&, const MapArr& depth, MapArr& vmap)   Map.h:31-37, Map.cu:73-86 */
extern "C" int xs_create_vmap(const float *intr4, const float *depth, size_t depth_step, int rows, int cols, float *vmap, size_t vmap_step,
                              void *stream) {
    if (!intr4 || !depth || !vmap) return xs_set_error(hipErrorInvalidValue, "xs_create_vmap: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    dim3 block(64, 4), grid(div_up(cols, 64), div_up(rows, 4));
    hipLaunchKernelGGL(k_vmap, grid, block, 0, (hipStream_t)stream, (const cfloat *)depth, depth_step, rows, cols, (cfloat *)vmap, vmap_step,
                       1.f / intr4[0], 1.f / intr4[1], intr4[2], intr4[3]);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nmap(int rows, int cols, const cfloat *vmap, cfloat *nmap, size_t mstep) {
    const int u = threadIdx.x + blockIdx.x * blockDim.x;
    const int v = threadIdx.y + blockIdx.y * blockDim.y;
    if (u >= cols || v >= rows) return;
    if (u == cols - 1 || v == rows - 1) { row_ptr(nmap, mstep, v)[u] = cfloat(qnan_f(), 0.f); return; }
    cfloat3 v00, v01, v10;
    v00.x = row_ptr(vmap, mstep, v)[u];
    v01.x = row_ptr(vmap, mstep, v)[u + 1];
    v10.x = row_ptr(vmap, mstep, v + 1)[u];
    if (!isnan(v00.x.re) && !isnan(v01.x.re) && !isnan(v10.x.re)) {
        v00.y = row_ptr(vmap, mstep, v + rows)[u];
        v01.y = row_ptr(vmap, mstep, v + rows)[u + 1];
        v10.y = row_ptr(vmap, mstep, v + 1 + rows)[u];
        v00.z = row_ptr(vmap, mstep, v + 2 * rows)[u];
        v01.z = row_ptr(vmap, mstep, v + 2 * rows)[u + 1];
        v10.z = row_ptr(vmap, mstep, v + 1 + 2 * rows)[u];
        const cfloat3 r = normalized(cross(v01 - v00, v10 - v00));
        row_ptr(nmap, mstep, v)[u] = r.x;
        row_ptr(nmap, mstep, v + rows)[u] = r.y;
        row_ptr(nmap, mstep, v + 2 * rows)[u] = r.z;
    } else
        row_ptr(nmap, mstep, v)[u] = cfloat(qnan_f(), 0.f);
}

/* createNMap(const MapArr& vmap, MapArr& nmap)   Map.h:39-44, Map.cu:89-102; rows = map rows / 3 */
extern "C" int xs_create_nmap(const float *vmap, float *nmap, size_t map_step, int rows, int cols, void *stream) {
    if (!vmap || !nmap) return xs_set_error(hipErrorInvalidValue, "xs_create_nmap: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    dim3 block(64, 4), grid(div_up(cols, 64), div_up(rows, 4));
    hipLaunchKernelGGL(k_nmap, grid, block, 0, (hipStream_t)stream, rows, cols, (const cfloat *)vmap, (cfloat *)nmap, map_step);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// createVMap + createNMap for every pyramid level in one launch (SurfaceMeasure,
// KinectFusionReconstruction.cpp:290-296).  A vertex is a function of its own depth pixel, so the
// normal's two neighbour vertices are recomputed from depth(u+1, v) and depth(u, v+1) with the same
// three operations instead of being read back from the vertex map: same values, no dependency between
// the two maps, one kernel.  blockIdx.z = level.
struct VnLevel {
    const cfloat *depth; size_t dstep; cfloat *vmap; cfloat *nmap; size_t mstep;
    int rows, cols; float fx_inv, fy_inv, cx, cy;
    float *vreal, *nreal; size_t rstep;   // optional: the real parts again, as float planes (xs_create_vnmaps_real)
};
struct VnArgs { VnLevel lv[3]; int levels; };
__device__ __forceinline__ bool vertex_of(const VnLevel &L, int u, int v, cfloat3 &p) {
    cfloat z = row_ptr(L.depth, L.dstep, v)[u];
    z /= 1000.f;
    if (z.re == 0) return false;
    p.x = z * (float(u) - L.cx) * L.fx_inv;
    p.y = z * (float(v) - L.cy) * L.fy_inv;
    p.z = z;
    return true;
}
__global__ void __launch_bounds__(256) k_vnmaps(const VnArgs a) {
    const VnLevel &L = a.lv[blockIdx.z];
    const int u = threadIdx.x + blockIdx.x * blockDim.x;
    const int v = threadIdx.y + blockIdx.y * blockDim.y;
    if (u >= L.cols || v >= L.rows) return;
    cfloat3 v00;
    const bool ok00 = vertex_of(L, u, v, v00);
    if (ok00) {
        row_ptr(L.vmap, L.mstep, v)[u] = v00.x;
        row_ptr(L.vmap, L.mstep, v + L.rows)[u] = v00.y;
        row_ptr(L.vmap, L.mstep, v + L.rows * 2)[u] = v00.z;
        if (L.vreal) {
            row_ptr(L.vreal, L.rstep, v)[u] = v00.x.re;
            row_ptr(L.vreal, L.rstep, v + L.rows)[u] = v00.y.re;
            row_ptr(L.vreal, L.rstep, v + L.rows * 2)[u] = v00.z.re;
        }
    } else {
        row_ptr(L.vmap, L.mstep, v)[u] = cfloat(qnan_f(), 0.f);
        if (L.vreal) row_ptr(L.vreal, L.rstep, v)[u] = qnan_f();
    }
    const cfloat nan_c(qnan_f(), 0.f);
    if (u == L.cols - 1 || v == L.rows - 1) {
        row_ptr(L.nmap, L.mstep, v)[u] = nan_c;
        if (L.nreal) row_ptr(L.nreal, L.rstep, v)[u] = qnan_f();
        return;
    }
    cfloat3 v01, v10;
    if (ok00 && vertex_of(L, u + 1, v, v01) && vertex_of(L, u, v + 1, v10)) {
        const cfloat3 r = normalized(cross(v01 - v00, v10 - v00));
        row_ptr(L.nmap, L.mstep, v)[u] = r.x;
        row_ptr(L.nmap, L.mstep, v + L.rows)[u] = r.y;
        row_ptr(L.nmap, L.mstep, v + 2 * L.rows)[u] = r.z;
        if (L.nreal) {
            row_ptr(L.nreal, L.rstep, v)[u] = r.x.re;
            row_ptr(L.nreal, L.rstep, v + L.rows)[u] = r.y.re;
            row_ptr(L.nreal, L.rstep, v + 2 * L.rows)[u] = r.z.re;
        }
    } else {
        row_ptr(L.nmap, L.mstep, v)[u] = nan_c;
        if (L.nreal) row_ptr(L.nreal, L.rstep, v)[u] = qnan_f();
    }
}
/* Vertex and normal maps of all pyramid levels (1..3) in one launch: what createVMap(intr(level), depth[level],
 * vmap[level]) followed by createNMap(vmap[level], nmap[level]) produce for level = 0 .. levels-1 (Map.h:31-44).
 * intr4s: levels x {fx, fy, cx, cy} already divided per level; rows0 / cols0: level-0 size, level l is
 * (rows0 >> l) x (cols0 >> l); steps in bytes per level. */
extern "C" int xs_create_vnmaps(int levels, const float *intr4s, const float *const *depths, const size_t *depth_steps, int rows0, int cols0,
                                float *const *vmaps, float *const *nmaps, const size_t *map_steps, void *stream) {
    return xs_create_vnmaps_real(levels, intr4s, depths, depth_steps, rows0, cols0, vmaps, nmaps, map_steps, nullptr, nullptr, nullptr, stream);
}
/* The same launch, additionally writing the real parts of both maps as float planes (vreal[l] / nreal[l]: 3 x rows(l) rows of
 * cols(l) floats, row pitch real_steps[l] bytes; NaN sentinel in the x plane as in the complex maps).  A depth image is real, so
 * the imaginary parts of the current-frame maps are zeros: the ICP reduction can read half the bytes (xs_icp_accumulate*_real).
 * All three arrays NULL = xs_create_vnmaps. */
extern "C" int xs_create_vnmaps_real(int levels, const float *intr4s, const float *const *depths, const size_t *depth_steps, int rows0, int cols0,
                                     float *const *vmaps, float *const *nmaps, const size_t *map_steps, float *const *vreal, float *const *nreal,
                                     const size_t *real_steps, void *stream) {
    if (levels < 1 || levels > 3 || !intr4s || !depths || !depth_steps || !vmaps || !nmaps || !map_steps)
        return xs_set_error(hipErrorInvalidValue, "xs_create_vnmaps: bad arguments");
    if ((vreal == nullptr) != (nreal == nullptr) || (vreal == nullptr) != (real_steps == nullptr))
        return xs_set_error(hipErrorInvalidValue, "xs_create_vnmaps_real: pass all three real-plane arrays or none");
    if (rows0 <= 0 || cols0 <= 0) return 0;
    VnArgs a;
    a.levels = levels;
    for (int l = 0; l < levels; ++l) {
        if (!depths[l] || !vmaps[l] || !nmaps[l]) return xs_set_error(hipErrorInvalidValue, "xs_create_vnmaps: null pointer");
        VnLevel &L = a.lv[l];
        L.depth = (const cfloat *)depths[l]; L.dstep = depth_steps[l];
        L.vmap = (cfloat *)vmaps[l]; L.nmap = (cfloat *)nmaps[l]; L.mstep = map_steps[l];
        L.rows = rows0 >> l; L.cols = cols0 >> l;
        L.fx_inv = 1.f / intr4s[4 * l]; L.fy_inv = 1.f / intr4s[4 * l + 1]; L.cx = intr4s[4 * l + 2]; L.cy = intr4s[4 * l + 3];
        L.vreal = vreal ? vreal[l] : nullptr; L.nreal = nreal ? nreal[l] : nullptr; L.rstep = real_steps ? real_steps[l] : 0;
        if (vreal && (!L.vreal || !L.nreal)) return xs_set_error(hipErrorInvalidValue, "xs_create_vnmaps_real: null pointer");
    }
    for (int l = levels; l < 3; ++l) a.lv[l] = a.lv[0];
    dim3 block(64, 4), grid(div_up(cols0, 64), div_up(rows0, 4), levels);
    hipLaunchKernelGGL(k_vnmaps, grid, block, 0, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
template <bool NORMALIZE>
__global__ void __launch_bounds__(256) k_resize(int drows, int dcols, int srows, const cfloat *in, size_t istep, cfloat *out, size_t ostep) {
    const int x = threadIdx.x + blockIdx.x * blockDim.x;
    const int y = threadIdx.y + blockIdx.y * blockDim.y;
    if (x >= dcols || y >= drows) return;
    const int xs_ = x * 2, ys = y * 2;
    const cfloat x00 = row_ptr(in, istep, ys)[xs_], x01 = row_ptr(in, istep, ys)[xs_ + 1];
    const cfloat x10 = row_ptr(in, istep, ys + 1)[xs_], x11 = row_ptr(in, istep, ys + 1)[xs_ + 1];
    if (isnan(x00.re) || isnan(x01.re) || isnan(x10.re) || isnan(x11.re)) {
        row_ptr(out, ostep, y)[x] = cfloat(qnan_f(), 0.f);
        return;
    }
    cfloat3 n;
    n.x = (x00 + x01 + x10 + x11) / 4.0f;
    const cfloat y00 = row_ptr(in, istep, ys + srows)[xs_], y01 = row_ptr(in, istep, ys + srows)[xs_ + 1];
    const cfloat y10 = row_ptr(in, istep, ys + srows + 1)[xs_], y11 = row_ptr(in, istep, ys + srows + 1)[xs_ + 1];
    n.y = (y00 + y01 + y10 + y11) / 4.0f;
    const cfloat z00 = row_ptr(in, istep, ys + 2 * srows)[xs_], z01 = row_ptr(in, istep, ys + 2 * srows)[xs_ + 1];
    const cfloat z10 = row_ptr(in, istep, ys + 2 * srows + 1)[xs_], z11 = row_ptr(in, istep, ys + 2 * srows + 1)[xs_ + 1];
    n.z = (z00 + z01 + z10 + z11) / 4.0f;
    if (NORMALIZE) n = normalized(n);
    row_ptr(out, ostep, y)[x] = n.x;
    row_ptr(out, ostep, y + drows)[x] = n.y;
    row_ptr(out, ostep, y + 2 * drows)[x] = n.z;
}

static int resize_map(bool normalize, const float *in, size_t istep, int srows, int scols, float *out, size_t ostep, void *stream) {
    if (!in || !out) return xs_set_error(hipErrorInvalidValue, "xs_resize_map: null pointer");
    int drows = srows / 2, dcols = scols / 2;
    if (drows <= 0 || dcols <= 0) return 0;
    dim3 block(64, 4), grid(div_up(dcols, 64), div_up(drows, 4));
    if (normalize)
        hipLaunchKernelGGL(k_resize<true>, grid, block, 0, (hipStream_t)stream, drows, dcols, srows, (const cfloat *)in, istep, (cfloat *)out, ostep);
    else
        hipLaunchKernelGGL(k_resize<false>, grid, block, 0, (hipStream_t)stream, drows, dcols, srows, (const cfloat *)in, istep, (cfloat *)out, ostep);
    XS_CHECK(hipGetLastError());
    return 0;
}
// Both halvings of both model maps in one launch (the orchestrator's per-frame pyramid: resizeVMap and
// resizeNMap for levels 1 and 2, KinectFusionReconstruction.cpp:272-277).  A thread owns one level-2
// pixel: it forms the four level-1 pixels under it (each from its own 2x2 of level 0, exactly as
// k_resize does), stores them, and averages them again from registers — the values it would read back.
// blockIdx.z picks the map (0: vertices, 1: normals, renormalised at both levels).
#include "xs_pyramid.h"   // PyramidArgs, resize_two_levels: shared with the raycast kernel, which can build the pyramid of its own tile
__global__ void __launch_bounds__(256) k_resize_pyramid(const PyramidArgs a) {
    const int x2 = threadIdx.x + blockIdx.x * blockDim.x;
    const int y2 = threadIdx.y + blockIdx.y * blockDim.y;
    const int rows1 = a.rows0 / 2, cols1 = a.cols0 / 2;
    if (2 * x2 >= cols1 || 2 * y2 >= rows1) return;
    if (blockIdx.z == 0) resize_two_levels<false>(a, 0, x2, y2);
    else resize_two_levels<true>(a, 1, x2, y2);
}

/* The orchestrator's model-map pyramid in one launch: level 1 and level 2 of the vertex map (as
 * resizeVMap twice) and of the normal map (as resizeNMap twice), Map.h:46-54.  rows0 / cols0: one plane
 * of the level-0 maps; all level-0 maps share in_step, level-1 mid_step, level-2 out_step. */
extern "C" int xs_resize_pyramid(const float *vmap0, const float *nmap0, size_t in_step, int rows0, int cols0, float *vmap1, float *nmap1,
                                 size_t mid_step, float *vmap2, float *nmap2, size_t out_step, void *stream) {
    return xs_resize_pyramid_ex(vmap0, nmap0, in_step, rows0, cols0, vmap1, nmap1, mid_step, vmap2, nmap2, out_step, nullptr, stream);
}
// (the same with a completion event: it rides on the dispatch — hipExtLaunchKernelGGL's stop event — so another stream can wait for the end
// of a frame's raycast + pyramid without a marker packet in this one; NULL = none)
extern "C" int xs_resize_pyramid_ex(const float *vmap0, const float *nmap0, size_t in_step, int rows0, int cols0, float *vmap1, float *nmap1,
                                    size_t mid_step, float *vmap2, float *nmap2, size_t out_step, void *completion_event, void *stream) {
    if (!vmap0 || !nmap0 || !vmap1 || !nmap1 || !vmap2 || !nmap2) return xs_set_error(hipErrorInvalidValue, "xs_resize_pyramid: null pointer");
    const int rows1 = rows0 / 2, cols1 = cols0 / 2;
    if (rows1 <= 0 || cols1 <= 0) return 0;
    PyramidArgs a;
    a.in[0] = (const cfloat *)vmap0; a.in[1] = (const cfloat *)nmap0;
    a.mid[0] = (cfloat *)vmap1; a.mid[1] = (cfloat *)nmap1;
    a.out[0] = (cfloat *)vmap2; a.out[1] = (cfloat *)nmap2;
    a.istep = in_step; a.mstep = mid_step; a.ostep = out_step; a.rows0 = rows0; a.cols0 = cols0;
    dim3 block(64, 4), grid(div_up(div_up(cols1, 2), 64), div_up(div_up(rows1, 2), 4), 2);
    if (completion_event) hipExtLaunchKernelGGL(k_resize_pyramid, grid, block, 0, (hipStream_t)stream, nullptr, (hipEvent_t)completion_event, 0, a);
    else hipLaunchKernelGGL(k_resize_pyramid, grid, block, 0, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}

/* resizeVMap / resizeNMap(const MapArr& input, MapArr& output)   Map.h:46-54, Map.cu:233-259
 * src_rows = rows of ONE plane of the input.  The reference synchronises here (:248). */
extern "C" int xs_resize_vmap(const float *in, size_t in_step, int src_rows, int src_cols, float *out, size_t out_step, void *stream) {
    return resize_map(false, in, in_step, src_rows, src_cols, out, out_step, stream);
}
extern "C" int xs_resize_nmap(const float *in, size_t in_step, int src_rows, int src_cols, float *out, size_t out_step, void *stream) {
    return resize_map(true, in, in_step, src_rows, src_cols, out, out_step, stream);
}
