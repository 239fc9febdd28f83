// xs_tsdf.hip — TSDF volume kernels for gfx950: initialise, depth scaling, per-voxel complex
// integrate.  Replaces XKinectFusion/src/TsdfFusion.cu:4-43 (initVolume), :68-82
// (scaleDepthKernal), :85-201 (tsdfFusionKernal / integrateTsdfVolume).
//
// Volume layout (TsdfVolume.cpp:17-20): value f32, weight i32, grad f32, each a pitched 2-D
// array with rows = Y*Z and cols = X; voxel (x,y,z) at row (y + z*Y), column x.  A z-slab
// [z0, z1) owned by one GPU is the row range [z0*Y, z1*Y); the pointers passed here are the
// base of the slab's own storage and z runs over global indices.
//
// Integrate design (CDNA4).  A wave is 64 consecutive x of one (y, z-chunk): every load and
// store of value/weight/grad is a 256 B coalesced row segment.  A thread walks z.  The part
// of Rv2c*v_g that does not depend on z is formed once per thread in the reference's own
// operation order (x term + y term), so each z step adds one product and the translation —
// the same additions the reference performs, hence the same bits.  A voxel whose projection
// falls outside the image by more than one pixel is rejected on the un-divided coordinates
// (u*fx vs c*(bound)), before the two IEEE divides; anything within a pixel of the border
// takes the exact path.  Only voxels that pass the reference's predicate touch memory.
#include <hip/hip_ext.h>
#include <string.h>
#include <stdio.h>
#include "xs_device.h"
#include "xs_mailbox.h"
#include "xs_signmap.h"
#include "xs_env.h"
#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include "../../include/xslam_amd.h"

using namespace xs;
// The record hand-offs below (relaxed agent-scope stores + s_waitcnt vmcnt(0) + a relaxed ticket, no release fence) are correct because
// gfx942 / gfx950 implement an agent-scope atomic store as a write-through (sc1) store that is acknowledged from memory; that is
// outside the HIP / LLVM memory model, so the file refuses to build for anything else rather than publish stale records there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "write-through record publish: gfx942 / gfx950 only (use a release fence + acq_rel ticket on other targets)"
#endif

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_init_volume(float *value, int *weight, float *grad, size_t step, int X, long long rows) {
    // one thread per 4 consecutive x of one row; rows = Y * (z1 - z0)
    int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    long long row = blockIdx.y;
    for (; row < rows; row += gridDim.y) {
        if (x4 + 3 < X && (step % 16) == 0) {
            float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(row_ptr(value, step, 0) + (size_t)row * (step / 4) + x4) = z4;
            *reinterpret_cast<int4 *>(row_ptr(weight, step, 0) + (size_t)row * (step / 4) + x4) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<float4 *>(row_ptr(grad, step, 0) + (size_t)row * (step / 4) + x4) = z4;
        } else {
            for (int x = x4; x < X && x < x4 + 4; ++x) {
                ((float *)((char *)value + (size_t)row * step))[x] = 0.f;
                ((int *)((char *)weight + (size_t)row * step))[x] = 0;
                ((float *)((char *)grad + (size_t)row * step))[x] = 0.f;
            }
        }
    }
}

extern "C" int xs_init_volume(float *value, int *weight, float *grad, size_t step_bytes, const int *res, int z0, int z1, void *stream) {
    if (!value || !weight || !grad || !res || z1 < z0) return xs_set_error(hipErrorInvalidValue, "xs_init_volume: bad argument");
    long long rows = (long long)res[1] * (z1 - z0);
    if (rows == 0 || res[0] == 0) return 0;
    dim3 block(64, 1);
    dim3 grid(div_up(div_up(res[0], 4), 64), (unsigned)(rows < 65535 ? rows : 65535));
    hipLaunchKernelGGL(k_init_volume, grid, block, 0, (hipStream_t)stream, value, weight, grad, step_bytes, res[0], rows);
    XS_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// scaleDepthKernal (TsdfFusion.cu:68-82) + what the integrate kernel wants to know about the frame: its largest valid depth and the
// smallest and the largest VALID scaled depth per DEPTH_TILE x DEPTH_TILE pixel tile and per 64 x 32 pixel block of tiles ("super tile":
// what one workgroup covers): {lo, hi}, lo NEGATED where the tile holds an invalid pixel (depth 0) — a tile with a hole or a speckle still
// says how near its valid pixels come (round 5: boxes that see such tiles stream their free planes with a validity test instead of walking
// them); a tile without a valid pixel is {-inf, 0}.  A wave takes a strip of 64 pixels x
// DEPTH_TILE rows — every row a coalesced 128-byte read and 256-byte write — lane l carries its column's min / max down the rows, three
// cross-lane steps fold the eight columns of a tile, three more the eight tiles of the strip, and LDS the workgroup's four strips.
enum { DEPTH_TILE = 8, SUPER_W = 64, SUPER_H = 4 * DEPTH_TILE };
struct DepthTile { float lo, hi; };   // over the tile's VALID pixels that lie in the image; lo < 0: |lo| is that minimum and the tile also holds an invalid pixel
__host__ __device__ __forceinline__ float depth_tile_encode(float lo_valid, bool has_invalid) { return has_invalid ? -lo_valid : lo_valid; }
struct DepthTiles {                   // view of the table: tiles [ty][tx], then super tiles [sy][sx]
    const DepthTile *tiles, *supers; int tiles_x, tiles_y, supers_x, supers_y;
};
static inline size_t depth_tiles_count(int rows, int cols) {
    return (size_t)((cols + DEPTH_TILE - 1) / DEPTH_TILE) * ((rows + DEPTH_TILE - 1) / DEPTH_TILE) + (size_t)((cols + SUPER_W - 1) / SUPER_W) * ((rows + SUPER_H - 1) / SUPER_H);
}
static inline DepthTiles depth_tiles_view(const void *buf, int rows, int cols) {
    DepthTiles t;
    t.tiles_x = (cols + DEPTH_TILE - 1) / DEPTH_TILE; t.tiles_y = (rows + DEPTH_TILE - 1) / DEPTH_TILE;
    t.supers_x = (cols + SUPER_W - 1) / SUPER_W; t.supers_y = (rows + SUPER_H - 1) / SUPER_H;
    t.tiles = static_cast<const DepthTile *>(buf); t.supers = t.tiles ? t.tiles + (size_t)t.tiles_x * t.tiles_y : nullptr;
    return t;
}
template <class Src> __device__ __forceinline__ float scaled_depth_of(Src v);
template <> __device__ __forceinline__ float scaled_depth_of<uint16_t>(uint16_t v) {
    const int Dp = v;
    return (Dp > 5000 || Dp < 200) ? 0.f : float(Dp) / 1000.f;   // metres
}
template <> __device__ __forceinline__ float scaled_depth_of<float>(float v) { return v; }   // already scaled: tiles only
// grid (ceil(cols / 64), ceil(rows / 32)), block 256: wave w of workgroup (bx, by) takes pixels [64 bx, +64) x rows [32 by + 8 w, +8)
template <class Src>
__global__ void __launch_bounds__(256) k_scale_depth(const Src *depth, size_t dstep, int rows, int cols, float *scaled, size_t sstep,
                                                     unsigned *max_bits, DepthTile *tiles, int tiles_x, int tiles_y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, ty = blockIdx.y * 4 + wave;
    float lo = __builtin_inff(), hi = 0.f;   // over the valid pixels
    int inv = 0;                             // an invalid pixel seen
#pragma unroll
    for (int r = 0; r < DEPTH_TILE; ++r) {
        const int y = ty * DEPTH_TILE + r;
        if (x < cols && y < rows) {
            const float v = scaled_depth_of<Src>(row_ptr(depth, dstep, y)[x]);
            if (scaled) row_ptr(scaled, sstep, y)[x] = v;
            if (v > 0.f) lo = fminf(lo, v); else inv = 1;
            hi = fmaxf(hi, v);
        }
    }
    __shared__ float s_lo[4], s_hi[4];
    __shared__ int s_inv[4];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); inv |= __shfl_xor(inv, off, 64);
        if (off == DEPTH_TILE / 2 && tiles && (lane & (DEPTH_TILE - 1)) == 0 && x < cols && ty < tiles_y)
            tiles[ty * tiles_x + (x / DEPTH_TILE)] = DepthTile{depth_tile_encode(lo, inv != 0), hi};
    }
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_inv[wave] = inv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3])); hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
        inv = s_inv[0] | s_inv[1] | s_inv[2] | s_inv[3];
        if (tiles) tiles[(size_t)tiles_x * tiles_y + blockIdx.y * gridDim.x + blockIdx.x] = DepthTile{depth_tile_encode(lo, inv != 0), hi};
        const unsigned b = __float_as_uint(hi);  // non-negative floats order like their bit patterns
        if (max_bits && b) atomicMax(max_bits, b);
    }
}
template <class Src>
static void launch_scale_depth(const Src *src, size_t src_step, int rows, int cols, float *scaled, size_t scaled_step, float *max_dev, void *tiles_dev, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_depth<Src>, dim3(div_up(cols, SUPER_W), div_up(rows, SUPER_H)), dim3(256), 0, st, src, src_step, rows, cols, scaled, scaled_step,
                       (unsigned *)max_dev, (DepthTile *)tiles_dev, div_up(cols, DEPTH_TILE), div_up(rows, DEPTH_TILE));
}

/* max_dev: optional device float that receives max(scaled) via atomicMax on its bits; the caller
 * zeroes it before the launch (used by integrate to stop walking behind the farthest surface).
 * tiles_dev: optional table of xs_depth_tiles_bytes(rows, cols) bytes that receives the per-tile depth range
 * (xs_integrate_set_depth_tiles hands it to the integrate calls). */
extern "C" size_t xs_depth_tiles_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return depth_tiles_count(rows, cols) * sizeof(DepthTile);
}
extern "C" int xs_scale_depth_tiles(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step,
                                    float *max_dev, void *tiles_dev, void *stream) {
    if (!depth || !scaled) return xs_set_error(hipErrorInvalidValue, "xs_scale_depth: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    launch_scale_depth(depth, depth_step, rows, cols, scaled, scaled_step, max_dev, tiles_dev, (hipStream_t)stream);
    XS_CHECK(hipGetLastError());
    return 0;
}
extern "C" int xs_scale_depth_max(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step,
                                  float *max_dev, void *stream) {
    return xs_scale_depth_tiles(depth, depth_step, rows, cols, scaled, scaled_step, max_dev, nullptr, stream);
}
/* the tile table of an image that is already scaled (what xs_integrate_scaled does itself when nobody handed it one) */
extern "C" int xs_depth_tiles(const float *scaled, size_t scaled_step, int rows, int cols, void *tiles_dev, void *stream) {
    if (!scaled || !tiles_dev) return xs_set_error(hipErrorInvalidValue, "xs_depth_tiles: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    launch_scale_depth(scaled, scaled_step, rows, cols, (float *)nullptr, (size_t)0, (float *)nullptr, tiles_dev, (hipStream_t)stream);
    XS_CHECK(hipGetLastError());
    return 0;
}
extern "C" int xs_scale_depth(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step, void *stream) {
    return xs_scale_depth_max(depth, depth_step, rows, cols, scaled, scaled_step, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// Brick geometry: 64 (x) x 4 (y) x 8 (z) voxels = one 256-thread workgroup, lane = x, so every
// volume access of a wave is one 256-byte row segment.
#ifndef XS_BRICK_X
#define XS_BRICK_X 32
#endif
enum { BRICK_X = XS_BRICK_X, BRICK_Y = 256 / XS_BRICK_X, BRICK_Z = 8 };   // thread t of the workgroup: column t % BRICK_X, row t / BRICK_X

// Half-spaces of the (padded) view frustum in the volume's voxel-index space, built on the
// host and passed as kernel arguments (wave-uniform: they live in scalar registers).  Along any
// direction the camera-frame position is affine in the voxel index (real parts), so each side
// of the image window, the camera plane and the far limit is
//     alpha + bx*i + by*j + bz*k >= -slack,   (i, j, k) = voxel index + 0.5.
// The window is the reference's in-image test (TsdfFusion.cu:123-124) padded by three pixels,
// the slack is 2e-3 of the term magnitudes: voxels outside provably fail the exact tests;
// voxels inside still take them.  Plane 5 is the far limit c <= cfar; cfar comes from the
// frame's largest depth on the device (alpha[5] holds everything but cfar).
struct Frustum {
    float alpha[6], bx[6], by[6], bz[6], slack[6];
};

// The same half-spaces solved for k, as the column clip wants them: plane p bounds k at
//     root_p(i, j) = A + B*i + C*j     (kind +1: k >= root, -1: k <= root),
// or, where the plane does not depend on k (kind 0), excludes the column when A + B*i + C*j < 0.
// A already holds the slack.  Three scalars per plane instead of six: the integrate kernels are
// short of scalar registers (72 at eight waves per SIMD).
struct ClipPlanes {
    float A[6], B[6], C[6];
    float far_scale;  // what a unit of the far limit adds to A[5]
    int kinds;        // two bits per plane: 0 = no k dependence, 1 = lower bound, 2 = upper bound
};

struct IntegrateArgs {
    Frustum fr;
    ClipPlanes cp;
    const float *depth; size_t dstep; int drows, dcols;
    float *value; int *weight; float *grad; size_t vstep;
    int X, Y, Z;        // full resolution
    int z0, z1;         // slab owned by this launch; storage starts at z0
    int zchunk;         // z range per blockIdx.z (column-walk kernel)
    float tranc_dist, tranc_dist_inv; int max_weight;
    MatS33 R; cfloat3 t;
    Intr intr; float voxel_size, threshold;
    unsigned long long *updated;  // optional device counter
    const float *depth_max;       // optional: largest valid depth of the frame (device)
    int *brick_list; unsigned *brick_count;  // work list of the two-phase path: the region the integrate kernel reads, the workspace header
    unsigned *count_room;                    // the brick kernel's update counts, a word per workgroup (room_count_add)
    unsigned list_cap, pair_word;            // entries a list region holds; header word of the ListPair that goes with brick_list (PAIR_*)
    int bricks_x, bricks_y, bricks_z, brick_z;  // brick_z: planes per brick (runtime; BRICK_Z by default)
    unsigned kflags;              // KF_*
    float near_margin;            // pixels: how near a half-integer a streamed voxel's approximate image coordinate may come before the exact path decides (stream_margins)
    const unsigned *mailbox; unsigned mailbox_seq;   // posted pose: what k_pose_gate polls ...
    unsigned *pose_dev;                              // ... and where it leaves {cmd, 24 floats} for k_integrate_bricks<., true>
    unsigned char *signmap;       // xs_signmap.h buffer (whole-volume launches) or null: bricks that receive a negative value are marked
    DepthTiles dt;                // per-tile depth range of the frame (k_scale_depth), for k_classify_boxes
    unsigned *box_class;          // [list entry][BOXES_PER_BRICK] box_word (k_classify_boxes) or null: every box takes the exact walk
    size_t probe_offset;          // XS_EXPERIMENTS + XS_WG_TIMES only: bytes from box_class to the record area
};
// KF_ALWAYS_STORE: write every updated voxel's three words even where the bits do not change (measurement aid)
// KF_COUNT_CLASSES: the classification counts its boxes in the workspace header, words CLASS_COUNT_WORD + 0..3, + 6, + 7: boxes wholly free /
//                   wholly empty / with planes to walk, planes walked, planes streamed with the in-image test (EDGE), with the validity test (SPECKLE)
// KF_FAR_FIRST:     the list is taken from its end (see k_integrate_bricks)
// KF_NO_TESTED_STREAM: the numeric shortcuts of the EDGE / SPECKLE classes are not proven for this launch (sensor too large or imaginary pose
//                   parts too large: stream_margins): such boxes take the exact walk
enum { KF_ALWAYS_STORE = 1u, KF_COUNT_CLASSES = 2u, KF_FAR_FIRST = 4u, KF_NO_TESTED_STREAM = 8u };
enum { CLASS_COUNT_WORD = 48 };

namespace {
// device side of the frustum: add the far limit (see struct Frustum).  Behind the farthest
// surface by more than the truncation band nothing is written: v_c_1 = Dp*(xl, yl, 1) lies on
// the voxel's own ray, so sdf = (Dp - c) * |v_c|/c with |v_c|/c >= 1 and Dp <= Dmax; c > Dmax +
// trunc therefore gives sdf < -trunc.  Without a frame maximum the valid-depth gate of
// scaleDepth (5 m, TsdfFusion.cu:77) bounds Dp.
__device__ __forceinline__ float far_limit(const IntegrateArgs &a) {
    const float dmax = a.depth_max ? *a.depth_max : 5.0f;
    return dmax * 1.0001f + 1.05f * a.tranc_dist;
}
__device__ __forceinline__ Frustum device_frustum(const IntegrateArgs &a) {
    Frustum f = a.fr;
    f.alpha[5] += far_limit(a);
    return f;
}
// z interval [zb, ze) of column (x, y) that can pass the tests, padded by two voxels.  Branch-free per
// lane: the plane kinds are wave-uniform.
__device__ __forceinline__ void clip_column(const ClipPlanes &c, float far, int x, int y, int &zb, int &ze) {
    float lo = (float)zb, hi = (float)ze;
    const float i = x + 0.5f, j = y + 0.5f;
    bool empty = false;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float a0 = (p == 5) ? c.A[5] + far * c.far_scale : c.A[p];
        const float root = a0 + c.B[p] * i + c.C[p] * j;
        const int kind = (c.kinds >> (2 * p)) & 3;
        if (kind == 1) lo = fmaxf(lo, root);
        else if (kind == 2) hi = fminf(hi, root);
        else empty = empty || (root < 0.f);
    }
    zb = max(zb, (int)floorf(lo - 0.5f) - 2);
    ze = empty ? zb : min(ze, (int)ceilf(hi - 0.5f) + 3);
}
// can any voxel of the index box [x0,x1) x [y0,y1) x [z0,z1) pass?  (all corners outside one
// half-space => no)
__device__ __forceinline__ bool box_may_pass(const Frustum &f, int x0, int x1, int y0, int y1, int z0, int z1) {
    const float xa = x0 + 0.5f, xb = x1 - 0.5f, ya = y0 + 0.5f, yb = y1 - 0.5f, za = z0 + 0.5f, zb = z1 - 0.5f;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        // maximum of the affine form over the box: pick the favourable end per axis
        const float m = f.alpha[p] + (f.bx[p] > 0 ? f.bx[p] * xb : f.bx[p] * xa) + (f.by[p] > 0 ? f.by[p] * yb : f.by[p] * ya) +
                        (f.bz[p] > 0 ? f.bz[p] * zb : f.bz[p] * za);
        if (m < -f.slack[p] * 1.5f) return false;
    }
    return true;
}

// The reference's per-voxel body (TsdfFusion.cu:110-168) for one voxel whose current state
// (value, grad, weight) has already been loaded.  Returns true and the new state if the voxel is
// written.
struct PoseRT { MatS33 R; cfloat3 t; };   // volume-to-camera pose: the kernel argument's, or the one a posted launch took from its mailbox
struct VoxelCtx {
    cfloat base[3];
    float fx, fy, cx, cy, ulo, uhi, vlo, vhi;
};
struct VoxelProj {  // what the projection hands to the update
    cfloat3 v_c; cfloat image_x, image_y, Dp; float c;
};
// how project_voxel reads the scaled depth image: plain global loads (64-bit per-lane addresses), or buffer loads through a
// wave-uniform descriptor with a 32-bit per-lane byte offset (no 64-bit multiply-add per gather)
struct DepthGlobal {
    const float *depth; size_t dstep;
    struct __attribute__((packed, aligned(4))) pair { float a, b; };
    __device__ __forceinline__ float one(int y, int x) const { return row_ptr(depth, dstep, y)[x]; }
    __device__ __forceinline__ void two(int y, int x, float &a, float &b) const {
        const pair r = *reinterpret_cast<const pair *>(row_ptr(depth, dstep, y) + x); a = r.a; b = r.b;
    }
};
struct DepthBuffer {
    __amdgpu_buffer_rsrc_t rsrc; int dstep;
    __device__ __forceinline__ float one(int y, int x) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, y * dstep + x * 4, 0, 0));
    }
    __device__ __forceinline__ void two(int y, int x, float &a, float &b) const {
        const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, y * dstep + x * 4, 0, 0);
        a = __builtin_bit_cast(float, r[0]); b = __builtin_bit_cast(float, r[1]);
    }
};
enum { BUFFER_RSRC_FLAGS = 0x00020000 };   // gfx9 raw buffer, dword 3: 32-bit data format
// phase 1 (TsdfFusion.cu:110-143): project the voxel, fetch its depth.  false = not written.
// In two halves so that a caller can put other work between the address and the use of the depth (integrate_span_grouped):
// voxel_pixel — camera-frame point, image coordinates, the pixel (TsdfFusion.cu:110-127); pixel_depth — the depth sample(s) as loaded
// words; voxel_depth — Dp from those words (:128-143).  project_voxel is the three in a row.
struct VoxelPixel { int coo_x, coo_y, near_x, near_y; };
__device__ __forceinline__ bool voxel_pixel(const IntegrateArgs &a, const PoseRT &ps, const VoxelCtx &k, int z, VoxelProj &o, VoxelPixel &px_) {
    const float vgz = (z + 0.5f) * a.voxel_size;
    o.v_c.x = (k.base[0] + ps.R.data[0].z * vgz) + ps.t.x;
    o.v_c.y = (k.base[1] + ps.R.data[1].z * vgz) + ps.t.y;
    o.v_c.z = (k.base[2] + ps.R.data[2].z * vgz) + ps.t.z;
    const float c = o.v_c.z.re;
    o.c = c;
    if (c < 0) return false;  // Re(1/v_c.z) < 0
    const cfloat px = o.v_c.x * k.fx, py = o.v_c.y * k.fy;
    // early reject before the divides: |true image coordinate - (px/c + cx)| << 1 pixel
    // (one min-chain and a single compare instead of four compare-and-branch pairs)
    const float margin = fminf(fminf(px.re - k.ulo * c, k.uhi * c - px.re), fminf(py.re - k.vlo * c, k.vhi * c - py.re));
    if (c > 0 && margin < 0) return false;
    const cfloat inv_z = 1.0f / o.v_c.z;
    o.image_x = px * inv_z + k.cx;
    o.image_y = py * inv_z + k.cy;
    px_.coo_x = __float2int_rd(o.image_x.re - 0.5f);
    px_.coo_y = __float2int_rd(o.image_y.re - 0.5f);
    if (!(px_.coo_x > 1 && px_.coo_y > 1 && px_.coo_x < a.dcols - 1 && px_.coo_y < a.drows - 1)) return false;
    px_.near_x = __float2int_rn(o.image_x.re); px_.near_y = __float2int_rn(o.image_y.re);
    return true;
}
// what a column's voxels share: the z-invariant part of dot(R.row, v_g): (row.x*vgx) + (row.y*vgy) — v_g has zero imaginary part, so each
// complex product is (re*vg, im*vg) — and the conservative image window in un-divided form (one pixel of slack on each side)
__device__ __forceinline__ VoxelCtx voxel_ctx(const IntegrateArgs &a, const PoseRT &ps, int x, int y) {
    const float vgx = (x + 0.5f) * a.voxel_size;
    const float vgy = (y + 0.5f) * a.voxel_size;
    VoxelCtx k;
#pragma unroll
    for (int r = 0; r < 3; ++r) k.base[r] = ps.R.data[r].x * vgx + ps.R.data[r].y * vgy;
    k.fx = a.intr.fx; k.fy = a.intr.fy; k.cx = a.intr.cx; k.cy = a.intr.cy;
    k.ulo = 1.5f - k.cx; k.uhi = (a.dcols - 0.5f) - k.cx + 1.0f;
    k.vlo = 1.5f - k.cy; k.vhi = (a.drows - 0.5f) - k.cy + 1.0f;
    return k;
}
template <bool BILINEAR> struct DepthWords;
template <> struct DepthWords<false> { float n; };
template <> struct DepthWords<true> { float d00, d10, d01, d11; };
template <bool BILINEAR, class Depth>
__device__ __forceinline__ void pixel_depth(const VoxelPixel &q, const Depth &dimg, DepthWords<BILINEAR> &w) {
    if constexpr (BILINEAR) {
        // the four corners as two 8-byte loads (the pair (coo_x, coo_x + 1) is contiguous; 4-byte alignment is all a
        // global load needs); the nearest pixel is always one of them — coo = floor(image - 0.5), so rn(image) is coo or
        // coo + 1 on each axis (ties included) — and is picked from the registers instead of being gathered a fifth time:
        // two address-unit trips per voxel where there were five
        dimg.two(q.coo_y, q.coo_x, w.d00, w.d10);
        dimg.two(q.coo_y + 1, q.coo_x, w.d01, w.d11);
    } else {
        w.n = dimg.one(q.near_y, q.near_x);
    }
}
template <bool BILINEAR>
__device__ __forceinline__ bool voxel_depth(const IntegrateArgs &a, const VoxelPixel &q, const DepthWords<BILINEAR> &w, VoxelProj &o) {
    cfloat Dp;
    if constexpr (BILINEAR) {
        const float d00 = w.d00, d10 = w.d10, d01 = w.d01, d11 = w.d11;
        const int coo_x = q.coo_x, coo_y = q.coo_y;
        const float n0 = q.near_x == coo_x ? d00 : d10, n1 = q.near_x == coo_x ? d01 : d11;
        Dp = cfloat(q.near_y == coo_y ? n0 : n1, 0.0f);
        const float gmax = fmaxf(d00, fmaxf(d01, fmaxf(d10, d11)));
        const float gmin = fminf(d00, fminf(d01, fminf(d10, d11)));
        if (gmax - gmin < a.threshold && d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
            const cfloat one(1.0f, 0.0f);
            const cfloat fa = o.image_x - cfloat(coo_x + 0.5f, 0.0f);
            const cfloat fb = o.image_y - cfloat(coo_y + 0.5f, 0.0f);
            Dp = d00 * (one - fa) * (one - fb) + d10 * fa * (one - fb) + d01 * (one - fa) * fb + d11 * fa * fb;
        }
    } else Dp = cfloat(w.n, 0.0f);
    o.Dp = Dp;
    return Dp.re > 0;  // the update needs Re Dp > 0 (TsdfFusion.cu:150)
}
template <bool BILINEAR, class Depth>
__device__ __forceinline__ bool project_voxel(const IntegrateArgs &a, const PoseRT &ps, const VoxelCtx &k, int z, VoxelProj &o, const Depth &dimg) {
    VoxelPixel q;
    if (!voxel_pixel(a, ps, k, z, o, q)) return false;
    DepthWords<BILINEAR> w;
    pixel_depth<BILINEAR>(q, dimg, w);
    return voxel_depth<BILINEAR>(a, q, w, o);
}
// phase 2 (TsdfFusion.cu:144-167): signed distance, truncation, running mean.
__device__ __forceinline__ bool update_voxel(const IntegrateArgs &a, const VoxelCtx &k, const VoxelProj &p, float pre_v, float pre_g,
                                             int pre_w, float &out_v, float &out_g, int &out_w) {
    // v_c_1 = Dp*(xl, yl, 1) lies on the voxel's own ray (xl = v_c.x / v_c.z), so
    // sdf = |v_c_1| - |v_c| = (Dp - c) * |v_c|/c with |v_c|/c >= 1.  Beyond the truncation band by
    // a safe margin (0.1 % of the band + 10 um, ~30x the float error of the two norms) the side is
    // decided by the depth difference alone: behind the surface nothing is written, in front the
    // value is the truncated constant (1, 0) — exactly what the reference computes — and the
    // two complex norms (6 complex products, 2 square roots, 6 divides) are only evaluated
    // inside the band.
    const float depth_diff = p.Dp.re - p.c;
    const float band = a.tranc_dist * 1.001f + 1e-5f;
    if (depth_diff < -band) return false;
    cfloat tsdf(1.0f, 0.0f);
    if (!(depth_diff > band)) {
        const cfloat xl = (p.image_x - k.cx) / k.fx;
        const cfloat yl = (p.image_y - k.cy) / k.fy;
        const cfloat3 v_c_1 = mk3(p.Dp * xl, p.Dp * yl, p.Dp);
        const cfloat sdf = norm(v_c_1) - norm(p.v_c);
        if (!(sdf.re >= -a.tranc_dist)) return false;
        if (!(sdf.re > a.tranc_dist)) tsdf = sdf * a.tranc_dist_inv;
    }
    // running mean (TsdfFusion.cu:161-167): (prev * w + tsdf) / (w + 1), component-wise by a real
    // divisor.  x / x is exactly 1 and 0 / x is exactly 0 (sign kept) in IEEE arithmetic, so the two
    // divides are skipped where the numerator equals the divisor or is zero — the steady state of
    // free space (value 1, derivative 0), i.e. most written voxels.
    const cfloat tsdf_prev = unpack_tsdf(pre_v, pre_g);
    const cfloat num = tsdf_prev * __int2float_rn(pre_w) + 1.0f * tsdf;
    const float den = __int2float_rn(pre_w + 1);
    const bool div_v = !(num.re == den), div_g = !(num.im == 0.0f);
    out_v = 1.0f; out_g = num.im;
    if (__builtin_amdgcn_ballot_w64(div_v || div_g) != 0) {  // wave-uniform: whole waves of steady free space skip both divides
        if (div_v) out_v = num.re / den;
        if (div_g) out_g = num.im / den;
    }
    out_w = min(pre_w + 1, a.max_weight);
    return true;
}
// update_voxel in two halves, for the kernel that decides first and touches the volume afterwards.
// voxel_tsdf is everything of TsdfFusion.cu:144-159 — it needs no voxel state: false = not written, else the frame's tsdf;
// running_mean is :161-167 on the loaded state.  Same operations in the same order as update_voxel.
__device__ __forceinline__ bool voxel_tsdf(const IntegrateArgs &a, const VoxelCtx &k, const VoxelProj &p, cfloat &tsdf) {
    const float depth_diff = p.Dp.re - p.c;
    const float band = a.tranc_dist * 1.001f + 1e-5f;
    if (depth_diff < -band) return false;
    tsdf = cfloat(1.0f, 0.0f);
    if (!(depth_diff > band)) {
        const cfloat xl = (p.image_x - k.cx) / k.fx;
        const cfloat yl = (p.image_y - k.cy) / k.fy;
        const cfloat3 v_c_1 = mk3(p.Dp * xl, p.Dp * yl, p.Dp);
        const cfloat sdf = norm(v_c_1) - norm(p.v_c);
        if (!(sdf.re >= -a.tranc_dist)) return false;
        if (!(sdf.re > a.tranc_dist)) tsdf = sdf * a.tranc_dist_inv;
    }
    return true;
}
__device__ __forceinline__ void running_mean(int max_weight, cfloat tsdf, float pre_v, float pre_g, int pre_w, float &out_v,
                                             float &out_g, int &out_w) {
    const cfloat tsdf_prev = unpack_tsdf(pre_v, pre_g);
    const cfloat num = tsdf_prev * __int2float_rn(pre_w) + 1.0f * tsdf;
    const float den = __int2float_rn(pre_w + 1);
    const bool div_v = !(num.re == den), div_g = !(num.im == 0.0f);
    out_v = 1.0f; out_g = num.im;
    if (__builtin_amdgcn_ballot_w64(div_v || div_g) != 0) {  // (see update_voxel)
        if (div_v) out_v = num.re / den;
        if (div_g) out_g = num.im / den;
    }
    out_w = min(pre_w + 1, max_weight);
}
template <bool BILINEAR>
__device__ __forceinline__ bool integrate_voxel(const IntegrateArgs &a, const PoseRT &ps, const VoxelCtx &k, int z, float pre_v, float pre_g, int pre_w,
                                                float &out_v, float &out_g, int &out_w) {
    VoxelProj p;
    if (!project_voxel<BILINEAR>(a, ps, k, z, p, DepthGlobal{a.depth, a.dstep})) return false;
    return update_voxel(a, k, p, pre_v, pre_g, pre_w, out_v, out_g, out_w);
}

// z in [zb, ze) of column (x, y).  The voxel's current state (coalesced 256 B rows) is requested
// before the projection arithmetic so its HBM latency runs under it.  (Two planes per trip with
// six reads in flight was measured slower: 102 VGPRs halve the resident waves.)
// OFF32 (the brick kernel): the three arrays are addressed as a wave-uniform base (the brick's first voxel: scalar registers) + one 32-bit
// byte offset per lane that advances by a plane per trip — one VALU instruction per trip for the addresses instead of the six of three
// 64-bit pointers.  ubase: byte offset of the brick's voxel (0, 0, zb0) in each array; zb0: the brick's first plane.
template <bool BILINEAR, bool OFF32 = false, bool SIGN = false>   // SIGN: a.signmap is marked (a template parameter: the walk is bound by instruction issue)
__device__ __forceinline__ unsigned integrate_span(const IntegrateArgs &a, const PoseRT &ps, int x, int y, int zb, int ze, size_t ubase = 0, int zb0 = 0, unsigned lane_off = 0) {
    unsigned n_upd = 0;
    const VoxelCtx k = voxel_ctx(a, ps, x, y);
    const size_t row = (size_t)(zb - a.z0) * a.Y + y;
    float *pos = row_ptr(a.value, a.vstep, 0) + row * (a.vstep / 4) + x;
    int *wpos = row_ptr(a.weight, a.vstep, 0) + row * (a.vstep / 4) + x;
    float *gpos = row_ptr(a.grad, a.vstep, 0) + row * (a.vstep / 4) + x;
    const size_t zstride = (size_t)a.Y * (a.vstep / 4);
    const unsigned always = (a.kflags & KF_ALWAYS_STORE) ? 1u : 0u;
    float vmin = 0.0f;   // smallest value written by this span (sign map: one v_min per written voxel, one test per span)
    if constexpr (OFF32) {
        char *bv = reinterpret_cast<char *>(a.value) + ubase, *bw = reinterpret_cast<char *>(a.weight) + ubase, *bg = reinterpret_cast<char *>(a.grad) + ubase;
        const unsigned plane = (unsigned)a.Y * (unsigned)a.vstep;
        unsigned off = lane_off + (unsigned)(zb - zb0) * plane;
        for (int z = zb; z < ze; ++z, off += plane) {
            float *pos = reinterpret_cast<float *>(bv + off), *gpos = reinterpret_cast<float *>(bg + off);
            int *wpos = reinterpret_cast<int *>(bw + off);
            const float v0 = *pos, g0 = *gpos;
            const int w0 = *wpos;
            float ov, og; int ow;
            if (integrate_voxel<BILINEAR>(a, ps, k, z, v0, g0, w0, ov, og, ow)) {
                if ((__float_as_uint(ov) ^ __float_as_uint(v0)) | always) *pos = ov;
                if ((unsigned)(ow ^ w0) | always) *wpos = ow;
                if ((__float_as_uint(og) ^ __float_as_uint(g0)) | always) *gpos = og;
                if (SIGN) vmin = fminf(vmin, ov);
                ++n_upd;
            }
            else asm volatile("" ::"v"(v0), "v"(g0), "v"(w0));
        }
        if (SIGN && vmin < 0.0f) signmap_mark_span(a.signmap, x, y, zb, ze);
        return n_upd;
    }
    // (Requesting the state of voxel z+1 one trip ahead, in front of or behind the depth gather, was
    // measured slower: loads return in issue order and the extra live registers cost a wave.)
    for (int z = zb; z < ze; ++z, pos += zstride, wpos += zstride, gpos += zstride) {
        const float v0 = *pos, g0 = *gpos;
        const int w0 = *wpos;
        float ov, og; int ow;
        if (integrate_voxel<BILINEAR>(a, ps, k, z, v0, g0, w0, ov, og, ow)) {
            // only the words whose bits change are stored (same volume, fewer bytes: in free space in front of a surface the running
            // mean of (1, 0) with (1, 0) is (1, 0) again, and a saturated weight stays)
            if ((__float_as_uint(ov) ^ __float_as_uint(v0)) | always) *pos = ov;
            if ((unsigned)(ow ^ w0) | always) *wpos = ow;
            if ((__float_as_uint(og) ^ __float_as_uint(g0)) | always) *gpos = og;
            if (SIGN) vmin = fminf(vmin, ov);
            ++n_upd;
        }
        else asm volatile("" ::"v"(v0), "v"(g0), "v"(w0));
        // (the empty asm consumes the three loads on the paths that left early: without it they are
        // still in flight at the loop head, and the wait the compiler puts there to protect their
        // registers also waits for the previous trip's stores to be acknowledged)
    }
    if (SIGN && vmin < 0.0f) signmap_mark_span(a.signmap, x, y, zb, ze);
    return n_upd;
}



// ---- what a brick is, before any of its voxels is touched ----------------------------------------------------------------------
// The reference evaluates every voxel (TsdfFusion.cu:110-167); most written voxels of a frame lie in free space well in front of the
// surface, where the result is known beforehand: tsdf = (1, 0) and the running mean.  k_classify_boxes decides that for each wave-sized
// box of every listed brick (WX columns x WY rows x the brick's planes: what one wave of k_integrate_bricks walks) from the box's eight
// corners and the frame's per-tile depth range:
//   FREE   every voxel passes the in-image test (:123-124) and sees a valid depth farther than its own c by more than the truncation
//          band (+ margin): each is written with tsdf = (1, 0) (:150-159) — no projection, no divide, no gather: a streaming update;
//   EMPTY  every voxel projects outside the image, or onto depths nearer than its own c by more than the band (or invalid): none is
//          written (:123-124, :150) — the box is skipped;
//   MIXED  anything else: the exact walk.
// Conservative by construction: the box's projections lie in the hull of its corners' (c > 0 over the box), the pixel range is padded
// by 1.5 px (the exact path's nearest / bilinear picks reach at most half a pixel past a projection; the rest covers the float
// difference between this projection and the exact one, ~1e-3 px), depths by 2e-4 m (~50x the float error of c at 8 m).
// A class byte holds for every pose whose camera-frame coordinates differ from the classified pose's by at most (dX, dY, dC) anywhere
// in the volume (BoxSlack; zero for a launch classified with its own pose): the pads grow by what such a pose can move a projection
// and a depth, and the host accepts a list for another pose only after checking exactly that (xs_integrate_list_covers).
#ifndef XS_FREE_CHUNK
#define XS_FREE_CHUNK 1
#endif
enum { BOX_MIXED = 0, BOX_FREE = 1, BOX_EMPTY = 2, BOX_MAX_TILES = 128, FREE_CHUNK = XS_FREE_CHUNK };
enum { BOX_WX = BRICK_X < 64 ? BRICK_X : 64, BOX_WY = 64 / BOX_WX, BOXES_PER_BRICK = 4 };   // a wave's part of a brick
struct BoxSlack { float dX, dY, dC; };
__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }    // quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }    // quad_perm [2, 3, 0, 1]
__device__ __forceinline__ float dpp_half_mirror(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true)); }   // lane i <- 7 - i of its 8
__device__ __forceinline__ float dpp_row_mirror(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true)); }    // lane i <- 15 - i of its 16
template <bool MAX> __device__ __forceinline__ float pick(float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); }
template <bool MAX> __device__ __forceinline__ float fold8(float v) {   // over each group of eight lanes, result in all of them
    v = pick<MAX>(v, dpp_xor1(v)); v = pick<MAX>(v, dpp_xor2(v)); return pick<MAX>(v, dpp_half_mirror(v));
}
template <bool MAX> __device__ __forceinline__ float fold64(float v) {  // over the wave, result wave-uniform
    v = fold8<MAX>(v); v = pick<MAX>(v, dpp_row_mirror(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return pick<MAX>(pick<MAX>(r0, r1), pick<MAX>(r2, r3));
}
// A box's verdict, plane by plane: along the box's z the camera depth c is affine, the depth range of the pixels the box can see is one
// pair (lo, hi) for the whole box — so the planes in front of everything (FREE) are the ones at the near end, the planes behind
// everything (EMPTY) the ones at the far end, and what lies between takes the per-voxel walk: two counts and the direction.
//   word = free planes | empty planes << 8 | (c falls with z) << 16 | EDGE << 17
// A surface across the planes puts its truncation band (six or seven planes wide) into one or two bricks' boxes: only those planes are walked.
//
// EDGE (round 5): the box's pixel range leaves the image — a box on the view frustum's side.  Its "free" planes are free WHERE THEY ARE IN
// THE IMAGE: every voxel of them that passes the reference's in-image test (TsdfFusion.cu:123-124) sees a valid depth farther away than its
// own c by more than the band (the range's depth bounds are taken over the part of it that lies in the image), so it is written with
// tsdf = (1, 0) like any free voxel, and the voxels that fail the test are not written: what is left of the per-voxel work is that test
// (integrate_edge_column).  Before, such a box walked every plane it could not call empty — and the frustum's sides pass through free
// space almost everywhere: 72 % of the planes walked on the benchmark scene at 512^3 and at 1024^3 (profiles/tools/emulate_box_classes.py,
// profiles/r05_box_classes_emulated.txt); with the pose slack's pads on the pixel range (classes decided ahead of the final pose) a 32-voxel
// box at 1024^3 is on an edge more often than not.  An EDGE box needs c >= EDGE_CMIN over its corners: the test's shortcut is derived
// for voxels that are not at the camera (integrate_edge_column).
// SPECKLE (round 5): the box sees an INVALID pixel (a hole, a speckle: depth 0) among valid ones that all lie far behind it.  Its "free" planes
// are free WHERE THE VOXEL'S OWN PIXEL IS VALID: such a voxel is written with tsdf = (1, 0) (whether the reference takes the nearest
// pixel or interpolates — it interpolates only where all four neighbours are valid, and the result lies between them), a voxel whose
// nearest pixel is invalid is not written (Dp = 0: TsdfFusion.cu:128-150).  What is left of the per-voxel work is finding that pixel
// and one depth gather (integrate_valid_column); with 0.2 % of a frame's pixels invalid nearly every box sees one, and before this they
// all walked (bench.py roofline_s2.noisy).
enum : unsigned { BOX_EDGE_BIT = 1u << 17, BOX_SPECKLE_BIT = 1u << 18, BOX_NO_STREAM_BIT = 1u << 19 };   // NO_STREAM: see list_append (a brick in two entries)
#define XS_EDGE_CMIN 0.05f
__device__ __forceinline__ unsigned box_word(int n_free, int n_empty, int falls, bool edge = false, bool speckle = false) {
    return (unsigned)n_free | ((unsigned)n_empty << 8) | ((unsigned)falls << 16) | (edge ? (unsigned)BOX_EDGE_BIT : 0u) | (speckle ? (unsigned)BOX_SPECKLE_BIT : 0u);
}
__device__ __forceinline__ float dpp_fold4min(float v) { v = fminf(v, dpp_xor1(v)); return fminf(v, dpp_xor2(v)); }
__device__ __forceinline__ float dpp_fold4max(float v) { v = fmaxf(v, dpp_xor1(v)); return fmaxf(v, dpp_xor2(v)); }
// box = voxel indices [x0, x1) x [y0, y1) x [z0, z1), not empty.  Eight lanes per box (a wave classifies eight boxes at once): lane
// c of the group evaluates corner c, cross-lane steps fold the group, the group's lanes share the tile reads and then take a plane
// each; every lane of a group returns the group's word.
__device__ __forceinline__ unsigned classify_box(const IntegrateArgs &a, const BoxSlack &sl, int x0, int x1, int y0, int y1, int z0, int z1, int corner) {
    const int nz = z1 - z0;
    const unsigned all_walk = box_word(0, 0, 0), all_empty = box_word(0, nz, 0);
    // voxel centres, in the exact path's own units: (index + 0.5) * voxel_size
    const float vx = (((corner & 1) ? x1 - 1 : x0) + 0.5f) * a.voxel_size;
    const float vy = (((corner & 2) ? y1 - 1 : y0) + 0.5f) * a.voxel_size;
    const float vz = (((corner & 4) ? z1 - 1 : z0) + 0.5f) * a.voxel_size;
    const float X = ((a.R.data[0].x.re * vx + a.R.data[0].y.re * vy) + a.R.data[0].z.re * vz) + a.t.x.re;
    const float Y = ((a.R.data[1].x.re * vx + a.R.data[1].y.re * vy) + a.R.data[1].z.re * vz) + a.t.y.re;
    const float c = ((a.R.data[2].x.re * vx + a.R.data[2].y.re * vy) + a.R.data[2].z.re * vz) + a.t.z.re;
    const float cmin = fold8<false>(c) - sl.dC, cmax = fold8<true>(c) + sl.dC;
    if (!(cmin > 1e-3f)) return all_walk;            // the camera plane cuts the box (or nearly): no hull argument
    const float rc = __builtin_amdgcn_rcpf(c);
    const float un = X * rc, vn = Y * rc;            // normalised image coordinates of the corner
    const float u = a.intr.fx * un + a.intr.cx, v = a.intr.fy * vn + a.intr.cy;
    // what a pose within the slack can move a projection by: |d(X/c)| <= (dX + |X/c| dC) / (c - dC)
    const float rcm = __builtin_amdgcn_rcpf(cmin);
    const float pad_u = 1.5f + fabsf(a.intr.fx) * (sl.dX + fold8<true>(fabsf(un)) * sl.dC) * rcm * 1.01f;
    const float pad_v = 1.5f + fabsf(a.intr.fy) * (sl.dY + fold8<true>(fabsf(vn)) * sl.dC) * rcm * 1.01f;
    const float ulo = fold8<false>(u) - pad_u, uhi = fold8<true>(u) + pad_u, vlo = fold8<false>(v) - pad_v, vhi = fold8<true>(v) + pad_v;
    if (!(ulo > -1e6f && uhi < 1e6f && vlo > -1e6f && vhi < 1e6f)) return all_walk;   // (NaN / overflow: take the exact walk)
    const int px0 = __float2int_rd(ulo), px1 = __float2int_ru(uhi), py0 = __float2int_rd(vlo), py1 = __float2int_ru(vhi);
    const bool inside = px0 >= 2 && py0 >= 2 && px1 <= a.dcols - 2 && py1 <= a.drows - 2;
    // the part of the pixel range that lies in the image (what is outside is never written)
    const int qx0 = max(px0, 0), qx1 = min(px1, a.dcols - 1), qy0 = max(py0, 0), qy1 = min(py1, a.drows - 1);
    if (qx0 > qx1 || qy0 > qy1) return all_empty;
    // the depth range over the pixel range: from the tiles, or — a box near the camera covers hundreds of them — from the super tiles
    int tx0 = qx0 / DEPTH_TILE, tx1 = qx1 / DEPTH_TILE, ty0 = qy0 / DEPTH_TILE, ty1 = qy1 / DEPTH_TILE, pitch = a.dt.tiles_x;
    const DepthTile *table = a.dt.tiles;
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > BOX_MAX_TILES) {
        tx0 = qx0 / SUPER_W; tx1 = qx1 / SUPER_W; ty0 = qy0 / SUPER_H; ty1 = qy1 / SUPER_H; pitch = a.dt.supers_x; table = a.dt.supers;
        if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 4 * BOX_MAX_TILES) return all_walk;
    }
    // (four rows of the range per trip, requested together — a row index past the range repeats its last row, which changes neither bound:
    // a box of the benchmark scene sees 3 x 7 tiles with the slack's pads, i.e. one trip of one round of loads where row by row took three)
    float lo = __builtin_inff(), hi = 0.f, neg = 0.f;   // lo: over the valid pixels; neg < 0: a tile of the range holds an invalid one
    for (int ty = ty0; ty <= ty1; ty += 4) {
        const int r0 = ty * pitch, r1 = min(ty + 1, ty1) * pitch, r2 = min(ty + 2, ty1) * pitch, r3 = min(ty + 3, ty1) * pitch;
        for (int tx = tx0 + corner; tx <= tx1; tx += 8) {
            const DepthTile t0 = table[r0 + tx], t1 = table[r1 + tx], t2 = table[r2 + tx], t3 = table[r3 + tx];
            lo = fminf(lo, fminf(fminf(fabsf(t0.lo), fabsf(t1.lo)), fminf(fabsf(t2.lo), fabsf(t3.lo))));
            neg = fminf(neg, fminf(fminf(t0.lo, t1.lo), fminf(t2.lo, t3.lo)));
            hi = fmaxf(hi, fmaxf(fmaxf(t0.hi, t1.hi), fmaxf(t2.hi, t3.hi)));
        }
    }
    lo = fold8<false>(lo); hi = fold8<true>(hi);
    const bool speckle = fold8<false>(neg) < 0.f;             // the box can see an invalid pixel
    const float band = (a.tranc_dist * 1.001f + 1e-5f) + 2e-4f;   // the walk's own band (update_voxel) + margin
    if (cmin - hi > band) return all_empty;                 // behind everything the box can see (hi = 0: nothing valid there)
    // plain free space needs the whole range in the image and valid; else the free planes are streamed with the tests that are left
    // (BOX_EDGE_BIT / BOX_SPECKLE_BIT), which are derived for voxels that are not at the camera
    const bool tested = !inside || speckle;
    const bool may_stream = !tested || (cmin >= XS_EDGE_CMIN && !(a.kflags & KF_NO_TESTED_STREAM));   // (cmin carries the pose slack: it bounds every covered pose's c)
    if (may_stream && lo - cmax > band) return box_word(nz, 0, 0, !inside, speckle);   // in front of everything valid it can see
    if (nz > 8) return all_walk;                            // (more planes per brick than lanes per box: a tuning configuration)
    // Plane by plane.  The four corners of the box's first plane are lanes 0 - 3 (corner bit 2 clear), of its last plane lanes 4 - 7; c moves
    // by dzc per plane at every (x, y).  Lane j takes plane j with the c range of the first plane's corners shifted j planes along.
    const float q_lo = dpp_fold4min(c), q_hi = dpp_fold4max(c);                 // per quad: this plane's corners
    const float o_lo = dpp_half_mirror(q_lo), o_hi = dpp_half_mirror(q_hi);     // the other quad's
    const float c0_lo = (corner & 4) ? o_lo : q_lo, c0_hi = (corner & 4) ? o_hi : q_hi;   // first plane
    const float dzc = a.R.data[2].z.re * a.voxel_size;
    const float pl_lo = (c0_lo + (float)corner * dzc) - sl.dC - 1e-5f, pl_hi = (c0_hi + (float)corner * dzc) + sl.dC + 1e-5f;   // plane `corner`'s c range
    const bool p_ok = corner < nz;
    const bool p_free = p_ok && may_stream && lo - pl_hi > band, p_empty = p_ok && pl_lo - hi > band;
    const int shift = (int)(threadIdx.x & 63u) & ~7;          // the group's first lane in the wave
    const unsigned fm = (unsigned)(__builtin_amdgcn_ballot_w64(p_free) >> shift) & 0xffu, em = (unsigned)(__builtin_amdgcn_ballot_w64(p_empty) >> shift) & 0xffu;
    const int falls = dzc < 0.f ? 1 : 0;
    int n_free, n_empty;
    if (!falls) {   // c grows with z: free planes from the first plane on, empty ones from the last plane back
        n_free = __builtin_ctz(~fm | 0x100u);
        n_empty = __builtin_clz((~em & ((1u << nz) - 1u)) << (32 - nz) | (1u << (31 - nz)));
    } else {
        n_free = __builtin_clz((~fm & ((1u << nz) - 1u)) << (32 - nz) | (1u << (31 - nz)));
        n_empty = __builtin_ctz(~em | 0x100u);
    }
    n_free = min(n_free, nz); n_empty = min(n_empty, nz - n_free);
    return box_word(n_free, n_empty, falls, !inside && n_free > 0, speckle && n_free > 0);
}
// FREE box: voxels (x, y, zb .. ze - 1) of this lane's column (which lies in the volume).  A rolling pipeline over groups of FREE_CHUNK
// planes: the next group's state is requested before the current group is updated and stored, so the wave always has reads in flight
// (a wave's memory operations complete in issue order: the stores of one group would otherwise stand between two groups of reads).
// Addressed like the OFF32 walk: wave-uniform array bases + one 32-bit byte offset per lane and plane, shared by the three arrays.
// off: the column's first voxel (plane zb); plane: bytes between planes.
struct FreeGroup { float v[FREE_CHUNK], g[FREE_CHUNK]; int w[FREE_CHUNK]; };
__device__ __forceinline__ void free_group_load(FreeGroup &s, const char *bv, const char *bw, const char *bg, unsigned off, unsigned plane, int left) {
#pragma unroll
    for (int j = 0; j < FREE_CHUNK; ++j)
        if (j < left) {
            s.v[j] = *reinterpret_cast<const float *>(bv + (off + j * plane));
            s.g[j] = *reinterpret_cast<const float *>(bg + (off + j * plane));
            s.w[j] = *reinterpret_cast<const int *>(bw + (off + j * plane));
        }
}
__device__ __forceinline__ void free_group_store(const IntegrateArgs &a, const FreeGroup &s, char *bv, char *bw, char *bg, unsigned off, unsigned plane, int left,
                                                 unsigned always) {
#pragma unroll
    for (int j = 0; j < FREE_CHUNK; ++j)
        if (j < left) {
            float ov, og; int ow;
            running_mean(a.max_weight, cfloat(1.0f, 0.0f), s.v[j], s.g[j], s.w[j], ov, og, ow);
            if ((__float_as_uint(ov) ^ __float_as_uint(s.v[j])) | always) *reinterpret_cast<float *>(bv + (off + j * plane)) = ov;
            if ((unsigned)(ow ^ s.w[j]) | always) *reinterpret_cast<int *>(bw + (off + j * plane)) = ow;
            if ((__float_as_uint(og) ^ __float_as_uint(s.g[j])) | always) *reinterpret_cast<float *>(bg + (off + j * plane)) = og;
        }
}
// (The streamed planes never mark the sign map: the running mean of a value with (1, 0) is negative only if the value was — and then the
// launch that wrote that negative value marked the brick, the map's bytes are never cleared, and a value that reached the array any other
// way obliges its owner to rebuild the map (xs_signmap.h).  One v_min per voxel and a dependent byte load per column less; 1024^3 -2 us.)
__device__ __forceinline__ unsigned integrate_free_column(const IntegrateArgs &a, char *bv, char *bw, char *bg, unsigned off, unsigned plane, int zb, int ze) {
    const unsigned always = (a.kflags & KF_ALWAYS_STORE) ? 1u : 0u;
    FreeGroup A, B;
    free_group_load(A, bv, bw, bg, off, plane, ze - zb);
#pragma unroll 1
    for (int z = zb; z < ze; z += 2 * FREE_CHUNK, off += 2 * FREE_CHUNK * plane) {
        free_group_load(B, bv, bw, bg, off + FREE_CHUNK * plane, plane, ze - z - FREE_CHUNK);
        free_group_store(a, A, bv, bw, bg, off, plane, ze - z, always);
        free_group_load(A, bv, bw, bg, off + 2 * FREE_CHUNK * plane, plane, ze - z - 2 * FREE_CHUNK);
        free_group_store(a, B, bv, bw, bg, off + FREE_CHUNK * plane, plane, ze - z - FREE_CHUNK, always);
    }
    return (unsigned)(ze - zb);
}


// EDGE box: like integrate_free_column, but a voxel is updated only if it passes the reference's in-image test (TsdfFusion.cu:123-124) —
// the one per-voxel decision left (see BOX_EDGE_BIT).  The exact path (voxel_pixel) decides it from
//     image_x = Re(px * (1 / v_c.z)) + cx,   coo_x = floor(image_x - 0.5),   1 < coo_x < cols - 1   <=>   2.5 <= image_x < cols - 0.5
// (and the same in y).  Here the real parts X, Y, c of v_c are formed with the exact path's own operations (the real part of each complex
// sum and product: the same bits), and the window is tested on the un-divided coordinates with a margin of EDGE_MARGIN pixels: the float
// image_x differs from fx X / c + cx in real arithmetic by < 3e-4 px (three roundings of magnitudes <= 640, the imaginary products ~1e-14),
// the un-divided terms by < 1e-4 px — two hundred times inside the margin, at any c >= XS_EDGE_CMIN.  A voxel inside the window by the
// margin is in the image, one outside by the margin is not, and the few in between (a strip 2 x EDGE_MARGIN px wide along the image
// border: a few per cent of the rows of an edge box) run the exact test itself — wave-uniformly skipped where no lane needs it.
#define XS_EDGE_MARGIN 0.03125f
struct EdgeWindow { float ulo, uhi, vlo, vhi; };   // the window shrunk by the margin, relative to the principal point
__device__ __forceinline__ bool edge_in_image(const IntegrateArgs &a, const PoseRT &ps, const VoxelCtx &k, const EdgeWindow &w, int z) {
    const float vgz = (z + 0.5f) * a.voxel_size;
    const float X = (k.base[0].re + ps.R.data[0].z.re * vgz) + ps.t.x.re;
    const float Y = (k.base[1].re + ps.R.data[1].z.re * vgz) + ps.t.y.re;
    const float c = (k.base[2].re + ps.R.data[2].z.re * vgz) + ps.t.z.re;
    const float pxr = X * k.fx, pyr = Y * k.fy;
    const float d = fminf(fminf(pxr - w.ulo * c, w.uhi * c - pxr), fminf(pyr - w.vlo * c, w.vhi * c - pyr));   // >= 0: inside by the margin
    bool in = d >= 0.0f;
    const bool unsure = !in && !(d < -2.0f * XS_EDGE_MARGIN * c);   // (NaN: unsure — the exact test decides)
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
        if (unsure) { VoxelProj p; VoxelPixel q; in = voxel_pixel(a, ps, k, z, p, q); }
    }
    return in;
}
__device__ __forceinline__ unsigned integrate_edge_column(const IntegrateArgs &a, const PoseRT &ps, char *bv, char *bw, char *bg, unsigned off, unsigned plane,
                                                          int x, int y, int zb, int ze) {
    const unsigned always = (a.kflags & KF_ALWAYS_STORE) ? 1u : 0u;
    const VoxelCtx k = voxel_ctx(a, ps, x, y);
    EdgeWindow w;
    w.ulo = (2.5f + XS_EDGE_MARGIN) - k.cx; w.uhi = ((a.dcols - 0.5f) - XS_EDGE_MARGIN) - k.cx;
    w.vlo = (2.5f + XS_EDGE_MARGIN) - k.cy; w.vhi = ((a.drows - 0.5f) - XS_EDGE_MARGIN) - k.cy;
    unsigned n = 0;
    // the free column's rolling pipeline, one plane per group: the next plane's test and state request go out before this plane's stores
    FreeGroup A, B;
    bool inA = edge_in_image(a, ps, k, w, zb), inB = false;
    if (inA) free_group_load(A, bv, bw, bg, off, plane, 1);
#pragma unroll 1
    for (int z = zb; z < ze; z += 2, off += 2 * plane) {
        inB = z + 1 < ze && edge_in_image(a, ps, k, w, z + 1);
        if (inB) free_group_load(B, bv, bw, bg, off + plane, plane, 1);
        if (inA) { free_group_store(a, A, bv, bw, bg, off, plane, 1, always); ++n; }
        inA = z + 2 < ze && edge_in_image(a, ps, k, w, z + 2);
        if (inA) free_group_load(A, bv, bw, bg, off + 2 * plane, plane, 1);
        if (inB) { free_group_store(a, B, bv, bw, bg, off + plane, plane, 1, always); ++n; }
    }
    return n;
}


// SPECKLE box (and EDGE + SPECKLE): a voxel of a free plane is written iff it is in the image AND its nearest pixel is valid (BOX_SPECKLE_BIT).
// The exact path finds that pixel from image_x = Re(px * (1 / v_c.z)) + cx, near_x = rn(image_x).  Here u = pxr * rcp(c) + cx from the same
// real parts (edge_in_image has the argument): the two differ by < 2.5e-4 px — the exact one carries five roundings of magnitudes <= 640
// (1.4e-4), this one v_rcp_f32's ulp, a product and a sum (1.0e-4) — so wherever u lies further than the margin (1 / 2048 px at that size) from a
// half-integer both round to the same pixel, and the ~0.2 % of voxels that lie nearer (one plane of a wave in eight) take the exact
// voxel_pixel, wave-uniformly skipped where no lane needs it.  THOSE FIGURES ARE THE 640 x 480 SENSOR'S: every rounding above is half an ulp
// of a magnitude up to E = max(cols + |cx|, rows + |cy|), so the margin is a launch parameter, a.near_margin = 8 ulp(E) (never below
// 1 / 2048: E < 1024 gives exactly that), against a worst case of 4 ulp(E) by the count above and 2 ulp(E) measured over 2e7 random voxels
// per width; and the argument drops the imaginary parts' contribution to the real part of px * (1 / v_c.z), which is
// <= E (Im c / c)^2 + |fx Im X| |Im c| / c^2 (1e-7 px for a first-order seed of 1e-7): stream_margins bounds it with c >= XS_EDGE_CMIN from
// the launch's pose and, where it exceeds a sixteenth of the margin or E >= 8192, sets KF_NO_TESTED_STREAM — no box is then classed EDGE or
// SPECKLE, they walk (INTEGRATION.md "Sensor size").  The in-image test is edge_in_image's (skipped for a box whose range is
// inside the image: need_image false).  One 4-byte depth gather per voxel; the state loads go out with it, a plane ahead of the stores.
#define XS_NEAR_MARGIN_MIN 0.00048828125f
struct ValidSlot { float v, g, d; int w; bool in; };
template <class Depth>
__device__ __forceinline__ void valid_slot_request(ValidSlot &s, const IntegrateArgs &a, const PoseRT &ps, const VoxelCtx &k, const EdgeWindow &win, bool need_image,
                                                   const Depth &dimg, const char *bv, const char *bw, const char *bg, unsigned off, int z, bool live) {
    s.in = false;
    if (__builtin_amdgcn_ballot_w64(live) == 0) return;
    const float vgz = (z + 0.5f) * a.voxel_size;
    const float X = (k.base[0].re + ps.R.data[0].z.re * vgz) + ps.t.x.re;
    const float Y = (k.base[1].re + ps.R.data[1].z.re * vgz) + ps.t.y.re;
    const float c = (k.base[2].re + ps.R.data[2].z.re * vgz) + ps.t.z.re;
    const float pxr = X * k.fx, pyr = Y * k.fy;
    bool in = live, unsure = false;
    if (need_image) {
        const float d = fminf(fminf(pxr - win.ulo * c, win.uhi * c - pxr), fminf(pyr - win.vlo * c, win.vhi * c - pyr));
        in = live && d >= 0.0f;
        unsure = live && !in && !(d < -2.0f * XS_EDGE_MARGIN * c);
    }
    const float rc = __builtin_amdgcn_rcpf(c);
    const float u = pxr * rc + k.cx, v = pyr * rc + k.cy;
    const float ru = rintf(u), rv = rintf(v);
    const float sure = 0.5f - a.near_margin;
    unsure = unsure || (in && !(fabsf(u - ru) < sure && fabsf(v - rv) < sure));   // (NaN: unsure)
    int nx = (int)ru, ny = (int)rv;
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
        if (unsure) { VoxelProj p; VoxelPixel q; in = voxel_pixel(a, ps, k, z, p, q); nx = q.near_x; ny = q.near_y; }
    }
    s.in = in;
    if (in) {
        s.d = dimg.one(ny, nx);
        s.v = *reinterpret_cast<const float *>(bv + off); s.g = *reinterpret_cast<const float *>(bg + off); s.w = *reinterpret_cast<const int *>(bw + off);
    }
}
__device__ __forceinline__ unsigned valid_slot_retire(const ValidSlot &s, const IntegrateArgs &a, char *bv, char *bw, char *bg, unsigned off, unsigned always) {
    if (!(s.in && s.d > 0.0f)) return 0u;   // (Dp = 0: TsdfFusion.cu:150)
    float ov, og; int ow;
    running_mean(a.max_weight, cfloat(1.0f, 0.0f), s.v, s.g, s.w, ov, og, ow);
    if ((__float_as_uint(ov) ^ __float_as_uint(s.v)) | always) *reinterpret_cast<float *>(bv + off) = ov;
    if ((unsigned)(ow ^ s.w) | always) *reinterpret_cast<int *>(bw + off) = ow;
    if ((__float_as_uint(og) ^ __float_as_uint(s.g)) | always) *reinterpret_cast<float *>(bg + off) = og;
    return 1u;
}
__device__ __forceinline__ unsigned integrate_valid_column(const IntegrateArgs &a, const PoseRT &ps, bool need_image, char *bv, char *bw, char *bg, unsigned off,
                                                           unsigned plane, int x, int y, int zb, int ze) {
    const unsigned always = (a.kflags & KF_ALWAYS_STORE) ? 1u : 0u;
    const VoxelCtx k = voxel_ctx(a, ps, x, y);
    EdgeWindow w;
    w.ulo = (2.5f + XS_EDGE_MARGIN) - k.cx; w.uhi = ((a.dcols - 0.5f) - XS_EDGE_MARGIN) - k.cx;
    w.vlo = (2.5f + XS_EDGE_MARGIN) - k.cy; w.vhi = ((a.drows - 0.5f) - XS_EDGE_MARGIN) - k.cy;
    const DepthGlobal dimg{a.depth, a.dstep};
    unsigned n = 0;
    ValidSlot A, B;
    valid_slot_request(A, a, ps, k, w, need_image, dimg, bv, bw, bg, off, zb, true);
#pragma unroll 1
    for (int z = zb; z < ze; z += 2, off += 2 * plane) {
        valid_slot_request(B, a, ps, k, w, need_image, dimg, bv, bw, bg, off + plane, z + 1, z + 1 < ze);
        n += valid_slot_retire(A, a, bv, bw, bg, off, always);
        valid_slot_request(A, a, ps, k, w, need_image, dimg, bv, bw, bg, off + 2 * plane, z + 2, z + 2 < ze);
        n += valid_slot_retire(B, a, bv, bw, bg, off + plane, always);
    }
    return n;
}

}  // namespace

// One atomic per workgroup (a same-address atomic costs ~12 ns on this chip: one per wave of a
// few thousand bricks would serialise into tens of microseconds), spread over `nslots` words.
__device__ __forceinline__ void block_count_add(unsigned n_upd, unsigned long long *slots, unsigned nslots) {
    __shared__ unsigned s_cnt[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const unsigned s = wave_sum_u32(n_upd);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const unsigned t = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (t) atomicAdd(slots + (b % nslots), (unsigned long long)t);
    }
}
// The brick kernel's update counts: one word per workgroup (COUNT_ROOM_WORDS of them, behind the workspace's 256-byte header), each added to
// by the one workgroup that owns it.  Round 5: they were 16 words in ONE cache line of the header, and atomics to a line are served one at a
// time whatever the word (~8 ns each): a launch whose 2 200 workgroups with work each add once could not end before ~3 + 2 200 x 0.008 =
// 21 us, whatever else was done to it (bricks in two entries, a smaller grid: no change until the counts moved;
// profiles/r05_ab_count_room.txt) — and at 1024^3 8 192 workgroups x 8 ns = 65 us of an 80 us launch.
enum { COUNT_ROOM_WORDS = 8192, WS_HEADER_BYTES = 256, WS_LIST_OFFSET = WS_HEADER_BYTES + COUNT_ROOM_WORDS * 4 };
__device__ __forceinline__ void room_count_add(unsigned n_upd, unsigned *room) {
    __shared__ unsigned s_cnt[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const unsigned s = wave_sum_u32(n_upd);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const unsigned t = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        if (t) atomicAdd(room + (blockIdx.x % COUNT_ROOM_WORDS), t);   // (an atomic: a grid may be larger than the room)
    }
}
enum { FOLD_COUNT_BLOCK = 1024 };
__global__ void __launch_bounds__(FOLD_COUNT_BLOCK) k_fold_count(unsigned *room, unsigned long long *updated) {
    // device-scope exchanges (which also leave the words zero for the next launch): the words were written by atomics from every XCD,
    // and a plain load here can be served from this XCD's L2 copy of a line an earlier fold pulled in — observed as lost counts when the
    // counts shared a line with the brick count that every workgroup reads
    // (all of a thread's exchanges go out before the first result is used: one round trip, where a loop that adds as it goes takes eight)
    unsigned v[COUNT_ROOM_WORDS / FOLD_COUNT_BLOCK];
#pragma unroll
    for (int k = 0; k < COUNT_ROOM_WORDS / FOLD_COUNT_BLOCK; ++k) v[k] = atomicExch(room + threadIdx.x + k * FOLD_COUNT_BLOCK, 0u);
    unsigned long long t = 0;
#pragma unroll
    for (int k = 0; k < COUNT_ROOM_WORDS / FOLD_COUNT_BLOCK; ++k) t += v[k];
    __shared__ unsigned long long s_v[FOLD_COUNT_BLOCK / 64];
    t = wave_sum_u64(t);
    if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long all = 0;
        for (int k = 0; k < FOLD_COUNT_BLOCK / 64; ++k) all += s_v[k];
        if (all) atomicAdd(updated, all);
    }
}

// ---- path 1: column walk (no workspace): thread (x, y) walks its clipped z interval ---------
template <bool BILINEAR, bool SIGN = false>
__global__ void __launch_bounds__(256) k_integrate(const IntegrateArgs a) {
    const int x = threadIdx.x + blockIdx.x * 64;
    const int y = threadIdx.y + blockIdx.y * 4;
    unsigned n_upd = 0;
    if (x < a.X && y < a.Y) {
        int zb = a.z0 + blockIdx.z * a.zchunk;
        int ze = min(zb + a.zchunk, a.z1);
        clip_column(a.cp, far_limit(a), x, y, zb, ze);
        if (zb < ze) n_upd = integrate_span<BILINEAR, false, SIGN>(a, PoseRT{a.R, a.t}, x, y, zb, ze);
    }
    if (a.updated) block_count_add(n_upd, a.updated, 1);
}

// ---- path 2: brick work list ------------------------------------------------------------------
// Phase A: one thread per brick tests it against the padded frustum (and the far limit) and
// appends survivors to a list — a wave ballots, one atomic per wave.  Phase B: resident
// workgroups stride over the list; each handles one 32x8x8 brick (a wave: 32 columns x 2 rows — 128-byte row segments;
// 64x4 bricks list 15 % more of them on the benchmark scene, 2 224 against 1 892, i.e. more than the 2 048 resident workgroups,
// and 16x16 ones stream a quarter slower: profiles/tools/ab_brick_shape.sh).  The voxels that can be
// written (a few % of an ICL-like volume) are thereby spread over every CU instead of being
// concentrated in the few columns that cross the frustum.
// (round 5: one reservation per WORKGROUP of 1 024 threads instead of one per wave — same-address atomics are served one at a time, ~12 ns
// each, and the listed bricks of a 1024^3 frustum sit in ~600 waves: 7 of the kernel's 8.9 us; entries stay in thread order, i.e. plane order)
enum { CLASSIFY_BRICKS_BLOCK = 1024 };
__global__ void __launch_bounds__(CLASSIFY_BRICKS_BLOCK) k_classify_bricks(const IntegrateArgs a) {
    const int nb = a.bricks_x * a.bricks_y * a.bricks_z;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = false;
    if (b < nb) {
        const int bx = b % a.bricks_x, by = (b / a.bricks_x) % a.bricks_y, bz = b / (a.bricks_x * a.bricks_y);
        const int x0 = bx * BRICK_X, y0 = by * BRICK_Y, z0 = a.z0 + bz * a.brick_z;
        const Frustum f = device_frustum(a);
        active = box_may_pass(f, x0, min(x0 + BRICK_X, a.X), y0, min(y0 + BRICK_Y, a.Y), z0, min(z0 + a.brick_z, a.z1));
    }
    const unsigned long long m = __ballot(active);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned s_n[CLASSIFY_BRICKS_BLOCK / 64], s_base;
    if (lane == 0) s_n[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { const unsigned n = s_n[w]; s_n[w] = total; total += n; }   // exclusive prefix: each wave's place
        s_base = total ? atomicAdd(a.brick_count, total) : 0u;
    }
    __syncthreads();
    if (active) {  // packed (bx, by, bz), 10 bits each: the consumer decodes with shifts
        const int bx = b % a.bricks_x, by = (b / a.bricks_x) % a.bricks_y, bz = b / (a.bricks_x * a.bricks_y);
        a.brick_list[s_base + s_n[wave] + __popcll(m & ((1ull << lane) - 1ull))] = bx | (by << 10) | (bz << 20);
    }
}

// ---- the list and its order ---------------------------------------------------------------------------------------------------
// A list is two runs in a region of list_cap entries: nwalk entries from the region's front and nother entries from its back, the two
// counts in one 8-byte word of the workspace header (ListPair) — so that a kernel reserves room in both runs with ONE atomic, and no
// entry's place depends on a total that is only known when every workgroup has reported.  Entry e of the list is region[e] for
// e < nwalk, else region[list_cap - 1 - (e - nwalk)] (list_at).  k_classify_bricks alone fills only the front run, in plane order.
//
// With the boxes' classes known the front run holds the bricks that have planes to walk voxel by voxel and the back run the others, and
// k_integrate_bricks takes the list front to back.  Why: a launch with no more bricks than resident workgroups (the benchmark scene:
// 1 860 bricks, 1 536 workgroups resident at once, the rest as soon as a slot is free) gives every brick its own workgroup and the
// dispatcher deals consecutive workgroups round the CUs, so entries e, e + 256, e + 512 ... share a CU.  Half of the bricks are walked —
// instruction issue, ~0.4 us of SIMD time per plane — and in plane order a CU's share of those ranges from 8 to 192 planes (mean 114):
// the launch lasts as long as the fullest CU (23 us where the emptiest is done after 10: profiles/r04_integrate_wg_times.txt).  With the
// walked bricks in front every CU gets 3 or 4 of them and the streaming ones run beside (S1 kernel 26.7 -> 24.7 us).  Larger launches
// run several rounds of workgroups, which balances them by itself; there the order puts the issue-bound walks in front and lets the
// streaming fill in around them instead of leaving a tail of walks (S2 0.143 -> 0.130 ms, S1 at 1024^3 78.8 -> 75.3 us).
enum { PAIR_PRIMARY = 0, PAIR_SECOND = 52 };   // header words of the two ListPairs: k_classify_bricks'; k_classify_boxes'
__device__ __forceinline__ unsigned list_at(unsigned e, unsigned nwalk, unsigned cap) { return e < nwalk ? e : cap - 1u - (e - nwalk); }
__device__ __forceinline__ unsigned long long list_reserve(unsigned *pair, unsigned n_walk, unsigned n_other) {   // -> the runs' lengths before
    return atomicAdd(reinterpret_cast<unsigned long long *>(pair), (unsigned long long)n_walk | ((unsigned long long)n_other << 32));
}
// One brick's four boxes by the 32 lanes of a half-wave (8 lanes per box, see classify_box); the box's word in all of its eight lanes, bit
// 31 set when the box has planes to walk.  KF_COUNT_CLASSES: counted in the header.
__device__ __forceinline__ unsigned classify_brick_boxes(const IntegrateArgs &a, const BoxSlack &sl, int b, int box, int corner) {
    const int bx = b & 1023, by = (b >> 10) & 1023, bz = b >> 20;
    const int wx0 = bx * BRICK_X + (box * 64) % BRICK_X, wy0 = by * BRICK_Y + (box * 64) / BRICK_X;
    const int zb0 = a.z0 + bz * a.brick_z, ze0 = min(zb0 + a.brick_z, a.z1);
    const int nz = ze0 - zb0;
    unsigned word = box_word(0, nz, 0);   // a wave without a column of the brick in the volume: nothing to write
    if (wx0 < a.X && wy0 < a.Y) word = classify_box(a, sl, wx0, min(wx0 + BOX_WX, a.X), wy0, min(wy0 + BOX_WY, a.Y), zb0, ze0, corner);
    const int nf = (int)(word & 0xff), ne = (int)((word >> 8) & 0xff);
    if (corner == 0 && (a.kflags & KF_COUNT_CLASSES)) {   // boxes wholly free / wholly empty / with planes to walk; + the planes walked
        atomicAdd(a.brick_count + CLASS_COUNT_WORD + (nf == nz ? 0 : ne == nz ? 1 : 2), 1u);
        atomicAdd(a.brick_count + CLASS_COUNT_WORD + 3, (unsigned)(nz - nf - ne));
        if (word & BOX_SPECKLE_BIT) atomicAdd(a.brick_count + CLASS_COUNT_WORD + 7, (unsigned)nf);   // planes streamed with the validity test (with or without the in-image test)
        else if (word & BOX_EDGE_BIT) atomicAdd(a.brick_count + CLASS_COUNT_WORD + 6, (unsigned)nf);   // planes streamed with the in-image test alone (words + 4, + 5: the second ListPair)
    }
    return word | (nf + ne < nz ? 1u << 31 : 0u);
}
// What a workgroup does with the n bricks it holds in LDS (s_brick; their boxes' words in s_word, bit 31 = planes to walk): the first
// wave ranks them inside the two runs, reserves room for both with one atomic, and all threads write the entries and the classes.
// Call from every thread; s_pos: n words; s_base: 2 words.
//
// A brick in TWO entries (split, small launches only): a wave that walks a whole box inside a surface's band is one chain of ~8 x 1 500
// instructions, and a wave alone on its SIMD issues at half the SIMD's rate (profiles/r05_valu_issue_calibration.txt: 2.3 ns per instruction
// against 1.2 with two waves) — a launch whose bricks are all resident at once lasts as long as that chain (17-19 us of scene S1's 21) while
// most SIMDs hold one such wave or none.  A brick with a box that walks >= SPLIT_MIN_PLANES planes therefore takes two neighbouring entries
// of the front run: the first streams the boxes' free planes and walks the half of each box's walked planes next to them, the second walks
// the half next to the empty planes and does not stream (BOX_NO_STREAM_BIT).  The halves are expressed in the class words themselves: the first entry counts the second's planes as empty,
// the second counts the first's as free-and-not-to-be-streamed; every voxel is still visited once.
#ifndef XS_SPLIT_MIN_PLANES
#define XS_SPLIT_MIN_PLANES 6
#endif
enum { SPLIT_MIN_PLANES = XS_SPLIT_MIN_PLANES, POS_OTHER_BIT = 1u << 31, POS_SPLIT_BIT = 1u << 30 };
__device__ __forceinline__ int box_walked_planes(unsigned word, int nz) { return nz - (int)(word & 0xffu) - (int)((word >> 8) & 0xffu); }
__device__ __forceinline__ void list_append(const IntegrateArgs &a, unsigned *pair, int *region, unsigned n, const int *s_brick, const unsigned *s_word,
                                            unsigned *s_pos, unsigned *s_base, bool split) {
    const unsigned tid = threadIdx.x;
    if (tid < 64u) {
        unsigned nw = 0, no = 0;
        const unsigned long long below = (1ull << tid) - 1ull;
        for (unsigned c = 0; c < n; c += 64u) {
            const unsigned k = c + tid;
            const bool live = k < n;
            const bool walk = live && ((s_word[k * 4] | s_word[k * 4 + 1] | s_word[k * 4 + 2] | s_word[k * 4 + 3]) >> 31) != 0u;
            bool heavy = false;
            if (split && walk) {
                const int nz = min(a.brick_z, a.z1 - (a.z0 + (s_brick[k] >> 20) * a.brick_z));
                const int most = max(max(box_walked_planes(s_word[k * 4], nz), box_walked_planes(s_word[k * 4 + 1], nz)),
                                     max(box_walked_planes(s_word[k * 4 + 2], nz), box_walked_planes(s_word[k * 4 + 3], nz)));
                heavy = most >= SPLIT_MIN_PLANES;
            }
            const unsigned long long m1 = __ballot(walk), m2 = __ballot(live && !walk), mh = __ballot(heavy);
            if (live) s_pos[k] = walk ? (nw + (unsigned)__popcll(m1 & below) + (unsigned)__popcll(mh & below)) | (heavy ? (unsigned)POS_SPLIT_BIT : 0u)
                                      : (unsigned)POS_OTHER_BIT | (no + (unsigned)__popcll(m2 & below));
            nw += (unsigned)__popcll(m1) + (unsigned)__popcll(mh); no += (unsigned)__popcll(m2);
        }
        if (tid == 0 && n) {
            const unsigned long long was = list_reserve(pair, nw, no);
            s_base[0] = (unsigned)was; s_base[1] = (unsigned)(was >> 32);
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < n * BOXES_PER_BRICK; i += blockDim.x) {
        const unsigned k = i / BOXES_PER_BRICK, r = s_pos[k];
        const unsigned pos = (r >> 31) ? a.list_cap - 1u - (s_base[1] + (r & 0x7fffffffu)) : s_base[0] + (r & ~(unsigned)POS_SPLIT_BIT);
        const unsigned word = s_word[i] & 0x7fffffffu;
        if ((r >> 31) == 0u && (r & POS_SPLIT_BIT)) {
            const int nz = min(a.brick_z, a.z1 - (a.z0 + (s_brick[k] >> 20) * a.brick_z));
            const int wk = box_walked_planes(word, nz), h = wk >> 1;
            a.box_class[(size_t)pos * BOXES_PER_BRICK + i % BOXES_PER_BRICK] = word + ((unsigned)h << 8);             // the second entry's planes: not this one's
            a.box_class[(size_t)(pos + 1u) * BOXES_PER_BRICK + i % BOXES_PER_BRICK] =
                ((word & ~(0xffu | BOX_EDGE_BIT | BOX_SPECKLE_BIT)) | ((word & 0xffu) + (unsigned)(wk - h))) | BOX_NO_STREAM_BIT;
            if (i % BOXES_PER_BRICK == 0) { region[pos] = s_brick[k]; region[pos + 1u] = s_brick[k]; }
            continue;
        }
        a.box_class[(size_t)pos * BOXES_PER_BRICK + i % BOXES_PER_BRICK] = word;
        if (i % BOXES_PER_BRICK == 0) region[pos] = s_brick[k];
    }
}

// The boxes' classes for the list k_classify_bricks left (the primary pair and region), and the list again in the order
// k_integrate_bricks takes it (the second pair and region: ord).  Eight lanes per box, 8 bricks per workgroup and trip; a reservation
// per 8 bricks while the launch is small, per ORDER_BATCH = 32 beyond (S2's 23 000 bricks then take 720 atomics instead of 2 900; ~12 ns each on one
// address).  Batches are runs of list neighbours and arrive roughly in list order, so each run of the ordered list keeps the plane order
// of the first — which matters: the same bricks dealt round the workgroups of a fused kernel (frustum test + classes in one launch, no
// faster: 12.5 us against 3.8 + 8.4) came out in an order that made the S2 launch 9 % slower (profiles/r04_ab_classify_fused.txt).
// The second pair is zero when the kernel starts (the launcher clears it).
struct BoxOrder { int *list; };
#ifndef XS_SPLIT_HEAVY
#define XS_SPLIT_HEAVY 1
#endif
#ifndef XS_ORDER_BATCH
#define XS_ORDER_BATCH 32   // bricks per reservation beyond 4 096 listed ones.  A workgroup takes its batch in rounds of 8 bricks (8 lanes per box, 4 boxes): 64
                            // was eight dependent rounds (18 us at 1024^3's 12.5 K bricks), 32 is four — 1024^3 tracking +0.5-1 %, scene S2's whole call
                            // 0.173 -> 0.169 ms; 16 doubles the same-address atomics again and loses (profiles/r05_ab_order_batch.txt)
#endif
enum { ORDER_BATCH = XS_ORDER_BATCH };
// (1 024-thread workgroups — a batch of 32 in ONE round — measured 14.5 us against 15.1 at 1024^3: the rounds are not what the kernel waits for.)
__global__ void __launch_bounds__(256) k_classify_boxes(const IntegrateArgs a, const BoxSlack sl, const BoxOrder ord) {
    const unsigned nwalk = a.brick_count[PAIR_PRIMARY], count = nwalk + a.brick_count[PAIR_PRIMARY + 1];
    const int tid = (int)threadIdx.x, corner = tid & 7, box = (tid >> 3) & (BOXES_PER_BRICK - 1), slot = tid >> 5;
    __shared__ unsigned s_word[ORDER_BATCH * BOXES_PER_BRICK], s_pos[ORDER_BATCH], s_base[2];
    __shared__ int s_brick[ORDER_BATCH];
    const unsigned batch = count <= 4096u ? 8u : (unsigned)ORDER_BATCH;
    // (a launch that small has every brick resident at once and lasts as long as its longest wave: list_append, "a brick in TWO entries";
    // a brick with >= SPLIT_MIN_PLANES >= 6 planes leaves room for both entries in a region laid out for 2-plane bricks)
    const bool split = XS_SPLIT_HEAVY != 0 && count <= 4096u && a.brick_z >= SPLIT_MIN_PLANES && SPLIT_MIN_PLANES >= 4;
    for (unsigned e0 = blockIdx.x * batch; e0 < count; e0 += gridDim.x * batch) {
        const unsigned n = min(batch, count - e0);
        for (unsigned r = 0; r < n; r += 8u) {
            const unsigned k = r + slot;
            if (k >= n) continue;    // (whole groups of 32 lanes)
            const int b = a.brick_list[list_at(e0 + k, nwalk, a.list_cap)];
            const unsigned word = classify_brick_boxes(a, sl, b, box, corner);
            if (corner == 0) {
                s_word[k * BOXES_PER_BRICK + box] = word;
                if (box == 0) s_brick[k] = b;
            }
        }
        __syncthreads();
        list_append(a, a.brick_count + PAIR_SECOND, ord.list, n, s_brick, s_word, s_pos, s_base, split);
        __syncthreads();
    }
}

// One wave waits at the mailbox for the pose of a posted integrate launch and leaves it in device memory for the launch behind it on the
// stream: a single poller (two thousand workgroups polling one line themselves serialise at the memory side: measured, 87 us instead of 27).
__global__ void __launch_bounds__(64) k_pose_gate(const unsigned *mailbox, unsigned seq, unsigned *pose_dev) {
    __shared__ unsigned s_mail[MAILBOX_WORDS];
    mailbox_wait(mailbox, seq, s_mail, (int)threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) pose_dev[0] = s_mail[1];
    if (threadIdx.x < 24) pose_dev[1 + threadIdx.x] = s_mail[mailbox_word_of((int)threadIdx.x)];
}

// POSTED: the launch is enqueued before its pose exists — behind the classification, which already runs behind the last ICP launch — and
// takes R / t from what k_pose_gate, one wave in front of it on the stream, read out of a mailbox the host posts the final pose to
// (xs_mailbox.h; xs_icp_post_pose): what is left between the last ICP
// reduction and the first integrated voxel is the host's solve + one posted write + one poll instead of those plus a kernel launch
// (~16 us -> ~4).  The frustum planes (brick test, column clip) stay those of the pose the list was classified with, widened: they
// only bound the voxels that take the exact tests, and the host posts only after checking that the final pose's planes lie inside
// them (xs_integrate_pose_covered); otherwise it posts an abandon command and the launch leaves without touching the volume.
#ifndef XS_INTEGRATE_WAVES
// Workgroups per CU = waves per SIMD the brick kernel is compiled for.  8 (64 VGPRs, 78 SGPRs) was round 3's choice: with the free-space
// path and the class look-up next to the walk the instance the pipeline runs then spills 36-48 bytes per lane, and a spill is not free
// even when it is rare: every wave's scratch lines are written back to HBM (2-3 KB per wave, ~15 MB per S1 launch: the traffic counters,
// profiles/r04_integrate_pmc.json).  At 6 (80 VGPRs, 106 SGPRs) no instance has scratch and far fewer scalars live in VGPR lanes: the S1
// kernel inside the pipeline 33 -> 28 us, S2 unchanged (profiles/r04_ab_integrate_waves.txt).
#define XS_INTEGRATE_WAVES 6
#endif
template <bool BILINEAR, bool POSTED = false, bool SIGN = false>
__global__ void __launch_bounds__(256, XS_INTEGRATE_WAVES) k_integrate_bricks(const IntegrateArgs a) {
    PoseRT ps{a.R, a.t};
    if constexpr (POSTED) {
        // the pose k_pose_gate (the launch in front of this one) took from the mailbox: word 0 = command (0: run), words 1..24 = R, t
        const unsigned *pd = a.pose_dev;
        if (__builtin_amdgcn_readfirstlane((int)pd[0]) != 0) return;   // abandoned (or the gate gave up): nothing is written, nothing is counted
        auto f = [&](int i) { return __int_as_float(__builtin_amdgcn_readfirstlane((int)pd[1 + i])); };
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ps.R.data[r].x = cfloat(f(6 * r + 0), f(6 * r + 1));
            ps.R.data[r].y = cfloat(f(6 * r + 2), f(6 * r + 3));
            ps.R.data[r].z = cfloat(f(6 * r + 4), f(6 * r + 5));
        }
        ps.t.x = cfloat(f(18), f(19)); ps.t.y = cfloat(f(20), f(21)); ps.t.z = cfloat(f(22), f(23));
    }
#if defined(XS_EXPERIMENTS) && defined(XS_WG_TIMES)   // measurement only: every workgroup's begin / end on the 100 MHz wall clock, 512 KiB into the tile room of the workspace
    const unsigned long long probe_t0 = wall_clock64();
#endif
    const unsigned nwalk = a.brick_count[a.pair_word], count = nwalk + a.brick_count[a.pair_word + 1];   // (see "the list and its order")
    unsigned n_upd = 0;
    const float far = far_limit(a);
    // The column clip's 20 plane scalars are used once per brick, outside the voxel loop.  Held in scalar registers for the
    // whole kernel they push the voxel loop's own operands out (99 spilled scalars, ~30 v_readlane per voxel to fetch them
    // back — VALU slots, and this kernel is bound by VALU issue: 101 M wave instructions per S2 launch): they live in LDS.
    __shared__ ClipPlanes s_cp;
    float cp_word = 0.f;
    if (threadIdx.y == 0 && threadIdx.x < (int)(sizeof(ClipPlanes) / 4)) cp_word = reinterpret_cast<const float *>(&a.cp)[threadIdx.x];
    // (A bit-reversed entry order for launches with fewer bricks than workgroups — so that list neighbours, the bricks of one surface, land
    // on different CUs — measured 29 -> 42 us on S1: the dispatcher deals consecutive workgroups round the CUs, so the eight workgroups of
    // a CU already take entries 256 apart; profiles/r04_integrate_wg_times.txt.)
    const unsigned first = blockIdx.x, stride = gridDim.x;
    // A list without classes (k_classify_bricks alone) is in plane order (z slowest).  Where the camera looks up the z axis the bricks of
    // the surfaces — the walked, expensive ones — come last, and a launch ends with a tail of them, bound by instruction issue, after the
    // free-space bricks have streamed: taken from its far end the list starts with them and the streaming fills in around (KF_FAR_FIRST,
    // set by the launcher from the pose; never for a list that is ordered by class).
    const bool far_first = (a.kflags & KF_FAR_FIRST) != 0;
    auto entry = [&](unsigned e) { return list_at(far_first ? count - 1u - e : e, nwalk, a.list_cap); };
    // the first entry and its class are requested together, in front of the barrier
    int b_next = 0, cls_next = BOX_MIXED;
    if (first < count) {
        b_next = a.brick_list[entry(first)];
        if (a.box_class) cls_next = (int)a.box_class[entry(first) * BOXES_PER_BRICK + threadIdx.y];
    }
    if (threadIdx.y == 0 && threadIdx.x < (int)(sizeof(ClipPlanes) / 4)) reinterpret_cast<float *>(&s_cp)[threadIdx.x] = cp_word;
    __syncthreads();
    for (unsigned e = first; e < count; e += stride) {
        // (the list was written by the classification kernel, so the compiler reads it with a vector load — back to a scalar,
        // or every address derived from it would be a 64-bit vector quantity)
        if (e != first) {
            b_next = a.brick_list[entry(e)];
            if (a.box_class) cls_next = (int)a.box_class[entry(e) * BOXES_PER_BRICK + threadIdx.y];
        }
        const int b = __builtin_amdgcn_readfirstlane(b_next);
        const int cls_now = cls_next;
        const int bx = b & 1023, by = (b >> 10) & 1023, bz = b >> 20;
        const int t256 = (int)(threadIdx.y * 64 + threadIdx.x);
        const int lx = t256 % BRICK_X, ly = t256 / BRICK_X;
        const int x = bx * BRICK_X + lx, y = by * BRICK_Y + ly;
        int walk_lo = a.z0, walk_hi = a.z1;   // the planes this wave walks voxel by voxel (all of the brick's unless the box's word says otherwise)
        {
            if (a.box_class) {   // what k_classify_boxes found for this wave's part of the brick: free planes at one end, empty ones at the other
                const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane(cls_now);
                const int zb0 = a.z0 + bz * a.brick_z, ze0 = min(zb0 + a.brick_z, a.z1);
                const int nf = (int)(word & 0xffu), ne = (int)((word >> 8) & 0xffu);
                const bool falls = ((word >> 16) & 1u) != 0u;
                const int f0 = falls ? ze0 - nf : zb0, f1 = falls ? ze0 : zb0 + nf;       // free planes [f0, f1)
                walk_lo = falls ? zb0 + ne : zb0 + nf; walk_hi = falls ? ze0 - nf : ze0 - ne;
                if (nf > 0 && !(word & BOX_NO_STREAM_BIT) && x < a.X && y < a.Y) {
                    const size_t ubase = ((size_t)(f0 - a.z0) * a.Y + (size_t)by * BRICK_Y) * a.vstep + (size_t)bx * BRICK_X * 4;
                    char *fv = reinterpret_cast<char *>(a.value) + ubase, *fw = reinterpret_cast<char *>(a.weight) + ubase, *fg = reinterpret_cast<char *>(a.grad) + ubase;
                    const unsigned foff = (unsigned)ly * (unsigned)a.vstep + (unsigned)lx * 4u, fplane = (unsigned)a.Y * (unsigned)a.vstep;
                    if (word & BOX_SPECKLE_BIT) n_upd += integrate_valid_column(a, ps, (word & BOX_EDGE_BIT) != 0u, fv, fw, fg, foff, fplane, x, y, f0, f1);   // (a box that sees an invalid pixel)
                    else if (word & BOX_EDGE_BIT) n_upd += integrate_edge_column(a, ps, fv, fw, fg, foff, fplane, x, y, f0, f1);   // (a box on the frustum's side)
                    else n_upd += integrate_free_column(a, fv, fw, fg, foff, fplane, f0, f1);
                }
                if (walk_lo >= walk_hi) continue;
            }
        }
        if (x < a.X && y < a.Y) {
            const int zb0 = a.z0 + bz * a.brick_z;
            int zb = zb0, ze = min(zb + a.brick_z, a.z1);
            clip_column(s_cp, far, x, y, zb, ze);
            zb = max(zb, walk_lo); ze = min(ze, walk_hi);
            if (zb < ze) {
                const size_t ubase = ((size_t)(zb0 - a.z0) * a.Y + (size_t)by * BRICK_Y) * a.vstep + (size_t)bx * BRICK_X * 4;
                n_upd += integrate_span<BILINEAR, true, SIGN>(a, ps, x, y, zb, ze, ubase, zb0, (unsigned)ly * (unsigned)a.vstep + (unsigned)lx * 4u);
            }
        }
    }
#if defined(XS_EXPERIMENTS) && defined(XS_WG_TIMES)
    if (a.box_class && threadIdx.x == 0) {
        unsigned *rec = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(a.box_class) + a.probe_offset) + 4u * (blockIdx.x * 4 + threadIdx.y);
        rec[0] = (unsigned)probe_t0; rec[1] = (unsigned)wall_clock64(); rec[2] = blockIdx.x < count ? (unsigned)a.box_class[entry(blockIdx.x) * BOXES_PER_BRICK + threadIdx.y] : 99u; rec[3] = (n_upd & 0xffu) | ((__builtin_amdgcn_s_getreg(0xF814) & 0xfu) << 8) | (__builtin_amdgcn_s_getreg(0xF804) << 16);   // + XCC_ID, HW_ID (wave, SIMD, pipe, CU, SH, SE)
    }
#endif
    if (a.updated) room_count_add(n_upd, a.count_room);
}


static void load_mat(const float *p, MatS33 &m) {
    for (int r = 0; r < 3; ++r) {
        m.data[r].x = cfloat(p[r * 6 + 0], p[r * 6 + 1]);
        m.data[r].y = cfloat(p[r * 6 + 2], p[r * 6 + 3]);
        m.data[r].z = cfloat(p[r * 6 + 4], p[r * 6 + 5]);
    }
}
static void load_vec(const float *p, cfloat3 &v) { v.x = cfloat(p[0], p[1]); v.y = cfloat(p[2], p[3]); v.z = cfloat(p[4], p[5]); }


static void set_plane(Frustum &f, int P, const float T[3], const float M[3][3], float cxk, float cyk, float czk, float sl) {
    f.alpha[P] = cxk * T[0] + cyk * T[1] + czk * T[2];  // cxk*X + cyk*Y + czk*c >= -sl
    f.bx[P] = cxk * M[0][0] + cyk * M[1][0] + czk * M[2][0];
    f.by[P] = cxk * M[0][1] + cyk * M[1][1] + czk * M[2][1];
    f.bz[P] = cxk * M[0][2] + cyk * M[1][2] + czk * M[2][2];
    f.slack[P] = sl;
}
static void host_frustum(IntegrateArgs &a, float slack_scale = 1.0f) {
    Frustum &f = a.fr;
    const float vs = a.voxel_size, fx = a.intr.fx, fy = a.intr.fy;
    const float ul = (1.5f - a.intr.cx) - 2.f, uh = ((a.dcols - 0.5f) - a.intr.cx + 1.0f) + 2.f;
    const float vl = (1.5f - a.intr.cy) - 2.f, vh = ((a.drows - 0.5f) - a.intr.cy + 1.0f) + 2.f;
    // camera coordinates (real parts): p_r = t_r + vs * (R_r0*i + R_r1*j + R_r2*k)
    const float T[3] = {a.t.x.re, a.t.y.re, a.t.z.re};
    const float M[3][3] = {{a.R.data[0].x.re * vs, a.R.data[0].y.re * vs, a.R.data[0].z.re * vs},
                           {a.R.data[1].x.re * vs, a.R.data[1].y.re * vs, a.R.data[1].z.re * vs},
                           {a.R.data[2].x.re * vs, a.R.data[2].y.re * vs, a.R.data[2].z.re * vs}};
    const float ext = (float)std::max(a.X, std::max(a.Y, a.Z));
    const float mag0 = fabsf(T[0]) + (fabsf(M[0][0]) + fabsf(M[0][1]) + fabsf(M[0][2])) * ext;
    const float mag1 = fabsf(T[1]) + (fabsf(M[1][0]) + fabsf(M[1][1]) + fabsf(M[1][2])) * ext;
    const float mag2 = fabsf(T[2]) + (fabsf(M[2][0]) + fabsf(M[2][1]) + fabsf(M[2][2])) * ext;
    const float rel = 2e-3f * slack_scale;
    set_plane(f, 0, T, M, 0.f, 0.f, 1.f, rel * mag2);                                      // c >= 0
    set_plane(f, 1, T, M, fx, 0.f, -ul, rel * (fabsf(fx) * mag0 + fabsf(ul) * mag2));      // fx*X >= ul*c
    set_plane(f, 2, T, M, -fx, 0.f, uh, rel * (fabsf(fx) * mag0 + fabsf(uh) * mag2));      // fx*X <= uh*c
    set_plane(f, 3, T, M, 0.f, fy, -vl, rel * (fabsf(fy) * mag1 + fabsf(vl) * mag2));      // fy*Y >= vl*c
    set_plane(f, 4, T, M, 0.f, -fy, vh, rel * (fabsf(fy) * mag1 + fabsf(vh) * mag2));      // fy*Y <= vh*c
    set_plane(f, 5, T, M, 0.f, 0.f, -1.f, rel * mag2);                                     // c <= cfar (cfar added on the device)
    // the column clip's form of the same planes (see ClipPlanes)
    ClipPlanes &c = a.cp;
    c.kinds = 0; c.far_scale = 1.0f;
    for (int p = 0; p < 6; ++p) {
        // a slope along z too small to divide by is dropped from the column clip; what it could
        // contribute over the whole column goes into the slack instead
        const float scale = fabsf(f.alpha[p]) + (fabsf(f.bx[p]) + fabsf(f.by[p])) * ext + f.slack[p] + (p == 5 ? 6.f : 0.f);
        if (fabsf(f.bz[p]) * ext <= 1e-6f * scale) {
            f.slack[p] += fabsf(f.bz[p]) * ext;
            c.A[p] = f.alpha[p] + f.slack[p]; c.B[p] = f.bx[p]; c.C[p] = f.by[p];
        } else {
            const double s = -1.0 / (double)f.bz[p];
            c.A[p] = (float)(((double)f.alpha[p] + (double)f.slack[p]) * s);
            c.B[p] = (float)(f.bx[p] * s); c.C[p] = (float)(f.by[p] * s);
            if (p == 5) c.far_scale = (float)s;
            c.kinds |= (f.bz[p] > 0.f ? 1 : 2) << (2 * p);
        }
    }
}

enum { TILE_ROOM_BYTES = 1 << 20 };   // the workspace's own tile table: images of up to 131 072 tiles (e.g. 4096 x 2048 pixels); larger ones take the exact walk everywhere
// workspace: 256-byte header | the update counts' room (COUNT_ROOM_WORDS words) | brick list (int per brick) | box classes (BOXES_PER_BRICK words per list entry) | the call's own depth tiles
static size_t workspace_bricks(const int *res, int nz) { return (size_t)div_up(res[0], BRICK_X) * div_up(res[1], BRICK_Y) * div_up(nz, 2); }   // room for 2-plane bricks
static size_t workspace_list_bytes(const int *res, int nz) { return (WS_LIST_OFFSET + workspace_bricks(res, nz) * sizeof(int) + 255) & ~(size_t)255; }
static size_t workspace_class_bytes(const int *res, int nz) { return (workspace_bricks(res, nz) * BOXES_PER_BRICK * sizeof(unsigned) + 255) & ~(size_t)255; }
// ... | the list in the order the integrate kernel takes it (k_classify_boxes), after the tile room
static size_t workspace_order_offset(const int *res, int nz) { return workspace_list_bytes(res, nz) + workspace_class_bytes(res, nz) + TILE_ROOM_BYTES; }
extern "C" size_t xs_integrate_workspace_bytes(const int *res, int nz) {
    if (!res || nz <= 0) return 0;
    return workspace_order_offset(res, nz) + ((workspace_bricks(res, nz) * sizeof(int) + 255) & ~(size_t)255);
}
// What xs_integrate_classify* last classified into a workspace: the pose and the slack its box classes were padded for, and everything else
// they depend on — the slab, the volume, the camera, the band, the tile table.  The integrate call with XS_INTEGRATE_LIST_IS_READY that follows
// on that workspace uses those classes only if all of that is ITS OWN and its pose lies within the slack (box_slack_covers) — checked here,
// whatever the caller says — and decides the boxes again with its own pose otherwise.  A process-wide table keyed by the workspace (round 6;
// it was a per-thread record holding workspace, pose and slack only): classifying on one host thread and integrating on another works, and
// a list classified for another slab, frame size or tile table of the same workspace is not mistaken for this call's.
struct ClassesAhead {
    const void *workspace, *tiles;
    float R18[18], t6[6], slack_scale, voxel_size, tranc_dist, intr[4];
    int res[3], z0, z1, rows, cols;
};
namespace {
std::mutex g_ca_mu;
ClassesAhead g_ca_tab[16];
int g_ca_n = 0, g_ca_next = 0;
void classes_ahead_forget(const void *workspace) {
    std::lock_guard<std::mutex> lk(g_ca_mu);
    for (int i = 0; i < g_ca_n; ++i) if (g_ca_tab[i].workspace == workspace) g_ca_tab[i].workspace = nullptr;
}
void classes_ahead_put(const ClassesAhead &r) {
    std::lock_guard<std::mutex> lk(g_ca_mu);
    for (int i = 0; i < g_ca_n; ++i) if (g_ca_tab[i].workspace == r.workspace || g_ca_tab[i].workspace == nullptr) { g_ca_tab[i] = r; return; }
    if (g_ca_n < 16) { g_ca_tab[g_ca_n++] = r; return; }
    g_ca_tab[g_ca_next] = r; g_ca_next = (g_ca_next + 1) % 16;   // (more than 16 workspaces with classes pending: the oldest record goes — its call classifies again)
}
bool classes_ahead_take(const void *workspace, ClassesAhead &out) {   // the record is consumed
    std::lock_guard<std::mutex> lk(g_ca_mu);
    for (int i = 0; i < g_ca_n; ++i)
        if (g_ca_tab[i].workspace == workspace) { out = g_ca_tab[i]; g_ca_tab[i].workspace = nullptr; return true; }
    return false;
}
}  // namespace

/* The two tiny launches xs_integrate_scaled wraps around its kernels, for a caller that takes them off its critical path
 * (xs_integrate_scaled_ex with XS_INTEGRATE_HEADER_IS_CLEAR | XS_INTEGRATE_NO_FOLD): the clear of the workspace's 256-byte
 * header (brick count + update-count slots) before the classification, and the fold of the update-count slots into
 * updated_dev after the integrate kernel. */
extern "C" int xs_integrate_workspace_clear(void *workspace, void *stream) {
    if (!workspace) return xs_set_error(hipErrorInvalidValue, "xs_integrate_workspace_clear: null pointer");
    XS_CHECK(hipMemsetAsync(workspace, 0, WS_LIST_OFFSET, (hipStream_t)stream));
    return 0;
}
extern "C" int xs_integrate_fold_counts(void *workspace, unsigned long long *updated_dev, void *stream) {
    if (!workspace || !updated_dev) return xs_set_error(hipErrorInvalidValue, "xs_integrate_fold_counts: null pointer");
    hipLaunchKernelGGL(k_fold_count, dim3(1), dim3(FOLD_COUNT_BLOCK), 0, (hipStream_t)stream, reinterpret_cast<unsigned *>((char *)workspace + WS_HEADER_BYTES), updated_dev);
    XS_CHECK(hipGetLastError());
    return 0;
}
// What a pose whose box classes a list classified with slack_scale still holds for may differ by from the list's pose, in camera-frame
// metres anywhere in the volume.  Sideways (X, Y) as much as the frustum planes' extra slack lets a pose move (2e-3 of the coordinate
// magnitudes per unit of slack_scale, doubled: 4.4 cm for the benchmark volume at slack 2 — the last ICP update of a scene that slides,
// like S1 along its wall, is centimetres): that only widens a box's pixel range (~8 px at 2.5 m).  Along the viewing axis (C) a third of
// it (6.6 mm): that one thickens the band round every surface in which boxes take the per-voxel walk.  xs_integrate_list_covers
// checks a pose against exactly these (bit 1 of its result).
static BoxSlack box_slack(const IntegrateArgs &a, float slack_scale) {
    const float vs = a.voxel_size, ext = (float)std::max(a.X, std::max(a.Y, a.Z));
    auto mag = [&](const cfloat3 &row, float t) { return fabsf(t) + (fabsf(row.x.re) + fabsf(row.y.re) + fabsf(row.z.re)) * vs * ext; };
    const float k = 2e-3f * (slack_scale - 1.0f);
    static const float lateral = exp_env_float("XS_BOX_SLACK_LATERAL", 2.0f);   // (tuning aids of an XS_EXPERIMENTS build: xs_env.h)
    static const float axial = exp_env_float("XS_BOX_SLACK_AXIAL", 0.3f);
    BoxSlack sl;
    sl.dX = lateral * k * mag(a.R.data[0], a.t.x.re); sl.dY = lateral * k * mag(a.R.data[1], a.t.y.re); sl.dC = axial * k * mag(a.R.data[2], a.t.z.re);
    return sl;
}
// largest camera-frame coordinate difference between two poses over the volume's voxels, per axis
static void pose_delta(const IntegrateArgs &l, const IntegrateArgs &f, const int *res, double d[3]) {
    const cfloat3 *lr = l.R.data, *fr = f.R.data;
    const float lt[3] = {l.t.x.re, l.t.y.re, l.t.z.re}, ft[3] = {f.t.x.re, f.t.y.re, f.t.z.re};
    for (int r = 0; r < 3; ++r)
        d[r] = fabs((double)lt[r] - ft[r]) + (fabs((double)lr[r].x.re - fr[r].x.re) * res[0] + fabs((double)lr[r].y.re - fr[r].y.re) * res[1] +
                                               fabs((double)lr[r].z.re - fr[r].z.re) * res[2]) * (double)l.voxel_size;
}
static bool box_slack_covers(const IntegrateArgs &l, const IntegrateArgs &f, const int *res, float slack_scale) {
    const BoxSlack sl = box_slack(l, slack_scale);
    double d[3];
    pose_delta(l, f, res, d);
    return d[0] <= 0.9 * sl.dX && d[1] <= 0.9 * sl.dY && d[2] <= 0.9 * sl.dC;   // (a tenth left for the float evaluation on the device)
}
// camera depth grows with z: the far bricks end the list, and the list is taken from that end (KF_FAR_FIRST)
static bool far_end_first(const IntegrateArgs &a) {
    static const char *env_order = exp_env_str("XS_INTEGRATE_ORDER");   // A/B aid: "near" / "far" force the list direction
    return env_order ? !strcmp(env_order, "far") : a.R.data[2].z.re > 0.0f;
}
// the workspace's header, primary list region and capacity
static void bind_workspace(IntegrateArgs &a, const int *res, int nz, void *workspace) {
    a.brick_count = (unsigned *)workspace;
    a.count_room = (unsigned *)((char *)workspace + WS_HEADER_BYTES);
    a.brick_list = (int *)((char *)workspace + WS_LIST_OFFSET);
    a.list_cap = (unsigned)workspace_bricks(res, nz); a.pair_word = PAIR_PRIMARY;
}
static void bind_classes(IntegrateArgs &a, const int *res, int nz, void *workspace) {
    a.box_class = reinterpret_cast<unsigned *>((char *)workspace + workspace_list_bytes(res, nz));
    a.probe_offset = workspace_class_bytes(res, nz) + TILE_ROOM_BYTES / 4;
}
// ... and the list k_classify_boxes has ordered, which the classes are filed under
static void bind_ordered_list(IntegrateArgs &a, const int *res, int nz, void *workspace) {
    bind_classes(a, res, nz, workspace);
    a.brick_list = reinterpret_cast<int *>((char *)workspace + workspace_order_offset(res, nz)); a.pair_word = PAIR_SECOND;
}
// k_classify_boxes for the list that is there (the primary one): classes and order go to the second pair and region, which a then names.
// done: completion event to ride on the dispatch, or null.
static bool launch_box_classes(IntegrateArgs &a, const int *res, int nz, void *workspace, const DepthTile *tiles, const BoxSlack &sl, hipStream_t st,
                               bool second_pair_is_clear, hipEvent_t done = nullptr) {
    if (!tiles) return false;
    a.dt = depth_tiles_view(tiles, a.drows, a.dcols);
    bind_classes(a, res, nz, workspace);
    // (a list classed a second time — XS_INTEGRATE_RECLASSIFY_BOXES — finds the first classification's counts there; behind a header clear the pair is zero)
    // (... and its class counters: words CLASS_COUNT_WORD .. + 6 hold the counters, the second pair between them — one fill)
    if (!second_pair_is_clear && hipMemsetAsync(a.brick_count + CLASS_COUNT_WORD, 0, 8 * sizeof(unsigned), st) != hipSuccess) return false;
    const int nb = a.bricks_x * a.bricks_y * a.bricks_z;
    BoxOrder ord;
    ord.list = reinterpret_cast<int *>((char *)workspace + workspace_order_offset(res, nz));
    const dim3 grid(div_up(nb, 8) < 2048 ? div_up(nb, 8) : 2048);
    if (done) hipExtLaunchKernelGGL(k_classify_boxes, grid, dim3(256), 0, st, nullptr, done, 0, a, sl, ord);
    else hipLaunchKernelGGL(k_classify_boxes, grid, dim3(256), 0, st, a, sl, ord);
    a.brick_list = ord.list; a.pair_word = PAIR_SECOND;
    return true;
}
// The classification of a launch (header clear): the bricks against the frustum, then — with a tile table — their boxes' classes and the
// list in order; true = a.box_class is being written.  done: completion event to ride on the last dispatch, or null.
static bool launch_classification(IntegrateArgs &a, const int *res, int nz, void *workspace, const DepthTile *tiles, const BoxSlack &sl, hipStream_t st,
                                  hipEvent_t done = nullptr) {
    const int nb = a.bricks_x * a.bricks_y * a.bricks_z;
    if (done && !tiles) hipExtLaunchKernelGGL(k_classify_bricks, dim3(div_up(nb, CLASSIFY_BRICKS_BLOCK)), dim3(CLASSIFY_BRICKS_BLOCK), 0, st, nullptr, done, 0, a);
    else hipLaunchKernelGGL(k_classify_bricks, dim3(div_up(nb, CLASSIFY_BRICKS_BLOCK)), dim3(CLASSIFY_BRICKS_BLOCK), 0, st, a);
    return launch_box_classes(a, res, nz, workspace, tiles, sl, st, true, done);
}
// the pose-dependent part of the arguments the classification needs (what xs_integrate_scaled_ex sets up for it)
// Where the streamed classes' numeric shortcuts hold (valid_slot_request has the derivation): the margin for this sensor, and whether the
// launch may class boxes EDGE / SPECKLE at all.  Needs a.drows / dcols / intr / R / t / X / Y / Z / voxel_size; host only.
static void stream_margins(IntegrateArgs &a) {
    const float E = fmaxf((float)a.dcols + fabsf(a.intr.cx), (float)a.drows + fabsf(a.intr.cy));
    int e = 0;
    (void)frexpf(E, &e);                                       // E = m 2^e, m in [0.5, 1): ulp(E) = 2^(e - 24)
    a.near_margin = fmaxf(ldexpf(1.0f, e - 21), XS_NEAR_MARGIN_MIN);   // 8 ulp(E)
    const double ext[3] = {(double)a.X * a.voxel_size, (double)a.Y * a.voxel_size, (double)a.Z * a.voxel_size};
    auto im_row = [&](const cfloat3 &r, const cfloat &t) { return fabs((double)r.x.im) * ext[0] + fabs((double)r.y.im) * ext[1] + fabs((double)r.z.im) * ext[2] + fabs((double)t.im); };
    const double imX = im_row(a.R.data[0], a.t.x), imY = im_row(a.R.data[1], a.t.y), imC = im_row(a.R.data[2], a.t.z);
    const double cmin = XS_EDGE_CMIN;
    const double bound = (double)E * (imC / cmin) * (imC / cmin) + fmax(fabs((double)a.intr.fx) * imX, fabs((double)a.intr.fy) * imY) * imC / (cmin * cmin);
    if (!(E < 8192.0f) || !(bound <= (double)a.near_margin / 16.0)) a.kflags |= KF_NO_TESTED_STREAM;   // (NaN: no)
}
static void classify_args(IntegrateArgs &a, int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18,
                          const float *tv2c6, float tranc_dist, int z0, int z1, const float *depth_max_dev) {
    memset(&a, 0, sizeof(a));
    a.drows = rows; a.dcols = cols;
    a.X = res[0]; a.Y = res[1]; a.Z = res[2]; a.z0 = z0; a.z1 = z1;
    a.tranc_dist = tranc_dist; a.tranc_dist_inv = 1.0f / tranc_dist;
    load_mat(Rv2c18, a.R); load_vec(tv2c6, a.t);
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.voxel_size = voxel_size; a.depth_max = depth_max_dev;
    host_frustum(a);
    stream_margins(a);
    static const int env_bz = exp_env_int("XS_BRICK_Z", 0);  // tuning aid (as in xs_integrate_scaled_ex)
    a.brick_z = (env_bz >= 2 && env_bz <= 64) ? env_bz : BRICK_Z;
    a.bricks_x = div_up(a.X, BRICK_X); a.bricks_y = div_up(a.Y, BRICK_Y); a.bricks_z = div_up(z1 - z0, a.brick_z);
}
/* The brick classification of an integrate call on its own, for a pose that is only NEARLY the one the call will be made with — the
 * orchestrator enqueues it behind the last ICP launch, with the pose that launch started from, so that the list is there when the
 * final pose is: the launch latency of the classification (and half of the integrate kernel's) leaves the frame's critical path.
 * The frustum's slack is multiplied by slack_scale (> 1): the list then holds every brick any pose covered by
 * xs_integrate_list_covers(..., this pose, slack_scale, that pose) would list.  Clears the workspace header first unless flags has
 * XS_INTEGRATE_HEADER_IS_CLEAR.  Follow with xs_integrate_scaled_ex(..., XS_INTEGRATE_LIST_IS_READY | XS_INTEGRATE_HEADER_IS_CLEAR, ...)
 * on the same stream, workspace, volume slab and image size.  No synchronisation. */
extern "C" int xs_integrate_classify(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6,
                                     float tranc_dist, int z0, int z1, const float *depth_max_dev, void *workspace, float slack_scale, unsigned flags,
                                     void *stream) {
    xs_integrate_opts o = {};   // the plain entry point: the list alone (no tile table: the integrate call decides the boxes itself)
    o.struct_bytes = sizeof(o); o.flags = flags; o.mailbox_slack = 2.0f;
    return xs_integrate_classify_ex(rows, cols, intr4, res, voxel_size, Rv2c18, tv2c6, tranc_dist, z0, z1, depth_max_dev, workspace, slack_scale, &o, stream);
}
extern "C" int xs_integrate_classify_ex(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6,
                                        float tranc_dist, int z0, int z1, const float *depth_max_dev, void *workspace, float slack_scale,
                                        const xs_integrate_opts *opts, void *stream) {
    if (!opts || opts->struct_bytes != sizeof(xs_integrate_opts)) return xs_set_error(hipErrorInvalidValue, "xs_integrate_classify_ex: opts->struct_bytes is not sizeof(xs_integrate_opts)");
    const unsigned flags = opts->flags;
    const DepthTile *depth_tiles = static_cast<const DepthTile *>(opts->depth_tiles);
    if (!intr4 || !res || !Rv2c18 || !tv2c6 || !workspace || !(slack_scale >= 1.0f))
        return xs_set_error(hipErrorInvalidValue, "xs_integrate_classify: bad argument");
    if (z0 < 0 || z1 > res[2] || z1 <= z0 || res[0] <= 0 || res[1] <= 0) return xs_set_error(hipErrorInvalidValue, "xs_integrate_classify: bad slab");
    IntegrateArgs a;
    classify_args(a, rows, cols, intr4, res, voxel_size, Rv2c18, tv2c6, tranc_dist, z0, z1, depth_max_dev);
    if (!(a.bricks_x <= 1024 && a.bricks_y <= 1024 && a.bricks_z <= 2047)) return xs_set_error(hipErrorInvalidValue, "xs_integrate_classify: volume too large for the brick list");
    for (int p = 0; p < 6; ++p) a.fr.slack[p] *= slack_scale;
    bind_workspace(a, res, z1 - z0, workspace);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & XS_INTEGRATE_HEADER_IS_CLEAR)) XS_CHECK(hipMemsetAsync(a.brick_count, 0, WS_LIST_OFFSET, st));
    // the boxes' classes, valid for every pose xs_integrate_list_covers accepts for this list (needs the frame's tile table:
    // xs_integrate_set_depth_tiles; without it the integrate call classifies with its own pose)
    static const bool env_no_tiles = exp_env_set("XS_INTEGRATE_NO_TILES");
    classes_ahead_forget(workspace);
    // (the 32-bit-offset condition on the tightest pitch: the integrate call tests the real one and decides the boxes itself when it disagrees)
    const bool off32 = ((size_t)a.brick_z * a.Y + BRICK_Y) * ((size_t)res[0] * 4) < (1ull << 32);
    const bool boxes = !env_no_tiles && !(flags & XS_INTEGRATE_NO_TILES) && off32 && depth_tiles;
    if (flags & XS_INTEGRATE_COUNT_CLASSES) a.kflags |= KF_COUNT_CLASSES;
    // (the caller's completion event — xs_integrate_set_classify_event — rides on the dispatch)
    if (launch_classification(a, res, z1 - z0, workspace, boxes ? depth_tiles : nullptr, box_slack(a, slack_scale), st, (hipEvent_t)opts->stop_event)) {
        ClassesAhead r;
        r.workspace = workspace; r.tiles = depth_tiles; r.slack_scale = slack_scale; r.voxel_size = voxel_size; r.tranc_dist = tranc_dist;
        memcpy(r.R18, Rv2c18, sizeof(r.R18)); memcpy(r.t6, tv2c6, sizeof(r.t6)); memcpy(r.intr, intr4, sizeof(r.intr)); memcpy(r.res, res, sizeof(r.res));
        r.z0 = z0; r.z1 = z1; r.rows = rows; r.cols = cols;
        classes_ahead_put(r);
    }
    XS_CHECK(hipGetLastError());
    return 0;
}
/* Host only: 1 if a list classified with (Rv2c18_list, tv2c6_list, slack_scale) holds every brick the classification of
 * (Rv2c18, tv2c6) with the standard slack would — each half-space of the second pose, anywhere in the volume, lies inside the
 * first one's widened half-space — else 0 (then classify again: xs_integrate_scaled_ex without XS_INTEGRATE_LIST_IS_READY). */
extern "C" int xs_integrate_list_covers(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18_list,
                                        const float *tv2c6_list, float slack_scale, const float *Rv2c18, const float *tv2c6) {
    if (!intr4 || !res || !Rv2c18_list || !tv2c6_list || !Rv2c18 || !tv2c6) return 0;
    IntegrateArgs l, f;
    classify_args(l, rows, cols, intr4, res, voxel_size, Rv2c18_list, tv2c6_list, 1.0f, 0, res[2], nullptr);
    classify_args(f, rows, cols, intr4, res, voxel_size, Rv2c18, tv2c6, 1.0f, 0, res[2], nullptr);
    for (int p = 0; p < 6; ++p) {
        // a brick passes plane p when max over the brick of (alpha + b . index) >= -1.5 slack (box_may_pass); the two maxima differ
        // by at most the largest pointwise difference of the two affine forms over the volume
        const double d = fabs((double)l.fr.alpha[p] - f.fr.alpha[p]) + fabs((double)l.fr.bx[p] - f.fr.bx[p]) * res[0] +
                         fabs((double)l.fr.by[p] - f.fr.by[p]) * res[1] + fabs((double)l.fr.bz[p] - f.fr.bz[p]) * res[2];
        const double room = 1.5 * ((double)slack_scale * l.fr.slack[p] - f.fr.slack[p]);
        if (!(d <= 0.9 * room)) return 0;   // (a tenth of the room left for the kernel's float evaluation of the forms)
    }
    const bool tested_ok = !(f.kflags & KF_NO_TESTED_STREAM) || (l.kflags & KF_NO_TESTED_STREAM);   // (stream_margins: EDGE / SPECKLE classes need the final pose to allow them)
    return box_slack_covers(l, f, res, slack_scale) && tested_ok ? 3 : 1;   // bit 1: the boxes' classes hold for the pose too
}
/* Host only: the stricter cover test a POSTED integrate launch needs — it keeps the list pose's widened planes for its column clip too, where
 * a voxel is kept when alpha + b . index >= -slack (the brick test of xs_integrate_list_covers allows 1.5 slack): 1 if every half-space of
 * (Rv2c18, tv2c6), anywhere in the volume, lies inside that of (Rv2c18_list, tv2c6_list) widened by slack_scale.  Implies xs_integrate_list_covers. */
extern "C" int xs_integrate_pose_covered(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18_list,
                                         const float *tv2c6_list, float slack_scale, const float *Rv2c18, const float *tv2c6) {
    if (!intr4 || !res || !Rv2c18_list || !tv2c6_list || !Rv2c18 || !tv2c6) return 0;
    IntegrateArgs l, f;
    classify_args(l, rows, cols, intr4, res, voxel_size, Rv2c18_list, tv2c6_list, 1.0f, 0, res[2], nullptr);
    classify_args(f, rows, cols, intr4, res, voxel_size, Rv2c18, tv2c6, 1.0f, 0, res[2], nullptr);
    for (int p = 0; p < 6; ++p) {
        const double d = fabs((double)l.fr.alpha[p] - f.fr.alpha[p]) + fabs((double)l.fr.bx[p] - f.fr.bx[p]) * res[0] +
                         fabs((double)l.fr.by[p] - f.fr.by[p]) * res[1] + fabs((double)l.fr.bz[p] - f.fr.bz[p]) * res[2];
        const double room = (double)slack_scale * l.fr.slack[p] - f.fr.slack[p];
        if (!(d <= 0.8 * room)) return 0;   // (a fifth of the room left for the float evaluation of the forms and of the roots the clip solves them for)
    }
    const bool tested_ok = !(f.kflags & KF_NO_TESTED_STREAM) || (l.kflags & KF_NO_TESTED_STREAM);
    return box_slack_covers(l, f, res, slack_scale) && tested_ok ? 1 : 0;   // (a posted launch cannot classify again: it needs both)
}
extern "C" int xs_integrate_scaled(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                                   const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist,
                                   float *value, int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                                   unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, void *stream) {
    return xs_integrate_scaled_ex(depth_scaled, scaled_step, rows, cols, intr4, max_weight, res, voxel_size, Rv2c18, tv2c6, tranc_dist, value, weight,
                                  grad, vol_step, threshold, z0, z1, updated_dev, depth_max_dev, workspace, 0u, stream);
}
extern "C" int xs_integrate_scaled_ex(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                                      const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist,
                                      float *value, int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                                      unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, unsigned flags, void *stream) {
    xs_integrate_opts o = {};   // the plain entry point: flags alone
    o.struct_bytes = sizeof(o); o.flags = flags; o.mailbox_slack = 2.0f;
    return xs_integrate_scaled_ex2(depth_scaled, scaled_step, rows, cols, intr4, max_weight, res, voxel_size, Rv2c18, tv2c6, tranc_dist, value, weight, grad,
                                   vol_step, threshold, z0, z1, updated_dev, depth_max_dev, workspace, &o, stream);
}
extern "C" int xs_integrate_scaled_ex2(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                                       const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist,
                                       float *value, int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                                       unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, const xs_integrate_opts *opts,
                                       void *stream) {
    if (!opts || opts->struct_bytes != sizeof(xs_integrate_opts)) return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled_ex2: opts->struct_bytes is not sizeof(xs_integrate_opts)");
    const unsigned flags = opts->flags;
    const hipEvent_t ev0 = (hipEvent_t)opts->start_event, ev1 = (hipEvent_t)opts->stop_event;
    const DepthTile *depth_tiles = static_cast<const DepthTile *>(opts->depth_tiles);
    if (!depth_scaled || !intr4 || !res || !Rv2c18 || !tv2c6 || !value || !weight || !grad)
        return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled: null pointer");
    if ((flags & (XS_INTEGRATE_HEADER_IS_CLEAR | XS_INTEGRATE_NO_FOLD | XS_INTEGRATE_LIST_IS_READY)) && !workspace)
        return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled_ex: the flags concern the workspace path");
    if (z0 < 0 || z1 > res[2] || z1 < z0 || (vol_step % 4) != 0 || vol_step < (size_t)res[0] * 4)
        return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled: bad slab or pitch");
    hipStream_t st = (hipStream_t)stream;
    if (z1 == z0 || res[0] == 0 || res[1] == 0) {   // nothing to launch: the caller's timing / completion events are recorded all the same
        if (ev0) XS_CHECK(hipEventRecord(ev0, st));
        if (ev1) XS_CHECK(hipEventRecord(ev1, st));
        return 0;
    }
    IntegrateArgs a;
    a.depth = depth_scaled; a.dstep = scaled_step; a.drows = rows; a.dcols = cols;
    a.value = value; a.weight = weight; a.grad = grad; a.vstep = vol_step;
    a.X = res[0]; a.Y = res[1]; a.Z = res[2]; a.z0 = z0; a.z1 = z1;
    a.tranc_dist = tranc_dist; a.tranc_dist_inv = 1.0f / tranc_dist; a.max_weight = max_weight;
    load_mat(Rv2c18, a.R); load_vec(tv2c6, a.t);
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.voxel_size = voxel_size; a.threshold = threshold; a.updated = updated_dev; a.depth_max = depth_max_dev;
    a.brick_list = nullptr; a.brick_count = nullptr; a.count_room = nullptr; a.kflags = 0;
    static const bool env_always = exp_env_set("XS_INTEGRATE_ALWAYS_STORE");   // measurement aid, as the flag
    if (env_always || (flags & XS_INTEGRATE_ALWAYS_STORE)) a.kflags |= KF_ALWAYS_STORE;
    if (flags & XS_INTEGRATE_COUNT_CLASSES) a.kflags |= KF_COUNT_CLASSES;
    if (far_end_first(a)) a.kflags |= KF_FAR_FIRST;
    stream_margins(a);
    const bool posted = (flags & XS_INTEGRATE_POSE_POSTED) != 0;
    a.mailbox = nullptr; a.mailbox_seq = 0; a.pose_dev = nullptr;
    a.dt = depth_tiles_view(nullptr, rows, cols); a.box_class = nullptr;
    a.signmap = static_cast<unsigned char *>(opts->signmap);   // (a slab launch marks the bricks of its own planes: the map is indexed by whole-volume coordinates)
    if (posted) {
        if (!workspace || !(flags & XS_INTEGRATE_LIST_IS_READY) || !opts->pose_mailbox || !opts->pose_dev)
            return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled_ex: a posted launch needs the classified list and a mailbox (xs_integrate_set_pose_mailbox)");
        a.mailbox = (const unsigned *)opts->pose_mailbox; a.mailbox_seq = opts->mailbox_seq; a.pose_dev = (unsigned *)opts->pose_dev;
    }
    host_frustum(a, posted ? opts->mailbox_slack : 1.0f);
    const int nz = z1 - z0;
    static const int env_bz = exp_env_int("XS_BRICK_Z", 0);  // tuning aid
    a.brick_z = (env_bz >= 2 && env_bz <= 64) ? env_bz : BRICK_Z;
    a.bricks_x = div_up(a.X, BRICK_X); a.bricks_y = div_up(a.Y, BRICK_Y); a.bricks_z = div_up(nz, a.brick_z);
    a.zchunk = nz;
    dim3 block(64, 4);
    // The brick kernel addresses a brick's voxels as a wave-uniform base (the brick's first voxel) + one 32-bit byte offset per lane: a brick
    // must span less than 4 GiB of an array — always, short of absurd pitches (rows of more than 128 K floats at Y = 1024); such a volume takes
    // the column walk below.  (Round 6: the brick kernel's 64-bit-pointer instances, which no test could reach, are gone.)
    const bool off32 = ((size_t)a.brick_z * a.Y + BRICK_Y) * a.vstep < (1ull << 32);
    if (workspace && off32 && a.bricks_x <= 1024 && a.bricks_y <= 1024 && a.bricks_z <= 2047) {  // packed brick ids: 10 + 10 + 11 bits
        bind_workspace(a, res, nz, workspace);
        const int nb = a.bricks_x * a.bricks_y * a.bricks_z;
        const bool sign = a.signmap != nullptr;
        // the boxes' classes (free space / nothing to write / exact walk) and the list's order: those xs_integrate_classify left for this
        // list, or decided here with the launch's own pose — from the caller's tile table (xs_integrate_set_depth_tiles) or one built here,
        // in the workspace
        static const bool env_no_tiles = exp_env_set("XS_INTEGRATE_NO_TILES");   // A/B aid, as the flag
        // the classes xs_integrate_classify* left hold for this launch only if its pose lies within the slack they were padded for: checked
        // here (a posted launch is handed the list's own pose and receives a covered one through its mailbox: xs_integrate_pose_covered)
        ClassesAhead ca;
        bool classes_ahead = (flags & XS_INTEGRATE_LIST_IS_READY) && classes_ahead_take(workspace, ca) && !(flags & XS_INTEGRATE_RECLASSIFY_BOXES);
        if (classes_ahead)   // ... for this slab of this volume, this camera and band, this frame's tile table?
            classes_ahead = ca.z0 == z0 && ca.z1 == z1 && !memcmp(ca.res, res, sizeof(ca.res)) && ca.rows == rows && ca.cols == cols && !memcmp(ca.intr, intr4, sizeof(ca.intr)) &&
                            ca.voxel_size == voxel_size && ca.tranc_dist == tranc_dist && (ca.tiles == depth_tiles || posted);
        if (classes_ahead && !posted) {
            IntegrateArgs l = a;
            load_mat(ca.R18, l.R); load_vec(ca.t6, l.t);
            // (classes decided for a pose whose imaginary parts passed stream_margins do not hold for one whose do not)
            classes_ahead = box_slack_covers(l, a, res, ca.slack_scale) && !(a.kflags & KF_NO_TESTED_STREAM);
        }
        const bool use_tiles = !env_no_tiles && !(flags & XS_INTEGRATE_NO_TILES);
        auto tile_table = [&]() -> const DepthTile * {   // the caller's, or one built in the workspace's tile room
            if (depth_tiles || xs_depth_tiles_bytes(rows, cols) > TILE_ROOM_BYTES) return depth_tiles;
            DepthTile *own = reinterpret_cast<DepthTile *>((char *)workspace + workspace_list_bytes(res, nz) + workspace_class_bytes(res, nz));
            launch_scale_depth(depth_scaled, scaled_step, rows, cols, (float *)nullptr, (size_t)0, (float *)nullptr, own, st);
            return own;
        };
        if (!(flags & XS_INTEGRATE_LIST_IS_READY)) {   // (else: xs_integrate_classify has run on this stream for a covering pose)
            if (!(flags & XS_INTEGRATE_HEADER_IS_CLEAR))
                XS_CHECK(hipMemsetAsync(a.brick_count, 0, WS_LIST_OFFSET, st));  // list pairs + the update counts' room: one fill
            launch_classification(a, res, nz, workspace, use_tiles ? tile_table() : nullptr, BoxSlack{0.f, 0.f, 0.f}, st);
        } else if (use_tiles) {
            if (classes_ahead) bind_ordered_list(a, res, nz, workspace);
            else if (!posted) launch_box_classes(a, res, nz, workspace, tile_table(), BoxSlack{0.f, 0.f, 0.f}, st, false);   // (a posted launch has no pose yet to classify with)
        }
        if (a.box_class) a.kflags &= ~(unsigned)KF_FAR_FIRST;   // the list is ordered by class
        // resident workgroups stride over the list: 256 CUs x 8
        static const int env_g = exp_env_int("XS_BRICK_GRID", 0);
        const int gmax = env_g > 0 ? env_g : 8192;
        const int g = nb < gmax ? nb : gmax;
        // profiling: the event pair rides on the dispatch packet itself (hipExtLaunchKernelGGL: start / stop are
        // the kernel's own begin / end timestamps), so it adds no marker packets to the stream and times what
        // rocprofv3 times
        // (either event may be null: a completion event alone lets another stream wait for this kernel without a marker packet)
        void (*kern)(const IntegrateArgs) = sign ? (threshold > 0.0f ? k_integrate_bricks<true, false, true> : k_integrate_bricks<false, false, true>)
                                                 : (threshold > 0.0f ? k_integrate_bricks<true> : k_integrate_bricks<false>);
        if (posted) {
            kern = sign ? (threshold > 0.0f ? k_integrate_bricks<true, true, true> : k_integrate_bricks<false, true, true>)
                        : (threshold > 0.0f ? k_integrate_bricks<true, true> : k_integrate_bricks<false, true>);
            hipLaunchKernelGGL(k_pose_gate, dim3(1), dim3(64), 0, st, a.mailbox, a.mailbox_seq, a.pose_dev);
        }
        static const int env_lds = exp_env_int("XS_INTEGRATE_DYN_LDS", 0);   // experiment: dynamic LDS bytes per workgroup = a cap on the workgroups resident per CU
        if (ev0 || ev1) hipExtLaunchKernelGGL(kern, dim3(g), block, env_lds, st, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(kern, dim3(g), block, env_lds, st, a);
        if (updated_dev && !(flags & XS_INTEGRATE_NO_FOLD))
            hipLaunchKernelGGL(k_fold_count, dim3(1), dim3(FOLD_COUNT_BLOCK), 0, st, a.count_room, updated_dev);
    } else {
        if (posted) return xs_set_error(hipErrorInvalidValue, "xs_integrate_scaled_ex: posted launch without a brick list");
        int gx = div_up(a.X, 64), gy = div_up(a.Y, 4), zsplit = 1;
        while ((long long)gx * gy * zsplit < 4096 && zsplit < nz && nz / (zsplit * 2) >= 16) zsplit *= 2;
        a.zchunk = div_up(nz, zsplit);
        dim3 grid(gx, gy, div_up(nz, a.zchunk));
        // (the events xs_integrate_set_timing_events handed over ride on this dispatch too: a caller that waits on the stop event — the
        // orchestrator's auxiliary stream does — must find it recorded whichever kernel ran)
        void (*kern)(const IntegrateArgs) = a.signmap ? (threshold > 0.0f ? k_integrate<true, true> : k_integrate<false, true>)
                                                      : (threshold > 0.0f ? k_integrate<true> : k_integrate<false>);
        if (ev0 || ev1) hipExtLaunchKernelGGL(kern, grid, block, 0, st, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(kern, grid, block, 0, st, a);
    }
    XS_CHECK(hipGetLastError());
    return 0;
}

// The reference's launcher (TsdfFusion.cu:173-201): u16 millimetres in, scale then integrate.
// depth_scaled is caller-owned workspace (rows x cols floats) — the reference allocates and
// frees it on every call (:180,:198); here it is resident.
extern "C" int xs_integrate_tsdf_volume(const uint16_t *depth, size_t depth_step, int rows, int cols, const float *intr4, int max_weight,
                                        const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist,
                                        float *value, int *weight, float *grad, size_t vol_step, float *depth_scaled,
                                        size_t scaled_step, float threshold, int z0, int z1, unsigned long long *updated_dev,
                                        void *stream) {
    int rc = xs_scale_depth(depth, depth_step, rows, cols, depth_scaled, scaled_step, stream);
    if (rc) return rc;
    return xs_integrate_scaled(depth_scaled, scaled_step, rows, cols, intr4, max_weight, res, voxel_size, Rv2c18, tv2c6, tranc_dist, value,
                               weight, grad, vol_step, threshold, z0, z1, updated_dev, nullptr, nullptr, stream);
}

// ==========================================================================================
// Dual-complex Hessian kernel and its real-valued twin.  Replaces TsdfFusion.cu:204-283
// (ComputeLocalTsdfHessianKernel) + :286-331 (ComputeLocalTsdf_hessian) and :335-410 + :412-447
// (ComputeLocalTsdfLossKernel / ComputeLocalTsdf_loss).
//
// The reference writes four N^3 scratch volumes (value / grad / hessian / count) and then runs
// four thrust::reduce passes over them: five full-volume passes.  Here the per-voxel terms are
// folded on chip — lane registers, wave shuffles, one LDS exchange, one record per workgroup,
// and the last workgroup adds the records in index order — so the only N^3 traffic is the 4 B
// per voxel of the dense ground-truth TSDF (algorithmic bytes 4*N^3 + 2*W*H).  Sums are kept
// in double and rounded once; thrust's float tree order is unspecified in the reference.
struct HessArgs {
    const float *depth; size_t dstep; int drows, dcols;
    int X, Y, Z, z0, z1, zchunk;
    float voxel_size, tranc_dist, tranc_dist_inv;
    Intr intr;
    const float *gt;      // dense, unpitched: index z*Y*X + y*X + x, storage starts at z0
    double *partials;     // [blocks][8]
    unsigned *ticket;
    double *out;          // hessian: {loss, grad, hessian, count}; loss: {loss, count}
    float *real_out, *grad_out, *hess_out; int *count_out;  // optional per-voxel volumes (same indexing as gt)
    int tiles_x, tiles_y, tiles_z;  // (64 x 4 x zchunk) tiles — (256 x 4 x zchunk) when wide; workgroups stride over them
    double *publish; unsigned long long publish_seq;   // optional, host-coherent pinned memory: the last workgroup stores the sums there + the word [32] = seq
    const unsigned *mailbox; unsigned mailbox_seq;     // k_tsdf_gauss_newton<true>: the six poses arrive through a mailbox (xs_gn_post_poses)
    int il;               // wide == 2: consecutive planes a workgroup takes together before its neighbours' (1, 2, 4, 8)
    int wide;             // a lane scans four x-neighbours with 16-byte loads (X % 4 == 0 and gt 16-byte aligned): for_band_voxels_wide; 2: planes interleaved
};
struct HessPoseD { MatD33 R; dcfloat3 t; };
struct HessPoseF { float R[9]; float t[3]; };

// The arrival ticket: bits 0 .. 15 count the workgroups that arrived (at most XS_TSDF_REDUCE_MAX_BLOCKS = 4096 per launch), bits 16 .. 31 those of
// them that LEFT without summing (k_tsdf_gauss_newton<true>: told to, or their poses never came).  Every workgroup of a launch arrives exactly once,
// whatever it did, so the last one always exists: it puts the ticket back to zero and publishes — the sums, or, if any workgroup left (some may
// have seen their poses just before the deadline and others not), the sequence word with bit 63 set and no sums.
enum { XS_TSDF_REDUCE_MAX_BLOCKS_C = 4096 };   // workgroups per launch (they stride over the tiles); records of up to 32 doubles
enum : unsigned { TICKET_ARRIVED = 1u, TICKET_LEFT = 0x10000u, TICKET_COUNT_MASK = 0xffffu };
template <int NV>
__device__ __forceinline__ void block_fold_and_finish(double (&v)[NV], double *partials, unsigned *ticket, double *out, double *publish = nullptr,
                                                      unsigned long long publish_seq = 0, bool left = false) {
    constexpr int STRIDE = NV <= 8 ? 8 : 32;  // doubles per workgroup record
    __shared__ double sm[4][NV];
    __shared__ unsigned s_last;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned nblocks = gridDim.x * gridDim.y * gridDim.z;
    const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (!left) {   // (workgroup-uniform)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double s = wave_sum_f64(v[k]);
            if (lane == 0) sm[wave][k] = s;
        }
        __syncthreads();
        if (tid < NV) {
            const double s = ((sm[0][tid] + sm[1][tid]) + sm[2][tid]) + sm[3][tid];
            __hip_atomic_store(&partials[(size_t)bid * STRIDE + tid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // The record went out with agent-scope write-through stores from lanes of wave 0 (NV <= 64): once they are acknowledged the
    // record is in memory, and the same wave's first lane takes the ticket — no release fence, whose write-back of the whole
    // L2 per workgroup is what used to cap the grid at 1024 workgroups (xs_icp.hip has the measurements)
    static_assert(NV <= 64, "the record is stored by one wave");
    static_assert(XS_TSDF_REDUCE_MAX_BLOCKS_C <= (int)TICKET_COUNT_MASK, "the ticket counts arrivals in sixteen bits");
    if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        const unsigned tk = __hip_atomic_fetch_add(ticket, left ? TICKET_ARRIVED + TICKET_LEFT : TICKET_ARRIVED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk & TICKET_COUNT_MASK) != nblocks - 1 ? 0u : ((tk >> 16) != 0 || left ? 2u : 1u);   // 2: the last one of a launch some workgroup left
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (s_last == 2u) {
        if (tid == 0) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (publish) __hip_atomic_store(reinterpret_cast<unsigned long long *>(publish) + 32, publish_seq | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (s_last) {
        // After the acquire + barrier plain loads see every record.  All 256 threads take part: thread (g, c)
        // adds column c of records g, g + G, g + 2G, ... in that order with 16 loads in flight, and the G row
        // groups are then added in group order — fixed association, deterministic.  (One wave reading the
        // records one dependent load at a time took longer than the rest of the kernel.)
        constexpr int G = 256 / STRIDE;
        __shared__ double s_red[G][STRIDE];
        const int c = tid % STRIDE, g = tid / STRIDE;
        const double *p = partials + c;
        double s = 0.0;
        unsigned b = g;
        for (; b + G * 15 < nblocks; b += G * 16) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(b + G * k) * STRIDE];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += v[k];
        }
        for (; b < nblocks; b += G) s += p[(size_t)b * STRIDE];
        s_red[g][c] = s;
        __syncthreads();
        if (tid < NV) {
            double t = s_red[0][tid];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) t += s_red[gg][tid];
            out[tid] = t;
            // the host's copy: system-scope stores into pinned memory by lanes of wave 0 (NV <= 64), then — once they are acknowledged — the
            // sequence word by its first lane: a host that sees the word sees the sums (the protocol of the ICP records, xs_icp.hip)
            if (publish) __hip_atomic_store(&publish[tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // the ticket goes back to zero for the next launch on this workspace (xs_tsdf_reduce_workspace_init zeroes it once): no fill per launch
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (publish && wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(publish) + 32, publish_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The dense ground-truth read is the kernels' only N^3 traffic, and a plain z loop keeps one 4-byte load
// per lane in flight (measured: 0.85 TB/s for the Hessian kernel at 512^3).  Here a lane requests thirty-two
// planes of its column at once and keeps a bit per plane whose voxel is in the band (gt != 0, |gt| <= 0.95).
// The band is a sheet: where it lies across the columns every lane holds a handful of band voxels, but where it
// runs ALONG them (a wall parallel to the z axis) six lanes of a wave hold thirty-two each and the other
// fifty-eight none — a wave that let every lane work through its own voxels ran the dual-complex body at a tenth
// of its lanes (the relocalisation pass of a box room at 1024^3: 4.7 ms against 1.4 ms for a wall across z).  So
// the voxels are dealt out again: plane by plane, the lanes that hold a band voxel append its coordinates to a
// per-wave queue in LDS (one ballot + one prefix count per plane), and whenever sixty-four are waiting every lane
// takes one — full lanes whatever the sheet's orientation; the queue runs on across chunks and tiles and is
// drained once at the end.  The order in which a lane's double sums meet their terms differs from the reference's
// thrust::reduce (unspecified there) by association only.
// (Scanning the slab as one flat array — contiguous 8 KB per wave — streamed only 6 % faster; four columns per
// lane with 16-byte loads no faster either.)
struct BandQueue {
    enum { CAP = 128 };                 // entries per wave: at most 63 left over + 64 appended
    unsigned long long (*q)[CAP];       // [wave][CAP] in LDS: x | y << 21 | z << 42
    unsigned head, tail;
};
template <class F>
__device__ __forceinline__ void band_queue_take(const HessArgs &a, BandQueue &Q, int wave, int lane, unsigned count, F &&body) {
    // lanes 0 .. count - 1 take the entries head .. head + count - 1 (the appends are this wave's own: LDS operations of
    // a wave complete in order, and the compiler may not move memory accesses across the asm)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((unsigned)lane < count) {
        const unsigned long long e = Q.q[wave][(Q.head + lane) % BandQueue::CAP];
        const int x = (int)(e & 0x1fffff), y = (int)((e >> 21) & 0x1fffff), z = (int)(e >> 42);
        const size_t index = ((size_t)(z - a.z0) * a.Y + y) * a.X + x;
        body(x, y, z, index, a.gt[index]);   // (re-read: a cache hit, instead of thirty-two live registers)
    }
    Q.head += count;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before a later append may reuse the slots
}
// the queue's side of the scan: the lanes whose voxel (bit b of their mask) is in the band append it; sixty-four waiting are dealt out
template <class F, class XYZ>
__device__ __forceinline__ void band_queue_append(const HessArgs &a, BandQueue &Q, int wave, int lane, bool mine, XYZ &&xyz, F &&body) {
    const unsigned long long who = __ballot(mine);
    if (!who) return;
    if (mine) {
        const unsigned pos = Q.tail + __popcll(who & ((1ull << lane) - 1ull));
        Q.q[wave][pos % BandQueue::CAP] = xyz();
    }
    Q.tail += (unsigned)__popcll(who);
    if (Q.tail - Q.head >= 64u) band_queue_take(a, Q, wave, lane, 64u, body);
}
// The ground truth is read ONCE per launch and is larger than the 256 MiB Infinity Cache (512 MiB at 512^3, 4 GiB at 1024^3): the scan's
// loads are NONTEMPORAL (round 6, profiles/r06_hess_scan.txt).  With the default policy every line read is allocated in the cache, and
// allocating evicts — what the predecessor left dirty first: the same scan streamed 3.0-3.8 TB/s behind a kernel that had written 1 GiB and
// 4.1-5.9 TB/s back to back, against 5.9-6.4 TB/s either way without allocation.
typedef float xs_f4 __attribute__((ext_vector_type(4)));
template <class F>
__device__ __forceinline__ void for_band_voxels(const HessArgs &a, BandQueue &Q, int wave, int lane, int x, int y, bool in_volume, int zb, int ze,
                                                F &&body) {
    constexpr int ZB = 32;
    const size_t plane = (size_t)a.Y * a.X;
    const size_t first = in_volume ? (size_t)(zb - a.z0) * plane + (size_t)y * a.X + x : 0;
    for (int zc = zb; zc < ze; zc += ZB) {
        const float *col = a.gt + first + (size_t)(zc - zb) * plane;
        unsigned mask = 0;
        if (in_volume) {
#pragma unroll
            for (int j = 0; j < ZB; ++j) {
                const float g = (zc + j < ze) ? col[(size_t)j * plane] : 0.f;   // (the fallback for rows that are no multiple of 16 bytes: as round 5 measured it)
                if (!(g == 0 || fabsf(g) > 0.95)) mask |= 1u << j;
            }
        }
        if (!__ballot(mask != 0)) continue;                       // free space: the usual case
        for (int j = 0; j < ZB; ++j)
            band_queue_append(a, Q, wave, lane, (mask >> j) & 1u,
                              [&] { return (unsigned long long)x | ((unsigned long long)y << 21) | ((unsigned long long)(zc + j) << 42); }, body);
    }
}
// Round 6: sixteen bytes per lane.  A lane takes FOUR x-neighbours (x0 .. x0 + 3: a wave reads 1 KiB of a row per instruction, a
// workgroup four rows) and requests eight planes at once — the same 32 registers and 32 mask bits as the column form, bit 4 j + k =
// voxel (x0 + k, y, zc + j), dealt out through the same queue.  Scan alone (a map with one wall, nothing but the read, the band test and
// the ballots; profiles/tools/scan_probe.hip on one MI355X): 4.4-4.6 TB/s for the column form at 512^3 (5.1-5.2 at 1024^3), 5.9 (5.4-5.9)
// for this shape, 6.2-6.3 (6.4) for this shape with nontemporal loads; a flat sweep of the array, which knows no coordinates: 6.5 (6.7).
template <class F>
__device__ __forceinline__ void for_band_voxels_wide(const HessArgs &a, BandQueue &Q, int wave, int lane, int x0, int y, bool in_volume, int zfirst, int zstep, int zend,
                                                     F &&body) {
    // this tile's planes: zfirst, zfirst + zstep, ... below a.z1 — INTERLEAVED with the other workgroups that share its columns.  A band is
    // a sheet a few planes thick: cut into runs of consecutive planes, a wall across z would put all of its voxels into the one run that
    // holds it (1 workgroup in 16 at 512^3 did the whole dual-complex evaluation: 0.145 ms against 0.117); plane by plane it goes to seven.
    constexpr int ZB = 8;
    const size_t plane = (size_t)a.Y * a.X;
    const int il = a.il;                                       // consecutive planes taken together (1, 2, 4 or 8); zstep counts such groups
    auto zof = [&](int zc, int j) { return zc + (j % il) + (j / il) * il * zstep; };
    const float *col = a.gt + (in_volume ? (size_t)(zfirst - a.z0) * plane + (size_t)y * a.X + x0 : 0);
    for (int zc = zfirst; zc < zend; zc += ZB * zstep, col += (size_t)ZB * zstep * plane) {
        unsigned mask = 0;
        if (in_volume) {
            xs_f4 v[ZB];
#pragma unroll
            for (int j = 0; j < ZB; ++j)
                v[j] = (zof(zc, j) < zend) ? __builtin_nontemporal_load(reinterpret_cast<const xs_f4 *>(col + (size_t)(zof(zc, j) - zc) * plane)) : xs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < ZB; ++j) {
                if (!(v[j].x == 0 || fabsf(v[j].x) > 0.95)) mask |= 1u << (4 * j);
                if (!(v[j].y == 0 || fabsf(v[j].y) > 0.95)) mask |= 2u << (4 * j);
                if (!(v[j].z == 0 || fabsf(v[j].z) > 0.95)) mask |= 4u << (4 * j);
                if (!(v[j].w == 0 || fabsf(v[j].w) > 0.95)) mask |= 8u << (4 * j);
            }
        }
        if (!__ballot(mask != 0)) continue;                       // free space: the usual case
        for (int b = 0; b < 32; ++b)
            band_queue_append(a, Q, wave, lane, (mask >> b) & 1u,
                              [&] { return (unsigned long long)(x0 + (b & 3)) | ((unsigned long long)y << 21) | ((unsigned long long)zof(zc, b >> 2) << 42); }, body);
    }
}
// the tile walk the three kernels share: a bounded number of workgroups (each pays a ticket when it retires) stride over the
// (64 x 4 x zchunk) tiles — (256 x 4 x zchunk) in the wide form — one column (four) per lane, one row of columns per wave
template <class F>
__device__ __forceinline__ void walk_band(const HessArgs &a, F &&body) {
    __shared__ unsigned long long s_queue[4][BandQueue::CAP];
    BandQueue Q{s_queue, 0u, 0u};
    const int lane = threadIdx.x, wave = threadIdx.y;            // blockDim = (64, 4)
    const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z;
    const bool skew = a.wide == 2 && (gridDim.x % (unsigned)a.tiles_z) == 0;   // (else consecutive rounds already land in different z groups)
    for (int tile = blockIdx.x, round = 0; tile < ntiles; tile += gridDim.x, ++round) {
        if (a.wide == 2) {   // z group fastest: tile = column * G + g, planes a.z0 + (g + k G) il + (0 .. il - 1)
            const int G = a.tiles_z, column = tile / G, g = (tile % G + (skew ? round : 0)) % G;
            const int x0 = 4 * (int)threadIdx.x + (column % a.tiles_x) * 256, y = threadIdx.y + (column / a.tiles_x) * 4;
            for_band_voxels_wide(a, Q, wave, lane, x0, y, x0 < a.X && y < a.Y, a.z0 + g * a.il, G, a.z1, body);
            continue;
        }
        const int y = threadIdx.y + ((tile / a.tiles_x) % a.tiles_y) * 4;
        const int zb = a.z0 + (tile / (a.tiles_x * a.tiles_y)) * a.zchunk, ze = min(zb + a.zchunk, a.z1);
        if (a.wide) {
            const int x0 = 4 * (int)threadIdx.x + (tile % a.tiles_x) * 256;
            for_band_voxels_wide(a, Q, wave, lane, x0, y, x0 < a.X && y < a.Y, zb, 1, ze, body);
        } else {
            const int x = threadIdx.x + (tile % a.tiles_x) * 64;
            for_band_voxels(a, Q, wave, lane, x, y, x < a.X && y < a.Y, zb, ze, body);
        }
    }
    band_queue_take(a, Q, wave, lane, Q.tail - Q.head, body);   // what is left: fewer than sixty-four
}

// Three waves per SIMD (168 VGPRs; left alone the compiler takes 176 = two waves; four = 128 VGPRs spill): the band's dual-complex evaluation is
// VALU work that only another wave's scan can hide.  0.1035 -> 0.1005 ms at 512^3 alternating on one box; four waves 0.1215 (profiles/r06_hess_scan.txt 6).
#ifndef XS_HESS_WAVES_PER_EU
#define XS_HESS_WAVES_PER_EU 3
#endif
#define XS_HESS_OCC __attribute__((amdgpu_waves_per_eu(XS_HESS_WAVES_PER_EU, XS_HESS_WAVES_PER_EU)))
__global__ void __launch_bounds__(256) XS_HESS_OCC k_tsdf_hessian(const HessArgs a, const HessPoseD Pk) {
    __shared__ HessPoseD P;   // 48 floats of pose through LDS rather than through vector registers (see k_tsdf_gauss_newton)
    {
        const float *src = reinterpret_cast<const float *>(&Pk);
        float *dst = reinterpret_cast<float *>(&P);
        for (int i = threadIdx.y * 64 + threadIdx.x; i < (int)(sizeof(HessPoseD) / sizeof(float)); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    walk_band(a, [&](int xq, int yq, int z, size_t index, float gt) {
        const dcfloat gt_tsdf(gt);
        const dcfloat vgx((float(xq) + 0.5f) * a.voxel_size);
        const dcfloat vgy((float(yq) + 0.5f) * a.voxel_size);
        const dcfloat vgz((float(z) + 0.5f) * a.voxel_size);
        dcfloat3 v_g; v_g.x = vgx; v_g.y = vgy; v_g.z = vgz;
        dcfloat3 v_c;
        v_c.x = dot(P.R.data[0], v_g) + P.t.x;
        v_c.y = dot(P.R.data[1], v_g) + P.t.y;
        v_c.z = dot(P.R.data[2], v_g) + P.t.z;
        const dcfloat inv_z = dcfloat(1.0f) / v_c.z;
        if (inv_z.value() < 0) return;
        const dcfloat image_x = v_c.x * inv_z * a.intr.fx + a.intr.cx;
        const dcfloat image_y = v_c.y * inv_z * a.intr.fy + a.intr.cy;
        const int coo_x = __float2int_rd(image_x.value() - 0.5f), coo_y = __float2int_rd(image_y.value() - 0.5f);
        if (!(coo_x > 1 && coo_y > 1 && coo_x < a.dcols - 1 && coo_y < a.drows - 1)) return;
        const int near_x = __float2int_rn(image_x.value()), near_y = __float2int_rn(image_y.value());
        dcfloat Dp(row_ptr(a.depth, a.dstep, near_y)[near_x]);
        const dcfloat d00(row_ptr(a.depth, a.dstep, coo_y)[coo_x]), d10(row_ptr(a.depth, a.dstep, coo_y)[coo_x + 1]);
        const dcfloat d01(row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x]), d11(row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x + 1]);
        if (d00.value() != 0.0f && d01.value() != 0.0f && d10.value() != 0.0f && d11.value() != 0.0f) {  // :248-251, threshold unused
            const dcfloat one(1.0f);
            const dcfloat fa = image_x - dcfloat(float(coo_x) + 0.5f);
            const dcfloat fb = image_y - dcfloat(float(coo_y) + 0.5f);
            Dp = d00 * (one - fa) * (one - fb) + d10 * fa * (one - fb) + d01 * (one - fa) * fb + d11 * fa * fb;
        }
        if (Dp.value() > 5 || Dp.value() < 0.2) return;
        const dcfloat xl = (image_x - a.intr.cx) / a.intr.fx;
        const dcfloat yl = (image_y - a.intr.cy) / a.intr.fy;
        dcfloat3 v_c_1; v_c_1.x = Dp * xl; v_c_1.y = Dp * yl; v_c_1.z = Dp;
        const dcfloat distance = norm(v_c_1) - norm(v_c);
        const dcfloat gt_distance = gt_tsdf * a.tranc_dist;
        const dcfloat error = (distance - gt_distance) * a.tranc_dist_inv;
        if (fabsf(error.value()) > 1) return;
        const dcfloat loss = error * error;
        if (a.real_out) {
            a.real_out[index] = loss.value(); a.grad_out[index] = loss.grad();
            a.hess_out[index] = loss.hessian(); a.count_out[index] = 1;
        }
        acc[0] += loss.value(); acc[1] += loss.grad(); acc[2] += loss.hessian(); acc[3] += 1.0;
    });
    block_fold_and_finish<4>(acc, a.partials, a.ticket, a.out);
}

__global__ void __launch_bounds__(256) k_tsdf_loss(const HessArgs a, const HessPoseF P) {
    double acc[2] = {0.0, 0.0};
    walk_band(a, [&](int xq, int yq, int z, size_t index, float gt_tsdf) {
        const float vgx = (float(xq) + 0.5f) * a.voxel_size, vgy = (float(yq) + 0.5f) * a.voxel_size, vgz = (float(z) + 0.5f) * a.voxel_size;
        const float vcx = (P.R[0] * vgx + P.R[1] * vgy + P.R[2] * vgz) + P.t[0];
        const float vcy = (P.R[3] * vgx + P.R[4] * vgy + P.R[5] * vgz) + P.t[1];
        const float vcz = (P.R[6] * vgx + P.R[7] * vgy + P.R[8] * vgz) + P.t[2];
        const float inv_z = 1.0f / vcz;
        if (inv_z < 0) return;
        const float image_x = vcx * inv_z * a.intr.fx + a.intr.cx;
        const float image_y = vcy * inv_z * a.intr.fy + a.intr.cy;
        const int coo_x = __float2int_rd(image_x - 0.5f), coo_y = __float2int_rd(image_y - 0.5f);
        if (!(coo_x > 1 && coo_y > 1 && coo_x < a.dcols - 1 && coo_y < a.drows - 1)) return;
        const int near_x = __float2int_rn(image_x), near_y = __float2int_rn(image_y);
        float Dp = row_ptr(a.depth, a.dstep, near_y)[near_x];
        const float d00 = row_ptr(a.depth, a.dstep, coo_y)[coo_x], d10 = row_ptr(a.depth, a.dstep, coo_y)[coo_x + 1];
        const float d01 = row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x], d11 = row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x + 1];
        if (d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
            const float one = 1.0f;
            const float fa = image_x - (float(coo_x) + 0.5f), fb = image_y - (float(coo_y) + 0.5f);
            Dp = d00 * (one - fa) * (one - fb) + d10 * fa * (one - fb) + d01 * (one - fa) * fb + d11 * fa * fb;
        }
        if (Dp > 5 || Dp < 0.2) return;
        const float xl = (image_x - a.intr.cx) / a.intr.fx, yl = (image_y - a.intr.cy) / a.intr.fy;
        const float v1x = Dp * xl, v1y = Dp * yl, v1z = Dp;
        const float distance = sqrtf(v1x * v1x + v1y * v1y + v1z * v1z) - sqrtf(vcx * vcx + vcy * vcy + vcz * vcz);
        const float gt_distance = gt_tsdf * a.tranc_dist;
        const float error = (distance - gt_distance) * a.tranc_dist_inv;
        if (fabsf(error) > 1) return;
        const float loss = error * error;
        if (a.real_out) { a.real_out[index] = loss; a.count_out[index] = 1; }
        acc[0] += loss; acc[1] += 1.0;
    });
    block_fold_and_finish<2>(acc, a.partials, a.ticket, a.out);
}

// ---- first-order CSFD Gauss-Newton terms of the same residual (BASELINE config 5) ----------------
// The residual of ComputeLocalTsdfHessianKernel (TsdfFusion.cu:204-283) evaluated in complex<float>
// for six poses at once — the pose seeded with i*h along each of its six degrees of freedom — so one
// pass over the volume yields, per voxel, the residual r = Re(error) and the six derivative parts
// d_k = Im(error_k) = h * dr/dtheta_k, and on chip the sums a Gauss-Newton step needs:
//   out[0..20]  sum d_j d_k (upper triangle, rows j <= k),  out[21..26]  sum d_k r,
//   out[27]     sum r^2,                                    out[28]      voxel count
// (the caller divides by h^2 / h).  The reference has no such kernel; its commented ComputeTSDF_hessian
// (KinectFusionReconstruction.cpp:404-434) takes one seeded direction per call and would need 6 passes
// and 6 N^3 scratch volumes for the same matrix.
struct GnPoses { MatS33 R[6]; cfloat3 t[6]; };
__device__ __forceinline__ bool tsdf_error_c(const HessArgs &a, const MatS33 &R, const cfloat3 &t, float vgx, float vgy, float vgz, float gt,
                                             cfloat &error) {
    cfloat3 v_g; v_g.x = cfloat(vgx); v_g.y = cfloat(vgy); v_g.z = cfloat(vgz);
    cfloat3 v_c;
    v_c.x = dot(R.data[0], v_g) + t.x;
    v_c.y = dot(R.data[1], v_g) + t.y;
    v_c.z = dot(R.data[2], v_g) + t.z;
    const cfloat inv_z = cfloat(1.0f) / v_c.z;
    if (inv_z.re < 0) return false;
    const cfloat image_x = v_c.x * inv_z * a.intr.fx + a.intr.cx;
    const cfloat image_y = v_c.y * inv_z * a.intr.fy + a.intr.cy;
    const int coo_x = __float2int_rd(image_x.re - 0.5f), coo_y = __float2int_rd(image_y.re - 0.5f);
    if (!(coo_x > 1 && coo_y > 1 && coo_x < a.dcols - 1 && coo_y < a.drows - 1)) return false;
    const int near_x = __float2int_rn(image_x.re), near_y = __float2int_rn(image_y.re);
    cfloat Dp(row_ptr(a.depth, a.dstep, near_y)[near_x]);
    const float d00 = row_ptr(a.depth, a.dstep, coo_y)[coo_x], d10 = row_ptr(a.depth, a.dstep, coo_y)[coo_x + 1];
    const float d01 = row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x], d11 = row_ptr(a.depth, a.dstep, coo_y + 1)[coo_x + 1];
    if (d00 != 0.0f && d01 != 0.0f && d10 != 0.0f && d11 != 0.0f) {
        const cfloat one(1.0f);
        const cfloat fa = image_x - cfloat(float(coo_x) + 0.5f);
        const cfloat fb = image_y - cfloat(float(coo_y) + 0.5f);
        Dp = d00 * (one - fa) * (one - fb) + d10 * fa * (one - fb) + d01 * (one - fa) * fb + d11 * fa * fb;
    }
    if (Dp.re > 5 || Dp.re < 0.2) return false;
    const cfloat xl = (image_x - a.intr.cx) / a.intr.fx;
    const cfloat yl = (image_y - a.intr.cy) / a.intr.fy;
    const cfloat3 v_c_1 = mk3(Dp * xl, Dp * yl, Dp);
    const cfloat distance = norm(v_c_1) - norm(v_c);
    const cfloat gt_distance = cfloat(gt) * a.tranc_dist;
    error = (distance - gt_distance) * a.tranc_dist_inv;
    return !(fabsf(error.re) > 1);
}
// POSTED: the launch was enqueued before its poses existed (the host is still solving the previous pass): wave 0 polls the mailbox — six pose
// mailboxes of xs_mailbox.h in a row, written in order, so box 5 carrying the sequence number means boxes 0 .. 4 do — and fills P from it.
// A workgroup that is told to leave (cmd 1) or whose poses never come sums nothing but still takes its arrival ticket, marked: the launch's last
// workgroup then publishes the sequence number with bit 63 set instead of sums (block_fold_and_finish) — one record per launch whatever happened,
// and the ticket back at zero.
#ifdef XS_GN_WAVES_PER_EU   // experiment switch (three waves per SIMD = 168 VGPRs spill 31 registers here: 206 -> 172 relocalisations/s; left at the compiler's 228 = two waves)
#define XS_GN_OCC __attribute__((amdgpu_waves_per_eu(XS_GN_WAVES_PER_EU, XS_GN_WAVES_PER_EU)))
#else
#define XS_GN_OCC
#endif
template <bool POSTED>
__global__ void __launch_bounds__(256) XS_GN_OCC k_tsdf_gauss_newton(const HessArgs a, const GnPoses Pk) {
    // The six poses (144 floats) do not fit the scalar registers next to everything else, and the compiler then keeps them
    // in vector registers for the whole kernel (256 of them: one wave per SIMD).  They go through LDS instead: broadcast
    // reads where an evaluation needs them.
    __shared__ GnPoses P;
    bool left = false;
    if constexpr (POSTED) {
        __shared__ unsigned s_mail[xs::MAILBOX_WORDS];
        __shared__ unsigned s_cmd;
        if (threadIdx.y == 0) {
            const int lane = threadIdx.x;
            const unsigned long long t_resident = wall_clock64();   // (100 MHz: what this launch waits for its poses is the host's side of the loop)
            // (a workgroup of this launch has left already — told to, or it waited its second out: the ones that become resident later do not wait theirs)
            const bool somebody_left = (__hip_atomic_load(a.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) != 0;
            if (!somebody_left) xs::mailbox_wait(a.mailbox + 5 * xs::MAILBOX_WORDS, a.mailbox_seq, s_mail, lane);
            if (a.publish && blockIdx.x == 0 && lane == 0)   // word [30] of the record: ticks from resident to poses seen (the record's sequence word follows ~0.8 ms later)
                __hip_atomic_store(&a.publish[30], (double)(wall_clock64() - t_resident), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned cmd = somebody_left ? 1u : (unsigned)__builtin_amdgcn_readfirstlane((int)s_mail[1]);
            float *dst = reinterpret_cast<float *>(&P);
            for (int k = 0; k < 6 && cmd == 0; ++k) {   // (issued after box 5's sequence words were seen: complete payloads)
                const unsigned w = __hip_atomic_load(a.mailbox + k * xs::MAILBOX_WORDS + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned s0 = __builtin_amdgcn_readlane(w, 0), s1 = __builtin_amdgcn_readlane(w, 8), s2 = __builtin_amdgcn_readlane(w, 16),
                               s3 = __builtin_amdgcn_readlane(w, 24);
                if (s0 != a.mailbox_seq || s1 != a.mailbox_seq || s2 != a.mailbox_seq || s3 != a.mailbox_seq) { cmd = 2; break; }
                const int f = xs::mailbox_float_of(lane & 31);   // (xs_mailbox.h: four sectors, each {seq, payload})
                if (lane < 32 && f >= 0) dst[f < 18 ? 18 * k + f : 108 + 6 * k + (f - 18)] = __uint_as_float(w);
            }
            if (lane == 0) s_cmd = cmd;
        }
        __syncthreads();
        left = s_cmd != 0;   // (the workgroup still arrives: block_fold_and_finish)
    } else {
        const float *src = reinterpret_cast<const float *>(&Pk);
        float *dst = reinterpret_cast<float *>(&P);
        for (int i = threadIdx.y * 64 + threadIdx.x; i < (int)(sizeof(GnPoses) / sizeof(float)); i += 256) dst[i] = src[i];
        __syncthreads();
    }
    double acc[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) acc[k] = 0.0;
    if (!left) walk_band(a, [&](int xq, int yq, int z, size_t index, float gt) {
        const float vgx = (float(xq) + 0.5f) * a.voxel_size, vgy = (float(yq) + 0.5f) * a.voxel_size, vgz = (float(z) + 0.5f) * a.voxel_size;
        cfloat e[6];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ok = ok && tsdf_error_c(a, P.R[k], P.t[k], vgx, vgy, vgz, gt, e[k]);
            // one evaluation at a time: left to itself the scheduler interleaves all six (256 registers, one wave per SIMD)
            asm volatile("" : "+v"(e[k].re), "+v"(e[k].im) :: "memory");
        }
        if (!ok) return;  // a voxel counts only if every seeded evaluation keeps it (they share their real parts)
        const double r = (double)e[0].re;
        int s = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int k = j; k < 6; ++k) acc[s++] += (double)e[j].im * (double)e[k].im;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[21 + k] += (double)e[k].im * r;
        acc[27] += r * r;
        acc[28] += 1.0;
    });
    block_fold_and_finish<29>(acc, a.partials, a.ticket, a.out, a.publish, a.publish_seq, left);
}

enum { XS_TSDF_REDUCE_MAX_BLOCKS = XS_TSDF_REDUCE_MAX_BLOCKS_C };
extern "C" size_t xs_tsdf_reduce_workspace_bytes(void) { return (size_t)XS_TSDF_REDUCE_MAX_BLOCKS * 32 * sizeof(double) + 256; }
/* Zero the workspace's arrival ticket once after allocation (any zero fill of the first 256 bytes does): every launch of the three residual kernels
 * leaves it zero — their last workgroup resets it — so a launch needs no fill of its own (round 6: that fill was a dispatch in front of every pass).
 * One launch at a time per workspace.  After a launch that did not complete (a device fault), initialise again. */
extern "C" int xs_tsdf_reduce_workspace_init(void *workspace, void *stream) {
    if (!workspace) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_reduce_workspace_init: null pointer");
    XS_CHECK(hipMemsetAsync(workspace, 0, 256, (hipStream_t)stream));
    return 0;
}

static int hess_common(HessArgs &a, const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                       float voxel_size, float tranc_dist, const float *gt, int z0, int z1, void *workspace, double *out_dev, dim3 &grid,
                       void *stream, bool heavy_body) {
    if (!depth_scaled || !intr4 || !res || !gt || !workspace || !out_dev) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_hessian/loss: null pointer");
    if (z0 < 0 || z1 > res[2] || z1 <= z0) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_hessian/loss: bad slab");
    a.depth = depth_scaled; a.dstep = scaled_step; a.drows = rows; a.dcols = cols;
    a.X = res[0]; a.Y = res[1]; a.Z = res[2]; a.z0 = z0; a.z1 = z1;
    a.voxel_size = voxel_size; a.tranc_dist = tranc_dist; a.tranc_dist_inv = 1.0f / tranc_dist;
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.publish = nullptr; a.publish_seq = 0; a.mailbox = nullptr; a.mailbox_seq = 0;
    a.gt = gt; a.ticket = (unsigned *)workspace; a.partials = (double *)((char *)workspace + 256); a.out = out_dev;
    // Tiles of 64 x 4 columns x zchunk planes, one column per lane (256 x 4 with four columns per lane: a.wide); the workgroups stride
    // over them.  (While every workgroup paid an L2 write-back for its record, 4096 of them halved the streaming rate against 1024; the
    // records now leave with write-through stores: block_fold_and_finish.)
    static const int env_blocks = exp_env_int("XS_HESS_BLOCKS", 0);  // tuning aid
    static const bool env_narrow = exp_env_set("XS_HESS_NARROW");    // A/B aid: the one-column-per-lane scan whatever the shape
    // sixteen bytes per lane where the rows allow it (for_band_voxels_wide): X a multiple of four and the slab's first voxel 16-byte aligned
    // Which planes a workgroup takes: runs of consecutive planes stream fastest (eight ADJACENT planes per request: 6.1 TB/s scan alone at
    // 512^3 against 5.7 for planes four apart), but a band is a thin sheet and a wall across z then lies in ONE workgroup's run per column — the
    // kernels whose band voxels are expensive (dual-complex Hessian, six-pose Gauss-Newton) take five interleaved groups of planes (below), the
    // loss kernel takes runs (profiles/r06_hess_scan.txt).
    static const int env_ilg = exp_env_int("XS_HESS_IL", 1);   // tuning aid: consecutive planes a group takes together (pairs measured the same or worse)
    a.il = (env_ilg == 2 || env_ilg == 4 || env_ilg == 8) ? env_ilg : 1;
    static const int env_il = exp_env_int("XS_HESS_INTERLEAVE", -1);   // A/B aid: 0 = runs, 1 = interleaved, whatever the kernel
    const bool interleave = env_il < 0 ? heavy_body : env_il != 0;
    a.wide = (a.X % 4 == 0 && (reinterpret_cast<uintptr_t>(gt) % 16) == 0 && !env_narrow) ? (interleave ? 2 : 1) : 0;
    int gx = div_up(a.X, a.wide ? 256 : 64), gy = div_up(a.Y, 4), nz = z1 - z0, zsplit = 1;
    // one workgroup per column of tiles while that gives 1024 .. 4096 of them (512^3: 1024, 1024^3: 4096 — measured best:
    // the Gauss-Newton pass at 1024^3 runs 15 % faster with 4096 workgroups walking one column each than with 1024 walking
    // four); fewer columns are split along z, more are strided over
    const long long cols_xy = (long long)gx * gy;
    // (in the bare scan 4096 workgroups streamed 4 % faster than 1024; in the kernels, which pay a record and a ticket per workgroup, 8-12 % slower)
    const int cap = env_blocks > 0 && env_blocks <= XS_TSDF_REDUCE_MAX_BLOCKS ? env_blocks
                    : a.wide == 2 ? (int)XS_TSDF_REDUCE_MAX_BLOCKS   // (one workgroup per tile up to 4096: 512^3 has 1024 tiles, 1024^3 4096)
                    : a.wide ? (int)(cols_xy < 1024 ? 1024 : (cols_xy > XS_TSDF_REDUCE_MAX_BLOCKS ? XS_TSDF_REDUCE_MAX_BLOCKS : cols_xy))
                    : (int)(cols_xy < 1024 ? 1024 : (cols_xy > XS_TSDF_REDUCE_MAX_BLOCKS ? XS_TSDF_REDUCE_MAX_BLOCKS : cols_xy));
    if (a.wide == 2) {
        // FIVE z groups per column of tiles, their planes interleaved one by one (a tile: 256 x 4 columns x every fifth plane).  A kernel is as
        // slow as its busiest wave, and a wave that lies IN a surface holds nothing but band voxels: a wall across z is a sheet one or two planes
        // thick — plane by plane it goes to different groups; a floor (a wall along z and x) fills whole rows of a column — with whole columns per
        // wave (round 5: 64 x 1024 voxels at 1024^3; 256 x 1024 with four columns per lane) the box room's floor kept a few dozen waves busy long
        // after the rest had left: 0.89 ms per Gauss-Newton pass of the relocalisation workload then, 2.0 ms with four columns per lane and whole
        // columns, 0.80 ms now.  More, smaller groups balance better and stream worse (32-plane groups: 0.139 ms for the Hessian kernel at 512^3
        // against 0.097).  FIVE, not four: the eight requests of a batch lie G planes apart, and with a power of two between them (4 MiB at 512^3,
        // 16 MiB at 1024^3) the scan alone loses 10 % at 512^3 (0.098 against 0.087 ms with three or six groups: the requests of a lane fall on the
        // same memory channels) — five keeps the balance of four and the rate of an odd stride: Gauss-Newton 1024^3 0.79 -> 0.73 ms, relocalisation
        // 206 -> 209 frames/s, Hessian 512^3 0.098 -> 0.097 (profiles/r06_hess_scan.txt 5, 7).  Workgroup b takes tiles b, b + grid, ... of an
        // enumeration with the z group fastest, skewed by one group per round (walk_band).
        static const int env_zt = exp_env_int("XS_HESS_TILE_PLANES", 0);   // tuning aid: planes per group
        int G = nz >= 80 ? 5 : (nz >= 48 ? 3 : (nz >= 24 ? 2 : 1));
        if (env_zt >= 8) { G = 1; while (G * 2 * env_zt <= nz) G *= 2; }
        static const int env_g = exp_env_int("XS_HESS_GROUPS", 0);   // tuning aid: G itself (any number)
        if (env_g >= 1 && env_g * 8 <= nz) G = env_g;
        a.tiles_x = gx; a.tiles_y = gy; a.tiles_z = G; a.zchunk = div_up(nz, G);
    } else {
        while ((long long)gx * gy * zsplit < cap && zsplit < nz && nz / (zsplit * 2) >= 16) zsplit *= 2;
        a.zchunk = div_up(nz, zsplit);
        a.tiles_x = gx; a.tiles_y = gy; a.tiles_z = div_up(nz, a.zchunk);
    }
    const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.tiles_z;
    grid = dim3((unsigned)(ntiles < cap ? ntiles : cap));
    if ((long long)grid.x * grid.y * grid.z > XS_TSDF_REDUCE_MAX_BLOCKS) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_hessian/loss: volume too large for the reduce workspace");
    return 0;
}

/* float4 ComputeLocalTsdf_hessian(const PtrStepSz<ushort>& depth, const Intr&, DeviceArray2D<float>& depthScaled,
 *     const int3& res, float voxel_size, const MatD33& Rv2c, const devDComplex3& tv2c, float tranc_dist,
 *     float threshold, float k, thrustDvec<float>& gt, real, grad, hessian, thrustDvec<int>& count)
 *                                                        TsdfFusion.h:55-60, TsdfFusion.cu:286-331
 * depth_scaled: output of xs_scale_depth.  Rv2c36 / tv2c12: MatD33 / devDComplex3 as groups of
 * (re.re, re.im, im.re, im.im).  gt: dense unpitched TSDF of the slab [z0, z1).  out4_dev: 4
 * doubles {loss, grad, hessian, count} (the reference returns them narrowed to float4).  The
 * four per-voxel volumes are optional (all four or none).  threshold and k are unused by the
 * reference kernel.  No synchronisation. */
extern "C" int xs_compute_local_tsdf_hessian(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4,
                                             const int *res, float voxel_size, const float *Rv2c36, const float *tv2c12,
                                             float tranc_dist, const float *gt, float *real_out, float *grad_out, float *hess_out,
                                             int *count_out, int z0, int z1, void *workspace, double *out4_dev, void *stream) {
    HessArgs a; dim3 grid;
    int rc = hess_common(a, depth_scaled, scaled_step, rows, cols, intr4, res, voxel_size, tranc_dist, gt, z0, z1, workspace, out4_dev, grid, stream, true);
    if (rc) return rc;
    if (!Rv2c36 || !tv2c12) return xs_set_error(hipErrorInvalidValue, "xs_compute_local_tsdf_hessian: null pose");
    const bool all = real_out && grad_out && hess_out && count_out, none = !real_out && !grad_out && !hess_out && !count_out;
    if (!all && !none) return xs_set_error(hipErrorInvalidValue, "xs_compute_local_tsdf_hessian: pass all four volumes or none");
    a.real_out = real_out; a.grad_out = grad_out; a.hess_out = hess_out; a.count_out = count_out;
    HessPoseD P;
    for (int r = 0; r < 3; ++r) {
        const float *p = Rv2c36 + r * 12;
        P.R.data[r].x = dcfloat(p[0], p[1], p[2], p[3]);
        P.R.data[r].y = dcfloat(p[4], p[5], p[6], p[7]);
        P.R.data[r].z = dcfloat(p[8], p[9], p[10], p[11]);
    }
    P.t.x = dcfloat(tv2c12[0], tv2c12[1], tv2c12[2], tv2c12[3]);
    P.t.y = dcfloat(tv2c12[4], tv2c12[5], tv2c12[6], tv2c12[7]);
    P.t.z = dcfloat(tv2c12[8], tv2c12[9], tv2c12[10], tv2c12[11]);
    hipLaunchKernelGGL(k_tsdf_hessian, grid, dim3(64, 4), 0, (hipStream_t)stream, a, P);
    XS_CHECK(hipGetLastError());
    return 0;
}

/* float2 ComputeLocalTsdf_loss(..., const Mat33& Rv2c, const float3& tv2c, ..., gt, real, count)
 *                                                        TsdfFusion.h:48-52, TsdfFusion.cu:412-447
 * out2_dev: {loss, count} as doubles. */
extern "C" int xs_compute_local_tsdf_loss(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4,
                                          const int *res, float voxel_size, const float *Rv2c9, const float *tv2c3, float tranc_dist,
                                          const float *gt, float *real_out, int *count_out, int z0, int z1, void *workspace,
                                          double *out2_dev, void *stream) {
    HessArgs a; dim3 grid;
    int rc = hess_common(a, depth_scaled, scaled_step, rows, cols, intr4, res, voxel_size, tranc_dist, gt, z0, z1, workspace, out2_dev, grid, stream, false);
    if (rc) return rc;
    if (!Rv2c9 || !tv2c3) return xs_set_error(hipErrorInvalidValue, "xs_compute_local_tsdf_loss: null pose");
    if ((real_out == nullptr) != (count_out == nullptr)) return xs_set_error(hipErrorInvalidValue, "xs_compute_local_tsdf_loss: pass both volumes or none");
    a.real_out = real_out; a.grad_out = nullptr; a.hess_out = nullptr; a.count_out = count_out;
    HessPoseF P;
    for (int i = 0; i < 9; ++i) P.R[i] = Rv2c9[i];
    for (int i = 0; i < 3; ++i) P.t[i] = tv2c3[i];
    hipLaunchKernelGGL(k_tsdf_loss, grid, dim3(64, 4), 0, (hipStream_t)stream, a, P);
    XS_CHECK(hipGetLastError());
    return 0;
}

/* First-order CSFD Gauss-Newton terms of the Hessian kernel's residual for six seeded poses in one pass
 * (BASELINE config 5; no counterpart launcher in the reference).  Rv2c108 / tv2c36: six MatS33 / devComplex3
 * (pose k carries i*h on degree of freedom k; real parts equal).  out29_dev: 29 doubles — sum d_j d_k for
 * j <= k (21, row-major upper triangle), sum d_k r (6), sum r^2, count — with d_k = Im(error_k), r =
 * Re(error_0); voxels are those with gt != 0, |gt| <= 0.95 that pass the kernel's gates for all six poses.
 * gt / depth_scaled / slab arguments as xs_compute_local_tsdf_hessian.  No synchronisation. */
extern "C" int xs_tsdf_gauss_newton_terms(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                                          float voxel_size, const float *Rv2c108, const float *tv2c36, float tranc_dist, const float *gt, int z0,
                                          int z1, void *workspace, double *out29_dev, void *stream) {
    return xs_tsdf_gauss_newton_terms_ex(depth_scaled, scaled_step, rows, cols, intr4, res, voxel_size, Rv2c108, tv2c36, tranc_dist, gt, z0, z1, workspace,
                                         out29_dev, nullptr, stream);
}
/* ... with the loop protocol of the ICP iterations (opts; NULL = none of it):
 *   publish_host / publish_seq   host-coherent pinned memory of xs_gn_publish_bytes(): the last workgroup stores the 29 sums there and then the
 *                                64-bit word [32] = publish_seq — the host spins on that word instead of copying and draining the stream;
 *   pose_mailbox / mailbox_seq   Rv2c108 / tv2c36 NULL: the launch is enqueued before its poses exist and takes them from the mailbox
 *                                (xs_icp_mailbox_alloc; xs_gn_post_poses writes it).  cmd 1 or a pose that never comes (about a second): nothing is
 *                                summed, publish word = publish_seq | 1 << 63. */
extern "C" int xs_tsdf_gauss_newton_terms_ex(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                                             float voxel_size, const float *Rv2c108, const float *tv2c36, float tranc_dist, const float *gt, int z0,
                                             int z1, void *workspace, double *out29_dev, const xs_gn_opts *opts, void *stream) {
    if (opts && opts->struct_bytes != sizeof(xs_gn_opts)) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_gauss_newton_terms_ex: opts->struct_bytes is not sizeof(xs_gn_opts)");
    HessArgs a; dim3 grid;
    int rc = hess_common(a, depth_scaled, scaled_step, rows, cols, intr4, res, voxel_size, tranc_dist, gt, z0, z1, workspace, out29_dev, grid, stream, true);
    if (rc) return rc;
    const bool posted = opts && opts->pose_mailbox && !Rv2c108 && !tv2c36;
    if (!posted && (!Rv2c108 || !tv2c36)) return xs_set_error(hipErrorInvalidValue, "xs_tsdf_gauss_newton_terms: give the six poses or a mailbox");
    a.real_out = nullptr; a.grad_out = nullptr; a.hess_out = nullptr; a.count_out = nullptr;
    if (opts) { a.publish = opts->publish_host; a.publish_seq = opts->publish_seq; }
    GnPoses P;
    memset(&P, 0, sizeof(P));
    if (posted) {
        a.mailbox = static_cast<const unsigned *>(opts->pose_mailbox); a.mailbox_seq = opts->mailbox_seq;
        hipLaunchKernelGGL(k_tsdf_gauss_newton<true>, grid, dim3(64, 4), 0, (hipStream_t)stream, a, P);
    } else {
        for (int k = 0; k < 6; ++k) { load_mat(Rv2c108 + 18 * k, P.R[k]); load_vec(tv2c36 + 6 * k, P.t[k]); }
        hipLaunchKernelGGL(k_tsdf_gauss_newton<false>, grid, dim3(64, 4), 0, (hipStream_t)stream, a, P);
    }
    XS_CHECK(hipGetLastError());
    return 0;
}
extern "C" size_t xs_gn_publish_bytes(void) { return 33 * sizeof(double); }
extern "C" size_t xs_gn_mailbox_bytes(void) { return 6 * xs::MAILBOX_WORDS * sizeof(unsigned); }
/* Host: the six seeded poses (or a command: cmd 1 = leave) for the launch that polls `mailbox_host` for `mailbox_seq` — six mailboxes of the
 * xs_icp_post_pose layout in a row.  The launch polls the LAST box and then reads all six, checking each one's sequence words: so the payloads of
 * all six go out first, then the sequence words of boxes 0 .. 4, then those of box 5 — three store fences (the mailbox is write-combining
 * BAR memory on the CPU side: stores may pass each other between fences) instead of the twelve of six xs_icp_post_pose calls, each of which
 * drains the write-combining buffers while the launch waits.  With MOVDIR64B (xs_mailbox.h): twelve direct 64-byte writes and one fence. */
extern "C" void xs_gn_post_poses(void *mailbox_host, const float *Rv2c108, const float *tv2c36, unsigned mailbox_seq, int cmd) {
    static const bool twelve = exp_env_set("XS_GN_POST_TWELVE_FENCES");   // A/B aid: box by box, as until round 6
    if (twelve) {
        for (int k = 0; k < 6; ++k)
            xs_icp_post_pose(static_cast<char *>(mailbox_host) + (size_t)k * xs::MAILBOX_WORDS * sizeof(unsigned), Rv2c108 ? Rv2c108 + 18 * k : nullptr,
                             tv2c36 ? tv2c36 + 6 * k : nullptr, mailbox_seq, cmd);
        return;
    }
    static const bool direct = mailbox_cpu_has_direct_store() && !exp_env_set("XS_MAILBOX_NO_DIRECT_STORE");
    alignas(64) unsigned img[6][xs::MAILBOX_WORDS];
    for (int k = 0; k < 6; ++k) mailbox_image(img[k], Rv2c108 ? Rv2c108 + 18 * k : nullptr, tv2c36 ? tv2c36 + 6 * k : nullptr, mailbox_seq, cmd);
    volatile unsigned *base = static_cast<volatile unsigned *>(mailbox_host);
    if (direct && (reinterpret_cast<uintptr_t>(mailbox_host) % 64) == 0) {
        // MOVDIR64B: every line is one write with its sequence word inside; boxes 0 .. 4, one fence, box 5 (the one the launch polls)
        mailbox_store_fence();
        for (int k = 0; k < 5; ++k) {
            mailbox_direct_store_64(const_cast<unsigned *>(base) + (size_t)k * xs::MAILBOX_WORDS, img[k]);
            mailbox_direct_store_64(const_cast<unsigned *>(base) + (size_t)k * xs::MAILBOX_WORDS + 16, img[k] + 16);
        }
        mailbox_store_fence();
        mailbox_direct_store_64(const_cast<unsigned *>(base) + 5 * (size_t)xs::MAILBOX_WORDS, img[5]);
        mailbox_direct_store_64(const_cast<unsigned *>(base) + 5 * (size_t)xs::MAILBOX_WORDS + 16, img[5] + 16);
        return;
    }
    for (int k = 0; k < 6; ++k)
        for (int i = 0; i < xs::MAILBOX_WORDS; ++i) if (i % 8 != 0) base[(size_t)k * xs::MAILBOX_WORDS + i] = img[k][i];
    mailbox_store_fence();
    for (int k = 0; k < 5; ++k) for (int i = 0; i < xs::MAILBOX_WORDS; i += 8) base[(size_t)k * xs::MAILBOX_WORDS + i] = mailbox_seq;
    mailbox_store_fence();
    for (int i = 0; i < xs::MAILBOX_WORDS; i += 8) base[5 * (size_t)xs::MAILBOX_WORDS + i] = mailbox_seq;
    mailbox_store_fence();
}
__global__ void k_publish_sums(const double *sums, int n, double *publish, unsigned long long seq) {
    const int tid = threadIdx.x;
    if (tid < n) __hip_atomic_store(&publish[tid], sums[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(publish) + 32, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
/* Shard mode: the sums as they stand in device memory AFTER the stream's all-reduce, published the same way (n <= 32 doubles, then word [32] = seq). */
extern "C" int xs_gn_publish_sums(const double *sums_dev, int n, double *publish_host, unsigned long long seq, void *stream) {
    if (!sums_dev || !publish_host || n < 1 || n > 32) return xs_set_error(hipErrorInvalidValue, "xs_gn_publish_sums: bad argument");
    hipLaunchKernelGGL(k_publish_sums, dim3(1), dim3(64), 0, (hipStream_t)stream, sums_dev, n, publish_host, seq);
    XS_CHECK(hipGetLastError());
    return 0;
}
