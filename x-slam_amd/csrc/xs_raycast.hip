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
#include <hip/hip_ext.h>
#include "xs_device.h"
#include "xs_env.h"
#include "xs_signmap.h"
#include "xs_pyramid.h"
#include <type_traits>
#include "../../include/xslam_amd.h"

using namespace xs;
xs::ConstDiv xs_const_div_get(float c);   // xs_constdiv.hip

struct RaycastArgs {
    MatS33 Rc2v; cfloat3 tc2v; MatS33 Rv2w; cfloat3 tv2w;
    int X, Y, Z;
    float voxel_size, time_step;
    float inv_vs_lo, inv_vs_hi;  // 1/voxel_size nudged 4 ulp down / up (march index shortcut when the constant is not prepared)
    ConstDiv dv;                 // division by voxel_size (xs_const_div_prepare)
    int cols, rows;
    const float *value; const float *grad; size_t vstep;
    Intr intr;
    cfloat *vmap; cfloat *nmap; size_t mstep;
    int zs0, zs1;  // z planes resident behind value/grad: storage starts at plane zs0 (whole volume: 0, Z)
    int z0, z1;    // z planes this launch owns (slab mode): only steps whose sample lands here are evaluated
    int wshift;    // log2 of the wave's pixel-tile width (tile = 2^wshift x 64/2^wshift pixels)
    float *cross_t; // two-kernel path: per pixel, the march time of the step before the crossing (or < 0)
    int *keys;     // slab mode: per pixel, (step << 1 | no_hit) of the first event among owned steps, INT_MAX if none
    unsigned long long *hits;
    int *steps;    // optional (measurement): per pixel, the march iterations the reference's loop (RayCaster.cu:222-247) runs for this ray
    SignMap sm;    // sm.dil != null (the march kernels): evaluate only the iterations the sign map leaves (xs_signmap.h)
    float sm_dt; int sm_rounds;   // sign map: spacing of the per-wave samples along the tile's centre ray; 64 * sm_rounds of them
    PyramidArgs pyr;              // pyr.mid[0] != null (the fused one-launch form): every workgroup also builds levels 1 and 2 of both maps under its own tile
};

namespace {
__device__ __forceinline__ int sgn(float v) { return (0.0f < v) - (v < 0.0f); }

// floor(fl(p / vs)) — getVoxel, RayCaster.cu:80-86 — without the IEEE divide: the two products bracket
// fl(p / vs) (the reciprocals are 4 ulp either side of 1/vs), so when their floors agree that is the
// answer; otherwise (the quotient lies within ~8 ulp of an integer, ~1e-4 of the samples) the
// divide is done.
__device__ __forceinline__ int cvt_flr(float v) {  // (int)floorf(v) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
template <bool SHORT>
__device__ __forceinline__ int voxel_index(float p, float vs, float r_lo, float r_hi, const ConstDiv &dv) {
    // SHORT (the march kernels, when xs_const_div_prepare has checked voxel_size over all 2^32 operands): the exact short division,
    // six branch-free instructions; else two reciprocal products that bracket the quotient, with the divide where they disagree.
    // (Measured, round 3: march 48-51 us against 52-53; the crossing kernel and the integrate band keep their IEEE divides — the
    // compiler hoists the divisor's half of that sequence (v_rcp, v_div_scale, refinement) out of the loop as the divisor is a launch
    // constant, so a divide costs ~6 instructions there, and the short form made the crossing 8-18 % SLOWER: profiles/r03_ab_const_div.txt)
    if (SHORT) return floor_div_by<true>(p, dv);
    const int f_lo = cvt_flr(p * r_lo), f_hi = cvt_flr(p * r_hi);
    if (__builtin_expect(f_lo == f_hi, 1)) return f_lo;
    return __float2int_rd(p / vs);
}

// OFF32: the resident planes of one array span at most 4 GiB, so a voxel's byte offset fits 32 bits
// (two 24-bit multiply-adds and a shift instead of 64-bit multiplies; the load takes a scalar base +
// 32-bit lane offset).  The kernels are VALU-bound: the march spends a fifth of its instructions on
// addresses otherwise.
template <bool OFF32>
struct Vol {
    const float *value; const float *grad; size_t vstep; int X, Y, Z; float vs; int zs0, zs1;
    __device__ __forceinline__ unsigned offset32(int x, int y, int z) const {
        const unsigned row = __umul24((unsigned)(z - zs0), (unsigned)Y) + (unsigned)y;  // < 2^24 for every supported volume
        return (__umul24(row, (unsigned)(vstep >> 2)) + (unsigned)x) << 2;
    }
    __device__ __forceinline__ float value_at(unsigned off) const { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(value) + off); }
    __device__ __forceinline__ float load_value(int x, int y, int z) const {
        if (OFF32) return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(value) + offset32(x, y, z));
        return row_ptr(value, vstep, Y * (z - zs0) + y)[x];
    }
    __device__ __forceinline__ float read_value(int x, int y, int z) const { return load_value(x, y, z) + 1e-5f; }  // RayCaster.cu:76
    __device__ __forceinline__ cfloat read(int x, int y, int z) const {  // readTsdf, :69-78
        // readTsdf wraps its indices with % resolution; every caller here passes 0 <= index < resolution
        // (interp() rejects cells outside [1, N-2] before touching the +1 neighbours), where the wrap is
        // the identity — and an integer modulo by a run-time value costs ~35 VALU ops, 192 of them per hit
        z = min(max(z, zs0), zs1 - 1);  // stay inside the resident planes (a no-op unless a slab's halo were too thin)
        cfloat r;
        if (OFF32) {
            const unsigned off = offset32(x, y, z);
            r = unpack_tsdf(*reinterpret_cast<const float *>(reinterpret_cast<const char *>(value) + off),
                            *reinterpret_cast<const float *>(reinterpret_cast<const char *>(grad) + off));
        } else
            r = unpack_tsdf(row_ptr(value, vstep, Y * (z - zs0) + y)[x], row_ptr(grad, vstep, Y * (z - zs0) + y)[x]);
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
        // the corners (gx, ·, ·) and (gx + 1, ·, ·) are neighbours in memory: one 8-byte load per array fetches both (a global
        // load needs 4-byte alignment only) — 8 address-unit trips per sample instead of 16
        cfloat lo00, hi00, lo01, hi01, lo10, hi10, lo11, hi11;
        read2(gx, gy + 0, gz + 0, lo00, hi00); read2(gx, gy + 0, gz + 1, lo01, hi01);
        read2(gx, gy + 1, gz + 0, lo10, hi10); read2(gx, gy + 1, gz + 1, lo11, hi11);
        return lo00 * a1 * b1 * c1 + lo01 * a1 * b1 * c0 + lo10 * a1 * b0 * c1 + lo11 * a1 * b0 * c0 +
               hi00 * a0 * b1 * c1 + hi01 * a0 * b1 * c0 + hi10 * a0 * b0 * c1 + hi11 * a0 * b0 * c0;
    }
    // Two samples at once: both cells are located first, then the sixteen 8-byte loads of the two go out together and the
    // two blends follow — one memory round trip where two calls of interp() make two (the crossing kernel is a chain of
    // eight dependent samples per pixel: Ft / Ftdt, then the three central differences of the normal).  Same values as
    // interp(p), interp(q): a sample outside [1, N-2] is NaN; its loads go to a resident cell instead and are not used.
    struct Cell { int gx, gy, gz; bool ok; cfloat a0, b0, c0; };
    __device__ __forceinline__ Cell locate(const cfloat3 &p) const {
        Cell c;
        int gx = __float2int_rd(p.x.re / vs), gy = __float2int_rd(p.y.re / vs), gz = __float2int_rd(p.z.re / vs);
        c.ok = !(gx <= 0 || gx >= X - 1) && !(gy <= 0 || gy >= Y - 1) && !(gz <= 0 || gz >= Z - 1);
        const float vx = (gx + 0.5f) * vs, vy = (gy + 0.5f) * vs, vz = (gz + 0.5f) * vs;
        gx += -(sgn(vx - p.x.re) + 1) >> 1;
        gy += -(sgn(vy - p.y.re) + 1) >> 1;
        gz += -(sgn(vz - p.z.re) + 1) >> 1;
        c.a0 = (p.x - (gx + 0.5f) * vs) / vs;
        c.b0 = (p.y - (gy + 0.5f) * vs) / vs;
        c.c0 = (p.z - (gz + 0.5f) * vs) / vs;
        c.gx = c.ok ? gx : 0; c.gy = c.ok ? gy : 0; c.gz = c.ok ? gz : zs0;
        return c;
    }
    __device__ __forceinline__ cfloat blend(const Cell &c, const cfloat (&lo)[4], const cfloat (&hi)[4]) const {
        const cfloat one(1.0f, 0.0f);
        const cfloat a1 = one - c.a0, b1 = one - c.b0, c1 = one - c.c0;
        const cfloat r = lo[0] * a1 * b1 * c1 + lo[1] * a1 * b1 * c.c0 + lo[2] * a1 * c.b0 * c1 + lo[3] * a1 * c.b0 * c.c0 +
                         hi[0] * c.a0 * b1 * c1 + hi[1] * c.a0 * b1 * c.c0 + hi[2] * c.a0 * c.b0 * c1 + hi[3] * c.a0 * c.b0 * c.c0;
        return c.ok ? r : cfloat(qnan_f(), 0.f);
    }
    __device__ __forceinline__ void interp2(const cfloat3 &p, const cfloat3 &q, cfloat &Fp, cfloat &Fq) const {
        const Cell cp = locate(p), cq = locate(q);
        cfloat lp[4], hp[4], lq[4], hq[4];
        read2(cp.gx, cp.gy + 0, cp.gz + 0, lp[0], hp[0]); read2(cp.gx, cp.gy + 0, cp.gz + 1, lp[1], hp[1]);
        read2(cp.gx, cp.gy + 1, cp.gz + 0, lp[2], hp[2]); read2(cp.gx, cp.gy + 1, cp.gz + 1, lp[3], hp[3]);
        read2(cq.gx, cq.gy + 0, cq.gz + 0, lq[0], hq[0]); read2(cq.gx, cq.gy + 0, cq.gz + 1, lq[1], hq[1]);
        read2(cq.gx, cq.gy + 1, cq.gz + 0, lq[2], hq[2]); read2(cq.gx, cq.gy + 1, cq.gz + 1, lq[3], hq[3]);
        Fp = blend(cp, lp, hp);
        Fq = blend(cq, lq, hq);
    }
    // read(x, y, z) and read(x + 1, y, z)
    __device__ __forceinline__ void read2(int x, int y, int z, cfloat &lo, cfloat &hi) const {
        struct __attribute__((packed, aligned(4))) pair { float a, b; };
        z = min(max(z, zs0), zs1 - 1);
        const pair *pv, *pg;
        if (OFF32) {
            const unsigned off = offset32(x, y, z);
            pv = reinterpret_cast<const pair *>(reinterpret_cast<const char *>(value) + off);
            pg = reinterpret_cast<const pair *>(reinterpret_cast<const char *>(grad) + off);
        } else {
            pv = reinterpret_cast<const pair *>(row_ptr(value, vstep, Y * (z - zs0) + y) + x);
            pg = reinterpret_cast<const pair *>(row_ptr(grad, vstep, Y * (z - zs0) + y) + x);
        }
        const pair v = *pv, g = *pg;
        lo = unpack_tsdf(v.a, g.a); lo += 1e-5f;
        hi = unpack_tsdf(v.b, g.b); hi += 1e-5f;
    }
};
}  // namespace

// MODE 0: whole ray in one kernel; 1: slab (multi-GPU), whole ray in one kernel; 2: march only, records the crossing
// time; 3: crossing only (trilinear samples, vertex, normal) for the pixels MODE 2 marked; 4 / 5: the slab march and
// the slab crossing as two kernels (what xs_raycast_slab launches: the march then runs at eight waves per SIMD with
// eight gathers in flight, like MODE 2).
// (crossing kernels: five waves per SIMD — a 640 x 480 frame is 4 800 waves on 1 024 SIMDs, all resident at once; the
// two-samples-per-round-trip form would otherwise take 104 registers, i.e. four)
#ifndef XS_RAYCAST_CROSS_WAVES
#define XS_RAYCAST_CROSS_WAVES 5
#endif
constexpr int raycast_min_waves(int mode, bool map) { return mode == 3 || mode == 5 || (mode == 0 && map) ? XS_RAYCAST_CROSS_WAVES : 1; }
template <int MODE, bool OFF32, bool SHORT, bool MAP = false>   // MAP (MODE 2 or 0): the march evaluates only the iterations the sign map (a.sm) leaves
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(raycast_min_waves(MODE, MAP)))) k_raycast(const RaycastArgs a) {
    constexpr bool SLAB = MODE == 1 || MODE == 4 || MODE == 5;
    constexpr bool PAIRS = MODE == 3 || MODE == 5 || (MODE == 0 && MAP);   // the crossing takes its eight samples two at a time (interp2)
    // lane -> pixel inside an 8x8 tile; 4 waves -> 16x16 tile per workgroup.  Workgroups are dealt
    // round-robin over the 8 XCDs (each with its own 4 MB L2): the linear id is remapped so that
    // XCD k marches one contiguous band of image tiles — an eighth of the frustum, which fits its
    // L2 — instead of every XCD pulling every line.  Placement only affects speed.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wx = 1 << a.wshift, wy = 64 >> a.wshift;  // wave tile; the workgroup covers 2 x 2 of them
    const int tiles_x = (a.cols + 2 * wx - 1) / (2 * wx), tiles_y = (a.rows + 2 * wy - 1) / (2 * wy), ntiles = tiles_x * tiles_y;
    const int per_xcd = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const bool tile_ok = tile < ntiles;
    const int x = (tile % tiles_x) * 2 * wx + (wave & 1) * wx + (lane & (wx - 1));
    const int y = (tile / tiles_x) * 2 * wy + (wave >> 1) * wy + (lane >> a.wshift);
    unsigned hit = 0;
    // MAP: which iterations of the reference's loop this wave's rays have to evaluate, as a wave-uniform bit mask (bit k = iteration k)
    constexpr int UW = SIGNMAP_MAX_STEPS / 64;
    unsigned long long unsafe[UW];
#pragma unroll
    for (int w = 0; w < UW; ++w) unsafe[w] = 0ull;
    if (MAP && tile_ok) {
        // Sign map, one pass per WAVE.  The 64 lanes sample the ray through the centre of the wave's pixel tile at t_i = 0.2 + i * dt — every
        // sample of the ray in one or two byte gathers from a table the L2 holds.  Every ray of the tile stays within t * delta of the
        // centre ray (delta: the tile's half diagonal in normalised image coordinates — normalising vectors of length >= 1 does not
        // stretch their difference), and the host chose dt with dt + (5.2 + dt) * delta <= 0.9 brick edges: every point any of the tile's
        // rays reaches at a parameter in [t_i, t_i + dt] lies in sample i's brick or one of its 26 neighbours, so a clear byte (xs_signmap.h)
        // says that no iteration whose sample falls into that stretch can see an event, on any of these rays.
        const int x0 = (tile % tiles_x) * 2 * wx + (wave & 1) * wx, y0 = (tile / tiles_x) * 2 * wy + (wave >> 1) * wy;
        cfloat3 rc;
        rc.x = cfloat(((float)x0 + 0.5f * (float)(wx - 1) - a.intr.cx) / a.intr.fx);
        rc.y = cfloat(((float)y0 + 0.5f * (float)(wy - 1) - a.intr.cy) / a.intr.fy);
        rc.z = cfloat(1.f);
        const cfloat3 dc = normalized((a.Rc2v * rc + a.tc2v) - a.tc2v);
        const float inv_edge = 1.0f / (a.voxel_size * (float)(1 << a.sm.shift));
        unsigned long long flagged[2] = {~0ull, ~0ull};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h < a.sm_rounds) {
                const float tt = 0.2f + a.sm_dt * (float)(lane + 64 * h);
                const int qx = cvt_flr((a.tc2v.x.re + dc.x.re * tt) * inv_edge), qy = cvt_flr((a.tc2v.y.re + dc.y.re * tt) * inv_edge),
                          qz = cvt_flr((a.tc2v.z.re + dc.z.re * tt) * inv_edge);
                const bool in = (unsigned)qx < (unsigned)a.sm.nx && (unsigned)qy < (unsigned)a.sm.ny && (unsigned)qz < (unsigned)a.sm.nz;
                const unsigned char f = a.sm.dil[in ? (qz * a.sm.ny + qy) * a.sm.nx + qx : 0];
                flagged[h] = __builtin_amdgcn_ballot_w64(!in || f);
            }
        }
        // Iteration k (lane k % 64 of word k / 64) samples the volume at parameter t[k + 1] and holds that sample against the one at t[k]
        // (the clamped start voxel for k = 0) — the table holds the reference's running float sum t[0] = 0.2, t[j + 1] = t[j] + time_step.
        // It is marked for evaluation when t[k + 1] lies in the stretch of a flagged sample (within 2 % of a stretch's end, of the
        // neighbouring one as well: the float error of the position is ~1e-6 of it) — and iteration 0 also when the start, t[0], does.
        // That alone would miss one event: the '- to +' that ends a march (RayCaster.cu:243) is decided by the PREVIOUS sample, and a ray
        // that leaves a negative region has t[k] flagged and t[k + 1] clear.  But then iteration k - 1 was evaluated and its sample — this
        // iteration's previous one — is in hand: the march does not jump while any of the wave's unfinished rays carries a negative value
        // out of a batch (the loops below).  Every iteration that is jumped over therefore has a positive previous sample, which is what
        // the jump assumes (prev = 1).  (Marking every iteration whose t[k] OR t[k + 1] is flagged, the simpler static rule, costs one more
        // iteration behind every flagged run a ray leaves: raycast alone 42.2 us against 41.4, profiles/r04_ab_signmap_rule.txt.)
        const float inv_dt = 1.0f / a.sm_dt;
        auto is_set = [&](int i) -> bool {   // sample i flagged (or none such: beyond the samples taken)
            return (unsigned)i >= (unsigned)(64 * a.sm_rounds) || (((i < 64 ? flagged[0] : flagged[1]) >> (i & 63)) & 1ull) != 0ull;
        };
#pragma unroll
        for (int w = 0; w < UW; ++w) {
            const int k = 64 * w + lane;
            if (64 * w <= a.sm.nt - 2) {   // (wave-uniform)
                const bool runs = k <= a.sm.nt - 2;   // iteration k runs while t[k] < max_time, i.e. k <= nt - 2
                auto flagged_at = [&](float tau) -> bool {
                    const float xq = (tau - 0.2f) * inv_dt;
                    const int i = cvt_flr(xq);
                    const float fr = xq - (float)i;
                    return is_set(i) || (fr < 0.02f && is_set(i - 1)) || (fr > 0.98f && is_set(i + 1));
                };
#if defined(XS_EXPERIMENTS) && defined(XS_SIGNMAP_BOTH)   // measurement only: the simpler static rule (see above)
                const bool uns = flagged_at(a.sm.t[runs ? k + 1 : 0]) || (k == 0 ? is_set(0) : flagged_at(a.sm.t[runs ? k : 0]));
#else
                const bool uns = flagged_at(a.sm.t[runs ? k + 1 : 0]) || (k == 0 && is_set(0));   // (t[0] = 0.2 is sample 0's own position)
#endif
                unsafe[w] = __builtin_amdgcn_ballot_w64(runs && uns);
            }
        }
    }
    // first iteration >= k that has to be evaluated, or -1 (wave-uniform)
    auto next_unsafe = [&](int k) -> int {
        int r = -1;
#pragma unroll
        for (int w = UW - 1; w >= 0; --w) {
            const unsigned long long m = (k >> 6) < w ? unsafe[w] : ((k >> 6) == w ? unsafe[w] & (~0ull << (k & 63)) : 0ull);
            r = m ? 64 * w + (int)__builtin_ctzll(m) : r;
        }
        return r;
    };
    if (tile_ok && x < a.cols && y < a.rows) {
        int key = 0x7fffffff, step_index = 0;
        if (MODE == 1 || MODE == 4) {
            // contributions of the ranks are added (as int32 bit patterns) after the first event
            // along each ray has been agreed on: start from all-zero entries
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                row_ptr(a.vmap, a.mstep, y + p * a.rows)[x] = cfloat(0.f, 0.f);
                row_ptr(a.nmap, a.mstep, y + p * a.rows)[x] = cfloat(0.f, 0.f);
            }
        } else if (MODE != 3 && MODE != 5) {
            row_ptr(a.vmap, a.mstep, y)[x] = cfloat(qnan_f(), 0.f);
            row_ptr(a.nmap, a.mstep, y)[x] = cfloat(qnan_f(), 0.f);
        }
        Vol<OFF32> vol{a.value, a.grad, a.vstep, a.X, a.Y, a.Z, a.voxel_size, a.zs0, a.zs1};
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
        // zero crossing between time_curr and tn (RayCaster.cu:247-306): returns 1 if a vertex was written
        auto crossing = [&](float tc, float tn) -> int {
            cfloat Ftdt, Ft;
            if (PAIRS) {   // the crossing kernels: both samples in one round trip
                vol.interp2(ray_start + ray_dir * tn, ray_start + ray_dir * tc, Ftdt, Ft);
                if (isnan(Ftdt.re) || isnan(Ft.re)) return 0;
            } else {
                Ftdt = vol.interp(ray_start + ray_dir * tn);
                if (isnan(Ftdt.re)) return 0;
                Ft = vol.interp(ray_start + ray_dir * tc);
                if (isnan(Ft.re)) return 0;
            }
            const cfloat coef = Ft / (Ftdt - Ft);
            if (Ft.re < 0.0f || Ftdt.re > 0.0f) return 0;
            const cfloat Ts = tc - time_step * coef;
            const cfloat3 vertex_found = ray_start + ray_dir * Ts;
            const cfloat3 vw = a.Rv2w * vertex_found + a.tv2w;
            row_ptr(a.vmap, a.mstep, y)[x] = vw.x;
            row_ptr(a.vmap, a.mstep, y + a.rows)[x] = vw.y;
            row_ptr(a.vmap, a.mstep, y + 2 * a.rows)[x] = vw.z;
            const int vx = __float2int_rd(vertex_found.x.re / vs);
            const int vy = __float2int_rd(vertex_found.y.re / vs);
            const int vz = __float2int_rd(vertex_found.z.re / vs);
            if (SLAB) row_ptr(a.nmap, a.mstep, y)[x] = cfloat(qnan_f(), 0.f);  // until a normal is written
            if (vx > 1 && vy > 1 && vz > 1 && vx < a.X - 2 && vy < a.Y - 2 && vz < a.Z - 2) {
                cfloat3 t, n;
                const float half = vs * 0.5f;
                if (PAIRS) {
                    cfloat3 u;
                    cfloat F1, F2;
                    // (one pair at a time: left to itself the scheduler issues all 48 loads of the six samples at once and
                    // keeps their 96 values and 36 weights live — 284 registers, one wave per SIMD)
                    t = vertex_found; t.x += half; u = vertex_found; u.x -= half; vol.interp2(t, u, F1, F2); n.x = F1 - F2;
                    asm volatile("" : "+v"(n.x.re), "+v"(n.x.im) :: "memory");
                    t = vertex_found; t.y += half; u = vertex_found; u.y -= half; vol.interp2(t, u, F1, F2); n.y = F1 - F2;
                    asm volatile("" : "+v"(n.y.re), "+v"(n.y.im) :: "memory");
                    t = vertex_found; t.z += half; u = vertex_found; u.z -= half; vol.interp2(t, u, F1, F2); n.z = F1 - F2;
                } else {
                t = vertex_found; t.x += half; const cfloat Fx1 = vol.interp(t);
                t = vertex_found; t.x -= half; const cfloat Fx2 = vol.interp(t);
                n.x = Fx1 - Fx2;
                t = vertex_found; t.y += half; const cfloat Fy1 = vol.interp(t);
                t = vertex_found; t.y -= half; const cfloat Fy2 = vol.interp(t);
                n.y = Fy1 - Fy2;
                t = vertex_found; t.z += half; const cfloat Fz1 = vol.interp(t);
                t = vertex_found; t.z -= half; const cfloat Fz2 = vol.interp(t);
                n.z = Fz1 - Fz2;
                }
                if (squarednorm(n).re == 0) return 1;
                const cfloat3 n_g = a.Rv2w * normalized(n);
                row_ptr(a.nmap, a.mstep, y)[x] = n_g.x;
                row_ptr(a.nmap, a.mstep, y + a.rows)[x] = n_g.y;
                row_ptr(a.nmap, a.mstep, y + 2 * a.rows)[x] = n_g.z;
            }
            return 1;
        };
        if (MODE == 3) {
            const float tc = a.cross_t[y * a.cols + x];
            if (tc >= 0.f) hit = crossing(tc, tc + time_step);
        } else if (MODE == 5) {
            // the slab march left (time of the step before the crossing, marker 1) in the vertex map's x plane where
            // this rank's first event is a + to - crossing, and the provisional key (step << 1 | 1)
            key = a.keys[y * a.cols + x];
            const cfloat mark = row_ptr(a.vmap, a.mstep, y)[x];
            if (mark.im == 1.0f) {
                row_ptr(a.vmap, a.mstep, y)[x] = cfloat(0.f, 0.f);
                if (crossing(mark.re, mark.re + time_step)) { key = key & ~1; hit = 1; }
            }
        } else if (MODE == 4) {
            // Slab march, eight steps at a time.  Every rank walks the same steps (the times are a float running sum)
            // and looks only at those whose sample voxel it owns; the sample before (at most 3 planes away) is in
            // the halo it also stores.  Per step, in order: a sample outside the volume ends the march without an
            // event; an owned step whose previous sample is not stored ends it with (step << 1 | 1) (never expected:
            // halo too thin); - to + ends it with (step << 1 | 1); + to - ends it with a crossing for MODE 5 to
            // evaluate; any other owned step leaves "no event".
            int pz = gz;
            float pval = (gz >= a.zs0 && gz < a.zs1) ? vol.read_value(gx, gy, gz) : 0.f;
            bool pstored = gz >= a.zs0 && gz < a.zs1;
            bool done = false;
            if (MAP) {
                // The same march over the iterations this rank's sign map leaves (the wave's bit mask, top of the kernel): the map knows the
                // negative voxels of the planes this rank stores — owned slab + halo, marked by its own integrate calls — and an iteration
                // whose sample lies elsewhere is not this rank's to evaluate anyway.  Behind a jump the previous sample is taken as stored
                // and positive: it lies in a brick without negative voxels, within one step of an owned sample, i.e. inside the halo.
                int k = next_unsafe(0);
                if (k != 0) { pval = 1.0f; pstored = true; }
                while (k >= 0) {
                    constexpr int NS = 8;
                    float val[NS];
                    unsigned term = 0, own = 0, stored = 0;
                    const float tb = a.sm.t[k];
                    if (!done) {
                        float t = tb;
#pragma unroll
                        for (int j = 0; j < NS; ++j) {
                            const float tn = t + time_step;
                            const int jx = voxel_index<SHORT>(sx + dx * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const int jy = voxel_index<SHORT>(sy + dy * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const int jz = voxel_index<SHORT>(sz + dz * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const bool inb = (unsigned)jx < (unsigned)a.X && (unsigned)jy < (unsigned)a.Y && (unsigned)jz < (unsigned)a.Z;
                            const bool st = inb && jz >= a.zs0 && jz < a.zs1;
                            term |= ((t < max_time) && inb ? 0u : 1u) << j;
                            own |= ((inb && jz >= a.z0 && jz < a.z1) ? 1u : 0u) << j;
                            stored |= (st ? 1u : 0u) << j;
                            if (OFF32) val[j] = vol.value_at(st ? vol.offset32(jx, jy, jz) : 0u) + 1e-5f;
                            else val[j] = vol.read_value(st ? jx : 0, st ? jy : 0, st ? jz : a.zs0);
                            t += time_step;
                        }
                        unsigned thin = 0, up = 0, down = 0;
                        float prev = pval;
                        bool prev_st = pstored;
#pragma unroll
                        for (int j = 0; j < NS; ++j) {
                            const bool o = (own >> j) & 1u;
                            thin |= ((o && !prev_st) ? 1u : 0u) << j;
                            up |= ((o && prev_st && prev < 0.f && val[j] > 0.f) ? 1u : 0u) << j;
                            down |= ((o && prev_st && prev > 0.f && val[j] < 0.f) ? 1u : 0u) << j;
                            prev = val[j];
                            prev_st = (stored >> j) & 1u;
                        }
                        const unsigned ev = term | thin | up | down;
                        if (ev) {
                            const int e = __ffs(ev) - 1;
                            if (!((term >> e) & 1u)) {
                                key = ((k + e) << 1) | 1;
                                if ((down >> e) & 1u) {
                                    float tce = tb;   // time_curr of iteration k + e: the same e additions once more
                                    for (int j = 0; j < e; ++j) tce += time_step;
                                    row_ptr(a.vmap, a.mstep, y)[x] = cfloat(tce, 1.0f);  // for MODE 5
                                }
                            }
                            done = true;
                        }
                        pval = val[NS - 1];
                        pstored = (stored >> (NS - 1)) & 1u;
                    }
                    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
                    // no jump while an unfinished ray of the wave is inside a negative region (its next iteration may be the '- to +' exit)
                    const bool inside = __builtin_amdgcn_ballot_w64(!done && pval < 0.f) != 0ull;
                    const int kn = inside ? (k + NS <= a.sm.nt - 2 ? k + NS : -1) : next_unsafe(k + NS);
                    if (kn != k + NS) { pval = 1.0f; pstored = true; }
                    k = kn;
                }
                done = true;
            }
            while (!done && time_curr < max_time) {
                constexpr int NS = 8;
                float tc[NS], val[NS];
                unsigned term = 0, own = 0, stored = 0;
                float t = time_curr;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    tc[j] = t;
                    const float tn = t + time_step;
                    const int jx = voxel_index<SHORT>(sx + dx * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    const int jy = voxel_index<SHORT>(sy + dy * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    const int jz = voxel_index<SHORT>(sz + dz * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    const bool inb = (unsigned)jx < (unsigned)a.X && (unsigned)jy < (unsigned)a.Y && (unsigned)jz < (unsigned)a.Z;
                    const bool st = inb && jz >= a.zs0 && jz < a.zs1;
                    term |= ((t < max_time) && inb ? 0u : 1u) << j;
                    own |= ((inb && jz >= a.z0 && jz < a.z1) ? 1u : 0u) << j;
                    stored |= (st ? 1u : 0u) << j;
                    if (OFF32) val[j] = vol.value_at(st ? vol.offset32(jx, jy, jz) : 0u) + 1e-5f;
                    else val[j] = vol.read_value(st ? jx : 0, st ? jy : 0, st ? jz : a.zs0);
                    t += time_step;
                }
                unsigned thin = 0, up = 0, down = 0;
                float prev = pval;
                bool prev_st = pstored;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const bool o = (own >> j) & 1u;
                    thin |= ((o && !prev_st) ? 1u : 0u) << j;
                    up |= ((o && prev_st && prev < 0.f && val[j] > 0.f) ? 1u : 0u) << j;
                    down |= ((o && prev_st && prev > 0.f && val[j] < 0.f) ? 1u : 0u) << j;
                    prev = val[j];
                    prev_st = (stored >> j) & 1u;
                }
                const unsigned ev = term | thin | up | down;
                if (ev) {
                    const int e = __ffs(ev) - 1;
                    if (!((term >> e) & 1u)) {
                        key = ((step_index + e) << 1) | 1;
                        if ((down >> e) & 1u) {
                            float tce = tc[0];
#pragma unroll
                            for (int j = 1; j < NS; ++j) tce = (e == j) ? tc[j] : tce;
                            row_ptr(a.vmap, a.mstep, y)[x] = cfloat(tce, 1.0f);  // for MODE 5
                        }
                    }
                    done = true;
                }
                pval = val[NS - 1];
                pstored = (stored >> (NS - 1)) & 1u;
                time_curr = t;
                step_index += NS;
            }
            (void)pz;
        } else if (SLAB) {
            // slab mode: every rank walks the same sequence of steps (the times are a float running
            // sum, so all of them must be taken) but evaluates only those whose sample voxel it owns;
            // the previous sample lies at most 3 planes away, inside the halo it also stores
            int pgx = gx, pgy = gy, pgz = gz;
            for (; time_curr < max_time; time_curr += time_step, ++step_index) {
                const float tn = time_curr + time_step;
                gx = voxel_index<SHORT>(sx + dx * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);  // floor(p / vs) without the divide
                gy = voxel_index<SHORT>(sy + dy * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                gz = voxel_index<SHORT>(sz + dz * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                if (!(gx >= 0 && gy >= 0 && gz >= 0 && gx < a.X && gy < a.Y && gz < a.Z)) break;
                const bool owned = gz >= a.z0 && gz < a.z1;
                const int qx = pgx, qy = pgy, qz = pgz;
                pgx = gx; pgy = gy; pgz = gz;
                if (!owned) continue;
                key = (step_index << 1) | 1;  // provisional: an event without a vertex
                if (qz < a.zs0 || qz >= a.zs1) break;  // halo too thin: never expected
                const float tsdf_prev = vol.read_value(qx, qy, qz);
                const float tsdf = vol.read_value(gx, gy, gz);
                if (tsdf_prev < 0.f && tsdf > 0.f) break;
                if (tsdf_prev > 0.f && tsdf < 0.f) {
                    if (crossing(time_curr, tn)) { key = step_index << 1; hit = 1; }
                    break;
                }
                key = 0x7fffffff;  // no event at this step
            }
        } else {
            // The sample positions do not depend on the loaded values, so four steps are issued at a
            // time (four gathers in flight per lane instead of one dependent chain) and then tested
            // in order; whatever lies behind the first event is discarded.  The times are the same
            // float running sum the one-step loop forms.
            float cross = -1.f;
            bool done = false;
            int nsteps = 0;   // iterations of the reference's loop so far (only stored when a.steps is given)
            if (MAP) {
                // The march over the iterations the wave has to evaluate, eight at a time; stretches in between are jumped over.  An iteration
                // that is left out has its sample in a brick without negative voxels, so the value the reference carries out of it is positive:
                // the first iteration behind a jump starts from "positive" (its magnitude never matters), iteration 0 from the clamped start
                // voxel as in the reference.  Within a batch the times are the reference's own running sum from t[k].
                constexpr int NS = 8;
                int k = next_unsafe(0);
                float prev = 1.0f;
                if (k == 0) prev = vol.read_value(gx, gy, gz);
                bool finished = false;
                nsteps = a.sm.nt - 1;   // (no event: the loop runs until its own condition ends it)
                while (k >= 0) {
                    float val[NS];
                    unsigned oob = 0;
                    const float tb = a.sm.t[k];
                    if (!finished) {
                        float t = tb;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const float tn = t + time_step;
                            const int jx = voxel_index<SHORT>(sx + dx * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const int jy = voxel_index<SHORT>(sy + dy * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const int jz = voxel_index<SHORT>(sz + dz * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                            const bool ok = (t < max_time) && (unsigned)jx < (unsigned)a.X && (unsigned)jy < (unsigned)a.Y && (unsigned)jz < (unsigned)a.Z;
                            oob |= (ok ? 0u : 1u) << i;
                            if (OFF32) val[i] = vol.value_at(ok ? vol.offset32(jx, jy, jz) : 0u) + 1e-5f;
                            else val[i] = vol.read_value(ok ? jx : 0, ok ? jy : 0, ok ? jz : a.zs0);
                            t += time_step;
                        }
                        unsigned down = 0, up = 0;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            down |= ((prev > 0.f && val[i] < 0.f) ? 1u : 0u) << i;
                            up |= ((prev < 0.f && val[i] > 0.f) ? 1u : 0u) << i;
                            prev = val[i];
                        }
                        const unsigned ev = oob | down | up;
                        if (ev) {
                            const int e = __ffs(ev) - 1;
                            float tce = tb;   // time_curr of iteration k + e: the same e additions once more (once per ray)
                            for (int i = 0; i < e; ++i) tce += time_step;
                            if (!((oob >> e) & 1u) && ((down >> e) & 1u)) cross = tce;
                            nsteps = k + e + (tce < max_time ? 1 : 0);   // the reference's loop runs iteration k + e unless its own condition ends it first
                            finished = true;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) break;
                    // no jump while an unfinished ray of the wave is inside a negative region (its next iteration may be the '- to +' exit)
                    const bool inside = __builtin_amdgcn_ballot_w64(!finished && prev < 0.f) != 0ull;
                    const int kn = inside ? (k + NS <= a.sm.nt - 2 ? k + NS : -1) : next_unsafe(k + NS);
                    if (kn != k + NS) prev = 1.0f;
                    k = kn;
                }
                if (MODE == 0 && cross >= 0.f) hit = crossing(cross, cross + time_step);   // (behind the loop: the wave's hit lanes together)
                done = true;
                time_curr = max_time;
            }
            float tsdf = vol.read_value(gx, gy, gz);
            while (!done && time_curr < max_time) {
                // eight steps at once, straight-line: positions, clamped (always valid) gathers, then one
                // event mask — out of range / past the end, - to + (no vertex), + to - (crossing) — whose
                // lowest set bit is the first event along the ray; one branch per eight steps
                constexpr int NS = 8;
                float tc[NS], val[NS];
                unsigned oob = 0;
                float t = time_curr;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    tc[j] = t;
                    const float tn = t + time_step;
                    const int jx = voxel_index<SHORT>(sx + dx * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    const int jy = voxel_index<SHORT>(sy + dy * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    const int jz = voxel_index<SHORT>(sz + dz * tn, vs, a.inv_vs_lo, a.inv_vs_hi, a.dv);
                    // (one unsigned compare per axis; a step outside reads voxel (0, 0, first plane), always resident,
                    // and its value is never looked at)
                    const bool ok = (t < max_time) && (unsigned)jx < (unsigned)a.X && (unsigned)jy < (unsigned)a.Y && (unsigned)jz < (unsigned)a.Z;
                    oob |= (ok ? 0u : 1u) << j;
                    if (OFF32) val[j] = vol.value_at(ok ? vol.offset32(jx, jy, jz) : 0u) + 1e-5f;
                    else val[j] = vol.read_value(ok ? jx : 0, ok ? jy : 0, ok ? jz : a.zs0);
                    t += time_step;
                }
                unsigned down = 0, up = 0;  // + to -, - to +
                float prev = tsdf;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    down |= ((prev > 0.f && val[j] < 0.f) ? 1u : 0u) << j;
                    up |= ((prev < 0.f && val[j] > 0.f) ? 1u : 0u) << j;
                    prev = val[j];
                }
                const unsigned ev = oob | down | up;
                if (ev) {
                    const int e = __ffs(ev) - 1;
                    float tce = tc[0];
#pragma unroll
                    for (int j = 1; j < NS; ++j) tce = (e == j) ? tc[j] : tce;
                    if (!((oob >> e) & 1u) && ((down >> e) & 1u)) {
                        if (MODE == 2) cross = tce; else hit = crossing(tce, tce + time_step);
                    }
                    // the reference's loop runs iteration e of this batch unless its own condition (time_curr < max_time) ends it first
                    nsteps += e + (tce < max_time ? 1 : 0);
                    done = true;
                } else
                    nsteps += NS;
                tsdf = val[NS - 1];
                time_curr = t;
            }
            if (MODE == 2 || (MAP && a.cross_t)) a.cross_t[y * a.cols + x] = cross;   // (the fused launch fills the workspace too: same contract)
            if (a.steps) a.steps[y * a.cols + x] = nsteps;
        }
        if (SLAB) a.keys[y * a.cols + x] = key;  // (MODE 5 rewrites the key it read: even where the crossing gave a vertex)
    }
    if (MODE == 0 && a.pyr.mid[0]) {
        // The model-map pyramid (resizeVMap / resizeNMap twice, KinectFusionReconstruction.cpp:272-277) of this workgroup's own tile: a
        // level-2 pixel needs the 4 x 4 level-0 pixels under it, and the tile's origin and size are multiples of four — so the two
        // halvings need nothing another workgroup writes, and the frame's tail loses a launch (~16 us: dispatch + a kernel that reads the
        // maps back from memory).  The tile's vertices / normals went out with plain stores: once they have completed (release at
        // workgroup scope + barrier) the lanes of wave 0 read them back through this CU's L1 / L2.  One lane per level-2 pixel and map,
        // through the pyramid kernel's own function: the same values.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // level 1: the tile's 64 pixels of each map, one per lane of waves 0 and 1 (wave 0: vertices, wave 1: normals); their values
        // pass to level 2 through LDS: the 16 pixels of each map, one per lane of the first half of wave 0
        __shared__ cfloat3 s_l1[2][64];
        __shared__ unsigned char s_ok1[2][64];
        const int rows1 = a.pyr.rows0 / 2, cols1 = a.pyr.cols0 / 2;
        const int n1x = wx;                                        // level-1 pixels per tile: (2 wx / 2) x (2 wy / 2) = wx x wy = 64
        const int x1o = (tile % tiles_x) * wx, y1o = (tile / tiles_x) * wy;   // the tile's first level-1 pixel
        if (tile_ok && threadIdx.x < 128) {
            const int m = (int)threadIdx.x >> 6, q = (int)threadIdx.x & 63;
            const int x1 = x1o + q % n1x, y1 = y1o + q / n1x;
            cfloat3 n;
            bool ok = false;
            if (x1 < cols1 && y1 < rows1) ok = m == 0 ? pyramid_level1_pixel<false>(a.pyr, 0, x1, y1, n) : pyramid_level1_pixel<true>(a.pyr, 1, x1, y1, n);
            if (ok) s_l1[m][q] = n;
            s_ok1[m][q] = ok ? 1 : 0;
        }
        __syncthreads();
        if (tile_ok && threadIdx.x < 32) {
            const int m = (int)threadIdx.x >> 4, r = (int)threadIdx.x & 15;
            const int n2x = n1x >> 1;
            const int lx = r % n2x, ly = r / n2x;                  // level-2 pixel inside the tile
            const int x2 = (x1o >> 1) + lx, y2 = (y1o >> 1) + ly;
            if (x2 < cols1 / 2 && y2 < rows1 / 2) {
                cfloat3 l1[4];
                bool ok[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int q = (2 * ly + (c >> 1)) * n1x + 2 * lx + (c & 1);
                    ok[c] = s_ok1[m][q] != 0;
                    if (ok[c]) l1[c] = s_l1[m][q];
                }
                if (m == 0) pyramid_level2_pixel<false>(a.pyr, 0, x2, y2, l1, ok);
                else pyramid_level2_pixel<true>(a.pyr, 1, x2, y2, l1, ok);
            }
        }
    }
    if (a.hits) {
        // one atomic per workgroup: same-address atomics cost ~12 ns each at the memory side, and one
        // per wave (4800 of them) would by itself take longer than the kernel
        __shared__ unsigned s_hits[4];
        const unsigned s = wave_sum_u32(hit);
        if (lane == 0) s_hits[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = (s_hits[0] + s_hits[1]) + (s_hits[2] + s_hits[3]);
            if (t) atomicAdd(a.hits, (unsigned long long)t);
        }
    }
}


static int ray_wshift() {
    static const int env_ws = exp_env_int("XS_RAY_WSHIFT", 3);
    return (env_ws >= 0 && env_ws <= 6) ? env_ws : 3;
}
// spacing of the per-wave sign-map samples (k_raycast, MAP): dt + (5.2 + dt) * delta <= 0.9 brick edges, delta = the pixel tile's half
// diagonal in normalised image coordinates (+ 1 %); at most two samples per lane must reach from t = 0.2 past max_time = 5
static bool sign_map_spacing(float fx, float fy, int wshift, float voxel_size, int shift, float &dt_out, int &rounds_out) {
    const float hx = 0.5f * (float)((1 << wshift) - 1) / fabsf(fx), hy = 0.5f * (float)((64 >> wshift) - 1) / fabsf(fy);
    const float delta = 1.01f * sqrtf(hx * hx + hy * hy) + 1e-6f, edge = voxel_size * (float)(1 << shift);
    const float dt = (0.9f * edge - 5.2f * delta) / (1.0f + delta);
    const int need = dt > 0.0f ? (int)ceilf(4.9f / dt) + 1 : 1 << 30;
    if (!(need <= 128)) return false;
    dt_out = dt; rounds_out = need <= 64 ? 1 : 2;
    return true;
}
extern "C" int xs_raycast_signmap_shift(const float *intr4, float voxel_size, float tranc_dist) {
    if (!intr4 || !(voxel_size > 0.0f) || signmap_steps(tranc_dist * 0.8f) == 0) return 0;
    float dt; int rounds;
    for (int shift = 2; shift <= 6; ++shift)
        if (sign_map_spacing(intr4[0], intr4[1], ray_wshift(), voxel_size, shift, dt, rounds)) return shift;
    return 0;
}


// do the resident planes of one volume array span at most 4 GiB (and the 24-bit products hold)?
static bool fits32(const RaycastArgs &a) {
    const unsigned long long rows = (unsigned long long)(a.zs1 - a.zs0) * (unsigned long long)a.Y;
    return (a.vstep % 4) == 0 && rows < (1ull << 24) && (a.vstep >> 2) < (1ull << 24) && rows * a.vstep <= (1ull << 32);
}
static void set_inv_vs(RaycastArgs &a) {
    float lo = 1.0f / a.voxel_size, hi = lo;
    for (int i = 0; i < 4; ++i) { lo = nextafterf(lo, 0.f); hi = nextafterf(hi, INFINITY); }
    a.inv_vs_lo = lo; a.inv_vs_hi = hi;
    a.dv = xs_const_div_get(a.voxel_size);
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
 * vertex.  workspace: optional rows*cols floats; with it the ray is split into a march kernel
 * and a crossing kernel (same arithmetic, higher occupancy).  No synchronisation (neither does
 * the reference, :367). */
// the reference's launcher proper: no sign map, no pyramid, no event (xs_raycast_ex takes those as an options struct)
extern "C" int xs_raycast(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                          float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                          float *vmap, float *nmap, size_t map_step, int rows, int cols, unsigned long long *hits_dev, float *workspace,
                          void *stream) {
    xs_raycast_opts o = {};
    o.struct_bytes = sizeof(o);
    return xs_raycast_ex(intr4, Rc2v18, tc2v6, Rv2w18, tv2w6, tranc_dist, res, voxel_size, value, grad, vol_step, vmap, nmap, map_step, rows, cols, hits_dev,
                         workspace, &o, stream);
}
extern "C" int xs_raycast_ex(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                             float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                             float *vmap, float *nmap, size_t map_step, int rows, int cols, unsigned long long *hits_dev, float *workspace,
                             xs_raycast_opts *opts, void *stream) {
    xs_raycast_opts none = {};
    none.struct_bytes = sizeof(none);
    xs_raycast_opts &o = opts ? *opts : none;
    if (o.struct_bytes != sizeof(xs_raycast_opts)) return xs_set_error(hipErrorInvalidValue, "xs_raycast_ex: opts->struct_bytes is not sizeof(xs_raycast_opts)");
    o.pyramid_built = 0;
    if (!intr4 || !Rc2v18 || !tc2v6 || !Rv2w18 || !tv2w6 || !res || !value || !grad || !vmap || !nmap)
        return xs_set_error(hipErrorInvalidValue, "xs_raycast: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    RaycastArgs a;
    ld_mat(Rc2v18, a.Rc2v); ld_vec(tc2v6, a.tc2v); ld_mat(Rv2w18, a.Rv2w); ld_vec(tv2w6, a.tv2w);
    a.X = res[0]; a.Y = res[1]; a.Z = res[2];
    a.voxel_size = voxel_size;
    a.time_step = tranc_dist * 0.8f;  // RayCaster.cu:350
    set_inv_vs(a);
    a.cols = cols; a.rows = rows;
    a.value = value; a.grad = grad; a.vstep = vol_step;
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.vmap = (cfloat *)vmap; a.nmap = (cfloat *)nmap; a.mstep = map_step;
    a.zs0 = 0; a.zs1 = res[2]; a.z0 = 0; a.z1 = res[2]; a.keys = nullptr;
    a.hits = hits_dev; a.cross_t = workspace; a.steps = o.steps_dev;
    a.sm = SignMap{};
    a.pyr = PyramidArgs{};
    if (o.signmap && workspace) {
        // the map's time table was written for one truncation distance (xs_signmap_reset): any other would resume at wrong times
        if (o.signmap_tranc_dist * 0.8f != a.time_step || o.signmap_shift < 2 || o.signmap_shift > 6)
            return xs_set_error(hipErrorInvalidValue, "xs_raycast: the sign map was prepared for another truncation distance");
        static thread_local float nt_for = 0.0f;
        static thread_local int nt = 0;
        if (nt_for != a.time_step) { nt = signmap_steps(a.time_step); nt_for = a.time_step; }
        if (nt == 0) return xs_set_error(hipErrorInvalidValue, "xs_raycast: the march has more steps than the sign map's time table holds");
        a.sm = signmap_view(const_cast<void *>(o.signmap), res, o.signmap_shift, nt);
    }
    a.wshift = ray_wshift();
    a.sm_dt = 0.0f; a.sm_rounds = 0;
    if (a.sm.dil && !sign_map_spacing(a.intr.fx, a.intr.fy, a.wshift, voxel_size, a.sm.shift, a.sm_dt, a.sm_rounds))
        return xs_set_error(hipErrorInvalidValue, "xs_raycast: the sign map's bricks are too small for a wave's pixel tile (xs_raycast_signmap_shift gives a usable shift)");
    dim3 block(256), grid(div_up(div_up(cols, 2 << a.wshift) * div_up(rows, 128 >> a.wshift), 8) * 8);
    const bool off32 = fits32(a);
    if (workspace) {
        // march (few registers, many waves, eight gathers in flight per lane) then the crossings
        a.hits = nullptr;
        static const bool env_pair = exp_env_set("XS_RAY_MAP_TWO_KERNELS");   // measurement: the mapped march and the crossing as two launches
        if (a.sm.dil && !env_pair) {
            // with the sign map the march is a few batches long: it and the crossing are one launch (five waves per SIMD hold the
            // frame's 4 800 waves either way) — no second dispatch, no crossing-time plane written and read back
            a.hits = hits_dev;
            static const bool env_no_pyr = exp_env_set("XS_RAY_NO_PYRAMID");   // A/B aid: the pyramid stays a launch of its own
            if (o.pyr_vmap1 && o.pyr_nmap1 && o.pyr_vmap2 && o.pyr_nmap2 && !env_no_pyr) {
                a.pyr.mid[0] = (cfloat *)o.pyr_vmap1; a.pyr.mid[1] = (cfloat *)o.pyr_nmap1; a.pyr.out[0] = (cfloat *)o.pyr_vmap2; a.pyr.out[1] = (cfloat *)o.pyr_nmap2;
                a.pyr.mstep = o.pyr_step1; a.pyr.ostep = o.pyr_step2;
                a.pyr.in[0] = a.vmap; a.pyr.in[1] = a.nmap; a.pyr.istep = map_step; a.pyr.rows0 = rows; a.pyr.cols0 = cols;
                o.pyramid_built = 1;
            }
            void (*kern)(const RaycastArgs) = off32 ? ((a.dv.ok & 2u) ? k_raycast<0, true, true, true> : k_raycast<0, true, false, true>) : k_raycast<0, false, false, true>;
            if (o.completion_event) hipExtLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)stream, nullptr, (hipEvent_t)o.completion_event, 0, a);
            else hipLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)stream, a);
            XS_CHECK(hipGetLastError());
            return 0;
        }
        if (a.sm.dil) {
            if (off32) hipLaunchKernelGGL(((a.dv.ok & 2u) ? k_raycast<2, true, true, true> : k_raycast<2, true, false, true>), grid, block, 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((k_raycast<2, false, false, true>), grid, block, 0, (hipStream_t)stream, a);
        } else if (off32) hipLaunchKernelGGL(((a.dv.ok & 2u) ? k_raycast<2, true, true> : k_raycast<2, true, false>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_raycast<2, false, false>), grid, block, 0, (hipStream_t)stream, a);
        a.hits = hits_dev;
        if (off32) hipLaunchKernelGGL((k_raycast<3, true, false>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_raycast<3, false, false>), grid, block, 0, (hipStream_t)stream, a);
    } else if (off32)
        hipLaunchKernelGGL((k_raycast<0, true, false>), grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((k_raycast<0, false, false>), grid, block, 0, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}

/* Slab form of raycast for a z-sharded volume (no reference counterpart: the reference is
 * single-GPU; semantics are those of RayCaster.cu:197-310 restricted to one rank's steps).
 * value / grad hold planes [zs0, zs1) (owned slab + halo); only march steps whose sample voxel
 * lies in the owned planes [z0, z1) are evaluated.  keys_dev[rows*cols] receives, per pixel,
 * (step << 1 | no_vertex) of the first event among those steps or INT_MAX; vmap / nmap receive
 * this rank's vertex / normal for its own event and zeros elsewhere.  The caller takes the
 * minimum of the keys over ranks, calls xs_raycast_compose_mask, adds the maps over ranks as
 * int32, then xs_raycast_compose_finish.  Halo needed: 6 planes. */
extern "C" int xs_raycast_slab(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                               float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                               int zs0, int zs1, int z0, int z1, float *vmap, float *nmap, size_t map_step, int rows, int cols,
                               int *keys_dev, void *stream) {
    xs_raycast_opts o = {};
    o.struct_bytes = sizeof(o);
    return xs_raycast_slab_ex(intr4, Rc2v18, tc2v6, Rv2w18, tv2w6, tranc_dist, res, voxel_size, value, grad, vol_step, zs0, zs1, z0, z1, vmap, nmap, map_step, rows,
                              cols, keys_dev, &o, stream);
}
extern "C" int xs_raycast_slab_ex(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                                  float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                                  int zs0, int zs1, int z0, int z1, float *vmap, float *nmap, size_t map_step, int rows, int cols,
                                  int *keys_dev, const xs_raycast_opts *opts, void *stream) {
    xs_raycast_opts none = {};
    none.struct_bytes = sizeof(none);
    const xs_raycast_opts &o = opts ? *opts : none;
    if (o.struct_bytes != sizeof(xs_raycast_opts)) return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab_ex: opts->struct_bytes is not sizeof(xs_raycast_opts)");
    if (!intr4 || !Rc2v18 || !tc2v6 || !Rv2w18 || !tv2w6 || !res || !value || !grad || !vmap || !nmap || !keys_dev)
        return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab: null pointer");
    if (zs0 < 0 || zs1 > res[2] || z0 < zs0 || z1 > zs1 || z1 < z0) return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab: bad slab");
    if (rows <= 0 || cols <= 0) return 0;
    RaycastArgs a;
    ld_mat(Rc2v18, a.Rc2v); ld_vec(tc2v6, a.tc2v); ld_mat(Rv2w18, a.Rv2w); ld_vec(tv2w6, a.tv2w);
    a.X = res[0]; a.Y = res[1]; a.Z = res[2];
    a.voxel_size = voxel_size;
    a.time_step = tranc_dist * 0.8f;
    set_inv_vs(a);
    a.cols = cols; a.rows = rows;
    a.value = value; a.grad = grad; a.vstep = vol_step;
    a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.vmap = (cfloat *)vmap; a.nmap = (cfloat *)nmap; a.mstep = map_step;
    a.zs0 = zs0; a.zs1 = zs1; a.z0 = z0; a.z1 = z1; a.keys = keys_dev;
    a.hits = nullptr; a.cross_t = nullptr; a.steps = nullptr;
    a.sm = SignMap{};
    a.pyr = PyramidArgs{};
    a.sm_dt = 0.0f; a.sm_rounds = 0;
    if (o.signmap) {
        // this rank's sign map (marked by its own integrate calls: owned slab + halo): the march evaluates the iterations it leaves
        if (o.signmap_tranc_dist * 0.8f != a.time_step || o.signmap_shift < 2 || o.signmap_shift > 6)
            return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab: the sign map was prepared for another truncation distance");
        const int nt = signmap_steps(a.time_step);
        if (nt == 0) return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab: the march has more steps than the sign map's time table holds");
        a.sm = signmap_view(const_cast<void *>(o.signmap), res, o.signmap_shift, nt);
    }
    a.wshift = ray_wshift();
    dim3 block(256), grid(div_up(div_up(cols, 2 << a.wshift) * div_up(rows, 128 >> a.wshift), 8) * 8);
    if (a.sm.dil && !sign_map_spacing(a.intr.fx, a.intr.fy, a.wshift, voxel_size, a.sm.shift, a.sm_dt, a.sm_rounds))
        return xs_set_error(hipErrorInvalidValue, "xs_raycast_slab: the sign map's bricks are too small for a wave's pixel tile (xs_raycast_signmap_shift gives a usable shift)");
    if (fits32(a)) {
        if (a.sm.dil) hipLaunchKernelGGL(((a.dv.ok & 2u) ? k_raycast<4, true, true, true> : k_raycast<4, true, false, true>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(((a.dv.ok & 2u) ? k_raycast<4, true, true> : k_raycast<4, true, false>), grid, block, 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL((k_raycast<5, true, false>), grid, block, 0, (hipStream_t)stream, a);
    } else {
        if (a.sm.dil) hipLaunchKernelGGL((k_raycast<4, false, false, true>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_raycast<4, false, false>), grid, block, 0, (hipStream_t)stream, a);
        hipLaunchKernelGGL((k_raycast<5, false, false>), grid, block, 0, (hipStream_t)stream, a);
    }
    XS_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) k_compose_mask(const int *own_keys, const int *min_keys, cfloat *vmap, cfloat *nmap, size_t mstep, int rows, int cols) {
    const int x = threadIdx.x + blockIdx.x * 64, y = threadIdx.y + blockIdx.y * 4;
    if (x >= cols || y >= rows) return;
    const int k = own_keys[y * cols + x], m = min_keys[y * cols + x];
    if (k == m && !(k & 1) && k != 0x7fffffff) return;  // this rank holds the ray's first event and it has a vertex
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        row_ptr(vmap, mstep, y + p * rows)[x] = cfloat(0.f, 0.f);
        row_ptr(nmap, mstep, y + p * rows)[x] = cfloat(0.f, 0.f);
    }
}
__global__ void __launch_bounds__(256) k_compose_finish(const int *min_keys, cfloat *vmap, cfloat *nmap, size_t mstep, int rows, int cols,
                                                        unsigned long long *hits) {
    // a few hundred workgroups stride over the 64x4 pixel tiles and add their hit count with ONE atomic each
    // (one per wave to the same word — 4800 of them at ~12 ns — made this the second longest kernel of the frame)
    const int tiles_x = (cols + 63) / 64, tiles_y = (rows + 3) / 4;
    unsigned hit = 0;
    for (int tile = blockIdx.x; tile < tiles_x * tiles_y; tile += gridDim.x) {
        const int x = threadIdx.x + (tile % tiles_x) * 64, y = threadIdx.y + (tile / tiles_x) * 4;
        if (x < cols && y < rows) {
            const int m = min_keys[y * cols + x];
            if ((m & 1) || m == 0x7fffffff) {  // no vertex on this ray: the NaN sentinel of RayCaster.cu:204-205
                row_ptr(vmap, mstep, y)[x] = cfloat(qnan_f(), 0.f);
                row_ptr(nmap, mstep, y)[x] = cfloat(qnan_f(), 0.f);
            } else
                ++hit;
        }
    }
    if (hits) {
        __shared__ unsigned s_hits[4];
        const int tid = threadIdx.y * 64 + threadIdx.x;
        const unsigned s = wave_sum_u32(hit);
        if ((tid & 63) == 0) s_hits[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            const unsigned t = (s_hits[0] + s_hits[1]) + (s_hits[2] + s_hits[3]);
            if (t) atomicAdd(hits, (unsigned long long)t);
        }
    }
}
// ---- owner-compacted exchange of the composite (round 4) ---------------------------------------------------------------------------
// Adding the ranks' maps moves every pixel's 48 bytes through a ring all-reduce although one rank at most has anything but zeros there:
// 2 (N - 1) / N x 14.7 MB per rank and frame.  Instead every rank packs the pixels it owns — {pixel index, vertex (3 complex), normal
// (3 complex)}: 13 words — the ranks' packs are gathered (variable sizes: the counts travel first, as an int32 sum with one non-zero
// entry per rank) and a scatter writes them into the maps: (N - 1) / N x 52 bytes x hits received per rank, about half.
enum { COMPOSE_ENTRY_WORDS = 13 };
__global__ void __launch_bounds__(256) k_compose_pack(const int *own_keys, const int *min_keys, const cfloat *vmap, const cfloat *nmap, size_t mstep, int rows,
                                                      int cols, unsigned *entries, int *count) {
    // one workgroup-wide slot reservation per 256 pixels: a ballot per wave, one LDS exchange, one atomic
    __shared__ unsigned s_base[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int p0 = blockIdx.x * 256; p0 < rows * cols; p0 += gridDim.x * 256) {
        const int p = p0 + tid;
        bool mine = false;
        int x = 0, y = 0;
        if (p < rows * cols) {
            const int k = own_keys[p], m = min_keys[p];
            mine = k == m && !(k & 1) && k != 0x7fffffff;
            y = p / cols; x = p - y * cols;
        }
        const unsigned long long b = __builtin_amdgcn_ballot_w64(mine);
        if (lane == 0) s_base[wave] = (unsigned)__popcll(b);
        __syncthreads();
        if (tid == 0) {
            const unsigned n0 = s_base[0], n1 = s_base[1], n2 = s_base[2], n3 = s_base[3];
            const unsigned tot = n0 + n1 + n2 + n3;
            const unsigned base = tot ? (unsigned)atomicAdd(count, (int)tot) : 0u;
            s_base[0] = base; s_base[1] = base + n0; s_base[2] = base + n0 + n1; s_base[3] = base + n0 + n1 + n2;
        }
        __syncthreads();
        if (mine) {
            unsigned *e = entries + (size_t)(s_base[wave] + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) * COMPOSE_ENTRY_WORDS;
            e[0] = (unsigned)p;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const cfloat v = row_ptr(vmap, mstep, y + q * rows)[x], n = row_ptr(nmap, mstep, y + q * rows)[x];
                e[1 + 2 * q] = __float_as_uint(v.re); e[2 + 2 * q] = __float_as_uint(v.im);
                e[7 + 2 * q] = __float_as_uint(n.re); e[8 + 2 * q] = __float_as_uint(n.im);
            }
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_compose_scatter(const unsigned *entries, long n, cfloat *vmap, cfloat *nmap, size_t mstep, int rows, int cols) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const unsigned *e = entries + i * COMPOSE_ENTRY_WORDS;
        const unsigned p = e[0];
        if (p >= (unsigned)(rows * cols)) continue;   // (never: the packs hold pixel indices)
        const int y = (int)(p / (unsigned)cols), x = (int)(p - (unsigned)y * (unsigned)cols);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            row_ptr(vmap, mstep, y + q * rows)[x] = cfloat(__uint_as_float(e[1 + 2 * q]), __uint_as_float(e[2 + 2 * q]));
            row_ptr(nmap, mstep, y + q * rows)[x] = cfloat(__uint_as_float(e[7 + 2 * q]), __uint_as_float(e[8 + 2 * q]));
        }
    }
}
/* The pixels this rank owns with a vertex (own key == min key, event with a vertex), packed: entries_dev receives 13 32-bit words per
 * pixel {pixel index y * cols + x, vertex x / y / z (re, im), normal x / y / z (re, im)} in no particular order, *count_dev is advanced
 * by their number (the caller zeroes it; room for rows * cols entries). */
extern "C" int xs_raycast_compose_pack(const int *own_keys_dev, const int *min_keys_dev, const float *vmap, const float *nmap, size_t map_step, int rows,
                                       int cols, void *entries_dev, int *count_dev, void *stream) {
    if (!own_keys_dev || !min_keys_dev || !vmap || !nmap || !entries_dev || !count_dev) return xs_set_error(hipErrorInvalidValue, "xs_raycast_compose_pack: null pointer");
    if (rows <= 0 || cols <= 0) return 0;
    const int blocks = div_up(rows * cols, 256);
    hipLaunchKernelGGL(k_compose_pack, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, (hipStream_t)stream, own_keys_dev, min_keys_dev, (const cfloat *)vmap,
                       (const cfloat *)nmap, map_step, rows, cols, (unsigned *)entries_dev, count_dev);
    XS_CHECK(hipGetLastError());
    return 0;
}
/* n packed entries (any ranks') written into the maps */
extern "C" int xs_raycast_compose_scatter(const void *entries_dev, long n, float *vmap, float *nmap, size_t map_step, int rows, int cols, void *stream) {
    if (!entries_dev || !vmap || !nmap) return xs_set_error(hipErrorInvalidValue, "xs_raycast_compose_scatter: null pointer");
    if (n <= 0 || rows <= 0 || cols <= 0) return 0;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_compose_scatter, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, (const unsigned *)entries_dev, n, (cfloat *)vmap,
                       (cfloat *)nmap, map_step, rows, cols);
    XS_CHECK(hipGetLastError());
    return 0;
}
extern "C" size_t xs_raycast_compose_entry_bytes(void) { return COMPOSE_ENTRY_WORDS * sizeof(unsigned); }
/* keep this rank's vertex / normal only where it owns the ray's first event (own key == min key) */
extern "C" int xs_raycast_compose_mask(const int *own_keys_dev, const int *min_keys_dev, float *vmap, float *nmap, size_t map_step, int rows,
                                       int cols, void *stream) {
    if (!own_keys_dev || !min_keys_dev || !vmap || !nmap) return xs_set_error(hipErrorInvalidValue, "xs_raycast_compose_mask: null pointer");
    dim3 block(64, 4), grid(div_up(cols, 64), div_up(rows, 4));
    hipLaunchKernelGGL(k_compose_mask, grid, block, 0, (hipStream_t)stream, own_keys_dev, min_keys_dev, (cfloat *)vmap, (cfloat *)nmap, map_step, rows, cols);
    XS_CHECK(hipGetLastError());
    return 0;
}
/* after the maps were added over ranks: write the NaN sentinel where the first event has no vertex */
extern "C" int xs_raycast_compose_finish(const int *min_keys_dev, float *vmap, float *nmap, size_t map_step, int rows, int cols,
                                         unsigned long long *hits_dev, void *stream) {
    if (!min_keys_dev || !vmap || !nmap) return xs_set_error(hipErrorInvalidValue, "xs_raycast_compose_finish: null pointer");
    const int ntiles = div_up(cols, 64) * div_up(rows, 4);
    dim3 block(64, 4), grid(ntiles < 256 ? ntiles : 256);
    hipLaunchKernelGGL(k_compose_finish, grid, block, 0, (hipStream_t)stream, min_keys_dev, (cfloat *)vmap, (cfloat *)nmap, map_step, rows, cols, hits_dev);
    XS_CHECK(hipGetLastError());
    return 0;
}
