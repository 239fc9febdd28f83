// xs_icp.hip — ICP point-to-plane normal equations for gfx950.  Replaces
// XKinectFusion/src/ICP.cu:166-281 (Combined: search_newton + operator()), :5-118 (LDS tree
// reductions), :120-164 (TranformReduction second kernel) and :365-429 (estimateCombined).
//
// Reference shape: one pixel per thread, 27 complex products each pushed through an LDS tree
// (27 x 2 barriers + 8 steps), 518 KB of per-block partials, and a second 27-block launch to
// add them up; the tree's tail relies on 32-wide warp-synchronous execution (ICP.cu:40-65).
// Here: one 64-pixel tile per wave, eight waves per workgroup (every level of a 640 x 480 frame; larger images fall back to four
// waves striding over the tiles with the sums in registers).  The 27 complex products of a pixel (complex float, ICP.cu:273) go
// to an LDS tile as floats and are added there in double — lanes in a fixed interleaved order, then the waves in order — as the
// reference accumulates in double (ICP.cu:274); the workgroup writes one 448-byte record with write-through stores and takes a
// ticket; the last workgroup to arrive adds the records in a fixed order, so the result is deterministic and there is no
// second launch.  The 55 sums go to device memory, or straight into host-coherent pinned memory with a completion word behind them
// (done_flag), or — the orchestrator's form since round 6 — as 55 stores of {launch number, sum} that need no word behind them
// (XS_ICP_PUBLISH_PAIRS).  How each phase was measured: profiles/tools/trace_icp.sh, profiles/r02_icp_phases.txt.
#include <string.h>
#include "xs_device.h"
#include "xs_env.h"
#include "xs_mailbox.h"
#include "xs_icp_solve.h"
#include "../../include/xslam_amd.h"

using namespace xs;
// The record hand-offs below (relaxed agent-scope stores + s_waitcnt vmcnt(0) + a relaxed ticket, no release fence) are correct because
// gfx942 / gfx950 implement an agent-scope atomic store as a write-through (sc1) store that is acknowledged from memory; that is
// outside the HIP / LLVM memory model, so the file refuses to build for anything else rather than publish stale records there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "write-through record publish: gfx942 / gfx950 only (use a release fence + acq_rel ticket on other targets)"
#endif

struct IcpArgs {
    MatS33 Rcurr; cfloat3 tcurr;
    const cfloat *vmap_curr; const cfloat *nmap_curr;
    // optional: the same two maps' real parts as float planes (their imaginary parts are zeros when the depth image is real:
    // xs_create_vnmaps_real) — read instead of the complex maps, half the bytes
    const float *vreal; const float *nreal; size_t rstep;
    MatS33 Rprev_inv; cfloat3 tprev;
    Intr intr;
    const cfloat *vmap_g_prev; const cfloat *nmap_g_prev;
    size_t mstep;
    float distThres, angleThres;
    int cols, rows;
    int y0, y1;             // pixel rows this launch covers (row sharding across GPUs)
    double *partials;       // [gridDim.x][56]
    unsigned *ticket;       // zeroed before the launch
    double *out;            // 54 sums (27 x re,im) + [54] = inlier count
    unsigned long long *done_flag; unsigned long long done_seq;  // optional: host-visible completion word
    int pairs;              // out is host-coherent pinned memory taking 55 x {u64 done_seq, double sum}: every sum carries its own sequence word (XS_ICP_PUBLISH_PAIRS)
    // optional device-side pose update (xs_icp_iterate): the last workgroup solves for the increment and
    // composes it into *pose, which the next launch reads instead of Rcurr / tcurr above
    IcpPoseState *pose; IcpPoseState *pose_host; int load_pose;
    // optional posted pose (xs_icp_accumulate_posted): the launch was enqueued before its pose was known and
    // picks it up from a 128-byte mailbox in host-coherent pinned memory once the host has posted mailbox_seq
    const unsigned *mailbox; unsigned mailbox_seq;
    // optional host fold (xs_icp_accumulate_records): every workgroup stores its record straight into host-coherent
    // pinned memory, sequence number last, and leaves; the host adds the records (xs_icp_sum_records)
    double *host_records; unsigned long long record_seq;
};

namespace {
// ICP.cu:196-244
// the six current-frame values of a pixel: they depend on nothing but the pixel, so the one-tile-per-wave instance requests them
// before it even has its pose (a posted launch spends ~3 us waiting for it)
__device__ __forceinline__ void load_vertex(const IcpArgs &a, int x, int y, cfloat3 &vcurr) {
    if (a.vreal) {   // wave-uniform
        vcurr.x = cfloat(row_ptr(a.vreal, a.rstep, y)[x], 0.0f);
        vcurr.y = cfloat(row_ptr(a.vreal, a.rstep, y + a.rows)[x], 0.0f);
        vcurr.z = cfloat(row_ptr(a.vreal, a.rstep, y + 2 * a.rows)[x], 0.0f);
        return;
    }
    vcurr.x = row_ptr(a.vmap_curr, a.mstep, y)[x];
    vcurr.y = row_ptr(a.vmap_curr, a.mstep, y + a.rows)[x];
    vcurr.z = row_ptr(a.vmap_curr, a.mstep, y + 2 * a.rows)[x];
}
__device__ __forceinline__ void load_normal(const IcpArgs &a, int x, int y, cfloat3 &ncurr) {
    // all three are requested before the sentinel is looked at (the y / z planes of an invalid pixel are allocated, merely
    // unused): one memory round trip instead of two
    if (a.nreal) {
        ncurr.x = cfloat(row_ptr(a.nreal, a.rstep, y)[x], 0.0f);
        ncurr.y = cfloat(row_ptr(a.nreal, a.rstep, y + a.rows)[x], 0.0f);
        ncurr.z = cfloat(row_ptr(a.nreal, a.rstep, y + 2 * a.rows)[x], 0.0f);
        return;
    }
    ncurr.x = row_ptr(a.nmap_curr, a.mstep, y)[x];
    ncurr.y = row_ptr(a.nmap_curr, a.mstep, y + a.rows)[x];
    ncurr.z = row_ptr(a.nmap_curr, a.mstep, y + 2 * a.rows)[x];
}
__device__ __forceinline__ void load_current(const IcpArgs &a, int x, int y, cfloat3 &ncurr, cfloat3 &vcurr) {
    load_normal(a, x, y, ncurr);
    load_vertex(a, x, y, vcurr);
}
// The two gates of ICP.cu:232-241 compare the REAL part of a complex square root with a threshold and use nothing else of
// it.  For z = a + ib, Re sqrt z = sqrt((|z| + a) / 2) lies in [sqrt(max(a, 0)), sqrt(|a| + |b|)], and the float sequence
// the reference runs (hypot, sqrt, atan2, cos, one product — or the short form of xs_complex.h) lands within a few ulp of it:
// with a margin three hundred times that (1e-5 relative on the squares) the comparison is settled from a and b alone unless
// the value sits on the threshold, and only then does the lane run the square root itself (~330 instructions when the
// argument is outside the short form's cone, which is where matched pixels live: |d| of a millimetre, sine ~ 0).  The same
// booleans as the full evaluation for every input (xs_icp_gate_selftest sweeps the neighbourhood of the threshold).
__device__ __forceinline__ bool re_sqrt_exceeds(cfloat z, float thres, bool or_equal) {
    const float lo2 = fmaxf(z.re, 0.0f), hi2 = fabsf(z.re) + fabsf(z.im), t2 = thres * thres;
    constexpr float M = 1e-5f;
    if (thres > 0x1p-60f && thres < 0x1p60f) {   // NaN operands fail both tests and take the full path
        if (hi2 * (1.0f + M) < t2) return false;
        if (lo2 > t2 * (1.0f + M)) return true;
    }
    const float r = sqrt(z).re;
    return or_equal ? r >= thres : r > thres;
}
__device__ __forceinline__ bool search_loaded(const IcpArgs &a, const MatS33 &Rcurr, const cfloat3 &tcurr, const cfloat3 &ncurr, const cfloat3 &vcurr,
                                              cfloat3 &n, cfloat3 &d, cfloat3 &s) {
    if (isnan(ncurr.x.re)) return false;
    const cfloat3 vcurr_g = Rcurr * vcurr + tcurr;
    const cfloat3 vcp = a.Rprev_inv * (vcurr_g - a.tprev);
    const float cpx = vcp.x.re, cpy = vcp.y.re, cpz = vcp.z.re;
    const int ux = __float2int_rn(cpx * a.intr.fx / cpz + a.intr.cx);
    const int uy = __float2int_rn(cpy * a.intr.fy / cpz + a.intr.cy);
    if (ux < 0 || uy < 0 || ux >= a.cols || uy >= a.rows || cpz < 0) return false;
    cfloat3 nprev_g, vprev_g;  // likewise: the six model-map values of the matched pixel together
    nprev_g.x = row_ptr(a.nmap_g_prev, a.mstep, uy)[ux];
    nprev_g.y = row_ptr(a.nmap_g_prev, a.mstep, uy + a.rows)[ux];
    nprev_g.z = row_ptr(a.nmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
    vprev_g.x = row_ptr(a.vmap_g_prev, a.mstep, uy)[ux];
    vprev_g.y = row_ptr(a.vmap_g_prev, a.mstep, uy + a.rows)[ux];
    vprev_g.z = row_ptr(a.vmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
    if (isnan(nprev_g.x.re)) return false;
    if (re_sqrt_exceeds(squarednorm(vprev_g - vcurr_g), a.distThres, false)) return false;   // norm(...).re > distThres
    const cfloat3 ncurr_g = Rcurr * ncurr;
    if (re_sqrt_exceeds(squarednorm(cross(ncurr_g, nprev_g)), a.angleThres, true)) return false;   // norm(...).re >= angleThres
    n = nprev_g; d = vprev_g; s = vcurr_g;
    return true;
}
// The same search for a wave that already holds its pixel's vertex (requested before the pose was known): the three normal
// values are requested here, and their sentinel is looked at only after the model-map gather — they are not needed before,
// and asking for them up front with the vertices doubled the burst of requests (14.7 MB at level 0) that the workgroups'
// mailbox polls then queued behind.  Same outcome for every pixel: the gate only moves past statements without side effects
// (a NaN vertex converts to pixel (0, 0), which is in range).
__device__ __forceinline__ bool search_vertex_loaded(const IcpArgs &a, const MatS33 &Rcurr, const cfloat3 &tcurr, int x, int y, const cfloat3 &vcurr,
                                                     cfloat3 &n, cfloat3 &d, cfloat3 &s) {
    cfloat3 ncurr;
    load_normal(a, x, y, ncurr);
    const cfloat3 vcurr_g = Rcurr * vcurr + tcurr;
    const cfloat3 vcp = a.Rprev_inv * (vcurr_g - a.tprev);
    const float cpx = vcp.x.re, cpy = vcp.y.re, cpz = vcp.z.re;
    // (an invalid pixel's vertex is NaN and its normal's sentinel has not been looked at yet: the conversion of a NaN is
    // kept out of the picture — such a pixel reads model pixel (0, 0) and is rejected below)
    const float fu = cpx * a.intr.fx / cpz + a.intr.cx, fv = cpy * a.intr.fy / cpz + a.intr.cy;
    const int ux = fu == fu ? __float2int_rn(fu) : 0;
    const int uy = fv == fv ? __float2int_rn(fv) : 0;
    if (ux < 0 || uy < 0 || ux >= a.cols || uy >= a.rows || cpz < 0) return false;
    cfloat3 nprev_g, vprev_g;
    nprev_g.x = row_ptr(a.nmap_g_prev, a.mstep, uy)[ux];
    nprev_g.y = row_ptr(a.nmap_g_prev, a.mstep, uy + a.rows)[ux];
    nprev_g.z = row_ptr(a.nmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
    vprev_g.x = row_ptr(a.vmap_g_prev, a.mstep, uy)[ux];
    vprev_g.y = row_ptr(a.vmap_g_prev, a.mstep, uy + a.rows)[ux];
    vprev_g.z = row_ptr(a.vmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
    if (isnan(ncurr.x.re)) return false;      // ICP.cu:202-204
    if (isnan(nprev_g.x.re)) return false;
    if (re_sqrt_exceeds(squarednorm(vprev_g - vcurr_g), a.distThres, false)) return false;   // norm(...).re > distThres
    const cfloat3 ncurr_g = Rcurr * ncurr;
    if (re_sqrt_exceeds(squarednorm(cross(ncurr_g, nprev_g)), a.angleThres, true)) return false;   // norm(...).re >= angleThres
    n = nprev_g; d = vprev_g; s = vcurr_g;
    return true;
}
// search_vertex_loaded in two stages, for a wave that works through two tiles (the balanced level-0 instance): stage 1 asks for the
// pixel's normal, projects it and asks for the six model-map values — of pixel (0, 0) where the projection leaves the image: resident,
// never used — without a branch, so that the requests of the second tile go out before the first tile's answers are looked at (two tiles
// one after the other were two dependent chains of memory round trips: the pixel phase of those waves took 6.2 us against 4.4); stage 2
// applies the rejection tests in the order of ICP.cu:202-241.  Same outcome for every pixel as search_vertex_loaded.
struct SearchStage { cfloat3 ncurr, vcurr_g, nprev_g, vprev_g; bool inb; };
__device__ __forceinline__ void search_issue(const IcpArgs &a, const MatS33 &Rcurr, const cfloat3 &tcurr, int x, int y, const cfloat3 &vcurr, SearchStage &st) {
    load_normal(a, x, y, st.ncurr);
    st.vcurr_g = Rcurr * vcurr + tcurr;
    const cfloat3 vcp = a.Rprev_inv * (st.vcurr_g - a.tprev);
    const float cpx = vcp.x.re, cpy = vcp.y.re, cpz = vcp.z.re;
    const float fu = cpx * a.intr.fx / cpz + a.intr.cx, fv = cpy * a.intr.fy / cpz + a.intr.cy;
    int ux = fu == fu ? __float2int_rn(fu) : 0;
    int uy = fv == fv ? __float2int_rn(fv) : 0;
    st.inb = !(ux < 0 || uy < 0 || ux >= a.cols || uy >= a.rows || cpz < 0);
    ux = st.inb ? ux : 0; uy = st.inb ? uy : 0;
    st.nprev_g.x = row_ptr(a.nmap_g_prev, a.mstep, uy)[ux];
    st.nprev_g.y = row_ptr(a.nmap_g_prev, a.mstep, uy + a.rows)[ux];
    st.nprev_g.z = row_ptr(a.nmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
    st.vprev_g.x = row_ptr(a.vmap_g_prev, a.mstep, uy)[ux];
    st.vprev_g.y = row_ptr(a.vmap_g_prev, a.mstep, uy + a.rows)[ux];
    st.vprev_g.z = row_ptr(a.vmap_g_prev, a.mstep, uy + 2 * a.rows)[ux];
}
__device__ __forceinline__ bool search_finish(const IcpArgs &a, const MatS33 &Rcurr, const SearchStage &st, cfloat3 &n, cfloat3 &d, cfloat3 &s) {
    if (!st.inb) return false;
    if (isnan(st.ncurr.x.re)) return false;      // ICP.cu:202-204
    if (isnan(st.nprev_g.x.re)) return false;
    if (re_sqrt_exceeds(squarednorm(st.vprev_g - st.vcurr_g), a.distThres, false)) return false;   // norm(...).re > distThres
    const cfloat3 ncurr_g = Rcurr * st.ncurr;
    if (re_sqrt_exceeds(squarednorm(cross(ncurr_g, st.nprev_g)), a.angleThres, true)) return false;   // norm(...).re >= angleThres
    n = st.nprev_g; d = st.vprev_g; s = st.vcurr_g;
    return true;
}
__device__ __forceinline__ bool search(const IcpArgs &a, const MatS33 &Rcurr, const cfloat3 &tcurr, int x, int y, cfloat3 &n, cfloat3 &d,
                                       cfloat3 &s) {
    cfloat3 ncurr, vcurr;
    load_current(a, x, y, ncurr, vcurr);
    return search_loaded(a, Rcurr, tcurr, ncurr, vcurr, n, d, s);
}
constexpr int NS = 54;      // 27 complex sums
constexpr int NP = 56;      // partial record: 54 sums + count + pad
}  // namespace

// host-visible completion word: push everything out, then publish the sequence number
__device__ __forceinline__ void publish_done(const IcpArgs &a) {
    // (a system-scope release store is the write-back of what precedes it + the store; a __threadfence_system() in front of it
    // wrote back a second time and invalidated for nothing)
    __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Phase stamps of every workgroup (experiment build -DXS_ICP_TRACE, profiles/tools/trace_icp.sh): 100 MHz wall clock at entry, pose ready,
// pixels done, fold done, record stored, released, ticket back, (last workgroup:) acquired, gathered, result out.
#ifdef XS_ICP_TRACE
__device__ unsigned long long g_icp_trace[768 * 16];
#define XS_STAMP(i) do { if (threadIdx.x == 0) { g_icp_trace[blockIdx.x * 16 + (i)] = wall_clock64(); \
    if ((i) == 0) g_icp_trace[blockIdx.x * 16 + 10] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | __builtin_amdgcn_s_getreg(0xF804); } } while (0)
extern "C" int xs_debug_icp_trace(unsigned long long *out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_icp_trace), sizeof(g_icp_trace));
}
#else
#define XS_STAMP(i) do { } while (0)
#endif

// the value held by the neighbouring lane (lane ^ 1): two quad-permute moves, no LDS
__device__ __forceinline__ double pair_swap(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)u, 0xB1, 0xF, 0xF, true);          // quad_perm [1, 0, 3, 2]
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// the per-wave partial sums of a workgroup, added in wave order
template <int WAVES>
__device__ __forceinline__ double wave_order_sum(const double (*smem)[NP], int k) {
    double t = smem[0][k];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) t += smem[w][k];
    return t;
}

// Pose mailbox (xs_icp_post_pose writes it, k_icp<POSE_POSTED> polls it; xs_icp_mailbox_alloc puts it in
// device memory the CPU reaches through the large BAR, so polling stays off the PCIe link — 512
// workgroups polling pinned host memory cost 50 us per iteration): four 32-byte sectors, each starting
// with the sequence number, carrying cmd (0 = run, 1 = abandon the launch) and the 18 floats of Rcurr followed by the 6 of
// tcurr — layout and reasons in xs_mailbox.h.  The load that finds the launch's number in all four sectors holds the payload.
enum { POSE_ARGS = 0, POSE_DEVICE = 1, POSE_POSTED = 2 };
constexpr unsigned long long kIcpTimeoutBit = 1ull << 63;

// POSE_ARGS: Rcurr / tcurr are kernel arguments (xs_icp_accumulate, first iteration of xs_icp_iterate).
// POSE_DEVICE: they are what k_icp_solve left in device memory after the previous iteration.
// POSE_POSTED: the host posts them while this launch is already resident — the launch latency of an
// iteration (~4 us) overlaps the previous iteration's epilogue and the host's solve.
#ifdef XS_ICP_WAVES_PER_EU   // experiment switch (profiles/tools/ab_icp_occupancy.sh): force the occupancy of the reduction kernel
#define XS_ICP_OCC __attribute__((amdgpu_waves_per_eu(XS_ICP_WAVES_PER_EU, XS_ICP_WAVES_PER_EU)))
#else
#define XS_ICP_OCC
#endif
// WAVES = 4: lanes stride over the tiles and carry the 27 complex sums in registers (54 doubles, 190 VGPRs: two waves per SIMD) —
// any image size.  WAVES = 8: one tile per wave, chosen whenever all tiles can be resident at once (a 640 x 480 level 0 has
// 4 800): nothing is carried from tile to tile, so the products go straight from the row to the LDS fold as floats and the kernel
// needs 80-104 registers.  PASSES: the LDS tile holds all 55 values of the eight waves at once (123 KB: one workgroup per CU —
// launches of up to 256 workgroups: levels 1 and 2), 28 in two passes (64 KB, two per CU at four waves per SIMD, up to 512) or 19
// in three (45 KB, three per CU at six waves per SIMD, 80 VGPRs: up to 768 — level 0's 600).
// BAL (round 3; PASSES = 2): a launch whose tiles outnumber the resident waves by up to a quarter — level 0 of a 640 x 480 frame: 4 800 tiles
// for 4 096 — deals them out evenly instead of launching a third workgroup on some CUs: workgroup b takes ntiles / grid tiles, the first
// ntiles % grid one more, and its first WAVES / 4 waves work through a second tile (in lock step: search_issue / search_finish).  With 600
// one-tile-per-wave workgroups 88 of the 256 CUs held three of them (24 tiles on the busiest against 18.75 on average) and the launch lasted
// as long as those.  WAVES = 8: 512 workgroups (two per CU) of nine or ten tiles — 22.3 -> 20.4 us per launch inside the loop; WAVES = 16
// (the default): 256 workgroups (one per CU, 158 KB of LDS) of 18 or 19 — half the records for the last workgroup to add and half the
// tickets: 18.9 us (profiles/r03_ab_icp_balanced_fps.txt).  The partition is a function of the launch geometry alone, so the sums keep one
// fixed association (slot order = tile order inside a workgroup).
template <int POSE_SRC, int WAVES, int PASSES, bool BAL = false>
__global__ void __launch_bounds__(64 * WAVES)
    __attribute__((amdgpu_waves_per_eu(WAVES == 4 || PASSES == 1 ? 2 : (PASSES == 2 ? 4 : 6), WAVES == 4 || PASSES == 1 ? 2 : (PASSES == 2 ? 4 : 6))))
    k_icp(const IcpArgs a) {
    static_assert(!BAL || ((WAVES == 8 || WAVES == 16) && PASSES == 2), "the balanced instances are the eight- and sixteen-wave, two-pass ones");
    constexpr int EXTRA = BAL ? WAVES / 4 : 0;   // waves that work through a second tile
    constexpr int SLOTS = WAVES + EXTRA;         // tiles a workgroup can take
    XS_STAMP(0);
#ifndef XS_ICP_NO_PRIO
    // an ICP launch is a chain of dependent steps on few waves; the announced next frame's bilateral filter (HintNextFrame) shares its
    // SIMDs with them and is throughput work: these waves go first at the issue arbiter
    __builtin_amdgcn_s_setprio(3);
#endif
    // one tile per wave: the tile's current-frame vertices are on their way before the pose is (a pixel outside the image reads
    // pixel (0, y0): resident, never used)
    cfloat3 pre_v, pre_v2;
    bool pre_ok = false, pre_ok2 = false;
    int pre_x = 0, pre_y = a.y0, pre_x2 = 0, pre_y2 = a.y0;
    if constexpr (WAVES >= 8) {
        const int tiles_x0 = (a.cols + 63) / 64;
        const int nt0 = tiles_x0 * (a.y1 - a.y0);
        int t0 = blockIdx.x * WAVES + (threadIdx.x >> 6), t1 = nt0;
        if constexpr (BAL) {
            const int base = nt0 / (int)gridDim.x, rem = nt0 % (int)gridDim.x;
            const int start = (int)blockIdx.x * base + min((int)blockIdx.x, rem), cnt = base + ((int)blockIdx.x < rem ? 1 : 0);
            const int w = threadIdx.x >> 6;
            t0 = w < cnt ? start + w : nt0;
            t1 = WAVES + w < cnt ? start + WAVES + w : nt0;
        }
        const int y = a.y0 + t0 / tiles_x0, x = (t0 % tiles_x0) * 64 + (threadIdx.x & 63);
        pre_ok = t0 < nt0 && x < a.cols;
        if (pre_ok) { pre_x = x; pre_y = y; }
#ifndef XS_ICP_NO_PREFETCH
        load_vertex(a, pre_x, pre_y, pre_v);
#endif
        if constexpr (BAL) {
            if (threadIdx.x < 64 * EXTRA) {   // the first waves: a second tile
                const int y2 = a.y0 + t1 / tiles_x0, x2 = (t1 % tiles_x0) * 64 + (threadIdx.x & 63);
                pre_ok2 = t1 < nt0 && x2 < a.cols;
                if (pre_ok2) { pre_x2 = x2; pre_y2 = y2; }
                load_vertex(a, pre_x2, pre_y2, pre_v2);
            }
        }
    }
    MatS33 Rcurr = a.Rcurr;
    cfloat3 tcurr = a.tcurr;
    // (readfirstlane: the 24 floats are wave-uniform and belong in scalar registers, like the kernel
    // arguments they replace)
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    if (POSE_SRC == POSE_DEVICE) {
        // the pose left by the previous launch on this stream; a failed solve ends the loop
        if (a.pose->status != 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0 && a.done_flag) publish_done(a);
            return;
        }
        const float *pr = a.pose->R, *pt = a.pose->t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Rcurr.data[r].x = cfloat(uni(pr[6 * r + 0]), uni(pr[6 * r + 1]));
            Rcurr.data[r].y = cfloat(uni(pr[6 * r + 2]), uni(pr[6 * r + 3]));
            Rcurr.data[r].z = cfloat(uni(pr[6 * r + 4]), uni(pr[6 * r + 5]));
        }
        tcurr.x = cfloat(uni(pt[0]), uni(pt[1])); tcurr.y = cfloat(uni(pt[2]), uni(pt[3])); tcurr.z = cfloat(uni(pt[4]), uni(pt[5]));
    }
    if (POSE_SRC == POSE_POSTED) {
        // One wave per workgroup polls the mailbox (system-scope loads: never cached) until
        // both lines carry this launch's sequence number, then hands the 32 words to the others through
        // LDS.  Bounded: after MAILBOX_MAX_POLLS the launch gives up and says so in the completion word.
        __shared__ unsigned s_mail[MAILBOX_WORDS];
        if (threadIdx.x < 64) mailbox_wait(a.mailbox, a.mailbox_seq, s_mail, (int)threadIdx.x);   // (xs_mailbox.h)
        __syncthreads();
        const unsigned cmd = s_mail[1];
        if (cmd != 0) {
            if (cmd == 2 && threadIdx.x == 0 && a.done_flag)
                __hip_atomic_store(a.done_flag, a.done_seq | kIcpTimeoutBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (cmd == 2 && threadIdx.x == 0 && a.pairs)   // (the first pair's sequence word carries the give-up)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.out), a.done_seq | kIcpTimeoutBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (cmd == 2 && threadIdx.x == 0 && a.host_records)   // host fold: the give-up shows in the record's sequence word
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_records) + (size_t)blockIdx.x * NP + (NP - 1),
                                   a.record_seq | kIcpTimeoutBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        auto word = [&](int i) { return uni(__uint_as_float(s_mail[mailbox_word_of(i)])); };
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Rcurr.data[r].x = cfloat(word(6 * r + 0), word(6 * r + 1));
            Rcurr.data[r].y = cfloat(word(6 * r + 2), word(6 * r + 3));
            Rcurr.data[r].z = cfloat(word(6 * r + 4), word(6 * r + 5));
        }
        tcurr.x = cfloat(word(18), word(19)); tcurr.y = cfloat(word(20), word(21)); tcurr.z = cfloat(word(22), word(23));
    }
    XS_STAMP(1);
    // 64 consecutive columns per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (a.cols + 63) / 64;
    const int ntiles = tiles_x * (a.y1 - a.y0);
    __shared__ double smem[SLOTS][NP];
    // one LDS buffer for the fold's tile and, afterwards, the last workgroup's row-group sums
    constexpr int PV = (NS + 1 + PASSES - 1) / PASSES;   // eight-wave instance: 55, 28 or 19 values per pass
    constexpr int RS = 68;                               // ... in rows 68 floats apart
    constexpr int TILE_BYTES = WAVES == 4 ? 4 * 28 * 65 * 8 : SLOTS * PV * RS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_tile[TILE_BYTES];
    if constexpr (WAVES == 4) {
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    double cnt = 0.0;
    // workgroups stride over (row, column-tile) pairs
    for (int t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
        const int y = a.y0 + t / tiles_x;
        const int x = (t % tiles_x) * 64 + lane;
        if (x >= a.cols) continue;
        cfloat3 n, d, s;
        if (!search(a, Rcurr, tcurr, x, y, n, d, s)) continue;
        cfloat row[7];
        const cfloat3 cr = cross(s, n);  // ICP.cu:257-259
        row[0] = cr.x; row[1] = cr.y; row[2] = cr.z;
        row[3] = n.x; row[4] = n.y; row[5] = n.z;
        row[6] = dot(n, d - s);
        int shift = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 7; ++j) {
                const cfloat p = row[i] * row[j];
                acc[2 * shift] += (double)p.re;
                acc[2 * shift + 1] += (double)p.im;
                ++shift;
            }
        cnt += 1.0;
    }
    // Fold the 256 lanes of the workgroup: every lane parks its 55 values (54 sums + count) in an
    // LDS tile [wave][value][lane] (rows padded to 65 doubles: conflict-free both ways), then four
    // threads per value each add one wave's 64 entries in lane order and the four results are
    // added in wave order — fixed association, no cross-lane shuffles (55 dependent 6-step
    // ds_bpermute chains cost ~20 us here).  Two passes of 28 values keep the tile at 58 KB.
    double (*tile)[28][65] = reinterpret_cast<double (*)[28][65]>(lds_tile);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 28; ++kk) {
            const int k = half * 28 + kk;
            if (k < NS) tile[wave][kk][lane] = acc[k];
            else if (k == NS) tile[wave][kk][lane] = cnt;
        }
        __syncthreads();
        const int kk = threadIdx.x >> 2, q = threadIdx.x & 3;
        if (kk < 28 && half * 28 + kk <= NS) {
            double sacc = 0.0;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) sacc += tile[q][kk][i];
            smem[q][half * 28 + kk] = sacc;
        }
    }
    } else {
    // one tile per wave: the row's 27 complex products (complex<f32>, ICP.cu:273) go to the LDS tile as they are formed —
    // floats, one row per (wave, value), in one, two or three passes — and are added in double, lane order then wave order, as above
    cfloat row[7];
    float one = 0.0f;
    cfloat row2[BAL ? 7 : 1];
    float one2 = 0.0f;
    auto fill_row = [](cfloat (&r)[7], const cfloat3 &n, const cfloat3 &d, const cfloat3 &s) {
        const cfloat3 cr = cross(s, n);  // ICP.cu:257-259
        r[0] = cr.x; r[1] = cr.y; r[2] = cr.z;
        r[3] = n.x; r[4] = n.y; r[5] = n.z;
        r[6] = dot(n, d - s);
    };
    if constexpr (BAL) {
        // both tiles of waves 0 and 1 in lock step (search_issue / search_finish above); the other waves' single tile the same way
        SearchStage sa, sb;
        search_issue(a, Rcurr, tcurr, pre_x, pre_y, pre_v, sa);
        const bool two = wave < EXTRA;   // (wave-uniform)
        if (two) search_issue(a, Rcurr, tcurr, pre_x2, pre_y2, pre_v2, sb);
        cfloat3 n, d, s;
#pragma unroll
        for (int i = 0; i < 7; ++i) { row[i] = cfloat(0.0f, 0.0f); row2[i] = cfloat(0.0f, 0.0f); }   // ICP.cu:262: a rejected pixel contributes zeros
        if (pre_ok && search_finish(a, Rcurr, sa, n, d, s)) { fill_row(row, n, d, s); one = 1.0f; }
        if (two && pre_ok2 && search_finish(a, Rcurr, sb, n, d, s)) { fill_row(row2, n, d, s); one2 = 1.0f; }   // a workgroup with nine tiles leaves wave 1's second slot all zeros
    } else {
        bool ok = false;
        cfloat3 n, d, s;
#ifdef XS_ICP_NO_PREFETCH
        load_vertex(a, pre_x, pre_y, pre_v);
#endif
        if (pre_ok) ok = search_vertex_loaded(a, Rcurr, tcurr, pre_x, pre_y, pre_v, n, d, s);
        if (ok) {
            fill_row(row, n, d, s);
            one = 1.0f;
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) row[i] = cfloat(0.0f, 0.0f);   // ICP.cu:262: a rejected pixel contributes zeros
        }
    }
    XS_STAMP(2);
    // Rows of 64 floats, one per (wave, value), 68 floats apart: 16-byte aligned with an odd number of 16-byte chunks between
    // rows, so the wave's row-wise stores and the adders' 128-bit loads (eight consecutive lanes = eight consecutive rows, or
    // four rows x two halves) are both free of bank conflicts.  A row is added by two lanes, 32 entries each, when the
    // workgroup has the threads for it (two or three passes), by one otherwise.
    constexpr int SPLIT = 16 * PV <= 512 ? 2 : 1;   // (the balanced instance's 280 rows: a few lanes take a second one)
    float *tile = reinterpret_cast<float *>(lds_tile);
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass) __syncthreads();
        int shift = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 7; ++j) {
                const int kre = 2 * shift, kim = 2 * shift + 1;
                if (kre / PV == pass || kim / PV == pass) {
                    const cfloat p = row[i] * row[j];
                    if (kre / PV == pass) tile[(wave * PV + kre % PV) * RS + lane] = p.re;
                    if (kim / PV == pass) tile[(wave * PV + kim % PV) * RS + lane] = p.im;
                    if constexpr (BAL) {
                        if (wave < EXTRA) {
                            const cfloat p2 = row2[i] * row2[j];
                            if (kre / PV == pass) tile[((WAVES + wave) * PV + kre % PV) * RS + lane] = p2.re;
                            if (kim / PV == pass) tile[((WAVES + wave) * PV + kim % PV) * RS + lane] = p2.im;
                        }
                    }
                }
                ++shift;
            }
        if (pass == NS / PV) {
            tile[(wave * PV + NS % PV) * RS + lane] = one;
            if constexpr (BAL) { if (wave < EXTRA) tile[((WAVES + wave) * PV + NS % PV) * RS + lane] = one2; }
        }
        XS_STAMP(11 + 2 * (pass & 1));   // (trace build: this wave's products of the pass are on their way to LDS)
        __syncthreads();
        const int h = threadIdx.x % SPLIT;
#pragma unroll
        for (int r = threadIdx.x / SPLIT; r < SLOTS * PV; r += (64 * WAVES) / SPLIT) {
            // four interleaved partial sums (entries i, i + 4, ... each) per lane; with two lanes per row the halves are added
            // partial by partial, then ((s0 + s1) + (s2 + s3)): the same fixed association every launch.  The second lane
            // walks its eight chunks starting from the fifth, which keeps the pair on different banks.
            const float4 *p = reinterpret_cast<const float4 *>(tile + r * RS);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            if constexpr (SPLIT == 1) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float4 f = p[c];
                    s0 += (double)f.x; s1 += (double)f.y; s2 += (double)f.z; s3 += (double)f.w;
                }
            } else {
                const float4 *pa = p + 12 * h, *pb = p + 4 + 4 * h;   // lane 0: chunks 0-3 then 4-7; lane 1: 12-15 then 8-11
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 f = pa[c];
                    s0 += (double)f.x; s1 += (double)f.y; s2 += (double)f.z; s3 += (double)f.w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 f = pb[c];
                    s0 += (double)f.x; s1 += (double)f.y; s2 += (double)f.z; s3 += (double)f.w;
                }
                s0 += pair_swap(s0); s1 += pair_swap(s1); s2 += pair_swap(s2); s3 += pair_swap(s3);
            }
            const int w = r / PV, v = r % PV;
            if (h == 0 && pass * PV + v <= NS) smem[w][pass * PV + v] = (s0 + s1) + (s2 + s3);
        }
        XS_STAMP(12 + 2 * (pass & 1));   // (trace build: this wave's rows of the pass are added)
    }
    }
    __syncthreads();
    XS_STAMP(3);
    if (a.host_records) {
        // Host fold: the record goes straight to host-coherent pinned memory (28 16-byte stores of wave 0), then — once
        // those stores are acknowledged and released at system scope — its last word receives the launch's sequence
        // number, and the workgroup is done: no ticket, no write-back of this XCD's L2 for another workgroup to read,
        // no last workgroup gathering 512 records from memory.  The host spins on the sequence words and adds the
        // records in index order (xs_icp_sum_records): the same deterministic association for every launch.
        if (threadIdx.x < 64) {
            struct alignas(16) d2 { double x, y; };
            if (threadIdx.x < NP / 2) {
                const int k0 = 2 * threadIdx.x, k1 = k0 + 1;
                d2 v;
                v.x = wave_order_sum<SLOTS>(smem, k0);
                v.y = k1 <= NS ? wave_order_sum<SLOTS>(smem, k1) : 0.0;
                if (k1 <= NS) reinterpret_cast<d2 *>(a.host_records)[(size_t)blockIdx.x * (NP / 2) + threadIdx.x] = v;
                else a.host_records[(size_t)blockIdx.x * NP + k0] = v.x;      // the count; the pad word is the sequence slot
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.host_records) + (size_t)blockIdx.x * NP + (NP - 1), a.record_seq,
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (threadIdx.x < NP / 2) {
        // The record: 28 lanes of wave 0 store two doubles each (the pad entry is written as zero) with agent-scope
        // write-through stores — they go through this XCD's L2 to memory on their own, so nothing is left for a write-back of
        // the whole L2 (what an agent-scope release fence does here: 0.6 us when one workgroup does it, up to 3 us when
        // six hundred do).
        const int k0 = 2 * threadIdx.x, k1 = k0 + 1;
        double *rec = a.partials + (size_t)blockIdx.x * NP;
        __hip_atomic_store(rec + k0, wave_order_sum<SLOTS>(smem, k0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + k1, k1 <= NS ? wave_order_sum<SLOTS>(smem, k1) : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // publish: the storing wave waits until its write-through stores are acknowledged, then its first lane takes a ticket
    // (same wave, program order: no barrier in between); the atomic is performed at agent scope after the records are in memory
    __shared__ unsigned s_last;
    if (threadIdx.x < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XS_STAMP(4);
    }
    if (threadIdx.x == 0) {
        XS_STAMP(5);
        const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == gridDim.x - 1) ? 1u : 0u;
        XS_STAMP(6);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // the ticket re-arms itself: the next launch on this stream starts from zero
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // holds the barrier until the invalidate has completed
            XS_STAMP(7);
        }
    }
    __syncthreads();
    if (s_last) {
        // after the acquire (this CU's L1 invalidated) + barrier, plain loads see every record.  The records
        // come from memory (other XCDs wrote them), so the sum is latency-bound: a record is 28 16-byte
        // chunks; thread (g, q) adds chunk q of records g, g+9, g+18, ... in that order with 32 loads in
        // flight, and the row groups are then added in group order — a fixed association, so
        // deterministic — leaving two or three memory round trips where a plain loop had sixteen.  (Level 0's 600 records
        // are 269 KB through one CU's 64 B / clk: ~1.8 us of the 3.4 us this takes is that, whatever the depth.)
        struct alignas(16) d2 { double x, y; };
        // (eight waves: twice the threads, so eighteen row groups with sixteen loads each in flight — the same bytes in
        // flight per workgroup at half the registers per lane, which is what lets this instance run at four waves per SIMD)
        constexpr int G = WAVES == 16 ? 36 : (WAVES == 8 ? 18 : 9), DEPTH = WAVES == 16 ? 8 : (WAVES == 8 ? 16 : 32);
        static_assert(TILE_BYTES >= G * 28 * 16, "row-group sums live in the fold's tile");
        d2 (*s_red)[28] = reinterpret_cast<d2 (*)[28]>(lds_tile);   // every wave is past the fold (barriers above)
        const int q = threadIdx.x % 28, g = threadIdx.x / 28;
        if (g < G) {
            const d2 *p = reinterpret_cast<const d2 *>(a.partials) + q;
            d2 acc2 = {0.0, 0.0};
            unsigned b = g;
            const unsigned nb = gridDim.x;
            for (; b + G * (DEPTH - 1) < nb; b += G * DEPTH) {
                d2 v[DEPTH];
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) v[k] = p[(size_t)(b + G * k) * (NP / 2)];
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) { acc2.x += v[k].x; acc2.y += v[k].y; }
            }
            {   // the tail, still issued together
                d2 v[DEPTH];
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) {
                    const unsigned bb = b + G * k;
                    v[k] = bb < nb ? p[(size_t)bb * (NP / 2)] : d2{0.0, 0.0};
                }
#pragma unroll
                for (int k = 0; k < DEPTH; ++k)
                    if (b + G * k < nb) { acc2.x += v[k].x; acc2.y += v[k].y; }
            }
            s_red[g][q] = acc2;
        }
        __syncthreads();
        XS_STAMP(8);
        if (threadIdx.x < 28) {
            d2 t = s_red[0][threadIdx.x];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) {
                t.x += s_red[gg][threadIdx.x].x; t.y += s_red[gg][threadIdx.x].y;
                // (both chains advance together: left alone the compiler sinks the y chain into the guarded store below, keeps all G
                // y values live across the x chain — 36 x 4 registers at sixteen waves — and spills them: five dependent scratch reloads,
                // 2-3.5 us at the very end of every level-0 launch, profiles/r03_icp_phases.txt)
#ifndef XS_ICP_TAIL_BASELINE
                asm volatile("" : "+v"(t.x), "+v"(t.y));
#endif
            }
            if (a.pairs) {
                // Every sum leaves as ONE 16-byte store {sequence number, sum}: a store of one lane cannot be seen in halves, so the host
                // needs no word that is ordered behind the others — and the kernel no wait for its stores' acknowledgement, barrier and
                // release store between the sums and that word (profiles/r06_ab_icp_publish_pairs.txt).
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
                ull2 *pairs = reinterpret_cast<ull2 *>(a.out);
                __builtin_nontemporal_store(ull2{a.done_seq, (unsigned long long)__double_as_longlong(t.x)}, pairs + 2 * threadIdx.x);
                if (2 * threadIdx.x + 1 < NS + 1)
                    __builtin_nontemporal_store(ull2{a.done_seq, (unsigned long long)__double_as_longlong(t.y)}, pairs + 2 * threadIdx.x + 1);
            } else {
                a.out[2 * threadIdx.x] = t.x;
                if (2 * threadIdx.x + 1 < NS + 1) a.out[2 * threadIdx.x + 1] = t.y;
            }
        }
        if (a.done_flag) {
            // out (and the flag) may live in host-coherent pinned memory: push the sums out, then
            // publish the sequence number the host is spinning on — no copy, no stream sync
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) publish_done(a);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        XS_STAMP(9);
    }
}

// ---- pose update on the device (xs_icp_iterate) ---------------------------------------------
// KinectFusionReconstruction.cpp:203-224 as a one-workgroup kernel behind the reduction: wave 1 takes
// the determinant gate while wave 0 factors and solves (one matrix row per lane); three lanes then
// evaluate the three angles' sin / cos side by side and form one row each of Rinc, Rinc*tcurr + tinc
// and Rinc*Rcurr.  It lives in its own kernel because the double-precision solve wants ~100 registers
// of its own: inside k_icp it cost the pixel loop its second wave per SIMD.
struct IcpSolveArgs {
    const double *sums;     // the 55 values the reduction just wrote
    double *sums_host;      // optional host-visible copy of them (per-iteration log)
    MatS33 Rcurr; cfloat3 tcurr; int load_pose;  // load_pose: start from these instead of *pose
    IcpPoseState *pose, *pose_host;
    unsigned long long *done_flag; unsigned long long done_seq;
};
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) k_icp_solve(const IcpSolveArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!a.load_pose && a.pose->status != 0) {  // the loop ended at an earlier iteration
        if (threadIdx.x == 0 && a.done_flag) {
            __threadfence_system();
            __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __shared__ double s_sums[NP];
    __shared__ double s_det;
    __shared__ cfloat s_res[6], s_sn[3], s_cs[3];
    if (threadIdx.x < NS + 1) {
        const double v = a.sums[threadIdx.x];
        s_sums[threadIdx.x] = v;
        if (a.sums_host) a.sums_host[threadIdx.x] = v;
    }
    __syncthreads();
    if (wave == 1) {
        const double det = icp_solve::real_determinant6_wave(s_sums, lane);
        if (lane == 0) s_det = det;
    } else {
        const cdouble sol = icp_solve::llt_solve6_wave(s_sums, lane);
        if (lane < 6) s_res[lane] = cfloat((float)sol.re, (float)sol.im);
        // (same wave: the three angles are in LDS once its writes have landed)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < 3) icp_solve::csincos(s_res[lane], s_sn[lane], s_cs[lane]);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int i = threadIdx.x;
        MatS33 Rcurr = a.Rcurr;
        cfloat3 tcurr = a.tcurr;
        int iters = 1;
        if (!a.load_pose) {
            const cfloat *pr = reinterpret_cast<const cfloat *>(a.pose->R), *pt = reinterpret_cast<const cfloat *>(a.pose->t);
#pragma unroll
            for (int r = 0; r < 3; ++r) { Rcurr.data[r].x = pr[3 * r]; Rcurr.data[r].y = pr[3 * r + 1]; Rcurr.data[r].z = pr[3 * r + 2]; }
            tcurr.x = pt[0]; tcurr.y = pt[1]; tcurr.z = pt[2];
            iters = a.pose->iters + 1;
        }
        const double det = s_det;
        const int status = (det != det) ? 2 : (fabs(det) < 1e-15 ? 1 : 0);
        icp_solve::Row3 Rn;
        cfloat tn;
        if (status == 0) icp_solve::compose_pose_rows(i, s_res, s_sn, s_cs, Rcurr, tcurr, Rn, tn);
        else {  // the pose stays where it was
            const cfloat3 r = i == 0 ? Rcurr.data[0] : (i == 1 ? Rcurr.data[1] : Rcurr.data[2]);
            Rn.v[0] = r.x; Rn.v[1] = r.y; Rn.v[2] = r.z;
            tn = i == 0 ? tcurr.x : (i == 1 ? tcurr.y : tcurr.z);
        }
        // every lane has read the old pose by now (same wave, program order), so it can be overwritten
        __builtin_amdgcn_wave_barrier();
        IcpPoseState *dst[2] = {a.pose, a.pose_host};
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            IcpPoseState *ps = dst[d];
            if (!ps) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) { ps->R[6 * i + 2 * j] = Rn.v[j].re; ps->R[6 * i + 2 * j + 1] = Rn.v[j].im; }
            ps->t[2 * i] = tn.re; ps->t[2 * i + 1] = tn.im;
            if (i == 0) { ps->status = status; ps->iters = iters; ps->det = det; }
        }
    }
    if (a.done_flag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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

enum { XS_ICP_MAX_BLOCKS = 768 };  // records a launch may write = eight-wave workgroups resident at once (three per CU: 51 KB of LDS, 79 VGPRs)

extern "C" size_t xs_icp_workspace_bytes(void) { return (size_t)XS_ICP_MAX_BLOCKS * NP * sizeof(double) + 256; }
/* zero the arrival ticket once after allocating the workspace (launches re-arm it themselves) */
extern "C" int xs_icp_workspace_init(void *workspace, void *stream) {
    if (!workspace) return xs_set_error(hipErrorInvalidValue, "xs_icp_workspace_init: null pointer");
    XS_CHECK(hipMemsetAsync(workspace, 0, 256, (hipStream_t)stream));
    return 0;
}

// workgroups (= records) of a launch over pixel rows [y0, y1), and the kernel shape: eight waves with one 64-pixel tile each
// while all of them are resident at once (two or three such workgroups per CU, see PASSES: up to 768, i.e. 6 144 tiles — every
// level of a 640 x 480 frame), else four waves striding over the tiles with the sums in registers (any size)
static int icp_blocks(int cols, int y0, int y1, int *waves = nullptr) {
    const int tiles = div_up(cols, 64) * (y1 - y0);
    // more tiles than the 4 096 waves a launch can keep resident at this kernel's register / LDS budget, by up to a quarter (level 0 of a
    // 640 x 480 frame: 4 800): one sixteen-wave workgroup per CU, 18 or 19 tiles each (k_icp<., 16, 2, true>; `waves` 17 marks it) —
    // 256 records for the last workgroup to add instead of 512 or 600.  XS_ICP_BALANCED (measurement aid): 1 = 512 eight-wave workgroups
    // of nine or ten tiles (`waves` 9), 0 = the one-tile-per-wave launch.
    static const int mode = product_env_int("XS_ICP_BALANCED", 2);   // (a documented switch: INTEGRATION.md "Environment"; tests/test_publish_stress_gpu.py runs both balanced instances)
    if (mode != 0 && tiles > 8 * 512 && tiles <= 10 * 512) {
        if (waves) *waves = mode == 1 ? 9 : 17;
        return mode == 1 ? 512 : 256;
    }
    int w = 8, blocks = div_up(tiles, 8);
    if (blocks > XS_ICP_MAX_BLOCKS) {
        w = 4;
        blocks = div_up(tiles, 4);
        if (blocks > 512) blocks = 512;
    }
    if (waves) *waves = w;
    return blocks < 1 ? 1 : blocks;
}
template <int POSE_SRC>
static void icp_dispatch(int waves, int blocks, hipStream_t st, const IcpArgs &a) {
    if (waves == 9) hipLaunchKernelGGL((k_icp<POSE_SRC, 8, 2, true>), dim3(blocks), dim3(512), 0, st, a);
    else if (waves == 17) hipLaunchKernelGGL((k_icp<POSE_SRC, 16, 2, true>), dim3(blocks), dim3(1024), 0, st, a);
    else if (waves == 8 && blocks <= 256) hipLaunchKernelGGL((k_icp<POSE_SRC, 8, 1>), dim3(blocks), dim3(512), 0, st, a);
    else if (waves == 8 && blocks <= 512) hipLaunchKernelGGL((k_icp<POSE_SRC, 8, 2>), dim3(blocks), dim3(512), 0, st, a);
    else if (waves == 8) hipLaunchKernelGGL((k_icp<POSE_SRC, 8, 3>), dim3(blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_icp<POSE_SRC, 4, 2>), dim3(blocks), dim3(256), 0, st, a);
}

// shared launcher of xs_icp_accumulate / xs_icp_iterate
static int icp_launch(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr, const float *Rprev_inv18,
                      const float *tprev6, const float *intr4, const float *vmap_g_prev, const float *nmap_g_prev, size_t map_step, int rows,
                      int cols, float distThres, float angleThres, int y0, int y1, void *workspace, double *sums_dev,
                      unsigned long long *done_flag, unsigned long long done_seq, IcpPoseState *pose, IcpPoseState *pose_host, double *sums_host,
                      void *stream, const char *who, const void *mailbox = nullptr, unsigned mailbox_seq = 0, double *host_records = nullptr,
                      unsigned long long record_seq = 0, const float *vreal = nullptr, const float *nreal = nullptr, size_t rstep = 0) {
    if ((!pose && !mailbox && (!Rcurr18 || !tcurr6)) || !vmap_curr || !nmap_curr || !Rprev_inv18 || !tprev6 || !intr4 || !vmap_g_prev || !nmap_g_prev ||
        (!host_records && (!workspace || !sums_dev)))
        return xs_set_error(hipErrorInvalidValue, who);
    if (y0 < 0 || y1 > rows || y1 < y0) return xs_set_error(hipErrorInvalidValue, "xs_icp: bad row range");
    IcpArgs a;
    if (Rcurr18) { ld_mat(Rcurr18, a.Rcurr); ld_vec(tcurr6, a.tcurr); }
    else { memset(&a.Rcurr, 0, sizeof(a.Rcurr)); memset(&a.tcurr, 0, sizeof(a.tcurr)); }
    ld_mat(Rprev_inv18, a.Rprev_inv); ld_vec(tprev6, a.tprev);
    a.vmap_curr = (const cfloat *)vmap_curr; a.nmap_curr = (const cfloat *)nmap_curr;
    a.vmap_g_prev = (const cfloat *)vmap_g_prev; a.nmap_g_prev = (const cfloat *)nmap_g_prev;
    a.mstep = map_step; a.intr = Intr{intr4[0], intr4[1], intr4[2], intr4[3]};
    a.vreal = vreal; a.nreal = nreal; a.rstep = rstep;
    a.distThres = distThres; a.angleThres = angleThres; a.cols = cols; a.rows = rows; a.y0 = y0; a.y1 = y1;
    a.ticket = (unsigned *)workspace;
    a.partials = workspace ? (double *)((char *)workspace + 256) : nullptr;
    a.host_records = host_records; a.record_seq = record_seq;
    a.out = sums_dev; a.done_flag = done_flag; a.done_seq = done_seq;
    a.pairs = 0;
    if (done_flag == XS_ICP_PUBLISH_PAIRS) {
        if (pose || host_records) return xs_set_error(hipErrorInvalidValue, "xs_icp: XS_ICP_PUBLISH_PAIRS goes with the plain and posted accumulate calls only");
        a.pairs = 1; a.done_flag = nullptr;
    }
    a.pose = pose; a.pose_host = pose_host; a.load_pose = (pose && Rcurr18) ? 1 : 0;
    a.mailbox = nullptr; a.mailbox_seq = 0;
    int waves = 4;
    const int blocks = icp_blocks(cols, y0, y1, &waves);
    // the ticket word must be zero on first use (xs_icp_workspace_init); every launch leaves it zero
    if (pose) {
        // the reduction, then the pose update it feeds; the completion word belongs to the second kernel
        IcpSolveArgs sa;
        sa.sums = sums_dev; sa.sums_host = sums_host; sa.Rcurr = a.Rcurr; sa.tcurr = a.tcurr; sa.load_pose = a.load_pose;
        sa.pose = pose; sa.pose_host = pose_host; sa.done_flag = done_flag; sa.done_seq = done_seq;
        a.done_flag = nullptr;
        if (a.load_pose) icp_dispatch<POSE_ARGS>(waves, blocks, (hipStream_t)stream, a);
        else icp_dispatch<POSE_DEVICE>(waves, blocks, (hipStream_t)stream, a);
        hipLaunchKernelGGL(k_icp_solve, dim3(1), dim3(128), 0, (hipStream_t)stream, sa);
    } else if (mailbox) {
        a.mailbox = (const unsigned *)mailbox; a.mailbox_seq = mailbox_seq;
        icp_dispatch<POSE_POSTED>(waves, blocks, (hipStream_t)stream, a);
    } else
        icp_dispatch<POSE_ARGS>(waves, blocks, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}

/* estimateCombined(const MatS33& Rcurr, const devComplex3& tcurr, const MapArr& vmap_curr,
 *     const MapArr& nmap_curr, const MatS33& Rprev_inv, const devComplex3& tprev, const Intr&,
 *     const MapArr& vmap_g_prev, const MapArr& nmap_g_prev, float distThres, float angleThres,
 *     DeviceArray2D<devComplexICP>& gbuf, DeviceArray<devComplexICP>& mbuf,
 *     hostComplexICP* A, hostComplexICP* b)                 ICP.h:24-31, ICP.cu:365-429
 * Device half: enqueue the reduction; sums_dev receives 55 doubles = the 27 complex<double>
 * sums in the reference's mbuf order (ICP.cu:266-280) followed by the inlier count.
 * workspace: xs_icp_workspace_bytes() bytes of device memory (replaces gbuf).  [y0, y1): pixel
 * rows covered (0, rows for one GPU).  done_flag (optional, with sums_dev in host-coherent pinned
 * memory): after the sums are written the kernel stores done_seq there with system-scope release,
 * so a host thread can spin on it instead of copying and synchronising.  No synchronisation. */
extern "C" int xs_icp_accumulate(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                                 const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                                 const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres,
                                 int y0, int y1, void *workspace, double *sums_dev, unsigned long long *done_flag,
                                 unsigned long long done_seq, void *stream) {
    if (!Rcurr18 || !tcurr6) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate: null pointer");
    return icp_launch(Rcurr18, tcurr6, vmap_curr, nmap_curr, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                      distThres, angleThres, y0, y1, workspace, sums_dev, done_flag, done_seq, nullptr, nullptr, nullptr, stream,
                      "xs_icp_accumulate: null pointer");
}

/* xs_icp_accumulate with the current-frame maps' real parts given as float planes too (xs_create_vnmaps_real: 3 x rows rows of
 * cols floats, pitch real_step bytes): the kernel reads those instead of vmap_curr / nmap_curr — whose imaginary parts must be
 * zeros, as they are for maps made from a real depth image — and forms (value, 0).  Same sums (an imaginary part that was -0 in the
 * complex map is +0 here).  vmap_curr / nmap_curr may be NULL when the planes are given. */
extern "C" int xs_icp_accumulate_real(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                                      const float *vmap_curr_real, const float *nmap_curr_real, size_t real_step, const float *Rprev_inv18,
                                      const float *tprev6, const float *intr4, const float *vmap_g_prev, const float *nmap_g_prev, size_t map_step,
                                      int rows, int cols, float distThres, float angleThres, int y0, int y1, void *workspace, double *sums_dev,
                                      unsigned long long *done_flag, unsigned long long done_seq, void *stream) {
    if (!Rcurr18 || !tcurr6) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_real: null pointer");
    if (!vmap_curr_real || !nmap_curr_real || real_step < (size_t)cols * 4) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_real: real planes");
    const float *vc = vmap_curr ? vmap_curr : vmap_curr_real, *nc = nmap_curr ? nmap_curr : nmap_curr_real;   // (only tested for null)
    return icp_launch(Rcurr18, tcurr6, vc, nc, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols, distThres, angleThres,
                      y0, y1, workspace, sums_dev, done_flag, done_seq, nullptr, nullptr, nullptr, stream, "xs_icp_accumulate_real: null pointer",
                      nullptr, 0, nullptr, 0, vmap_curr_real, nmap_curr_real, real_step);
}
extern "C" int xs_icp_accumulate_posted_real(const void *mailbox, unsigned mailbox_seq, const float *vmap_curr, const float *nmap_curr,
                                             const float *vmap_curr_real, const float *nmap_curr_real, size_t real_step, const float *Rprev_inv18,
                                             const float *tprev6, const float *intr4, const float *vmap_g_prev, const float *nmap_g_prev,
                                             size_t map_step, int rows, int cols, float distThres, float angleThres, int y0, int y1, void *workspace,
                                             double *sums_dev, unsigned long long *done_flag, unsigned long long done_seq, void *stream) {
    if (!mailbox) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_posted_real: null pointer");
    if (!vmap_curr_real || !nmap_curr_real || real_step < (size_t)cols * 4) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_posted_real: real planes");
    const float *vc = vmap_curr ? vmap_curr : vmap_curr_real, *nc = nmap_curr ? nmap_curr : nmap_curr_real;
    return icp_launch(nullptr, nullptr, vc, nc, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols, distThres, angleThres,
                      y0, y1, workspace, sums_dev, done_flag, done_seq, nullptr, nullptr, nullptr, stream, "xs_icp_accumulate_posted_real: null pointer",
                      mailbox, mailbox_seq, nullptr, 0, vmap_curr_real, nmap_curr_real, real_step);
}

/* estimateCombined with the pose posted after the launch.  The launch is enqueued while the previous
 * iteration is still running; its workgroups become resident, poll `mailbox` (xs_icp_mailbox_bytes()
 * bytes of host-coherent pinned memory, zero before first use) and start on the pixels as soon as
 * xs_icp_post_pose has written Rcurr / tcurr with sequence number mailbox_seq — what is left of an
 * iteration's turnaround is the host's solve and one PCIe read instead of a kernel launch.
 * xs_icp_post_pose(..., cmd = 1) makes the launch return without touching anything (the host left
 * the loop: singular system).  A launch whose pose never arrives gives up after about a second and
 * stores done_seq | 1<<63 to done_flag; re-initialise the workspace (xs_icp_workspace_init) after that.
 * Everything else as xs_icp_accumulate. */
/* estimateCombined with the final addition on the host.  Same kernel and the same per-workgroup records as xs_icp_accumulate, but
 * every workgroup stores its record (56 doubles: 54 sums, inlier count, sequence word) straight into `records_host` — host-coherent
 * pinned memory of xs_icp_records_bytes() bytes — and leaves; the sequence word is written last (system-scope release).  Nothing is
 * gathered on the device: the launch has no ticket, no second pass over the records and no completion word.  The host adds the records
 * with xs_icp_sum_records, which waits for each record's sequence word in index order.  Pose: Rcurr18 / tcurr6, or both NULL with a
 * mailbox (as xs_icp_accumulate_posted).  seq: non-zero, different from the previous launch's on the same buffer. */
extern "C" size_t xs_icp_records_bytes(void) { return (size_t)XS_ICP_MAX_BLOCKS * NP * sizeof(double); }
extern "C" int xs_icp_records_count(int cols, int y0, int y1) { return icp_blocks(cols, y0, y1); }
extern "C" int xs_icp_accumulate_records(const float *Rcurr18, const float *tcurr6, const void *mailbox, unsigned mailbox_seq, const float *vmap_curr,
                                         const float *nmap_curr, const float *Rprev_inv18, const float *tprev6, const float *intr4,
                                         const float *vmap_g_prev, const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres,
                                         float angleThres, int y0, int y1, double *records_host, unsigned long long seq, void *stream) {
    if (!records_host || seq == 0 || (seq >> 63)) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_records: bad records / seq");
    if ((Rcurr18 == nullptr) != (tcurr6 == nullptr) || (!Rcurr18 && !mailbox))
        return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_records: give a pose or a mailbox");
    return icp_launch(Rcurr18, tcurr6, vmap_curr, nmap_curr, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols, distThres,
                      angleThres, y0, y1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream, "xs_icp_accumulate_records: null pointer",
                      Rcurr18 ? nullptr : mailbox, mailbox_seq, records_host, seq);
}
/* Host half: waits (spinning) until record i carries `seq`, adds it, for i = 0 .. count - 1 — one fixed order, so the 55 results are
 * the same bits for the same records whatever order the workgroups finished in.  sums55: 54 sums + inlier count.  Returns 0; 1 if a
 * record reports that its launch gave up waiting for a posted pose; -1 after max_spins polls of one record (<= 0: no limit). */
extern "C" int xs_icp_sum_records(const double *records_host, int count, unsigned long long seq, double *sums55, long long max_spins) {
    if (!records_host || !sums55 || count < 1) return -1;
    double acc[NP];
    for (int k = 0; k < NP; ++k) acc[k] = 0.0;
    for (int i = 0; i < count; ++i) {
        const double *rec = records_host + (size_t)i * NP;
        const volatile unsigned long long *flag = reinterpret_cast<const volatile unsigned long long *>(rec + (NP - 1));
        long long spins = 0;
        unsigned long long seen;
        while ((seen = *flag) != seq) {
            if (seen == (seq | kIcpTimeoutBit)) return 1;
            if (max_spins > 0 && ++spins > max_spins) return -1;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        for (int k = 0; k < NP - 1; ++k) acc[k] += rec[k];
    }
    for (int k = 0; k < NP - 1; ++k) sums55[k] = acc[k];
    return 0;
}

extern "C" size_t xs_icp_mailbox_bytes(void) { return MAILBOX_WORDS * sizeof(unsigned); }
extern "C" int xs_icp_accumulate_posted(const void *mailbox, unsigned mailbox_seq, const float *vmap_curr, const float *nmap_curr,
                                        const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                                        const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres,
                                        int y0, int y1, void *workspace, double *sums_dev, unsigned long long *done_flag,
                                        unsigned long long done_seq, void *stream) {
    if (!mailbox) return xs_set_error(hipErrorInvalidValue, "xs_icp_accumulate_posted: null pointer");
    return icp_launch(nullptr, nullptr, vmap_curr, nmap_curr, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                      distThres, angleThres, y0, y1, workspace, sums_dev, done_flag, done_seq, nullptr, nullptr, nullptr, stream,
                      "xs_icp_accumulate_posted: null pointer", mailbox, mailbox_seq);
}
/* host half of XS_ICP_PUBLISH_PAIRS: every pair {sequence number, sum} arrives as one 16-byte write; done when all 55 carry `seq` */
extern "C" int xs_icp_wait_pairs(const void *pairs_host, unsigned long long seq, double *sums55, long long max_spins) {
    const volatile unsigned long long *p = static_cast<const volatile unsigned long long *>(pairs_host);
    for (long long spins = 0;; ++spins) {
        int have = 0;
        for (int i = 0; i < NS + 1; ++i) have += p[2 * i] == seq;
        if (have == NS + 1) break;
        if (p[0] == (seq | kIcpTimeoutBit)) return 1;
        if (spins >= max_spins) return 2;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < NS + 1; ++i) {
        const unsigned long long bits = p[2 * i + 1];
        memcpy(&sums55[i], &bits, sizeof(double));
    }
    return 0;
}
/* Host side of the mailbox (xs_mailbox.h has the layout and the reasons).  With MOVDIR64B each of the two 64-byte lines goes out as one write,
 * sequence words and payload together; without it: the payload first, a store fence, the four sequence words, a store fence (the poller accepts
 * a 32-byte sector only with its sequence word). */
extern "C" void xs_icp_post_pose(void *mailbox_host, const float *Rcurr18, const float *tcurr6, unsigned mailbox_seq, int cmd) {
    static const bool direct = mailbox_cpu_has_direct_store() && !exp_env_set("XS_MAILBOX_NO_DIRECT_STORE");
    alignas(64) unsigned img[MAILBOX_WORDS];
    mailbox_image(img, Rcurr18, tcurr6, mailbox_seq, cmd);
    if (direct && (reinterpret_cast<uintptr_t>(mailbox_host) % 64) == 0) {
        mailbox_store_fence();   // (behind whatever this thread posted before: nothing is pending, so this costs nothing)
        mailbox_direct_store_64(mailbox_host, img);
        mailbox_direct_store_64(static_cast<char *>(mailbox_host) + 64, img + 16);
        return;
    }
    mailbox_store_fenced((volatile unsigned *)mailbox_host, img);
}
/* A mailbox where polling is cheapest: fine-grained device memory the CPU writes through the large
 * BAR (512 workgroups then poll local memory, not the PCIe link), or — without a large BAR —
 * host-coherent pinned memory.  Zeroed.  *in_device_memory (optional) says which one it is. */
extern "C" int xs_icp_mailbox_alloc(void **mailbox, int *in_device_memory) {
    if (!mailbox) return xs_set_error(hipErrorInvalidValue, "xs_icp_mailbox_alloc: null pointer");
    int dev = 0, large_bar = 0;
    XS_CHECK(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) { large_bar = 0; (void)hipGetLastError(); }
    const char *force_host = getenv("XS_ICP_MAILBOX_HOST");   // (a documented switch: INTEGRATION.md "Environment")
    void *p = nullptr;
    if (large_bar && !(force_host && force_host[0] == '1') &&
        hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
        XS_CHECK(hipMemset(p, 0, 4096));
        XS_CHECK(hipDeviceSynchronize());
        if (in_device_memory) *in_device_memory = 1;
    } else {
        (void)hipGetLastError();
        XS_CHECK(hipHostMalloc(&p, 4096, hipHostMallocCoherent | hipHostMallocMapped));
        memset(p, 0, 4096);
        if (in_device_memory) *in_device_memory = 0;
    }
    *mailbox = p;
    return 0;
}
extern "C" int xs_icp_mailbox_free(void *mailbox, int in_device_memory) {
    if (!mailbox) return 0;
    if (in_device_memory) XS_CHECK(hipFree(mailbox));
    else XS_CHECK(hipHostFree(mailbox));
    return 0;
}

/* One whole ICP iteration on the device: estimateCombined (ICP.cu:365-429) followed by the pose
 * update the reference's host performs before the next launch (KinectFusionReconstruction.cpp:203-224:
 * determinant gate, complex<double> LLT solve, cast to complex<float>, AngleAxis Z*Y*X, Rcurr / tcurr
 * composition).  pose_state: xs_icp_pose_state_bytes() bytes of device memory holding
 * {float R[18]; float t[6]; int status; int iters; double det; double pad[2]} (128 bytes); when Rcurr18 / tcurr6 are
 * given the launch starts from them (first iteration of a frame) and otherwise from the state the
 * previous launch on the stream left.  status: 0 ok, 1 |det| < 1e-15, 2 NaN; once non-zero the
 * following launches return at once (the reference leaves PoseEstimate there).  sums_dev: device
 * memory, as in xs_icp_accumulate; sums_host / pose_state_host (optional, host-coherent pinned
 * memory) receive a copy of the 55 sums and of the state after every solve, then done_seq is
 * published to done_flag (optional).  Two launches (reduction, one-workgroup solve); no synchronisation. */
extern "C" size_t xs_icp_pose_state_bytes(void) { return sizeof(IcpPoseState); }
extern "C" int xs_icp_iterate(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                              const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                              const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres,
                              void *workspace, double *sums_dev, double *sums_host, void *pose_state, void *pose_state_host,
                              unsigned long long *done_flag, unsigned long long done_seq, void *stream) {
    if (!pose_state || ((Rcurr18 == nullptr) != (tcurr6 == nullptr))) return xs_set_error(hipErrorInvalidValue, "xs_icp_iterate: null pointer");
    return icp_launch(Rcurr18, tcurr6, vmap_curr, nmap_curr, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step, rows, cols,
                      distThres, angleThres, 0, rows, workspace, sums_dev, done_flag, done_seq, (IcpPoseState *)pose_state,
                      (IcpPoseState *)pose_state_host, sums_host, stream, "xs_icp_iterate: null pointer");
}

/* estimateCombined whole (ICP.cu:365-429): the reduction, then — where the reference synchronises the device and downloads (:414-417) —
 * the kernel's last workgroup has written the 27 sums and the inlier count straight into host-coherent pinned memory, each with the launch's
 * number in the same 16-byte store (XS_ICP_PUBLISH_PAIRS), and the host spins until all of them carry it: the call returns when the launch, and
 * with it everything before it on the stream, has completed, without a copy or a stream drain (round 6: ~10 us less per iteration for a caller
 * that changes nothing).  Unpacks
 * into the symmetric 6x6 A (A[i*6+j] = A[j*6+i]) and b, both complex<double> (re, im) pairs.  sums_dev (optional): also receives the 55
 * doubles, by an asynchronous copy the call does not wait for.  inliers may be NULL.  One call at a time per host thread. */
extern "C" int xs_estimate_combined(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                                    const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                                    const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres,
                                    void *workspace, double *sums_dev, double *A72_host, double *b12_host, long long *inliers,
                                    void *stream) {
    static thread_local double *host = nullptr;            // the 55 sums as doubles (what xs_icp_unpack and the optional copy read)
    static thread_local void *pairs = nullptr;             // XS_ICP_PUBLISH_PAIRS: 55 x {sequence number, sum} written by the launch
    static thread_local unsigned long long seq = 0;
    if (!host) {
        XS_CHECK(hipHostMalloc((void **)&host, 64 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
        memset(host, 0, 64 * sizeof(double));
        XS_CHECK(hipHostMalloc(&pairs, XS_ICP_PAIRS_BYTES, hipHostMallocCoherent | hipHostMallocMapped));
        memset(pairs, 0, XS_ICP_PAIRS_BYTES);
    }
    ++seq;
    int rc = xs_icp_accumulate(Rcurr18, tcurr6, vmap_curr, nmap_curr, Rprev_inv18, tprev6, intr4, vmap_g_prev, nmap_g_prev, map_step,
                               rows, cols, distThres, angleThres, 0, rows, workspace, static_cast<double *>(pairs), XS_ICP_PUBLISH_PAIRS, seq, stream);
    if (rc) return rc;
    if (rows <= 0 || cols <= 0) { XS_CHECK(hipStreamSynchronize((hipStream_t)stream)); memset(host, 0, 55 * sizeof(double)); }   // (nothing was launched: nothing publishes)
    else if (xs_icp_wait_pairs(pairs, seq, host, 4000000000LL) != 0)
        return xs_set_error(hipErrorLaunchTimeOut, "xs_estimate_combined: the launch never published its sums");
    xs_icp_unpack(host, A72_host, b12_host);
    if (inliers) *inliers = (long long)host[54];
    if (sums_dev) XS_CHECK(hipMemcpyAsync(sums_dev, host, 55 * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

/* ICP.cu:419-428 on the host: 27 (re, im) sums -> symmetric A[36] and b[6], complex<double> */
extern "C" void xs_icp_unpack(const double *sums54, double *A72, double *b12) {
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const double re = sums54[2 * shift], im = sums54[2 * shift + 1];
            ++shift;
            if (j == 6) { b12[2 * i] = re; b12[2 * i + 1] = im; }
            else {
                A72[2 * (j * 6 + i)] = re; A72[2 * (j * 6 + i) + 1] = im;
                A72[2 * (i * 6 + j)] = re; A72[2 * (i * 6 + j) + 1] = im;
            }
        }
}

// ---- self-test of the gate shortcut ---------------------------------------------------------
__global__ void k_icp_gate_check(const float *z2n, int n, float thres, int or_equal, unsigned *counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cfloat z(z2n[2 * i], z2n[2 * i + 1]);
    const bool fast = re_sqrt_exceeds(z, thres, or_equal != 0);
    const float r = sqrt(z).re;
    const bool full = or_equal ? r >= thres : r > thres;
    if (fast != full) atomicAdd(&counts[0], 1u);
    const float lo2 = fmaxf(z.re, 0.0f), hi2 = fabsf(z.re) + fabsf(z.im), t2 = thres * thres;
    if (!(hi2 * (1.0f + 1e-5f) < t2) && !(lo2 > t2 * (1.0f + 1e-5f))) atomicAdd(&counts[1], 1u);
}
/* Gate shortcut of the ICP search (re_sqrt_exceeds) against the full complex square root on n complex values (device, re / im
 * interleaved): counts_dev[0] = values on which the two disagree (must be 0), counts_dev[1] = values the shortcut could not
 * settle from its bounds.  counts_dev: two zeroed 32-bit words of device memory.  Test hook; no synchronisation. */
extern "C" int xs_icp_gate_selftest(const float *z2n_dev, int n, float thres, int or_equal, unsigned *counts_dev, void *stream) {
    if (!z2n_dev || !counts_dev || n < 0) return xs_set_error(hipErrorInvalidValue, "xs_icp_gate_selftest: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_icp_gate_check, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, z2n_dev, n, thres, or_equal, counts_dev);
    XS_CHECK(hipGetLastError());
    return 0;
}
