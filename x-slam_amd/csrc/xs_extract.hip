// xs_extract.hip — surface point / normal extraction from the TSDF volume for gfx950.  Replaces
// XKinectFusion/src/ExtractPointCloud.cu: Scanner / extractKernel (:25-185), extractPoints (:188-211),
// ExtractNormals / extractNormalsKernel (:214-341), extractNormals (:344-362).  Real-valued export
// (the complex part of the volume is not read), SURVEY.md section 8(f) #4.
//
// Reference shape: a 32x6 workgroup walks z; per plane each thread finds up to three zero crossings
// (towards +x, +y, +z), a 32-wide warp scans the counts through volatile shared memory (relying on
// warp-synchronous execution, ExtractPointCloud.h:34-49), takes a slot with one atomicAdd per warp per
// plane, and copies the points out through three shared arrays; the order of the output depends on
// the order the atomics land in.
//
// Here: lane = x (64 consecutive voxels of a row: every volume read is one coalesced 256-byte
// segment), a workgroup covers 64x4 columns and walks its z range.  Two passes over the same code:
// pass 1 counts the crossings of every workgroup, one small kernel turns the counts into exclusive
// offsets, pass 2 recomputes the crossings and writes them at offset + (plane order, wave order, lane
// order) positions computed with wave ballots — no shared-memory staging, no warp-synchronous
// assumptions, and the output order is deterministic (workgroup, then z, then y, then x, then
// direction).  The value volume is read twice (2 x 4 B per voxel): an export path, not a per-frame one.
#include "xs_device.h"
#include "../../include/xslam_amd.h"

using namespace xs;

struct ExtractArgs {
    const float *value; size_t vstep;
    int X, Y, Z;      // full resolution
    int zs0;          // first stored plane (value points at it)
    int z0, z1;       // planes whose crossings this launch reports (z1 <= Z - 1: plane z looks at z + 1)
    float voxel_size;
    unsigned *block_counts;   // [blocks] pass 1 out / pass 2 in (exclusive offsets)
    float *out;               // xyz triples
    unsigned capacity;        // points that fit in out
};

namespace {
__device__ __forceinline__ float fetch(const ExtractArgs &a, int x, int y, int z) {  // Scanner::fetch, :38-49 (the % wrap is the identity here)
    return row_ptr(a.value, a.vstep, a.Y * (z - a.zs0) + y)[x];
}
// the up-to-three crossings of voxel (x, y, z), ExtractPointCloud.cu:69-117; returns their number
__device__ __forceinline__ int crossings(const ExtractArgs &a, int x, int y, int z, float (&p)[3][3]) {
    int n = 0;
    if (!(x < a.X - 1 && y < a.Y - 1)) return 0;
    const float F = fetch(a, x, y, z);
    if (!(F < 0.99f)) return 0;
    const float Vx = (x + 0.5f) * a.voxel_size, Vy = (y + 0.5f) * a.voxel_size, Vz = (z + 0.5f) * a.voxel_size;
    {
        const float Fn = fetch(a, x + 1, y, z);
        if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) { p[n][0] = Vx - (F / (Fn - F)) * a.voxel_size; p[n][1] = Vy; p[n][2] = Vz; ++n; }
    }
    {
        const float Fn = fetch(a, x, y + 1, z);
        if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) { p[n][0] = Vx; p[n][1] = Vy - (F / (Fn - F)) * a.voxel_size; p[n][2] = Vz; ++n; }
    }
    {
        const float Fn = fetch(a, x, y, z + 1);
        if (Fn < 0.99f && ((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) { p[n][0] = Vx; p[n][1] = Vy; p[n][2] = Vz - (F / (Fn - F)) * a.voxel_size; ++n; }
    }
    return n;
}
}  // namespace

// PASS 1 counts, PASS 2 writes.  One workgroup = 64 (x) x 4 (y) columns, all of [z0, z1).
template <int PASS>
__global__ void __launch_bounds__(256) k_extract(const ExtractArgs a) {
    const int x = threadIdx.x + blockIdx.x * 64, y = threadIdx.y + blockIdx.y * 4;
    const int lane = threadIdx.x, wave = threadIdx.y;
    const unsigned bid = blockIdx.x + gridDim.x * blockIdx.y;
    __shared__ unsigned s_wave[4];
    unsigned running = (PASS == 2) ? a.block_counts[bid] : 0u;  // next free slot of this workgroup
    const bool inside = x < a.X && y < a.Y;
    for (int z = a.z0; z < a.z1; ++z) {
        float p[3][3];
        const int n = inside ? crossings(a, x, y, z, p) : 0;
        // points of the wave in (lane, direction) order: lanes before this one contribute their counts
        const unsigned long long m1 = __ballot(n > 0), m2 = __ballot(n > 1), m3 = __ballot(n > 2);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned before = __popcll(m1 & below) + __popcll(m2 & below) + __popcll(m3 & below);
        const unsigned wave_total = __popcll(m1) + __popcll(m2) + __popcll(m3);
        if (lane == 0) s_wave[wave] = wave_total;
        __syncthreads();
        unsigned wave_base = 0, plane_total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) wave_base += s_wave[w]; plane_total += s_wave[w]; }
        if (PASS == 2) {
            const unsigned at = running + wave_base + before;
            for (int l = 0; l < n; ++l)
                if (at + l < a.capacity) { a.out[3 * (size_t)(at + l)] = p[l][0]; a.out[3 * (size_t)(at + l) + 1] = p[l][1]; a.out[3 * (size_t)(at + l) + 2] = p[l][2]; }
        }
        running += plane_total;
        __syncthreads();
    }
    if (PASS == 1 && threadIdx.x == 0 && threadIdx.y == 0) a.block_counts[bid] = running;
}

// exclusive scan of the per-workgroup counts (a few thousand entries): one workgroup, wave scans + LDS
__global__ void __launch_bounds__(256) k_extract_scan(unsigned *counts, unsigned n, unsigned *total) {
    __shared__ unsigned s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned carry = 0;
    for (unsigned base = 0; base < n; base += 256) {
        const unsigned i = base + tid;
        const unsigned v = i < n ? counts[i] : 0u;
        unsigned s = v;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(s, off, 64);
            if (lane >= off) s += t;
        }
        if (lane == 63) s_part[wave] = s;
        __syncthreads();
        unsigned wbase = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) wbase += s_part[w]; chunk += s_part[w]; }
        if (i < n) counts[i] = carry + wbase + s - v;
        carry += chunk;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

extern "C" size_t xs_extract_workspace_bytes(const int *res) {
    if (!res) return 0;
    return 256 + (size_t)div_up(res[0], 64) * div_up(res[1], 4) * sizeof(unsigned);
}

/* size_t extractPoints(const PtrStep<float>& value_volume, const PtrStep<int>& weight_volume,
 *     const PtrStep<float>& grad_volume, const int3& volume_resolution, float voxel_size,
 *     PtrSz<float3> output)                              ExtractPointCloud.h:19-20, .cu:188-211
 * Zero crossings of the TSDF between axis neighbours (both samples < 0.99, opposite signs), linearly
 * interpolated, for voxels of planes [z0, z1) (z1 <= res[2] - 1; the whole volume: 0, res[2] - 1).
 * value points at stored plane zs0 (a z-slab; 0 for the whole volume) and must hold planes up to z1.
 * The weight and grad volumes of the reference signature are not read by the reference either.
 * points_dev: capacity x 3 floats.  *count_host receives min(found, capacity) — the reference's return
 * value — and, if found_host is given, the number found.  Synchronises the stream (as the reference). */
extern "C" int xs_extract_points(const float *value, size_t vol_step, const int *res, float voxel_size, int zs0, int z0, int z1,
                                 float *points_dev, size_t capacity, void *workspace, size_t *count_host, size_t *found_host,
                                 void *stream) {
    if (!value || !res || !points_dev || !workspace || !count_host) return xs_set_error(hipErrorInvalidValue, "xs_extract_points: null pointer");
    if (z0 < zs0 || z1 > res[2] - 1 || z1 < z0 || (vol_step % 4) != 0) return xs_set_error(hipErrorInvalidValue, "xs_extract_points: bad plane range");
    *count_host = 0;
    if (found_host) *found_host = 0;
    if (z1 == z0 || res[0] < 2 || res[1] < 2) return 0;
    ExtractArgs a;
    a.value = value; a.vstep = vol_step; a.X = res[0]; a.Y = res[1]; a.Z = res[2]; a.zs0 = zs0; a.z0 = z0; a.z1 = z1;
    a.voxel_size = voxel_size; a.out = points_dev;
    a.capacity = capacity > 0xffffffffull ? 0xffffffffu : (unsigned)capacity;
    unsigned *total = (unsigned *)workspace;
    a.block_counts = (unsigned *)((char *)workspace + 256);
    hipStream_t st = (hipStream_t)stream;
    dim3 block(64, 4), grid(div_up(a.X, 64), div_up(a.Y, 4));
    const unsigned nblocks = grid.x * grid.y;
    hipLaunchKernelGGL(k_extract<1>, grid, block, 0, st, a);
    hipLaunchKernelGGL(k_extract_scan, dim3(1), dim3(256), 0, st, a.block_counts, nblocks, total);
    hipLaunchKernelGGL(k_extract<2>, grid, block, 0, st, a);
    XS_CHECK(hipGetLastError());
    unsigned h = 0;
    XS_CHECK(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, st));
    XS_CHECK(hipStreamSynchronize(st));
    if (found_host) *found_host = h;
    *count_host = h < a.capacity ? h : a.capacity;
    return 0;
}

// ---- normals at the extracted points ---------------------------------------------------------
struct NormalArgs {
    const float *value; size_t vstep; int X, Y, Z, zs0, zs1;
    float voxel_size;
    const float *points; float *normals; size_t n;
};
namespace {
__device__ __forceinline__ float read_tsdf(const NormalArgs &a, int x, int y, int z) {  // ExtractNormals::readTsdf, :226-238
    z = min(max(z, a.zs0), a.zs1 - 1);  // stay inside the resident planes (identity for the whole volume)
    return row_ptr(a.value, a.vstep, a.Y * (z - a.zs0) + y)[x];
}
__device__ __forceinline__ float interp(const NormalArgs &a, float px, float py, float pz) {  // :312-340
    const float vs = a.voxel_size;
    int gx = __float2int_rd(px / vs), gy = __float2int_rd(py / vs), gz = __float2int_rd(pz / vs);
    const float vx = (gx + 0.5f) * vs, vy = (gy + 0.5f) * vs, vz = (gz + 0.5f) * vs;
    gx = (px < vx) ? (gx - 1) : gx;
    gy = (py < vy) ? (gy - 1) : gy;
    gz = (pz < vz) ? (gz - 1) : gz;
    const float fa = (px - (gx + 0.5f) * vs) / vs, fb = (py - (gy + 0.5f) * vs) / vs, fc = (pz - (gz + 0.5f) * vs) / vs;
    return read_tsdf(a, gx + 0, gy + 0, gz + 0) * (1 - fa) * (1 - fb) * (1 - fc) + read_tsdf(a, gx + 0, gy + 0, gz + 1) * (1 - fa) * (1 - fb) * fc +
           read_tsdf(a, gx + 0, gy + 1, gz + 0) * (1 - fa) * fb * (1 - fc) + read_tsdf(a, gx + 0, gy + 1, gz + 1) * (1 - fa) * fb * fc +
           read_tsdf(a, gx + 1, gy + 0, gz + 0) * fa * (1 - fb) * (1 - fc) + read_tsdf(a, gx + 1, gy + 0, gz + 1) * fa * (1 - fb) * fc +
           read_tsdf(a, gx + 1, gy + 1, gz + 0) * fa * fb * (1 - fc) + read_tsdf(a, gx + 1, gy + 1, gz + 1) * fa * fb * fc;
}
}  // namespace

__global__ void __launch_bounds__(256) k_extract_normals(const NormalArgs a) {
    const size_t idx = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    if (idx >= a.n) return;
    const float px = a.points[3 * idx], py = a.points[3 * idx + 1], pz = a.points[3 * idx + 2];
    const float vs = a.voxel_size;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int gx = __float2int_rd(px / vs), gy = __float2int_rd(py / vs), gz = __float2int_rd(pz / vs);
    if (gx > 1 && gy > 1 && gz > 1 && gx < a.X - 2 && gy < a.Y - 2 && gz < a.Z - 2) {
        nx = interp(a, px + vs, py, pz) - interp(a, px - vs, py, pz);
        ny = interp(a, px, py + vs, pz) - interp(a, px, py - vs, pz);
        nz = interp(a, px, py, pz + vs) - interp(a, px, py, pz - vs);
        const float norm = nx * nx + ny * ny + nz * nz;  // the reference divides by the squared length (:303-304)
        nx = nx / norm; ny = ny / norm; nz = nz / norm;
    }
    a.normals[3 * idx] = nx; a.normals[3 * idx + 1] = ny; a.normals[3 * idx + 2] = nz;
}

/* void extractNormals(value, weight, grad, volume_resolution, voxel_size, PtrSz<float3> points,
 *     PtrSz<float3> normal)                               ExtractPointCloud.h:22-23, .cu:344-362
 * Central differences of the trilinearly interpolated TSDF one voxel either side of each point, divided
 * by their squared length as in the reference; (0, 0, 0) within two voxels of the border.  value holds
 * stored planes [zs0, zs1) (whole volume: 0, res[2]); samples outside are clamped to them.  No sync. */
extern "C" int xs_extract_normals(const float *value, size_t vol_step, const int *res, float voxel_size, int zs0, int zs1,
                                  const float *points_dev, size_t n, float *normals_dev, void *stream) {
    if (!value || !res || !points_dev || !normals_dev) return xs_set_error(hipErrorInvalidValue, "xs_extract_normals: null pointer");
    if (n == 0) return 0;
    NormalArgs a;
    a.value = value; a.vstep = vol_step; a.X = res[0]; a.Y = res[1]; a.Z = res[2]; a.zs0 = zs0; a.zs1 = zs1;
    a.voxel_size = voxel_size; a.points = points_dev; a.normals = normals_dev; a.n = n;
    hipLaunchKernelGGL(k_extract_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    XS_CHECK(hipGetLastError());
    return 0;
}
