// xs_launchers.hpp — the reference's kernel-launcher API, re-created over the C ABI
// (include/xslam_amd.h) so orchestrator code written against
//   XKinectFusion/include/{TsdfVolume.h:16, TsdfFusion.h:40-60, RayCaster.h:21-25, ICP.h:24-31,
//                          Map.h:16-54}
// links against libxslam_hip.so with the same names, argument order and meaning.  Errors print
// and exit(-1) like cudaSafeCall (Common/include/cx.h:124-130).  Synchronisation follows the
// reference: initVolume, integrateTsdfVolume, resizeV/NMap and estimateCombined return after the
// stream has drained; raycast, bilateralFilter, pyrDown, createV/NMap do not.
#pragma once
#include "../../include/xslam_amd.h"
#include "xs_types.hpp"
#include "host_algebra.hpp"

namespace xs_host {
inline void check_rc(int rc, const char *what) {
    if (rc != 0) {
        printf("HIP error(%s): %s\n", what, xs_last_error());
        exit(-1);
    }
}
inline void sync() { hipSafeCall(hipStreamSynchronize(current_stream())); }
// scratch the launchers need (the reference allocates gbuf / thrust vectors per call): one set per HOST THREAD, like current_stream() — two
// threads driving two streams must not share a workspace (a reduce workspace serves one launch at a time)
struct Scratch {
    DeviceArray<unsigned char> icp_ws, reduce_ws, integrate_ws;
    DeviceArray<double> sums;
    DeviceArray<float> ray_ws;
    static Scratch &get() { static thread_local Scratch s; return s; }
    void *icp() {
        if (icp_ws.size() != xs_icp_workspace_bytes()) {
            icp_ws.create(xs_icp_workspace_bytes());
            check_rc(xs_icp_workspace_init(icp_ws.ptr(), current_stream()), "icp workspace");
        }
        return icp_ws.ptr();
    }
    // the integrate kernel's brick list, box classes and tile table for a volume of this size (include/xslam_amd.h: xs_integrate_workspace_bytes)
    void *integrate(const int *res) {
        const size_t bytes = xs_integrate_workspace_bytes(res, res[2]);
        if (integrate_ws.size() != bytes) integrate_ws.create(bytes);
        return integrate_ws.ptr();
    }
    // the raycast's crossing-time plane (rows x cols floats): with it the ray is a march kernel + a crossing kernel (same maps, higher occupancy)
    float *ray(int rows, int cols) { if (ray_ws.size() != (size_t)rows * cols) ray_ws.create((size_t)rows * cols); return ray_ws.ptr(); }
    void *reduce() {
        if (reduce_ws.size() != xs_tsdf_reduce_workspace_bytes()) {
            reduce_ws.create(xs_tsdf_reduce_workspace_bytes());
            check_rc(xs_tsdf_reduce_workspace_init(reduce_ws.ptr(), current_stream()), "reduce workspace");
        }
        return reduce_ws.ptr();
    }
    double *sum_buf() { if (sums.size() != 64) sums.create(64); return sums.ptr(); }
};
}  // namespace xs_host

// TsdfVolume.h:16
inline void initVolume(PtrStep<short> /*volume: allocated but never read in the reference*/, PtrStep<float> value_volume,
                       PtrStep<int> weight_volume, PtrStep<float> grad_volume, const int3 &volume_resolution) {
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    xs_host::check_rc(xs_init_volume(value_volume.data, weight_volume.data, grad_volume.data, value_volume.step, res, 0, res[2],
                                     xs_host::current_stream()), "initVolume");
    xs_host::sync();
}

// TsdfFusion.h:40-45.  depthScaled is kept resident (create() is a no-op when the size is unchanged); the two launches of
// TsdfFusion.cu:173-201 — scaleDepthKernal, tsdfFusionKernal — with the second one walking the brick list the library keeps in the
// per-thread scratch (the same voxels bit for bit as the walk over every voxel: tests/test_integrate_gpu.py).
inline void integrateTsdfVolume(const PtrStepSz<ushort> &depth, const Intr &intr, int max_weight, const int3 &volume_resolution,
                                float voxel_size, const MatS33 &Rv2c, const devComplex3 &tv2c, const devComplex3 & /*tc2v*/,
                                float tranc_dist, PtrStep<float> value_volume, PtrStep<int> weight_volume, PtrStep<float> grad_volume,
                                DeviceArray2D<float> &depthScaled, int /*frame_id*/, float threshold = 0.0f, float /*k*/ = 0.0f,
                                unsigned long long *updated_dev = nullptr, bool synchronise = true) {
    depthScaled.create(depth.rows, depth.cols);
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    hipStream_t st = xs_host::current_stream();
    xs_host::check_rc(xs_scale_depth(depth.data, depth.step, depth.rows, depth.cols, depthScaled.ptr(), depthScaled.step(), st), "scaleDepth");
    xs_host::check_rc(xs_integrate_scaled(depthScaled.ptr(), depthScaled.step(), depth.rows, depth.cols, &intr.fx, max_weight, res, voxel_size,
                                          &Rv2c.data[0].x.re, &tv2c.x.re, tranc_dist, value_volume.data, weight_volume.data, grad_volume.data,
                                          value_volume.step, threshold, 0, res[2], updated_dev, nullptr, xs_host::Scratch::get().integrate(res), st),
                      "integrateTsdfVolume");
    if (synchronise) xs_host::sync();
}

// RayCaster.h:21-25 (no synchronisation, RayCaster.cu:367)
inline void raycast(const Intr &intr, const MatS33 &Rc2v, const devComplex3 &tc2v, const MatS33 &Rv2w, const devComplex3 &tv2w,
                    float tranc_dist, const int3 &volume_resolution, float voxel_size, const PtrStep<float> &value_volume,
                    const PtrStep<float> &grad_volume, MapArr &vmap, MapArr &nmap, unsigned long long *hits_dev = nullptr,
                    float *workspace = nullptr, xs_raycast_opts *opts = nullptr) {   // opts: sign map, pyramid outputs, completion event (xs_raycast_ex)
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    xs_host::check_rc(xs_raycast_ex(&intr.fx, &Rc2v.data[0].x.re, &tc2v.x.re, &Rv2w.data[0].x.re, &tv2w.x.re, tranc_dist, res, voxel_size,
                                    value_volume.data, grad_volume.data, value_volume.step, &vmap.ptr()->re, &nmap.ptr()->re, vmap.step(),
                                    vmap.rows() / 3, vmap.cols(), hits_dev, workspace ? workspace : xs_host::Scratch::get().ray(vmap.rows() / 3, vmap.cols()), opts,
                                    xs_host::current_stream()), "raycast");
}

// ICP.h:24-31.  gbuf / mbuf are accepted for signature compatibility; the single-launch
// reduction keeps its own workspace.  Returns when the launch has completed (the host spins on the
// pinned record its last workgroup writes: what the reference's cudaDeviceSynchronize + download give).
inline void estimateCombined(const MatS33 &Rcurr, const devComplex3 &tcurr, const MapArr &vmap_curr, const MapArr &nmap_curr,
                             const MatS33 &Rprev_inv, const devComplex3 &tprev, const Intr &intr, const MapArr &vmap_g_prev,
                             const MapArr &nmap_g_prev, float distThres, float angleThres, DeviceArray2D<devComplexICP> & /*gbuf*/,
                             DeviceArray<devComplexICP> & /*mbuf*/, hostComplexICP *matrixA_host, hostComplexICP *vectorB_host,
                             long long *inliers = nullptr) {
    auto &S = xs_host::Scratch::get();
    xs_host::check_rc(xs_estimate_combined(&Rcurr.data[0].x.re, &tcurr.x.re, &vmap_curr.ptr()->re, &nmap_curr.ptr()->re,
                                           &Rprev_inv.data[0].x.re, &tprev.x.re, &intr.fx, &vmap_g_prev.ptr()->re, &nmap_g_prev.ptr()->re,
                                           vmap_curr.step(), vmap_curr.rows() / 3, vmap_curr.cols(), distThres, angleThres, S.icp(),
                                           nullptr /* no device copy of the sums: they arrive in pinned memory */, reinterpret_cast<double *>(matrixA_host), reinterpret_cast<double *>(vectorB_host),
                                           inliers, xs_host::current_stream()), "estimateCombined");
}

// Map.h:16-54
inline void bilateralFilter(const DeviceArray2D<ushort> &src, MapArr &dst) {
    xs_host::check_rc(xs_bilateral_filter(src.ptr(), src.step(), src.rows(), src.cols(), &dst.ptr()->re, dst.step(), xs_host::current_stream()), "bilateralFilter");
}
inline void pyrDown(const MapArr &src, MapArr &dst) {
    dst.create(src.rows() / 2, src.cols() / 2);
    xs_host::check_rc(xs_pyr_down(&src.ptr()->re, src.step(), src.rows(), src.cols(), &dst.ptr()->re, dst.step(), xs_host::current_stream()), "pyrDown");
}
inline void createVMap(const Intr &intr, const MapArr &depth, MapArr &vmap) {
    vmap.create(depth.rows() * 3, depth.cols());
    xs_host::check_rc(xs_create_vmap(&intr.fx, &depth.ptr()->re, depth.step(), depth.rows(), depth.cols(), &vmap.ptr()->re, vmap.step(), xs_host::current_stream()), "createVMap");
}
inline void createNMap(const MapArr &vmap, MapArr &nmap) {
    nmap.create(vmap.rows(), vmap.cols());
    xs_host::check_rc(xs_create_nmap(&vmap.ptr()->re, &nmap.ptr()->re, vmap.step(), vmap.rows() / 3, vmap.cols(), xs_host::current_stream()), "createNMap");
}
inline void resizeVMap(const MapArr &input, MapArr &output, bool synchronise = true) {
    output.create((input.rows() / 3 / 2) * 3, input.cols() / 2);
    xs_host::check_rc(xs_resize_vmap(&input.ptr()->re, input.step(), input.rows() / 3, input.cols(), &output.ptr()->re, output.step(), xs_host::current_stream()), "resizeVMap");
    if (synchronise) xs_host::sync();
}
inline void resizeNMap(const MapArr &input, MapArr &output, bool synchronise = true) {
    output.create((input.rows() / 3 / 2) * 3, input.cols() / 2);
    xs_host::check_rc(xs_resize_nmap(&input.ptr()->re, input.step(), input.rows() / 3, input.cols(), &output.ptr()->re, output.step(), xs_host::current_stream()), "resizeNMap");
    if (synchronise) xs_host::sync();
}

// TsdfFusion.h:48-60.  The thrust scratch vectors of the reference become optional dense device
// arrays (pass empty arrays to skip the per-voxel volumes); gt is a dense unpitched device array.
inline float4 ComputeLocalTsdf_hessian(const PtrStepSz<ushort> &depth, const Intr &intr, DeviceArray2D<float> &depthScaled,
                                        const int3 &volume_resolution, float voxel_size, const MatD33 &Rv2c, const devDComplex3 &tv2c,
                                        float tranc_dist, float /*threshold*/, float /*k*/, DeviceArray<float> &gt_vec,
                                        DeviceArray<float> &real_vec, DeviceArray<float> &grad_vec, DeviceArray<float> &hessian_vec,
                                        DeviceArray<int> &count_vec) {
    depthScaled.create(depth.rows, depth.cols);
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    auto &S = xs_host::Scratch::get();
    hipStream_t st = xs_host::current_stream();
    xs_host::check_rc(xs_scale_depth(depth.data, depth.step, depth.rows, depth.cols, depthScaled.ptr(), depthScaled.step(), st), "scaleDepth");
    const bool vols = !real_vec.empty() && !grad_vec.empty() && !hessian_vec.empty() && !count_vec.empty();
    xs_host::check_rc(xs_compute_local_tsdf_hessian(depthScaled.ptr(), depthScaled.step(), depth.rows, depth.cols, &intr.fx, res, voxel_size,
                                                    &Rv2c.data[0].x.re_re, &tv2c.x.re_re, tranc_dist, gt_vec.ptr(),
                                                    vols ? real_vec.ptr() : nullptr, vols ? grad_vec.ptr() : nullptr,
                                                    vols ? hessian_vec.ptr() : nullptr, vols ? count_vec.ptr() : nullptr, 0, res[2],
                                                    S.reduce(), S.sum_buf(), st), "ComputeLocalTsdf_hessian");
    double h[4];
    hipSafeCall(hipMemcpyAsync(h, S.sum_buf(), sizeof(h), hipMemcpyDeviceToHost, st));
    xs_host::sync();
    return float4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
inline float2 ComputeLocalTsdf_loss(const PtrStepSz<ushort> &depth, const Intr &intr, DeviceArray2D<float> &depthScaled,
                                     const int3 &volume_resolution, float voxel_size, const Mat33 &Rv2c, const float3 &tv2c,
                                     float tranc_dist, float /*threshold*/, float /*k*/, DeviceArray<float> &gt_vec,
                                     DeviceArray<float> &real_vec, DeviceArray<int> &count_vec) {
    depthScaled.create(depth.rows, depth.cols);
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    auto &S = xs_host::Scratch::get();
    hipStream_t st = xs_host::current_stream();
    xs_host::check_rc(xs_scale_depth(depth.data, depth.step, depth.rows, depth.cols, depthScaled.ptr(), depthScaled.step(), st), "scaleDepth");
    const bool vols = !real_vec.empty() && !count_vec.empty();
    xs_host::check_rc(xs_compute_local_tsdf_loss(depthScaled.ptr(), depthScaled.step(), depth.rows, depth.cols, &intr.fx, res, voxel_size,
                                                 &Rv2c.data[0].x, &tv2c.x, tranc_dist, gt_vec.ptr(), vols ? real_vec.ptr() : nullptr,
                                                 vols ? count_vec.ptr() : nullptr, 0, res[2], S.reduce(), S.sum_buf(), st), "ComputeLocalTsdf_loss");
    double h[2];
    hipSafeCall(hipMemcpyAsync(h, S.sum_buf(), sizeof(h), hipMemcpyDeviceToHost, st));
    xs_host::sync();
    return float2{(float)h[0], (float)h[1]};
}

// --- ExtractPointCloud.h:19-23 (surface export; real-valued) ---------------------------------------
// extractPoints returns the number of points stored (at most output.size), extractNormals fills one
// normal per point.  Both return after the stream has drained, as the reference.
inline size_t extractPoints(const PtrStep<float> &value_volume, const PtrStep<int> & /*weight_volume*/, const PtrStep<float> & /*grad_volume*/,
                            const int3 &volume_resolution, float voxel_size, PtrSz<float3> output, int zs0 = 0, int z0 = 0, int z1 = -1) {
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    static thread_local DeviceArray<unsigned char> ws;
    if (ws.size() < xs_extract_workspace_bytes(res)) ws.create(xs_extract_workspace_bytes(res));
    size_t count = 0;
    xs_host::check_rc(xs_extract_points(value_volume.data, value_volume.step, res, voxel_size, zs0, z0, z1 < 0 ? res[2] - 1 : z1,
                                        reinterpret_cast<float *>(output.data), output.size, ws.ptr(), &count, nullptr, xs_host::current_stream()),
                      "extractPoints");
    return count;
}
inline void extractNormals(const PtrStep<float> &value_volume, const PtrStep<int> & /*weight_volume*/, const PtrStep<float> & /*grad_volume*/,
                           const int3 &volume_resolution, float voxel_size, PtrSz<float3> points, PtrSz<float3> normal, int zs0 = 0, int zs1 = -1) {
    const int res[3] = {volume_resolution.x, volume_resolution.y, volume_resolution.z};
    xs_host::check_rc(xs_extract_normals(value_volume.data, value_volume.step, res, voxel_size, zs0, zs1 < 0 ? res[2] : zs1,
                                         reinterpret_cast<const float *>(points.data), points.size, reinterpret_cast<float *>(normal.data),
                                         xs_host::current_stream()), "extractNormals");
    xs_host::sync();
}
