// reference_shape.cpp — see reference_shape.hpp.  Every device-side step below is a call of a
// reference-signature launcher with the reference's own argument list; the host algebra between
// them is host_algebra.hpp's (the Eigen expressions of the reference written out).  Cited lines are
// XKinectFusion/src/KinectFusionReconstruction.cpp.
#include "reference_shape.hpp"
#include "../../include/xslam_amd_pipeline.h"
#include <cmath>
#include <cstring>
#include <exception>
#include <iostream>

using namespace xs_host;

// :9-73
void ReferenceCallShape::SetYamlParameters(const FlatYaml &config) {
    volume_resolution = Vector3i(config.as<int>("tsdf_size_x"), config.as<int>("tsdf_size_y"), config.as<int>("tsdf_size_z"));
    voxel_size = config.as<float>("tsdf_voxel_size");
    max_integration_weight = config.as<int>("max_integration_weight");
    const float thres_range = config.as<float>("thres_range");

    world2camera = Matrix4cf::Identity();
    // :22, the seed line the reference keeps commented: world2camera(0, 3).imag(H_) — here a parameter, as in the orchestrator
    const int seed_row = config.as<int>("csfd_seed_row", -1), seed_col = config.as<int>("csfd_seed_col", -1);
    if (seed_row >= 0 && seed_row < 4 && seed_col >= 0 && seed_col < 4) world2camera(seed_row, seed_col).imag(config.as<float>("csfd_seed_h", (float)H_));
    world2camera_record.clear();
    world2camera_record.reserve(10000);
    world2camera_record.push_back(world2camera);
    world2volume = Matrix4cf::Identity();
    const float r_x = config.as<float>("r_x") / 180.0f * float(M_PI), r_y = config.as<float>("r_y") / 180.0f * float(M_PI),
                r_z = config.as<float>("r_z") / 180.0f * float(M_PI);
    const Matrix3cf rotation = (angle_axis(hostComplex(r_x, 0.f), 0) * angle_axis(hostComplex(r_y, 0.f), 1)) * angle_axis(hostComplex(r_z, 0.f), 2);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) world2volume(i, j) = hostComplex(rotation(i, j).real(), 0.f);
    world2volume(0, 3) = hostComplex(config.as<float>("init_x"), 0.f);
    world2volume(1, 3) = hostComplex(config.as<float>("init_y"), 0.f);
    world2volume(2, 3) = hostComplex(config.as<float>("init_z"), 0.f);

    depth_width = config.as<int>("depth_width");
    depth_height = config.as<int>("depth_height");
    kinect_intrinsic = Intr(config.as<float>("fx"), config.as<float>("fy"), config.as<float>("cx"), config.as<float>("cy"));

    num_levels = config.as<int>("num_levels");
    if (num_levels > 3) {
        std::cout << "sorry, the max supported multi-level = 3" << "\n";
        num_levels = 3;
    }
    const int iters[] = {5, 4, 3};
    std::copy(iters, iters + num_levels, icp_iterations);
    distThres = config.as<float>("distThres");
    angleThres = float(sin(config.as<float>("angleThres") / 180.f * M_PI));
    biInterpolate_threshold = config.as<float>("biInterpolate_threshold");
    trunc_logistic_k = config.as<float>("trunc_logistic_k", 0.f);

    AllocateBuffers();
    tsdf_volume_d_ptr = new TsdfVolume(volume_resolution, voxel_size, thres_range);
    frame_id = 0;
    frame_step = config.as<int>("frame_step", 1);
}

// :75-106 (the buffers nothing reads — newTSDF_volume, jacobi_buf, hessian_buf — are left out)
void ReferenceCallShape::AllocateBuffers() {
    for (auto *maps : {&depths_curr_d, &vmaps_curr_d, &nmaps_curr_d, &vmaps_g_prev_d, &nmaps_g_prev_d}) maps->resize(num_levels);
    for (int i = 0; i < num_levels; ++i) {
        const int pyr_rows = depth_height >> i, pyr_cols = depth_width >> i;
        depths_curr_d[i].create(pyr_rows, pyr_cols);
        for (auto *maps : {&vmaps_curr_d, &nmaps_curr_d, &vmaps_g_prev_d, &nmaps_g_prev_d}) (*maps)[i].create(pyr_rows * 3, pyr_cols);
    }
    g_buf.create(27, 20 * 60);
    sum_buf.create(27);
}

// :147-159
int ReferenceCallShape::ProcessFrame(const DeviceArray2D<ushort> &depth_frame_d) {
    const int aligned = AlignDepthToReconstruction(depth_frame_d);
    if (frame_id > 0 && !aligned) {
        std::cout << "Frame align failed!" << std::endl;
        return 0;
    }
    IntegrateFrame(depth_frame_d);
    frame_id += frame_step;
    return 1;
}

// :280-299 with SmoothDepthFrame (:125-145) in place
void ReferenceCallShape::SurfaceMeasure(const DeviceArray2D<ushort> &depth_frame_d) {
    if (depth_frame_d.rows() <= 0 || depth_frame_d.cols() <= 0) {
        std::cout << "error: KinectFusionReconstruction::SmoothDepthFrame, input map is empty" << std::endl;
        return;
    }
    MapArr &smooth = depths_curr_d[0];
    if (smooth.rows() != depth_frame_d.rows() || smooth.cols() != depth_frame_d.cols()) {
        smooth.release();
        smooth.create(depth_frame_d.rows(), depth_frame_d.cols());
    }
    bilateralFilter(depth_frame_d, smooth);
    for (int i = 1; i < num_levels; ++i) pyrDown(depths_curr_d[i - 1], depths_curr_d[i]);
    for (int i = 0; i < num_levels; ++i) {
        createVMap(kinect_intrinsic(i), depths_curr_d[i], vmaps_curr_d[i]);
        createNMap(vmaps_curr_d[i], nmaps_curr_d[i]);
    }
}

// :161-175
int ReferenceCallShape::AlignDepthToReconstruction(const DeviceArray2D<ushort> &depth_frame_d) {
    SurfaceMeasure(depth_frame_d);
    const Matrix4cf c2w_prev = inverse(world2camera_record.back());
    const Matrix3frm Rprev = GetRotation(c2w_prev);
    const Vector3cf tprev = GetTranslation(c2w_prev);
    return PoseEstimate(Rprev, tprev, inverse(Rprev), tprev);
}

// :177-235 — one estimateCombined per iteration, each returning after the stream has drained with the 6x6 system on the host
int ReferenceCallShape::PoseEstimate(Matrix3frm Rcurr, Vector3cf tcurr, Matrix3frm Rprev_inv, Vector3cf tprev) {
    icp_log.clear();
    if (frame_id == 0) return 0;
    Matrix4cf c2w_curr = inverse(world2camera_record.back());
    const MatS33 &device_Rprev_inv = device_cast<MatS33>(Rprev_inv);
    const devComplex3 &device_tprev = device_cast<devComplex3>(tprev);
    for (int level_index = num_levels - 1; level_index >= 0; --level_index) {
        for (int iter = 0; iter < icp_iterations[level_index]; ++iter) {
            hostComplexICP A[36], b[6];
            estimateCombined(device_cast<MatS33>(Rcurr), device_cast<devComplex3>(tcurr), vmaps_curr_d[level_index], nmaps_curr_d[level_index],
                             device_Rprev_inv, device_tprev, kinect_intrinsic(level_index), vmaps_g_prev_d[level_index], nmaps_g_prev_d[level_index],
                             distThres, angleThres, g_buf, sum_buf, A, b);
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 7; ++j) {
                    const hostComplexICP v = j == 6 ? b[i] : A[i * 6 + j];
                    icp_log.push_back(v.real());
                    icp_log.push_back(v.imag());
                }
            const double det = real_determinant6(A);
            if (fabs(det) < 1e-15 || std::isnan(det)) {
                if (std::isnan(det)) std::cout << "qnan det" << std::endl;
                else std::cout << "eps det: " << fabs(det) << std::endl;
                return 0;
            }
            hostComplexICP sol[6];
            llt_solve6(A, b, sol);
            hostComplex result[6];
            for (int i = 0; i < 6; ++i) result[i] = hostComplex((float)sol[i].real(), (float)sol[i].imag());
            const Matrix3cf Rinc = (angle_axis(result[2], 2) * angle_axis(result[1], 1)) * angle_axis(result[0], 0);
            const Vector3cf rotated = Rinc * tcurr;
            for (int i = 0; i < 3; ++i) tcurr[i] = rotated[i] + result[3 + i];
            Rcurr = Rinc * Rcurr;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) c2w_curr(i, j) = Rcurr(i, j);
                c2w_curr(i, 3) = tcurr[i];
            }
            c2w_curr(3, 3) = hostComplex(1.f, 0.f);
        }
    }
    world2camera = inverse(c2w_curr);
    world2camera_record.push_back(world2camera);
    return 1;
}

// :237-278 — integrateTsdfVolume drains the stream, the raycast does not, every resize does
int ReferenceCallShape::IntegrateFrame(const DeviceArray2D<ushort> &depth_frame_d) {
    const Matrix4cf c2w = inverse(world2camera_record.back());
    const Matrix4cf c2v = world2volume * c2w;
    const Matrix4cf v2c = inverse(c2v);
    Vector3cf tc2v = GetTranslation(c2v), tv2c = GetTranslation(v2c);
    Matrix3frm Rv2c = GetRotation(v2c);
    int3 volume_res;
    volume_res.x = volume_resolution.x(); volume_res.y = volume_resolution.y(); volume_res.z = volume_resolution.z();
    integrateTsdfVolume(depth_frame_d, kinect_intrinsic, max_integration_weight, volume_res, voxel_size, device_cast<MatS33>(Rv2c),
                        device_cast<devComplex3>(tv2c), device_cast<devComplex3>(tc2v), tsdf_volume_d_ptr->getTsdfTruncDist(),
                        tsdf_volume_d_ptr->value(), tsdf_volume_d_ptr->weight(), tsdf_volume_d_ptr->grad(), depthRawScaled_d, frame_id,
                        biInterpolate_threshold, trunc_logistic_k);
    CalculatePointCloud(vmaps_g_prev_d[0], nmaps_g_prev_d[0]);
    for (int i = 1; i < num_levels; ++i) {
        resizeVMap(vmaps_g_prev_d[i - 1], vmaps_g_prev_d[i]);
        resizeNMap(nmaps_g_prev_d[i - 1], nmaps_g_prev_d[i]);
    }
    return 1;
}

// :302-332
void ReferenceCallShape::CalculatePointCloud(MapArr &xyz_g_d, MapArr &normal_g_d) {
    const Matrix4cf c2v = world2volume * inverse(world2camera);
    const Matrix4cf v2w = inverse(world2volume);
    Matrix3frm Rc2v = GetRotation(c2v), Rv2w = GetRotation(v2w);
    Vector3cf tc2v = GetTranslation(c2v), tv2w = GetTranslation(v2w);
    int3 volume_res;
    volume_res.x = volume_resolution.x(); volume_res.y = volume_resolution.y(); volume_res.z = volume_resolution.z();
    raycast(kinect_intrinsic, device_cast<MatS33>(Rc2v), device_cast<devComplex3>(tc2v), device_cast<MatS33>(Rv2w), device_cast<devComplex3>(tv2w),
            tsdf_volume_d_ptr->getTsdfTruncDist(), volume_res, voxel_size, tsdf_volume_d_ptr->value(), tsdf_volume_d_ptr->grad(), xyz_g_d, normal_g_d);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// C ABI (include/xslam_amd_pipeline.h, "reference call shape")
typedef ReferenceCallShape RS;
extern "C" {

void *xs_refshape_create(const char *yaml_text) {
    try {
        RS *k = new RS();
        k->SetYamlParameters(FlatYaml::Load(yaml_text ? yaml_text : ""));
        return k;
    } catch (const std::exception &e) {
        printf("xs_refshape_create: %s\n", e.what());
        return nullptr;
    }
}
void xs_refshape_destroy(void *h) { delete (RS *)h; }
// main.cpp:50-58: DeviceArray2D<ushort>::upload of the host frame, then ProcessFrame
int xs_refshape_process_frame_host(void *h, const uint16_t *depth_host) {
    RS *k = (RS *)h;
    DeviceArray2D<ushort> depth_frame_d;
    depth_frame_d.upload(depth_host, k->depth_width * sizeof(ushort), k->depth_height, k->depth_width);
    return k->ProcessFrame(depth_frame_d);
}
// a frame that is already resident (the benchmark's timed region is main.cpp:57-60's: ProcessFrame alone)
int xs_refshape_process_frame(void *h, const uint16_t *depth_dev, size_t step_bytes) {
    RS *k = (RS *)h;
    DeviceArray2D<ushort> view(k->depth_height, k->depth_width, (void *)depth_dev, step_bytes);
    return k->ProcessFrame(view);
}
int xs_refshape_frame_id(void *h) { return ((RS *)h)->frame_id; }
void xs_refshape_get_world2camera(void *h, int idx, float *out32) {
    RS *k = (RS *)h;
    if (idx < 0) idx += (int)k->world2camera_record.size();
    std::memcpy(out32, &k->world2camera_record[idx], 32 * sizeof(float));
}
int xs_refshape_download_volume(void *h, float *value, int *weight, float *grad) {
    RS *k = (RS *)h;
    const int X = k->volume_resolution[0];
    if (value) k->tsdf_volume_d_ptr->value().download(value, X * sizeof(float));
    if (weight) k->tsdf_volume_d_ptr->weight().download(weight, X * sizeof(int));
    if (grad) k->tsdf_volume_d_ptr->grad().download(grad, X * sizeof(float));
    return 0;
}
int xs_refshape_download_map(void *h, int which, int level, float *out) {
    RS *k = (RS *)h;
    if (level < 0 || level >= k->num_levels || which < 0 || which > 4) return -1;
    std::vector<MapArr> *sets[5] = {&k->depths_curr_d, &k->vmaps_curr_d, &k->nmaps_curr_d, &k->vmaps_g_prev_d, &k->nmaps_g_prev_d};
    MapArr &m = (*sets[which])[level];
    m.download(out, m.cols() * sizeof(devComplex));
    return 0;
}
int xs_refshape_icp_log(void *h, double *out, int capacity) {
    RS *k = (RS *)h;
    const int n = (int)k->icp_log.size();
    if (out && capacity >= n && n) std::memcpy(out, k->icp_log.data(), n * sizeof(double));
    return n;
}

// ComputeLocalTsdf_hessian / ComputeLocalTsdf_loss through their TsdfFusion.h:48-60 signatures: depth u16 on the device, gt a dense device
// array of X*Y*Z floats; with_volumes != 0 also fills the per-voxel scratch volumes the reference's thrust vectors hold (returned in
// volumes_host: real | grad | hessian | count-as-float, 4 x N^3, or real | count for the loss) — NULL skips the download.
int xs_refshape_hessian(const uint16_t *depth_dev, size_t depth_step, int rows, int cols, const float *intr4, const int *res3, float voxel_size,
                        const float *Rv2c36, const float *tv2c12, float tranc_dist, const float *gt_dev, int with_volumes, float *out4,
                        float *volumes_host) {
    const size_t n = (size_t)res3[0] * res3[1] * res3[2];
    DeviceArray2D<float> depthScaled;
    DeviceArray<float> gt_vec(const_cast<float *>(gt_dev), n), real_vec, grad_vec, hessian_vec;
    DeviceArray<int> count_vec;
    if (with_volumes) {   // thrust's resize(voxel_num, 0) of a fresh vector (TsdfFusion.cu:293-300): the kernel writes only the voxels it counts
        real_vec.create(n); grad_vec.create(n); hessian_vec.create(n); count_vec.create(n);
        for (void *p : {(void *)real_vec.ptr(), (void *)grad_vec.ptr(), (void *)hessian_vec.ptr(), (void *)count_vec.ptr()})
            hipSafeCall(hipMemsetAsync(p, 0, n * 4, current_stream()));
    }
    MatD33 R; devDComplex3 t;
    std::memcpy(&R, Rv2c36, sizeof(R));
    std::memcpy(&t, tv2c12, sizeof(t));
    int3 volume_res; volume_res.x = res3[0]; volume_res.y = res3[1]; volume_res.z = res3[2];
    const float4 r = ComputeLocalTsdf_hessian(PtrStepSz<ushort>(rows, cols, const_cast<ushort *>(depth_dev), depth_step), Intr(intr4[0], intr4[1], intr4[2], intr4[3]),
                                              depthScaled, volume_res, voxel_size, R, t, tranc_dist, 0.f, 0.f, gt_vec, real_vec, grad_vec, hessian_vec, count_vec);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
    if (with_volumes && volumes_host) {
        real_vec.download(volumes_host); grad_vec.download(volumes_host + n); hessian_vec.download(volumes_host + 2 * n);
        count_vec.download(reinterpret_cast<int *>(volumes_host + 3 * n));
    }
    return 0;
}
int xs_refshape_loss(const uint16_t *depth_dev, size_t depth_step, int rows, int cols, const float *intr4, const int *res3, float voxel_size,
                     const float *Rv2c9, const float *tv2c3, float tranc_dist, const float *gt_dev, int with_volumes, float *out2, float *volumes_host) {
    const size_t n = (size_t)res3[0] * res3[1] * res3[2];
    DeviceArray2D<float> depthScaled;
    DeviceArray<float> gt_vec(const_cast<float *>(gt_dev), n), real_vec;
    DeviceArray<int> count_vec;
    if (with_volumes) {
        real_vec.create(n); count_vec.create(n);
        hipSafeCall(hipMemsetAsync(real_vec.ptr(), 0, n * 4, current_stream()));
        hipSafeCall(hipMemsetAsync(count_vec.ptr(), 0, n * 4, current_stream()));
    }
    Mat33 R;
    std::memcpy(&R, Rv2c9, sizeof(R));
    float3 t; t.x = tv2c3[0]; t.y = tv2c3[1]; t.z = tv2c3[2];
    int3 volume_res; volume_res.x = res3[0]; volume_res.y = res3[1]; volume_res.z = res3[2];
    const float2 r = ComputeLocalTsdf_loss(PtrStepSz<ushort>(rows, cols, const_cast<ushort *>(depth_dev), depth_step), Intr(intr4[0], intr4[1], intr4[2], intr4[3]),
                                           depthScaled, volume_res, voxel_size, R, t, tranc_dist, 0.f, 0.f, gt_vec, real_vec, count_vec);
    out2[0] = r.x; out2[1] = r.y;
    if (with_volumes && volumes_host) { real_vec.download(volumes_host); count_vec.download(reinterpret_cast<int *>(volumes_host + n)); }
    return 0;
}

}  // extern "C"
