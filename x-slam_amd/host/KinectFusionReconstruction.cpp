// KinectFusionReconstruction.cpp — per-frame pipeline on one MI355X; follows the control flow
// of XKinectFusion/src/KinectFusionReconstruction.cpp:9-332 (cited per method) over the
// launcher shim.  One stream, no per-frame allocation, synchronisation only where the host
// needs a result (the 6x6 normal equations of each ICP iteration).
#include "KinectFusionReconstruction.h"
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

using namespace xs_host;

KinectFusionReconstruction::KinectFusionReconstruction() {
    depth_width = 0;
    depth_height = 0;
    hipSafeCall(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking));
    hipSafeCall(hipEventCreateWithFlags(&surface_done_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&integrate_done_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&scale_done_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&classify_done_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&tail_done_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&surface_done_next_, hipEventDisableTiming));
    hipSafeCall(hipEventCreateWithFlags(&scale_done_next_, hipEventDisableTiming));
    hipSafeCall(hipHostMalloc((void **)&pinned_counters_, COUNTER_RING * 2 * sizeof(unsigned long long)));
    hipSafeCall(hipHostMalloc((void **)&pinned_sums_, PINNED_DOUBLES * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < PINNED_DOUBLES; ++i) pinned_sums_[i] = 0.0;
    hipSafeCall(hipHostMalloc((void **)&pinned_pairs_, XS_ICP_PAIRS_BYTES, hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(pinned_pairs_, 0, XS_ICP_PAIRS_BYTES);
    hipSafeCall(hipHostMalloc((void **)&pinned_records_, xs_icp_records_bytes(), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(pinned_records_, 0, xs_icp_records_bytes());
}

void KinectFusionReconstruction::SetSharding(int rank, int count, collective_fn fn, void *user) {
    shard_rank = rank; shard_count = count < 1 ? 1 : count; collective = fn; collective_user = user;
}

KinectFusionReconstruction::~KinectFusionReconstruction() {
    if (aux_stream_) { (void)hipStreamSynchronize(aux_stream_); (void)hipStreamDestroy(aux_stream_); }
    if (surface_done_) (void)hipEventDestroy(surface_done_);
    if (integrate_done_) (void)hipEventDestroy(integrate_done_);
    if (tail_done_) (void)hipEventDestroy(tail_done_);
    if (surface_done_next_) (void)hipEventDestroy(surface_done_next_);
    if (scale_done_next_) (void)hipEventDestroy(scale_done_next_);
    if (scale_done_) (void)hipEventDestroy(scale_done_);
    if (classify_done_) (void)hipEventDestroy(classify_done_);
    if (pinned_counters_) (void)hipHostFree(pinned_counters_);
    if (gather_counts_host_) (void)hipHostFree(gather_counts_host_);
    if (pinned_sums_) (void)hipHostFree(pinned_sums_);
    if (pinned_pairs_) (void)hipHostFree(pinned_pairs_);
    if (pinned_records_) (void)hipHostFree(pinned_records_);
    if (icp_mailbox_) (void)xs_icp_mailbox_free(icp_mailbox_, icp_mailbox_in_device_);
    if (integrate_mailbox_) (void)xs_icp_mailbox_free(integrate_mailbox_, integrate_mailbox_in_device_);
    if (gn_mailbox_) (void)xs_icp_mailbox_free(gn_mailbox_, gn_mailbox_in_device_);
    if (gn_publish_) (void)hipHostFree(gn_publish_);
    for (int i = 0; i < 2; ++i) {
        if (ingest_pinned_[i]) { (void)hipEventSynchronize(ingest_done_[i]); (void)hipHostFree(ingest_pinned_[i]); (void)hipEventDestroy(ingest_done_[i]); }
    }
    if (tsdf_volume_d_ptr) ReleaseBuffers();
    for (auto &slot : prof_ring_)
        for (int s = 0; s < ST_COUNT; ++s) { (void)hipEventDestroy(slot.ev[s][0]); (void)hipEventDestroy(slot.ev[s][1]); }
}

// reference :9-73
void KinectFusionReconstruction::SetYamlParameters(const FlatYaml &config_) {
    this->config = config_;
    const int resolutionX = config.as<int>("tsdf_size_x");
    const int resolutionY = config.as<int>("tsdf_size_y");
    const int resolutionZ = config.as<int>("tsdf_size_z");
    volume_resolution = Vector3i(resolutionX, resolutionY, resolutionZ);
    voxel_size = config.as<float>("tsdf_voxel_size");
    max_integration_weight = config.as<int>("max_integration_weight");
    const float thres_range = config.as<float>("thres_range");

    csfd_seed_row = config.as<int>("csfd_seed_row", -1);
    csfd_seed_col = config.as<int>("csfd_seed_col", -1);
    csfd_seed_h = config.as<float>("csfd_seed_h", (float)H_);

    world2camera = Matrix4cf::Identity();
    if (csfd_seed_row >= 0 && csfd_seed_row < 4 && csfd_seed_col >= 0 && csfd_seed_col < 4)
        world2camera(csfd_seed_row, csfd_seed_col).imag(csfd_seed_h);  // the line the reference leaves commented (:22)
    world2camera_record.clear();
    world2camera_record.reserve(10000);
    world2camera_record.push_back(world2camera);
    world2volume = Matrix4cf::Identity();
    const float init_x = config.as<float>("init_x"), init_y = config.as<float>("init_y"), init_z = config.as<float>("init_z");
    const float r_x = config.as<float>("r_x") / 180.0f * float(M_PI);
    const float r_y = config.as<float>("r_y") / 180.0f * float(M_PI);
    const float r_z = config.as<float>("r_z") / 180.0f * float(M_PI);
    // Rx * Ry * Rz as rotation matrices (the reference multiplies real AngleAxisf objects, i.e.
    // quaternions; identical for the shipped r_x = r_y = r_z = 0)
    const Matrix3cf rotation = (angle_axis(hostComplex(r_x, 0.f), 0) * angle_axis(hostComplex(r_y, 0.f), 1)) * angle_axis(hostComplex(r_z, 0.f), 2);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) world2volume(i, j) = hostComplex(rotation(i, j).real(), 0.f);
    world2volume(0, 3) = hostComplex(init_x, 0.f);
    world2volume(1, 3) = hostComplex(init_y, 0.f);
    world2volume(2, 3) = hostComplex(init_z, 0.f);

    depth_width = config.as<int>("depth_width");
    depth_height = config.as<int>("depth_height");
    kinect_intrinsic.fx = config.as<float>("fx");
    kinect_intrinsic.fy = config.as<float>("fy");
    kinect_intrinsic.cx = config.as<float>("cx");
    kinect_intrinsic.cy = config.as<float>("cy");

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

    // planes owned / stored by this rank (whole volume when not sharded)
    zo0 = (int)((long long)resolutionZ * shard_rank / shard_count);
    zo1 = (int)((long long)resolutionZ * (shard_rank + 1) / shard_count);
    zs0 = shard_count > 1 ? std::max(0, zo0 - HALO) : 0;
    zs1 = shard_count > 1 ? std::min(resolutionZ, zo1 + HALO) : resolutionZ;
    icp_solve_on_device = config.as<bool>("icp_solve_on_device", false);
    icp_shard_rows = config.as<bool>("icp_shard_rows", false);
    icp_post_pose = config.as<bool>("icp_post_pose", true);
    icp_real_current_maps = config.as<bool>("icp_real_current_maps", true);
    integrate_classify_ahead = config.as<bool>("integrate_classify_ahead", true);
    integrate_classify_beside_icp = config.as<bool>("integrate_classify_beside_icp", false);
    integrate_classify_early = std::max(0, config.as<int>("integrate_classify_early", 0));
    integrate_classify_predicted = config.as<bool>("integrate_classify_predicted", false);
    integrate_classify_slack = std::max(1.0f, config.as<float>("integrate_classify_slack", 2.0f));
    integrate_post_pose = config.as<bool>("integrate_post_pose", false);
    integrate_post_early = config.as<bool>("integrate_post_early", false);
    raycast_sign_map = config.as<bool>("raycast_sign_map", true);
    raycast_sign_map_shift = std::min(6, std::max(0, config.as<int>("raycast_sign_map_shift", 0)));   // 0: the finest usable one
    icp_lookahead = config.as<int>("icp_lookahead", 1);
    icp_host_fold = config.as<bool>("icp_host_fold", false);
    icp_publish_pairs = config.as<bool>("icp_publish_pairs", true);
    force_shard_composite = config.as<bool>("force_shard_composite", false);
    shard_composite_gather = config.as<bool>("shard_composite_gather", true);
    raycast_builds_pyramid = config.as<bool>("raycast_builds_pyramid", true);
    gn_post_pose = config.as<bool>("gn_post_pose", true);
    gn_publish_sharded = config.as<bool>("gn_publish_sharded", true);
    profile_integrate_every = std::max(1, config.as<int>("profile_integrate_every", 4));
    AllocateBuffers();
    tsdf_volume_d_ptr = new TsdfVolume(Vector3i(resolutionX, resolutionY, zs1 - zs0), voxel_size, thres_range);
    if (sign_map_on()) {
        // the sign map of the ray march (include/xslam_amd.h), with the finest bricks the march can use for this camera and volume — or none
        const int finest = xs_raycast_signmap_shift(&kinect_intrinsic.fx, voxel_size, tsdf_volume_d_ptr->getTsdfTruncDist());
        raycast_sign_map_shift = finest ? std::max(raycast_sign_map_shift, finest) : 0;   // (a finer one than the march can use is coarsened)
        const int res[3] = {volume_resolution[0], volume_resolution[1], volume_resolution[2]};
        const size_t bytes = raycast_sign_map_shift ? xs_signmap_bytes(res, raycast_sign_map_shift) : 0;
        if (bytes) { sign_map_.create(bytes); RebuildSignMap(); }   // (from the volume as it is: empty)
        else raycast_sign_map = false;
    }

    use_gtPose = config.as<bool>("flag_use_gtPose", false);
    gt_poses.resize(0);
    frame_id = 0;
    frame_step = config.as<int>("frame_step", 1);
}

// reference :75-106 (the unused newTSDF_volume, jacobi and hessian buffers are not allocated)
void KinectFusionReconstruction::AllocateBuffers() {
    depths_curr_d.resize(num_levels);
    vmaps_curr_d.resize(num_levels);
    nmaps_curr_d.resize(num_levels);
    vmaps_g_prev_d.resize(num_levels);
    nmaps_g_prev_d.resize(num_levels);
    for (int i = 0; i < num_levels; ++i) {
        const int pyr_rows = depth_height >> i, pyr_cols = depth_width >> i;
        depths_curr_d[i].create(pyr_rows, pyr_cols);
        vmaps_curr_d[i].create(pyr_rows * 3, pyr_cols);
        nmaps_curr_d[i].create(pyr_rows * 3, pyr_cols);
        if (i == 0) {
            // the level-0 model maps share one allocation, vertex planes then normal planes: the sharded
            // raycast composite adds both with a single all-reduce
            const size_t step = ((size_t)pyr_cols * sizeof(devComplex) + 255) / 256 * 256;
            maps_prev0_block_.create(2 * (size_t)pyr_rows * 3 * step);
            vmaps_g_prev_d[0] = MapArr(pyr_rows * 3, pyr_cols, maps_prev0_block_.ptr(), step);
            nmaps_g_prev_d[0] = MapArr(pyr_rows * 3, pyr_cols, maps_prev0_block_.ptr() + (size_t)pyr_rows * 3 * step, step);
        } else {
            vmaps_g_prev_d[i].create(pyr_rows * 3, pyr_cols);
            nmaps_g_prev_d[i].create(pyr_rows * 3, pyr_cols);
        }
    }
    depthRawScaled_d.create(depth_height, depth_width);
    // the constant the raycast march divides by, checked once (all 2^32 operands, ~2 ms): it then takes floor(p / voxel_size) with the
    // short division; a constant that fails its check keeps the bracketed reciprocals + divide, same results
    xs_const_div_prepare(voxel_size);
    {
        const int res[3] = {volume_resolution[0], volume_resolution[1], volume_resolution[2]};
        integrate_ws_.create(xs_integrate_workspace_bytes(res, zs1 - zs0));
        icp_ws_.create(xs_icp_workspace_bytes());
        check_rc(xs_icp_workspace_init(icp_ws_.ptr(), current_stream()), "icp workspace");
        if (!icp_mailbox_) check_rc(xs_icp_mailbox_alloc(&icp_mailbox_, &icp_mailbox_in_device_), "icp mailbox");
        if (!integrate_mailbox_) check_rc(xs_icp_mailbox_alloc(&integrate_mailbox_, &integrate_mailbox_in_device_), "integrate mailbox");
        posted_pose_.create(32);
        icp_sums_.create(64);
        icp_pose_.create(xs_icp_pose_state_bytes());
        ray_ws_.create((size_t)depth_width * depth_height);
        if (shard_count > 1 || force_shard_composite) {
            ray_keys_.create((size_t)depth_width * depth_height);
            ray_min_keys_.create((size_t)depth_width * depth_height);
        }
    }
    counters_.create(2 * COUNTER_RING);
    hipSafeCall(hipMemsetAsync(counters_.ptr(), 0, 2 * COUNTER_RING * sizeof(unsigned long long), current_stream()));
    depth_max_.create(4);
    depth_tiles_.create(xs_depth_tiles_bytes(depth_height, depth_width));
}

// reference :108-123
void KinectFusionReconstruction::ReleaseBuffers() {
    for (int i = 0; i < (int)depths_curr_d.size(); ++i) {
        depths_curr_d[i].release();
        vmaps_curr_d[i].release();
        nmaps_curr_d[i].release();
        vmaps_g_prev_d[i].release();
        nmaps_g_prev_d[i].release();
    }
    maps_prev0_block_.release();
    g_buf.release();
    sum_buf.release();
    depthRawScaled_d.release();
    for (auto &m : vreal_curr_d) m.release();
    for (auto &m : nreal_curr_d) m.release();
    real_maps_valid_ = false;
    delete tsdf_volume_d_ptr;
    tsdf_volume_d_ptr = nullptr;
}

// reference :125-145
int KinectFusionReconstruction::SmoothDepthFrame(MapArr &dst_d, const DeviceArray2D<ushort> &src_d) {
    if (src_d.rows() <= 0 || src_d.cols() <= 0) {
        std::cout << "error: KinectFusionReconstruction::SmoothDepthFrame, input map is empty" << std::endl;
        return 0;
    }
    if (dst_d.rows() != src_d.rows() || dst_d.cols() != src_d.cols()) {
        dst_d.release();
        dst_d.create(src_d.rows(), src_d.cols());
    }
    bilateralFilter(src_d, dst_d);
    return 1;
}

// reference :147-159.  On a failed alignment the reference returns without advancing frame_id,
// and its demo retries the same frame forever; the status is returned
// the same way here and the caller decides.
int KinectFusionReconstruction::ProcessFrame(const DeviceArray2D<ushort> &depth_frame_d) {
    const int align_return = AlignDepthToReconstruction(depth_frame_d, false);
    if (frame_id > 0 && !align_return) {
        std::cout << "Frame align failed!" << std::endl;
        return 0;
    }
    IntegrateFrame(depth_frame_d);
    frame_id += frame_step;
    if (profiling) end_profiled_frame();
    return 1;
}

// Depth ingest (main.cpp:50-58: imread -> upload -> ProcessFrame).  The frame goes through one of two
// host-pinned staging buffers and an asynchronous copy on the second stream — the stream the map
// preparation and the depth scaling run on, so they follow the copy with no host wait, and all of it
// runs under the previous frame's tail.  A caller that decodes straight into IngestBuffer() (or passes
// any pinned pointer) skips the staging copy.
ushort *KinectFusionReconstruction::IngestBuffer() {
    const int slot = ingest_seq_ & 1;
    const size_t bytes = (size_t)depth_width * depth_height * sizeof(ushort);
    if (!ingest_pinned_[slot]) {
        hipSafeCall(hipHostMalloc((void **)&ingest_pinned_[slot], bytes));
        hipSafeCall(hipEventCreateWithFlags(&ingest_done_[slot], hipEventDisableTiming));
    } else
        hipSafeCall(hipEventSynchronize(ingest_done_[slot]));  // the copy out of this slot two frames ago
    return ingest_pinned_[slot];
}
int KinectFusionReconstruction::ProcessFrameHost(const ushort *depth_host) {
    if (depth_width <= 0 || depth_height <= 0 || !depth_host) return 0;
    const size_t row = (size_t)depth_width * sizeof(ushort);
    const ushort *src = depth_host;
    int slot = -1;
    if (depth_host == ingest_pinned_[0] || depth_host == ingest_pinned_[1]) slot = depth_host == ingest_pinned_[0] ? 0 : 1;
    else {
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, depth_host) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!pinned) {
            (void)hipGetLastError();  // a pageable pointer is reported as an error: not one
            ushort *stage = IngestBuffer();
            slot = ingest_seq_ & 1;
            std::memcpy(stage, depth_host, row * depth_height);
            src = stage;
        }
    }
    depth_ingest_d_.create(depth_height, depth_width);
    hipSafeCall(hipMemcpy2DAsync(depth_ingest_d_.ptr(), depth_ingest_d_.step(), src, row, row, depth_height, hipMemcpyHostToDevice, aux_stream_));
    if (slot >= 0) {
        hipSafeCall(hipEventRecord(ingest_done_[slot], aux_stream_));
        if (slot == (ingest_seq_ & 1)) ++ingest_seq_;
    }
    return ProcessFrame(depth_ingest_d_);
}

// reference :161-175
int KinectFusionReconstruction::AlignDepthToReconstruction(const DeviceArray2D<ushort> &depth_frame_d, bool /*use_LM*/) {
    SurfaceMeasure(depth_frame_d);
    struct AtExit { KinectFusionReconstruction *k; ~AtExit() { k->EnqueueAnnouncedFrame(true); } } announced_at_exit{this};   // (whatever path is taken below)
    if (use_gtPose) return 1;
    Matrix4cf c2w_prev = inverse(world2camera_record.back());
    Matrix3frm Rprev = GetRotation(c2w_prev);
    Vector3cf tprev = GetTranslation(c2w_prev);
    Matrix3frm Rprev_inv = inverse(Rprev);
    Matrix3frm Rcurr = Rprev;
    Vector3cf tcurr = tprev;
    return PoseEstimate(Rcurr, tcurr, Rprev_inv, tprev);
}

// reference :177-235
int KinectFusionReconstruction::PoseEstimate(Matrix3frm Rcurr, Vector3cf tcurr, Matrix3frm Rprev_inv, Vector3cf tprev) {
    icp_log.clear();
    if (!list_predicted_) list_ready_ = false;   // (a list classified for the predicted pose, on the auxiliary stream, stays: SurfaceMeasure)
    if (frame_id == 0) return 0;
    Matrix4cf c2w_prev = inverse(world2camera_record.back());
    Matrix4cf c2w_curr = c2w_prev;
    auto &device_Rprev_inv = device_cast<MatS33>(Rprev_inv);
    auto &device_tprev = device_cast<devComplex3>(tprev);
    int total_iters = 0;
    for (int l = 0; l < num_levels; ++l) total_iters += icp_iterations[l];
    const bool icp_local = shard_count == 1 || !icp_shard_rows;  // this rank evaluates every pixel row itself
    if (icp_solve_on_device && icp_local && !profiling_icp_sync && total_iters >= 1 && total_iters <= ICP_LOG_MAX)
        return PoseEstimateOnDevice(Rcurr, tcurr, Rprev_inv, tprev, c2w_curr, total_iters);
    // posted mode: launch n + 1 is enqueued before the host waits for launch n, and gets its pose through
    // the mailbox after the solve (xs_icp_accumulate_posted)
    // (only with the mailbox in device memory behind a large BAR: 512 workgroups polling pinned host memory
    // over PCIe cost more than the launch latency they would save)
    const bool posted = icp_post_pose && icp_local && !profiling_icp_sync && total_iters >= 1 && icp_mailbox_ && icp_mailbox_in_device_;
    std::vector<int> level_of;
    for (int level_index = num_levels - 1; level_index >= 0; --level_index)
        for (int iter = 0; iter < icp_iterations[level_index]; ++iter) level_of.push_back(level_index);
    void *mailbox = icp_mailbox_;
    auto launch_local = [&](int level, const MatS33 *R, const devComplex3 *t, unsigned mail_seq) {
        MapArr &vc = vmaps_curr_d[level], &nc = nmaps_curr_d[level], &vp = vmaps_g_prev_d[level], &np_ = nmaps_g_prev_d[level];
        const Intr k = kinect_intrinsic(level);
        const int rows = vc.rows() / 3, cols = vc.cols();
        unsigned long long *flag = icp_publish_pairs ? XS_ICP_PUBLISH_PAIRS : reinterpret_cast<unsigned long long *>(pinned_sums_ + 56);
        double *sums_out = icp_publish_pairs ? reinterpret_cast<double *>(pinned_pairs_) : pinned_sums_;
        const unsigned long long seq = ++icp_seq_;
        if (icp_host_fold) {
            check_rc(xs_icp_accumulate_records(R ? &R->data[0].x.re : nullptr, R ? &t->x.re : nullptr, R ? nullptr : mailbox, mail_seq, &vc.ptr()->re,
                                               &nc.ptr()->re, &device_Rprev_inv.data[0].x.re, &device_tprev.x.re, &k.fx, &vp.ptr()->re, &np_.ptr()->re,
                                               vc.step(), rows, cols, distThres, angleThres, 0, rows, pinned_records_, seq, current_stream()),
                     "estimateCombined (records)");
            return seq;
        }
        const bool real = real_maps_valid_ && level < (int)vreal_curr_d.size();
        if (R && real)
            check_rc(xs_icp_accumulate_real(&R->data[0].x.re, &t->x.re, &vc.ptr()->re, &nc.ptr()->re, vreal_curr_d[level].ptr(), nreal_curr_d[level].ptr(),
                                            vreal_curr_d[level].step(), &device_Rprev_inv.data[0].x.re, &device_tprev.x.re, &k.fx, &vp.ptr()->re,
                                            &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres, 0, rows, icp_ws_.ptr(), sums_out, flag, seq,
                                            current_stream()), "estimateCombined");
        else if (R)
            check_rc(xs_icp_accumulate(&R->data[0].x.re, &t->x.re, &vc.ptr()->re, &nc.ptr()->re, &device_Rprev_inv.data[0].x.re,
                                       &device_tprev.x.re, &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres,
                                       0, rows, icp_ws_.ptr(), sums_out, flag, seq, current_stream()), "estimateCombined");
        else if (real)
            check_rc(xs_icp_accumulate_posted_real(mailbox, mail_seq, &vc.ptr()->re, &nc.ptr()->re, vreal_curr_d[level].ptr(), nreal_curr_d[level].ptr(),
                                                   vreal_curr_d[level].step(), &device_Rprev_inv.data[0].x.re, &device_tprev.x.re, &k.fx,
                                                   &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres, 0, rows, icp_ws_.ptr(),
                                                   sums_out, flag, seq, current_stream()), "estimateCombined (posted)");
        else
            check_rc(xs_icp_accumulate_posted(mailbox, mail_seq, &vc.ptr()->re, &nc.ptr()->re, &device_Rprev_inv.data[0].x.re,
                                              &device_tprev.x.re, &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres,
                                              angleThres, 0, rows, icp_ws_.ptr(), sums_out, flag, seq, current_stream()),
                     "estimateCombined (posted)");
        return seq;
    };
    stage_begin(ST_ICP);
    // posted mode: which launches are in the queue, their completion and mailbox sequence numbers
    std::vector<unsigned long long> seq_of(total_iters, 0);
    std::vector<unsigned> mail_of(total_iters, 0);
    int enqueued = 0;
    auto enqueue_through = [&](int last, MatS33 *R0, devComplex3 *t0) {
        for (; enqueued < total_iters && enqueued <= last; ++enqueued) {
            if (enqueued == 0) { seq_of[0] = launch_local(level_of[0], R0, t0, 0); continue; }
            // the mailbox sequence number is the low word of the launch's completion number; a poller accepts any post at or after its
            // own number in the modulo-2^32 order, so the numbers must stay monotone there: 0 (the mailbox's initial word) is skipped
            // by spending one completion number, not replaced by an out-of-order value
            if ((unsigned)(icp_seq_ + 1) == 0u) ++icp_seq_;
            const unsigned m = (unsigned)(icp_seq_ + 1);
            mail_of[enqueued] = m;
            seq_of[enqueued] = launch_local(level_of[enqueued], nullptr, nullptr, m);
        }
    };
    auto t_prev = std::chrono::steady_clock::now();
    for (int n = 0; n < total_iters; ++n) {
        {
            const int level_index = level_of[n];
            auto &device_Rcurr = device_cast<MatS33>(Rcurr);
            auto &device_tcurr = device_cast<devComplex3>(tcurr);
            hostComplexICP A[36], b[6];
            long long inliers = 0;
            bool next_enqueued = false;
            unsigned next_mail_seq = 0, last_mail_seq = 0;
            if (posted) {
                // launches n + 1 .. n + icp_lookahead are in the queue before the host waits for launch n: each becomes resident the
                // moment its predecessor retires and polls the mailbox for its own sequence number.  (Default 1.  A kernel trace
                // shows ~4 us between the 8 us launches of levels 2 and 1, but keeping 2, 3, 5 or all 11 launches ahead — the
                // frame's first ones are enqueued while the GPU still works on the previous frame's raycast — measured the same
                // 2 900-2 950 frames/s as one: those gaps are the profiler's.)
                enqueue_through(n + std::max(1, icp_lookahead), &device_Rcurr, &device_tcurr);
                EnqueueAnnouncedFrame(false);   // (one stage of the announced next frame's map preparation per iteration, behind this frame's ICP launches)
                // the last launch is in the queue: the integrate call's brick classification goes in behind it, for the pose that
                // launch starts from — the final one differs by the last level-0 update, which IntegrateFrame checks is covered
                // (with integrate_post_pose the integrate kernel follows the classification into the queue, gated on its mailbox: the host's three
                // launches — classification, gate, integrate — go in while the last, 18 us, ICP launch runs; integrate_post_early puts them in one
                // iteration earlier, with planes from a pose two updates old: 16 % of those were not covered at slack 2)
                // (integrate_classify_early = k: k iterations before the last, i.e. for a pose k + 1 updates old — on the auxiliary stream the
                // classification is then finished long before the final pose is, and the integrate launch needs no wait packet)
                if ((integrate_post_pose && integrate_post_early ? enqueued == total_iters : n == std::max(0, total_iters - 1 - integrate_classify_early)) && !list_ready_ && integrate_classify_ahead && integrate_split())
                    ClassifyAhead(Rcurr, tcurr);
                const unsigned long long seq = seq_of[n];
                if (n + 1 < total_iters) {
                    next_mail_seq = mail_of[n + 1];
                    last_mail_seq = mail_of[enqueued - 1];   // an abandon command with this number releases every launch in the queue
                    next_enqueued = true;
                }
                volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(pinned_sums_ + 56);
                auto launch_gave_up = [&]() {   // never expected: the launch gave up on its pose
                    if (next_enqueued) xs_icp_post_pose(mailbox, nullptr, nullptr, last_mail_seq, 1);
                    if (posted_pending_) {   // the integrate launch gated on this frame's pose leaves too — before the stream is drained, or the drain waits out its gate
                        xs_icp_post_pose(integrate_mailbox_, nullptr, nullptr, posted_seq_, 1);
                        posted_pending_ = false;
                    }
                    hipSafeCall(hipStreamSynchronize(current_stream()));
                    check_rc(xs_icp_workspace_init(icp_ws_.ptr(), current_stream()), "icp workspace");
                    stage_end(ST_ICP);
                    AbandonClassifiedList();
                    std::cout << "error::KinectFusionReconstruction, ICP launch timed out waiting for its pose" << std::endl;
                    return 0;
                };
                if (icp_host_fold) {
                    // the workgroups' records arrive in pinned memory; add them here, in index order
                    const int level_cols = vmaps_curr_d[level_index].cols(), level_rows = vmaps_curr_d[level_index].rows() / 3;
                    if (xs_icp_sum_records(pinned_records_, xs_icp_records_count(level_cols, 0, level_rows), seq, pinned_sums_, 2000000000LL) != 0)
                        return launch_gave_up();
                } else if (icp_publish_pairs) {
                    if (xs_icp_wait_pairs(pinned_pairs_, seq, pinned_sums_, 2000000000LL) != 0) return launch_gave_up();
                } else {
                    long spins = 0;
                    unsigned long long seen;
                    while ((seen = *flag) != seq) {
                        if (seen == (seq | (1ull << 63)) || ++spins > 2000000000L) return launch_gave_up();
#if defined(__x86_64__)
                        __builtin_ia32_pause();
#endif
                    }
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                }
                xs_icp_unpack(pinned_sums_, reinterpret_cast<double *>(A), reinterpret_cast<double *>(b));
                inliers = (long long)pinned_sums_[54];
            } else
                icp_normal_equations(device_Rcurr, device_tcurr, device_Rprev_inv, device_tprev, level_index, A, b, &inliers);
            if (level_index < 3) {   // this iteration's sums are in: the period since the previous iteration's were
                const auto t_now = std::chrono::steady_clock::now();
                const int slot = n == 0 ? 3 : level_index;   // (the frame's first iteration also waits for the previous frame's tail and the map preparation)
                icp_level_us[slot] += std::chrono::duration<double, std::micro>(t_now - t_prev).count();
                ++icp_level_calls[slot];
                t_prev = t_now;
                t_last_sums_ = t_now;
            }
            // The solve and the post come first: the enqueued launch is waiting for them.
            hostComplexICP sol[6];
            llt_solve6(A, b, sol);
            hostComplex result[6];
            for (int i = 0; i < 6; ++i) result[i] = hostComplex((float)sol[i].real(), (float)sol[i].imag());
            const hostComplex alpha = result[0], beta = result[1], gamma = result[2];
            const Matrix3cf Rinc = (angle_axis(gamma, 2) * angle_axis(beta, 1)) * angle_axis(alpha, 0);
            Vector3cf tinc; tinc[0] = result[3]; tinc[1] = result[4]; tinc[2] = result[5];
            const Vector3cf rt = Rinc * tcurr;
            Vector3cf tnext;
            for (int i = 0; i < 3; ++i) tnext[i] = rt[i] + tinc[i];
            const Matrix3frm Rnext = Rinc * Rcurr;
            // the reference's determinant gate (:203-210): a 6x6 LU in double, a fraction of a microsecond, so it is taken
            // before the post — the enqueued launch of a singular system is told to leave (cmd 1) instead of running
            // unobserved on maps the next frame's preparation may already be rewriting
            const double det = real_determinant6(A);
            bool singular = fabs(det) < 1e-15 || std::isnan(det);
            if (debug_fail_icp_iteration_ == n) { singular = true; debug_fail_icp_iteration_ = -1; }   // (test aid)
            if (debug_post_delay_us_[1] > 0) {   // (test aid: a slow host — the enqueued launch keeps polling)
                debug_post_rng_ = debug_post_rng_ * 1664525u + 1013904223u;
                const int span = debug_post_delay_us_[1] - debug_post_delay_us_[0] + 1;
                const int us = debug_post_delay_us_[0] + (int)((debug_post_rng_ >> 8) % (unsigned)(span > 0 ? span : 1));
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(us);
                while (std::chrono::steady_clock::now() < until) {}
            }
            if (next_enqueued) {
                if (singular) xs_icp_post_pose(mailbox, nullptr, nullptr, last_mail_seq, 1);
                else xs_icp_post_pose(mailbox, &device_cast<MatS33>(Rnext).data[0].x.re, &device_cast<devComplex3>(tnext).x.re, next_mail_seq, 0);
            }
            {   // diagnostics: re-pack the 27 sums in launch order
                int shift = 0;
                for (int i = 0; i < 6; ++i)
                    for (int j = i; j < 7; ++j, ++shift) {
                        const hostComplexICP v = (j == 6) ? b[i] : A[i * 6 + j];
                        icp_log.push_back(v.real());
                        icp_log.push_back(v.imag());
                    }
                icp_log.push_back((double)inliers);
            }
            if (singular) {
                if (std::isnan(det)) std::cout << "qnan det" << std::endl;
                else std::cout << "eps det: " << fabs(det) << std::endl;
                stage_end(ST_ICP);
                AbandonClassifiedList();
                return 0;
            }
            tcurr = tnext;
            Rcurr = Rnext;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) c2w_curr(i, j) = Rcurr(i, j);
                c2w_curr(i, 3) = tcurr[i];
            }
            c2w_curr(3, 3) = hostComplex(1.f, 0.f);
        }
    }
    stage_end(ST_ICP);
    world2camera = inverse(c2w_curr);
    world2camera_record.push_back(world2camera);
    return 1;
}

// The same loop with the pose update on the device (xs_icp_iterate): the launches of all levels and
// iterations queue back to back, each leaving the pose for the next in device memory, and the host
// waits once per frame for the final pose instead of once per iteration for the 27 sums.
int KinectFusionReconstruction::PoseEstimateOnDevice(Matrix3frm Rcurr, Vector3cf tcurr, const Matrix3frm &Rprev_inv, const Vector3cf &tprev,
                                                     Matrix4cf c2w_curr, int total_iters) {
    struct PoseState { float R[18]; float t[6]; int status; int iters; double det; double pad[2]; };
    static_assert(sizeof(PoseState) == 128, "pose state layout (xs_icp_iterate)");
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(pinned_sums_ + 56);
    PoseState *pose_host = reinterpret_cast<PoseState *>(pinned_sums_ + 64);
    double *log_host = pinned_sums_ + 128;
    const unsigned long long seq = ++icp_seq_;
    hipStream_t st = current_stream();
    stage_begin(ST_ICP);
    int n = 0;
    for (int level_index = num_levels - 1; level_index >= 0; --level_index) {
        MapArr &vc = vmaps_curr_d[level_index], &nc = nmaps_curr_d[level_index];
        MapArr &vp = vmaps_g_prev_d[level_index], &np_ = nmaps_g_prev_d[level_index];
        const Intr k = kinect_intrinsic(level_index);
        const int rows = vc.rows() / 3, cols = vc.cols();
        for (int iter = 0; iter < icp_iterations[level_index]; ++iter, ++n) {
            const bool first = (n == 0), last = (n == total_iters - 1);
            check_rc(xs_icp_iterate(first ? Rcurr.data() : nullptr, first ? tcurr.data() : nullptr, &vc.ptr()->re, &nc.ptr()->re,
                                    Rprev_inv.data(), tprev.data(), &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres,
                                    angleThres, icp_ws_.ptr(), icp_sums_.ptr(), log_host + 64 * n, icp_pose_.ptr(), pose_host,
                                    last ? reinterpret_cast<unsigned long long *>(pinned_sums_ + 56) : nullptr, seq, st),
                     "xs_icp_iterate");
        }
    }
    long spins = 0;
    while (*flag != seq) {
        if (++spins > 2000000000L) { hipSafeCall(hipStreamSynchronize(st)); break; }  // never expected: fall back to a real wait
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    stage_end(ST_ICP);
    const PoseState ps = *pose_host;
    for (int i = 0; i < ps.iters && i < total_iters; ++i)
        for (int j = 0; j < 55; ++j) icp_log.push_back(log_host[64 * i + j]);
    if (ps.status != 0) {
        if (ps.status == 2) std::cout << "qnan det" << std::endl;
        else std::cout << "eps det: " << fabs(ps.det) << std::endl;
        return 0;
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c2w_curr(i, j) = hostComplex(ps.R[(i * 3 + j) * 2], ps.R[(i * 3 + j) * 2 + 1]);
        c2w_curr(i, 3) = hostComplex(ps.t[2 * i], ps.t[2 * i + 1]);
    }
    c2w_curr(3, 3) = hostComplex(1.f, 0.f);
    world2camera = inverse(c2w_curr);
    world2camera_record.push_back(world2camera);
    return 1;
}

// estimateCombined (ICP.cu:365-429) on this rank's pixel rows, summed over ranks, then unpacked
void KinectFusionReconstruction::icp_normal_equations(const MatS33 &Rcurr, const devComplex3 &tcurr, const MatS33 &Rprev_inv,
                                                      const devComplex3 &tprev, int level, hostComplexICP *A, hostComplexICP *b,
                                                      long long *inliers) {
    MapArr &vc = vmaps_curr_d[level], &nc = nmaps_curr_d[level], &vp = vmaps_g_prev_d[level], &np_ = nmaps_g_prev_d[level];
    const Intr k = kinect_intrinsic(level);
    const int rows = vc.rows() / 3, cols = vc.cols();
    const bool icp_local = shard_count == 1 || !icp_shard_rows;
    const int y0 = icp_local ? 0 : (int)((long long)rows * shard_rank / shard_count);
    const int y1 = icp_local ? rows : (int)((long long)rows * (shard_rank + 1) / shard_count);
    hipStream_t st = current_stream();
    if (icp_local && !profiling_icp_sync) {
        // single GPU: the kernel writes the sums straight into host-coherent pinned memory and then
        // publishes a sequence number; the host spins on it (no copy kernel, no stream synchronise)
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(pinned_sums_ + 56);
        const unsigned long long seq = ++icp_seq_;
        if (icp_host_fold) {
            check_rc(xs_icp_accumulate_records(&Rcurr.data[0].x.re, &tcurr.x.re, nullptr, 0, &vc.ptr()->re, &nc.ptr()->re, &Rprev_inv.data[0].x.re,
                                               &tprev.x.re, &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres, y0, y1,
                                               pinned_records_, seq, st), "estimateCombined (records)");
            if (xs_icp_sum_records(pinned_records_, xs_icp_records_count(cols, y0, y1), seq, pinned_sums_, 2000000000LL) != 0) {
                printf("HIP error(estimateCombined): the ICP records never arrived\n");
                exit(-1);
            }
        } else {
            check_rc(xs_icp_accumulate(&Rcurr.data[0].x.re, &tcurr.x.re, &vc.ptr()->re, &nc.ptr()->re, &Rprev_inv.data[0].x.re, &tprev.x.re,
                                       &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres, y0, y1,
                                       icp_ws_.ptr(), pinned_sums_, reinterpret_cast<unsigned long long *>(pinned_sums_ + 56), seq, st),
                     "estimateCombined");
            long spins = 0;
            while (*flag != seq) {
                if (++spins > 2000000000L) { hipSafeCall(hipStreamSynchronize(st)); break; }  // never expected: fall back to a real wait
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    } else {
        check_rc(xs_icp_accumulate(&Rcurr.data[0].x.re, &tcurr.x.re, &vc.ptr()->re, &nc.ptr()->re, &Rprev_inv.data[0].x.re, &tprev.x.re,
                                   &k.fx, &vp.ptr()->re, &np_.ptr()->re, vc.step(), rows, cols, distThres, angleThres, y0, y1,
                                   icp_ws_.ptr(), icp_sums_.ptr(), nullptr, 0, st), "estimateCombined");
        if (!icp_local && collective) collective(collective_user, 0, icp_sums_.ptr(), 55);  // the 440-byte all-reduce
        hipSafeCall(hipMemcpyAsync(pinned_sums_, icp_sums_.ptr(), 55 * sizeof(double), hipMemcpyDeviceToHost, st));
        hipSafeCall(hipStreamSynchronize(st));
    }
    xs_icp_unpack(pinned_sums_, reinterpret_cast<double *>(A), reinterpret_cast<double *>(b));
    if (inliers) *inliers = (long long)pinned_sums_[54];
}

// Profiling level 1 attaches a start / stop event pair to the integrate kernel's dispatch (the roofline's kernel_ms).  The START event costs the
// launch call 4.8 us of host time (9.1 us against 4.2 with the completion event alone: bench.py's tail_host_us), on the critical path
// between the last ICP reduction and the raycast — 3 % of the frame rate it is there to annotate.  So the pair rides on every
// profile_integrate_every-th frame only (default 4; level 2, the per-stage pass, times every frame): the mean kernel time is that of the
// sampled launches.
bool KinectFusionReconstruction::IntegrateKernelTimedThisFrame() const {
    return profiling && (profiling_stages || profile_integrate_every <= 1 || counter_frame_ % profile_integrate_every == 0);
}
// one integrate call per frame (the volume is not sharded): its header clear and count fold can leave the main stream
bool KinectFusionReconstruction::integrate_split() const { return zs0 == zo0 && zs1 == zo1 && integrate_ws_.ptr() != nullptr; }
// the voxel count of the last integrate call still sits in the workspace header: fold it into its frame's counter slot
// (on `st`, which must be ordered behind that call: the main stream, or the auxiliary one after integrate_done_)
void KinectFusionReconstruction::flush_pending_fold(hipStream_t st) {
    if (!pending_fold_) return;
    check_rc(xs_integrate_fold_counts(integrate_ws_.ptr(), pending_fold_, st), "integrate count");
    pending_fold_ = nullptr;
}

// reference :237-278
// the volume-to-camera pose a list is classified for, from a camera-to-world one
void KinectFusionReconstruction::SetListPose(const Matrix4cf &c2w) {
    Matrix4cf v2c = inverse(world2volume * c2w);
    Matrix3frm Rv2c = GetRotation(v2c);
    Vector3cf tv2c = GetTranslation(v2c);
    std::memcpy(list_Rv2c_, &device_cast<MatS33>(Rv2c).data[0].x.re, sizeof(list_Rv2c_));
    std::memcpy(list_tv2c_, &device_cast<devComplex3>(tv2c).x.re, sizeof(list_tv2c_));
}
// xs_integrate_classify for the camera pose (Rcurr, tcurr) = camera-to-world, on the main stream (behind the ICP launches)
void KinectFusionReconstruction::ClassifyAhead(const Matrix3frm &Rcurr, const Vector3cf &tcurr) {
    Matrix4cf c2w;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) c2w(i, j) = hostComplex(i == j ? 1.f : 0.f, 0.f);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c2w(i, j) = Rcurr(i, j);
        c2w(i, 3) = tcurr[i];
    }
    SetListPose(c2w);
    // integrate_classify_beside_icp (off by default: measured, no gain — see the header): on the auxiliary stream instead; everything the two
    // classification kernels read — the scaled depth's maximum, the tile table, the cleared header — was written on that stream.  The integrate
    // launch then waits for their completion event.
    const bool beside = integrate_classify_beside_icp && aux_stream_ && integrate_header_clear_;
    hipStream_t st = beside ? aux_stream_ : current_stream();
    // the scaled depth's maximum and the cleared header come from the auxiliary stream
    if (!beside && scale_recorded_ && hipEventQuery(scale_done_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(st, scale_done_, 0));
    EnqueueClassification(st, beside);
    EnqueuePostedIntegrate();
}
// the two classification kernels for list_Rv2c_ / list_tv2c_ on stream st; with_event: their completion rides on the last dispatch
// (classify_done_: the stream that integrates is another one)
void KinectFusionReconstruction::EnqueueClassification(hipStream_t st, bool with_event) {
    const int res[3] = {volume_resolution.x(), volume_resolution.y(), volume_resolution.z()};
    xs_integrate_opts o = {};
    o.struct_bytes = sizeof(o);
    o.flags = integrate_header_clear_ ? XS_INTEGRATE_HEADER_IS_CLEAR : 0u;
    o.depth_tiles = depth_tiles_.ptr();   // the boxes' classes are decided here too, with the slack's pads (the integrate call checks its pose against them)
    o.stop_event = with_event ? classify_done_ : nullptr;
    check_rc(xs_integrate_classify_ex(depth_height, depth_width, &kinect_intrinsic.fx, res, voxel_size, list_Rv2c_, list_tv2c_,
                                      tsdf_volume_d_ptr->getTsdfTruncDist(), zo0, zo1, depth_max_.ptr(), integrate_ws_.ptr(), integrate_classify_slack,
                                      &o, st), "integrate classification");
    classify_recorded_ = with_event;
    list_ready_ = true;
}
// integrate_classify_predicted (off by default): the classification for the pose the frame is EXPECTED to end at — the previous pose moved
// on by the previous frame's motion — at the frame's very start, on the auxiliary stream (behind the header clear and the depth scaling,
// i.e. under the previous frame's raycast and the first, small ICP launches); IntegrateFrame checks the final pose against the list's and
// the classes' slack as it does for a list classified behind the last ICP launch, and a frame that moved otherwise classifies again there.
// It would take the two classification kernels (4.8 + 7.8 us) off the chain between the last ICP reduction and the integrate kernel, but
// on the benchmark scene — which slides along a wall: the estimated trajectory jitters by more than the slack allows — only a third of
// the frames are covered at slack 2 and 72 % at slack 6, where the wider pads cost the integrate kernel 4 us; the others classify after
// the final pose, i.e. later than ClassifyAhead would have: no gain (profiles/r04_ab_classify_predicted.txt).  Same volume bit for bit
// either way (tested).
void KinectFusionReconstruction::ClassifyPredicted() {
    list_predicted_ = false;
    if (!integrate_classify_predicted || !integrate_classify_ahead || integrate_post_pose || !integrate_split() || !integrate_header_clear_ ||
        !aux_stream_ || world2camera_record.empty() || use_gtPose)
        return;
    Matrix4cf w2c = world2camera_record.back();
    if (world2camera_record.size() >= 2) {   // constant velocity: W(n+1) = (W(n) W(n-1)^-1) W(n)
        const Matrix4cf step = w2c * inverse(world2camera_record[world2camera_record.size() - 2]);
        w2c = step * w2c;
    }
    SetListPose(inverse(w2c));
    EnqueueClassification(aux_stream_, true);
    list_predicted_ = true;
}

// this frame's counter slot; entering a half of the ring clears that half (its frames were folded or abandoned at least COUNTER_RING / 2
// frames ago) — once per frame, whichever of the two integrate paths comes first
unsigned long long *KinectFusionReconstruction::PrepareFrameCounters(hipStream_t st) {
    if (counters_prepared_for_ != counter_frame_) {
        counters_prepared_for_ = counter_frame_;
        if (counter_frame_ % (COUNTER_RING / 2) == 0 && counter_frame_ > 0)
            hipSafeCall(hipMemsetAsync(frame_counters(), 0, (COUNTER_RING / 2) * 2 * sizeof(unsigned long long), st));
    }
    return frame_counters();
}

// Behind the classification: the integrate kernel itself, to take the final pose from its mailbox (k_integrate_bricks<., true>).  Everything
// IntegrateFrame does around its launch happens here; IntegrateFrame then only checks that the final pose is covered and posts it.
void KinectFusionReconstruction::EnqueuePostedIntegrate() {
    if (!integrate_post_pose || !integrate_split() || !integrate_mailbox_ || !integrate_mailbox_in_device_ || !list_ready_) return;
    if (zs0 != zo0 || zs1 != zo1) return;   // (halo bands: several calls per frame)
    hipStream_t st = current_stream();
    WaitForClassification(st);
    unsigned long long *counters = PrepareFrameCounters(st);
    const int res[3] = {volume_resolution.x(), volume_resolution.y(), volume_resolution.z()};
    DeviceArray2D<float> value = tsdf_volume_d_ptr->value(), grad = tsdf_volume_d_ptr->grad();
    DeviceArray2D<int> weight = tsdf_volume_d_ptr->weight();
    xs_integrate_opts o = {};
    o.struct_bytes = sizeof(o);
    hipEvent_t integrate_stop = integrate_done_;
    if (IntegrateKernelTimedThisFrame()) {
        integrate_stop = prof_ring_[prof_pending_].ev[ST_INTEGRATE][1];
        o.start_event = prof_ring_[prof_pending_].ev[ST_INTEGRATE][0];
        prof_ring_[prof_pending_].used[ST_INTEGRATE] = true;
    }
    o.stop_event = integrate_stop;
    if (++integrate_mail_seq_ == 0u) ++integrate_mail_seq_;
    posted_seq_ = integrate_mail_seq_;
    o.pose_mailbox = integrate_mailbox_; o.mailbox_seq = posted_seq_; o.mailbox_slack = integrate_classify_slack; o.pose_dev = posted_pose_.ptr();
    o.signmap = sign_map_ptr();
    o.depth_tiles = depth_tiles_.ptr();
    const bool split = integrate_header_clear_;   // header cleared and count folded on the auxiliary stream (SurfaceMeasure)
    o.flags = XS_INTEGRATE_POSE_POSTED | XS_INTEGRATE_LIST_IS_READY | XS_INTEGRATE_HEADER_IS_CLEAR | (split ? XS_INTEGRATE_NO_FOLD : 0u);
    check_rc(xs_integrate_scaled_ex2(depthRawScaled_d.ptr(), depthRawScaled_d.step(), depth_height, depth_width, &kinect_intrinsic.fx, max_integration_weight,
                                     res, voxel_size, list_Rv2c_, list_tv2c_, tsdf_volume_d_ptr->getTsdfTruncDist(), value.ptr(0), weight.ptr(0), grad.ptr(0),
                                     value.step(), biInterpolate_threshold, zo0, zo1, counters, depth_max_.ptr(), integrate_ws_.ptr(), &o, st),
             "integrateTsdfVolume (posted)");
    posted_pending_ = true;
    posted_at_ = std::chrono::steady_clock::now();
    posted_stop_ = integrate_stop;
    posted_split_ = split;
}

// A frame whose alignment fails after ClassifyAhead has run never reaches IntegrateFrame: the classification kernel is still in the main
// stream, has filled the workspace header, and reads the frame's depth maximum — while the retried frame's SurfaceMeasure clears that header
// and rewrites the maximum on the auxiliary stream, ordered only behind the previous frame's integrate kernel.  So the failure path (rare: a
// singular system, a launch that timed out) drains the main stream, clears the header there and forgets the list.
// the list and classes ClassifyAhead left on the auxiliary stream: the stream st reads them next
void KinectFusionReconstruction::WaitForClassification(hipStream_t st) {
    if (!classify_recorded_) return;
    classify_recorded_ = false;
    if (hipEventQuery(classify_done_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(st, classify_done_, 0));
}
void KinectFusionReconstruction::AbandonClassifiedList() {
    if (!list_ready_) return;
    list_ready_ = false;
    if (classify_recorded_) { classify_recorded_ = false; hipSafeCall(hipStreamSynchronize(aux_stream_)); }
    if (posted_pending_) {   // the integrate launch waiting for this frame's pose leaves without touching the volume
        xs_icp_post_pose(integrate_mailbox_, nullptr, nullptr, posted_seq_, 1);
        posted_pending_ = false;
    }
    check_rc(xs_integrate_workspace_clear(integrate_ws_.ptr(), current_stream()), "integrate workspace");
    hipSafeCall(hipStreamSynchronize(current_stream()));
    integrate_header_clear_ = true;
}
void KinectFusionReconstruction::DebugSetIcpSequence(unsigned long long v) {
    synchronize();
    icp_seq_ = v;
    // the mailbox holds the number of the last post: bring it up to date, or a launch numbered just above v would take the stale, smaller
    // number for a later one after the 2^32 wrap (an abandon command nobody is waiting for)
    if (icp_mailbox_ && (unsigned)v != 0u) xs_icp_post_pose(icp_mailbox_, nullptr, nullptr, (unsigned)v, 1);
}

int KinectFusionReconstruction::IntegrateFrame(const DeviceArray2D<ushort> &depth_frame_d) {
    if (use_gtPose) {
        Matrix4cf c2w = gt_poses[frame_id];
        world2camera = inverse(c2w);
        world2camera_record.back() = world2camera;
    }
    Matrix4cf c2w = inverse(world2camera_record.back());
    Matrix4cf c2v = world2volume * c2w;
    Matrix4cf v2c = inverse(c2v);
    Vector3cf tc2v = GetTranslation(c2v);
    auto &device_tc2v = device_cast<devComplex3>(tc2v);
    Matrix3frm Rv2c = GetRotation(v2c);
    auto &device_Rv2c = device_cast<MatS33>(Rv2c);
    Vector3cf tv2c = GetTranslation(v2c);
    auto &device_tv2c = device_cast<devComplex3>(tv2c);
    (void)device_tc2v;

    int3 volume_res;
    volume_res.x = volume_resolution.x();
    volume_res.y = volume_resolution.y();
    volume_res.z = volume_resolution.z();
    hipStream_t st = current_stream();
    const auto t_enter = std::chrono::steady_clock::now();
    auto t_call = t_enter, t_back = t_enter;
    unsigned long long *counters = PrepareFrameCounters(st);
    // A posted integrate launch is waiting in the stream for this pose (EnqueuePostedIntegrate): if the pose's frustum lies inside the planes
    // that launch was given, post it — the launch is the frame's integrate call; else tell it to leave and take the plain path below.
    bool integrated_by_post = false;
    if (posted_pending_) {
        posted_pending_ = false;
        const int res_[3] = {volume_res.x, volume_res.y, volume_res.z};
        // (a launch whose gate has waited long may have given up — MAILBOX_MAX_POLLS, about a second — and left without writing: a host that
        // took more than a quarter of that between enqueue and post tells it to leave and integrates the plain way, whatever the gate did)
        const bool in_time = std::chrono::steady_clock::now() - posted_at_ < std::chrono::milliseconds(250);
        if (in_time && xs_integrate_pose_covered(depth_frame_d.rows(), depth_frame_d.cols(), &kinect_intrinsic.fx, res_, voxel_size, list_Rv2c_, list_tv2c_,
                                                 integrate_classify_slack, &device_Rv2c.data[0].x.re, &device_tv2c.x.re)) {
            xs_icp_post_pose(integrate_mailbox_, &device_Rv2c.data[0].x.re, &device_tv2c.x.re, posted_seq_, 0);
            integrated_by_post = true;
            ++posted_accepted_;
            list_ready_ = false;
            if (posted_split_) { integrate_header_clear_ = false; pending_fold_ = counters; }
        } else {   // (never seen: the last ICP update moved the frustum further than the widened planes allow for)
            xs_icp_post_pose(integrate_mailbox_, nullptr, nullptr, posted_seq_, 1);
            ++posted_refused_;
            list_ready_ = false;
            check_rc(xs_integrate_workspace_clear(integrate_ws_.ptr(), st), "integrate workspace");
        }
    }
    // the depth scaling ran on the auxiliary stream behind the map preparation
    // (a wait is a packet the next kernel queues behind: none is enqueued for an event that has already completed — the
    // scaling finished under the ICP loop long ago)
    if (scale_recorded_ && hipEventQuery(scale_done_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(st, scale_done_, 0));
    float *depth_max_dev = depth_max_.ptr();  // filled with the scaled depth by SurfaceMeasure, on the auxiliary stream
    // integrateTsdfVolume (TsdfFusion.cu:173-201), its two launches timed separately
    const int res[3] = {volume_res.x, volume_res.y, volume_res.z};
    DeviceArray2D<float> value = tsdf_volume_d_ptr->value(), grad = tsdf_volume_d_ptr->grad();
    DeviceArray2D<int> weight = tsdf_volume_d_ptr->weight();
    // ST_INTEGRATE brackets the integrate kernel proper of the owned planes (the events are recorded
    // inside xs_integrate_scaled, after the brick classification): the figure the roofline uses
    // The integrate kernel's completion is what the auxiliary stream waits for before it touches the scaled depth / the
    // workspace header again.  With one integrate call per frame that completion event rides on the kernel's own dispatch
    // (its stop event — the profiling pair's when profiling) instead of a marker packet behind it, which the raycast
    // launch would queue behind (~5 us of every frame).
    hipEvent_t integrate_stop = integrate_done_;
    xs_integrate_opts o = {};   // everything the integrate calls below take besides their arguments proper (no per-thread setters)
    o.struct_bytes = sizeof(o);
    if (integrated_by_post) integrate_stop = posted_stop_;
    else if (IntegrateKernelTimedThisFrame()) {
        integrate_stop = prof_ring_[prof_pending_].ev[ST_INTEGRATE][1];
        o.start_event = prof_ring_[prof_pending_].ev[ST_INTEGRATE][0]; o.stop_event = integrate_stop;
        prof_ring_[prof_pending_].used[ST_INTEGRATE] = true;
    } else if (integrate_split())
        o.stop_event = integrate_stop;
    if (!integrated_by_post) {
        o.signmap = sign_map_ptr();   // (a rank of a sharded volume: the owned planes and both halo bands mark it)
        o.depth_tiles = depth_tiles_.ptr();
        // owned planes (counted), then the two halo bands every neighbour also integrates: the
        // update is per voxel and deterministic, so a halo voxel carries the owner's exact bits
        const int zr[3][2] = {{zo0, zo1}, {zs0, zo0}, {zo1, zs1}};
        for (int i = 0; i < 3; ++i) {
            const int za = zr[i][0], zb = zr[i][1];
            if (zb <= za) continue;
            const size_t off = (size_t)(za - zs0) * res[1];
            // header cleared and count folded on the auxiliary stream (SurfaceMeasure) when there is one call per frame
            const bool split = integrate_split() && integrate_header_clear_ && i == 0;
            unsigned list_flag = 0;
            if (i == 0 && list_ready_) {
                list_ready_ = false; list_predicted_ = false;
                WaitForClassification(st);
                const int covers = xs_integrate_list_covers(depth_frame_d.rows(), depth_frame_d.cols(), &kinect_intrinsic.fx, res, voxel_size, list_Rv2c_,
                                                            list_tv2c_, integrate_classify_slack, &device_Rv2c.data[0].x.re, &device_tv2c.x.re);
                ++list_cover_counts_[covers & 3];
                if (covers)   // (bit 1 clear: the list holds but the boxes' classes were padded for a nearer pose — they are decided again, the list stays)
                    list_flag = XS_INTEGRATE_LIST_IS_READY | XS_INTEGRATE_HEADER_IS_CLEAR | ((covers & 2) ? 0u : XS_INTEGRATE_RECLASSIFY_BOXES);
                else   // the last update moved the frustum further than the widened list allows for (never seen): start over
                    check_rc(xs_integrate_workspace_clear(integrate_ws_.ptr(), st), "integrate workspace");
            }
            o.flags = (split ? (XS_INTEGRATE_HEADER_IS_CLEAR | XS_INTEGRATE_NO_FOLD) : 0u) | list_flag;
            if (i == 0) t_call = std::chrono::steady_clock::now();
            check_rc(xs_integrate_scaled_ex2(depthRawScaled_d.ptr(), depthRawScaled_d.step(), depth_frame_d.rows(), depth_frame_d.cols(),
                                             &kinect_intrinsic.fx, max_integration_weight, res, voxel_size, &device_Rv2c.data[0].x.re,
                                             &device_tv2c.x.re, tsdf_volume_d_ptr->getTsdfTruncDist(), value.ptr((int)off), weight.ptr((int)off),
                                             grad.ptr((int)off), value.step(), biInterpolate_threshold, za, zb, i == 0 ? counters : nullptr,
                                             depth_max_dev, integrate_ws_.ptr(), &o, st),
                     "integrateTsdfVolume");
            if (split) { integrate_header_clear_ = false; pending_fold_ = counters; }
            if (i == 0) { t_back = std::chrono::steady_clock::now(); o.start_event = nullptr; o.stop_event = nullptr; }   // (the event pair rides on the owned planes' launch only)
        }
    }

    if (integrate_split()) integrate_done_now_ = integrate_stop;          // attached to the dispatch above
    else { hipSafeCall(hipEventRecord(integrate_done_, st)); integrate_done_now_ = integrate_done_; }   // after the halo calls too
    integrate_recorded_ = true;
    stage_begin(ST_RAYCAST);
    hits_counter_ = counters + 1;
    CalculatePointCloud(vmaps_g_prev_d[0], nmaps_g_prev_d[0]);
    hits_counter_ = nullptr;
    if (frame_id > 0 && !use_gtPose && !integrated_by_post) {
        const auto t_ray = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        tail_host_us[0] += us(t_last_sums_, t_enter); tail_host_us[1] += us(t_enter, t_call); tail_host_us[2] += us(t_call, t_back); tail_host_us[3] += us(t_back, t_ray);
        ++tail_host_calls;
    }
    stage_end(ST_RAYCAST);
    stage_begin(ST_RESIZE);
    ModelMapPyramid();
    stage_end(ST_RESIZE);
    ++counter_frame_;
    return 1;
}

// reference :272-277: resizeVMap / resizeNMap per level.  With three levels both halvings of both maps
// are one launch (xs_resize_pyramid: same values).
// levels 1 and 2 of the model maps allocated, and all three levels' vertex / normal maps of equal pitch (what the one-launch pyramid — its
// own kernel or the raycast's epilogue — wants): false = the per-level launches
bool KinectFusionReconstruction::PreparePyramidLevels() {
    if (num_levels != 3) return false;
    const int rows0 = vmaps_g_prev_d[0].rows() / 3, cols0 = vmaps_g_prev_d[0].cols();
    for (int i = 1; i < 3; ++i) {
        vmaps_g_prev_d[i].create((rows0 >> i) * 3, cols0 >> i);
        nmaps_g_prev_d[i].create((rows0 >> i) * 3, cols0 >> i);
    }
    return vmaps_g_prev_d[0].step() == nmaps_g_prev_d[0].step() && vmaps_g_prev_d[1].step() == nmaps_g_prev_d[1].step() &&
           vmaps_g_prev_d[2].step() == nmaps_g_prev_d[2].step();
}
void KinectFusionReconstruction::ModelMapPyramid() {
    if (pyramid_in_raycast_) { pyramid_in_raycast_ = false; return; }   // built by the raycast launch (CalculatePointCloud)
    if (num_levels == 3) {
        const int rows0 = vmaps_g_prev_d[0].rows() / 3, cols0 = vmaps_g_prev_d[0].cols();
        if (PreparePyramidLevels()) {
            // (its completion = the end of the frame's tail: what the announced next frame's map preparation waits for, HintNextFrame)
            check_rc(xs_resize_pyramid_ex(&vmaps_g_prev_d[0].ptr()->re, &nmaps_g_prev_d[0].ptr()->re, vmaps_g_prev_d[0].step(), rows0, cols0,
                                          &vmaps_g_prev_d[1].ptr()->re, &nmaps_g_prev_d[1].ptr()->re, vmaps_g_prev_d[1].step(),
                                          &vmaps_g_prev_d[2].ptr()->re, &nmaps_g_prev_d[2].ptr()->re, vmaps_g_prev_d[2].step(), tail_done_, current_stream()),
                     "resizeMap");
            tail_recorded_ = true;
            return;
        }
    }
    for (int i = 1; i < num_levels; ++i) {
        resizeVMap(vmaps_g_prev_d[i - 1], vmaps_g_prev_d[i], false);
        resizeNMap(nmaps_g_prev_d[i - 1], nmaps_g_prev_d[i], false);
    }
}

// The announced next frame's map preparation (HintNextFrame), on the second stream, in stages: called from the ICP loop (PoseEstimate), one
// stage per iteration, and with `all` at the end of AlignDepthToReconstruction for whatever is left (or where there is no ICP loop).
void KinectFusionReconstruction::EnqueueAnnouncedFrame(bool all) {
    // One stage per call from the ICP loop (the host has ~8 us to spare per iteration: one more launch next to the ICP launch it enqueues while
    // the GPU runs the previous one — all six at once delayed the frame's second pose by 15 us), the rest at once when `all`.
    while (next_hint_ptr_) {
        hipStream_t main_stream = current_stream();
        current_stream() = aux_stream_;
        // The NEXT frame's map preparation — bilateral filter, depth pyramid, vertex / normal maps, scaled depth — goes into the second set of
        // buffers: it depends on that depth image alone and runs while this frame's ICP launches (45 to 256 workgroups each, waiting on one
        // another) leave most of the GPU idle, instead of next to this frame's integrate and raycast, which the next frame's first ICP launch
        // waits for.  The sets change places around each stage only: the ICP launches enqueued in between read this frame's maps.
        SwapMapSets();
        const DeviceArray2D<ushort> next(depth_height, depth_width, const_cast<void *>(next_hint_ptr_), next_hint_step_);   // borrowed
        if (next_stage_ == 0) {
            if (depths_curr_d.size() != depths_next_d.size()) depths_curr_d.resize(depths_next_d.size());
            for (size_t i = 0; i < depths_curr_d.size(); ++i)
                if (depths_curr_d[i].rows() != depths_next_d[i].rows() || depths_curr_d[i].cols() != depths_next_d[i].cols())
                    depths_curr_d[i].create(depths_next_d[i].rows(), depths_next_d[i].cols());
            if (vmaps_curr_d.size() != vmaps_next_d.size()) { vmaps_curr_d.resize(vmaps_next_d.size()); nmaps_curr_d.resize(nmaps_next_d.size()); }
            if (depthRawScaled_d.rows() != depth_height || depthRawScaled_d.cols() != depth_width) depthRawScaled_d.create(depth_height, depth_width);
            if (!depth_max_.ptr()) depth_max_.create(4);
            if (!depth_tiles_.ptr()) depth_tiles_.create(xs_depth_tiles_bytes(depth_height, depth_width));
            // ... not before the previous frame's raycast and pyramid are through (the main stream's ICP launches start there): the event rides
            // on that pyramid's dispatch, the wait is a packet of this stream only
            if (tail_recorded_ && hipEventQuery(tail_done_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(aux_stream_, tail_done_, 0));
            // (this set's scaled depth was the previous frame's: that frame's integrate must be through with it — it is, wherever the pyramid's
            // event above exists; the explicit wait covers the configurations without one)
            if (integrate_recorded_ && hipEventQuery(integrate_done_now_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(aux_stream_, integrate_done_now_, 0));
            SmoothDepthFrame(depths_curr_d[0], next);
        } else if (next_stage_ < num_levels) {
            pyrDown(depths_curr_d[next_stage_ - 1], depths_curr_d[next_stage_]);
        } else if (next_stage_ == num_levels) {
            EnqueueMapsFromPyramid();
            hipSafeCall(hipEventRecord(surface_done_, aux_stream_));
        } else {
            EnqueueScale(next);
        }
        SwapMapSets();
        current_stream() = main_stream;
        if (++next_stage_ > num_levels + 1) {
            next_ready_ptr_ = next_hint_ptr_; next_ready_step_ = next_hint_step_; next_ready_ = true;
            next_hint_ptr_ = nullptr;
            next_stage_ = 0;
        }
        if (!all) break;
    }
}

// createVMap + createNMap of the current set's depth pyramid, on current_stream()
void KinectFusionReconstruction::EnqueueMapsFromPyramid() {
    {   // createVMap + createNMap per level (camera frame, +z forward), all levels in one launch
        Intr ks[3];
        const float *dp[3]; size_t ds[3], ms[3];
        float *vp[3], *np_[3];
        bool uniform = num_levels <= 3;
        for (int i = 0; i < num_levels && uniform; ++i) {
            ks[i] = kinect_intrinsic(i);
            vmaps_curr_d[i].create(depths_curr_d[i].rows() * 3, depths_curr_d[i].cols());
            nmaps_curr_d[i].create(depths_curr_d[i].rows() * 3, depths_curr_d[i].cols());
            dp[i] = &depths_curr_d[i].ptr()->re; ds[i] = depths_curr_d[i].step();
            vp[i] = &vmaps_curr_d[i].ptr()->re; np_[i] = &nmaps_curr_d[i].ptr()->re; ms[i] = vmaps_curr_d[i].step();
            uniform = depths_curr_d[i].rows() == (depth_height >> i) && depths_curr_d[i].cols() == (depth_width >> i) &&
                      nmaps_curr_d[i].step() == ms[i];
        }
        real_maps_valid_ = false;
        if (uniform && icp_real_current_maps) {
            float *vr[3], *nr[3]; size_t rs[3];
            vreal_curr_d.resize(num_levels); nreal_curr_d.resize(num_levels);
            for (int i = 0; i < num_levels; ++i) {
                vreal_curr_d[i].create(depths_curr_d[i].rows() * 3, depths_curr_d[i].cols());
                nreal_curr_d[i].create(depths_curr_d[i].rows() * 3, depths_curr_d[i].cols());
                vr[i] = vreal_curr_d[i].ptr(); nr[i] = nreal_curr_d[i].ptr(); rs[i] = vreal_curr_d[i].step();
                uniform = uniform && nreal_curr_d[i].step() == rs[i];
            }
            if (uniform) {
                check_rc(xs_create_vnmaps_real(num_levels, &ks[0].fx, dp, ds, depth_height, depth_width, vp, np_, ms, vr, nr, rs, current_stream()), "createVMap");
                real_maps_valid_ = true;
            }
        }
        if (real_maps_valid_) {
        } else if (uniform)
            check_rc(xs_create_vnmaps(num_levels, &ks[0].fx, dp, ds, depth_height, depth_width, vp, np_, ms, current_stream()), "createVMap");
        else
            for (int i = 0; i < num_levels; ++i) {
                createVMap(kinect_intrinsic(i), depths_curr_d[i], vmaps_curr_d[i]);
                createNMap(vmaps_curr_d[i], nmaps_curr_d[i]);
            }
    }
}

// scaleDepthKernal of integrateTsdfVolume (TsdfFusion.cu:182-187) into the current set, on the auxiliary stream
void KinectFusionReconstruction::EnqueueScale(const DeviceArray2D<ushort> &depth_frame_d) {
    hipSafeCall(hipMemsetAsync(depth_max_.ptr(), 0, sizeof(float), aux_stream_));
    // (+ the per-tile depth range the integrate call classifies its bricks with: free space / nothing to write / per-voxel walk)
    check_rc(xs_scale_depth_tiles(depth_frame_d.ptr(), depth_frame_d.step(), depth_frame_d.rows(), depth_frame_d.cols(), depthRawScaled_d.ptr(),
                                  depthRawScaled_d.step(), depth_max_.ptr(), depth_tiles_.ptr(), aux_stream_), "scaleDepth");
    hipSafeCall(hipEventRecord(scale_done_, aux_stream_));
    scale_recorded_ = true;
}

// the two sets of per-frame buffers (filtered depth pyramid, vertex / normal maps and their real planes, scaled depth and its maximum,
// the events that say when they are ready) change places
void KinectFusionReconstruction::SwapMapSets() {
    std::swap(depths_curr_d, depths_next_d);
    std::swap(vmaps_curr_d, vmaps_next_d);
    std::swap(nmaps_curr_d, nmaps_next_d);
    std::swap(vreal_curr_d, vreal_next_d);
    std::swap(nreal_curr_d, nreal_next_d);
    std::swap(depthRawScaled_d, depthRawScaled_next_d);
    std::swap(depth_max_, depth_max_next_);
    std::swap(depth_tiles_, depth_tiles_next_);
    std::swap(surface_done_, surface_done_next_);
    std::swap(scale_done_, scale_done_next_);
    std::swap(real_maps_valid_, real_maps_valid_next_);
    std::swap(scale_recorded_, scale_recorded_next_);
}

// reference :280-299
void KinectFusionReconstruction::SurfaceMeasure(const DeviceArray2D<ushort> &depth_frame_d) {
    if (depth_width <= 0 || depth_height <= 0) {
        std::cout << "error::KinectFusionReconstruction, not created yet" << std::endl;
        return;
    }
    // The current-frame maps depend only on the new depth image, which must be complete when
    // ProcessFrame is called: they are built on a second stream, so they run under whatever the
    // main stream still has in flight from the previous frame (raycast, pyramid), and the main
    // stream picks them up through an event before the first ICP launch.
    hipStream_t main_stream = current_stream();
    current_stream() = aux_stream_;
    stage_begin(ST_SURFACE);
    const bool adopted = next_ready_ && next_ready_ptr_ == (const void *)depth_frame_d.ptr() && next_ready_step_ == depth_frame_d.step();
    next_ready_ = false;
    if (adopted) {
        // this frame was announced while the previous one was being tracked (HintNextFrame): its maps and its scaled depth were built then, on
        // this same stream, in the second set of buffers — the sets change places, and so do the events that say when they were ready
        SwapMapSets();
    } else {
        SmoothDepthFrame(depths_curr_d[0], depth_frame_d);
        for (int i = 1; i < num_levels; ++i) pyrDown(depths_curr_d[i - 1], depths_curr_d[i]);
        EnqueueMapsFromPyramid();
    }
    stage_end(ST_SURFACE);
    // the ICP needs the maps and nothing else of this stream: the main stream picks them up here, not
    // behind the depth scaling below (two more kernels and their packets in front of the first ICP launch)
    if (!adopted) hipSafeCall(hipEventRecord(surface_done_, aux_stream_));
    // scaleDepthKernal of integrateTsdfVolume (TsdfFusion.cu:182-187) also depends on the depth image
    // alone: metres + the frame's largest valid depth, ready long before integrate asks for them
    // (the previous frame's integrate, possibly still running on the main stream, reads the same
    // buffers: wait for it — it is the first thing in that frame's tail)
    if (integrate_recorded_ && hipEventQuery(integrate_done_now_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(aux_stream_, integrate_done_now_, 0));
    // The two tiny launches that bracket an integrate call — the fold of the previous frame's voxel count and the clear of
    // the brick-list header — run here, off the main stream's dependent chain (each cost that chain a dispatch: ~14 us a
    // frame); the integrate call below is told so (IntegrateFrame).  One integrate call per frame only, i.e. not sharded.
    if (integrate_split()) {
        flush_pending_fold(aux_stream_);
        check_rc(xs_integrate_workspace_clear(integrate_ws_.ptr(), aux_stream_), "integrate workspace");
        integrate_header_clear_ = true;
    }
    if (!adopted) {
        stage_begin(ST_SCALE);
        EnqueueScale(depth_frame_d);
        stage_end(ST_SCALE);
    }
    list_ready_ = false;
    ClassifyPredicted();
    current_stream() = main_stream;
    // the main stream picks the maps up — without a wait packet when they are already there (the usual case once the
    // previous frame's tail is the longer of the two)
    if (hipEventQuery(surface_done_) != hipSuccess) hipSafeCall(hipStreamWaitEvent(main_stream, surface_done_, 0));
}

// reference :302-332
int KinectFusionReconstruction::CalculatePointCloud(MapArr &xyz_g_d, MapArr &normal_g_d) {
    Matrix4cf c2w = inverse(world2camera);
    Matrix4cf c2v = world2volume * c2w;
    Matrix4cf v2w = inverse(world2volume);
    Matrix3frm Rc2v = GetRotation(c2v);
    Vector3cf tc2v = GetTranslation(c2v);
    Matrix3frm Rv2w = GetRotation(v2w);
    Vector3cf tv2w = GetTranslation(v2w);
    auto &device_Rc2v = device_cast<MatS33>(Rc2v);
    auto &device_tc2v = device_cast<devComplex3>(tc2v);
    auto &device_Rv2w = device_cast<MatS33>(Rv2w);
    auto &device_tv2w = device_cast<devComplex3>(tv2w);
    int3 volume_res;
    volume_res.x = volume_resolution.x();
    volume_res.y = volume_resolution.y();
    volume_res.z = volume_resolution.z();
    if (sign_map_stale_) { RebuildSignMap(); sign_map_stale_ = false; }   // (xs_kf_volume_ptr handed the value array out since the last raycast)
    xs_raycast_opts ro = {};   // the sign map, the pyramid outputs and the completion event of the raycast calls below (no per-thread setters)
    ro.struct_bytes = sizeof(ro);
    ro.signmap = sign_map_ptr(); ro.signmap_shift = raycast_sign_map_shift; ro.signmap_tranc_dist = tsdf_volume_d_ptr->getTsdfTruncDist();
    if (shard_count == 1 && !force_shard_composite) {
        // the model-map pyramid rides in the raycast launch where it can (three levels, the level-0 model maps, one launch: the sign map's
        // form): ModelMapPyramid then has nothing to launch, and the tail's completion event rides on the raycast
        pyramid_in_raycast_ = false;
        const bool own_maps = &xyz_g_d == &vmaps_g_prev_d[0] && &normal_g_d == &nmaps_g_prev_d[0];
        if (raycast_builds_pyramid && own_maps && PreparePyramidLevels()) {
            ro.pyr_vmap1 = &vmaps_g_prev_d[1].ptr()->re; ro.pyr_nmap1 = &nmaps_g_prev_d[1].ptr()->re; ro.pyr_step1 = vmaps_g_prev_d[1].step();
            ro.pyr_vmap2 = &vmaps_g_prev_d[2].ptr()->re; ro.pyr_nmap2 = &nmaps_g_prev_d[2].ptr()->re; ro.pyr_step2 = vmaps_g_prev_d[2].step();
            ro.completion_event = tail_done_;
        }
        raycast(kinect_intrinsic, device_Rc2v, device_tc2v, device_Rv2w, device_tv2w, tsdf_volume_d_ptr->getTsdfTruncDist(), volume_res,
                voxel_size, tsdf_volume_d_ptr->value(), tsdf_volume_d_ptr->grad(), xyz_g_d, normal_g_d, hits_counter_, ray_ws_.ptr(), &ro);
        if (ro.pyramid_built) { pyramid_in_raycast_ = true; tail_recorded_ = true; }
        return 0;
    }
    // sharded: march this rank's planes, agree on the first event of every ray, add the winners
    const int res[3] = {volume_res.x, volume_res.y, volume_res.z};
    const int rows = xyz_g_d.rows() / 3, cols = xyz_g_d.cols();
    hipStream_t st = current_stream();
    DeviceArray2D<float> value = tsdf_volume_d_ptr->value(), grad = tsdf_volume_d_ptr->grad();
    check_rc(xs_raycast_slab_ex(&kinect_intrinsic.fx, &device_Rc2v.data[0].x.re, &device_tc2v.x.re, &device_Rv2w.data[0].x.re, &device_tv2w.x.re,
                                tsdf_volume_d_ptr->getTsdfTruncDist(), res, voxel_size, value.ptr(), grad.ptr(), value.step(), zs0, zs1, zo0, zo1,
                                &xyz_g_d.ptr()->re, &normal_g_d.ptr()->re, xyz_g_d.step(), rows, cols, ray_keys_.ptr(), &ro, st), "raycast");
    hipSafeCall(hipMemcpyAsync(ray_min_keys_.ptr(), ray_keys_.ptr(), (size_t)rows * cols * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (collective) collective(collective_user, 1, ray_min_keys_.ptr(), (long)rows * cols);
    // (what a rank receives: a ring all-reduce of S bytes over N ranks moves 2 (N - 1) / N x S through every rank, a gather the other ranks' parts)
    const double ring = shard_count > 1 ? 2.0 * (shard_count - 1) / shard_count : 0.0;
    composite_bytes_ += (long long)(ring * (double)rows * cols * sizeof(int));
    check_rc(xs_raycast_compose_mask(ray_keys_.ptr(), ray_min_keys_.ptr(), &xyz_g_d.ptr()->re, &normal_g_d.ptr()->re, xyz_g_d.step(), rows, cols, st), "raycast");
    if (collective && shard_composite_gather) {
        // Owner-compacted exchange: every rank packs the pixels it owns (52 bytes each), the counts travel as an int32 sum with one non-zero
        // entry per rank, the packs are gathered at the offsets every rank derives from the counts, and a scatter writes them into the maps.
        // One host wait per frame (the counts); half the bytes of adding the maps (a ring all-reduce moves every pixel's 48 bytes twice).
        const int n_ranks = shard_count;
        const size_t eb = xs_raycast_compose_entry_bytes();
        if (!gather_buf_.ptr()) {
            gather_buf_.create((size_t)rows * cols * eb);       // every pixel has at most one owner: the gathered packs fit
            pack_buf_.create((size_t)rows * cols * eb);
            gather_counts_.create((size_t)n_ranks);
            hipSafeCall(hipHostMalloc((void **)&gather_counts_host_, (size_t)n_ranks * sizeof(int)));
        }
        hipSafeCall(hipMemsetAsync(gather_counts_.ptr(), 0, (size_t)n_ranks * sizeof(int), st));
        check_rc(xs_raycast_compose_pack(ray_keys_.ptr(), ray_min_keys_.ptr(), &xyz_g_d.ptr()->re, &normal_g_d.ptr()->re, xyz_g_d.step(), rows, cols,
                                         pack_buf_.ptr(), gather_counts_.ptr() + shard_rank, st), "raycast");
        collective(collective_user, 2, gather_counts_.ptr(), n_ranks);
        hipSafeCall(hipMemcpyAsync(gather_counts_host_, gather_counts_.ptr(), (size_t)n_ranks * sizeof(int), hipMemcpyDeviceToHost, st));
        hipSafeCall(hipStreamSynchronize(st));
        std::vector<long long> desc((size_t)n_ranks + 2);
        desc[0] = (long long)(size_t)gather_buf_.ptr();
        desc[1] = 0;
        for (int r = 0; r < n_ranks; ++r) desc[(size_t)r + 2] = desc[(size_t)r + 1] + (long long)gather_counts_host_[r] * (long long)eb;
        const long long total = desc[(size_t)n_ranks + 1];
        if (total > (long long)rows * cols * (long long)eb) { std::cout << "error::KinectFusionReconstruction, composite: more owned pixels than pixels" << std::endl; exit(-1); }
        const long long mine = desc[(size_t)shard_rank + 2] - desc[(size_t)shard_rank + 1];
        if (mine > 0)
            hipSafeCall(hipMemcpyAsync(gather_buf_.ptr() + desc[(size_t)shard_rank + 1], pack_buf_.ptr(), (size_t)mine, hipMemcpyDeviceToDevice, st));
        collective(collective_user, 3, desc.data(), n_ranks);   // gather-v: afterwards every rank holds every rank's part
        composite_bytes_ += total - mine + (long long)(ring * n_ranks * sizeof(int));
        check_rc(xs_raycast_compose_scatter(gather_buf_.ptr(), (long)(total / (long long)eb), &xyz_g_d.ptr()->re, &normal_g_d.ptr()->re, xyz_g_d.step(), rows, cols, st), "raycast");
    } else if (collective) {
        const long words = (long)(xyz_g_d.step() / 4) * xyz_g_d.rows();
        const bool adjacent = normal_g_d.step() == xyz_g_d.step() && normal_g_d.rows() == xyz_g_d.rows() &&
                              (const char *)normal_g_d.ptr() == (const char *)xyz_g_d.ptr() + xyz_g_d.step() * (size_t)xyz_g_d.rows();
        if (adjacent) collective(collective_user, 2, xyz_g_d.ptr(), 2 * words);  // 14.7 MB, one collective
        else {
            collective(collective_user, 2, xyz_g_d.ptr(), words);
            collective(collective_user, 2, normal_g_d.ptr(), words);
        }
        composite_bytes_ += (long long)(ring * 2.0 * words * 4.0);
    }
    check_rc(xs_raycast_compose_finish(ray_min_keys_.ptr(), &xyz_g_d.ptr()->re, &normal_g_d.ptr()->re, xyz_g_d.step(), rows, cols,
                                       hits_counter_, st), "raycast");
    return 0;
}

// The six seeded volume-to-camera poses of one Gauss-Newton pass: v2c_k = inverse(se3Exp(i h e_k) * camera2volume), k = 0 .. 5.
// For a single seeded generator se3Exp takes its small-angle branch and is EXACTLY I + i h G_k (G_k: unit translation along axis k, or the hat matrix
// of axis k - 3), whose inverse is I - i h G_k up to a REAL term h^2 G_k^2 (1e-14: below the rounding of every entry it would touch).  So
//     v2c_k = v2c - i h (v2c G_k),        v2c = inverse(camera2volume) once,
// and v2c G_k is a column of v2c (translations) or two columns of its rotation swapped and signed (rotations): one 4x4 inverse and a few dozen products
// instead of six complex 4x4 products and six complex 4x4 cofactor inverses (4.9 -> 0.5 us on the build container's core; the host's side of a pass is
// what stands between two kernels).  The six poses share their real parts bit for bit by construction (the kernel counts a voxel only if every seeded
// evaluation keeps it); the imaginary parts equal those of the long form to rounding (tests/test_gauss_newton_gpu.py: the oracle twin, which inverts in
// double, and the analytic seeds of the per-pass test).
static void gn_seeded_poses(const xs_host::Matrix4cf &camera2volume, float R[6][18], float t[6][6]) {
    using namespace xs_host;
    const Matrix4cf v2c = inverse(camera2volume);
    const hostComplex ih(0.f, (float)H_);
    for (int k = 0; k < 6; ++k) {
        hostComplex Rk[3][3], tk[3];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Rk[i][j] = v2c.m[i][j]; tk[i] = v2c.m[i][3]; }
        if (k < 3) {
            for (int i = 0; i < 3; ++i) tk[i] = tk[i] - ih * v2c.m[i][k];                    // (v2c G_k): column 3 = column k of the rotation
        } else {
            const int a = k - 3, b = (a + 1) % 3, c = (a + 2) % 3;                           // hat(e_a): (c, b) = +1, (b, c) = -1
            for (int i = 0; i < 3; ++i) {
                Rk[i][b] = Rk[i][b] - ih * v2c.m[i][c];                                      // (R hat)(i, b) = R(i, c)
                Rk[i][c] = Rk[i][c] + ih * v2c.m[i][b];                                      // (R hat)(i, c) = -R(i, b)
            }
        }
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { R[k][(i * 3 + j) * 2] = Rk[i][j].real(); R[k][(i * 3 + j) * 2 + 1] = Rk[i][j].imag(); }
            t[k][2 * i] = tk[i].real(); t[k][2 * i + 1] = tk[i].imag();
        }
    }
}
// What every pass of a frame shares: the scaled depth (once per frame, not per pass) and this rank's owned planes as the dense array the
// kernel indexes (a pitched volume is packed first); the pinned record and the mailbox of the loop protocol.
const float *KinectFusionReconstruction::GaussNewtonPrepare(const DeviceArray2D<ushort> &depth_frame_d) {
    hipStream_t st = current_stream();
    DeviceArray2D<float> value = tsdf_volume_d_ptr->value();
    depthRawScaled_d.create(depth_frame_d.rows(), depth_frame_d.cols());
    check_rc(xs_scale_depth(depth_frame_d.ptr(), depth_frame_d.step(), depth_frame_d.rows(), depth_frame_d.cols(), depthRawScaled_d.ptr(),
                            depthRawScaled_d.step(), st), "scaleDepth");
    if (gn_sums_.size() < 32) gn_sums_.create(32);
    if (gn_ws_.size() < xs_tsdf_reduce_workspace_bytes()) {
        gn_ws_.create(xs_tsdf_reduce_workspace_bytes());
        check_rc(xs_tsdf_reduce_workspace_init(gn_ws_.ptr(), st), "reduce workspace");
    }
    if (!gn_publish_) {
        hipSafeCall(hipHostMalloc((void **)&gn_publish_, xs_gn_publish_bytes(), hipHostMallocCoherent | hipHostMallocMapped));
        std::memset(gn_publish_, 0, xs_gn_publish_bytes());
    }
    if (!gn_mailbox_ && gn_post_pose) check_rc(xs_icp_mailbox_alloc(&gn_mailbox_, &gn_mailbox_in_device_), "Gauss-Newton mailbox");
    const size_t row_bytes = (size_t)volume_resolution[0] * sizeof(float), plane_rows = (size_t)volume_resolution[1];
    const float *gt = reinterpret_cast<const float *>(reinterpret_cast<const char *>(value.ptr()) + (size_t)(zo0 - zs0) * plane_rows * value.step());
    if (value.step() != row_bytes) {
        const size_t rows = (size_t)(zo1 - zo0) * plane_rows;
        if (gn_dense_.size() < rows * volume_resolution[0]) gn_dense_.create(rows * volume_resolution[0]);
        hipSafeCall(hipMemcpy2DAsync(gn_dense_.ptr(), row_bytes, gt, value.step(), row_bytes, rows, hipMemcpyDeviceToDevice, st));
        gt = gn_dense_.ptr();
    }
    return gt;
}
// One pass enqueued: the kernel over the owned planes (poses as arguments, or — R null — from the mailbox with number mail_seq), in shard
// mode the all-reduce of the 29 sums on the stream and then their publication; single GPU: the kernel's last workgroup publishes.  The host
// reads the record with GaussNewtonWait(seq).
void KinectFusionReconstruction::GaussNewtonEnqueue(const DeviceArray2D<ushort> &depth_frame_d, const float *gt, const float (*R)[18], const float (*t)[6],
                                                    unsigned mail_seq, unsigned long long seq) {
    hipStream_t st = current_stream();
    const int res[3] = {volume_resolution[0], volume_resolution[1], volume_resolution[2]};
    const bool sharded = shard_count > 1 && collective;
    xs_gn_opts o = {};
    o.struct_bytes = sizeof(o);
    o.pose_mailbox = R ? nullptr : gn_mailbox_; o.mailbox_seq = mail_seq;
    if (!sharded) { o.publish_host = gn_publish_; o.publish_seq = seq; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profiling && gn_events_.size() < 64) {   // (level 1 or 2: the kernel's own duration, for bench.py's host_us_per_pass)
        hipSafeCall(hipEventCreate(&e0)); hipSafeCall(hipEventCreate(&e1));
        gn_events_.push_back({e0, e1});
        hipSafeCall(hipEventRecord(e0, st));
    }
    check_rc(xs_tsdf_gauss_newton_terms_ex(depthRawScaled_d.ptr(), depthRawScaled_d.step(), depth_frame_d.rows(), depth_frame_d.cols(), &kinect_intrinsic.fx,
                                           res, voxel_size, R ? &R[0][0] : nullptr, t ? &t[0][0] : nullptr, tsdf_volume_d_ptr->getTsdfTruncDist(), gt, zo0,
                                           zo1, gn_ws_.ptr(), gn_sums_.ptr(), &o, st), "GaussNewtonTerms");
    if (e1) hipSafeCall(hipEventRecord(e1, st));
    if (sharded) {
        collective(collective_user, 0, gn_sums_.ptr(), 29);
        if (gn_publish_sharded) check_rc(xs_gn_publish_sums(gn_sums_.ptr(), 29, gn_publish_, seq, st), "GaussNewtonTerms");
        else {   // (YAML gn_publish_sharded: false — the round-5 way: copy + stream drain, then the record is filled by the host itself)
            hipSafeCall(hipMemcpyAsync(gn_publish_, gn_sums_.ptr(), 29 * sizeof(double), hipMemcpyDeviceToHost, st));
            hipSafeCall(hipStreamSynchronize(st));
            reinterpret_cast<volatile unsigned long long *>(gn_publish_)[32] = seq;
        }
    }
}
// spins on the record's sequence word; false if the launch reported that it left without summing (abandoned, or its poses never came)
bool KinectFusionReconstruction::GaussNewtonWait(unsigned long long seq, double out29[29]) {
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(gn_publish_) + 32;
    unsigned long long seen;
    long spins = 0;
    while ((seen = *flag) != seq) {
        if (seen == (seq | (1ull << 63))) return false;
        if (++spins > 4000000000L) { std::cout << "error::KinectFusionReconstruction, Gauss-Newton pass never published its sums" << std::endl; exit(-1); }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const double ih = 1.0 / (double)(float)H_;
    for (int i = 0; i < 21; ++i) out29[i] = gn_publish_[i] * ih * ih;
    for (int i = 21; i < 27; ++i) out29[i] = gn_publish_[i] * ih;
    out29[27] = gn_publish_[27]; out29[28] = gn_publish_[28];
    return true;
}
void KinectFusionReconstruction::GaussNewtonCollectEvents() {
    for (auto &e : gn_events_) {
        float ms = 0.f;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { gn_kernel_ms += ms; ++gn_kernel_calls; }
        (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    gn_events_.clear();
}

// BASELINE config 5 (see the header): one pass of xs_tsdf_gauss_newton_terms for the six seeded poses
int KinectFusionReconstruction::GaussNewtonTerms(const DeviceArray2D<ushort> &depth_frame_d, const Matrix4cf &camera2volume, double out29[29]) {
    if (!tsdf_volume_d_ptr) return 0;
    const float *gt = GaussNewtonPrepare(depth_frame_d);
    float R[6][18], t[6][6];
    gn_seeded_poses(camera2volume, R, t);
    const unsigned long long seq = ++gn_seq_;
    GaussNewtonEnqueue(depth_frame_d, gt, R, t, 0u, seq);
    const bool ok = GaussNewtonWait(seq, out29);
    GaussNewtonCollectEvents();
    return ok ? 1 : 0;
}

// The Gauss-Newton loop with the ICP loop's protocol (round 6): the depth is scaled once per frame; a pass's sums reach the host through a
// pinned record the kernel's last workgroup writes (no copy, no stream drain); and pass n + 1 is in the queue BEFORE the host waits for pass
// n — its kernel resident, polling a mailbox for the six poses the host posts after the solve — so what stands between two kernels is the
// record's way to the host, the 6x6 solve, six pose inversions and one posted write (YAML gn_post_pose, default true; single GPU with a mailbox
// in device memory — a sharded rank enqueues pass n + 1 after the solve, its all-reduce on the stream in front of the publication).
int KinectFusionReconstruction::RelocalizeGaussNewton(const DeviceArray2D<ushort> &depth_frame_d, Matrix4cf &camera2volume, int iterations,
                                                      float damping, std::vector<double> *loss_history) {
    if (!tsdf_volume_d_ptr) return 0;
    const int passes = iterations + (loss_history ? 1 : 0);   // (the last one only reports the loss the loop ended at)
    if (passes <= 0) return 1;
    const float *gt = GaussNewtonPrepare(depth_frame_d);
    const bool ahead = gn_post_pose && shard_count == 1 && gn_mailbox_ && gn_mailbox_in_device_;
    float R[6][18], t[6][6];
    gn_seeded_poses(camera2volume, R, t);
    unsigned long long seq = ++gn_seq_;
    GaussNewtonEnqueue(depth_frame_d, gt, R, t, 0u, seq);
    int rc = 1;
    const auto t_begin = std::chrono::steady_clock::now();
    int done = 0;
    for (int p = 0; p < passes; ++p) {
        unsigned long long next_seq = 0;
        unsigned next_mail = 0;
        if (ahead && p + 1 < passes) {
            next_seq = ++gn_seq_; next_mail = ++gn_mail_seq_;
            if (next_mail == 0) next_mail = ++gn_mail_seq_;   // (0 is the mailbox's initial content)
            GaussNewtonEnqueue(depth_frame_d, gt, nullptr, nullptr, next_mail, next_seq);
        }
        auto leave = [&](int code) {   // the loop ends here: a launch that is waiting for its poses is told to leave, and has left before its buffers are reused
            if (next_seq) {
                xs_gn_post_poses(gn_mailbox_, nullptr, nullptr, next_mail, 1);
                double ignore[29];
                (void)GaussNewtonWait(next_seq, ignore);
            }
            rc = code;
        };
        double s[29];
        if (!GaussNewtonWait(seq, s)) { leave(0); break; }
        ++done;
        if (p > 0 && ahead) { gn_poll_us += gn_publish_[30] * 0.01; ++gn_poll_passes; }   // (this pass was enqueued ahead: what its kernel waited for its poses, 100 MHz ticks)
        if (loss_history) loss_history->push_back(s[28] > 0 ? s[27] / s[28] : 0.0);
        if (p == iterations) break;                           // the final loss pass
        if (s[28] < 6) { leave(0); break; }                   // nothing to align to
        double A[36], b[6], x[6];
        int q = 0;
        for (int j = 0; j < 6; ++j)
            for (int k = j; k < 6; ++k, ++q) { A[j * 6 + k] = s[q]; A[k * 6 + j] = s[q]; }
        for (int k = 0; k < 6; ++k) { A[k * 6 + k] *= 1.0 + (double)damping; b[k] = -s[21 + k]; }
        if (!solve_spd6(A, b, x)) { leave(0); break; }
        hostComplex xi[6];
        for (int k = 0; k < 6; ++k) xi[k] = hostComplex((float)x[k], 0.f);
        camera2volume = se3Exp(xi) * camera2volume;
        if (p + 1 < passes) {
            gn_seeded_poses(camera2volume, R, t);
            if (next_seq) { xs_gn_post_poses(gn_mailbox_, &R[0][0], &t[0][0], next_mail, 0); seq = next_seq; }
            else { seq = ++gn_seq_; GaussNewtonEnqueue(depth_frame_d, gt, R, t, 0u, seq); }
        }
    }
    gn_pass_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    gn_passes += done;
    GaussNewtonCollectEvents();
    return rc;
}

// reference :334-372
KinectFusionReconstruction::CPointCloud KinectFusionReconstruction::ExportPointCloud(int max_buffer) {
    CPointCloud res;
    if (max_buffer <= 0 || !tsdf_volume_d_ptr) return res;
    DeviceArray<float3> cloud_buffer, normal_buffer;
    cloud_buffer.create(max_buffer);
    normal_buffer.create(max_buffer);
    int3 volume_res;
    volume_res.x = volume_resolution.x();
    volume_res.y = volume_resolution.y();
    volume_res.z = volume_resolution.z();
    // a rank of a sharded run reports the crossings of the planes it owns (the +z neighbour of its last
    // plane is in its halo); the single-GPU case is the whole volume
    const int z1 = std::min(zo1, volume_res.z - 1);
    PtrSz<float3> cloud; cloud.data = cloud_buffer.ptr(); cloud.size = (size_t)max_buffer;
    const size_t num_points = extractPoints(tsdf_volume_d_ptr->value(), tsdf_volume_d_ptr->weight(), tsdf_volume_d_ptr->grad(), volume_res,
                                            voxel_size, cloud, zs0, zo0, std::max(z1, zo0));
    if (num_points == 0) return res;
    cloud.size = num_points;
    PtrSz<float3> normal; normal.data = normal_buffer.ptr(); normal.size = num_points;
    extractNormals(tsdf_volume_d_ptr->value(), tsdf_volume_d_ptr->weight(), tsdf_volume_d_ptr->grad(), volume_res, voxel_size, cloud, normal,
                   zs0, zs1);
    res.positions.resize(3 * num_points);
    res.normals.resize(3 * num_points);
    hipSafeCall(hipMemcpy(res.positions.data(), cloud_buffer.ptr(), num_points * sizeof(float3), hipMemcpyDeviceToHost));
    hipSafeCall(hipMemcpy(res.normals.data(), normal_buffer.ptr(), num_points * sizeof(float3), hipMemcpyDeviceToHost));
    return res;
}
bool KinectFusionReconstruction::CPointCloud::exportPly(const std::string &filename) const {
    std::ofstream file_out{filename};
    if (!file_out.is_open()) return false;
    file_out << "ply\nformat ascii 1.0\ncomment Created by myself\nelement vertex " << size() << "\n";
    file_out << "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\nend_header\n";
    for (size_t i = 0; i < size(); ++i)
        file_out << positions[3 * i] << " " << positions[3 * i + 1] << " " << positions[3 * i + 2] << " " << normals[3 * i] << " "
                 << normals[3 * i + 1] << " " << normals[3 * i + 2] << "\n";
    return true;
}

void KinectFusionReconstruction::synchronize() { hipSafeCall(hipStreamSynchronize(current_stream())); }

long long KinectFusionReconstruction::lastUpdatedVoxels() { return last_frame_counter(0); }
long long KinectFusionReconstruction::lastRaycastHits() { return last_frame_counter(1); }
long long KinectFusionReconstruction::last_frame_counter(int which) {
    if (counter_frame_ == 0) return 0;
    unsigned long long h[2] = {0, 0};
    synchronize();
    flush_pending_fold(current_stream());
    hipSafeCall(hipStreamSynchronize(current_stream()));
    hipSafeCall(hipMemcpy(h, counters_.ptr() + 2 * (size_t)((counter_frame_ - 1) % COUNTER_RING), sizeof(h), hipMemcpyDeviceToHost));
    return (long long)h[which];
}

// ---- volume checkpoint --------------------------------------------------------------------
// reference :438-447 writes raw float32 values; here X*Y*Z of them (the reference's count uses
// res[2] twice)
void KinectFusionReconstruction::saveTSDFVolume(const std::string &tsdf_filename) {
    std::vector<float> tsdf;
    tsdf_volume_d_ptr->downloadTSDFWithoutGrad(tsdf);
    std::ofstream f(tsdf_filename, std::ios::binary);
    f.write(reinterpret_cast<const char *>(tsdf.data()), (std::streamsize)(tsdf.size() * sizeof(float)));
}
namespace {
// Volume checkpoint, version 2.  Layout: header, n_poses x Matrix4cf, then value / grad / weight of the stored planes
// [zs0, zs1) as dense rows of X elements (a rank of a sharded run saves and restores its own planes).
struct CkptHeader {
    char magic[8];
    int res[3];
    float voxel_size, tranc_dist;
    int frame_id, n_poses;
    int zs0, zs1;        // planes held in this file
    int shard_rank, shard_count;
};
const int CKPT_MAX_POSES = 1 << 24;   // a sanity bound on the pose record (16 M frames), not a format limit
// one device <- host copy of a dense array into the EXISTING pitched buffer (no reallocation, no temporaries)
template <class T>
void restore_rows(DeviceArray2D<T> dst, const std::vector<T> &src, int cols, size_t rows) {
    hipSafeCall(hipMemcpy2D(dst.ptr(), dst.step(), src.data(), (size_t)cols * sizeof(T), (size_t)cols * sizeof(T), rows, hipMemcpyHostToDevice));
}
}  // namespace
void KinectFusionReconstruction::saveCheckpoint(const std::string &filename) {
    std::vector<float> v, g;
    std::vector<int> w;
    tsdf_volume_d_ptr->downloadTSDFWithGrad(v, g);
    tsdf_volume_d_ptr->downloadWeight(w);
    CkptHeader h{};
    std::snprintf(h.magic, sizeof(h.magic), "XSTSDF2");
    for (int i = 0; i < 3; ++i) h.res[i] = volume_resolution[i];
    h.voxel_size = voxel_size; h.tranc_dist = tsdf_volume_d_ptr->getTsdfTruncDist(); h.frame_id = frame_id;
    h.n_poses = (int)world2camera_record.size();
    h.zs0 = zs0; h.zs1 = zs1; h.shard_rank = shard_rank; h.shard_count = shard_count;
    std::ofstream f(filename, std::ios::binary);
    f.write(reinterpret_cast<const char *>(&h), sizeof(h));
    f.write(reinterpret_cast<const char *>(world2camera_record.data()), (std::streamsize)(h.n_poses * sizeof(Matrix4cf)));
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * 4));
    f.write(reinterpret_cast<const char *>(g.data()), (std::streamsize)(g.size() * 4));
    f.write(reinterpret_cast<const char *>(w.data()), (std::streamsize)(w.size() * 4));
}
// Nothing of *this is touched until the whole file has been read and validated: magic, volume geometry (resolution,
// voxel size, truncation distance), the planes it holds against the planes this instance stores, a sane pose count and
// the exact file length.  Returns false (state unchanged) on any mismatch.
// The sign map from the volume alone (allocation, checkpoint, anything that wrote the value array without the integrate kernels).
void KinectFusionReconstruction::RebuildSignMap() {
    if (!sign_map_on() || !sign_map_.ptr() || !tsdf_volume_d_ptr) return;
    const int res[3] = {volume_resolution.x(), volume_resolution.y(), volume_resolution.z()};
    DeviceArray2D<float> value = tsdf_volume_d_ptr->value();
    check_rc(xs_signmap_rebuild_slab(sign_map_.ptr(), res, raycast_sign_map_shift, tsdf_volume_d_ptr->getTsdfTruncDist(), value.ptr(0), value.step(),
                                     zs0, zs1, current_stream()), "sign map");
}

bool KinectFusionReconstruction::loadCheckpoint(const std::string &filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f || !tsdf_volume_d_ptr) return false;
    f.seekg(0, std::ios::end);
    const long long file_bytes = (long long)f.tellg();
    f.seekg(0, std::ios::beg);
    CkptHeader h{};
    if (file_bytes < (long long)sizeof(h)) return false;
    f.read(reinterpret_cast<char *>(&h), sizeof(h));
    if (!f || std::memcmp(h.magic, "XSTSDF2", 8) != 0) return false;
    for (int i = 0; i < 3; ++i) if (h.res[i] != volume_resolution[i]) return false;
    if (h.voxel_size != voxel_size || h.tranc_dist != tsdf_volume_d_ptr->getTsdfTruncDist()) return false;
    if (h.zs0 != zs0 || h.zs1 != zs1 || h.shard_rank != shard_rank || h.shard_count != shard_count) return false;
    if (h.n_poses < 1 || h.n_poses > CKPT_MAX_POSES || h.frame_id < 0) return false;
    const int X = h.res[0];
    const size_t rows = (size_t)h.res[1] * (size_t)(h.zs1 - h.zs0), n = rows * (size_t)X;
    const long long expect = (long long)sizeof(h) + (long long)h.n_poses * (long long)sizeof(Matrix4cf) + 3LL * (long long)n * 4LL;
    if (file_bytes != expect) return false;            // truncated or trailing bytes
    std::vector<Matrix4cf> poses((size_t)h.n_poses);
    f.read(reinterpret_cast<char *>(poses.data()), (std::streamsize)(poses.size() * sizeof(Matrix4cf)));
    std::vector<float> v(n), g(n);
    std::vector<int> w(n);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * 4));
    f.read(reinterpret_cast<char *>(g.data()), (std::streamsize)(n * 4));
    f.read(reinterpret_cast<char *>(w.data()), (std::streamsize)(n * 4));
    if (!f) return false;
    DeviceArray2D<float> dv = tsdf_volume_d_ptr->value(), dg = tsdf_volume_d_ptr->grad();
    DeviceArray2D<int> dw = tsdf_volume_d_ptr->weight();
    if ((size_t)dv.rows() != rows || dv.cols() != X || (size_t)dg.rows() != rows || (size_t)dw.rows() != rows) return false;
    // validated: commit
    synchronize();
    restore_rows(dv, v, X, rows);
    restore_rows(dg, g, X, rows);
    restore_rows(dw, w, X, rows);
    RebuildSignMap();    // the volume was written behind the integrate kernels' back
    world2camera_record.swap(poses);
    world2camera = world2camera_record.back();
    frame_id = h.frame_id;
    // previous-frame maps are derived state: regenerate them from the restored volume and pose (in a sharded run every
    // rank must load its own file before the next frame: the raycast composite is a collective)
    CalculatePointCloud(vmaps_g_prev_d[0], nmaps_g_prev_d[0]);
    ModelMapPyramid();
    synchronize();
    return true;
}

// ---- per-stage HIP event timing (on the stream the kernels are launched on) -----------------
void KinectFusionReconstruction::set_profiling(int level) {
    const bool on = level > 0;
    if (on && prof_ring_.empty()) {
        prof_ring_.resize(PROF_RING);
        for (auto &slot : prof_ring_)
            for (int s = 0; s < ST_COUNT; ++s) {
                hipSafeCall(hipEventCreate(&slot.ev[s][0]));
                hipSafeCall(hipEventCreate(&slot.ev[s][1]));
                slot.used[s] = false;
            }
    }
    if (profiling) collect_stage_times();
    profiling = on;
    profiling_stages = level > 1;
}
void KinectFusionReconstruction::stage_begin(int st) {
    if (!profiling_stages) return;
    hipSafeCall(hipEventRecord(prof_ring_[prof_pending_].ev[st][0], current_stream()));
}
void KinectFusionReconstruction::stage_end(int st) {
    if (!profiling_stages) return;
    hipSafeCall(hipEventRecord(prof_ring_[prof_pending_].ev[st][1], current_stream()));
    prof_ring_[prof_pending_].used[st] = true;
}
void KinectFusionReconstruction::end_profiled_frame() {
    // nothing is enqueued here: the frame's counters stay in their ring slot until the pending frames are folded
    if (++prof_pending_ == PROF_RING) collect_stage_times();
}
void KinectFusionReconstruction::collect_stage_times() {
    flush_pending_fold(current_stream());
    synchronize();
    if (prof_pending_ > 0)
        hipSafeCall(hipMemcpy(pinned_counters_, counters_.ptr(), COUNTER_RING * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int f = 0; f < prof_pending_; ++f) {
        // pending frame f is the (prof_pending_ - f)-th most recent integrated frame
        const long long fr = counter_frame_ - prof_pending_ + f;
        if (fr >= 0) {
            cum_updated += (long long)pinned_counters_[2 * (fr % COUNTER_RING)];
            cum_hits += (long long)pinned_counters_[2 * (fr % COUNTER_RING) + 1];
        }
        for (int s = 0; s < ST_COUNT; ++s) {
            if (!prof_ring_[f].used[s]) continue;
            float ms = 0.f;
            hipSafeCall(hipEventElapsedTime(&ms, prof_ring_[f].ev[s][0], prof_ring_[f].ev[s][1]));
            stage_ms[s] += ms;
            stage_calls[s] += 1;
            prof_ring_[f].used[s] = false;
        }
    }
    prof_pending_ = 0;
}
