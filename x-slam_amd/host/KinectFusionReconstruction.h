// KinectFusionReconstruction.h — host orchestrator of the XKinectFusion CSFD pipeline on one
// MI355X.  Same class name, public fields and method names as the reference
// (XKinectFusion/include/KinectFusionReconstruction.h:19-174), so a caller of the reference's
// class (Experiments/test_xkinect_fusion/main.cpp:38-58) drives this one the same way:
//     KinectFusionReconstruction kinfu;  kinfu.SetYamlParameters(config);
//     kinfu.ProcessFrame(depth_frame_d);  kinfu.world2camera_record.back();
// Host algebra comes from host_algebra.hpp (no Eigen), config from flat_yaml.hpp (no yaml-cpp).
//
// Additions: the CSFD seed is a parameter (keys csfd_seed_row / csfd_seed_col / csfd_seed_h; the
// reference has the seed line commented out, KinectFusionReconstruction.cpp:22); per-stage HIP
// event timing; counters of updated voxels and raycast hits; volume checkpoint save / load.
#pragma once
#include "TsdfVolume.h"
#include "flat_yaml.hpp"
#include <chrono>
#include <string>
#include <vector>

class KinectFusionReconstruction {
public:
    typedef xs_host::Matrix3cf Matrix3frm;  // row-major complex 3x3
    typedef xs_host::Matrix4cf Matrix4cf;
    typedef xs_host::Vector3cf Vector3cf;
    enum Stage { ST_SURFACE = 0, ST_ICP, ST_SCALE, ST_INTEGRATE, ST_RAYCAST, ST_RESIZE, ST_COUNT };

    xs_host::FlatYaml config;
    int num_levels = 3;
    Matrix4cf world2camera;
    Matrix4cf world2volume;
    std::vector<Matrix4cf> world2camera_record;
    int frame_id = 0;
    int frame_step = 1;
    std::vector<Matrix4cf> gt_poses{};  // camera-to-world; complex so a CSFD seed can ride on a given pose

    Intr kinect_intrinsic;
    int depth_width = 0, depth_height = 0;

    Vector3i volume_resolution;
    float voxel_size{};
    TsdfVolume *tsdf_volume_d_ptr = nullptr;
    int max_integration_weight = 0;

    int icp_iterations[3]{};
    float distThres{};
    float angleThres{};
    DeviceArray2D<devComplexICP> g_buf;  // kept for signature compatibility; unused by the fused reduction
    DeviceArray<devComplexICP> sum_buf;

    float trunc_logistic_k = 0.f;
    float biInterpolate_threshold{0.005f};

    std::vector<DeviceArray2D<devComplex>> depths_curr_d, vmaps_curr_d, nmaps_curr_d, vmaps_g_prev_d, nmaps_g_prev_d;
    DeviceArray2D<float> depthRawScaled_d;
    // the current-frame maps' real parts as float planes (their imaginary parts are zeros: the depth image is real) — what the
    // ICP reduction reads (half the bytes); YAML icp_real_current_maps, default true
    std::vector<DeviceArray2D<float>> vreal_curr_d, nreal_curr_d;
    bool icp_real_current_maps = true;
    // the integrate call's brick classification enqueued behind the last ICP launch, with the pose that launch starts from (YAML
    // integrate_classify_ahead, default true; single GPU with the posted ICP loop): IntegrateFrame then finds the list ready
    bool integrate_classify_ahead = true;
    // ... on the auxiliary stream, beside that launch instead of behind it (YAML integrate_classify_beside_icp, default FALSE).  Measured, no
    // gain: a level-0 ICP launch fills the chip (one 16-wave workgroup per CU, all of its registers and 158 of 160 KB of LDS), so the two
    // classification kernels do not run beside it but in front of its workgroups — the launch they share the chip with takes 26 us
    // instead of 17.7 — and an integrate launch that has to wait for another stream's event starts 10 us later than one that follows its
    // own queue (profiles/r04_ab_classify_beside_icp.txt).
    bool integrate_classify_beside_icp = false;
    // the classification at the frame's start, for the pose a constant-velocity model predicts, on the auxiliary stream (YAML
    // integrate_classify_predicted, default FALSE: measured, no gain on the benchmark scene; not with integrate_post_pose or ground-truth
    // poses): see ClassifyPredicted
    bool integrate_classify_predicted = false;
    bool list_predicted_ = false;              // the list that is ready was classified for a predicted pose (this frame's SurfaceMeasure)
    int integrate_classify_early = 0;          // YAML integrate_classify_early: that many ICP iterations before the last (a pose that many more updates old)
    long long list_cover_counts_[4] = {0, 0, 0, 0};   // what xs_integrate_list_covers said of the lists classified ahead (xs_kf_list_cover_counts)
    // the integrate kernel itself enqueued behind that classification, handed the final pose through a mailbox of its own and a one-wave gate
    // kernel (YAML integrate_post_pose, default FALSE; needs integrate_classify_ahead and a mailbox in device memory).  Round 3, first form: all
    // three launches went in one ICP iteration early (YAML integrate_post_early) — 16 % of the frames were then not covered by the planes of a
    // pose two updates old at slack 2 and fell back: +0.7 % (profiles/r03_ab_integrate_post.txt).  As it stands they go in while the last ICP launch
    // runs, with the pose that launch starts from (as the classification alone does): 4.7 % of the frames still fail the posted launch's stricter
    // coverage test and the kernel starts ~8 us earlier on the others: +0.8 % frames/s, inside the noise (profiles/r03_ab_integrate_post_late.txt)
    // — still off by default; xs_kf_posted_integrate_counts says how many launches were given their pose and how many were told to leave.
    bool integrate_post_pose = false;
    long long composite_bytes_ = 0;   // bytes this rank received through the raycast composite's collectives so far (ring all-reduce: 2 (N - 1) / N x size; gather: the other ranks' parts)
    long long posted_accepted_ = 0, posted_refused_ = 0;   // posted integrate launches that were given their pose / told to leave (xs_kf_posted_integrate_counts)
    bool integrate_post_early = false;       // YAML integrate_post_early: the posted launch goes in one ICP iteration earlier (list from a pose two updates old)
    float integrate_classify_slack = 2.0f;   // YAML integrate_classify_slack: how much wider than its own the list's frustum slack is (1 = every frame falls back)
    // The sign map of the ray march (include/xslam_amd.h, csrc/xs_signmap.h; YAML raycast_sign_map, default true; raycast_sign_map_shift, default 0 =
    // the finest bricks the march can use: 8^3 voxels at 512^3): the integrate kernel marks the bricks it writes negative values into, the march starts every ray at the first
    // step that can end it — the same maps bit for bit (tests/test_signmap_gpu.py), about a fifth of the volume reads.  A rank of a sharded
    // volume keeps a map of the planes it stores (owned slab + halo) and its slab march evaluates the iterations that map leaves.
    bool raycast_sign_map = true;
    int raycast_sign_map_shift = 0;
    void RebuildSignMap();   // call after writing the value array through xs_kf_volume_ptr
    // the value array's address has been handed out (xs_kf_volume_ptr): whoever holds it may write the volume behind the integrate kernels'
    // back, after which the sign map is no longer a superset — it is rebuilt from the volume in front of the next raycast
    void MarkSignMapStale() { sign_map_stale_ = true; }
    bool sign_map_stale_ = false;
    // Look-ahead of the map preparation (no counterpart in the reference, whose main.cpp:50-58 reads, uploads and processes one frame at a
    // time): the caller names the depth image it will pass to the NEXT ProcessFrame call — device memory, unchanged until then — and that
    // frame's bilateral filter and depth pyramid are built during this frame's ICP loop.  A ProcessFrame call with any other image simply
    // prepares its own maps as always.  Same results bit for bit (tests/test_pipeline_gpu.py).
    void HintNextFrame(const ushort *depth_dev, size_t step_bytes) {
        if (next_stage_ != 0) EnqueueAnnouncedFrame(true);   // (an announcement half enqueued is finished first: never expected)
        next_hint_ptr_ = depth_dev; next_hint_step_ = step_bytes;
    }
    void EnqueueAnnouncedFrame(bool all);
    int next_stage_ = 0;
    void EnqueueMapsFromPyramid();
    void EnqueueScale(const DeviceArray2D<ushort> &depth_frame_d);
    void SwapMapSets();
    bool list_ready_ = false;
    float list_Rv2c_[18]{}, list_tv2c_[6]{};
    void ClassifyAhead(const Matrix3frm &Rcurr, const Vector3cf &tcurr);
    void ClassifyPredicted();
    void SetListPose(const Matrix4cf &c2w);
    void EnqueueClassification(hipStream_t st, bool with_event);
    bool real_maps_valid_ = false;

    bool use_gtPose = false;

    // CSFD seed: i*h on world2camera(row, col); row < 0 disables
    int csfd_seed_row = -1, csfd_seed_col = -1;
    float csfd_seed_h = 1e-7f;

    // z-slab / pixel-row sharding across GPUs (no counterpart in the single-GPU
    // reference).  Rank r of G owns z planes [zo0, zo1) and stores [zs0, zs1) = owned + halo; it
    // integrates its own storage, evaluates the ICP rows [r*H/G, (r+1)*H/G) of each level and the
    // march steps of every ray that land in its planes.  collective(user, op, dev_ptr, count)
    // must all-reduce in place over the ranks on the current stream: op 0 = sum of doubles,
    // 1 = min of int32, 2 = sum of int32.
    typedef void (*collective_fn)(void *user, int op, void *dev_ptr, long count);
    enum { HALO = 6 };
    int shard_rank = 0, shard_count = 1;
    int zo0 = 0, zo1 = 0, zs0 = 0, zs1 = 0;
    collective_fn collective = nullptr;
    void *collective_user = nullptr;
    void SetSharding(int rank, int count, collective_fn fn, void *user);  // call before SetYamlParameters

    // false (default): the reference's shape, one host solve per ICP iteration (the host spins on a
    // completion word the kernel writes into pinned memory; with icp_post_pose below the next launch is
    // already resident when the solve ends: ICP stage 0.24 ms).  true: the pose update runs on the device
    // (xs_icp_iterate) and the host waits once per frame — slower on this machine (a serial
    // double-precision Cholesky + substitution is ~9000 cycles for one wave, plus a second launch per
    // iteration: 0.35 ms), kept for hosts that cannot spin.  YAML key icp_solve_on_device.
    bool icp_solve_on_device = false;
    // true (default): each iteration's launch is enqueued while the previous one is still running and
    // receives its pose through a pinned-memory mailbox once the host has solved for it
    // (xs_icp_accumulate_posted / xs_icp_post_pose): the launch latency leaves the per-iteration
    // turnaround.  Same kernel arithmetic, same poses.  false: launch after the solve (the reference's
    // order).  Applies when this rank evaluates whole images (not with icp_shard_rows).  YAML key
    // icp_post_pose.
    bool icp_post_pose = true;
    int icp_lookahead = 1;   // posted launches kept in the queue ahead of the one the host waits for (>= 1; more measured the same)
    // The final addition of an ICP reduction on the host (xs_icp_accumulate_records / xs_icp_sum_records): every
    // workgroup writes its record of 55 partial sums straight into pinned host memory and leaves; the host — which is
    // spinning for the result anyway — adds the records in index order.  Takes the cross-XCD gather (write-back, ticket,
    // last workgroup reading every record from memory) out of the launch.  Measured no faster than the device gather
    // (profiles/r02_ab_icp_fold.txt: 2 270-2 540 against 2 530-2 550 frames/s — 90-512 workgroups each pushing 448 bytes
    // and a system-scope release over PCIe cost what the agent-scope publish + final sum cost), so it is off by default.
    // YAML key icp_host_fold.
    bool icp_host_fold = false;
    // Every ICP sum reaches the host as one 16-byte store {sequence number, sum} (XS_ICP_PUBLISH_PAIRS, include/xslam_amd.h): no completion word that
    // has to be ordered behind the sums — the kernel's last workgroup skips the wait for its stores' acknowledgement, a barrier and a release store
    // per launch.  YAML key icp_publish_pairs; the launches of the host-solve loop only (not icp_host_fold, not the device solve).
    bool icp_publish_pairs = true;
    // Sharded runs (SetSharding): false (default) — every rank evaluates the whole ICP itself; all ranks hold
    // the same current-frame maps and the composited previous-frame maps, the reduction is deterministic,
    // so they reach the same pose bit for bit with no collective inside the ICP loop.  true — pixel rows
    // are split across ranks and the 55 sums are all-reduced every iteration: 12
    // latency-bound collectives + host waits per frame to save < 0.1 ms of kernel time.  YAML key
    // icp_shard_rows.
    bool icp_shard_rows = false;
    // test aid: take the slab raycast + composite (and its collectives) even with a single rank
    bool force_shard_composite = false;

    // instrumentation
    bool profiling = false;
    double stage_ms[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    long long stage_calls[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    std::vector<double> icp_log;  // per ICP iteration of the last frame: 54 sums + inliers
    long long cum_updated = 0, cum_hits = 0;  // summed over profiled frames (read back with the stage times)

    KinectFusionReconstruction();
    ~KinectFusionReconstruction();

    int getFrame() const { return frame_id; }
    int getVolumeSize() const { return volume_resolution[0] * volume_resolution[1] * volume_resolution[2]; }
    Matrix4cf getCamera2Volume() { return world2volume * xs_host::inverse(world2camera); }

    void AllocateBuffers();
    void ReleaseBuffers();
    void SetYamlParameters(const xs_host::FlatYaml &config_);
    int ProcessFrame(const DeviceArray2D<ushort> &depth_frame_d);
    int ProcessFrameHost(const ushort *depth_host);  // dense rows of depth_width u16; pinned or pageable
    ushort *IngestBuffer();                          // next host-pinned staging buffer (decode into it, then ProcessFrameHost(it))
    void SurfaceMeasure(const DeviceArray2D<ushort> &depth_frame_d);
    int PoseEstimate(Matrix3frm Rcurr, Vector3cf tcurr, Matrix3frm Rprev_inv, Vector3cf tprev);
    int AlignDepthToReconstruction(const DeviceArray2D<ushort> &depth_frame_d, bool use_LM = false);
    int IntegrateFrame(const DeviceArray2D<ushort> &depth_frame_d);
    static int SmoothDepthFrame(MapArr &dst_d, const DeviceArray2D<ushort> &src_d);
    int CalculatePointCloud(MapArr &xyz_g_d, MapArr &normal_g_d);
    void ModelMapPyramid();

    static inline Vector3cf GetTranslation(Matrix4cf &trans_mat) { return xs_host::GetTranslation(trans_mat); }
    static inline Matrix3frm GetRotation(Matrix4cf &trans_mat) { return xs_host::GetRotation(trans_mat); }

    // Pose refinement against the map by Gauss-Newton on the TSDF residual of ComputeLocalTsdf_hessian, with the
    // Jacobian from first-order CSFD: six poses seeded with i*h along the generators of a left perturbation
    // camera2volume <- se3Exp(xi) * camera2volume, all evaluated in one pass over the volume
    // (xs_tsdf_gauss_newton_terms).  BASELINE config 5; the reference only has the commented single-direction
    // ComputeTSDF_loss / ComputeTSDF_hessian (:374-434).  A sharded rank evaluates its slab and the 29 sums are
    // all-reduced.  GaussNewtonTerms fills {JtJ upper triangle (21), Jtr (6), sum r^2, count}, already divided by h.
    int GaussNewtonTerms(const DeviceArray2D<ushort> &depth_frame_d, const Matrix4cf &camera2volume, double out29[29]);
    int RelocalizeGaussNewton(const DeviceArray2D<ushort> &depth_frame_d, Matrix4cf &camera2volume, int iterations, float damping,
                              std::vector<double> *loss_history = nullptr);
    // the loop protocol of the Gauss-Newton passes (see RelocalizeGaussNewton): YAML gn_post_pose, default true
    bool gn_post_pose = true;
    bool gn_publish_sharded = true;   // shard mode: the all-reduced sums reach the host through a publish kernel + pinned record (false: copy + stream drain)
    double gn_pass_us = 0, gn_kernel_ms = 0;       // wall clock of the passes RelocalizeGaussNewton ran (from its first kernel enqueued to its last sums seen) / kernel durations of the profiled passes
    long long gn_passes = 0, gn_kernel_calls = 0;
    double gn_poll_us = 0;             // summed over the passes enqueued ahead: what the resident kernel waited for its poses (its own 100 MHz clock): the host's side of a pass
    long long gn_poll_passes = 0;
    const float *GaussNewtonPrepare(const DeviceArray2D<ushort> &depth_frame_d);
    void GaussNewtonEnqueue(const DeviceArray2D<ushort> &depth_frame_d, const float *gt, const float (*R)[18], const float (*t)[6], unsigned mail_seq,
                            unsigned long long seq);
    bool GaussNewtonWait(unsigned long long seq, double out29[29]);
    void GaussNewtonCollectEvents();

    // ExportPointCloud (reference :334-372, main.cpp:78-80): zero-crossing points of the TSDF with
    // normals, at most max_buffer of them; a sharded rank exports the planes it owns.
    struct CPointCloud {
        std::vector<float> positions, normals;   // xyz triples
        size_t size() const { return positions.size() / 3; }
        bool exportPly(const std::string &filename) const;  // CPointCloud.cpp:42-67: ascii, x y z nx ny nz
    };
    CPointCloud ExportPointCloud(int max_buffer);

    // volume checkpoint: raw float32 value (+ grad, + int32 weight), X*Y*Z each, dense
    // (the reference's saveTSDFVolume writes value only and res[0]*res[2]*res[2] floats,
    // KinectFusionReconstruction.cpp:438-447)
    void saveTSDFVolume(const std::string &tsdf_filename);
    void saveCheckpoint(const std::string &filename);
    bool loadCheckpoint(const std::string &filename);

    // counters of the last frame (synchronise the stream)
    long long lastUpdatedVoxels();
    long long lastRaycastHits();
    long long last_frame_counter(int which);
    void synchronize();

private:
    // Per-frame diagnostics counters ([0] updated voxels, [1] raycast hits) live in a device ring, one slot
    // per frame; half of the ring is cleared by one fill every COUNTER_RING / 2 frames instead of one fill
    // (and, when profiling, one download) per frame on the critical stream.
    enum { COUNTER_RING = 512 };
    DeviceArray<unsigned long long> counters_;  // [COUNTER_RING][2]
    long long counter_frame_ = 0;               // frames integrated so far = next slot to use
    unsigned long long *frame_counters() { return counters_.ptr() + 2 * (size_t)(counter_frame_ % COUNTER_RING); }
    unsigned long long *hits_counter_ = nullptr;  // set while IntegrateFrame runs the raycast
    DeviceArray<float> depth_max_;              // [0] largest valid depth of the frame (scale kernel -> integrate far clip)
    DeviceArray<unsigned char> integrate_ws_;  // brick work list of the integrate kernel
    DeviceArray<unsigned char> icp_ws_;        // per-workgroup partial records of the ICP reduction
    DeviceArray<double> icp_sums_;             // 27 complex sums + inlier count
    DeviceArray<unsigned char> icp_pose_;      // device-resident pose of the ICP loop (xs_icp_iterate)
    DeviceArray<double> gn_sums_;              // 29 Gauss-Newton sums (+ pad)
    double *gn_publish_ = nullptr;             // pinned: the 29 sums + sequence word a pass publishes (xs_gn_publish_bytes)
    void *gn_mailbox_ = nullptr;               // the six-pose mailbox of the passes enqueued ahead (xs_gn_post_poses)
    int gn_mailbox_in_device_ = 0;
    unsigned long long gn_seq_ = 0;
    unsigned gn_mail_seq_ = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gn_events_;
    DeviceArray<float> gn_dense_;              // packed copy of the owned planes when the volume is pitched
    DeviceArray<unsigned char> gn_ws_;         // reduce workspace of the Gauss-Newton / Hessian kernels
    DeviceArray2D<ushort> depth_ingest_d_;     // device copy of a host frame (ProcessFrameHost)
    ushort *ingest_pinned_[2] = {nullptr, nullptr};
    hipEvent_t ingest_done_[2] = {nullptr, nullptr};
    int ingest_seq_ = 0;
    DeviceArray<unsigned char> maps_prev0_block_;  // level-0 model vertex + normal maps, one allocation
    DeviceArray<float> ray_ws_;                // raycast: crossing time per pixel (march kernel -> crossing kernel)
    hipEvent_t tail_done_ = nullptr;           // completion of a frame's model-map pyramid (rides on its dispatch)
    bool tail_recorded_ = false;
    // the second set of per-frame buffers: the announced next frame's maps are built into it (HintNextFrame, SwapMapSets)
    std::vector<MapArr> depths_next_d, vmaps_next_d, nmaps_next_d;
    std::vector<DeviceArray2D<float>> vreal_next_d, nreal_next_d;
    DeviceArray2D<float> depthRawScaled_next_d;
    DeviceArray<float> depth_max_next_;
    // the model-map pyramid inside the raycast launch (YAML raycast_builds_pyramid, default true; single GPU, three levels, sign map on)
    bool raycast_builds_pyramid = true;
    int profile_integrate_every = 4;   // YAML profile_integrate_every: at profiling level 1 the integrate kernel's event pair rides on every n-th frame (IntegrateKernelTimedThisFrame)
    bool IntegrateKernelTimedThisFrame() const;
    bool pyramid_in_raycast_ = false;   // this frame's raycast launch built levels 1 and 2: ModelMapPyramid has nothing to do
    bool PreparePyramidLevels();
    // shard mode, raycast composite by owner-compacted exchange (YAML shard_composite_gather, default true; false = the int32 sum of the maps)
    bool shard_composite_gather = true;
    DeviceArray<unsigned char> gather_buf_, pack_buf_;   // the ranks' packed owned pixels (52 bytes each), all of them / this rank's
    DeviceArray<int> gather_counts_;                     // [rank] = owned pixels with a vertex
    int *gather_counts_host_ = nullptr;                  // pinned
    DeviceArray<unsigned char> depth_tiles_, depth_tiles_next_;   // per-tile depth range of the frame (xs_scale_depth_tiles -> the integrate call's brick classes)
    hipEvent_t surface_done_next_ = nullptr, scale_done_next_ = nullptr;
    bool real_maps_valid_next_ = false, scale_recorded_next_ = false;
    const void *next_hint_ptr_ = nullptr, *next_ready_ptr_ = nullptr;
    size_t next_hint_step_ = 0, next_ready_step_ = 0;
    bool next_ready_ = false;
    DeviceArray<unsigned char> sign_map_;      // raycast: bricks that may hold a negative voxel (single GPU)
    bool sign_map_on() const { return raycast_sign_map; }   // (every rank of a sharded volume keeps one for the planes it stores)
    unsigned char *sign_map_ptr() { return sign_map_on() ? sign_map_.ptr() : nullptr; }
    DeviceArray<int> ray_keys_, ray_min_keys_; // sharded raycast: first-event keys (own, agreed)
    // host-coherent: [0..54] sums + count, [56] completion sequence word, [64..80) pose state of the
    // device-side loop, [128 + 64*n ..) the 55 values of its iteration n
    enum { ICP_LOG_MAX = 62, PINNED_DOUBLES = 128 + 64 * ICP_LOG_MAX };
    double *pinned_sums_ = nullptr;
    unsigned long long *pinned_pairs_ = nullptr;   // XS_ICP_PAIRS_BYTES of host-coherent pinned memory (icp_publish_pairs)
    // the integrate call's header clear / count fold taken off the main stream (SurfaceMeasure, IntegrateFrame)
    bool integrate_split() const;
    void flush_pending_fold(hipStream_t st);
    bool integrate_header_clear_ = false;
    unsigned long long *pending_fold_ = nullptr;
    double *pinned_records_ = nullptr;   // xs_icp_records_bytes() of host-coherent pinned memory (icp_host_fold)
    void *icp_mailbox_ = nullptr;              // pose mailbox of the posted ICP launches (xs_icp_mailbox_alloc)
    void *integrate_mailbox_ = nullptr;        // ... and the posted integrate launch's own (never rewritten while its kernel may still poll)
    int integrate_mailbox_in_device_ = 0;
    unsigned integrate_mail_seq_ = 0;
    DeviceArray<unsigned> posted_pose_;        // {command, 24 floats}: the gate kernel's hand-over to the posted integrate launch
    bool posted_pending_ = false;              // a posted integrate launch is in the stream, waiting for its pose
    std::chrono::steady_clock::time_point posted_at_{};   // when it was enqueued (a pose that comes too late is not posted: the gate may have given up)
    unsigned posted_seq_ = 0;
    hipEvent_t posted_stop_ = nullptr;         // its completion event
    bool posted_split_ = false;
    void EnqueuePostedIntegrate();
    long long counters_prepared_for_ = -1;     // the frame whose counter slot has been prepared (ring half cleared)
    unsigned long long *PrepareFrameCounters(hipStream_t st);
    int icp_mailbox_in_device_ = 0;
    unsigned long long icp_seq_ = 0;
public:
    // host wall clock of the ICP loop per pyramid level: from the completion of one iteration's sums to the completion of the next one's
    // (kernel + completion word over PCIe + solve + post) — SURVEY 8(d): "ICP: report us per iteration".  Always on: two clock reads.
    double icp_level_us[4] = {0, 0, 0, 0};          // [3]: the frame's first iteration, which also waits for the stream to drain the previous frame's tail
    long long icp_level_calls[4] = {0, 0, 0, 0};
    // host clock of the frame's tail (xs_kf_tail_host_times), microseconds summed since the last reset: [0] the last ICP sums seen ->
    // IntegrateFrame entered (solve, pose algebra), [1] entered -> the integrate launch call (transforms, cover test), [2] that call itself
    // (xs_integrate_scaled_ex2), [3] its return -> the raycast launch's return
    double tail_host_us[4] = {0, 0, 0, 0};
    long long tail_host_calls = 0;
    std::chrono::steady_clock::time_point t_last_sums_{};
    // test aids: start the launch sequence numbers at `v` (the mailbox is told); make the determinant gate fail at iteration n of the
    // next PoseEstimate (-1: off)
    void DebugSetIcpSequence(unsigned long long v);
    int debug_fail_icp_iteration_ = -1;
    int debug_post_delay_us_[2] = {0, 0};   // test aid: a random host sleep of [min, max] microseconds in front of every pose post (a slow host)
    unsigned debug_post_rng_ = 12345u;
private:
    void AbandonClassifiedList();
    void WaitForClassification(hipStream_t st);
    hipEvent_t classify_done_ = nullptr;       // completion of ClassifyAhead's launches on the auxiliary stream (rides on the last dispatch)
    bool classify_recorded_ = false;
    hipStream_t aux_stream_ = nullptr;         // surface measure of frame k+1 runs here, under raycast / pyramid of frame k
    hipEvent_t surface_done_ = nullptr, integrate_done_ = nullptr, scale_done_ = nullptr;
    hipEvent_t integrate_done_now_ = nullptr;   // the event that marks the last integrate call's completion (integrate_done_, or its dispatch's stop event)
    bool integrate_recorded_ = false, scale_recorded_ = false;
    bool profiling_icp_sync = false;           // true: copy + stream synchronise instead of the spin (debug aid)
    int PoseEstimateOnDevice(Matrix3frm Rcurr, Vector3cf tcurr, const Matrix3frm &Rprev_inv, const Vector3cf &tprev, Matrix4cf c2w_curr,
                             int total_iters);
    void icp_normal_equations(const MatS33 &Rcurr, const devComplex3 &tcurr, const MatS33 &Rprev_inv, const devComplex3 &tprev, int level,
                              hostComplexICP *A, hostComplexICP *b, long long *inliers);
    // deferred profiling: one slot of events + one pinned counter record per frame, folded into
    // stage_ms / cum_* only when somebody asks (no per-frame synchronisation)
    enum { PROF_RING = COUNTER_RING / 2 };
    struct ProfSlot { hipEvent_t ev[ST_COUNT][2]; bool used[ST_COUNT]; };
    std::vector<ProfSlot> prof_ring_;
    int prof_pending_ = 0;
    unsigned long long *pinned_counters_ = nullptr;  // [COUNTER_RING][2]: the ring, downloaded when the pending frames are folded
    void stage_begin(int st);
    void stage_end(int st);
    void end_profiled_frame();
public:
    void collect_stage_times();  // synchronises and folds the pending frames
    // level 0: off.  1: the integrate kernel's own event pair + the counters (what bench.py's timed region
    // needs; nothing extra on the stream between kernels).  2: an event pair around every stage as well —
    // each record is a packet the next kernel has to queue behind, so stage times come from their own pass.
    void set_profiling(int level);
    bool profiling_stages = false;
private:
};
