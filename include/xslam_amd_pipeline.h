/* xslam_amd_pipeline.h — C ABI of the host orchestrator (libxslam_host.so): the C++ class
 * KinectFusionReconstruction (x-slam_amd/host/, mirroring
 * XKinectFusion/include/KinectFusionReconstruction.h:19-174) behind an opaque handle, for
 * callers that are not C++ (bench.py, the parity tests).  Config is the reference's flat YAML
 * (Experiments/test_xkinect_fusion/configs/ICL_traj2.yaml) passed as text, plus the optional
 * keys csfd_seed_row / csfd_seed_col / csfd_seed_h.
 * 4x4 complex matrices cross as 32 floats, row-major (re, im) pairs.  Errors inside the
 * pipeline print and exit(-1), as the reference's cudaSafeCall does (Common/include/cx.h:124-130).
 */
#ifndef XSLAM_AMD_PIPELINE_H
#define XSLAM_AMD_PIPELINE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Host DoubleComplex (x-slam_amd/host/DoubleComplex.h, mirroring DeviceArray/include/DoubleComplex.h:15-95)
 * over arrays of n groups (re.re, re.im, im.re, im.im).  op: 0 add 1 sub 2 mul 3 div 4 sqrt 5 abs 6 exp
 * 7 log 8 sin 9 cos 10 pow(x, y.re.re) 11 f1(x, y) = (x + y)^2 12 conj 13 norm 14 (x > y, x < y).  CPU only. */
int xs_host_double_complex_table(int op, long n, const float *a, const float *b, float *out);
/* Host instantiation of the complex<float> operators (x-slam_amd/csrc/xs_complex.h is host + device, like
 * DeviceArray/include/cuda_complex.hpp:12-16): n interleaved (re, im) pairs, op codes of xs_complex_table
 * (include/xslam_amd.h).  Returns 0, -1 for a bad op.  CPU only. */
int xs_host_complex_table(int op, long n, const float *a, const float *b, float *out);

/* Flat "key: value" config reader (x-slam_amd/host/flat_yaml.hpp; stands in for yaml-cpp's
 * config["k"].as<T>() of KinectFusionReconstruction.cpp:12-72): copies the value of key into out.
 * Returns its length, -1 if the key is absent.  CPU only. */
int xs_flat_yaml_get(const char *yaml_text, const char *key, char *out, int capacity);

/* hipStream_t (as void*) every later call enqueues on; NULL = default stream */
void xs_kf_set_stream(void *stream);
/* KinectFusionReconstruction() + SetYamlParameters(config)     KinectFusionReconstruction.cpp:4-73
 * returns NULL (and prints) on a missing key */
void *xs_kf_create(const char *yaml_text);
/* Sharded instance (the reference is single-GPU): rank r of count owns the z planes
 * [r*Z/count, (r+1)*Z/count) (+ 6 halo planes each side) and the pixel rows [r*H/count, ...) of
 * every ICP level.  collective(user, op, dev_ptr, count) must all-reduce dev_ptr in place over
 * the ranks, ordered on the current stream: op 0 = sum of doubles (the 27 complex<double>
 * normal-equation sums + inlier count), 1 = min of int32 (first raycast event per pixel),
 * 2 = sum of int32 (the owned-pixel counts of the raycast composite; with shard_composite_gather: false the vertex / normal maps as bit
 * patterns), 3 = gather of variable-size parts: dev_ptr is then a HOST array of count + 2 64-bit words — [0] the device buffer's address,
 * [1 + r] .. [2 + r] the byte range of rank r's part in it — this rank's part is in place when the call is made, every rank's when the
 * operation has completed on the stream (count = the number of ranks; the parts are whole 52-byte entries).
 * xs_kf_download_volume then returns the stored planes only (xs_kf_shard_planes). */
void *xs_kf_create_sharded(const char *yaml_text, int rank, int count, void (*collective)(void *user, int op, void *dev_ptr, long n),
                           void *user);
void xs_kf_shard_planes(void *kf, int *owned2, int *stored2);
void xs_kf_destroy(void *kf);
/* gt_poses (camera-to-world) for flag_use_gtPose: n matrices of 32 floats   .h:36, .cpp:239-247 */
void xs_kf_set_gt_poses(void *kf, int n, const float *c2w32);
/* ProcessFrame(const DeviceArray2D<ushort>& depth_frame_d)     .cpp:147-159
 * depth_dev: u16 millimetres resident in device memory, row pitch step_bytes.  Returns 1, or 0
 * when the alignment failed (frame_id is then not advanced, as in the reference). */
int xs_kf_process_frame(void *kf, const uint16_t *depth_dev, size_t step_bytes);
/* Look-ahead (no reference counterpart: main.cpp:50-58 handles one frame at a time): the device depth image the NEXT xs_kf_process_frame
 * call will be given — it must stay unchanged until that call.  Its maps (bilateral filter, depth pyramid, vertex / normal maps, scaled depth) are then built
 * during the current frame's ICP loop, when the GPU is mostly idle, into a second set of buffers.  A call with any other image prepares its own maps as always; results are the same bits
 * either way.  Call before xs_kf_process_frame of the current frame. */
void xs_kf_hint_next_frame(void *kf, const uint16_t *depth_dev, size_t step_bytes);
/* the reference demo's upload + ProcessFrame (main.cpp:50-58): host buffer, dense rows.  The frame is copied
 * asynchronously on the pipeline's second stream (through a pinned staging buffer unless depth_host is
 * itself pinned), where the map preparation follows it without a host wait. */
int xs_kf_process_frame_host(void *kf, const uint16_t *depth_host);
/* the next of two host-pinned staging buffers (depth_width x depth_height u16): decode a frame straight into
 * it and pass it to xs_kf_process_frame_host — no staging copy.  Blocks only if the copy out of that buffer
 * two frames ago has not finished. */
uint16_t *xs_kf_ingest_buffer(void *kf);
/* getCamera2Volume()  .h:137: world2volume * world2camera^-1 of the latest pose, 32 floats */
void xs_kf_get_camera2volume(void *kf, float *out32);
/* Pose refinement against the map (BASELINE config 5): Gauss-Newton on the residual of ComputeLocalTsdf_hessian
 * with the Jacobian from first-order CSFD — six poses seeded with i*1e-7 along the generators of
 * camera2volume <- se3Exp(xi) * camera2volume, one pass over the volume (xs_tsdf_gauss_newton_terms); a sharded
 * rank evaluates its slab and the sums are all-reduced.  gauss_newton_terms: out29 = J^T J upper triangle (21),
 * J^T r (6), sum r^2, count.  relocalize: `iterations` steps of (J^T J + damping * diag) delta = -J^T r from c2v32
 * (updated in place); loss_out (optional, iterations + 1 doubles): mean squared residual before each step and at
 * the end.  Both return 1, or 0 when there is nothing to align to. */
int xs_kf_gauss_newton_terms(void *kf, const uint16_t *depth_dev, size_t step_bytes, const float *c2v32, double *out29);
int xs_kf_relocalize(void *kf, const uint16_t *depth_dev, size_t step_bytes, float *c2v32, int iterations, float damping, double *loss_out);
/* ExportPointCloud(max_buffer)  .cpp:334-372 (+ CPointCloud::exportPly, main.cpp:78-80): zero-crossing points of
 * the TSDF with normals, at most max_buffer; xyz triples into the host arrays (either may be NULL); returns
 * the number of points.  A sharded rank exports the planes it owns.  export_ply writes the reference's
 * ascii format (x y z nx ny nz) and returns the number of points, -1 if the file cannot be opened. */
long long xs_kf_export_point_cloud(void *kf, int max_buffer, float *points_host, float *normals_host);
long long xs_kf_export_ply(void *kf, int max_buffer, const char *filename);
void xs_kf_synchronize(void *kf);

int xs_kf_frame_id(void *kf);
int xs_kf_num_poses(void *kf);                                   /* world2camera_record.size() */
void xs_kf_get_world2camera(void *kf, int idx, float *out32);    /* idx < 0 counts from the back */
float xs_kf_tranc_dist(void *kf);
long long xs_kf_last_updated_voxels(void *kf);                   /* U of the last integrate */
long long xs_kf_last_raycast_hits(void *kf);
/* per ICP iteration of the last frame: 54 sums (27 x re, im) + inlier count; returns #doubles */
int xs_kf_icp_log(void *kf, double *out, int capacity);

/* dense host copies (X*Y*Z each; any pointer may be NULL)   TsdfVolume.cpp:64-77 */
int xs_kf_download_volume(void *kf, float *value, int *weight, float *grad);
/* which: 0 depths_curr 1 vmaps_curr 2 nmaps_curr 3 vmaps_g_prev 4 nmaps_g_prev; out: planes x
 * rows x cols x (re, im), dense */
int xs_kf_download_map(void *kf, int which, int level, float *out);
/* device pointers + pitch of the live arrays: which 0 value 1 weight 2 grad.  Asking for the value array (which 0) marks the ray march's
 * sign map stale: it is rebuilt from the volume in front of the next raycast (one pass over the value array), so a caller that writes
 * values through the pointer needs no further call; a caller that only reads pays that pass once per request. */
void *xs_kf_volume_ptr(void *kf, int which, size_t *step_bytes);

/* HIP-event timing.  level 0: off.  1: the integrate kernel's own event pair (stage 3) and the per-frame
 * counters — nothing extra on the stream between kernels.  2: an event pair around every stage too
 * (0 surface 1 icp 2 scale 3 integrate 4 raycast 5 resize); every record is a packet the next kernel
 * queues behind, so a frame is ~10 % slower at level 2 than at level 1. */
void xs_kf_set_profiling(void *kf, int level);
void xs_kf_stage_times(void *kf, double *ms6, long long *calls6);
/* voxels written / raycast hits summed over the frames processed with profiling on */
void xs_kf_cumulative_counters(void *kf, long long *updated, long long *hits);
void xs_kf_reset_stage_times(void *kf);
/* Host wall clock of the ICP loop per pyramid level (index = level, 0 = full resolution; index 3 = the first iteration of each frame,
 * kept apart because it also waits for the stream to drain the previous frame's tail): microseconds summed over the iterations run since
 * the last xs_kf_reset_stage_times, and their count — the period from one iteration's sums arriving to the next one's (kernel, completion
 * word, 6x6 solve, pose post).  ICP.cu:395-417 is a launch + device sync + copy per iteration.  Always on. */
void xs_kf_icp_iteration_times(void *kf, double *us4, long long *calls4);
/* Host wall clock of a tracked frame's tail since the last xs_kf_reset_stage_times, microseconds summed over *frames frames: us4[0] the last ICP
 * sums seen -> IntegrateFrame entered (6x6 solve, pose algebra), [1] entered -> the integrate launch call (transforms, cover test), [2] the
 * launch call itself, [3] its return -> the raycast launch's return. */
void xs_kf_tail_host_times(void *kf, double *us4, long long *frames);
/* The Gauss-Newton passes xs_kf_relocalize has run since the last reset: *pass_us = host wall clock summed over *passes passes (from a loop's
 * first kernel enqueued to its last sums seen: kernel + the record's way to the host + the 6x6 solve + the six pose inversions + the post);
 * *kernel_ms / *kernel_calls = the kernels' own durations, of passes run with profiling on (xs_kf_set_profiling >= 1: an event pair around each).
 * reset != 0 zeroes the four.  pass - kernel is what the host's side of a pass costs (bench.py: workloads.reloc.host_us_per_pass). */
void xs_kf_gn_times(void *kf, double *pass_us, long long *passes, double *kernel_ms, long long *kernel_calls, int reset);
/* The host's side of a Gauss-Newton pass as the DEVICE sees it: *poll_us = microseconds the kernels that were enqueued ahead waited for their poses (from
 * resident — i.e. the previous pass's kernel gone — to poses seen, on the device's 100 MHz clock), summed over *poll_passes such passes since the last
 * reset: the record's way to the host, the solve, the six pose inversions, the post and its way back.  No events involved. */
void xs_kf_gn_poll_times(void *kf, double *poll_us, long long *poll_passes, int reset);
/* YAML gn_post_pose at run time (0: every Gauss-Newton pass launched with its poses as arguments, after the solve — a kernel's event pair then times
 * the kernel alone; a pass enqueued ahead also spends the host's turnaround polling its mailbox inside the pair) */
void xs_kf_set_gn_post_pose(void *kf, int on);
/* Test aids.  Start the ICP launch sequence numbers at v (exercises the 2^32 wrap of the mailbox numbers); make the determinant gate of
 * iteration n (0-based, over all levels) of the next alignment fail as for a singular system (KinectFusionReconstruction.cpp:203-210). */
void xs_kf_debug_set_icp_sequence(void *kf, unsigned long long v);
void xs_kf_debug_fail_icp_iteration(void *kf, int n);
/* test aid: a random host sleep of [min_us, max_us] microseconds in front of every ICP pose post (0, 0 = none): a slow host must neither
 * time a resident launch out nor change a pose */
void xs_kf_debug_post_delay(void *kf, int min_us, int max_us);
/* Rebuilds the sign map of the ray march (xslam_amd.h) from the volume now (in shard mode: from the planes this rank stores, on every
 * rank).  xs_kf_volume_ptr(kf, 0, .) schedules the same in front of the next raycast and loadCheckpoint does it itself, so this is
 * only for a caller that kept the pointer and wrote through it again later.  No-op with raycast_sign_map: false. */
void xs_kf_rebuild_sign_map(void *kf);
/* shard mode: bytes this rank has received through the raycast composite's collectives since creation (a ring all-reduce of S bytes over N
 * ranks counted as 2 (N - 1) / N x S, the gather of the owned pixels as the other ranks' parts) */
long long xs_kf_composite_bytes(void *kf);
/* integrate_post_pose: how many posted integrate launches were given their pose, and how many were told to leave because the final pose was
 * not covered by the planes they had been given (those frames took the plain call) */
void xs_kf_posted_integrate_counts(void *kf, long long *accepted, long long *refused);
/* How often the brick list and box classes decided ahead of a frame's final pose (behind its last ICP launch) held for that pose: counts4[0]
 * neither (everything classified again), [1] the list only (the boxes decided again with the final pose), [3] both. */
void xs_kf_list_cover_counts(void *kf, long long *counts4);

/* volume checkpoint (value + grad + weight + poses)   cf. saveTSDFVolume, .cpp:438-447 */
int xs_kf_save_checkpoint(void *kf, const char *path);
int xs_kf_load_checkpoint(void *kf, const char *path);
int xs_kf_save_tsdf_volume(void *kf, const char *path);

/* ---- The reference's own call shape (x-slam_amd/host/reference_shape.hpp) ---------------------------------------------------------
 * The per-frame sequence of Experiments/test_xkinect_fusion/main.cpp:46-60 + KinectFusionReconstruction.cpp:147-332 over nothing but the
 * reference-signature launchers (bilateralFilter, pyrDown, createVMap / createNMap, estimateCombined per iteration, integrateTsdfVolume,
 * raycast, resizeVMap / resizeNMap — x-slam_amd/host/xs_launchers.hpp) with the reference's argument lists and a stream drain wherever
 * the reference's launcher drains the device: what swapping the library and changing nothing else gives.  Same results as xs_kf_* bit
 * for bit (tests/test_reference_shape_gpu.py); bench.py times it as legs.reference_call_shape.  Config: the same flat YAML. */
void *xs_refshape_create(const char *yaml_text);
void xs_refshape_destroy(void *rs);
/* main.cpp:50-58: DeviceArray2D<ushort>::upload of a host frame (dense rows), then ProcessFrame */
int xs_refshape_process_frame_host(void *rs, const uint16_t *depth_host);
/* ProcessFrame on a frame already in device memory (main.cpp:57-60 times ProcessFrame alone) */
int xs_refshape_process_frame(void *rs, const uint16_t *depth_dev, size_t step_bytes);
int xs_refshape_frame_id(void *rs);
void xs_refshape_get_world2camera(void *rs, int idx, float *out32);
int xs_refshape_download_volume(void *rs, float *value, int *weight, float *grad);
int xs_refshape_download_map(void *rs, int which, int level, float *out);   /* as xs_kf_download_map */
/* per ICP iteration of the last frame: the 54 sums estimateCombined returned (no inlier count: ICP.h:24-31 has none); returns #doubles */
int xs_refshape_icp_log(void *rs, double *out, int capacity);
/* ComputeLocalTsdf_hessian / ComputeLocalTsdf_loss through their TsdfFusion.h:48-60 signatures (PtrStepSz<ushort> depth on the device,
 * MatD33 = 36 floats / devDComplex3 = 12 floats, or Mat33 = 9 / float3 = 3; gt_dev: X*Y*Z floats, dense).  out4 = the returned float4
 * {loss, gradient, second derivative, count}, out2 = float2 {loss, count}.  with_volumes: the per-voxel scratch volumes of the reference's
 * thrust vectors are filled too and copied to volumes_host (real | grad | hessian | count as int32: 4 x N^3 words; loss: real | count). */
int xs_refshape_hessian(const uint16_t *depth_dev, size_t depth_step, int rows, int cols, const float *intr4, const int *res3, float voxel_size,
                        const float *Rv2c36, const float *tv2c12, float tranc_dist, const float *gt_dev, int with_volumes, float *out4,
                        float *volumes_host);
int xs_refshape_loss(const uint16_t *depth_dev, size_t depth_step, int rows, int cols, const float *intr4, const int *res3, float voxel_size,
                     const float *Rv2c9, const float *tv2c3, float tranc_dist, const float *gt_dev, int with_volumes, float *out2,
                     float *volumes_host);

#ifdef __cplusplus
}
#endif
#endif
