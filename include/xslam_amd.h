/* xslam_amd.h — C ABI of the MI355X-native XKinectFusion CSFD hot path (libxslam_hip.so).
 *
 * Every entry point replaces one launcher function of the reference's kernel-launcher API
 * (XKinectFusion/include/{TsdfVolume.h,TsdfFusion.h,RayCaster.h,ICP.h,Map.h}); the original
 * signature each one stands in for is cited per function.  Plain pointers and sizes only:
 *
 *   - device pointers are raw HIP device addresses; `step` / `*_step` is the row pitch in
 *     BYTES, as in the reference's PtrStep<T> (DeviceArray/include/kernel_containers.hpp:53-55)
 *   - complex values are (re, im) float pairs; MatS33 = 9 pairs row-major (18 floats),
 *     devComplex3 = 3 pairs (6 floats), the layouts device_cast<> reinterprets
 *     (XKinectFusion/include/Internal.h:42-45, 63-65, 146-148); MatD33 / devDComplex3 are
 *     9 / 3 groups of (re.re, re.im, im.re, im.im) (cuda_double_complex.hpp:24-31)
 *   - Intr is float[4] = {fx, fy, cx, cy} (Internal.h:49-59); int3 is int[3] = {x, y, z}
 *   - a vertex/normal map is 3 stacked planes of rows x cols complex (Map.cu:23-25)
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue
 *     work and return; nothing synchronises unless the function says so.  The reference
 *     synchronises the device inside most launchers.
 *   - return value: 0 on success, otherwise the hipError_t code; xs_last_error() gives text.
 *     The reference prints and exit(-1)s on any CUDA error (Common/include/cx.h:125-130);
 *     the C++ shim above this ABI reproduces that.
 */
#ifndef XSLAM_AMD_H
#define XSLAM_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *xs_last_error(void);

/* floor(p / voxel_size) in the raycast march without a divide (a reciprocal product with one fused residual correction), used ONLY
 * for constants that xs_const_div_prepare has checked — on the device, every one of the 2^32 float operands against the divide (~2 ms
 * and one synchronisation per new constant; the verdict is kept in a process-wide table).  Returns bit 1: floor(short form) ==
 * floor(divide) for all |x| <= 2^60 (what the march uses), bit 0: the quotients are bit-identical for 2^-60 <= |x| <= 2^60 and +-0
 * (recorded only); 0: no device / constant outside [2^-20, 2^20].  xs_raycast called with a voxel_size that was never prepared
 * (xs_const_div_state == 0) brackets the quotient with two reciprocal products and divides where they disagree — same results.
 * The orchestrator prepares voxel_size in AllocateBuffers. */
unsigned xs_const_div_prepare(float c);
unsigned xs_const_div_state(float c);
int xs_const_div_enable(int on);   /* test aid: 0 = every launcher divides whatever has been prepared; returns the previous setting */
/* ABI version, bumped on any signature change */
int xs_abi_version(void);

/* ---- TSDF volume ---------------------------------------------------------------------- */
/* initVolume(PtrStep<short2>, PtrStep<float> value, PtrStep<int> weight, PtrStep<float> grad,
 *            const int3&)                                    TsdfVolume.h:16, TsdfFusion.cu:34-43
 * Zero-fills the z-slab [z0, z1) whose storage starts at the given pointers. */
int xs_init_volume(float *value, int *weight, float *grad, size_t step_bytes, const int *res, int z0, int z1, void *stream);

/* scaleDepthKernal launch inside integrateTsdfVolume        TsdfFusion.cu:68-82, :182-187
 * u16 millimetres -> float metres, 0 outside [200, 5000]. */
int xs_scale_depth(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step, void *stream);
/* Same, and max_dev (a device float the caller zeroed beforehand) receives the largest valid
 * depth of the frame, for xs_integrate_scaled's far clipping. */
int xs_scale_depth_max(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step,
                       float *max_dev, void *stream);
/* Same, and tiles_dev (xs_depth_tiles_bytes(rows, cols) bytes, or NULL) receives the smallest and the largest scaled depth of every
 * 8 x 8 pixel tile {float lo, hi} (an invalid pixel counts as 0).  No counterpart in the reference: the integrate kernel classifies
 * whole bricks from it (free space in front of every surface it can see: written with tsdf = (1, 0) without a projection; behind
 * everything it can see: skipped) and takes the reference's per-voxel path (TsdfFusion.cu:110-167) for the rest — same volume.
 * xs_integrate_opts.depth_tiles hands the table to an xs_integrate_scaled_ex2 / xs_integrate_classify_ex call (NULL: the call builds its
 * own from the scaled image, in its workspace); xs_depth_tiles builds it from an image that is already scaled. */
size_t xs_depth_tiles_bytes(int rows, int cols);
int xs_scale_depth_tiles(const uint16_t *depth, size_t depth_step, int rows, int cols, float *scaled, size_t scaled_step,
                         float *max_dev, void *tiles_dev, void *stream);
int xs_depth_tiles(const float *scaled, size_t scaled_step, int rows, int cols, void *tiles_dev, void *stream);

/* integrateTsdfVolume(const PtrStepSz<ushort>& depth, const Intr&, int max_weight, const int3& res,
 *     float voxel_size, const MatS33& Rv2c, const devComplex3& tv2c, const devComplex3& tc2v,
 *     float tranc_dist, PtrStep<float> value, PtrStep<int> weight, PtrStep<float> grad,
 *     DeviceArray2D<float>& depthScaled, int frame_id, float threshold, float k)
 *                                                            TsdfFusion.h:40-45, TsdfFusion.cu:173-201
 * tc2v, frame_id and k are unused by the reference kernel and are not part of this ABI.
 * depth_scaled: caller-owned rows x cols float workspace.  [z0, z1): the z-slab this device
 * owns (0, res[2] for the whole volume); value/weight/grad point at the slab's storage.
 * updated_dev: optional device counter incremented by the number of voxels written. */
int xs_integrate_tsdf_volume(const uint16_t *depth, size_t depth_step, int rows, int cols, const float *intr4, int max_weight,
                             const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist,
                             float *value, int *weight, float *grad, size_t vol_step, float *depth_scaled, size_t scaled_step,
                             float threshold, int z0, int z1, unsigned long long *updated_dev, void *stream);
/* Same, from an already scaled depth image (tsdfFusionKernal alone, TsdfFusion.cu:85-171).
 * depth_max_dev: optional device float = largest valid depth of the frame (xs_scale_depth_max);
 * lets a column stop behind the farthest surface.  NULL = walk the whole frustum. */
int xs_integrate_scaled(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                        const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist, float *value,
                        int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                        unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, void *stream);
/* xs_integrate_scaled with its two bracketing launches under the caller's control (both are tiny, but each is a dependent
 * dispatch on the stream: ~14 us of a 390 us frame).  flags: XS_INTEGRATE_HEADER_IS_CLEAR — the caller has cleared the
 * workspace header since the previous call (xs_integrate_workspace_clear, on any stream ordered before this call);
 * XS_INTEGRATE_NO_FOLD — the voxel count stays in the workspace (a word per workgroup, behind the header) until the caller folds it into updated_dev
 * (xs_integrate_fold_counts, ordered after this call and before the next clear).  updated_dev non-NULL still enables
 * the counting.  flags = 0 is xs_integrate_scaled. */
#define XS_INTEGRATE_HEADER_IS_CLEAR 1u
#define XS_INTEGRATE_NO_FOLD 2u
#define XS_INTEGRATE_ALWAYS_STORE 8u    /* store all three words of every updated voxel, also where their bits do not change (the default stores only words that change: same volume, fewer bytes) */
#define XS_INTEGRATE_POSE_POSTED 16u     /* the kernel takes its pose from a mailbox (xs_integrate_opts.pose_mailbox; posted with xs_icp_post_pose): see below */
#define XS_INTEGRATE_NO_TILES 32u        /* every brick takes the exact per-voxel walk: no free-space / nothing-to-write classification from the depth tiles (A/B and tests; same volume either way) */
#define XS_INTEGRATE_COUNT_CLASSES 64u   /* the kernel counts the wave-sized boxes it classified: 32-bit words 48 / 49 / 50 of the workspace = free / nothing to write / exact walk (cleared with the header; tests and bench figures) */
#define XS_INTEGRATE_RECLASSIFY_BOXES 128u /* with XS_INTEGRATE_LIST_IS_READY: the brick list holds for this pose but the box classes xs_integrate_classify left do not (xs_integrate_list_covers returned 1, not 3): classify the boxes again, with this pose */
#define XS_INTEGRATE_LIST_IS_READY 4u   /* xs_integrate_classify has produced the brick list on this stream (see there) */
int xs_integrate_scaled_ex(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                           const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist, float *value,
                           int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                           unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, unsigned flags, void *stream);
int xs_integrate_workspace_clear(void *workspace, void *stream);
int xs_integrate_fold_counts(void *workspace, unsigned long long *updated_dev, void *stream);
/* The brick classification of an integrate call ahead of the call, for a pose that is only NEARLY the final one (the orchestrator
 * enqueues it behind the last ICP launch with the pose that launch starts from: the list is there when the final pose is, and
 * the classification's launch latency leaves the frame's critical path).  The frustum's slack is multiplied by slack_scale (>= 1);
 * xs_integrate_list_covers (host only) says whether such a list holds every brick the final pose would list — if so, call
 * xs_integrate_scaled_ex with XS_INTEGRATE_LIST_IS_READY (| XS_INTEGRATE_HEADER_IS_CLEAR) on the same stream / workspace / slab /
 * image size, else clear the header again (xs_integrate_workspace_clear) and call it without.  flags of xs_integrate_classify:
 * XS_INTEGRATE_HEADER_IS_CLEAR as above.  Results are those of the plain call, bit for bit (the list is a superset; every voxel
 * still takes the exact tests with the final pose).
 * With a depth-tile table named (xs_integrate_opts.depth_tiles) xs_integrate_classify_ex also decides the boxes' classes (free space /
 * nothing to write / per-voxel walk), padded for every pose within the slack's allowances; xs_integrate_list_covers returns 0 (the
 * list does not hold), 1 (the list holds, the classes do not: add XS_INTEGRATE_RECLASSIFY_BOXES to the call's flags) or 3 (both hold). */
/* The integrate kernel enqueued before its pose exists (flag XS_INTEGRATE_POSE_POSTED, with XS_INTEGRATE_LIST_IS_READY | XS_INTEGRATE_HEADER_IS_CLEAR):
 * xs_integrate_opts.pose_mailbox / mailbox_seq / mailbox_slack / pose_dev name — for an xs_integrate_scaled_ex2 call — the mailbox
 * (xs_icp_mailbox_alloc: one of its own, not the ICP loop's) that a one-wave gate kernel in front of the integrate kernel polls, the sequence number
 * it waits for, the factor the call widens the frustum planes by, and 128 bytes of device memory (pose_dev) through which the gate hands the pose on; that call is given the pose the brick list was classified with (xs_integrate_classify) and uses it for the planes only.
 * When the final pose is known: if xs_integrate_pose_covered(..., list pose, slack_scale, final pose) post it with
 * xs_icp_post_pose(mailbox, Rv2c18, tv2c6, seq, 0) — the kernel then integrates with exactly that pose, same volume as a plain call — else post
 * xs_icp_post_pose(mailbox, NULL, NULL, seq, 1): the launch leaves without touching the volume, and a plain call follows.  Every posted launch must
 * be answered by exactly one post (it gives up after ~1 s otherwise).  Sequence numbers: non-zero, increasing per mailbox. */
int xs_integrate_pose_covered(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18_list,
                              const float *tv2c6_list, float slack_scale, const float *Rv2c18, const float *tv2c6);
int xs_integrate_classify(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6,
                          float tranc_dist, int z0, int z1, const float *depth_max_dev, void *workspace, float slack_scale, unsigned flags,
                          void *stream);
int xs_integrate_list_covers(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18_list,
                             const float *tv2c6_list, float slack_scale, const float *Rv2c18, const float *tv2c6);
/* Device workspace for xs_integrate_scaled's brick work list, for a slab of nz planes.  With a
 * workspace the kernel first lists the 64x4x8-voxel bricks that can intersect the frustum and
 * then spreads them over all CUs; without one (NULL) each column walks its own clipped range. */
size_t xs_integrate_workspace_bytes(const int *res, int nz);

/* ---- Options structs ---------------------------------------------------------------------------------------------------------------
 * Everything a call takes besides its arguments proper — depth-tile table, sign map, events, pose mailbox, pyramid outputs — travels in an
 * options struct that is an argument of the call: xs_integrate_scaled_ex2 / xs_integrate_classify_ex / xs_raycast_ex / xs_raycast_slab_ex /
 * xs_resize_pyramid_ex.  The library keeps NO per-thread state (ABI version 2: the xs_*_set_* functions of version 1 are gone), so nothing
 * a previous call set, or an early return forgot to clear, can leak into a call.  The entry points without a struct (xs_integrate_scaled*,
 * xs_integrate_classify, xs_raycast, xs_raycast_slab, xs_resize_pyramid) are the same calls with an empty one.
 * Zero-initialise a struct, set struct_bytes = sizeof(it), fill what applies.
 *   events      hipEvent_t riding on a kernel's own dispatch (its begin / end: no marker packets on the stream); a stop event alone is a completion
 *               event another stream can wait on.
 *
 * The library remembers, per WORKSPACE, what xs_integrate_classify* last classified into it (pose, slack, slab, volume, camera, band, tile
 * table): a following integrate call with XS_INTEGRATE_LIST_IS_READY on that workspace uses those box classes only if all of that is its
 * own and ITS pose lies within the slack they were padded for, and decides the boxes again with its own pose otherwise —
 * XS_INTEGRATE_RECLASSIFY_BOXES is a hint that saves the check, not a correctness requirement (the brick LIST itself is the caller's claim:
 * xs_integrate_list_covers).  Classifying on one host thread and integrating on another is fine. */
typedef struct xs_integrate_opts {
    unsigned struct_bytes;           /* sizeof(xs_integrate_opts) */
    unsigned flags;                  /* XS_INTEGRATE_* */
    const void *depth_tiles;         /* this frame's xs_scale_depth_tiles table, or NULL (the call builds its own in its workspace) */
    void *signmap;                   /* the sign map the call marks, or NULL */
    void *start_event, *stop_event;  /* hipEvent_t riding on the integrate kernel's dispatch (classify: stop_event = its last dispatch); NULL = none */
    const void *pose_mailbox;        /* XS_INTEGRATE_POSE_POSTED: the mailbox the gate polls ... */
    unsigned mailbox_seq;            /* ... the sequence number it waits for ... */
    float mailbox_slack;             /* ... the factor the call widens the list pose's frustum planes by ... */
    void *pose_dev;                  /* ... and 128 bytes of device memory through which the gate hands the pose on */
} xs_integrate_opts;
int xs_integrate_scaled_ex2(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                            const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist, float *value,
                            int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                            unsigned long long *updated_dev, const float *depth_max_dev, void *workspace, const xs_integrate_opts *opts, void *stream);
int xs_integrate_classify_ex(int rows, int cols, const float *intr4, const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6,
                             float tranc_dist, int z0, int z1, const float *depth_max_dev, void *workspace, float slack_scale,
                             const xs_integrate_opts *opts, void *stream);
typedef struct xs_raycast_opts {
    unsigned struct_bytes;           /* sizeof(xs_raycast_opts) */
    int signmap_shift;               /* the sign map's brick shift (2..6) */
    const void *signmap;             /* the sign map the march goes by, or NULL (march from t = 0.2) */
    float signmap_tranc_dist;        /* the truncation distance the map was reset for */
    float *pyr_vmap1, *pyr_nmap1;    /* model-map pyramid built by the one-launch form (all four or none) */
    size_t pyr_step1;
    float *pyr_vmap2, *pyr_nmap2;
    size_t pyr_step2;
    void *completion_event;          /* hipEvent_t riding on the one-launch form's dispatch, or NULL */
    int *steps_dev;                  /* measurement: rows x cols ints receiving every ray's march length, or NULL */
    int pyramid_built;               /* OUT: 1 if the call built the pyramid (else launch xs_resize_pyramid) */
} xs_raycast_opts;
int xs_raycast_ex(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                  float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                  float *vmap, float *nmap, size_t map_step, int rows, int cols, unsigned long long *hits_dev, float *workspace,
                  xs_raycast_opts *opts, void *stream);
int xs_raycast_slab_ex(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                       float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                       int zs0, int zs1, int z0, int z1, float *vmap, float *nmap, size_t map_step, int rows, int cols, int *keys_dev,
                       const xs_raycast_opts *opts, void *stream);
int xs_resize_pyramid_ex(const float *vmap0, const float *nmap0, size_t in_step, int rows0, int cols0, float *vmap1, float *nmap1,
                         size_t mid_step, float *vmap2, float *nmap2, size_t out_step, void *completion_event, void *stream);

/* ---- Dual-complex Hessian / real loss over the volume ----------------------------------- */
size_t xs_tsdf_reduce_workspace_bytes(void);
/* The workspace of the three residual kernels (Hessian, loss, Gauss-Newton terms): its first 256 bytes must be ZERO on first use — call this once after
 * allocating it (or zero-fill it) — and every launch leaves them zero (the last workgroup resets the arrival ticket), so a launch carries no fill of
 * its own.  One launch at a time per workspace; initialise again after a launch that did not complete.  (ABI 2; ABI 1 filled the ticket per launch.) */
int xs_tsdf_reduce_workspace_init(void *workspace, void *stream);
/* float4 ComputeLocalTsdf_hessian(depth, Intr, depthScaled, res, voxel_size, const MatD33& Rv2c,
 *     const devDComplex3& tv2c, tranc_dist, threshold, k, gt, real, grad, hessian, count)
 *                                                  TsdfFusion.h:55-60, TsdfFusion.cu:204-331
 * One fused pass: the reference's kernel + 4 thrust::reduce.  out4_dev = {loss, grad, hessian,
 * count} as doubles.  gt: dense unpitched TSDF of the slab [z0, z1).  The four per-voxel
 * volumes (indexed like gt) are optional: all four or none.  threshold / k are unused by the
 * reference kernel and absent here. */
int xs_compute_local_tsdf_hessian(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4,
                                  const int *res, float voxel_size, const float *Rv2c36, const float *tv2c12, float tranc_dist,
                                  const float *gt, float *real_out, float *grad_out, float *hess_out, int *count_out, int z0, int z1,
                                  void *workspace, double *out4_dev, void *stream);
/* float2 ComputeLocalTsdf_loss(..., const Mat33& Rv2c, const float3& tv2c, ..., gt, real, count)
 *                                                  TsdfFusion.h:48-52, TsdfFusion.cu:335-447 */
int xs_compute_local_tsdf_loss(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                               float voxel_size, const float *Rv2c9, const float *tv2c3, float tranc_dist, const float *gt,
                               float *real_out, int *count_out, int z0, int z1, void *workspace, double *out2_dev, void *stream);

/* ---- Depth / vertex / normal maps (Map.h:16-54) ------------------------------------------ */
/* bilateralFilter(const DeviceArray2D<ushort>& src, MapArr& dst)            Map.cu:155-199, 262-271 */
int xs_bilateral_filter(const uint16_t *src, size_t src_step, int rows, int cols, float *dst, size_t dst_step, void *stream);
/* pyrDown(const MapArr& src, MapArr& dst); dst is (rows/2) x (cols/2)        Map.cu:202-230, 274-283 */
int xs_pyr_down(const float *src, size_t src_step, int src_rows, int src_cols, float *dst, size_t dst_step, void *stream);
/* createVMap(const Intr&, const MapArr& depth, MapArr& vmap)                 Map.cu:8-29, 73-86 */
int xs_create_vmap(const float *intr4, const float *depth, size_t depth_step, int rows, int cols, float *vmap, size_t vmap_step,
                   void *stream);
/* createNMap(const MapArr& vmap, MapArr& nmap); rows = rows of one plane     Map.cu:32-70, 89-102 */
int xs_create_nmap(const float *vmap, float *nmap, size_t map_step, int rows, int cols, void *stream);
/* createVMap + createNMap of every pyramid level (1..3) in one launch — the six calls SurfaceMeasure makes
 * per frame (KinectFusionReconstruction.cpp:290-296); same values.  intr4s: levels x {fx, fy, cx, cy}
 * (already divided per level); level l is (rows0 >> l) x (cols0 >> l); pointer / pitch arrays per level. */
int xs_create_vnmaps(int levels, const float *intr4s, const float *const *depths, const size_t *depth_steps, int rows0, int cols0,
                     float *const *vmaps, float *const *nmaps, const size_t *map_steps, void *stream);
/* xs_create_vnmaps that also writes the real parts of both maps as float planes (vreal[l] / nreal[l]: 3 x rows(l) rows of cols(l)
 * floats, pitch real_steps[l] bytes, NaN sentinel in the x plane).  The current-frame maps come from a real depth image, so their
 * imaginary parts are zeros, and the ICP reduction can read these planes instead (xs_icp_accumulate_real /
 * xs_icp_accumulate_posted_real: half the bytes of its current-frame reads).  The three arrays are all given or all NULL. */
int xs_create_vnmaps_real(int levels, const float *intr4s, const float *const *depths, const size_t *depth_steps, int rows0, int cols0,
                          float *const *vmaps, float *const *nmaps, const size_t *map_steps, float *const *vreal, float *const *nreal,
                          const size_t *real_steps, void *stream);
/* resizeVMap / resizeNMap(const MapArr& in, MapArr& out); src_rows = rows of one input plane
 *                                                                            Map.cu:105-152, 233-259 */
int xs_resize_vmap(const float *in, size_t in_step, int src_rows, int src_cols, float *out, size_t out_step, void *stream);
int xs_resize_nmap(const float *in, size_t in_step, int src_rows, int src_cols, float *out, size_t out_step, void *stream);
/* Levels 1 and 2 of both model maps in one launch (what the orchestrator does with two resizeVMap and
 * two resizeNMap calls per frame, KinectFusionReconstruction.cpp:272-277); same values as the four
 * separate calls.  rows0 / cols0: one plane of the level-0 maps. */
int xs_resize_pyramid(const float *vmap0, const float *nmap0, size_t in_step, int rows0, int cols0, float *vmap1, float *nmap1,
                      size_t mid_step, float *vmap2, float *nmap2, size_t out_step, void *stream);
/* (xs_resize_pyramid_ex, below: the same with a completion event riding on the dispatch) */

/* ---- Raycast ------------------------------------------------------------------------------ */
/* raycast(const Intr&, const MatS33& Rc2v, const devComplex3& tc2v, const MatS33& Rv2w,
 *         const devComplex3& tv2w, float tranc_dist, const int3& res, float voxel_size,
 *         const PtrStep<float>& value, const PtrStep<float>& grad, MapArr& vmap, MapArr& nmap)
 *                                                  RayCaster.h:21-25, RayCaster.cu:197-368
 * rows / cols: one map plane.  hits_dev: optional device counter of pixels given a vertex.
 * workspace: optional rows*cols device floats (march kernel + crossing kernel instead of one). */
int xs_raycast(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
               float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
               float *vmap, float *nmap, size_t map_step, int rows, int cols, unsigned long long *hits_dev, float *workspace,
               void *stream);
/* (Measurement hook of bench.py, SURVEY 8(d) "Raycast: report Mrays/s and steps/ray": xs_raycast_opts.steps_dev, a device buffer of
 * rows x cols ints that xs_raycast_ex fills with each ray's march length — the iterations the reference's loop, RayCaster.cu:222-247, runs for it.) */

/* ---- The sign map: where a ray's march may start (no counterpart in the reference) ---------------------------------------------
 * RayCaster.cu:222-247 evaluates every step of every ray from t = 0.2 m.  Each event that can end that loop — a sample outside the
 * volume, a + to - crossing, a - to + crossing — needs a sample outside or a NEGATIVE voxel on one side of the step.  The sign map keeps
 * one byte per brick of 2^shift voxels a side, set once a negative value has been written into the brick (never cleared but by reset /
 * rebuild: a superset); the march (single GPU, or a rank's slab march) samples it along the ray, finds the stretch that cannot hold an event, and resumes the
 * reference's loop behind it at the float time that loop would have there (a table of its running sum) with the value it would
 * carry — the same crossing, vertex and normal bits, about a fifth of the volume reads (march kernel 48 -> 22 us on the benchmark scene).  The owner of the volume is responsible for
 * the map seeing every write: xs_integrate_scaled_ex2 marks it when xs_integrate_opts.signmap names it; after any other write into the value
 * array (a checkpoint, a host upload) call xs_signmap_rebuild, after xs_init_volume call xs_signmap_reset.
 *   xs_raycast_signmap_shift   the smallest brick shift the march can use for these intrinsics, voxel size and truncation distance — the
 *                        finest map, measured fastest — or 0 if none (the bricks must outgrow a wave's pixel tile at 5 m; the march at most
 *                        320 steps long)
 *   xs_signmap_bytes     size of the device buffer for a resolution and brick shift (2..6; 3 = 8^3 voxels), 0 if invalid
 *   xs_signmap_reset     empty map + the time table for tranc_dist (time step 0.8 * tranc_dist, RayCaster.cu:350)
 *   xs_signmap_rebuild   reset, then mark every brick of `value` that holds a negative voxel
 *   xs_integrate_opts.signmap  the map an xs_integrate_scaled_ex2 call marks (slab launches mark the bricks of their own planes; NULL = none).  EVERY
 *                              integrate call on a volume that has a map must name it: the streamed planes of a later call do not re-mark
 *                              what an unmarked earlier call wrote
 *   xs_raycast_opts.signmap / signmap_shift / signmap_tranc_dist   the map an xs_raycast_ex (with a workspace) / xs_raycast_slab_ex call marches
 *                              by — a rank of a sharded volume passes the map its own integrate calls (owned slab + halo) marked; it must
 *                              have been reset for the same tranc_dist, else the call refuses; NULL = march from t = 0.2 */
int xs_raycast_signmap_shift(const float *intr4, float voxel_size, float tranc_dist);
size_t xs_signmap_bytes(const int *res, int shift);
int xs_signmap_reset(void *signmap, const int *res, int shift, float tranc_dist, void *stream);
int xs_signmap_rebuild(void *signmap, const int *res, int shift, float tranc_dist, const float *value, size_t vol_step, void *stream);
int xs_signmap_rebuild_slab(void *signmap, const int *res, int shift, float tranc_dist, const float *value, size_t vol_step, int zs0, int zs1,
                            void *stream);   /* one rank's storage of a z-sharded volume: value holds planes [zs0, zs1) */

/* The one-launch form of xs_raycast (workspace + sign map) can build the model-map pyramid too — levels 1 and 2 of the vertex and of the
 * normal map, the values xs_resize_pyramid writes (resizeVMap / resizeNMap twice, Map.h:46-54): every workgroup halves its own pixel tile
 * twice, so the frame's tail loses a launch.  xs_raycast_opts.pyr_* name the four maps (NULLs: none); .pyramid_built tells whether the call
 * built them (0: launch xs_resize_pyramid as before); .completion_event rides on that launch's dispatch (its completion), or NULL. */

/* Slab form for a z-sharded volume (the reference is single-GPU; per-ray semantics are those of
 * RayCaster.cu:197-310).  value / grad hold planes [zs0, zs1) = owned slab + halo (6 planes);
 * only march steps whose sample voxel lies in the owned planes [z0, z1) are evaluated.
 * keys_dev[rows*cols]: per pixel (step << 1 | no_vertex) of the first event among those steps, or
 * INT_MAX.  vmap / nmap: this rank's vertex / normal for its own event, zeros elsewhere.
 * Protocol: keys -> all-reduce(min) -> xs_raycast_compose_mask -> all-reduce(sum) of the maps as
 * int32 -> xs_raycast_compose_finish. */
int xs_raycast_slab(const float *intr4, const float *Rc2v18, const float *tc2v6, const float *Rv2w18, const float *tv2w6,
                    float tranc_dist, const int *res, float voxel_size, const float *value, const float *grad, size_t vol_step,
                    int zs0, int zs1, int z0, int z1, float *vmap, float *nmap, size_t map_step, int rows, int cols, int *keys_dev,
                    void *stream);
int xs_raycast_compose_mask(const int *own_keys_dev, const int *min_keys_dev, float *vmap, float *nmap, size_t map_step, int rows,
                            int cols, void *stream);
int xs_raycast_compose_finish(const int *min_keys_dev, float *vmap, float *nmap, size_t map_step, int rows, int cols,
                              unsigned long long *hits_dev, void *stream);
/* The composite without the sum of the maps (round 4): xs_raycast_compose_pack packs the pixels this rank owns with a vertex (own key ==
 * min key) into entries of xs_raycast_compose_entry_bytes() = 52 bytes {pixel index, vertex (3 complex), normal (3 complex)} and advances
 * *count_dev (zeroed by the caller; room for rows * cols entries) by their number; the caller gathers the ranks' packs (variable sizes)
 * and hands all of them to xs_raycast_compose_scatter, which writes them into the maps; then xs_raycast_compose_finish as before.  Per
 * rank and frame a ring all-reduce of the maps moves 2 (N - 1) / N x 14.7 MB, the gathered packs (N - 1) / N x 52 B x hits: half. */
size_t xs_raycast_compose_entry_bytes(void);
int xs_raycast_compose_pack(const int *own_keys_dev, const int *min_keys_dev, const float *vmap, const float *nmap, size_t map_step, int rows, int cols,
                            void *entries_dev, int *count_dev, void *stream);
int xs_raycast_compose_scatter(const void *entries_dev, long n, float *vmap, float *nmap, size_t map_step, int rows, int cols, void *stream);

/* First-order CSFD Gauss-Newton terms of ComputeLocalTsdfHessianKernel's residual for six seeded poses in one
 * pass over the volume (BASELINE config 5; the reference has only the single-direction kernels above).
 * Rv2c108 / tv2c36: six MatS33 / devComplex3, pose k carrying i*h on degree of freedom k (equal real parts).
 * out29_dev: sum d_j d_k for j <= k (21, row-major upper triangle), sum d_k r (6), sum r^2, count, with
 * d_k = Im(error_k) = h * dr/dtheta_k and r = Re(error_0), over voxels with gt != 0, |gt| <= 0.95 that pass the
 * kernel's gates for all six poses.  Other arguments as xs_compute_local_tsdf_hessian.  No synchronisation. */
int xs_tsdf_gauss_newton_terms(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                               float voxel_size, const float *Rv2c108, const float *tv2c36, float tranc_dist, const float *gt, int z0, int z1,
                               void *workspace, double *out29_dev, void *stream);
/* ... with the loop protocol the ICP iterations have (round 6): opts NULL = plain xs_tsdf_gauss_newton_terms.
 *   publish_host / publish_seq   host-coherent pinned memory of xs_gn_publish_bytes() bytes (hipHostMalloc, coherent + mapped): the launch's last
 *                                workgroup stores the 29 sums there and then the 64-bit word [32] = publish_seq; the host spins on that word
 *                                instead of hipMemcpyAsync + hipStreamSynchronize.  Use a different number for every launch on a buffer.
 *   pose_mailbox / mailbox_seq   with Rv2c108 = tv2c36 = NULL: the launch is enqueued BEFORE its poses exist (the host is still solving the
 *                                previous pass) and takes them from the mailbox — xs_icp_mailbox_alloc memory, written by xs_gn_post_poses with
 *                                the same number.  xs_gn_post_poses(..., cmd = 1) makes the launch leave; so does a pose that never comes (about a
 *                                second): nothing is summed, out29_dev is not written and the publish word becomes publish_seq | 1 << 63 — written by
 *                                the launch's LAST workgroup like the sums would be (every workgroup arrives whatever it did, so a launch publishes
 *                                exactly one record and leaves the workspace's ticket at zero; a launch of which only some workgroups saw their poses
 *                                before the deadline counts as left).  A posted launch also leaves, in
 *                                double [30] of the record, the 100 MHz ticks its first workgroup waited for the poses (from resident to poses seen:
 *                                the host's side of the loop as the device sees it). */
typedef struct xs_gn_opts {
    unsigned struct_bytes;            /* sizeof(xs_gn_opts) */
    unsigned mailbox_seq;
    const void *pose_mailbox;
    double *publish_host;
    unsigned long long publish_seq;
} xs_gn_opts;
int xs_tsdf_gauss_newton_terms_ex(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, const int *res,
                                  float voxel_size, const float *Rv2c108, const float *tv2c36, float tranc_dist, const float *gt, int z0, int z1,
                                  void *workspace, double *out29_dev, const xs_gn_opts *opts, void *stream);
size_t xs_gn_publish_bytes(void);     /* 33 doubles */
size_t xs_gn_mailbox_bytes(void);     /* six pose mailboxes in a row (xs_icp_mailbox_alloc's 4096 bytes hold them) */
/* host: the six seeded poses (or cmd 1 = leave) for the launch polling mailbox_host for mailbox_seq */
void xs_gn_post_poses(void *mailbox_host, const float *Rv2c108, const float *tv2c36, unsigned mailbox_seq, int cmd);
/* shard mode: the n <= 32 sums as they stand in device memory behind the stream's all-reduce, published like the kernel's own (word [32] = seq) */
int xs_gn_publish_sums(const double *sums_dev, int n, double *publish_host, unsigned long long seq, void *stream);

/* ---- surface extraction (export; real-valued) ------------------------------------------------ */
size_t xs_extract_workspace_bytes(const int *res);
/* size_t extractPoints(value_volume, weight_volume, grad_volume, volume_resolution, voxel_size,
 * PtrSz<float3> output)                       ExtractPointCloud.h:19-20, ExtractPointCloud.cu:25-211
 * Zero crossings of the TSDF between axis neighbours (+x, +y, +z; both samples < 0.99, opposite signs),
 * linearly interpolated, for the voxels of planes [z0, z1), z1 <= res[2] - 1 (whole volume: 0, res[2] - 1).
 * value points at stored plane zs0 (0 for the whole volume) and holds planes up to z1 inclusive.  The weight
 * and grad volumes of the reference signature are not read there either.  points_dev: capacity x 3 floats,
 * filled in a deterministic order (workgroup, z, y, x, direction; the reference's order depends on its
 * atomics).  *count_host = min(found, capacity), the reference's return value; *found_host (optional) =
 * found.  workspace: xs_extract_workspace_bytes(res).  Synchronises the stream, as the reference does. */
int xs_extract_points(const float *value, size_t vol_step, const int *res, float voxel_size, int zs0, int z0, int z1, float *points_dev,
                      size_t capacity, void *workspace, size_t *count_host, size_t *found_host, void *stream);
/* void extractNormals(value_volume, weight_volume, grad_volume, volume_resolution, voxel_size,
 * PtrSz<float3> points, PtrSz<float3> normal)            ExtractPointCloud.h:22-23, .cu:214-362
 * Central differences of the trilinearly interpolated TSDF one voxel either side of each point, divided by
 * their squared length as in the reference; (0, 0, 0) within two voxels of the border.  value holds stored
 * planes [zs0, zs1) (whole volume: 0, res[2]).  No synchronisation. */
int xs_extract_normals(const float *value, size_t vol_step, const int *res, float voxel_size, int zs0, int zs1, const float *points_dev,
                       size_t n, float *normals_dev, void *stream);

/* ---- ICP normal equations ----------------------------------------------------------------- */
size_t xs_icp_workspace_bytes(void);
/* zero the workspace's arrival ticket once after allocation; every launch leaves it zero */
int xs_icp_workspace_init(void *workspace, void *stream);
/* Device half of estimateCombined (ICP.h:24-31, ICP.cu:166-281 + 120-164): sums_dev receives 55
 * doubles — the 27 complex<double> sums in the reference's mbuf order, then the inlier count.
 * workspace replaces gbuf.  [y0, y1): pixel rows covered.  done_flag (optional; with sums_dev in
 * host-coherent pinned memory): the kernel stores done_seq there, system-scope release, after the
 * sums — the host may spin on it instead of copying + synchronising.  No synchronisation.
 * done_flag == XS_ICP_PUBLISH_PAIRS (xs_icp_accumulate, _real, _posted, _posted_real): sums_dev is host-coherent pinned memory of
 * XS_ICP_PAIRS_BYTES taking 55 pairs {u64 done_seq, double sum} — every sum leaves as one 16-byte store that carries its own sequence
 * word, so no completion word has to be ordered behind the sums (on the device: no wait for the stores' acknowledgement, no barrier, no
 * release store).  The host spins until all 55 sequence words equal done_seq (xs_icp_wait_pairs does); a posted launch that gave up
 * writes done_seq | 1 << 63 into the first pair's word.  Use a different done_seq for every launch on a buffer. */
#define XS_ICP_PUBLISH_PAIRS ((unsigned long long *)(size_t)1)
#define XS_ICP_PAIRS_BYTES (55 * 16)
/* host: spin until every pair of `pairs_host` carries `seq`, then copy the 55 sums out.  0 = done, 1 = the launch gave up (its poses never
 * came), 2 = nothing within max_spins polls. */
int xs_icp_wait_pairs(const void *pairs_host, unsigned long long seq, double *sums55, long long max_spins);
int xs_icp_accumulate(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                      const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                      const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres, int y0, int y1,
                      void *workspace, double *sums_dev, unsigned long long *done_flag, unsigned long long done_seq, void *stream);
/* estimateCombined with the final addition left to the host (same reference interface, ICP.h:24-31; it replaces TranformReduction,
 * ICP.cu:120-164, by a loop on the CPU that is waiting for the result anyway).  Every workgroup stores its record — 56 doubles: the 54
 * partial sums, the inlier count, a sequence word — straight into records_host (host-coherent pinned memory, xs_icp_records_bytes()
 * bytes, zero before first use) and leaves; the sequence word is written last, system-scope release.  No workspace, no ticket, no
 * completion word.  Pose: Rcurr18 / tcurr6, or both NULL with a mailbox and its sequence number (xs_icp_accumulate_posted's
 * protocol).  seq: non-zero, below 2^63, different from the previous launch's on the same buffer.
 * xs_icp_records_count: records a launch over pixel rows [y0, y1) of a cols-wide level writes.
 * xs_icp_sum_records (host only): waits for record 0 .. count - 1 in turn and adds them in that order (deterministic); sums55 = 54
 * sums + inlier count.  Returns 0; 1 if the launch reports that it gave up waiting for a posted pose; -1 after max_spins polls of one
 * record (max_spins <= 0: wait for ever). */
size_t xs_icp_records_bytes(void);
int xs_icp_records_count(int cols, int y0, int y1);
int xs_icp_accumulate_records(const float *Rcurr18, const float *tcurr6, const void *mailbox, unsigned mailbox_seq, const float *vmap_curr,
                              const float *nmap_curr, const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                              const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres, int y0, int y1,
                              double *records_host, unsigned long long seq, void *stream);
int xs_icp_sum_records(const double *records_host, int count, unsigned long long seq, double *sums55, long long max_spins);

/* Test hook for the gate shortcut of the correspondence search: the two rejection tests of ICP.cu:232-241 use only the real
 * part of a complex square root, which the kernel settles from bounds on it unless the value sits on the threshold (then it
 * runs the square root).  Compares that decision with the full evaluation on n complex values (device memory, re / im
 * interleaved): counts_dev[0] = disagreements (must stay 0), counts_dev[1] = values that needed the square root.
 * counts_dev = two zeroed 32-bit words in device memory.  No synchronisation. */
int xs_icp_gate_selftest(const float *z2n_dev, int n, float thres, int or_equal, unsigned *counts_dev, void *stream);
/* estimateCombined with the pose posted AFTER the launch (same reference interface as above; it
 * replaces the launch latency between two iterations of KinectFusionReconstruction.cpp:187-225).
 * The call is made while the previous iteration is still running; the launch becomes resident behind
 * it, polls `mailbox` — xs_icp_mailbox_bytes() (128) bytes, 64-byte aligned, that the host can write and
 * the device read coherently (xs_icp_mailbox_alloc), zero before first use — and starts on the pixels once xs_icp_post_pose(mailbox, Rcurr18, tcurr6, mailbox_seq, 0)
 * has been called; cmd = 1 makes the launch return without touching anything (the host left the loop).
 * Several launches may be queued behind one another, with mailbox sequence numbers that increase in launch order (compared
 * as signed 32-bit distances): each waits for its own number; an abandon command (cmd = 1) posted with a LATER launch's number
 * releases every queued launch up to that one, so one post empties the queue.
 * A launch whose pose never arrives gives up after about a second and stores done_seq | 1<<63 to
 * done_flag; call xs_icp_workspace_init again after that.  One mailbox serves one stream: post
 * sequence numbers in launch order, each only after the previous launch's sums were consumed. */
size_t xs_icp_mailbox_bytes(void);
/* a zeroed mailbox where polling is cheapest: fine-grained device memory the CPU writes through the
 * large PCIe BAR (the workgroups then poll local memory), else host-coherent pinned memory */
int xs_icp_mailbox_alloc(void **mailbox, int *in_device_memory);
int xs_icp_mailbox_free(void *mailbox, int in_device_memory);
int xs_icp_accumulate_posted(const void *mailbox, unsigned mailbox_seq, const float *vmap_curr, const float *nmap_curr,
                             const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                             const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres, int y0,
                             int y1, void *workspace, double *sums_dev, unsigned long long *done_flag, unsigned long long done_seq,
                             void *stream);
/* host: writes the mailbox (four 32-byte sectors, each its sequence word + payload: two 64-byte lines).  On a CPU with MOVDIR64B each line is one direct 64-byte
 * store; otherwise payload, store fence, sequence words, store fence.  One post per posted launch, from one host thread per mailbox. */
void xs_icp_post_pose(void *mailbox_host, const float *Rcurr18, const float *tcurr6, unsigned mailbox_seq, int cmd);
/* xs_icp_accumulate / xs_icp_accumulate_posted reading the current-frame maps' real parts from float planes (xs_create_vnmaps_real;
 * 3 x rows rows of cols floats, pitch real_step bytes) instead of the complex maps, whose imaginary parts must be zeros — true for
 * maps made from a real depth image, i.e. always in this pipeline.  Half the bytes of the kernel's current-frame reads; the same sums
 * (an imaginary part that was -0 in the complex map is +0 here).  vmap_curr / nmap_curr may be NULL. */
int xs_icp_accumulate_real(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                           const float *vmap_curr_real, const float *nmap_curr_real, size_t real_step, const float *Rprev_inv18,
                           const float *tprev6, const float *intr4, const float *vmap_g_prev, const float *nmap_g_prev, size_t map_step,
                           int rows, int cols, float distThres, float angleThres, int y0, int y1, void *workspace, double *sums_dev,
                           unsigned long long *done_flag, unsigned long long done_seq, void *stream);
int xs_icp_accumulate_posted_real(const void *mailbox, unsigned mailbox_seq, const float *vmap_curr, const float *nmap_curr,
                                  const float *vmap_curr_real, const float *nmap_curr_real, size_t real_step, const float *Rprev_inv18,
                                  const float *tprev6, const float *intr4, const float *vmap_g_prev, const float *nmap_g_prev,
                                  size_t map_step, int rows, int cols, float distThres, float angleThres, int y0, int y1, void *workspace,
                                  double *sums_dev, unsigned long long *done_flag, unsigned long long done_seq, void *stream);
/* One whole ICP iteration without leaving the device: estimateCombined followed by the pose update
 * the reference's host performs before the next launch (KinectFusionReconstruction.cpp:203-224:
 * A.real().determinant() gate, complex<double> llt().solve, cast to complex<float>, AngleAxis
 * Z*Y*X, tcurr = Rinc*tcurr + tinc, Rcurr = Rinc*Rcurr).  pose_state: xs_icp_pose_state_bytes()
 * (128) bytes of device memory laid out {float R[18]; float t[6]; int status; int iters; double det;
 * double pad[2]}.  With Rcurr18 / tcurr6 given the launch starts from them (first iteration of a
 * frame), with both NULL from the state the previous launch on the stream left.  status: 0 ok,
 * 1 |det| < 1e-15, 2 NaN det — once non-zero the following launches return at once, which is where
 * the reference leaves PoseEstimate.  sums_dev: device memory, as above.  sums_host / pose_state_host
 * (optional, host-coherent pinned memory) receive a copy of the 55 sums and of the state after every
 * solve; done_flag / done_seq as above (published after both).  Each call enqueues the reduction and a
 * one-workgroup solve kernel; the iterations of a frame queue back to back: no synchronisation, one
 * host wait per frame instead of one per iteration. */
size_t xs_icp_pose_state_bytes(void);
int xs_icp_iterate(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                   const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                   const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres, void *workspace,
                   double *sums_dev, double *sums_host, void *pose_state, void *pose_state_host, unsigned long long *done_flag,
                   unsigned long long done_seq, void *stream);
/* estimateCombined(...) whole: accumulate, wait for the launch (the host spins on a pinned record the kernel's last workgroup writes: everything
 * before it on the stream has completed when the call returns — what the reference's device synchronisation gives, without the stream drain),
 * unpack into the symmetric A (36 complex<double>, A[i*6+j] = A[j*6+i]) and b (6).  sums_dev: optional device copy of the 55 doubles.
 *                                                                              ICP.cu:365-429 */
int xs_estimate_combined(const float *Rcurr18, const float *tcurr6, const float *vmap_curr, const float *nmap_curr,
                         const float *Rprev_inv18, const float *tprev6, const float *intr4, const float *vmap_g_prev,
                         const float *nmap_g_prev, size_t map_step, int rows, int cols, float distThres, float angleThres,
                         void *workspace, double *sums_dev, double *A72_host, double *b12_host, long long *inliers, void *stream);
/* ICP.cu:419-428 on host data: 27 (re, im) sums -> A[36], b[6] */
void xs_icp_unpack(const double *sums54, double *A72, double *b12);

/* ---- DeviceArray complex math over arrays -------------------------------------------------- */
/* Experiments/test_CSFD/main.cpp:18-86 over arrays: which 0 mul 1 div 2 exp 3 sin 4 pow(.,3);
 * our 0 = *_raw, 1 = *_our. */
int xs_csfd_array_op(int which, int our, const float *a, const float *b, float *out, long n, void *stream);
/* f1(x, y) = (x + y)^2 in dual-complex arithmetic (test_CSFD/main.cpp:8-11) */
int xs_dcsfd_f1(const float *x, const float *y, float *out, long n, void *stream);
/* elementwise complex<float> (dual = 0) / d_complex<float> (dual = 1) operator tables
 * (cuda_complex.hpp:100-881, cuda_double_complex.hpp:137-260); op codes listed in DESIGN.md */
int xs_complex_table(int dual, int op, const float *a, const float *b, float *out, long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XSLAM_AMD_H */
