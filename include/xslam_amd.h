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
/* Same, from an already scaled depth image (tsdfFusionKernal alone, TsdfFusion.cu:85-171). */
int xs_integrate_scaled(const float *depth_scaled, size_t scaled_step, int rows, int cols, const float *intr4, int max_weight,
                        const int *res, float voxel_size, const float *Rv2c18, const float *tv2c6, float tranc_dist, float *value,
                        int *weight, float *grad, size_t vol_step, float threshold, int z0, int z1,
                        unsigned long long *updated_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XSLAM_AMD_H */
