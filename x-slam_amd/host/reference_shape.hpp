// reference_shape.hpp — the reference's OWN per-frame call sequence, restated over nothing but the
// reference-signature layer (xs_launchers.hpp, TsdfVolume.h, device_array.hpp, host_algebra.hpp):
//   Experiments/test_xkinect_fusion/main.cpp:46-60   upload -> timer -> ProcessFrame
//   XKinectFusion/src/KinectFusionReconstruction.cpp:147-332
// i.e. what a maintainer gets by swapping the library and changing nothing else: one stream, about
// 40 launches a frame, a stream drain wherever the reference's launcher drains the device
// (integrateTsdfVolume, every estimateCombined, every resizeVMap / resizeNMap), no look-ahead, no
// options struct, no posted poses, no pyramid in the raycast, no sign map, no extension argument of
// any launcher.  KinectFusionReconstruction (the redesigned orchestrator beside this file) must give
// the same bits; tests/test_reference_shape_gpu.py holds the two side by side, and bench.py times
// this class as legs.reference_call_shape so the line separates what the library swap alone gives
// from what the redesigned orchestrator adds.
#pragma once
#include "TsdfVolume.h"
#include "flat_yaml.hpp"
#include <vector>

class ReferenceCallShape {
public:
    typedef xs_host::Matrix3cf Matrix3frm;
    typedef xs_host::Matrix4cf Matrix4cf;
    typedef xs_host::Vector3cf Vector3cf;

    int num_levels = 3;
    Matrix4cf world2camera, world2volume;
    std::vector<Matrix4cf> world2camera_record;
    int frame_id = 0, frame_step = 1;
    Intr kinect_intrinsic;
    int depth_width = 0, depth_height = 0;
    Vector3i volume_resolution;
    float voxel_size = 0.f;
    TsdfVolume *tsdf_volume_d_ptr = nullptr;
    int max_integration_weight = 0;
    int icp_iterations[3] = {0, 0, 0};
    float distThres = 0.f, angleThres = 0.f;
    float biInterpolate_threshold = 0.005f, trunc_logistic_k = 0.f;
    DeviceArray2D<devComplexICP> g_buf;
    DeviceArray<devComplexICP> sum_buf;
    std::vector<MapArr> depths_curr_d, vmaps_curr_d, nmaps_curr_d, vmaps_g_prev_d, nmaps_g_prev_d;
    DeviceArray2D<float> depthRawScaled_d;
    std::vector<double> icp_log;   // per iteration of the last frame: the 27 sums as the launcher returned them (A upper triangle | b, row by row)

    ~ReferenceCallShape() { delete tsdf_volume_d_ptr; }
    void SetYamlParameters(const xs_host::FlatYaml &config);
    int ProcessFrame(const DeviceArray2D<ushort> &depth_frame_d);
    Matrix4cf getCamera2Volume() { return world2volume * xs_host::inverse(world2camera); }
    int getFrame() const { return frame_id; }

private:
    void AllocateBuffers();
    void SurfaceMeasure(const DeviceArray2D<ushort> &depth_frame_d);
    int AlignDepthToReconstruction(const DeviceArray2D<ushort> &depth_frame_d);
    int PoseEstimate(Matrix3frm Rcurr, Vector3cf tcurr, Matrix3frm Rprev_inv, Vector3cf tprev);
    int IntegrateFrame(const DeviceArray2D<ushort> &depth_frame_d);
    void CalculatePointCloud(MapArr &xyz_g_d, MapArr &normal_g_d);
};
