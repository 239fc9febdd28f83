// TsdfVolume.h — owner of the TSDF arrays; same public methods as the reference's TsdfVolume
// (XKinectFusion/include/TsdfVolume.h:18-60, src/TsdfVolume.cpp:11-77).  Layout per array:
// rows = Y*Z, cols = X, voxel (x, y, z) at row (y + z*Y).  Only the three live arrays are
// allocated (value f32, weight i32, grad f32 = 12 B / voxel); the reference's fourth
// DeviceArray2D<int> volume_ is never read and is not allocated here.
#pragma once
#include "xs_launchers.hpp"
#include <algorithm>
#include <vector>

struct Vector3i { int v[3]; Vector3i() : v{0, 0, 0} {} Vector3i(int x, int y, int z) : v{x, y, z} {} int &operator()(int i) { return v[i]; } int operator()(int i) const { return v[i]; } int x() const { return v[0]; } int y() const { return v[1]; } int z() const { return v[2]; } int operator[](int i) const { return v[i]; } };

class TsdfVolume {
    float voxel_size_{};
    Vector3i resolution_;
    DeviceArray2D<float> value_volume_;
    DeviceArray2D<int> weight_volume_;
    DeviceArray2D<float> grad_volume_;
    float tranc_dist_{};

public:
    TsdfVolume(Vector3i resolution, float voxel_size, float thres_range) : resolution_(resolution) {
        const int vx = resolution_(0), vy = resolution_(1), vz = resolution_(2);
        value_volume_.create(vy * vz, vx);
        weight_volume_.create(vy * vz, vx);
        grad_volume_.create(vy * vz, vx);
        setVoxelSize(voxel_size);
        const float default_tranc_dist = voxel_size * thres_range;  // metres
        setTsdfTruncDist(default_tranc_dist);
        reset();
    }
    // (xs_const_div_prepare: the raycast march's floor(p / voxel_size) takes the verified short division — once per constant and process, ~2 ms)
    void setVoxelSize(float voxel_size) { voxel_size_ = voxel_size; setTsdfTruncDist(tranc_dist_); (void)xs_const_div_prepare(voxel_size); }
    // never less than 2.1 voxels (TsdfVolume.cpp:35-38)
    void setTsdfTruncDist(float distance) { tranc_dist_ = std::max(distance, 2.1f * voxel_size_); }
    float getTsdfTruncDist() const { return tranc_dist_; }
    DeviceArray2D<float> value() const { return value_volume_; }
    DeviceArray2D<int> weight() const { return weight_volume_; }
    DeviceArray2D<float> grad() const { return grad_volume_; }
    void reset() {
        int3 r; r.x = resolution_(0); r.y = resolution_(1); r.z = resolution_(2);
        initVolume(PtrStep<short>(), value_volume_, weight_volume_, grad_volume_, r);
    }
    void downloadTSDFWithGrad(std::vector<float> &tsdf, std::vector<float> &grad) const {
        tsdf.resize((size_t)value_volume_.cols() * value_volume_.rows());
        grad.resize(tsdf.size());
        value_volume_.download(&tsdf[0], value_volume_.cols() * sizeof(float));
        grad_volume_.download(&grad[0], value_volume_.cols() * sizeof(float));
    }
    void downloadTSDFWithoutGrad(std::vector<float> &tsdf) const {
        tsdf.resize((size_t)value_volume_.cols() * value_volume_.rows());
        value_volume_.download(&tsdf[0], value_volume_.cols() * sizeof(float));
    }
    void downloadWeight(std::vector<int> &weight) const {
        weight.resize((size_t)weight_volume_.cols() * weight_volume_.rows());
        weight_volume_.download(&weight[0], weight_volume_.cols() * sizeof(int));
    }
};
