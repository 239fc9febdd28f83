// xs_pipeline_capi.cpp — C ABI over KinectFusionReconstruction (include/xslam_amd_pipeline.h).
#include "../../include/xslam_amd_pipeline.h"
#include "../csrc/xs_complex.h"
#include "DoubleComplex.h"
#include "KinectFusionReconstruction.h"
#include <cstring>
#include <exception>

typedef KinectFusionReconstruction KF;
using xs_host::Matrix4cf;

extern "C" {

// host DoubleComplex over arrays of (re.re, re.im, im.re, im.im) groups
int xs_host_double_complex_table(int op, long n, const float *a, const float *b, float *out) {
    for (long i = 0; i < n; ++i) {
        const DoubleComplex x(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]), y(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
        DoubleComplex r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = sqrt(x); break;
            case 5: r = DoubleComplex(abs(x)); break;
            case 6: r = exp(x); break;
            case 7: r = log(x); break;
            case 8: r = sin(x); break;
            case 9: r = cos(x); break;
            case 10: r = pow(x, y.real().real()); break;
            case 11: { const DoubleComplex s = x + y; r = s * s; break; }  // f1 of test_CSFD/main.cpp:8-11
            case 12: r = conj(x); break;
            case 13: r = DoubleComplex(norm(x)); break;
            case 14: r = DoubleComplex(SingleComplex((x > y) ? 1.f : 0.f, (x < y) ? 1.f : 0.f)); break;
            default: return -1;
        }
        out[4 * i] = r.real().real(); out[4 * i + 1] = r.real().imag(); out[4 * i + 2] = r.imag().real(); out[4 * i + 3] = r.imag().imag();
    }
    return 0;
}

// csrc/xs_complex.h compiled for the host (the DeviceArray operator API is __host__ __device__ in the
// reference too, cuda_complex.hpp:12-16): n interleaved (re, im) pairs, op codes of xs_complex_table
int xs_host_complex_table(int op, long n, const float *a, const float *b, float *out) {
    using namespace xs;
    for (long i = 0; i < n; ++i) {
        const cfloat x(a[2 * i], a[2 * i + 1]), y(b[2 * i], b[2 * i + 1]);
        cfloat r(0.f, 0.f);
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x / y; break;
            case 4: r = sqrt(x); break;
            case 5: r = cfloat(abs(x), 0.f); break;
            case 6: r = exp(x); break;
            case 7: r = log(x); break;
            case 8: r = pow(x, y); break;
            case 9: r = sin(x); break;
            case 10: r = cos(x); break;
            case 11: r = sinh(x); break;
            case 12: r = cosh(x); break;
            case 13: r = sin_new(x); break;
            case 14: r = sinh_new(x); break;
            case 15: r = cfloat(norm(x), 0.f); break;
            case 16: r = cfloat(arg(x), 0.f); break;
            case 17: r = conj(x); break;
            case 18: r = polar(x.re, y.re); break;
            case 19: r = x / y.re; break;
            case 20: r = y.re / x; break;
            case 21: r = x * y.re; break;
            case 22: r = y.re - x; break;
            case 23: r = proj(x); break;
            case 24: r = log10(x); break;
            case 25: r = tanh(x); break;
            case 26: r = tan(x); break;
            case 27: r = asinh(x); break;
            case 28: r = acosh(x); break;
            case 29: r = atanh(x); break;
            case 30: r = asin(x); break;
            case 31: r = acos(x); break;
            case 32: r = atan(x); break;
            default: return -1;
        }
        out[2 * i] = r.re; out[2 * i + 1] = r.im;
    }
    return 0;
}

// flat-YAML reader probe (CPU only): value of `key` in `yaml_text` as the reader sees it
int xs_flat_yaml_get(const char *yaml_text, const char *key, char *out, int capacity) {
    try {
        const xs_host::FlatYaml y = xs_host::FlatYaml::Load(yaml_text ? yaml_text : "");
        if (!y.has(key)) return -1;
        const std::string v = y.as<std::string>(key);
        if ((int)v.size() + 1 > capacity) return -2;
        std::memcpy(out, v.c_str(), v.size() + 1);
        return (int)v.size();
    } catch (const std::exception &) {
        return -3;
    }
}

void xs_kf_set_stream(void *stream) { xs_host::current_stream() = (hipStream_t)stream; }

void *xs_kf_create_sharded(const char *yaml_text, int rank, int count, void (*collective)(void *, int, void *, long), void *user) {
    try {
        KF *k = new KF();
        k->SetSharding(rank, count, collective, user);
        k->SetYamlParameters(xs_host::FlatYaml::Load(yaml_text ? yaml_text : ""));
        return k;
    } catch (const std::exception &e) {
        printf("xs_kf_create_sharded: %s\n", e.what());
        return nullptr;
    }
}
void xs_kf_shard_planes(void *kf, int *owned2, int *stored2) {
    KF *k = (KF *)kf;
    if (owned2) { owned2[0] = k->zo0; owned2[1] = k->zo1; }
    if (stored2) { stored2[0] = k->zs0; stored2[1] = k->zs1; }
}
void *xs_kf_create(const char *yaml_text) {
    try {
        KF *k = new KF();
        k->SetYamlParameters(xs_host::FlatYaml::Load(yaml_text ? yaml_text : ""));
        return k;
    } catch (const std::exception &e) {
        printf("xs_kf_create: %s\n", e.what());
        return nullptr;
    }
}
void xs_kf_destroy(void *kf) { delete (KF *)kf; }

void xs_kf_set_gt_poses(void *kf, int n, const float *c2w32) {
    KF *k = (KF *)kf;
    k->gt_poses.resize(n);
    for (int i = 0; i < n; ++i) std::memcpy(static_cast<void *>(&k->gt_poses[i]), c2w32 + 32 * i, 32 * sizeof(float));
}

int xs_kf_process_frame(void *kf, const uint16_t *depth_dev, size_t step_bytes) {
    KF *k = (KF *)kf;
    DeviceArray2D<ushort> view(k->depth_height, k->depth_width, (void *)depth_dev, step_bytes);  // borrowed, not counted
    return k->ProcessFrame(view);
}
void xs_kf_hint_next_frame(void *kf, const uint16_t *depth_dev, size_t step_bytes) { ((KF *)kf)->HintNextFrame(depth_dev, step_bytes); }
int xs_kf_process_frame_host(void *kf, const uint16_t *depth_host) { return ((KF *)kf)->ProcessFrameHost(depth_host); }
uint16_t *xs_kf_ingest_buffer(void *kf) { return ((KF *)kf)->IngestBuffer(); }
void xs_kf_get_camera2volume(void *kf, float *out32) {
    const auto m = ((KF *)kf)->getCamera2Volume();
    std::memcpy(out32, &m, 32 * sizeof(float));
}
static DeviceArray2D<ushort> wrap_depth(KF *k, const uint16_t *depth_dev, size_t step_bytes) {
    return DeviceArray2D<ushort>(k->depth_height, k->depth_width, const_cast<uint16_t *>(depth_dev), step_bytes);
}
int xs_kf_gauss_newton_terms(void *kf, const uint16_t *depth_dev, size_t step_bytes, const float *c2v32, double *out29) {
    KF *k = (KF *)kf;
    xs_host::Matrix4cf m;
    std::memcpy(static_cast<void *>(&m), c2v32, 32 * sizeof(float));
    return k->GaussNewtonTerms(wrap_depth(k, depth_dev, step_bytes), m, out29);
}
int xs_kf_relocalize(void *kf, const uint16_t *depth_dev, size_t step_bytes, float *c2v32, int iterations, float damping, double *loss_out) {
    KF *k = (KF *)kf;
    xs_host::Matrix4cf m;
    std::memcpy(static_cast<void *>(&m), c2v32, 32 * sizeof(float));
    std::vector<double> hist;
    const int rc = k->RelocalizeGaussNewton(wrap_depth(k, depth_dev, step_bytes), m, iterations, damping, loss_out ? &hist : nullptr);
    std::memcpy(c2v32, &m, 32 * sizeof(float));
    if (loss_out) for (size_t i = 0; i < hist.size() && i <= (size_t)iterations; ++i) loss_out[i] = hist[i];
    return rc;
}
long long xs_kf_export_point_cloud(void *kf, int max_buffer, float *points_host, float *normals_host) {
    const auto pc = ((KF *)kf)->ExportPointCloud(max_buffer);
    if (points_host && pc.size()) std::memcpy(points_host, pc.positions.data(), pc.positions.size() * sizeof(float));
    if (normals_host && pc.size()) std::memcpy(normals_host, pc.normals.data(), pc.normals.size() * sizeof(float));
    return (long long)pc.size();
}
long long xs_kf_export_ply(void *kf, int max_buffer, const char *filename) {
    const auto pc = ((KF *)kf)->ExportPointCloud(max_buffer);
    return pc.exportPly(filename) ? (long long)pc.size() : -1;
}
void xs_kf_synchronize(void *kf) { ((KF *)kf)->synchronize(); }

int xs_kf_frame_id(void *kf) { return ((KF *)kf)->frame_id; }
int xs_kf_num_poses(void *kf) { return (int)((KF *)kf)->world2camera_record.size(); }
void xs_kf_get_world2camera(void *kf, int idx, float *out32) {
    KF *k = (KF *)kf;
    if (idx < 0) idx += (int)k->world2camera_record.size();
    std::memcpy(out32, &k->world2camera_record[idx], 32 * sizeof(float));
}
float xs_kf_tranc_dist(void *kf) { return ((KF *)kf)->tsdf_volume_d_ptr->getTsdfTruncDist(); }
long long xs_kf_last_updated_voxels(void *kf) { return ((KF *)kf)->lastUpdatedVoxels(); }
long long xs_kf_last_raycast_hits(void *kf) { return ((KF *)kf)->lastRaycastHits(); }
int xs_kf_icp_log(void *kf, double *out, int capacity) {
    KF *k = (KF *)kf;
    const int n = (int)k->icp_log.size();
    if (out && capacity >= n && n) std::memcpy(out, k->icp_log.data(), n * sizeof(double));
    return n;
}

int xs_kf_download_volume(void *kf, float *value, int *weight, float *grad) {
    KF *k = (KF *)kf;
    const int X = k->volume_resolution[0];
    if (value) k->tsdf_volume_d_ptr->value().download(value, X * sizeof(float));
    if (weight) k->tsdf_volume_d_ptr->weight().download(weight, X * sizeof(int));
    if (grad) k->tsdf_volume_d_ptr->grad().download(grad, X * sizeof(float));
    return 0;
}
int xs_kf_download_map(void *kf, int which, int level, float *out) {
    KF *k = (KF *)kf;
    if (level < 0 || level >= k->num_levels) return -1;
    MapArr *m = nullptr;
    switch (which) {
        case 0: m = &k->depths_curr_d[level]; break;
        case 1: m = &k->vmaps_curr_d[level]; break;
        case 2: m = &k->nmaps_curr_d[level]; break;
        case 3: m = &k->vmaps_g_prev_d[level]; break;
        case 4: m = &k->nmaps_g_prev_d[level]; break;
        default: return -1;
    }
    m->download(out, m->cols() * sizeof(devComplex));
    return 0;
}
void *xs_kf_volume_ptr(void *kf, int which, size_t *step_bytes) {
    KF *k = (KF *)kf;
    if (which == 1) { auto a = k->tsdf_volume_d_ptr->weight(); if (step_bytes) *step_bytes = a.step(); return a.ptr(); }
    auto a = which == 0 ? k->tsdf_volume_d_ptr->value() : k->tsdf_volume_d_ptr->grad();
    if (step_bytes) *step_bytes = a.step();
    if (which == 0) k->MarkSignMapStale();   // a caller that writes values through this pointer need not know about the sign map
    return a.ptr();
}

void xs_kf_set_profiling(void *kf, int level) { ((KF *)kf)->set_profiling(level); }
void xs_kf_stage_times(void *kf, double *ms6, long long *calls6) {
    KF *k = (KF *)kf;
    if (k->profiling) k->collect_stage_times();
    for (int i = 0; i < KF::ST_COUNT; ++i) { if (ms6) ms6[i] = k->stage_ms[i]; if (calls6) calls6[i] = k->stage_calls[i]; }
}
void xs_kf_cumulative_counters(void *kf, long long *updated, long long *hits) {
    KF *k = (KF *)kf;
    if (k->profiling) k->collect_stage_times();
    if (updated) *updated = k->cum_updated;
    if (hits) *hits = k->cum_hits;
}
void xs_kf_reset_stage_times(void *kf) {
    KF *k = (KF *)kf;
    if (k->profiling) k->collect_stage_times();
    k->cum_updated = 0; k->cum_hits = 0;
    for (int i = 0; i < KF::ST_COUNT; ++i) { k->stage_ms[i] = 0; k->stage_calls[i] = 0; }
    for (int i = 0; i < 4; ++i) { k->icp_level_us[i] = 0; k->icp_level_calls[i] = 0; k->tail_host_us[i] = 0; }
    k->tail_host_calls = 0;
}
void xs_kf_tail_host_times(void *kf, double *us4, long long *frames) {
    KF *k = (KF *)kf;
    for (int i = 0; i < 4; ++i) if (us4) us4[i] = k->tail_host_us[i];
    if (frames) *frames = k->tail_host_calls;
}
void xs_kf_icp_iteration_times(void *kf, double *us4, long long *calls4) {
    KF *k = (KF *)kf;
    for (int i = 0; i < 4; ++i) { if (us4) us4[i] = k->icp_level_us[i]; if (calls4) calls4[i] = k->icp_level_calls[i]; }
}
void xs_kf_set_gn_post_pose(void *kf, int on) { ((KF *)kf)->gn_post_pose = on != 0; }
void xs_kf_gn_poll_times(void *kf, double *poll_us, long long *poll_passes, int reset) {
    KF *k = (KF *)kf;
    if (poll_us) *poll_us = k->gn_poll_us;
    if (poll_passes) *poll_passes = k->gn_poll_passes;
    if (reset) { k->gn_poll_us = 0; k->gn_poll_passes = 0; }
}
void xs_kf_gn_times(void *kf, double *pass_us, long long *passes, double *kernel_ms, long long *kernel_calls, int reset) {
    KF *k = (KF *)kf;
    if (pass_us) *pass_us = k->gn_pass_us;
    if (passes) *passes = k->gn_passes;
    if (kernel_ms) *kernel_ms = k->gn_kernel_ms;
    if (kernel_calls) *kernel_calls = k->gn_kernel_calls;
    if (reset) { k->gn_pass_us = 0; k->gn_passes = 0; k->gn_kernel_ms = 0; k->gn_kernel_calls = 0; }
}
void xs_kf_debug_set_icp_sequence(void *kf, unsigned long long v) { ((KF *)kf)->DebugSetIcpSequence(v); }
void xs_kf_debug_fail_icp_iteration(void *kf, int n) { ((KF *)kf)->debug_fail_icp_iteration_ = n; }
void xs_kf_debug_post_delay(void *kf, int min_us, int max_us) { ((KF *)kf)->debug_post_delay_us_[0] = min_us; ((KF *)kf)->debug_post_delay_us_[1] = max_us; }
void xs_kf_rebuild_sign_map(void *kf) { ((KF *)kf)->RebuildSignMap(); }
long long xs_kf_composite_bytes(void *kf) { return ((KF *)kf)->composite_bytes_; }
void xs_kf_list_cover_counts(void *kf, long long *counts4) {
    if (counts4) for (int i = 0; i < 4; ++i) counts4[i] = ((KF *)kf)->list_cover_counts_[i];
}
void xs_kf_posted_integrate_counts(void *kf, long long *accepted, long long *refused) {
    if (accepted) *accepted = ((KF *)kf)->posted_accepted_;
    if (refused) *refused = ((KF *)kf)->posted_refused_;
}

int xs_kf_save_checkpoint(void *kf, const char *path) { ((KF *)kf)->saveCheckpoint(path); return 0; }
int xs_kf_load_checkpoint(void *kf, const char *path) { return ((KF *)kf)->loadCheckpoint(path) ? 0 : -1; }
int xs_kf_save_tsdf_volume(void *kf, const char *path) { ((KF *)kf)->saveTSDFVolume(path); return 0; }

}  // extern "C"
