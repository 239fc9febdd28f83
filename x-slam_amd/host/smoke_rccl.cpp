// smoke_rccl.cpp — a C++ host running the sharded orchestrator over RCCL with no Python anywhere: world = 1 by default
// (the collectives are single-rank all-reduces), or N ranks as N processes sharing the unique id through a file:
//     ./smoke_rccl                       one rank on device 0
//     ./smoke_rccl <rank> <count> <id-file>     rank of count, device = rank (rank 0 writes the id file, the others wait for it)
// Renders the corner of a room, runs four frames in shard mode (force_shard_composite for count = 1) and
// checks that tracking holds and the collectives ran.  Exit code 0 on success.
#include "../../include/xslam_amd_pipeline.h"
#include "../../include/xslam_amd_rccl.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <hip/hip_runtime_api.h>
#include <string>
#include <thread>
#include <vector>

static long g_calls = 0;
static void *g_comm = nullptr;
static void counted(void *user, int op, void *ptr, long n) { ++g_calls; xs_rccl_collective(user, op, ptr, n); }

int main(int argc, char **argv) {
    const int rank = argc > 2 ? atoi(argv[1]) : 0, count = argc > 2 ? atoi(argv[2]) : 1;
    const std::string idfile = argc > 3 ? argv[3] : "";
    if (hipSetDevice(rank) != hipSuccess) { printf("hipSetDevice(%d) failed\n", rank); return 2; }
    unsigned char id[XS_RCCL_UNIQUE_ID_BYTES];
    if (rank == 0) {
        if (xs_rccl_get_unique_id(id) != 0) { printf("%s\n", xs_rccl_last_error()); return 2; }
        if (count > 1) { std::ofstream f(idfile + ".tmp", std::ios::binary); f.write((const char *)id, sizeof(id)); f.close(); rename((idfile + ".tmp").c_str(), idfile.c_str()); }
    } else {
        for (int i = 0; i < 600; ++i) {
            std::ifstream f(idfile, std::ios::binary);
            if (f && f.read((char *)id, sizeof(id))) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (i == 599) { printf("no unique id in %s\n", idfile.c_str()); return 2; }
        }
    }
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return 2;
    g_comm = xs_rccl_comm_create(id, rank, count, st);
    if (!g_comm) { printf("%s\n", xs_rccl_last_error()); return 2; }
    xs_kf_set_stream(st);
    const int n = 128, W = 640, H = 480;
    char yaml[2048];
    snprintf(yaml, sizeof(yaml),
             "tsdf_size_x: %d\ntsdf_size_y: %d\ntsdf_size_z: %d\ntsdf_voxel_size: %f\nmax_integration_weight: 100\nthres_range: 3\n"
             "init_x: 3.2\ninit_y: 3.2\ninit_z: 3.2\nr_x: 0\nr_y: 0\nr_z: 0\ndepth_width: %d\ndepth_height: %d\nfx: 481.2\nfy: -480.0\n"
             "cx: 319.5\ncy: 239.5\nnum_levels: 3\ndistThres: 0.10\nangleThres: 15\nbiInterpolate_threshold: 0\ntrunc_logistic_k: 0\n"
             "flag_use_gtPose: false\nframe_step: 1\ncsfd_seed_row: 2\ncsfd_seed_col: 3\ncsfd_seed_h: 1e-7\nicp_shard_rows: true\n%s",
             n, n, n, 7.68 / n, W, H, count == 1 ? "force_shard_composite: true\n" : "");
    void *kf = xs_kf_create_sharded(yaml, rank, count, counted, g_comm);
    if (!kf) { printf("xs_kf_create_sharded failed\n"); return 2; }
    // the corner of a room seen from inside (front wall z = 2.5 m, side wall x = 1.2 m, a third plane y = 0.9 m): three
    // orthogonal planes in view constrain all six degrees of freedom; depth = z of the first hit along the pixel's ray; static camera
    std::vector<unsigned short> depth((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double xl = (x - 319.5) / 481.2, yl = (y - 239.5) / -480.0;
            double z = 2.5;
            if (xl > 0 && 1.2 / xl < z) z = 1.2 / xl;
            if (yl > 0 && 0.9 / yl < z) z = 0.9 / yl;
            depth[(size_t)y * W + x] = (unsigned short)(1000.0 * z + 0.5);
        }
    unsigned short *dd = nullptr;
    if (hipMalloc((void **)&dd, depth.size() * 2) != hipSuccess) return 2;
    if (hipMemcpy(dd, depth.data(), depth.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return 2;
    int ok = 1;
    for (int k = 0; k < 4 && ok; ++k) ok = xs_kf_process_frame(kf, dd, (size_t)W * 2);
    xs_kf_synchronize(kf);
    float pose[32];
    xs_kf_get_world2camera(kf, -1, pose);
    const long long U = xs_kf_last_updated_voxels(kf), hits = xs_kf_last_raycast_hits(kf);
    double drift = 0;
    for (int i = 0; i < 3; ++i) drift += (double)pose[(i * 4 + 3) * 2] * pose[(i * 4 + 3) * 2];
    printf("{\"rank\": %d, \"count\": %d, \"rccl_version\": %d, \"tracked\": %d, \"collective_calls\": %ld, \"U_last\": %lld, \"hits_last\": %lld, "
           "\"static_camera_translation_sq\": %.3e, \"composite_bytes\": %lld, \"world2camera\": [", rank, count, xs_rccl_version(), ok, g_calls, U, hits, drift,
           xs_kf_composite_bytes(kf));
    for (int i = 0; i < 32; ++i) printf("%s%.9g", i ? ", " : "", (double)pose[i]);   // (9 significant digits: a float round-trips)
    printf("]}\n");
    xs_kf_destroy(kf);
    (void)hipFree(dd);
    xs_rccl_comm_destroy(g_comm);
    // four frames: 4 raycast composites (3 collectives each: the min of the keys, the sum of the owned-pixel counts, the gather of the
    // packed owned pixels); with more than one rank also 3 tracked frames x 12 ICP all-reduces
    // (a single rank evaluates every pixel row itself).  The camera does not move: the estimate may settle a few millimetres off
    // (6 cm voxels round the room's corners), not more.
    const long expect_calls = 4 * 3 + (count > 1 ? 3 * 12 : 0);
    const bool good = ok == 1 && g_calls == expect_calls && hits > 0.8 * W * H && drift < 1e-4;
    return good ? 0 : 1;
}
