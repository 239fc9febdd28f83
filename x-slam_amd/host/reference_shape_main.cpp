// reference_shape_main.cpp — the demo of Experiments/test_xkinect_fusion/main.cpp:17-84 over this library,
// in the reference's own call shape (reference_shape.hpp): read the flat YAML, then per frame
//   read depth -> DeviceArray2D<ushort>::upload -> timer -> ProcessFrame -> timer -> log the pose
// and "mean frame time" at the end.  OpenCV is not in the image, so a frame is a raw little-endian
// u16 file `<dataset_dir>depth/<id>.u16` (depth_width x depth_height millimetres, what
// Dataset::getDepthData hands over after imread / factor / flip) instead of a PNG; poses are
// written as `<output_dir>slam/frame-%06d.pose.txt`, four rows of four reals (main.cpp:8-14).
//   usage: reference_shape <config.yaml>
#include "reference_shape.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sys/stat.h>

static bool read_frame(const std::string &path, std::vector<ushort> &pixels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.read(reinterpret_cast<char *>(pixels.data()), (std::streamsize)(pixels.size() * sizeof(ushort)));
    return (size_t)f.gcount() == pixels.size() * sizeof(ushort);
}

int main(int argc, char *argv[]) {
    std::cout << "Demo of XKinectFusion" << std::endl;
    if (argc < 2) {
        std::cout << "please enter the config file name" << "\n";
        return -1;
    }
    const xs_host::FlatYaml config = xs_host::FlatYaml::LoadFile(argv[1]);
    const std::string dataset_dir = config.as<std::string>("dataset_dir"), output_path = config.as<std::string>("output_dir");
    const int end_frame = config.as<int>("end_frame");
    const bool log_slam_pose = config.as<bool>("log_slam_pose", false);
    std::cout << "initialize kinect fusion......" << std::endl;
    ReferenceCallShape kinfu;
    kinfu.SetYamlParameters(config);
    const int start_frame = config.as<int>("start_frame", 0);   // (Dataset.cpp:74-96: the dataset's entry i is file start_frame + i)
    if (log_slam_pose) { mkdir(output_path.c_str(), 0755); mkdir((output_path + "slam/").c_str(), 0755); }
    std::vector<ushort> pixels((size_t)kinfu.depth_width * kinfu.depth_height);
    double total_time = 0;
    int frames = 0;
    std::cout << "start slam!" << std::endl;
    while (kinfu.frame_id < end_frame) {
        const int frame_id = kinfu.frame_id;
        if (!read_frame(dataset_dir + "depth/" + std::to_string(start_frame + frame_id) + ".u16", pixels)) {
            std::cout << "cannot read frame " << frame_id << "\n";
            return -1;
        }
        DeviceArray2D<ushort> depth_frame_d;
        depth_frame_d.upload(pixels.data(), kinfu.depth_width * sizeof(ushort), kinfu.depth_height, kinfu.depth_width);
        const auto t0 = std::chrono::steady_clock::now();
        const int ok = kinfu.ProcessFrame(depth_frame_d);
        total_time += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++frames;
        if (!ok) return -2;   // (the reference's loop would retry the same frame for ever: SURVEY.md appendix B)
        if (log_slam_pose) {
            const xs_host::Matrix4cf pose_c2w = xs_host::inverse(kinfu.world2camera_record.back());
            char name[64];
            snprintf(name, sizeof(name), "frame-%06d.pose.txt", frame_id);
            std::ofstream out(output_path + "slam/" + name);
            out.precision(9);
            for (int i = 0; i < 4; ++i) {
                for (int j = 0; j < 4; ++j) out << pose_c2w(i, j).real() << (j == 3 ? "\n" : " ");
            }
        }
    }
    printf("mean frame time = %.3f ms\n", total_time / (frames ? frames : 1));
    return 0;
}
