// cornell_bunny.cpp -- the reference's executable (main.cu:39-194) on the drop-in host API:
// scene recipe (rtcuda/cornell_bunny.hpp) -> Bvh -> Scene -> Camera -> render() -> image.ppm.
// Unlike main.cu, which hard-codes 600 x 600 x 10 and "../bun_zipper.ply", sizes and paths are arguments.
//
//   make -C rtcuda_amd/csrc example
//   examples/cornell_bunny [--devices 0,1,...] [width height spp [bun_zipper.ply [image.ppm [matte|full_bsdf|four_bunnies|sixteen_lights]]]]
// --devices: several GPUs of the node behind the one render() call (rt_render_multi; the reference drives one device);
// a device may be listed twice, e.g. `--devices 0,0` on a one-GPU box.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "rtcuda/cornell_bunny.hpp"

int main(int argc, char **argv) {
    std::vector<int> devices;
    if (argc > 2 && strcmp(argv[1], "--devices") == 0) {
        for (const char *q = argv[2]; *q;) {
            char *end = nullptr;
            long v = strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
        }
        argv += 2;
        argc -= 2;
    }
    const int width = argc > 1 ? atoi(argv[1]) : 600, height = argc > 2 ? atoi(argv[2]) : 600;
    const int num_samples = argc > 3 ? atoi(argv[3]) : 10, max_bounces = 10;  // main.cu:168-170
    const std::string ply = argc > 4 ? argv[4] : "data/bun_zipper.ply";
    const std::string out = argc > 5 ? argv[5] : "image.ppm";
    const std::string variant = argc > 6 ? argv[6] : "matte";
    try {
        using rtcuda::CornellBunny;
        CornellBunny::Variant v = variant == "full_bsdf" ? CornellBunny::FULL_BSDF
                                  : variant == "four_bunnies" ? CornellBunny::FOUR_BUNNIES
                                  : variant == "sixteen_lights" ? CornellBunny::SIXTEEN_LIGHTS : CornellBunny::MATTE;
        // stage lines as the reference's driver prints them (main.cu:59-85,172-192; profiler.hpp), so the logs line up;
        // the BVH stage is one line here (the reference's constructor prints five: bvh.cuh:36-218)
        CornellBunny recipe(ply, v, true);
        Scene scene = recipe.scene();
        profiler.start("Constructing BVH");
        prepare(scene);
        profiler.stop();
        Camera camera = CornellBunny::camera((float)width / (float)height);
        std::vector<Vec3> framebuffer;
        rt_stats st;
        profiler.start("Rendering");
        if (devices.empty()) render(width, height, num_samples, max_bounces, camera, scene, framebuffer, 1, &st);
        else render(width, height, num_samples, max_bounces, camera, scene, framebuffer, devices, 1, &st);
        profiler.stop();
        if (!devices.empty()) std::cout << "rendered as " << st.reserved[3] << " device shard(s)" << std::endl;
        std::cout << "render loop " << st.seconds_render * 1e3 << " ms, "
                  << (double)width * height * num_samples / st.seconds_render / 1e6 << " Msamples/s" << std::endl;
        profiler.start("Writing image");
        rtcuda::write_ppm(out, width, height, framebuffer);
        profiler.stop();
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
