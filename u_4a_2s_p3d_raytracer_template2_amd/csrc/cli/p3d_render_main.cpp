// p3d_render -- command-line front end: the offline branch of the reference's main()
// (RT/main.cpp:949-976: init_scene -> renderScene -> save image) on an MI355X.
//   p3d_render <scene.p3f> [--res W H] [--accel 0|1|2] [--depth D] [--spp N] [--seed S]
//              [--device K | --gpus N] [--out image.png|image.ppm] [--counters] [--soft-shadow] [--fuzzy-reflection]
// --gpus N: devices 0..N-1 each render every N-th block of 16 rows, one RCCL gather to device 0.
// Defaults are the reference's: resolution / accel / spp from the file, MAX_DEPTH 4.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../host/p3d_scene.h"

using namespace p3d_host;

static int save_ppm(const char* path, const std::vector<uint8_t>& img, int w, int h) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    for (int y = h - 1; y >= 0; y--)                 // img_Data is bottom row first
        fwrite(img.data() + (size_t)y * w * 3, 1, (size_t)w * 3, f);
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s scene.p3f [--res W H] [--accel A] [--depth D] [--spp N] [--seed S] "
                        "[--device K | --gpus N] [--out file.ppm] [--counters] [--soft-shadow] [--fuzzy-reflection]\n", argv[0]);
        return 2;
    }
    RenderOptions opt;
    int rw = 0, rh = 0;
    std::string out = "RT_Output.png";                       // the reference's file name, RT/main.cpp:968
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n); exit(2); } };
        if (a == "--res") { need(2); rw = atoi(argv[++i]); rh = atoi(argv[++i]); }
        else if (a == "--accel") { need(1); opt.accel = atoi(argv[++i]); }
        else if (a == "--depth") { need(1); opt.max_depth = atoi(argv[++i]); }
        else if (a == "--spp") { need(1); opt.spp = atoi(argv[++i]); }
        else if (a == "--seed") { need(1); opt.seed = (unsigned)strtoul(argv[++i], nullptr, 10); }
        else if (a == "--device") { need(1); opt.device = atoi(argv[++i]); }
        else if (a == "--gpus") { need(1); opt.gpus = atoi(argv[++i]); }
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--counters") opt.counters = true;
        else if (a == "--soft-shadow") opt.SOFT_SHADOW = true;
        else if (a == "--fuzzy-reflection") opt.FUZZY_REFLECTION = true;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    Scene scene;
    if (!scene.load_p3f(argv[1])) { fprintf(stderr, "Error opening P3F file: %s\n", scene.parse_error().c_str()); return 1; }
    if (rw > 0 && rh > 0) scene.GetCamera()->SetResolution(rw, rh);
    printf("Scene loaded: %d objects, %d lights, %dx%d\n", scene.getNumObjects(), scene.getNumLights(),
           scene.GetCamera()->GetResX(), scene.GetCamera()->GetResY());
    RenderResult res;
    std::string err;
    auto t0 = std::chrono::high_resolution_clock::now();
    int rc = renderScene(scene, opt, false, false, res, &err);
    auto t1 = std::chrono::high_resolution_clock::now();
    if (rc) { fprintf(stderr, "render failed (%d): %s\n", rc, err.c_str()); return 1; }
    printf("Done: %.3f ms on the device stream, %.3f s wall incl. BVH build and upload\n", res.kernel_ms,
           std::chrono::duration<double>(t1 - t0).count());
    if (opt.counters) {
        unsigned long long rays = res.counters.closest_queries + res.counters.shadow_queries;
        printf("rays=%llu (closest %llu, shadow %llu) box=%llu sph=%llu tri=%llu\n", rays,
               (unsigned long long)res.counters.closest_queries, (unsigned long long)res.counters.shadow_queries,
               (unsigned long long)res.counters.box_tests, (unsigned long long)res.counters.sphere_tests,
               (unsigned long long)res.counters.tri_tests);
    }
    const int W = scene.GetCamera()->GetResX(), H = scene.GetCamera()->GetResY();
    const bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
    if (ppm ? save_ppm(out.c_str(), res.img_Data, W, H) : save_png(out.c_str(), res.img_Data.data(), W, H)) {
        fprintf(stderr, "Error saving Image file\n");
        return 1;
    }
    printf("Image file created: %s\n", out.c_str());
    return 0;
}
