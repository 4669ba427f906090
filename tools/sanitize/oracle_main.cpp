#include <cstdio>
#include <cstring>
#include <vector>
#include "p3d_oracle.h"
extern "C" int pto_render(int, int, int, float, float, float, float, int, float*, float*);
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        p3o_scene* sc = p3o_scene_load(argv[i]);
        if (!sc) { printf("%s: load failed\n", argv[i]); continue; }
        p3o_scene_set_resolution(sc, 48, 32);
        for (int mode = 0; mode < 6; mode++) {
            p3o_params p; memset(&p, 0, sizeof p);
            p.max_depth = 4; p.accel = mode % 3; p.spp = mode >= 3 ? 2 : 0; p.seed = 5; p.threads = mode == 1 ? 4 : 1;
            p.soft_shadow = mode & 1; p.fuzzy_reflection = (mode >> 1) & 1;
            std::vector<uint8_t> rgb(48 * 32 * 3); std::vector<float> f(48 * 32 * 3); std::vector<int32_t> h(48 * 32);
            p3o_counters c;
            int rc = p3o_render(sc, &p, rgb.data(), f.data(), h.data(), &c);
            if (rc) printf("%s mode %d rc %d\n", argv[i], mode, rc);
        }
        printf("%s ok\n", argv[i]);
        p3o_scene_free(sc);
    }
    return 0;
}
