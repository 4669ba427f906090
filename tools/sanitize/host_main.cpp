#include <cstdio>
#include <vector>
#include "host/p3d_scene.h"
#include "scene_flatten.h"
#include "bvh_builder.h"
#include "grid_builder.h"
using namespace p3d_host;
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        Scene sc;
        if (!sc.load_p3f(argv[i])) { printf("%s: load failed: %s\n", argv[i], sc.parse_error().c_str()); continue; }
        Scene::Flat flat; sc.flatten(flat);
        p3d::FlatScene F; std::string why = p3d::flatten_scene(flat.desc, F);
        std::vector<p3d::NodePair> nodes; std::vector<uint32_t> refs; p3d::BvhStats st;
        p3d::build_bvh(F.build_prims, p3d::BvhOptions(), nodes, refs, st);
        {   // what the upload does with the tree: 32-byte quantised node pairs, typed leaves with and without direct references
            p3d::QuantisedNodes Q; p3d::quantise_nodes(nodes, Q);
            for (int direct = 0; direct < 2; direct++) {
                std::vector<p3d::NodePair> n2 = nodes; p3d::FlatScene F2 = F; p3d::TypedLeaves T;
                p3d::type_leaves(n2, refs, F2, T, direct != 0);
            }
            p3d::GridHost g; std::vector<p3d::GridPrim> gp; p3d::grid_prims_from_desc(flat.desc, gp); p3d::build_grid(gp, g);
        }
        p3d_camera cam; sc.GetCamera()->describe(&cam);
        std::vector<float> smp((size_t)32 * 32 * 4 * 4);
        generate_samples(7, 32, 32, 2, cam.aperture, smp.data());
        std::vector<uint8_t> img((size_t)64 * 48 * 3, 128);
        save_png("/tmp/p3d_san/out.png", img.data(), 64, 48);
        printf("%s: %d objects, %zu nodes depth %u %s\n", argv[i], sc.getNumObjects(), nodes.size(), st.max_depth, why.c_str());
    }
    // malformed inputs
    FILE* f = fopen("/tmp/p3d_san/bad.p3f", "w"); fputs("v\nfrom 1 2\ns 1 2\np 3\n1 2 3\nfoo bar\n", f); fclose(f);
    Scene bad; printf("bad: %d %s\n", (int)bad.load_p3f("/tmp/p3d_san/bad.p3f"), bad.parse_error().c_str());
    Scene none; printf("missing: %d\n", (int)none.load_p3f("/tmp/p3d_san/nope.p3f"));
    return 0;
}
