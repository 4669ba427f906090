// ref_skybox_harness.cpp -- TEST INFRASTRUCTURE: calls the reference's own Scene::GetSkyboxColor (RT/scene.cpp:383-461,
// compiled unchanged into oracle/_ref/scene_383_461.o) on a cube map the test supplies.  The images live in a private
// member of Scene (RT/scene.h:190-195) that only Scene::LoadSkybox fills, and LoadSkybox needs DevIL; so this one
// translation unit includes the reference's header with `private` spelled `public` and fills the member itself.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#define private public
#include "scene.h"
#undef private

extern "C" {

// faces: right, left, top, bottom, front, back (RT/scene.cpp:337); returns rgb for each of n directions
void ref_skybox_colors(const uint8_t* const faces[6], const unsigned* res_x, const unsigned* res_y, const unsigned* bpp,
                       int n, const float* dirs3, float* rgb3) {
    Scene sc;
    for (int i = 0; i < 6; i++) {
        sc.skybox_img[i].img = (ILubyte*)faces[i];
        sc.skybox_img[i].resX = res_x[i]; sc.skybox_img[i].resY = res_y[i]; sc.skybox_img[i].BPP = bpp[i];
    }
    for (int k = 0; k < n; k++) {
        Vector o(0.0f, 0.0f, 0.0f), d(dirs3[3 * k], dirs3[3 * k + 1], dirs3[3 * k + 2]);
        Ray r(o, d);
        Color c = sc.GetSkyboxColor(r);
        rgb3[3 * k] = c.r(); rgb3[3 * k + 1] = c.g(); rgb3[3 * k + 2] = c.b();
    }
}

}  // extern "C"
