// p3d_scene.h -- host-side C++ mirror of the reference's scene surface (RT/scene.h,
// RT/camera.h, RT/vector.h, RT/color.h; RT/ = /root/reference/P3D_RayTracer_Template2/).
//
// Same class names, method names and argument meaning as the reference so that code written
// against Scene / Camera / Material / Light / Object keeps compiling, but a new
// implementation: objects are plain data holders (the intersectors live on the GPU), the
// loader is a tokenizer instead of iostream extraction, and Scene::flatten() produces the
// p3d_scene_desc the C-ABI (include/p3d_hip.h) consumes.
#ifndef P3D_HOST_SCENE_H
#define P3D_HOST_SCENE_H

#include <cstdint>
#include <string>
#include <vector>

#include "p3d_hip.h"

namespace p3d_host {

class Vector {                                   // RT/vector.h:12-49
public:
    float x = 0, y = 0, z = 0;
    Vector() {}
    Vector(float a, float b, float c) : x(a), y(b), z(c) {}
    float length() const;
    Vector& normalize();
    Vector operator+(const Vector& v) const { return Vector(x + v.x, y + v.y, z + v.z); }
    Vector operator-(const Vector& v) const { return Vector(x - v.x, y - v.y, z - v.z); }
    Vector operator*(float f) const { return Vector(x * f, y * f, z * f); }
    Vector operator/(float f) const { return Vector(x / f, y / f, z / f); }
    float operator*(const Vector& v) const { return x * v.x + y * v.y + z * v.z; }      // inner product
    Vector operator%(const Vector& v) const {                                            // cross product
        return Vector(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x);
    }
};

class Color {                                    // RT/color.h:11-67
    float R = 0, G = 0, B = 0;
public:
    Color() {}
    Color(float r_, float g_, float b_) : R(r_), G(g_), B(b_) {}
    float r() const { return R; }
    float g() const { return G; }
    float b() const { return B; }
};

struct Ray { Vector origin, direction; };        // RT/ray.h (ids/time are unused by rendering)

struct AABB { Vector min = Vector(-1, -1, -1), max = Vector(1, 1, 1); };   // RT/boundingBox.cpp:8-12

typedef enum { NONE, GRID_ACC, BVH_ACC } accelerator;   // RT/scene.h:18

class Material {                                 // RT/scene.h:23-55
public:
    Material() {}
    Material(const Color& c, float Kd, const Color& cs, float Ks, float Shine, float T, float ior)
        : m_diffColor(c), m_specColor(cs), m_Refl(Ks), m_T(T), m_Diff(Kd), m_Shine(Shine), m_Spec(Ks), m_RIndex(ior) {}
    void SetDiffColor(const Color& c) { m_diffColor = c; }
    Color GetDiffColor() const { return m_diffColor; }
    void SetSpecColor(const Color& c) { m_specColor = c; }
    Color GetSpecColor() const { return m_specColor; }
    void SetDiffuse(float v) { m_Diff = v; }
    void SetSpecular(float v) { m_Spec = v; }
    void SetShine(float v) { m_Shine = v; }
    void SetReflection(float v) { m_Refl = v; }
    void SetTransmittance(float v) { m_T = v; }
    void SetRefrIndex(float v) { m_RIndex = v; }
    float GetSpecular() const { return m_Spec; }
    float GetDiffuse() const { return m_Diff; }
    float GetShine() const { return m_Shine; }
    float GetReflection() const { return m_Refl; }
    float GetTransmittance() const { return m_T; }
    float GetRefrIndex() const { return m_RIndex; }
private:
    Color m_diffColor = Color(0.2f, 0.2f, 0.2f), m_specColor = Color(1.0f, 1.0f, 1.0f);
    float m_Refl = 1.0f, m_T = 0.0f, m_Diff = 0.2f, m_Shine = 20.0f, m_Spec = 0.8f, m_RIndex = 1.0f;
};

class Light {                                    // RT/scene.h:57-65
public:
    Light(const Vector& pos, const Color& col) : position(pos), color(col) {}
    Vector position;
    Color color;
};

class Object {                                   // RT/scene.h:67-83
public:
    virtual ~Object() {}
    Material* GetMaterial() const { return m_Material; }
    void SetMaterial(Material* m) { m_Material = m; }
    virtual AABB GetBoundingBox() const { return AABB(); }        // planes keep the default
    Vector getCentroid() const { AABB b = GetBoundingBox(); return (b.min + b.max) / 2; }
    virtual int kind() const = 0;                                  // P3D_SPHERE ...
    virtual void flatten(float out12[12]) const = 0;               // p3d_scene_desc::prim_data record
protected:
    Material* m_Material = nullptr;
};

class Plane : public Object {                    // RT/scene.cpp:90-147
public:
    Plane(const Vector& PNc, float Dc) : PN(PNc), D(Dc) {}
    Plane(const Vector& P0, const Vector& P1, const Vector& P2);
    int kind() const override { return P3D_PLANE; }
    void flatten(float o[12]) const override;
    Vector PN; float D = 0;
};

class Triangle : public Object {                 // RT/scene.cpp:10-50
public:
    Triangle(const Vector& P0, const Vector& P1, const Vector& P2);
    AABB GetBoundingBox() const override;
    int kind() const override { return P3D_TRIANGLE; }
    void flatten(float o[12]) const override;
    Vector points[3];
};

class Sphere : public Object {                   // RT/scene.h:116-131
public:
    Sphere(const Vector& c, float r) : center(c), radius(r) {}
    AABB GetBoundingBox() const override;
    int kind() const override { return P3D_SPHERE; }
    void flatten(float o[12]) const override;
    Vector center; float radius;
};

class aaBox : public Object {                    // RT/scene.cpp:188-196
public:
    aaBox(const Vector& mn, const Vector& mx) : min(mn), max(mx) {}
    AABB GetBoundingBox() const override { AABB b; b.min = min; b.max = max; return b; }
    int kind() const override { return P3D_BOX; }
    void flatten(float o[12]) const override;
    Vector min, max;
};

class Camera {                                   // RT/camera.h:14-128
public:
    Camera(Vector from, Vector At, Vector Up, float angle, float hither, float yon, int ResX, int ResY,
           float Aperture_ratio, float Focal_ratio, float t0 = 0.0f, float t1 = 0.0f);
    Vector GetEye() const { return eye; }
    int GetResX() const { return res_x; }
    int GetResY() const { return res_y; }
    float GetFov() const { return fovy; }
    float GetPlaneDist() const { return plane_dist; }
    float GetFar() const { return vfar; }
    float GetAperture() const { return aperture; }
    float GetFocalRatio() const { return focal_ratio; }
    void SetShutterTime(float a, float b) { time0 = a; time1 = b; }
    void SetEye(Vector from);                                       // RT/camera.h:80-89
    void SetResolution(int ResX, int ResY);                         // extension: SURVEY Q14 override
    Ray PrimaryRay(const Vector& pixel_sample) const;               // RT/camera.h:91-108
    Ray PrimaryRay(const Vector& lens_sample, const Vector& pixel_sample) const;   // RT/camera.h:110-127
    void describe(p3d_camera* out) const;                           // POD for the C-ABI
private:
    void derive(bool renormalise_n);
    Vector eye, at, up, u, v, n;
    float fovy, vnear, vfar, plane_dist = 1, focal_ratio, aperture = 0, aperture_ratio;
    float w = 0, h = 0;
    int res_x, res_y;
    float time0, time1;
};

class Scene {                                    // RT/scene.h:148-197
public:
    Scene() {}
    virtual ~Scene();
    Camera* GetCamera() const { return camera; }
    Color GetBackgroundColor() const { return bgColor; }
    bool GetSkyBoxFlg() const { return SkyBoxFlg; }
    unsigned int GetSamplesPerPixel() const { return samples_per_pixel; }
    accelerator GetAccelStruct() const { return accel_struc_type; }
    void SetBackgroundColor(Color c) { bgColor = c; }
    void SetSkyBoxFlg(bool f) { SkyBoxFlg = f; }
    // what Scene::LoadSkybox (RT/scene.cpp:333-381) fills from six image files through DevIL: here the caller brings the
    // decoded faces (right, left, top, bottom, front, back; rows bottom-up; 3 or 4 bytes per pixel).  Copied.
    void SetSkybox(const uint8_t* const faces[6], const uint32_t res_x[6], const uint32_t res_y[6], const uint32_t bytes_per_pixel[6]);
    bool HasSkybox() const { return !skybox_img[0].img.empty(); }
    struct CubeFace { std::vector<uint8_t> img; uint32_t resX = 0, resY = 0, BPP = 3; };
    const CubeFace& GetSkyboxFace(int i) const { return skybox_img[i]; }
    void SetCamera(Camera* c) { delete camera; camera = c; }
    void SetAccelStruct(accelerator a) { accel_struc_type = a; }
    void SetSamplesPerPixel(unsigned int spp) { samples_per_pixel = spp; }
    int getNumObjects() const { return (int)objects.size(); }
    void addObject(Object* o) { objects.push_back(o); }
    Object* getObject(unsigned int i) const { return i < objects.size() ? objects[i] : nullptr; }
    int getNumLights() const { return (int)lights.size(); }
    void addLight(Light* l) { lights.push_back(l); }
    Light* getLight(unsigned int i) const { return i < lights.size() ? lights[i] : nullptr; }
    bool load_p3f(const char* name);             // RT/scene.cpp:476-675 grammar
    const std::string& parse_error() const { return parse_err; }

    // ---- flattening for the C-ABI (scene order preserved: it is the tie-break key)
    struct Flat {
        std::vector<uint32_t> prim_type, prim_material;
        std::vector<float> prim_data, materials, lights;
        p3d_scene_desc desc;
    };
    void flatten(Flat& out) const;
private:
    std::vector<Object*> objects;
    std::vector<Light*> lights;
    std::vector<Material*> materials;            // owned (the reference leaks them)
    Camera* camera = nullptr;
    Color bgColor;
    unsigned int samples_per_pixel = 0;
    accelerator accel_struc_type = NONE;
    bool SkyBoxFlg = false;
    CubeFace skybox_img[6];                       // RT/scene.h:190-195
    std::string parse_err;
};

// Pixel / lens samples of the anti-aliased path in the reference's libc rand() order
// (RT/main.cpp:747,776-801; RT/maths.h:67-70): out[res_y][res_x][spp*spp][4].
void generate_samples(unsigned seed, int res_x, int res_y, int spp, float aperture, float* out);

// renderScene() drop-in (RT/main.cpp:732-832): renders `scene` on one device through the C-ABI.
struct RenderOptions {
    int max_depth = 4;        // MAX_DEPTH
    int accel = -1;           // -1 = scene->GetAccelStruct()
    int spp = -1;             // -1 = scene->GetSamplesPerPixel()
    unsigned seed = 12345;    // replaces time(NULL)
    int device = 0;
    int gpus = 1;             // > 1: devices 0..gpus-1 render interleaved 16-row blocks, one RCCL gather to device 0
    bool counters = false;
    // the reference's distribution-ray-tracing globals (RT/main.cpp:41,43); ANTI_ALIASING and
    // DEPTH_OF_FIELD follow spp > 0 as in RT/main.cpp:943-944
    bool SOFT_SHADOW = false;
    bool FUZZY_REFLECTION = false;
    // misses return Scene::GetSkyboxColor(ray) (RT/scene.cpp:383-461; never called by the reference) when the
    // scene carries a cube map (Scene::SetSkybox): off by default, like in the reference's rayTracing()
    bool SKYBOX = false;
};
struct RenderResult {
    std::vector<uint8_t> img_Data;   // RGB8, bottom row first (RT/main.cpp:76)
    std::vector<float> colors;       // optional float RGB
    std::vector<int32_t> hit_id;     // optional
    p3d_counters counters{};
    float kernel_ms = 0;
};
// saveImgFile("RT_Output.png") of RT/main.cpp:261-276 without DevIL: 8-bit RGB PNG, one IDAT of stored
// (uncompressed) deflate blocks.  img_Data is bottom row first; the file is written top row first.
// Returns 0, or -1 when the file cannot be written.
int save_png(const char* path, const uint8_t* img_Data, int width, int height);

int renderScene(const Scene& scene, const RenderOptions& opt, bool want_colors, bool want_hit, RenderResult& out,
                std::string* err);

}  // namespace p3d_host
#endif
