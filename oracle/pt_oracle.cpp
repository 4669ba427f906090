// pt_oracle.cpp -- CPU restatement of the GLSL progressive path tracer of the reference
// (PT/ = /root/reference/GPU_PathTracer_template/: common.glsl, P3D_RT.glsl).  TEST
// INFRASTRUCTURE ONLY.  PARITY UNPINNED: there is no GLSL compiler or GL driver in this image
// and the reference holds no output of this shader except a screenshot (PT/shadertoy.png), so
// this file pins nothing against the reference; it is an independent second implementation that
// the HIP kernel is compared with (bit-exact integer hash RNG, statistical agreement of images).
//
// GLSL built-ins are taken at their specification formulas (normalize = v / sqrt(v.v),
// reflect = I - 2 dot(N,I) N, mix = a(1-t) + bt, pow/exp/sin/cos/tan = libm float), function
// arguments are evaluated left to right (GLSL 4.x, section 6.1.1), `out` parameters a callee
// does not write keep the caller's value (what every inlining GLSL compiler does).
// The Shadertoy inputs are fixed to: no mouse (iMouse = 0, iMouseButton = 0), iTime of frame k =
// time0 + k*dt, buffer A = float RGBA holding the gamma-encoded running mean and the frame count.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct v2 { float x, y; };
struct v3 { float x, y, z; };
inline v3 V(float a, float b, float c) { v3 r = {a, b, c}; return r; }
inline v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
inline v3 operator*(v3 a, float f) { return V(a.x * f, a.y * f, a.z * f); }
inline v3 operator*(float f, v3 a) { return V(f * a.x, f * a.y, f * a.z); }
inline v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
inline v3 operator/(v3 a, float f) { return V(a.x / f, a.y / f, a.z / f); }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(v3 a) { return sqrtf(dot(a, a)); }
inline v3 normalize(v3 a) { return a / length(a); }
inline v3 mix(v3 a, v3 b, float t) { return a * (1.0f - t) + b * t; }
inline v3 reflect(v3 I, v3 N) { return I - 2.0f * dot(N, I) * N; }
inline v3 vpow(v3 a, float e) { return V(powf(a.x, e), powf(a.y, e), powf(a.z, e)); }

const float pi = 3.14159265358979f;       // PT/common.glsl:1
const float epsilon = 0.001f;             // PT/common.glsl:2

inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

uint32_t baseHash(uint32_t px, uint32_t py) {                                // PT/common.glsl:31-36
    uint32_t qx = 1103515245U * ((px >> 1U) ^ py);
    uint32_t qy = 1103515245U * ((py >> 1U) ^ px);
    uint32_t h32 = 1103515245U * (qx ^ (qy >> 3U));
    return h32 ^ (h32 >> 16);
}
inline uint32_t seed_hash(float& seed) {      // baseHash(floatBitsToUint(vec2(seed += 0.1, seed += 0.1)))
    seed += 0.1f; float a = seed;
    seed += 0.1f; float b = seed;
    return baseHash(fbits(a), fbits(b));
}
float hash1(float& seed) { return (float)seed_hash(seed) / (float)0xffffffffU; }              // :38-41
v2 hash2(float& seed) {                                                                      // :43-47
    uint32_t n = seed_hash(seed);
    v2 r = {(float)(n & 0x7fffffffU) / (float)0x7fffffff, (float)((n * 48271U) & 0x7fffffffU) / (float)0x7fffffff};
    return r;
}
v3 hash3(float& seed) {                                                                      // :49-54
    uint32_t n = seed_hash(seed);
    return V((float)(n & 0x7fffffffU) / (float)0x7fffffff, (float)((n * 16807U) & 0x7fffffffU) / (float)0x7fffffff,
             (float)((n * 48271U) & 0x7fffffffU) / (float)0x7fffffff);
}
v2 randomInUnitDisk(float& seed) {                                                           // :71-76
    v2 h = hash2(seed);
    float phi = h.y * 6.28318530718f;
    float r = sqrtf(h.x * 1.0f);
    v2 o = {r * sinf(phi), r * cosf(phi)};
    return o;
}
v3 randomInUnitSphere(float& seed) {                                                         // :78-84
    v3 h = hash3(seed) * V(2.0f, 6.28318530718f, 1.0f) - V(1.0f, 0.0f, 0.0f);
    float phi = h.y;
    float r = powf(h.z, 1.0f / 3.0f);
    float s = sqrtf(1.0f - h.x * h.x);
    return V(r * (s * sinf(phi)), r * (s * cosf(phi)), r * h.x);
}
v3 randomUnitVector(float& seed) { return normalize(randomInUnitSphere(seed)); }

struct Ray { v3 o, d; float t; };
struct Camera { v3 eye, u, v, n; float width, height, lensRadius, planeDist, focusDist, time0, time1; };
enum { MT_DIFFUSE = 0, MT_METAL = 1, MT_DIALECTRIC = 2 };
struct Material { int type; v3 albedo, specColor, emissive; float roughness, refIdx; v3 refractColor; };
struct HitRecord { v3 pos, normal; float t; Material material; };

Material diffuseM(v3 albedo) { Material m; m.type = MT_DIFFUSE; m.albedo = albedo; m.specColor = V(0, 0, 0); m.roughness = 1.0f; m.refIdx = 1.0f; m.refractColor = V(0, 0, 0); m.emissive = V(0, 0, 0); return m; }
Material metalM(v3 spec, float rough) { Material m; memset(&m, 0, sizeof m); m.type = MT_METAL; m.albedo = V(0, 0, 0); m.specColor = spec; m.roughness = rough; m.emissive = V(0, 0, 0); return m; }
Material dialectricM(v3 refr, float idx, float rough) { Material m; m.type = MT_DIALECTRIC; m.albedo = V(1, 1, 1); m.specColor = V(0.04f, 0.04f, 0.04f); m.refIdx = idx; m.refractColor = refr; m.roughness = rough; m.emissive = V(0, 0, 0); return m; }

struct Frame { float res_x, res_y, iTime; float gSeed; float mouse_x, mouse_y; };

Camera createCamera(const Frame& F, v3 eye, v3 at, v3 up, float fovy, float aspect, float aperture, float focusDist,
                    float time0, float time1) {                                              // :101-128
    Camera cam;
    if (aperture == 0.0f) cam.focusDist = 1.0f; else cam.focusDist = focusDist;
    v3 w = eye - at;
    cam.planeDist = length(w);
    cam.height = 2.0f * cam.planeDist * tanf(fovy * pi / 180.0f * 0.5f);
    cam.width = aspect * cam.height;
    cam.lensRadius = aperture * 0.5f * cam.width / F.res_x;
    cam.eye = eye;
    cam.n = normalize(w);
    cam.u = normalize(cross(up, cam.n));
    cam.v = cross(cam.n, cam.u);
    cam.time0 = time0; cam.time1 = time1;
    return cam;
}
Ray getRay(Frame& F, const Camera& cam, v2 ps) {                                            // :130-146
    v2 d = randomInUnitDisk(F.gSeed);
    v2 ls = {cam.lensRadius * d.x, cam.lensRadius * d.y};
    float time = cam.time0 + hash1(F.gSeed) * (cam.time1 - cam.time0);
    v3 p = V(cam.width * (ps.x / F.res_x - 0.5f) * cam.focusDist, cam.height * (ps.y / F.res_y - 0.5f) * cam.focusDist, 0.0f);
    v3 eye_offset = cam.eye + cam.u * ls.x + cam.v * ls.y;
    v3 dir = cam.u * (p.x - ls.x) + cam.v * (p.y - ls.y) + cam.n * (-cam.focusDist * cam.planeDist);
    Ray r; r.o = eye_offset; r.d = normalize(dir); r.t = time;
    return r;
}

bool hit_triangle(v3 v0, v3 v1, v3 v2_, const Ray& r, float tmin, float tmax, HitRecord& rec) {   // :334-376
    v3 e1 = v1 - v0, e2 = v2_ - v0;
    v3 pv = cross(r.d, e2);
    float det = dot(pv, e1);
    if (det > -0.0000001f && det < 0.0000001f) return false;
    float inv = 1.0f / det;
    v3 tv = r.o - v0;
    float u = inv * dot(tv, pv);
    if (u < 0.0f || u > 1.0f) return false;
    v3 qv = cross(tv, e1);
    float v = inv * dot(r.d, qv);
    if (v < 0.0f || v > 1.0f) return false;        // (sic) v alone, not u+v: both triangles are full parallelograms
    float t = inv * dot(e2, qv);
    if (t < tmax && t > tmin) {
        rec.t = t; rec.normal = normalize(cross(e1, e2)); rec.pos = r.o + r.d * rec.t;
        return true;
    }
    return false;
}
bool hit_sphere(v3 c, float radius, const Ray& r, float tmin, float tmax, HitRecord& rec) {    // :423-460
    v3 L = r.o - c;
    float b = dot(L, r.d);
    float cc = dot(L, L) - radius * radius;
    if (cc > 0.0f && b > 0.0f) return false;
    float disc = b * b - cc;
    if (disc < 0.0f) return false;
    float t = -b - sqrtf(disc);
    if (t < 0.0f) t = -b + sqrtf(disc);
    if (t < tmax && t > tmin) {
        rec.t = t; rec.pos = r.o + r.d * rec.t;
        rec.normal = radius >= 0.0f ? normalize(rec.pos - c) : normalize(c - rec.pos);
        return true;
    }
    return false;
}
bool hit_movingSphere(v3 c0, v3 c1, float radius, float t0, float t1, const Ray& r, float tmin, float tmax,
                      HitRecord& rec) {                                                     // :462-500
    v3 center = c0 + (c1 - c0) * ((r.t - t0) / (t1 - t0));
    v3 L = r.o - center;
    float B = dot(L, r.d), C = dot(L, L) - radius * radius;
    if (C > 0.0f && B > 0.0f) return false;
    float delta = B * B - C;
    if (delta < 0.0f) return false;
    float t = -B - sqrtf(delta);
    if (t < 0.0f) t = -B + sqrtf(delta);
    if (t < tmax && t > tmin) {
        rec.t = t; rec.pos = r.o + r.d * rec.t;
        rec.normal = radius >= 0.0f ? ((rec.pos - center) / radius) : normalize(center - rec.pos);
        return true;
    }
    return false;
}

bool hit_world(Frame& F, const Ray& r, float tmin, float tmax, HitRecord& rec) {            // PT/P3D_RT.glsl:12-180
    bool hit = false;
    rec.t = tmax;
    if (hit_triangle(V(-10.0f, -0.01f, 10.0f), V(10.0f, -0.01f, 10.0f), V(-10.0f, -0.01f, -10.0f), r, tmin, rec.t, rec)) { hit = true; rec.material = diffuseM(V(0.2f, 0.2f, 0.2f)); }
    if (hit_triangle(V(-10.0f, -0.01f, -10.0f), V(10.0f, -0.01f, 10.0f), V(10.0f, -0.01f, -10.0f), r, tmin, rec.t, rec)) { hit = true; rec.material = diffuseM(V(0.2f, 0.2f, 0.2f)); }
    if (hit_sphere(V(-4.0f, 1.0f, 0.0f), 1.0f, r, tmin, rec.t, rec)) { hit = true; rec.material = diffuseM(V(0.4f, 0.2f, 0.1f)); }
    if (hit_sphere(V(4.0f, 1.0f, 0.0f), 1.0f, r, tmin, rec.t, rec)) { hit = true; rec.material = metalM(V(0.7f, 0.6f, 0.5f), 0.0f); }
    if (hit_sphere(V(0.0f, 1.0f, 0.0f), 1.0f, r, tmin, rec.t, rec)) { hit = true; rec.material = dialectricM(V(0, 0, 0), 1.333f, 0.0f); }
    if (hit_sphere(V(0.0f, 1.0f, 0.0f), -0.5f, r, tmin, rec.t, rec)) { hit = true; rec.material = dialectricM(V(0, 0, 0), 1.333f, 0.0f); }
    const int numxy = 5;
    for (int x = -numxy; x < numxy; ++x)
        for (int y = -numxy; y < numxy; ++y) {
            float fx = (float)x, fy = (float)y;
            float seed = fx + fy / 1000.0f;
            v3 rand1 = hash3(seed);
            v3 center = V(fx + 0.9f * rand1.x, 0.2f, fy + 0.9f * rand1.y);
            float choose = rand1.z;
            if (length(center - V(4.0f, 0.2f, 0.0f)) > 0.9f) {
                if (choose < 0.3f) {
                    v3 center1 = center + V(0.0f, hash1(F.gSeed) * 0.5f, 0.0f);      // consumes the pixel's RNG on every call
                    if (hit_movingSphere(center, center1, 0.2f, 0.0f, 1.0f, r, tmin, rec.t, rec)) { hit = true; v3 a = hash3(seed); v3 b = hash3(seed); rec.material = diffuseM(a * b); }
                } else if (choose < 0.5f) {
                    if (hit_sphere(center, 0.2f, r, tmin, rec.t, rec)) { hit = true; v3 a = hash3(seed); v3 b = hash3(seed); rec.material = diffuseM(a * b); }
                } else if (choose < 0.7f) {
                    if (hit_sphere(center, 0.2f, r, tmin, rec.t, rec)) { hit = true; rec.material = metalM((hash3(seed) + V(1.0f, 1.0f, 1.0f)) * 0.5f, 0.0f); }
                } else if (choose < 0.9f) {
                    if (hit_sphere(center, 0.2f, r, tmin, rec.t, rec)) { hit = true; v3 a = (hash3(seed) + V(1.0f, 1.0f, 1.0f)) * 0.5f; float rg = hash1(seed); rec.material = metalM(a, rg); }
                } else {
                    if (hit_sphere(center, 0.2f, r, tmin, rec.t, rec)) { hit = true; rec.material = dialectricM(hash3(seed), 1.2f, 0.0f); }
                }
            }
        }
    return hit;
}

float schlick(float cosine, float r0) { r0 = r0 * r0; return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f); }   // :210-215

bool scatter(Frame& F, const Ray& rIn, const HitRecord& rec, v3& atten, Ray& rS) {          // PT/common.glsl:217-324
    v3 precise = rec.pos + rec.normal * epsilon;
    if (rec.material.type == MT_DIFFUSE) {
        v3 S = rec.pos + rec.normal + randomUnitVector(F.gSeed);
        v3 dir = normalize(S - rec.pos);
        rS.o = precise; rS.d = normalize(dir); rS.t = rIn.t;
        atten = rec.material.albedo * fmaxf(dot(rS.d, rec.normal), 0.0f) / pi;
        return true;
    }
    if (rec.material.type == MT_METAL) {
        v3 dir = normalize(rIn.d - 2.0f * dot(rIn.d, rec.normal) * rec.normal);
        dir = dir + rec.material.roughness * randomInUnitSphere(F.gSeed);
        rS.o = precise; rS.d = dir; rS.t = rIn.t;
        atten = rec.material.specColor;
        return true;
    }
    if (rec.material.type == MT_DIALECTRIC) {
        atten = rec.material.albedo;
        v3 outwardNormal; float niOverNt, cosine, etaI, etaT;
        if (dot(rIn.d, rec.normal) > 0.0f) {
            outwardNormal = -rec.normal; niOverNt = rec.material.refIdx; cosine = dot(rIn.d, rec.normal);
            etaI = rec.material.refIdx; etaT = 1.0f;
        } else {
            outwardNormal = rec.normal; niOverNt = 1.0f / rec.material.refIdx; cosine = -dot(rIn.d, rec.normal);
            etaI = 1.0f; etaT = rec.material.refIdx;
        }
        float reflectProb;
        float r0 = (etaI - etaT) / (etaI + etaT);
        float k = 1.0f - niOverNt * niOverNt * (1.0f - cosine * cosine);
        if (k < 0.0f) reflectProb = 1.0f; else reflectProb = schlick(cosine, r0);
        if (hash1(F.gSeed) < reflectProb) {
            v3 dir = reflect(rIn.d, rec.normal);
            dir = dir + rec.material.roughness * randomInUnitSphere(F.gSeed);
            precise = rec.pos + outwardNormal * epsilon;
            rS.o = precise; rS.d = dir; rS.t = rIn.t;            // "normalize(dir);" discards its result
        } else {
            v3 refracted = normalize(niOverNt * rIn.d + (niOverNt * cosine - sqrtf(k)) * outwardNormal);
            refracted = mix(refracted, normalize(outwardNormal + randomInUnitSphere(F.gSeed)),
                            rec.material.roughness * rec.material.roughness);
            precise = rec.pos - outwardNormal * epsilon;
            v3 ab = V(expf(rec.material.refractColor.x * -rec.t), expf(rec.material.refractColor.y * -rec.t),
                      expf(rec.material.refractColor.z * -rec.t));
            atten = atten * ab;
            rS.o = precise; rS.d = refracted; rS.t = rIn.t;
        }
        return true;
    }
    return false;
}

v3 directlighting(Frame& F, v3 lpos, v3 lcol, const Ray& r, const HitRecord& rec) {         // PT/P3D_RT.glsl:182-232
    v3 colorOut = V(0, 0, 0);
    v3 lightDir = normalize(lpos - rec.pos);
    float dotRec = fmaxf(dot(rec.normal, lightDir), 0.0f);
    if (dotRec > 0.0f) {
        Ray feeler; feeler.o = rec.pos + epsilon * rec.normal; feeler.d = lightDir; feeler.t = 0.0f;
        float size = length(lightDir);          // (sic) the length of the NORMALISED direction: occluders within ~1 unit only
        HitRecord dummy;
        if (hit_world(F, feeler, 0.0f, size, dummy)) return colorOut;
        v3 specCol, diffCol; float shininess, diffuse, specular;
        if (rec.material.type == MT_DIFFUSE) { specCol = V(0.1f, 0.1f, 0.1f); diffCol = rec.material.albedo; shininess = 10.0f; diffuse = 1.0f; specular = 0.0f; }
        else if (rec.material.type == MT_METAL) { specCol = rec.material.albedo; diffCol = V(0, 0, 0); shininess = 100.0f; diffuse = 0.0f; specular = 1.0f; }
        else { specCol = V(0.004f, 0.004f, 0.004f); diffCol = V(0, 0, 0); shininess = 100.0f; diffuse = 0.0f; specular = 1.0f; }
        lightDir = normalize(lightDir);
        v3 H = normalize(lightDir - r.d);
        diffCol = (lcol * diffCol) * fmaxf(0.0f, dot(rec.normal, lightDir));
        specCol = (lcol * specCol) * powf(fmaxf(0.0f, dot(rec.normal, H)), shininess);
        colorOut = diffCol * diffuse + specCol * specular;
    }
    return colorOut;
}

v3 rayColor(Frame& F, Ray r) {                                                              // PT/P3D_RT.glsl:236-284
    HitRecord rec; memset(&rec, 0, sizeof rec);
    v3 col = V(0, 0, 0), throughput = V(1, 1, 1);
    for (int i = 0; i < 10; ++i) {
        if (hit_world(F, r, 0.001f, 10000.0f, rec)) {
            col = col + directlighting(F, V(-10.0f, 15.0f, 0.0f), V(1, 1, 1), r, rec) * throughput;
            col = col + directlighting(F, V(8.0f, 15.0f, 3.0f), V(1, 1, 1), r, rec) * throughput;
            col = col + directlighting(F, V(1.0f, 15.0f, -9.0f), V(1, 1, 1), r, rec) * throughput;
            Ray sr; v3 atten;
            if (scatter(F, r, rec, atten, sr)) { r = sr; throughput = throughput * atten; }
        } else {
            float t = 0.8f * (r.d.y + 1.0f);
            col = col + throughput * mix(V(1, 1, 1), V(0.5f, 0.7f, 1.0f), t);
            break;
        }
    }
    return col;
}

// one mainImage() evaluation without the accumulation: linear radiance of pixel (x,y) in frame iTime
v3 sample_color(float res_x, float res_y, int x, int y, float iTime, float mouse_px, float mouse_py) {                      // PT/P3D_RT.glsl:286-343
    Frame F; F.res_x = res_x; F.res_y = res_y; F.iTime = iTime;
    float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;
    F.gSeed = (float)baseHash(fbits(fcx), fbits(fcy)) / (float)0xffffffffU + iTime;
    float mx = mouse_px / res_x, my = mouse_py / res_y;    // iMouse.xy / iResolution.xy (0 = never clicked)
    mx = mx * 2.0f - 1.0f;
    v3 camPos = V(mx * 10.0f, my * 5.0f, 8.0f);
    Camera cam = createCamera(F, camPos, V(0.0f, 0.0f, -1.0f), V(0.0f, 1.0f, 0.0f), 60.0f, res_x / res_y, 0.0f, 1.0f, 0.0f, 1.0f);
    v2 j = hash2(F.gSeed);
    v2 ps = {fcx + j.x, fcy + j.y};
    return rayColor(F, getRay(F, cam, ps));
}

}  // namespace

extern "C" {

uint32_t pto_base_hash(uint32_t a, uint32_t b) { return baseHash(a, b); }
void pto_hash_stream(float seed, int n, float* out3n, float* seed_out) {
    for (int i = 0; i < n; i++) { v3 h = hash3(seed); out3n[3 * i] = h.x; out3n[3 * i + 1] = h.y; out3n[3 * i + 2] = h.z; }
    *seed_out = seed;
}
void pto_sample(int res_x, int res_y, int x, int y, float iTime, float mouse_px, float mouse_py, float* rgb) {
    v3 c = sample_color((float)res_x, (float)res_y, x, y, iTime, mouse_px, mouse_py);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
// n_frames mainImage() evaluations with the Shadertoy accumulation (PT/P3D_RT.glsl:334-365):
// rgba = gamma-encoded running mean + frame count; linear_sum (optional) = plain sum of the
// per-frame linear colours (what a multi-GPU sample split reduces).  Rows [y0,y1), threads >= 1.
void pto_render(int res_x, int res_y, int n_frames, float time0, float dt, float mouse_px, float mouse_py, int threads,
                float* rgba, float* linear_sum) {
    auto rows = [&](int ya, int yb) {
        for (int y = ya; y < yb; y++)
            for (int x = 0; x < res_x; x++) {
                float prev[4] = {0, 0, 0, 0};
                v3 sum = V(0, 0, 0);
                for (int k = 0; k < n_frames; k++) {
                    float iTime = time0 + (float)k * dt;
                    v3 color = sample_color((float)res_x, (float)res_y, x, y, iTime, mouse_px, mouse_py);
                    sum = sum + color;
                    v3 prevLinear = vpow(V(prev[0], prev[1], prev[2]), 2.2f);
                    float w = prev[3] + 1.0f;
                    color = mix(prevLinear, color, 1.0f / w);
                    v3 g = vpow(color, 1.0f / 2.2f);
                    prev[0] = g.x; prev[1] = g.y; prev[2] = g.z; prev[3] = w;
                }
                size_t p = (size_t)y * res_x + x;
                memcpy(rgba + 4 * p, prev, sizeof prev);
                if (linear_sum) { linear_sum[3 * p] = sum.x; linear_sum[3 * p + 1] = sum.y; linear_sum[3 * p + 2] = sum.z; }
            }
    };
    if (threads <= 1) { rows(0, res_y); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([&, t]() { for (int y = t; y < res_y; y += threads) rows(y, y + 1); });
    for (auto& th : pool) th.join();
}

}  // extern "C"
