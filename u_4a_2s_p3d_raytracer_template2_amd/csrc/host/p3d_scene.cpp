// p3d_scene.cpp -- host scene layer: see p3d_scene.h.
#include "p3d_scene.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

extern "C" int p3d_internal_set_error(int code, const char* msg);

namespace p3d_host {

static const float kEpsilon = 0.001f;                       // RT/macros.h:1
static const float kPI = 3.141592653589793238462f;          // RT/maths.h:7

float Vector::length() const { return std::sqrt(x * x + y * y + z * z); }
Vector& Vector::normalize() {                               // RT/vector.cpp:66-71
    float l = 1.0 / length();
    x *= l; y *= l; z *= l;
    return *this;
}

// ---------------------------------------------------------------- objects
Plane::Plane(const Vector& P0, const Vector& P1, const Vector& P2) {   // RT/scene.cpp:95-115
    PN = (P1 - P0) % (P2 - P0);
    if (PN.length() == 0.0) {
        fprintf(stderr, "DEGENERATED PLANE!\n");
    } else {
        PN.normalize();
        D = PN * P0 * (-1);
    }
}
void Plane::flatten(float o[12]) const {
    memset(o, 0, 12 * sizeof(float));
    o[0] = PN.x; o[1] = PN.y; o[2] = PN.z; o[3] = D;
}

Triangle::Triangle(const Vector& P0, const Vector& P1, const Vector& P2) { points[0] = P0; points[1] = P1; points[2] = P2; }
AABB Triangle::GetBoundingBox() const {                     // RT/scene.cpp:27-44
    AABB b;
    b.min = Vector(std::fmin(std::fmin(points[0].x, points[1].x), points[2].x) - kEpsilon,
                   std::fmin(std::fmin(points[0].y, points[1].y), points[2].y) - kEpsilon,
                   std::fmin(std::fmin(points[0].z, points[1].z), points[2].z) - kEpsilon);
    b.max = Vector(std::fmax(std::fmax(points[0].x, points[1].x), points[2].x) + kEpsilon,
                   std::fmax(std::fmax(points[0].y, points[1].y), points[2].y) + kEpsilon,
                   std::fmax(std::fmax(points[0].z, points[1].z), points[2].z) + kEpsilon);
    return b;
}
void Triangle::flatten(float o[12]) const {
    memset(o, 0, 12 * sizeof(float));
    for (int i = 0; i < 3; i++) { o[3 * i] = points[i].x; o[3 * i + 1] = points[i].y; o[3 * i + 2] = points[i].z; }
}

AABB Sphere::GetBoundingBox() const {                       // RT/scene.cpp:180-186
    AABB b;
    b.min = Vector(center.x - radius, center.y - radius, center.z - radius);
    b.max = Vector(center.x + radius, center.y + radius, center.z + radius);
    return b;
}
void Sphere::flatten(float o[12]) const {
    memset(o, 0, 12 * sizeof(float));
    o[0] = center.x; o[1] = center.y; o[2] = center.z; o[3] = radius;
}
void aaBox::flatten(float o[12]) const {
    memset(o, 0, 12 * sizeof(float));
    o[0] = min.x; o[1] = min.y; o[2] = min.z; o[3] = max.x; o[4] = max.y; o[5] = max.z;
}

// ---------------------------------------------------------------- camera
Camera::Camera(Vector from, Vector At, Vector Up, float angle, float hither, float yon, int ResX, int ResY,
               float Aperture_ratio, float Focal_ratio, float t0, float t1)
    : eye(from), at(At), up(Up), fovy(angle), vnear(hither), vfar(yon), focal_ratio(Focal_ratio),
      aperture_ratio(Aperture_ratio), res_x(ResX), res_y(ResY), time0(t0), time1(t1) {
    derive(true);
}
void Camera::derive(bool renormalise_n) {
    n = eye - at;                                           // RT/camera.h:47-54
    plane_dist = n.length();
    n = n / plane_dist;
    u = up % n;
    u = u / u.length();
    v = n % u;
    // the reference's constructor runs "ze = n.normalize()" AFTER u and v exist, which
    // re-normalises n in place (RT/camera.h:55); SetEye does not (RT/camera.h:80-89)
    if (renormalise_n) n.normalize();
    h = 2 * plane_dist * std::tan((kPI * fovy / 180) / 2.0f);     // float tan, RT/camera.h:62
    w = ((float)res_x / res_y) * h;
    aperture = aperture_ratio * (w / res_x);
}
void Camera::SetEye(Vector from) {
    // RT/camera.h:80-89 recomputes the frame only; w and h keep their constructor values there.
    float w0 = w, h0 = h, ap0 = aperture;
    eye = from;
    derive(false);
    w = w0; h = h0; aperture = ap0;
}
void Camera::SetResolution(int ResX, int ResY) {
    res_x = ResX; res_y = ResY;                 // h depends on the view angle only
    w = ((float)res_x / res_y) * h;
    aperture = aperture_ratio * (w / res_x);
}
Ray Camera::PrimaryRay(const Vector& ps) const {
    Vector vX = u * w * (ps.x / res_x - 0.5f);
    Vector vY = v * h * (ps.y / res_y - 0.5f);
    Vector vZ = n * -plane_dist;
    Ray r; r.origin = eye; r.direction = vX + vY + vZ;
    r.direction.normalize();
    return r;
}
Ray Camera::PrimaryRay(const Vector& ls, const Vector& ps) const {
    Vector p(w * (ps.x / res_x - 0.5f) * focal_ratio, h * (ps.y / res_y - 0.5f) * focal_ratio, 0);
    Ray r;
    r.direction = u * (p.x - ls.x) + v * (p.y - ls.y) + n * (-focal_ratio * plane_dist);
    r.direction.normalize();
    r.origin = eye + (u * ls.x) + (v * ls.y);
    return r;
}
void Camera::describe(p3d_camera* o) const {
    o->eye[0] = eye.x; o->eye[1] = eye.y; o->eye[2] = eye.z;
    o->u[0] = u.x; o->u[1] = u.y; o->u[2] = u.z;
    o->v[0] = v.x; o->v[1] = v.y; o->v[2] = v.z;
    o->n[0] = n.x; o->n[1] = n.y; o->n[2] = n.z;
    o->w = w; o->h = h; o->plane_dist = plane_dist; o->aperture = aperture; o->focal_ratio = focal_ratio;
    o->res_x = res_x; o->res_y = res_y;
}

// ---------------------------------------------------------------- scene + loader
Scene::~Scene() {
    for (auto* o : objects) delete o;
    for (auto* l : lights) delete l;
    for (auto* m : materials) delete m;
    delete camera;
}

namespace {
struct Tokens {
    std::vector<std::string> tok; std::vector<size_t> line; size_t i = 0;
    bool more() const { return i < tok.size(); }
    const char* next() { static const char* empty = ""; return i < tok.size() ? tok[i++].c_str() : empty; }
    float f() { return strtof(next(), nullptr); }          // iostream >> float
    float df() { return (float)strtod(next(), nullptr); }  // iostream >> double, narrowed by the Material ctor
    long l() { return strtol(next(), nullptr, 10); }
    Vector v() { float a = f(), b = f(), c = f(); return Vector(a, b, c); }
    Color c() { float a = f(), b = f(), c2 = f(); return Color(a, b, c2); }
    void skip_line() { size_t li = line[i - 1]; while (i < tok.size() && line[i] == li) i++; }
};
}  // namespace

bool Scene::load_p3f(const char* name) {
    std::ifstream file(name, std::ios::in);
    if (!file) { parse_err = std::string("cannot open ") + name; return false; }
    Tokens T;
    std::string ln; size_t li = 0;
    while (std::getline(file, ln)) {
        std::istringstream is(ln); std::string w;
        while (is >> w) { T.tok.push_back(w); T.line.push_back(li); }
        li++;
    }
    Material* material = nullptr;
    while (T.more()) {
        std::string cmd = T.next();
        if (cmd == "accel") SetAccelStruct((accelerator)(unsigned)T.l());
        else if (cmd == "spp") SetSamplesPerPixel((unsigned)T.l());
        else if (cmd == "f") {
            Color cd = T.c(); float Kd = T.df(); Color cs = T.c();
            float Ks = T.df(), Shine = T.df(), Tr = T.df(), ior = T.df();
            material = new Material(cd, Kd, cs, Ks, Shine, Tr, ior);
            materials.push_back(material);
        } else if (cmd == "s") {
            Vector c = T.v(); float r = T.f();
            Sphere* s = new Sphere(c, r);
            if (material) s->SetMaterial(material);
            addObject(s);
        } else if (cmd == "box") {
            Vector a = T.v(), b = T.v();
            aaBox* bx = new aaBox(a, b);
            if (material) bx->SetMaterial(material);
            addObject(bx);
        } else if (cmd == "p") {
            unsigned nv = (unsigned)T.l();
            if (nv != 3) { parse_err = "Unsupported number of vertices."; fprintf(stderr, "%s\n", parse_err.c_str()); break; }
            Vector a = T.v(), b = T.v(), c = T.v();
            Triangle* t = new Triangle(a, b, c);
            if (material) t->SetMaterial(material);
            addObject(t);
        } else if (cmd == "mesh") {
            unsigned nv = (unsigned)T.l(), nf = (unsigned)T.l();
            std::vector<Vector> vs(nv);
            for (unsigned k = 0; k < nv; k++) vs[k] = T.v();
            for (unsigned k = 0; k < nf; k++) {
                unsigned P0 = (unsigned)T.l(), P1 = (unsigned)T.l(), P2 = (unsigned)T.l();
                if (P0 > 0) { P0 -= 1; P1 -= 1; P2 -= 1; }          // 1-based, RT/scene.cpp:570-579
                else { P0 += nv; P1 += nv; P2 += nv; }
                if (P0 >= nv || P1 >= nv || P2 >= nv) { parse_err = "mesh index out of range"; return false; }
                Triangle* t = new Triangle(vs[P0], vs[P1], vs[P2]);
                if (material) t->SetMaterial(material);
                addObject(t);
            }
        } else if (cmd == "pl") {
            Vector a = T.v(), b = T.v(), c = T.v();
            Plane* p = new Plane(a, b, c);
            if (material) p->SetMaterial(material);
            addObject(p);
        } else if (cmd == "l") {
            Vector pos = T.v(); Color col = T.c();
            addLight(new Light(pos, col));
        } else if (cmd == "v") {
            T.next(); Vector from = T.v();
            T.next(); Vector at = T.v();
            T.next(); Vector up = T.v();
            T.next(); float fov = T.f();
            T.next(); float hither = T.f();
            T.next(); int xres = (int)T.l(); int yres = (int)T.l();
            T.next(); float ap = T.f();
            T.next(); float foc = T.f();
            SetCamera(new Camera(from, at, up, fov, hither, 100.0 * hither, xres, yres, ap, foc));
        } else if (cmd == "bclr") SetBackgroundColor(T.c());
        else if (cmd == "env") { T.next(); SetSkyBoxFlg(true); }   // skybox lookup is dead code
        else if (cmd[0] == '#') T.skip_line();
        else {
            parse_err = "unknown command '" + cmd + "'.";
            fprintf(stderr, "%s\n", parse_err.c_str());
            break;                                                   // the reference stops parsing here
        }
    }
    return camera != nullptr;
}

void Scene::flatten(Flat& F) const {
    F.prim_type.clear(); F.prim_material.clear(); F.prim_data.clear(); F.materials.clear(); F.lights.clear();
    std::vector<const Material*> mats(materials.begin(), materials.end());
    Material fallback;                       // objects declared before any "f" line
    bool need_fallback = false;
    for (auto* o : objects) if (!o->GetMaterial()) need_fallback = true;
    auto mat_index = [&](const Material* m) -> uint32_t {
        if (!m) return (uint32_t)mats.size();
        for (size_t i = 0; i < mats.size(); i++) if (mats[i] == m) return (uint32_t)i;
        mats.push_back(m);                   // material set by SetMaterial from outside the loader
        return (uint32_t)mats.size() - 1;
    };
    for (auto* o : objects) {
        float rec[12];
        o->flatten(rec);
        F.prim_type.push_back((uint32_t)o->kind());
        F.prim_data.insert(F.prim_data.end(), rec, rec + 12);
        F.prim_material.push_back(mat_index(o->GetMaterial()));
    }
    if (need_fallback) {
        uint32_t fb = (uint32_t)mats.size();
        for (size_t i = 0; i < objects.size(); i++) if (!objects[i]->GetMaterial()) F.prim_material[i] = fb;
        mats.push_back(&fallback);
    }
    for (const Material* m : mats) {
        Color d = m->GetDiffColor(), s = m->GetSpecColor();
        float rec[12] = {d.r(), d.g(), d.b(), m->GetDiffuse(), s.r(), s.g(), s.b(), m->GetSpecular(),
                         m->GetShine(), m->GetTransmittance(), m->GetRefrIndex(), m->GetReflection()};
        F.materials.insert(F.materials.end(), rec, rec + 12);
    }
    for (auto* l : lights) {
        float rec[6] = {l->position.x, l->position.y, l->position.z, l->color.r(), l->color.g(), l->color.b()};
        F.lights.insert(F.lights.end(), rec, rec + 6);
    }
    p3d_scene_desc& d = F.desc;
    d.n_prims = (uint32_t)F.prim_type.size();
    d.prim_type = F.prim_type.data(); d.prim_data = F.prim_data.data(); d.prim_material = F.prim_material.data();
    d.n_materials = (uint32_t)(F.materials.size() / 12); d.materials = F.materials.data();
    d.n_lights = (uint32_t)(F.lights.size() / 6); d.lights = F.lights.data();
    d.background[0] = bgColor.r(); d.background[1] = bgColor.g(); d.background[2] = bgColor.b();
}

// ---------------------------------------------------------------- sample stream
static inline float rand_float() { return ((float)rand() / ((float)RAND_MAX + 1.0)); }   // RT/maths.h:67-70

void generate_samples(unsigned seed, int res_x, int res_y, int spp, float aperture, float* out) {
    srand(seed);                                             // set_rand_seed(), RT/main.cpp:747
    size_t k = 0;
    for (int y = 0; y < res_y; y++)
        for (int x = 0; x < res_x; x++)
            for (int i = 0; i < spp; i++)
                for (int j = 0; j < spp; j++) {
                    float px = x + (i + rand_float()) / spp;               // RT/main.cpp:781-782
                    float py = y + (j + rand_float()) / spp;
                    float dx, dy;
                    do {
                        // sampleUnitDisk(): Vector(rand_float(), rand_float(), 0.0) * 2 - Vector(1,1,0).
                        // The two draws are constructor ARGUMENTS; g++ (and MSVC) evaluate them right
                        // to left, so the first draw lands in y (checked against the oracle build).
                        float ry = rand_float();
                        float rx = rand_float();
                        dx = rx * 2 - 1.0f; dy = ry * 2 - 1.0f;
                    } while (dx * dx + dy * dy + 0.0f * 0.0f >= 1.0);
                    out[k++] = px; out[k++] = py;
                    out[k++] = dx * aperture; out[k++] = dy * aperture;    // cameralens = disk * aperture
                }
}

// ---------------------------------------------------------------- PNG output (RT/main.cpp:261-276)
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    if (!table[1]) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}
}  // namespace

int save_png(const char* path, const uint8_t* img, int w, int h) {
    if (!path || !img || w <= 0 || h <= 0) return -1;
    // raw scanlines: filter byte 0 + RGB, top row first (img_Data is bottom row first, RT/main.cpp:76)
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (int y = h - 1; y >= 0; y--) {
        raw.push_back(0);
        raw.insert(raw.end(), img + (size_t)y * w * 3, img + (size_t)(y + 1) * w * 3);
    }
    // zlib stream of stored blocks (<= 65535 bytes each) + Adler-32
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size() || at == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        const bool last = at + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        at += n;
        if (last) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", std::vector<uint8_t>());
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

// ---------------------------------------------------------------- renderScene drop-in
namespace {

// the frame on opt.gpus devices of this process: scene + BVH replicated, interleaved row blocks per rank
// (p3d_render rank / world), ONE gather per plane into device 0 (p3d_gather_all over RCCL), row order
// restored there (p3d_deinterleave), then the download the single-GPU path also pays
int render_multi_gpu(const Scene::Flat& flat, const p3d_camera& cam, p3d_render_params prm, const RenderOptions& opt,
                     bool want_colors, bool want_hit, RenderResult& out) {
    const int n = opt.gpus, row_block = 16;
    std::vector<p3d_scene*> dev(n, nullptr);
    std::vector<p3d_comm*> comm(n, nullptr);
    struct Plane { int bpp; void* host; std::vector<void*> tile; void* gathered = nullptr; void* frame = nullptr; };
    std::vector<Plane> planes;
    int rc = P3D_OK;
    auto cleanup = [&]() {
        for (auto& pl : planes) {
            for (int r = 1; r < n && r < (int)pl.tile.size(); r++) if (dev[r]) p3d_device_free(dev[r], pl.tile[r]);
            if (dev[0]) { p3d_device_free(dev[0], pl.gathered); p3d_device_free(dev[0], pl.frame); }
        }
        for (auto* c : comm) p3d_comm_destroy(c);
        for (auto* d : dev) p3d_scene_destroy(d);
    };
    for (int r = 0; r < n && !rc; r++) rc = p3d_scene_create(&flat.desc, nullptr, r, &dev[r]);
    if (!rc) rc = p3d_comm_create_all(nullptr, n, comm.data());
    if (rc) { cleanup(); return rc; }
    const size_t npx = (size_t)cam.res_x * cam.res_y;
    const size_t tile_px = (size_t)p3d_local_rows(cam.res_y, row_block, n) * cam.res_x;
    out.img_Data.assign(npx * 3, 0);
    planes.push_back(Plane{3, out.img_Data.data(), {}});
    if (want_colors) { out.colors.assign(npx * 3, 0.0f); planes.push_back(Plane{12, out.colors.data(), {}}); }
    if (want_hit) { out.hit_id.assign(npx, -1); planes.push_back(Plane{4, out.hit_id.data(), {}}); }
    for (auto& pl : planes) {
        pl.tile.assign(n, nullptr);
        if (!rc) rc = p3d_device_alloc(dev[0], (uint64_t)tile_px * pl.bpp * n, &pl.gathered);
        if (!rc) rc = p3d_device_alloc(dev[0], (uint64_t)npx * pl.bpp, &pl.frame);
        pl.tile[0] = pl.gathered;                                   // rank 0 renders straight into its slot
        for (int r = 1; r < n && !rc; r++) rc = p3d_device_alloc(dev[r], (uint64_t)tile_px * pl.bpp, &pl.tile[r]);
    }
    if (!rc) rc = p3d_timer_begin(dev[0]);
    for (int r = 0; r < n && !rc; r++) {
        prm.rank = r; prm.world = n; prm.row_block = row_block;
        p3d_outputs o;
        o.memory = 1; o.rgb8 = (uint8_t*)planes[0].tile[r]; o.rgb32f = nullptr; o.hit_id = nullptr;
        for (auto& pl : planes) {
            if (pl.bpp == 12) o.rgb32f = (float*)pl.tile[r];
            if (pl.bpp == 4) o.hit_id = (int32_t*)pl.tile[r];
        }
        rc = p3d_render(dev[r], &cam, &prm, &o);
    }
    for (auto& pl : planes) {
        std::vector<const void*> tiles(pl.tile.begin(), pl.tile.end());
        if (!rc) rc = p3d_gather_all(comm.data(), dev.data(), tiles.data(), n, pl.gathered, (uint64_t)tile_px * pl.bpp);
        if (!rc) rc = p3d_deinterleave(dev[0], pl.gathered, pl.frame, cam.res_x, cam.res_y, row_block, n, pl.bpp, 0);
    }
    if (!rc) rc = p3d_timer_end(dev[0], &out.kernel_ms);
    for (auto& pl : planes)
        if (!rc) rc = p3d_download(dev[0], pl.host, pl.frame, (uint64_t)npx * pl.bpp);
    if (!rc && opt.counters) {
        memset(&out.counters, 0, sizeof out.counters);
        for (int r = 0; r < n && !rc; r++) {
            p3d_counters c;
            rc = p3d_get_counters(dev[r], &c);
            out.counters.closest_queries += c.closest_queries; out.counters.shadow_queries += c.shadow_queries;
            out.counters.box_tests += c.box_tests; out.counters.sphere_tests += c.sphere_tests;
            out.counters.tri_tests += c.tri_tests; out.counters.aabox_tests += c.aabox_tests;
            out.counters.plane_tests += c.plane_tests; out.counters.pixels += c.pixels;
        }
    }
    std::string keep = rc ? p3d_last_error() : "";
    cleanup();
    if (rc) p3d_internal_set_error(rc, keep.c_str());
    return rc;
}

}  // namespace

void Scene::SetSkybox(const uint8_t* const faces[6], const uint32_t res_x[6], const uint32_t res_y[6], const uint32_t bytes_per_pixel[6]) {
    for (int i = 0; i < 6; i++) {
        skybox_img[i].resX = res_x[i]; skybox_img[i].resY = res_y[i]; skybox_img[i].BPP = bytes_per_pixel[i];
        skybox_img[i].img.assign(faces[i], faces[i] + (size_t)res_x[i] * res_y[i] * bytes_per_pixel[i]);
    }
}

namespace {
int upload_skybox(const Scene& scene, p3d_scene* dev) {
    const uint8_t* faces[6]; uint32_t rx[6], ry[6], bpp[6];
    for (int i = 0; i < 6; i++) {
        const Scene::CubeFace& f = scene.GetSkyboxFace(i);
        faces[i] = f.img.data(); rx[i] = f.resX; ry[i] = f.resY; bpp[i] = f.BPP;
    }
    return p3d_scene_set_skybox(dev, faces, rx, ry, bpp);
}
}  // namespace

int renderScene(const Scene& scene, const RenderOptions& opt, bool want_colors, bool want_hit, RenderResult& out,
                std::string* err) {
    auto bad = [&](int rc) { if (err) *err = p3d_last_error(); return rc; };
    if (!scene.GetCamera()) { if (err) *err = "scene has no camera"; return P3D_ERR_ARG; }
    if (opt.gpus < 1 || opt.gpus > 64) { if (err) *err = "gpus must be in 1..64"; return P3D_ERR_ARG; }
    Scene::Flat flat;
    scene.flatten(flat);
    p3d_camera cam;
    scene.GetCamera()->describe(&cam);
    p3d_render_params prm;
    memset(&prm, 0, sizeof prm);
    prm.max_depth = opt.max_depth;
    prm.accel = opt.accel < 0 ? (int)scene.GetAccelStruct() : opt.accel;
    prm.spp = opt.spp < 0 ? (int)scene.GetSamplesPerPixel() : opt.spp;
    prm.world = 1; prm.rank = 0; prm.row_block = 16;
    prm.flags = opt.counters ? P3D_FLAG_COUNTERS : 0;
    prm.features = (opt.SOFT_SHADOW ? P3D_FEATURE_SOFT_SHADOW : 0u) | (opt.FUZZY_REFLECTION ? P3D_FEATURE_FUZZY_REFLECTION : 0u);
    if (opt.SKYBOX) {
        if (!scene.HasSkybox()) { if (err) *err = "RenderOptions::SKYBOX without Scene::SetSkybox()"; return P3D_ERR_STATE; }
        if (opt.gpus > 1) { if (err) *err = "the skybox switch is served on one GPU"; return P3D_ERR_ARG; }
        prm.features |= P3D_FEATURE_SKYBOX;
    }
    prm.seed = opt.seed;
    std::vector<float> samples;
    if (prm.spp > 0) {
        samples.resize((size_t)cam.res_x * cam.res_y * prm.spp * prm.spp * 4);
        generate_samples(opt.seed, cam.res_x, cam.res_y, prm.spp, cam.aperture, samples.data());
        prm.samples = samples.data();
    }
    if (opt.gpus > 1) {
        int rc = render_multi_gpu(flat, cam, prm, opt, want_colors, want_hit, out);
        return rc ? bad(rc) : P3D_OK;
    }
    p3d_scene* dev = nullptr;
    int rc = p3d_scene_create(&flat.desc, nullptr, opt.device, &dev);
    if (rc) return bad(rc);
    if (opt.SKYBOX && (rc = upload_skybox(scene, dev)) != 0) { p3d_scene_destroy(dev); return bad(rc); }
    size_t npx = (size_t)cam.res_x * cam.res_y;
    out.img_Data.assign(npx * 3, 0);
    if (want_colors) out.colors.assign(npx * 3, 0.0f);
    if (want_hit) out.hit_id.assign(npx, -1);
    p3d_outputs o;
    o.rgb8 = out.img_Data.data(); o.rgb32f = want_colors ? out.colors.data() : nullptr;
    o.hit_id = want_hit ? out.hit_id.data() : nullptr; o.memory = 0;
    rc = p3d_timer_begin(dev);
    if (!rc) rc = p3d_render(dev, &cam, &prm, &o);
    if (!rc) rc = p3d_timer_end(dev, &out.kernel_ms);
    if (!rc && opt.counters) rc = p3d_get_counters(dev, &out.counters);
    if (rc) { bad(rc); p3d_scene_destroy(dev); return rc; }
    p3d_scene_destroy(dev);
    return P3D_OK;
}

}  // namespace p3d_host
