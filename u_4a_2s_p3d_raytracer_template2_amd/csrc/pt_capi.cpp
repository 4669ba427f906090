// pt_capi.cpp -- C-ABI of the path tracer (include/p3d_pathtracer.h): host-side createCamera()
// (PT/common.glsl:101-128) and mainImage()'s camera set-up (PT/P3D_RT.glsl:290-333), buffers,
// launches of pt_kernels.hip.
#include "p3d_pathtracer.h"
#include "p3d_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

namespace p3dpt {
struct f3 { float x, y, z; };
struct Cam { f3 eye, u, v, n; float width, height, lensRadius, planeDist, focusDist, time0, time1; };
struct PtLaunch {
    Cam cam; float res_x, res_y; int32_t ires_x, ires_y; int32_t n_frames, first_frame, frame_stride;
    float time0, dt; float* rgba; float* linear;
    int32_t n_chunks, chunk_frames; float* partial;
};
hipError_t launch_pt_frames(const PtLaunch& P, hipStream_t stream);
hipError_t launch_pt_hash(uint32_t n, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t stream);
}  // namespace p3dpt

using namespace p3dpt;

// error text shared with the Whitted entry points (p3d_last_error)
extern "C" int p3d_internal_set_error(int code, const char* msg);

namespace {
int fail(int code, const std::string& m) { return p3d_internal_set_error(code, m.c_str()); }
#define PT_TRY(expr)                                                                       \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(P3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

f3 sub(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
f3 cross(f3 a, f3 b) { return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
float len(f3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
f3 norm(f3 a) { float l = len(a); return f3{a.x / l, a.y / l, a.z / l}; }
}  // namespace

struct p3d_pt {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* d_rgba = nullptr; size_t rgba_cap = 0;
    void* d_linear = nullptr; size_t linear_cap = 0;
    void* d_partial = nullptr; size_t partial_cap = 0;      // partial linear sums of the frame runs (linear-only requests)
};

extern "C" {

int p3d_pt_create(int device, p3d_pt** out) {
    if (!out) return fail(P3D_ERR_ARG, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(P3D_ERR_ARG, "device index out of range");
    PT_TRY(hipSetDevice(device));
    p3d_pt* h = new p3d_pt();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) { p3d_pt_destroy(h); return fail(P3D_ERR_HIP, "stream/event creation failed"); }
    h->stream = h->own_stream;
    *out = h;
    return P3D_OK;
}
int p3d_pt_destroy(p3d_pt* h) {
    if (!h) return P3D_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); }
    if (h->d_rgba) (void)hipFree(h->d_rgba);
    if (h->d_linear) (void)hipFree(h->d_linear);
    if (h->d_partial) (void)hipFree(h->d_partial);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return P3D_OK;
}
// device and stream a handle is bound to (for p3d_comm.cpp)
int p3d_internal_pt_binding(p3d_pt* h, int* device, void** stream) {
    if (!h) return fail(P3D_ERR_ARG, "handle is NULL");
    *device = h->device; *stream = (void*)h->stream;
    return P3D_OK;
}
int p3d_pt_set_stream(p3d_pt* h, void* s) {
    if (!h) return fail(P3D_ERR_ARG, "handle is NULL");
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return P3D_OK;
}
int p3d_pt_sync(p3d_pt* h) {
    if (!h) return fail(P3D_ERR_ARG, "handle is NULL");
    PT_TRY(hipSetDevice(h->device));
    PT_TRY(hipStreamSynchronize(h->stream));
    return P3D_OK;
}
int p3d_pt_timer_begin(p3d_pt* h) {
    if (!h) return fail(P3D_ERR_ARG, "handle is NULL");
    PT_TRY(hipSetDevice(h->device));
    PT_TRY(hipEventRecord(h->ev0, h->stream));
    return P3D_OK;
}
int p3d_pt_timer_end(p3d_pt* h, float* ms) {
    if (!h || !ms) return fail(P3D_ERR_ARG, "NULL argument");
    PT_TRY(hipSetDevice(h->device));
    PT_TRY(hipEventRecord(h->ev1, h->stream));
    PT_TRY(hipEventSynchronize(h->ev1));
    PT_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return P3D_OK;
}

int p3d_pt_render(p3d_pt* h, const p3d_pt_params* p, const p3d_pt_outputs* o) {
    if (!h || !p || !o) return fail(P3D_ERR_ARG, "NULL argument");
    if (p->res_x <= 0 || p->res_y <= 0 || p->n_frames <= 0 || p->frame_stride <= 0 || p->first_frame < 0)
        return fail(P3D_ERR_ARG, "bad resolution / frame range");
    PT_TRY(hipSetDevice(h->device));
    PtLaunch L;
    memset(&L, 0, sizeof L);
    L.res_x = (float)p->res_x; L.res_y = (float)p->res_y; L.ires_x = p->res_x; L.ires_y = p->res_y;
    L.n_frames = p->n_frames; L.first_frame = p->first_frame; L.frame_stride = p->frame_stride;
    L.time0 = p->time0; L.dt = p->dt;
    {   // mainImage() camera (ORBIT_CAMERA / SHOWCASE_DOF are false in the reference), then createCamera()
        float mx = p->mouse_x / L.res_x, my = p->mouse_y / L.res_y;
        mx = mx * 2.0f - 1.0f;
        const f3 eye = {mx * 10.0f, my * 5.0f, 8.0f}, at = {0.0f, 0.0f, -1.0f}, up = {0.0f, 1.0f, 0.0f};
        const float pi = 3.14159265358979f, fovy = 60.0f, aperture = 0.0f, aspect = L.res_x / L.res_y;
        Cam& c = L.cam;
        c.focusDist = 1.0f;                               // aperture == 0: pinhole, focus on the view plane
        const f3 w = sub(eye, at);
        c.planeDist = len(w);
        c.height = 2.0f * c.planeDist * std::tan(fovy * pi / 180.0f * 0.5f);
        c.width = aspect * c.height;
        c.lensRadius = aperture * 0.5f * c.width / L.res_x;
        c.eye = eye; c.n = norm(w); c.u = norm(cross(up, c.n)); c.v = cross(c.n, c.u);
        c.time0 = 0.0f; c.time1 = 1.0f;
    }
    const size_t npx = (size_t)p->res_x * p->res_y;
    if (o->memory == 1) { L.rgba = o->rgba; L.linear = o->linear; }
    else {
        if (o->rgba) {
            if (h->rgba_cap < npx * 16) { if (h->d_rgba) (void)hipFree(h->d_rgba); h->d_rgba = nullptr; PT_TRY(hipMalloc(&h->d_rgba, npx * 16)); h->rgba_cap = npx * 16; }
            L.rgba = (float*)h->d_rgba;
        }
        if (o->linear) {
            if (h->linear_cap < npx * 12) { if (h->d_linear) (void)hipFree(h->d_linear); h->d_linear = nullptr; PT_TRY(hipMalloc(&h->d_linear, npx * 12)); h->linear_cap = npx * 12; }
            L.linear = (float*)h->d_linear;
        }
    }
    // Linear sums only: no recurrence runs from frame to frame, so a strip's frames are cut into runs: the launch is then no longer as long as its most expensive strip (pt_kernels.hip).  Sums are added in run order.
    if (!L.rgba && L.linear && p->n_frames >= 32) {
        // up to 32 runs of >= 8 frames (1080p x 256 frames: 16 x 16 76.2 ms, 32 x 8 72.9, 64 x 4 71.9, 256 x 1 75.3: tools/r03/exp34.sh)
        const int chunks = std::max(1, std::min(32, p->n_frames / 8));
        const size_t bytes = (size_t)chunks * npx * 12;
        if (h->partial_cap < bytes) {
            if (h->d_partial) (void)hipFree(h->d_partial);
            h->d_partial = nullptr; h->partial_cap = 0;
            PT_TRY(hipMalloc(&h->d_partial, bytes));
            h->partial_cap = bytes;
        }
        L.n_chunks = chunks; L.chunk_frames = (p->n_frames + chunks - 1) / chunks; L.partial = (float*)h->d_partial;
    }
    PT_TRY(launch_pt_frames(L, h->stream));
    if (o->memory != 1) {
        if (o->rgba) PT_TRY(hipMemcpyAsync(o->rgba, L.rgba, npx * 16, hipMemcpyDeviceToHost, h->stream));
        if (o->linear) PT_TRY(hipMemcpyAsync(o->linear, L.linear, npx * 12, hipMemcpyDeviceToHost, h->stream));
        PT_TRY(hipStreamSynchronize(h->stream));
    }
    return P3D_OK;
}

int p3d_pt_debug_hash(int device, uint32_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    if (!a || !b || !out) return fail(P3D_ERR_ARG, "NULL argument");
    if (n == 0) return P3D_OK;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    PT_TRY(hipSetDevice(device));
    uint32_t *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void**)&da, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&db, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, n * 4);
    if (e == hipSuccess) e = hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_pt_hash(n, da, db, dout, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) return fail(P3D_ERR_HIP, hipGetErrorString(e));
    return P3D_OK;
}

}  // extern "C"
