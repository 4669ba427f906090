/*
 * p3d_pathtracer.h -- C-ABI of the MI355X version of the reference's progressive path tracer
 * (PT/ = GPU_PathTracer_template/: P3D_RT.glsl + common.glsl, a Shadertoy fragment shader;
 * SURVEY.md section 8f row 1, BASELINE config 5).  Part of libp3d_hip.so.
 *
 * The shader has no host code: Shadertoy calls mainImage() (PT/P3D_RT.glsl:286) once per pixel
 * per frame with iTime, iMouse and the previous contents of buffer A.  p3d_pt_render() is that
 * loop for `n_frames` frames: frame j has index k = first_frame + j * frame_stride and
 * iTime = time0 + k * dt; each evaluation seeds the integer-hash RNG from the pixel and iTime
 * (PT/P3D_RT.glsl:288), traces one path (rayColor, :234-282) and
 *   - rgba   accumulates exactly like the shader (PT/P3D_RT.glsl:345-365): gamma-encoded running
 *            mean through toLinear / mix(prev, c, 1/w) / toGamma, frame count in .w;
 *   - linear holds the plain sum of the per-frame linear colours: what ranks reduce (sum) when the
 *            samples of one image are split across GPUs (first_frame = rank, frame_stride = world).
 * Fixed Shadertoy inputs: iMouseButton = 0; iMouse = (mouse_x, mouse_y) in pixels, (0,0) = never
 * clicked (camera at (-10, 0, 8)).
 *
 * PARITY UNPINNED: no GLSL compiler or GL driver exists in the build image and the reference
 * holds no output of this shader except a screenshot, so the only check is agreement with the
 * independent CPU restatement in oracle/pt_oracle.cpp (tests/test_gpu_pathtracer.py).
 */
#ifndef P3D_PATHTRACER_H
#define P3D_PATHTRACER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p3d_pt p3d_pt;

typedef struct p3d_pt_params {
    int32_t res_x, res_y;          /* iResolution.xy                                          */
    int32_t n_frames;              /* mainImage() evaluations per pixel in this call          */
    int32_t first_frame;           /* frame index of the first one                            */
    int32_t frame_stride;          /* index step (1 on one GPU; world size when samples are split) */
    float   time0, dt;             /* iTime of frame k = time0 + k*dt                         */
    float   mouse_x, mouse_y;      /* iMouse.xy in pixels                                     */
} p3d_pt_params;

/* rgba: [res_y][res_x][4] floats (row 0 = bottom, like gl_FragCoord), linear: [res_y][res_x][3];
 * either may be NULL. memory: 0 = host pointers (call returns when done), 1 = device pointers
 * (call only enqueues on the handle's stream). */
typedef struct p3d_pt_outputs {
    float*  rgba;
    float*  linear;
    int32_t memory;
} p3d_pt_outputs;

int p3d_pt_create(int device, p3d_pt** out);
int p3d_pt_destroy(p3d_pt* pt);
int p3d_pt_set_stream(p3d_pt* pt, void* hip_stream);
int p3d_pt_render(p3d_pt* pt, const p3d_pt_params* params, const p3d_pt_outputs* out);
int p3d_pt_sync(p3d_pt* pt);
/* HIP-event bracket on the handle's stream, like p3d_timer_begin/end */
int p3d_pt_timer_begin(p3d_pt* pt);
int p3d_pt_timer_end(p3d_pt* pt, float* elapsed_ms);
/* Multi-GPU form: the samples of one image are split over the ranks (first_frame =
 * rank, frame_stride = world) and the `linear` sums are added up on rank 0: ncclReduce(sum) over `comm`
 * (see p3d_hip.h), in place on a device buffer of `count` floats, enqueued on the handle's stream. */
struct p3d_comm;
int p3d_pt_reduce_sum(struct p3d_comm* comm, p3d_pt* pt, float* linear, uint64_t count);
/* integer hash of PT/common.glsl:31-36 evaluated on the device (known-answer probe) */
int p3d_pt_debug_hash(int device, uint32_t n, const uint32_t* a, const uint32_t* b, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
