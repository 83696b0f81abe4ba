// Per-view parameters, frame buffers and the host-side launch functions that
// the C ABI layer calls.  Kernels live in pt_exact.hip (fp64, built without
// multiply-add contraction) and pt_fast.hip (fp32 GI wavefront path).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_scene_dev.h"

struct ViewParams {
    double eye[3];
    double zoom, sa, ca;      // sa, ca: sin / cos of the view angle (host libm)
    int32_t size;             // image rows = cols
    int32_t rb;               // 3x3 ray bundle
    double off;               // bundle offset: 0.01 (main.cpp:116) or 0.1 (mainOMP.cpp:117)
    uint32_t seed, view;
    int32_t model;
    uint32_t part, n_parts;   // rows of this call: 8-row bands b with b % n_parts == part (n_parts >= 1)
};

// Does image row i belong to this call?
__host__ __device__ inline bool own_row(uint32_t part, uint32_t n_parts, int i) {
    return n_parts <= 1u || ((uint32_t)i >> 3) % n_parts == part;
}

struct FrameBuffers {
    double *mpnt;             // 3 per pixel: direct pass result, main.cpp:31
    double *gclr, *gnrm, *ghit;   // G-buffers, main.cpp:32-34
    int32_t *rec_found, *rec_tri;
    double *rec_t, *rec_u, *rec_v;
    double *gi_acc;           // 3 per pixel: running `indirect` of globalIllumination
    double *contrib;          // 3 per (sample of the batch, pixel): r1 * castRay(iray)
    double *rgb;              // 3 per pixel, final linear colour
    uint8_t *bgr;             // 3 per pixel
    unsigned long long *counters;   // RC_N ray counters
};


// Outputs of the primary pass per sub-ray (one sub-ray without -rb, nine with); the combine
// kernel folds them into the frame buffers in the reference's order.  Index = pixel * nsub + sub.
struct SubBuffers {
    double *paint;             // 3: castRay's value
    double *g;                 // 9: mclr, mnrm, mhit candidate
    int32_t *found, *tri;
    double *tuv;               // 3: t, u, v
    int32_t *again;            // mainOMP model: the second search of mainOMP.cpp:60 has to be run for this sub-ray
    int32_t *mirror;           // the hit is reflective: `paint` holds the hit's own term until the bounce is added
    double *mdir;              // 3: direction of the mirror ray (its origin is the hit point in `g`)
    uint32_t *again_list;      // the sub-rays with `again` set, in no particular order (counters[RC_AGAIN] of them)
    float4 *forms;             // per view, not per sub-ray: the packet test's table, 4 arrays of n_ref entries (pt_packet.h)
};

// Queues and slots of the fp32 GI wavefront path (pt_fast.hip).
struct FastBuffers {
    float4 *rays_o;           // origin xyz, index bits of the triangle the ray starts on (never reported as a hit)
    float4 *rays_d;           // direction xyz, weight r1 (cos theta)
    uint32_t *ray_slot;       // the ray's code: sample_in_pass * 64 + pixel of its 8x8 tile
    float4 *hits;             // t, u, v, triangle index bits (-1: miss)
    float4 *sp_pos;           // shading points (MAIN model): position xyz, triangle index bits
    float4 *sp_nrm;           // normal xyz, queue index of the ray (bits)
    float4 *sp_base;          // r1 * 0.6 * txt * albedo / pi, index bits of the point's first shadow ray
    float4 *sh_o, *sh_d;      // shadow rays (MAIN model), a point's rays one after the other in light order: origin xyz, index bits of
                              // the point's triangle | direction xyz, distance to the light -- replaced by 1.0f / 0.0f (light
                              // visible / not) when the ray has been traced
    float4 *contrib;          // per queue entry: r1 * radiance of the sample ray
    uint32_t *counts;         // [0] rays in the queue, [1] shading points, [2] shadow rays
    uint32_t *tile_base;      // per 8x8 tile in queue order: pixels with a primary hit before it (exclusive prefix sum; total at the end)
    unsigned long long *totals;   // [0] GI rays, [1] shadow rays of the view so far
    uint32_t *heads;          // queue heads of the closest-hit kernel, then of the shadow kernel (pt_persistent.h)
    unsigned long long *work; // [0] nodes visited, [1] triangles tested (counting build of the closest-hit kernel)
    size_t max_rays;
};

// One wave handles one 8x8 pixel tile, so that the rays of a wave stay close.
static inline int tile_threads(int size) { int t = (size + 7) >> 3; return t * t * 64; }

// ---- pt_exact.hip ----
#define PT_EXACT_WAVES 4096u       // one-wave blocks of a launch of an fp64 search kernel (each owns a stack overflow region)
// s2, ev_a, ev_b (optional): a second stream and two events for the part of the pass that can run beside the rest
int exact_primary(const DevScene &S, const ViewParams &V, const FrameBuffers &F, const SubBuffers &B, hipStream_t s, hipStream_t s2 = nullptr,
                  hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr);
int exact_gi(const DevScene &S, const ViewParams &V, const FrameBuffers &F, int s0, int nb, hipStream_t s);
int gi_accumulate(const FrameBuffers &F, int size, int nb, int first_batch, hipStream_t s);
int resolve_frame(const FrameBuffers &F, int size, int gi, int nsamples, hipStream_t s);
int exact_trace(const DevScene &S, size_t n, const double *src, const double *dir, const double *tmin_in,
                int32_t *found, int32_t *tri, double *t, double *u, double *v, int research, hipStream_t s);
int camera_rays(const ViewParams &V, double x, double y, double *dir, hipStream_t s);
int hemisphere_samples(uint32_t seed, uint32_t view, uint32_t pixel, int n, double *out, hipStream_t s);

// ---- pt_fast.hip ----
#define PT_TRACE_WAVES (256 * 32)   // waves of a trace launch: enough to fill every CU; surplus waves find the queue empty
// Rays of one GI pass (one launch of every stage).  Larger passes amortise the ramp and the tail of the persistent
// traversal launch and put more samples of a tile next to each other in the sorted queue: 16 Mi -> 64 Mi rays per
// pass is +2.3 % whole-job on the 1 024-spp benchmark (128 Mi: another +0.3 to +0.8 %, 256 Mi: nothing more); the queues of a 64 Mi-ray pass
// take 11 GB.  PT_TUNE_PASS_RAYS overrides it (tests use a small value to force several passes).
#define PT_FAST_RAYS_PER_PASS (64u << 20)
typedef int (*PtAllocFn)(void *ctx, size_t bytes, void **out);
int fast_alloc(FastBuffers Q[2], size_t pixels, int batch, int n_light, PtAllocFn alloc, void *ctx);
// All GI passes of one view: fills F.gi_acc.  trace_ms / launches: device time and
// count of the closest-hit traversal launches; rays / shadow: query counts.
int fast_gi(const DevScene &S, const ViewParams &V, const FrameBuffers &F, const FastBuffers Q[2], int nsamples, int batch,
            hipStream_t s, hipStream_t s_gen, std::vector<hipEvent_t> &events, float *trace_ms, uint64_t *launches, uint64_t *rays, uint64_t *shadow,
            uint64_t *work);
int fast_trace(const DevScene &S, size_t n, const double *src, const double *dir, const double *tmin_in,
               int32_t *found, int32_t *tri, double *t, double *u, double *v, hipStream_t s);
// The same search on device-resident rays with its scratch (queue heads, overflow stack) in a fixed array of the
// code object: no allocation.  Sizes of that array: waves of a launch, stack entries per lane (>= fp32 tree depth + 2),
// the most rays one call takes, and the device ordinals the library keeps an address for.
#define PT_STATIC_WAVES 1024u
#define PT_STATIC_DEPTH 64u
#define PT_STATIC_MAX_RAYS ((size_t)1 << 24)
#define PT_STATIC_MAX_DEVICES 64
int static_scratch_setup(int device);      // once per device, with that device current (pt_init)
int fast_trace_static(const DevScene &S, int device, size_t n, const double *src, const double *dir, const double *tmin_in,
                      int32_t *found, int32_t *mesh, int32_t *tri, double *t, double *u, double *v, hipStream_t s);
