/*
 * pathtracing_amd.h -- C ABI of the MI355X render library (libptamd.so).
 *
 * The reference path tracer has no plugin or FFI layer: its render path is
 * entered by plain function calls that read globals (obj, light, tree, gi, rb)
 * and write the global 512x512 BGR image.  Each entry point below names the
 * reference call it stands in for (paths relative to the reference checkout).
 *
 * Conventions: every function returns PT_OK (0) or a negative PtStatus; no C++
 * exception crosses this boundary; pt_last_error() returns a thread-local
 * message for the last failing call on the calling thread.  Pointers in
 * PtSceneDesc are borrowed only for the duration of pt_scene_upload().
 * All host buffers are plain caller-owned memory.
 */
#ifndef PATHTRACING_AMD_H
#define PATHTRACING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument / inconsistent scene description */
    PT_ERR_NO_DEVICE = -2,    /* no usable gfx950 device, or pt_init not called */
    PT_ERR_HIP = -3,          /* a HIP runtime call failed                      */
    PT_ERR_OOM = -4,
    PT_ERR_LIMIT = -5         /* a documented size limit was exceeded           */
} PtStatus;

/* Shading model: the two reference front ends do not compute the same image. */
enum { PT_MODEL_MAIN = 0,     /* main.cpp:41-147   (shadow rays, 4 GI samples)  */
       PT_MODEL_OMP = 1 };    /* mainOMP.cpp:37-136 (no shadows, 16 GI samples) */

/* Arithmetic of the GI sample rays.  Primary, mirror and their shadow rays are
   always traced in fp64 with the reference's operation order. */
enum { PT_GI_F64_EXACT = 0,   /* fp64, reference search semantics, bit-reproducible */
       PT_GI_F32_FAST = 1 };  /* fp32 wavefront queues, nearest-hit kd traversal; with PT_MODEL_MAIN a
                                 shadow-ray queue entry per point and light in front of it (any number
                                 of lights; passes shrink past 16 so that the queue stays below 2^30) */

/* One kd-tree node as built by the host (util/kdtree.cpp:74-82), flattened:
   children are adjacent (right == left + 1). */
typedef struct PtKdNode {
    double lo[3], hi[3];      /* Box::bnds, util/kdtree.cpp:23            */
    double split;             /* KDNode::splt                              */
    int32_t axis;             /* KDNode::axis, -1 for a leaf               */
    int32_t left;             /* index of the left child, -1 for a leaf    */
    int32_t first, count;     /* KDNode::ind as a slice of PtSceneDesc.refs */
} PtKdNode;

typedef struct PtTexture { int32_t rows, cols; const uint8_t *bgr; } PtTexture;      /* cv::Mat, 8UC3 */
typedef struct PtLight {                                   /* util/lighting.cpp:9-15 */
    int32_t type;             /* 'p' point, 'd' directional */
    int32_t reserved;
    double mag, pos[3], color[3];
} PtLight;

/* Scene description handed over by the host reader + kd build.  Triangle
   t belongs to mesh tri_mesh[t] and is triangle tri_local[t] of it, so that
   (mesh, tri) pairs can be reported exactly as the reference's `pii tind`. */
typedef struct PtSceneDesc {
    uint32_t struct_size;     /* sizeof(PtSceneDesc) */
    uint32_t n_mesh, n_vert, n_norm, n_uv, n_tri, n_tex, n_light, n_node, n_ref;
    const double *vert;       /* 3 * n_vert */
    const double *norm;       /* 3 * n_norm */
    const double *uv;         /* 2 * n_uv   */
    const int32_t *tri_vi, *tri_ni, *tri_ti;   /* 3 * n_tri each, indices into the pools, -1 = absent */
    const int32_t *tri_tm;    /* -99999: untextured; k >= 0: texture k, colour interpolated from the
                                 three vertex texels; k < 0: texture -(k+1), texel looked up per hit
                                 (util/polygonMesh.cpp:89-117) */
    const int32_t *tri_reft;  /* 1 = mirror */
    const int32_t *tri_mesh, *tri_local;
    const double *mesh_albedo;/* 3 * n_mesh */
    const PtTexture *tex;
    const PtLight *lights;
    double background[3];     /* main.cpp:16 */
    const PtKdNode *nodes;    /* node 0 is the root */
    const int32_t *refs;      /* n_ref triangle indices (into the n_tri arrays) */
} PtSceneDesc;

typedef struct PtCamera { double eye[3], angle, zoom; } PtCamera;   /* main.cpp:16,19 */

typedef struct PtOptions {
    uint32_t struct_size;     /* sizeof(PtOptions) */
    int32_t gi, rb;           /* the -gi / -rb flags, main.cpp:196-197 */
    int32_t model;            /* PT_MODEL_* */
    int32_t spp;              /* GI samples per pixel, 0 = reference default (4 MAIN / 16 OMP) */
    int32_t gi_precision;     /* PT_GI_* */
    uint32_t seed;            /* key of the counter RNG */
    uint32_t view;            /* second key word: index of the view in a multi-view job */
    int32_t size;             /* image rows = cols; 0 = 512 (main.cpp:14) */
    uint32_t flags;           /* PT_FLAG_* */
    uint32_t part, n_parts;   /* one view split over n_parts workers: this call renders only the 8-row bands b
                                 with b % n_parts == part -- renderParallel() deals the rows of a view round-robin
                                 to its workers (mainOMP.cpp:169-174); here the unit is a band of 8 rows, the
                                 height of the device's pixel tiles.  Every other row of every output is zero
                                 (AOV ids -1).  n_parts 0 or 1: the whole view. */
} PtOptions;

/* Count the nodes visited and triangles tested by the GI closest-hit kernel
   (PT_GI_F32_FAST only).  Slower; used to price the kernel's algorithmic bytes. */
#define PT_FLAG_COUNT_WORK 1u

/* Optional per-pixel outputs of pt_render_view (any pointer may be NULL).
   The primary record is that of the ray whose G-buffer the GI pass uses:
   the last sub-ray for PT_MODEL_MAIN, the centre ray for PT_MODEL_OMP. */
typedef struct PtAov {
    int32_t *mesh, *tri;      /* (mesh, tri) of the primary hit, -1 on a miss */
    double *t, *u, *v;        /* hit distance (-1 on a miss) and barycentrics */
    double *rgb;              /* 3 per pixel, linear, before quantisation     */
    double *gclr, *gnrm, *ghit; /* mclr / mnrm / mhit, main.cpp:32-34         */
} PtAov;

/* Ray counters of one or more views; closest-hit and any-hit queries that the
   image needs.  The second search of mainOMP.cpp:60 is reported apart, once
   per light and hit as the reference performs it, whether or not the device
   had to run it to know its result. */
typedef struct PtStats {
    uint64_t primary, shadow, mirror, gi, omp_retrace;
    double ms_total;          /* wall time of the call                     */
    double ms_gi_trace;       /* device time of the GI closest-hit kernels */
    uint64_t gi_trace_launches;
    uint64_t gi_nodes_visited, gi_tris_tested;   /* only with PT_FLAG_COUNT_WORK */
    double ms_primary;        /* device time of the primary pass (ray generation, fp64 search, shading, mirror bounce) */
} PtStats;

typedef struct PtScene PtScene;

/* Select the device(s) of the calling process: device >= 0 that one; device < 0 every visible device -- scenes are
   then uploaded to the first of them and pt_render_views shards its views over all (the library copies the scene to
   the others with one RCCL broadcast when it first needs them).  mainOMP.cpp:344-346 (MPI_Init, world size) is the
   nearest counterpart.  Fails with PT_ERR_NO_DEVICE when a selected device is not a gfx950 GPU. */
int pt_init(int device);
int pt_shutdown(void);
int pt_device_count(void);    /* devices selected by pt_init, 0 before it */
const char *pt_last_error(void);
const char *pt_version(void);

/* Copies the scene and its kd-tree to the device; stands in for the globals
   filled by init() and buildTree() (main.cpp:200-201, mainOMP.cpp:338-339). */
int pt_scene_upload(const PtSceneDesc *desc, PtScene **out);
int pt_scene_free(PtScene *scene);

/* Sizes of the device copies.  dev_* describe the fp32 copy of the tree walked by
   the GI fast path: the host's nodes plus the nodes made by subdividing fat leaves. */
typedef struct PtSceneInfo {
    uint64_t host_nodes, host_refs, dev_nodes, dev_refs, dev_leaves, dev_max_leaf, dev_interior_refs, dev_bytes;
    int32_t host_depth, dev_depth;
} PtSceneInfo;
int pt_scene_info(const PtScene *scene, PtSceneInfo *out);

/* Same, from / to one self-contained byte blob, so that one process can hand
   the scene to others (e.g. an RCCL broadcast) without re-reading files. */
int pt_scene_blob_size(const PtSceneDesc *desc, size_t *bytes);
int pt_scene_blob_write(const PtSceneDesc *desc, void *blob, size_t bytes);
int pt_scene_upload_blob(const void *blob, size_t bytes, PtScene **out);

/* A built scene as one position-independent block of device data (everything pt_scene_upload prepares, the fp32 tree
   included), for handing it to another GPU or process without preparing it again -- every MPI rank of the reference
   reads the scene and builds the tree for itself (mainOMP.cpp:338-339).  dst / src are host memory, or device memory
   of the current device when the *_is_device flag is set (the payload of an RCCL broadcast can be used in place). */
int pt_scene_slab_size(const PtScene *scene, size_t *bytes);
int pt_scene_slab_export(const PtScene *scene, void *dst, size_t bytes, int dst_is_device);
int pt_scene_upload_slab(const void *src, size_t bytes, int src_is_device, PtScene **out);

/* render(eye, zoom, angle) of main.cpp:109-147 (PT_MODEL_MAIN) or the per-pixel
   render() + row gather of mainOMP.cpp:111-136,185-197 (PT_MODEL_OMP).
   bgr_out receives size*size*3 bytes, exactly what `image` would hold. */
int pt_render_view(PtScene *scene, const PtCamera *cam, const PtOptions *opt,
                   uint8_t *bgr_out, PtAov *aov, PtStats *stats);

/* As pt_render_view but the frame stays on the device: d_bgr is device memory
   of size*size*3 bytes on the current device (for a device-side frame gather). */
int pt_render_view_device(PtScene *scene, const PtCamera *cam, const PtOptions *opt,
                          void *d_bgr, PtStats *stats);

/* The view loop of renderMultiple() (mainOMP.cpp:225-247) for the views this
   process owns: n cameras -> n frames of size*size*3 bytes each, view index
   of camera i = view0 + i * view_stride (used as the RNG key word).
   Views are taken from the list one after another by worker threads (the reference deals rows to MPI workers,
   mainOMP.cpp:169-174; here the unit is the view): after pt_init(-1) every GPU has its own copy of the scene and copies
   its frames into `frames` itself, and on each GPU up to four workers share that copy when a view is small (under
   8 Mi GI sample rays per worker's view, e.g. 512 x 512 at 16 spp), so that several views are in flight and the GPU stays
   full; each worker keeps a workspace of its own.  The frames are the same bytes for any number of GPUs and workers. */
int pt_render_views(PtScene *scene, const PtCamera *cams, size_t n, uint32_t view0, uint32_t view_stride,
                    const PtOptions *opt, uint8_t *frames, PtStats *stats);

/* search() of util/kdtree.cpp:216-257 on n caller-supplied rays: src, dir are
   3 doubles per ray, tmin_in the initial bound (FLT_MAX for a closest-hit
   query, the light distance for a shadow query).  Outputs per ray: found,
   (mesh, tri), t (= tmin_in when nothing was found), u, v.
   mode 0: reference semantics in fp64; mode 1: fp32 nearest-hit traversal
   (the GI fast path; same outputs, t/u/v widened to double); mode 2: mode 0
   followed, for a ray that hit, by the second search of the same ray bounded
   by that hit which mainOMP.cpp:60 makes per light -- the outputs are the
   second search's (found = it found something closer; t = the first hit's
   distance when it did not, with mesh = tri = -1). */
int pt_trace(PtScene *scene, size_t n, const double *src, const double *dir, const double *tmin_in, int mode,
             int32_t *found, int32_t *mesh, int32_t *tri, double *t, double *u, double *v);

/* pt_trace mode 1 (the fp32 nearest-hit search) for callers that cannot allocate: tight serving loops, harnesses
   that forbid hipMalloc.  (Not for a stream under graph capture: the call synchronises its own stream and reads a
   word back.)  It has no counterpart in the reference.  All ten arrays
   are DEVICE memory on the device chosen by pt_init (layouts as in pt_trace; d_mesh / d_tri receive the (mesh, tri)
   pair); the call runs on the scene's stream and returns when the outputs are complete.  It calls no hipMalloc,
   hipFree or other allocating routine: the kernel's scratch (queue heads and the traversal overflow stack) is a
   fixed 32 MiB array inside the library's code object, one copy per GPU for the life of the process.
   Limits, refused with PT_ERR_LIMIT: n <= 2^24 rays per call; fp32 device tree at most 62 levels deep
   (PtSceneInfo.dev_depth).  Every call on a GPU uses that GPU's one array, so the library serialises the calls on a
   GPU, whatever scenes they are for; the caller's current device is left as it was. */
int pt_trace_device(PtScene *scene, size_t n, const double *d_src, const double *d_dir, const double *d_tmin_in,
                    int32_t *d_found, int32_t *d_mesh, int32_t *d_tri, double *d_t, double *d_u, double *d_v);

/* Test helpers: the camera ray directions of one view (3 doubles per pixel;
   sub-pixel offset (x, y) as in main.cpp:116-121) and the hemisphere samples of
   the counter RNG (3 doubles per sample: x, r1, z of util/lighting.cpp:53-59). */
int pt_camera_rays(const PtCamera *cam, int size, double x, double y, double *dir_out);
int pt_hemisphere_samples(uint32_t seed, uint32_t view, uint32_t pixel, int n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACING_AMD_H */
