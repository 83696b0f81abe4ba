// Device-resident scene: what the kernels see.  Filled by the upload code in
// pt_abi.hip from a PtSceneDesc; all pointers are device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pathtracing_amd.h"

// Traversal half of a kd node (fp64 path); the box is kept apart because it is
// only read when a node's own triangle produces a candidate hit.
struct NodeA {
    double split;
    int32_t axis;       // -1: leaf
    int32_t left;       // right child = left + 1
    int32_t first;      // first entry of the node's slice in the ref arrays
    int32_t count;
};

// fp32 node of the GI fast path, 16 bytes, four to a 64-byte line (layout and
// encoding: pt_trace_f32.h).  While the tree is being built `link` holds
// (left child index | axis << 30) and first/count index the reference list;
// encode_node() turns that into the byte-offset form the kernels read.
struct NodeF {
    float split;
    uint32_t link;      // built: left | axis << 30, leaf 0xFFFFFFFF; encoded: left * 16 | axis, leaf 0xFFFFFFFF
    uint32_t first;     // built: first reference;              encoded: byte offset of the first triangle record
    uint32_t count;     // built: number of references;         encoded: byte offset one past the last record
};
static inline NodeF encode_node(NodeF n) {
    NodeF o = n;
    if (n.link != 0xFFFFFFFFu) o.link = ((n.link & 0x3FFFFFFFu) << 4) | (n.link >> 30);
    o.first = n.first * 48u;
    o.count = (n.first + n.count) * 48u;
    return o;
}

struct DevScene {
    // ---- fp64 reference-semantics path ----
    const NodeA *nodeA;
    const double *nodeBox;     // 6 per node: lo, hi
    const double *refTri;      // 9 per ref: v0, v1 - v0, v2 - v0
    const int32_t *refId;      // triangle index per ref
    double *ovf64;             // overflow of the fp64 search's stack: [block][entry][field][lane], ovf64_entries entries of 3 doubles
    uint32_t ovf64_entries;    //   per lane for each of the PT_EXACT_WAVES blocks the fp64 kernels are launched with
    // ---- fp32 fast path ----
    const NodeF *nodeF;
    const uint4 *quadF;        // the same tree as quad records (pt_trace_quad.h); has_quads: it exists
    int32_t has_quads;
    const float4 *refTriF;     // 3 float4 per ref: (v0, id bits), (e2.x, e1.x, e2.y, e1.y), (e2.z, e1.z, 0, 0)
    uint2 *ovf;                // traversal overflow stack: [wave][entry][lane], ovf_depth entries per lane, ovf_waves waves
    uint32_t ovf_depth, ovf_waves;
    const double *triGeo;      // 9 per triangle (by triangle index): v0, v1 - v0, v2 - v0
    const float4 *triNgF;      // per triangle: unit geometric normal (e1 x e2), 0
    // ---- shading data ----
    const int32_t *tri_ni, *tri_ti, *tri_tm, *tri_reft, *tri_mesh, *tri_local;
    const double *norm, *uv, *albedo;
    // GI fast path: everything surfaceProperties needs of a triangle in one 64-byte record (4 float4, four requests):
    // n0.xyz n1.x | n1.yz n2.xy | n2.z uv0.xy uv1.x | uv1.y uv2.xy code
    // (a missing uv index is stored as (0, 0), which is what the reference uses for it).  code = mesh << tex_bits | t with
    // t = 0: no texture, 1 + k: texture k looked up at the three vertices, 1 + n_tex + k: texture k looked up at the hit.
    // meshAlbF: albedo per mesh (rgb, 0).  texF: per texture (rows, cols, index of its first texel in texels4, 0);
    // texels4: one word per texel, B | G << 8 | R << 16, so a texel is one request instead of three.
    const float4 *triShadeF;
    const float4 *meshAlbF;
    const int4 *texF;
    const uint32_t *texels4;
    int32_t n_tex, tex_bits;
    const uint8_t *texels;
    const int64_t *tex_off;
    const int32_t *tex_rows, *tex_cols;
    const PtLight *lights;
    const float4 *lightsF;     // fp32 lights, two float4 each: (position, type), (mag * colour, 0)
    int32_t n_light, n_node, n_ref, n_tri;
    // Limits a kernel ran into: bit 0 the fp32 traversal's overflow stack was full (an entry was lost), bit 1 a ray was
    // ended by the bound on node steps, bit 2 the fp64 search's overflow region was full.  The host reads the word after
    // every call and reports PT_ERR_LIMIT; a result produced under a tripped limit is never handed out as good.
    unsigned long long *err;
    int32_t any_mirror;        // some triangle is reflective (the mirror pass of the primary kernel is skipped otherwise)
    double bg[3];
    double root_lo[3], root_hi[3];
};

// The immutable arrays of a DevScene live in ONE device allocation (the scene slab); these are the pointer fields that
// point into it.  A slab is position independent: its header holds a DevScene whose pointer fields are byte offsets
// from the slab's start, so a copy of the slab on another GPU (or in another process) becomes a scene by adding its
// own base address -- no host preprocessing, no second build of the fp32 tree.
#define PT_SLAB_POINTERS(X) \
    X(nodeA) X(nodeBox) X(refTri) X(refId) X(nodeF) X(quadF) X(refTriF) X(triGeo) X(triNgF) X(tri_ni) X(tri_ti) X(tri_tm) X(tri_reft) \
    X(tri_mesh) X(tri_local) X(norm) X(uv) X(albedo) X(triShadeF) X(texels) X(tex_off) X(tex_rows) X(tex_cols) X(lights) X(lightsF) \
    X(meshAlbF) X(texF) X(texels4)

enum { PT_LIMIT_F32_STACK = 1, PT_LIMIT_F32_STEPS = 2, PT_LIMIT_F64_STACK = 4 };
// ray counters of a view, the work counters the persistent primary / mirror / shading kernels draw their blocks from, the limit
// word, and the length of the list of sub-rays whose second search (mainOMP.cpp:60) has to be run
enum { RC_PRIMARY = 0, RC_SHADOW = 1, RC_MIRROR = 2, RC_GI = 3, RC_RETRACE = 4, RC_WORK_PRIMARY = 5, RC_WORK_MIRROR = 6, RC_LIMITS = 7, RC_WORK_SHADE = 8, RC_AGAIN = 9, RC_N = 10 };
// The packets of the primary search are handed out from RC_HEAD_REGIONS work counters, one per contiguous band of the
// image, each in a cache line of its own behind the counters above (updates of ONE word run at some 90 per microsecond
// on the whole chip: at 37 888 packets per view a single counter was a quarter of the kernel's time on small scenes).
// Words 1 .. 5 of a band's line: the rays per class counted by the blocks of that group (blockIdx % 8).
enum { RC_HEADS = 32, RC_HEAD_STRIDE = 16, RC_HEAD_REGIONS = 8, RC_WORDS = RC_HEADS + RC_HEAD_REGIONS * RC_HEAD_STRIDE };
