// C ABI of libptamd.so (include/pathtracing_amd.h): scene upload, view
// rendering and raw ray queries on one MI355X.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pathtracing_amd.h"
#include "pt_frame.h"
#include "pt_refine.h"

#define PT_STACK_MAX 64   // per-lane stack of the fp64 search (pt_exact.h)

namespace {

thread_local std::string g_error;
int g_device = -1;

int fail(int code, const std::string &msg) { g_error = msg; return code; }

// Every entry point that may allocate host memory runs inside this: nothing is thrown across the C boundary.
template <typename F>
int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try { return fail(PT_ERR_OOM, "out of host memory"); } catch (...) { return PT_ERR_OOM; }
    } catch (const std::exception &e) {
        try { return fail(PT_ERR_INVALID, std::string("unexpected exception: ") + e.what()); } catch (...) { return PT_ERR_INVALID; }
    } catch (...) {
        return PT_ERR_INVALID;
    }
}

float bits_to_float(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP,                      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

struct Workspace {
    int size = 0, batch = 0, nsub = 0;
    bool shadow = false;              // the shadow-ray queue of the main.cpp model's fp32 GI path is allocated
    FrameBuffers F{};
    FastBuffers Q[2] = {};
    SubBuffers B{};
    std::vector<void *> owned;
};

}  // namespace

// Header of a scene slab (the one device allocation that holds every immutable array of a scene).
#define PT_SLAB_MAX_ARRAYS 32
struct SlabHeader {
    char magic[8];                    // "PTSLAB04"
    uint64_t total_bytes;             // header + arrays
    uint64_t array_bytes[PT_SLAB_MAX_ARRAYS];   // size of each array, in the order of PT_SLAB_POINTERS
    uint64_t body_check;              // slab_body_check() of everything after the header
    DevScene S;                       // pointer fields of PT_SLAB_POINTERS: byte offsets from the slab's start; scratch pointers null
    RefineStats refine;
    int32_t max_depth;                // host tree
    uint32_t n_tri, n_mesh;
    uint64_t check;                   // header_check() of this header with this field zero
};

struct PtScene {
    DevScene S{};
    int device = 0;
    uint8_t *slab = nullptr;          // device memory: SlabHeader, then the arrays
    size_t slab_bytes = 0;
    std::vector<void *> owned;        // the scene's scratch allocations (overflow stacks, limit word)
    Workspace ws;
    hipStream_t stream = nullptr;
    hipStream_t stream_gen = nullptr;         // ray generation of the next GI pass runs here, beside the current pass
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;   // around the primary pass of a view
    hipEvent_t ev_a = nullptr, ev_b = nullptr;     // fork and join of the part of the primary pass that runs on stream_gen
    std::vector<hipEvent_t> pass_events;      // a pair per GI pass of a view (fp32 path), grown on demand
    int max_depth = 0;
    RefineStats refine;
    uint64_t dev_bytes = 0;
    uint32_t n_tri = 0, n_mesh = 0;
    std::vector<int32_t> tri_mesh, tri_local;   // host copies for (mesh, tri) reporting
    std::vector<PtScene *> replicas;            // copies of this scene on the other initialised devices (pt_render_views)
    std::vector<PtScene *> sharers;             // further view workers on this device: own scratch, streams and workspace around this scene's slab
    bool slab_borrowed = false;                 // the slab belongs to the scene this one is a sharer of
    std::mutex busy;                            // calls on one scene are serialised: it has one workspace, one stream pair
};

// Checksum of a slab's body (everything after the header), the same number on the host and on the device: the sum over
// the body's 64-bit words of a mix of (word, position).  A sum, so that any number of threads can each take a part.
__host__ __device__ inline uint64_t slab_word_mix(uint64_t w, uint64_t i) {
    uint64_t z = w + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_slab_body_check(const uint64_t *words, size_t n, unsigned long long *out) {
    uint64_t a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += slab_word_mix(words[i], i);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor((unsigned long long)a, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)a);
}

namespace {

std::vector<int> g_devices;

constexpr size_t kSlabBody = (sizeof(SlabHeader) + 255) & ~(size_t)255;       // where the arrays begin

uint64_t slab_body_check_host(const uint8_t *slab, size_t total) {
    const size_t n = (total - kSlabBody) / 8;
    const uint8_t *body = slab + kSlabBody;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    std::vector<uint64_t> part(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            uint64_t a = 0;
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) { uint64_t w; std::memcpy(&w, body + 8 * i, 8); a += slab_word_mix(w, i); }
            part[t] = a;
        });
    uint64_t a = 0;
    for (unsigned t = 0; t < nt; t++) { th[t].join(); a += part[t]; }
    return a;
}

// The same over a slab in memory of the current device.
int slab_body_check_device(const uint8_t *slab, size_t total, uint64_t *out) {
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, sizeof *d) != hipSuccess) return fail(PT_ERR_OOM, "out of device memory (slab check)");
    hipError_t e = hipMemset(d, 0, sizeof *d);
    const size_t n = (total - kSlabBody) / 8;
    if (e == hipSuccess && n) {
        hipLaunchKernelGGL(k_slab_body_check, dim3(2048), dim3(256), 0, 0, (const uint64_t *)(slab + kSlabBody), n, d);
        e = hipGetLastError();
    }
    unsigned long long v = 0;
    if (e == hipSuccess) e = hipMemcpy(&v, d, sizeof v, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PT_ERR_HIP, std::string("slab check: ") + hipGetErrorString(e));
    *out = v;
    return PT_OK;
}           // the devices pt_init selected; g_device is the first of them

// FNV-1a over the header's bytes as they lie in memory (padding included: a header is only ever moved with memcpy),
// the check field itself left out.
uint64_t header_check(const SlabHeader &h) {
    const uint8_t *p = (const uint8_t *)&h;
    const size_t skip = offsetof(SlabHeader, check);
    uint64_t a = 0xCBF29CE484222325ull;
    for (size_t i = 0; i < sizeof h; i++)
        if (i < skip || i >= skip + sizeof h.check) a = (a ^ p[i]) * 0x100000001B3ull;
    return a;
}

// Host image of a slab under construction: arrays are appended at 256-byte boundaries and their offsets recorded in the
// header's DevScene.
struct SlabBuilder {
    std::vector<uint8_t> bytes;
    std::vector<std::pair<size_t, size_t>> placed;         // (offset, bytes) of every array added
    SlabHeader hdr;
    SlabBuilder() {
        std::memset((void *)&hdr, 0, sizeof hdr);          // padding bytes too: the header is hashed as bytes
        bytes.resize((sizeof(SlabHeader) + 255) & ~(size_t)255);
    }
    template <typename T>
    void add(const T *src, size_t n, const T *DevScene::*field) {
        const size_t off = bytes.size(), nb = n * sizeof(T);
        bytes.resize((off + (nb ? nb : 1) + 255) & ~(size_t)255);
        if (nb) std::memcpy(bytes.data() + off, src, nb);
        placed.push_back({off, nb});
        hdr.S.*field = (const T *)(uintptr_t)off;
    }
};
template <typename T>
int ws_alloc(Workspace &ws, size_t n, T **out) {
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, (n ? n : 1) * sizeof(T)));
    ws.owned.push_back(d);
    *out = (T *)d;
    return PT_OK;
}

int ws_alloc_cb(void *ctx, size_t bytes, void **out) {
    Workspace &ws = *(Workspace *)ctx;
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    ws.owned.push_back(*out);
    return PT_OK;
}

void ws_release(Workspace &ws) {
    for (void *p : ws.owned) (void)hipFree(p);
    ws = Workspace();
}

int validate_desc(const PtSceneDesc *d) {
    if (!d) return fail(PT_ERR_INVALID, "scene description is null");
    if (d->struct_size != sizeof(PtSceneDesc)) return fail(PT_ERR_INVALID, "PtSceneDesc.struct_size mismatch");
    if (d->n_node == 0 || !d->nodes) return fail(PT_ERR_INVALID, "scene has no kd-tree (n_node == 0)");
    if (d->n_tri && (!d->tri_vi || !d->tri_ni || !d->tri_ti || !d->tri_tm || !d->tri_reft || !d->tri_mesh || !d->tri_local))
        return fail(PT_ERR_INVALID, "triangle arrays missing");
    if ((d->n_vert && !d->vert) || (d->n_norm && !d->norm) || (d->n_uv && !d->uv) || (d->n_mesh && !d->mesh_albedo) ||
        (d->n_tex && !d->tex) || (d->n_light && !d->lights) || (d->n_ref && !d->refs))
        return fail(PT_ERR_INVALID, "a pool pointer is null while its count is not zero");
    for (uint32_t t = 0; t < d->n_tri; t++) {
        for (int k = 0; k < 3; k++) {
            int32_t v = d->tri_vi[3 * t + k], n = d->tri_ni[3 * t + k], u = d->tri_ti[3 * t + k];
            if (v < 0 || (uint32_t)v >= d->n_vert || n < 0 || (uint32_t)n >= d->n_norm || u < -1 ||
                (u >= 0 && (uint32_t)u >= d->n_uv))
                return fail(PT_ERR_INVALID, "triangle " + std::to_string(t) + ": index out of range");
        }
        int32_t tm = d->tri_tm[t];
        if (tm != -99999) {
            int32_t k = tm >= 0 ? tm : -(tm + 1);
            if ((uint32_t)k >= d->n_tex) return fail(PT_ERR_INVALID, "triangle " + std::to_string(t) + ": texture index out of range");
            if (tm < 0)
                for (int c = 0; c < 3; c++)
                    if (d->tri_ti[3 * t + c] < 0)
                        return fail(PT_ERR_INVALID, "triangle " + std::to_string(t) + ": per-hit texture without uv");
        }
        if (d->tri_mesh[t] < 0 || (uint32_t)d->tri_mesh[t] >= d->n_mesh)
            return fail(PT_ERR_INVALID, "triangle " + std::to_string(t) + ": mesh index out of range");
    }
    for (uint32_t r = 0; r < d->n_ref; r++)
        if (d->refs[r] < 0 || (uint32_t)d->refs[r] >= d->n_tri) return fail(PT_ERR_INVALID, "kd reference out of range");
    for (uint32_t k = 0; k < d->n_tex; k++)
        if (d->tex[k].rows <= 0 || d->tex[k].cols <= 0 || !d->tex[k].bgr) return fail(PT_ERR_INVALID, "empty texture");
    return PT_OK;
}

// Depth of the tree below node 0 and structural sanity (children in range, every
// node reachable once, slices inside the ref array).
int tree_depth(const PtSceneDesc *d, int *depth_out) {
    std::vector<std::pair<int32_t, int>> todo;
    std::vector<char> seen(d->n_node, 0);
    todo.push_back({0, 0});
    int maxd = 0;
    while (!todo.empty()) {
        auto [k, dep] = todo.back();
        todo.pop_back();
        if (k < 0 || (uint32_t)k >= d->n_node || seen[k]) return fail(PT_ERR_INVALID, "kd-tree links are not a tree");
        seen[k] = 1;
        const PtKdNode &n = d->nodes[k];
        if (n.count < 0 || n.first < 0 || (uint64_t)n.first + (uint64_t)n.count > d->n_ref)
            return fail(PT_ERR_INVALID, "kd node slice outside the reference array");
        if (dep > maxd) maxd = dep;
        if (n.axis >= 0) {
            if (n.axis > 2) return fail(PT_ERR_INVALID, "kd node axis out of range");
            todo.push_back({n.left, dep + 1});
            todo.push_back({n.left + 1, dep + 1});
        }
    }
    *depth_out = maxd;
    return PT_OK;
}

// Everything the host prepares of a scene, as the host image of its slab: the fp64 tree and reference lists, the fp32
// tree built from the triangles (pt_refine.hip), per-triangle shading records, textures, lights.
int build_slab(const PtSceneDesc *d, SlabBuilder &B) {
    int rc = validate_desc(d);
    if (rc) return rc;
    int depth = 0;
    rc = tree_depth(d, &depth);
    if (rc) return rc;
    if (depth + 1 > PT_STACK_MAX)
        return fail(PT_ERR_LIMIT, "kd-tree depth " + std::to_string(depth) + " exceeds the traversal stack (" +
                                      std::to_string(PT_STACK_MAX) + ")");
    SlabHeader &H = B.hdr;
    H.max_depth = depth;
    H.n_tri = d->n_tri; H.n_mesh = d->n_mesh;
    DevScene &S = H.S;

    // ---- tree: fp64 traversal nodes + boxes ----
    std::vector<NodeA> na(d->n_node);
    std::vector<double> nbox((size_t)d->n_node * 6);
    for (uint32_t k = 0; k < d->n_node; k++) {
        const PtKdNode &n = d->nodes[k];
        na[k].split = n.split; na[k].axis = n.axis; na[k].left = n.left; na[k].first = n.first; na[k].count = n.count;
        for (int c = 0; c < 3; c++) { nbox[6 * (size_t)k + c] = n.lo[c]; nbox[6 * (size_t)k + 3 + c] = n.hi[c]; }
    }
    // ---- fp32 tree (pt_refine.hip) ----
    std::vector<NodeF> nf;
    std::vector<int32_t> frefs;
    RefineParams rp;
    // PT_TUNE_REFINE: 0 = the host tree as it is, 1 = host tree with subdivided leaves, 2 (default) = rebuilt
    if (const char *e = getenv("PT_TUNE_REFINE")) { rp.enable = atoi(e) != 0; rp.rebuild = atoi(e) == 2; }
    if (const char *e = getenv("PT_TUNE_CI")) rp.ci = (float)atof(e);
    if (const char *e = getenv("PT_TUNE_LEAF")) rp.leaf_keep = atoi(e);
    if (const char *e = getenv("PT_TUNE_EMPTY")) rp.empty_bonus = (float)atof(e);
    if (const char *e = getenv("PT_TUNE_CLIP")) rp.clip = atoi(e) != 0;
    if (const char *e = getenv("PT_TUNE_DEPTH")) { int k = atoi(e); if (k >= 8 && k <= 60) rp.max_depth = k; }
    {
        const RefineStats r = refine_tree(d, rp, nf, frefs);    // member by member: a struct copy need not keep H's zeroed padding
        H.refine.nodes = r.nodes; H.refine.refs = r.refs; H.refine.leaves = r.leaves; H.refine.leaf_refs = r.leaf_refs;
        H.refine.interior_refs = r.interior_refs; H.refine.new_nodes = r.new_nodes; H.refine.max_leaf = r.max_leaf;
        H.refine.max_depth = r.max_depth;
    }
    if (nf.size() >= (1u << 28) || (uint64_t)frefs.size() * 48u >= (1ull << 32))
        return fail(PT_ERR_LIMIT, "kd-tree too large for 32-bit byte offsets (2^28 nodes, 2^32 / 48 references)");
    // ---- per-reference triangle data in node order: v0, v1 - v0, v2 - v0 ----
    std::vector<double> rt((size_t)d->n_ref * 9);
    auto tri9 = [&](int32_t t, double *o) {
        const double *a = d->vert + 3 * (size_t)d->tri_vi[3 * (size_t)t];
        const double *b = d->vert + 3 * (size_t)d->tri_vi[3 * (size_t)t + 1];
        const double *c = d->vert + 3 * (size_t)d->tri_vi[3 * (size_t)t + 2];
        for (int k = 0; k < 3; k++) { o[k] = a[k]; o[3 + k] = b[k] - a[k]; o[6 + k] = c[k] - a[k]; }
    };
    for (uint32_t r = 0; r < d->n_ref; r++) tri9(d->refs[r], &rt[(size_t)r * 9]);
    const std::vector<uint4> quads = quad_records(nf);
    if ((uint64_t)quads.size() * 16u >= (1ull << 32)) return fail(PT_ERR_LIMIT, "kd-tree too large for 32-bit byte offsets (quad records)");
    for (NodeF &n : nf) n = encode_node(n);
    // ---- per-triangle geometry by triangle index (origin-triangle tests of the GI fast path) ----
    std::vector<double> tgeo((size_t)d->n_tri * 9);
    std::vector<float4> tng(d->n_tri);
    for (uint32_t t = 0; t < d->n_tri; t++) {
        double *o = &tgeo[(size_t)t * 9];
        tri9((int32_t)t, o);
        double nx = o[4] * o[8] - o[5] * o[7], ny = o[5] * o[6] - o[3] * o[8], nz = o[3] * o[7] - o[4] * o[6];
        double l = std::sqrt(nx * nx + ny * ny + nz * nz);
        l = l > 0 ? 1.0 / l : 0.0;
        tng[t] = make_float4((float)(nx * l), (float)(ny * l), (float)(nz * l), 0.f);
    }
    std::vector<float4> rtf(frefs.size() * 3);
    for (size_t r = 0; r < frefs.size(); r++) {
        double o[9];
        tri9(frefs[r], o);
        rtf[3 * r] = make_float4((float)o[0], (float)o[1], (float)o[2], bits_to_float(frefs[r]));
        // the two edges interleaved, e2 first: the pairs the packed-fp32 triangle test multiplies
        rtf[3 * r + 1] = make_float4((float)o[6], (float)o[3], (float)o[7], (float)o[4]);
        rtf[3 * r + 2] = make_float4((float)o[8], (float)o[5], 0.f, 0.f);
    }
    // ---- textures flattened into one byte pool ----
    std::vector<int64_t> toff(d->n_tex);
    std::vector<int32_t> trows(d->n_tex), tcols(d->n_tex);
    std::vector<uint8_t> texels;
    for (uint32_t k = 0; k < d->n_tex; k++) {
        toff[k] = (int64_t)texels.size();
        trows[k] = d->tex[k].rows; tcols[k] = d->tex[k].cols;
        size_t nb = (size_t)trows[k] * tcols[k] * 3;
        texels.insert(texels.end(), d->tex[k].bgr, d->tex[k].bgr + nb);
    }
    B.add(na.data(), na.size(), &DevScene::nodeA);
    B.add(nbox.data(), nbox.size(), &DevScene::nodeBox);
    B.add(nf.data(), nf.size(), &DevScene::nodeF);
    B.add(quads.data(), quads.size(), &DevScene::quadF);
    S.has_quads = quads.empty() ? 0 : 1;
    B.add(rt.data(), rt.size(), &DevScene::refTri);
    B.add(rtf.data(), rtf.size(), &DevScene::refTriF);
    B.add(tgeo.data(), tgeo.size(), &DevScene::triGeo);
    B.add(tng.data(), tng.size(), &DevScene::triNgF);
    B.add(d->refs, (size_t)d->n_ref, &DevScene::refId);
    B.add(d->tri_ni, (size_t)d->n_tri * 3, &DevScene::tri_ni);
    B.add(d->tri_ti, (size_t)d->n_tri * 3, &DevScene::tri_ti);
    B.add(d->tri_tm, (size_t)d->n_tri, &DevScene::tri_tm);
    B.add(d->tri_reft, (size_t)d->n_tri, &DevScene::tri_reft);
    B.add(d->tri_mesh, (size_t)d->n_tri, &DevScene::tri_mesh);
    B.add(d->tri_local, (size_t)d->n_tri, &DevScene::tri_local);
    B.add(d->norm, (size_t)d->n_norm * 3, &DevScene::norm);
    {
        // the fp32 shading record: mesh and texture mode share one word (see DevScene::triShadeF)
        int tex_bits = 1;
        while ((1u << tex_bits) < 2u * d->n_tex + 1u) tex_bits++;
        if (tex_bits > 24 || (uint64_t)d->n_mesh >= (1ull << (31 - tex_bits)))
            return fail(PT_ERR_LIMIT, "more meshes and textures than the fp32 shading record has bits for");
        S.n_tex = (int32_t)d->n_tex; S.tex_bits = tex_bits;
        std::vector<float4> sh((size_t)d->n_tri * 4);
        for (uint32_t t = 0; t < d->n_tri; t++) {
            float n[9], uv[6];
            for (int c = 0; c < 3; c++) {
                const double *nn = d->norm + 3 * (size_t)d->tri_ni[3 * (size_t)t + c];
                for (int k = 0; k < 3; k++) n[3 * c + k] = (float)nn[k];
                const int32_t ui = d->tri_ti[3 * (size_t)t + c];
                uv[2 * c] = ui >= 0 ? (float)d->uv[2 * (size_t)ui] : 0.f;
                uv[2 * c + 1] = ui >= 0 ? (float)d->uv[2 * (size_t)ui + 1] : 0.f;
            }
            const int32_t tm = d->tri_tm[t];
            const uint32_t mode = tm == -99999 ? 0u : (tm >= 0 ? 1u + (uint32_t)tm : 1u + d->n_tex + (uint32_t)(-(tm + 1)));
            const uint32_t code = (uint32_t)d->tri_mesh[t] << tex_bits | mode;
            float4 *o = &sh[(size_t)t * 4];
            o[0] = make_float4(n[0], n[1], n[2], n[3]);
            o[1] = make_float4(n[4], n[5], n[6], n[7]);
            o[2] = make_float4(n[8], uv[0], uv[1], uv[2]);
            o[3] = make_float4(uv[3], uv[4], uv[5], bits_to_float((int32_t)code));
        }
        B.add(sh.data(), sh.size(), &DevScene::triShadeF);
        std::vector<float4> alb(d->n_mesh);
        for (uint32_t m = 0; m < d->n_mesh; m++)
            alb[m] = make_float4((float)d->mesh_albedo[3 * (size_t)m], (float)d->mesh_albedo[3 * (size_t)m + 1], (float)d->mesh_albedo[3 * (size_t)m + 2], 0.f);
        B.add(alb.data(), alb.size(), &DevScene::meshAlbF);
        if (texels.size() / 3 >= (1ull << 31)) return fail(PT_ERR_LIMIT, "more than 2^31 texels");
        std::vector<int4> tf(d->n_tex);
        std::vector<uint32_t> t4(texels.size() / 3);
        for (uint32_t k = 0; k < d->n_tex; k++) tf[k] = make_int4(trows[k], tcols[k], (int)(toff[k] / 3), 0);
        for (size_t k = 0; k < t4.size(); k++) t4[k] = (uint32_t)texels[3 * k] | (uint32_t)texels[3 * k + 1] << 8 | (uint32_t)texels[3 * k + 2] << 16;
        B.add(tf.data(), tf.size(), &DevScene::texF);
        B.add(t4.data(), t4.size(), &DevScene::texels4);
    }
    B.add(d->uv, (size_t)d->n_uv * 2, &DevScene::uv);
    B.add(d->mesh_albedo, (size_t)d->n_mesh * 3, &DevScene::albedo);
    B.add(texels.data(), texels.size(), &DevScene::texels);
    B.add(toff.data(), toff.size(), &DevScene::tex_off);
    B.add(trows.data(), trows.size(), &DevScene::tex_rows);
    B.add(tcols.data(), tcols.size(), &DevScene::tex_cols);
    B.add(d->lights, (size_t)d->n_light, &DevScene::lights);
    {
        std::vector<float4> lf(2 * (size_t)d->n_light);
        for (uint32_t k = 0; k < d->n_light; k++) {
            const PtLight &L = d->lights[k];
            lf[2 * k] = make_float4((float)L.pos[0], (float)L.pos[1], (float)L.pos[2], bits_to_float(L.type));
            lf[2 * k + 1] = make_float4((float)(L.mag * L.color[0]), (float)(L.mag * L.color[1]), (float)(L.mag * L.color[2]), 0.f);
        }
        B.add(lf.data(), lf.size(), &DevScene::lightsF);
    }
    S.any_mirror = 0;
    for (uint32_t t = 0; t < d->n_tri; t++) S.any_mirror |= d->tri_reft[t] != 0;
    S.n_light = (int32_t)d->n_light; S.n_node = (int32_t)d->n_node; S.n_ref = (int32_t)d->n_ref; S.n_tri = (int32_t)d->n_tri;
    for (int k = 0; k < 3; k++) {
        S.bg[k] = d->background[k];
        S.root_lo[k] = d->nodes[0].lo[k];
        S.root_hi[k] = d->nodes[0].hi[k];
    }
    // sizes of the scratch every copy of the scene allocates for itself
    // a binary descent leaves at most one entry per level on the stack, a quad record up to three per two levels
    S.ovf_depth = (uint32_t)H.refine.max_depth * 3u / 2u + 4u;
    // PT_TUNE_OVF_DEPTH: a smaller stack than the tree needs, for the test of the limit report
    if (const char *e = getenv("PT_TUNE_OVF_DEPTH")) { int k = atoi(e); if (k >= 1 && (uint32_t)k < S.ovf_depth) S.ovf_depth = (uint32_t)k; }
    S.ovf_waves = PT_TRACE_WAVES;
    S.ovf64_entries = (uint32_t)depth + 2u;
    std::memcpy(H.magic, "PTSLAB04", 8);
    H.total_bytes = B.bytes.size();
    {
        size_t k = 0;
#define X(f) { const size_t off = (size_t)(uintptr_t)S.f; for (const auto &pl : B.placed) if (pl.first == off) H.array_bytes[k] = pl.second; k++; }
        PT_SLAB_POINTERS(X)
#undef X
        static_assert(sizeof(uint64_t) * PT_SLAB_MAX_ARRAYS == sizeof H.array_bytes, "array table");
    }
    H.body_check = slab_body_check_host(B.bytes.data(), B.bytes.size());
    H.check = header_check(H);
    std::memcpy(B.bytes.data(), &H, sizeof H);
    return PT_OK;
}

// Is this the header of a slab that this version of the library wrote, intact, and do the arrays it describes lie inside
// the `bytes` of the block and have the sizes the kernels will assume from the scene's counts?  (What is IN the arrays --
// node links, triangle offsets, indices -- is covered by the body checksum, compared once the block is on the device.)
int check_slab_header(const SlabHeader &H, size_t bytes) {
    if (std::memcmp(H.magic, "PTSLAB04", 8) != 0 || H.total_bytes != bytes || bytes < kSlabBody || (bytes & 7) || H.check != header_check(H))
        return fail(PT_ERR_INVALID, "not a scene slab of this library version, or damaged");
    const DevScene &S = H.S;
    if (S.n_node < 1 || S.n_ref < 0 || S.n_tri < 0 || S.n_light < 0 || (uint64_t)S.n_tri != H.n_tri)
        return fail(PT_ERR_INVALID, "scene slab: counts out of range");
    size_t k = 0;
#define X(f) { const uint64_t off = (uint64_t)(uintptr_t)S.f, nb = H.array_bytes[k++]; \
               if (off < kSlabBody || off > bytes || nb > bytes - off) return fail(PT_ERR_INVALID, "scene slab: array " #f " outside the block"); }
    PT_SLAB_POINTERS(X)
#undef X
    // sizes the kernels rely on, from the counts they index with
    struct Want { uint64_t have, want; const char *what; };
    k = 0;
    uint64_t have[PT_SLAB_MAX_ARRAYS] = {};
#define X(f) have[k] = H.array_bytes[k]; k++;
    PT_SLAB_POINTERS(X)
#undef X
    k = 0;
#define X(f) const size_t i_##f = k++;
    PT_SLAB_POINTERS(X)
#undef X
    const uint64_t nn = (uint64_t)S.n_node, nr = (uint64_t)S.n_ref, nt = (uint64_t)S.n_tri, nl = (uint64_t)S.n_light;
    const Want wants[] = {
        {have[i_nodeA], nn * sizeof(NodeA), "nodeA"}, {have[i_nodeBox], nn * 48, "nodeBox"}, {have[i_refTri], nr * 72, "refTri"},
        {have[i_refId], nr * 4, "refId"}, {have[i_nodeF], H.refine.nodes * sizeof(NodeF), "nodeF"}, {have[i_refTriF], H.refine.refs * 48, "refTriF"},
        {have[i_triGeo], nt * 72, "triGeo"}, {have[i_triNgF], nt * 16, "triNgF"}, {have[i_tri_ni], nt * 12, "tri_ni"}, {have[i_tri_ti], nt * 12, "tri_ti"},
        {have[i_tri_tm], nt * 4, "tri_tm"}, {have[i_tri_reft], nt * 4, "tri_reft"}, {have[i_tri_mesh], nt * 4, "tri_mesh"},
        {have[i_tri_local], nt * 4, "tri_local"}, {have[i_triShadeF], nt * 64, "triShadeF"}, {have[i_meshAlbF], (uint64_t)H.n_mesh * 16, "meshAlbF"}, {have[i_albedo], (uint64_t)H.n_mesh * 24, "albedo"},
        {have[i_lights], nl * sizeof(PtLight), "lights"}, {have[i_lightsF], nl * 32, "lightsF"},
    };
    for (const Want &w : wants)
        if (w.have != w.want) return fail(PT_ERR_INVALID, std::string("scene slab: array ") + w.what + " does not have the size its count asks for");
    if (have[i_tex_off] / 8 != have[i_tex_rows] / 4 || have[i_tex_rows] != have[i_tex_cols] || (S.has_quads && have[i_quadF] < 16) ||
        S.n_tex < 0 || have[i_tex_rows] != (uint64_t)S.n_tex * 4 || have[i_texF] != (uint64_t)S.n_tex * 16 || have[i_texels4] / 4 != have[i_texels] / 3 ||
        S.tex_bits < 1 || S.tex_bits > 24 || (1ull << S.tex_bits) < 2ull * (uint64_t)S.n_tex + 1ull || H.n_mesh >= (1ull << (31 - S.tex_bits)))
        return fail(PT_ERR_INVALID, "scene slab: texture or quad tables inconsistent");
    // scratch every copy allocates from these: keep them in the range build_slab can produce
    if (S.ovf_waves != PT_TRACE_WAVES || S.ovf_depth < 1 || S.ovf_depth > 4096 || S.ovf64_entries < 2 || S.ovf64_entries > PT_STACK_MAX + 2 ||
        H.max_depth < 0 || H.max_depth >= PT_STACK_MAX)
        return fail(PT_ERR_INVALID, "scene slab: scratch sizes out of range");
    return PT_OK;
}

// A scene on the current device around a slab that is already there (`slab`, owned by the scene from here on unless
// `borrowed`: then it stays with the scene that lends it): binds the pointers, allocates the scene's own scratch,
// streams and events, fetches the two host-side index tables.
int finish_scene(uint8_t *slab, const SlabHeader &H, int device, PtScene **out, bool borrowed = false) {
    struct Free { void operator()(PtScene *p) const { pt_scene_free(p); } };
    std::unique_ptr<PtScene, Free> holder(new (std::nothrow) PtScene());
    PtScene *sc = holder.get();
    if (!sc) { if (!borrowed) (void)hipFree(slab); return fail(PT_ERR_OOM, "out of host memory"); }
    sc->slab_borrowed = borrowed;
    sc->slab = slab; sc->slab_bytes = (size_t)H.total_bytes; sc->device = device;
    sc->dev_bytes = H.total_bytes;
    sc->max_depth = H.max_depth; sc->refine = H.refine; sc->n_tri = H.n_tri; sc->n_mesh = H.n_mesh;
    DevScene &S = sc->S;
    S = H.S;
#define X(f) S.f = (decltype(S.f))(slab + (uintptr_t)H.S.f);
    PT_SLAB_POINTERS(X)
#undef X
    auto scratch = [&](size_t bytes, void **p) -> int {
        if (hipMalloc(p, bytes) != hipSuccess) return fail(PT_ERR_OOM, "out of device memory (scene scratch)");
        sc->owned.push_back(*p);
        sc->dev_bytes += bytes;
        return PT_OK;
    };
    int rc;
    // overflow stack of the fp32 traversal: one column of (tree depth + 2) entries per lane of every wave of a trace launch
    if ((rc = scratch((size_t)S.ovf_waves * S.ovf_depth * 64 * sizeof(uint2), (void **)&S.ovf))) return rc;
    // overflow of the fp64 search's stack: beyond its LDS ring a lane needs at most one entry per tree level
    if ((rc = scratch((size_t)PT_EXACT_WAVES * S.ovf64_entries * 64 * 3 * sizeof(double), (void **)&S.ovf64))) return rc;
    if ((rc = scratch(sizeof(unsigned long long), (void **)&S.err))) return rc;
    HIPCHK(hipMemset(S.err, 0, sizeof(unsigned long long)));
    sc->tri_mesh.resize(H.n_tri); sc->tri_local.resize(H.n_tri);
    if (H.n_tri) {
        HIPCHK(hipMemcpy(sc->tri_mesh.data(), S.tri_mesh, (size_t)H.n_tri * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(sc->tri_local.data(), S.tri_local, (size_t)H.n_tri * 4, hipMemcpyDeviceToHost));
    }
    hipError_t e = hipStreamCreate(&sc->stream);
    if (e == hipSuccess) e = hipStreamCreate(&sc->stream_gen);
    if (e == hipSuccess) e = hipEventCreate(&sc->ev0);
    if (e == hipSuccess) e = hipEventCreate(&sc->ev1);
    if (e == hipSuccess) e = hipEventCreate(&sc->ev_p0);
    if (e == hipSuccess) e = hipEventCreate(&sc->ev_p1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_b, hipEventDisableTiming);
    if (e != hipSuccess) return fail(PT_ERR_HIP, std::string("stream/event creation: ") + hipGetErrorString(e));
    *out = holder.release();
    return PT_OK;
}

// A scene on device g_device from a slab image in host or device memory.
int scene_from_slab(const void *src, size_t bytes, bool src_is_device, PtScene **out) {
    if (g_device < 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not been called");
    if (!out || !src) return fail(PT_ERR_INVALID, "null pointer");
    if (bytes < sizeof(SlabHeader)) return fail(PT_ERR_INVALID, "not a scene slab (too small)");
    SlabHeader H;
    if (src_is_device) HIPCHK(hipMemcpy(&H, src, sizeof H, hipMemcpyDeviceToHost));
    else std::memcpy(&H, src, sizeof H);
    int rc = check_slab_header(H, bytes);
    if (rc) return rc;
    HIPCHK(hipSetDevice(g_device));
    void *slab = nullptr;
    if (hipMalloc(&slab, bytes) != hipSuccess) return fail(PT_ERR_OOM, "out of device memory (scene slab)");
    hipError_t e = hipMemcpy(slab, src, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();      // the caller may free or reuse `src` as soon as this returns
    if (e != hipSuccess) { (void)hipFree(slab); return fail(PT_ERR_HIP, std::string("scene slab copy: ") + hipGetErrorString(e)); }
    // the body as it arrived: a block that another process or rank built travels unchecked otherwise, and the kernels
    // follow its node links and triangle offsets without looking
    uint64_t body = 0;
    rc = slab_body_check_device((const uint8_t *)slab, bytes, &body);
    if (!rc && body != H.body_check) rc = fail(PT_ERR_INVALID, "scene slab: the arrays do not match the header's checksum (damaged or truncated in transit)");
    if (rc) { (void)hipFree(slab); return rc; }
    return finish_scene((uint8_t *)slab, H, g_device, out);
}

int upload(const PtSceneDesc *d, PtScene **out) {
    if (g_device < 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not been called");
    if (!out) return fail(PT_ERR_INVALID, "out pointer is null");
    SlabBuilder B;
    int rc = build_slab(d, B);
    if (rc) return rc;
    return scene_from_slab(B.bytes.data(), B.bytes.size(), false, out);
}

// The limit word of the scene after the kernels of a call have finished (the caller has synchronised the scene's
// stream): zero, or the call fails.  The word is cleared for the next call.
std::string limit_message(const char *what, unsigned long long e) {
    std::string msg = std::string(what) + ": a kernel limit was exceeded, the result is not valid (";
    if (e & PT_LIMIT_F32_STACK) msg += "fp32 traversal overflow stack full; ";
    if (e & PT_LIMIT_F32_STEPS) msg += "fp32 traversal ended a ray at the bound on node steps; ";
    if (e & PT_LIMIT_F64_STACK) msg += "fp64 search overflow stack full; ";
    return msg + "flags " + std::to_string(e) + ")";
}
int check_limits(PtScene *sc, const char *what) {
    unsigned long long e = 0;
    HIPCHK(hipMemcpyAsync(&e, sc->S.err, sizeof e, hipMemcpyDeviceToHost, sc->stream));
    HIPCHK(hipStreamSynchronize(sc->stream));
    if (!e) return PT_OK;
    HIPCHK(hipMemsetAsync(sc->S.err, 0, sizeof e, sc->stream));
    return fail(PT_ERR_LIMIT, limit_message(what, e));
}

int ensure_workspace(PtScene *sc, int size, int batch, bool fast, int nsub, bool shadow) {
    Workspace &ws = sc->ws;
    bool have_fast = ws.Q[0].rays_o != nullptr;
    shadow = shadow && fast;
    if (ws.size == size && ws.batch >= batch && (!fast || have_fast) && ws.nsub >= nsub && (!shadow || ws.shadow)) return PT_OK;
    if (ws.size == size) {
        if (batch < ws.batch) batch = ws.batch;
        if (nsub < ws.nsub) nsub = ws.nsub;
        fast = fast || have_fast;
        shadow = shadow || ws.shadow;
    }
    ws_release(ws);
    size_t P = (size_t)size * size;
    FrameBuffers &F = ws.F;
#define WS(expr) do { int rc_ = (expr); if (rc_) { ws_release(ws); return rc_; } } while (0)
    WS(ws_alloc(ws, P * 3, &F.mpnt));
    WS(ws_alloc(ws, P * 3, &F.gclr));
    WS(ws_alloc(ws, P * 3, &F.gnrm));
    WS(ws_alloc(ws, P * 3, &F.ghit));
    WS(ws_alloc(ws, P, &F.rec_found));
    WS(ws_alloc(ws, P, &F.rec_tri));
    WS(ws_alloc(ws, P, &F.rec_t));
    WS(ws_alloc(ws, P, &F.rec_u));
    WS(ws_alloc(ws, P, &F.rec_v));
    WS(ws_alloc(ws, P * 3, &F.gi_acc));
    WS(ws_alloc(ws, P * 3 * (size_t)batch, &F.contrib));
    WS(ws_alloc(ws, P * 3, &F.rgb));
    WS(ws_alloc(ws, P * 3, &F.bgr));
    WS(ws_alloc(ws, (size_t)RC_WORDS, &F.counters));
    WS(ws_alloc(ws, P * nsub * 3, &ws.B.paint));
    WS(ws_alloc(ws, P * nsub * 9, &ws.B.g));
    WS(ws_alloc(ws, P * nsub, &ws.B.found));
    WS(ws_alloc(ws, P * nsub, &ws.B.tri));
    WS(ws_alloc(ws, P * nsub * 3, &ws.B.tuv));
    WS(ws_alloc(ws, P * nsub, &ws.B.again));
    WS(ws_alloc(ws, P * nsub, &ws.B.again_list));
    WS(ws_alloc(ws, P * nsub, &ws.B.mirror));
    WS(ws_alloc(ws, P * nsub * 3, &ws.B.mdir));
    WS(ws_alloc(ws, (size_t)4 * (size_t)(sc->S.n_ref > 0 ? sc->S.n_ref : 0), &ws.B.forms));
    if (fast) WS(fast_alloc(ws.Q, P, batch, shadow ? (sc->S.n_light > 1 ? sc->S.n_light : 1) : 0, ws_alloc_cb, &ws));
#undef WS
    ws.size = size; ws.batch = batch; ws.nsub = nsub; ws.shadow = shadow;
    return PT_OK;
}

int check_options(const PtOptions *o, int *size, int *nsamples) {
    if (!o) return fail(PT_ERR_INVALID, "options are null");
    if (o->struct_size != sizeof(PtOptions)) return fail(PT_ERR_INVALID, "PtOptions.struct_size mismatch");
    if (o->model != PT_MODEL_MAIN && o->model != PT_MODEL_OMP) return fail(PT_ERR_INVALID, "unknown shading model");
    if (o->gi_precision != PT_GI_F64_EXACT && o->gi_precision != PT_GI_F32_FAST) return fail(PT_ERR_INVALID, "unknown gi_precision");
    if (o->spp < 0) return fail(PT_ERR_INVALID, "spp is negative");
    if (o->n_parts > 1 && o->part >= o->n_parts) return fail(PT_ERR_INVALID, "part must be below n_parts");
    int s = o->size == 0 ? 512 : o->size;
    if (s < 1 || s > 8192) return fail(PT_ERR_LIMIT, "image size must be in 1..8192");
    *size = s;
    *nsamples = o->spp > 0 ? o->spp : (o->model == PT_MODEL_MAIN ? 4 : 16);   // main.cpp:129, mainOMP.cpp:132
    return PT_OK;
}

// One view on the scene's stream; leaves the frame in ws.F (bgr, rgb, AOVs).
int render_on_device(PtScene *sc, const PtCamera *cam, const PtOptions *o, PtStats *st, float *gi_ms, uint64_t *gi_launches) {
    int size, N;
    int rc = check_options(o, &size, &N);
    if (rc) return rc;
    if (!cam) return fail(PT_ERR_INVALID, "camera is null");
    HIPCHK(hipSetDevice(sc->device));          // the current device belongs to the calling thread, not to the process
    const bool fast = o->gi && o->gi_precision == PT_GI_F32_FAST;
    size_t P = (size_t)size * size;
    // samples per GI pass: bounded so that per-pass buffers stay at a few hundred MB
    int batch = 1;
    if (o->gi) {
        size_t cap = fast ? (size_t)PT_FAST_RAYS_PER_PASS : ((size_t)16 << 18);
        if (fast)
            if (const char *e = getenv("PT_TUNE_PASS_RAYS")) {
                const long long k = atoll(e);
                if (k >= 1 && k <= (long long)(128u << 20)) cap = (size_t)k;
            }
        // main.cpp model: a pass also queues a shadow ray per shading point and light; passes shrink past 16 lights
        // so that that queue stays below 2^30 entries
        if (fast && o->model == PT_MODEL_MAIN && sc->S.n_light > 16) {
            const size_t most = ((size_t)1 << 30) / (size_t)sc->S.n_light;
            if (cap > most) cap = most;
        }
        batch = (int)(cap / P);
        if (batch < 1) batch = 1;
        if (batch > N) batch = N;
    }
    rc = ensure_workspace(sc, size, batch, fast, o->rb ? 9 : 1, o->model == PT_MODEL_MAIN);
    if (rc) return rc;
    FrameBuffers &F = sc->ws.F;
    ViewParams V{};
    for (int k = 0; k < 3; k++) V.eye[k] = cam->eye[k];
    V.zoom = cam->zoom;
    V.sa = std::sin(cam->angle);
    V.ca = std::cos(cam->angle);
    V.size = size; V.rb = o->rb ? 1 : 0;
    V.off = o->model == PT_MODEL_MAIN ? 0.01 : 0.1;
    V.seed = o->seed; V.view = o->view; V.model = o->model;
    V.part = o->n_parts > 1 ? o->part : 0; V.n_parts = o->n_parts > 1 ? o->n_parts : 1;
    hipStream_t s = sc->stream;
    HIPCHK(hipMemsetAsync(F.counters, 0, RC_WORDS * sizeof(unsigned long long), s));
    // the kernels of a view report tripped limits in the last word of the view's counter block, read with the counters
    DevScene S = sc->S;
    S.err = F.counters + RC_LIMITS;
    if (st) HIPCHK(hipEventRecord(sc->ev_p0, s));
    if (exact_primary(S, V, F, sc->ws.B, s, sc->stream_gen, sc->ev_a, sc->ev_b)) return fail(PT_ERR_HIP, "primary pass launch failed");
    if (st) HIPCHK(hipEventRecord(sc->ev_p1, s));
    *gi_ms = 0; *gi_launches = 0;
    uint64_t fast_rays = 0, fast_shadow = 0, fast_work[2] = {0, 0};
    const bool count_work = fast && (o->flags & PT_FLAG_COUNT_WORK);
    if (o->gi) {
        if (fast) {
            rc = fast_gi(S, V, F, sc->ws.Q, N, batch, s, sc->stream_gen, sc->pass_events, gi_ms, gi_launches, &fast_rays, &fast_shadow,
                         count_work ? fast_work : nullptr);
            if (rc) return fail(rc, std::string("GI wavefront pass failed: ") + hipGetErrorString(hipGetLastError()));
        } else {
            for (int s0 = 0; s0 < N; s0 += batch) {
                int nb = N - s0 < batch ? N - s0 : batch;
                HIPCHK(hipEventRecord(sc->ev0, s));
                if (exact_gi(S, V, F, s0, nb, s)) return fail(PT_ERR_HIP, "GI pass launch failed");
                HIPCHK(hipEventRecord(sc->ev1, s));
                if (gi_accumulate(F, size, nb, s0 == 0, s)) return fail(PT_ERR_HIP, "GI accumulation launch failed");
                HIPCHK(hipEventSynchronize(sc->ev1));
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, sc->ev0, sc->ev1));
                *gi_ms += ms; (*gi_launches)++;
            }
        }
    }
    if (resolve_frame(F, size, o->gi ? 1 : 0, N, s)) return fail(PT_ERR_HIP, "resolve launch failed");
    unsigned long long c[RC_WORDS];
    HIPCHK(hipMemcpyAsync(c, F.counters, sizeof c, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (unsigned h = 0; h < RC_HEAD_REGIONS; h++)                // the kernels count rays per group of blocks
        for (unsigned k = 0; k < RC_WORK_PRIMARY; k++) c[k] += c[RC_HEADS + h * RC_HEAD_STRIDE + 1 + k];
    if (c[RC_LIMITS]) return fail(PT_ERR_LIMIT, limit_message("render", c[RC_LIMITS]));
    if (st) {
        float pms = 0;
        HIPCHK(hipEventElapsedTime(&pms, sc->ev_p0, sc->ev_p1));
        st->ms_primary += pms;
        st->primary += c[RC_PRIMARY]; st->shadow += c[RC_SHADOW]; st->mirror += c[RC_MIRROR];
        st->gi += c[RC_GI] + fast_rays; st->shadow += fast_shadow; st->omp_retrace += c[RC_RETRACE];
        st->ms_gi_trace += *gi_ms; st->gi_trace_launches += *gi_launches;
        st->gi_nodes_visited += fast_work[0]; st->gi_tris_tested += fast_work[1];
    }
    return PT_OK;
}

// ---- copies of a scene on the other initialised devices -------------------------------------------------------
// RCCL is loaded when it is first needed (a process that runs on one GPU, or under torch.distributed with its own
// RCCL, never touches it).
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};
Rccl g_rccl;

int rccl_comms(const std::vector<int> &devs) {
    Rccl &R = g_rccl;
    if (!R.lib) {
        R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) return fail(PT_ERR_NO_DEVICE, std::string("RCCL could not be loaded: ") + dlerror());
#define SYM(field, name) R.field = (decltype(R.field))dlsym(R.lib, name); if (!R.field) return fail(PT_ERR_NO_DEVICE, "RCCL lacks " name)
        SYM(CommInitAll, "ncclCommInitAll"); SYM(Broadcast, "ncclBroadcast"); SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    }
    if (R.devices == devs) return PT_OK;
    for (ncclComm_t c : R.comms) (void)R.CommDestroy(c);
    R.comms.assign(devs.size(), nullptr);
    R.devices.clear();
    ncclResult_t r = R.CommInitAll(R.comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) { R.comms.clear(); return fail(PT_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r)); }
    R.devices = devs;
    return PT_OK;
}

// The devices the view loop runs on: those of pt_init, or -- PT_TUNE_REPLICAS=k, a rehearsal for machines with one
// GPU -- k workers that all use the first device.
std::vector<int> worker_devices() {
    std::vector<int> devs = g_devices;
    if (const char *e = getenv("PT_TUNE_REPLICAS")) {
        const int k = atoi(e);
        if (k >= 1 && k <= 16) devs.assign((size_t)k, g_device);
    }
    return devs;
}

// Makes sure `sc` has a copy on every worker device but the first.  The slab -- every immutable array of the scene,
// the fp32 tree included -- goes from the first device to the others in ONE broadcast over xGMI (RCCL; a single
// collective of the scene's size, no host round trip, nothing rebuilt).  Workers that share the first device (the
// rehearsal) get a device-to-device copy, or, with PT_TUNE_RCCL_SELF=1, a broadcast in a communicator of one rank, so
// that the RCCL calls are exercised on a one-GPU machine too.
int ensure_replicas(PtScene *sc, const std::vector<int> &devs) {
    if (sc->replicas.size() + 1 == devs.size()) {
        bool same = true;
        for (size_t k = 1; k < devs.size(); k++) same = same && sc->replicas[k - 1]->device == devs[k];
        if (same) return PT_OK;
    }
    for (PtScene *r : sc->replicas) pt_scene_free(r);
    sc->replicas.clear();
    if (devs.size() <= 1) return PT_OK;
    SlabHeader H;
    HIPCHK(hipSetDevice(sc->device));
    HIPCHK(hipMemcpy(&H, sc->slab, sizeof H, hipMemcpyDeviceToHost));
    bool distinct = true;
    for (size_t a = 0; a < devs.size(); a++)
        for (size_t b = a + 1; b < devs.size(); b++) distinct = distinct && devs[a] != devs[b];
    const bool self = !distinct && getenv("PT_TUNE_RCCL_SELF") && atoi(getenv("PT_TUNE_RCCL_SELF")) == 1;
    std::vector<uint8_t *> slabs(devs.size(), nullptr);
    slabs[0] = sc->slab;
    auto release = [&]() { for (size_t k = 1; k < slabs.size(); k++) if (slabs[k]) { (void)hipSetDevice(devs[k]); (void)hipFree(slabs[k]); } (void)hipSetDevice(sc->device); };
    for (size_t k = 1; k < devs.size(); k++) {
        if (hipSetDevice(devs[k]) != hipSuccess || hipMalloc((void **)&slabs[k], sc->slab_bytes) != hipSuccess) {
            release();
            return fail(PT_ERR_OOM, "out of device memory (scene copy on device " + std::to_string(devs[k]) + ")");
        }
    }
    int rc = PT_OK;
    if (distinct) {
        rc = rccl_comms(devs);
        if (!rc) {
            Rccl &R = g_rccl;
            ncclResult_t r = R.GroupStart();
            for (size_t k = 0; k < devs.size() && r == ncclSuccess; k++) {
                (void)hipSetDevice(devs[k]);
                r = R.Broadcast(sc->slab, slabs[k], sc->slab_bytes, ncclUint8, 0, R.comms[k], nullptr);
            }
            ncclResult_t r2 = R.GroupEnd();
            if (r == ncclSuccess) r = r2;
            for (size_t k = 0; k < devs.size(); k++) { (void)hipSetDevice(devs[k]); (void)hipDeviceSynchronize(); }
            if (r != ncclSuccess) rc = fail(PT_ERR_HIP, std::string("ncclBroadcast of the scene: ") + R.GetErrorString(r));
        }
    } else if (self) {
        rc = rccl_comms(std::vector<int>{sc->device});
        for (size_t k = 1; k < devs.size() && !rc; k++) {
            ncclResult_t r = g_rccl.Broadcast(sc->slab, slabs[k], sc->slab_bytes, ncclUint8, 0, g_rccl.comms[0], nullptr);
            if (r != ncclSuccess) rc = fail(PT_ERR_HIP, std::string("ncclBroadcast of the scene: ") + g_rccl.GetErrorString(r));
        }
        (void)hipDeviceSynchronize();
    } else {
        for (size_t k = 1; k < devs.size() && !rc; k++)
            if (hipMemcpy(slabs[k], sc->slab, sc->slab_bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = fail(PT_ERR_HIP, "scene copy failed");
    }
    if (rc) { release(); return rc; }
    for (size_t k = 1; k < devs.size(); k++) {
        (void)hipSetDevice(devs[k]);
        PtScene *r = nullptr;
        uint8_t *slab = slabs[k];
        slabs[k] = nullptr;                                  // owned by the copy (or freed by finish_scene) from here on
        rc = finish_scene(slab, H, devs[k], &r);
        if (rc) { release(); for (PtScene *q : sc->replicas) pt_scene_free(q); sc->replicas.clear(); return rc; }
        sc->replicas.push_back(r);
    }
    HIPCHK(hipSetDevice(sc->device));
    return PT_OK;
}

// Makes sure `base` has at least k - 1 sharers: view workers on its device that use its slab (the scene is immutable)
// with scratch, streams and workspace of their own.
int ensure_sharers(PtScene *base, size_t k) {
    if (base->sharers.size() + 1 >= k) return PT_OK;
    HIPCHK(hipSetDevice(base->device));
    SlabHeader H;
    HIPCHK(hipMemcpy(&H, base->slab, sizeof H, hipMemcpyDeviceToHost));
    while (base->sharers.size() + 1 < k) {
        PtScene *r = nullptr;
        int rc = finish_scene(base->slab, H, base->device, &r, true);
        if (rc) return rc;
        base->sharers.push_back(r);
    }
    return PT_OK;
}

// View workers per device copy of the scene.  One view of a few million rays does not fill the GPU from end to end:
// its kernels have ramps and tails, the fp64 primary pass and the fp32 traversal are bound by different units, and
// the host submits between them; two to four views in flight on streams of their own fill those gaps (car, 16 spp,
// -rb: 585 -> 691 views/s with four).  A view whose GI pass alone is tens of millions of rays gains nothing and
// its workspace is gigabytes, so it keeps one worker.  PT_TUNE_WORKERS overrides the choice.
size_t workers_per_device(size_t views_per_device, int size, int gi, int spp) {
    if (const char *e = getenv("PT_TUNE_WORKERS")) {
        const int k = atoi(e);
        if (k >= 1 && k <= 8) return (size_t)k;
    }
    const size_t load = (size_t)size * size * (size_t)(gi ? (spp > 1 ? spp : 1) : 1);
    size_t k = ((size_t)32 << 20) / (load ? load : 1);
    k = k < 1 ? 1 : (k > 4 ? 4 : k);
    return k < views_per_device ? k : (views_per_device ? views_per_device : 1);
}

template <typename T>
int fetch(PtScene *sc, T *host, const T *dev, size_t n) {
    if (!host) return PT_OK;
    HIPCHK(hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, sc->stream));
    return PT_OK;
}

}  // namespace

extern "C" {

const char *pt_last_error(void) { return g_error.c_str(); }
const char *pt_version(void) { return "pathtracing_amd 0.1 (gfx950)"; }

int pt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device >= n) return fail(PT_ERR_NO_DEVICE, "device ordinal " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
    // device >= 0: that device.  device < 0: every visible device -- scenes are uploaded to the first, and
    // pt_render_views shards its views over all of them.
    std::vector<int> devs;
    for (int dev = device < 0 ? 0 : device; dev < (device < 0 ? n : device + 1); dev++) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(PT_ERR_NO_DEVICE, std::string("device ") + std::to_string(dev) + " is " + prop.gcnArchName +
                                              ", this library is built for gfx950 only");
        HIPCHK(hipSetDevice(dev));
        // address of this device's copy of the fixed scratch array (pt_trace_device)
        if (int rc = static_scratch_setup(dev))
            return fail(rc, "the address of the library's scratch array could not be taken on device " + std::to_string(dev));
        devs.push_back(dev);
    }
    HIPCHK(hipSetDevice(devs[0]));
    g_devices = devs;
    g_device = devs[0];
    return PT_OK;
}

int pt_shutdown(void) { g_device = -1; g_devices.clear(); return PT_OK; }

int pt_device_count(void) { return (int)g_devices.size(); }

int pt_scene_upload(const PtSceneDesc *desc, PtScene **out) { return guarded([&] { return upload(desc, out); }); }

int pt_scene_info(const PtScene *sc, PtSceneInfo *out) {
    if (!sc || !out) return fail(PT_ERR_INVALID, "pt_scene_info: null argument");
    std::memset(out, 0, sizeof *out);
    out->host_nodes = (uint64_t)sc->S.n_node; out->host_refs = (uint64_t)sc->S.n_ref;
    out->dev_nodes = sc->refine.nodes; out->dev_refs = sc->refine.refs; out->dev_leaves = sc->refine.leaves;
    out->dev_max_leaf = sc->refine.max_leaf; out->dev_interior_refs = sc->refine.interior_refs;
    out->dev_bytes = sc->dev_bytes;
    out->host_depth = sc->max_depth; out->dev_depth = sc->refine.max_depth;
    return PT_OK;
}

int pt_scene_free(PtScene *sc) {
    if (!sc) return PT_OK;
    for (PtScene *r : sc->replicas) pt_scene_free(r);
    for (PtScene *r : sc->sharers) pt_scene_free(r);
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(sc->device);
    ws_release(sc->ws);
    for (void *p : sc->owned) (void)hipFree(p);
    if (sc->slab && !sc->slab_borrowed) (void)hipFree(sc->slab);
    if (sc->ev0) (void)hipEventDestroy(sc->ev0);
    if (sc->ev1) (void)hipEventDestroy(sc->ev1);
    if (sc->ev_p0) (void)hipEventDestroy(sc->ev_p0);
    if (sc->ev_p1) (void)hipEventDestroy(sc->ev_p1);
    if (sc->ev_a) (void)hipEventDestroy(sc->ev_a);
    if (sc->ev_b) (void)hipEventDestroy(sc->ev_b);
    for (hipEvent_t e : sc->pass_events) (void)hipEventDestroy(e);
    if (sc->stream_gen) (void)hipStreamDestroy(sc->stream_gen);
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete sc;
    return PT_OK;
}

// ---- blob form: header, then the arrays of PtSceneDesc back to back (8-byte aligned) ----
struct BlobHeader {
    char magic[8];            // "PTBLOB02"
    uint32_t counts[9];       // n_mesh .. n_ref in PtSceneDesc order
    uint32_t reserved;
    double background[3];
    uint64_t total_bytes;
    uint64_t checksum;        // blob_checksum() of the whole blob with this field zero
};

static size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

// A blob travels between processes and machines (broadcast, files): any damage on the way must be refused, also where
// it would still pass the structural checks (a flipped bit in a coordinate).  64-bit multiply-rotate hash over the
// blob's 8-byte words; the sections are padded to 8 bytes, so the blob is a whole number of words.
static uint64_t blob_checksum(const uint8_t *p, size_t bytes) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
    for (size_t o = 0; o + 8 <= bytes; o += 8) {
        uint64_t w;
        std::memcpy(&w, p + o, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h = (h << 27) | (h >> 37);
    }
    return h ^ (h >> 33);
}

static size_t blob_bytes(const PtSceneDesc *d) {
    size_t n = sizeof(BlobHeader);
    n += pad8((size_t)d->n_vert * 24) + pad8((size_t)d->n_norm * 24) + pad8((size_t)d->n_uv * 16);
    n += 3 * pad8((size_t)d->n_tri * 12) + 4 * pad8((size_t)d->n_tri * 4);
    n += pad8((size_t)d->n_mesh * 24);
    n += pad8((size_t)d->n_tex * 8);
    for (uint32_t k = 0; k < d->n_tex; k++) n += pad8((size_t)d->tex[k].rows * d->tex[k].cols * 3);
    n += pad8((size_t)d->n_light * sizeof(PtLight));
    n += pad8((size_t)d->n_node * sizeof(PtKdNode)) + pad8((size_t)d->n_ref * 4);
    return n;
}

int pt_scene_blob_size(const PtSceneDesc *desc, size_t *bytes) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (!bytes) return fail(PT_ERR_INVALID, "bytes pointer is null");
    *bytes = blob_bytes(desc);
    return PT_OK;
}

int pt_scene_blob_write(const PtSceneDesc *d, void *blob, size_t bytes) { return guarded([&]() -> int {
    int rc = validate_desc(d);
    if (rc) return rc;
    if (!blob || bytes < blob_bytes(d)) return fail(PT_ERR_INVALID, "blob buffer too small");
    uint8_t *p = (uint8_t *)blob;
    std::memset(p, 0, blob_bytes(d));
    BlobHeader h{};
    std::memcpy(h.magic, "PTBLOB02", 8);
    const uint32_t c[9] = {d->n_mesh, d->n_vert, d->n_norm, d->n_uv, d->n_tri, d->n_tex, d->n_light, d->n_node, d->n_ref};
    std::memcpy(h.counts, c, sizeof c);
    for (int k = 0; k < 3; k++) h.background[k] = d->background[k];
    h.total_bytes = blob_bytes(d);
    std::memcpy(p, &h, sizeof h);
    size_t o = sizeof h;
    auto put = [&](const void *src, size_t n) { if (n) std::memcpy(p + o, src, n); o += pad8(n); };
    put(d->vert, (size_t)d->n_vert * 24); put(d->norm, (size_t)d->n_norm * 24); put(d->uv, (size_t)d->n_uv * 16);
    put(d->tri_vi, (size_t)d->n_tri * 12); put(d->tri_ni, (size_t)d->n_tri * 12); put(d->tri_ti, (size_t)d->n_tri * 12);
    put(d->tri_tm, (size_t)d->n_tri * 4); put(d->tri_reft, (size_t)d->n_tri * 4);
    put(d->tri_mesh, (size_t)d->n_tri * 4); put(d->tri_local, (size_t)d->n_tri * 4);
    put(d->mesh_albedo, (size_t)d->n_mesh * 24);
    std::vector<int32_t> dims(2 * (size_t)d->n_tex);
    for (uint32_t k = 0; k < d->n_tex; k++) { dims[2 * k] = d->tex[k].rows; dims[2 * k + 1] = d->tex[k].cols; }
    put(dims.data(), dims.size() * 4);
    for (uint32_t k = 0; k < d->n_tex; k++) put(d->tex[k].bgr, (size_t)d->tex[k].rows * d->tex[k].cols * 3);
    put(d->lights, (size_t)d->n_light * sizeof(PtLight));
    put(d->nodes, (size_t)d->n_node * sizeof(PtKdNode));
    put(d->refs, (size_t)d->n_ref * 4);
    h.checksum = blob_checksum(p, (size_t)h.total_bytes);      // the header in place still has a zero checksum
    std::memcpy(p, &h, sizeof h);
    return PT_OK;
}); }

int pt_scene_upload_blob(const void *blob, size_t bytes, PtScene **out) { return guarded([&]() -> int {
    if (!blob || bytes < sizeof(BlobHeader)) return fail(PT_ERR_INVALID, "blob too small");
    const uint8_t *p = (const uint8_t *)blob;
    BlobHeader h;
    std::memcpy(&h, p, sizeof h);
    if (std::memcmp(h.magic, "PTBLOB02", 8) != 0 || h.total_bytes > bytes || h.total_bytes < sizeof h || (h.total_bytes & 7))
        return fail(PT_ERR_INVALID, "not a scene blob, or truncated");
    {
        // checksum of the blob with the checksum field zero, computed without copying the blob: header apart
        BlobHeader z = h;
        z.checksum = 0;
        std::vector<uint8_t> head(sizeof z);
        std::memcpy(head.data(), &z, sizeof z);
        // hash = fold over words; fold the zeroed header first, then the rest, exactly as blob_checksum does over the whole
        uint64_t acc = 0x9E3779B97F4A7C15ull ^ (size_t)h.total_bytes;
        auto fold = [&](const uint8_t *q, size_t n) {
            for (size_t o = 0; o + 8 <= n; o += 8) {
                uint64_t w;
                std::memcpy(&w, q + o, 8);
                acc = (acc ^ w) * 0xFF51AFD7ED558CCDull;
                acc = (acc << 27) | (acc >> 37);
            }
        };
        static_assert(sizeof(BlobHeader) % 8 == 0, "the header is a whole number of checksum words");
        fold(head.data(), sizeof z);
        fold(p + sizeof z, (size_t)h.total_bytes - sizeof z);
        if ((acc ^ (acc >> 33)) != h.checksum) return fail(PT_ERR_INVALID, "scene blob is damaged (checksum mismatch)");
    }
    PtSceneDesc d{};
    d.struct_size = sizeof d;
    d.n_mesh = h.counts[0]; d.n_vert = h.counts[1]; d.n_norm = h.counts[2]; d.n_uv = h.counts[3]; d.n_tri = h.counts[4];
    d.n_tex = h.counts[5]; d.n_light = h.counts[6]; d.n_node = h.counts[7]; d.n_ref = h.counts[8];
    for (int k = 0; k < 3; k++) d.background[k] = h.background[k];
    size_t o = sizeof h;
    bool ok = true;
    auto take = [&](size_t n) -> const void * {
        const void *q = p + o;
        if (o + pad8(n) > h.total_bytes) { ok = false; return p; }
        o += pad8(n);
        return q;
    };
    d.vert = (const double *)take((size_t)d.n_vert * 24); d.norm = (const double *)take((size_t)d.n_norm * 24);
    d.uv = (const double *)take((size_t)d.n_uv * 16);
    d.tri_vi = (const int32_t *)take((size_t)d.n_tri * 12); d.tri_ni = (const int32_t *)take((size_t)d.n_tri * 12);
    d.tri_ti = (const int32_t *)take((size_t)d.n_tri * 12);
    d.tri_tm = (const int32_t *)take((size_t)d.n_tri * 4); d.tri_reft = (const int32_t *)take((size_t)d.n_tri * 4);
    d.tri_mesh = (const int32_t *)take((size_t)d.n_tri * 4); d.tri_local = (const int32_t *)take((size_t)d.n_tri * 4);
    d.mesh_albedo = (const double *)take((size_t)d.n_mesh * 24);
    const int32_t *dims = (const int32_t *)take((size_t)d.n_tex * 8);
    std::vector<PtTexture> tex(d.n_tex);
    for (uint32_t k = 0; ok && k < d.n_tex; k++) {
        tex[k].rows = dims[2 * k]; tex[k].cols = dims[2 * k + 1];
        if (tex[k].rows <= 0 || tex[k].cols <= 0) { ok = false; break; }
        tex[k].bgr = (const uint8_t *)take((size_t)tex[k].rows * tex[k].cols * 3);
    }
    d.tex = tex.data();
    d.lights = (const PtLight *)take((size_t)d.n_light * sizeof(PtLight));
    d.nodes = (const PtKdNode *)take((size_t)d.n_node * sizeof(PtKdNode));
    d.refs = (const int32_t *)take((size_t)d.n_ref * 4);
    if (!ok) return fail(PT_ERR_INVALID, "scene blob is truncated");
    return upload(&d, out);
}); }

int pt_render_view(PtScene *sc, const PtCamera *cam, const PtOptions *opt, uint8_t *bgr_out, PtAov *aov, PtStats *stats) { return guarded([&]() -> int {
    if (!sc) return fail(PT_ERR_INVALID, "scene is null");
    std::lock_guard<std::mutex> lock(sc->busy);
    auto t0 = std::chrono::steady_clock::now();
    float gi_ms; uint64_t gi_l;
    int rc = render_on_device(sc, cam, opt, stats, &gi_ms, &gi_l);
    if (rc) return rc;
    size_t P = (size_t)sc->ws.size * sc->ws.size;
    FrameBuffers &F = sc->ws.F;
    std::vector<int32_t> tri;
    rc = fetch(sc, bgr_out, F.bgr, P * 3);
    if (rc) return rc;
    if (aov) {
        if (aov->mesh || aov->tri) { tri.resize(P); rc = fetch(sc, tri.data(), F.rec_tri, P); if (rc) return rc; }
        if ((rc = fetch(sc, aov->t, F.rec_t, P)) || (rc = fetch(sc, aov->u, F.rec_u, P)) || (rc = fetch(sc, aov->v, F.rec_v, P)) ||
            (rc = fetch(sc, aov->rgb, F.rgb, P * 3)) || (rc = fetch(sc, aov->gclr, F.gclr, P * 3)) ||
            (rc = fetch(sc, aov->gnrm, F.gnrm, P * 3)) || (rc = fetch(sc, aov->ghit, F.ghit, P * 3)))
            return rc;
    }
    HIPCHK(hipStreamSynchronize(sc->stream));
    if (aov && (aov->mesh || aov->tri))
        for (size_t p = 0; p < P; p++) {
            int32_t t = tri[p];
            if (aov->mesh) aov->mesh[p] = t < 0 ? -1 : sc->tri_mesh[t];
            if (aov->tri) aov->tri[p] = t < 0 ? -1 : sc->tri_local[t];
        }
    if (stats) stats->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}); }

int pt_render_view_device(PtScene *sc, const PtCamera *cam, const PtOptions *opt, void *d_bgr, PtStats *stats) { return guarded([&]() -> int {
    if (!sc || !d_bgr) return fail(PT_ERR_INVALID, "scene or output pointer is null");
    std::lock_guard<std::mutex> lock(sc->busy);
    auto t0 = std::chrono::steady_clock::now();
    float gi_ms; uint64_t gi_l;
    int rc = render_on_device(sc, cam, opt, stats, &gi_ms, &gi_l);
    if (rc) return rc;
    size_t P = (size_t)sc->ws.size * sc->ws.size;
    HIPCHK(hipMemcpyAsync(d_bgr, sc->ws.F.bgr, P * 3, hipMemcpyDeviceToDevice, sc->stream));
    HIPCHK(hipStreamSynchronize(sc->stream));
    if (stats) stats->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}); }

int pt_render_views(PtScene *sc, const PtCamera *cams, size_t n, uint32_t view0, uint32_t view_stride,
                    const PtOptions *opt, uint8_t *frames, PtStats *stats) { return guarded([&]() -> int {
    if (!sc || (n && (!cams || !frames))) return fail(PT_ERR_INVALID, "scene, cameras or frames pointer is null");
    int size, N;
    int rc = check_options(opt, &size, &N);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(sc->busy);
    auto t0 = std::chrono::steady_clock::now();
    const size_t frame_bytes = (size_t)size * size * 3;
    // One worker per device, each on its own copy of the scene: a worker takes the next view that nobody has taken yet
    // (views differ in cost by tens of percent, so dealing them out in advance -- the reference deals rows round-robin,
    // mainOMP.cpp:169-174 -- would leave GPUs waiting at the end) and copies each frame straight into the caller's array
    // (every GPU over its own PCIe link; nothing is routed through the first GPU).  The frames do not depend on the
    // number of workers or on who renders what: the RNG is keyed by the view index.
    const std::vector<int> devs = worker_devices();
    rc = ensure_replicas(sc, devs);
    if (rc) return rc;
    const size_t per_dev = workers_per_device((n + devs.size() - 1) / devs.size(), size, opt->gi, N);
    std::vector<PtScene *> workers;
    std::vector<int> wdev;
    for (size_t d = 0; d < devs.size(); d++) {
        PtScene *base = d == 0 ? sc : sc->replicas[d - 1];
        if ((rc = ensure_sharers(base, per_dev))) return rc;
        for (size_t k = 0; k < per_dev; k++) { workers.push_back(k == 0 ? base : base->sharers[k - 1]); wdev.push_back(devs[d]); }
    }
    (void)hipSetDevice(sc->device);
    const size_t W = workers.size();
    std::vector<int> wrc(W, PT_OK);
    std::vector<std::string> werr(W);
    std::vector<PtStats> wst(W);
    std::atomic<size_t> next_view{0};
    std::vector<long long> wfailed(W, -1);                 // the view a worker failed on (it takes no further view)
    // PT_TUNE_FAIL_VIEW=i: the first attempt at view i of the call fails (a test of the second attempt)
    long long inject = -1;
    if (const char *e = getenv("PT_TUNE_FAIL_VIEW")) inject = atoll(e);
    std::atomic<bool> injected{false};
    auto one_view = [&](PtScene *my, PtStats *st, size_t i) -> int {
        PtOptions o = *opt;
        o.view = view0 + (uint32_t)i * view_stride;
        if ((long long)i == inject && !injected.exchange(true)) return fail(PT_ERR_HIP, "injected failure (PT_TUNE_FAIL_VIEW)");
        float gi_ms; uint64_t gi_l;
        int r = render_on_device(my, &cams[i], &o, st, &gi_ms, &gi_l);
        // the copy is ordered on the scene's stream; the next view's kernels follow it
        if (!r && hipMemcpyAsync(frames + i * frame_bytes, my->ws.F.bgr, frame_bytes, hipMemcpyDeviceToHost, my->stream) != hipSuccess)
            r = fail(PT_ERR_HIP, "frame copy failed");
        return r;
    };
    auto work = [&](size_t w) {
        PtScene *my = workers[w];
        PtStats *st = stats ? &wst[w] : nullptr;
        if (st) std::memset(st, 0, sizeof *st);
        int r = PT_OK;
        if (hipSetDevice(my->device) != hipSuccess) r = fail(PT_ERR_HIP, "hipSetDevice failed in a view worker");
        for (size_t i; !r && (i = next_view.fetch_add(1)) < n;) {
            r = one_view(my, st, i);
            if (r) wfailed[w] = (long long)i;
        }
        // whatever happened, no copy into the caller's `frames` is in flight on this worker's stream when the call returns
        const hipError_t se = hipStreamSynchronize(my->stream);
        if (!r && se != hipSuccess) r = fail(PT_ERR_HIP, "stream synchronisation failed");
        wrc[w] = r;
        if (r) werr[w] = g_error;                          // the message lives in the worker's thread
    };
    if (W == 1) {
        work(0);
    } else {
        std::vector<std::thread> threads;
        for (size_t w = 0; w < W; w++) threads.emplace_back([&, w] { try { work(w); } catch (...) { wrc[w] = PT_ERR_OOM; werr[w] = "exception in a view worker"; } });
        for (std::thread &t : threads) t.join();
        (void)hipSetDevice(sc->device);
    }
    // A worker that failed stopped at its view; the others went on and rendered the rest (the failed worker's share included).
    // The failed view gets a second attempt on a worker that is in order (the reference has nothing of the kind: a
    // worker that dies leaves rank 0 waiting in MPI_Recv, mainOMP.cpp:188); a second failure, or no worker left, ends the call.
    for (size_t w = 0; w < W; w++) {
        if (!wrc[w]) continue;
        size_t good = W;
        for (size_t k = 0; k < W && good == W; k++) if (!wrc[k]) good = k;
        const std::string first = "view worker " + std::to_string(w) + " (device " + std::to_string(wdev[w]) + "): " + werr[w];
        if (good == W || wfailed[w] < 0) return fail(wrc[w], first);
        // views the failed worker never got to: the counter ran on, the other workers took them all; only its own view is open
        PtScene *my = workers[good];
        int r = hipSetDevice(my->device) == hipSuccess ? PT_OK : fail(PT_ERR_HIP, "hipSetDevice failed");
        if (!r) r = one_view(my, stats ? &wst[good] : nullptr, (size_t)wfailed[w]);
        const hipError_t se = hipStreamSynchronize(my->stream);
        if (!r && se != hipSuccess) r = fail(PT_ERR_HIP, "stream synchronisation failed");
        (void)hipSetDevice(sc->device);
        if (r) return fail(r, first + "; second attempt at view " + std::to_string(wfailed[w]) + " on worker " + std::to_string(good) + ": " + g_error);
        wrc[w] = PT_OK;
    }
    if (stats) {
        for (size_t w = 0; w < W; w++) {
            stats->primary += wst[w].primary; stats->shadow += wst[w].shadow; stats->mirror += wst[w].mirror; stats->gi += wst[w].gi;
            stats->omp_retrace += wst[w].omp_retrace; stats->ms_gi_trace += wst[w].ms_gi_trace;
            stats->gi_trace_launches += wst[w].gi_trace_launches; stats->gi_nodes_visited += wst[w].gi_nodes_visited;
            stats->gi_tris_tested += wst[w].gi_tris_tested; stats->ms_primary += wst[w].ms_primary;
        }
        stats->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PT_OK;
}); }

// The scene's slab -- all of its immutable device arrays in one position-independent block -- copied out to host or
// device memory, and a scene made from such a copy: how one process hands a built scene to another GPU or process
// (e.g. as the payload of an RCCL broadcast) without reading, checking or building anything twice.
int pt_scene_slab_size(const PtScene *sc, size_t *bytes) {
    if (!sc || !bytes) return fail(PT_ERR_INVALID, "pt_scene_slab_size: null argument");
    *bytes = sc->slab_bytes;
    return PT_OK;
}

int pt_scene_slab_export(const PtScene *sc, void *dst, size_t bytes, int dst_is_device) {
    if (!sc || !dst) return fail(PT_ERR_INVALID, "pt_scene_slab_export: null argument");
    if (bytes < sc->slab_bytes) return fail(PT_ERR_INVALID, "pt_scene_slab_export: destination too small");
    HIPCHK(hipSetDevice(sc->device));
    HIPCHK(hipMemcpy(dst, sc->slab, sc->slab_bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    // a device-to-device copy may still be running when hipMemcpy returns; the caller hands `dst` to another stream or
    // library (a collective) next, so the copy has to be complete
    HIPCHK(hipDeviceSynchronize());
    return PT_OK;
}

int pt_scene_upload_slab(const void *src, size_t bytes, int src_is_device, PtScene **out) {
    return guarded([&] { return scene_from_slab(src, bytes, src_is_device != 0, out); });
}

int pt_trace(PtScene *sc, size_t n, const double *src, const double *dir, const double *tmin_in, int mode,
             int32_t *found, int32_t *mesh, int32_t *tri, double *t, double *u, double *v) { return guarded([&]() -> int {
    if (!sc) return fail(PT_ERR_INVALID, "scene is null");
    if (mode < 0 || mode > 2) return fail(PT_ERR_INVALID, "pt_trace: mode must be 0, 1 or 2");
    if (n == 0) return PT_OK;
    if (!src || !dir || !tmin_in || !found || !mesh || !tri || !t || !u || !v) return fail(PT_ERR_INVALID, "pt_trace: null array");
    if (n > ((size_t)1 << 28)) return fail(PT_ERR_LIMIT, "pt_trace: more than 2^28 rays in one call");
    std::lock_guard<std::mutex> lock(sc->busy);
    HIPCHK(hipSetDevice(sc->device));
    double *d_src = nullptr, *d_dir = nullptr, *d_tmin = nullptr, *d_t = nullptr, *d_u = nullptr, *d_v = nullptr;
    int32_t *d_found = nullptr, *d_tri = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
    auto alloc = [&](void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) tmp.push_back(*p);
        return e;
    };
#define TR(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    TR(alloc((void **)&d_src, n * 24)); TR(alloc((void **)&d_dir, n * 24)); TR(alloc((void **)&d_tmin, n * 8));
    TR(alloc((void **)&d_t, n * 8)); TR(alloc((void **)&d_u, n * 8)); TR(alloc((void **)&d_v, n * 8));
    TR(alloc((void **)&d_found, n * 4)); TR(alloc((void **)&d_tri, n * 4));
    hipStream_t s = sc->stream;
    TR(hipMemcpyAsync(d_src, src, n * 24, hipMemcpyHostToDevice, s));
    TR(hipMemcpyAsync(d_dir, dir, n * 24, hipMemcpyHostToDevice, s));
    TR(hipMemcpyAsync(d_tmin, tmin_in, n * 8, hipMemcpyHostToDevice, s));
    if (mode == 1 ? fast_trace(sc->S, n, d_src, d_dir, d_tmin, d_found, d_tri, d_t, d_u, d_v, s)
                  : exact_trace(sc->S, n, d_src, d_dir, d_tmin, d_found, d_tri, d_t, d_u, d_v, mode == 2, s)) {
        cleanup();
        return fail(PT_ERR_HIP, std::string("pt_trace launch: ") + hipGetErrorString(hipGetLastError()));
    }
    TR(hipMemcpyAsync(found, d_found, n * 4, hipMemcpyDeviceToHost, s));
    TR(hipMemcpyAsync(tri, d_tri, n * 4, hipMemcpyDeviceToHost, s));
    TR(hipMemcpyAsync(t, d_t, n * 8, hipMemcpyDeviceToHost, s));
    TR(hipMemcpyAsync(u, d_u, n * 8, hipMemcpyDeviceToHost, s));
    TR(hipMemcpyAsync(v, d_v, n * 8, hipMemcpyDeviceToHost, s));
    TR(hipStreamSynchronize(s));
#undef TR
    cleanup();
    if (int lrc = check_limits(sc, "pt_trace")) return lrc;
    for (size_t k = 0; k < n; k++) {
        int32_t g = tri[k];
        mesh[k] = g < 0 ? -1 : sc->tri_mesh[g];
        tri[k] = g < 0 ? -1 : sc->tri_local[g];
    }
    return PT_OK;
}); }

// No allocation on this path: no hipMalloc / hipFree, no std::vector, no std::string unless the call fails.
// The fixed scratch array is one per GPU, so the calls on one GPU are serialised whatever scene they are for
// (a static mutex per device ordinal); the caller's current device is put back before the call returns.
static std::mutex g_static_scratch_mutex[PT_STATIC_MAX_DEVICES];
int pt_trace_device(PtScene *sc, size_t n, const double *d_src, const double *d_dir, const double *d_tmin_in,
                    int32_t *d_found, int32_t *d_mesh, int32_t *d_tri, double *d_t, double *d_u, double *d_v) { return guarded([&]() -> int {
    if (!sc) return fail(PT_ERR_INVALID, "scene is null");
    if (g_device < 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not been called");
    if (n == 0) return PT_OK;
    if (!d_src || !d_dir || !d_tmin_in || !d_found || !d_mesh || !d_tri || !d_t || !d_u || !d_v)
        return fail(PT_ERR_INVALID, "pt_trace_device: null array");
    if (n > PT_STATIC_MAX_RAYS) return fail(PT_ERR_LIMIT, "pt_trace_device: more than 2^24 rays in one call");
    if ((uint32_t)sc->refine.max_depth + 2u > PT_STATIC_DEPTH)            // this path walks the binary records: one stack entry per level
        return fail(PT_ERR_LIMIT, "pt_trace_device: the scene's fp32 tree is deeper than the fixed scratch stack (62 levels)");
    if (sc->device < 0 || sc->device >= PT_STATIC_MAX_DEVICES) return fail(PT_ERR_LIMIT, "pt_trace_device: device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> device_lock(g_static_scratch_mutex[sc->device]);   // one scratch array per GPU
    std::lock_guard<std::mutex> lock(sc->busy);                                    // one scene, one stream
    int before = -1;
    (void)hipGetDevice(&before);
    HIPCHK(hipSetDevice(sc->device));
    int rc = fast_trace_static(sc->S, sc->device, n, d_src, d_dir, d_tmin_in, d_found, d_mesh, d_tri, d_t, d_u, d_v, sc->stream);
    if (rc == PT_ERR_HIP) rc = fail(rc, std::string("pt_trace_device: ") + hipGetErrorString(hipGetLastError()));
    else if (rc) rc = fail(rc, "pt_trace_device: no scratch address for this device (pt_init), or a size limit");
    else rc = check_limits(sc, "pt_trace_device");
    if (before >= 0 && before != sc->device) (void)hipSetDevice(before);
    return rc;
}); }

int pt_camera_rays(const PtCamera *cam, int size, double x, double y, double *dir_out) {
    if (g_device < 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not been called");
    if (!cam || !dir_out || size < 1 || size > 8192) return fail(PT_ERR_INVALID, "pt_camera_rays: bad argument");
    size_t P = (size_t)size * size;
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, P * 24));
    ViewParams V{};
    for (int k = 0; k < 3; k++) V.eye[k] = cam->eye[k];
    V.zoom = cam->zoom; V.sa = std::sin(cam->angle); V.ca = std::cos(cam->angle); V.size = size;
    int lrc = camera_rays(V, x, y, d, 0);
    hipError_t e = lrc ? hipErrorLaunchFailure : hipMemcpy(dir_out, d, P * 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PT_ERR_HIP, std::string("pt_camera_rays: ") + hipGetErrorString(e));
    return PT_OK;
}

int pt_hemisphere_samples(uint32_t seed, uint32_t view, uint32_t pixel, int n, double *out) {
    if (g_device < 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not been called");
    if (!out || n < 1 || n > (1 << 24)) return fail(PT_ERR_INVALID, "pt_hemisphere_samples: bad argument");
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, (size_t)n * 24));
    int lrc = hemisphere_samples(seed, view, pixel, n, d, 0);
    hipError_t e = lrc ? hipErrorLaunchFailure : hipMemcpy(out, d, (size_t)n * 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PT_ERR_HIP, std::string("pt_hemisphere_samples: ") + hipGetErrorString(e));
    return PT_OK;
}

}  // extern "C"
