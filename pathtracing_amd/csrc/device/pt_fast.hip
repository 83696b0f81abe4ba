// fp32 wavefront path for the GI sample rays (the bulk of the rays at any
// useful sample count):
//
//   k_gi_gen        G-buffer + counter RNG -> compacted ray queue (SoA)
//   k_trace_closest kd traversal, nearest hit, per-lane short stack in LDS
//   k_shade_gi      surface properties + lights; OMP model: writes the sample's
//                   contribution; MAIN model: compacts hits into a shading-point queue
//   k_shadow_gi     MAIN model: any-hit traversal towards every light
//   k_gi_sum        ordered per-pixel sum of the pass's contributions (fp64)
//
// Compaction between stages is a wave ballot + prefix popcount + one atomicAdd
// per wave.  Every sample owns a contribution slot, so the image does not depend
// on the order in which waves append to the queues.
#include <float.h>
#include <stdlib.h>

#include "pt_frame.h"
#include "pt_vec.h"

#pragma clang fp contract(fast)

#define FS_SLOTS 8             // short-stack entries per lane (LDS); deeper pushes drop the oldest entry
#define FS_MAX_STEPS 100000    // hard bound on node visits per ray: every wave terminates

struct HitF { float t, u, v; int tri; };

__device__ __forceinline__ float sel3(uint32_t a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// Nearest hit (ANY = false) or first hit (ANY = true) of the ray o + t*d, 0 < t < tmax,
// ignoring triangle `skip`.  Triangles referenced by interior nodes are tested on
// the way down; a leaf's triangles when it is reached.  Front-to-back order with
// the exit test "segment start >= best t" makes the result the exact nearest hit
// over the triangles of all visited nodes.
//
// The stack holds (far child, far segment end) only: a popped segment starts
// where the previous leaf's segment ended.  It is a ring of FS_SLOTS entries per
// lane in LDS; when it overflows the oldest entry is dropped and, once the ring
// runs empty, traversal restarts from the root at the current segment start.
//
// COUNT adds two per-lane counters (nodes visited, triangles tested) for the
// work census that prices the kernel's algorithmic bytes; the timed build has
// COUNT = false.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool trace_f32(const DevScene &S, float ox, float oy, float oz, float dx, float dy, float dz,
                                          float tmax, int skip, uint2 *stk, HitF &h, uint32_t *work = nullptr) {
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float ent, ext;
    {
        float lx = (float)S.root_lo[0], ly = (float)S.root_lo[1], lz = (float)S.root_lo[2];
        float hx = (float)S.root_hi[0], hy = (float)S.root_hi[1], hz = (float)S.root_hi[2];
        float a0 = (lx - ox) * ix, a1 = (hx - ox) * ix;
        float b0 = (ly - oy) * iy, b1 = (hy - oy) * iy;
        float c0 = (lz - oz) * iz, c1 = (hz - oz) * iz;
        ent = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
        ext = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tmax));
        if (!(ent <= ext)) return false;
    }
    const float root_ext = ext;
    float tmin = tmax;
    bool found = false;
    uint32_t curr = 0, top = 0, cnt = 0;
    bool dropped = false;
    int steps = 0;
    for (;;) {
        NodeF n = S.nodeF[curr];
        for (;;) {
            if (COUNT) { work[0]++; work[1] += n.count; }
            if (n.count) {
                // The next triangle's 48 bytes are requested before the current one is
                // tested, so that two loads per lane are in flight along the list.
                const float4 *tp = S.refTriF + (size_t)n.first * 3;
                float4 A = tp[0], B = tp[1], C = tp[2];
                for (uint32_t i = 0; i < n.count; i++) {
                    tp += (i + 1 < n.count) ? 3 : 0;
                    float4 An = tp[0], Bn = tp[1], Cn = tp[2];
                    float px = dy * C.z - dz * C.y, py = dz * C.x - dx * C.z, pz = dx * C.y - dy * C.x;
                    float det = px * B.x + py * B.y + pz * B.z;
                    float inv = 1.0f / det;
                    float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
                    float u = (px * tx + py * ty + pz * tz) * inv;
                    float qx = ty * B.z - tz * B.y, qy = tz * B.x - tx * B.z, qz = tx * B.y - ty * B.x;
                    float v = (qx * dx + qy * dy + qz * dz) * inv;
                    float t = (qx * C.x + qy * C.y + qz * C.z) * inv;
                    int id = __float_as_int(A.w);
                    if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > 0.0f && t < tmin && id != skip) {
                        tmin = t; h.t = t; h.u = u; h.v = v; h.tri = id;
                        found = true;
                        if (ANY) return true;
                    }
                    A = An; B = Bn; C = Cn;
                }
            }
            if (n.link == 0xFFFFFFFFu) break;
            uint32_t axis = n.link >> 30, left = n.link & 0x3FFFFFFFu;
            float oa = sel3(axis, ox, oy, oz), ia = sel3(axis, ix, iy, iz);
            float tmid = (n.split - oa) * ia;
            uint32_t side = oa > n.split ? 1u : 0u;
            uint32_t nr = left + side, fr = left + 1u - side;
            if (!(tmid < ext) || tmid < 0.0f) {
                curr = nr;
            } else if (tmid <= ent) {
                curr = fr;
            } else {
                if (tmid < tmin) {
                    stk[top * 64] = make_uint2(fr, __float_as_uint(ext));
                    top = (top + 1) & (FS_SLOTS - 1);
                    if (cnt == FS_SLOTS) dropped = true; else cnt++;
                }
                curr = nr;
                ext = tmid;
            }
            n = S.nodeF[curr];
            if (++steps > FS_MAX_STEPS) return found;
        }
        ent = ext;                       // the next segment starts where this leaf's ended
        if (ent >= tmin) break;
        if (cnt > 0) {
            top = (top - 1) & (FS_SLOTS - 1);
            cnt--;
            uint2 e = stk[top * 64];
            curr = e.x; ext = __uint_as_float(e.y);
        } else if (dropped) {
            dropped = false;
            curr = 0; ext = root_ext;
        } else {
            break;
        }
        if (++steps > FS_MAX_STEPS) break;
    }
    return found;
}

// ---- stage 1: ray generation from the G-buffer --------------------------------
__device__ __forceinline__ bool tile_pixel_f(int linear, int size, int &i, int &j) {
    int tiles_x = (size + 7) >> 3;
    int tile = linear >> 6, lane = linear & 63;
    int ti = tile / tiles_x, tj = tile - ti * tiles_x;
    i = ti * 8 + (lane >> 3);
    j = tj * 8 + (lane & 7);
    return i < size && j < size;
}

// createCoordSystem + sample direction, util/lighting.cpp:42-59 and main.cpp:94-96,
// in fp64 so that the fast path shoots the same sample directions as the fp64 path.
__device__ __forceinline__ D3 gi_direction(D3 nrm, uint32_t seed, uint32_t view, uint32_t pixel, uint32_t sample, double &r1) {
#pragma clang fp contract(off)
    D3 Nt, Nb;
    if (fabs(nrm.x) > fabs(nrm.y))
        Nt = ddivs(d3(nrm.z, 0, -nrm.x), sqrt(nrm.x * nrm.x + nrm.z * nrm.z));
    else
        Nt = ddivs(d3(0, -nrm.z, nrm.y), sqrt(nrm.y * nrm.y + nrm.z * nrm.z));
    Nb = dcross(nrm, Nt);
    D3 sm = sample_hemisphere(seed, view, pixel, sample, r1);
    return d3(sm.x * Nb.x + sm.y * nrm.x + sm.z * Nt.x,
              sm.x * Nb.y + sm.y * nrm.y + sm.z * Nt.y,
              sm.x * Nb.z + sm.y * nrm.z + sm.z * Nt.z);
}

__global__ void __launch_bounds__(64) k_gi_gen(ViewParams V, FrameBuffers F, FastBuffers Q, int s0, int tthreads) {
    int lin = blockIdx.x * 64 + threadIdx.x;
    int sl = lin / tthreads;
    int i, j;
    bool live = tile_pixel_f(lin - sl * tthreads, V.size, i, j);
    size_t P = (size_t)V.size * V.size, p = live ? (size_t)i * V.size + j : 0;
    bool valid = false;
    float4 ro, rd;
    if (live) {
        D3 hit = dload(F.ghit + 3 * p);
        valid = !(hit.x == -99999);
        if (valid) {
            double r1;
            D3 w = gi_direction(dload(F.gnrm + 3 * p), V.seed, V.view, (uint32_t)p, (uint32_t)(s0 + sl), r1);
            ro = make_float4((float)hit.x, (float)hit.y, (float)hit.z, FLT_MAX);
            rd = make_float4((float)w.x, (float)w.y, (float)w.z, (float)r1);
        } else {
            Q.contrib[(size_t)sl * P + p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    unsigned long long mask = __ballot(valid);
    int lane = threadIdx.x & 63;
    uint32_t n = (uint32_t)__popcll(mask), base = 0;
    if (lane == 0 && n) base = atomicAdd(&Q.counts[0], n);
    base = __shfl(base, 0, 64);
    if (valid) {
        uint32_t r = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        Q.rays_o[r] = ro;
        Q.rays_d[r] = rd;
        Q.ray_slot[r] = (uint32_t)((size_t)sl * P + p);
        Q.ray_skip[r] = F.rec_tri[p];
    }
}

// ---- stage 2: closest hit -------------------------------------------------------
// Persistent waves.  Rays differ a lot in length (a ray that threads a cluster
// of fat leaves costs tens of times the mean), so a wave that owned 64 fixed
// rays would idle most of its lanes most of the time.  Instead every lane runs
// a small state machine and advances by ONE unit per loop turn -- one triangle
// test if its current node still has untested triangles, else one node step
// (descend / pop / restart) -- and lanes whose ray is finished draw new rays
// from the queue head (one atomicAdd per wave, when at least PT_REFILL_MIN
// lanes are idle).  A wave leaves when the queue is drained and all its lanes
// are done, so the grid always drains.
#define PT_REFILL_MIN 8
#define PT_QUEUE_CHUNK 128          // rays a wave takes from the queue head per atomicAdd (upper bound)
#define PT_TRACE_WAVES (256 * 32)     // enough waves to fill every CU; surplus waves find the queue empty

template <bool COUNT>
__global__ void __launch_bounds__(64) k_trace_closest(DevScene S, FastBuffers Q, uint32_t slot_mask, int refill_min, uint32_t chunk) {
    extern __shared__ uint2 stack[];            // (slot_mask + 1) * 64 entries
    const uint32_t lane = threadIdx.x & 63;
    uint2 *stk = stack + lane;
    const uint32_t n_rays = Q.counts[0];
    const float lx = (float)S.root_lo[0], ly = (float)S.root_lo[1], lz = (float)S.root_lo[2];
    const float hx = (float)S.root_hi[0], hy = (float)S.root_hi[1], hz = (float)S.root_hi[2];

    bool active = false, more = true;
    uint32_t r = 0, chunk_next = 0, chunk_end = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0;
    float tmin = 0, ent = 0, ext = 0, root_ext = 0, hu = 0, hv = 0;
    int htri = -1, skip = -1;
    uint32_t curr = 0, top = 0, cnt = 0, ti = 0, te = 0, steps = 0;
    bool dropped = false;
    NodeF n; n.split = 0; n.link = 0xFFFFFFFFu; n.first = 0; n.count = 0;
    uint32_t w_nodes = 0, w_tris = 0;

    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (more && __popcll(idle) >= refill_min) {
            // The wave owns a chunk [chunk_next, chunk_end) of the queue; one atomicAdd on
            // the queue head buys a whole chunk, so the head is touched once per `chunk`
            // rays instead of once per refill.
            if (chunk_next == chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&Q.counts[3], chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_next = base < n_rays ? base : n_rays;
                chunk_end = base + chunk < n_rays ? base + chunk : n_rays;
                if (chunk_next == chunk_end) more = false;
            }
            uint32_t want = (uint32_t)__popcll(idle), have = chunk_end - chunk_next;
            uint32_t take = want < have ? want : have;
            uint32_t base = chunk_next;
            chunk_next += take;
            if (!active) {
                uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                uint32_t mine = k < take ? base + k : n_rays;
                if (mine < n_rays) {
                    r = mine;
                    float4 o = Q.rays_o[r], d = Q.rays_d[r];
                    skip = Q.ray_skip[r];
                    ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
                    ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;
                    float a0 = (lx - ox) * ix, a1 = (hx - ox) * ix;
                    float b0 = (ly - oy) * iy, b1 = (hy - oy) * iy;
                    float c0 = (lz - oz) * iz, c1 = (hz - oz) * iz;
                    ent = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
                    ext = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), o.w));
                    tmin = o.w; htri = -1; hu = 0; hv = 0;
                    if (ent <= ext) {
                        active = true;
                        root_ext = ext;
                        curr = 0; top = 0; cnt = 0; dropped = false; steps = 0;
                        n = S.nodeF[0];
                        ti = n.first; te = n.first + n.count;
                        if (COUNT) { w_nodes++; w_tris += n.count; }
                    } else {
                        Q.hits[r] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
                    }
                }
            }
        }
        if (!__any(active)) {
            if (!more) break;
            continue;
        }
        if (active) {
            if (ti < te) {
                // ---- one triangle of the current node ----
                const float4 *tp = (const float4 *)((const char *)S.refTriF + ti * 48u);   // 32-bit offset: < 2^32 bytes, checked at upload
                float4 A = tp[0], B = tp[1], C = tp[2];
                ti++;
                float px = dy * C.z - dz * C.y, py = dz * C.x - dx * C.z, pz = dx * C.y - dy * C.x;
                float det = px * B.x + py * B.y + pz * B.z;
                float inv = __builtin_amdgcn_rcpf(det);
                float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
                float u = (px * tx + py * ty + pz * tz) * inv;
                float qx = ty * B.z - tz * B.y, qy = tz * B.x - tx * B.z, qz = tx * B.y - ty * B.x;
                float v = (qx * dx + qy * dy + qz * dz) * inv;
                float t = (qx * C.x + qy * C.y + qz * C.z) * inv;
                int id = __float_as_int(A.w);
                if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > 0.0f && t < tmin && id != skip) {
                    tmin = t; hu = u; hv = v; htri = id;
                }
            } else {
                // ---- one node step ----
                bool done = false;
                if (n.link == 0xFFFFFFFFu) {
                    ent = ext;                      // the next segment starts where this leaf's ended
                    if (ent >= tmin) {
                        done = true;
                    } else if (cnt > 0) {
                        top = (top - 1) & slot_mask;
                        cnt--;
                        uint2 e = stk[top * 64];
                        curr = e.x; ext = __uint_as_float(e.y);
                    } else if (dropped) {
                        dropped = false;
                        curr = 0; ext = root_ext;
                    } else {
                        done = true;
                    }
                } else {
                    uint32_t axis = n.link >> 30, left = n.link & 0x3FFFFFFFu;
                    float oa = sel3(axis, ox, oy, oz), ia = sel3(axis, ix, iy, iz);
                    float tmid = (n.split - oa) * ia;
                    uint32_t side = oa > n.split ? 1u : 0u;
                    uint32_t nr = left + side, fr = left + 1u - side;
                    if (!(tmid < ext) || tmid < 0.0f) {
                        curr = nr;
                    } else if (tmid <= ent) {
                        curr = fr;
                    } else {
                        if (tmid < tmin) {
                            stk[top * 64] = make_uint2(fr, __float_as_uint(ext));
                            top = (top + 1) & slot_mask;
                            if (cnt > slot_mask) dropped = true; else cnt++;
                        }
                        curr = nr;
                        ext = tmid;
                    }
                }
                if (++steps > FS_MAX_STEPS) done = true;
                if (done) {
                    Q.hits[r] = make_float4(tmin, hu, hv, __int_as_float(htri));
                    active = false;
                } else {
                    n = *(const NodeF *)((const char *)S.nodeF + curr * 16u);
                    ti = n.first; te = n.first + n.count;
                    if (COUNT) { w_nodes++; w_tris += n.count; }
                }
            }
        }
    }
    if (COUNT) {
        uint32_t w[2] = {w_nodes, w_tris};
        for (int k = 0; k < 2; k++) {
            uint32_t v = w[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) atomicAdd(&Q.work[k], (unsigned long long)v);
        }
    }
}

// ---- stage 3: shading of the GI hits ---------------------------------------------
struct SurfF { float nx, ny, nz, tr, tg, tb; };

__device__ __forceinline__ int clampi_f(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ const uint8_t *texel_f(const DevScene &S, int tex, int px, int py) {
    int rows = S.tex_rows[tex], cols = S.tex_cols[tex];
    int row = clampi_f(py, 0, rows - 1), col = clampi_f(px, 0, cols - 1);
    return S.texels + S.tex_off[tex] + ((size_t)row * cols + col) * 3;
}

// surfaceProperties (util/polygonMesh.cpp:81-118) in fp32.
__device__ __forceinline__ SurfF surface_f(const DevScene &S, int tri, float bu, float bv) {
    SurfF o;
    const int32_t *N = S.tri_ni + (size_t)tri * 3;
    float w = 1.0f - bu - bv;
    const float *n0 = S.normF + (size_t)N[0] * 3, *n1 = S.normF + (size_t)N[1] * 3, *n2 = S.normF + (size_t)N[2] * 3;
    float nx = w * n0[0] + bu * n1[0] + bv * n2[0];
    float ny = w * n0[1] + bu * n1[1] + bv * n2[1];
    float nz = w * n0[2] + bu * n1[2] + bv * n2[2];
    float il = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    o.nx = nx * il; o.ny = ny * il; o.nz = nz * il;
    int mn = S.tri_tm[tri];
    if (mn == -99999) { o.tr = o.tg = o.tb = 1.0f; return o; }
    const int32_t *T = S.tri_ti + (size_t)tri * 3;
    if (mn < 0) {
        mn = -(mn + 1);
        float rows = (float)S.tex_rows[mn];
        const float *a = S.uvF + (size_t)T[0] * 2, *b = S.uvF + (size_t)T[1] * 2, *c = S.uvF + (size_t)T[2] * 2;
        int r = (int)((a[0] * w + b[0] * bu + c[0] * bv) * rows);
        int cc = (int)((a[1] * w + b[1] * bu + c[1] * bv) * rows);
        const uint8_t *p = texel_f(S, mn, r, cc);
        o.tr = p[2] * (1.0f / 256.0f); o.tg = p[1] * (1.0f / 256.0f); o.tb = p[0] * (1.0f / 256.0f);
    } else {
        int rows = S.tex_rows[mn], cols = S.tex_cols[mn];
        float cr = 0, cg = 0, cb = 0;
        float wt[3] = {w, bu, bv};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int r = 0, c = 0;
            if (T[i] != -1) {
                r = (int)(S.uvF[(size_t)T[i] * 2] * rows);
                c = (int)(S.uvF[(size_t)T[i] * 2 + 1] * cols);
            }
            const uint8_t *p = texel_f(S, mn, r, c);
            cr += wt[i] * p[2]; cg += wt[i] * p[1]; cb += wt[i] * p[0];
        }
        o.tr = cr * (1.0f / 256.0f); o.tg = cg * (1.0f / 256.0f); o.tb = cb * (1.0f / 256.0f);
    }
    return o;
}

// Light::illuminate (util/lighting.cpp:24-36) in fp32: direction, distance, radiance.
__device__ __forceinline__ void light_f(const PtLight &L, float hx, float hy, float hz, float &lx, float &ly, float &lz,
                                        float &dis, float &sr, float &sg, float &sb) {
    float px = (float)L.pos[0], py = (float)L.pos[1], pz = (float)L.pos[2], mag = (float)L.mag;
    if (L.type == 'd') {
        float il = 1.0f / sqrtf(px * px + py * py + pz * pz);
        lx = px * il; ly = py * il; lz = pz * il;
        dis = FLT_MAX;
        sr = mag * (float)L.color[0]; sg = mag * (float)L.color[1]; sb = mag * (float)L.color[2];
    } else {
        lx = px - hx; ly = py - hy; lz = pz - hz;
        dis = sqrtf(lx * lx + ly * ly + lz * lz);
        float il = 1.0f / dis;
        lx *= il; ly *= il; lz *= il;
        float k = mag / (4.0f * dis * 3.14159265358979f);
        sr = k * (float)L.color[0]; sg = k * (float)L.color[1]; sb = k * (float)L.color[2];
    }
}

template <int MODEL>
__global__ void __launch_bounds__(64) k_shade_gi(DevScene S, FastBuffers Q) {
    uint32_t r = blockIdx.x * 64 + threadIdx.x;
    bool in_range = r < Q.counts[0];
    bool hit = false;
    float4 sp_pos, sp_nrm, sp_base;
    if (in_range) {
        float4 h = Q.hits[r];
        int tri = __float_as_int(h.w);
        uint32_t slot = Q.ray_slot[r];
        if (tri < 0) {
            Q.contrib[slot] = make_float4(0.f, 0.f, 0.f, 0.f);     // castRay returns 0 on a miss at depth -1
        } else {
            float4 o = Q.rays_o[r], d = Q.rays_d[r];
            SurfF sf = surface_f(S, tri, h.y, h.z);
            const double *ap = S.albedo + (size_t)S.tri_mesh[tri] * 3;
            float ar = sf.tr * (float)ap[0], ag = sf.tg * (float)ap[1], ab = sf.tb * (float)ap[2];
            float hx = o.x + h.x * d.x + 0.0001f * sf.nx, hy = o.y + h.x * d.y + 0.0001f * sf.ny,
                  hz = o.z + h.x * d.z + 0.0001f * sf.nz;
            if (MODEL == PT_MODEL_OMP) {
                // mainOMP.cpp:55-62,74-75: every light is visible; r1 * txt*albedo*direct/pi
                float dr = 0, dg = 0, db = 0;
                for (int k = 0; k < S.n_light; k++) {
                    float lx, ly, lz, dis, sr, sg, sb;
                    light_f(S.lights[k], hx, hy, hz, lx, ly, lz, dis, sr, sg, sb);
                    float c = fmaxf(0.0f, lx * sf.nx + ly * sf.ny + lz * sf.nz);
                    dr += c * sr; dg += c * sg; db += c * sb;
                }
                float k = d.w * (1.0f / 3.14159265358979f);
                Q.contrib[slot] = make_float4(k * ar * dr, k * ag * dg, k * ab * db, 0.f);
            } else {
                hit = true;
                float k = d.w * 0.6f * (1.0f / 3.14159265358979f);            // main.cpp:77 with ref = 0
                sp_pos = make_float4(hx, hy, hz, __int_as_float(tri));
                sp_nrm = make_float4(sf.nx, sf.ny, sf.nz, __uint_as_float(slot));
                sp_base = make_float4(k * ar, k * ag, k * ab, 0.f);
            }
        }
    }
    if (MODEL == PT_MODEL_MAIN) {
        unsigned long long mask = __ballot(hit);
        int lane = threadIdx.x & 63;
        uint32_t n = (uint32_t)__popcll(mask), base = 0;
        if (lane == 0 && n) base = atomicAdd(&Q.counts[1], n);
        base = __shfl(base, 0, 64);
        if (hit) {
            uint32_t q = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            Q.sp_pos[q] = sp_pos; Q.sp_nrm[q] = sp_nrm; Q.sp_base[q] = sp_base;
        }
    }
}

// ---- stage 4 (MAIN model): shadow rays of the GI hits ------------------------------
__global__ void __launch_bounds__(64) k_shadow_gi(DevScene S, FastBuffers Q) {
    __shared__ uint2 stack[FS_SLOTS * 64];
    uint32_t q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Q.counts[1]) return;
    float4 pos = Q.sp_pos[q], nrm = Q.sp_nrm[q], base = Q.sp_base[q];
    int tri = __float_as_int(pos.w);
    uint32_t shadow = 0;
    float dr = 0, dg = 0, db = 0;
    for (int k = 0; k < S.n_light; k++) {
        float lx, ly, lz, dis, sr, sg, sb;
        light_f(S.lights[k], pos.x, pos.y, pos.z, lx, ly, lz, dis, sr, sg, sb);
        float c = lx * nrm.x + ly * nrm.y + lz * nrm.z;
        if (c > 0.0f) {                                           // a light behind the surface adds 0 whatever the shadow ray says
            HitF h;
            shadow++;
            bool occluded = trace_f32<true, false>(S, pos.x, pos.y, pos.z, lx, ly, lz, dis, tri, stack + (threadIdx.x & 63), h);
            if (!occluded) { dr += c * sr; dg += c * sg; db += c * sb; }
        }
    }
    Q.contrib[__float_as_uint(nrm.w)] = make_float4(base.x * dr, base.y * dg, base.z * db, 0.f);
    // shadow-ray census: one atomic per wave
    for (int o = 32; o > 0; o >>= 1) shadow += __shfl_xor(shadow, o, 64);
    if ((threadIdx.x & 63) == 0 && shadow) atomicAdd(&Q.counts[2], shadow);
}

// ---- stage 5: indirect += r1 * L over the pass's samples, in sample order ------------
__global__ void k_gi_sum(FrameBuffers F, FastBuffers Q, int size, int nb, int first_batch) {
    size_t P = (size_t)size * size, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double ax = 0, ay = 0, az = 0;
    if (!first_batch) { ax = F.gi_acc[3 * p]; ay = F.gi_acc[3 * p + 1]; az = F.gi_acc[3 * p + 2]; }
    for (int s = 0; s < nb; s++) {
        float4 c = Q.contrib[(size_t)s * P + p];
        ax += (double)c.x; ay += (double)c.y; az += (double)c.z;
    }
    F.gi_acc[3 * p] = ax; F.gi_acc[3 * p + 1] = ay; F.gi_acc[3 * p + 2] = az;
}

// ---- raw queries for tests: fp64 in/out around the fp32 traversal -------------------
__global__ void __launch_bounds__(64) k_trace_fast_f64io(DevScene S, size_t n, const double *src, const double *dir,
                                                         const double *tmin_in, int32_t *found, int32_t *tri, double *t,
                                                         double *u, double *v) {
    __shared__ uint2 stack[FS_SLOTS * 64];
    size_t k = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= n) return;
    HitF h; h.t = 0; h.u = 0; h.v = 0; h.tri = -1;
    double tm = tmin_in[k];
    float tmax = tm >= (double)FLT_MAX ? FLT_MAX : (float)tm;
    bool ok = trace_f32<false, false>(S, (float)src[3 * k], (float)src[3 * k + 1], (float)src[3 * k + 2], (float)dir[3 * k],
                               (float)dir[3 * k + 1], (float)dir[3 * k + 2], tmax, -1, stack + (threadIdx.x & 63), h);
    found[k] = ok; tri[k] = h.tri; t[k] = ok ? (double)h.t : tm; u[k] = h.u; v[k] = h.v;
}

// ---- host side ------------------------------------------------------------------------
#define FQ(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return PT_ERR_HIP; } while (0)

int fast_alloc(FastBuffers &Q, size_t pixels, int batch, PtAllocFn alloc, void *ctx) {
    size_t n = pixels * (size_t)batch;
    if (n > 0xFFFFFF00u) return PT_ERR_LIMIT;
    Q.max_rays = n;
    int rc;
#define FA(field, type) if ((rc = alloc(ctx, n * sizeof(type), (void **)&Q.field))) return rc
    FA(rays_o, float4); FA(rays_d, float4); FA(ray_slot, uint32_t); FA(ray_skip, int32_t); FA(hits, float4);
    FA(sp_pos, float4); FA(sp_nrm, float4); FA(sp_base, float4); FA(contrib, float4);
#undef FA
    if ((rc = alloc(ctx, 8 * sizeof(uint32_t), (void **)&Q.counts))) return rc;
    if ((rc = alloc(ctx, 2 * sizeof(unsigned long long), (void **)&Q.work))) return rc;
    return PT_OK;
}

// Launch parameters of the closest-hit kernel.  The defaults are what the
// measurements in profiles/ were taken with; the PT_TUNE_* environment
// variables exist for sweeps.
struct TraceTune { int slots, refill, waves, chunk; };
static TraceTune trace_tune() {
    static TraceTune t = [] {
        TraceTune v{FS_SLOTS, PT_REFILL_MIN, PT_TRACE_WAVES, PT_QUEUE_CHUNK};
        if (const char *e = getenv("PT_TUNE_SLOTS")) { int k = atoi(e); if (k == 4 || k == 8 || k == 16 || k == 32) v.slots = k; }
        if (const char *e = getenv("PT_TUNE_REFILL")) { int k = atoi(e); if (k >= 1 && k <= 64) v.refill = k; }
        if (const char *e = getenv("PT_TUNE_CHUNK")) { int k = atoi(e); if (k >= 64 && k <= 65536) v.chunk = k; }
        if (const char *e = getenv("PT_TUNE_WAVES")) { int k = atoi(e); if (k >= 64 && k <= (1 << 20)) v.waves = k; }
        return v;
    }();
    return t;
}

int fast_gi(const DevScene &S, const ViewParams &V, const FrameBuffers &F, const FastBuffers &Q, int nsamples, int batch,
            hipStream_t s, hipEvent_t e0, hipEvent_t e1, float *trace_ms, uint64_t *launches, uint64_t *rays, uint64_t *shadow,
            uint64_t *work /* null, or [2]: nodes visited, triangles tested by the closest-hit launches */) {
    const int tthreads = tile_threads(V.size);
    const size_t P = (size_t)V.size * V.size;
    const TraceTune tune = trace_tune();
    *trace_ms = 0; *launches = 0; *rays = 0; *shadow = 0;
    if (work) FQ(hipMemsetAsync(Q.work, 0, 2 * sizeof(unsigned long long), s));
    for (int s0 = 0; s0 < nsamples; s0 += batch) {
        int nb = nsamples - s0 < batch ? nsamples - s0 : batch;
        if ((size_t)nb * P > Q.max_rays) return PT_ERR_LIMIT;
        unsigned blocks = (unsigned)(((size_t)tthreads * nb) / 64);
        unsigned qblocks = (unsigned)((P * nb + 63) / 64);
        FQ(hipMemsetAsync(Q.counts, 0, 8 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_gi_gen, dim3(blocks), dim3(64), 0, s, V, F, Q, s0, tthreads);
        FQ(hipEventRecord(e0, s));
        unsigned tblocks = qblocks < (unsigned)tune.waves ? qblocks : (unsigned)tune.waves;
        size_t lds = (size_t)tune.slots * 64 * sizeof(uint2);
        // queue chunk per atomicAdd: large enough to keep the queue head cold, small enough
        // that every resident wave gets several chunks
        uint32_t chunk = (uint32_t)((P * nb) / ((size_t)tune.waves * 4));
        chunk = chunk < 64 ? 64 : (chunk > (uint32_t)tune.chunk ? (uint32_t)tune.chunk : chunk);
        if (work)
            hipLaunchKernelGGL(k_trace_closest<true>, dim3(tblocks), dim3(64), lds, s, S, Q, (uint32_t)tune.slots - 1u, tune.refill, chunk);
        else
            hipLaunchKernelGGL(k_trace_closest<false>, dim3(tblocks), dim3(64), lds, s, S, Q, (uint32_t)tune.slots - 1u, tune.refill, chunk);
        FQ(hipEventRecord(e1, s));
        if (V.model == PT_MODEL_MAIN) {
            hipLaunchKernelGGL(k_shade_gi<PT_MODEL_MAIN>, dim3(qblocks), dim3(64), 0, s, S, Q);
            hipLaunchKernelGGL(k_shadow_gi, dim3(qblocks), dim3(64), 0, s, S, Q);
        } else {
            hipLaunchKernelGGL(k_shade_gi<PT_MODEL_OMP>, dim3(qblocks), dim3(64), 0, s, S, Q);
        }
        hipLaunchKernelGGL(k_gi_sum, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, F, Q, V.size, nb, s0 == 0);
        FQ(hipGetLastError());
        uint32_t c[8];
        FQ(hipMemcpyAsync(c, Q.counts, sizeof c, hipMemcpyDeviceToHost, s));
        FQ(hipStreamSynchronize(s));
        float ms = 0;
        FQ(hipEventElapsedTime(&ms, e0, e1));
        *trace_ms += ms; (*launches)++;
        *rays += c[0]; *shadow += c[2];
    }
    if (work) {
        unsigned long long w[2];
        FQ(hipMemcpy(w, Q.work, sizeof w, hipMemcpyDeviceToHost));
        work[0] = w[0]; work[1] = w[1];
    }
    return PT_OK;
}

int fast_trace(const DevScene &S, size_t n, const double *src, const double *dir, const double *tmin_in,
               int32_t *found, int32_t *tri, double *t, double *u, double *v, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_fast_f64io, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, S, n, src, dir, tmin_in, found, tri, t, u, v);
    return hipGetLastError() == hipSuccess ? PT_OK : PT_ERR_HIP;
}
