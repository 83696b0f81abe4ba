// fp32 kd traversal as a per-lane state machine (the persistent-wave loop that
// drives it is in pt_persistent.h).  This file walks binary node records; pt_trace_quad.h
// is the same machine over quad records (two levels per record), which the kernels use
// whenever the tree has them.
//
// Why this shape.  Rays differ a lot in length (a ray that threads a cluster of
// fat nodes costs tens of times the mean), and the lanes of a wave are in
// different nodes all the time.  A wave that owned 64 fixed rays and ran
// "descend ... test leaf" loops would keep ~4 % of its lanes busy (measured).
// Here every lane advances by a unit or two per turn of the loop -- one triangle
// test if its current node still has untested triangles, then one node step
// (descend / pop / finish) if it has none left -- and a lane whose ray is finished
// takes the next one from the work queue.  Both kinds of unit are straight-line,
// predicated blocks that every lane of the wave walks through.
//
// Node record (16 bytes, four to a 64-byte line):
//   split | link | tri_off | tri_end
//   link    : 0xFFFFFFFF for a leaf, else (byte offset of the left child) | axis;
//             children are adjacent and the pair is 32-byte aligned: right = left + 16 = left ^ 16
//   tri_off : byte offset of the node's first 48-byte triangle record
//   tri_end : byte offset one past its last
// Triangle record (48 bytes): v0.xyz, triangle index | e2.x, e1.x, e2.y, e1.y | e2.z, e1.z, 0, 0
// (e1 = v1 - v0, e2 = v2 - v0; the edges are interleaved for the packed-fp32 test).
//
// The traversal is an exact nearest-hit search over the triangles of all visited
// nodes: triangles held by interior nodes are tested on the way down, a leaf's
// when it is reached; segments are visited front to back and the search ends when
// the next segment starts at or behind the best hit.  The stack holds (far child,
// far segment end) only -- a popped segment starts where the previous leaf's
// segment ended.  Its newest `slots - 1` entries live in a ring per lane in LDS;
// when the ring is full its oldest entry moves to the lane's column of an overflow
// stack in device memory and comes back when the ring has run empty, so no entry
// is ever lost (a deep tree overflows an 8-entry ring on a few percent of the
// rays).  A ring that dropped entries and restarted from the root instead cannot
// make progress through cells of zero length along the ray, which have to be
// visited (see the node step).
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

#include "pt_scene_dev.h"

#define FS_MAX_STEPS 200000    // hard bound on node steps per ray: every lane, so every wave, terminates

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct LaneState {
    float ox, oy, oz, ix, iy, iz;
    f32x2 dtx, dty, dtz;                // .x: ray direction component; .y: scratch of the triangle test (o - v0)
    float tmin, ent, ext, hu, hv;
    int htri, skip;
    uint32_t top, cs, novf;             // ring: free slot, (node steps << 8 | entries in the ring), entries in the overflow stack
    float nsplit;                       // current node
    uint32_t nlink, toff, tend;
};

struct RootBox { float lx, ly, lz, hx, hy, hz; };

__device__ __forceinline__ RootBox root_box(const DevScene &S) {
    RootBox b;
    b.lx = (float)S.root_lo[0]; b.ly = (float)S.root_lo[1]; b.lz = (float)S.root_lo[2];
    b.hx = (float)S.root_hi[0]; b.hy = (float)S.root_hi[1]; b.hz = (float)S.root_hi[2];
    return b;
}

__device__ __forceinline__ uint32_t tris_here(const LaneState &L) { return (L.tend - L.toff) / 48u; }

__device__ __forceinline__ float sel3(uint32_t a, float x, float y, float z) { return a == 0 ? x : (a == 1 ? y : z); }

// One Moeller-Trumbore test (two-sided) of the 48-byte triangle record at byte offset `toff` against the lane's ray;
// updates the lane's best hit and returns whether this triangle became it.  LS: LaneState or LaneStateQ (same ray fields).
// Both cross products and all four dot products run as packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32): lane .x of a
// pair works on P = d x e2, lane .y on Q = T x e1.
template <class LS>
__device__ __forceinline__ bool triangle_unit(LS &L, const DevScene &S, uint32_t toff) {
    const float4 *tp = (const float4 *)((const char *)S.refTriF + toff);
    const float4 A = tp[0], B = tp[1], C = tp[2];
    const f32x2 ex = {B.x, B.y}, ey = {B.z, B.w}, ez = {C.x, C.y};                 // (e2, e1) per axis
    L.dtx.y = L.ox - A.x; L.dty.y = L.oy - A.y; L.dtz.y = L.oz - A.z;              // (d, T = o - v0)
    const f32x2 dtx = L.dtx, dty = L.dty, dtz = L.dtz;
    const f32x2 pqx = dty * ez - dtz * ey, pqy = dtz * ex - dtx * ez, pqz = dtx * ey - dty * ex;   // (P, Q)
    const f32x2 dn = pqx * ex.yx + pqy * ey.yx + pqz * ez.yx;                      // (det = P . e1, Q . e2)
    const f32x2 uvn = pqx * dtx.yx + pqy * dty.yx + pqz * dtz.yx;                  // (P . T, Q . d)
    const float inv = __builtin_amdgcn_rcpf(dn.x);
    const float u = uvn.x * inv, v = uvn.y * inv, t = dn.y * inv;
    const int id = __float_as_int(A.w);
    // every comparison is false for a NaN, so a degenerate triangle never hits
    const bool hit = (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > 0.0f) & (t < L.tmin) & (id != L.skip);
    L.tmin = hit ? t : L.tmin; L.hu = hit ? u : L.hu; L.hv = hit ? v : L.hv; L.htri = hit ? id : L.htri;
    return hit;
}

// The ray into the lane state and its segment inside the scene's box; false when it misses the box.
template <class LS>
__device__ __forceinline__ bool ray_unit(LS &L, const RootBox &b, float ox, float oy, float oz, float dx, float dy, float dz, float tmax, int skip) {
    L.ox = ox; L.oy = oy; L.oz = oz; L.dtx.x = dx; L.dty.x = dy; L.dtz.x = dz;
    L.ix = __builtin_amdgcn_rcpf(dx); L.iy = __builtin_amdgcn_rcpf(dy); L.iz = __builtin_amdgcn_rcpf(dz);
    float a0 = (b.lx - ox) * L.ix, a1 = (b.hx - ox) * L.ix;
    float b0 = (b.ly - oy) * L.iy, b1 = (b.hy - oy) * L.iy;
    float c0 = (b.lz - oz) * L.iz, c1 = (b.hz - oz) * L.iz;
    L.ent = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fmaxf(fminf(c0, c1), 0.0f));
    L.ext = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fminf(fmaxf(c0, c1), tmax));
    L.tmin = tmax; L.htri = -1; L.hu = 0; L.hv = 0; L.skip = skip;
    L.top = 0; L.cs = 0; L.novf = 0;
    return L.ent <= L.ext;
}

__device__ __forceinline__ void load_node(LaneState &L, const DevScene &S, uint32_t off) {
    const uint4 n = *(const uint4 *)((const char *)S.nodeF + off);
    L.nsplit = __uint_as_float(n.x); L.nlink = n.y; L.toff = n.z; L.tend = n.w;
}

// Start a ray; false when it misses the scene's box (nothing to traverse).
__device__ __forceinline__ bool lane_begin(LaneState &L, const DevScene &S, const RootBox &b, float ox, float oy, float oz,
                                           float dx, float dy, float dz, float tmax, int skip) {
    if (!ray_unit(L, b, ox, oy, oz, dx, dy, dz, tmax, skip)) return false;
    load_node(L, S, 0);
    return true;
}

// One unit of work for this lane.  Returns true when the ray is finished
// (nearest hit in tmin / hu / hv / htri, htri < 0 for a miss; with ANY the first
// hit found ends the ray).  `stk` points at this lane's column of the LDS ring,
// entries 64 apart; slot_mask = slots - 1 >= 1.  One slot is kept free so that the
// push below can be an unconditional store.  `ovf` points at this lane's column
// of the overflow stack, entries 64 apart.
__device__ __forceinline__ uint2 *overflow_column(const DevScene &S) {
    return S.ovf + (size_t)blockIdx.x * S.ovf_depth * 64 + (threadIdx.x & 63);
}

template <bool ANY, bool COUNT>
__device__ __forceinline__ bool lane_turn(LaneState &L, const DevScene &S, uint2 *stk, uint32_t slot_mask, uint32_t *work, bool active = true) {
    bool done = false;
    if (active & (L.toff < L.tend)) {
        // ---- one triangle of the current node ----
        const bool hit = triangle_unit(L, S, L.toff);
        L.toff += 48u;
        if (ANY) done = hit;
    }
    if (active & !(L.toff < L.tend) & !done) {
    // ---- one node step ----
    const bool leaf = L.nlink == 0xFFFFFFFFu;
    const uint32_t axis = L.nlink & 3u, left = L.nlink & ~15u;
    const float oa = sel3(axis, L.ox, L.oy, L.oz), ia = sel3(axis, L.ix, L.iy, L.iz);
    const float tmid = (L.nsplit - oa) * ia;
    const uint32_t nr = left + (oa > L.nsplit ? 16u : 0u), fr = nr ^ 16u;
    // A plane met exactly at the segment's start or end sends the ray into BOTH children: a cell of
    // zero length along the ray may be the thin cell that holds an axis-aligned wall or floor
    // (the tree builder isolates those), and its triangles have to be tested.
    const bool near_only = !(tmid <= L.ext) | (tmid < 0.0f);    // also when tmid is NaN (ray in the split plane)
    const bool far_only = !near_only & (tmid < L.ent);
    const bool both = !near_only & !far_only;
    const bool push = !leaf & both & (tmid < L.tmin);
    // leaf: the next segment starts where this one ended
    const uint32_t cnt = L.cs & 0xFFu;
    const bool finished = leaf & (L.ext >= L.tmin);
    const bool cont = leaf & !finished;
    const bool pop_ring = cont & (cnt != 0);
    const bool pop_ovf = cont & (cnt == 0) & (L.novf != 0);
    const bool exhausted = cont & (cnt == 0) & (L.novf == 0);
    const bool pop = pop_ring | pop_ovf;
    const bool full = cnt == slot_mask;
    const uint32_t ptop = (L.top - 1u) & slot_mask;
    uint2 e = stk[ptop * 64];                                      // harmless when not popping
    if (pop_ovf | (push & full)) {                                 // rare: the ring ran empty or is full
        uint2 *ovf = overflow_column(S);
        if (pop_ovf) {
            L.novf--;
            e = ovf[(size_t)L.novf * 64];
        } else if (L.novf < S.ovf_depth) {                         // always true for a tree of the depth the stack was sized for
            ovf[(size_t)L.novf * 64] = stk[((L.top + 1u) & slot_mask) * 64];      // the oldest entry leaves the ring
            L.novf++;
        } else {
            atomicOr((unsigned int *)S.err, (unsigned int)PT_LIMIT_F32_STACK);   // the entry is lost: the call fails with PT_ERR_LIMIT
        }
    }
    stk[L.top * 64] = make_uint2(fr, __float_as_uint(L.ext));      // slot `top` is always free
    // next node: a leaf that neither pops nor restarts loads the root again, harmlessly (the ray is done)
    const uint32_t curr = leaf ? (pop ? e.x : 0u) : (far_only ? fr : nr);
    const float next_ext = leaf ? (pop ? __uint_as_float(e.y) : L.ext) : (both ? tmid : L.ext);
    L.ent = leaf ? L.ext : L.ent;
    L.ext = next_ext;
    // ring bookkeeping: a push onto a full ring takes the slot its oldest entry has just left
    L.top = push ? ((L.top + 1u) & slot_mask) : (pop_ring ? ptop : L.top);
    L.cs += 256u + (uint32_t)(push & !full) - (uint32_t)pop_ring;
    const bool runaway = L.cs > ((uint32_t)FS_MAX_STEPS << 8);
    if (runaway) atomicOr((unsigned int *)S.err, (unsigned int)PT_LIMIT_F32_STEPS);          // never seen; ends the ray so that the wave ends
    done = finished | exhausted | runaway;
    load_node(L, S, curr);
    if (COUNT && !done) { work[0]++; work[1] += tris_here(L); }
    }
    return done;
}
