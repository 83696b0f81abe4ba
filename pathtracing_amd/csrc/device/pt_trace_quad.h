// fp32 kd traversal over "quad" records: one 16-byte record holds an interior node's split plane AND the split planes
// of its two children, so one request (what bounds the kernel) takes a ray down two levels of the tree.
//
// Quad record:  s0 | s1 | s2 | link        s0: the node's plane, s1 / s2: the left / right child's
//   link = (byte offset of the group of four 16-byte records of the grandchildren, 64-byte aligned) | a2 << 4 | a1 << 2 | a0
//   a0: the node's axis (0..2); a1, a2: the children's axes, 3 = that child is a leaf
//   group: [left-left, left-right, right-left, right-right]; a leaf child's record sits in the first slot of its half
// Leaf record:  tri_off | tri_end | 0 | 0xFFFFFFFF   (byte offsets into the 48-byte triangle records)
// Everything else -- the triangle test, the ring stack with its overflow column, front-to-back order, both children on
// equality -- is as in pt_trace_f32.h; the visiting order of the leaves is the same, so the results are the same bit for bit.
#pragma once
#include "pt_trace_f32.h"

#ifndef PT_QUAD_MERGED
#define PT_QUAD_MERGED 0
#endif

struct LaneStateQ {
    float ox, oy, oz, ix, iy, iz;
    f32x2 dtx, dty, dtz;
    float tmin, ent, ext, hu, hv;
    int htri, skip;
    uint32_t top, cs, novf;
    uint32_t n0, n1, n2, nlink;         // current record: quad (s0, s1, s2, link) or leaf (tri_off, tri_end, -, ~0)
};

__device__ __forceinline__ void load_quad(LaneStateQ &L, const DevScene &S, uint32_t off) {
    const uint4 n = *(const uint4 *)((const char *)S.quadF + off);
    L.n0 = n.x; L.n1 = n.y; L.n2 = n.z; L.nlink = n.w;
}

__device__ __forceinline__ bool lane_begin(LaneStateQ &L, const DevScene &S, const RootBox &b, float ox, float oy, float oz,
                                             float dx, float dy, float dz, float tmax, int skip) {
    if (!ray_unit(L, b, ox, oy, oz, dx, dy, dz, tmax, skip)) return false;
    load_quad(L, S, 0);
    return true;
}

// Triangles of the record the lane stands on (0 for a quad): what the counting build of a kernel adds per arrival.
__device__ __forceinline__ uint32_t tris_here(const LaneStateQ &L) { return L.nlink == 0xFFFFFFFFu ? (L.n1 - L.n0) / 48u : 0u; }

// Push (node, ext) when `pred`: the ring keeps one slot free; when it is full its oldest entry moves to the overflow column.
__device__ __forceinline__ void push_q(LaneStateQ &L, const DevScene &S, uint2 *stk, uint32_t slot_mask, bool pred, uint32_t node, float ext) {
    if (pred) {
        if ((L.cs & 0xFFu) == slot_mask) {
            uint2 *ovf = overflow_column(S);
            if (L.novf < S.ovf_depth) { ovf[(size_t)L.novf * 64] = stk[((L.top + 1u) & slot_mask) * 64]; L.novf++; }
            else atomicOr((unsigned int *)S.err, (unsigned int)PT_LIMIT_F32_STACK);
        } else {
            L.cs++;
        }
        stk[L.top * 64] = make_uint2(node, __float_as_uint(ext));
        L.top = (L.top + 1u) & slot_mask;
    }
}

// One level of the descent: a segment [ent, ext] of the ray against the plane `s` on axis `a` (a == 3: no plane, a leaf).
// first: 0 / 1 = the left / right side is visited first; both: the other side follows from `tmid` on.
__device__ __forceinline__ void plane_q(const LaneStateQ &L, uint32_t a, float s, float ent, float ext, uint32_t &first, bool &both, float &tmid) {
    const float oa = sel3(a, L.ox, L.oy, L.oz), ia = sel3(a, L.ix, L.iy, L.iz);
    tmid = (s - oa) * ia;
    const uint32_t nr = oa > s ? 1u : 0u;
    const bool leaf = a == 3u;
    const bool near_only = leaf | !(tmid <= ext) | (tmid < 0.0f);
    const bool far_only = !near_only & (tmid < ent);
    both = !near_only & !far_only;
    first = leaf ? 0u : (far_only ? nr ^ 1u : nr);
}

template <bool ANY, bool COUNT>
__device__ __forceinline__ bool lane_turn(LaneStateQ &L, const DevScene &S, uint2 *stk, uint32_t slot_mask, uint32_t *work, bool active = true) {
    bool done = false;
    const bool at_leaf = L.nlink == 0xFFFFFFFFu;
    if (active & at_leaf & (L.n0 < L.n1)) {
        const bool hit = triangle_unit(L, S, L.n0);
        L.n0 += 48u;
        if (ANY) done = hit;
    }
    if (active & !(at_leaf & (L.n0 < L.n1)) & !done) {
        uint32_t curr;
#if PT_QUAD_MERGED
        // One straight-line block for both kinds of step (a wave nearly always holds lanes of both kinds, and an if / else
        // is walked twice by such a wave): the quad step's numbers are formed by every lane and used by the lanes on a quad
        // record (q), the pop by the lanes on a finished leaf.
        {
            const bool q = !at_leaf;
            const uint32_t link = L.nlink, base = link & ~63u;
            const float s0 = __uint_as_float(L.n0), s1 = __uint_as_float(L.n1), s2 = __uint_as_float(L.n2);
            uint32_t h1; bool both0; float t0;
            plane_q(L, link & 3u, s0, L.ent, L.ext, h1, both0, t0);
            const float e1 = both0 ? t0 : L.ext;
            const bool second = q & both0 & (t0 < L.tmin);
            uint32_t ga, gb; bool botha, bothb; float ta, tb;
            plane_q(L, (link >> (h1 ? 4 : 2)) & 3u, h1 ? s2 : s1, L.ent, e1, ga, botha, ta);
            plane_q(L, (link >> (h1 ? 2 : 4)) & 3u, h1 ? s1 : s2, t0, L.ext, gb, bothb, tb);
            const uint32_t slot_a = base + (h1 * 2u + ga) * 16u, slot_b = base + ((h1 ^ 1u) * 2u + gb) * 16u;
            push_q(L, S, stk, slot_mask, second & bothb & (tb < L.tmin), slot_b ^ 16u, L.ext);
            push_q(L, S, stk, slot_mask, second, slot_b, bothb ? tb : L.ext);
            push_q(L, S, stk, slot_mask, q & botha & (ta < L.tmin), slot_a ^ 16u, e1);
            // leaf done: the next segment starts where this one ended
            const uint32_t cnt = L.cs & 0xFFu;
            const bool finished = at_leaf & (L.ext >= L.tmin);
            const bool cont = at_leaf & !finished;
            const bool pop_ring = cont & (cnt != 0);
            const bool pop_ovf = cont & (cnt == 0) & (L.novf != 0);
            const bool exhausted = cont & (cnt == 0) & (L.novf == 0);
            const uint32_t ptop = (L.top - 1u) & slot_mask;
            uint2 e = stk[ptop * 64];
            if (pop_ovf) { L.novf--; e = overflow_column(S)[(size_t)L.novf * 64]; }
            const bool pop = pop_ring | pop_ovf;
            curr = q ? slot_a : (pop ? e.x : 0u);
            const float ext_q = botha ? ta : e1, ext_l = pop ? __uint_as_float(e.y) : L.ext;
            L.ent = q ? L.ent : L.ext;
            L.ext = q ? ext_q : ext_l;
            L.top = pop_ring ? ptop : L.top;
            L.cs += 256u - (uint32_t)pop_ring;
            done = finished | exhausted;
        }
#else
        if (!at_leaf) {
            // ---- two levels down: up to four grandchildren in front-to-back order; the first is taken, the others pushed ----
            const uint32_t link = L.nlink, base = link & ~63u;
            const float s0 = __uint_as_float(L.n0), s1 = __uint_as_float(L.n1), s2 = __uint_as_float(L.n2);
            uint32_t h1; bool both0; float t0;
            plane_q(L, link & 3u, s0, L.ent, L.ext, h1, both0, t0);
            const float e1 = both0 ? t0 : L.ext;                       // end of the first half's segment
            const bool second = both0 & (t0 < L.tmin);                 // the other half is visited afterwards
            uint32_t ga, gb; bool botha, bothb; float ta, tb;
            plane_q(L, (link >> (h1 ? 4 : 2)) & 3u, h1 ? s2 : s1, L.ent, e1, ga, botha, ta);
            plane_q(L, (link >> (h1 ? 2 : 4)) & 3u, h1 ? s1 : s2, t0, L.ext, gb, bothb, tb);
            const uint32_t slot_a = base + (h1 * 2u + ga) * 16u, slot_b = base + ((h1 ^ 1u) * 2u + gb) * 16u;
            // pushes, farthest first
            push_q(L, S, stk, slot_mask, second & bothb & (tb < L.tmin), slot_b ^ 16u, L.ext);
            push_q(L, S, stk, slot_mask, second, slot_b, bothb ? tb : L.ext);
            push_q(L, S, stk, slot_mask, botha & (ta < L.tmin), slot_a ^ 16u, e1);
            curr = slot_a;
            L.ext = botha ? ta : e1;
            L.cs += 256u;
        } else {
            // ---- leaf done: the next segment starts where this one ended ----
            const uint32_t cnt = L.cs & 0xFFu;
            const bool finished = L.ext >= L.tmin;
            const bool pop_ring = !finished & (cnt != 0);
            const bool pop_ovf = !finished & (cnt == 0) & (L.novf != 0);
            const bool exhausted = !finished & (cnt == 0) & (L.novf == 0);
            const uint32_t ptop = (L.top - 1u) & slot_mask;
            uint2 e = stk[ptop * 64];
            if (pop_ovf) { L.novf--; e = overflow_column(S)[(size_t)L.novf * 64]; }
            curr = (pop_ring | pop_ovf) ? e.x : 0u;
            L.ent = L.ext;
            L.ext = (pop_ring | pop_ovf) ? __uint_as_float(e.y) : L.ext;
            L.top = pop_ring ? ptop : L.top;
            L.cs += 256u - (uint32_t)pop_ring;
            done = finished | exhausted;
        }
#endif
        const bool runaway = L.cs > ((uint32_t)FS_MAX_STEPS << 8);
        if (runaway) atomicOr((unsigned int *)S.err, (unsigned int)PT_LIMIT_F32_STEPS);
        done = done | runaway;
        load_quad(L, S, curr);
        if (COUNT && !done) { work[0]++; work[1] += tris_here(L); }
    }
    return done;
}
