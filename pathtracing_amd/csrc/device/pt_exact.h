// fp64 device functions with the reference's exact semantics: ray/triangle
// test, kd search, surface properties, lights, castRay.  A thread owns a ray
// from search to radiance; the traversal stack is a per-lane ring in LDS that
// spills its oldest entry to a column in device memory when it is full.
#pragma once
#include <float.h>

#include "pt_scene_dev.h"
#include "pt_vec.h"

#define PT_STACK_MAX 64   // >= tree depth, checked when the scene is uploaded

// Stack of search() (util/kdtree.cpp:203-208,223-250): entries (node, ent, ext).  The newest PT_RING64 entries of a
// lane live in LDS, laid out [entry][lane] per field (8-byte accesses at stride 8: no bank conflicts whatever the
// lanes' depths); when the ring is full the oldest entry moves to the lane's column of the block's overflow region
// in device memory and is popped from there once the ring has run empty.  The stack never holds more than one
// entry per tree level, and the overflow region is sized for the tree's depth at upload, so nothing is ever dropped.
#define PT_RING64 8
#define PT_RING64_LDS_BYTES (PT_RING64 * 64 * 20)
struct StackD {
    double *ent, *ext;        // LDS, this lane's column; entry k at [k * 64]
    uint32_t *node;
    double *ovf;              // device memory, this lane's column; field f of entry e at [(3 * e + f) * 64]
    uint32_t top, cnt, novf, cap;
    unsigned long long *err;
    PT_DEV void clear() { top = 0; cnt = 0; novf = 0; }
    PT_DEV uint32_t size() const { return cnt + novf; }
    PT_DEV void push(int n, double e, double x) {
        if (cnt == PT_RING64) {                      // full: `top` is also the oldest entry's slot
            if (novf < cap) {
                double *o = ovf + (size_t)novf * 192;
                o[0] = __longlong_as_double((long long)node[top * 64]); o[64] = ent[top * 64]; o[128] = ext[top * 64];
                novf++;
            } else {
                atomicOr(err, (unsigned long long)PT_LIMIT_F64_STACK);     // cannot happen for a tree of the uploaded depth
            }
            cnt--;
        }
        node[top * 64] = (uint32_t)n; ent[top * 64] = e; ext[top * 64] = x;
        top = (top + 1u) & (PT_RING64 - 1u);
        cnt++;
    }
    PT_DEV void pop(int &n, double &e, double &x) {  // size() > 0
        if (cnt == 0) {
            novf--;
            const double *o = ovf + (size_t)novf * 192;
            n = (int)__double_as_longlong(o[0]); e = o[64]; x = o[128];
            return;
        }
        top = (top - 1u) & (PT_RING64 - 1u);
        cnt--;
        n = (int)node[top * 64]; e = ent[top * 64]; x = ext[top * 64];
    }
    // smallest `ent` among the entries left on the stack
    PT_DEV double min_ent() const {
        double m = DBL_MAX;
        for (uint32_t i = 0; i < cnt; i++) { const double e = ent[((top - 1u - i) & (PT_RING64 - 1u)) * 64]; if (e < m) m = e; }
        for (uint32_t i = 0; i < novf; i++) { const double e = ovf[(size_t)i * 192 + 64]; if (e < m) m = e; }
        return m;
    }
};
// The calling wave's stack: `lds` is the block's PT_RING64_LDS_BYTES of shared memory (one wave per block).
PT_DEV StackD make_stack(const DevScene &S, double *lds) {
    const uint32_t lane = threadIdx.x & 63u;
    StackD st;
    st.ent = lds + lane; st.ext = lds + PT_RING64 * 64 + lane;
    st.node = (uint32_t *)(lds + 2 * PT_RING64 * 64) + lane;
    st.ovf = S.ovf64 + (size_t)blockIdx.x * S.ovf64_entries * 192 + lane;
    st.cap = S.ovf64_entries; st.err = S.err;
    st.clear();
    return st;
}

struct HitD { int32_t tri; double tmin, u, v; };

struct RayCounts { uint32_t n[RC_N]; };

// Loads through the scalar unit, for addresses that are the same in every active lane of the wave (constant
// address space: the scene is read-only for the kernels).  The rays of a wave are neighbours -- the nine sub-rays
// of a bundle, the pixels of a tile -- and are in the same node testing the same triangle most of the time; the fp64
// search is bound by per-lane load requests (72 bytes of triangle = 5 requests per lane and test), and a scalar load
// is one request per wave.
typedef const double __attribute__((address_space(4))) *PtUniformD;
typedef const int32_t __attribute__((address_space(4))) *PtUniformI;
PT_DEV D3 uload3(const double *p) {
    PtUniformD q = (PtUniformD)(uintptr_t)p;
    return d3(q[0], q[1], q[2]);
}

// Moeller-Trumbore as util/polygonMesh.cpp:54-74 (two-sided, three divides).
// v0, e1 = v1 - v0, e2 = v2 - v0 are precomputed in fp64 on the host with the
// same subtractions the reference performs per call.
PT_DEV bool tri_intersect(const D3 &v0, const D3 &ba, const D3 &ca, const RayD &ray, double &t, double &bu, double &bv) {
    D3 P = dcross(ray.dir, ca);
    double det = ddot(P, ba);
    D3 T = dsub(ray.src, v0);
    double u = ddot(P, T) / det;
    if (u < 0.0 || u > 1.0) return false;
    D3 Q = dcross(T, ba);
    double v = ddot(Q, ray.dir) / det;
    if (v < 0.0 || u + v > 1.0) return false;
    t = ddot(Q, ca) / det;
    bu = u; bv = v;
    return true;
}

// Box::intersect, util/kdtree.cpp:30-49 (ent/ext come from the x and y slabs only).  The reciprocal direction and the
// sign bits of the Ray (util/polygonMesh.cpp:14-19) are only ever read here, so they are formed here -- the same
// divisions -- instead of being carried through the search in registers.
PT_DEV bool box_intersect(const double lo[3], const double hi[3], const RayD &r, double &ent, double &ext) {
    const double ix = 1.0 / r.dir.x, iy = 1.0 / r.dir.y;
    const bool sx = __double_as_longlong(r.dir.x) < 0, sy = __double_as_longlong(r.dir.y) < 0;
    double tmin = ((sx ? hi[0] : lo[0]) - r.src.x) * ix;
    double tmax = ((sx ? lo[0] : hi[0]) - r.src.x) * ix;
    double ymin = ((sy ? hi[1] : lo[1]) - r.src.y) * iy;
    double ymax = ((sy ? lo[1] : hi[1]) - r.src.y) * iy;
    if ((tmin > ymax) || (ymin > tmax)) return false;
    if (ymin > tmin) tmin = ymin;
    if (ymax < tmax) tmax = ymax;
    const double iz = 1.0 / r.dir.z;
    const bool sz = __double_as_longlong(r.dir.z) < 0;
    double zmin = ((sz ? hi[2] : lo[2]) - r.src.z) * iz;
    double zmax = ((sz ? lo[2] : hi[2]) - r.src.z) * iz;
    if ((tmin > zmax) || (zmin > tmax)) return false;
    ent = tmin; ext = tmax;
    return true;
}

// KDNode::searchNode, util/kdtree.cpp:180-197: a hit only counts when the hit
// point lies inside the node's own box (inclusive on all three axes).
// The node's list, fetched through the scalar unit, for the lanes that are all in node `node0`.
PT_DEV bool search_node_uniform(const DevScene &S, int node0, int first0, int count0, const RayD &ray, HitD &h) {
    bool found = false;
    const double *tri = S.refTri + (size_t)first0 * 9;
    for (int i = 0; i < count0; i++, tri += 9) {
        double t, bu, bv;
        if (tri_intersect(uload3(tri), uload3(tri + 3), uload3(tri + 6), ray, t, bu, bv) && t > 0 && t < h.tmin) {
            D3 hit = dadd(ray.src, dscale(t, ray.dir));
            const double *b = S.nodeBox + (size_t)node0 * 6;
            const D3 lo = uload3(b), hi = uload3(b + 3);
            if (!(hit.x >= lo.x && hit.y >= lo.y && hit.z >= lo.z && hit.x <= hi.x && hit.y <= hi.y && hit.z <= hi.z))
                continue;
            found = true;
            h.tmin = t; h.u = bu; h.v = bv;
            h.tri = ((PtUniformI)(uintptr_t)S.refId)[first0 + i];
        }
    }
    return found;
}

// The rays of a wave are neighbours (sub-rays of a bundle, pixels of a tile) and are often all in one node: then the
// node's list comes through the scalar unit, one request per wave; otherwise each lane loads for itself.  (Taking the
// distinct nodes of a wave one after the other through the scalar unit instead was slower on every scene: 1.86 against
// 1.60 ms for the bundles of an atrium view.)
PT_DEV bool search_node(const DevScene &S, int node, const NodeA &na, const RayD &ray, HitD &h) {
    const int node0 = __builtin_amdgcn_readfirstlane(node);
    if (__all(node == node0)) {
        // every active lane is in this node: its list comes through the scalar unit, once for the wave
        const int first0 = __builtin_amdgcn_readfirstlane(na.first), count0 = __builtin_amdgcn_readfirstlane(na.count);
        return search_node_uniform(S, node0, first0, count0, ray, h);
    }
    bool found = false;
    const double *tri = S.refTri + (size_t)na.first * 9;
    for (int i = 0; i < na.count; i++, tri += 9) {
        double t, bu, bv;
        if (tri_intersect(dload(tri), dload(tri + 3), dload(tri + 6), ray, t, bu, bv) && t > 0 && t < h.tmin) {
            D3 hit = dadd(ray.src, dscale(t, ray.dir));
            const double *b = S.nodeBox + (size_t)node * 6;
            if (!(hit.x >= b[0] && hit.y >= b[1] && hit.z >= b[2] && hit.x <= b[3] && hit.y <= b[4] && hit.z <= b[5]))
                continue;
            found = true;
            h.tmin = t; h.u = bu; h.v = bv;
            h.tri = S.refId[na.first + i];
        }
    }
    return found;
}

// A node of the fp64 tree; through the scalar unit when every active lane wants the same one.
PT_DEV NodeA fetch_node(const DevScene &S, int curr) {
    const int curr0 = __builtin_amdgcn_readfirstlane(curr);
    if (__all(curr == curr0)) {
        typedef const NodeA __attribute__((address_space(4))) *PtUniformNode;
        PtUniformNode q = (PtUniformNode)(uintptr_t)S.nodeA + curr0;
        NodeA n;
        n.split = q->split; n.axis = q->axis; n.left = q->left; n.first = q->first; n.count = q->count;
        return n;
    }
    return S.nodeA[curr];
}

// What a search notes for the sake of a later search of the SAME ray bounded by the hit it returned
// (mainOMP.cpp:60 searches the camera ray again, once per light):
//   far_max   the largest tmid among the steps that took the far child only (util/kdtree.cpp:242-243)
//   stack_min the smallest entry distance left on the stack when the search ended
// The later search makes the decisions of this one again under a bound that is no larger.  A smaller bound can only
// (a) leave far children unpushed, (b) stop at a popped entry sooner -- both make it visit fewer of the same nodes --
// (c) turn a far-only step into a near step, if that step's tmid lies beyond the bound, or (d) go on to a stack entry
// that this search never reached because it returned first, if that entry starts before the bound.  While it stays on
// this search's nodes every triangle it meets was tested here and either failed for a reason that does not involve
// the bound or had t >= the bound this search ended with (the returned hit itself: t == bound), so `t < tmin` fails
// again, bit for bit.  Hence: unless far_max > t_hit (c) or stack_min < t_hit (d), the later search finds nothing and
// leaves the hit record alone, and need not be run.  tests/research_check.c checks this rule on the CPU.
struct SearchNote {
    double far_max, stack_min;
    PT_DEV bool may_find_closer(double t_hit) const { return far_max > t_hit || stack_min < t_hit; }
};

// search(), util/kdtree.cpp:216-257.  h.tmin is the in/out bound.  NOTED: also fills *note.
template <bool NOTED>
PT_DEV bool search_ref_t(const DevScene &S, const RayD &ray, HitD &h, SearchNote *note, StackD &st) {
    double ent, ext;
    if (NOTED) { note->far_max = -DBL_MAX; note->stack_min = DBL_MAX; }
    if (!box_intersect(S.root_lo, S.root_hi, ray, ent, ext)) return false;
    st.clear();
    int curr = 0;                                   // the root entry is taken at once instead of pushed and popped
    bool found = false, at_leaf = false;
    for (;;) {
        if (ent >= h.tmin) break;                   // util/kdtree.cpp:229-231: ends the whole search
        NodeA na = fetch_node(S, curr);
        while (na.axis >= 0) {
            if (na.count > 0) found |= search_node(S, curr, na, ray, h);
            double s = na.split;
            int a = na.axis;
            double da = dget(ray.dir, a), sa = dget(ray.src, a);
            double tdir = da == 0 ? 0.0000001 : da;
            double tmid = (s - sa) / tdir;
            int nr = na.left, fr = na.left + 1;
            if (sa > s) { int x = nr; nr = fr; fr = x; }
            if (tmid >= ext || tmid < 0)
                curr = nr;
            else if (tmid <= ent && tmid <= h.tmin) {
                if (NOTED && tmid > note->far_max) note->far_max = tmid;
                curr = fr;
            } else {
                if (tmid <= h.tmin) st.push(fr, tmid, ext);
                curr = nr;
                ext = tmid;
            }
            na = fetch_node(S, curr);
        }
        if (na.count > 0 && search_node(S, curr, na, ray, h)) { at_leaf = true; break; }
        if (st.size() == 0) break;
        st.pop(curr, ent, ext);
    }
    if (NOTED) note->stack_min = st.min_ent();
    return at_leaf || found;
}
PT_DEV bool search_ref(const DevScene &S, const RayD &ray, HitD &h, StackD &st) { return search_ref_t<false>(S, ray, h, nullptr, st); }

// ---- shading ------------------------------------------------------------------

PT_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv::Mat::at<Vec3b>(Point(x, y)) is row y, column x; indices are clamped into
// the image (the reference does not check them).
PT_DEV const uint8_t *texel(const DevScene &S, int tex, int px, int py) {
    int rows = S.tex_rows[tex], cols = S.tex_cols[tex];
    int row = clampi(py, 0, rows - 1), col = clampi(px, 0, cols - 1);
    return S.texels + S.tex_off[tex] + ((size_t)row * cols + col) * 3;
}

// PolygonMesh::surfaceProperties, util/polygonMesh.cpp:81-118, in two parts: the interpolated normal, which the light
// loop needs, and the texture colour and mirror flag, which are only needed once the lights are done (nothing in
// between depends on them, so fetching them later changes no value and keeps six registers free during the searches
// of the light loop).
PT_DEV D3 surface_normal(const DevScene &S, int tri, double bu, double bv) {
    const int32_t *N = S.tri_ni + (size_t)tri * 3;
    double w = 1.0 - bu - bv;
    D3 n0 = dload(S.norm + (size_t)N[0] * 3), n1 = dload(S.norm + (size_t)N[1] * 3), n2 = dload(S.norm + (size_t)N[2] * 3);
    return dunit(dadd(dadd(dscale(w, n0), dscale(bu, n1)), dscale(bv, n2)));
}
PT_DEV void surface_texture(const DevScene &S, int tri, double bu, double bv, D3 &txt, int &reft) {
    double w = 1.0 - bu - bv;
    int mn = S.tri_tm[tri];
    reft = S.tri_reft[tri];
    if (mn == -99999) { txt = d3(1, 1, 1); return; }
    const int32_t *T = S.tri_ti + (size_t)tri * 3;
    if (mn < 0) {
        mn = -(mn + 1);
        int rows = S.tex_rows[mn];
        const double *a = S.uv + (size_t)T[0] * 2, *b = S.uv + (size_t)T[1] * 2, *c = S.uv + (size_t)T[2] * 2;
        int r = (int)((a[0] * w + b[0] * bu + c[0] * bv) * rows);
        int cc = (int)((a[1] * w + b[1] * bu + c[1] * bv) * rows);
        const uint8_t *p = texel(S, mn, r, cc);
        txt = ddivs(d3((double)p[2], (double)p[1], (double)p[0]), 256.0);
    } else {
        int rows = S.tex_rows[mn], cols = S.tex_cols[mn];
        D3 col[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            int r = 0, c = 0;
            if (T[i] != -1) {
                r = (int)(S.uv[(size_t)T[i] * 2] * rows);
                c = (int)(S.uv[(size_t)T[i] * 2 + 1] * cols);
            }
            int px = r < 0 ? 0 : r; if (px > rows) px = rows;
            int py = c < 0 ? 0 : c; if (py > cols) py = cols;
            const uint8_t *p = texel(S, mn, px, py);
            col[i] = d3((double)p[2], (double)p[1], (double)p[0]);
        }
        txt = ddivs(dadd(dadd(dscale(w, col[0]), dscale(bu, col[1])), dscale(bv, col[2])), 256.0);
    }
}

// Light::illuminate, util/lighting.cpp:24-36, in two parts: direction + distance, and the radiance, which depends on the
// light and the distance only.
PT_DEV void illuminate_dir(const PtLight &L, D3 hit, D3 &ldir, double &dis) {
    D3 pos = d3(L.pos[0], L.pos[1], L.pos[2]);
    ldir = d3(0, 0, 0); dis = 0;
    if (L.type == 'd') {
        dis = (double)FLT_MAX;
        ldir = dunit(pos);
    } else if (L.type == 'p') {
        ldir = dsub(pos, hit);
        dis = dlen(ldir);
        ldir = dunit(ldir);
    }
}
PT_DEV D3 illuminate_shade(const PtLight &L, double dis) {
    D3 color = d3(L.color[0], L.color[1], L.color[2]);
    if (L.type == 'd') return dscale(L.mag, color);
    if (L.type == 'p') return ddivs(dscale(L.mag, color), 4 * dis * 3.14159265358979323846);
    return d3(0, 0, 0);
}

// createCoordSystem, util/lighting.cpp:42-48.
PT_DEV void coord_system(D3 N, D3 &Nt, D3 &Nb) {
    if (fabs(N.x) > fabs(N.y))
        Nt = ddivs(d3(N.z, 0, -N.x), sqrt(N.x * N.x + N.z * N.z));
    else
        Nt = ddivs(d3(0, -N.z, N.y), sqrt(N.y * N.y + N.z * N.z));
    Nb = dcross(N, Nt);
}

struct GBufD { D3 clr, nrm, hit; };

// Primary-hit record of the depth-0 call.
struct PrimRec { int32_t found, tri; double t, u, v; };

// What castRay computes of a hit before its mirror bounce: search, surface properties, hit point, the light loop.
struct CastHit { HitD h; D3 nrm, txt, hit, direct, alb; int reft; };

// castRay after its search found the hit c.h: surface properties, hit point, the light loop
// (main.cpp:51-66 for MODEL == PT_MODEL_MAIN, mainOMP.cpp:47-63 for PT_MODEL_OMP).
// retrace_clear (mainOMP model): the search of mainOMP.cpp:60 is known to find nothing (SearchNote).
// SEARCH = false (mainOMP model only): the caller vouches for retrace_clear, no search is compiled in and `st` is not used.
template <int MODEL, bool SEARCH = true>
PT_DEV void cast_lights(const DevScene &S, const RayD &ray, StackD &st, RayCounts &rc, CastHit &c, bool retrace_clear) {
    static_assert(SEARCH || MODEL == PT_MODEL_OMP, "only the mainOMP model can do without its searches");
    HitD &h = c.h;
    c.nrm = surface_normal(S, h.tri, h.u, h.v);
    c.hit = dadd(dadd(ray.src, dscale(h.tmin, ray.dir)), dscale(0.0001, c.nrm));
    D3 direct = d3(0, 0, 0);
    for (int k = 0; k < S.n_light; k++) {
        // Light::illuminate (util/lighting.cpp:24-36) around the visibility search: direction and distance before it,
        // the radiance after it (a function of the light and the distance only), so that it is not held in registers
        // across the search.
        D3 ldir; double dis;
        illuminate_dir(S.lights[k], c.hit, ldir, dis);
        bool vis;
        if (MODEL == PT_MODEL_MAIN) {                        // main.cpp:62-63: a real shadow ray
            RayD lray = make_ray(c.hit, ldir);
            HitD sh; sh.tmin = dis; sh.tri = -1; sh.u = 0; sh.v = 0;
            vis = !search_ref(S, lray, sh, st);
            rc.n[RC_SHADOW]++;
        } else {                                             // mainOMP.cpp:60: re-traces `ray` bounded by its own hit
            // known to find nothing by the note of the first search, or because the last one found nothing and so left
            // h as it was, which makes the next one the same search
            if (SEARCH && !retrace_clear) retrace_clear = !search_ref(S, ray, h, st);
            vis = SEARCH ? retrace_clear : true;
            rc.n[RC_RETRACE]++;
        }
        double cosv = ddot(ldir, c.nrm);
        double mx = (0.0 < cosv) ? cosv : 0.0;
        direct = dadd(direct, dscale((vis ? 1.0 : 0.0) * mx, illuminate_shade(S.lights[k], dis)));
    }
    c.direct = direct;
    surface_texture(S, h.tri, h.u, h.v, c.txt, c.reft);
    const double *ap = S.albedo + (size_t)S.tri_mesh[h.tri] * 3;
    c.alb = d3(ap[0], ap[1], ap[2]);
}

// The first part of castRay, main.cpp:41-66 (MODEL == PT_MODEL_MAIN) / mainOMP.cpp:37-63 (PT_MODEL_OMP):
// false on a miss (c.h then holds the search's record).
template <int MODEL>
PT_DEV bool cast_hit(const DevScene &S, const RayD &ray, StackD &st, RayCounts &rc, CastHit &c) {
    HitD &h = c.h;
    h.tmin = (double)FLT_MAX; h.tri = -1; h.u = 0; h.v = 0;
    SearchNote note;
    bool ok = MODEL == PT_MODEL_OMP ? search_ref_t<true>(S, ray, h, &note, st) : search_ref(S, ray, h, st);
    if (!ok) return false;
    cast_lights<MODEL>(S, ray, st, rc, c, MODEL == PT_MODEL_OMP && !note.may_find_closer(h.tmin));
    return true;
}

// The value castRay returns for a hit whose mirror bounce (if any) returned `ref`: main.cpp:77 / mainOMP.cpp:74-76.
template <int MODEL>
PT_DEV D3 cast_value(const CastHit &c, int depth, D3 ref) {
    if (MODEL == PT_MODEL_OMP && depth == -1)
        return ddivs(dmul(dmul(c.txt, c.alb), c.direct), 3.14159265358979323846);        // mainOMP.cpp:74-75
    D3 a = ddivs(dmul(dmul(dscale(0.6, c.txt), c.alb), c.direct), 3.14159265358979323846); // main.cpp:77, mainOMP.cpp:76
    return dadd(a, dscale(0.4, ref));
}

// castRay at depth != 0 (a GI sample ray, or the mirror bounce of a primary hit): no G-buffer, no further bounce.
template <int MODEL>
PT_DEV D3 cast_secondary(const DevScene &S, const RayD &ray, int depth, StackD &st, RayCounts &rc) {
    CastHit c;
    if (!cast_hit<MODEL>(S, ray, st, rc, c)) return d3(0, 0, 0);
    return cast_value<MODEL>(c, depth, d3(0, 0, 0));
}

// Camera ray of main.cpp:119-121 / mainOMP.cpp:121-123; sa, ca = sin, cos of the
// view angle, computed once on the host with libm as the reference does.
PT_DEV RayD camera_ray(D3 src, double zoom, double sa, double ca, int rows, int i, int j, double x, double y) {
    D3 pxl = d3(0.5 - (double)(j + x) / rows, 0.5 - (double)(i + y) / rows, zoom);
    D3 snk = d3(src.x + sa * pxl.x + ca * pxl.z, src.y + pxl.y, src.z - ca * pxl.x + sa * pxl.z);
    return make_ray(src, dunit(dsub(snk, src)));
}
