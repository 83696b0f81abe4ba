// search() of util/kdtree.cpp:216-257 for the rays of one wave that leave the same point (camera rays), walked as ONE
// packet.  The wave stands in one node at a time: the node and its triangle list come through the scalar unit, a
// 64-bit mask says which rays are in the node, and rays that part ways with the wave wait in a stack entry until the
// wave comes back for them.  Every ray makes exactly the decisions of the reference's search on its own numbers and
// meets its nodes and triangles in the order of its own search (rays that share their origin agree on which child of
// a node is the near one), so hit records are those of search_ref_t bit for bit; only the order in which DIFFERENT
// rays advance changes.  tests/packet_check.c states the same walk in plain C and checks it against the oracle.
//
// Before the rays test the triangles of a node, the packet as a whole is tested against them, one triangle per lane
// (packet_cull): a triangle that provably none of the packet's rays can pass is left out.
#pragma once
#include "pt_exact.h"

#define PT_PKT_LDS 8                       // stack entries whose per-lane numbers live in LDS; deeper ones in device memory
#define PT_PKT_CULL_MIN 3                  // lists shorter than this are tested without the packet test

PT_DEV double wave_min_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
PT_DEV double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
PT_DEV double uniform_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Bounds of a packet's directions in the camera's frame (Z forward, X right, Y up: main.cpp:119-120): every direction
// d of the packet is  c Z + al X + be Y + r  with c in [c0, c1], al in [a0, a1], be in [b0, b1], |r|_2 <= rr.
// The remainder r is measured per ray, so the bounds hold for any frame, orthonormal or not; in the camera's own a
// packet of neighbouring pixels is an axis-aligned box.
struct PacketBounds { D3 Z, X, Y; double c0, c1, a0, a1, b0, b1, rr; };

// Every lane of the wave must call this; lanes that are not `live` do not count.
PT_DEV PacketBounds packet_bounds(double sa, double ca, const D3 &dir, bool live) {
    PacketBounds B;
    B.Z = d3(ca, 0, sa); B.X = d3(sa, 0, -ca); B.Y = d3(0, 1, 0);
    const double c = ddot(dir, B.Z), al = ddot(dir, B.X), be = ddot(dir, B.Y);
    const D3 r = dsub(dir, dadd(dadd(dscale(c, B.Z), dscale(al, B.X)), dscale(be, B.Y)));
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    B.c0 = uniform_d(wave_min_d(live ? c : inf)); B.c1 = uniform_d(wave_max_d(live ? c : -inf));
    B.a0 = uniform_d(wave_min_d(live ? al : inf)); B.a1 = uniform_d(wave_max_d(live ? al : -inf));
    B.b0 = uniform_d(wave_min_d(live ? be : inf)); B.b1 = uniform_d(wave_max_d(live ? be : -inf));
    B.rr = 2.0 * uniform_d(wave_max_d(live ? dlen(r) : 0.0)) + 1e-15;
    return B;
}

PT_DEV double dl1(const D3 &a) { return fabs(a.x) + fabs(a.y) + fabs(a.z); }
// Range of the linear form d . A over the packet's directions.
PT_DEV void form_range(const PacketBounds &B, const D3 &A, double &lo, double &hi) {
    const double f0 = ddot(B.Z, A), fa = ddot(B.X, A), fb = ddot(B.Y, A);
    const double x0 = B.c0 * f0, x1 = B.c1 * f0, y0 = B.a0 * fa, y1 = B.a1 * fa, z0 = B.b0 * fb, z1 = B.b1 * fb;
    const double r = B.rr * dl1(A);
    lo = fmin(x0, x1) + fmin(y0, y1) + fmin(z0, z1) - r;
    hi = fmax(x0, x1) + fmax(y0, y1) + fmax(z0, z1) + r;
}

// true: no ray that leaves `o` with a direction inside B can pass the reference's tests on the triangle (v0, e1, e2),
// so the reference rejects it for every ray of the packet and leaving it out changes nothing.
//
// With T = o - v0 the reference computes, per ray of direction d (util/polygonMesh.cpp:56-72),
//     det = (d x e2) . e1 = d . (e2 x e1)      u = un / det,  un = (d x e2) . T  = d . (e2 x T)
//     v = vn / det,  vn = (T x e1) . d         t = tn / det,  tn = (T x e1) . e2
// and rejects when u < 0, u > 1, v < 0, u + v > 1 or (in searchNode, util/kdtree.cpp:186) t <= 0.  det, un, vn and
// w = un + vn - det are linear in d, so their ranges over the packet follow from the packet's bounds; tn does not
// depend on d.  The triangle is left out when the sign of det is the same for the whole packet and, with that sign,
// un < 0, or vn < 0, or w > 0, or tn < 0 holds for the whole packet -- each by a margin:
// every quantity is a sum of at most six products x_i y_j z_k of three vectors, so M = |x|_1 |y|_1 |z|_1 bounds the sum of
// their magnitudes, and any fp64 evaluation of it (the reference's, this one's, the range arithmetic on top) is within
// 40 * 2^-53 * M < 1e-14 M of the exact value.  The margins are 1e-12 M, and M is required to lie in [1e-100, 1e100], so a
// quantity that clears its margin has the claimed sign as the reference computes it, and the quotients the reference
// forms from such quantities are at least 1e-12 in magnitude away from 0 (resp. from 1 for u + v): no rounding, underflow or
// NaN can turn the comparison.  Triangles outside that range of magnitudes, and packets for which det changes sign, are kept.
PT_DEV bool packet_cull(const PacketBounds &B, const D3 &o, const D3 &v0, const D3 &e1, const D3 &e2) {
    const double K = 1e-12;
    const D3 T = dsub(o, v0);
    const D3 Nn = dcross(e2, e1), A = dcross(e2, T), Bv = dcross(T, e1);
    const D3 Cw = dsub(dadd(A, Bv), Nn);
    const double tn = ddot(Bv, e2);
    const double n1 = dl1(e1), n2 = dl1(e2), nT = dl1(T);
    const double Md = 2.0 * n1 * n2, Mu = 2.0 * n2 * nT, Mv = 2.0 * nT * n1, Mt = nT * n1 * n2;
    if (!(Md >= 1e-100 && Md <= 1e100 && Mu >= 1e-100 && Mu <= 1e100 && Mv >= 1e-100 && Mv <= 1e100 && Mt <= 1e100)) return false;
    const double md = K * Md, mu = K * Mu, mv = K * Mv, mw = K * (Md + Mu + Mv), mt = K * Mt;
    double dlo, dhi, ulo, uhi, vlo, vhi, wlo, whi;
    form_range(B, Nn, dlo, dhi);
    const bool pos = dlo > md, neg = dhi < -md;
    form_range(B, A, ulo, uhi);
    form_range(B, Bv, vlo, vhi);
    form_range(B, Cw, wlo, whi);
    const bool out_pos = uhi < -mu || vhi < -mv || wlo > mw || tn < -mt;
    const bool out_neg = ulo > mu || vlo > mv || whi < -mw || tn > mt;
    return (pos && out_pos) || (neg && out_neg);
}

// The packet's stack.  An entry is a node that some rays still have to visit: `push` are the rays that pushed it
// (util/kdtree.cpp:246-247; their entry and exit distances are kept per lane), `defer` the rays that took the far child
// only (:243-244) while others took the near one first -- they wait here with their numbers untouched.  The node and the
// two masks of entry k sit in lane k of five registers; the per-lane distances of the first PT_PKT_LDS entries in LDS
// ([entry][lane], conflict-free), those of deeper entries in the block's overflow region in device memory.
struct PacketStack {
    double *lds;              // this lane's column: ent of entry k at [k * 64], ext at [(PT_PKT_LDS + k) * 64]
    double *ovf;              // this lane's column: ent of entry k >= PT_PKT_LDS at [(k - PT_PKT_LDS) * 128], ext 64 further
    int v_node, v_push_lo, v_push_hi, v_defer_lo, v_defer_hi;
    uint32_t sp;
    PT_DEV void push(int node, unsigned long long pushm, unsigned long long deferm, bool mine, double ent, double ext) {
        if (mine) {
            if (sp < PT_PKT_LDS) { lds[sp * 64] = ent; lds[(PT_PKT_LDS + sp) * 64] = ext; }
            else { ovf[(size_t)(sp - PT_PKT_LDS) * 128] = ent; ovf[(size_t)(sp - PT_PKT_LDS) * 128 + 64] = ext; }
        }
        if ((threadIdx.x & 63u) == sp) {
            v_node = node;
            v_push_lo = (int)(uint32_t)pushm; v_push_hi = (int)(uint32_t)(pushm >> 32);
            v_defer_lo = (int)(uint32_t)deferm; v_defer_hi = (int)(uint32_t)(deferm >> 32);
        }
        sp++;
    }
    PT_DEV void pop(int &node, unsigned long long &pushm, unsigned long long &deferm) {
        sp--;
        node = __builtin_amdgcn_readlane(v_node, sp);
        pushm = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(v_push_lo, sp) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(v_push_hi, sp) << 32;
        deferm = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(v_defer_lo, sp) |
                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(v_defer_hi, sp) << 32;
    }
    // distances of the entry just popped (entry `sp`), for a lane of its push mask
    PT_DEV void popped(double &ent, double &ext) const {
        if (sp < PT_PKT_LDS) { ent = lds[sp * 64]; ext = lds[(PT_PKT_LDS + sp) * 64]; }
        else { ent = ovf[(size_t)(sp - PT_PKT_LDS) * 128]; ext = ovf[(size_t)(sp - PT_PKT_LDS) * 128 + 64]; }
    }
};
// `lds`: the block's shared memory (one wave per block; 2 * PT_PKT_LDS * 64 doubles are used).
PT_DEV PacketStack make_packet_stack(const DevScene &S, double *lds) {
    const uint32_t lane = threadIdx.x & 63u;
    PacketStack st;
    st.lds = lds + lane;
    st.ovf = S.ovf64 + (size_t)blockIdx.x * S.ovf64_entries * 192 + lane;
    st.v_node = st.v_push_lo = st.v_push_hi = st.v_defer_lo = st.v_defer_hi = 0;
    st.sp = 0;
    return st;
}

// KDNode::searchNode (util/kdtree.cpp:180-197) for the rays `in` of the packet on the list of node `node0`.
PT_DEV bool packet_node(const DevScene &S, int node0, int first0, int count0, const RayD &ray, bool in, const PacketBounds &B, HitD &h) {
    const uint32_t lane = threadIdx.x & 63u;
    bool improved = false;
    for (int base = 0; base < count0; base += 64) {
        const int n = count0 - base < 64 ? count0 - base : 64;
        unsigned long long surv = n == 64 ? ~0ull : (1ull << n) - 1ull;
        if (count0 >= PT_PKT_CULL_MIN) {
            bool keep = false;
            if ((int)lane < n) {
                const double *tri = S.refTri + (size_t)(first0 + base + (int)lane) * 9;
                keep = !packet_cull(B, ray.src, dload(tri), dload(tri + 3), dload(tri + 6));
            }
            surv = __ballot(keep);
        }
        while (surv) {
            const int k = first0 + base + (int)__builtin_ctzll(surv);
            surv &= surv - 1ull;
            if (in) {
                const double *tri = S.refTri + (size_t)k * 9;
                double t, bu, bv;
                if (tri_intersect(uload3(tri), uload3(tri + 3), uload3(tri + 6), ray, t, bu, bv) && t > 0 && t < h.tmin) {
                    const D3 hit = dadd(ray.src, dscale(t, ray.dir));
                    const double *b = S.nodeBox + (size_t)node0 * 6;
                    const D3 lo = uload3(b), hi = uload3(b + 3);
                    if (hit.x >= lo.x && hit.y >= lo.y && hit.z >= lo.z && hit.x <= hi.x && hit.y <= hi.y && hit.z <= hi.z) {
                        improved = true;
                        h.tmin = t; h.u = bu; h.v = bv;
                        h.tri = ((PtUniformI)(uintptr_t)S.refId)[k];
                    }
                }
            }
        }
    }
    return improved;
}

// The search of every `live` ray of the wave: src (the same for all) and `dir`; h.tmin is the in/out bound as in
// search_ref_t, whose result and notes this returns.  Every lane of the wave must call it.
template <bool NOTED>
PT_DEV bool packet_search(const DevScene &S, const RayD &ray, bool live, const PacketBounds &B, HitD &h, SearchNote *note, double *lds) {
    const uint32_t lane = threadIdx.x & 63u;
    PacketStack st = make_packet_stack(S, lds);
    double ent = 0, ext = 0;
    if (NOTED) { note->far_max = -DBL_MAX; note->stack_min = DBL_MAX; }
    bool in = live && box_intersect(S.root_lo, S.root_hi, ray, ent, ext) && !(ent >= h.tmin);
    bool finished = !in;                            // this ray's search has ended (or never began)
    bool found = false, at_leaf = false;
    unsigned long long act = __ballot(in);
    int curr = 0;
    for (;;) {
        if (!act) {
            if (!NOTED && !__ballot(!finished)) break;
            while (st.sp > 0 && !act) {
                unsigned long long pushm, deferm;
                st.pop(curr, pushm, deferm);
                in = (deferm >> lane) & 1ull;
                if ((pushm >> lane) & 1ull) {
                    double e, x;
                    st.popped(e, x);
                    if (finished) { if (NOTED && e < note->stack_min) note->stack_min = e; }
                    else if (e >= h.tmin) finished = true;          // util/kdtree.cpp:229-231: ends this ray's search
                    else { ent = e; ext = x; in = true; }
                }
                act = __ballot(in);
            }
            if (!act) break;
        }
        typedef const NodeA __attribute__((address_space(4))) *PtUniformNode;
        const PtUniformNode q = (PtUniformNode)(uintptr_t)S.nodeA + curr;
        const double split = q->split;
        const int axis = q->axis, left = q->left, first = q->first, count = q->count;
        bool improved = false;
        if (count > 0) improved = packet_node(S, curr, first, count, ray, in, B, h);
        if (axis < 0) {                              // a leaf: the rays that improved their hit here are done (:252-253)
            if (in && improved) { at_leaf = true; finished = true; }
            act = 0;
            continue;
        }
        found |= in && improved;
        const double da = axis == 0 ? ray.dir.x : (axis == 1 ? ray.dir.y : ray.dir.z);
        const double sa = axis == 0 ? ray.src.x : (axis == 1 ? ray.src.y : ray.src.z);
        const double tdir = da == 0 ? 0.0000001 : da;
        const double tmid = (split - sa) / tdir;
        const bool near_only = tmid >= ext || tmid < 0;
        const bool far_only = !near_only && tmid <= ent && tmid <= h.tmin;
        const bool push = !near_only && !far_only && tmid <= h.tmin;
        const unsigned long long nearm = __ballot(in && !far_only), farm = __ballot(in && far_only), pushm = __ballot(in && push);
        if (NOTED && in && far_only && tmid > note->far_max) note->far_max = tmid;
        const bool swap = __builtin_amdgcn_readfirstlane((int)(sa > split)) != 0;     // the same for every ray: they share src
        const int nr = swap ? left + 1 : left, fr = swap ? left : left + 1;
        if (nearm) {
            if (farm | pushm) {
                if (st.sp >= 64u || (st.sp >= PT_PKT_LDS && (st.sp - PT_PKT_LDS) * 2u + 2u > S.ovf64_entries * 3u)) {
                    atomicOr(S.err, (unsigned long long)PT_LIMIT_F64_STACK);      // cannot happen for a tree of the uploaded depth
                    return false;
                }
                st.push(fr, pushm, farm, in && push, tmid, ext);
            }
            if (in && !near_only && !far_only) ext = tmid;
            in = in && !far_only;
            curr = nr; act = nearm;
        } else {
            curr = fr; act = farm;
        }
    }
    return at_leaf || found;
}
