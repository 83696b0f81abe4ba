// search() of util/kdtree.cpp:216-257 for the rays of one wave that leave the same point (camera rays), walked as ONE
// packet.  The wave stands in one node at a time: the node and its triangle list come through the scalar unit, a
// 64-bit mask says which rays are in the node, and rays that part ways with the wave wait in a stack entry until the
// wave comes back for them.  Every ray makes exactly the decisions of the reference's search on its own numbers and
// meets its nodes and triangles in the order of its own search (rays that share their origin agree on which child of
// a node is the near one), so hit records are those of search_ref_t bit for bit; only the order in which DIFFERENT
// rays advance changes.  tests/packet_check.c states the same walk in plain C and checks it against the oracle.
//
// Before the rays test the triangles of a node, the packet as a whole is tested against them, one triangle per lane
// (packet_cull): a triangle that provably none of the packet's rays can pass is left out.  What that test needs of a
// triangle depends on the camera only, and is written once per view for every entry of the reference lists (k_packet_prep).
#pragma once
#include "pt_exact.h"

#define PT_PKT_LDS 8                       // stack entries whose per-lane numbers live in LDS; deeper ones in device memory
#define PT_PKT_LDS_BYTES (2 * PT_PKT_LDS * 64 * 8)   // shared memory of a block of the search kernel
#define PT_PKT_CULL_MIN 1                  // lists shorter than this are tested without the packet test
#ifndef PT_PKT_PRE_TRI
#define PT_PKT_PRE_TRI 0                   // the next triangle of a list is fetched while the current one is tested
#endif
#ifndef PT_PKT_PRE_NODE
#define PT_PKT_PRE_NODE 0                  // both children of a node are fetched before its list is gone through
#endif
#ifndef PT_PKT_WAVES
#define PT_PKT_WAVES 4                     // waves per SIMD the search kernel is compiled for
#endif

// PT_PKT_CLOCKS (an instrumented build for tools/pkt_clocks.sh, never the product): wave clocks per part of the walk.
// Slots: 0 pop, 1 node fetch, 2 packet test of a list, 3 triangle fetch + the reference's test, 4 the step to the
// children, 5 everything else of a packet (set-up, results), 6 the first packet test of a packet (counted apart from 2),
// 7 the whole first packet of a wave (also counted in the other slots).
#ifdef PT_PKT_CLOCKS
__device__ unsigned long long g_pkt_clk[16];
struct PktClk {
    unsigned long long acc[8], cnt[8], t, t_wave; bool first;
    PT_DEV void reset() { for (int k = 0; k < 8; k++) acc[k] = cnt[k] = 0; t = t_wave = __builtin_amdgcn_s_memtime(); first = true; }
    PT_DEV void packet_done() { if (cnt[7] == 0) { acc[7] = __builtin_amdgcn_s_memtime() - t_wave; cnt[7] = 1; } }
    PT_DEV void lap(int slot) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc[slot] += n - t; cnt[slot]++; t = n; }
    PT_DEV void lap2() { lap(first ? 6 : 2); first = false; }
    PT_DEV void flush() {
        if ((threadIdx.x & 63u) == 0)
            for (int k = 0; k < 8; k++) { atomicAdd(&g_pkt_clk[k], acc[k]); atomicAdd(&g_pkt_clk[8 + k], cnt[k]); }
    }
};
#else
struct PktClk { bool first; PT_DEV void reset() {} PT_DEV void lap(int) {} PT_DEV void lap2() {} PT_DEV void packet_done() {} PT_DEV void flush() {} };
#endif

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

// The packet test: true means that no ray that leaves `o` with a direction inside the packet's bounds can pass the
// reference's tests on the triangle (v0, e1, e2), so the reference rejects it for every ray of the packet and leaving
// it out changes nothing.
//
// With T = o - v0 the reference computes, per ray of direction d (util/polygonMesh.cpp:56-72),
//     det = (d x e2) . e1 = d . (e2 x e1)      u = un / det,  un = (d x e2) . T  = d . (e2 x T)
//     v = vn / det,  vn = (T x e1) . d         t = tn / det,  tn = (T x e1) . e2
// and rejects when u < 0, u > 1, v < 0, u + v > 1 or (in searchNode, util/kdtree.cpp:186) t <= 0.  det, un, vn and
// w = un + vn - det are linear in d, so their ranges over the packet follow from the packet's bounds; tn does not
// depend on d.  The triangle is left out when the sign of det is the same for the whole packet and, with that sign,
// un < 0, or vn < 0, or w > 0, or tn < 0 holds for the whole packet -- each by a margin:
// every quantity is a sum of at most six products x_i y_j z_k of three vectors, so M = |x|_1 |y|_1 |z|_1 bounds the sum of
// their magnitudes, and any fp64 evaluation of it (the reference's, this one's) is within 40 * 2^-53 * M < 1e-14 M of the
// exact value.  A quantity that is at least 1e-12 M away from zero therefore has its sign as the reference computes it,
// and the quotients the reference forms from such quantities are at least 1e-12 in magnitude away from 0 (resp. from 1
// for u + v): no rounding, underflow or NaN can turn the comparison.  M is required to lie in [1e-100, 1e100];
// triangles outside that range of magnitudes, and packets for which det changes sign, are kept.
//
// All camera rays of a view leave the same point, so everything but the packet's bounds is a property of the (view,
// reference) pair: k_packet_prep writes, once per view and reference, the four linear forms in the camera's frame, each
// divided by its M (so that the margin is a constant and nothing can overflow), rounded to fp32, and the sign class of
// tn.  The packet then evaluates the forms' ranges in fp32: 3 multiplications by bounds of magnitude <= 1.5, a min / max
// and two additions per end, on inputs rounded once -- within 23 * 2^-24 < 1.4e-6 of the fp64 value, to which the
// remainder term of the bounds adds less than 1e-7 (checked per packet) -- and asks for a margin of 1e-5 where fp64
// needs 1e-12.  What it gives up is triangles that miss the packet by less than 1e-5 of their own size.
#define PT_PKT_MARGIN 1e-5f
struct PacketForms { float4 det, un, vn, w; };      // xyz: the form on the camera's Z, X, Y; det.w: sign class of tn (+1, -1, 0)

PT_DEV PacketForms packet_forms(const D3 &Z, const D3 &X, const D3 &Y, const D3 &o, const D3 &v0, const D3 &e1, const D3 &e2) {
    const D3 T = dsub(o, v0);
    const D3 Nn = dcross(e2, e1), A = dcross(e2, T), Bv = dcross(T, e1);
    const D3 Cw = dsub(dadd(A, Bv), Nn);
    const double tn = ddot(Bv, e2);
    const double n1 = dl1(e1), n2 = dl1(e2), nT = dl1(T);
    const double Md = 2.0 * n1 * n2, Mu = 2.0 * n2 * nT, Mv = 2.0 * nT * n1, Mt = nT * n1 * n2, Mw = Md + Mu + Mv;
    PacketForms F;
    const float nan = __int_as_float(0x7FC00000);
    if (!(Md >= 1e-100 && Md <= 1e100 && Mu >= 1e-100 && Mu <= 1e100 && Mv >= 1e-100 && Mv <= 1e100 && Mt <= 1e100)) {
        F.det = F.un = F.vn = F.w = make_float4(nan, nan, nan, 0.f);      // every comparison fails: kept
        return F;
    }
    const double mt = 1e-12 * Mt;
    F.det = make_float4((float)(ddot(Nn, Z) / Md), (float)(ddot(Nn, X) / Md), (float)(ddot(Nn, Y) / Md), tn > mt ? 1.f : (tn < -mt ? -1.f : 0.f));
    F.un = make_float4((float)(ddot(A, Z) / Mu), (float)(ddot(A, X) / Mu), (float)(ddot(A, Y) / Mu), 0.f);
    F.vn = make_float4((float)(ddot(Bv, Z) / Mv), (float)(ddot(Bv, X) / Mv), (float)(ddot(Bv, Y) / Mv), 0.f);
    F.w = make_float4((float)(ddot(Cw, Z) / Mw), (float)(ddot(Cw, X) / Mw), (float)(ddot(Cw, Y) / Mw), 0.f);
    return F;
}

// The packet's bounds as the fp32 test reads them; ok: the remainder term is small enough to be covered by the margin.
struct PacketBoundsF { float c0, c1, a0, a1, b0, b1; bool ok; };
PT_DEV PacketBoundsF packet_bounds_f(const PacketBounds &B) {
    PacketBoundsF b;
    b.c0 = (float)B.c0; b.c1 = (float)B.c1; b.a0 = (float)B.a0; b.a1 = (float)B.a1; b.b0 = (float)B.b0; b.b1 = (float)B.b1;
    b.ok = B.rr < 1e-7 && fabs(B.c0) <= 1.5 && fabs(B.c1) <= 1.5 && fabs(B.a0) <= 1.5 && fabs(B.a1) <= 1.5 && fabs(B.b0) <= 1.5 && fabs(B.b1) <= 1.5;
    return b;
}
PT_DEV void form_range_f(const PacketBoundsF &b, const float4 &f, float &lo, float &hi) {
    const float x0 = b.c0 * f.x, x1 = b.c1 * f.x, y0 = b.a0 * f.y, y1 = b.a1 * f.y, z0 = b.b0 * f.z, z1 = b.b1 * f.z;
    lo = fminf(x0, x1) + fminf(y0, y1) + fminf(z0, z1);
    hi = fmaxf(x0, x1) + fmaxf(y0, y1) + fmaxf(z0, z1);
}
PT_DEV bool packet_cull(const PacketBoundsF &b, const float4 &fd, const float4 &fu, const float4 &fv, const float4 &fw) {
    const float K = PT_PKT_MARGIN;
    float dlo, dhi, ulo, uhi, vlo, vhi, wlo, whi;
    form_range_f(b, fd, dlo, dhi);
    form_range_f(b, fu, ulo, uhi);
    form_range_f(b, fv, vlo, vhi);
    form_range_f(b, fw, wlo, whi);
    const bool pos = dlo > K, neg = dhi < -K;
    const bool out_pos = uhi < -K || vhi < -K || wlo > K || fd.w < 0.f;
    const bool out_neg = ulo > K || vlo > K || whi < -K || fd.w > 0.f;
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

// A triangle of a reference list and a node of the tree, fetched through the scalar unit (the whole wave wants the same one).
struct TriU { D3 v0, e1, e2; int32_t id; };
PT_DEV TriU load_tri_u(const DevScene &S, int k) {
    const double *tri = S.refTri + (size_t)k * 9;
    TriU t;
    t.v0 = uload3(tri); t.e1 = uload3(tri + 3); t.e2 = uload3(tri + 6);
    t.id = ((PtUniformI)(uintptr_t)S.refId)[k];
    return t;
}
PT_DEV NodeA load_node_u(const DevScene &S, int k) {
    typedef const NodeA __attribute__((address_space(4))) *PtUniformNode;
    const PtUniformNode q = (PtUniformNode)(uintptr_t)S.nodeA + k;
    NodeA n;
    n.split = q->split; n.axis = q->axis; n.left = q->left; n.first = q->first; n.count = q->count;
    return n;
}

// The rays of a packet as the walk reads them: the origin (one for the whole wave), the direction, and per axis the
// divisor of util/kdtree.cpp:236 -- all plain values, so that picking a component by the node's axis is a select
// between registers.
struct PacketRay { double ox, oy, oz, dx, dy, dz, tx, ty, tz; };
PT_DEV PacketRay packet_ray(const RayD &r) {
    PacketRay p;
    p.ox = uniform_d(r.src.x); p.oy = uniform_d(r.src.y); p.oz = uniform_d(r.src.z);
    p.dx = r.dir.x; p.dy = r.dir.y; p.dz = r.dir.z;
    p.tx = p.dx == 0 ? 0.0000001 : p.dx; p.ty = p.dy == 0 ? 0.0000001 : p.dy; p.tz = p.dz == 0 ? 0.0000001 : p.dz;
    return p;
}

// KDNode::searchNode (util/kdtree.cpp:180-197) for the rays `in` of the packet on the list of node `node0`.
// forms: the view's table of k_packet_prep, four arrays of n_ref entries (det, un, vn, w).
// second: this ray is in its second search (see packet_search): a triangle that would improve on the bound is reported, the
// hit record stays as the first search left it.
PT_DEV bool packet_node(const DevScene &S, int node0, int first0, int count0, const RayD &ray, bool in, bool second, const PacketBoundsF &B,
                        const float4 *forms, HitD &h, PktClk &C) {
    const uint32_t lane = threadIdx.x & 63u;
    bool improved = false;
    for (int base = 0; base < count0; base += 64) {
        const int n = count0 - base < 64 ? count0 - base : 64;
        unsigned long long surv = n == 64 ? ~0ull : (1ull << n) - 1ull;
        if (B.ok && count0 >= PT_PKT_CULL_MIN) {
            bool keep = false;
            if ((int)lane < n) {
                const float4 *f = forms + (size_t)(first0 + base + (int)lane);
                const size_t nr = (size_t)S.n_ref;
                keep = !packet_cull(B, f[0], f[nr], f[2 * nr], f[3 * nr]);
            }
            surv = __ballot(keep);
        }
        C.lap2();
        if (!surv) continue;
        // the triangles that are left, one after the other for the whole wave; the next one is on its way while this one is tested
        int k = first0 + base + (int)__builtin_ctzll(surv);
        surv &= surv - 1ull;
        TriU cur = load_tri_u(S, k);
        for (;;) {
            const bool more = surv != 0;
            TriU nxt = cur;
            if (PT_PKT_PRE_TRI && more) {
                k = first0 + base + (int)__builtin_ctzll(surv);
                surv &= surv - 1ull;
                nxt = load_tri_u(S, k);
            }
            if (in) {
                double t, bu, bv;
                if (tri_intersect(cur.v0, cur.e1, cur.e2, ray, t, bu, bv) && t > 0 && t < h.tmin) {
                    const D3 hit = dadd(ray.src, dscale(t, ray.dir));
                    const double *b = S.nodeBox + (size_t)node0 * 6;
                    const D3 lo = uload3(b), hi = uload3(b + 3);
                    if (hit.x >= lo.x && hit.y >= lo.y && hit.z >= lo.z && hit.x <= hi.x && hit.y <= hi.y && hit.z <= hi.z) {
                        improved = true;
                        if (!second) { h.tmin = t; h.u = bu; h.v = bv; h.tri = cur.id; }
                    }
                }
            }
            C.lap(3);
            if (!more) break;
            if (!PT_PKT_PRE_TRI) {
                k = first0 + base + (int)__builtin_ctzll(surv);
                surv &= surv - 1ull;
                nxt = load_tri_u(S, k);
            }
            cur = nxt;
        }
    }
    return improved;
}

// The search of every `live` ray of the wave: src (the same for all) and `dir`; h.tmin is the in/out bound as in
// search_ref_t, whose result this returns.  Every lane of the wave must call it.
//
// NOTED (mainOMP model): the walk also answers the question of SearchNote -- can the second search of mainOMP.cpp:60, the
// same ray bounded by the hit just found, find anything? -- and answers it by doing what is left of that search.  The second
// search repeats the first one's decisions and finds nothing on the way, unless a far-only step lay beyond the hit (far_max,
// as in search_ref_t); having arrived where the first search returned, it goes on with the entries the first search left on
// the stack.  The walk is standing right there with those entries in hand: a ray that has its hit (state `second`) takes up
// an entry that starts before the hit under the second search's own rules (the bound is the hit's distance), ends at one
// that starts AT the hit (util/kdtree.cpp:229-231), and passes over one that starts beyond it (the second search never
// pushed it, :246).  Only a triangle that would improve on the bound makes the note say "run it" (stack_min = -DBL_MAX);
// otherwise stack_min stays +DBL_MAX.  tests/packet_check.c states this in C and checks, ray by ray, that the second search
// as written is the first one's walk thinned out followed by exactly this resumed walk: on the atrium 1 095 of 6.2 M camera
// rays left such entries, each resumed walk was 2 or 3 nodes long, none found anything.
template <bool NOTED>
PT_DEV bool packet_search(const DevScene &S, const RayD &ray_in, bool live, const PacketBoundsF &B, const float4 *forms, HitD &h, SearchNote *note, double *lds, PktClk &C) {
    const uint32_t lane = threadIdx.x & 63u;
    const PacketRay pr = packet_ray(ray_in);
    const RayD ray = make_ray(d3(pr.ox, pr.oy, pr.oz), d3(pr.dx, pr.dy, pr.dz));
    PacketStack st = make_packet_stack(S, lds);
    double ent = 0, ext = 0;
    if (NOTED) { note->far_max = -DBL_MAX; note->stack_min = DBL_MAX; }
    bool in = live && box_intersect(S.root_lo, S.root_hi, ray, ent, ext) && !(ent >= h.tmin);
    bool finished = !in;                            // this ray's first search has ended (or never began)
    bool second = false;                            // ... with a hit at a leaf, and its second search has not ended
    bool found = false, at_leaf = false;
    unsigned long long act = __ballot(in);
    int curr = 0;
    C.lap(5);
    C.first = true;
    // One loop with one back edge: a turn pops an entry if no ray is in the current node, and then, if some ray is in
    // the (new) current node, does that node.
    for (;;) {
        if (!act) {
            if (st.sp == 0 || !__ballot(!finished || second)) break;
            unsigned long long pushm, deferm;
            st.pop(curr, pushm, deferm);
            in = (deferm >> lane) & 1ull;
            if ((pushm >> lane) & 1ull) {
                double e, x;
                st.popped(e, x);
                if (!finished) {
                    if (e >= h.tmin) finished = true;               // util/kdtree.cpp:229-231: ends this ray's search (and a second one at the same entry)
                    else { ent = e; ext = x; in = true; }
                } else if (NOTED && second) {
                    if (e < h.tmin) { ent = e; ext = x; in = true; }
                    else if (e == h.tmin) second = false;
                }
            }
            act = __ballot(in);
            C.lap(0);
        }
        if (act) {
            const NodeA nd = load_node_u(S, curr);
            const int axis = nd.axis, left = nd.left;
            bool improved = false;
            C.lap(1);
            if (nd.count > 0) improved = packet_node(S, curr, nd.first, nd.count, ray, in, NOTED && second, B, forms, h, C);
            if (NOTED && in && second && improved) {    // the second search would find something: it has to be run as written
                note->stack_min = -DBL_MAX;
                second = false; in = false; improved = false;
            }
            if (axis < 0) {                          // a leaf: the rays that improved their hit here are done (:252-253)
                if (in && improved) { at_leaf = true; finished = true; second = NOTED; }
                act = 0;
            } else {
                found |= in && improved;
                const double sa = axis == 0 ? pr.ox : (axis == 1 ? pr.oy : pr.oz);
                const double tdir = axis == 0 ? pr.tx : (axis == 1 ? pr.ty : pr.tz);
                const double tmid = (nd.split - sa) / tdir;
                const bool near_only = tmid >= ext || tmid < 0;
                const bool far_only = !near_only && tmid <= ent && tmid <= h.tmin;
                const bool push = !near_only && !far_only && tmid <= h.tmin;
                const unsigned long long nearm = __ballot(in && !far_only), farm = __ballot(in && far_only), pushm = __ballot(in && push);
                if (NOTED && in && !second && far_only && tmid > note->far_max) note->far_max = tmid;
                const bool swap = sa > nd.split;     // the same for every ray: they share src
                const int nr = swap ? left + 1 : left, fr = swap ? left : left + 1;
                bool overflow = false;
                if (nearm) {
                    if (farm | pushm) {
                        overflow = st.sp >= 64u || (st.sp >= PT_PKT_LDS && (st.sp - PT_PKT_LDS) * 2u + 2u > S.ovf64_entries * 3u);
                        if (!overflow) st.push(fr, pushm, farm, in && push, tmid, ext);
                    }
                    if (in && !near_only && !far_only) ext = tmid;
                    in = in && !far_only;
                    curr = nr; act = nearm;
                } else {
                    curr = fr; act = farm;           // (both masks empty: the rays that were here have all just left the walk)
                }
                if (overflow) {                      // cannot happen for a tree of the uploaded depth; ends the walk, the call fails
                    atomicOr(S.err, (unsigned long long)PT_LIMIT_F64_STACK);
                    st.sp = 0; act = 0; finished = true; second = false;
                }
            }
            C.lap(4);
        }
    }
    return at_leaf || found;
}
