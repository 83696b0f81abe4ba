/*
 * Test helper (built by tests/test_packet_rule.py, never by the product): a CPU model of the device's packet search,
 * checked ray by ray against the oracle's search() restatement.
 *
 * The device searches the camera rays of one wave -- up to 64 rays that leave the same point -- as ONE packet: the
 * wave stands in one node at a time, a mask says which rays are in it, and the rays that part ways wait in a stack
 * entry until the wave comes back for them.  Each ray still makes exactly the decisions of util/kdtree.cpp:216-257 on
 * its own numbers, and meets its nodes and triangles in the order of its own search.  Before the rays of the packet
 * test the triangles of a node, the packet as a whole is tested against them (one triangle per lane): a triangle that
 * provably none of the packet's rays can hit is left out.  This file states both parts in plain C:
 *     packet_search()   the walk; for mainOMP.cpp:60 (see research_check.c) it notes far-only steps beyond the hit and goes
 *                       on, for a ray that has its hit, with the part of the second search that lies behind the point where
 *                       the first one returned -- the walk is standing there with the ray's stack entries in hand
 *     packet_cull()     the conservative packet-vs-triangle test as the proof is written: fp64, margin 1e-12
 *     packet_cull_f()   the test the device runs: the same linear forms per (camera, triangle), divided by their
 *                       magnitude bounds and rounded to fp32, ranges evaluated in fp32, margin 1e-5; every triangle
 *                       it leaves out must also be left out by packet_cull() (counted in `loose`, which must be 0)
 * and packet_check() runs them on the camera rays of a view, packets formed as the device forms them, and counts every
 * ray whose result differs from search_ref()/search_noted() in any bit.
 */
#include <stddef.h>

#include "research_check.c"

#define PW 64
typedef unsigned long long u64;

typedef struct {
    long packets, rays, node_visits, lists, list_entries, survivors, lane_tests, any_pass, pushes, pops, max_depth;
    long ref_node_visits, ref_tri_tests, loose, kept64, n_again, n_again_old, n_found2, n_resumed_nodes;
    long depth_hist[64];
} PStats;

/* Bounds of a packet's directions in a frame (d0, a, b): every direction d of the packet satisfies
   d = c d0 + al a + be b + r  with c in [c0, c1], al in [a0, a1], be in [b0, b1] and |r|_2 <= rr. */
typedef struct { V3 d0, a, b; double c0, c1, a0, a1, b0, b1, rr; } Bundle;

/* The frame is the camera's (forward, right, up of main.cpp:119-120), in which a packet of neighbouring pixels is an
   axis-aligned box; any frame gives valid bounds, since the remainder r is measured, not assumed. */
static Bundle bundle_of(const Ray *rays, u64 live, double sa, double ca) {
    Bundle B; memset(&B, 0, sizeof B);
    B.d0 = v3(ca, 0, sa); B.a = v3(sa, 0, -ca); B.b = v3(0, 1, 0);
    B.c0 = B.a0 = B.b0 = INFINITY; B.c1 = B.a1 = B.b1 = -INFINITY;
    for (int l = 0; l < PW; l++) if (live >> l & 1) {
        V3 d = rays[l].dir;
        double c = vdot(d, B.d0), al = vdot(d, B.a), be = vdot(d, B.b);
        V3 r = vsub(d, vadd(vadd(vscale(c, B.d0), vscale(al, B.a)), vscale(be, B.b)));
        double rr = vlen(r);
        if (c < B.c0) B.c0 = c; if (c > B.c1) B.c1 = c;
        if (al < B.a0) B.a0 = al; if (al > B.a1) B.a1 = al;
        if (be < B.b0) B.b0 = be; if (be > B.b1) B.b1 = be;
        if (rr > B.rr) B.rr = rr;
    }
    B.rr = 2.0 * B.rr + 1e-15;
    return B;
}

static inline double l1(V3 a) { return fabs(a.x) + fabs(a.y) + fabs(a.z); }
static inline void form_range(const Bundle *B, V3 A, double *lo, double *hi) {
    double f0 = vdot(B->d0, A), fa = vdot(B->a, A), fb = vdot(B->b, A);
    double x0 = B->c0 * f0, x1 = B->c1 * f0, y0 = B->a0 * fa, y1 = B->a1 * fa, z0 = B->b0 * fb, z1 = B->b1 * fb;
    double r = B->rr * l1(A);
    *lo = fmin(x0, x1) + fmin(y0, y1) + fmin(z0, z1) - r;
    *hi = fmax(x0, x1) + fmax(y0, y1) + fmax(z0, z1) + r;
}

/* 1: no ray of the packet (origin o, directions bounded by B) can pass the reference's tests on this triangle. */
static int packet_cull(const Bundle *B, V3 o, V3 v0, V3 e1, V3 e2) {
    const double K = 1e-12;
    V3 T = vsub(o, v0);
    V3 Nn = vcross(e2, e1), A = vcross(e2, T), Bv = vcross(T, e1);
    V3 Cw = vsub(vadd(A, Bv), Nn);
    double tn = vdot(Bv, e2);
    double n1 = l1(e1), n2 = l1(e2), nT = l1(T);
    double Md = 2.0 * n1 * n2, Mu = 2.0 * n2 * nT, Mv = 2.0 * nT * n1, Mt = nT * n1 * n2;
    if (!(Md >= 1e-100 && Md <= 1e100 && Mu >= 1e-100 && Mu <= 1e100 && Mv >= 1e-100 && Mv <= 1e100 && Mt <= 1e100)) return 0;
    double md = K * Md, mu = K * Mu, mv = K * Mv, mw = K * (Md + Mu + Mv), mt = K * Mt;
    double dlo, dhi, ulo, uhi, vlo, vhi, wlo, whi;
    form_range(B, Nn, &dlo, &dhi);
    int pos = dlo > md, neg = dhi < -md;
    if (!pos && !neg) return 0;
    form_range(B, A, &ulo, &uhi);
    form_range(B, Bv, &vlo, &vhi);
    form_range(B, Cw, &wlo, &whi);
    if (pos) return uhi < -mu || vhi < -mv || wlo > mw || tn < -mt;
    return ulo > mu || vlo > mv || whi < -mw || tn > mt;
}

/* ---- the second search, ray by ray, for the structural check below ------------------------------------------------ */
#define TRACE_MAX 4096
typedef struct { const KD *n[TRACE_MAX]; int cnt; } Trace;
static inline void trace_add(Trace *t, const KD *n) { if (t->cnt < TRACE_MAX) t->n[t->cnt] = n; t->cnt++; }

/* search() of util/kdtree.cpp:216-257 with the nodes it visits written down; on return the entries left on its stack
   (bottom first) are in sn / se / sx, *left of them.  With resume != 0 the search does not start at the root but goes on
   from such a stack under the rules of the second search described at packet_search(): the bound h->tmin stays, an entry
   that starts at the bound ends the search, one that starts beyond it is passed over; a triangle that would improve on
   the bound ends it with 2. */
static int search_traced(const Scene *s, const Ray *ray, Hit *h, Trace *tr, const KD **sn, double *se, double *sx, int *left, int resume) {
    double ent, ext;
    const KD *curr = s->root;
    Cnt c; memset(&c, 0, sizeof c);
    int sp = resume ? *left : 0, found = 0, ret = -1;
    const double bound = h->tmin;
    if (!resume) {
        if (!box_intersect(curr->lo, curr->hi, ray, &ent, &ext)) { *left = 0; return 0; }
        sn[sp] = curr; se[sp] = ent; sx[sp] = ext; sp++;
    }
    while (sp > 0 && ret < 0) {
        sp--;
        curr = sn[sp]; ent = se[sp]; ext = sx[sp];
        if (resume) {
            if (ent > bound) continue;
            if (ent == bound) break;
        } else if (ent >= h->tmin) break;
        while (!curr->leaf) {
            trace_add(tr, curr);
            Hit before = *h;
            int f = search_node(s, curr, ray, h, &c);
            if (resume && f) { *h = before; *left = sp; return 2; }
            found |= f;
            double sp_ = curr->splt;
            int a = curr->axis;
            double da = vget(ray->dir, a);
            double tdir = da == 0 ? 0.0000001 : da;
            double tmid = (sp_ - vget(ray->src, a)) / tdir;
            const KD *nr = curr->left, *fr = curr->right;
            if (vget(ray->src, a) > sp_) { const KD *x = nr; nr = fr; fr = x; }
            if (tmid >= ext || tmid < 0)
                curr = nr;
            else if (tmid <= ent && tmid <= h->tmin)
                curr = fr;
            else {
                if (tmid <= h->tmin) { sn[sp] = fr; se[sp] = tmid; sx[sp] = ext; sp++; }
                curr = nr;
                ext = tmid;
            }
        }
        trace_add(tr, curr);
        Hit before = *h;
        if (search_node(s, curr, ray, h, &c)) {
            if (resume) { *h = before; *left = sp; return 2; }
            ret = 1;
        }
    }
    *left = sp;
    return ret > 0 ? 1 : found;
}

/* For a ray whose first search returned a hit AT A LEAF: is the second search as written (from the root, bounded by the
   hit) the first search's walk, thinned out, followed by exactly the walk that resumes from the first search's stack?
   0: yes.  Also hands out the resumed walk for comparison with what the packet walk did for this ray. */
static int second_search_structure(const Scene *s, const Ray *ray, Trace *resumed, int *resumed_ret) {
    static _Thread_local Trace t1, t2;
    const KD *sn[STK]; double se[STK], sx[STK];
    const KD *qn[STK]; double qe[STK], qx[STK];
    int left = 0, qleft = 0;
    Hit h1; h1.tmin = FLT_MAX; h1.m = h1.t = -1; h1.u = h1.v = 0;
    t1.cnt = t2.cnt = resumed->cnt = 0;
    if (!search_traced(s, ray, &h1, &t1, sn, se, sx, &left, 0)) return -1;
    Hit h2 = h1; h2.m = h2.t = -1; h2.u = h2.v = 0;
    const int f2 = search_traced(s, ray, &h2, &t2, qn, qe, qx, &qleft, 0);
    Hit h3 = h1;
    *resumed_ret = search_traced(s, ray, &h3, resumed, sn, se, sx, &left, 1);
    if (t1.cnt > TRACE_MAX || t2.cnt > TRACE_MAX || resumed->cnt > TRACE_MAX) return -2;
    if (f2) return *resumed_ret == 2 ? 0 : 1;              /* it finds something: the resumed walk must have met it (suspect) */
    if (*resumed_ret == 2) return 2;
    if (resumed->cnt > t2.cnt) return 3;
    const int head = t2.cnt - resumed->cnt;
    for (int i = 0; i < resumed->cnt; i++) if (t2.n[head + i] != resumed->n[i]) return 4;
    int j = 0;                                              /* the head: a subsequence of the first search's walk */
    for (int i = 0; i < head; i++) {
        while (j < t1.cnt && t1.n[j] != t2.n[i]) j++;
        if (j == t1.cnt) return 5;
        j++;
    }
    return 0;
}

/* The device's form of the test (pt_packet.h: packet_forms, packet_bounds_f, packet_cull). */
typedef struct { float f[4][3]; float tsign; } Forms;
static Forms packet_forms(const Bundle *B, V3 o, V3 v0, V3 e1, V3 e2) {
    Forms F; memset(&F, 0, sizeof F);
    V3 T = vsub(o, v0);
    V3 Nn = vcross(e2, e1), A = vcross(e2, T), Bv = vcross(T, e1);
    V3 Cw = vsub(vadd(A, Bv), Nn);
    double tn = vdot(Bv, e2);
    double n1 = l1(e1), n2 = l1(e2), nT = l1(T);
    double Md = 2.0 * n1 * n2, Mu = 2.0 * n2 * nT, Mv = 2.0 * nT * n1, Mt = nT * n1 * n2, Mw = Md + Mu + Mv;
    if (!(Md >= 1e-100 && Md <= 1e100 && Mu >= 1e-100 && Mu <= 1e100 && Mv >= 1e-100 && Mv <= 1e100 && Mt <= 1e100)) {
        for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) F.f[k][c] = NAN;
        return F;
    }
    double mt = 1e-12 * Mt;
    const V3 form[4] = { Nn, A, Bv, Cw };
    const double M[4] = { Md, Mu, Mv, Mw };
    for (int k = 0; k < 4; k++) {
        F.f[k][0] = (float)(vdot(form[k], B->d0) / M[k]);
        F.f[k][1] = (float)(vdot(form[k], B->a) / M[k]);
        F.f[k][2] = (float)(vdot(form[k], B->b) / M[k]);
    }
    F.tsign = tn > mt ? 1.f : (tn < -mt ? -1.f : 0.f);
    return F;
}
static inline void form_range_f(const float b[6], const float f[3], float *lo, float *hi) {
    float x0 = b[0] * f[0], x1 = b[1] * f[0], y0 = b[2] * f[1], y1 = b[3] * f[1], z0 = b[4] * f[2], z1 = b[5] * f[2];
    *lo = fminf(x0, x1) + fminf(y0, y1) + fminf(z0, z1);
    *hi = fmaxf(x0, x1) + fmaxf(y0, y1) + fmaxf(z0, z1);
}
static int packet_cull_f(const Bundle *B, const Forms *F) {
    const float K = 1e-5f;
    const float b[6] = { (float)B->c0, (float)B->c1, (float)B->a0, (float)B->a1, (float)B->b0, (float)B->b1 };
    if (!(B->rr < 1e-7 && fabs(B->c0) <= 1.5 && fabs(B->c1) <= 1.5 && fabs(B->a0) <= 1.5 && fabs(B->a1) <= 1.5 &&
          fabs(B->b0) <= 1.5 && fabs(B->b1) <= 1.5))
        return 0;
    float lo[4], hi[4];
    for (int k = 0; k < 4; k++) form_range_f(b, F->f[k], &lo[k], &hi[k]);
    int pos = lo[0] > K, neg = hi[0] < -K;
    int out_pos = hi[1] < -K || hi[2] < -K || lo[3] > K || F->tsign < 0.f;
    int out_neg = lo[1] > K || lo[2] > K || hi[3] < -K || F->tsign > 0.f;
    return (pos && out_pos) || (neg && out_neg);
}

typedef struct { const KD *node; u64 push, defer; double ent[PW], ext[PW]; } PEntry;

/* One node's list for the rays in `act`; returns the mask of rays that improved their hit here. */
/* m2: the rays that are in their SECOND search (mainOMP.cpp:60, bounded by the hit of the first): a triangle that would
   improve on the bound is only reported (bit in the result), the hit record stays as the first search left it. */
static u64 packet_node(const Scene *s, const KD *nd, const Ray *rays, u64 act, u64 m2, const Bundle *B, Hit *h, PStats *ps, int cull) {
    u64 improved = 0;
    ps->lists++;
    for (int i = 0; i < nd->n; i++) {
        const Mesh *me = &s->mesh[nd->m[i]];
        ps->list_entries++;
        if (cull) {
            const int *V = me->tris[nd->t[i]].vi;
            V3 v0 = me->vert[V[0]], e1 = vsub(me->vert[V[1]], v0), e2 = vsub(me->vert[V[2]], v0), o = rays[__builtin_ctzll(act)].src;
            const int out64 = packet_cull(B, o, v0, e1, e2);
            if (!out64) ps->kept64++;
            const Forms F = packet_forms(B, o, v0, e1, e2);
            if (packet_cull_f(B, &F)) {
                if (!out64) ps->loose++;               /* the fp32 test claimed more than the fp64 test proves */
                continue;
            }
        }
        ps->survivors++;
        int any = 0;
        for (int l = 0; l < PW; l++) if (act >> l & 1) {
            double t, bu, bv;
            ps->lane_tests++;
            if (tri_intersect(me, nd->t[i], &rays[l], &t, &bu, &bv)) {
                any = 1;
                if (t > 0 && t < h[l].tmin) {
                    V3 hit = vadd(rays[l].src, vscale(t, rays[l].dir));
                    if (!(hit.x >= nd->lo.x && hit.y >= nd->lo.y && hit.z >= nd->lo.z &&
                          hit.x <= nd->hi.x && hit.y <= nd->hi.y && hit.z <= nd->hi.z))
                        continue;
                    improved |= 1ull << l;
                    if (m2 >> l & 1) continue;
                    h[l].tmin = t; h[l].u = bu; h[l].v = bv; h[l].m = nd->m[i]; h[l].t = nd->t[i];
                }
            }
        }
        ps->any_pass += any;
    }
    return improved;
}

/* The packet walk.  rays[l] for l in live share their origin.  ret[l] = what search() returns for ray l. */
/* States of a ray of the packet.  FIRST: in its first search.  OVER: that search ended without a hit (or never began).
   SECOND: the first search returned a hit at a leaf; the ray now stands where the second search of mainOMP.cpp:60 -- the
   same ray bounded by that hit -- stands after repeating the first one's decisions (research_check.c: it repeats them and
   finds nothing, unless a far-only step lay beyond the hit, which far_max reports), and goes on from there on the entries
   the first search left on the stack: an entry that starts before the hit is visited under the second search's own rules,
   one that starts AT the hit ends it (util/kdtree.cpp:229-231), one that starts beyond it was never pushed by the second
   search (:246) and is passed over.  ENDED: the second search has ended, or has met a triangle that improves on the
   bound (`suspect`: then the device runs the second search as written). */
enum { FIRST = 0, OVER = 1, SECOND = 2, ENDED = 3 };

static void packet_search(const Scene *s, const Ray *rays, u64 live, double sa, double ca, Hit *h, int *ret, Note *note, int *suspect_out, Trace *second, PStats *ps, int cull) {
    static _Thread_local PEntry stack[STK];
    double ent[PW], ext[PW];
    int state[PW];
    u64 act = 0, found = 0, suspect = 0;
    int sp = 0;
    const KD *curr = s->root;
    Bundle B = bundle_of(rays, live, sa, ca);
    for (int l = 0; l < PW; l++) {
        ret[l] = 0; state[l] = OVER;
        note[l].far_max = -INFINITY; note[l].stack_min = INFINITY;
        if (!(live >> l & 1)) continue;
        if (box_intersect(curr->lo, curr->hi, &rays[l], &ent[l], &ext[l]) && !(ent[l] >= h[l].tmin)) { act |= 1ull << l; state[l] = FIRST; }
    }
    ps->packets++;
    for (;;) {
        if (!act) {
            while (sp > 0 && !act) {
                const PEntry *e = &stack[--sp];
                ps->pops++;
                for (int l = 0; l < PW; l++) {
                    if (e->push >> l & 1) {
                        if (state[l] == FIRST) {
                            if (e->ent[l] >= h[l].tmin) {                               /* util/kdtree.cpp:229-231: the first search ends */
                                /* with a hit from an interior node the second search ends at this very entry too */
                                state[l] = (found >> l & 1) ? ENDED : OVER;
                            } else { ent[l] = e->ent[l]; ext[l] = e->ext[l]; act |= 1ull << l; }
                        } else if (state[l] == SECOND) {
                            if (e->ent[l] < h[l].tmin) { ent[l] = e->ent[l]; ext[l] = e->ext[l]; act |= 1ull << l; }
                            else if (e->ent[l] == h[l].tmin) state[l] = ENDED;
                        }
                    } else if (e->defer >> l & 1) act |= 1ull << l;
                }
                curr = e->node;
            }
            if (!act) break;
        }
        ps->node_visits++;
        u64 m2 = 0;
        for (int l = 0; l < PW; l++) if ((act >> l & 1) && state[l] == SECOND) { m2 |= 1ull << l; trace_add(&second[l], curr); }
        u64 improved = curr->n > 0 ? packet_node(s, curr, rays, act, m2, &B, h, ps, cull) : 0;
        for (int l = 0; l < PW; l++) if (improved & m2 & (1ull << l)) { suspect |= 1ull << l; state[l] = ENDED; }
        act &= ~(improved & m2);
        if (curr->leaf) {
            for (int l = 0; l < PW; l++) if ((improved & ~m2) >> l & 1) { ret[l] = 1; state[l] = SECOND; }
            act = 0;
            continue;
        }
        found |= improved & ~m2;
        if (!act) continue;
        u64 nearm = 0, faronly = 0, pushm = 0;
        double tmidv[PW];
        const int a = curr->axis;
        const double sp_ = curr->splt;
        const int swap = vget(rays[__builtin_ctzll(act)].src, a) > sp_;
        for (int l = 0; l < PW; l++) if (act >> l & 1) {
            double da = vget(rays[l].dir, a);
            double tdir = da == 0 ? 0.0000001 : da;
            double tmid = (sp_ - vget(rays[l].src, a)) / tdir;
            tmidv[l] = tmid;
            if (tmid >= ext[l] || tmid < 0) nearm |= 1ull << l;
            else if (tmid <= ent[l] && tmid <= h[l].tmin) {
                faronly |= 1ull << l;
                if (state[l] == FIRST && tmid > note[l].far_max) note[l].far_max = tmid;
            } else {
                nearm |= 1ull << l;
                if (tmid <= h[l].tmin) pushm |= 1ull << l;
            }
        }
        const KD *nr = swap ? curr->right : curr->left, *fr = swap ? curr->left : curr->right;
        if (nearm) {
            if (faronly | pushm) {
                if (sp >= STK) { fprintf(stderr, "packet: stack overflow\n"); abort(); }
                PEntry *e = &stack[sp++];
                e->node = fr; e->push = pushm; e->defer = faronly;
                for (int l = 0; l < PW; l++) if (pushm >> l & 1) { e->ent[l] = tmidv[l]; e->ext[l] = ext[l]; }
                ps->pushes++;
                if (sp > ps->max_depth) ps->max_depth = sp;
            }
            /* the rays that take the near child with the far one to follow (pushed or, beyond their hit, dropped) */
            for (int l = 0; l < PW; l++)
                if ((nearm >> l & 1) && !(tmidv[l] >= ext[l] || tmidv[l] < 0)) ext[l] = tmidv[l];
            curr = nr; act = nearm;
        } else {
            curr = fr; act = faronly;
        }
        ps->depth_hist[sp < 63 ? sp : 63]++;
    }
    for (int l = 0; l < PW; l++) if ((live >> l & 1) && !ret[l]) ret[l] = (int)(found >> l & 1);
    for (int l = 0; l < PW; l++) suspect_out[l] = (int)(suspect >> l & 1);
}

/* Camera rays of one view in the device's packets: with rb, strips of 7 pixels of a row x 9 sub-rays (63 lanes);
   without, 8x8 pixel tiles.  Rows row0, row0 + row_step, ... (tiles: tile rows).  out: [0] rays, [1] rays whose result
   differs from search_ref (must be 0), [2] rays whose far_max differs from search_noted's, or whose second search as written
   finds something although the walk says it need not be run (must be 0), then the PStats. */
void packet_check(const Scene *s, const double eye[3], double angle, double zoom, int size, int rb, double off,
                  int row0, int row_step, int cull, long *out, PStats *stats) {
    V3 src = v3(eye[0], eye[1], eye[2]);
    double sa = sin(angle), ca = cos(angle);
    long n_rays = 0, n_diff = 0, n_note = 0;
    PStats total; memset(&total, 0, sizeof total);
    const int nrows = rb ? size : (size + 7) / 8;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_rays, n_diff, n_note)
    for (int r = row0; r < nrows; r += row_step) {
        PStats ps; memset(&ps, 0, sizeof ps);
        const int nseg = rb ? (size + 6) / 7 : (size + 7) / 8;
        for (int seg = 0; seg < nseg; seg++) {
            Ray rays[PW]; u64 live = 0;
            memset(rays, 0, sizeof rays);
            for (int l = 0; l < PW; l++) {
                int i, j; double x = 0, y = 0;
                if (rb) {
                    if (l >= 63) continue;
                    int px = l / 9, sub = l % 9;
                    i = r; j = seg * 7 + px;
                    const double o0 = -off, o1 = o0 + off, o2 = o1 + off;
                    const int xi = sub % 3, yi = sub / 3;
                    x = xi == 0 ? o0 : (xi == 1 ? o1 : o2); y = yi == 0 ? o0 : (yi == 1 ? o1 : o2);
                } else {
                    i = r * 8 + (l >> 3); j = seg * 8 + (l & 7);
                }
                if (i >= size || j >= size) continue;
                rays[l] = camera_ray(src, zoom, sa, ca, size, i, j, x, y);
                live |= 1ull << l;
            }
            if (!live) continue;
            Hit h[PW]; int ret[PW], suspect[PW]; Note note[PW];
            static _Thread_local Trace second[PW], resumed;
            for (int l = 0; l < PW; l++) second[l].cnt = 0;
            for (int l = 0; l < PW; l++) { h[l].tmin = FLT_MAX; h[l].m = h[l].t = -1; h[l].u = h[l].v = 0; }
            packet_search(s, rays, live, sa, ca, h, ret, note, suspect, second, &ps, cull);
            for (int l = 0; l < PW; l++) if (live >> l & 1) {
                Hit a; Note na; Cnt c; memset(&c, 0, sizeof c);
                a.tmin = FLT_MAX; a.m = a.t = -1; a.u = a.v = 0;
                Hit b = a;
                int fa = search_ref(s, &rays[l], &a, &c), fb = search_noted(s, &rays[l], &b, &na);
                ps.ref_node_visits += c.node_visits; ps.ref_tri_tests += c.tri_tests;
                n_rays++;
                if (fa != ret[l] || memcmp(&a, &h[l], sizeof a) != 0) n_diff++;
                /* the second search of mainOMP.cpp:60 as written, against what the packet walk says about it */
                int again = 0, found2 = 0;
                if (fa) {
                    Hit a2 = a; Cnt c2; memset(&c2, 0, sizeof c2);
                    a2.m = a2.t = -1; a2.u = a2.v = 0;                      /* the bound stays: the first hit's distance */
                    found2 = search_ref(s, &rays[l], &a2, &c2);
                    again = note[l].far_max > a.tmin || suspect[l];
                    ps.n_again += again; ps.n_again_old += na.far_max > a.tmin || na.stack_min < a.tmin; ps.n_found2 += found2;
                }
                if (fb != fa || na.far_max != note[l].far_max || (found2 && !again)) n_note++;
                /* structure: whenever no far-only step lay beyond the hit, the second search as written ends with exactly the
                   walk that resumes from the first search's stack, and the packet walk made that very walk for this ray */
                if (fa && !(na.far_max > a.tmin)) {
                    int rret = 0;
                    const int st = second_search_structure(s, &rays[l], &resumed, &rret);
                    if (st == -1) { /* the hit came from an interior node only: nothing to resume */ }
                    else if (st != 0) n_note++;
                    if (st >= 0 && ret[l] && (h[l].m >= 0)) {
                        int same = resumed.cnt == second[l].cnt && (rret == 2) == (suspect[l] != 0);
                        for (int i = 0; same && i < resumed.cnt && i < TRACE_MAX; i++) same = resumed.n[i] == second[l].n[i];
                        ps.n_resumed_nodes += resumed.cnt;
                        if (!same) n_note++;
                    }
                }
            }
        }
#pragma omp critical
        {
            long *t = (long *)&total, *p = (long *)&ps;
            for (size_t k = 0; k < sizeof(PStats) / sizeof(long); k++)
                if (k == offsetof(PStats, max_depth) / sizeof(long)) { if (p[k] > t[k]) t[k] = p[k]; } else t[k] += p[k];
        }
    }
    total.rays = n_rays;
    out[0] = n_rays; out[1] = n_diff; out[2] = n_note;
    *stats = total;
}
