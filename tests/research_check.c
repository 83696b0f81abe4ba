/*
 * Test helper (built by tests/test_research_rule.py, never by the product): checks on the CPU, over many rays, the rule
 * by which the device skips the second search of mainOMP.cpp:60.
 *
 * The mainOMP front end searches the camera ray again, bounded by the hit the first search returned.  The device
 * notes two numbers during the first search:
 *     far_max   = the largest tmid of the steps that took the far child only   (util/kdtree.cpp:242-243)
 *     stack_min = the smallest entry distance still on the stack when the search ended
 * and claims: if neither far_max > t_hit nor stack_min < t_hit, the second search finds nothing.
 * This file restates the search with those notes, checks the restatement against the oracle's own search_ref on every
 * ray, runs the oracle's second search, and reports every ray where the claim fails.
 */
#include "../oracle/pt_oracle.c"

typedef struct { double far_max, stack_min; } Note;

static int search_noted(const Scene *s, const Ray *ray, Hit *h, Note *note) {
    double ent, ext;
    const KD *curr = s->root;
    Cnt c; memset(&c, 0, sizeof c);
    note->far_max = -INFINITY; note->stack_min = INFINITY;
    if (!box_intersect(curr->lo, curr->hi, ray, &ent, &ext)) return 0;
    const KD *sn[STK]; double se[STK], sx[STK];
    int sp = 0, found = 0, ret = -1;
    sn[sp] = curr; se[sp] = ent; sx[sp] = ext; sp++;
    while (sp > 0 && ret < 0) {
        sp--;
        curr = sn[sp]; ent = se[sp]; ext = sx[sp];
        if (ent >= h->tmin) break;
        while (!curr->leaf) {
            found |= search_node(s, curr, ray, h, &c);
            double sp_ = curr->splt;
            int a = curr->axis;
            double da = vget(ray->dir, a);
            double tdir = da == 0 ? 0.0000001 : da;
            double tmid = (sp_ - vget(ray->src, a)) / tdir;
            const KD *nr = curr->left, *fr = curr->right;
            if (vget(ray->src, a) > sp_) { const KD *x = nr; nr = fr; fr = x; }
            if (tmid >= ext || tmid < 0)
                curr = nr;
            else if (tmid <= ent && tmid <= h->tmin) {
                if (tmid > note->far_max) note->far_max = tmid;
                curr = fr;
            } else {
                if (tmid <= h->tmin) { sn[sp] = fr; se[sp] = tmid; sx[sp] = ext; sp++; }
                curr = nr;
                ext = tmid;
            }
        }
        if (search_node(s, curr, ray, h, &c)) ret = 1;
    }
    for (int i = 0; i < sp; i++)
        if (se[i] < note->stack_min) note->stack_min = se[i];
    return ret > 0 ? 1 : found;
}

/* out[0] rays that hit, [1] of those: rule says "search again", [2] second searches that found something,
   [3] restatement differs from search_ref (must be 0), [4] rule said "skip" but the second search found something (must be 0) */
void research_check(const Scene *s, long n, const double *src, const double *dir, long *out) {
    Cnt c; memset(&c, 0, sizeof c);
    for (int k = 0; k < 5; k++) out[k] = 0;
    for (long i = 0; i < n; i++) {
        Ray r = mkray(v3(src[3 * i], src[3 * i + 1], src[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        Hit a, b; Note note;
        a.tmin = FLT_MAX; a.m = a.t = -1; a.u = a.v = 0; b = a;
        int fa = search_ref(s, &r, &a, &c), fb = search_noted(s, &r, &b, &note);
        if (fa != fb || memcmp(&a, &b, sizeof a) != 0) out[3]++;
        if (!fa) continue;
        out[0]++;
        int again = note.far_max > a.tmin || note.stack_min < a.tmin;
        Hit h2 = a;
        int f2 = search_ref(s, &r, &h2, &c);
        out[1] += again; out[2] += f2;
        if (!again && (f2 || memcmp(&h2, &a, sizeof a) != 0)) out[4]++;
    }
}
