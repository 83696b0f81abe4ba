// Device-side kd-tree of the fp32 GI path.
//
// The host hands over the reference's tree (centroid-sweep SAH, axis by depth,
// pull-up; util/kdtree.cpp:88-172).  The fp64 reference-semantics path walks that
// tree node for node, because the reference's results depend on it.  The GI
// sample rays only need the nearest hit, which no kd-tree changes -- a tree
// decides WHICH triangles are tested, not what the nearest one is -- and on the
// GPU the price of a tree is plain: every triangle test is three divergent
// 16-byte loads per lane plus ~30 VALU instructions, every node step one load
// plus ~45.  The reference's tree is expensive in those terms (leaves stop as
// soon as the centroid heuristic sees no gain, so leaves of tens to hundreds of
// triangles are common; the split axis is fixed by depth; triangles pulled up
// into interior nodes are tested by every ray that passes).
//
// Three variants, selected at upload (PT_TUNE_REFINE; kept so that their cost can be compared):
//   0  the host tree as it is;
//   1  the host tree, every leaf with more than a few triangles subdivided;
//   2  (default) a tree built from the triangles: surface-area-heuristic sweep over
//      the triangles' bounds on all three axes, with split clipping.
// Measured on MI355X with the closest-hit kernel as it was when the variants were
// compared, 64 spp: atrium-262k 0.91 / 1.69 / 2.6 Grays/s, car 2.9 / 4.8 / 7.2 Grays/s.
// The build runs on up to 16 host threads (subtrees below a fixed depth are
// independent jobs, spliced in job order, so the tree does not depend on the
// thread count).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "pt_refine.h"

namespace {

struct Box3 { float lo[3], hi[3]; };

inline float area(const Box3 &b) {
    float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return 2.0f * (x * y + y * z + z * x);
}

// A triangle as a node of the build sees it: its index and the bounds of the part of it that lies inside the node's box.
struct Ref { int32_t tri; Box3 b; };

// A subtree put aside by the top of the build, to be built by a worker thread.
struct Job { uint32_t slot; Box3 box; std::vector<Ref> ids; int depth; };

// Sutherland-Hodgman: the part of a convex polygon on one side of an axis-aligned plane.
inline int clip_plane(const double (*in)[3], int n, int axis, double plane, bool keep_above, double (*out)[3]) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        const double *a = in[i], *b = in[(i + 1) % n];
        const bool ia = keep_above ? a[axis] >= plane : a[axis] <= plane, ib = keep_above ? b[axis] >= plane : b[axis] <= plane;
        if (ia) { for (int k = 0; k < 3; k++) out[m][k] = a[k]; m++; }
        if (ia != ib) {
            const double t = (plane - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; k++) out[m][k] = a[k] + t * (b[k] - a[k]);
            out[m][axis] = plane;
            m++;
        }
    }
    return m;
}

struct Builder {
    const PtSceneDesc *d;
    RefineParams prm;
    std::vector<Box3> own_tbox;
    const std::vector<Box3> &tbox;        // per triangle: bounds of its fp32 vertices, slightly widened
    std::vector<NodeF> &nodes;
    std::vector<int32_t> &refs;
    RefineStats st{};
    std::vector<Job> *jobs = nullptr;     // when set: subtrees at `job_depth` with enough triangles are deferred
    int job_depth = 0;

    // worker: shares the triangle bounds of `top`
    Builder(const Builder &top, std::vector<NodeF> &n, std::vector<int32_t> &r)
        : d(top.d), prm(top.prm), tbox(top.tbox), nodes(n), refs(r) {}

    Builder(const PtSceneDesc *desc, const RefineParams &p, std::vector<NodeF> &n, std::vector<int32_t> &r)
        : d(desc), prm(p), tbox(own_tbox), nodes(n), refs(r) {
        std::vector<Box3> &tbox = own_tbox;
        tbox.resize(d->n_tri);
        for (uint32_t t = 0; t < d->n_tri; t++) {
            Box3 b;
            for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
            for (int c = 0; c < 3; c++) {
                const double *v = d->vert + 3 * (size_t)d->tri_vi[3 * (size_t)t + c];
                for (int k = 0; k < 3; k++) {
                    float f = (float)v[k];
                    b.lo[k] = std::min(b.lo[k], f);
                    b.hi[k] = std::max(b.hi[k], f);
                }
            }
            // the kernel intersects v0 + e1, v0 + e2 in fp32: allow for a few ulps of difference
            for (int k = 0; k < 3; k++) {
                float m = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) * 4.8e-7f + 1e-30f;
                b.lo[k] -= m; b.hi[k] += m;
            }
            tbox[t] = b;
        }
    }

    // The allowance for the kernel's fp32 arithmetic (it intersects v0 + e1, v0 + e2 rounded to fp32): a few ulps of the
    // largest coordinate involved.
    static float slack(float lo, float hi) { return std::max(std::fabs(lo), std::fabs(hi)) * 4.8e-7f + 1e-30f; }

    // Bounds of the part of triangle `t` inside `box` ("split clipping": the triangle itself is cut by the box, in
    // fp64, not its bounding box -- a slanted triangle's bounding box reaches into cells the triangle never enters, and
    // every such cell would list it and test it).  The box is widened by the fp32 allowance before the cut and the
    // result widened again, so a cell never loses a triangle that the kernel's arithmetic could still hit inside it.
    // False: no part of the triangle is inside.
    bool clipped_bounds(int32_t t, const Box3 &box, Box3 &out) const {
        double poly[2][12][3];
        int n = 3, cur = 0;
        for (int c = 0; c < 3; c++) {
            const double *v = d->vert + 3 * (size_t)d->tri_vi[3 * (size_t)t + c];
            for (int k = 0; k < 3; k++) poly[0][c][k] = v[k];
        }
        for (int k = 0; k < 3 && n; k++) {
            const float m = slack(box.lo[k], box.hi[k]);
            n = clip_plane(poly[cur], n, k, (double)box.lo[k] - (double)m, true, poly[cur ^ 1]); cur ^= 1;
            if (!n) break;
            n = clip_plane(poly[cur], n, k, (double)box.hi[k] + (double)m, false, poly[cur ^ 1]); cur ^= 1;
        }
        if (!n) return false;
        for (int k = 0; k < 3; k++) {
            double lo = poly[cur][0][k], hi = lo;
            for (int i = 1; i < n; i++) { lo = std::min(lo, poly[cur][i][k]); hi = std::max(hi, poly[cur][i][k]); }
            float fl = std::nextafter((float)lo, -INFINITY), fh = std::nextafter((float)hi, INFINITY);
            const float m = slack(fl, fh);
            out.lo[k] = std::max(fl - m, box.lo[k]);
            out.hi[k] = std::min(fh + m, box.hi[k]);
            if (out.lo[k] > out.hi[k]) { out.lo[k] = out.hi[k] = std::min(std::max(fl, box.lo[k]), box.hi[k]); }
        }
        return true;
    }

    // The triangles `tris` as references inside `box` (bounding boxes cut by the box; the exact cut comes with the splits).
    std::vector<Ref> make_refs(const int32_t *tris, size_t n, const Box3 &box) const {
        std::vector<Ref> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i].tri = tris[i];
            for (int k = 0; k < 3; k++) {
                out[i].b.lo[k] = std::max(tbox[tris[i]].lo[k], box.lo[k]);
                out[i].b.hi[k] = std::min(tbox[tris[i]].hi[k], box.hi[k]);
            }
        }
        return out;
    }

    // Emit node `slot` for the triangles `ids` inside `box`; children go to fresh adjacent slots.
    void build(uint32_t slot, const Box3 &box, std::vector<Ref> &ids, int depth) {
        const size_t n = ids.size();
        if (jobs && depth == job_depth && n > 4096) {
            jobs->push_back(Job{slot, box, std::move(ids), depth});
            return;
        }
        int best_axis = -1;
        float best_plane = 0, best_cost = prm.ci * (float)n;
        if (n > (size_t)prm.leaf_keep && depth < prm.max_depth) {
            const float sa = area(box);
            std::vector<float> los(n), his(n);
            for (int a = 0; a < 3 && sa > 0; a++) {
                const float blo = box.lo[a], bhi = box.hi[a];
                if (!(bhi > blo)) continue;
                for (size_t i = 0; i < n; i++) {
                    los[i] = std::max(ids[i].b.lo[a], blo);
                    his[i] = std::min(ids[i].b.hi[a], bhi);
                }
                std::sort(los.begin(), los.end());
                std::sort(his.begin(), his.end());
                Box3 lb = box, rb = box;
                // candidate planes: every clipped bound strictly inside the box
                size_t il = 0, ih = 0;
                auto consider = [&](float p) {
                    if (!(p > blo && p < bhi)) return;
                    // left gets triangles starting below p, right those ending above p
                    while (il < n && los[il] < p) il++;
                    while (ih < n && his[ih] <= p) ih++;
                    size_t nl = il, nr = n - ih;
                    lb.hi[a] = p; rb.lo[a] = p;
                    float c = prm.ct + prm.ci * (area(lb) * (float)nl + area(rb) * (float)nr) / sa;
                    if (nl == 0 || nr == 0) c *= prm.empty_bonus;
                    if (c < best_cost) { best_cost = c; best_axis = a; best_plane = p; }
                };
                // merged ascending walk over both bound lists
                size_t a_i = 0, b_i = 0;
                while (a_i < n || b_i < n) {
                    float p = (b_i >= n || (a_i < n && los[a_i] <= his[b_i])) ? los[a_i++] : his[b_i++];
                    consider(p);
                }
            }
        }
        NodeF &o = nodes[slot];
        if (best_axis < 0) {
            o.split = 0; o.link = 0xFFFFFFFFu;
            o.first = (uint32_t)refs.size(); o.count = (uint32_t)n;
            for (const Ref &r : ids) refs.push_back(r.tri);
            st.leaves++; st.leaf_refs += n;
            st.max_leaf = std::max<uint64_t>(st.max_leaf, n);
            st.max_depth = std::max(st.max_depth, depth);
            return;
        }
        Box3 lb = box, rb = box;
        lb.hi[best_axis] = best_plane; rb.lo[best_axis] = best_plane;
        std::vector<Ref> L, R;
        for (const Ref &t : ids) {
            const float lo = std::max(t.b.lo[best_axis], box.lo[best_axis]), hi = std::min(t.b.hi[best_axis], box.hi[best_axis]);
            bool l = lo < best_plane, r = hi > best_plane;
            if (!l && !r) l = true;                       // flat in the plane
            if (l && r && prm.clip) {
                // on both sides by its bounds: each child gets the bounds of the part that is really inside it, or nothing
                Ref a = t, b = t;
                const bool in_l = clipped_bounds(t.tri, lb, a.b), in_r = clipped_bounds(t.tri, rb, b.b);
                if (in_l || in_r) {
                    if (in_l) L.push_back(a);
                    if (in_r) R.push_back(b);
                    continue;
                }
            }
            Ref c = t;
            if (l) { c.b.hi[best_axis] = std::min(c.b.hi[best_axis], best_plane); L.push_back(c); c = t; }
            if (r) { c.b.lo[best_axis] = std::max(c.b.lo[best_axis], best_plane); R.push_back(c); }
        }
        std::vector<Ref>().swap(ids);
        uint32_t left = (uint32_t)nodes.size();
        nodes.push_back(NodeF{});
        nodes.push_back(NodeF{});
        NodeF &o2 = nodes[slot];                          // nodes may have moved
        o2.split = best_plane;
        o2.link = left | ((uint32_t)best_axis << 30);
        o2.first = (uint32_t)refs.size(); o2.count = 0;
        st.new_nodes += 2;
        build(left, lb, L, depth + 1);
        build(left + 1, rb, R, depth + 1);
    }
};

// Build the subtrees the top of the build put aside, on up to 16 threads, and splice them in in job order
// (the result does not depend on the number of threads).
void run_jobs(Builder &top, std::vector<Job> &jobs) {
    struct Out { std::vector<NodeF> nodes; std::vector<int32_t> refs; RefineStats st{}; };
    std::vector<Out> outs(jobs.size());
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t j = next++; j < jobs.size(); j = next++) {
            outs[j].nodes.assign(1, NodeF{});
            Builder b(top, outs[j].nodes, outs[j].refs);
            b.build(0, jobs[j].box, jobs[j].ids, jobs[j].depth);
            outs[j].st = b.st;
        }
    };
    unsigned nt = std::min<unsigned>({16u, std::max(1u, std::thread::hardware_concurrency()), (unsigned)jobs.size()});
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (size_t j = 0; j < jobs.size(); j++) {
        Out &o = outs[j];
        const uint32_t node_off = (uint32_t)top.nodes.size() - 1u, ref_off = (uint32_t)top.refs.size();
        for (size_t k = 0; k < o.nodes.size(); k++) {
            NodeF n = o.nodes[k];
            if (n.link != 0xFFFFFFFFu) n.link = ((n.link & 0x3FFFFFFFu) + node_off) | (n.link & 0xC0000000u);
            n.first += ref_off;
            if (k == 0) top.nodes[jobs[j].slot] = n; else top.nodes.push_back(n);
        }
        top.refs.insert(top.refs.end(), o.refs.begin(), o.refs.end());
        top.st.leaves += o.st.leaves; top.st.leaf_refs += o.st.leaf_refs; top.st.new_nodes += o.st.new_nodes;
        top.st.max_leaf = std::max(top.st.max_leaf, o.st.max_leaf);
        top.st.max_depth = std::max(top.st.max_depth, o.st.max_depth);
    }
}

// Re-number the nodes so that every pair of children starts at an even index (32-byte aligned in
// the encoded array): the traversal then gets the far child as near ^ 16.  Depth-first, a pair
// placed when its parent is reached.
void align_pairs(std::vector<NodeF> &nodes) {
    std::vector<NodeF> out;
    out.reserve(nodes.size() + 2);
    out.push_back(nodes[0]);
    out.push_back(NodeF{0.f, 0xFFFFFFFFu, 0u, 0u});            // never referenced
    std::vector<std::pair<uint32_t, uint32_t>> todo;          // (old index, new index)
    todo.push_back({0u, 0u});
    while (!todo.empty()) {
        auto [o, n] = todo.back();
        todo.pop_back();
        const NodeF src = nodes[o];
        if (src.link == 0xFFFFFFFFu) continue;
        const uint32_t old_left = src.link & 0x3FFFFFFFu, axis = src.link >> 30, new_left = (uint32_t)out.size();
        out.push_back(nodes[old_left]);
        out.push_back(nodes[old_left + 1]);
        out[n].link = new_left | (axis << 30);
        todo.push_back({old_left + 1, new_left + 1});
        todo.push_back({old_left, new_left});
    }
    nodes.swap(out);
}

}  // namespace

RefineStats refine_tree(const PtSceneDesc *d, const RefineParams &prm, std::vector<NodeF> &nodes, std::vector<int32_t> &refs) {
    nodes.assign(d->n_node, NodeF{});
    refs.clear();
    refs.reserve((size_t)d->n_ref * 2);
    Builder b(d, prm, nodes, refs);
    // depth of every host node, for the depth bound of the subdivision
    std::vector<int> depth(d->n_node, 0);
    std::vector<uint32_t> order;
    order.push_back(0);
    for (size_t q = 0; q < order.size(); q++) {
        const PtKdNode &n = d->nodes[order[q]];
        if (n.axis >= 0) {
            depth[n.left] = depth[n.left + 1] = depth[order[q]] + 1;
            order.push_back((uint32_t)n.left);
            order.push_back((uint32_t)n.left + 1);
        }
    }
    if (prm.rebuild) {
        // Experiment: ignore the host tree and build the whole tree with the sweep above.
        nodes.assign(1, NodeF{});
        std::vector<int32_t> every(d->n_tri);
        for (uint32_t t = 0; t < d->n_tri; t++) every[t] = (int32_t)t;
        Box3 box;
        for (int c = 0; c < 3; c++) { box.lo[c] = (float)d->nodes[0].lo[c]; box.hi[c] = (float)d->nodes[0].hi[c]; }
        std::vector<Ref> all = b.make_refs(every.data(), every.size(), box);
        std::vector<Job> jobs;
        b.jobs = &jobs; b.job_depth = 6;
        b.build(0, box, all, 0);
        b.jobs = nullptr;
        run_jobs(b, jobs);
        align_pairs(nodes);
        b.st.nodes = nodes.size();
        b.st.refs = refs.size();
        return b.st;
    }
    for (uint32_t k : order) {
        const PtKdNode &n = d->nodes[k];
        if (n.axis >= 0) {                                 // interior node of the host tree: copied
            NodeF &o = nodes[k];
            o.split = (float)n.split;
            o.link = (uint32_t)n.left | ((uint32_t)n.axis << 30);
            o.first = (uint32_t)refs.size(); o.count = (uint32_t)n.count;
            refs.insert(refs.end(), d->refs + n.first, d->refs + n.first + n.count);
            b.st.interior_refs += (uint64_t)n.count;
            continue;
        }
        Box3 box;
        for (int c = 0; c < 3; c++) { box.lo[c] = (float)n.lo[c]; box.hi[c] = (float)n.hi[c]; }
        std::vector<Ref> ids = b.make_refs(d->refs + n.first, (size_t)n.count, box);
        if (!prm.enable) {
            NodeF &o = nodes[k];
            o.split = 0; o.link = 0xFFFFFFFFu; o.first = (uint32_t)refs.size(); o.count = (uint32_t)ids.size();
            refs.insert(refs.end(), d->refs + n.first, d->refs + n.first + n.count);
            b.st.leaves++; b.st.leaf_refs += ids.size();
            b.st.max_leaf = std::max<uint64_t>(b.st.max_leaf, ids.size());
            b.st.max_depth = std::max(b.st.max_depth, depth[k]);
            continue;
        }
        b.build(k, box, ids, depth[k]);
    }
    align_pairs(nodes);
    b.st.nodes = nodes.size();
    b.st.refs = refs.size();
    return b.st;
}

std::vector<uint4> quad_records(const std::vector<NodeF> &nodes) {
    std::vector<uint4> q;
    for (const NodeF &n : nodes)
        if (n.link != 0xFFFFFFFFu && n.count != 0) return q;             // triangles in an interior node
    const uint4 unused = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);              // an empty leaf; never referenced
    auto leaf_record = [](const NodeF &n) { return make_uint4(n.first * 48u, (n.first + n.count) * 48u, 0u, 0xFFFFFFFFu); };
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    q.assign(4, unused);                                                  // the root's record, padded to a line
    std::vector<std::pair<uint32_t, uint32_t>> todo;                      // (node, record to fill); depth first
    todo.push_back({0u, 0u});
    while (!todo.empty()) {
        const auto [ni, slot] = todo.back();
        todo.pop_back();
        const NodeF n = nodes[ni];
        if (n.link == 0xFFFFFFFFu) { q[slot] = leaf_record(n); continue; }
        const uint32_t left = n.link & 0x3FFFFFFFu, base = (uint32_t)q.size();     // a multiple of four: 64-byte aligned
        q.resize(q.size() + 4, unused);
        uint32_t code[2];
        float split[2];
        for (int h = 1; h >= 0; h--) {                                    // right half first, so that the left subtree is laid out first
            const NodeF c = nodes[left + (uint32_t)h];
            if (c.link == 0xFFFFFFFFu) {
                code[h] = 3u; split[h] = 0.f;
                q[base + 2u * (uint32_t)h] = leaf_record(c);
            } else {
                code[h] = c.link >> 30; split[h] = c.split;
                const uint32_t cl = c.link & 0x3FFFFFFFu;
                todo.push_back({cl + 1u, base + 2u * (uint32_t)h + 1u});
                todo.push_back({cl, base + 2u * (uint32_t)h});
            }
        }
        q[slot] = make_uint4(bits(n.split), bits(split[0]), bits(split[1]), (base * 16u) | (code[1] << 4) | (code[0] << 2) | (n.link >> 30));
    }
    // every link must stay inside the array (the kernel follows them unchecked)
    for (const uint4 &r : q)
        if (r.w != 0xFFFFFFFFu && (uint64_t)(r.w & ~63u) + 64u > (uint64_t)q.size() * 16u) return std::vector<uint4>();
    return q;
}
