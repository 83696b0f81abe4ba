// Host kd-tree build and scene flattening.
//
// The tree is the reference's: a centroid-sweep SAH split per node with the
// axis cycling by depth, triangles assigned by vertex extent, and after both
// children are built the triangles they have in common are pulled up into the
// parent (util/kdtree.cpp:88-172, buildTree :262-281).  It is produced directly
// as an index-linked array whose sibling pairs are adjacent, which is what the
// device consumes.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <deque>

#include "scene.hpp"

namespace pth {

namespace {

constexpr int kLeafMax = 3;          // LEAF, util/kdtree.cpp:14
constexpr double kTrav = 5.0;        // util/kdtree.cpp:68
constexpr double kInst = 5.0;        // util/kdtree.cpp:69

struct Box { double lo[3], hi[3]; };

double area(const Box &b) {           // util/kdtree.cpp:54-59
    double l = b.hi[0] - b.lo[0], w = b.hi[1] - b.lo[1], h = b.hi[2] - b.lo[2];
    return 2.0 * (l * w + w * h + l * h);
}

struct Work {
    Box box;
    double split = 0;
    int axis = -1;
    int left = -1, right = -1;       // indices into Builder::pool
    std::vector<int32_t> ids;        // global triangle ids currently held
};

struct Builder {
    const Flat &g;
    std::vector<double> cent;        // 3 per triangle, util/reader.cpp:113
    std::deque<Work> pool;
    PthTreeStats st{};

    explicit Builder(const Flat &flat) : g(flat) {
        size_t n = g.tri_mesh.size();
        cent.resize(3 * n);
        for (size_t t = 0; t < n; t++) {
            const double *a = &g.vert[3 * g.tri_vi[3 * t]], *b = &g.vert[3 * g.tri_vi[3 * t + 1]],
                         *c = &g.vert[3 * g.tri_vi[3 * t + 2]];
            for (int k = 0; k < 3; k++) cent[3 * t + k] = ((a[k] + b[k]) + c[k]) / 3.0;
        }
    }

    void leaf(int idx, int depth) {
        (void)depth;
        st.naive += (int64_t)pool[idx].ids.size();
        st.leaves++;
        st.max_leaf = std::max<int64_t>(st.max_leaf, (int64_t)pool[idx].ids.size());
    }

    void build(int idx, int depth) {
        st.nodes++;
        st.max_depth = std::max<int64_t>(st.max_depth, depth);
        const size_t n = pool[idx].ids.size();
        if (n <= (size_t)kLeafMax) return leaf(idx, depth);
        const int axis = depth % 3;
        // candidate planes: the triangle centroids in ascending order along the axis
        std::vector<double> cand(n);
        for (size_t i = 0; i < n; i++) cand[i] = cent[3 * (size_t)pool[idx].ids[i] + axis];
        std::sort(cand.begin(), cand.end());
        const Box box = pool[idx].box;
        Box lb = box, rb = box;
        const double sa = area(box);
        double best = DBL_MAX, split = 0;
        for (size_t i = 0; i < n; i++) {
            lb.hi[axis] = cand[i];
            rb.lo[axis] = cand[i];
            double lsa = area(lb), rsa = area(rb);
            double cost = kTrav + lsa / sa * (double)(int)(i + 1) * kInst + rsa / sa * (double)(n - i - 1) * kInst;
            if (cost < best) { best = cost; split = cand[i]; }
        }
        if (kInst * (double)n <= best) return leaf(idx, depth);
        lb.hi[axis] = split;
        rb.lo[axis] = split;
        Work L, R;
        L.box = lb; R.box = rb;
        for (int32_t t : pool[idx].ids) {
            bool l = false, r = false;
            for (int k = 0; k < 3; k++) {
                double c = g.vert[3 * (size_t)g.tri_vi[3 * (size_t)t + k] + axis];
                l = l || c <= split;
                r = r || c >= split;
            }
            if (l) L.ids.push_back(t);
            if (r) R.ids.push_back(t);
        }
        if (std::max(L.ids.size(), R.ids.size()) == n) return leaf(idx, depth);
        const int li = (int)pool.size();
        pool.push_back(std::move(L));
        const int ri = (int)pool.size();
        pool.push_back(std::move(R));
        pool[idx].axis = axis;
        pool[idx].split = split;
        pool[idx].left = li;
        pool[idx].right = ri;
        build(li, depth + 1);
        build(ri, depth + 1);
        // Pull-up: what both children still hold moves here.  The left child keeps
        // its order; the right child ends up in ascending (mesh, triangle) order,
        // which is ascending global id.
        std::vector<int32_t> rs = pool[ri].ids;
        std::sort(rs.begin(), rs.end());
        rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
        std::vector<char> taken(rs.size(), 0);
        std::vector<int32_t> up, lkeep;
        for (int32_t t : pool[li].ids) {
            auto it = std::lower_bound(rs.begin(), rs.end(), t);
            size_t k = (size_t)(it - rs.begin());
            if (it != rs.end() && *it == t && !taken[k]) { taken[k] = 1; up.push_back(t); }
            else lkeep.push_back(t);
        }
        std::vector<int32_t> rkeep;
        for (size_t k = 0; k < rs.size(); k++) if (!taken[k]) rkeep.push_back(rs[k]);
        pool[idx].ids = std::move(up);
        pool[li].ids = std::move(lkeep);
        pool[ri].ids = std::move(rkeep);
        st.optimized += (int64_t)(pool[li].ids.size() + pool[ri].ids.size());
    }
};

}  // namespace

// Concatenate the meshes into global pools (indices rebased, texture indices
// made global) -- the layout PtSceneDesc describes.
void flatten_geometry(const Scene &sc, Flat &f) {
    f = Flat();
    int32_t vb = 0, nb = 0, ub = 0, tb = 0, gb = 0;
    for (size_t m = 0; m < sc.meshes.size(); m++) {
        const Mesh &me = sc.meshes[m];
        f.mesh_tri_base.push_back(gb);
        for (int k = 0; k < 3; k++) f.mesh_albedo.push_back(me.albedo[k]);
        for (const D3 &v : me.vert) f.vert.insert(f.vert.end(), v.begin(), v.end());
        for (const D3 &v : me.norm) f.norm.insert(f.norm.end(), v.begin(), v.end());
        for (const auto &v : me.uv) f.uv.insert(f.uv.end(), v.begin(), v.end());
        for (size_t t = 0; t < me.tris.size(); t++) {
            const Triangle &tr = me.tris[t];
            for (int k = 0; k < 3; k++) {
                f.tri_vi.push_back(tr.vi[k] + vb);
                f.tri_ni.push_back(tr.ni[k] < 0 ? -1 : tr.ni[k] + nb);
                f.tri_ti.push_back(tr.ti[k] < 0 ? -1 : tr.ti[k] + ub);
            }
            int32_t tm = tr.tm;
            if (tm != -99999) tm = tm >= 0 ? tm + tb : -((-(tm + 1)) + tb) - 1;
            f.tri_tm.push_back(tm);
            f.tri_reft.push_back(tr.reft);
            f.tri_mesh.push_back((int32_t)m);
            f.tri_local.push_back((int32_t)t);
        }
        for (const Texture &tx : me.tmaps) f.tex.push_back(PtTexture{tx.rows, tx.cols, tx.bgr.data()});
        vb += (int32_t)me.vert.size(); nb += (int32_t)me.norm.size(); ub += (int32_t)me.uv.size();
        tb += (int32_t)me.tmaps.size(); gb += (int32_t)me.tris.size();
    }
    for (const Light &L : sc.lights) {
        PtLight pl{};
        pl.type = L.type; pl.mag = L.mag;
        for (int k = 0; k < 3; k++) { pl.pos[k] = L.pos[k]; pl.color[k] = L.color[k]; }
        f.lights.push_back(pl);
    }
}

void build_tree(const Scene &sc, Flat &f, PthTreeStats &stats) {
    auto t0 = std::chrono::steady_clock::now();
    flatten_geometry(sc, f);
    Builder b(f);
    Work root;
    // Root box: extent of all vertices and the origin, padded by 0.01 (util/kdtree.cpp:270-278).
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (size_t i = 0; i < f.vert.size() / 3; i++)
        for (int k = 0; k < 3; k++) {
            mn[k] = std::min(mn[k], f.vert[3 * i + k]);
            mx[k] = std::max(mx[k], f.vert[3 * i + k]);
        }
    for (int k = 0; k < 3; k++) { root.box.lo[k] = mn[k] - 0.01; root.box.hi[k] = mx[k] + 0.01; }
    root.ids.resize(f.tri_mesh.size());
    for (size_t t = 0; t < root.ids.size(); t++) root.ids[t] = (int32_t)t;
    b.pool.push_back(std::move(root));
    b.st.triangles = (int64_t)f.tri_mesh.size();
    b.build(0, 0);
    b.st.root_entries = (int64_t)b.pool[0].ids.size();

    // Emit: node 0 = root, every interior node's children in adjacent slots,
    // left subtree laid out before the right one.
    f.nodes.assign(b.pool.size(), PtKdNode{});
    f.refs.clear();
    std::vector<std::pair<int, int>> todo;   // (pool index, output slot)
    todo.push_back({0, 0});
    int next = 1;
    while (!todo.empty()) {
        auto [pi, slot] = todo.back();
        todo.pop_back();
        const Work &w = b.pool[pi];
        PtKdNode &o = f.nodes[slot];
        for (int k = 0; k < 3; k++) { o.lo[k] = w.box.lo[k]; o.hi[k] = w.box.hi[k]; }
        o.split = w.left < 0 ? 0.0 : w.split;
        o.axis = w.left < 0 ? -1 : w.axis;
        o.first = (int32_t)f.refs.size();
        o.count = (int32_t)w.ids.size();
        f.refs.insert(f.refs.end(), w.ids.begin(), w.ids.end());
        if (w.left >= 0) {
            o.left = next;
            todo.push_back({w.right, next + 1});   // popped second
            todo.push_back({w.left, next});        // popped first
            next += 2;
        } else {
            o.left = -1;
        }
    }
    f.built = true;
    b.st.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats = b.st;
}

}  // namespace pth
