// Host-side checks of the fp32 tree build (pathtracing_amd/csrc/device/pt_refine.hip), run without a GPU by
// tests/test_refine_host.py:
//   * the quad records (two tree levels per record) describe the same tree as the binary records: the same leaves with
//     the same triangle ranges in the same left-to-right order, every link inside the array;
//   * split clipping never loses a triangle: points sampled on every triangle lie in leaves that list the triangle
//     (checked with and without clipping; the two builds are different trees, since the SAH sees different bounds).
// Prints one line of numbers; exit code 0 = all checks hold.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "pathtracing_amd.h"
#include "pathtracing_host.h"
#include "../pathtracing_amd/csrc/device/pt_refine.h"

static float f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct Leaf { uint32_t first, count; };

static void binary_leaves(const std::vector<NodeF> &n, uint32_t i, std::vector<Leaf> &out) {
    if (n[i].link == 0xFFFFFFFFu) { out.push_back(Leaf{n[i].first, n[i].count}); return; }
    const uint32_t l = n[i].link & 0x3FFFFFFFu;
    binary_leaves(n, l, out);
    binary_leaves(n, l + 1, out);
}

static bool quad_leaves(const std::vector<uint4> &q, uint32_t i, std::vector<Leaf> &out, const std::vector<NodeF> &n, uint32_t ni) {
    const uint4 r = q[i];
    if (r.w == 0xFFFFFFFFu) {
        if (n[ni].link != 0xFFFFFFFFu) return false;
        out.push_back(Leaf{r.x / 48u, (r.y - r.x) / 48u});
        return true;
    }
    if (n[ni].link == 0xFFFFFFFFu) return false;
    const uint32_t base = (r.w & ~63u) / 16u, left = n[ni].link & 0x3FFFFFFFu;
    if ((uint64_t)base + 4u > q.size() || (r.w & 3u) != (n[ni].link >> 30) || f32(r.x) != n[ni].split) return false;
    for (uint32_t h = 0; h < 2; h++) {
        const uint32_t code = (r.w >> (2u + 2u * h)) & 3u, ci = left + h;
        if (code == 3u) {
            if (!quad_leaves(q, base + 2u * h, out, n, ci)) return false;
        } else {
            if (n[ci].link == 0xFFFFFFFFu || code != (n[ci].link >> 30) || f32(h ? r.z : r.y) != n[ci].split) return false;
            const uint32_t cl = n[ci].link & 0x3FFFFFFFu;
            if (!quad_leaves(q, base + 2u * h, out, n, cl) || !quad_leaves(q, base + 2u * h + 1u, out, n, cl + 1u)) return false;
        }
    }
    return true;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    PthScene *hs = nullptr;
    if (pth_scene_load_packed(argv[1], &hs)) { std::printf("load failed: %s\n", pth_last_error()); return 2; }
    PthTreeStats ts;
    if (pth_scene_build_tree(hs, &ts)) return 2;
    PtSceneDesc d;
    if (pth_scene_desc(hs, &d)) return 2;
    int bad = 0;
    for (int clip = 0; clip < 2; clip++) {
        RefineParams p;
        p.clip = clip != 0;
        std::vector<NodeF> nodes;
        std::vector<int32_t> list;
        const RefineStats st = refine_tree(&d, p, nodes, list);
        // quad records against binary records
        const std::vector<uint4> q = quad_records(nodes);
        std::vector<Leaf> lb, lq;
        binary_leaves(nodes, 0, lb);
        const bool walk = !q.empty() && quad_leaves(q, 0, lq, nodes, 0);
        bool same = walk && lb.size() == lq.size();
        for (size_t k = 0; same && k < lb.size(); k++) same = lb[k].first == lq[k].first && lb[k].count == lq[k].count;
        if (!same) { std::printf("clip %d: quad records differ from the binary tree\n", clip); bad++; }
        // every triangle, at points sampled on it, is listed by the leaf the point lies in
        std::mt19937 rng(12345);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        uint64_t lost = 0, tested = 0, near_plane = 0;
        for (uint32_t t = 0; t < d.n_tri; t++) {
            const double *a = d.vert + 3 * (size_t)d.tri_vi[3 * (size_t)t], *b = d.vert + 3 * (size_t)d.tri_vi[3 * (size_t)t + 1],
                         *c = d.vert + 3 * (size_t)d.tri_vi[3 * (size_t)t + 2];
            for (int s = 0; s < 12; s++) {
                double u = U(rng), v = U(rng);
                if (u + v > 1.0) { u = 1.0 - u; v = 1.0 - v; }
                float P[3];
                for (int k = 0; k < 3; k++) P[k] = (float)(a[k] + u * (b[k] - a[k]) + v * (c[k] - a[k]));
                uint32_t i = 0;
                bool close = false;
                while (nodes[i].link != 0xFFFFFFFFu) {
                    const uint32_t ax = nodes[i].link >> 30, l = nodes[i].link & 0x3FFFFFFFu;
                    if (std::fabs(P[ax] - nodes[i].split) <= 1e-5f * (1.0f + std::fabs(nodes[i].split))) close = true;   // either side may hold it
                    i = l + (P[ax] > nodes[i].split ? 1u : 0u);
                }
                if (close) { near_plane++; continue; }
                tested++;
                bool found = false;
                for (uint32_t r = 0; r < nodes[i].count && !found; r++) found = list[nodes[i].first + r] == (int32_t)t;
                if (!found) lost++;
            }
        }
        if (lost) { std::printf("clip %d: %llu of %llu sampled points lie in a leaf that does not list their triangle\n", clip,
                                (unsigned long long)lost, (unsigned long long)tested); bad++; }
        std::printf("clip %d: nodes %llu refs %llu leaves %llu depth %d quads %zu points %llu (+%llu next to a plane)\n", clip,
                    (unsigned long long)st.nodes, (unsigned long long)st.refs, (unsigned long long)st.leaves, st.max_depth, q.size(),
                    (unsigned long long)tested, (unsigned long long)near_plane);
    }
    pth_scene_free(hs);
    return bad ? 1 : 0;
}
