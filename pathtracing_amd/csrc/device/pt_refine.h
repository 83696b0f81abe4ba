// Subdivision of fat kd leaves for the fp32 GI path (see pt_refine.hip).
#pragma once
#include <stdint.h>

#include <vector>

#include "pt_scene_dev.h"

struct RefineParams {
    bool enable = true;
    bool rebuild = true;        // build the whole fp32 tree from the triangles (false: keep the host tree, subdivide its leaves)
    float ct = 1.0f;            // cost of one traversal step
    float ci = 1.2f;            // cost of one triangle test
    float empty_bonus = 1.0f;   // factor on splits that cut off empty space (0.8 was 5 % slower on the atrium, 17 % on car)
    bool clip = true;           // split clipping: a triangle's bounds in a child are those of the part of it inside the child
    int leaf_keep = 2;          // leaves with at most this many triangles are left alone
    int max_depth = 56;         // depth bound, counted from the root of the host tree
};

struct RefineStats {
    uint64_t nodes = 0, refs = 0, leaves = 0, leaf_refs = 0, interior_refs = 0, new_nodes = 0, max_leaf = 0;
    int max_depth = 0;
};

// nodes: the device node array (host nodes keep their indices, new nodes are
// appended in adjacent pairs); refs: triangle indices in node order.
RefineStats refine_tree(const PtSceneDesc *d, const RefineParams &prm, std::vector<NodeF> &nodes, std::vector<int32_t> &refs);

// The built tree (as refine_tree leaves it) as quad records, two levels per record (layout: pt_trace_quad.h).
// Empty when the tree keeps triangles in interior nodes, which quad records cannot express.
std::vector<uint4> quad_records(const std::vector<NodeF> &nodes);
