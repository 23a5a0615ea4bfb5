// Host-side SAH k-d tree builder of the engine (product code).
//
// Produces exactly the tree rt::KdTree<Triangle> builds in the reference
// (/root/reference/Source/RT/KdTree.hpp:87-264): same split choices, same leaf contents and
// order, same DFS node numbering, in the reference's 8-byte node encoding
// (Source/RT/KdTree.h:22-65).  The flattened device layout is derived from it in gigl_rt.hip (flatten_nodes).
#pragma once
#include <cstdint>
#include <vector>

namespace gigl {

struct KdNode {          // word0: split_pos bits | single_prim_id | prims_offset
    uint32_t w0, w1;     // word1: bit0 is_leaf; interior: bits1-2 axis, bits3-31 above_child;
};                       //        leaf: bits1-31 n_prims

struct KdBuildParams {   // KdTree ctor arguments, Source/RT/KdTree.h:14-16
    int inters_cost = 10;
    int trav_cost = 1;
    int max_depth = 30;
    uint32_t min_num_prims = 2;
    float empty_bonus = 0.25f;
    int n_threads = 0;   // 0: use hardware concurrency (capped); the tree does not depend on it
};

struct KdTreeHost {
    std::vector<KdNode> nodes;
    std::vector<uint32_t> leaf_prim_ids;
    float box_min[3], box_max[3];
    int max_depth = 0;       // effective (after the "negative -> auto" rule, KdTree.hpp:97-100)
    int actual_depth = 0;    // KdTree.hpp:129
    uint32_t leaf_count = 0;
    double prims_in_leaves = 0;
    bool degenerate_split_guard_hit = false;  // see kd_build.cpp: reference would index out of range
};

// verts: float[3*n_verts]; tris: uint32[4*n_tris] (i0,i1,i2,mat).  Returns false and fills err on
// invalid input (index out of range, non-finite vertex, max_depth > 31).
bool kd_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
              const KdBuildParams& p, KdTreeHost& out, const char** err);

}  // namespace gigl
