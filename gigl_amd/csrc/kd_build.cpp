// Host-side SAH k-d tree builder (product code).  See kd_build.h.
//
// The split search follows /root/reference/Source/RT/KdTree.hpp:139-264 decision for decision so
// that the tree is the reference's tree.  The reference recurses serially and sorts the three axes
// on three OpenMP threads (:166); here whole subtrees near the root are built as independent tasks
// into private node arrays and spliced back in DFS order, which yields the same node numbering
// (bottom child = parent + 1, Source/RT/KdTree.hpp:254-262) while using every host core.
//
// Bounding-edge ties are ordered by std::sort exactly as in the reference (:179); the tie order
// decides which coplanar primitives fall below or above a split (:229-247), so the same
// comparator on the same input order is part of the contract.
#include "kd_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <thread>

namespace gigl {
namespace {

struct Box3 {
    float mn[3], mx[3];
};

// Select-style min/max (GLM 0.9.5.4 func_common.inl:377,395), as BBox::extend uses them.
inline float sel_min(float x, float y) { return x < y ? x : y; }
inline float sel_max(float x, float y) { return x > y ? x : y; }

struct BoundEdge {  // Source/RT/KdTree.h:67-77
    enum Type : int { END = 0, START = 1, BOTH = 2, NONE = 3 };
    float axis_pos;
    uint32_t prim_id;
    Type type;
    bool operator<(const BoundEdge& r) const {  // KdTree.hpp:73-80
        if (axis_pos != r.axis_pos) return axis_pos < r.axis_pos;
        return type < r.type;
    }
};

struct Subtree {
    std::vector<KdNode> nodes;
    std::vector<uint32_t> leaf_ids;
    int actual_depth = 0;
    uint32_t leaf_count = 0;
    double prims_in_leaves = 0;
    bool guard_hit = false;
};

struct Shared {
    const Box3* prim_boxes;
    KdBuildParams p;
    int max_depth;
    int spawn_depth;        // subtrees rooted above this depth are built as parallel tasks
    uint32_t spawn_min_prims;
};

struct Scratch {
    std::vector<BoundEdge> edges[3];
    std::vector<uint32_t> prims_bottom, prims_top;
    void size_for(uint32_t n, int levels) {
        for (int a = 0; a < 3; ++a) edges[a].resize(2 * (size_t)n);
        prims_bottom.resize(n);
        prims_top.resize((size_t)n * (size_t)(levels + 1));
    }
};

inline KdNode make_leaf(uint32_t n_prims, const uint32_t* ids, Subtree& st) {  // KdTree.hpp:14-30
    KdNode n;
    n.w1 = 1u | (n_prims << 1);
    if (n_prims == 0) {
        n.w0 = 0;
    } else if (n_prims == 1) {
        n.w0 = ids[0];
    } else {
        n.w0 = (uint32_t)st.leaf_ids.size();
        st.leaf_ids.insert(st.leaf_ids.end(), ids, ids + n_prims);
    }
    st.prims_in_leaves += n_prims;
    ++st.leaf_count;
    return n;
}

inline float box_area(const Box3& b) {  // BBox::computeArea, Source/Common/BBox.cpp:57-60
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2.0f * (dx * dy + (dx + dy) * dz);
}

void build_task(const Shared& sh, Subtree& st, const Box3& node_box, std::vector<uint32_t>&& prims,
                int depth, int bad_refines);

// Splices a finished child subtree behind the nodes already in `st`.
void splice(Subtree& st, const Subtree& c) {
    const uint32_t node_base = (uint32_t)st.nodes.size();
    const uint32_t leaf_base = (uint32_t)st.leaf_ids.size();
    st.nodes.reserve(st.nodes.size() + c.nodes.size());
    for (const KdNode& n : c.nodes) {
        KdNode m = n;
        if (m.w1 & 1u) {
            if ((m.w1 >> 1) >= 2) m.w0 += leaf_base;
        } else {
            m.w1 = (m.w1 & 7u) | (((m.w1 >> 3) + node_base) << 3);
        }
        st.nodes.push_back(m);
    }
    st.leaf_ids.insert(st.leaf_ids.end(), c.leaf_ids.begin(), c.leaf_ids.end());
    st.actual_depth = std::max(st.actual_depth, c.actual_depth);
    st.leaf_count += c.leaf_count;
    st.prims_in_leaves += c.prims_in_leaves;
    st.guard_hit = st.guard_hit || c.guard_hit;
}

// Serial recursion inside one task; node ids are local to `st`.  KdTree.hpp:139-264.
void build_rec(const Shared& sh, Subtree& st, Scratch& sc, const Box3& node_box, const uint32_t* overlap,
               uint32_t n_prims, int depth, int bad_refines, uint32_t* prims_bottom, uint32_t* prims_top) {
    const uint32_t node_id = (uint32_t)st.nodes.size();
    st.nodes.push_back(KdNode{0, 1});
    st.actual_depth = std::max(st.actual_depth, depth + 1);
    if (n_prims <= sh.p.min_num_prims || depth == sh.max_depth) {
        st.nodes[node_id] = make_leaf(n_prims, overlap, st);
        return;
    }
    float best_cost = FLT_MAX;
    uint32_t best_axis = UINT32_MAX, best_edge_id = UINT32_MAX;
    const float inv_total_sa = 1.0f / box_area(node_box);
    const float diag[3] = {node_box.mx[0] - node_box.mn[0], node_box.mx[1] - node_box.mn[1],
                           node_box.mx[2] - node_box.mn[2]};
    auto sort_axis = [&](int ax) {
        BoundEdge* E = sc.edges[ax].data();
        for (uint32_t i = 0; i < n_prims; ++i) {
            const uint32_t prim_id = overlap[i];
            const Box3& pb = sh.prim_boxes[prim_id];
            const bool is_flat = pb.mn[ax] == pb.mx[ax];
            E[2 * i] = BoundEdge{pb.mn[ax], prim_id, is_flat ? BoundEdge::BOTH : BoundEdge::START};
            E[2 * i + 1] = BoundEdge{pb.mx[ax], prim_id, is_flat ? BoundEdge::NONE : BoundEdge::END};
        }
        std::sort(E, E + 2 * (size_t)n_prims);
    };
    if (n_prims >= 65536 && sh.p.n_threads > 1) {  // the reference's 3-thread axis sort (:166)
        std::thread t1(sort_axis, 1), t2(sort_axis, 2);
        sort_axis(0);
        t1.join();
        t2.join();
    } else {
        for (int ax = 0; ax < 3; ++ax) sort_axis(ax);
    }
    for (int axis = 0; axis < 3; ++axis) {  // :181-215
        const BoundEdge* E = sc.edges[axis].data();
        uint32_t n_below = 0, n_above = n_prims;
        const float bmin = node_box.mn[axis], bmax = node_box.mx[axis];
        const float d2 = diag[(axis + 1) % 3], d3 = diag[(axis + 2) % 3];
        for (uint32_t i = 0; i < 2 * n_prims; ++i) {
            switch (E[i].type) {
                case BoundEdge::BOTH: ++n_below;  // fallthrough
                case BoundEdge::END: --n_above;   // fallthrough
                case BoundEdge::START: break;
                case BoundEdge::NONE: continue;
            }
            const float edge_pos = E[i].axis_pos;
            if (edge_pos > bmin && edge_pos < bmax) {
                const float below_sa = 2 * (d2 * d3 + (edge_pos - bmin) * (d2 + d3));
                const float above_sa = 2 * (d2 * d3 + (bmax - edge_pos) * (d2 + d3));
                const float bns = (0 == n_above || 0 == n_below) ? sh.p.empty_bonus : 0.0f;
                const float cost = (1.0f - bns) * (n_below * below_sa + n_above * above_sa);
                if (cost < best_cost) {
                    best_axis = (uint32_t)axis;
                    best_cost = cost;
                    best_edge_id = i;
                }
            }
            if (BoundEdge::START == E[i].type) ++n_below;
        }
    }
    best_cost = sh.p.trav_cost + (sh.p.inters_cost * inv_total_sa) * best_cost;  // :218
    const uint32_t old_cost = (uint32_t)sh.p.inters_cost * n_prims;
    if (best_cost > old_cost) ++bad_refines;
    // :221.  With no bounding edge strictly inside the node box and n_prims >= 16 the reference
    // goes on to index m_edges[UINT32_MAX]; the engine makes a leaf instead and records it.
    const bool no_split = best_axis == UINT32_MAX;
    if ((best_cost > 4.0f * old_cost && n_prims < 16) || 3 == bad_refines || no_split) {
        if (no_split && !((best_cost > 4.0f * old_cost && n_prims < 16) || 3 == bad_refines)) st.guard_hit = true;
        st.nodes[node_id] = make_leaf(n_prims, overlap, st);
        return;
    }
    const BoundEdge* E = sc.edges[best_axis].data();
    uint32_t n_bottom = 0, n_top = 0;  // :229-247
    for (uint32_t k = 0; k < best_edge_id; ++k)
        if (E[k].type == BoundEdge::START || E[k].type == BoundEdge::BOTH) prims_bottom[n_bottom++] = E[k].prim_id;
    if (E[best_edge_id].type == BoundEdge::BOTH) prims_bottom[n_bottom++] = E[best_edge_id].prim_id;
    for (uint32_t k = best_edge_id + 1; k < 2 * n_prims; ++k)
        if (E[k].type == BoundEdge::END || E[k].type == BoundEdge::BOTH) prims_top[n_top++] = E[k].prim_id;
    const float split_pos = E[best_edge_id].axis_pos;
    Box3 bottom_box = node_box, top_box = node_box;
    bottom_box.mx[best_axis] = split_pos;
    top_box.mn[best_axis] = split_pos;

    const bool spawn = depth < sh.spawn_depth && n_prims >= sh.spawn_min_prims;
    if (spawn) {
        // Children become independent tasks with private scratch; spliced back bottom-first.
        std::vector<uint32_t> bl(prims_bottom, prims_bottom + n_bottom), tl(prims_top, prims_top + n_top);
        Subtree sb, stp;
        std::thread tb([&]() { build_task(sh, sb, bottom_box, std::move(bl), depth + 1, bad_refines); });
        build_task(sh, stp, top_box, std::move(tl), depth + 1, bad_refines);
        tb.join();
        splice(st, sb);
        const uint32_t top_child_id = (uint32_t)st.nodes.size();
        uint32_t w0;
        memcpy(&w0, &split_pos, 4);
        st.nodes[node_id] = KdNode{w0, (best_axis << 1) | (top_child_id << 3)};  // KdTree.hpp:32-37
        splice(st, stp);
        return;
    }
    build_rec(sh, st, sc, bottom_box, prims_bottom, n_bottom, depth + 1, bad_refines, prims_bottom,
              prims_top + n_prims);
    const uint32_t top_child_id = (uint32_t)st.nodes.size();
    uint32_t w0;
    memcpy(&w0, &split_pos, 4);
    st.nodes[node_id] = KdNode{w0, (best_axis << 1) | (top_child_id << 3)};
    build_rec(sh, st, sc, top_box, prims_top, n_top, depth + 1, bad_refines, prims_bottom, prims_top + n_prims);
}

void build_task(const Shared& sh, Subtree& st, const Box3& node_box, std::vector<uint32_t>&& prims, int depth,
                int bad_refines) {
    Scratch sc;
    const uint32_t n = (uint32_t)prims.size();
    sc.size_for(n, sh.max_depth - depth);
    std::copy(prims.begin(), prims.end(), sc.prims_bottom.begin());
    std::vector<uint32_t>().swap(prims);
    build_rec(sh, st, sc, node_box, sc.prims_bottom.data(), n, depth, bad_refines, sc.prims_bottom.data(),
              sc.prims_top.data());
}

}  // namespace

bool kd_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
              const KdBuildParams& p_in, KdTreeHost& out, const char** err) {
    static const char* e_idx = "triangle vertex index out of range";
    static const char* e_fin = "vertex coordinate is not finite";
    static const char* e_depth = "max_depth above 31 overflows the 32-entry traversal stack";
    static const char* e_big = "more than 2^29-1 nodes or 2^31-1 primitives do not fit the node encoding";
    KdBuildParams p = p_in;
    p.min_num_prims = std::max(p.min_num_prims, 1u);  // KdTree.hpp:91
    for (size_t i = 0; i < 3 * (size_t)n_verts; ++i)
        if (!std::isfinite(verts[i])) { *err = e_fin; return false; }
    for (size_t i = 0; i < (size_t)n_tris; ++i)
        for (int k = 0; k < 3; ++k)
            if (tris[4 * i + k] >= n_verts) { *err = e_idx; return false; }
    if (n_tris > 0x1FFFFFFFu) { *err = e_big; return false; }
    int max_depth = p.max_depth;
    if (max_depth < 0) max_depth = (int)(8 + 1.3f * log2f((float)n_tris));  // KdTree.hpp:97-100
    // The traversal stack has 32 entries (KdTree.hpp:276); one entry per interior ancestor.
    if (max_depth > 31) { *err = e_depth; return false; }
    if (p.n_threads <= 0) p.n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));

    std::vector<Box3> prim_boxes(n_tris);
    Box3 tree_box;
    for (int a = 0; a < 3; ++a) { tree_box.mn[a] = FLT_MAX; tree_box.mx[a] = -FLT_MAX; }
    std::vector<uint32_t> all(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        all[i] = i;
        Box3 b;
        for (int a = 0; a < 3; ++a) { b.mn[a] = FLT_MAX; b.mx[a] = -FLT_MAX; }
        for (int k = 0; k < 3; ++k) {  // Triangle::computeBBox, Source/RT/RTBase.cpp:142-148
            const float* v = verts + 3 * (size_t)tris[4 * (size_t)i + k];
            for (int a = 0; a < 3; ++a) { b.mn[a] = sel_min(b.mn[a], v[a]); b.mx[a] = sel_max(b.mx[a], v[a]); }
        }
        prim_boxes[i] = b;
        for (int a = 0; a < 3; ++a) {  // BBox::extend(BBox), Source/Common/BBox.cpp:38-41
            tree_box.mn[a] = sel_min(tree_box.mn[a], b.mn[a]); tree_box.mx[a] = sel_max(tree_box.mx[a], b.mn[a]);
            tree_box.mn[a] = sel_min(tree_box.mn[a], b.mx[a]); tree_box.mx[a] = sel_max(tree_box.mx[a], b.mx[a]);
        }
    }
    Shared sh;
    sh.prim_boxes = prim_boxes.data();
    sh.p = p;
    sh.max_depth = max_depth;
    sh.spawn_depth = 0;
    if (p.n_threads > 1) {
        int d = 0;
        while ((1 << d) < 2 * p.n_threads) ++d;
        sh.spawn_depth = d;
    }
    sh.spawn_min_prims = 4096;
    Subtree st;
    if (n_tris > 0) build_task(sh, st, tree_box, std::move(all), 0, 0);
    if (st.nodes.size() > 0x1FFFFFFFu) { *err = e_big; return false; }
    out.nodes.swap(st.nodes);
    out.leaf_prim_ids.swap(st.leaf_ids);
    for (int a = 0; a < 3; ++a) { out.box_min[a] = tree_box.mn[a]; out.box_max[a] = tree_box.mx[a]; }
    out.max_depth = max_depth;
    out.actual_depth = st.actual_depth;
    out.leaf_count = st.leaf_count;
    out.prims_in_leaves = st.prims_in_leaves;
    out.degenerate_split_guard_hit = st.guard_hit;
    return true;
}

}  // namespace gigl
