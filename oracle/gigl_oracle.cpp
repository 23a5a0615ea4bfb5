/*
 * gigl_oracle.cpp — CPU ORACLE (test infrastructure, NOT product; see gigl_oracle.h).
 * PARITY UNPINNED against the reference itself (gigl_oracle.h says why).
 *
 * A restatement of the reference's ray-query path in plain scalar C++ (no GLM).
 * Every arithmetic expression keeps the reference's operation order; build with
 * -ffp-contract=off and without fast-math (the reference is MSVC /fp:precise,
 * GIGL.vcxproj:84-100).  Citations are path:line under /root/reference.
 */
#include "gigl_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef unsigned int uint;

/* ---- vec3 with GLM 0.9.5.4 operation order --------------------------------------- */
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline void set_comp(V3& v, int a, float f) { if (a == 0) v.x = f; else if (a == 1) v.y = f; else v.z = f; }
/* Source/Include/GLM/detail/func_geometric.inl:66-73: products first, then x+y, then +z */
inline float dot(V3 a, V3 b) { V3 t = a * b; return t.x + t.y + t.z; }
/* func_geometric.inl:224-227 */
inline V3 cross(V3 x, V3 y) {
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:112-113 */
inline float length(V3 v) { float sqr = v.x * v.x + v.y * v.y + v.z * v.z; return sqrtf(sqr); }
/* func_geometric.inl:263-264 with inversesqrt = 1/sqrt (func_exponential.inl:226-229) */
inline V3 normalize(V3 v) { float sqr = v.x * v.x + v.y * v.y + v.z * v.z; return v * (1.0f / sqrtf(sqr)); }
/* Source/Include/GLM/detail/func_common.inl:377,395 — select-style, NOT fminf/fmaxf */
inline float gmin(float x, float y) { return x < y ? x : y; }
inline float gmax(float x, float y) { return x > y ? x : y; }

/* Source/Common/Constants.h:6-9, :24-26, :38-42 */
const float TWO_PI = 6.28318548f, PI = 3.14159274f, INV_PI = 0.318309873f;
const float RAY_OFFSET = 1E-4f, TRI_EPS = 1E-4f, SURVIVAL_P_RR = 0.95f;
const float MAX_FOG_HEIGHT = 548.8f, HG_G = 0.25f;
const int N_GI_BOUNCES = 3, N_OCTAVES = 6;

/* Source/RT/RTBase.h:12 */
const float EPS = FLT_EPSILON;

/* Source/RT/RTBase.h:18-29.  `1 << exp` is undefined for exp >= 31; x86 shl masks the
 * count to 5 bits, which is what the survey-time build did, so the shift
 * count is masked explicitly here. */
inline float rescaled_eps(float t) {
    int e;
    (void)frexpf(t, &e);
    ++e;
    if (e >= 0) {
        return EPS * (float)(int)(1u << ((unsigned)e & 31u));
    } else {
        return EPS / (float)(int)(1u << ((unsigned)(-e) & 31u));
    }
}

/* ---- BBox: Source/Common/BBox.cpp -------------------------------------------------- */
struct Box {
    V3 b[2];
    Box() { b[0] = v3(FLT_MAX, FLT_MAX, FLT_MAX); b[1] = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX); } /* :17-20 */
    Box(V3 mn, V3 mx) { b[0] = mn; b[1] = mx; }
    void extend(V3 p) {                                                                     /* :31-36 */
        b[0].x = gmin(b[0].x, p.x); b[1].x = gmax(b[1].x, p.x);
        b[0].y = gmin(b[0].y, p.y); b[1].y = gmax(b[1].y, p.y);
        b[0].z = gmin(b[0].z, p.z); b[1].z = gmax(b[1].z, p.z);
    }
    void extend(const Box& o) { extend(o.b[0]); extend(o.b[1]); }                          /* :38-41 */
    V3 dims() const { return b[1] - b[0]; }                                                 /* :49-51 */
    float area() const { V3 d = dims(); return 2.0f * (d.x * d.y + (d.x + d.y) * d.z); }    /* :57-60 */
};

/* Source/RT/RTBase.h:81-105 */
struct Ray {
    V3 o, d, inv_d;
    float t_min, t_max;
    /* Intersection (RTBase.h:95-100, RTBase.cpp:133) + what the hit record needs */
    float dist;
    V3 normal;
    uint tri_id;
    float u, v;
    /* exact per-ray work counters for the roofline */
    uint nodes_visited, tri_tests;
};
inline Ray make_ray(V3 o, V3 d) {                                    /* RTBase.cpp:111-115 */
    Ray r;
    r.o = o; r.d = d;
    r.inv_d = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.t_min = 0.0f; r.t_max = FLT_MAX;
    r.dist = FLT_MAX; r.normal = v3(0, 0, 0); r.tri_id = 0xFFFFFFFFu; r.u = r.v = 0.0f;
    r.nodes_visited = r.tri_tests = 0;
    return r;
}
inline Ray make_ray_offset(V3 o, V3 d, V3 n) {                       /* RTBase.cpp:117-122 */
    return make_ray(o + RAY_OFFSET * n, d);
}
inline V3 pt_at(const Ray& r, float t) { return r.o + t * r.d; }     /* RTBase.cpp:124-126 */

/* Source/Common/BBox.cpp:81-109 (Williams et al.) */
inline bool box_intersect(const Box& bx, const Ray& ray, float& entr, float& exit) {
    float t_entr, t_exit, t_y_min, t_y_max, t_z_min, t_z_max;
    const int sx = ray.inv_d.x >= 0.0f, sy = ray.inv_d.y >= 0.0f, sz = ray.inv_d.z >= 0.0f;
    t_entr  = (bx.b[1 - sx].x - ray.o.x) * ray.inv_d.x;
    t_exit  = (bx.b[sx].x - ray.o.x) * ray.inv_d.x;
    t_y_min = (bx.b[1 - sy].y - ray.o.y) * ray.inv_d.y;
    t_y_max = (bx.b[sy].y - ray.o.y) * ray.inv_d.y;
    if ((t_entr > t_y_max) || (t_y_min > t_exit)) { entr = FLT_MAX; exit = -FLT_MAX; return false; }
    t_entr = gmax(t_entr, t_y_min);
    t_exit = gmin(t_exit, t_y_max);
    t_z_min = (bx.b[1 - sz].z - ray.o.z) * ray.inv_d.z;
    t_z_max = (bx.b[sz].z - ray.o.z) * ray.inv_d.z;
    if ((t_entr > t_z_max) || (t_z_min > t_exit)) { entr = FLT_MAX; exit = -FLT_MAX; return false; }
    t_entr = gmax(t_entr, t_z_min);
    t_exit = gmin(t_exit, t_z_max);
    if (t_entr < ray.t_max && t_exit > ray.t_min) {
        entr = t_entr; exit = t_exit;
        return entr < FLT_MAX;                                        /* IntDist::operator bool, :13-15 */
    }
    entr = FLT_MAX; exit = -FLT_MAX;
    return false;
}

/* ---- k-d tree: Source/RT/KdTree.h, KdTree.hpp -------------------------------------- */
struct Node { uint w0, w1; };                                        /* KdTree.h:22-65 */
inline bool  n_is_leaf(const Node& n) { return (n.w1 & 1u) != 0; }
inline uint  n_axis(const Node& n) { return (n.w1 >> 1) & 3u; }
inline uint  n_above(const Node& n) { return n.w1 >> 3; }
inline uint  n_nprims(const Node& n) { return n.w1 >> 1; }
inline float n_split(const Node& n) { float f; memcpy(&f, &n.w0, 4); return f; }

struct BoundEdge {                                                   /* KdTree.h:67-77 */
    enum Type { END = 0, START = 1, BOTH = 2, NONE = 3 };
    float axis_pos; uint prim_id; Type type;
    bool operator<(const BoundEdge& r) const {                       /* KdTree.hpp:73-80 */
        if (axis_pos != r.axis_pos) return axis_pos < r.axis_pos;
        return type < r.type;
    }
};

}  // namespace

struct ora_scene {
    std::vector<V3> verts;
    std::vector<uint> tris;               /* 4 per triangle: i0 i1 i2 mat */
    uint n_tris;
    /* tree */
    int max_depth, actual_depth, inters_cost, trav_cost;
    uint min_num_prims;
    float empty_bonus, avg_prims_per_leaf;
    Box tree_box;
    uint next_node_id, leaf_count;
    std::vector<Node> nodes;
    std::vector<uint> leaf_prim_ids;
    /* build scratch */
    std::vector<Box> prim_boxes;
    std::vector<BoundEdge> edges[3];
};

namespace {

/* Leaf node ctor: KdTree.hpp:14-30 */
inline Node make_leaf(uint n_prims, const uint* ids, std::vector<uint>& leaf_prim_ids) {
    Node n;
    n.w1 = 1u | (n_prims << 1);
    if (n_prims == 0) n.w0 = 0;
    else if (n_prims == 1) n.w0 = ids[0];
    else {
        n.w0 = (uint)leaf_prim_ids.size();
        leaf_prim_ids.insert(leaf_prim_ids.end(), ids, ids + n_prims);
    }
    return n;
}
/* Interior node ctor: KdTree.hpp:32-37 */
inline Node make_interior(uint axis, float split, uint above) {
    Node n;
    memcpy(&n.w0, &split, 4);
    n.w1 = (axis << 1) | (above << 3);
    return n;
}

/* KdTree.hpp:139-264 */
void build_tree(ora_scene& T, uint node_id, const Box& node_box, const uint* overlap, uint n_prims,
                int depth, int bad_refines, uint* prims_bottom, uint* prims_top) {
    if (T.next_node_id == (uint)T.nodes.size()) {
        uint na = std::max<uint>(((uint)T.nodes.size() * 3) / 2, 64u);
        T.nodes.resize(na);
    }
    ++T.next_node_id;
    T.actual_depth = std::max(T.actual_depth, depth + 1);
    if (n_prims <= T.min_num_prims || depth == T.max_depth) {
        T.nodes[node_id] = make_leaf(n_prims, overlap, T.leaf_prim_ids);
        T.avg_prims_per_leaf += n_prims;
        ++T.leaf_count;
        return;
    }
    float best_cost = FLT_MAX;
    uint best_axis = UINT32_MAX, best_edge_id = UINT32_MAX;
    const float inv_total_sa = 1.0f / node_box.area();
    const V3 diag = node_box.dims();
    /* :166-180 — three independent sorts; the reference runs them on 3 OpenMP threads,
     * which does not change the result. */
    for (int ax = 0; ax < 3; ++ax) {
        BoundEdge* E = T.edges[ax].data();
        for (uint i = 0; i < n_prims; ++i) {
            const uint prim_id = overlap[i];
            const Box& pb = T.prim_boxes[prim_id];
            const float lo = comp(pb.b[0], ax), hi = comp(pb.b[1], ax);
            const bool is_flat = lo == hi;
            E[2 * i].axis_pos = lo; E[2 * i].prim_id = prim_id;
            E[2 * i].type = is_flat ? BoundEdge::BOTH : BoundEdge::START;
            E[2 * i + 1].axis_pos = hi; E[2 * i + 1].prim_id = prim_id;
            E[2 * i + 1].type = is_flat ? BoundEdge::NONE : BoundEdge::END;
        }
        std::sort(E, E + 2 * n_prims);   /* same libstdc++ introsort as the survey-time build */
    }
    for (int axis = 0; axis < 3; ++axis) {                            /* :181-215 */
        const BoundEdge* E = T.edges[axis].data();
        uint n_below = 0, n_above = n_prims;
        const float bmin = comp(node_box.b[0], axis), bmax = comp(node_box.b[1], axis);
        for (uint i = 0; i < 2 * n_prims; ++i) {
            switch (E[i].type) {
                case BoundEdge::BOTH:  ++n_below;   /* fallthrough */
                case BoundEdge::END:   --n_above;   /* fallthrough */
                case BoundEdge::START: break;
                case BoundEdge::NONE:  continue;
            }
            const float edge_pos = E[i].axis_pos;
            if (edge_pos > bmin && edge_pos < bmax) {
                const int a2 = (axis + 1) % 3, a3 = (axis + 2) % 3;
                const float d2 = comp(diag, a2), d3 = comp(diag, a3);
                const float below_sa = 2 * (d2 * d3 + (edge_pos - bmin) * (d2 + d3));
                const float above_sa = 2 * (d2 * d3 + (bmax - edge_pos) * (d2 + d3));
                const float bns = (0 == n_above || 0 == n_below) ? T.empty_bonus : 0.0f;
                const float cost = (1.0f - bns) * (n_below * below_sa + n_above * above_sa);
                if (cost < best_cost) { best_axis = axis; best_cost = cost; best_edge_id = i; }
            }
            if (BoundEdge::START == E[i].type) ++n_below;
        }
    }
    best_cost = T.trav_cost + (T.inters_cost * inv_total_sa) * best_cost;       /* :218 */
    const uint old_cost = T.inters_cost * n_prims;
    if (best_cost > old_cost) ++bad_refines;
    /* :221.  best_axis == UINT32_MAX (no edge strictly inside the box) with n_prims >= 16
     * indexes m_edges out of range in the reference; the restatement makes a leaf. */
    if ((best_cost > 4.0f * old_cost && n_prims < 16) || 3 == bad_refines || best_axis == UINT32_MAX) {
        T.nodes[node_id] = make_leaf(n_prims, overlap, T.leaf_prim_ids);
        T.avg_prims_per_leaf += n_prims;
        ++T.leaf_count;
        return;
    }
    const BoundEdge* E = T.edges[best_axis].data();
    uint n_bottom = 0, n_top = 0;                                      /* :229-247 */
    for (uint k = 0; k < best_edge_id; ++k)
        if (E[k].type == BoundEdge::START || E[k].type == BoundEdge::BOTH) prims_bottom[n_bottom++] = E[k].prim_id;
    if (E[best_edge_id].type == BoundEdge::BOTH) prims_bottom[n_bottom++] = E[best_edge_id].prim_id;
    for (uint k = best_edge_id + 1; k < 2 * n_prims; ++k)
        if (E[k].type == BoundEdge::END || E[k].type == BoundEdge::BOTH) prims_top[n_top++] = E[k].prim_id;
    const float split_pos = E[best_edge_id].axis_pos;
    V3 top_min = node_box.b[0], bottom_max = node_box.b[1];
    set_comp(top_min, best_axis, split_pos);
    set_comp(bottom_max, best_axis, split_pos);
    const Box bottom_box(node_box.b[0], bottom_max);
    build_tree(T, node_id + 1, bottom_box, prims_bottom, n_bottom, depth + 1, bad_refines,
               prims_bottom, prims_top + n_prims);
    const uint top_child_id = T.next_node_id;
    T.nodes[node_id] = make_interior(best_axis, split_pos, top_child_id);
    const Box top_box(top_min, node_box.b[1]);
    build_tree(T, top_child_id, top_box, prims_top, n_top, depth + 1, bad_refines,
               prims_bottom, prims_top + n_prims);
}

/* Source/RT/RTBase.cpp:150-177 (Möller–Trumbore), plus id/u/v capture */
inline bool tri_intersect(const ora_scene& S, uint tri, Ray& ray) {
    ++ray.tri_tests;
    const uint* ix = &S.tris[4 * (size_t)tri];
    const V3 pt0 = S.verts[ix[0]], pt1 = S.verts[ix[1]], pt2 = S.verts[ix[2]];
    const V3 edge0 = pt1 - pt0, edge1 = pt2 - pt0;
    const V3 pVec = cross(ray.d, edge1);
    const float det = dot(edge0, pVec);
    if (det != 0.0f) {
        const float invDet = 1.0f / det;
        const V3 tVec = ray.o - pt0;
        const V3 qVec = cross(tVec, edge0);
        const float u = dot(tVec, pVec) * invDet;
        const float v = dot(ray.d, qVec) * invDet;
        if (u >= -TRI_EPS && v >= -TRI_EPS && (u + v) <= 1.0f + 2.0f * TRI_EPS) {
            const float dist = dot(edge1, qVec) * invDet;
            if (ray.t_min < dist && dist < ray.t_max && dist < ray.dist) {
                ray.dist = dist;
                ray.normal = cross(edge0, edge1);
                ray.tri_id = tri; ray.u = u; ray.v = v;
                return true;
            }
        }
    }
    return false;
}

/* Source/RT/KdTree.hpp:266-347 */
bool kd_intersect(const ora_scene& S, Ray& ray, bool is_vis_ray) {
    float entr, exit;
    if (!box_intersect(S.tree_box, ray, entr, exit)) return false;
    {
        const float nmin = gmax(ray.t_min, entr - rescaled_eps(entr));
        const float nmax = gmin(ray.t_max, exit + rescaled_eps(exit));
        ray.t_min = nmin; ray.t_max = nmax;
    }
    struct TraceNode { float t_min, t_max; uint id; } stack[32];
    int stack_pos = -1;
    const Node* nodes = S.nodes.data();
    const Node* node = &nodes[0];
    while (node) {
        if (ray.dist < ray.t_min) break;
        ++ray.nodes_visited;
        if (!n_is_leaf(*node)) {
            const uint axis = n_axis(*node);
            const float split = n_split(*node);
            const Node* children[2] = {&nodes[n_above(*node)], node + 1};
            const float oa = comp(ray.o, axis), da = comp(ray.d, axis);
            const bool bottom_first = (oa < split) || (oa == split && da <= 0.0f);
            if (bottom_first) std::swap(children[0], children[1]);
            const float t_split = (split - oa) * comp(ray.inv_d, axis);
            if (t_split > ray.t_max || t_split <= 0.0f) {
                node = children[0];
            } else if (t_split < ray.t_min) {
                node = children[1];
            } else {
                const uint second = (uint)(children[1] - nodes);
                const float new_t_min = t_split - rescaled_eps(t_split);
                ++stack_pos;
                stack[stack_pos].t_min = new_t_min; stack[stack_pos].t_max = ray.t_max; stack[stack_pos].id = second;
                node = children[0];
                ray.t_max = t_split + rescaled_eps(t_split);
            }
        } else {
            const uint n_prims = n_nprims(*node);
            if (1u == n_prims) {
                if (tri_intersect(S, node->w0, ray)) return true;
            } else {
                bool hit = false;
                for (uint i = 0; i < n_prims; ++i) {
                    if (tri_intersect(S, S.leaf_prim_ids[node->w0 + i], ray)) {
                        if (is_vis_ray) return true;
                        hit = true;
                    }
                }
                if (hit) return true;
            }
            if (stack_pos >= 0) {
                node = &nodes[stack[stack_pos].id];
                ray.t_min = stack[stack_pos].t_min;
                ray.t_max = stack[stack_pos].t_max;
                --stack_pos;
            } else break;
        }
    }
    return false;
}

/* Source/Common/Scene.cpp:234-243 */
inline bool scene_trace(const ora_scene& S, Ray& ray, bool is_vis_ray) {
    if (kd_intersect(S, ray, is_vis_ray)) {
        if (is_vis_ray) return true;
        ray.normal = normalize(ray.normal);
        return true;
    }
    return false;
}

struct Hit16 { uint tri_id; float t, u, v; };

inline Ray load_ray(const float* ro, const float* rd, uint64_t i) {
    Ray r = make_ray(v3(ro[4 * i], ro[4 * i + 1], ro[4 * i + 2]), v3(rd[4 * i], rd[4 * i + 1], rd[4 * i + 2]));
    r.t_min = ro[4 * i + 3]; r.t_max = rd[4 * i + 3];               /* setValidRange, RTBase.cpp:128-131 */
    return r;
}

/* ---- fog: Source/Fog/DensityField.cpp, Source/Common/Interpolation.hpp ------------- */
inline float lerp1(float p0, float p1, float t) { return (1.0f - t) * p0 + t * p1; }  /* Interpolation.hpp:3-6 */
inline float fog_texel(const ora_fog& F, int x, int y, int z) {                       /* DensityField.cpp:204-211 */
    if (x >= 0 && x < F.res[0] && y >= 0 && y < F.res[1] && z >= 0 && z < F.res[2])
        return F.dens[x + y * F.res[0] + z * F.res[0] * F.res[1]] / 255.0f;
    return 0.0f;
}
/* the filter of DensityField.cpp:180-201 from a normalized position */
inline float fog_filter(const ora_fog& F, V3 n_pos) {
    V3 tc = n_pos * v3((float)F.res[0], (float)F.res[1], (float)F.res[2]);
    tc = tc - v3(0.5f, 0.5f, 0.5f);
    const int x0 = (int)floorf(tc.x), x1 = (int)ceilf(tc.x);
    const int y0 = (int)floorf(tc.y), y1 = (int)ceilf(tc.y);
    const int z0 = (int)floorf(tc.z), z1 = (int)ceilf(tc.z);
    const float tx = tc.x - x0, ty = tc.y - y0, tz = tc.z - z0;
    const float d000 = fog_texel(F, x0, y0, z0), d100 = fog_texel(F, x1, y0, z0);
    const float d010 = fog_texel(F, x0, y1, z0), d110 = fog_texel(F, x1, y1, z0);
    const float d001 = fog_texel(F, x0, y0, z1), d101 = fog_texel(F, x1, y0, z1);
    const float d011 = fog_texel(F, x0, y1, z1), d111 = fog_texel(F, x1, y1, z1);
    /* Interpolation.hpp:8-23: x, then y, then z */
    const float i00 = lerp1(d000, d100, tx), i10 = lerp1(d010, d110, tx);
    const float i0 = lerp1(i00, i10, ty);
    const float i01 = lerp1(d001, d101, tx), i11 = lerp1(d011, d111, tx);
    const float i1 = lerp1(i01, i11, ty);
    return lerp1(i0, i1, tz);
}
inline V3 fog_min(const ora_fog& F) { return v3(F.box[0], F.box[1], F.box[2]); }
inline V3 fog_max(const ora_fog& F) { return v3(F.box[3], F.box[4], F.box[5]); }
/* DensityField::sampleDensity, DensityField.cpp:177-202; computeNormPos BBox.cpp:53-55 */
inline float sample_density(const ora_fog& F, V3 pos) {
    const V3 n_pos = (pos - fog_min(F)) / (fog_max(F) - fog_min(F));
    return fog_filter(F, n_pos);
}

/* GLSL min/max (Source/Shaders/Surface.frag:146-153): min(x,y) = y<x?y:x, max(x,y) = x<y?y:x */
inline float smin(float x, float y) { return y < x ? y : x; }
inline float smax(float x, float y) { return x < y ? y : x; }

/* Surface.frag:139-157 */
inline bool glsl_intersect_bbox(const ora_fog& F, V3 ro, V3 rd, float max_dist, float& t_min, float& t_max) {
    const V3 inv = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    const V3 b0 = fog_min(F), b1 = fog_max(F);
    float t0 = (b0.x - ro.x) * inv.x, t1 = (b1.x - ro.x) * inv.x;
    t_min = smin(t0, t1); t_max = smax(t0, t1);
    t0 = (b0.y - ro.y) * inv.y; t1 = (b1.y - ro.y) * inv.y;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    t0 = (b0.z - ro.z) * inv.z; t1 = (b1.z - ro.z) * inv.z;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    return t_max > smax(t_min, 0.0f) && t_min < max_dist;
}
/* calcFogDens + sampleExtK, Surface.frag:159-169: n_pos = (w_pos - min) * inv_fog_dims with
 * inv_fog_dims = 1/(max-min) set by the host (Source/GIGL.cpp:83); texture() := fog_filter */
inline float glsl_sample_ext_k(const ora_fog& F, V3 inv_dims, V3 w_pos) {
    const V3 n_pos = (w_pos - fog_min(F)) * inv_dims;
    return F.ext_k * fog_filter(F, n_pos);
}
/* rayMarch, Surface.frag:172-188 */
inline float glsl_ray_march(const ora_fog& F, V3 inv_dims, V3 ro, V3 rd, float t_min, float t_max) {
    const float dt = (t_max - t_min) / (float)F.n_intervals;
    float prev = glsl_sample_ext_k(F, inv_dims, ro + t_min * rd);
    float total = 0.0f;
    for (int i = 1; i <= F.n_intervals; ++i) {
        const float t = t_min + (float)i * dt;
        const float cur = glsl_sample_ext_k(F, inv_dims, ro + t * rd);
        total += 0.5f * (prev + cur);
        prev = cur;
    }
    return expf(-(total * dt));
}
/* calcTransm, Surface.frag:190-203 */
inline float glsl_calc_transm(const ora_fog& F, V3 p, V3 dir, float dist_sq) {
    float transm = 1.0f;
    if (F.ext_k > 0.0f) {
        const V3 dm = fog_max(F) - fog_min(F);
        const V3 inv_dims = v3(1.0f / dm.x, 1.0f / dm.y, 1.0f / dm.z);
        const float max_dist = sqrtf(dist_sq);
        float t_min, t_max;
        if (glsl_intersect_bbox(F, p, dir, max_dist, t_min, t_max)) {
            t_min = smax(t_min, 0.0f);
            t_max = smin(t_max, max_dist);
            transm = glsl_ray_march(F, inv_dims, p, dir, t_min, t_max);
        }
    }
    return transm;
}

/* ---- sampling helpers: Source/RT/RTBase.h:33-50, RTBase.cpp:27-36 ------------------- */
struct M3 { V3 c0, c1, c2; };                                        /* column-major glm::mat3 */
/* Source/Include/GLM/detail/type_mat3x3.inl:593-596 */
inline V3 mul(const M3& m, V3 v) {
    return v3(m.c0.x * v.x + m.c1.x * v.y + m.c2.x * v.z,
              m.c0.y * v.x + m.c1.y * v.y + m.c2.y * v.z,
              m.c0.z * v.x + m.c1.z * v.y + m.c2.z * v.z);
}
inline M3 span_tangent_space(V3 normal) {                            /* RTBase.h:33-41 */
    const V3 c1 = cross(normal, v3(0.0f, 0.0f, 1.0f));
    const V3 c2 = cross(normal, v3(0.0f, 1.0f, 0.0f));
    const V3 c3 = length(c1) > length(c2) ? c1 : c2;
    const V3 tangent = normalize(-c3);
    const V3 bitangent = cross(normal, tangent);
    M3 m = {tangent, bitangent, normal};
    return m;
}
inline V3 gen_vec_sin_cos(float phi, float sin_the, float cos_the) { /* RTBase.h:44-50 */
    return v3(sin_the * cosf(phi), sin_the * sinf(phi), cos_the);
}
inline V3 gen_cos_rand_hemi_dir(float u1, float u2) {                /* RTBase.cpp:27-36 */
    const float phi = TWO_PI * u1;
    const float sin_the = sqrtf(1.0f - u2);
    return gen_vec_sin_cos(phi, sin_the, sqrtf(u2));
}
inline V3 gen_cos_pow_rand_hemi_dir(float u1, float u2, float n) {   /* RTBase.cpp:39-49 */
    const float phi = TWO_PI * u1;
    const float cos_the = powf(u2, 1.0f / (n + 1.0f));
    const float sin_the = sqrtf(1.0f - cos_the * cos_the);
    return gen_vec_sin_cos(phi, sin_the, cos_the);
}

/* ---- GLM 0.9.5.4 simplex(vec3): Source/Include/GLM/gtc/noise.inl:679-751 ----------- */
struct V4 { float x, y, z, w; };
/* Source/Include/GLM/detail/_noise.hpp:36-45, :72-75 */
inline float mod289f(float x) { return x - floorf(x * 1.0f / 289.0f) * 289.0f; }
inline float permutef(float x) { return mod289f(((x * 34.0f) + 1.0f) * x); }
inline float taylor_inv_sqrt(float r) { return (float)1.79284291400159 - (float)0.85373472095314 * r; }
inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

}  // namespace

/* ===================================== C interface ================================== */
extern "C" {

ora_scene* ora_scene_create(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                            int inters_cost, int trav_cost, int max_depth, uint32_t min_num_prims,
                            float empty_bonus) {
    ora_scene* S = new ora_scene();
    S->verts.resize(n_verts);
    for (uint32_t i = 0; i < n_verts; ++i) S->verts[i] = v3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    S->tris.assign(tris, tris + 4 * (size_t)n_tris);
    S->n_tris = n_tris;
    /* KdTree.hpp:87-132 */
    S->max_depth = max_depth; S->actual_depth = 0;
    S->min_num_prims = std::max(min_num_prims, 1u);
    S->avg_prims_per_leaf = 0.0f; S->inters_cost = inters_cost; S->trav_cost = trav_cost;
    S->empty_bonus = empty_bonus; S->next_node_id = 0; S->leaf_count = 0;
    const uint n = n_tris;
    if (S->max_depth < 0) S->max_depth = (int)(8 + 1.3f * log2f((float)n));
    S->prim_boxes.resize(n);
    std::vector<uint> prims_bottom(n);
    std::vector<uint> prims_top((size_t)n * (S->max_depth + 1));
    for (int a = 0; a < 3; ++a) S->edges[a].resize(2 * (size_t)n);
    for (uint i = 0; i < n; ++i) {
        prims_bottom[i] = i;
        Box b;                                                        /* Triangle::computeBBox, RTBase.cpp:142-148 */
        for (int k = 0; k < 3; ++k) b.extend(S->verts[S->tris[4 * (size_t)i + k]]);
        S->prim_boxes[i] = b;
        S->tree_box.extend(b);
    }
    if (n > 0)
        build_tree(*S, 0, S->tree_box, prims_bottom.data(), n, 0, 0, prims_bottom.data(), prims_top.data());
    S->nodes.resize(S->next_node_id);
    S->nodes.shrink_to_fit();
    for (int a = 0; a < 3; ++a) std::vector<BoundEdge>().swap(S->edges[a]);
    std::vector<Box>().swap(S->prim_boxes);
    if (S->leaf_count) S->avg_prims_per_leaf /= S->leaf_count;
    return S;
}

/* A scene over a tree that was built elsewhere (the engine's exported tree, in the reference's node
 * encoding): for scenes whose serial build here would take minutes (10 M triangles), SURVEY.md §7
 * allows "the build's own tree as ground truth for large scenes".  Traversal (KdTree.hpp:266-347) and
 * every caller of it work on it as on a scene built by ora_scene_create; the tree box is the union
 * of the triangles' boxes, as the constructor computes it (KdTree.hpp:100-106). */
ora_scene* ora_scene_from_tree(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                               const uint32_t* nodes_words, uint32_t n_nodes, const uint32_t* leaf_prim_ids,
                               uint32_t n_leaf_prim_ids) {
    ora_scene* S = new ora_scene();
    S->verts.resize(n_verts);
    for (uint32_t i = 0; i < n_verts; ++i) S->verts[i] = v3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    S->tris.assign(tris, tris + 4 * (size_t)n_tris);
    S->n_tris = n_tris;
    S->max_depth = S->actual_depth = 0; S->min_num_prims = 1; S->inters_cost = S->trav_cost = 0;
    S->empty_bonus = 0.0f; S->avg_prims_per_leaf = 0.0f; S->leaf_count = 0;
    for (uint32_t i = 0; i < n_tris; ++i) {
        Box b;                                                        /* Triangle::computeBBox, RTBase.cpp:142-148 */
        for (int k = 0; k < 3; ++k) b.extend(S->verts[S->tris[4 * (size_t)i + k]]);
        S->tree_box.extend(b);
    }
    S->nodes.resize(n_nodes);
    memcpy(S->nodes.data(), nodes_words, (size_t)n_nodes * 8);
    S->leaf_prim_ids.assign(leaf_prim_ids, leaf_prim_ids + n_leaf_prim_ids);
    S->next_node_id = n_nodes;
    double prims = 0.0;
    for (uint32_t i = 0; i < n_nodes; ++i)
        if (n_is_leaf(S->nodes[i])) { ++S->leaf_count; prims += n_nprims(S->nodes[i]); }
    if (S->leaf_count) S->avg_prims_per_leaf = (float)(prims / S->leaf_count);
    return S;
}

/* Scene::trace on ONE ray, with everything the caller's Ray holds afterwards: the bool, the hit
 * record, the normal as Scene::trace leaves it (normalized for closest-hit rays, cross(e0,e1) for
 * visibility rays; Scene.cpp:234-243) and the ray's (t_min, t_max) as KdTree::intersect leaves
 * them (KdTree.hpp:273-274, :310, :337-338). */
int ora_trace_one(const ora_scene* s, const float o[3], const float d[3], float t_min, float t_max, int is_vis_ray,
                  void* hit16, float normal_out[3], float range_out[2]) {
    Ray r = make_ray(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    r.t_min = t_min; r.t_max = t_max;
    const bool hit = scene_trace(*s, r, is_vis_ray != 0);
    Hit16 h;
    if (hit) { h.tri_id = r.tri_id; h.t = r.dist; h.u = r.u; h.v = r.v; }
    else { h.tri_id = 0xFFFFFFFFu; h.t = FLT_MAX; h.u = 0.0f; h.v = 0.0f; }
    if (hit16) memcpy(hit16, &h, 16);
    if (normal_out) { normal_out[0] = r.normal.x; normal_out[1] = r.normal.y; normal_out[2] = r.normal.z; }
    if (range_out) { range_out[0] = r.t_min; range_out[1] = r.t_max; }
    return hit ? 1 : 0;
}

void ora_scene_destroy(ora_scene* s) { delete s; }
uint32_t ora_scene_n_nodes(const ora_scene* s) { return (uint32_t)s->nodes.size(); }
uint32_t ora_scene_n_leaf_prim_ids(const ora_scene* s) { return (uint32_t)s->leaf_prim_ids.size(); }
int ora_scene_depth(const ora_scene* s) { return s->actual_depth; }
uint32_t ora_scene_leaf_count(const ora_scene* s) { return s->leaf_count; }
float ora_scene_avg_prims_per_leaf(const ora_scene* s) { return s->avg_prims_per_leaf; }
void ora_scene_bbox(const ora_scene* s, float out[6]) {
    out[0] = s->tree_box.b[0].x; out[1] = s->tree_box.b[0].y; out[2] = s->tree_box.b[0].z;
    out[3] = s->tree_box.b[1].x; out[4] = s->tree_box.b[1].y; out[5] = s->tree_box.b[1].z;
}
void ora_scene_copy_nodes(const ora_scene* s, uint32_t* out) { memcpy(out, s->nodes.data(), s->nodes.size() * 8); }
void ora_scene_copy_leaf_prim_ids(const ora_scene* s, uint32_t* out) {
    memcpy(out, s->leaf_prim_ids.data(), s->leaf_prim_ids.size() * 4);
}

float ora_rescaled_eps(float t) { return rescaled_eps(t); }

int ora_bbox_intersect(const float box[6], const float o[3], const float d[3], float t_min, float t_max,
                       float* entr, float* exit) {
    Box b(v3(box[0], box[1], box[2]), v3(box[3], box[4], box[5]));
    Ray r = make_ray(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    r.t_min = t_min; r.t_max = t_max;
    return box_intersect(b, r, *entr, *exit) ? 1 : 0;
}

void ora_trace_closest(const ora_scene* s, const float* ro, const float* rd, uint64_t n, void* hits16,
                       uint32_t* counters, int n_threads) {
    Hit16* H = (Hit16*)hits16;
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Ray r = load_ray(ro, rd, (uint64_t)i);
        const bool hit = scene_trace(*s, r, false);
        Hit16 h;
        if (hit) { h.tri_id = r.tri_id; h.t = r.dist; h.u = r.u; h.v = r.v; }
        else { h.tri_id = 0xFFFFFFFFu; h.t = FLT_MAX; h.u = 0.0f; h.v = 0.0f; }
        H[i] = h;
        if (counters) { counters[2 * i] = r.nodes_visited; counters[2 * i + 1] = r.tri_tests; }
    }
}

void ora_trace_any(const ora_scene* s, const float* ro, const float* rd, uint64_t n, uint8_t* occluded,
                   uint32_t* counters, int n_threads) {
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        Ray r = load_ray(ro, rd, (uint64_t)i);
        occluded[i] = scene_trace(*s, r, true) ? 1 : 0;
        if (counters) { counters[2 * i] = r.nodes_visited; counters[2 * i + 1] = r.tri_tests; }
    }
}

void ora_camera_setup(const float pos[3], const float dir[3], const float up[3], float h_fov, int res_x,
                      int res_y, float cam_out[12]) {
    /* Source/Common/Camera.cpp:15-29 */
    const float aspect_ratio = (float)res_x / (float)res_y;
    const V3 D = v3(dir[0], dir[1], dir[2]), U = v3(up[0], up[1], up[2]);
    const V3 fwd = normalize(D);
    const V3 right = normalize(cross(fwd, U));
    const V3 upa = normalize(cross(right, fwd));
    const V3 row_vec = ((2.0f * tanf(0.5f * h_fov)) * right) * aspect_ratio;
    const V3 col_vec = (2.0f * tanf(0.5f * h_fov)) * upa;
    const V3 bl = (fwd - 0.5f * row_vec) - 0.5f * col_vec;
    const V3 sx = row_vec / (float)res_x, sy = col_vec / (float)res_y;
    const float o[12] = {pos[0], pos[1], pos[2], bl.x, bl.y, bl.z, sx.x, sx.y, sx.z, sy.x, sy.y, sy.z};
    memcpy(cam_out, o, sizeof(o));
}

void ora_primary_rays(const float cam[12], int res_x, uint64_t first, uint64_t n, float* ro, float* rd) {
    const V3 pos = v3(cam[0], cam[1], cam[2]), bl = v3(cam[3], cam[4], cam[5]);
    const V3 sx = v3(cam[6], cam[7], cam[8]), sy = v3(cam[9], cam[10], cam[11]);
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t p = first + k;
        const float x = (int)(p % (uint64_t)res_x) + 0.5f, y = (int)(p / (uint64_t)res_x) + 0.5f;
        const V3 dir = (bl + x * sx) + y * sy;                        /* Camera.cpp:66 */
        const V3 d = normalize(dir);
        ro[4 * k] = pos.x; ro[4 * k + 1] = pos.y; ro[4 * k + 2] = pos.z; ro[4 * k + 3] = 0.0f;
        rd[4 * k] = d.x; rd[4 * k + 1] = d.y; rd[4 * k + 2] = d.z; rd[4 * k + 3] = FLT_MAX;
    }
}

void ora_shadow_batch(const ora_scene* s, const float* po, const float* pd, const void* prim_hits16,
                      uint64_t n_pts, const float* lights, uint32_t n_lights, uint8_t* occluded,
                      const ora_fog* fog, float* transm, int n_threads) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n_pts; ++i) {
        uint8_t* occ = occluded + (size_t)i * n_lights;
        float* tr = transm ? transm + (size_t)i * n_lights : NULL;
        const Hit16 h = H[i];
        if (h.tri_id == 0xFFFFFFFFu) {
            for (uint32_t l = 0; l < n_lights; ++l) { occ[l] = 0xFF; if (tr) tr[l] = 1.0f; }
            continue;
        }
        const V3 o = v3(po[4 * i], po[4 * i + 1], po[4 * i + 2]);
        const V3 d = v3(pd[4 * i], pd[4 * i + 1], pd[4 * i + 2]);
        const V3 p = o + h.t * d;                                     /* getPtAtDist */
        const uint* ix = &s->tris[4 * (size_t)h.tri_id];
        const V3 pt0 = s->verts[ix[0]];
        const V3 e0 = s->verts[ix[1]] - pt0, e1 = s->verts[ix[2]] - pt0;
        V3 N = normalize(cross(e0, e1));                              /* Scene.cpp:238 */
        if (dot(d, N) > 0.0f) N = -N;
        for (uint32_t l = 0; l < n_lights; ++l) {
            const V3 L = v3(lights[3 * l], lights[3 * l + 1], lights[3 * l + 2]);
            const V3 dirv = L - p;
            const float dist_sq = dot(dirv, dirv);
            const float dist = sqrtf(dist_sq);
            const V3 w = dirv * (1.0f / sqrtf(dist_sq));
            Ray r = make_ray_offset(p, w, N);
            r.t_min = 0.0f; r.t_max = dist;
            occ[l] = scene_trace(*s, r, true) ? 1 : 0;
            if (tr) tr[l] = fog ? glsl_calc_transm(*fog, p, w, dist_sq) : 1.0f;
        }
    }
}

/* The shader's order (Surface.frag:250-256, :270-288): visibility first, calcTransm only for visible
 * lights.  occluded as ora_shadow_batch; transm = 0 for an occluded pair (nothing arrives from that
 * light), calcTransm for a visible one, 1.0 for points whose primary ray missed.  sums (optional):
 * [0] pairs traced, [1] nodes visited, [2] triangle tests, [3] pairs marched - the exact counters the
 * algorithmic-bytes definition of SURVEY.md 8(d) is stated on. */
void ora_shadow_batch_fused(const ora_scene* s, const float* po, const float* pd, const void* prim_hits16,
                            uint64_t n_pts, const float* lights, uint32_t n_lights, uint8_t* occluded,
                            const ora_fog* fog, float* transm, uint64_t* sums, int n_threads) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
    uint64_t n_pairs = 0, n_nodes = 0, n_tests = 0, n_marched = 0;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt) if (nt > 1) reduction(+ : n_pairs, n_nodes, n_tests, n_marched)
    for (int64_t i = 0; i < (int64_t)n_pts; ++i) {
        uint8_t* occ = occluded + (size_t)i * n_lights;
        float* tr = transm ? transm + (size_t)i * n_lights : NULL;
        const Hit16 h = H[i];
        if (h.tri_id == 0xFFFFFFFFu) {
            for (uint32_t l = 0; l < n_lights; ++l) { occ[l] = 0xFF; if (tr) tr[l] = 1.0f; }
            continue;
        }
        const V3 o = v3(po[4 * i], po[4 * i + 1], po[4 * i + 2]);
        const V3 d = v3(pd[4 * i], pd[4 * i + 1], pd[4 * i + 2]);
        const V3 p = o + h.t * d;
        const uint* ix = &s->tris[4 * (size_t)h.tri_id];
        const V3 pt0 = s->verts[ix[0]];
        const V3 e0 = s->verts[ix[1]] - pt0, e1 = s->verts[ix[2]] - pt0;
        V3 N = normalize(cross(e0, e1));
        if (dot(d, N) > 0.0f) N = -N;
        for (uint32_t l = 0; l < n_lights; ++l) {
            const V3 L = v3(lights[3 * l], lights[3 * l + 1], lights[3 * l + 2]);
            const V3 dirv = L - p;
            const float dist_sq = dot(dirv, dirv);
            const float dist = sqrtf(dist_sq);
            const V3 w = dirv * (1.0f / sqrtf(dist_sq));
            Ray r = make_ray_offset(p, w, N);
            r.t_min = 0.0f; r.t_max = dist;
            const bool blocked = scene_trace(*s, r, true);
            occ[l] = blocked ? 1 : 0;
            ++n_pairs; n_nodes += r.nodes_visited; n_tests += r.tri_tests;
            if (tr) {
                if (blocked) tr[l] = 0.0f;
                else { tr[l] = fog ? glsl_calc_transm(*fog, p, w, dist_sq) : 1.0f; ++n_marched; }
            }
        }
    }
    if (sums) { sums[0] = n_pairs; sums[1] = n_nodes; sums[2] = n_tests; sums[3] = n_marched; }
}

int ora_max_threads(void) { return omp_get_max_threads(); }

float ora_sample_density(const ora_fog* fog, const float pos[3]) {
    return sample_density(*fog, v3(pos[0], pos[1], pos[2]));
}

void ora_transmittance(const ora_fog* fog, const float* sp, const float* sd, uint64_t n, float* transm,
                       int n_threads) {
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        transm[i] = glsl_calc_transm(*fog, v3(sp[4 * i], sp[4 * i + 1], sp[4 * i + 2]),
                                     v3(sd[4 * i], sd[4 * i + 1], sd[4 * i + 2]), sd[4 * i + 3]);
    }
}

/* DensityField::computePiDensity, Source/Fog/DensityField.cpp:271-327.  out: res_x*res_y*res_z
 * floats, index x + y*res_x + z*res_x*res_y.  PARITY UNPINNED against the reference (its 268 MB
 * pi_df.3dt was never digested); every piece it is made of is pinned. */
void ora_compute_pi_density(const ora_scene* s, const ora_fog* fog, const float cam[12], int res_x, int res_y,
                            float* out, int n_threads) {
    const V3 pos = v3(cam[0], cam[1], cam[2]), bl = v3(cam[3], cam[4], cam[5]);
    const V3 sx = v3(cam[6], cam[7], cam[8]), sy = v3(cam[9], cam[10], cam[11]);
    const Box fbox(fog_min(*fog), fog_max(*fog));
    const int res_z = fog->res[2];
    const size_t plane = (size_t)res_x * res_y;
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nt) if (nt > 1)
    for (int y = 0; y < res_y; ++y)
        for (int x = 0; x < res_x; ++x) {
            const V3 dir = (bl + (x + 0.5f) * sx) + (y + 0.5f) * sy;      /* Camera.cpp:66 */
            Ray ray = make_ray(pos, normalize(dir));
            float entr, exit;
            if (box_intersect(fbox, ray, entr, exit)) {
                scene_trace(*s, ray, false);
                const float t_min = gmax(entr, 0.0f);
                const float t_max = gmin(exit, ray.dist);
                const int n_intervals = res_z * 4;
                const float dt = (t_max - t_min) / n_intervals;
                float prev_dens = sample_density(*fog, ray.o + t_min * ray.d);
                float dens = 0.0f;
                for (int i = 1; i <= n_intervals; ++i) {
                    const float t = t_min + i * dt;
                    const float curr_dens = sample_density(*fog, ray.o + t * ray.d);
                    dens += 0.5f * (curr_dens + prev_dens);
                    prev_dens = curr_dens;
                    if (2 == i % 4) out[x + (size_t)y * res_x + (size_t)(i / 4) * plane] = dens * dt;
                }
            } else {
                for (int z = 0; z < res_z; ++z) out[x + (size_t)y * res_x + (size_t)z * plane] = 0.0f;
            }
        }
}

/* glm::simplex(vec3), Source/Include/GLM/gtc/noise.inl:679-751, scalarised */
static float glm_simplex3(V3 v) {
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    const float Dx = 0.0f, Dy = 0.5f, Dz = 1.0f, Dw = 2.0f;
    /* First corner */
    const float s0 = dot(v, v3(Cy, Cy, Cy));
    V3 i = v3(floorf(v.x + s0), floorf(v.y + s0), floorf(v.z + s0));
    const float s1 = dot(i, v3(Cx, Cx, Cx));
    const V3 x0 = v3(v.x - i.x + s1, v.y - i.y + s1, v.z - i.z + s1);
    /* Other corners: g = step(x0.yzx, x0); l = 1 - g; i1 = min(g, l.zxy); i2 = max(g, l.zxy) */
    const V3 g = v3(stepf(x0.y, x0.x), stepf(x0.z, x0.y), stepf(x0.x, x0.z));
    const V3 l = v3(1.0f - g.x, 1.0f - g.y, 1.0f - g.z);
    const V3 i1 = v3(gmin(g.x, l.z), gmin(g.y, l.x), gmin(g.z, l.y));
    const V3 i2 = v3(gmax(g.x, l.z), gmax(g.y, l.x), gmax(g.z, l.y));
    const V3 x1 = v3(x0.x - i1.x + Cx, x0.y - i1.y + Cx, x0.z - i1.z + Cx);
    const V3 x2 = v3(x0.x - i2.x + Cy, x0.y - i2.y + Cy, x0.z - i2.z + Cy);
    const V3 x3 = v3(x0.x - Dy, x0.y - Dy, x0.z - Dy);
    /* Permutations */
    i = v3(mod289f(i.x), mod289f(i.y), mod289f(i.z));
    float p[4];
    {
        const float zz[4] = {0.0f, i1.z, i2.z, 1.0f}, yy[4] = {0.0f, i1.y, i2.y, 1.0f}, xx[4] = {0.0f, i1.x, i2.x, 1.0f};
        for (int k = 0; k < 4; ++k)
            p[k] = permutef(permutef(permutef(i.z + zz[k]) + i.y + yy[k]) + i.x + xx[k]);
    }
    /* Gradients */
    const float n_ = (float)0.142857142857;
    const float nsx = n_ * Dw - Dx, nsy = n_ * Dy - Dz, nsz = n_ * Dz - Dx;   /* ns = n_ * D.wyz - D.xzx */
    float gx[4], gy[4], gh[4];
    float b0[4], b1[4], sa0[4], sa1[4];
    for (int k = 0; k < 4; ++k) {
        const float j = p[k] - 49.0f * floorf(p[k] * nsz * nsz);
        const float x_ = floorf(j * nsz);
        const float y_ = floorf(j - 7.0f * x_);
        const float x = x_ * nsx + nsy, y = y_ * nsx + nsy;
        const float h = 1.0f - fabsf(x) - fabsf(y);
        gx[k] = x; gy[k] = y; gh[k] = h;
    }
    /* b0 = (x.xy, y.xy), b1 = (x.zw, y.zw) */
    b0[0] = gx[0]; b0[1] = gx[1]; b0[2] = gy[0]; b0[3] = gy[1];
    b1[0] = gx[2]; b1[1] = gx[3]; b1[2] = gy[2]; b1[3] = gy[3];
    for (int k = 0; k < 4; ++k) {
        sa0[k] = floorf(b0[k]) * 2.0f + 1.0f;
        sa1[k] = floorf(b1[k]) * 2.0f + 1.0f;
    }
    float sh[4];
    for (int k = 0; k < 4; ++k) sh[k] = -stepf(gh[k], 0.0f);
    /* a0 = b0.xzyw + s0.xzyw * sh.xxyy ; a1 = b1.xzyw + s1.xzyw * sh.zzww */
    const float a0[4] = {b0[0] + sa0[0] * sh[0], b0[2] + sa0[2] * sh[0], b0[1] + sa0[1] * sh[1], b0[3] + sa0[3] * sh[1]};
    const float a1[4] = {b1[0] + sa1[0] * sh[2], b1[2] + sa1[2] * sh[2], b1[1] + sa1[1] * sh[3], b1[3] + sa1[3] * sh[3]};
    V3 p0 = v3(a0[0], a0[1], gh[0]), p1 = v3(a0[2], a0[3], gh[1]);
    V3 p2 = v3(a1[0], a1[1], gh[2]), p3 = v3(a1[2], a1[3], gh[3]);
    /* Normalise gradients */
    p0 = p0 * taylor_inv_sqrt(dot(p0, p0)); p1 = p1 * taylor_inv_sqrt(dot(p1, p1));
    p2 = p2 * taylor_inv_sqrt(dot(p2, p2)); p3 = p3 * taylor_inv_sqrt(dot(p3, p3));
    /* Mix final noise value */
    float m[4] = {gmax(0.6f - dot(x0, x0), 0.0f), gmax(0.6f - dot(x1, x1), 0.0f),
                  gmax(0.6f - dot(x2, x2), 0.0f), gmax(0.6f - dot(x3, x3), 0.0f)};
    for (int k = 0; k < 4; ++k) m[k] = m[k] * m[k];
    const float q[4] = {dot(p0, x0), dot(p1, x1), dot(p2, x2), dot(p3, x3)};
    /* dot(m*m, q) on vec4: (x+y)+(z+w), func_geometric.inl:78-82 */
    const float t[4] = {m[0] * m[0] * q[0], m[1] * m[1] * q[1], m[2] * m[2] * q[2], m[3] * m[3] * q[3]};
    return 42.0f * ((t[0] + t[1]) + (t[2] + t[3]));
}

void ora_make_density(const int res[3], float freq, float ampl, uint8_t* out) {
    /* Source/Fog/DensityField.cpp:27-76 */
    float max_dens = 0.0f;
    const float x_norm = 1.0f / (res[0] - 1), y_norm = 1.0f / (res[1] - 1), z_norm = 1.0f / (res[2] - 1);
    for (int z = 0; z < res[2]; ++z)
        for (int y = 0; y < res[1]; ++y)
            for (int x = 0; x < res[0]; ++x) {
                const V3 norm_pos = v3(x * x_norm, y * y_norm, z * z_norm);
                float sum = 0.0f, curr_freq = freq, curr_ampl = ampl;
                for (int oct = 0; oct < N_OCTAVES; ++oct) {
                    float val = glm_simplex3(curr_freq * norm_pos);
                    val = 0.25f * (val + 1.0f);
                    sum += curr_ampl * val;
                    curr_freq *= 2.0f;
                    curr_ampl /= 2.0f;
                }
                const float dens_val = sum / N_OCTAVES;
                const uint8_t byte_dens = (uint8_t)(255.0f * dens_val);
                max_dens = gmax(max_dens, dens_val);
                out[x + y * res[0] + z * res[0] * res[1]] = byte_dens;
            }
    const float inv_max_dens = 1.0f / max_dens;
    const size_t n = (size_t)res[0] * res[1] * res[2];
    for (size_t k = 0; k < n; ++k) {
        const uint8_t old_b = out[k];
        const float new_dens = gmin(old_b * inv_max_dens / 255.0f, 1.0f);
        out[k] = (uint8_t)(255.0f * new_dens);
    }
}

}  /* extern "C" */

namespace {
/* Where the walk's random numbers come from.  Stream mode: one caller-supplied sequence consumed in
 * walk order (the reference's single global std::mt19937, Source/Common/Random.cpp:11-18).
 * Substream mode (the build's own definition, for the batched device walk): path p draws from a
 * counter-based stream of its own, u(p, k) = top 24 bits of splitmix64(key(p) + k * gamma) * 2^-24 with
 * key(p) = splitmix64(seed ^ splitmix64(p)), so paths can be walked in any order or all at once. */
inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct WalkDraw {
    const float* rng = nullptr; uint64_t n_rng = 0, rp = 0;   /* stream mode */
    bool substreams = false; uint64_t seed = 0, key = 0; uint32_t k = 0;
    bool starved = false;
    uint64_t n_drawn = 0;
    void begin_path(uint p) {
        if (substreams) { key = splitmix64(seed ^ splitmix64((uint64_t)p)); k = 0; }
    }
    float operator()() {
        ++n_drawn;
        if (substreams) {
            const uint64_t z = splitmix64(key + (uint64_t)k * 0x9E3779B97F4A7C15ull);
            ++k;
            return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-8f;   /* 2^-24: [0, 1) */
        }
        if (rp < n_rng) return rng[rp++];
        starved = true;
        return 0.0f;
    }
};
/* The walk's transcendental functions.  Stream mode calls the host's float libm (as the reference calls
 * MSVC's).  Substream mode rounds the double-precision function to float: the correctly rounded float
 * result (up to double rounding at exact ties of the double result, probability ~2^-29 per call),
 * which the device can reproduce bit for bit; a float libm differs between platforms in the last place. */
struct WalkMath {
    bool cr;
    float cos_(float x) const { return cr ? (float)cos((double)x) : cosf(x); }
    float sin_(float x) const { return cr ? (float)sin((double)x) : sinf(x); }
    float log_(float x) const { return cr ? (float)log((double)x) : logf(x); }
    float pow_(float x, float y) const { return cr ? (float)pow((double)x, (double)y) : powf(x, y); }
    V3 vec_sin_cos(float phi, float sin_the, float cos_the) const {                  /* RTBase.h:44-50 */
        return v3(sin_the * cos_(phi), sin_the * sin_(phi), cos_the);
    }
};

/* Source/RT/PhotonTracer.cpp:78-188.  PARITY UNPINNED. */
int photon_walk(const ora_scene* s, const float* materials, const ora_fog* fog,
                float abs_k, float sca_k, float maj_ext_k, int fog_enabled, const float ppl_pos[3],
                const float ppl_intens[3], const float shoot_dir[3], int max_vpl_count,
                WalkDraw& draw, const WalkMath M, void* vpls_out, uint32_t* n_paths_out) {
    struct VPL96 { float w_pos[3]; uint type; float intens[3]; float sca_k; float w_inc[3]; uint path_id;
                   float w_norm[3]; uint pad1; float k_d[3]; uint pad2; float k_s[3]; float n_s; };
    VPL96* out = (VPL96*)vpls_out;
    bool& starved = draw.starved;
    const V3 src = v3(ppl_pos[0], ppl_pos[1], ppl_pos[2]);
    const V3 src_int = v3(ppl_intens[0], ppl_intens[1], ppl_intens[2]);
    const V3 sdir = v3(shoot_dir[0], shoot_dir[1], shoot_dir[2]);
    Box fog_box;
    if (fog) fog_box = Box(fog_min(*fog), fog_max(*fog));
    const float ext_k_sum = abs_k + sca_k;                            /* FogVolume.cpp:34,44 */
    int size = 0;
    uint n_paths = 0;
    do {
        ++n_paths;
        draw.begin_path(n_paths - 1);
        /* :86-88 — the draw happens only when source.z <= MAX_FOG_HEIGHT; `up` and the
         * tangent space built from it are never used. */
        if (src.z <= MAX_FOG_HEIGHT) { (void)draw(); }
        (void)sdir;
        /* :90 — both arguments are RNG draws with unspecified order; g++ (and MSVC) on
         * x86-64 evaluate right to left, so the FIRST draw is u1 (cos_the), the second u0. */
        const float u1 = draw();
        const float u0 = draw();
        const float phi = TWO_PI * u0;
        const float sin_the = sqrtf(1.0f - u1 * u1);
        const V3 dir = M.vec_sin_cos(phi, sin_the, u1);
        const float pdf = 0.5f * INV_PI;
        Ray ray = make_ray(src, dir);
        const float ia = 1.0f / pdf;
        V3 attenuation = v3(ia, ia, ia);
        for (int n_bounces = 1; n_bounces <= N_GI_BOUNCES; ++n_bounces) {
            const bool hit = scene_trace(*s, ray, false);
            /* Scene::traceFog (Scene.cpp:245-251) sees the ray's t_min/t_max as traversal left them */
            float f_entr = FLT_MAX, f_exit = -FLT_MAX;
            const bool fog_hit = (fog_enabled && fog) ? box_intersect(fog_box, ray, f_entr, f_exit) : false;
            if (fog_hit) {
                ray.t_min = gmax(f_entr, 0.0f);
                ray.t_max = gmin(f_exit, ray.dist);
                /* calcEventDistWT, :62-76 */
                bool event = false;
                float t_event = 0.0f;
                if (0.0f != maj_ext_k) {
                    float d_event = ray.t_min, p_event = 0.0f;
                    while (d_event < ray.t_max && p_event < draw()) {
                        const float dt = -M.log_(1.0f - draw()) / maj_ext_k;
                        d_event += dt;
                        const V3 s_pos = pt_at(ray, d_event);
                        const float ext_k = ext_k_sum * sample_density(*fog, s_pos);
                        p_event = ext_k / maj_ext_k;
                        if (starved) break;
                    }
                    event = d_event < ray.t_max;
                    t_event = d_event;
                }
                if (event) {
                    const V3 event_pt = pt_at(ray, t_event);
                    const float sk = sca_k * sample_density(*fog, event_pt);       /* FogVolume.cpp:38-41 */
                    const float sca_albedo = sca_k / ext_k_sum;
                    if (draw() < sca_albedo) {
                        const V3 I = ray.d;
                        const V3 vi = src_int * attenuation;
                        VPL96 v; memset(&v, 0, sizeof(v));                           /* PointLight.cpp:15-17 */
                        v.w_pos[0] = event_pt.x; v.w_pos[1] = event_pt.y; v.w_pos[2] = event_pt.z; v.type = 1;
                        v.intens[0] = vi.x; v.intens[1] = vi.y; v.intens[2] = vi.z; v.sca_k = sk;
                        v.w_inc[0] = I.x; v.w_inc[1] = I.y; v.w_inc[2] = I.z; v.path_id = n_paths - 1;
                        out[size++] = v;
                        if (size == max_vpl_count) break;
                        /* HGPhaseFunc::importanceSample, :49-58 */
                        const float hphi = TWO_PI * draw();
                        const float uu = draw();
                        const float nn = 1.0f - HG_G * HG_G;
                        const float dd = 1.0f - HG_G + 2.0f * HG_G * uu;
                        const float q = nn / dd;
                        const float cos_the = 0.5f / fabsf(HG_G) * (1.0f + HG_G * HG_G - q * q);
                        const float sin_t = sqrtf(1.0f - cos_the * cos_the);
                        const M3 m2m = span_tangent_space(I);
                        const V3 O = mul(m2m, M.vec_sin_cos(hphi, sin_t, cos_the));
                        attenuation = attenuation * 1.0f;
                        ray = make_ray(event_pt, O);
                        continue;
                    } else break;
                }
            }
            if (hit) {
                const V3 hit_pt = pt_at(ray, ray.dist);
                const V3 norm = ray.normal;
                if (dot(ray.d, ray.normal) >= 0.0f || hit_pt.z > MAX_FOG_HEIGHT) break;
                const float* mat = materials + 12 * (size_t)s->tris[4 * (size_t)ray.tri_id + 3];
                const V3 k_d = v3(mat[0], mat[1], mat[2]), k_s = v3(mat[4], mat[5], mat[6]);
                const float n_s = mat[7];
                if (mat[8] > 0.0f || mat[9] > 0.0f || mat[10] > 0.0f) break;          /* isEmissive */
                if (draw() < SURVIVAL_P_RR) {
                    attenuation = attenuation / SURVIVAL_P_RR;
                    const V3 I = -ray.d;
                    const V3 vi = src_int * attenuation;
                    const V3 lifted = hit_pt + 1E-4f * norm;                            /* PointLight.cpp:21 */
                    VPL96 v; memset(&v, 0, sizeof(v));
                    v.w_pos[0] = lifted.x; v.w_pos[1] = lifted.y; v.w_pos[2] = lifted.z; v.type = 2;
                    v.intens[0] = vi.x; v.intens[1] = vi.y; v.intens[2] = vi.z;
                    v.w_inc[0] = I.x; v.w_inc[1] = I.y; v.w_inc[2] = I.z; v.path_id = n_paths - 1;
                    v.w_norm[0] = norm.x; v.w_norm[1] = norm.y; v.w_norm[2] = norm.z;
                    v.k_d[0] = k_d.x; v.k_d[1] = k_d.y; v.k_d[2] = k_d.z;
                    v.k_s[0] = k_s.x; v.k_s[1] = k_s.y; v.k_s[2] = k_s.z; v.n_s = n_s;
                    out[size++] = v;
                    if (size == max_vpl_count) break;
                    /* importanceSampleBRDF, RTBase.cpp:70-109 */
                    const float bu1 = draw(), bu2 = draw();
                    const V3 lum = v3(0.2126f, 0.7152f, 0.0722f);
                    const float dw = dot(k_d, lum), gw = dot(k_s, lum);
                    const float sp_diffuse = dw / (dw + gw);
                    const float u_rand = draw();
                    V3 O, throughput;
                    if (u_rand < sp_diffuse) {
                        const M3 m2m = span_tangent_space(norm);
                        /* genCosRandHemiDir, RTBase.cpp:27-36 */
                        const V3 micro_O = M.vec_sin_cos(TWO_PI * bu1, sqrtf(1.0f - bu2), sqrtf(bu2));
                        O = mul(m2m, micro_O);
                        throughput = (k_d * PI) / (micro_O.z * sp_diffuse);
                    } else {
                        const V3 S = -I + (2.0f * dot(I, norm)) * norm;                 /* reflect, :22-24 */
                        const M3 m2m = span_tangent_space(S);
                        /* genCosPowRandHemiDir, RTBase.cpp:39-49 */
                        const float lobe_cos = M.pow_(bu2, 1.0f / (n_s + 1.0f));
                        const V3 micro_O = M.vec_sin_cos(TWO_PI * bu1, sqrtf(1.0f - lobe_cos * lobe_cos), lobe_cos);
                        O = mul(m2m, micro_O);
                        const float sp_glossy = 1.0f - sp_diffuse;
                        throughput = (k_s * TWO_PI) / ((n_s + 1.0f) * sp_glossy);
                    }
                    const float cos_the = dot(O, norm);
                    attenuation = attenuation * (throughput * cos_the);
                    ray = make_ray_offset(hit_pt, O, norm);
                    continue;
                } else break;
            } else break;
        }
        if (starved) break;
    } while (size < max_vpl_count && n_paths < (uint)(100 * max_vpl_count));
    if (n_paths_out) *n_paths_out = n_paths;
    if (starved) return -1;
    if (size < max_vpl_count) return 0;                                /* :181-184 */
    /* LightArray::normalizeIntensity, Source/VPL/LightArray.hpp:46-66 */
    const float inv_n_paths = 1.0f / (int)n_paths;
    uint old_path_id = out[0].path_id, new_path_id = 0;
    for (int i = 0; i < size; ++i) {
        if (old_path_id == out[i].path_id) out[i].path_id = new_path_id;
        else { old_path_id = out[i].path_id; out[i].path_id = ++new_path_id; }
        out[i].intens[0] *= inv_n_paths; out[i].intens[1] *= inv_n_paths; out[i].intens[2] *= inv_n_paths;
    }
    return size;
}
}  /* namespace */

extern "C" {

/* PhotonTracer::trace with an explicit RNG stream consumed in walk order (the reference's order). */
int ora_photon_trace(const ora_scene* s, const float* materials, uint32_t n_mat, const ora_fog* fog,
                     float abs_k, float sca_k, float maj_ext_k, int fog_enabled, const float ppl_pos[3],
                     const float ppl_intens[3], const float shoot_dir[3], int max_vpl_count,
                     const float* rng, uint64_t n_rng, void* vpls_out, uint32_t* n_paths_out,
                     uint64_t* n_rng_used_out) {
    (void)n_mat;
    WalkDraw draw;
    draw.rng = rng; draw.n_rng = n_rng;
    const int n = photon_walk(s, materials, fog, abs_k, sca_k, maj_ext_k, fog_enabled, ppl_pos, ppl_intens,
                              shoot_dir, max_vpl_count, draw, WalkMath{false}, vpls_out, n_paths_out);
    if (n_rng_used_out) *n_rng_used_out = draw.rp;
    return n;
}

/* The same walk in substream mode (WalkDraw, WalkMath above): what gigl_photon_trace_batch computes. */
int ora_photon_trace_substreams(const ora_scene* s, const float* materials, uint32_t n_mat, const ora_fog* fog,
                                float abs_k, float sca_k, float maj_ext_k, int fog_enabled,
                                const float ppl_pos[3], const float ppl_intens[3], const float shoot_dir[3],
                                int max_vpl_count, uint64_t seed, void* vpls_out, uint32_t* n_paths_out,
                                uint64_t* n_draws_out) {
    (void)n_mat;
    WalkDraw draw;
    draw.substreams = true; draw.seed = seed;
    const int n = photon_walk(s, materials, fog, abs_k, sca_k, maj_ext_k, fog_enabled, ppl_pos, ppl_intens,
                              shoot_dir, max_vpl_count, draw, WalkMath{true}, vpls_out, n_paths_out);
    if (n_draws_out) *n_draws_out = draw.n_drawn;
    return n;
}

}  /* extern "C" */

/* ---- software shading stage: Source/Shaders/Surface.frag:205-343, Combine.frag:62-77 ---------
 * PARITY UNPINNED: GLSL cannot run here.  Stand-ins for the GL pipeline, all stated in DESIGN.md
 * "Shading stage": the G-buffer position is the primary hit point o + t*d, the G-buffer normal is
 * the geometric normal turned to face the camera, shadow-map visibility is the any-hit result of
 * the shadow batch, texture(pi_dens) is nearest in x,y (pixel centre = texel centre) and linear in
 * z with edge clamping (DensityField.cpp:376-382). */
namespace {
struct VPL96 { float w_pos[3]; uint type; float intens[3]; float sca_k; float w_inc[3]; uint path_id;
               float w_norm[3]; uint pad1; float k_d[3]; uint pad2; float k_s[3]; float n_s; };
inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
/* phongBRDF, Surface.frag:205-217; reflect(I,N) = I - 2*dot(N,I)*N */
inline V3 glsl_phong_brdf(V3 I, V3 N, V3 O, V3 k_d, V3 k_s, float n_s) {
    V3 brdf = k_d;
    if (n_s > 0.0f) {
        const V3 mI = -I;
        const V3 S = mI - (2.0f * dot(N, mI)) * N;
        const float c = dot(O, S);
        if (c > 0.0f) brdf = brdf + k_s * powf(c, n_s);
    }
    return brdf;
}
/* evalPhaseHG, Surface.frag:219-223 */
inline float glsl_phase_hg(float cos_the) {
    const float g = 0.25f;
    const float base = 1.0f + g * g - 2.0f * g * cos_the;
    return 0.25f * 0.318309873f * (1.0f - g * g) / (base * sqrtf(base));
}
/* computeLe, Surface.frag:225-242 */
inline V3 glsl_compute_le(const VPL96& v, V3 O, float sca_albedo) {
    if (v.type == 1u) return (glsl_phase_hg(dot(O, ld3(v.w_inc))) * sca_albedo) * ld3(v.intens);
    if (v.type == 2u) {
        const float cos_out = smax(0.0f, dot(O, ld3(v.w_norm)));
        return (glsl_phong_brdf(ld3(v.w_inc), ld3(v.w_norm), O, ld3(v.k_d), ld3(v.k_s), v.n_s) * ld3(v.intens)) * cos_out;
    }
    return v3(0.0f, 0.0f, 0.0f);
}
inline float pi_dens_fetch(const float* pi, int res_x, int res_y, int res_z, int x, int y, float z) {
    const size_t plane = (size_t)res_x * res_y;
    float tz = z * (float)res_z - 0.5f;
    tz = smin(smax(tz, 0.0f), (float)(res_z - 1));                   /* CLAMP_TO_EDGE */
    const float fz = floorf(tz);
    const int z0 = (int)fz, z1 = z0 + 1 < res_z ? z0 + 1 : res_z - 1;
    const float f = tz - fz;
    const float a = pi[x + (size_t)y * res_x + (size_t)z0 * plane], b = pi[x + (size_t)y * res_x + (size_t)z1 * plane];
    return (1.0f - f) * a + f * b;
}
}  // namespace

extern "C" {

void ora_shade_surface(const ora_scene* s, const ora_fog* fog, const float cam_pos[3], int res_x, int res_y,
                       uint64_t first, const float* po, const float* pd, const void* prim_hits16, uint64_t n_pts,
                       const float* materials, const float* ppls, uint32_t n_ppls, const void* vpls,
                       uint32_t n_vpls, const uint8_t* occluded, const float* transm, const float* pi_dens,
                       int pi_res_z, float sca_albedo, int gi_enabled, int clamp_rsq, int frame_id,
                       float* accum_rgba, float* fog_dist, int n_threads) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const VPL96* V = (const VPL96*)vpls;
    const uint32_t n_lights = n_ppls + n_vpls;
    const V3 cam = ld3(cam_pos);
    const float ext_k = fog ? fog->ext_k : 0.0f;
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n_pts; ++i) {
        V3 col = v3(0.0f, 0.0f, 0.0f);
        float f_min = 0.0f, f_max = 0.0f;
        const Hit16 h = H[i];
        if (frame_id < 30 && h.tri_id != 0xFFFFFFFFu) {                 /* Surface.frag:292 */
            const V3 o = ld3(po + 4 * i), d = ld3(pd + 4 * i);
            const V3 w_pos = o + h.t * d;
            const V3 ray_d = normalize(w_pos - cam);                      /* :296 */
            float transm_frag = 1.0f;
            if (ext_k > 0.0f) {                                           /* :297-315 */
                const float t_frag = length(w_pos - cam);
                float t_min, t_max;
                if (glsl_intersect_bbox(*fog, cam, ray_d, t_frag, t_min, t_max)) {
                    t_min = smax(t_min, 0.0f);
                    t_max = smin(t_max, t_frag);
                    const uint64_t pix = first + (uint64_t)i;
                    const float z = (t_frag - t_min) / (t_max - t_min);
                    const float opt_depth = ext_k * pi_dens_fetch(pi_dens, res_x, res_y, pi_res_z,
                                                                  (int)(pix % (uint64_t)res_x), (int)(pix / (uint64_t)res_x), z);
                    transm_frag = expf(-opt_depth);
                } else { t_min = t_max = 0.0f; }
                f_min = t_min; f_max = t_max;
            }
            if (transm_frag > 0.001f) {                                    /* :316-331 */
                const uint* ix = &s->tris[4 * (size_t)h.tri_id];
                const V3 pt0 = s->verts[ix[0]];
                V3 N = normalize(cross(s->verts[ix[1]] - pt0, s->verts[ix[2]] - pt0));
                if (dot(d, N) > 0.0f) N = -N;
                const float* m = materials + 12 * (size_t)ix[3];
                const V3 k_d = ld3(m), k_s = ld3(m + 4);
                const float n_s = m[7];
                const V3 O = -ray_d;
                for (uint32_t l = 0; l < n_lights; ++l) {
                    const bool is_ppl = l < n_ppls;
                    if (is_ppl ? !clamp_rsq : !gi_enabled) continue;
                    if (occluded[(size_t)i * n_lights + l]) continue;     /* visibility > 0.0, :253/:276 */
                    const V3 L = is_ppl ? ld3(ppls + 8 * l) : ld3(V[l - n_ppls].w_pos);
                    const V3 dd = w_pos - L;
                    const float dist_sq = dot(dd, dd);
                    const V3 I = normalize(-dd);
                    const float tr = transm ? transm[(size_t)i * n_lights + l] : 1.0f;
                    V3 Li;
                    if (is_ppl) {                                          /* :255-258 */
                        const float falloff = 1.0f / dist_sq;
                        Li = (tr * ld3(ppls + 8 * l + 4)) * falloff;
                    } else {                                               /* :278-281 */
                        const float falloff = 1.0f / smax(dist_sq, 75.0f * 75.0f);
                        Li = (tr * glsl_compute_le(V[l - n_ppls], -I, sca_albedo)) * falloff;
                    }
                    const float cos_inc = smax(0.0f, dot(I, N));
                    const V3 c = (glsl_phong_brdf(I, N, O, k_d, k_s, n_s) * Li) * cos_inc;
                    col = col + transm_frag * c;                           /* :322, :328 */
                }
            }
        }
        if (frame_id < 30) {
            float* a = accum_rgba + 4 * i;
            if (frame_id > 0) { col.x += a[0]; col.y += a[1]; col.z += a[2]; }   /* :332-337 */
            a[0] = col.x; a[1] = col.y; a[2] = col.z; a[3] = 1.0f;                /* :339 */
            if (fog_dist && ext_k > 0.0f) { fog_dist[2 * i] = f_min; fog_dist[2 * i + 1] = f_max; }
        }
    }
}

/* Combine.frag:72-76 without the volume term: divide by the frame count, then tone-map. */
void ora_tonemap(const float* accum_rgba, uint64_t n, int frame_id, int exposure, float* rgb) {
    const float div = (float)(1 + (frame_id < 30 ? frame_id : 30));
    for (uint64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = 1.0f - expf(-(float)exposure * (accum_rgba[4 * i + c] / div));
}

}  /* extern "C" */

/* ---- Volume.frag main() (Source/Shaders/Volume.frag:268-314).  PARITY UNPINNED, same stand-ins as
 * the surface pass; shadow-map visibility = any-hit rays from the sample point (ora_shadow_points). */
namespace {
struct VolPixel { V3 ray_d; float t_min, t_max, z_offset; int tx, ty; bool fog; };
struct VolSample { V3 pos; float transm; bool valid; };
inline VolPixel volume_pixel(const float* po, const float* pd, const Hit16* H, const float* fog_dist,
                             const float* rnd_offsets, V3 cam, int res_x, int frame_id, int hx, int hy) {
    VolPixel px;
    const size_t full = (size_t)(2 * hx + 1) + (size_t)(2 * hy + 1) * res_x;     /* :90, :95 */
    px.t_min = fog_dist[2 * full]; px.t_max = fog_dist[2 * full + 1];
    px.fog = px.t_max > 0.0f && frame_id < 30;                                     /* :270, :275 */
    px.ray_d = v3(0.0f, 0.0f, 0.0f);
    px.z_offset = 0.0f;
    /* gl_FragCoord.xy / CAM_RES with `#define CAM_RES 1024 / 2` = (xy / 1024) / 2 (:4, :285) */
    px.tx = hx >> 1; px.ty = hy >> 1;
    if (px.fog) {
        const V3 w_pos = ld3(po + 4 * full) + H[full].t * ld3(pd + 4 * full);
        px.ray_d = normalize(w_pos - cam);                                         /* :280 */
        px.z_offset = rnd_offsets[hx + (size_t)hy * (res_x / 2)];                  /* :282 */
    }
    return px;
}
inline VolSample volume_sample(const ora_fog& F, const VolPixel& px, V3 cam, const float* halton,
                               const float* pi_dens, int res_x, int res_y, int pi_res_z, int n_samples,
                               int frame_id, int s) {
    VolSample smp;
    const float hz = halton[s + n_samples * frame_id] + px.z_offset;
    const float z = smin(smax(hz, 0.0f), 0.995f);                                  /* :286 */
    const float t = px.t_min + z * (px.t_max - px.t_min);
    smp.pos = cam + t * px.ray_d;
    smp.transm = expf(-(F.ext_k * pi_dens_fetch(pi_dens, res_x, res_y, pi_res_z, px.tx, px.ty, z)));
    smp.valid = smp.transm > 0.001f;                                               /* :292 */
    return smp;
}
/* calcTransm, Volume.frag:159-181 */
inline float glsl_calc_transm_vol(const ora_fog& F, V3 p, float density, V3 dir, float dist_sq, int transm_opt) {
    if (!transm_opt) return glsl_calc_transm(F, p, dir, dist_sq);
    float transm = 1.0f;
    if (F.ext_k > 0.0f) {
        const float max_dist = sqrtf(dist_sq);
        float t_min, t_max;
        if (glsl_intersect_bbox(F, p, dir, max_dist, t_min, t_max)) {
            t_min = smax(t_min, 0.0f);
            t_max = smin(t_max, max_dist);
            transm = expf(-(F.ext_k * density * (t_max - t_min)));
        }
    }
    return transm;
}
}  // namespace

extern "C" {

/* Source/Common/Halton.hpp:5-18, base 2 */
void ora_halton2(float* seq, int n) {
    for (int i = 0; i < n; ++i) {
        seq[i] = 0.0f;
        int den = 2, t = i + 1;
        while (t > 0) {
            seq[i] += 1.0f / den * (t % 2);
            t /= 2;
            den *= 2;
        }
    }
}

void ora_volume_samples(const ora_fog* fog, const float cam_pos[3], int res_x, int res_y, const float* po,
                        const float* pd, const void* prim_hits16, const float* fog_dist, const float* rnd_offsets,
                        const float* halton, const float* pi_dens, int pi_res_z, int gi_enabled, int frame_id,
                        float* points) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const V3 cam = ld3(cam_pos);
    const int hw = res_x / 2, hh = res_y / 2, n_samples = gi_enabled ? 8 : 32;
    for (int hy = 0; hy < hh; ++hy)
        for (int hx = 0; hx < hw; ++hx) {
            const VolPixel px = volume_pixel(po, pd, H, fog_dist, rnd_offsets, cam, res_x, frame_id, hx, hy);
            for (int s = 0; s < n_samples; ++s) {
                float* out = points + 4 * (((size_t)hx + (size_t)hy * hw) * n_samples + s);
                out[0] = out[1] = out[2] = out[3] = 0.0f;
                if (!px.fog) continue;
                const VolSample smp = volume_sample(*fog, px, cam, halton, pi_dens, res_x, res_y, pi_res_z,
                                                    n_samples, frame_id, s);
                if (smp.valid) { out[0] = smp.pos.x; out[1] = smp.pos.y; out[2] = smp.pos.z; out[3] = 1.0f; }
            }
        }
}

/* any-hit rays from explicit points to lights: Ray{p, normalize(L - p)} (RTBase.cpp:111-115), range
 * (0, |L - p|); points with w <= 0 emit no rays (bytes 0xFF). */
void ora_shadow_points(const ora_scene* s, const float* points, uint64_t n_pts, const float* lights,
                       uint32_t n_lights, uint8_t* occluded, int n_threads) {
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt) if (nt > 1)
    for (int64_t i = 0; i < (int64_t)n_pts; ++i) {
        uint8_t* occ = occluded + (size_t)i * n_lights;
        if (!(points[4 * i + 3] > 0.0f)) { for (uint32_t l = 0; l < n_lights; ++l) occ[l] = 0xFF; continue; }
        const V3 p = ld3(points + 4 * i);
        for (uint32_t l = 0; l < n_lights; ++l) {
            const V3 dirv = ld3(lights + 3 * l) - p;
            const float dist_sq = dot(dirv, dirv);
            Ray r = make_ray(p, dirv * (1.0f / sqrtf(dist_sq)));
            r.t_min = 0.0f; r.t_max = sqrtf(dist_sq);
            occ[l] = scene_trace(*s, r, true) ? 1 : 0;
        }
    }
}

void ora_shade_volume(const ora_fog* fog, const float cam_pos[3], int res_x, int res_y, const float* po,
                      const float* pd, const void* prim_hits16, const float* fog_dist, const float* rnd_offsets,
                      const float* halton, const float* pi_dens, int pi_res_z, const float* ppls, uint32_t n_ppls,
                      const void* vpls, uint32_t n_vpls, const uint8_t* occluded, float sca_k, float sca_albedo,
                      int gi_enabled, int clamp_rsq, int transm_opt, int frame_id, float* vol_comp_rgba,
                      int n_threads) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const VPL96* V = (const VPL96*)vpls;
    const uint32_t n_lights = n_ppls + n_vpls;
    const V3 cam = ld3(cam_pos);
    const int hw = res_x / 2, hh = res_y / 2, n_samples = gi_enabled ? 8 : 32;
    const V3 dm = fog_max(*fog) - fog_min(*fog);
    const V3 inv_dims = v3(1.0f / dm.x, 1.0f / dm.y, 1.0f / dm.z);
    const int nt = n_threads < 1 ? 1 : n_threads;
    (void)nt;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nt) if (nt > 1)
    for (int hy = 0; hy < hh; ++hy)
        for (int hx = 0; hx < hw; ++hx) {
            const size_t pix = (size_t)hx + (size_t)hy * hw;
            const VolPixel px = volume_pixel(po, pd, H, fog_dist, rnd_offsets, cam, res_x, frame_id, hx, hy);
            V3 col = v3(0.0f, 0.0f, 0.0f);
            if (px.fog) {
                const V3 O = -px.ray_d;
                for (int s = 0; s < n_samples; ++s) {
                    const VolSample smp = volume_sample(*fog, px, cam, halton, pi_dens, res_x, res_y, pi_res_z,
                                                        n_samples, frame_id, s);
                    if (!smp.valid) continue;
                    const float density = fog_filter(*fog, (smp.pos - fog_min(*fog)) * inv_dims);   /* :121-126 */
                    for (uint32_t l = 0; l < n_lights; ++l) {
                        const bool is_ppl = l < n_ppls;
                        if (is_ppl ? !clamp_rsq : !gi_enabled) continue;
                        if (occluded[(pix * n_samples + s) * n_lights + l]) continue;
                        const V3 L = is_ppl ? ld3(ppls + 8 * l) : ld3(V[l - n_ppls].w_pos);
                        const V3 dd = smp.pos - L;
                        const float dist_sq = dot(dd, dd);
                        const V3 I = normalize(-dd);
                        const float tr = glsl_calc_transm_vol(*fog, smp.pos, density, I, dist_sq, transm_opt);
                        V3 Li;
                        if (is_ppl) Li = (tr * ld3(ppls + 8 * l + 4)) * (1.0f / dist_sq);
                        else Li = (tr * glsl_compute_le(V[l - n_ppls], -I, sca_albedo)) * (1.0f / smax(dist_sq, 75.0f * 75.0f));
                        const float cos_the = -dot(I, O);
                        const V3 c = ((glsl_phase_hg(cos_the) * sca_k) * density) * Li;
                        col = col + smp.transm * c;
                    }
                }
                col = col * ((px.t_max - px.t_min) / (float)n_samples);
            }
            float* out = vol_comp_rgba + 4 * pix;
            out[0] = col.x; out[1] = col.y; out[2] = col.z; out[3] = 1.0f;
        }
}

}  /* extern "C" */

/* ---- Combine.frag (Source/Shaders/Combine.frag:23-77): bilateral upsampling of the half-resolution
 * volume term, accumulation, normalisation, tone mapping.  PARITY UNPINNED.  Stand-ins: depth_buf =
 * the GL depth of the primary hit point (lookAt + glm::perspective, Camera.cpp:30-31,
 * matrix_transform.inl:236-243) quantised to DEPTH_COMPONENT24 (GLTexture2D.h:27), 1.0 where the
 * primary ray missed; texelFetch outside the image = 0 (robust access); texture(vol_comp) = bilinear,
 * CLAMP_TO_BORDER with a zero border (GLTexture2D.hpp:22-33). */
namespace {
inline float gl_depth24(V3 w_pos, V3 cam, V3 fwd, float t_near, float t_far) {
    const float z_e = -dot(fwd, w_pos - cam);
    const float A = -(t_far + t_near) / (t_far - t_near), B = -(2.0f * t_far * t_near) / (t_far - t_near);
    const float z_ndc = (A * z_e + B) / -z_e;
    const float d = 0.5f * z_ndc + 0.5f;
    return floorf(d * 16777215.0f + 0.5f) / 16777215.0f;
}
struct C4 { float r, g, b; };
inline C4 vol_texel(const float* vol, int hw, int hh, int x, int y) {
    if (x < 0 || x >= hw || y < 0 || y >= hh) return C4{0.0f, 0.0f, 0.0f};
    const float* p = vol + 4 * ((size_t)x + (size_t)y * hw);
    return C4{p[0], p[1], p[2]};
}
inline C4 vol_bilinear(const float* vol, int hw, int hh, float u, float v) {   /* u, v in texels, centre at +0.5 */
    const float fu = u - 0.5f, fv = v - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float a = fu - x0f, b = fv - y0f;
    const C4 t00 = vol_texel(vol, hw, hh, x0, y0), t10 = vol_texel(vol, hw, hh, x0 + 1, y0);
    const C4 t01 = vol_texel(vol, hw, hh, x0, y0 + 1), t11 = vol_texel(vol, hw, hh, x0 + 1, y0 + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return C4{((w00 * t00.r + w10 * t10.r) + w01 * t01.r) + w11 * t11.r,
              ((w00 * t00.g + w10 * t10.g) + w01 * t01.g) + w11 * t11.g,
              ((w00 * t00.b + w10 * t10.b) + w01 * t01.b) + w11 * t11.b};
}
}  // namespace

extern "C" {

void ora_depth_buffer(const float cam_pos[3], const float cam_dir[3], float t_near, float t_far, const float* po,
                      const float* pd, const void* prim_hits16, uint64_t n, float* depth) {
    const Hit16* H = (const Hit16*)prim_hits16;
    const V3 cam = ld3(cam_pos), fwd = normalize(ld3(cam_dir));
    for (uint64_t i = 0; i < n; ++i)
        depth[i] = H[i].tri_id == 0xFFFFFFFFu ? 1.0f
                                              : gl_depth24(ld3(po + 4 * i) + H[i].t * ld3(pd + 4 * i), cam, fwd, t_near, t_far);
}

void ora_combine(int res_x, int res_y, const float* depth, const float* vol_comp_rgba, float ext_k, int frame_id,
                 int exposure, float* accum_rgba, float* rgb) {
    const int hw = res_x / 2, hh = res_y / 2;
    std::vector<float> upd;
    const bool add = frame_id < 30 && ext_k > 0.0f;                              /* :65 */
    if (add) upd.assign(accum_rgba, accum_rgba + 4 * (size_t)res_x * res_y);
    const float div = (float)(1 + (frame_id < 30 ? frame_id : 30));
    for (int y = 0; y < res_y; ++y)
        for (int x = 0; x < res_x; ++x) {
            const size_t i = (size_t)x + (size_t)y * res_x;
            float c[3] = {accum_rgba[4 * i], accum_rgba[4 * i + 1], accum_rgba[4 * i + 2]};
            if (add) {                                                           /* bilateralUpsampling, :24-49 */
                const float center_depth = depth[i];
                float accum_weight = 0.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int px = x + dx, py = y + dy;
                        const float dpt = (px < 0 || px >= res_x || py < 0 || py >= res_y) ? 0.0f
                                                                                         : depth[(size_t)px + (size_t)py * res_x];
                        /* n_pos = vec2(pos) / CAM_RES; in half-resolution texels: pos / 2 */
                        const C4 col = vol_bilinear(vol_comp_rgba, hw, hh, ((float)px / (float)res_x) * (float)hw,
                                                    ((float)py / (float)res_y) * (float)hh);
                        const float weight = (1.0f - 0.3f * (float)abs(dx)) * (1.0f - 0.3f * (float)abs(dy)) /
                                             (smax(dpt - center_depth, 0.0f) * 1000.0f + 1.0f);
                        accum_weight += weight;
                        ar += weight * col.r; ag += weight * col.g; ab += weight * col.b;
                    }
                c[0] += ar / accum_weight; c[1] += ag / accum_weight; c[2] += ab / accum_weight;
                upd[4 * i] = c[0]; upd[4 * i + 1] = c[1]; upd[4 * i + 2] = c[2]; upd[4 * i + 3] = 1.0f;   /* :69 */
            }
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = 1.0f - expf(-(float)exposure * (c[k] / div));     /* :73-75 */
        }
    if (add) memcpy(accum_rgba, upd.data(), upd.size() * sizeof(float));
}

}  /* extern "C" */
