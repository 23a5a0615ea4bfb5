// gigl_trace_body.h — the traversal + intersection body shared by every kernel that walks the tree:
// trace_kernel (gigl_rt.hip: all ray batches) and photon_small_kernel (gigl_photon.hip: a whole small VPL set in
// one workgroup).  Device code only; everything has internal linkage.  See gigl_rt.hip's header for the
// arithmetic contract and the execution model.
#pragma once
#include "gigl_internal.h"
#include "kd_layout.h"

namespace {
using namespace gd;

struct ShadowSeg { float px, py, pz, wx, wy, wz, dist_sq, dist; float ox, oy, oz; };

// Per-point half of the shadow-ray construction (DESIGN.md "Shadow batch"): p = o + t*d; N =
// normalize(cross(e0,e1)) turned to face the primary ray; origin of the shadow rays = p + RAY_OFFSET * N.
// valid = 0 where the primary ray missed.
__device__ __forceinline__ bool shadow_point(const TraceParams& P, unsigned pt, float& px, float& py, float& pz,
                                             float& ox, float& oy, float& oz) {
    const uint4 h = P.prim_hits[pt];
    if (h.x == kMiss) return false;
    const float4 o = P.prim_o[pt], d = P.prim_d[pt];
    const float t = __uint_as_float(h.y);
    px = o.x + t * d.x; py = o.y + t * d.y; pz = o.z + t * d.z;  // getPtAtDist, RTBase.cpp:124-126
    const float4 a = P.tris[kTriStride * (size_t)h.x], b = P.tris[kTriStride * (size_t)h.x + 1];
    const float e1z = P.tris[kTriStride * (size_t)h.x + 2].x;
    const float e0x = a.w, e0y = b.x, e0z = b.y, e1x = b.z, e1y = b.w;
    float nx = e0y * e1z - e1y * e0z, ny = e0z * e1x - e1z * e0x, nz = e0x * e1y - e1x * e0y;
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    nx = nx * inv; ny = ny * inv; nz = nz * inv;
    if ((d.x * nx + d.y * ny) + d.z * nz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    ox = px + kRayOffset * nx; oy = py + kRayOffset * ny; oz = pz + kRayOffset * nz;   // Ray{p, w, N}, RTBase.cpp:117-122
    return true;
}

// The (point, light) pair: w = normalize(L - p); range (0, |L - p|).  With P.pt_cache the per-point half
// comes from the two float4 a pre-pass stored per point (the same values: it ran shadow_point once per
// point instead of once per pair).
__device__ __forceinline__ bool shadow_segment(const TraceParams& P, unsigned pt, unsigned l, ShadowSeg& s) {
    if (P.pt_cache) {
        const float4 a = P.pt_cache[2 * (size_t)pt];
        if (!(a.w > 0.0f)) return false;
        const float4 b = P.pt_cache[2 * (size_t)pt + 1];
        s.px = a.x; s.py = a.y; s.pz = a.z; s.ox = b.x; s.oy = b.y; s.oz = b.z;
    } else if (!shadow_point(P, pt, s.px, s.py, s.pz, s.ox, s.oy, s.oz)) return false;
    const float vx = P.lights[3 * l] - s.px, vy = P.lights[3 * l + 1] - s.py, vz = P.lights[3 * l + 2] - s.pz;
    s.dist_sq = (vx * vx + vy * vy) + vz * vz;
    s.dist = sqrtf(s.dist_sq);
    const float iv = 1.0f / sqrtf(s.dist_sq);
    s.wx = vx * iv; s.wy = vy * iv; s.wz = vz * iv;
    return true;
}
__device__ __forceinline__ bool shadow_segment(const TraceParams& P, unsigned pair, ShadowSeg& s) {
    const unsigned pt = pair / P.n_lights;
    return shadow_segment(P, pt, pair - pt * P.n_lights, s);
}

// Rays and hit records are touched once: streamed with the nontemporal hint so that they do not
// displace the tree and the triangles from L2 / Infinity Cache.
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef unsigned vuint4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
#if GIGL_STREAM_HINT
    const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stream_store(uint4* p, uint4 v) {
#if GIGL_STREAM_HINT
    vuint4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<vuint4*>(p));
#else
    *p = v;
#endif
}

// idx: the ray (GEN 0), the pixel (GEN 1) or the point of a (point, light) pair (GEN 2, 3; l = its light)
template <int GEN>
__device__ __forceinline__ bool fetch_ray(const TraceParams& P, unsigned idx, unsigned l, RayReg& r) {
    if (GEN == 0) {
        const float4 o = stream_load(P.o_tmin + idx), d = stream_load(P.d_tmax + idx);
        r.ox = o.x; r.oy = o.y; r.oz = o.z;
        set_dir(r, d.x, d.y, d.z);
        r.tmin = o.w; r.tmax = d.w;
        return true;
    } else if (GEN == 1) {
        float dx, dy, dz;
        primary_ray(P.cam, P.res_x, P.first + idx, dx, dy, dz);
        r.ox = P.cam[0]; r.oy = P.cam[1]; r.oz = P.cam[2];
        set_dir(r, dx, dy, dz);
        r.tmin = 0.0f; r.tmax = FLT_MAX;
        return true;
    } else if (GEN == 2) {
        ShadowSeg s;
        if (!shadow_segment(P, idx, l, s)) return false;
        r.ox = s.ox; r.oy = s.oy; r.oz = s.oz;
        set_dir(r, s.wx, s.wy, s.wz);
        r.tmin = 0.0f; r.tmax = s.dist;
        return true;
    } else {
        // visibility of light l from a point in the volume: Ray{p, normalize(L - p)} (no normal,
        // so no origin offset), range (0, |L - p|)
        const float4 p = P.points[idx];
        if (!(p.w > 0.0f)) return false;
        const float vx = P.lights[3 * l] - p.x, vy = P.lights[3 * l + 1] - p.y, vz = P.lights[3 * l + 2] - p.z;
        const float dist_sq = (vx * vx + vy * vy) + vz * vz;
        const float iv = 1.0f / sqrtf(dist_sq);
        r.ox = p.x; r.oy = p.y; r.oz = p.z;
        set_dir(r, vx * iv, vy * iv, vz * iv);
        r.tmin = 0.0f; r.tmax = sqrtf(dist_sq);
        return true;
    }
}

// Flattened triangle record: (p0.xyz, e0.x) (e0.yz, e1.xy) (e1.z, id).  In the scene's table a record
// is one 64-byte line per triangle id (its third float4 continues with the material id); in a leaf's
// private block (kd_layout.h) it is 40 bytes, three per 128-byte line, 8-byte aligned.
struct TriRec { float4 a, b; float e1z; unsigned tag; };
typedef float tri_f4 __attribute__((ext_vector_type(4), aligned(8)));
typedef float tri_f2 __attribute__((ext_vector_type(2), aligned(8)));

// All three loads are issued before any use: one memory round trip per triangle (left to itself
// the compiler sinks the p0 load below the det != 0 test, which makes it two).
__device__ __forceinline__ TriRec load_tri(const char* __restrict__ p) {
    TriRec t;
    // plain (cached) loads: the nontemporal hint on these costs 60 %, the records live in L2 / Infinity Cache
    const tri_f4 a = *reinterpret_cast<const tri_f4*>(p), b = *reinterpret_cast<const tri_f4*>(p + 16);
    const tri_f2 c = *reinterpret_cast<const tri_f2*>(p + 32);
    t.a = make_float4(a.x, a.y, a.z, a.w); t.b = make_float4(b.x, b.y, b.z, b.w);
    t.e1z = c.x; t.tag = __float_as_uint(c.y);
    asm volatile("" : "+v"(t.a.x), "+v"(t.a.y), "+v"(t.a.z), "+v"(t.a.w));
    return t;
}
// Table and leaf blocks share one allocation, so a triangle record is named by one 32-bit offset in
// 8-byte units from the table's start, whichever of the two it lives in (32 GB of range).
__device__ __forceinline__ const char* tri_ptr(const float4* tris, unsigned off8) {
    return reinterpret_cast<const char*>(tris) + (size_t)off8 * 8u;
}

// Triangle::intersect, RTBase.cpp:150-177.  Edges are stored precomputed (pt1-pt0, pt2-pt0 are
// the same float subtractions the reference performs per test).
__device__ __forceinline__ bool tri_test(const TriRec& T, unsigned id, const RayReg& r, float& best_t,
                                         float& best_u, float& best_v, unsigned& best_id) {
    const float p0x = T.a.x, p0y = T.a.y, p0z = T.a.z, e0x = T.a.w, e0y = T.b.x, e0z = T.b.y, e1x = T.b.z,
                e1y = T.b.w, e1z = T.e1z;
    const float pvx = r.dy * e1z - e1y * r.dz, pvy = r.dz * e1x - e1z * r.dx, pvz = r.dx * e1y - e1x * r.dy;
    const float det = (e0x * pvx + e0y * pvy) + e0z * pvz;
#if GIGL_BRANCHLESS_TRI
    // The reference nests these tests (det != 0, then u/v, then the distance window); none of them
    // has a side effect, so evaluating all of them and AND-ing the outcomes accepts exactly the
    // same hits while the wave runs one straight-line block instead of three nested masks.
    const float inv_det = 1.0f / det;
    const float tx = r.ox - p0x, ty = r.oy - p0y, tz = r.oz - p0z;
    const float qx = ty * e0z - e0y * tz, qy = tz * e0x - e0z * tx, qz = tx * e0y - e0x * ty;
    const float u = ((tx * pvx + ty * pvy) + tz * pvz) * inv_det;
    const float v = ((r.dx * qx + r.dy * qy) + r.dz * qz) * inv_det;
    const float dist = ((e1x * qx + e1y * qy) + e1z * qz) * inv_det;
    const bool ok = (det != 0.0f) & (u >= -kTriEps) & (v >= -kTriEps) & ((u + v) <= 1.0f + 2.0f * kTriEps) &
                    (r.tmin < dist) & (dist < r.tmax) & (dist < best_t);
    if (ok) { best_t = dist; best_u = u; best_v = v; best_id = id; }
    return ok;
#else
    if (det != 0.0f) {
        const float inv_det = 1.0f / det;
        const float tx = r.ox - p0x, ty = r.oy - p0y, tz = r.oz - p0z;
        const float qx = ty * e0z - e0y * tz, qy = tz * e0x - e0z * tx, qz = tx * e0y - e0x * ty;
        const float u = ((tx * pvx + ty * pvy) + tz * pvz) * inv_det;
        const float v = ((r.dx * qx + r.dy * qy) + r.dz * qz) * inv_det;
        if (u >= -kTriEps && v >= -kTriEps && (u + v) <= 1.0f + 2.0f * kTriEps) {
            const float dist = ((e1x * qx + e1y * qy) + e1z * qz) * inv_det;
            if (r.tmin < dist && dist < r.tmax && dist < best_t) {
                best_t = dist; best_u = u; best_v = v; best_id = id;
                return true;
            }
        }
    }
    return false;
#endif
}

// Spill-area accesses are opaque to the optimizer on purpose: merged with the LDS path they
// turn every stack access into a flat_* instruction.
typedef unsigned spill_t __attribute__((ext_vector_type(4)));
// A spilled entry also keeps the t_max it was pushed with (the fourth dword is free), so popping it
// is one load: entries in LDS rebuild t_max from the entry below them instead.
// The s_nop covers the wait states gfx9 needs between a VMEM store of more than 64 bits and a VALU
// write to its data registers: the compiler's hazard recognizer does not look inside inline asm, and
// `e` is dead (free for the next VALU result) as soon as the asm ends.
__device__ __forceinline__ void spill_store(uint4* p, unsigned a, unsigned b, unsigned c, unsigned d) {
    spill_t e; e.x = a; e.y = b; e.z = c; e.w = d;
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(e) : "memory");
}
__device__ __forceinline__ void spill_load(const uint4* p, unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    spill_t e;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(p) : "memory");
    a = e.x; b = e.y; c = e.z; d = e.w;
}

// ---- the traversal kernel ---------------------------------------------------------------------
// One kernel body for both device layouts (kd_layout.h); the wave pool, the refill, the stack, the
// T/L phase scheduling, the triangle rounds, the outputs and the instrumented build are shared, and
// only the node step and the leaf decoding differ (`if constexpr (TRIPLES)`).  Per ray the sequence
// of reference operations (KdTree.hpp:266-347) is the same under either layout.
//
// TRIPLES = false: pair-addressed 8-byte nodes (layouts 0 and 1; same topology and numbering as the
// reference's m_nodes, KdTree.h:22-65):
//   interior: w0 = split_pos, w1 = slot of the children pair << 2 | axis (bottom child first)
//   leaf:     w1 bit 31 = 1 (so "still descending" is one signed compare), bits 29-30 = kind,
//             bits 0-28 = payload
//             kind 0: empty          kind 1: one prim, w0 = id
//             kind 2: two prims, w0 = id0, payload = id1 (saves the m_leaf_prim_ids round trip)
//             kind 3: n >= 3 prims, w0 = offset into leaf_ids, payload = n
//   An interior step requests both children (one aligned 16-byte pair) before it decides which one
//   to enter, so the decision's arithmetic overlaps the memory round trip.
//
// TRIPLES = true: triple records (layout 2).  One 16-byte record holds a node at an even depth and
// both of its children, so one load serves two levels of the descent: half the per-lane loads and
// half the dependent round trips.  Lane state: lvl 0 = `rec` is the record of the current node
// (interior record, or leaf record when (rec.w & 3) == 3); lvl 1 = the current node is a child
// inside a record: rec.x = its slot (split or leaf payload), rec.y = type | block unit << 3.
// Stack entries are (t_split, w0, w1): w1 bit 2 clear = a lvl-1 node (w0 = slot, w1 = rec.y), set =
// a record still to be loaded (unit = w1 >> 3).
//
// A lane without a ray always holds a leaf, so the T-phase test needs no have_ray term.
//
// XCHG (experiment, closest-hit over explicit rays, 256-lane workgroups): rays change lanes.  Every P.xchg_every
// rounds the four waves of a workgroup meet and, lane column by lane column (lane l of waves 0..3), hand their
// rays round so that the descending ones end up in the low waves and the ones at leaves in the high waves: a wave
// then runs its T-phase, or its L-phase, with most of its lanes.  A ray's traversal stack does not move: the ray
// carries a slot number (the wave and lane whose LDS column and spill area it was given) — the bank a stack access
// falls into depends on the lane alone, so the accesses stay conflict free.  What moves is the register state,
// 24 dwords per ray, through one 4 KB transient buffer in six passes.
// STATS builds: the 12 counters of gigl_debug_last_stats sit at P.stats[0..11]; three 1024-bucket histograms over the
// 100 MHz real-time counter (rays started, rounds, lanes holding a ray at a round; of every 16th wave) follow at P.stats[16...]
constexpr unsigned kStatsTimeline = 16u;
constexpr unsigned kStatsWords = kStatsTimeline + 3u * 1024u;

// SOLO (photon_small_kernel): the workgroup is one wave whose whole batch is rays [0, P.n), P.n <= 64: no work counter
// (two atomic round trips saved per call), and a caller that calls again in the same kernel says that the epsilon
// table in LDS is already filled.
template <bool TRIPLES, bool ANY, int GEN, int STACK_N, int BLOCK, bool STATS = false, bool XCHG = false, bool SOLO = false>
__device__ __forceinline__ void trace_body(const TraceParams& P, bool eps_ready = false) {
    static_assert(!SOLO || BLOCK == 64, "a solo batch belongs to a one-wave workgroup");
    static_assert(!XCHG || (TRIPLES && !ANY && GEN == 0 && BLOCK == 256), "the exchange is built for one configuration");
    // stack entry = (t_split, the far child's two words): the far child is in registers when the
    // entry is pushed, so a pop costs no global load (pairs) or at most the record's (triples)
    __shared__ uint2 s_stack[STACK_N * BLOCK];     // (t_split, w1)
    __shared__ unsigned s_stack0[STACK_N * BLOCK]; // w0
    __shared__ float s_eps[kEpsTable];
    __shared__ uint4 s_xbuf[XCHG ? BLOCK : 1];      // XCHG: four dwords of every lane's state at a time
    __shared__ unsigned s_xcls[XCHG ? 64 : 1];      // XCHG: per lane column, one class byte per wave
    __shared__ unsigned s_xalive[XCHG ? 4 : 1];     // XCHG: per wave, whether it still has rays or may get some
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    if (!(SOLO && eps_ready)) {
        for (int i = (int)tid; i < kEpsTable; i += BLOCK) s_eps[i] = eps_from_exp(i - kEpsBias + 1);
        __syncthreads();
    }
#if GIGL_EPS_ALU
    // For frexp exponents in [-31, 29] (|t| in [2^-32, 2^29), and 0/inf/NaN, whose exponent reads 0) the
    // reference's value is 2^(e - 23) whichever branch of RTBase.h:24-28 it takes: assembled in two
    // instructions, off the LDS round trip; the wrap-around cases beyond that range read the table.
    auto eps_of = [&](float t) {
        const int fe = __builtin_amdgcn_frexp_expf(t);
        float v = __uint_as_float((unsigned)(fe + 105) << 23);
        const bool odd = (unsigned)(fe + 31) > 60u;
        if (__ballot(odd) != 0ull) {
            asm volatile("" ::: "memory");
            if (odd) v = s_eps[(unsigned)(fe + kEpsBias)];
        }
        return v;
    };
#else
    auto eps_of = [&](float t) { return s_eps[(unsigned)(__builtin_amdgcn_frexp_expf(t) + kEpsBias)]; };
#endif
    // XCHG: `slot` travels with the ray (the lane a ray sits in changes, its stack's place does not)
    unsigned slot = tid;
    uint2* my_stack = s_stack + tid;
    unsigned* my_stack0 = s_stack0 + tid;
    uint4* my_spill = P.spill + ((size_t)blockIdx.x * BLOCK + tid);
    const uint4* __restrict__ recs = P.recs;
    const float4* __restrict__ tris = P.tris;

    RayReg r;
    float root_tmax = 0.0f;
    float best_t = FLT_MAX, best_u = 0.0f, best_v = 0.0f;
    unsigned best_id = kMiss;
    unsigned ray_idx = 0;
    int sp = -1;
    bool have_ray = false, pool_empty = false;
    // wave-local slice of the ray index space, refilled CHUNK rays at a time with one atomic
    unsigned pool_next = 0, pool_end = 0, last_base = 0;
    // any-hit "group" mode: a wave owns 64 consecutive rays and writes their bits with one store
    unsigned group_base = kMiss;
    bool occ = false;
    const bool group_mode = ANY && !P.any_atomic;
    // node state (see above); a lane without a ray holds an empty leaf
    uint2 nd = make_uint2(0u, kLeafBit);
    uint4 rec = make_uint4(0u, 0u, 0u, 3u);
    unsigned lvl = 0, leaf_i = 0;

    auto descending = [&]() -> bool {
        if constexpr (TRIPLES) return ((lvl == 0u ? rec.w : rec.y) & 3u) != 3u;
        else return (int)nd.y >= 0;
    };
    auto push = [&](float t_split, unsigned w0, unsigned w1) {   // call before r.tmax shrinks
        ++sp;
        if (STACK_N >= 32 || sp < STACK_N) {
            my_stack[sp * BLOCK] = make_uint2(__float_as_uint(t_split), w1);
            my_stack0[sp * BLOCK] = w0;
        } else spill_store(my_spill + (size_t)(sp - STACK_N) * P.spill_stride, __float_as_uint(t_split), w0, w1,
                           __float_as_uint(r.tmax));
    };
    // pop, KdTree.hpp:334-339.  The reference stores (t_min, t_max, id); t_min is t_split - eps(t_split)
    // (:305) and the stored t_max is always the t_max set when the entry below was pushed,
    // t_split' + eps(t_split') (:310), or the root t_max for the bottom entry, so both are rebuilt
    // from the stored t_split values.
    auto pop = [&](unsigned& w0, unsigned& w1) {
        float ts;
        if (STACK_N >= 32 || sp < STACK_N) {
            const uint2 e = my_stack[sp * BLOCK];
            ts = __uint_as_float(e.x);
            w0 = my_stack0[sp * BLOCK];
            w1 = e.y;
            if (sp > 0) {
                const float tb = __uint_as_float(my_stack[(sp - 1) * BLOCK].x);
                r.tmax = tb + eps_of(tb);
            } else {
                r.tmax = root_tmax;
            }
        } else {
            unsigned a, d;
            spill_load(my_spill + (size_t)(sp - STACK_N) * P.spill_stride, a, w0, w1, d);
            ts = __uint_as_float(a);
            r.tmax = __uint_as_float(d);
        }
        r.tmin = ts - eps_of(ts);
        --sp;
    };
    auto finish = [&](bool hit) {
        if (ANY) {
            if (group_mode) occ = hit;
            else if (hit) atomicOr(((unsigned*)P.bits) + (ray_idx >> 5), 1u << (ray_idx & 31u));
        } else {
            stream_store(P.hits + ray_idx, hit ? make_uint4(best_id, __float_as_uint(best_t), __float_as_uint(best_u),
                                                            __float_as_uint(best_v))
                                               : make_uint4(kMiss, __float_as_uint(FLT_MAX), 0u, 0u));
            if (P.ranges) {   // gigl_trace_one: the ray's range as KdTree::intersect leaves it, and the hit triangle
                P.ranges[ray_idx] = make_float2(r.tmin, r.tmax);
                if (P.hit_tri && hit) {
                    const float4* t = tris + kTriStride * (size_t)best_id;
                    P.hit_tri[3 * ray_idx] = t[0]; P.hit_tri[3 * ray_idx + 1] = t[1]; P.hit_tri[3 * ray_idx + 2] = t[2];
                }
            }
        }
        have_ray = false;
        if constexpr (TRIPLES) { rec.w = 3u; lvl = 0u; }   // pairs: nd already is the leaf the ray ended in
    };

    // one interior step, KdTree.hpp:288-311: which child first, which children at all
    struct Step { float t_split; bool bottom_first, both, take_top; };
    auto decide = [&](unsigned axis, float split) -> Step {
        const float oa = sel3(axis, r.ox, r.oy, r.oz);
        const float ia = sel3(axis, r.ix, r.iy, r.iz);
        // KdTree.hpp:289-291: below the plane, or on it and not heading up.  The direction only
        // matters for an origin exactly on the plane, so it is selected on that (wave-uniform,
        // rarely taken) branch only.
        bool bottom_first = oa < split;
        if (__ballot(oa == split) != 0ull) {
            asm volatile("" ::: "memory");  // keeps this a real branch (else it is if-converted)
            bottom_first = bottom_first | ((oa == split) & (sel3(axis, r.dx, r.dy, r.dz) <= 0.0f));
        }
        Step s;
        s.t_split = (split - oa) * ia;
        const bool only_first = (s.t_split > r.tmax) | (s.t_split <= 0.0f);
        const bool only_second = s.t_split < r.tmin;
        s.bottom_first = bottom_first;
        s.both = !(only_first | only_second);
        s.take_top = bottom_first == (!only_first & only_second);   // the child entered now is the top one
        return s;
    };

    // STATS build only (tools/sweep.py stats=1): wave-level round counts and lane-level work counts
    const unsigned long long st_t0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned st_rounds = 0, st_t_iters = 0, st_t_lanes = 0, st_l_rounds = 0, st_l_lanes = 0, st_refills = 0,
             st_refill_lanes = 0, st_rays_lanes = 0, st_moved = 0;
    // timeline (every 16th wave, so that the histogram's atomics do not slow the launch down): per-bucket sums kept in
    // registers and flushed when the real-time counter enters the next 82 us bucket
    const bool st_sampled = STATS && P.stats && (((blockIdx.x * (BLOCK / 64u) + (tid >> 6)) & 15u) == 0u);
    unsigned st_bucket = STATS ? ((unsigned)(st_t0 >> 13) & 1023u) : 0u, st_b_started = 0, st_b_rounds = 0, st_b_lanes = 0;
    auto st_flush = [&]() {
        if (lane == 0 && st_b_rounds) {
            atomicAdd(P.stats + kStatsTimeline + st_bucket, (unsigned long long)st_b_started);
            atomicAdd(P.stats + kStatsTimeline + 1024u + st_bucket, (unsigned long long)st_b_rounds);
            atomicAdd(P.stats + kStatsTimeline + 2048u + st_bucket, (unsigned long long)st_b_lanes);
        }
        st_b_started = 0; st_b_rounds = 0; st_b_lanes = 0;
    };
    unsigned xround = 0;   // XCHG: rounds of this wave; the four waves of a workgroup count alike
    const int t_exit = P.t_exit;

    // Per-lane state machine.  A lane with a ray is either Traversing (at an interior node) or at a
    // Leaf (leaf_i = next primitive to test).  Every round of the outer loop runs a T-phase (interior
    // steps, repeated while at least t_exit lanes are still descending, so a few stragglers do not
    // stall the lanes already waiting at leaves) and an L-phase (one pair of triangle tests per lane
    // at a leaf).
    for (;;) {
        const unsigned long long idle = __ballot(!have_ray);
        const int n_idle = __popcll(idle);
        if (STATS) {
            ++st_rounds; st_rays_lanes += 64 - n_idle;
            if (st_sampled) {   // timeline: rounds and lanes holding a ray per 82 us of the real-time counter
                const unsigned b = (unsigned)(__builtin_amdgcn_s_memrealtime() >> 13) & 1023u;
                if (b != st_bucket) { st_flush(); st_bucket = b; }
                ++st_b_rounds; st_b_lanes += 64 - n_idle;
            }
        }
        if (n_idle >= P.refill_thresh && !pool_empty) {
            if (STATS) { ++st_refills; st_refill_lanes += n_idle; }
            if (group_mode) {
                const unsigned long long m = __ballot(occ);
                if (group_base != kMiss && lane == 0) P.bits[group_base >> 6] = m;
                occ = false;
                group_base = kMiss;
            }
            if (pool_next == pool_end) {
                // Guided self-scheduling: full chunks while plenty of rays remain, smaller ones
                // (multiples of 64) towards the end, so that the last chunk a wave takes does not
                // keep it busy long after the others have drained.  The request is sized from the
                // counter value this wave's previous request returned: stale, but it only sizes
                // the request (reading the counter with a load instead costs 45 %: the line is
                // owned by the atomics).
                unsigned base = 0, take_n = P.chunk;
                if (P.guided) {
                    const unsigned rem = last_base < P.n ? P.n - last_base : 0u;
                    const unsigned g = (rem / P.guided) & ~63u;
                    take_n = g < 64u ? 64u : (g < P.chunk ? g : P.chunk);
                }
                if constexpr (SOLO) {   // the first request is the whole batch, the second finds nothing
                    base = last_base;
                    take_n = P.n;
                    last_base = P.n;
                } else {
                    if (lane == 0) base = atomicAdd(P.counter, take_n);
                    base = __builtin_amdgcn_readfirstlane(base);
                    last_base = base;
                }
                pool_next = base < P.n ? base : P.n;
                pool_end = (P.n - pool_next) < take_n ? P.n : pool_next + take_n;
                if (pool_next == pool_end) pool_empty = true;
            }
            if (!pool_empty) {
                const unsigned avail = pool_end - pool_next;
                const unsigned take = (unsigned)n_idle < avail ? (unsigned)n_idle : avail;
                if (group_mode) group_base = pool_next;
                if (!have_ray) {
                    const unsigned rank = __popcll(idle & ((1ull << lane) - 1ull));
                    if (rank < take) {
                        unsigned idx = pool_next + rank;
                        if (GEN == 0 && P.perm) idx = P.perm[idx];
                        unsigned src = idx, light = 0u;   // what fetch_ray reads: the ray / pixel, or (point, light)
                        if (GEN == 2 || GEN == 3) {
                            if (P.n_pts) {
                                // light-major order: the lanes of a wave hold neighbouring points and one light, so
                                // their rays run side by side through the tree; results go to the pair's own bit
                                light = idx / P.n_pts;
                                src = idx - light * P.n_pts;
                                idx = src * P.n_lights + light;
                            } else {
                                src = idx / P.n_lights;
                                light = idx - src * P.n_lights;
                            }
                        }
                        ray_idx = idx;
                        best_t = FLT_MAX; best_u = 0.0f; best_v = 0.0f; best_id = kMiss;
                        bool go = fetch_ray<GEN>(P, src, light, r);
                        if (go) {
                            // KdTree.hpp:269-274
                            float entr, exit;
                            go = box_intersect(P.box, r, entr, exit);
                            if (go) {
                                const float nmin = gmax(r.tmin, entr - eps_of(entr));
                                const float nmax = gmin(r.tmax, exit + eps_of(exit));
                                r.tmin = nmin; r.tmax = nmax;
                                root_tmax = nmax;
                                go = !(best_t < r.tmin);  // KdTree.hpp:280
                            }
                        }
                        if (go) {
                            have_ray = true; sp = -1; leaf_i = 0;
                            if constexpr (TRIPLES) { rec = P.root_rec; lvl = 0; }
                            else nd = P.root;
                        } else finish(false);
                    }
                }
                pool_next = __builtin_amdgcn_readfirstlane(pool_next + take);
                if (STATS && st_sampled) st_b_started += take;
            }
        }
        if constexpr (XCHG) {
            if (xround++ % P.xchg_every == 0u) {
                // classes of this lane column: 0 descending, 1 no ray, 2 at a leaf
                const unsigned w = tid >> 6;
                const unsigned cls = have_ray ? (descending() ? 0u : 2u) : 1u;
                reinterpret_cast<unsigned char*>(s_xcls)[lane * 4u + w] = (unsigned char)cls;
                if (lane == 0u) s_xalive[w] = (__ballot(have_ray) != 0ull || !pool_empty) ? 1u : 0u;
                __syncthreads();
                const unsigned col = s_xcls[lane];
                const unsigned alive = s_xalive[0] | s_xalive[1] | s_xalive[2] | s_xalive[3];
                if (!alive) break;   // the same for all four waves: none has rays, none can get any
                const unsigned c0 = col & 255u, c1 = (col >> 8) & 255u, c2 = (col >> 16) & 255u, c3 = col >> 24;
                // stable sort of the column by class: where each wave's ray goes
                const unsigned d0 = (c1 < c0) + (c2 < c0) + (c3 < c0);
                const unsigned d1 = (c0 <= c1) + (c2 < c1) + (c3 < c1);
                const unsigned d2 = (c0 <= c2) + (c1 <= c2) + (c3 < c2);
                const unsigned src = d0 == w ? 0u : (d1 == w ? 1u : (d2 == w ? 2u : 3u));
                const bool moved = src != w;
                if (STATS && moved) ++st_moved;
                const unsigned from = src * 64u + lane;
                uint4 q;
#define GIGL_XPASS(a, b, c, d)                                   \
                q = make_uint4(a, b, c, d);                      \
                s_xbuf[tid] = q;                                 \
                __syncthreads();                                 \
                if (moved) q = s_xbuf[from];                     \
                __syncthreads();
                GIGL_XPASS(__float_as_uint(r.ox), __float_as_uint(r.oy), __float_as_uint(r.oz), __float_as_uint(r.dx))
                r.ox = __uint_as_float(q.x); r.oy = __uint_as_float(q.y); r.oz = __uint_as_float(q.z); r.dx = __uint_as_float(q.w);
                GIGL_XPASS(__float_as_uint(r.dy), __float_as_uint(r.dz), __float_as_uint(r.ix), __float_as_uint(r.iy))
                r.dy = __uint_as_float(q.x); r.dz = __uint_as_float(q.y); r.ix = __uint_as_float(q.z); r.iy = __uint_as_float(q.w);
                GIGL_XPASS(__float_as_uint(r.iz), __float_as_uint(r.tmin), __float_as_uint(r.tmax), __float_as_uint(root_tmax))
                r.iz = __uint_as_float(q.x); r.tmin = __uint_as_float(q.y); r.tmax = __uint_as_float(q.z); root_tmax = __uint_as_float(q.w);
                GIGL_XPASS(__float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v), best_id)
                best_t = __uint_as_float(q.x); best_u = __uint_as_float(q.y); best_v = __uint_as_float(q.z); best_id = q.w;
                GIGL_XPASS(rec.x, rec.y, rec.z, rec.w)
                rec = q;
                GIGL_XPASS(ray_idx, leaf_i, (unsigned)(sp + 1) | (lvl << 8) | ((have_ray ? 1u : 0u) << 9), slot)
#undef GIGL_XPASS
                ray_idx = q.x; leaf_i = q.y;
                sp = (int)(q.z & 255u) - 1; lvl = (q.z >> 8) & 1u; have_ray = ((q.z >> 9) & 1u) != 0u;
                slot = q.w;
                my_stack = s_stack + slot;
                my_stack0 = s_stack0 + slot;
                my_spill = P.spill + ((size_t)blockIdx.x * BLOCK + slot);
            }
        } else {
            if (__ballot(have_ray) == 0ull) {
                if (pool_empty) break;
                continue;
            }
        }
        // XCHG: a phase that only a few lanes of this wave would take part in is left out; their rays wait for the
        // next exchange, which hands them to a wave where they are many (the phase with more lanes always runs)
        bool run_t = true, run_l = true;
        if constexpr (XCHG) {
            const int n_t = __popcll(__ballot(descending())), n_l = __popcll(__ballot(have_ray && !descending()));
            run_t = n_t >= P.xchg_min || n_t >= n_l;
            run_l = n_l >= P.xchg_min || n_l > n_t;
        }
        // ---- T-phase: KdTree.hpp:284-311 ----
        for (;;) {
            if (XCHG && !run_t) break;
            const bool is_t = descending();
            const unsigned long long m_t = __ballot(is_t);
            if (m_t == 0ull) break;
            if (STATS) { ++st_t_iters; st_t_lanes += __popcll(m_t); }
            if (is_t) {
                if constexpr (TRIPLES) {   // up to two levels and exactly one record load per lane and iteration
#if GIGL_EARLY_TOUCH
                    // Experiment: the four records one of which this iteration ends up loading share a 128-byte line
                    // (kd_layout.cpp); one dword of it is requested before the two steps are decided, so that the
                    // decisions' arithmetic overlaps the miss.  `touch` stays allocated until its data has arrived
                    // (the s_waitcnt below names it): a load's destination register must not be reused in flight.
                    unsigned touch = 0u;
                    const bool touched = lvl == 0u;
                    if (touched) asm volatile("global_load_dword %0, %1, off" : "=v"(touch) : "v"(recs + ((rec.w >> 6) << 1)));
#endif
                    if (lvl == 0u) {   // a node with both children in its record: no load
                        const Step s = decide(rec.w & 3u, __uint_as_float(rec.x));
                        const unsigned type_a = (rec.w >> 2) & 3u, type_b = (rec.w >> 4) & 3u;
                        const unsigned blk = (rec.w >> 6) << 1;                    // unit of A's children
                        const unsigned blk_b = blk + (type_a != 3u ? 2u : 0u);     // unit of B's children
                        const unsigned desc_a = type_a | (blk << 3), desc_b = type_b | (blk_b << 3);
                        if (s.both) {      // the far child waits on the stack
                            push(s.t_split, s.bottom_first ? rec.z : rec.y, s.bottom_first ? desc_b : desc_a);
                            r.tmax = s.t_split + eps_of(s.t_split);
                        }
                        rec.x = s.take_top ? rec.z : rec.y;
                        rec.y = s.take_top ? desc_b : desc_a;
                        lvl = 1u;
                    }
                    if ((rec.y & 3u) != 3u) {   // a child inside a record: its children are records of their own
                        const Step s = decide(rec.y & 3u, __uint_as_float(rec.x));
                        const unsigned blk = rec.y >> 3;
                        if (s.both) {
                            push(s.t_split, 0u, 4u | ((blk + (s.bottom_first ? 1u : 0u)) << 3));
                            r.tmax = s.t_split + eps_of(s.t_split);
                        }
                        rec = recs[blk + (s.take_top ? 1u : 0u)];
                        lvl = 0u;
                    }
#if GIGL_EARLY_TOUCH
                    if (touched) asm volatile("s_waitcnt vmcnt(0)" ::"v"(touch));
#endif
                } else {
                    // slot of the bottom child; the top child is the next slot
                    const uint4 kids = *reinterpret_cast<const uint4*>(P.nodes + (nd.y >> 2));
                    const Step s = decide(nd.y & 3u, __uint_as_float(nd.x));
                    if (s.both) {  // both children: the far one waits on the stack
                        push(s.t_split, s.bottom_first ? kids.z : kids.x, s.bottom_first ? kids.w : kids.y);
                        r.tmax = s.t_split + eps_of(s.t_split);
                    }
                    nd = s.take_top ? make_uint2(kids.z, kids.w) : make_uint2(kids.x, kids.y);
                }
            }
            if ((int)__popcll(m_t) < t_exit) break;
        }
        // ---- L-phase: KdTree.hpp:312-343.  Prims are tested in m_leaf_prim_ids order, two per
        // round; both triangles' loads are issued before the first test. ----
        if (have_ray && !descending() && (!XCHG || run_l)) {
            unsigned n_prims;
            if constexpr (TRIPLES) {
                if (lvl == 1u) {   // a leaf inside a record: its payload names the prims
                    const unsigned payload = rec.x, kind = payload & 3u;
                    if (kind == 2u) rec = recs[payload >> 2];
                    else if (kind == 3u)   // a triangle block named by the payload (kd_layout.h): no leaf record to fetch
                        rec = make_uint4(payload >> 7, 0u, 0u, 3u | (((payload >> 2) & 31u) << 2) | gigl::kLeafBlockFlag);
                    else rec = make_uint4(payload >> 2, 0u, 0u, kind == 1u ? (3u | (1u << 2)) : 3u);
                    lvl = 0u;
                }
                n_prims = (rec.w >> 2) & 0x1FFFFFFFu;   // bit 31: the leaf reads its private triangle block
            } else {
                const unsigned kind = (nd.y >> 29) & 3u;
                n_prims = kind == 3u ? (nd.y & kLeafPayload) : kind;
            }
            bool any_done = false;
            if (STATS) {   // L rounds that test triangles, and the lanes that do
                const unsigned long long m_l = __ballot(leaf_i < n_prims);
                if (m_l) { ++st_l_rounds; st_l_lanes += __popcll(m_l); }
            }
            if (leaf_i < n_prims) {
#if GIGL_POP_TOUCH
                // Experiment: a lane that starts testing a leaf's triangles requests the record its next pop will load
                // (the top stack entry names it), so that the pop's miss overlaps the triangle loads and tests.  One dword
                // into `touch`, which stays allocated until the triangle data, requested after it, has arrived.
                unsigned touch = 0u;
                if constexpr (TRIPLES) {
                    if (leaf_i == 0u && sp >= 0 && sp < STACK_N) {
                        const unsigned w1n = my_stack[sp * BLOCK].y;
                        if (w1n & 4u) asm volatile("global_load_dword %0, %1, off" : "=v"(touch) : "v"(recs + (w1n >> 3)));
                    }
                }
#endif
                unsigned id0 = 0u, id1 = 0u;   // table mode: triangle ids; block mode: record offsets in 8-byte units
                const bool two = leaf_i + 1 < n_prims;
                bool in_block = false;
                if constexpr (TRIPLES) {
                    in_block = (int)rec.w < 0;
                    if (in_block) {
                        // record i of the block: line i / 3, 40-byte slot i % 3; read in list order, so a
                        // leaf costs ceil(n / 3) consecutive lines and no id list
                        const unsigned i1 = two ? leaf_i + 1u : leaf_i;
                        const unsigned base8 = P.blocks_off8 + rec.x * 16u;
                        id0 = base8 + (leaf_i / 3u) * 16u + (leaf_i % 3u) * 5u;
                        id1 = base8 + (i1 / 3u) * 16u + (i1 % 3u) * 5u;
                    } else if (n_prims <= 3u) {
                        id0 = leaf_i == 0u ? rec.x : rec.z;
                        id1 = rec.y;
                    } else {
                        const uint2 two_ids = reinterpret_cast<const uint2*>(recs + rec.x)[leaf_i >> 1];
                        id0 = two_ids.x; id1 = two_ids.y;
                    }
                } else {
                    if (n_prims >= 3u) {
                        // lists are 8-byte aligned and padded to an even length (kd_layout.h)
                        const uint2 two_ids = *reinterpret_cast<const uint2*>(P.leaf_ids + nd.x + leaf_i);
                        id0 = two_ids.x; id1 = two_ids.y;
                    } else {
                        id0 = nd.x;
                        id1 = nd.y & kLeafPayload;
                    }
                }
                if (!two) id1 = id0;
                const TriRec t0 = load_tri(tri_ptr(tris, in_block ? id0 : id0 * (kTriStride * 2u)));
                const TriRec t1 = load_tri(tri_ptr(tris, in_block ? id1 : id1 * (kTriStride * 2u)));
                bool hit = tri_test(t0, t0.tag, r, best_t, best_u, best_v, best_id);
                if (!(ANY && hit) && two) hit = tri_test(t1, t1.tag, r, best_t, best_u, best_v, best_id) || hit;
                any_done = ANY && hit;
                leaf_i += 2;
#if GIGL_POP_TOUCH
                asm volatile("" ::"v"(touch));
#endif
            }
            if (any_done) {
                finish(true);
            } else if (leaf_i >= n_prims) {
                if (best_id != kMiss) {
                    finish(true);  // closest hit inside this leaf's cell: KdTree.hpp:320,332
                } else if (sp >= 0) {
                    unsigned w0, w1;
                    pop(w0, w1);
                    if (best_t < r.tmin) {   // KdTree.hpp:280
                        finish(false);
                    } else {
                        leaf_i = 0;
                        if constexpr (TRIPLES) {
                            if (w1 & 4u) { rec = recs[w1 >> 3]; lvl = 0u; }
                            else { rec.x = w0; rec.y = w1; lvl = 1u; }
                        } else {
                            nd = make_uint2(w0, w1);
                        }
                    }
                } else {
                    finish(false);
                }
            }
        }
    }
    if (group_mode) {
        const unsigned long long m = __ballot(occ);
        if (group_base != kMiss && lane == 0) P.bits[group_base >> 6] = m;
    }
    // launches on a workspace that is kept (gigl_trace_one, the no-allocation entry point) leave the work counter
    // at zero for the next one, so that no memset has to precede them (one that a HIP graph would have to replay):
    // every wave checks out once it has taken its last request, and the last one out clears both words
    if (P.reset_counter && lane == 0) {
        const unsigned n_waves = gridDim.x * (BLOCK / 64);
        if (atomicAdd(P.counter + 1, 1u) == n_waves - 1u) {
            P.counter[1] = 0u;
            P.counter[0] = 0u;
        }
    }
    if (STATS && st_sampled) st_flush();
    if (STATS && lane == 0 && P.stats) {
        atomicAdd(P.stats + 0, (unsigned long long)st_rounds);
        atomicAdd(P.stats + 1, (unsigned long long)st_t_iters);
        atomicAdd(P.stats + 2, (unsigned long long)st_t_lanes);
        atomicAdd(P.stats + 3, (unsigned long long)st_l_rounds);
        atomicAdd(P.stats + 4, (unsigned long long)st_l_lanes);
        atomicAdd(P.stats + 5, (unsigned long long)(XCHG ? st_moved : st_refills));   // XCHG builds: lanes whose ray changed at an exchange
        atomicAdd(P.stats + 6, (unsigned long long)st_refill_lanes);
        atomicAdd(P.stats + 7, (unsigned long long)st_rays_lanes);
        // when the waves start and stop, on the 100 MHz real-time counter: the drain tail of the
        // launch is (latest end) - (mean end)
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        atomicMax(P.stats + 8, ~st_t0);   // earliest start, complemented (the slot starts at 0)
        atomicMax(P.stats + 9, t_end);
        atomicAdd(P.stats + 10, t_end);
        atomicAdd(P.stats + 11, 1ull);
    }
}

}  // namespace
