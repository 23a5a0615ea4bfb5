// gigl_walk_lean.h — a second, lean form of the traversal for photon_small_kernel (gigl_photon.hip): one ray per
// lane, walked to the end in one divergent loop.  Device code only; uses the helpers of gigl_trace_body.h
// (fetch_ray, load_tri, tri_test) and gigl_internal.h, and nothing of trace_body's own state machine.
#pragma once
#include "gigl_trace_body.h"

namespace {
using namespace gd;

// ---- one ray per lane, walked to the end in one divergent loop ------------------------------------------
// photon_small_kernel's walk (triple records, closest hit): a wave there holds 4 to 32 rays and nothing to refill
// them from, and what it costs is the dependent chain of its longest ray at the issue rate of a lone wave (one
// instruction every 8-13 cycles), so the instructions per node are what counts.  trace_body's rounds (ballots, refill
// test, T and L phases taking turns, two triangles per round) are built for 64 busy lanes and a work counter; here a
// lane steps through its own ray — interior steps and leaves as they come, a leaf's triangles in one go — with a
// 32-entry stack in LDS (no spill area) and the epsilon assembled in registers.  Per ray the sequence of reference
// operations (KdTree.hpp:266-347), the order of the triangle tests and every expression are trace_body's, so the
// records are the same bit for bit (tests/test_gpu_vpl_batch.py compares both walks with the oracle's).
// Outputs as trace_body's finish() (P.ranges / P.hit_tri optional).  Call with the whole wave; `active` lanes hold a ray.
template <int BLOCK>
__device__ __forceinline__ void walk_lean(const TraceParams& P, bool active, unsigned ray_idx) {
    constexpr int kDepth = 32;
    __shared__ uint2 s_lstack[kDepth * BLOCK];       // (t_split, w1)
    __shared__ unsigned s_lstack0[kDepth * BLOCK];   // w0
    if (!active) return;
    uint2* my_stack = s_lstack + threadIdx.x;
    unsigned* my_stack0 = s_lstack0 + threadIdx.x;
    const uint4* __restrict__ recs = P.recs;
    const float4* __restrict__ tris = P.tris;
    auto eps_of = [&](float t) {   // getRescaledEPS, as trace_body assembles it; the wrap-around cases by the full function
        const int fe = __builtin_amdgcn_frexp_expf(t);
        float v = __uint_as_float((unsigned)(fe + 105) << 23);
        if ((unsigned)(fe + 31) > 60u) v = eps_from_exp(fe + 1);
        return v;
    };
    RayReg r;
    fetch_ray<0>(P, ray_idx, 0u, r);
    float root_tmax = 0.0f, best_t = FLT_MAX, best_u = 0.0f, best_v = 0.0f;
    unsigned best_id = kMiss;
    bool go;
    {   // KdTree.hpp:269-274
        float entr, exit;
        go = box_intersect(P.box, r, entr, exit);
        if (go) {
            const float nmin = gmax(r.tmin, entr - eps_of(entr));
            const float nmax = gmin(r.tmax, exit + eps_of(exit));
            r.tmin = nmin; r.tmax = nmax;
            root_tmax = nmax;
            go = !(best_t < r.tmin);   // KdTree.hpp:280
        }
    }
    uint4 rec = P.root_rec;
    unsigned lvl = 0u;
    int sp = -1;
    bool hit = false;
    auto decide = [&](unsigned axis, float split, float& t_split, bool& bottom_first, bool& both, bool& take_top) {
        const float oa = sel3(axis, r.ox, r.oy, r.oz);
        const float ia = sel3(axis, r.ix, r.iy, r.iz);
        bottom_first = (oa < split) | ((oa == split) & (sel3(axis, r.dx, r.dy, r.dz) <= 0.0f));   // KdTree.hpp:289-291
        t_split = (split - oa) * ia;
        const bool only_first = (t_split > r.tmax) | (t_split <= 0.0f);
        const bool only_second = t_split < r.tmin;
        both = !(only_first | only_second);
        take_top = bottom_first == (!only_first & only_second);
    };
    auto push = [&](float t_split, unsigned w0, unsigned w1) {
        ++sp;
        my_stack[sp * BLOCK] = make_uint2(__float_as_uint(t_split), w1);
        my_stack0[sp * BLOCK] = w0;
    };
    while (go) {
        if (((lvl == 0u ? rec.w : rec.y) & 3u) != 3u) {
            // ---- interior: KdTree.hpp:284-311, up to two levels per record ----
            float ts; bool bf, both, top;
            if (lvl == 0u) {
                decide(rec.w & 3u, __uint_as_float(rec.x), ts, bf, both, top);
                const unsigned type_a = (rec.w >> 2) & 3u, type_b = (rec.w >> 4) & 3u;
                const unsigned blk = (rec.w >> 6) << 1;
                const unsigned blk_b = blk + (type_a != 3u ? 2u : 0u);
                const unsigned desc_a = type_a | (blk << 3), desc_b = type_b | (blk_b << 3);
                if (both) {
                    push(ts, bf ? rec.z : rec.y, bf ? desc_b : desc_a);
                    r.tmax = ts + eps_of(ts);
                }
                rec.x = top ? rec.z : rec.y;
                rec.y = top ? desc_b : desc_a;
                lvl = 1u;
            }
            if ((rec.y & 3u) != 3u) {
                decide(rec.y & 3u, __uint_as_float(rec.x), ts, bf, both, top);
                const unsigned blk = rec.y >> 3;
                if (both) {
                    push(ts, 0u, 4u | ((blk + (bf ? 1u : 0u)) << 3));
                    r.tmax = ts + eps_of(ts);
                }
                rec = recs[blk + (top ? 1u : 0u)];
                lvl = 0u;
            }
            continue;
        }
        // ---- leaf: KdTree.hpp:312-343, prims in m_leaf_prim_ids order ----
        if (lvl == 1u) {
            const unsigned payload = rec.x, kind = payload & 3u;
            if (kind == 2u) rec = recs[payload >> 2];
            else if (kind == 3u) rec = make_uint4(payload >> 7, 0u, 0u, 3u | (((payload >> 2) & 31u) << 2) | gigl::kLeafBlockFlag);
            else rec = make_uint4(payload >> 2, 0u, 0u, kind == 1u ? (3u | (1u << 2)) : 3u);
            lvl = 0u;
        }
        const unsigned n_prims = (rec.w >> 2) & 0x1FFFFFFFu;
        const bool in_block = (int)rec.w < 0;
        for (unsigned i = 0; i < n_prims; ++i) {
            unsigned off8;
            if (in_block) off8 = P.blocks_off8 + rec.x * 16u + (i / 3u) * 16u + (i % 3u) * 5u;
            else if (n_prims <= 3u) off8 = (i == 0u ? rec.x : (i == 1u ? rec.y : rec.z)) * (kTriStride * 2u);
            else off8 = reinterpret_cast<const unsigned*>(recs + rec.x)[i] * (kTriStride * 2u);
            const TriRec t = load_tri(tri_ptr(tris, off8));
            tri_test(t, t.tag, r, best_t, best_u, best_v, best_id);
        }
        if (best_id != kMiss) { hit = true; break; }   // closest hit inside this leaf's cell: KdTree.hpp:320,332
        if (sp < 0) break;
        {   // pop, KdTree.hpp:334-339 (t_min and t_max rebuilt from the stored t_split values, as trace_body does)
            const uint2 e = my_stack[sp * BLOCK];
            const unsigned w0 = my_stack0[sp * BLOCK], w1 = e.y;
            const float ts = __uint_as_float(e.x);
            if (sp > 0) {
                const float tb = __uint_as_float(my_stack[(sp - 1) * BLOCK].x);
                r.tmax = tb + eps_of(tb);
            } else {
                r.tmax = root_tmax;
            }
            r.tmin = ts - eps_of(ts);
            --sp;
            if (best_t < r.tmin) break;   // KdTree.hpp:280
            if (w1 & 4u) { rec = recs[w1 >> 3]; lvl = 0u; }
            else { rec.x = w0; rec.y = w1; lvl = 1u; }
        }
    }
    P.hits[ray_idx] = hit ? make_uint4(best_id, __float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v))
                          : make_uint4(kMiss, __float_as_uint(FLT_MAX), 0u, 0u);
    if (P.ranges) P.ranges[ray_idx] = make_float2(r.tmin, r.tmax);
    if (hit && P.hit_tri) {
        const float4* t = tris + kTriStride * (size_t)best_id;
        P.hit_tri[3 * ray_idx] = t[0]; P.hit_tri[3 * ray_idx + 1] = t[1]; P.hit_tri[3 * ray_idx + 2] = t[2];
    }
}

}  // namespace
