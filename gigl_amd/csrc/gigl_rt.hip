// gigl_rt.hip — HIP kernels (gfx950) and C-ABI implementation of the ray-query engine.
//
// Device arithmetic restates, operation for operation, the reference's scalar float code:
//   BBox::intersect        /root/reference/Source/Common/BBox.cpp:81-109
//   getRescaledEPS         Source/RT/RTBase.h:18-29
//   KdTree::intersect      Source/RT/KdTree.hpp:266-347
//   Triangle::intersect    Source/RT/RTBase.cpp:150-177
//   Scene::trace           Source/Common/Scene.cpp:234-243
//   getPrimaryRay          Source/Common/Camera.cpp:65-68
//   sampleDensity          Source/Fog/DensityField.cpp:177-211
//   calcTransm / rayMarch  Source/Shaders/Surface.frag:139-203
// Build flags that are part of the contract: -ffp-contract=off (no FMA fusion; the reference is
// MSVC /fp:precise), correctly rounded f32 divide and sqrt (hipcc default), f32 denormals kept.
// min/max are written as selects (GLM func_common.inl:377,395), never fminf/fmaxf.
//
// Execution model: persistent wavefronts.  Each 64-lane wave takes ray indices from a global
// counter (one atomic per `chunk` rays), one ray per lane, and refills idle lanes (found with a
// ballot) once enough of them have retired, so long-tailed rays do not hold 63 idle lanes hostage.
// Lanes run a two-state machine (Traversing / at a Leaf, see trace_kernel).  The per-lane traversal
// stack holds (t_split, node) pairs — t_min/t_max are rebuilt from them on pop — with the first
// STACK_N entries in LDS (lane-interleaved, conflict-free ds_read/write_b64) and the rare deeper
// entries (0.04 % of pushes on the 1M-triangle scene) in a global spill area.
// Device data layout: flatten_nodes() (pair-addressed nodes in 128-byte treelets) and the 64-byte
// triangle records built in gigl_scene_create().
#include <hip/hip_runtime.h>

// A/B switches kept from tuning (tools/sweep.py over builds, same device, interleaved): the
// branch-free child-order predicate is worth 1-3 %; the branch-free triangle test costs 10 VGPRs,
// drops the kernel from 6 to 5 waves/SIMD and loses 4-10 % on coherent batches.  GIGL_EARLY_PAIR requests
// both children of a node before the step is decided (+14 % on coherent batches, +1.5 % on the
// incoherent bench batch, which is bound by the chip's L2-miss rate instead).
#ifndef GIGL_BRANCHLESS_TRI
#define GIGL_BRANCHLESS_TRI 0
#endif
#ifndef GIGL_BRANCHLESS_STEP
#define GIGL_BRANCHLESS_STEP 1
#endif
#ifndef GIGL_STREAM_HINT
#define GIGL_STREAM_HINT 1
#endif
#ifndef GIGL_EARLY_PAIR
#define GIGL_EARLY_PAIR 1
#endif

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gigl_rt.h"
#include "kd_build.h"

namespace {

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
constexpr float kRayOffset = 1E-4f;              // Source/Common/Constants.h:24
constexpr float kTriEps = 1E-4f;                 // Constants.h:25
constexpr unsigned kMiss = 0xFFFFFFFFu;
constexpr unsigned kLeafBit = 0x80000000u, kLeafPayload = 0x1FFFFFFFu;   // device node word 1, see trace_kernel
// float4 per flattened triangle record: 4 = one 64-byte line per triangle (a 48-byte record
// straddles two lines half of the time, and nearly every triangle fetch of an incoherent batch
// misses L2)
constexpr unsigned kTriStride = 4;

__device__ __forceinline__ float gmin(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float gmax(float x, float y) { return x > y ? x : y; }
__device__ __forceinline__ float sel3(unsigned a, float x, float y, float z) { return a == 0u ? x : (a == 1u ? y : z); }

// RTBase.h:18-29: frexp(t,&e); ++e; e >= 0 ? EPS * (1 << e) : EPS / (1 << -e), with `1 << n` evaluated
// as x86 shl does for the reference (count masked to 5 bits, 1 << 31 = INT_MIN).  frexp's exponent
// comes from v_frexp_exp_i32_f32 (0 for 0/inf/nan, as glibc).  Every operand is a power of two, so
// product and quotient are exact and the result is assembled directly: 2^(k-23) or 2^(-23-k),
// negative when k == 31.  (tests/test_gpu_parity.py::test_rescaled_eps_device checks 8k values
// against the oracle's literal evaluation.)
__device__ __forceinline__ float eps_from_exp(int e /* frexp exponent + 1 */) {
    const unsigned k = (unsigned)(e >= 0 ? e : -e) & 31u;
    const unsigned ex = e >= 0 ? 104u + k : 104u - k;   // biased exponent of 2^(-23 +/- k)
    return __uint_as_float((ex << 23) | (k == 31u ? 0x80000000u : 0u));
}
__device__ __forceinline__ float rescaled_eps(float t) { return eps_from_exp(__builtin_amdgcn_frexp_expf(t) + 1); }
// trace_kernel keeps the function as a table in LDS, indexed by frexp exponent + kEpsBias: the
// exponent of a float (denormals included) lies in [-148, 128].
constexpr int kEpsBias = 148, kEpsTable = 277;

struct RayReg {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, tmin, tmax;
};

__device__ __forceinline__ void set_dir(RayReg& r, float dx, float dy, float dz) {
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;  // Ray ctor, RTBase.cpp:111-113
}

struct BoxF { float mnx, mny, mnz, mxx, mxy, mxz; };

// BBox.cpp:81-109
__device__ __forceinline__ bool box_intersect(const BoxF& b, const RayReg& r, float& entr, float& exit) {
    const bool sx = r.ix >= 0.0f, sy = r.iy >= 0.0f, sz = r.iz >= 0.0f;
    float t_entr = ((sx ? b.mnx : b.mxx) - r.ox) * r.ix;
    float t_exit = ((sx ? b.mxx : b.mnx) - r.ox) * r.ix;
    const float t_y_min = ((sy ? b.mny : b.mxy) - r.oy) * r.iy;
    const float t_y_max = ((sy ? b.mxy : b.mny) - r.oy) * r.iy;
    if ((t_entr > t_y_max) || (t_y_min > t_exit)) return false;
    t_entr = gmax(t_entr, t_y_min);
    t_exit = gmin(t_exit, t_y_max);
    const float t_z_min = ((sz ? b.mnz : b.mxz) - r.oz) * r.iz;
    const float t_z_max = ((sz ? b.mxz : b.mnz) - r.oz) * r.iz;
    if ((t_entr > t_z_max) || (t_z_min > t_exit)) return false;
    t_entr = gmax(t_entr, t_z_min);
    t_exit = gmin(t_exit, t_z_max);
    if (t_entr < r.tmax && t_exit > r.tmin) {
        entr = t_entr; exit = t_exit;
        return t_entr < FLT_MAX;  // IntDist::operator bool, BBox.cpp:13-15
    }
    return false;
}

struct TraceParams {
    const uint2* nodes;
    const unsigned* leaf_ids;
    const float4* tris;   // kTriStride float4 per triangle: (p0.xyz,e0.x) (e0.yz,e1.xy) (e1.z,mat,-,-) (-)
    BoxF box;
    unsigned n;
    unsigned* counter;
    uint2* spill;
    unsigned spill_stride;
    int refill_thresh;
    int any_atomic;
    unsigned chunk;   // rays a wave takes from the global counter per atomic (multiple of 64)
    unsigned guided;  // 0: fixed chunks; else requests shrink to remaining / guided near the end
    int t_exit;       // T-phase repeats while at least this many lanes are still descending
    // GEN 0: explicit rays
    const float4* o_tmin;
    const float4* d_tmax;
    // GEN 1: primary rays
    float cam[12];
    unsigned res_x, first;
    // GEN 2: shadow pairs
    const float4* prim_o;
    const float4* prim_d;
    const uint4* prim_hits;
    const float* lights;
    unsigned n_lights;
    // GEN 3: (point, light) pairs from explicit points (x, y, z, valid)
    const float4* points;
    // outputs
    uint4* hits;
    unsigned long long* bits;
    unsigned long long* stats;   // STATS build only: 12 counters
    float2* ranges;              // optional: the ray's (t_min, t_max) as KdTree::intersect leaves them
    const unsigned* perm;        // optional (GEN 0): work item i traces ray perm[i] (ray regrouping)
};

// getPrimaryRay(x + 0.5, y + 0.5), Camera.cpp:65-68; normalize = v * (1/sqrt(dot)) (GLM).
__device__ __forceinline__ void primary_ray(const float* cam, unsigned res_x, unsigned pix, float& dx, float& dy,
                                            float& dz) {
    const float x = (float)(int)(pix % res_x) + 0.5f, y = (float)(int)(pix / res_x) + 0.5f;
    const float vx = (cam[3] + x * cam[6]) + y * cam[9];
    const float vy = (cam[4] + x * cam[7]) + y * cam[10];
    const float vz = (cam[5] + x * cam[8]) + y * cam[11];
    const float sqr = vx * vx + vy * vy + vz * vz;
    const float inv = 1.0f / sqrtf(sqr);
    dx = vx * inv; dy = vy * inv; dz = vz * inv;
}

struct ShadowSeg { float px, py, pz, wx, wy, wz, dist_sq, dist; float nx, ny, nz; };

// Shadow-ray construction (DESIGN.md "Shadow batch"): p = o + t*d; N = normalize(cross(e0,e1))
// turned to face the primary ray; w = normalize(L - p); range (0, |L - p|).
__device__ __forceinline__ bool shadow_segment(const TraceParams& P, unsigned pair, ShadowSeg& s) {
    const unsigned pt = pair / P.n_lights, l = pair - pt * P.n_lights;
    const uint4 h = P.prim_hits[pt];
    if (h.x == kMiss) return false;
    const float4 o = P.prim_o[pt], d = P.prim_d[pt];
    const float t = __uint_as_float(h.y);
    s.px = o.x + t * d.x; s.py = o.y + t * d.y; s.pz = o.z + t * d.z;  // getPtAtDist, RTBase.cpp:124-126
    const float4 a = P.tris[kTriStride * (size_t)h.x], b = P.tris[kTriStride * (size_t)h.x + 1];
    const float e1z = P.tris[kTriStride * (size_t)h.x + 2].x;
    const float e0x = a.w, e0y = b.x, e0z = b.y, e1x = b.z, e1y = b.w;
    float nx = e0y * e1z - e1y * e0z, ny = e0z * e1x - e1z * e0x, nz = e0x * e1y - e1x * e0y;
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);
    nx = nx * inv; ny = ny * inv; nz = nz * inv;
    if ((d.x * nx + d.y * ny) + d.z * nz > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    s.nx = nx; s.ny = ny; s.nz = nz;
    const float vx = P.lights[3 * l] - s.px, vy = P.lights[3 * l + 1] - s.py, vz = P.lights[3 * l + 2] - s.pz;
    s.dist_sq = (vx * vx + vy * vy) + vz * vz;
    s.dist = sqrtf(s.dist_sq);
    const float iv = 1.0f / sqrtf(s.dist_sq);
    s.wx = vx * iv; s.wy = vy * iv; s.wz = vz * iv;
    return true;
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

template <int GEN>
__device__ __forceinline__ bool fetch_ray(const TraceParams& P, unsigned idx, RayReg& r) {
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
        if (!shadow_segment(P, idx, s)) return false;
        r.ox = s.px + kRayOffset * s.nx; r.oy = s.py + kRayOffset * s.ny; r.oz = s.pz + kRayOffset * s.nz;
        set_dir(r, s.wx, s.wy, s.wz);
        r.tmin = 0.0f; r.tmax = s.dist;
        return true;
    } else {
        // visibility of light l from a point in the volume: Ray{p, normalize(L - p)} (no normal,
        // so no origin offset), range (0, |L - p|)
        const unsigned pt = idx / P.n_lights, l = idx - pt * P.n_lights;
        const float4 p = P.points[pt];
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

// Flattened triangle record, one 64-byte line: (p0.xyz, e0.x) (e0.yz, e1.xy) (e1.z, mat, -, -) (-).
struct TriRec { float4 a, b; float e1z; };

// All three loads are issued before any use: one memory round trip per triangle (left to itself
// the compiler sinks the p0 load below the det != 0 test, which makes it two).
__device__ __forceinline__ TriRec load_tri(const float4* __restrict__ tris, unsigned id) {
    TriRec t;
    const float4* p = tris + kTriStride * (size_t)id;
    // plain (cached) loads: the same hint on these costs 60 %, the records live in L2 / Infinity Cache
    t.a = p[0]; t.b = p[1]; t.e1z = p[2].x;
    asm volatile("" : "+v"(t.a.x), "+v"(t.a.y), "+v"(t.a.z), "+v"(t.a.w));
    return t;
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
__device__ __forceinline__ void spill_store(uint2* p, uint2 e) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(e) : "memory");
}
__device__ __forceinline__ uint2 spill_load(const uint2* p) {
    uint2 e;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(p) : "memory");
    return e;
}

// Device node encoding (same topology and numbering as the reference's m_nodes, KdTree.h:22-65):
//   interior: w0 = split_pos, w1 = slot of the children pair << 2 | axis (bottom child first;
//             slots follow flatten_nodes' treelet order, not the reference's depth-first order)
//   leaf:     w1 bit 31 = 1 (so "still descending" is one signed compare), bits 29-30 = kind,
//             bits 0-28 = payload
//             kind 0: empty          kind 1: one prim, w0 = id
//             kind 2: two prims, w0 = id0, payload = id1 (saves the m_leaf_prim_ids round trip)
//             kind 3: n >= 3 prims, w0 = offset into leaf_ids, payload = n
// A lane without a ray always holds a leaf word, so the T-phase test needs no have_ray term.
template <bool ANY, int GEN, int STACK_N, int BLOCK, bool STATS = false>
__global__ void __launch_bounds__(BLOCK) trace_kernel(const TraceParams P) {
    __shared__ uint2 s_stack[STACK_N * BLOCK];
    __shared__ float s_eps[kEpsTable];
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    for (int i = (int)tid; i < kEpsTable; i += BLOCK) s_eps[i] = eps_from_exp(i - kEpsBias + 1);
    __syncthreads();
    auto eps_of = [&](float t) { return s_eps[(unsigned)(__builtin_amdgcn_frexp_expf(t) + kEpsBias)]; };
    uint2* const my_stack = s_stack + tid;
    uint2* const my_spill = P.spill + ((size_t)blockIdx.x * BLOCK + tid);

    RayReg r;
    float root_tmax = 0.0f;
    float best_t = FLT_MAX, best_u = 0.0f, best_v = 0.0f;
    unsigned best_id = kMiss;
    unsigned node = 0, ray_idx = 0;
    int sp = -1;
    bool have_ray = false, pool_empty = false;
    // wave-local slice of the ray index space, refilled CHUNK rays at a time with one atomic
    unsigned pool_next = 0, pool_end = 0, last_base = 0;
    // any-hit "group" mode: a wave owns 64 consecutive rays and writes their bits with one store
    unsigned group_base = kMiss;
    bool occ = false;
    const bool group_mode = ANY && !P.any_atomic;

    auto push = [&](float t_split, unsigned id) {
        ++sp;
        const uint2 e = make_uint2(__float_as_uint(t_split), id);
        if (STACK_N >= 32 || sp < STACK_N) my_stack[sp * BLOCK] = e;
        else spill_store(my_spill + (size_t)(sp - STACK_N) * P.spill_stride, e);
    };
    auto peek = [&](int k) -> uint2 {
        if (STACK_N >= 32 || k < STACK_N) return my_stack[k * BLOCK];
        return spill_load(my_spill + (size_t)(k - STACK_N) * P.spill_stride);
    };
    auto finish = [&](bool hit) {
        if (ANY) {
            if (group_mode) occ = hit;
            else if (hit) atomicOr(((unsigned*)P.bits) + (ray_idx >> 5), 1u << (ray_idx & 31u));
        } else {
            stream_store(P.hits + ray_idx, hit ? make_uint4(best_id, __float_as_uint(best_t), __float_as_uint(best_u),
                                                            __float_as_uint(best_v))
                                               : make_uint4(kMiss, __float_as_uint(FLT_MAX), 0u, 0u));
            if (P.ranges) P.ranges[ray_idx] = make_float2(r.tmin, r.tmax);
        }
        have_ray = false;
    };

    // Per-lane state machine.  A lane with a ray is either Traversing (nd is an interior node) or
    // at a Leaf (nd is a leaf, leaf_i = next primitive to test).  Every round of the outer loop runs
    // a T-phase (interior steps, repeated while at least t_exit lanes are still descending, so a
    // few stragglers do not stall the lanes already waiting at leaves) and an L-phase (one pair of
    // triangle tests per lane at a leaf).  Per ray the sequence of operations is the reference's.
    uint2 nd = make_uint2(0u, kLeafBit);
    unsigned leaf_i = 0;
    bool need_load = false;
    const int t_exit = P.t_exit;
    // STATS build only: wave-level round counts and lane-level work counts (tools/sweep.py stats=1)
    const unsigned long long st_t0 = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned st_rounds = 0, st_t_iters = 0, st_t_lanes = 0, st_l_rounds = 0, st_l_lanes = 0, st_refills = 0,
             st_refill_lanes = 0, st_rays_lanes = 0;

    for (;;) {
        const unsigned long long idle = __ballot(!have_ray);
        const int n_idle = __popcll(idle);
        if (STATS) { ++st_rounds; st_rays_lanes += 64 - n_idle; }
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
                if (lane == 0) base = atomicAdd(P.counter, take_n);
                base = __builtin_amdgcn_readfirstlane(base);
                last_base = base;
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
                        ray_idx = idx;
                        best_t = FLT_MAX; best_u = 0.0f; best_v = 0.0f; best_id = kMiss;
                        bool go = fetch_ray<GEN>(P, idx, r);
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
                        if (go) { have_ray = true; node = 0; sp = -1; need_load = true; leaf_i = 0; }
                        else finish(false);
                    }
                }
                pool_next = __builtin_amdgcn_readfirstlane(pool_next + take);
            }
        }
        if (__ballot(have_ray) == 0ull) {
            if (pool_empty) break;
            continue;
        }
        if (have_ray && need_load) {
            nd = P.nodes[node];
            need_load = false;
        }
        // ---- T-phase: KdTree.hpp:284-311 ----
        for (;;) {
            const bool is_t = (int)nd.y >= 0;
            const unsigned long long m_t = __ballot(is_t);
            if (m_t == 0ull) break;
            if (STATS) { ++st_t_iters; st_t_lanes += __popcll(m_t); }
            if (is_t) {
                const unsigned axis = nd.y & 3u;
                const unsigned pair = nd.y >> 2;   // slot of the bottom child; the top child is the next slot
#if GIGL_EARLY_PAIR
                // both children (one aligned 16-byte pair) are requested before the step is decided,
                // so the decision's arithmetic overlaps the memory round trip
                const uint4 kids = *reinterpret_cast<const uint4*>(P.nodes + pair);
#endif
                const float split = __uint_as_float(nd.x);
                const float oa = sel3(axis, r.ox, r.oy, r.oz);
                const float ia = sel3(axis, r.ix, r.iy, r.iz);
                // KdTree.hpp:289-291: below the plane, or on it and not heading up.  The direction
                // only matters for an origin exactly on the plane, so it is selected on that
                // (wave-uniform, rarely taken) branch only.
                bool bottom_first = oa < split;
                if (__ballot(oa == split) != 0ull) {
                    asm volatile("" ::: "memory");  // keeps this a real branch (else it is if-converted)
                    bottom_first = bottom_first | ((oa == split) & (sel3(axis, r.dx, r.dy, r.dz) <= 0.0f));
                }
                const float t_split = (split - oa) * ia;
                const bool only_first = (t_split > r.tmax) | (t_split <= 0.0f);
                const bool only_second = t_split < r.tmin;
                const bool take_second = !only_first & only_second;
                // first = bottom_first ? pair : pair + 1, second = the other slot
                if (!(only_first | only_second)) {  // both children: the far one waits on the stack
                    push(t_split, pair + (bottom_first ? 1u : 0u));
                    r.tmax = t_split + eps_of(t_split);
                }
#if GIGL_EARLY_PAIR
                nd = (bottom_first == take_second) ? make_uint2(kids.z, kids.w) : make_uint2(kids.x, kids.y);
#else
                node = pair + (bottom_first == take_second ? 1u : 0u);
                nd = P.nodes[node];
#endif
            }
            if ((int)__popcll(m_t) < t_exit) break;
        }
        // ---- L-phase: KdTree.hpp:312-343.  Prims are tested in m_leaf_prim_ids order, two per
        // round; both triangles' loads are issued before the first test. ----
        if (STATS) {
            const unsigned long long m_l = __ballot(have_ray && (int)nd.y < 0);
            if (m_l) { ++st_l_rounds; st_l_lanes += __popcll(m_l); }
        }
        if (have_ray && (int)nd.y < 0) {
            const unsigned kind = (nd.y >> 29) & 3u;
            const unsigned n_prims = kind == 3u ? (nd.y & kLeafPayload) : kind;
            bool any_done = false;
            if (leaf_i < n_prims) {
                unsigned id0, id1;
                const bool two = leaf_i + 1 < n_prims;
                if (kind == 3u) {
                    id0 = P.leaf_ids[nd.x + leaf_i];
                    id1 = two ? P.leaf_ids[nd.x + leaf_i + 1] : id0;
                } else {
                    id0 = nd.x;
                    id1 = two ? (nd.y & kLeafPayload) : id0;
                }
                const TriRec t0 = load_tri(P.tris, id0);
                const TriRec t1 = load_tri(P.tris, id1);
                bool hit = tri_test(t0, id0, r, best_t, best_u, best_v, best_id);
                if (!(ANY && hit) && two) hit = tri_test(t1, id1, r, best_t, best_u, best_v, best_id) || hit;
                any_done = ANY && hit;
                leaf_i += 2;
            }
            if (any_done) {
                finish(true);
            } else if (leaf_i >= n_prims) {
                if (best_id != kMiss) {
                    finish(true);  // closest hit inside this leaf's cell: KdTree.hpp:320,332
                } else if (sp >= 0) {
                    // ---- pop: KdTree.hpp:334-339.  The reference stores (t_min, t_max, id); t_min is
                    // t_split - eps(t_split) (:305) and the stored t_max is always the t_max set when
                    // the entry below was pushed, t_split' + eps(t_split') (:310), or the root t_max
                    // for the bottom entry, so both are rebuilt from the stored t_split values.
                    const uint2 e = peek(sp);
                    const float ts = __uint_as_float(e.x);
                    node = e.y;
                    r.tmin = ts - eps_of(ts);
                    if (sp > 0) {
                        const float tb = __uint_as_float(peek(sp - 1).x);
                        r.tmax = tb + eps_of(tb);
                    } else {
                        r.tmax = root_tmax;
                    }
                    --sp;
                    if (best_t < r.tmin) finish(false);  // KdTree.hpp:280
                    else { need_load = true; leaf_i = 0; }
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
    if (STATS && lane == 0 && P.stats) {
        atomicAdd(P.stats + 0, (unsigned long long)st_rounds);
        atomicAdd(P.stats + 1, (unsigned long long)st_t_iters);
        atomicAdd(P.stats + 2, (unsigned long long)st_t_lanes);
        atomicAdd(P.stats + 3, (unsigned long long)st_l_rounds);
        atomicAdd(P.stats + 4, (unsigned long long)st_l_lanes);
        atomicAdd(P.stats + 5, (unsigned long long)st_refills);
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

// ---- ray regrouping: coarse counting sort by (direction octant, Morton cell of the origin) -----
// Incoherent batches are traced through a permutation that puts rays which start in the same cell
// of the tree box and head into the same octant next to each other, so the lanes of a wave touch
// the same cache lines near the top of the tree and in the first leaves.  Results are written by
// original ray index; the order inside a bin (decided by atomics) never reaches the output.
constexpr unsigned kBinsPerBlock = 8192;

__device__ __forceinline__ unsigned spread3(unsigned v) {  // 0b...cba -> 0b..c00b00a
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256) ray_key_kernel(const float4* __restrict__ o_tmin,
                                                      const float4* __restrict__ d_tmax, unsigned n, BoxF box,
                                                      int bits, unsigned* __restrict__ keys,
                                                      unsigned* __restrict__ hist) {
    const float cells = (float)(1u << bits);
    const float sx = cells / (box.mxx - box.mnx), sy = cells / (box.mxy - box.mny), sz = cells / (box.mxz - box.mnz);
    const float top = cells - 1.0f;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 o = o_tmin[i], d = d_tmax[i];
        // fminf/fmaxf drop NaNs, so any input lands in a valid cell
        const unsigned qx = (unsigned)fminf(fmaxf((o.x - box.mnx) * sx, 0.0f), top);
        const unsigned qy = (unsigned)fminf(fmaxf((o.y - box.mny) * sy, 0.0f), top);
        const unsigned qz = (unsigned)fminf(fmaxf((o.z - box.mnz) * sz, 0.0f), top);
        const unsigned oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
        const unsigned key = (oct << (3 * bits)) | spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);
        keys[i] = key;
        atomicAdd(&hist[key], 1u);
    }
}

__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned* lds /*[256]*/, unsigned& total) {
    const unsigned t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        const unsigned add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[255];
    return lds[t] - v;
}

__global__ void __launch_bounds__(256) bin_block_sums_kernel(const unsigned* __restrict__ hist,
                                                             unsigned* __restrict__ block_sums) {
    __shared__ unsigned lds[256];
    const unsigned* h = hist + (size_t)blockIdx.x * kBinsPerBlock;
    unsigned sum = 0;
    for (unsigned k = threadIdx.x; k < kBinsPerBlock; k += 256u) sum += h[k];
    unsigned total;
    (void)block_exclusive_scan_256(sum, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// in: counts, out: exclusive start offsets (used as scatter cursors)
__global__ void __launch_bounds__(256) bin_scan_kernel(unsigned* __restrict__ hist,
                                                       const unsigned* __restrict__ block_sums) {
    __shared__ unsigned lds[256];
    unsigned before = 0;
    for (unsigned k = threadIdx.x; k < blockIdx.x; k += 256u) before += block_sums[k];
    unsigned base;
    (void)block_exclusive_scan_256(before, lds, base);
    __syncthreads();
    unsigned* h = hist + (size_t)blockIdx.x * kBinsPerBlock;
    constexpr unsigned per = kBinsPerBlock / 256;
    unsigned loc[per];
    unsigned sum = 0;
#pragma unroll
    for (unsigned k = 0; k < per; ++k) { loc[k] = h[threadIdx.x * per + k]; sum += loc[k]; }
    unsigned total;
    unsigned run = base + block_exclusive_scan_256(sum, lds, total);
#pragma unroll
    for (unsigned k = 0; k < per; ++k) { h[threadIdx.x * per + k] = run; run += loc[k]; }
}

__global__ void __launch_bounds__(256) ray_scatter_kernel(const unsigned* __restrict__ keys, unsigned n,
                                                          unsigned* __restrict__ cursor, unsigned* __restrict__ perm) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned pos = atomicAdd(&cursor[keys[i]], 1u);
        perm[pos] = i;
    }
}

// ---- ray generation as its own pass (for callers that want the rays) ----------------------
struct Cam12 { float c[12]; };
__global__ void __launch_bounds__(256) primary_rays_kernel(const Cam12 cam, unsigned res_x, unsigned first,
                                                           unsigned n, float4* o_tmin, float4* d_tmax) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        float dx, dy, dz;
        primary_ray(cam.c, res_x, first + i, dx, dy, dz);
        o_tmin[i] = make_float4(cam.c[0], cam.c[1], cam.c[2], 0.0f);
        d_tmax[i] = make_float4(dx, dy, dz, FLT_MAX);
    }
}

// ---- fog: sampleDensity and the GLSL ray march ----------------------------------------------
struct FogParams {
    // Corner bricks: brick (bx,by,bz), bx in [0,rx], holds the 8 texels (x,y,z) with x in
    // {bx-1, bx}, ... as bytes, byte index = dx + 2*dy + 4*dz, texels outside the volume 0
    // (DensityField::sample, DensityField.cpp:204-211).  One 8-byte load per trilinear sample.
    const uint2* bricks;
    int rx, ry, rz;
    float mnx, mny, mnz, mxx, mxy, mxz;
    float ext_k;
    int n_intervals;
};

// b / 255.0f (DensityField.cpp:210) without the divide: one Newton correction of b * (1/255) gives
// the correctly rounded quotient for every byte value (checked exhaustively,
// tests/test_host_logic.py::test_byte_unorm_is_exact_division).
__device__ __forceinline__ float byte_unorm(unsigned b) {
    const float rcp = 1.0f / 255.0f, fb = (float)b;
    const float r = fb * rcp;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, r, fb), rcp, r);
}
__device__ __forceinline__ float lerp1(float p0, float p1, float t) { return (1.0f - t) * p0 + t * p1; }
// DensityField.cpp:180-201 from a normalized position; lerp order x, y, z (Interpolation.hpp:8-23)
__device__ __forceinline__ float fog_filter(const FogParams& F, float nx, float ny, float nz) {
    const float tcx = nx * (float)F.rx - 0.5f, tcy = ny * (float)F.ry - 0.5f, tcz = nz * (float)F.rz - 0.5f;
    const float fx = floorf(tcx), fy = floorf(tcy), fz = floorf(tcz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx = tcx - (float)x0, ty = tcy - (float)y0, tz = tcz - (float)z0;
    // The reference's upper texel is ceil(tc), which equals x0 when tc is an integer; the brick's
    // is x0 + 1.  They differ only when the weight t is exactly 0, where (1-0)*a + 0*b = a for
    // either b, so the filtered value is the same bit for bit.
    const unsigned bx = (unsigned)x0 + 1u, by = (unsigned)y0 + 1u, bz = (unsigned)z0 + 1u;
    uint2 w = make_uint2(0u, 0u);   // every texel outside: the lerps still run (inf/NaN weights propagate)
    if (bx <= (unsigned)F.rx && by <= (unsigned)F.ry && bz <= (unsigned)F.rz)
        w = F.bricks[bx + (size_t)by * (unsigned)(F.rx + 1) + (size_t)bz * (unsigned)(F.rx + 1) * (unsigned)(F.ry + 1)];
    const float d000 = byte_unorm(w.x & 0xFFu), d100 = byte_unorm((w.x >> 8) & 0xFFu);
    const float d010 = byte_unorm((w.x >> 16) & 0xFFu), d110 = byte_unorm(w.x >> 24);
    const float d001 = byte_unorm(w.y & 0xFFu), d101 = byte_unorm((w.y >> 8) & 0xFFu);
    const float d011 = byte_unorm((w.y >> 16) & 0xFFu), d111 = byte_unorm(w.y >> 24);
    const float i0 = lerp1(lerp1(d000, d100, tx), lerp1(d010, d110, tx), ty);
    const float i1 = lerp1(lerp1(d001, d101, tx), lerp1(d011, d111, tx), ty);
    return lerp1(i0, i1, tz);
}
// GLSL min/max as selects (Surface.frag:146-153)
__device__ __forceinline__ float smin(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float smax(float x, float y) { return x < y ? y : x; }

// calcTransm + intersectBBox + rayMarch, Surface.frag:139-203
__device__ __forceinline__ float calc_transm(const FogParams& F, float px, float py, float pz, float wx, float wy,
                                             float wz, float dist_sq) {
    if (!(F.ext_k > 0.0f)) return 1.0f;
    const float max_dist = sqrtf(dist_sq);
    const float ivx = 1.0f / wx, ivy = 1.0f / wy, ivz = 1.0f / wz;
    float t0 = (F.mnx - px) * ivx, t1 = (F.mxx - px) * ivx;
    float t_min = smin(t0, t1), t_max = smax(t0, t1);
    t0 = (F.mny - py) * ivy; t1 = (F.mxy - py) * ivy;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    t0 = (F.mnz - pz) * ivz; t1 = (F.mxz - pz) * ivz;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    if (!(t_max > smax(t_min, 0.0f) && t_min < max_dist)) return 1.0f;
    t_min = smax(t_min, 0.0f);
    t_max = smin(t_max, max_dist);
    // inv_fog_dims = 1/(max-min), Source/GIGL.cpp:83
    const float idx = 1.0f / (F.mxx - F.mnx), idy = 1.0f / (F.mxy - F.mny), idz = 1.0f / (F.mxz - F.mnz);
    const float dt = (t_max - t_min) / (float)F.n_intervals;
    auto ext_at = [&](float t) {
        const float qx = px + t * wx, qy = py + t * wy, qz = pz + t * wz;
        return F.ext_k * fog_filter(F, (qx - F.mnx) * idx, (qy - F.mny) * idy, (qz - F.mnz) * idz);
    };
    float prev = ext_at(t_min), total = 0.0f;
    for (int i = 1; i <= F.n_intervals; ++i) {
        const float cur = ext_at(t_min + (float)i * dt);
        total += 0.5f * (prev + cur);
        prev = cur;
    }
    return expf(-(total * dt));
}

__global__ void __launch_bounds__(256) transm_segments_kernel(FogParams F, const float4* seg_p,
                                                              const float4* seg_d, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 p = seg_p[i], d = seg_d[i];
        out[i] = calc_transm(F, p.x, p.y, p.z, d.x, d.y, d.z, d.w);
    }
}

__global__ void __launch_bounds__(256) transm_shadow_kernel(FogParams F, TraceParams P, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < P.n; i += gridDim.x * 256u) {
        ShadowSeg s;
        float tr = 1.0f;
        if (shadow_segment(P, i, s)) tr = calc_transm(F, s.px, s.py, s.pz, s.wx, s.wy, s.wz, s.dist_sq);
        out[i] = tr;
    }
}

// DensityField::computePiDensity's march (Source/Fog/DensityField.cpp:288-321) for one pixel per
// lane, after the primary closest-hit pass has produced `hits`.  Stores of one z-plane are
// coalesced across the lanes of a wave (x fastest).
__global__ void __launch_bounds__(256) pi_density_kernel(FogParams F, Cam12 cam, unsigned res_x, unsigned n_pix,
                                                         const uint4* __restrict__ hits, float* __restrict__ out) {
    const BoxF fb = {F.mnx, F.mny, F.mnz, F.mxx, F.mxy, F.mxz};
    for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n_pix; p += gridDim.x * 256u) {
        RayReg r;
        float dx, dy, dz;
        primary_ray(cam.c, res_x, p, dx, dy, dz);
        r.ox = cam.c[0]; r.oy = cam.c[1]; r.oz = cam.c[2];
        set_dir(r, dx, dy, dz);
        r.tmin = 0.0f; r.tmax = FLT_MAX;
        float entr, exit;
        if (box_intersect(fb, r, entr, exit)) {
            const float dist = __uint_as_float(hits[p].y);   // FLT_MAX on a miss, as Ray::Intersection starts
            const float t_min = gmax(entr, 0.0f), t_max = gmin(exit, dist);
            const int n_intervals = F.rz * 4;
            const float dt = (t_max - t_min) / (float)n_intervals;
            const float ivx = F.mxx - F.mnx, ivy = F.mxy - F.mny, ivz = F.mxz - F.mnz;
            auto dens_at = [&](float t) {
                const float qx = r.ox + t * r.dx, qy = r.oy + t * r.dy, qz = r.oz + t * r.dz;
                return fog_filter(F, (qx - F.mnx) / ivx, (qy - F.mny) / ivy, (qz - F.mnz) / ivz);  // computeNormPos
            };
            float prev = dens_at(t_min), dens = 0.0f;
            for (int i = 1; i <= n_intervals; ++i) {
                const float cur = dens_at(t_min + (float)i * dt);
                dens += 0.5f * (cur + prev);
                prev = cur;
                if (2 == i % 4) out[p + (size_t)(i / 4) * n_pix] = dens * dt;
            }
        } else {
            for (int z = 0; z < F.rz; ++z) out[p + (size_t)z * n_pix] = 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// software shading stage: Surface.frag main() (Source/Shaders/Surface.frag:290-343) and
// Combine.frag's normalisation + tone mapping (:72-76) as compute kernels over the hit buffers,
// occlusion bits and transmittance the ray batches produce.  Stand-ins for the GL pipeline are
// listed in DESIGN.md "Shading stage"; the arithmetic mirrors oracle/gigl_oracle.cpp step by step.
// ------------------------------------------------------------------------------------------
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 ld3(const float* p) { return F3{p[0], p[1], p[2]}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float k) { return F3{a.x * k, a.y * k, a.z * k}; }
__device__ __forceinline__ F3 operator*(float k, F3 a) { return F3{k * a.x, k * a.y, k * a.z}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float len3(F3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
__device__ __forceinline__ F3 norm3(F3 v) { return v * (1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z)); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return F3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}

// phongBRDF, Surface.frag:205-217
__device__ __forceinline__ F3 phong_brdf(F3 I, F3 N, F3 O, F3 k_d, F3 k_s, float n_s) {
    F3 brdf = k_d;
    if (n_s > 0.0f) {
        const F3 mI = -I;
        const F3 S = mI - (2.0f * dot3(N, mI)) * N;   // reflect(-I, N)
        const float c = dot3(O, S);
        if (c > 0.0f) brdf = brdf + k_s * powf(c, n_s);
    }
    return brdf;
}
// evalPhaseHG, Surface.frag:219-223
__device__ __forceinline__ float phase_hg(float cos_the) {
    const float g = 0.25f;
    const float base = 1.0f + g * g - 2.0f * g * cos_the;
    return 0.25f * 0.318309873f * (1.0f - g * g) / (base * sqrtf(base));
}
// computeLe, Surface.frag:225-242.  v: one 96-byte VPL record as 24 words (PointLight.h:66-77)
__device__ __forceinline__ F3 compute_le(const float* v, F3 O, float sca_albedo) {
    const unsigned type = __float_as_uint(v[3]);
    if (type == 1u) return (phase_hg(dot3(O, ld3(v + 8))) * sca_albedo) * ld3(v + 4);
    if (type == 2u) {
        const F3 w_norm = ld3(v + 12);
        const float cos_out = smax(0.0f, dot3(O, w_norm));
        return (phong_brdf(ld3(v + 8), w_norm, O, ld3(v + 16), ld3(v + 20), v[23]) * ld3(v + 4)) * cos_out;
    }
    return f3(0.0f, 0.0f, 0.0f);
}

struct ShadeParams {
    float cam[3];
    unsigned res_x, res_y, first, n_pts, n_ppls, n_vpls;
    int gi_enabled, clamp_rsq, frame_id, pi_res_z;
    float sca_albedo;
    const float4* prim_o;
    const float4* prim_d;
    const uint4* hits;
    const float4* tris;
    const float* materials;          // 12 floats per material: k_d, pad, k_s, n_s, k_e, pad (RTBase.h:72-77)
    const float* ppls;               // 8 floats per light: w_pos, pad, intens, pad (std140)
    const float* vpls;               // 24 words per light
    const unsigned long long* bits;  // shadow batch occlusion bits, pair = point * n_lights + light
    const float* transm;             // shadow batch transmittance per pair, or NULL
    const float* pi_dens;            // computePiDensity's output, or NULL without fog
    float4* accum;
    float2* fog_dist;
};

constexpr int kShadeTile = 16;   // lights per transmittance tile

// One pixel per lane, lights in the reference's order (primary lights, then VPLs).  The pair-major
// transmittance array is read through a 256 x 16 LDS tile so that the global loads are coalesced.
__global__ void __launch_bounds__(256) shade_surface_kernel(FogParams F, ShadeParams S) {
    __shared__ float s_tr[256 * (kShadeTile + 1)];
    const unsigned tid = threadIdx.x;
    const unsigned pt0 = blockIdx.x * 256u, pt = pt0 + tid;
    const unsigned n_lights = S.n_ppls + S.n_vpls;
    const bool in_range = pt < S.n_pts;
    F3 col = f3(0.0f, 0.0f, 0.0f);
    float f_min = 0.0f, f_max = 0.0f, transm_frag = 1.0f;
    bool lit = false;
    F3 w_pos = col, N = col, O = col, k_d = col, k_s = col, d = col;
    float n_s = 0.0f;
    uint4 h = make_uint4(kMiss, 0u, 0u, 0u);
    if (in_range) h = S.hits[pt];
    if (in_range && S.frame_id < 30 && h.x != kMiss) {                       // Surface.frag:292
        const float4 o4 = S.prim_o[pt], d4 = S.prim_d[pt];
        d = f3(d4.x, d4.y, d4.z);
        w_pos = f3(o4.x, o4.y, o4.z) + __uint_as_float(h.y) * d;
        const F3 cam = f3(S.cam[0], S.cam[1], S.cam[2]);
        const F3 ray_d = norm3(w_pos - cam);                                 // :296
        if (F.ext_k > 0.0f) {                                                // :297-315
            const float t_frag = len3(w_pos - cam);
            const float ivx = 1.0f / ray_d.x, ivy = 1.0f / ray_d.y, ivz = 1.0f / ray_d.z;
            float t0 = (F.mnx - cam.x) * ivx, t1 = (F.mxx - cam.x) * ivx;
            float t_min = smin(t0, t1), t_max = smax(t0, t1);
            t0 = (F.mny - cam.y) * ivy; t1 = (F.mxy - cam.y) * ivy;
            t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
            t0 = (F.mnz - cam.z) * ivz; t1 = (F.mxz - cam.z) * ivz;
            t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
            if (t_max > smax(t_min, 0.0f) && t_min < t_frag) {
                t_min = smax(t_min, 0.0f);
                t_max = smin(t_max, t_frag);
                const unsigned pix = S.first + pt;
                const unsigned px = pix % S.res_x, py = pix / S.res_x;
                const float z = (t_frag - t_min) / (t_max - t_min);
                // texture(pi_dens): nearest in x,y, linear in z, CLAMP_TO_EDGE (DensityField.cpp:376-382)
                float tz = z * (float)S.pi_res_z - 0.5f;
                tz = smin(smax(tz, 0.0f), (float)(S.pi_res_z - 1));
                const float fz = floorf(tz);
                const int z0 = (int)fz, z1 = z0 + 1 < S.pi_res_z ? z0 + 1 : S.pi_res_z - 1;
                const float f = tz - fz;
                const size_t plane = (size_t)S.res_x * S.res_y, at = px + (size_t)py * S.res_x;
                const float a = S.pi_dens[at + (size_t)z0 * plane], b = S.pi_dens[at + (size_t)z1 * plane];
                const float opt_depth = F.ext_k * ((1.0f - f) * a + f * b);
                transm_frag = expf(-opt_depth);
            } else {
                t_min = t_max = 0.0f;
            }
            f_min = t_min; f_max = t_max;
        }
        if (transm_frag > 0.001f) {                                           // :316-318
            lit = true;
            const float4 ta = S.tris[kTriStride * (size_t)h.x], tb = S.tris[kTriStride * (size_t)h.x + 1];
            const float4 tc = S.tris[kTriStride * (size_t)h.x + 2];
            N = norm3(cross3(f3(ta.w, tb.x, tb.y), f3(tb.z, tb.w, tc.x)));
            if (dot3(d, N) > 0.0f) N = -N;
            const float* m = S.materials + 12 * (size_t)__float_as_uint(tc.y);
            k_d = ld3(m); k_s = ld3(m + 4); n_s = m[7];
            O = -ray_d;
        }
    }
    for (unsigned l0 = 0; l0 < n_lights; l0 += kShadeTile) {
        if (S.transm) {
            for (unsigned k = 0; k < (unsigned)kShadeTile; ++k) {
                const unsigned idx = tid + k * 256u, p = idx / kShadeTile, l = idx % kShadeTile;
                float v = 1.0f;
                if (pt0 + p < S.n_pts && l0 + l < n_lights) v = S.transm[(size_t)(pt0 + p) * n_lights + l0 + l];
                s_tr[p * (kShadeTile + 1) + l] = v;
            }
            __syncthreads();
        }
        if (lit) {
            const unsigned l_end = l0 + kShadeTile < n_lights ? l0 + kShadeTile : n_lights;
            for (unsigned l = l0; l < l_end; ++l) {
                const bool is_ppl = l < S.n_ppls;
                if (is_ppl ? !S.clamp_rsq : !S.gi_enabled) continue;                  // :319, :325
                const size_t pair = (size_t)pt * n_lights + l;
                if ((S.bits[pair >> 6] >> (pair & 63u)) & 1ull) continue;             // visibility > 0.0
                const float* v = S.vpls + 24 * (size_t)(is_ppl ? 0u : l - S.n_ppls);
                const F3 L = is_ppl ? ld3(S.ppls + 8 * l) : ld3(v);
                const F3 dd = w_pos - L;
                const float dist_sq = dot3(dd, dd);
                const F3 I = norm3(-dd);
                const float tr = S.transm ? s_tr[tid * (kShadeTile + 1) + (l - l0)] : 1.0f;
                F3 Li;
                if (is_ppl) {                                                          // :255-258
                    const float falloff = 1.0f / dist_sq;
                    Li = (tr * ld3(S.ppls + 8 * l + 4)) * falloff;
                } else {                                                               // :278-281
                    const float falloff = 1.0f / smax(dist_sq, 75.0f * 75.0f);
                    Li = (tr * compute_le(v, -I, S.sca_albedo)) * falloff;
                }
                const float cos_inc = smax(0.0f, dot3(I, N));
                const F3 c = (phong_brdf(I, N, O, k_d, k_s, n_s) * Li) * cos_inc;
                col = col + transm_frag * c;                                           // :322, :328
            }
        }
        if (S.transm) __syncthreads();
    }
    if (in_range && S.frame_id < 30) {
        if (S.frame_id > 0) {                                                          // :332-337
            const float4 a = S.accum[pt];
            col.x += a.x; col.y += a.y; col.z += a.z;
        }
        S.accum[pt] = make_float4(col.x, col.y, col.z, 1.0f);                          // :339
        if (S.fog_dist && F.ext_k > 0.0f) S.fog_dist[pt] = make_float2(f_min, f_max);  // recordRayDist, :314
    }
}

// ---- Volume.frag (Source/Shaders/Volume.frag:268-314): half-resolution in-scattering ----------
struct VolumeParams {
    float cam[3];
    unsigned res_x, res_y;            // full resolution (CAM_RES of Surface.frag); this pass runs at res/2
    unsigned n_samples;               // gi_enabled ? MAX_VOL_SAMP / 4 : MAX_VOL_SAMP
    unsigned n_ppls, n_vpls;
    int gi_enabled, clamp_rsq, transm_opt, frame_id, pi_res_z;
    float sca_k, sca_albedo;
    const float4* prim_o;             // full-resolution primary rays and hits (the G-buffer stand-in)
    const float4* prim_d;
    const uint4* hits;
    const float2* fog_dist;           // full resolution, written by shade_surface_kernel
    const float* rnd_offsets;         // res/2 x res/2
    const float* halton;              // MAX_FRAMES * MAX_VOL_SAMP values
    const float* pi_dens;
    const float* ppls;
    const float* vpls;
    const unsigned long long* bits;   // pair = (pixel * n_samples + sample) * n_lights + light
    float4* points;                   // volume_samples_kernel: (s_pos, valid)
    float4* vol_comp;                 // shade_volume_kernel: rgb, 1
};

struct VolSample { F3 pos; float transm; bool valid; };
struct VolPixel { F3 ray_d; float t_min, t_max, z_offset; unsigned tx, ty; bool fog; };

// per-pixel part of main(), Volume.frag:272-282
__device__ __forceinline__ VolPixel volume_pixel(const VolumeParams& V, unsigned hx, unsigned hy) {
    VolPixel px;
    const size_t full = (size_t)(2u * hx + 1u) + (size_t)(2u * hy + 1u) * V.res_x;   // texel (2x+1, 2y+1), :90, :95
    const float2 fd = V.fog_dist[full];
    px.t_min = fd.x; px.t_max = fd.y;
    px.fog = px.t_max > 0.0f && V.frame_id < 30;
    px.ray_d = f3(0.0f, 0.0f, 0.0f);
    px.z_offset = 0.0f;
    // gl_FragCoord.xy / CAM_RES with `#define CAM_RES 1024 / 2` is (xy / 1024) / 2 (Volume.frag:4,
    // :285): the lookup lands in the lower-left quarter of pi_dens, nearest texel = (hx/2, hy/2)
    px.tx = hx >> 1; px.ty = hy >> 1;
    if (px.fog) {
        const float4 o4 = V.prim_o[full], d4 = V.prim_d[full];
        const uint4 h = V.hits[full];
        const F3 w_pos = f3(o4.x, o4.y, o4.z) + __uint_as_float(h.y) * f3(d4.x, d4.y, d4.z);
        px.ray_d = norm3(w_pos - f3(V.cam[0], V.cam[1], V.cam[2]));
        px.z_offset = V.rnd_offsets[hx + (size_t)hy * (V.res_x / 2u)];
    }
    return px;
}
// per-sample part, Volume.frag:285-291
__device__ __forceinline__ VolSample volume_sample(const FogParams& F, const VolumeParams& V, const VolPixel& px,
                                                  unsigned s) {
    VolSample smp;
    const float hz = V.halton[s + V.n_samples * (unsigned)V.frame_id] + px.z_offset;
    const float z = smin(smax(hz, 0.0f), 0.995f);                                  // clamp
    const float t = px.t_min + z * (px.t_max - px.t_min);
    smp.pos = f3(V.cam[0], V.cam[1], V.cam[2]) + t * px.ray_d;
    float tz = z * (float)V.pi_res_z - 0.5f;
    tz = smin(smax(tz, 0.0f), (float)(V.pi_res_z - 1));
    const float fz = floorf(tz);
    const int z0 = (int)fz, z1 = z0 + 1 < V.pi_res_z ? z0 + 1 : V.pi_res_z - 1;
    const float f = tz - fz;
    const size_t plane = (size_t)V.res_x * V.res_y, at = px.tx + (size_t)px.ty * V.res_x;
    const float a = V.pi_dens[at + (size_t)z0 * plane], b = V.pi_dens[at + (size_t)z1 * plane];
    smp.transm = expf(-(F.ext_k * ((1.0f - f) * a + f * b)));
    smp.valid = smp.transm > 0.001f;                                               // :292
    return smp;
}

__global__ void __launch_bounds__(256) volume_samples_kernel(FogParams F, VolumeParams V) {
    const unsigned hw = V.res_x / 2u, n_pix = hw * (V.res_y / 2u);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_pix * V.n_samples; i += gridDim.x * 256u) {
        const unsigned pix = i / V.n_samples, s = i - pix * V.n_samples;
        const VolPixel px = volume_pixel(V, pix % hw, pix / hw);
        float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (px.fog) {
            const VolSample smp = volume_sample(F, V, px, s);
            if (smp.valid) out = make_float4(smp.pos.x, smp.pos.y, smp.pos.z, 1.0f);
        }
        V.points[i] = out;
    }
}

// calcTransm of Volume.frag:159-181 (transm_opt: constant medium at the sample point's density)
__device__ __forceinline__ float calc_transm_vol(const FogParams& F, F3 p, float density, F3 w, float dist_sq,
                                                 int transm_opt) {
    if (!transm_opt) return calc_transm(F, p.x, p.y, p.z, w.x, w.y, w.z, dist_sq);
    if (!(F.ext_k > 0.0f)) return 1.0f;
    const float max_dist = sqrtf(dist_sq);
    const float ivx = 1.0f / w.x, ivy = 1.0f / w.y, ivz = 1.0f / w.z;
    float t0 = (F.mnx - p.x) * ivx, t1 = (F.mxx - p.x) * ivx;
    float t_min = smin(t0, t1), t_max = smax(t0, t1);
    t0 = (F.mny - p.y) * ivy; t1 = (F.mxy - p.y) * ivy;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    t0 = (F.mnz - p.z) * ivz; t1 = (F.mxz - p.z) * ivz;
    t_min = smax(t_min, smin(t0, t1)); t_max = smin(t_max, smax(t0, t1));
    if (!(t_max > smax(t_min, 0.0f) && t_min < max_dist)) return 1.0f;
    t_min = smax(t_min, 0.0f);
    t_max = smin(t_max, max_dist);
    return expf(-(F.ext_k * density * (t_max - t_min)));
}

// One half-resolution pixel per lane; samples and lights in the shader's order.
__global__ void __launch_bounds__(256) shade_volume_kernel(FogParams F, VolumeParams V) {
    const unsigned hw = V.res_x / 2u, n_pix = hw * (V.res_y / 2u);
    const unsigned n_lights = V.n_ppls + V.n_vpls;
    const float idx = 1.0f / (F.mxx - F.mnx), idy = 1.0f / (F.mxy - F.mny), idz = 1.0f / (F.mxz - F.mnz);
    for (unsigned pix = blockIdx.x * 256u + threadIdx.x; pix < n_pix; pix += gridDim.x * 256u) {
        const VolPixel px = volume_pixel(V, pix % hw, pix / hw);
        F3 col = f3(0.0f, 0.0f, 0.0f);
        if (px.fog) {
            const F3 O = -px.ray_d;
            for (unsigned s = 0; s < V.n_samples; ++s) {
                const VolSample smp = volume_sample(F, V, px, s);
                if (!smp.valid) continue;
                // calcFogDens, Volume.frag:121-126
                const float density = fog_filter(F, (smp.pos.x - F.mnx) * idx, (smp.pos.y - F.mny) * idy,
                                                 (smp.pos.z - F.mnz) * idz);
                for (unsigned l = 0; l < n_lights; ++l) {
                    const bool is_ppl = l < V.n_ppls;
                    if (is_ppl ? !V.clamp_rsq : !V.gi_enabled) continue;                  // :293, :299
                    const size_t pair = ((size_t)pix * V.n_samples + s) * n_lights + l;
                    if ((V.bits[pair >> 6] >> (pair & 63u)) & 1ull) continue;             // visibility > 0.0
                    const float* v = V.vpls + 24 * (size_t)(is_ppl ? 0u : l - V.n_ppls);
                    const F3 L = is_ppl ? ld3(V.ppls + 8 * l) : ld3(v);
                    const F3 dd = smp.pos - L;
                    const float dist_sq = dot3(dd, dd);
                    const F3 I = norm3(-dd);
                    const float tr = calc_transm_vol(F, smp.pos, density, I, dist_sq, V.transm_opt);
                    F3 Li;
                    if (is_ppl) Li = (tr * ld3(V.ppls + 8 * l + 4)) * (1.0f / dist_sq);                 // :235-236
                    else Li = (tr * compute_le(v, -I, V.sca_albedo)) * (1.0f / smax(dist_sq, 75.0f * 75.0f));  // :258-259
                    const float cos_the = -dot3(I, O);
                    const F3 c = ((phase_hg(cos_the) * V.sca_k) * density) * Li;                        // :239, :262
                    col = col + smp.transm * c;                                                          // :296, :302
                }
            }
            col = col * ((px.t_max - px.t_min) / (float)V.n_samples);                                    // :308-309
        }
        V.vol_comp[pix] = make_float4(col.x, col.y, col.z, 1.0f);
    }
}

// ---- Combine.frag (Source/Shaders/Combine.frag:23-77) --------------------------------------------
// GL depth of the primary hit point: lookAt + glm::perspective (Camera.cpp:30-31), quantised to
// DEPTH_COMPONENT24 (GLTexture2D.h:27); 1.0 (the cleared value) where the primary ray missed.
__global__ void __launch_bounds__(256) depth_kernel(F3 cam, F3 fwd, float t_near, float t_far,
                                                    const float4* __restrict__ prim_o, const float4* __restrict__ prim_d,
                                                    const uint4* __restrict__ hits, unsigned n, float* __restrict__ depth) {
    const float A = -(t_far + t_near) / (t_far - t_near), B = -(2.0f * t_far * t_near) / (t_far - t_near);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint4 h = hits[i];
        float d = 1.0f;
        if (h.x != kMiss) {
            const float4 o = prim_o[i], dir = prim_d[i];
            const F3 w_pos = f3(o.x, o.y, o.z) + __uint_as_float(h.y) * f3(dir.x, dir.y, dir.z);
            const float z_e = -dot3(fwd, w_pos - cam);
            const float z_ndc = (A * z_e + B) / -z_e;
            d = floorf((0.5f * z_ndc + 0.5f) * 16777215.0f + 0.5f) / 16777215.0f;
        }
        depth[i] = d;
    }
}

__device__ __forceinline__ F3 vol_texel(const float4* vol, int hw, int hh, int x, int y) {  // CLAMP_TO_BORDER, border 0
    if (x < 0 || x >= hw || y < 0 || y >= hh) return f3(0.0f, 0.0f, 0.0f);
    const float4 t = vol[(size_t)x + (size_t)y * hw];
    return f3(t.x, t.y, t.z);
}

__global__ void __launch_bounds__(256) combine_kernel(unsigned res_x, unsigned res_y, const float* __restrict__ depth,
                                                      const float4* __restrict__ vol, int add, int frame_id,
                                                      int exposure, float4* accum, float* __restrict__ rgb) {
    const int hw = (int)(res_x / 2u), hh = (int)(res_y / 2u);
    const float div = (float)(1 + (frame_id < 30 ? frame_id : 30));
    const unsigned n = res_x * res_y;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const int x = (int)(i % res_x), y = (int)(i / res_x);
        const float4 a4 = accum[i];
        F3 c = f3(a4.x, a4.y, a4.z);
        if (add) {                                       // bilateralUpsampling, :24-49
            const float center_depth = depth[i];
            float accum_weight = 0.0f;
            F3 acc = f3(0.0f, 0.0f, 0.0f);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int px = x + dx, py = y + dy;
                    const bool inside = px >= 0 && px < (int)res_x && py >= 0 && py < (int)res_y;
                    const float dpt = inside ? depth[(size_t)px + (size_t)py * res_x] : 0.0f;
                    // texture(vol_comp, vec2(pos) / CAM_RES), bilinear
                    const float fu = ((float)px / (float)res_x) * (float)hw - 0.5f;
                    const float fv = ((float)py / (float)res_y) * (float)hh - 0.5f;
                    const float x0f = floorf(fu), y0f = floorf(fv);
                    const int x0 = (int)x0f, y0 = (int)y0f;
                    const float a = fu - x0f, b = fv - y0f;
                    const F3 t00 = vol_texel(vol, hw, hh, x0, y0), t10 = vol_texel(vol, hw, hh, x0 + 1, y0);
                    const F3 t01 = vol_texel(vol, hw, hh, x0, y0 + 1), t11 = vol_texel(vol, hw, hh, x0 + 1, y0 + 1);
                    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
                    const F3 col = f3(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x,
                                      ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
                                      ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z);
                    const float weight = (1.0f - 0.3f * (float)abs(dx)) * (1.0f - 0.3f * (float)abs(dy)) /
                                         (smax(dpt - center_depth, 0.0f) * 1000.0f + 1.0f);
                    accum_weight += weight;
                    acc = f3(acc.x + weight * col.x, acc.y + weight * col.y, acc.z + weight * col.z);
                }
            c = f3(c.x + acc.x / accum_weight, c.y + acc.y / accum_weight, c.z + acc.z / accum_weight);
            accum[i] = make_float4(c.x, c.y, c.z, 1.0f);                                      // :69
        }
        rgb[3 * (size_t)i] = 1.0f - expf(-(float)exposure * (c.x / div));                     // :73-75
        rgb[3 * (size_t)i + 1] = 1.0f - expf(-(float)exposure * (c.y / div));
        rgb[3 * (size_t)i + 2] = 1.0f - expf(-(float)exposure * (c.z / div));
    }
}

// Combine.frag:72-76 (without the volume term)
__global__ void __launch_bounds__(256) tonemap_kernel(const float4* __restrict__ accum, unsigned n, int frame_id,
                                                      int exposure, float* __restrict__ rgb) {
    const float div = (float)(1 + (frame_id < 30 ? frame_id : 30));
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 a = accum[i];
        rgb[3 * (size_t)i] = 1.0f - expf(-(float)exposure * (a.x / div));
        rgb[3 * (size_t)i + 1] = 1.0f - expf(-(float)exposure * (a.y / div));
        rgb[3 * (size_t)i + 2] = 1.0f - expf(-(float)exposure * (a.z / div));
    }
}

__global__ void __launch_bounds__(256) sample_density_kernel(FogParams F, const float* pos, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        // computeNormPos: (pos - min) / dims, BBox.cpp:53-55
        const float nx = (pos[3 * i] - F.mnx) / (F.mxx - F.mnx);
        const float ny = (pos[3 * i + 1] - F.mny) / (F.mxy - F.mny);
        const float nz = (pos[3 * i + 2] - F.mnz) / (F.mxz - F.mnz);
        out[i] = fog_filter(F, nx, ny, nz);
    }
}

__global__ void __launch_bounds__(256) eps_kernel(const float* t, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = rescaled_eps(t[i]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? GIGL_ERR_NO_DEVICE \
                                                                              : GIGL_ERR_HIP,      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

struct Options {
    int stack_lds = 12;       // LDS entries per lane: 8, 12, 16 or 32
    int refill_thresh = 32;   // idle lanes that trigger a refill (closest-hit / atomic any-hit)
    int any_atomic = -1;      // 0: any-hit waves own 64 consecutive rays, one store per wave; 1: lane refill +
                              // atomicOr; -1: choose by tree depth (deep trees have long-tailed rays)
    int blocks_per_cu = 0;    // 0: occupancy query
    int chunk = 256;          // rays per work-counter atomic
    int guided = 4;           // 0: fixed chunks; k: near the end a request is remaining / (k * waves)
    int t_exit = 16;          // see trace_kernel: T-phase exit threshold (1..64)
    int stats = 0;            // 1: closest-hit launches run the instrumented build and keep 8 counters
    int sort_rays = 0;        // explicit-ray batches: 1 regroup rays before tracing, 0 never, -1 deep tree + large
                              // batch.  Off by default: on MI355X the two atomic passes (2.9 ms per 2^24 rays)
                              // cost more than the 1.7 ms the regrouped traversal saves
    int sort_bits = 6;        // cells per axis = 2^sort_bits (4..6)
    int treelet_levels = 3;   // read at gigl_scene_create: levels of child pairs per cache-line treelet (1..5)
};
Options g_opt;
std::mutex g_opt_mu;
unsigned long long g_last_stats[12] = {0};

constexpr int kBlock = 256;
constexpr unsigned kMaxRays = 0xF0000000u;

}  // namespace

struct gigl_scene {
    int device = 0;
    int n_cus = 0;
    gigl::KdTreeHost tree;
    uint32_t n_tris = 0;
    uint2* d_nodes = nullptr;
    unsigned* d_leaf_ids = nullptr;
    float4* d_tris = nullptr;
    BoxF box;
    gigl_tree_stats stats;
    // staging for gigl_trace_one
    float4* d_one = nullptr;   // 2 float4 in, 1 uint4 out
    std::mutex one_mu;
};

struct gigl_fog {
    int device = 0;
    uint2* d_bricks = nullptr;
    FogParams fp;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

unsigned stream_grid(const gigl_scene* s, uint64_t n) {
    uint64_t g = (n + 255) / 256;
    const uint64_t cap = (uint64_t)(s ? s->n_cus : 256) * 8;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

template <bool ANY, int GEN, int STACK_N>
int launch_trace_t(const gigl_scene* s, TraceParams& P, const Options& opt, hipStream_t st) {
    auto kern = trace_kernel<ANY, GEN, STACK_N, kBlock>;
    if (!ANY && GEN == 0 && STACK_N == 12 && opt.stats) kern = trace_kernel<ANY, GEN, 12, kBlock, true>;
    int per_cu = opt.blocks_per_cu;
    if (per_cu <= 0) {
        static int cached[2] = {0, 0};  // per instantiation; [1] is the instrumented build
        int& c = cached[(!ANY && GEN == 0 && STACK_N == 12 && opt.stats) ? 1 : 0];
        if (c <= 0) {
            int q = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, kBlock, 0));
            c = q < 1 ? 1 : q;
        }
        per_cu = c;
    }
    unsigned grid = (unsigned)(s->n_cus * per_cu);
    if (getenv("GIGL_RT_DEBUG")) fprintf(stderr, "[gigl_rt] trace launch: %d blocks/CU x %d CUs\n", per_cu, s->n_cus);
    const unsigned need = (P.n + kBlock - 1) / kBlock;
    if (grid > need) grid = need;
    if (grid == 0) return GIGL_OK;
    const bool spill = STACK_N < 32 && s->tree.actual_depth - 1 > STACK_N;
    const size_t spill_entries = spill ? (size_t)grid * kBlock * (size_t)(32 - STACK_N) : 0;
    const size_t ws_bytes = 256 + spill_entries * sizeof(uint2);
    void* ws = nullptr;
    HIP_TRY(hipMallocAsync(&ws, ws_bytes, st));
    if (hipMemsetAsync(ws, 0, 256, st) != hipSuccess) {
        (void)hipFreeAsync(ws, st);
        return fail(GIGL_ERR_HIP, "hipMemsetAsync(workspace)");
    }
    P.counter = (unsigned*)ws;
    P.spill = (uint2*)((char*)ws + 256);
    P.spill_stride = grid * kBlock;
    P.guided = opt.guided > 0 ? grid * (kBlock / 64u) * (unsigned)opt.guided : 0u;
    P.stats = nullptr;
    if (!ANY && GEN == 0 && STACK_N == 12 && opt.stats) {
        P.stats = (unsigned long long*)((char*)ws + 128);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, st, P);
    hipError_t le = hipGetLastError();
    if (P.stats && le == hipSuccess) {  // diagnostic path: synchronous read-back
        std::lock_guard<std::mutex> lk(g_opt_mu);
        (void)hipMemcpyAsync(g_last_stats, P.stats, sizeof(g_last_stats), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    }
    hipError_t fe = hipFreeAsync(ws, st);
    if (le != hipSuccess) return fail(GIGL_ERR_HIP, std::string("trace kernel launch: ") + hipGetErrorString(le));
    if (fe != hipSuccess) return fail(GIGL_ERR_HIP, std::string("hipFreeAsync: ") + hipGetErrorString(fe));
    return GIGL_OK;
}

template <bool ANY, int GEN>
int launch_trace(const gigl_scene* s, TraceParams& P, hipStream_t st) {
    Options opt;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        opt = g_opt;
    }
    P.nodes = s->d_nodes;
    P.leaf_ids = s->d_leaf_ids;
    P.tris = s->d_tris;
    P.box = s->box;
    P.any_atomic = ANY ? (opt.any_atomic < 0 ? (s->tree.actual_depth > 20 ? 1 : 0) : opt.any_atomic) : 0;
    // ---- ray regrouping for explicit-ray batches (see ray_key_kernel) ----
    void* sort_ws = nullptr;
    P.perm = nullptr;
    if (GEN == 0 && (opt.sort_rays == 1 || (opt.sort_rays < 0 && s->tree.actual_depth > 20 && P.n >= (1u << 18)))) {
        const int bits = opt.sort_bits;
        const unsigned n_bins = 8u << (3 * bits);
        const unsigned n_blocks = n_bins / kBinsPerBlock;
        const size_t off_perm = ((size_t)P.n * 4 + 255) & ~(size_t)255;
        const size_t off_hist = 2 * off_perm;
        const size_t off_sums = off_hist + (size_t)n_bins * 4;
        const size_t total = off_sums + (size_t)n_blocks * 4 + 256;
        if (hipMallocAsync(&sort_ws, total, st) != hipSuccess) return fail(GIGL_ERR_HIP, "hipMallocAsync(regroup)");
        unsigned* keys = (unsigned*)sort_ws;
        unsigned* perm = (unsigned*)((char*)sort_ws + off_perm);
        unsigned* hist = (unsigned*)((char*)sort_ws + off_hist);
        unsigned* sums = (unsigned*)((char*)sort_ws + off_sums);
        hipError_t e = hipMemsetAsync(hist, 0, (size_t)n_bins * 4, st);
        if (e == hipSuccess) {
            const unsigned g = stream_grid(s, P.n);
            hipLaunchKernelGGL(ray_key_kernel, dim3(g), dim3(256), 0, st, P.o_tmin, P.d_tmax, P.n, s->box, bits, keys,
                               hist);
            hipLaunchKernelGGL(bin_block_sums_kernel, dim3(n_blocks), dim3(256), 0, st, hist, sums);
            hipLaunchKernelGGL(bin_scan_kernel, dim3(n_blocks), dim3(256), 0, st, hist, sums);
            hipLaunchKernelGGL(ray_scatter_kernel, dim3(g), dim3(256), 0, st, keys, P.n, hist, perm);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            (void)hipFreeAsync(sort_ws, st);
            return fail(GIGL_ERR_HIP, std::string("ray regrouping: ") + hipGetErrorString(e));
        }
        P.perm = perm;
        if (ANY) P.any_atomic = 1;  // bits of regrouped rays are not consecutive: OR them in
    }
    P.refill_thresh = (ANY && !P.any_atomic) ? 64 : opt.refill_thresh;
    if (P.refill_thresh < 1) P.refill_thresh = 1;
    if (P.refill_thresh > 64) P.refill_thresh = 64;
    P.chunk = (unsigned)opt.chunk;
    P.t_exit = opt.t_exit;
    int rc = GIGL_OK;
    if (ANY && P.any_atomic)
        if (hipMemsetAsync(P.bits, 0, ((size_t)P.n + 63) / 64 * 8, st) != hipSuccess)
            rc = fail(GIGL_ERR_HIP, "hipMemsetAsync(bits)");
    // smallest LDS stack that holds the deepest possible stack avoids the spill path entirely
    int sn = opt.stack_lds;
    const int deepest = s->tree.actual_depth - 1;
    if (deepest <= 8) sn = 8;
    else if (deepest <= 12 && sn >= 12) sn = 12;
    else if (deepest <= 16 && sn >= 16) sn = 16;
    if (rc == GIGL_OK) switch (sn) {
        case 8: rc = launch_trace_t<ANY, GEN, 8>(s, P, opt, st); break;
        case 12: rc = launch_trace_t<ANY, GEN, 12>(s, P, opt, st); break;
        case 16: rc = launch_trace_t<ANY, GEN, 16>(s, P, opt, st); break;
        case 32: rc = launch_trace_t<ANY, GEN, 32>(s, P, opt, st); break;
        default: rc = fail(GIGL_ERR_INVALID, "option stack_lds must be 8, 12, 16 or 32");
    }
    if (sort_ws && hipFreeAsync(sort_ws, st) != hipSuccess && rc == GIGL_OK) rc = fail(GIGL_ERR_HIP, "hipFreeAsync");
    return rc;
}

int check_batch(const gigl_scene* s, const void* a, const void* b, uint64_t n, const void* out) {
    if (!s) return fail(GIGL_ERR_INVALID, "scene is NULL");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (n == 0) return GIGL_OK;
    if (!a || !b || !out) return fail(GIGL_ERR_INVALID, "NULL ray or output pointer");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 rays per call");
    return GIGL_OK;
}


}  // namespace

namespace {

// Re-encodes one host node for the device (see trace_kernel): leaves get a kind field and 2-prim
// leaves carry their ids inline; interior nodes get the index of their children PAIR.
inline gigl::KdNode device_leaf(const gigl::KdTreeHost& T, const gigl::KdNode& h) {
    gigl::KdNode nd = h;
    const uint32_t np = h.w1 >> 1;
    if (np == 0) { nd.w0 = 0; nd.w1 = kLeafBit; }
    else if (np == 1) nd.w1 = kLeafBit | (1u << 29);
    else if (np == 2) {
        nd.w0 = T.leaf_prim_ids[h.w0];
        nd.w1 = kLeafBit | (2u << 29) | T.leaf_prim_ids[h.w0 + 1];
    } else nd.w1 = kLeafBit | (3u << 29) | np;
    return nd;
}

// Device node order.  The reference numbers nodes depth-first (bottom child = parent + 1,
// KdTree.hpp:254-262), which makes every step to a top child a jump to a far cache line.  Here the
// two children of a node sit side by side (one 16-byte pair; the parent stores the pair index) and
// pairs are laid out in treelets: a breadth-first walk of `levels` levels of pairs below a treelet
// root (1 + 2 + 4 = 7 pairs = 112 B for 3 levels) is stored contiguously and never straddles a
// 128-byte line, so up to three consecutive steps of a ray read one line.  Topology, leaf contents
// and traversal order are untouched; only addresses change.  Slot 0 is the root, slot 1 padding.
void flatten_nodes(const gigl::KdTreeHost& T, int levels, std::vector<gigl::KdNode>& out) {
    out.clear();
    const std::vector<gigl::KdNode>& H = T.nodes;
    if (H.empty()) {  // empty scene: a single empty leaf (KdTree.hpp:9-12)
        out.push_back(gigl::KdNode{0u, kLeafBit});
        out.push_back(gigl::KdNode{0u, kLeafBit});
        return;
    }
    out.reserve(H.size() + H.size() / 4 + 16);
    out.resize(2, gigl::KdNode{0u, kLeafBit});
    const unsigned max_pairs = (1u << levels) - 1u;           // pairs in a full treelet
    const unsigned line_pairs = 8;                            // 128 B / 16 B
    struct Item { uint32_t host; uint32_t slot; };            // interior host node, its device slot
    std::vector<Item> roots, cur, nxt;
    if (H[0].w1 & 1u) { out[0] = device_leaf(T, H[0]); return; }
    out[0] = gigl::KdNode{H[0].w0, (H[0].w1 >> 1) & 3u};
    roots.push_back(Item{0u, 0u});
    while (!roots.empty()) {
        const Item r = roots.back();
        roots.pop_back();
        // first pass: how many pairs will this treelet hold?
        unsigned n_pairs = 0;
        cur.assign(1, r);
        for (int lv = 0; lv < levels && !cur.empty(); ++lv) {
            nxt.clear();
            for (const Item& it : cur) {
                ++n_pairs;
                const uint32_t kids[2] = {it.host + 1u, H[it.host].w1 >> 3};
                for (uint32_t k : kids)
                    if (!(H[k].w1 & 1u)) nxt.push_back(Item{k, 0u});
            }
            cur.swap(nxt);
        }
        size_t pos = out.size() / 2;                           // in pairs
        if (max_pairs <= line_pairs && (pos % line_pairs) + n_pairs > line_pairs)
            pos = (pos + line_pairs - 1) / line_pairs * line_pairs;
        out.resize(2 * pos, gigl::KdNode{0u, kLeafBit});
        // second pass: emit breadth-first
        cur.assign(1, r);
        std::vector<Item> frontier;
        for (int lv = 0; lv < levels && !cur.empty(); ++lv) {
            nxt.clear();
            for (const Item& it : cur) {
                const uint32_t pair = (uint32_t)(out.size() / 2);
                out[it.slot].w1 = (out[it.slot].w1 & 3u) | ((uint32_t)(2 * pair) << 2);
                const uint32_t kids[2] = {it.host + 1u, H[it.host].w1 >> 3};
                for (uint32_t k : kids) {
                    const uint32_t slot = (uint32_t)out.size();
                    if (H[k].w1 & 1u) out.push_back(device_leaf(T, H[k]));
                    else {
                        out.push_back(gigl::KdNode{H[k].w0, (H[k].w1 >> 1) & 3u});
                        (lv + 1 < levels ? nxt : frontier).push_back(Item{k, slot});
                    }
                }
            }
            cur.swap(nxt);
        }
        // deeper treelets of this subtree follow it closely (depth-first over treelets)
        for (size_t i = frontier.size(); i-- > 0;) roots.push_back(frontier[i]);
    }
}

}  // namespace

extern "C" {

const char* gigl_last_error(void) { return g_err.c_str(); }

int gigl_device_count(int* count) {
    if (!count) return fail(GIGL_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(GIGL_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return GIGL_OK;
}

int gigl_debug_last_stats(uint64_t out[12]) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    for (int i = 0; i < 12; ++i) out[i] = g_last_stats[i];
    return GIGL_OK;
}

int gigl_set_option(const char* name, int value) {
    if (!name) return fail(GIGL_ERR_INVALID, "name is NULL");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    const std::string k(name);
    if (k == "stack_lds") {
        if (value != 8 && value != 12 && value != 16 && value != 32)
            return fail(GIGL_ERR_INVALID, "stack_lds must be 8, 12, 16 or 32");
        g_opt.stack_lds = value;
    } else if (k == "refill_thresh") {
        if (value < 1 || value > 64) return fail(GIGL_ERR_INVALID, "refill_thresh must be in 1..64");
        g_opt.refill_thresh = value;
    } else if (k == "any_atomic") {
        if (value < -1 || value > 1) return fail(GIGL_ERR_INVALID, "any_atomic must be -1, 0 or 1");
        g_opt.any_atomic = value;
    } else if (k == "sort_rays") {
        if (value < -1 || value > 1) return fail(GIGL_ERR_INVALID, "sort_rays must be -1, 0 or 1");
        g_opt.sort_rays = value;
    } else if (k == "treelet_levels") {
        if (value < 1 || value > 5) return fail(GIGL_ERR_INVALID, "treelet_levels must be in 1..5");
        g_opt.treelet_levels = value;
    } else if (k == "sort_bits") {
        if (value < 4 || value > 6) return fail(GIGL_ERR_INVALID, "sort_bits must be in 4..6");
        g_opt.sort_bits = value;
    } else if (k == "stats") {
        g_opt.stats = value ? 1 : 0;
    } else if (k == "t_exit") {
        if (value < 1 || value > 64) return fail(GIGL_ERR_INVALID, "t_exit must be in 1..64");
        g_opt.t_exit = value;
    } else if (k == "chunk") {
        if (value < 64 || value > 65536 || value % 64) return fail(GIGL_ERR_INVALID, "chunk must be a multiple of 64 in 64..65536");
        g_opt.chunk = value;
    } else if (k == "guided") {
        if (value < 0 || value > 64) return fail(GIGL_ERR_INVALID, "guided must be in 0..64");
        g_opt.guided = value;
    } else if (k == "blocks_per_cu") {
        if (value < 0 || value > 8) return fail(GIGL_ERR_INVALID, "blocks_per_cu must be in 0..8");
        g_opt.blocks_per_cu = value;
    } else {
        return fail(GIGL_ERR_INVALID, "unknown option " + k);
    }
    return GIGL_OK;
}

int gigl_scene_create(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                      int inters_cost, int trav_cost, int max_depth, uint32_t min_num_prims, float empty_bonus,
                      int device, gigl_scene** out) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if ((!verts && n_verts) || (!tris && n_tris)) return fail(GIGL_ERR_INVALID, "NULL vertex or triangle array");
    const bool host_only = device == GIGL_DEVICE_NONE;
    if (!host_only) {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
            return fail(GIGL_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
        if (device < 0 || device >= n_dev) return fail(GIGL_ERR_INVALID, "device index out of range");
    }

    int treelet_levels;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        treelet_levels = g_opt.treelet_levels;
    }
    gigl_scene* s = new gigl_scene();
    s->device = device;
    s->n_tris = n_tris;
    gigl::KdBuildParams bp;
    bp.inters_cost = inters_cost; bp.trav_cost = trav_cost; bp.max_depth = max_depth;
    bp.min_num_prims = min_num_prims; bp.empty_bonus = empty_bonus;
    const char* err = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!gigl::kd_build(verts, n_verts, tris, n_tris, bp, s->tree, &err)) {
        delete s;
        return fail(GIGL_ERR_BUILD, err ? err : "k-d tree build failed");
    }
    const auto t1 = std::chrono::steady_clock::now();
    {
        gigl_tree_stats& S = s->stats;
        S.n_nodes = (uint32_t)s->tree.nodes.size();
        S.n_leaf_prim_ids = (uint32_t)s->tree.leaf_prim_ids.size();
        S.leaf_count = s->tree.leaf_count;
        S.max_depth = s->tree.max_depth;
        S.actual_depth = s->tree.actual_depth;
        S.avg_prims_per_leaf = s->tree.leaf_count ? (float)(s->tree.prims_in_leaves / s->tree.leaf_count) : 0.0f;
        S.build_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        S.degenerate_split_guard_hit = s->tree.degenerate_split_guard_hit ? 1u : 0u;
        S.device_bytes = 0;
    }
    if (host_only) {  // tree only: bbox/stats/export work, every trace call reports GIGL_ERR_NO_DEVICE
        *out = s;
        return GIGL_OK;
    }

    // flattened triangle records: p0, e0 = p1 - p0, e1 = p2 - p0 (RTBase.cpp:151-155)
    std::vector<float> rec((size_t)n_tris * 4 * kTriStride, 0.0f);
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float* p0 = verts + 3 * (size_t)tris[4 * (size_t)i];
        const float* p1 = verts + 3 * (size_t)tris[4 * (size_t)i + 1];
        const float* p2 = verts + 3 * (size_t)tris[4 * (size_t)i + 2];
        float* r = &rec[(size_t)i * 4 * kTriStride];
        r[0] = p0[0]; r[1] = p0[1]; r[2] = p0[2];
        r[3] = p1[0] - p0[0]; r[4] = p1[1] - p0[1]; r[5] = p1[2] - p0[2];
        r[6] = p2[0] - p0[0]; r[7] = p2[1] - p0[1]; r[8] = p2[2] - p0[2];
        uint32_t mat = tris[4 * (size_t)i + 3];
        memcpy(&r[9], &mat, 4);
    }
    DeviceGuard g(device);
    if (!g.ok) { delete s; return fail(GIGL_ERR_NO_DEVICE, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete s; return fail(GIGL_ERR_HIP, "hipGetDeviceProperties failed"); }
    s->n_cus = prop.multiProcessorCount;
    // keep the stream-ordered pool's memory between calls
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
        uint64_t thr = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    }
    // ---- device node array: same topology, cache-line treelet order (see flatten_nodes) ----
    std::vector<gigl::KdNode> dn;
    flatten_nodes(s->tree, treelet_levels, dn);
    // device node word 1 keeps 29 bits for a triangle id / leaf size and 29 for a node slot
    if (n_tris > kLeafPayload || dn.size() >= (1u << 29) || s->tree.leaf_prim_ids.size() > kLeafPayload) {
        delete s;
        return fail(GIGL_ERR_TOO_LARGE, "scene exceeds the device node encoding (2^29 triangles / node slots)");
    }
    const size_t nb_nodes = std::max<size_t>(dn.size(), 2) * sizeof(uint2);
    const size_t nb_leaf = std::max<size_t>(s->tree.leaf_prim_ids.size(), 1) * 4;
    const size_t nb_tris = std::max<size_t>(rec.size(), 16) * 4;
    auto cleanup = [&]() {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_leaf_ids) (void)hipFree(s->d_leaf_ids);
        if (s->d_tris) (void)hipFree(s->d_tris);
        if (s->d_one) (void)hipFree(s->d_one);
        delete s;
    };
#define CREATE_TRY(expr)                                                                      \
    do {                                                                                      \
        hipError_t e2_ = (expr);                                                              \
        if (e2_ != hipSuccess) {                                                              \
            cleanup();                                                                        \
            return fail(GIGL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e2_));    \
        }                                                                                     \
    } while (0)
    CREATE_TRY(hipMalloc((void**)&s->d_nodes, nb_nodes));
    CREATE_TRY(hipMalloc((void**)&s->d_leaf_ids, nb_leaf));
    CREATE_TRY(hipMalloc((void**)&s->d_tris, nb_tris));
    CREATE_TRY(hipMalloc((void**)&s->d_one, 64));
    CREATE_TRY(hipMemcpy(s->d_nodes, dn.data(), dn.size() * 8, hipMemcpyHostToDevice));
    if (!s->tree.leaf_prim_ids.empty())
        CREATE_TRY(hipMemcpy(s->d_leaf_ids, s->tree.leaf_prim_ids.data(), s->tree.leaf_prim_ids.size() * 4,
                             hipMemcpyHostToDevice));
    if (!rec.empty()) CREATE_TRY(hipMemcpy(s->d_tris, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
#undef CREATE_TRY
    s->box = BoxF{s->tree.box_min[0], s->tree.box_min[1], s->tree.box_min[2],
                  s->tree.box_max[0], s->tree.box_max[1], s->tree.box_max[2]};
    s->stats.device_bytes = nb_nodes + nb_leaf + nb_tris;
    *out = s;
    return GIGL_OK;
}

int gigl_scene_destroy(gigl_scene* s) {
    if (!s) return GIGL_OK;
    if (s->device == GIGL_DEVICE_NONE) { delete s; return GIGL_OK; }
    DeviceGuard g(s->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_leaf_ids);
    (void)hipFree(s->d_tris);
    (void)hipFree(s->d_one);
    delete s;
    return GIGL_OK;
}

int gigl_scene_bbox(const gigl_scene* s, float out[6]) {
    if (!s || !out) return fail(GIGL_ERR_INVALID, "NULL argument");
    for (int a = 0; a < 3; ++a) { out[a] = s->tree.box_min[a]; out[3 + a] = s->tree.box_max[a]; }
    return GIGL_OK;
}

int gigl_scene_stats(const gigl_scene* s, gigl_tree_stats* out) {
    if (!s || !out) return fail(GIGL_ERR_INVALID, "NULL argument");
    *out = s->stats;
    return GIGL_OK;
}

int gigl_scene_export_tree(const gigl_scene* s, uint32_t* nodes_words, uint32_t* leaf_prim_ids) {
    if (!s) return fail(GIGL_ERR_INVALID, "scene is NULL");
    if (nodes_words && !s->tree.nodes.empty()) memcpy(nodes_words, s->tree.nodes.data(), s->tree.nodes.size() * 8);
    if (leaf_prim_ids && !s->tree.leaf_prim_ids.empty())
        memcpy(leaf_prim_ids, s->tree.leaf_prim_ids.data(), s->tree.leaf_prim_ids.size() * 4);
    return GIGL_OK;
}

int gigl_trace_closest_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                              gigl_hit* hits, void* stream) {
    int rc = check_batch(s, o_tmin, d_tmax, n, hits);
    if (rc != GIGL_OK || n == 0) return rc;
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    P.o_tmin = (const float4*)o_tmin;
    P.d_tmax = (const float4*)d_tmax;
    P.hits = (uint4*)hits;
    return launch_trace<false, 0>(s, P, (hipStream_t)stream);
}

int gigl_trace_any_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                          uint64_t* occluded_bits, void* stream) {
    int rc = check_batch(s, o_tmin, d_tmax, n, occluded_bits);
    if (rc != GIGL_OK || n == 0) return rc;
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    P.o_tmin = (const float4*)o_tmin;
    P.d_tmax = (const float4*)d_tmax;
    P.bits = (unsigned long long*)occluded_bits;
    return launch_trace<true, 0>(s, P, (hipStream_t)stream);
}

static int trace_host(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n, void* out,
                      size_t out_bytes, bool any) {
    int rc = check_batch(s, o_tmin, d_tmax, n, out);
    if (rc != GIGL_OK || n == 0) return rc;
    DeviceGuard g(s->device);
    void *d_o = nullptr, *d_d = nullptr, *d_out = nullptr;
    auto release = [&]() {
        if (d_o) (void)hipFree(d_o);
        if (d_d) (void)hipFree(d_d);
        if (d_out) (void)hipFree(d_out);
    };
#define HOST_TRY(expr)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            release();                                                                    \
            return fail(GIGL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                 \
    } while (0)
    HOST_TRY(hipMalloc(&d_o, n * 16));
    HOST_TRY(hipMalloc(&d_d, n * 16));
    HOST_TRY(hipMalloc(&d_out, out_bytes));
    HOST_TRY(hipMemcpy(d_o, o_tmin, n * 16, hipMemcpyHostToDevice));
    HOST_TRY(hipMemcpy(d_d, d_tmax, n * 16, hipMemcpyHostToDevice));
    rc = any ? gigl_trace_any_device(s, d_o, d_d, n, (uint64_t*)d_out, nullptr)
             : gigl_trace_closest_device(s, d_o, d_d, n, (gigl_hit*)d_out, nullptr);
    if (rc != GIGL_OK) { release(); return rc; }
    HOST_TRY(hipDeviceSynchronize());
    HOST_TRY(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
#undef HOST_TRY
    release();
    return GIGL_OK;
}

int gigl_trace_closest(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n, gigl_hit* hits) {
    return trace_host(s, o_tmin, d_tmax, n, hits, n * sizeof(gigl_hit), false);
}

int gigl_trace_any(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n,
                   uint64_t* occluded_bits) {
    return trace_host(s, o_tmin, d_tmax, n, occluded_bits, (n + 63) / 64 * 8, true);
}

int gigl_trace_one(const gigl_scene* s, const float o[3], const float d[3], float t_min, float t_max,
                   int is_vis_ray, int* hit_out, gigl_hit* hit_rec, float normal_out[3], float range_out[2]) {
    if (!s || !o || !d || !hit_out) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    // Scene::trace semantics on one ray: a closest-hit launch answers both modes (the bool of a
    // visibility ray equals the bool of the closest-hit query; only the early-out differs).
    const float ro[4] = {o[0], o[1], o[2], t_min}, rd[4] = {d[0], d[1], d[2], t_max};
    gigl_hit h;
    std::lock_guard<std::mutex> lk(const_cast<gigl_scene*>(s)->one_mu);
    DeviceGuard g(s->device);
    HIP_TRY(hipMemcpy(s->d_one, ro, 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_one + 1, rd, 16, hipMemcpyHostToDevice));
    TraceParams P = {};
    P.n = 1;
    P.o_tmin = s->d_one;
    P.d_tmax = s->d_one + 1;
    P.hits = (uint4*)(s->d_one + 2);
    P.ranges = (float2*)(s->d_one + 3);
    int rc = launch_trace<false, 0>(s, P, nullptr);
    if (rc != GIGL_OK) return rc;
    float back[8];
    HIP_TRY(hipMemcpy(back, s->d_one + 2, 32, hipMemcpyDeviceToHost));
    memcpy(&h, back, 16);
    if (range_out) { range_out[0] = back[4]; range_out[1] = back[5]; }
    *hit_out = h.tri_id != GIGL_MISS;
    if (hit_rec) *hit_rec = h;
    if (normal_out) {
        normal_out[0] = normal_out[1] = normal_out[2] = 0.0f;
        if (*hit_out) {
            // cross(edge0, edge1) (RTBase.cpp:170); normalized for non-visibility rays (Scene.cpp:238)
            const gigl::KdTreeHost& T = s->tree;
            (void)T;
            float rec[12];
            HIP_TRY(hipMemcpy(rec, s->d_tris + kTriStride * (size_t)h.tri_id, 48, hipMemcpyDeviceToHost));
            const float e0x = rec[3], e0y = rec[4], e0z = rec[5], e1x = rec[6], e1y = rec[7], e1z = rec[8];
            float nx = e0y * e1z - e1y * e0z, ny = e0z * e1x - e1z * e0x, nz = e0x * e1y - e1x * e0y;
            if (!is_vis_ray) {
                const float sqr = nx * nx + ny * ny + nz * nz;
                const float inv = 1.0f / sqrtf(sqr);
                nx = nx * inv; ny = ny * inv; nz = nz * inv;
            }
            normal_out[0] = nx; normal_out[1] = ny; normal_out[2] = nz;
        }
    }
    return GIGL_OK;
}

int gigl_camera_setup(const float pos[3], const float dir[3], const float up[3], float h_fov, int res_x,
                      int res_y, float cam_out[12]) {
    if (!pos || !dir || !up || !cam_out || res_x <= 0 || res_y <= 0) return fail(GIGL_ERR_INVALID, "bad camera");
    // Camera.cpp:15-29, GLM operation order
    auto norm = [](float v[3]) {
        const float sqr = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const float inv = 1.0f / sqrtf(sqr);
        v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
    };
    auto cross = [](const float x[3], const float y[3], float o[3]) {
        o[0] = x[1] * y[2] - y[1] * x[2]; o[1] = x[2] * y[0] - y[2] * x[0]; o[2] = x[0] * y[1] - y[0] * x[1];
    };
    const float aspect = (float)res_x / (float)res_y;
    float fwd[3] = {dir[0], dir[1], dir[2]}, right[3], upa[3];
    norm(fwd);
    cross(fwd, up, right); norm(right);
    cross(right, fwd, upa); norm(upa);
    const float k = 2.0f * tanf(0.5f * h_fov);
    for (int a = 0; a < 3; ++a) {
        const float row = (k * right[a]) * aspect, col = k * upa[a];
        cam_out[a] = pos[a];
        cam_out[3 + a] = (fwd[a] - 0.5f * row) - 0.5f * col;
        cam_out[6 + a] = row / (float)res_x;
        cam_out[9 + a] = col / (float)res_y;
    }
    return GIGL_OK;
}

int gigl_gen_primary_rays_device(const gigl_scene* s, const float cam[12], int res_x, uint64_t first, uint64_t n,
                                 void* o_tmin, void* d_tmax, void* stream) {
    if (!s || !cam || res_x <= 0) return fail(GIGL_ERR_INVALID, "bad argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (n == 0) return GIGL_OK;
    if (!o_tmin || !d_tmax) return fail(GIGL_ERR_INVALID, "NULL output");
    if (n > kMaxRays || first + n > 0xFFFFFFFFull) return fail(GIGL_ERR_TOO_LARGE, "pixel range too large");
    DeviceGuard g(s->device);
    Cam12 c;
    memcpy(c.c, cam, sizeof(c.c));
    hipLaunchKernelGGL(primary_rays_kernel, dim3(stream_grid(s, n)), dim3(256), 0, (hipStream_t)stream, c,
                       (unsigned)res_x, (unsigned)first, (unsigned)n, (float4*)o_tmin, (float4*)d_tmax);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_trace_primary_device(const gigl_scene* s, const float cam[12], int res_x, uint64_t first, uint64_t n,
                              gigl_hit* hits, void* stream) {
    if (!s || !cam || res_x <= 0) return fail(GIGL_ERR_INVALID, "bad argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (n == 0) return GIGL_OK;
    if (!hits) return fail(GIGL_ERR_INVALID, "NULL output");
    if (n > kMaxRays || first + n > 0xFFFFFFFFull) return fail(GIGL_ERR_TOO_LARGE, "pixel range too large");
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    memcpy(P.cam, cam, sizeof(P.cam));
    P.res_x = (unsigned)res_x;
    P.first = (unsigned)first;
    P.hits = (uint4*)hits;
    return launch_trace<false, 1>(s, P, (hipStream_t)stream);
}

int gigl_shadow_batch_device(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                             const gigl_hit* prim_hits, uint64_t n_pts, const float* lights, uint32_t n_lights,
                             uint64_t* occluded_bits, const gigl_fog* fog, float* transm, void* stream) {
    if (!s) return fail(GIGL_ERR_INVALID, "scene is NULL");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    const uint64_t n = n_pts * (uint64_t)n_lights;
    if (n == 0) return GIGL_OK;
    if (!prim_o_tmin || !prim_d_tmax || !prim_hits || !lights) return fail(GIGL_ERR_INVALID, "NULL input");
    if (!occluded_bits && !transm) return fail(GIGL_ERR_INVALID, "no output requested");
    if (transm && !fog) return fail(GIGL_ERR_INVALID, "transm requested without fog");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "n_pts * n_lights must stay below 0xF0000000");
    if (fog && fog->device != s->device) return fail(GIGL_ERR_INVALID, "fog and scene live on different devices");
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    P.prim_o = (const float4*)prim_o_tmin;
    P.prim_d = (const float4*)prim_d_tmax;
    P.prim_hits = (const uint4*)prim_hits;
    P.lights = lights;
    P.n_lights = n_lights;
    P.bits = (unsigned long long*)occluded_bits;
    if (occluded_bits) {
        int rc = launch_trace<true, 2>(s, P, (hipStream_t)stream);
        if (rc != GIGL_OK) return rc;
    }
    if (transm) {
        P.tris = s->d_tris;
        hipLaunchKernelGGL(transm_shadow_kernel, dim3(stream_grid(s, n)), dim3(256), 0, (hipStream_t)stream, fog->fp,
                           P, transm);
        HIP_TRY(hipGetLastError());
    }
    return GIGL_OK;
}

int gigl_shade_surface_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                              const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                              uint64_t n_pts, const float* materials, uint32_t n_materials, const float* ppls,
                              const void* vpls, const uint64_t* occluded_bits, const float* transm,
                              const float* pi_dens, float* accum_rgba, float* fog_dist, void* stream) {
    if (!s || !prm) return fail(GIGL_ERR_INVALID, "scene or params is NULL");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to shade on");
    if (n_pts == 0) return GIGL_OK;
    if (!prim_o_tmin || !prim_d_tmax || !prim_hits || !accum_rgba) return fail(GIGL_ERR_INVALID, "NULL input");
    if (prm->n_ppls < 0 || prm->n_vpls < 0) return fail(GIGL_ERR_INVALID, "negative light count");
    const uint64_t n_lights = (uint64_t)prm->n_ppls + (uint64_t)prm->n_vpls;
    if (n_lights > 0 && (!occluded_bits || !materials || n_materials == 0))
        return fail(GIGL_ERR_INVALID, "lights given without occlusion bits or materials");
    if ((prm->n_ppls > 0 && !ppls) || (prm->n_vpls > 0 && !vpls)) return fail(GIGL_ERR_INVALID, "NULL light array");
    if (n_pts > kMaxRays || n_pts * (n_lights ? n_lights : 1) > kMaxRays)
        return fail(GIGL_ERR_TOO_LARGE, "n_pts * n_lights must stay below 0xF0000000");
    if (prm->res_x <= 0 || prm->res_y <= 0 || prm->first + n_pts > (uint64_t)prm->res_x * (uint64_t)prm->res_y)
        return fail(GIGL_ERR_INVALID, "pixel range outside the image");
    if (fog && fog->device != s->device) return fail(GIGL_ERR_INVALID, "fog and scene live on different devices");
    if (fog && fog->fp.ext_k > 0.0f && (!pi_dens || prm->pi_res_z <= 0))
        return fail(GIGL_ERR_INVALID, "fog given without a preintegrated density volume");
    if (transm && !fog) return fail(GIGL_ERR_INVALID, "transm given without fog");
    DeviceGuard g(s->device);
    FogParams F = {};
    if (fog) F = fog->fp;
    ShadeParams S = {};
    for (int a = 0; a < 3; ++a) S.cam[a] = prm->cam_pos[a];
    S.res_x = (unsigned)prm->res_x; S.res_y = (unsigned)prm->res_y; S.first = (unsigned)prm->first;
    S.n_pts = (unsigned)n_pts; S.n_ppls = (unsigned)prm->n_ppls; S.n_vpls = (unsigned)prm->n_vpls;
    S.gi_enabled = prm->gi_enabled; S.clamp_rsq = prm->clamp_rsq; S.frame_id = prm->frame_id;
    S.pi_res_z = prm->pi_res_z; S.sca_albedo = prm->sca_albedo;
    S.prim_o = (const float4*)prim_o_tmin; S.prim_d = (const float4*)prim_d_tmax; S.hits = (const uint4*)prim_hits;
    S.tris = s->d_tris; S.materials = materials; S.ppls = ppls; S.vpls = (const float*)vpls;
    S.bits = (const unsigned long long*)occluded_bits; S.transm = transm; S.pi_dens = pi_dens;
    S.accum = (float4*)accum_rgba; S.fog_dist = (float2*)fog_dist;
    hipLaunchKernelGGL(shade_surface_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, S);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

namespace {
int volume_params(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm, VolumeParams& V) {
    if (!s || !fog || !prm) return fail(GIGL_ERR_INVALID, "scene, fog or params is NULL");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to shade on");
    if (fog->device != s->device) return fail(GIGL_ERR_INVALID, "fog and scene live on different devices");
    if (prm->res_x < 2 || prm->res_y < 2 || (prm->res_x & 1) || (prm->res_y & 1))
        return fail(GIGL_ERR_INVALID, "the volume pass needs an even resolution");
    if ((uint64_t)prm->res_x * (uint64_t)prm->res_y > (1ull << 28)) return fail(GIGL_ERR_TOO_LARGE, "at most 2^28 pixels");
    if (prm->n_ppls < 0 || prm->n_vpls < 0 || prm->pi_res_z <= 0) return fail(GIGL_ERR_INVALID, "bad light count or pi_res_z");
    if (prm->frame_id < 0) return fail(GIGL_ERR_INVALID, "negative frame_id");
    V = VolumeParams{};
    for (int a = 0; a < 3; ++a) V.cam[a] = prm->cam_pos[a];
    V.res_x = (unsigned)prm->res_x; V.res_y = (unsigned)prm->res_y;
    V.n_samples = prm->gi_enabled ? 8u : 32u;       // MAX_VOL_SAMP / 4 : MAX_VOL_SAMP, Volume.frag:277
    V.n_ppls = (unsigned)prm->n_ppls; V.n_vpls = (unsigned)prm->n_vpls;
    V.gi_enabled = prm->gi_enabled; V.clamp_rsq = prm->clamp_rsq; V.transm_opt = prm->transm_opt;
    V.frame_id = prm->frame_id; V.pi_res_z = prm->pi_res_z; V.sca_k = prm->sca_k; V.sca_albedo = prm->sca_albedo;
    return GIGL_OK;
}
}  // namespace

int gigl_volume_samples_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                               const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                               const float* fog_dist, const float* rnd_offsets, const float* halton,
                               const float* pi_dens, float* points, void* stream) {
    VolumeParams V;
    int rc = volume_params(s, fog, prm, V);
    if (rc != GIGL_OK) return rc;
    if (!prim_o_tmin || !prim_d_tmax || !prim_hits || !fog_dist || !rnd_offsets || !halton || !pi_dens || !points)
        return fail(GIGL_ERR_INVALID, "NULL pointer");
    DeviceGuard g(s->device);
    V.prim_o = (const float4*)prim_o_tmin; V.prim_d = (const float4*)prim_d_tmax; V.hits = (const uint4*)prim_hits;
    V.fog_dist = (const float2*)fog_dist; V.rnd_offsets = rnd_offsets; V.halton = halton; V.pi_dens = pi_dens;
    V.points = (float4*)points;
    const uint64_t n = (uint64_t)(V.res_x / 2) * (V.res_y / 2) * V.n_samples;
    hipLaunchKernelGGL(volume_samples_kernel, dim3(stream_grid(s, n)), dim3(256), 0, (hipStream_t)stream, fog->fp, V);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_shadow_points_device(const gigl_scene* s, const float* points, uint64_t n_pts, const float* lights,
                              uint32_t n_lights, uint64_t* occluded_bits, void* stream) {
    if (!s) return fail(GIGL_ERR_INVALID, "scene is NULL");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    const uint64_t n = n_pts * (uint64_t)n_lights;
    if (n == 0) return GIGL_OK;
    if (!points || !lights || !occluded_bits) return fail(GIGL_ERR_INVALID, "NULL pointer");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "n_pts * n_lights must stay below 0xF0000000");
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    P.points = (const float4*)points;
    P.lights = lights;
    P.n_lights = n_lights;
    P.bits = (unsigned long long*)occluded_bits;
    return launch_trace<true, 3>(s, P, (hipStream_t)stream);
}

int gigl_shade_volume_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                             const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                             const float* fog_dist, const float* rnd_offsets, const float* halton,
                             const float* pi_dens, const float* ppls, const void* vpls,
                             const uint64_t* occluded_bits, float* vol_comp_rgba, void* stream) {
    VolumeParams V;
    int rc = volume_params(s, fog, prm, V);
    if (rc != GIGL_OK) return rc;
    if (!prim_o_tmin || !prim_d_tmax || !prim_hits || !fog_dist || !rnd_offsets || !halton || !pi_dens || !vol_comp_rgba)
        return fail(GIGL_ERR_INVALID, "NULL pointer");
    if ((V.n_ppls + V.n_vpls) > 0 && !occluded_bits) return fail(GIGL_ERR_INVALID, "lights given without occlusion bits");
    if ((V.n_ppls > 0 && !ppls) || (V.n_vpls > 0 && !vpls)) return fail(GIGL_ERR_INVALID, "NULL light array");
    const uint64_t n_pix = (uint64_t)(V.res_x / 2) * (V.res_y / 2);
    if (n_pix * V.n_samples * (uint64_t)(V.n_ppls + V.n_vpls ? V.n_ppls + V.n_vpls : 1) > kMaxRays)
        return fail(GIGL_ERR_TOO_LARGE, "pixels * samples * lights must stay below 0xF0000000");
    DeviceGuard g(s->device);
    V.prim_o = (const float4*)prim_o_tmin; V.prim_d = (const float4*)prim_d_tmax; V.hits = (const uint4*)prim_hits;
    V.fog_dist = (const float2*)fog_dist; V.rnd_offsets = rnd_offsets; V.halton = halton; V.pi_dens = pi_dens;
    V.ppls = ppls; V.vpls = (const float*)vpls; V.bits = (const unsigned long long*)occluded_bits;
    V.vol_comp = (float4*)vol_comp_rgba;
    hipLaunchKernelGGL(shade_volume_kernel, dim3(stream_grid(s, n_pix)), dim3(256), 0, (hipStream_t)stream, fog->fp, V);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_depth_buffer_device(int device, const float cam_pos[3], const float cam_dir[3], float t_near, float t_far,
                             const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits, uint64_t n,
                             float* depth, void* stream) {
    if (n == 0) return GIGL_OK;
    if (!cam_pos || !cam_dir || !prim_o_tmin || !prim_d_tmax || !prim_hits || !depth) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 pixels per call");
    if (!(t_far > t_near) || !(t_near > 0.0f)) return fail(GIGL_ERR_INVALID, "need 0 < t_near < t_far");
    DeviceGuard g(device);
    if (!g.ok) return fail(GIGL_ERR_NO_DEVICE, "hipSetDevice failed");
    const float sq = cam_dir[0] * cam_dir[0] + cam_dir[1] * cam_dir[1] + cam_dir[2] * cam_dir[2];
    const float inv = 1.0f / sqrtf(sq);                      // glm::normalize, on the host as in Camera.cpp:21
    const F3 fwd = {cam_dir[0] * inv, cam_dir[1] * inv, cam_dir[2] * inv};
    const F3 cam = {cam_pos[0], cam_pos[1], cam_pos[2]};
    hipLaunchKernelGGL(depth_kernel, dim3(stream_grid(nullptr, n)), dim3(256), 0, (hipStream_t)stream, cam, fwd, t_near,
                       t_far, (const float4*)prim_o_tmin, (const float4*)prim_d_tmax, (const uint4*)prim_hits, (unsigned)n,
                       depth);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_combine_device(int device, int res_x, int res_y, const float* depth, const float* vol_comp_rgba, float ext_k,
                        int frame_id, int exposure, float* accum_rgba, float* rgb_out, void* stream) {
    if (res_x < 2 || res_y < 2 || (res_x & 1) || (res_y & 1)) return fail(GIGL_ERR_INVALID, "need an even resolution");
    if ((uint64_t)res_x * (uint64_t)res_y > (1ull << 28)) return fail(GIGL_ERR_TOO_LARGE, "at most 2^28 pixels");
    if (!accum_rgba || !rgb_out) return fail(GIGL_ERR_INVALID, "NULL argument");
    const int add = frame_id < 30 && ext_k > 0.0f;                                            // Combine.frag:65
    if (add && (!depth || !vol_comp_rgba)) return fail(GIGL_ERR_INVALID, "fog on: depth and vol_comp are required");
    DeviceGuard g(device);
    if (!g.ok) return fail(GIGL_ERR_NO_DEVICE, "hipSetDevice failed");
    hipLaunchKernelGGL(combine_kernel, dim3(stream_grid(nullptr, (uint64_t)res_x * res_y)), dim3(256), 0,
                       (hipStream_t)stream, (unsigned)res_x, (unsigned)res_y, depth, (const float4*)vol_comp_rgba, add,
                       frame_id, exposure, (float4*)accum_rgba, rgb_out);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_tonemap_device(int device, const float* accum_rgba, uint64_t n, int frame_id, int exposure, float* rgb_out,
                        void* stream) {
    if (n == 0) return GIGL_OK;
    if (!accum_rgba || !rgb_out) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 pixels per call");
    DeviceGuard g(device);
    if (!g.ok) return fail(GIGL_ERR_NO_DEVICE, "hipSetDevice failed");
    hipLaunchKernelGGL(tonemap_kernel, dim3(stream_grid(nullptr, n)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)accum_rgba, (unsigned)n, frame_id, exposure, rgb_out);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

int gigl_fog_create(const uint8_t* dens, const int res[3], const float box[6], float ext_k, int n_intervals,
                    int device, gigl_fog** out) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!dens || !res || !box) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || (int64_t)res[0] * res[1] * res[2] > (1ll << 30))
        return fail(GIGL_ERR_INVALID, "bad volume resolution");
    if (n_intervals < 1 || n_intervals > 4096) return fail(GIGL_ERR_INVALID, "n_intervals must be in 1..4096");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(GIGL_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(GIGL_ERR_INVALID, "device index out of range");
    DeviceGuard g(device);
    gigl_fog* f = new gigl_fog();
    f->device = device;
    // corner bricks, see FogParams
    const int rx = res[0], ry = res[1], rz = res[2];
    std::vector<uint64_t> bricks((size_t)(rx + 1) * (ry + 1) * (rz + 1));
    auto texel = [&](int x, int y, int z) -> uint64_t {
        if (x < 0 || x >= rx || y < 0 || y >= ry || z < 0 || z >= rz) return 0;
        return dens[x + (size_t)y * rx + (size_t)z * rx * ry];
    };
    for (int bz = 0; bz <= rz; ++bz)
        for (int by = 0; by <= ry; ++by)
            for (int bx = 0; bx <= rx; ++bx) {
                uint64_t w = 0;
                for (int c = 0; c < 8; ++c) w |= texel(bx - 1 + (c & 1), by - 1 + ((c >> 1) & 1), bz - 1 + (c >> 2)) << (8 * c);
                bricks[bx + (size_t)by * (rx + 1) + (size_t)bz * (rx + 1) * (ry + 1)] = w;
            }
    const size_t nb = bricks.size() * 8;
    hipError_t e = hipMalloc((void**)&f->d_bricks, nb);
    if (e == hipSuccess) e = hipMemcpy(f->d_bricks, bricks.data(), nb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (f->d_bricks) (void)hipFree(f->d_bricks);
        delete f;
        return fail(GIGL_ERR_HIP, std::string("fog upload: ") + hipGetErrorString(e));
    }
    f->fp = FogParams{f->d_bricks, res[0], res[1], res[2], box[0], box[1], box[2], box[3], box[4], box[5], ext_k,
                      n_intervals};
    *out = f;
    return GIGL_OK;
}

int gigl_fog_destroy(gigl_fog* f) {
    if (!f) return GIGL_OK;
    DeviceGuard g(f->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(f->d_bricks);
    delete f;
    return GIGL_OK;
}

int gigl_transmittance_device(const gigl_fog* f, const void* seg_p, const void* seg_d_distsq, uint64_t n,
                              float* transm, void* stream) {
    if (!f) return fail(GIGL_ERR_INVALID, "fog is NULL");
    if (n == 0) return GIGL_OK;
    if (!seg_p || !seg_d_distsq || !transm) return fail(GIGL_ERR_INVALID, "NULL pointer");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 segments per call");
    DeviceGuard g(f->device);
    hipLaunchKernelGGL(transm_segments_kernel, dim3(stream_grid(nullptr, n)), dim3(256), 0, (hipStream_t)stream,
                       f->fp, (const float4*)seg_p, (const float4*)seg_d_distsq, (unsigned)n, transm);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

static int map_host(int device, const float* in, size_t in_floats, uint64_t n, float* out,
                    void (*launch)(const float*, unsigned, float*, const void*), const void* ctx) {
    DeviceGuard g(device);
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, in_floats * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_floats * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch(d_in, (unsigned)n, d_out, ctx);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(GIGL_ERR_HIP, std::string("device map: ") + hipGetErrorString(e));
    return GIGL_OK;
}

int gigl_compute_pi_density_device(const gigl_scene* s, const gigl_fog* f, const float cam[12], int res_x,
                                   int res_y, float* pi_dens, void* stream) {
    if (!s || !f || !cam || res_x <= 0 || res_y <= 0) return fail(GIGL_ERR_INVALID, "bad argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (f->device != s->device) return fail(GIGL_ERR_INVALID, "fog and scene live on different devices");
    if (!pi_dens) return fail(GIGL_ERR_INVALID, "NULL output");
    const uint64_t n = (uint64_t)res_x * (uint64_t)res_y;
    if (n > (1ull << 28)) return fail(GIGL_ERR_TOO_LARGE, "at most 2^28 pixels per call");
    DeviceGuard g(s->device);
    hipStream_t st = (hipStream_t)stream;
    void* hits = nullptr;
    HIP_TRY(hipMallocAsync(&hits, n * sizeof(gigl_hit), st));
    int rc = gigl_trace_primary_device(s, cam, res_x, 0, n, (gigl_hit*)hits, stream);
    if (rc == GIGL_OK) {
        Cam12 c;
        memcpy(c.c, cam, sizeof(c.c));
        hipLaunchKernelGGL(pi_density_kernel, dim3(stream_grid(s, n)), dim3(256), 0, st, f->fp, c, (unsigned)res_x,
                           (unsigned)n, (const uint4*)hits, pi_dens);
        if (hipGetLastError() != hipSuccess) rc = fail(GIGL_ERR_HIP, "pi_density_kernel launch failed");
    }
    if (hipFreeAsync(hits, st) != hipSuccess && rc == GIGL_OK) rc = fail(GIGL_ERR_HIP, "hipFreeAsync");
    return rc;
}

int gigl_sample_density(const gigl_fog* f, const float* pos, uint64_t n, float* dens_out) {
    if (!f) return fail(GIGL_ERR_INVALID, "fog is NULL");
    if (n == 0) return GIGL_OK;
    if (!pos || !dens_out) return fail(GIGL_ERR_INVALID, "NULL pointer");
    if (n > (1u << 28)) return fail(GIGL_ERR_TOO_LARGE, "at most 2^28 positions per call");
    return map_host(f->device, pos, 3 * n, n, dens_out,
                    [](const float* in, unsigned m, float* o, const void* ctx) {
                        hipLaunchKernelGGL(sample_density_kernel, dim3((m + 255) / 256), dim3(256), 0, nullptr,
                                           *(const FogParams*)ctx, in, m, o);
                    },
                    &f->fp);
}

int gigl_debug_rescaled_eps(int device, const float* t, uint64_t n, float* out) {
    if (n == 0) return GIGL_OK;
    if (!t || !out) return fail(GIGL_ERR_INVALID, "NULL pointer");
    if (n > (1u << 28)) return fail(GIGL_ERR_TOO_LARGE, "at most 2^28 values per call");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(GIGL_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(GIGL_ERR_INVALID, "device index out of range");
    return map_host(device, t, n, n, out,
                    [](const float* in, unsigned m, float* o, const void*) {
                        hipLaunchKernelGGL(eps_kernel, dim3((m + 255) / 256), dim3(256), 0, nullptr, in, m, o);
                    },
                    nullptr);
}

}  // extern "C"
