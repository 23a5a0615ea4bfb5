// gigl_internal.h — shared by the three translation units of libgigl_rt.so (gigl_rt.hip: traversal,
// gigl_fog.hip: fog volume kernels, gigl_shade.hip: shading stage).  Device helpers are
// __forceinline__, so no device symbol crosses a translation unit.
#pragma once
#include <hip/hip_runtime.h>

// A/B switches kept from tuning (tools/sweep.py over builds, same device, interleaved): the
// branch-free triangle test costs 10 VGPRs, drops the kernel from 6 to 5 waves/SIMD and loses 4-10 %
// on coherent batches; the nontemporal hint on ray loads and hit stores is worth 1.5-3.7 %.
#ifndef GIGL_BRANCHLESS_TRI
#define GIGL_BRANCHLESS_TRI 0
#endif
// Register budget of the traversal kernel: the compiler is told to stay within what 6 waves per SIMD
// allow (80 VGPRs); 0 leaves the allocation to it.
#ifndef GIGL_WAVES_PER_EU
#define GIGL_WAVES_PER_EU 6
#endif
// getRescaledEPS in the traversal kernel: 1 = assembled from the frexp exponent in two instructions for the
// ordinary range, the LDS table only for the wrap-around cases (+0.9 %: the table read sat on the step's
// critical path); 0 = always the table.
#ifndef GIGL_EPS_ALU
#define GIGL_EPS_ALU 1
#endif
// LDS stack entries per lane of the default launches (12 bytes each; deeper entries spill to global memory).
// 8 with 6 waves per SIMD is the shipped pair; 7 / 7 and 6 / 8 are the occupancy experiments of DESIGN.md §4.1.
#ifndef GIGL_LDS_STACK
#define GIGL_LDS_STACK 8
#endif
// 1: explicit-ray closest-hit launches run the kernel in which rays change lanes (trace_body's XCHG); experiment
#ifndef GIGL_XCHG
#define GIGL_XCHG 0
#endif
// 1: the T-phase requests the line of the next records before it decides its two steps (experiment)
#ifndef GIGL_POP_TOUCH
#define GIGL_POP_TOUCH 0   // 1: request the next pop's record when a leaf's triangle tests start (experiment)
#endif
#ifndef GIGL_PHOTON_STAMPS
#define GIGL_PHOTON_STAMPS 0   // 1: photon_small_kernel stamps its phases on the real-time counter (GIGL_PHOTON_TIMING=1 prints them)
#endif
#ifndef GIGL_EARLY_TOUCH
#define GIGL_EARLY_TOUCH 0
#endif
// 1: the ray-march kernels (nine or 257 samples per lane, bound by vector-ALU issue) read byte / 255.0f from a
// 256-entry table in LDS instead of computing it: 152 M Cornell segments 4.16 -> 3.94 ms (+5.6 %)
#ifndef GIGL_FOG_LUT
#define GIGL_FOG_LUT 1
#endif
// the table, filled by a 256-lane workgroup from the same function it replaces
#if GIGL_FOG_LUT
#define GIGL_FOG_LUT_DECL                               \
    __shared__ float s_unorm[256];                      \
    s_unorm[threadIdx.x] = gd::byte_unorm(threadIdx.x); \
    __syncthreads();
#else
#define GIGL_FOG_LUT_DECL const float* s_unorm = nullptr;
#endif
#ifndef GIGL_STREAM_HINT
#define GIGL_STREAM_HINT 1
#endif

#include <atomic>
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

namespace gd {


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

struct Cam12 { float c[12]; };

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
// LUT: `lut` is a 256-entry table of byte_unorm in LDS (kernels that take nine samples per lane and are bound by
// vector-ALU issue: a table read replaces four of the five instructions per texel, exact by construction).
template <bool LUT = false>
__device__ __forceinline__ float fog_filter(const FogParams& F, float nx, float ny, float nz, const float* lut = nullptr) {
    const float tcx = nx * (float)F.rx - 0.5f, tcy = ny * (float)F.ry - 0.5f, tcz = nz * (float)F.rz - 0.5f;
    const float fx = floorf(tcx), fy = floorf(tcy), fz = floorf(tcz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx = tcx - (float)x0, ty = tcy - (float)y0, tz = tcz - (float)z0;
    // The reference's upper texel is ceil(tc), which equals x0 when tc is an integer; the brick's
    // is x0 + 1.  They differ only when the weight t is exactly 0, where (1-0)*a + 0*b = a for
    // either b, so the filtered value is the same bit for bit.
    const unsigned bx = (unsigned)x0 + 1u, by = (unsigned)y0 + 1u, bz = (unsigned)z0 + 1u;
    uint2 w = make_uint2(0u, 0u);   // every texel outside: the lerps still run (inf/NaN weights propagate)
    // 32-bit index: gigl_fog_create keeps the brick count below 2^31
    if ((bx <= (unsigned)F.rx) & (by <= (unsigned)F.ry) & (bz <= (unsigned)F.rz))
        w = F.bricks[bx + (by + bz * (unsigned)(F.ry + 1)) * (unsigned)(F.rx + 1)];
    auto un = [&](unsigned b) { return LUT ? lut[b] : byte_unorm(b); };
    const float d000 = un(w.x & 0xFFu), d100 = un((w.x >> 8) & 0xFFu);
    const float d010 = un((w.x >> 16) & 0xFFu), d110 = un(w.x >> 24);
    const float d001 = un(w.y & 0xFFu), d101 = un((w.y >> 8) & 0xFFu);
    const float d011 = un((w.y >> 16) & 0xFFu), d111 = un(w.y >> 24);
    const float i0 = lerp1(lerp1(d000, d100, tx), lerp1(d010, d110, tx), ty);
    const float i1 = lerp1(lerp1(d001, d101, tx), lerp1(d011, d111, tx), ty);
    return lerp1(i0, i1, tz);
}
// GLSL min/max as selects (Surface.frag:146-153)
__device__ __forceinline__ float smin(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float smax(float x, float y) { return x < y ? y : x; }

// calcTransm + intersectBBox + rayMarch, Surface.frag:139-203
template <bool LUT = false>
__device__ __forceinline__ float calc_transm(const FogParams& F, float px, float py, float pz, float wx, float wy,
                                             float wz, float dist_sq, const float* lut = nullptr) {
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
        return F.ext_k * fog_filter<LUT>(F, (qx - F.mnx) * idx, (qy - F.mny) * idy, (qz - F.mnz) * idz, lut);
    };
    float prev = ext_at(t_min), total = 0.0f;
    for (int i = 1; i <= F.n_intervals; ++i) {
        const float cur = ext_at(t_min + (float)i * dt);
        total += 0.5f * (prev + cur);
        prev = cur;
    }
    return expf(-(total * dt));
}

struct TraceParams {
    const uint2* nodes;
    uint2 root;           // the root node's two words (slot 0 of nodes)
    const uint4* recs;    // triple layout: the same buffer as 16-byte records
    uint4 root_rec;       // triple layout: record 0
    const unsigned* leaf_ids;
    const float4* tris;   // kTriStride float4 per triangle: (p0.xyz,e0.x) (e0.yz,e1.xy) (e1.z,id,mat,-) (-)
    unsigned blocks_off8;   // leaf triangle blocks (kd_layout.h): where their lines start behind the table, in 8-byte units
    BoxF box;
    unsigned n;
    unsigned* counter;
    uint4* spill;         // (t_split, w0, w1, -) entries beyond the LDS stack
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
    const float4* pt_cache;  // optional: per point (p, valid), (shadow-ray origin, -) from shadow_points_kernel
    unsigned n_pts;          // light-major order: work item i is pair (i % n_pts) * n_lights + i / n_pts; 0: pair order
    // GEN 3: (point, light) pairs from explicit points (x, y, z, valid)
    const float4* points;
    // outputs
    uint4* hits;
    unsigned long long* bits;
    unsigned long long* stats;   // STATS build only: 12 counters
    float2* ranges;              // optional: the ray's (t_min, t_max) as KdTree::intersect leaves them
    float4* hit_tri;             // optional, with ranges: the first 48 bytes of the hit triangle's record per ray
    unsigned reset_counter;      // kept workspaces: the last wave out zeroes the counter words (counter[0] work, counter[1] waves out)
    const unsigned* perm;        // optional (GEN 0): work item i traces ray perm[i] (ray regrouping)
    int xchg_min;                // XCHG kernels: a wave runs a phase for at least this many lanes (or for the larger group)
    unsigned xchg_every;         // XCHG kernels: the waves of a workgroup exchange rays every this many rounds (>= 1)
};

// ---- 3-vectors for the shading stage (operation order as the oracle's V3) ------------------
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

// ---- host side ------------------------------------------------------------------------------
int fail(int code, const std::string& msg);   // sets gigl_last_error(); defined in gigl_rt.hip
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? GIGL_ERR_NO_DEVICE \
                                                                              : GIGL_ERR_HIP,      \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

constexpr int kBlock = 256;
constexpr unsigned kMaxRays = 0xF0000000u;

}  // namespace gd

struct gigl_scene {
    int device = 0;
    int n_cus = 0;
    gigl::KdTreeHost tree;
    uint32_t n_tris = 0;
    uint2* d_nodes = nullptr;
    uint2 root = {0u, 0x80000000u};
    uint4 root_rec = {0u, 0u, 0u, 3u};
    bool triples = false;          // d_nodes holds the triple layout (trace_kernel<TRIPLES = true>)
    unsigned* d_leaf_ids = nullptr;
    float4* d_tris = nullptr;
    unsigned blocks_off8 = 0;      // leaf triangle blocks (kd_layout.h) start here in d_tris, in 8-byte units (0: none)
    size_t nb_nodes = 0, nb_leaf = 0, nb_tris = 0;   // bytes of the three device arrays
    bool replica = false;          // a copy made by scene_clone_to_device / adopted from a device blob: no host tree
    bool counted = false;          // counted among the device's scenes (the last one to go trims the library's pool)
    gd::BoxF box;
    gigl_tree_stats stats;
    int layout = 2;                // device layout in use (kd_layout.h): 0, 1 or 2 (2 may fall back to 1 at create)
    void* scratch = nullptr;       // this device's copy of the no-allocation entry point's static scratch array
    // gigl_trace_one: pinned, device-mapped staging (7 float4: ray in, hit record, range and hit
    // triangle out), a persistent workspace for its single-wave launch and a stream of its own, so a
    // call is one memcpy-free launch and one stream synchronisation
    float4* h_one = nullptr;       // host address
    float4* d_one = nullptr;       // the same memory as the device sees it
    void* one_ws = nullptr;        // 256-byte counter block (kept at zero by the kernel) + one wave's spill area
    hipStream_t one_stream = nullptr;
    std::mutex one_mu;
    // gigl_photon_trace_batch: a workspace kept between calls (grown on demand), used on one_stream under one_mu
    void* photon_ws = nullptr;
    size_t photon_ws_bytes = 0;
    unsigned photon_epoch = 0;           // the one-launch route's barrier count
    const void* photon_small_at = nullptr;   // where in photon_ws the one-launch route's ticket word was last cleared
    float photon_rate = 4.0f;            // small sets: paths a kept record costs, learnt from the sets made so far (sizes the launch)
    void* photon_h = nullptr;            // small sets (one launch): page-locked, device-mapped staging the kernel writes
    void* photon_h_dev = nullptr;        // records and totals into (host address / the same memory as the device sees it)
    std::vector<float> photon_mats;      // the materials last uploaded into photon_ws, and where
    const void* photon_mats_at = nullptr;
};

struct gigl_fog {
    int device = 0;
    uint2* d_bricks = nullptr;
    gd::FogParams fp;
    // gigl_sample_density on a few positions (the serial photon walk samples one at a time): pinned,
    // device-mapped staging and a stream of its own, so a call is one launch and one synchronisation
    static constexpr unsigned kStagePositions = 64;
    float* h_stage = nullptr;      // 3 * kStagePositions floats in, kStagePositions floats out
    float* d_stage = nullptr;
    hipStream_t stage_stream = nullptr;
    std::mutex stage_mu;
};

namespace gd {

// A copy of a device scene on another device (or the same one): the flattened tree and the triangle
// records are copied device to device, the host tree's vectors are not duplicated.  Defined in
// gigl_rt.hip, used by the multi-device entry points (gigl_multi.hip).
int scene_clone_to_device(const gigl_scene* src, int device, gigl_scene** out);
// gigl_rt.hip: closest-hit batch with per-ray final ranges and hit triangle records (gigl_photon.hip)
// With a workspace of the caller's (trace_workspace_bytes(ws_blocks) bytes, its first word zero at launch) the call
// neither allocates nor clears anything; the launch is held to ws_blocks workgroups.
int trace_closest_ranges(const gigl_scene* s, const float4* o_tmin, const float4* d_tmax, unsigned n, uint4* hits,
                         float2* ranges, float4* hit_tri, hipStream_t st, void* ws = nullptr, unsigned ws_blocks = 0);
size_t trace_workspace_bytes(unsigned blocks);
void scene_trace_params(const gigl_scene* s, TraceParams& P);
int option_photon_small();

// gigl_rt.hip: stream-ordered allocation from the library's own pool on `device` (release with hipFreeAsync)
hipError_t pool_alloc(int device, void** p, size_t bytes, hipStream_t st);
hipError_t pool_trim(int device);
void scene_count(int device, int delta);
extern std::atomic<unsigned long long> g_pool_allocs;

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

inline unsigned stream_grid(const gigl_scene* s, uint64_t n) {
    uint64_t g = (n + 255) / 256;
    const uint64_t cap = (uint64_t)(s ? s->n_cus : 256) * 8;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// Runs a float -> float kernel over host data (diagnostic / convenience entry points).
inline int map_host(int device, const float* in, size_t in_floats, uint64_t n, float* out,
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

}  // namespace gd
