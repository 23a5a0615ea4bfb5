// gigl_photon.hip — rt::PhotonTracer::trace (Source/RT/PhotonTracer.cpp:78-188) as ray batches.
//
// The reference walks one path at a time and draws every random number from one global generator
// (Source/Common/Random.cpp:11-18), so path p's numbers depend on how many paths 0..p-1 consumed.  The
// batched walk gives every path a random-number substream of its own (the build's definition, stated in
// include/gigl_rt.h and mirrored by the oracle's ora_photon_trace_substreams), walks a block of paths
// at once — one closest-hit batch per bounce through the same traversal kernel as every other query,
// one lane per path for Woodcock tracking, Russian roulette, VPL records and the next ray — and then
// keeps the VPLs of the first paths, in path order, up to max_vpl_count, exactly as the serial loop
// would have: the path that supplies VPL number max_vpl_count is the last one counted (:115, :159,
// :180), LightArray::normalizeIntensity (Source/VPL/LightArray.hpp:46-66) renumbers path ids and divides
// by the number of paths.  The serial single-stream walk stays available to a C++ host through
// gigl_trace_one (gigl_amd/host/gigl_host.hpp).
//
// cos, sin, log and pow are evaluated in double precision and rounded to float: the correctly rounded
// float result, the same on host and device, which no float libm guarantees.  Everything else is the
// float arithmetic of the reference in the reference's order (-ffp-contract=off).
#include <cstring>

#include <chrono>
#include "gigl_trace_body.h"   // the traversal body: photon_small_kernel walks the tree itself
#include "gigl_walk_lean.h"    // ... in this lean form when the scene is in triple records

using namespace gd;

namespace {

// Source/Common/Constants.h
constexpr float kTwoPi = 6.28318548f, kPi = 3.14159274f, kInvPi = 0.318309873f;
constexpr float kSurvivalRR = 0.95f, kMaxFogHeight = 548.8f, kHgG = 0.25f;
constexpr int kGiBounces = 3;

__host__ __device__ inline unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// draw k of path p: the top 24 bits of splitmix64(key(p) + k * gamma), scaled to [0, 1)
struct Draw {
    unsigned long long key;
    unsigned k;
    __device__ float operator()() {
        const unsigned long long z = splitmix64(key + (unsigned long long)k * 0x9E3779B97F4A7C15ull);
        ++k;
        return (float)(unsigned)(z >> 40) * 5.9604644775390625e-8f;
    }
};
__device__ __forceinline__ unsigned long long path_key(unsigned long long seed, unsigned path) {
    return splitmix64(seed ^ splitmix64((unsigned long long)path));
}

__device__ __forceinline__ float cr_cos(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float cr_sin(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float cr_log(float x) { return (float)log((double)x); }
__device__ __forceinline__ float cr_pow(float x, float y) { return (float)pow((double)x, (double)y); }

__device__ __forceinline__ F3 vec_sin_cos(float phi, float sin_the, float cos_the) {   // RTBase.h:44-50
    return f3(sin_the * cr_cos(phi), sin_the * cr_sin(phi), cos_the);
}
struct M3 { F3 c0, c1, c2; };
__device__ __forceinline__ F3 mul3(const M3& m, F3 v) {   // GLM type_mat3x3.inl:593-596
    return f3((m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
              (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z);
}
__device__ __forceinline__ M3 span_tangent_space(F3 normal) {   // RTBase.h:33-41
    const F3 c1 = cross3(normal, f3(0.0f, 0.0f, 1.0f)), c2 = cross3(normal, f3(0.0f, 1.0f, 0.0f));
    const F3 c3 = len3(c1) > len3(c2) ? c1 : c2;
    const F3 tangent = norm3(-c3);
    return M3{tangent, cross3(normal, tangent), normal};
}

struct PathState {
    float ax, ay, az;     // attenuation
    unsigned k;           // draws taken from the path's substream so far
    unsigned alive;       // the path still has a ray in flight
    unsigned n_vpl;       // VPL records written to the path's slots (at most kGiBounces)
    unsigned pad0, pad1;
};
struct VplRec { unsigned w[24]; };   // 96 B, PointLight.h:66-77 (std140)

struct WalkParams {
    // per path of the block
    PathState* state;
    float4* o_tmin;
    float4* d_tmax;
    const uint4* hits;
    const float2* ranges;
    const float4* hit_tri;
    VplRec* slots;            // kGiBounces per path
    unsigned first, n;        // path ids [first, first + n)
    unsigned long long seed;
    // scene
    const float* materials;   // 12 floats per material: k_d, -, k_s, n_s, k_e, -
    unsigned n_mat;
    FogParams fog;
    BoxF fog_box;
    int fog_on;
    float abs_k, sca_k, maj_ext_k;
    float src[3], src_int[3];
    unsigned* error;          // set when a hit triangle names a material beyond n_mat
    unsigned* trace_counter;  // the traversal launches' work counter: the walk's kernels leave it at zero for the next one
    unsigned* totals;         // the running totals (Acc, 8 words) and the error word behind them...
    unsigned n_total_words;   // ...cleared by the first block's start kernel when > 0
};

__device__ __forceinline__ void store_ray(const WalkParams& W, unsigned i, F3 o, F3 d) {
    W.o_tmin[i] = make_float4(o.x, o.y, o.z, 0.0f);        // Ray ctor: valid range (0, FLT_MAX), RTBase.cpp:111-115
    W.d_tmax[i] = make_float4(d.x, d.y, d.z, FLT_MAX);
}
// A path that has ended keeps a ray with an empty range, which leaves the traversal at the box test.
__device__ __forceinline__ void retire(const WalkParams& W, unsigned i, PathState& st) {
    st.alive = 0u;
    W.o_tmin[i].w = FLT_MAX;
    W.d_tmax[i].w = -FLT_MAX;
}

// PhotonTracer.cpp:84-93: the start of path i of the block
__device__ __forceinline__ void path_start(const WalkParams& W, unsigned i) {
    Draw draw{path_key(W.seed, W.first + i), 0u};
    if (W.src[2] <= kMaxFogHeight) (void)draw();   // :86 — the draw is made, its result never reaches the path (:88-92)
    const float u1 = draw(), u0 = draw();          // :90, arguments evaluated right to left
    const float sin_the = sqrtf(1.0f - u1 * u1);
    const F3 dir = vec_sin_cos(kTwoPi * u0, sin_the, u1);
    const float pdf = 0.5f * kInvPi;
    const float ia = 1.0f / pdf;
    PathState st;
    st.ax = ia; st.ay = ia; st.az = ia;
    st.k = draw.k; st.alive = 1u; st.n_vpl = 0u; st.pad0 = st.pad1 = 0u;
    W.state[i] = st;
    store_ray(W, i, f3(W.src[0], W.src[1], W.src[2]), dir);
}
__global__ void __launch_bounds__(256) photon_start_kernel(const WalkParams W) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) *W.trace_counter = 0u;
    if (i < W.n_total_words) W.totals[i] = 0u;
    if (i >= W.n) return;
    path_start(W, i);
}

__device__ __forceinline__ float sample_density(const FogParams& F, F3 p) {   // DensityField.cpp:177-202, BBox.cpp:53-55
    return fog_filter(F, (p.x - F.mnx) / (F.mxx - F.mnx), (p.y - F.mny) / (F.mxy - F.mny), (p.z - F.mnz) / (F.mxz - F.mnz));
}

// PhotonTracer.cpp:95-178: what happens to path i of the block after scene.trace(ray), one lane per path
__device__ __forceinline__ void path_bounce(const WalkParams& W, unsigned i, int last_bounce) {
    PathState st = W.state[i];
    if (!st.alive) return;
    Draw draw{path_key(W.seed, W.first + i), st.k};
    const float4 o4 = W.o_tmin[i], d4 = W.d_tmax[i];
    const F3 o = f3(o4.x, o4.y, o4.z), d = f3(d4.x, d4.y, d4.z);
    const uint4 h = W.hits[i];
    const bool hit = h.x != kMiss;
    const float dist = __uint_as_float(h.y);   // FLT_MAX when nothing was hit (Intersection's initial distance)
    const float2 rg = W.ranges[i];             // t_min, t_max as KdTree::intersect left them in the ray
    RayReg r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    set_dir(r, d.x, d.y, d.z);
    r.tmin = rg.x; r.tmax = rg.y;
    const F3 src_int = f3(W.src_int[0], W.src_int[1], W.src_int[2]);
    VplRec* const slots = W.slots + (size_t)i * kGiBounces;
    bool go_on = false;      // the bounce ended in `continue` (a new ray was made)
    bool surface = true;     // fall through to the surface event
    F3 new_o = o, new_d = d;
    float f_entr = FLT_MAX, f_exit = -FLT_MAX;
    const bool fog_hit = W.fog_on ? box_intersect(W.fog_box, r, f_entr, f_exit) : false;   // Scene::traceFog, Scene.cpp:245-251
    if (fog_hit) {
        const float t_lo = gmax(f_entr, 0.0f), t_hi = gmin(f_exit, dist);   // :101-102
        bool event = false;
        float t_event = 0.0f;
        if (0.0f != W.maj_ext_k) {   // calcEventDistWT, :62-76
            float d_event = t_lo, p_event = 0.0f;
            while (d_event < t_hi && p_event < draw()) {
                const float dt = -cr_log(1.0f - draw()) / W.maj_ext_k;
                d_event += dt;
                const F3 s_pos = o + d_event * d;
                const float ext_k = (W.abs_k + W.sca_k) * sample_density(W.fog, s_pos);   // FogVolume.cpp:32-36
                p_event = ext_k / W.maj_ext_k;
            }
            event = d_event < t_hi;
            t_event = d_event;
        }
        if (event) {
            surface = false;
            const F3 event_pt = o + t_event * d;
            const float sk = W.sca_k * sample_density(W.fog, event_pt);   // FogVolume.cpp:38-41
            const float sca_albedo = W.sca_k / (W.abs_k + W.sca_k);
            if (draw() < sca_albedo) {
                const F3 I = d;
                const F3 vi = src_int * f3(st.ax, st.ay, st.az);
                VplRec v;
                for (int k = 0; k < 24; ++k) v.w[k] = 0u;   // PointLight.cpp:15-17
                v.w[0] = __float_as_uint(event_pt.x); v.w[1] = __float_as_uint(event_pt.y); v.w[2] = __float_as_uint(event_pt.z);
                v.w[3] = 1u;
                v.w[4] = __float_as_uint(vi.x); v.w[5] = __float_as_uint(vi.y); v.w[6] = __float_as_uint(vi.z);
                v.w[7] = __float_as_uint(sk);
                v.w[8] = __float_as_uint(I.x); v.w[9] = __float_as_uint(I.y); v.w[10] = __float_as_uint(I.z);
                v.w[11] = W.first + i;
                slots[st.n_vpl++] = v;
                // HGPhaseFunc::importanceSample, :49-58
                const float hphi = kTwoPi * draw();
                const float uu = draw();
                const float nn = 1.0f - kHgG * kHgG;
                const float dd = 1.0f - kHgG + 2.0f * kHgG * uu;
                const float q = nn / dd;
                const float cos_the = 0.5f / fabsf(kHgG) * (1.0f + kHgG * kHgG - q * q);
                const float sin_t = sqrtf(1.0f - cos_the * cos_the);
                const F3 O = mul3(span_tangent_space(I), vec_sin_cos(hphi, sin_t, cos_the));
                st.ax = st.ax * 1.0f; st.ay = st.ay * 1.0f; st.az = st.az * 1.0f;   // throughput 1 (:57, :122)
                new_o = event_pt; new_d = O;   // Ray{event_pt, O}
                go_on = true;
            }
        }
    }
    if (surface && hit) {
        const F3 hit_pt = o + dist * d;
        const float4 ta = W.hit_tri[3 * (size_t)i], tb = W.hit_tri[3 * (size_t)i + 1], tc = W.hit_tri[3 * (size_t)i + 2];
        // Scene::trace leaves normalize(cross(e0, e1)) in the ray (RTBase.cpp:170, Scene.cpp:238)
        const F3 norm = norm3(cross3(f3(ta.w, tb.x, tb.y), f3(tb.z, tb.w, tc.x)));
        const unsigned mat_id = __float_as_uint(tc.z);
        if (mat_id >= W.n_mat) {
            *W.error = 1u;
        } else if (!(dot3(d, norm) >= 0.0f || hit_pt.z > kMaxFogHeight)) {   // :139
            const float* mat = W.materials + 12 * (size_t)mat_id;
            const F3 k_d = ld3(mat), k_s = ld3(mat + 4);
            const float n_s = mat[7];
            const bool emissive = mat[8] > 0.0f || mat[9] > 0.0f || mat[10] > 0.0f;
            if (!emissive && draw() < kSurvivalRR) {
                st.ax = st.ax / kSurvivalRR; st.ay = st.ay / kSurvivalRR; st.az = st.az / kSurvivalRR;
                const F3 I = -d;
                const F3 vi = src_int * f3(st.ax, st.ay, st.az);
                const F3 lifted = hit_pt + 1E-4f * norm;   // PointLight.cpp:21
                VplRec v;
                for (int k = 0; k < 24; ++k) v.w[k] = 0u;
                v.w[0] = __float_as_uint(lifted.x); v.w[1] = __float_as_uint(lifted.y); v.w[2] = __float_as_uint(lifted.z);
                v.w[3] = 2u;
                v.w[4] = __float_as_uint(vi.x); v.w[5] = __float_as_uint(vi.y); v.w[6] = __float_as_uint(vi.z);
                v.w[8] = __float_as_uint(I.x); v.w[9] = __float_as_uint(I.y); v.w[10] = __float_as_uint(I.z);
                v.w[11] = W.first + i;
                v.w[12] = __float_as_uint(norm.x); v.w[13] = __float_as_uint(norm.y); v.w[14] = __float_as_uint(norm.z);
                v.w[16] = __float_as_uint(k_d.x); v.w[17] = __float_as_uint(k_d.y); v.w[18] = __float_as_uint(k_d.z);
                v.w[20] = __float_as_uint(k_s.x); v.w[21] = __float_as_uint(k_s.y); v.w[22] = __float_as_uint(k_s.z);
                v.w[23] = __float_as_uint(n_s);
                slots[st.n_vpl++] = v;
                // PhongMaterial::importanceSampleBRDF, RTBase.cpp:70-109
                const float bu1 = draw(), bu2 = draw();
                const F3 lum = f3(0.2126f, 0.7152f, 0.0722f);
                const float dw = dot3(k_d, lum), gw = dot3(k_s, lum);
                const float sp_diffuse = dw / (dw + gw);
                const float u_rand = draw();
                F3 O, throughput;
                if (u_rand < sp_diffuse) {
                    const F3 micro_O = vec_sin_cos(kTwoPi * bu1, sqrtf(1.0f - bu2), sqrtf(bu2));   // RTBase.cpp:27-36
                    O = mul3(span_tangent_space(norm), micro_O);
                    const float den = micro_O.z * sp_diffuse;
                    throughput = f3(k_d.x * kPi / den, k_d.y * kPi / den, k_d.z * kPi / den);
                } else {
                    const F3 S = -I + (2.0f * dot3(I, norm)) * norm;   // reflect, RTBase.cpp:22-24
                    const float lobe_cos = cr_pow(bu2, 1.0f / (n_s + 1.0f));   // RTBase.cpp:39-49
                    const F3 micro_O = vec_sin_cos(kTwoPi * bu1, sqrtf(1.0f - lobe_cos * lobe_cos), lobe_cos);
                    O = mul3(span_tangent_space(S), micro_O);
                    const float den = (n_s + 1.0f) * (1.0f - sp_diffuse);
                    throughput = f3(k_s.x * kTwoPi / den, k_s.y * kTwoPi / den, k_s.z * kTwoPi / den);
                }
                const float cos_the = dot3(O, norm);
                const F3 tc3 = throughput * cos_the;
                st.ax = st.ax * tc3.x; st.ay = st.ay * tc3.y; st.az = st.az * tc3.z;
                new_o = hit_pt + kRayOffset * norm; new_d = O;   // Ray{hit_pt, O, norm}, RTBase.cpp:117-122
                go_on = true;
            }
        }
    }
    st.k = draw.k;
    if (go_on && !last_bounce) {
        store_ray(W, i, new_o, new_d);
    } else {
        retire(W, i, st);
    }
    W.state[i] = st;
}
__global__ void __launch_bounds__(256) photon_bounce_kernel(const WalkParams W, int last_bounce) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) *W.trace_counter = 0u;
    if (i >= W.n) return;
    path_bounce(W, i, last_bounce);
}

// ---- keeping the first max_vpl_count records in path order --------------------------------------
// Running totals over the blocks of paths walked so far.
struct Acc {
    unsigned size;        // VPL records kept
    unsigned contrib;     // paths that supplied at least one of them (the renumbered path id of the next one)
    unsigned n_paths;     // set with `done`: paths counted by the serial loop
    unsigned done;        // record number max_vpl_count has been written
    unsigned error;
    unsigned pad[3];
};

// per path: (records, 1 if any) packed as low/high halves of one 64-bit sum
__device__ __forceinline__ unsigned long long path_counts(const PathState& s) {
    return (unsigned long long)s.n_vpl | ((unsigned long long)(s.n_vpl ? 1u : 0u) << 32);
}
__device__ __forceinline__ unsigned long long block_exclusive_scan_u64(unsigned long long v, unsigned long long* lds,
                                                                       unsigned long long& total) {
    const unsigned t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        const unsigned long long add = t >= off ? lds[t - off] : 0ull;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[255];
    const unsigned long long excl = lds[t] - v;
    __syncthreads();
    return excl;
}
__global__ void __launch_bounds__(256) vpl_block_sums_kernel(const PathState* st, unsigned n, unsigned long long* sums) {
    __shared__ unsigned long long lds[256];
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    unsigned long long total;
    (void)block_exclusive_scan_u64(i < n ? path_counts(st[i]) : 0ull, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// one workgroup: block sums -> exclusive offsets
__global__ void __launch_bounds__(256) vpl_scan_sums_kernel(unsigned long long* sums, unsigned n_blocks) {
    __shared__ unsigned long long lds[256];
    unsigned long long run = 0ull;
    for (unsigned b0 = 0; b0 < n_blocks; b0 += 256u) {
        const unsigned b = b0 + threadIdx.x;
        const unsigned long long v = b < n_blocks ? sums[b] : 0ull;
        unsigned long long total;
        const unsigned long long excl = block_exclusive_scan_u64(v, lds, total);
        if (b < n_blocks) sums[b] = run + excl;
        run += total;
    }
    if (threadIdx.x == 0) sums[n_blocks] = run;   // the grand total, for vpl_advance_kernel
}
__global__ void __launch_bounds__(256) vpl_gather_kernel(const PathState* st, const VplRec* slots, unsigned first,
                                                         unsigned n, const unsigned long long* sums, unsigned max_vpl,
                                                         Acc* acc, VplRec* out) {
    __shared__ unsigned long long lds[256];
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long mine = i < n ? path_counts(st[i]) : 0ull;
    unsigned long long total;
    const unsigned long long excl = sums[blockIdx.x] + block_exclusive_scan_u64(mine, lds, total);
    const unsigned base = acc->size + (unsigned)(excl & 0xFFFFFFFFull);
    const unsigned new_id = acc->contrib + (unsigned)(excl >> 32);
    const unsigned cnt = (unsigned)(mine & 0xFFFFFFFFull);
    for (unsigned j = 0; j < cnt; ++j) {
        const unsigned pos = base + j;
        if (pos >= max_vpl) break;
        VplRec v = slots[(size_t)i * kGiBounces + j];
        v.w[11] = new_id;   // LightArray::normalizeIntensity's renumbering: ids count the contributing paths
        out[pos] = v;
        if (pos + 1u == max_vpl) {   // the serial loop stops inside this path: it is the last one counted
            acc->n_paths = first + i + 1u;
            acc->done = 1u;
        }
    }
}
// after the gather of a block of paths (single thread): totals for the next block
__global__ void vpl_advance_kernel(const unsigned long long* sums, unsigned n_blocks, Acc* acc, const unsigned* error) {
    const unsigned long long tot = sums[n_blocks];
    acc->size += (unsigned)(tot & 0xFFFFFFFFull);
    acc->contrib += (unsigned)(tot >> 32);
    if (*error) acc->error = 1u;
}
// launched after every block of paths; acts once, when the block has supplied record number max_vpl_count
__global__ void __launch_bounds__(256) vpl_normalize_kernel(VplRec* out, unsigned n, const Acc* acc) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !acc->done) return;
    const float inv = 1.0f / (float)(int)acc->n_paths;   // LightArray.hpp:47
    for (int k = 4; k < 7; ++k) out[i].w[k] = __float_as_uint(__uint_as_float(out[i].w[k]) * inv);
}


// ---- a whole small set in one launch ---------------------------------------------------------------
// The batched route above is a chain of a dozen dependent launches and a read-back per block of paths:
// 0.21 ms for the reference's own 150-VPL set.  For blocks of up to kSmallMaxPaths paths one launch does
// everything on the device: a workgroup is ONE wave that walks 1 to 16 paths on its first lanes (a wave takes, at
// every phase, as long as the longest of its paths; a few hundred such waves spread over the chip and none of them
// waits for another); per bounce the wave walks the tree itself (triple records: walk_lean, one ray per lane in one
// lean loop, gigl_walk_lean.h; the other layouts: trace_body in its SOLO form, its batch the wave's own rays, no work
// counter), then every lane advances its path, writing the records it makes straight into page-locked host memory.
// A path that holds records finally writes how many, the wave checks in and leaves; the last wave to check in raises
// a flag in host memory, which the host call polls instead of waiting for the launch to retire.  The ordered gather
// (where every record goes, which path completes the set and hence 1 / n_paths, the renumbering of the contributing
// paths: LightArray::normalizeIntensity, LightArray.hpp:46-66) is then a walk over a few hundred counts and a copy
// of the set's records on the host.  One launch, no device-to-host copy.  Same arithmetic, same order of records:
// byte-identical to the batched route.  What a set costs is the dependent chain of its slowest path
// (GIGL_PHOTON_STAMPS=1 builds stamp it: three tree walks and three path steps, some 8,000 instructions of a lone
// wave at one every 8-13 cycles: 50-57 us, against 20 us for the average wave) plus launch and flag latency.
constexpr unsigned kSmallMaxPaths = 4096;   // paths per launch: the page-locked slots hold three records for each
constexpr unsigned kSmallMaxVpls = 1024;    // largest set the one-launch route takes (4 x 1,024 paths per launch)

// the phases of a wave exchange their data through global memory: the fence completes the stores before the loads
// of the next phase are issued
__device__ __forceinline__ void phase_fence() {
    // one wave, one CU: workgroup scope is enough (the CU's vector cache is write-through and shared by its waves)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
struct SmallParams {
    unsigned* h_cnt;                 // page-locked host memory as the device sees it: per path of the launch, the records
                                     // it holds (the host zeroes it; only paths that hold any write theirs)
    unsigned* h_err;                 // ... the error word of the launch (path_bounce's), written by the last wave
    unsigned* h_flag;                // ... and the word the last wave sets to `token` once every wave has checked in
    unsigned token;
    unsigned* arrivals;              // device: waves that have checked in; only grows
    unsigned epoch;                  // its value when this launch starts
    unsigned lanes;                  // paths per wave: 1, 2, 4 ... 32
#if GIGL_PHOTON_STAMPS
    unsigned long long* stamps;      // 16 real-time stamps per wave (diagnostic build)
#endif
};

template <bool TRIPLES>
__global__ void __launch_bounds__(64) photon_small_kernel(const WalkParams W, const TraceParams P, const SmallParams K) {
    // a wave walks K.lanes paths (8, 16 or 32; any divisor of 64 works) on its first lanes: fewer paths per wave means fewer rounds of the
    // divergent loops (tree walk, Woodcock tracking) per wave, and the waves have the whole chip to spread over
    const unsigned L = K.lanes;
    const unsigned g = blockIdx.x, lane = threadIdx.x;
#if GIGL_PHOTON_STAMPS
    unsigned n_st = 0;
#define STAMP() do { if (lane == 0u) K.stamps[g * 16u + n_st] = __builtin_amdgcn_s_memrealtime(); ++n_st; } while (0)
#else
#define STAMP() do {} while (0)
#endif
    STAMP();
    const unsigned nb = W.n;
    const bool mine_path = lane < L && g * L + lane < nb;
    const unsigned i = mine_path ? g * L + lane : 0u;
    TraceParams Q = P;
    Q.o_tmin = P.o_tmin + g * L; Q.d_tmax = P.d_tmax + g * L;
    Q.hits = P.hits + g * L; Q.ranges = P.ranges + g * L; Q.hit_tri = P.hit_tri + 3u * g * L;
    Q.n = nb - g * L < L ? nb - g * L : L;
    Q.counter = nullptr;   // SOLO: no work counter
    // trace_body addresses its spill column by workgroup * 64 + lane; only the first L lanes of a wave ever hold a
    // ray, so the columns of consecutive waves are packed L apart (the host sizes the area for the paths, not for
    // 64 lanes per wave)
    Q.spill = P.spill - (size_t)g * (64u - L);
    if (mine_path) path_start(W, i);
    STAMP();
    for (int b = 1; b <= kGiBounces; ++b) {
        phase_fence();
        // triple records (the default layout): every lane walks its own ray in one lean loop; the other layouts go
        // through the shared body in its one-wave form
        if constexpr (TRIPLES) walk_lean<64>(Q, lane < Q.n, lane);
        else trace_body<TRIPLES, false, 0, 8, 64, false, false, true>(Q, b > 1);
        phase_fence();
        STAMP();
        if (mine_path) path_bounce(W, i, b == kGiBounces ? 1 : 0);
        STAMP();
    }
    // ---- the ordered gather (vpl_gather_kernel's) is the host's ----
    // W.slots is page-locked host memory here: the records went out as path_bounce made them.  A path that holds
    // any now says how many; then the wave checks in and leaves — nobody waits.  The wave that checks in last
    // raises the flag the host call polls, and the host lays the set out: it walks the paths' counts in order, which
    // gives every record its place, the path that supplies record number max_vpl (the last one the serial loop
    // counts, hence 1 / n_paths) and the renumbering of the contributing paths (LightArray::normalizeIntensity,
    // LightArray.hpp:46-66) — a few hundred words and some 150 records, microseconds of one core, where a lone wave
    // took 10-15 us over the same job (and an earlier form, a barrier in global memory behind which every wave placed
    // its own records, had the waves that were done early polling: a few hundred pollers slowed the ones still
    // walking down several times).
    const unsigned cnt = mine_path ? W.state[i].n_vpl : 0u;   // at most kGiBounces = 3
    if (cnt) K.h_cnt[g * L + lane] = cnt;
    STAMP();
    __threadfence_system();   // records and count are out (system scope) before the wave checks in
    unsigned ticket = 0u;
    if (lane == 0u) ticket = atomicAdd(K.arrivals, 1u);
    ticket = __shfl(ticket, 0);
    if (ticket != K.epoch + gridDim.x - 1u) return;
    __threadfence();   // what the others wrote (the error word) is seen behind the last check-in
    STAMP();
    if (lane == 0u) {
        *K.h_err = __hip_atomic_load(W.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *W.error = 0u;
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0u) __hip_atomic_store(K.h_flag, K.token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    STAMP();
}

// walk_lean over an explicit ray batch, one ray per lane, 64 per workgroup: the cross-check of the lean walk against
// the shipped kernel on batches of any size (tests), and the measure of what a lone ray costs in either form
__global__ void __launch_bounds__(64) lean_trace_kernel(const TraceParams P) {
    const unsigned i = blockIdx.x * 64u + threadIdx.x;
    walk_lean<64>(P, i < P.n, i);
}

}  // namespace

extern "C" int gigl_debug_trace_closest_lean_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                                                    gigl_hit* hits, void* stream) {
    if (!s || ((!o_tmin || !d_tmax || !hits) && n)) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (!s->triples) return fail(GIGL_ERR_INVALID, "the lean walk reads triple records (layout 2)");
    if (n > kMaxRays) return fail(GIGL_ERR_TOO_LARGE, "at most 0xF0000000 rays per call");
    if (n == 0) return GIGL_OK;
    DeviceGuard g(s->device);
    TraceParams P = {};
    scene_trace_params(s, P);
    P.o_tmin = (const float4*)o_tmin; P.d_tmax = (const float4*)d_tmax;
    P.hits = (uint4*)hits;
    P.n = (unsigned)n;
    hipLaunchKernelGGL(lean_trace_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, P);
    HIP_TRY(hipGetLastError());
    return GIGL_OK;
}

extern "C" int gigl_photon_trace_batch(const gigl_scene* s, const gigl_fog* fog, const float* materials, uint32_t n_mat,
                                       float abs_k, float sca_k, float maj_ext_k, int fog_enabled,
                                       const float ppl_pos[3], const float ppl_intens[3], const float shoot_dir[3],
                                       int max_vpl_count, uint64_t seed, uint32_t paths_per_batch, void* vpls_out,
                                       int* n_vpls_out, uint32_t* n_paths_out) {
    (void)shoot_dir;   // PhotonTracer.cpp:86-88 builds a frame from it that the walk never uses
#if GIGL_PHOTON_STAMPS
    const auto t_entry = std::chrono::steady_clock::now();
#endif
    if (n_vpls_out) *n_vpls_out = 0;
    if (n_paths_out) *n_paths_out = 0;
    if (!s || !materials || !ppl_pos || !ppl_intens || !vpls_out || !n_vpls_out)
        return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    if (max_vpl_count <= 0 || max_vpl_count > (1 << 22)) return fail(GIGL_ERR_INVALID, "max_vpl_count must be in 1..2^22");
    if (n_mat == 0) return fail(GIGL_ERR_INVALID, "no materials");
    const bool fog_on = fog_enabled && fog;
    if (fog && fog->device != s->device) return fail(GIGL_ERR_INVALID, "fog and scene live on different devices");
    const uint64_t path_limit = 100ull * (uint64_t)max_vpl_count;   // :180
    uint64_t per = paths_per_batch ? paths_per_batch : std::max<uint64_t>(4ull * (uint64_t)max_vpl_count, 1024);
    per = std::min<uint64_t>(std::min<uint64_t>(per, path_limit), 1u << 22);
    per = (per + 255) & ~255ull;
    const unsigned P = (unsigned)per, n_blocks = P / 256u;

    DeviceGuard g(s->device);
    // one allocation: state, rays, hits, ranges, hit triangles, VPL slots, kept records, scan sums, totals, materials
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_state = take((size_t)P * sizeof(PathState)), o_o = take((size_t)P * 16), o_d = take((size_t)P * 16);
    const size_t o_hits = take((size_t)P * 16), o_rng = take((size_t)P * 8), o_tri = take((size_t)P * 48);
    const size_t o_slots = take((size_t)P * kGiBounces * sizeof(VplRec));
    const size_t o_out = take((size_t)max_vpl_count * sizeof(VplRec));
    const size_t o_sums = take(((size_t)n_blocks + 1) * 8), o_acc = take(sizeof(Acc)), o_err = take(4);
    const size_t o_mat = take((size_t)n_mat * 48);
    // the traversal launches' workspace (work counter + stack spill area), so that they neither allocate nor clear
    const unsigned trace_blocks = std::min<unsigned>(n_blocks, (unsigned)s->n_cus * 6u);
    // a small set in one launch (photon_small_kernel)
    const int small_opt = option_photon_small();
    const bool small = (unsigned)max_vpl_count <= kSmallMaxVpls && P <= kSmallMaxPaths &&
                       (small_opt == 1 || (small_opt < 0 && paths_per_batch == 0));
    // its spill area: packed columns, one per path (photon_small_kernel), kSmallMaxPaths and a wave's worth at most
    const size_t o_tws = take(trace_workspace_bytes(std::max<unsigned>(trace_blocks, kSmallMaxPaths / 256u + 1u)));
    // the check-in counter of the one-launch route
    const size_t o_small = take(256);
    // the workspace and the stream belong to the scene and are kept between calls (a 150-VPL set is a
    // dozen short launches: allocating per call would cost as much as the walk); calls on one scene serialise
    gigl_scene* ms = const_cast<gigl_scene*>(s);
    std::lock_guard<std::mutex> lk(ms->one_mu);
    if (ms->photon_ws_bytes < off) {
        if (ms->photon_ws) { (void)hipStreamSynchronize(ms->one_stream); (void)hipFree(ms->photon_ws); }
        ms->photon_ws = nullptr; ms->photon_ws_bytes = 0;
        ms->photon_mats.clear(); ms->photon_mats_at = nullptr;
        HIP_TRY(hipMalloc(&ms->photon_ws, off));
        ms->photon_ws_bytes = off;
        ms->photon_small_at = nullptr;
    }
    char* const ws = (char*)ms->photon_ws;
    const hipStream_t st = ms->one_stream;
    int rc = GIGL_OK;
    Acc acc = {};
    auto run = [&]() -> int {
        // materials travel only when they differ from the last call's (a renderer passes the same ones every frame)
        if (ms->photon_mats.size() != (size_t)n_mat * 12 || ms->photon_mats_at != ws + o_mat ||
            std::memcmp(ms->photon_mats.data(), materials, (size_t)n_mat * 48) != 0) {
            ms->photon_mats.clear();   // no stale match if the copy fails
            HIP_TRY(hipMemcpyAsync(ws + o_mat, materials, (size_t)n_mat * 48, hipMemcpyHostToDevice, st));
            ms->photon_mats.assign(materials, materials + (size_t)n_mat * 12);
            ms->photon_mats_at = ws + o_mat;
        }
        WalkParams W = {};
        W.state = (PathState*)(ws + o_state);
        W.o_tmin = (float4*)(ws + o_o); W.d_tmax = (float4*)(ws + o_d);
        W.hits = (const uint4*)(ws + o_hits); W.ranges = (const float2*)(ws + o_rng); W.hit_tri = (const float4*)(ws + o_tri);
        W.slots = (VplRec*)(ws + o_slots);
        W.seed = seed;
        W.materials = (const float*)(ws + o_mat); W.n_mat = n_mat;
        W.fog_on = fog_on ? 1 : 0;
        if (fog) {
            W.fog = fog->fp;
            W.fog_box = BoxF{fog->fp.mnx, fog->fp.mny, fog->fp.mnz, fog->fp.mxx, fog->fp.mxy, fog->fp.mxz};
        }
        W.abs_k = abs_k; W.sca_k = sca_k; W.maj_ext_k = maj_ext_k;
        for (int a = 0; a < 3; ++a) { W.src[a] = ppl_pos[a]; W.src_int[a] = ppl_intens[a]; }
        W.error = (unsigned*)(ws + o_err);
        W.trace_counter = (unsigned*)(ws + o_tws);
        W.totals = (unsigned*)(ws + o_acc);
        static_assert(sizeof(Acc) == 32, "Acc is 8 words; the error word follows at the next 256-byte step");
        unsigned long long* sums = (unsigned long long*)(ws + o_sums);
        Acc* d_acc = (Acc*)(ws + o_acc);
        VplRec* d_out = (VplRec*)(ws + o_out);
        if (small) {
            // page-locked, device-mapped, coherent (uncached on the device side): the paths' record slots, their
            // counts, and the error and flag words — the host reads them as soon as the flag says every wave is done
            constexpr size_t h_slots_bytes = (size_t)kSmallMaxPaths * kGiBounces * sizeof(VplRec);
            constexpr size_t h_cnt_at = h_slots_bytes, h_words_at = h_cnt_at + (size_t)kSmallMaxPaths * 4;
            constexpr size_t h_bytes = h_words_at + 256;
            if (!ms->photon_h) {
                HIP_TRY(hipHostMalloc(&ms->photon_h, h_bytes, hipHostMallocMapped | hipHostMallocCoherent));
                HIP_TRY(hipHostGetDevicePointer(&ms->photon_h_dev, ms->photon_h, 0));
            }
            W.slots = (VplRec*)ms->photon_h_dev;   // path_bounce writes its records straight to the host
            TraceParams Q = {};
            scene_trace_params(s, Q);
            Q.o_tmin = W.o_tmin; Q.d_tmax = W.d_tmax;
            Q.hits = (uint4*)(ws + o_hits); Q.ranges = (float2*)(ws + o_rng); Q.hit_tri = (float4*)(ws + o_tri);
            Q.spill = (uint4*)(ws + o_tws + 256);
            // a wave of 8 to 32 rays: the T-phase runs until no lane descends any more (a threshold sized for 64 busy
            // lanes would end it after every step): -3 %
            Q.refill_thresh = 24; Q.chunk = 64u; Q.guided = 0u; Q.t_exit = 1;
            SmallParams K = {};
            K.h_cnt = (unsigned*)((char*)ms->photon_h_dev + h_cnt_at);
            K.h_err = (unsigned*)((char*)ms->photon_h_dev + h_words_at);
            K.h_flag = (unsigned*)((char*)ms->photon_h_dev + h_words_at + 128);
            K.arrivals = (unsigned*)(ws + o_small);
            // the check-in counter only grows (every launch adds one arrival per wave; the host keeps count); a new
            // workspace, or one whose layout moved with another block size, starts from zero, and so does the error
            // word, which the last wave clears again after reading it
            if (ms->photon_small_at != ws + o_small) {
                HIP_TRY(hipMemsetAsync(ws + o_small, 0, 64, st));
                HIP_TRY(hipMemsetAsync(W.error, 0, 4, st));
                ms->photon_small_at = ws + o_small;
                ms->photon_epoch = 0u;
            }
            // A launch walks what the set is expected to need and a quarter more (the workspace stays sized for P): a
            // set of the reference's size needs about 3 paths per record, so 640 paths rather than 1,024.  The estimate
            // follows the sets made so far.
            const unsigned Ps = std::min<unsigned>(P, std::max<unsigned>(256u,
                (unsigned)(1.25f * ms->photon_rate * (float)max_vpl_count + 127.0f) & ~127u));
            // Few paths per wave: a wave takes, at every phase, as long as the longest of its paths, and no wave waits
            // for another.  Measured on the 150-VPL set (640 paths): 1 / 2 / 4 / 8 paths per wave 0.088 / 0.082 / 0.083 /
            // 0.084 ms — beyond some 400 waves the slowest ones get slower (two and three waves to a CU), so the
            // launch takes the fewest paths per wave that keep it under 384 waves.
            K.lanes = 1u;
            while ((Ps + K.lanes - 1u) / K.lanes > 384u) K.lanes *= 2u;
            if (const char* e = getenv("GIGL_PHOTON_LANES")) { const unsigned l = (unsigned)atoi(e); if (l && l <= 32u && (l & (l - 1u)) == 0u) K.lanes = l; }
#if GIGL_PHOTON_STAMPS
            // stamps go to device memory and are copied back after the launch (written to host memory they queue up
            // behind one another on the way out, and the phases' fences wait for them)
            static unsigned long long* d_stamps = nullptr;
            static std::vector<unsigned long long> h_stamps_v;
            if (!d_stamps) HIP_TRY(hipMalloc((void**)&d_stamps, (size_t)kSmallMaxPaths * 16 * 8));
            h_stamps_v.assign((size_t)kSmallMaxPaths * 16, 0ull);
            unsigned long long* h_stamps = h_stamps_v.data();
            HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)kSmallMaxPaths * 16 * 8, st));
            K.stamps = d_stamps;
#endif
            auto kern = s->triples ? photon_small_kernel<true> : photon_small_kernel<false>;
            const VplRec* h_slots = (const VplRec*)ms->photon_h;
            unsigned* h_cnt = (unsigned*)((char*)ms->photon_h + h_cnt_at);
            const volatile unsigned* h_err = (const volatile unsigned*)((char*)ms->photon_h + h_words_at);
            volatile unsigned* h_flag = (volatile unsigned*)((char*)ms->photon_h + h_words_at + 128);
            VplRec* out = (VplRec*)vpls_out;
            uint64_t first = 0;
            while (first < path_limit) {
                const unsigned nb = (unsigned)std::min<uint64_t>(Ps, path_limit - first);
                const unsigned grid = (nb + K.lanes - 1u) / K.lanes;
                W.first = (unsigned)first; W.n = nb;
                Q.spill_stride = grid * K.lanes;   // packed columns (photon_small_kernel)
                K.epoch = ms->photon_epoch;
                ms->photon_epoch += grid;
                K.token = ms->photon_epoch | 0x80000000u;   // never 0, new with every launch
                *h_flag = 0u;
                std::memset(h_cnt, 0, (size_t)grid * K.lanes * 4);   // only the paths that hold records write their count
#if GIGL_PHOTON_STAMPS
                const auto tq0 = std::chrono::steady_clock::now();
#endif
                hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, st, W, Q, K);
                hipError_t le = hipGetLastError();
#if GIGL_PHOTON_STAMPS
                const auto tq1 = std::chrono::steady_clock::now();
#endif
                // The last wave raises the flag when the set and the totals are in host memory: the call returns then,
                // some 20 us before the runtime would report the launch retired.  Every few thousand polls the stream is
                // asked whether it failed or finished without the flag (a fault).
                if (le == hipSuccess) {
                    for (unsigned spins = 1u;; ++spins) {
                        if (__atomic_load_n((const unsigned*)h_flag, __ATOMIC_ACQUIRE) == K.token) break;
                        if ((spins & 4095u) == 0u) {
                            const hipError_t q = hipStreamQuery(st);
                            if (q != hipErrorNotReady) {   // retired (or failed): whatever it wrote is there now
                                le = q;
                                break;
                            }
                        }
                    }
                }
#if GIGL_PHOTON_STAMPS
                if (getenv("GIGL_PHOTON_TIMING")) {
                    (void)hipStreamSynchronize(st);
                    (void)hipMemcpy(h_stamps, d_stamps, (size_t)kSmallMaxPaths * 16 * 8, hipMemcpyDeviceToHost);
                    // per stamp: latest wave, relative to the earliest start; names of the intervals
                    static const char* nm[] = {"start", "path_start", "walk1", "bounce1", "walk2", "bounce2", "walk3", "bounce3",
                                               "count out", "last wave: checked in", "flag", "-"};
                    unsigned long long t0 = ~0ull;
                    for (unsigned w = 0; w < grid; ++w) t0 = std::min(t0, h_stamps[w * 16]);
                    fprintf(stderr, "[stamps] %u waves;", grid);
                    for (unsigned k = 0; k < 12; ++k) {
                        unsigned long long mx = 0; double mean = 0; unsigned cnt = 0;
                        for (unsigned w = 0; w < grid; ++w) if (h_stamps[w * 16 + k]) { mx = std::max(mx, h_stamps[w * 16 + k]); mean += (double)(h_stamps[w * 16 + k] - t0); ++cnt; }
                        if (cnt) fprintf(stderr, " %s %.1f/%.1f", nm[k], mean / cnt * 0.01, (double)(mx - t0) * 0.01);
                    }
                    fprintf(stderr, " (us since the first wave started: mean/latest over waves)\n");
                }
#endif
#if GIGL_PHOTON_STAMPS
                if (getenv("GIGL_PHOTON_TIMING")) {
                    const auto tq2 = std::chrono::steady_clock::now();
                    fprintf(stderr, "[photon_small] entry->launch %.1f us, launch call %.1f us, wait %.1f us\n",
                            std::chrono::duration<double, std::micro>(tq0 - t_entry).count(),
                            std::chrono::duration<double, std::micro>(tq1 - tq0).count(),
                            std::chrono::duration<double, std::micro>(tq2 - tq1).count());
                }
#endif
                if (le != hipSuccess) {
                    ms->photon_small_at = nullptr;   // the counter is no longer where the host thinks it is
                    return fail(GIGL_ERR_HIP, std::string("photon_small_kernel: ") + hipGetErrorString(le));
                }
                if (*h_err) return fail(GIGL_ERR_INVALID, "a hit triangle names a material beyond n_mat");
                // The ordered gather (vpl_gather_kernel's), on the host: the paths in order, every record to its place;
                // the path that supplies record number max_vpl_count is the last one the serial loop counts
                // (PhotonTracer.cpp:115, :159, :180); ids count the contributing paths (LightArray.hpp:46-66).
                for (unsigned p = 0; p < nb && !acc.done; ++p) {
                    const unsigned c = h_cnt[p];
                    if (!c) continue;
                    for (unsigned j = 0; j < c && !acc.done; ++j) {
                        VplRec& r = out[acc.size];
                        r = h_slots[(size_t)p * kGiBounces + j];
                        r.w[11] = acc.contrib;
                        if (++acc.size == (unsigned)max_vpl_count) {
                            acc.done = 1u;
                            acc.n_paths = (unsigned)first + p + 1u;
                        }
                    }
                    ++acc.contrib;
                }
                if (acc.done) break;
                first += nb;
            }
            if (!acc.done) {   // :181-184
                std::memset(vpls_out, 0, (size_t)max_vpl_count * sizeof(VplRec));
                if (n_paths_out) *n_paths_out = (uint32_t)path_limit;
                return GIGL_OK;
            }
            {   // LightArray::normalizeIntensity, LightArray.hpp:47 and PointLight.h:68: intensity / n_paths
                const float inv = 1.0f / (float)(int)acc.n_paths;
                for (int v = 0; v < max_vpl_count; ++v)
                    for (int c = 4; c < 7; ++c) {
                        float f;
                        std::memcpy(&f, &out[v].w[c], 4);
                        f = f * inv;
                        std::memcpy(&out[v].w[c], &f, 4);
                    }
            }
            ms->photon_rate = std::max(0.5f, 0.5f * ms->photon_rate + 0.5f * (float)acc.n_paths / (float)max_vpl_count);
            *n_vpls_out = max_vpl_count;
            if (n_paths_out) *n_paths_out = acc.n_paths;
            return GIGL_OK;
        }
        uint64_t first = 0;
        while (first < path_limit) {
            const unsigned nb = (unsigned)std::min<uint64_t>(P, path_limit - first);
            const unsigned grid = (nb + 255u) / 256u;
            W.first = (unsigned)first; W.n = nb;
            W.n_total_words = first == 0 ? (unsigned)((o_err + 4 - o_acc) / 4) : 0u;
            hipLaunchKernelGGL(photon_start_kernel, dim3(grid), dim3(256), 0, st, W);
            for (int b = 1; b <= kGiBounces; ++b) {
                int trc = trace_closest_ranges(s, W.o_tmin, W.d_tmax, nb, (uint4*)(ws + o_hits), (float2*)(ws + o_rng),
                                               (float4*)(ws + o_tri), st, ws + o_tws, trace_blocks);
                if (trc != GIGL_OK) return trc;
                hipLaunchKernelGGL(photon_bounce_kernel, dim3(grid), dim3(256), 0, st, W, b == kGiBounces ? 1 : 0);
            }
            hipLaunchKernelGGL(vpl_block_sums_kernel, dim3(grid), dim3(256), 0, st, W.state, nb, sums);
            hipLaunchKernelGGL(vpl_scan_sums_kernel, dim3(1), dim3(256), 0, st, sums, grid);
            hipLaunchKernelGGL(vpl_gather_kernel, dim3(grid), dim3(256), 0, st, W.state, W.slots, W.first, nb, sums,
                               (unsigned)max_vpl_count, d_acc, d_out);
            hipLaunchKernelGGL(vpl_advance_kernel, dim3(1), dim3(1), 0, st, sums, grid, d_acc, W.error);
            hipLaunchKernelGGL(vpl_normalize_kernel, dim3(((unsigned)max_vpl_count + 255u) / 256u), dim3(256), 0, st, d_out,
                               (unsigned)max_vpl_count, d_acc);
            HIP_TRY(hipGetLastError());
            // totals into the scene's page-locked staging (gigl_trace_one's, free while one_mu is held): a copy
            // into pageable memory would block the caller once more
            static_assert(sizeof(Acc) <= 7 * sizeof(float4), "Acc fits the scene's pinned staging");
            HIP_TRY(hipMemcpyAsync(ms->h_one, d_acc, sizeof(Acc), hipMemcpyDeviceToHost, st));
            // the first block of paths nearly always supplies the whole set: its records come back behind the
            // totals, one synchronisation for both
            const bool with_records = first == 0;
            if (with_records)
                HIP_TRY(hipMemcpyAsync(vpls_out, d_out, (size_t)max_vpl_count * sizeof(VplRec), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::memcpy(&acc, ms->h_one, sizeof(Acc));
            if (acc.error) return fail(GIGL_ERR_INVALID, "a hit triangle names a material beyond n_mat");
            if (acc.done) {
                if (!with_records) {
                    HIP_TRY(hipMemcpyAsync(vpls_out, d_out, (size_t)max_vpl_count * sizeof(VplRec), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                }
                break;
            }
            first += nb;
        }
        if (!acc.done) {   // :181-184: not enough VPLs within 100 * max_vpl_count paths: the array is cleared
            std::memset(vpls_out, 0, (size_t)max_vpl_count * sizeof(VplRec));   // what the first block's copy left there
            if (n_paths_out) *n_paths_out = (uint32_t)path_limit;
            return GIGL_OK;
        }
        *n_vpls_out = max_vpl_count;
        if (n_paths_out) *n_paths_out = acc.n_paths;
        return GIGL_OK;
    };
    rc = run();
    if (rc != GIGL_OK) (void)hipStreamSynchronize(st);   // nothing of a failed call may still be running on the workspace
    return rc;
}
