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
// stack holds (t_split, far child's node words) — t_min/t_max are rebuilt from t_split on pop, and
// a pop costs no global load — with the first STACK_N entries in LDS (lane-interleaved, conflict
// free) and the deeper entries (2 % of pushes on the 1M-triangle scene with 8 in LDS) in a global
// spill area.
// Device data layout: kd_layout.cpp (triple records, or pair-addressed nodes in clustered 128-byte
// lines or level treelets; trace_kernel walks either) and the 64-byte triangle records
// built in gigl_scene_create().
#include "gigl_trace_body.h"

using namespace gd;

namespace {

#if GIGL_WAVES_PER_EU
#define GIGL_WAVES_ATTR __attribute__((amdgpu_waves_per_eu((STACK_N == 8 || STACK_N == GIGL_LDS_STACK) ? GIGL_WAVES_PER_EU : 1, 8)))
#else
#define GIGL_WAVES_ATTR
#endif
template <bool TRIPLES, bool ANY, int GEN, int STACK_N, int BLOCK, bool STATS = false, bool XCHG = false>
__global__ void __launch_bounds__(BLOCK) GIGL_WAVES_ATTR trace_kernel(const TraceParams P) {
    trace_body<TRIPLES, ANY, GEN, STACK_N, BLOCK, STATS, XCHG>(P);   // gigl_trace_body.h
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
__global__ void __launch_bounds__(256) primary_rays_kernel(const Cam12 cam, unsigned res_x, unsigned first,
                                                           unsigned n, float4* o_tmin, float4* d_tmax) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        float dx, dy, dz;
        primary_ray(cam.c, res_x, first + i, dx, dy, dz);
        o_tmin[i] = make_float4(cam.c[0], cam.c[1], cam.c[2], 0.0f);
        d_tmax[i] = make_float4(dx, dy, dz, FLT_MAX);
    }
}

// The per-point half of every shadow pair, once per point instead of once per pair (150 lights per point
// in configs[1]): hit point, facing normal, offset origin.  Same operations, same results.
__global__ void __launch_bounds__(256) shadow_points_kernel(TraceParams P, unsigned n_pts, float4* cache) {
    for (unsigned pt = blockIdx.x * 256u + threadIdx.x; pt < n_pts; pt += gridDim.x * 256u) {
        float px = 0.0f, py = 0.0f, pz = 0.0f, ox = 0.0f, oy = 0.0f, oz = 0.0f;
        const bool ok = shadow_point(P, pt, px, py, pz, ox, oy, oz);
        cache[2 * (size_t)pt] = make_float4(px, py, pz, ok ? 1.0f : 0.0f);
        cache[2 * (size_t)pt + 1] = make_float4(ox, oy, oz, 0.0f);
    }
}

__global__ void __launch_bounds__(256) transm_shadow_kernel(FogParams F, TraceParams P, float* out) {
    GIGL_FOG_LUT_DECL
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < P.n; i += gridDim.x * 256u) {
        ShadowSeg s;
        float tr = 1.0f;
        if (shadow_segment(P, i, s)) tr = calc_transm<GIGL_FOG_LUT != 0>(F, s.px, s.py, s.pz, s.wx, s.wy, s.wz, s.dist_sq, s_unorm);
        out[i] = tr;
    }
}

// calcTransm for the pairs the any-hit pass found unoccluded only (the shader's order,
// Surface.frag:250-256).  37 % of the Cornell pairs are occluded; skipping them lane by lane would
// save nothing (a wave marches as long as its slowest lane), so every wave compacts: it walks its
// block of pairs 64 at a time, writes 0 for the occluded ones, queues the indices of the others in
// LDS (ballot + prefix count) and marches whenever 64 are queued, every lane busy.
constexpr unsigned kVisPairsPerWave = 2048;
__global__ void __launch_bounds__(256) transm_visible_kernel(FogParams F, TraceParams P, const unsigned long long* bits,
                                                             float* out) {
    __shared__ unsigned s_q[4][128];
    GIGL_FOG_LUT_DECL
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned* q = s_q[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned n_blocks = (P.n + kVisPairsPerWave - 1) / kVisPairsPerWave;
    auto march = [&](unsigned idx) {
        ShadowSeg s;
        float tr = 1.0f;   // points whose primary ray missed
        if (shadow_segment(P, idx, s)) tr = calc_transm<GIGL_FOG_LUT != 0>(F, s.px, s.py, s.pz, s.wx, s.wy, s.wz, s.dist_sq, s_unorm);
        out[idx] = tr;
    };
    for (unsigned blk = blockIdx.x * 4u + wave; blk < n_blocks; blk += gridDim.x * 4u) {
        const unsigned base = blk * kVisPairsPerWave;
        unsigned count = 0;
        for (unsigned c = 0; c < kVisPairsPerWave; c += 64u) {
            const unsigned idx = base + c + lane;
            if (base + c >= P.n) break;
            const unsigned long long word = bits[(base + c) >> 6];   // base is a multiple of 64: one word per round
            const bool valid = idx < P.n;
            const bool occluded = (word >> lane) & 1ull;
            if (valid && occluded) out[idx] = 0.0f;   // nothing arrives from an occluded light
            const unsigned long long m = __ballot(valid && !occluded);
            if (valid && !occluded) q[count + __popcll(m & lt)] = idx;
            count += __popcll(m);
            if (count >= 64u) {
                march(q[lane]);
                const unsigned rest = count - 64u;
                const unsigned moved = lane < rest ? q[64u + lane] : 0u;
                if (lane < rest) q[lane] = moved;
                count = rest;
            }
        }
        if (lane < count) march(q[lane]);
    }
}

__global__ void __launch_bounds__(256) eps_kernel(const float* t, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = rescaled_eps(t[i]);
}


}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;

int gd::fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

namespace {

struct Options {
    int stack_lds = 8;        // LDS entries per lane (12 bytes each): 8, 12, 16 or 32; 8 keeps 6 workgroups per CU
    int refill_thresh = 0;    // idle lanes that trigger a refill (closest-hit / atomic any-hit), 1..64; 0: the engine's choice:
                              // 24, or 40 on a tiny scene (see chunk)
    int any_atomic = -1;      // 0: any-hit waves own 64 consecutive rays, one store per wave; 1: lane refill +
                              // atomicOr; -1: the engine's choice (1)
    int blocks_per_cu = 0;    // 0: occupancy query
    int chunk = 0;            // rays per work-counter atomic; 0: the engine's choice (256; 512 for structures served
                              // from HBM, where a wave's rays live twice as long: +1.6 % on S(500), -2 % on S(158);
                              // 1,024 on a tiny scene — a structure of at most 1 MB, resident in every cache: its rays are
                              // so cheap that the waves' requests saturate the one counter they all go through (152 M
                              // Cornell shadow pairs: 76 requests per us in chunks of 256 against the ~88 per us one word
                              // serves).  Chunks of 1,024, no shrinking requests near the end, refills at 40 idle lanes:
                              // 7.88 -> 5.36 ms, profiles/r03_knob_sweep_cornell_shadow.txt)
    int guided = -1;          // 0: fixed chunks; k: near the end a request is remaining / (k * waves); -1: the engine's choice:
                              // 4, or 0 on a tiny scene (see chunk)
    int t_exit = 20;          // see trace_kernel: T-phase exit threshold (1..64)
    int stats = 0;            // 1: closest-hit launches run the instrumented build and keep 8 counters
    int sort_rays = 0;        // explicit-ray batches: 1 regroup rays before tracing, 0 never, -1 deep tree + large
                              // batch.  Off by default: on MI355X the two atomic passes (2.9 ms per 2^24 rays)
                              // cost more than the 1.7 ms the regrouped traversal saves
    int sort_bits = 6;        // cells per axis = 2^sort_bits (4..6)
    int treelet_levels = 3;   // read at gigl_scene_create: levels of child pairs per cache-line treelet (1..5)
    int leaf_blocks = -1;     // read at gigl_scene_create (triple layout): leaves with at least this many prims read
                              // a private block of triangle copies (kd_layout.h); 0 never; -1 the engine's choice:
                              // 2 when tree and triangle table exceed the 256 MB Infinity Cache anyway, else 6
                              // (only the big leaves: +14 % bytes on S(158), still cache resident, +1 %)
    int shadow_order = 1;     // shadow-pair batches (generator 2): 1 trace the pairs light by light (neighbouring
                              // points side by side in a wave: coherent rays), 0 in pair order (one point's lights
                              // side by side); the output layout is the same
    int build_threads = 0;    // read at gigl_scene_create: host threads of the tree build (0: min(16, cores)); the
                              // tree does not depend on it
    int xchg_every = 1;       // GIGL_XCHG builds: rounds between two exchanges of rays among the waves of a workgroup (0: the
                              // ordinary kernel)
    int xchg_min = 1;         // GIGL_XCHG builds: smallest group of lanes a wave runs a phase for (1: every phase runs)
    int photon_small = -1;    // gigl_photon_trace_batch: 1 sets of up to 1,024 VPLs run as one workgroup in one launch
                              // (gigl_photon.hip, photon_small_kernel), 0 never, -1 the engine's choice: when the
                              // caller leaves paths_per_batch to the engine
    int layout = 2;           // read at gigl_scene_create (kd_layout.h): 0 level treelets + separate id lists,
                              // 1 clustered lines with the id lists in their leaves' lines, 2 triple records
                              // (two levels per load; falls back to 1 beyond its encoding)
};
Options g_opt;
std::mutex g_opt_mu;
unsigned long long g_last_stats[kStatsWords] = {0};   // [0..11] the counters, [16...] the timeline histograms

// ---- the no-allocation entry point's scratch (gigl_trace_closest_noalloc_device) ----------------
// A file-scope device array: every GPU the library runs on holds its own copy, at its own address,
// until the process ends.  Host code never uses the array's name as an address: gigl_scene_create
// asks hipGetSymbolAddress for it once per device (scratch_for_device) and the pointer it returns
// is what the kernels, and hipMemsetAsync, are given.
constexpr unsigned kNoAllocBlocks = 256;          // workgroups of a no-allocation launch (one per CU)
constexpr unsigned kSpillDepth = 32 - 8;          // entries per lane beyond the 8 in LDS (stack of 32, KdTree.hpp:276)
constexpr size_t kScratchBytes = 256 + (size_t)kNoAllocBlocks * kBlock * kSpillDepth * sizeof(uint4);   // 24 MiB + 256 B
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

// Call with `device` current.  Returns the device address of this GPU's copy of `scratch`.
int scratch_for_device(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return fail(GIGL_ERR_INVALID, "device index beyond the scratch table");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(GIGL_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return GIGL_OK;
}

}  // namespace
namespace gd {
// Stream-ordered allocations (launch workspaces, staging of the host-pointer calls) come from a pool of the
// library's own, one per device, not from the device's default pool: with the default reuse policy the
// runtime hands a block whose hipFreeAsync is still pending on stream A to an allocation on stream B as
// soon as B has waited on any event of A, also one recorded BEFORE the launch that still uses the block
// (hipMemPoolReuseFollowEventDependencies; seen on ROCm 7.2 as two concurrent traversals sharing one work
// counter, tests/test_gpu_streams.py).  This pool reuses a block only on the stream that freed it, or
// once that free has completed.
std::mutex g_pool_mu;
hipMemPool_t g_pool[kMaxDevices] = {nullptr};
std::atomic<unsigned long long> g_pool_allocs{0};   // gigl_debug_pool_allocs: what a no-allocation call must leave unchanged
int g_scenes_on_device[kMaxDevices] = {0};          // under g_pool_mu: the last scene to go trims the pool

hipError_t pool_alloc(int device, void** p, size_t bytes, hipStream_t st) {
    *p = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    hipMemPool_t pool;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool[device]) {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = device;
            hipMemPool_t np = nullptr;
            hipError_t e = hipMemPoolCreate(&np, &props);
            if (e != hipSuccess) return e;
            uint64_t keep = UINT64_MAX;   // the pool keeps its memory between calls
            int off = 0;
            e = hipMemPoolSetAttribute(np, hipMemPoolAttrReleaseThreshold, &keep);
            if (e == hipSuccess) e = hipMemPoolSetAttribute(np, hipMemPoolReuseFollowEventDependencies, &off);
            if (e == hipSuccess) e = hipMemPoolSetAttribute(np, hipMemPoolReuseAllowInternalDependencies, &off);
            if (e != hipSuccess) {
                (void)hipMemPoolDestroy(np);
                return e;
            }
            g_pool[device] = np;
        }
        pool = g_pool[device];
    }
    g_pool_allocs.fetch_add(1, std::memory_order_relaxed);
    return hipMallocFromPoolAsync(p, bytes, pool, st);
}

// Returns the memory the pool holds but no allocation uses to the device (the pool keeps the largest workspaces
// a process ever asked for between calls: regrouping buffers, the pi-density hit buffer, the shadow point
// cache, the host-pointer pipeline's slots).
hipError_t pool_trim(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_pool[device]) return hipSuccess;
    return hipMemPoolTrimTo(g_pool[device], 0);
}
void scene_count(int device, int delta) {
    if (device < 0 || device >= kMaxDevices) return;
    bool trim = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_scenes_on_device[device] += delta;
        trim = delta < 0 && g_scenes_on_device[device] <= 0;
    }
    if (trim) (void)pool_trim(device);
}
}  // namespace gd
namespace {

// Workspace handed in by the caller of launch_trace instead of a stream-ordered allocation.
struct ExtWorkspace {
    void* base;           // 256 bytes of counters, then the spill area
    unsigned max_grid;    // workgroups the spill area was sized for (kSpillDepth entries per lane)
    bool counter_is_zero; // the caller's kernels on the same stream leave the work counter at zero: no memset
};

template <bool ANY, int GEN, int STACK_N, int BLOCK = kBlock>
int launch_trace_t(const gigl_scene* s, TraceParams& P, const Options& opt, hipStream_t st, const ExtWorkspace* ext) {
    auto kern = s->triples ? trace_kernel<true, ANY, GEN, STACK_N, BLOCK> : trace_kernel<false, ANY, GEN, STACK_N, BLOCK>;
    // instrumented builds: explicit-ray closest-hit with the default LDS stack only
    const bool stats = !ANY && GEN == 0 && (STACK_N == 8 || STACK_N == GIGL_LDS_STACK) && BLOCK == kBlock && opt.stats && !ext;
    if (stats) kern = s->triples ? trace_kernel<true, ANY, GEN, STACK_N, BLOCK, true> : trace_kernel<false, ANY, GEN, STACK_N, BLOCK, true>;
    bool xchg = false;
#if GIGL_XCHG
    if constexpr (!ANY && GEN == 0 && BLOCK == kBlock && STACK_N == GIGL_LDS_STACK) {
        if (s->triples && !ext && opt.xchg_every > 0) {
            xchg = true;
            kern = stats ? trace_kernel<true, false, 0, STACK_N, BLOCK, true, true> : trace_kernel<true, false, 0, STACK_N, BLOCK, false, true>;
            P.xchg_every = (unsigned)opt.xchg_every;
            P.xchg_min = opt.xchg_min;
        }
    }
#endif
    unsigned grid;
    if (ext) {
        grid = ext->max_grid;
    } else {
        int per_cu = opt.blocks_per_cu;
        if (per_cu <= 0) {
            // one cache per instantiation (this function is a template), layout and instrumented build;
            // the devices of a node are identical, so it is not keyed by device
            static std::atomic<int> cached[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
            std::atomic<int>& c = cached[(xchg ? 4 : 0) + (stats ? 2 : 0) + (s->triples ? 1 : 0)];
            int q = c.load(std::memory_order_relaxed);
            if (q <= 0) {
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, BLOCK, 0));
                q = q < 1 ? 1 : q;
                c.store(q, std::memory_order_relaxed);
            }
            per_cu = q;
        }
        grid = (unsigned)(s->n_cus * per_cu);
        if (getenv("GIGL_RT_DEBUG")) fprintf(stderr, "[gigl_rt] trace launch: %d blocks/CU x %d CUs\n", per_cu, s->n_cus);
    }
    const unsigned need = (P.n + BLOCK - 1) / BLOCK;
    if (grid > need) grid = need;
    if (grid == 0) return GIGL_OK;
    const bool spill = STACK_N < 32 && s->tree.actual_depth - 1 > STACK_N;
    // workspace: 256 bytes (work counter, finish ticket; instrumented builds: counters and timeline behind them), spill area
    const size_t hdr = stats ? ((128 + sizeof(g_last_stats) + 255) & ~(size_t)255) : 256;
    void* ws = nullptr;
    if (ext) {
        if (STACK_N < 8) return fail(GIGL_ERR_INVALID, "the external spill area holds 32 - 8 entries per lane");
        ws = ext->base;
    } else {
        const size_t spill_entries = spill ? (size_t)grid * BLOCK * (size_t)(32 - STACK_N) : 0;
        HIP_TRY(gd::pool_alloc(s->device, &ws, hdr + spill_entries * sizeof(uint4), st));
    }
    if (!P.reset_counter && !(ext && ext->counter_is_zero) && hipMemsetAsync(ws, 0, hdr, st) != hipSuccess) {
        if (!ext) (void)hipFreeAsync(ws, st);
        return fail(GIGL_ERR_HIP, "hipMemsetAsync(workspace)");
    }
    P.counter = (unsigned*)ws;
    P.spill = (uint4*)((char*)ws + hdr);
    P.spill_stride = grid * BLOCK;
    // guided requests only for batches that are long against the machine: a small batch would run
    // on minimum-size requests from its first ray (one atomic round trip per refill)
    P.guided = opt.guided > 0 ? grid * (BLOCK / 64u) * (unsigned)opt.guided : 0u;
    if ((uint64_t)P.n < 2ull * P.guided * P.chunk) P.guided = 0u;
    P.stats = nullptr;
    if (stats) P.stats = (unsigned long long*)((char*)ws + 128);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), 0, st, P);
    hipError_t le = hipGetLastError();
    if (P.stats && le == hipSuccess) {  // diagnostic path: synchronous read-back
        std::lock_guard<std::mutex> lk(g_opt_mu);
        (void)hipMemcpyAsync(g_last_stats, P.stats, sizeof(g_last_stats), hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    }
    hipError_t fe = ext ? hipSuccess : hipFreeAsync(ws, st);
    if (le != hipSuccess) return fail(GIGL_ERR_HIP, std::string("trace kernel launch: ") + hipGetErrorString(le));
    if (fe != hipSuccess) return fail(GIGL_ERR_HIP, std::string("hipFreeAsync: ") + hipGetErrorString(fe));
    return GIGL_OK;
}

template <bool ANY, int GEN>
int launch_trace(const gigl_scene* s, TraceParams& P, hipStream_t st, const ExtWorkspace* ext = nullptr) {
    Options opt;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        opt = g_opt;
    }
    if (ext) {   // no allocation of any kind: no regrouping pass, the spill area is sized for 8 LDS entries
        opt.sort_rays = 0;
        opt.stack_lds = 8;
    }
    P.nodes = s->d_nodes;
    P.root = s->root;
    P.recs = reinterpret_cast<const uint4*>(s->d_nodes);
    P.root_rec = s->root_rec;
    P.leaf_ids = s->d_leaf_ids ? s->d_leaf_ids : reinterpret_cast<const unsigned*>(s->d_nodes);  // clustered: lists in the node buffer
    P.tris = s->d_tris;
    P.blocks_off8 = s->blocks_off8;
    P.box = s->box;
    // any-hit output: lane refill + atomicOr unless the caller asks for the group mode (a wave owns 64
    // consecutive rays and stores one word); with the current kernels the refill mode wins on shallow
    // trees too (Cornell shadow batch 13.9 against 12.2 Grays/s)
    P.any_atomic = ANY ? (opt.any_atomic < 0 ? 1 : opt.any_atomic) : 0;
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
        if (gd::pool_alloc(s->device, &sort_ws, total, st) != hipSuccess) return fail(GIGL_ERR_HIP, "pool_alloc(regroup)");
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
    if (GEN == 2 || GEN == 3) {
        // light-major needs every lane to write its own bit (atomicOr mode) and both counts to fit the index maths
        if (!(ANY && P.any_atomic && opt.shadow_order == 1 && P.n_lights > 1u)) P.n_pts = 0u;
    }
    const bool tiny_scene = s->stats.device_bytes <= ((uint64_t)1 << 20);
    if (opt.refill_thresh <= 0) opt.refill_thresh = tiny_scene ? 40 : 24;
    if (opt.guided < 0) opt.guided = tiny_scene ? 0 : 4;
    P.refill_thresh = (ANY && !P.any_atomic) ? 64 : opt.refill_thresh;
    if (P.refill_thresh < 1) P.refill_thresh = 1;
    if (P.refill_thresh > 64) P.refill_thresh = 64;
    P.chunk = opt.chunk > 0 ? (unsigned)opt.chunk
                            : (s->stats.device_bytes > ((uint64_t)256 << 20) ? 512u : (tiny_scene ? 1024u : 256u));
    if (opt.chunk <= 0) {
        // a batch that is small against the machine: smaller requests, so that every resident wave gets some
        // (2^18 rays in requests of 256 would occupy a sixth of the waves, four rays deep per lane: 0.81 ms
        // against 0.63), but no more requests than about 4/3 per wave: they all go through one counter, and
        // cheap coherent rays are bound by it (1024^2 Cornell primary rays: 0.28 ms in requests of 64, 0.17 in 256)
        // (half as many on a tiny scene, whose rays are cheap everywhere: 1024^2 Cornell primary rays then keep 256)
        const uint64_t max_requests = (uint64_t)s->n_cus * (tiny_scene ? 16u : 32u);
        while (P.chunk > 64u && (uint64_t)P.n <= max_requests * (P.chunk / 2u)) P.chunk >>= 1;
    }
    P.t_exit = opt.t_exit;
    int rc = GIGL_OK;
    if (ANY && P.any_atomic)
        if (hipMemsetAsync(P.bits, 0, ((size_t)P.n + 63) / 64 * 8, st) != hipSuccess)
            rc = fail(GIGL_ERR_HIP, "hipMemsetAsync(bits)");
    // smallest LDS stack that holds the deepest possible stack avoids the spill path entirely
    int sn = opt.stack_lds;
    const int deepest = s->tree.actual_depth - 1;
    if (deepest <= 8 || ext) sn = 8;
    else if (deepest <= 12 && sn >= 12) sn = 12;
    else if (deepest <= 16 && sn >= 16) sn = 16;
    if (rc == GIGL_OK) switch (sn) {
        // GIGL_LDS_STACK (build switch, default 8): the depth the default launches run with; launches on a caller's
        // workspace keep 8, which their spill areas are sized for
        case 8: rc = ext ? launch_trace_t<ANY, GEN, 8>(s, P, opt, st, ext) : launch_trace_t<ANY, GEN, GIGL_LDS_STACK>(s, P, opt, st, ext); break;
        case 12: rc = launch_trace_t<ANY, GEN, 12>(s, P, opt, st, ext); break;
        case 16: rc = launch_trace_t<ANY, GEN, 16>(s, P, opt, st, ext); break;
        case 32: rc = launch_trace_t<ANY, GEN, 32>(s, P, opt, st, ext); break;
        default: rc = fail(GIGL_ERR_INVALID, "option stack_lds must be 8, 12, 16 or 32");
    }
    if (sort_ws && hipFreeAsync(sort_ws, st) != hipSuccess && rc == GIGL_OK) rc = fail(GIGL_ERR_HIP, "hipFreeAsync");
    return rc;
}

// Per-scene resources on the scene's device (current when called): gigl_trace_one's staging,
// workspace and stream, and this device's address of the no-allocation scratch array.
int scene_device_resources(gigl_scene* s) {
    HIP_TRY(hipHostMalloc((void**)&s->h_one, 7 * sizeof(float4), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&s->d_one, s->h_one, 0));
    const size_t one_ws_bytes = 256 + (size_t)64 * kSpillDepth * sizeof(uint4);
    HIP_TRY(hipMalloc(&s->one_ws, one_ws_bytes));
    HIP_TRY(hipMemset(s->one_ws, 0, 256));
    HIP_TRY(hipStreamCreateWithFlags(&s->one_stream, hipStreamNonBlocking));
    const int rc = scratch_for_device(s->device, &s->scratch);
    if (rc == GIGL_OK) { s->counted = true; gd::scene_count(s->device, +1); }
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

// Closest-hit batch that also leaves, per ray, the (t_min, t_max) KdTree::intersect ends with and the hit
// triangle's record: what Scene::trace's callers read from the ray afterwards (gigl_photon.hip).
int gd::option_photon_small() {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    return g_opt.photon_small;
}

// What every traversal launch is told about the scene (the rest of TraceParams is the batch's).
void gd::scene_trace_params(const gigl_scene* s, TraceParams& P) {
    P.nodes = s->d_nodes;
    P.root = s->root;
    P.recs = reinterpret_cast<const uint4*>(s->d_nodes);
    P.root_rec = s->root_rec;
    P.leaf_ids = s->d_leaf_ids ? s->d_leaf_ids : reinterpret_cast<const unsigned*>(s->d_nodes);  // clustered: lists in the node buffer
    P.tris = s->d_tris;
    P.blocks_off8 = s->blocks_off8;
    P.box = s->box;
}

size_t gd::trace_workspace_bytes(unsigned blocks) {
    return 256 + (size_t)blocks * kBlock * kSpillDepth * sizeof(uint4);
}

int gd::trace_closest_ranges(const gigl_scene* s, const float4* o_tmin, const float4* d_tmax, unsigned n, uint4* hits,
                             float2* ranges, float4* hit_tri, hipStream_t st, void* ws, unsigned ws_blocks) {
    if (n == 0) return GIGL_OK;
    TraceParams P = {};
    P.n = n;
    P.o_tmin = o_tmin;
    P.d_tmax = d_tmax;
    P.hits = hits;
    P.ranges = ranges;
    P.hit_tri = hit_tri;
    if (ws && ws_blocks) {
        const ExtWorkspace ext = {ws, ws_blocks, true};
        return launch_trace<false, 0>(s, P, st, &ext);
    }
    return launch_trace<false, 0>(s, P, st);
}

int gd::scene_clone_to_device(const gigl_scene* src, int device, gigl_scene** out) {
    *out = nullptr;
    if (!src || src->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_INVALID, "clone needs a device scene");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return fail(GIGL_ERR_INVALID, "device index out of range");
    gigl_scene* s = new gigl_scene();
    s->device = device;
    s->replica = true;
    s->n_tris = src->n_tris;
    s->n_cus = src->n_cus;
    for (int a = 0; a < 3; ++a) { s->tree.box_min[a] = src->tree.box_min[a]; s->tree.box_max[a] = src->tree.box_max[a]; }
    s->tree.max_depth = src->tree.max_depth; s->tree.actual_depth = src->tree.actual_depth;
    s->tree.leaf_count = src->tree.leaf_count; s->tree.prims_in_leaves = src->tree.prims_in_leaves;
    s->root = src->root; s->root_rec = src->root_rec; s->triples = src->triples; s->layout = src->layout;
    s->box = src->box; s->stats = src->stats;
    s->nb_nodes = src->nb_nodes; s->nb_leaf = src->nb_leaf; s->nb_tris = src->nb_tris;
    s->blocks_off8 = src->blocks_off8;
    DeviceGuard g(device);
    auto cleanup = [&]() {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_leaf_ids) (void)hipFree(s->d_leaf_ids);
        if (s->d_tris) (void)hipFree(s->d_tris);
        if (s->h_one) (void)hipHostFree(s->h_one);
        if (s->one_ws) (void)hipFree(s->one_ws);
        if (s->one_stream) (void)hipStreamDestroy(s->one_stream);
        delete s;
    };
    hipError_t e = g.ok ? hipSuccess : hipErrorInvalidDevice;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) s->n_cus = prop.multiProcessorCount;
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_nodes, s->nb_nodes);
    if (e == hipSuccess && s->nb_leaf) e = hipMalloc((void**)&s->d_leaf_ids, s->nb_leaf);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_tris, s->nb_tris);
    // the replicated tree travels device to device (over xGMI between two GPUs), not through the host
    if (e == hipSuccess) e = hipMemcpyPeer(s->d_nodes, device, src->d_nodes, src->device, s->nb_nodes);
    if (e == hipSuccess && s->nb_leaf) e = hipMemcpyPeer(s->d_leaf_ids, device, src->d_leaf_ids, src->device, s->nb_leaf);
    if (e == hipSuccess) e = hipMemcpyPeer(s->d_tris, device, src->d_tris, src->device, s->nb_tris);
    if (e != hipSuccess) {
        cleanup();
        return fail(GIGL_ERR_HIP, std::string("scene replica: ") + hipGetErrorString(e));
    }
    if (scene_device_resources(s) != GIGL_OK) { cleanup(); return GIGL_ERR_HIP; }
    *out = s;
    return GIGL_OK;
}

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

int gigl_debug_check_layout(const gigl_scene* s, int layout, int treelet_levels, uint64_t* bytes_out) {
    if (!s) return fail(GIGL_ERR_INVALID, "scene is NULL");
    if (s->replica) return fail(GIGL_ERR_INVALID, "a scene adopted from device arrays has no host tree to check against");
    if (layout < 0 || layout > 3 || treelet_levels < 1 || treelet_levels > 5) return fail(GIGL_ERR_INVALID, "bad layout");
    gigl::DeviceTree dt;
    if (layout == 3) dt.leaf_block_min = 2;   // layout 2 with leaf triangle blocks
    if (layout >= 2) {
        if (!gigl::layout_triples(s->tree, dt)) return fail(GIGL_ERR_TOO_LARGE, "tree exceeds the triple-record encoding");
    } else if (layout == 1) gigl::layout_clustered(s->tree, dt);
    else gigl::layout_levels(s->tree, treelet_levels, dt);
    const char* what = nullptr;
    if (!gigl::layout_verify(s->tree, dt, &what)) return fail(GIGL_ERR_BUILD, std::string("device layout differs from the host tree: ") + (what ? what : "?"));
    if (bytes_out) *bytes_out = dt.slots.size() * 8 + (dt.ids_inline ? 0 : dt.ids.size() * 4) + dt.block_ids.size() / 3 * 128;
    return GIGL_OK;
}

int gigl_debug_last_stats(uint64_t out[12]) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    for (int i = 0; i < 12; ++i) out[i] = g_last_stats[i];
    return GIGL_OK;
}

int gigl_debug_last_timeline(uint64_t out[3072]) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    for (unsigned i = 0; i < 3u * 1024u; ++i) out[i] = g_last_stats[kStatsTimeline + i];
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
        if (value < 0 || value > 64) return fail(GIGL_ERR_INVALID, "refill_thresh must be in 0..64");
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
    } else if (k == "layout") {
        if (value < 0 || value > 2) return fail(GIGL_ERR_INVALID, "layout must be 0, 1 or 2");
        g_opt.layout = value;
    } else if (k == "sort_bits") {
        if (value < 4 || value > 6) return fail(GIGL_ERR_INVALID, "sort_bits must be in 4..6");
        g_opt.sort_bits = value;
    } else if (k == "stats") {
        g_opt.stats = value ? 1 : 0;
    } else if (k == "t_exit") {
        if (value < 1 || value > 64) return fail(GIGL_ERR_INVALID, "t_exit must be in 1..64");
        g_opt.t_exit = value;
    } else if (k == "chunk") {
        if (value != 0 && (value < 64 || value > 65536 || value % 64)) return fail(GIGL_ERR_INVALID, "chunk must be 0 or a multiple of 64 in 64..65536");
        g_opt.chunk = value;
    } else if (k == "guided") {
        if (value < -1 || value > 64) return fail(GIGL_ERR_INVALID, "guided must be in -1..64");
        g_opt.guided = value;
    } else if (k == "leaf_blocks") {
        if (value < -1 || value == 1 || value > 64) return fail(GIGL_ERR_INVALID, "leaf_blocks must be -1, 0 or 2..64");
        g_opt.leaf_blocks = value;
    } else if (k == "shadow_order") {
        if (value < 0 || value > 1) return fail(GIGL_ERR_INVALID, "shadow_order must be 0 or 1");
        g_opt.shadow_order = value;
    } else if (k == "build_threads") {
        if (value < 0 || value > 256) return fail(GIGL_ERR_INVALID, "build_threads must be in 0..256");
        g_opt.build_threads = value;
    } else if (k == "xchg_every") {
        if (value < 0 || value > 64) return fail(GIGL_ERR_INVALID, "xchg_every must be in 0..64");
        g_opt.xchg_every = value;
    } else if (k == "xchg_min") {
        if (value < 1 || value > 64) return fail(GIGL_ERR_INVALID, "xchg_min must be in 1..64");
        g_opt.xchg_min = value;
    } else if (k == "photon_small") {
        if (value < -1 || value > 1) return fail(GIGL_ERR_INVALID, "photon_small must be -1, 0 or 1");
        g_opt.photon_small = value;
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

    int treelet_levels, layout, build_threads, leaf_blocks;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        treelet_levels = g_opt.treelet_levels;
        layout = g_opt.layout;
        build_threads = g_opt.build_threads;
        leaf_blocks = g_opt.leaf_blocks;
    }
    gigl_scene* s = new gigl_scene();
    s->device = device;
    s->n_tris = n_tris;
    gigl::KdBuildParams bp;
    bp.inters_cost = inters_cost; bp.trav_cost = trav_cost; bp.max_depth = max_depth;
    bp.min_num_prims = min_num_prims; bp.empty_bonus = empty_bonus;
    bp.n_threads = build_threads;
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
        // word 9: the triangle's id, as in a leaf block's record (the kernel takes the id of a tested triangle
        // from the record in both); word 10: material id (shading stage)
        uint32_t mat = tris[4 * (size_t)i + 3];
        memcpy(&r[9], &i, 4);
        memcpy(&r[10], &mat, 4);
    }
    DeviceGuard g(device);
    if (!g.ok) { delete s; return fail(GIGL_ERR_NO_DEVICE, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete s; return fail(GIGL_ERR_HIP, "hipGetDeviceProperties failed"); }
    s->n_cus = prop.multiProcessorCount;
    // ---- device tree: same topology, re-addressed for the kernels (kd_layout.h) ----
    gigl::DeviceTree dtree;
    if (layout == 2) {
        // Leaf triangle blocks multiply the triangle bytes (a triangle is copied into every leaf that lists
        // it): worth it only when the structure is served from HBM anyway.  Below the Infinity Cache's size
        // the shared 64-byte table stays on-die and wins (kernels).
        const size_t plain_bytes = s->tree.nodes.size() * 6 + (size_t)n_tris * 4 * kTriStride * 4;
        dtree.leaf_block_min = leaf_blocks < 0 ? (plain_bytes > (size_t)200 << 20 ? 2u : 6u) : (uint32_t)leaf_blocks;
        if (!gigl::layout_triples(s->tree, dtree)) { layout = 1; dtree = gigl::DeviceTree(); }   // beyond the record encoding: fall back
    }
    if (layout == 1) gigl::layout_clustered(s->tree, dtree);
    else if (layout == 0) gigl::layout_levels(s->tree, treelet_levels, dtree);
    s->triples = dtree.triples;
    s->layout = layout;
    const std::vector<gigl::KdNode>& dn = dtree.slots;
    const bool ids_inline = dtree.ids_inline;
    // device node word 1 keeps 29 bits for a triangle id / leaf size and 29 for a node slot
    if (n_tris > kLeafPayload || dn.size() >= (1u << 29) || s->tree.leaf_prim_ids.size() > kLeafPayload) {
        delete s;
        return fail(GIGL_ERR_TOO_LARGE, "scene exceeds the device node encoding (2^29 triangles / node slots)");
    }
    const size_t nb_nodes = std::max<size_t>(dn.size(), 2) * sizeof(uint2);
    const size_t nb_leaf = ids_inline ? 0 : std::max<size_t>(dtree.ids.size() + 2, 2) * 4;
    // the triangle table, then (triple layout with leaf triangle blocks) the blocks' 128-byte lines, in one
    // allocation: the kernel names a record in either by one 32-bit offset
    const size_t nb_table = (std::max<size_t>(rec.size(), 16) * 4 + 127) & ~(size_t)127;
    const size_t n_block_lines = dtree.block_ids.size() / 3;
    const size_t nb_blocks = n_block_lines ? n_block_lines * 128 + 128 : 0;   // a line of slack behind the last record
    const size_t nb_tris = nb_table + nb_blocks;
    if (nb_tris >= ((size_t)1 << 35)) { delete s; return fail(GIGL_ERR_TOO_LARGE, "triangle table and leaf blocks exceed 32 GB"); }
    auto cleanup = [&]() {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_leaf_ids) (void)hipFree(s->d_leaf_ids);
        if (s->d_tris) (void)hipFree(s->d_tris);
        if (s->h_one) (void)hipHostFree(s->h_one);
        if (s->one_ws) (void)hipFree(s->one_ws);
        if (s->one_stream) (void)hipStreamDestroy(s->one_stream);
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
    if (!ids_inline) CREATE_TRY(hipMalloc((void**)&s->d_leaf_ids, nb_leaf));
    CREATE_TRY(hipMalloc((void**)&s->d_tris, nb_tris));
    if (scene_device_resources(s) != GIGL_OK) { cleanup(); return GIGL_ERR_HIP; }
    CREATE_TRY(hipMemcpy(s->d_nodes, dn.data(), dn.size() * 8, hipMemcpyHostToDevice));
    if (!ids_inline && !dtree.ids.empty())
        CREATE_TRY(hipMemcpy(s->d_leaf_ids, dtree.ids.data(), dtree.ids.size() * 4, hipMemcpyHostToDevice));
    s->root = make_uint2(dn[0].w0, dn[0].w1);
    s->root_rec = make_uint4(dn[0].w0, dn[0].w1, dn[1].w0, dn[1].w1);
    if (!rec.empty()) CREATE_TRY(hipMemcpy(s->d_tris, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    if (!dtree.block_ids.empty()) {
        // leaf triangle blocks: 128-byte lines of three 40-byte records (p0, e0, e1, id), filled from the
        // table records above and uploaded in slabs so the staging copy stays small
        const size_t n_lines = n_block_lines;
        char* const d_blocks = reinterpret_cast<char*>(s->d_tris) + nb_table;
        s->blocks_off8 = (unsigned)(nb_table / 8);
        const size_t slab_lines = (size_t)1 << 20;
        std::vector<uint32_t> slab(std::min(n_lines, slab_lines) * 32);
        for (size_t l0 = 0; l0 < n_lines; l0 += slab_lines) {
            const size_t m = std::min(slab_lines, n_lines - l0);
            std::memset(slab.data(), 0, m * 128);
            for (size_t l = 0; l < m; ++l)
                for (int k = 0; k < 3; ++k) {
                    const uint32_t id = dtree.block_ids[(l0 + l) * 3 + k];
                    if (id == gigl::kLeafBlockPad) continue;
                    uint32_t* d = &slab[l * 32 + (size_t)k * 10];
                    std::memcpy(d, &rec[(size_t)id * 4 * kTriStride], 36);
                    d[9] = id;
                }
            CREATE_TRY(hipMemcpy(d_blocks + l0 * 128, slab.data(), m * 128, hipMemcpyHostToDevice));
        }
    }
#undef CREATE_TRY
    s->box = BoxF{s->tree.box_min[0], s->tree.box_min[1], s->tree.box_min[2],
                  s->tree.box_max[0], s->tree.box_max[1], s->tree.box_max[2]};
    s->stats.device_bytes = nb_nodes + nb_leaf + nb_tris;
    s->nb_nodes = nb_nodes; s->nb_leaf = nb_leaf; s->nb_tris = nb_tris;
    *out = s;
    return GIGL_OK;
}

int gigl_scene_export_device_blob(const gigl_scene* s, gigl_scene_blob* desc, void* ptrs_out[3]) {
    if (!s || !desc) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: nothing on a device to export");
    memset(desc, 0, sizeof(*desc));
    desc->magic = GIGL_BLOB_MAGIC;
    desc->version = 1;
    desc->nb_nodes = s->nb_nodes; desc->nb_leaf_ids = s->nb_leaf; desc->nb_tris = s->nb_tris;
    desc->layout = (uint32_t)s->layout; desc->triples = s->triples ? 1u : 0u;
    desc->blocks_off8 = s->blocks_off8; desc->n_tris = s->n_tris;
    desc->root[0] = s->root.x; desc->root[1] = s->root.y;
    desc->root_rec[0] = s->root_rec.x; desc->root_rec[1] = s->root_rec.y;
    desc->root_rec[2] = s->root_rec.z; desc->root_rec[3] = s->root_rec.w;
    for (int a = 0; a < 3; ++a) { desc->box[a] = s->tree.box_min[a]; desc->box[3 + a] = s->tree.box_max[a]; }
    desc->stats = s->stats;
    if (ptrs_out) { ptrs_out[0] = s->d_nodes; ptrs_out[1] = s->d_leaf_ids; ptrs_out[2] = s->d_tris; }
    return GIGL_OK;
}

int gigl_scene_create_from_device_blob(const gigl_scene_blob* desc, const void* nodes, const void* leaf_ids,
                                       const void* tris, int device, void* stream, gigl_scene** out) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || desc->magic != GIGL_BLOB_MAGIC || desc->version != 1) return fail(GIGL_ERR_INVALID, "not a scene blob of this library version");
    if (!nodes || !tris || (desc->nb_leaf_ids && !leaf_ids)) return fail(GIGL_ERR_INVALID, "NULL array pointer");
    // what the kernels assume of the arrays: at least the root's two slots, whole 8-byte node words, a triangle
    // table of 64-byte records for n_tris triangles in front of the leaf blocks, layouts 0..2
    if (desc->nb_nodes < 16 || desc->nb_nodes % 8 || desc->layout > 2u || (desc->triples != 0u) != (desc->layout == 2u) ||
        desc->nb_tris < (uint64_t)desc->n_tris * 4 * kTriStride * 4 || (uint64_t)desc->blocks_off8 * 8 > desc->nb_tris ||
        desc->stats.actual_depth < 0 || desc->stats.actual_depth > 32)   // the traversal stacks hold 32 entries
        return fail(GIGL_ERR_INVALID, "inconsistent scene blob header");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(GIGL_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(GIGL_ERR_INVALID, "device index out of range");
    gigl_scene* s = new gigl_scene();
    s->device = device;
    s->replica = true;
    s->n_tris = desc->n_tris;
    for (int a = 0; a < 3; ++a) { s->tree.box_min[a] = desc->box[a]; s->tree.box_max[a] = desc->box[3 + a]; }
    s->tree.max_depth = desc->stats.max_depth; s->tree.actual_depth = desc->stats.actual_depth;
    s->tree.leaf_count = desc->stats.leaf_count;
    s->root = make_uint2(desc->root[0], desc->root[1]);
    s->root_rec = make_uint4(desc->root_rec[0], desc->root_rec[1], desc->root_rec[2], desc->root_rec[3]);
    s->triples = desc->triples != 0u; s->layout = (int)desc->layout;
    s->box = BoxF{desc->box[0], desc->box[1], desc->box[2], desc->box[3], desc->box[4], desc->box[5]};
    s->stats = desc->stats;
    s->nb_nodes = desc->nb_nodes; s->nb_leaf = desc->nb_leaf_ids; s->nb_tris = desc->nb_tris;
    s->blocks_off8 = desc->blocks_off8;
    DeviceGuard g(device);
    auto cleanup = [&]() {
        if (s->d_nodes) (void)hipFree(s->d_nodes);
        if (s->d_leaf_ids) (void)hipFree(s->d_leaf_ids);
        if (s->d_tris) (void)hipFree(s->d_tris);
        if (s->h_one) (void)hipHostFree(s->h_one);
        if (s->one_ws) (void)hipFree(s->one_ws);
        if (s->one_stream) (void)hipStreamDestroy(s->one_stream);
        delete s;
    };
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = g.ok ? hipSuccess : hipErrorInvalidDevice;
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) s->n_cus = prop.multiProcessorCount;
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_nodes, s->nb_nodes);
    if (e == hipSuccess && s->nb_leaf) e = hipMalloc((void**)&s->d_leaf_ids, s->nb_leaf);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_tris, s->nb_tris);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_nodes, nodes, s->nb_nodes, hipMemcpyDefault, st);
    if (e == hipSuccess && s->nb_leaf) e = hipMemcpyAsync(s->d_leaf_ids, leaf_ids, s->nb_leaf, hipMemcpyDefault, st);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_tris, tris, s->nb_tris, hipMemcpyDefault, st);
    const hipError_t se = hipStreamSynchronize(st);   // the caller may free its arrays when this returns
    if (e == hipSuccess) e = se;
    if (e != hipSuccess) {
        cleanup();
        return fail(GIGL_ERR_HIP, std::string("scene from device blob: ") + hipGetErrorString(e));
    }
    if (scene_device_resources(s) != GIGL_OK) { cleanup(); return GIGL_ERR_HIP; }
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
    (void)hipHostFree(s->h_one);
    (void)hipFree(s->one_ws);
    (void)hipFree(s->photon_ws);
    if (s->photon_h) (void)hipHostFree(s->photon_h);
    if (s->one_stream) (void)hipStreamDestroy(s->one_stream);
    if (s->counted) gd::scene_count(s->device, -1);   // the last scene on a device returns the pool's idle memory
    delete s;
    return GIGL_OK;
}

int gigl_trim_pool(int device) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(GIGL_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= n_dev) return fail(GIGL_ERR_INVALID, "device index out of range");
    DeviceGuard g(device);
    HIP_TRY(gd::pool_trim(device));
    return GIGL_OK;
}

int gigl_debug_pool_allocs(uint64_t* count) {
    if (!count) return fail(GIGL_ERR_INVALID, "count is NULL");
    *count = gd::g_pool_allocs.load(std::memory_order_relaxed);
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
    if (s->replica) return fail(GIGL_ERR_INVALID, "a scene adopted from device arrays has no host tree to export");
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

// Host-pointer batches.  A batch of more than one chunk is pipelined over three buffer slots: chunk k's
// rays go up on one stream (in order, so the first chunk has the whole link) while chunk k-1 is traversed
// and its records come down on one of two alternating streams, so that with page-locked host memory the
// call runs near the rate of the slower link direction (32 B per ray up) instead of the sum of both copies
// and the traversal.  (Records on a stream of their own measured 13 % slower: the runtime maps streams
// onto few hardware queues.)  Pageable memory is staged by the runtime inside hipMemcpyAsync, which then
// blocks the caller: copies serialise, and only the traversal overlaps (chunk k-1's records are fetched
// after chunk k's launch for that reason).  Chunks are multiples of 64 rays, so any-hit chunks own whole
// 64-bit words of the output.  Buffers come from the library's pool, which keeps them between calls.
#ifndef GIGL_HOST_PIPELINE
#define GIGL_HOST_PIPELINE 1
#endif
static constexpr uint64_t kHostChunk = 1ull << 21;
static constexpr int kHostSlots = 3;

namespace {
struct HostPipe {
    hipStream_t s_in = nullptr, s_cmp[2] = {nullptr, nullptr};
    hipEvent_t ev_in[kHostSlots] = {}, ev_k[kHostSlots] = {}, ev_out[kHostSlots] = {};
    void* d_o[kHostSlots] = {};
    void* d_d[kHostSlots] = {};
    void* d_out[kHostSlots] = {};
    ~HostPipe() {
        for (hipStream_t st : {s_cmp[0], s_cmp[1], s_in})
            if (st) (void)hipStreamSynchronize(st);
        for (int k = 0; k < kHostSlots; ++k) {
            if (d_o[k]) (void)hipFreeAsync(d_o[k], s_in);
            if (d_d[k]) (void)hipFreeAsync(d_d[k], s_in);
            if (d_out[k]) (void)hipFreeAsync(d_out[k], s_in);
            for (hipEvent_t e : {ev_in[k], ev_k[k], ev_out[k]})
                if (e) (void)hipEventDestroy(e);
        }
        if (s_in) (void)hipStreamSynchronize(s_in);
        for (hipStream_t st : {s_cmp[0], s_cmp[1], s_in})
            if (st) (void)hipStreamDestroy(st);
    }
};
}  // namespace

// A batch of one chunk: copy, trace, copy on one stream, buffers from the pool.  Nothing to overlap, and no
// streams or events to create per call (they cost 0.15 ms, more than a small batch takes): the stream is the
// calling thread's own for the device, created on first use and kept.  (Not the null stream: it would order the
// call against every blocking stream of the process, serialise calls from two host threads, and break a stream
// capture in progress on another thread.)
namespace {
// Never destroyed: a thread's exit (the main thread's: process exit) may come after the HIP runtime has shut down.
struct ThreadStreams {
    hipStream_t st[kMaxDevices] = {nullptr};
};
hipError_t thread_stream(int device, hipStream_t* out) {
    static thread_local ThreadStreams ts;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!ts.st[device]) {
        const hipError_t e = hipStreamCreateWithFlags(&ts.st[device], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = ts.st[device];
    return hipSuccess;
}
}  // namespace

static int trace_host_small(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n, void* out, bool any) {
    const size_t out_bytes = any ? (size_t)((n + 63) / 64 * 8) : (size_t)n * sizeof(gigl_hit);
    hipStream_t st = nullptr;
    hipError_t e = thread_stream(s->device, &st);
    if (e != hipSuccess) return fail(GIGL_ERR_HIP, std::string("host-pointer batch: ") + hipGetErrorString(e));
    void* d[3] = {nullptr, nullptr, nullptr};
    auto release = [&]() {
        for (void* p : d)
            if (p) (void)hipFreeAsync(p, st);
    };
    e = gd::pool_alloc(s->device, &d[0], n * 16, st);
    if (e == hipSuccess) e = gd::pool_alloc(s->device, &d[1], n * 16, st);
    if (e == hipSuccess) e = gd::pool_alloc(s->device, &d[2], out_bytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d[0], o_tmin, n * 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d[1], d_tmax, n * 16, hipMemcpyHostToDevice, st);
    int rc = GIGL_OK;
    if (e == hipSuccess)
        rc = any ? gigl_trace_any_device(s, d[0], d[1], n, (uint64_t*)d[2], st)
                 : gigl_trace_closest_device(s, d[0], d[1], n, (gigl_hit*)d[2], st);
    if (e == hipSuccess && rc == GIGL_OK) e = hipMemcpyAsync(out, d[2], out_bytes, hipMemcpyDeviceToHost, st);
    const hipError_t se = hipStreamSynchronize(st);   // also on failure: nothing may still use the buffers
    release();
    if (rc != GIGL_OK) return rc;
    if (e == hipSuccess) e = se;
    if (e != hipSuccess) return fail(GIGL_ERR_HIP, std::string("host-pointer batch: ") + hipGetErrorString(e));
    return GIGL_OK;
}

static int trace_host(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n, void* out, bool any) {
    int rc = check_batch(s, o_tmin, d_tmax, n, out);
    if (rc != GIGL_OK || n == 0) return rc;
    DeviceGuard g(s->device);
    const bool piped = GIGL_HOST_PIPELINE && n > kHostChunk;
    if (!piped) return trace_host_small(s, o_tmin, d_tmax, n, out, any);
    const uint64_t chunk = kHostChunk;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const int S = (int)std::min<uint64_t>(kHostSlots, n_chunks);
    auto out_bytes = [&](uint64_t m) { return any ? (size_t)((m + 63) / 64 * 8) : (size_t)m * sizeof(gigl_hit); };
    HostPipe hp;
#define HOST_TRY(expr)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(GIGL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    HOST_TRY(hipStreamCreateWithFlags(&hp.s_in, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) HOST_TRY(hipStreamCreateWithFlags(&hp.s_cmp[k], hipStreamNonBlocking));
    for (int k = 0; k < S; ++k) {
        HOST_TRY(hipEventCreateWithFlags(&hp.ev_in[k], hipEventDisableTiming));
        HOST_TRY(hipEventCreateWithFlags(&hp.ev_k[k], hipEventDisableTiming));
        HOST_TRY(hipEventCreateWithFlags(&hp.ev_out[k], hipEventDisableTiming));
        HOST_TRY(gd::pool_alloc(s->device, &hp.d_o[k], chunk * 16, hp.s_in));
        HOST_TRY(gd::pool_alloc(s->device, &hp.d_d[k], chunk * 16, hp.s_in));
        HOST_TRY(gd::pool_alloc(s->device, &hp.d_out[k], out_bytes(chunk), hp.s_in));
    }
    // iteration k: rays of chunk k up, chunk k launched, records of chunk k-1 down
    for (uint64_t k = 0; k <= n_chunks; ++k) {
        if (k < n_chunks) {
            const int slot = (int)(k % (uint64_t)S);
            const uint64_t lo = k * chunk, m = std::min(chunk, n - lo);
            if (k >= (uint64_t)S) HOST_TRY(hipStreamWaitEvent(hp.s_in, hp.ev_k[slot], 0));   // chunk k-S has been traversed
            HOST_TRY(hipMemcpyAsync(hp.d_o[slot], o_tmin + lo * 4, m * 16, hipMemcpyHostToDevice, hp.s_in));
            HOST_TRY(hipMemcpyAsync(hp.d_d[slot], d_tmax + lo * 4, m * 16, hipMemcpyHostToDevice, hp.s_in));
            HOST_TRY(hipEventRecord(hp.ev_in[slot], hp.s_in));
            hipStream_t c = hp.s_cmp[k & 1];
            HOST_TRY(hipStreamWaitEvent(c, hp.ev_in[slot], 0));
            if (k >= (uint64_t)S) HOST_TRY(hipStreamWaitEvent(c, hp.ev_out[slot], 0));       // chunk k-S's records have left
            rc = any ? gigl_trace_any_device(s, hp.d_o[slot], hp.d_d[slot], m, (uint64_t*)hp.d_out[slot], c)
                     : gigl_trace_closest_device(s, hp.d_o[slot], hp.d_d[slot], m, (gigl_hit*)hp.d_out[slot], c);
            if (rc != GIGL_OK) return rc;
            HOST_TRY(hipEventRecord(hp.ev_k[slot], c));
        }
        if (k >= 1) {
            const int slot = (int)((k - 1) % (uint64_t)S);
            const uint64_t lo = (k - 1) * chunk, m = std::min(chunk, n - lo);
            char* dst = (char*)out + (any ? (size_t)(lo / 64 * 8) : (size_t)lo * sizeof(gigl_hit));
            hipStream_t c = hp.s_cmp[(k - 1) & 1];   // behind the chunk's traversal, in stream order
            HOST_TRY(hipMemcpyAsync(dst, hp.d_out[slot], out_bytes(m), hipMemcpyDeviceToHost, c));
            HOST_TRY(hipEventRecord(hp.ev_out[slot], c));
        }
    }
    for (hipStream_t c : {hp.s_cmp[0], hp.s_cmp[1]})
        if (c) HOST_TRY(hipStreamSynchronize(c));
#undef HOST_TRY
    return GIGL_OK;   // ~HostPipe drains the other streams and returns the buffers to the pool
}

int gigl_trace_closest(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n, gigl_hit* hits) {
    return trace_host(s, o_tmin, d_tmax, n, hits, false);
}

int gigl_trace_any(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n,
                   uint64_t* occluded_bits) {
    return trace_host(s, o_tmin, d_tmax, n, occluded_bits, true);
}

int gigl_trace_closest_noalloc_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                                      gigl_hit* hits, void* stream) {
    int rc = check_batch(s, o_tmin, d_tmax, n, hits);
    if (rc != GIGL_OK || n == 0) return rc;
    if (n > GIGL_NOALLOC_MAX_RAYS) return fail(GIGL_ERR_TOO_LARGE, "gigl_trace_closest_noalloc_device: at most 2^24 rays per call");
    if (!s->scratch) return fail(GIGL_ERR_HIP, "no scratch array on this device");
    DeviceGuard g(s->device);
    TraceParams P = {};
    P.n = (unsigned)n;
    P.o_tmin = (const float4*)o_tmin;
    P.d_tmax = (const float4*)d_tmax;
    P.hits = (uint4*)hits;
    P.reset_counter = 1u;   // the scratch array starts at zero and every launch leaves its counter words so: no memset
    const ExtWorkspace ext = {s->scratch, kNoAllocBlocks, false};
    return launch_trace<false, 0>(s, P, (hipStream_t)stream, &ext);
}

int gigl_debug_launch_info(const gigl_scene* s, int* blocks_per_cu, int* n_cus) {
    if (!s || !blocks_per_cu || !n_cus) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene");
    DeviceGuard g(s->device);
    auto kern = s->triples ? trace_kernel<true, false, 0, GIGL_LDS_STACK, kBlock> : trace_kernel<false, false, 0, GIGL_LDS_STACK, kBlock>;
    int q = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, kBlock, 0));
    *blocks_per_cu = q;
    *n_cus = s->n_cus;
    return GIGL_OK;
}

int gigl_scene_layout(const gigl_scene* s, int* layout_out) {
    if (!s || !layout_out) return fail(GIGL_ERR_INVALID, "NULL argument");
    *layout_out = s->layout;
    return GIGL_OK;
}

int gigl_trace_one(const gigl_scene* s, const float o[3], const float d[3], float t_min, float t_max,
                   int is_vis_ray, int* hit_out, gigl_hit* hit_rec, float normal_out[3], float range_out[2]) {
    if (!s || !o || !d || !hit_out) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (s->device == GIGL_DEVICE_NONE) return fail(GIGL_ERR_NO_DEVICE, "host-only scene: no device to trace on");
    // Scene::trace semantics on one ray: a closest-hit launch answers both modes (the bool of a
    // visibility ray equals the bool of the closest-hit query; only the early-out differs, so for a
    // visibility ray the record and the normal are those of the CLOSEST hit, where KdTree.hpp:328
    // would leave those of the first hit found — no reference caller passes true and reads them).
    // One single-wave launch reads the ray from pinned host memory and writes the record, the range
    // and the hit triangle's record back to it: no copy, no allocation, one synchronisation.
    gigl_scene* ms = const_cast<gigl_scene*>(s);
    std::lock_guard<std::mutex> lk(ms->one_mu);
    DeviceGuard g(s->device);
    float* h = reinterpret_cast<float*>(s->h_one);
    h[0] = o[0]; h[1] = o[1]; h[2] = o[2]; h[3] = t_min;
    h[4] = d[0]; h[5] = d[1]; h[6] = d[2]; h[7] = t_max;
    TraceParams P = {};
    P.n = 1;
    P.o_tmin = s->d_one;
    P.d_tmax = s->d_one + 1;
    P.hits = (uint4*)(s->d_one + 2);
    P.ranges = (float2*)(s->d_one + 3);
    P.hit_tri = s->d_one + 4;
    P.reset_counter = 1u;
    Options opt;
    {
        std::lock_guard<std::mutex> lk2(g_opt_mu);
        opt = g_opt;
    }
    P.nodes = s->d_nodes; P.root = s->root;
    P.recs = reinterpret_cast<const uint4*>(s->d_nodes); P.root_rec = s->root_rec;
    P.leaf_ids = s->d_leaf_ids ? s->d_leaf_ids : reinterpret_cast<const unsigned*>(s->d_nodes);
    P.tris = s->d_tris;
    P.blocks_off8 = s->blocks_off8;
    P.box = s->box;
    P.refill_thresh = 1; P.chunk = 64; P.t_exit = 1;
    opt.guided = 0; opt.stats = 0;
    const ExtWorkspace ext = {s->one_ws, 1u, false};
    int rc = launch_trace_t<false, 0, 8, 64>(s, P, opt, s->one_stream, &ext);
    if (rc != GIGL_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->one_stream));
    gigl_hit hrec;
    memcpy(&hrec, h + 8, 16);
    if (range_out) { range_out[0] = h[12]; range_out[1] = h[13]; }
    *hit_out = hrec.tri_id != GIGL_MISS;
    if (hit_rec) *hit_rec = hrec;
    if (normal_out) {
        normal_out[0] = normal_out[1] = normal_out[2] = 0.0f;
        if (*hit_out) {
            // cross(edge0, edge1) (RTBase.cpp:170); normalized for non-visibility rays (Scene.cpp:238)
            const float* rec = h + 16;
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

static int shadow_batch(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                        const gigl_hit* prim_hits, uint64_t n_pts, const float* lights, uint32_t n_lights,
                        uint64_t* occluded_bits, const gigl_fog* fog, float* transm, void* stream, bool fused) {
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
    P.n_pts = (unsigned)n_pts;   // launch_trace keeps it only where the light-major order applies
    P.bits = (unsigned long long*)occluded_bits;
    P.tris = s->d_tris;
    // the per-point half of the pairs, computed once per point (worth it from a few lights per point on)
    void* cache = nullptr;
    if (n_lights >= 4u) {
        HIP_TRY(gd::pool_alloc(s->device, &cache, (size_t)n_pts * 2 * sizeof(float4), (hipStream_t)stream));
        hipLaunchKernelGGL(shadow_points_kernel, dim3(stream_grid(s, n_pts)), dim3(256), 0, (hipStream_t)stream, P,
                           (unsigned)n_pts, (float4*)cache);
        P.pt_cache = (const float4*)cache;
    }
    struct FreeCache {   // released in stream order behind the launches that read it
        void* p; hipStream_t st;
        ~FreeCache() { if (p) (void)hipFreeAsync(p, st); }
    } free_cache{cache, (hipStream_t)stream};
    if (occluded_bits && transm && fused) {   // visibility first, then the march of the visible pairs only
        int rc = launch_trace<true, 2>(s, P, (hipStream_t)stream);
        if (rc != GIGL_OK) return rc;
        P.tris = s->d_tris;
        const uint64_t waves = (n + kVisPairsPerWave - 1) / kVisPairsPerWave;
        const uint64_t cap = (uint64_t)s->n_cus * 8;
        const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>((waves + 3) / 4, 1), cap);
        hipLaunchKernelGGL(transm_visible_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, fog->fp, P,
                           (const unsigned long long*)occluded_bits, transm);
        HIP_TRY(hipGetLastError());
        return GIGL_OK;
    }
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

int gigl_shadow_batch_device(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                             const gigl_hit* prim_hits, uint64_t n_pts, const float* lights, uint32_t n_lights,
                             uint64_t* occluded_bits, const gigl_fog* fog, float* transm, void* stream) {
    return shadow_batch(s, prim_o_tmin, prim_d_tmax, prim_hits, n_pts, lights, n_lights, occluded_bits, fog, transm,
                        stream, false);
}

int gigl_shadow_batch_fused_device(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                                   const gigl_hit* prim_hits, uint64_t n_pts, const float* lights, uint32_t n_lights,
                                   uint64_t* occluded_bits, const gigl_fog* fog, float* transm, void* stream) {
    if (!occluded_bits || !transm || !fog) return fail(GIGL_ERR_INVALID, "the fused shadow batch writes bits and transmittance: both outputs and the fog are required");
    return shadow_batch(s, prim_o_tmin, prim_d_tmax, prim_hits, n_pts, lights, n_lights, occluded_bits, fog, transm,
                        stream, true);
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
    P.n_pts = (unsigned)n_pts;
    P.bits = (unsigned long long*)occluded_bits;
    return launch_trace<true, 3>(s, P, (hipStream_t)stream);
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

