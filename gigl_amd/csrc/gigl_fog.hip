// gigl_fog.hip — fog volume kernels and their C-ABI entry points: sampleDensity
// (/root/reference/Source/Fog/DensityField.cpp:177-211), the GLSL calcTransm / rayMarch
// (Source/Shaders/Surface.frag:139-203) over explicit segments, and computePiDensity
// (Source/Fog/DensityField.cpp:271-327).  The shared device functions (fog_filter, calc_transm)
// are in gigl_internal.h.
#include "gigl_internal.h"

using namespace gd;

namespace {

__global__ void __launch_bounds__(256) transm_segments_kernel(FogParams F, const float4* seg_p,
                                                              const float4* seg_d, unsigned n, float* out) {
    GIGL_FOG_LUT_DECL
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 p = seg_p[i], d = seg_d[i];
        out[i] = calc_transm<GIGL_FOG_LUT != 0>(F, p.x, p.y, p.z, d.x, d.y, d.z, d.w, s_unorm);
    }
}

// DensityField::computePiDensity's march (Source/Fog/DensityField.cpp:288-321) for one pixel per
// lane, after the primary closest-hit pass has produced `hits`.  Stores of one z-plane are
// coalesced across the lanes of a wave (x fastest).
__global__ void __launch_bounds__(256) pi_density_kernel(FogParams F, Cam12 cam, unsigned res_x, unsigned n_pix,
                                                         const uint4* __restrict__ hits, float* __restrict__ out) {
    const BoxF fb = {F.mnx, F.mny, F.mnz, F.mxx, F.mxy, F.mxz};
    GIGL_FOG_LUT_DECL
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
                return fog_filter<GIGL_FOG_LUT != 0>(F, (qx - F.mnx) / ivx, (qy - F.mny) / ivy, (qz - F.mnz) / ivz, s_unorm);  // computeNormPos
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

__global__ void __launch_bounds__(256) sample_density_kernel(FogParams F, const float* pos, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        // computeNormPos: (pos - min) / dims, BBox.cpp:53-55
        const float nx = (pos[3 * i] - F.mnx) / (F.mxx - F.mnx);
        const float ny = (pos[3 * i + 1] - F.mny) / (F.mxy - F.mny);
        const float nz = (pos[3 * i + 2] - F.mnz) / (F.mxz - F.mnz);
        out[i] = fog_filter(F, nx, ny, nz);
    }
}

}  // namespace

extern "C" {

int gigl_fog_create(const uint8_t* dens, const int res[3], const float box[6], float ext_k, int n_intervals,
                    int device, gigl_fog** out) {
    if (!out) return fail(GIGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!dens || !res || !box) return fail(GIGL_ERR_INVALID, "NULL argument");
    if (res[0] <= 0 || res[1] <= 0 || res[2] <= 0 || (int64_t)res[0] * res[1] * res[2] > (1ll << 30))
        return fail(GIGL_ERR_INVALID, "bad volume resolution");
    if ((uint64_t)(res[0] + 1ll) * (uint64_t)(res[1] + 1ll) * (uint64_t)(res[2] + 1ll) >= (1ull << 31))
        return fail(GIGL_ERR_TOO_LARGE, "density volume too large for the brick table (2^31 bricks)");
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
    // optional fast path of gigl_sample_density: without it the call falls back to plain copies
    if (hipHostMalloc((void**)&f->h_stage, 4 * gigl_fog::kStagePositions * sizeof(float), hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer((void**)&f->d_stage, f->h_stage, 0) == hipSuccess &&
        hipStreamCreateWithFlags(&f->stage_stream, hipStreamNonBlocking) == hipSuccess) {
    } else {
        if (f->h_stage) (void)hipHostFree(f->h_stage);
        f->h_stage = f->d_stage = nullptr;
        f->stage_stream = nullptr;
        (void)hipGetLastError();
    }
    *out = f;
    return GIGL_OK;
}

int gigl_fog_destroy(gigl_fog* f) {
    if (!f) return GIGL_OK;
    DeviceGuard g(f->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(f->d_bricks);
    if (f->h_stage) (void)hipHostFree(f->h_stage);
    if (f->stage_stream) (void)hipStreamDestroy(f->stage_stream);
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
    HIP_TRY(gd::pool_alloc(s->device, &hits, n * sizeof(gigl_hit), st));
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
    if (n <= gigl_fog::kStagePositions && f->h_stage && f->stage_stream) {
        gigl_fog* mf = const_cast<gigl_fog*>(f);
        std::lock_guard<std::mutex> lk(mf->stage_mu);
        DeviceGuard g(f->device);
        memcpy(f->h_stage, pos, 3 * (size_t)n * sizeof(float));
        hipLaunchKernelGGL(sample_density_kernel, dim3(1), dim3(256), 0, f->stage_stream, f->fp, (const float*)f->d_stage,
                           (unsigned)n, f->d_stage + 3 * gigl_fog::kStagePositions);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(f->stage_stream));
        memcpy(dens_out, f->h_stage + 3 * gigl_fog::kStagePositions, (size_t)n * sizeof(float));
        return GIGL_OK;
    }
    return map_host(f->device, pos, 3 * n, n, dens_out,
                    [](const float* in, unsigned m, float* o, const void* ctx) {
                        hipLaunchKernelGGL(sample_density_kernel, dim3((m + 255) / 256), dim3(256), 0, nullptr,
                                           *(const FogParams*)ctx, in, m, o);
                    },
                    &f->fp);
}

}  // extern "C"
