// gigl_shade.hip — the software shading stage and its C-ABI entry points:
// Surface.frag main() (Source/Shaders/Surface.frag:290-343) and
// Combine.frag's normalisation + tone mapping (:72-76) as compute kernels over the hit buffers,
// occlusion bits and transmittance the ray batches produce.  Stand-ins for the GL pipeline are
// listed in DESIGN.md "Shading stage"; the arithmetic mirrors oracle/gigl_oracle.cpp step by step.
// ------------------------------------------------------------------------------------------
#include "gigl_internal.h"

using namespace gd;

namespace {

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
            const float* m = S.materials + 12 * (size_t)__float_as_uint(tc.z);   // word 10 of the record: material id
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

}  // namespace

extern "C" {

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

}  // extern "C"
