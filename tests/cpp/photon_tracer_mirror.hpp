// photon_tracer_mirror.hpp — TEST CODE: rt::PhotonTracer::trace (Source/RT/PhotonTracer.cpp:78-188) as the
// reference's own host would keep it, statement for statement, with GLM removed and the random numbers injected,
// every scene.trace / traceFog / sampleExtK answered by the engine through gigl_amd/host/gigl_host.hpp's Scene.
// It demonstrates "caller unchanged"  and gives tests/cpp/vpl_plumbing.cpp its serial walk;
// the product package does not carry it.
#pragma once
#include <functional>

#include "../../gigl_amd/host/gigl_host.hpp"

namespace gigl_host {
namespace rt {

struct mat3 { vec3 c0, c1, c2; };
inline vec3 mul(const mat3& m, vec3 v) {  // Source/Include/GLM/detail/type_mat3x3.inl:593-596
    return v3(m.c0.x * v.x + m.c1.x * v.y + m.c2.x * v.z, m.c0.y * v.x + m.c1.y * v.y + m.c2.y * v.z,
              m.c0.z * v.x + m.c1.z * v.y + m.c2.z * v.z);
}
inline mat3 spanTangentSpace(vec3 normal) {  // Source/RT/RTBase.h:33-41
    const vec3 c1 = cross(normal, v3(0.0f, 0.0f, 1.0f)), c2 = cross(normal, v3(0.0f, 1.0f, 0.0f));
    const vec3 c3 = length(c1) > length(c2) ? c1 : c2;
    const vec3 tangent = normalize(-c3);
    return mat3{tangent, cross(normal, tangent), normal};
}
inline vec3 genVecSinCos(float phi, float sin_the, float cos_the) {  // RTBase.h:44-50
    return v3(sin_the * std::cos(phi), sin_the * std::sin(phi), cos_the);
}

// rt::PhotonTracer::trace, Source/RT/PhotonTracer.cpp:78-188.  `rng` replaces UnitRNG::generate.
struct SerialPhotonTracer {
    static void trace(const Scene& scene, vec3 source_pos, vec3 source_intensity, vec3 /*shoot_dir*/,
                      int max_vpl_count, LightArray& la, const std::function<float()>& rng,
                      uint32_t* n_paths_out = nullptr) {
        la.clear();
        uint32_t n_paths = 0;
        do {
            ++n_paths;
            if (source_pos.z <= MAX_FOG_HEIGHT) (void)rng();              // :86 (result unused, :88-92)
            const float u1 = rng(), u0 = rng();                           // :90, right-to-left evaluation
            const float sin_the0 = std::sqrt(1.0f - u1 * u1);
            const vec3 dir = genVecSinCos(TWO_PI * u0, sin_the0, u1);
            const float pdf = 0.5f * INV_PI;
            Ray ray{source_pos, dir};
            const float ia = 1.0f / pdf;
            vec3 attenuation = v3(ia, ia, ia);
            for (int n_bounces = 1; n_bounces <= N_GI_BOUNCES; ++n_bounces) {
                const bool hit = scene.trace(ray);
                const IntDist fog_hit = scene.traceFog(ray);
                if (fog_hit) {
                    const float f_min = fog_hit.entr > 0.0f ? fog_hit.entr : 0.0f;
                    const float f_max = fog_hit.exit < ray.inters.distance ? fog_hit.exit : ray.inters.distance;
                    ray.setValidRange(f_min, f_max);
                    float t_event = 0.0f;
                    bool event = false;
                    const float maj = scene.getMajExtK();
                    if (0.0f != maj) {                                    // calcEventDistWT, :62-76
                        float d_event = ray.t_min, p_event = 0.0f;
                        while (d_event < ray.t_max && p_event < rng()) {
                            const float dt = -std::log(1.0f - rng()) / maj;
                            d_event += dt;
                            p_event = scene.sampleExtK(ray.getPtAtDist(d_event)) / maj;
                        }
                        event = d_event < ray.t_max;
                        t_event = d_event;
                    }
                    if (event) {
                        const vec3 event_pt = ray.getPtAtDist(t_event);
                        const float sca_k = scene.sampleScaK(event_pt);
                        if (rng() < scene.getScaAlbedo()) {
                            const vec3 I = ray.d;
                            VPL v;
                            std::memset(&v, 0, sizeof(v));
                            v.w_pos = event_pt; v.type = 1; v.intensity = source_intensity * attenuation;
                            v.sca_k = sca_k; v.w_inc = I; v.path_id = n_paths - 1;
                            la.addLight(v);
                            if (la.size() == max_vpl_count) break;
                            const float phi = TWO_PI * rng();             // HGPhaseFunc::importanceSample, :49-58
                            const float uu = rng();
                            const float nn = 1.0f - HG_G * HG_G, dd = 1.0f - HG_G + 2.0f * HG_G * uu;
                            const float q = nn / dd;
                            const float cos_the = 0.5f / std::fabs(HG_G) * (1.0f + HG_G * HG_G - q * q);
                            const float sin_the = std::sqrt(1.0f - cos_the * cos_the);
                            const vec3 O = mul(spanTangentSpace(I), genVecSinCos(phi, sin_the, cos_the));
                            attenuation = attenuation * 1.0f;
                            ray = Ray{event_pt, O};
                            continue;
                        } else break;
                    }
                }
                if (hit) {
                    const vec3 hit_pt = ray.getPtAtDist(ray.inters.distance);
                    const vec3 norm = ray.inters.normal;
                    if (dot(ray.d, norm) >= 0.0f || hit_pt.z > MAX_FOG_HEIGHT) break;
                    const PhongMaterial* mat = ray.inters.material;
                    if (mat->isEmissive()) break;
                    if (rng() < SURVIVAL_P_RR) {
                        attenuation = attenuation / SURVIVAL_P_RR;
                        const vec3 I = -ray.d;
                        VPL v;
                        std::memset(&v, 0, sizeof(v));
                        v.w_pos = hit_pt + 1E-4f * norm; v.type = 2; v.intensity = source_intensity * attenuation;
                        v.w_inc = I; v.path_id = n_paths - 1; v.w_norm = norm;
                        v.k_d = mat->k_d; v.k_s = mat->k_s; v.n_s = mat->n_s;
                        la.addLight(v);
                        if (la.size() == max_vpl_count) break;
                        const float bu1 = rng(), bu2 = rng();             // importanceSampleBRDF, RTBase.cpp:70-109
                        const vec3 lum = v3(0.2126f, 0.7152f, 0.0722f);
                        const float dw = dot(mat->k_d, lum), gw = dot(mat->k_s, lum);
                        const float sp_d = dw / (dw + gw);
                        const float u_rand = rng();
                        vec3 O, throughput;
                        if (u_rand < sp_d) {
                            const float sin_t = std::sqrt(1.0f - bu2);
                            const vec3 micro = genVecSinCos(TWO_PI * bu1, sin_t, std::sqrt(bu2));
                            O = mul(spanTangentSpace(norm), micro);
                            throughput = (mat->k_d * PI) / (micro.z * sp_d);
                        } else {
                            const vec3 S = -I + (2.0f * dot(I, norm)) * norm;
                            const float cos_t = std::pow(bu2, 1.0f / (mat->n_s + 1.0f));
                            const float sin_t = std::sqrt(1.0f - cos_t * cos_t);
                            O = mul(spanTangentSpace(S), genVecSinCos(TWO_PI * bu1, sin_t, cos_t));
                            throughput = (mat->k_s * TWO_PI) / ((mat->n_s + 1.0f) * (1.0f - sp_d));
                        }
                        attenuation = attenuation * (throughput * dot(O, norm));
                        ray = Ray{hit_pt, O, norm};
                        continue;
                    } else break;
                } else break;
            }
        } while (la.size() < max_vpl_count && n_paths < (uint32_t)(100 * max_vpl_count));
        if (n_paths_out) *n_paths_out = n_paths;
        if (la.size() < max_vpl_count) la.clear();                        // :181-184
        else la.normalizeIntensity((int)n_paths);
    }
};

}  // namespace rt
}  // namespace gigl_host
