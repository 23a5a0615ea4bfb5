// gigl_host.hpp — C++ host-side mirror of the reference's RT interface over the C ABI.
//
// Header-only.  Same names, argument meaning and return conventions as the reference so that
// its callers read unchanged:
//   rt::Ray                 /root/reference/Source/RT/RTBase.h:81-105, RTBase.cpp:111-133
//   BBox::IntDist           Source/Common/BBox.h:37-43
//   Scene::trace / traceFog / sampleExtK / sampleScaK / getMajExtK / getScaAlbedo
//                           Source/Common/Scene.h:33-47, Scene.cpp:209-251
//   VPL (96 B, std140)      Source/VPL/PointLight.h:47-78, PointLight.cpp:15-22
//   LightArray<VPL>         Source/VPL/LightArray.hpp:41-66
//   rt::PhotonTracer::traceBatch   the VPL set as ray batches (replaces the call at Source/Common/Renderer.cpp:223)
// Every ray query and every density sample goes to the GPU through include/gigl_rt.h; nothing
// here intersects or samples on the CPU.  (The serial PhotonTracer::trace walk a host keeps on top of
// Scene::trace is the reference's own function; a compiling copy lives with the tests, tests/cpp/.)
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gigl_rt.h"

namespace gigl_host {

struct vec3 {
    float x, y, z;
};
inline vec3 v3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// GLM 0.9.5.4 operation order (Source/Include/GLM/detail/func_geometric.inl:66-73, :224-227, :263-264)
inline float dot(vec3 a, vec3 b) { vec3 t = a * b; return t.x + t.y + t.z; }
inline vec3 cross(vec3 x, vec3 y) { return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline float length(vec3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z)); }

// Source/Common/Constants.h
constexpr float TWO_PI = 6.28318548f, PI = 3.14159274f, INV_PI = 0.318309873f;
constexpr float RAY_OFFSET = 1E-4f, SURVIVAL_P_RR = 0.95f, MAX_FOG_HEIGHT = 548.8f, HG_G = 0.25f;
constexpr int N_GI_BOUNCES = 3;

struct GiglError : std::runtime_error {
    int code;
    GiglError(int c, const char* m) : std::runtime_error(std::string("gigl_rt: ") + m), code(c) {}
};
inline void check(int rc) {
    if (rc != GIGL_OK) throw GiglError(rc, gigl_last_error());
}

// rt::PhongMaterial, Source/RT/RTBase.h:53-78 (48 B)
struct PhongMaterial {
    vec3 k_d; float pad; vec3 k_s; float n_s; vec3 k_e; float slack;
    bool isEmissive() const { return k_e.x > 0.0f || k_e.y > 0.0f || k_e.z > 0.0f; }
};

namespace rt {
struct Ray {
    struct Intersection {
        const PhongMaterial* material = nullptr;
        float distance = FLT_MAX;
        vec3 normal = {0.0f, 0.0f, 0.0f};
    } inters;
    vec3 o, d, inv_d;
    float t_min, t_max;
    Ray(vec3 origin, vec3 direction)
        : o(origin), d(direction), inv_d{1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z}, t_min(0.0f),
          t_max(FLT_MAX) {}
    Ray(vec3 origin, vec3 direction, vec3 normal) : Ray(origin + RAY_OFFSET * normal, direction) {}
    vec3 getPtAtDist(float t) const { return o + t * d; }
    void setValidRange(float mn, float mx) { t_min = mn; t_max = mx; }
};
}  // namespace rt

struct IntDist {  // BBox::IntDist
    float entr = FLT_MAX, exit = -FLT_MAX;
    explicit operator bool() const { return entr < FLT_MAX; }
};

// VPL, Source/VPL/PointLight.h:66-77
struct VPL {
    vec3 w_pos; uint32_t type; vec3 intensity; float sca_k; vec3 w_inc; uint32_t path_id;
    vec3 w_norm; uint32_t pad1; vec3 k_d; uint32_t pad2; vec3 k_s; float n_s;
};
static_assert(sizeof(VPL) == 96, "std140 VPL record");

struct LightArray {  // LightArray<VPL>, Source/VPL/LightArray.hpp
    std::vector<VPL> pls;
    int size() const { return (int)pls.size(); }
    void clear() { pls.clear(); }
    bool isEmpty() const { return pls.empty(); }
    void addLight(const VPL& v) { pls.push_back(v); }
    void normalizeIntensity(int n_paths) {  // :46-66
        const float inv = 1.0f / n_paths;
        uint32_t old_id = pls[0].path_id, new_id = 0;
        for (VPL& p : pls) {
            if (old_id == p.path_id) p.path_id = new_id;
            else { old_id = p.path_id; p.path_id = ++new_id; }
            p.intensity = p.intensity * inv;
        }
    }
};

class Scene {
public:
    Scene() = default;
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    ~Scene() {
        if (m_fog) gigl_fog_destroy(m_fog);
        if (m_scene) gigl_scene_destroy(m_scene);
    }
    // RT half of Scene::loadObjects (Source/Common/Scene.cpp:136-158) from already loaded arrays
    void setGeometry(const std::vector<float>& verts, const std::vector<uint32_t>& tris,
                     const std::vector<PhongMaterial>& materials, int device = 0) {
        m_tris = tris;
        m_materials = materials;
        check(gigl_scene_create(verts.data(), (uint32_t)(verts.size() / 3), tris.data(), (uint32_t)(tris.size() / 4),
                                10, 1, 30, 2, 0.25f, device, &m_scene));
        float b[6];
        check(gigl_scene_bbox(m_scene, b));
        m_box_min = v3(b[0], b[1], b[2]);
        m_box_max = v3(b[3], b[4], b[5]);
        m_device = device;
    }
    // Scene::addFog (Scene.cpp:164-192) with an already loaded density volume
    void addFog(const std::vector<uint8_t>& dens, const int res[3], float maj_ext_k, float abs_k, float sca_k) {
        vec3 mx = m_box_max;
        mx.y = std::min(mx.y, MAX_FOG_HEIGHT);
        m_fog_min = m_box_min;
        m_fog_max = mx;
        const float box[6] = {m_fog_min.x, m_fog_min.y, m_fog_min.z, mx.x, mx.y, mx.z};
        check(gigl_fog_create(dens.data(), res, box, abs_k + sca_k, 8, m_device, &m_fog));
        m_maj_ext_k = maj_ext_k; m_abs_k = abs_k; m_sca_k = sca_k;
        m_fog_enabled = true;
    }
    void toggleFog() { m_fog_enabled = !m_fog_enabled; }
    // Scene::trace, Scene.cpp:234-243.  Like the reference it leaves the ray's t_min/t_max as
    // KdTree::intersect left them (KdTree.hpp:273-274, :310, :337-338): callers read them.
    bool trace(rt::Ray& ray, bool is_vis_ray = false) const {
        int hit = 0;
        gigl_hit h;
        float n[3], range[2];
        const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
        check(gigl_trace_one(m_scene, o, d, ray.t_min, ray.t_max, is_vis_ray ? 1 : 0, &hit, &h, n, range));
        ray.t_min = range[0];
        ray.t_max = range[1];
        m_last_hit = h;
        if (hit) {
            ray.inters.distance = h.t;
            ray.inters.normal = v3(n[0], n[1], n[2]);
            ray.inters.material = &m_materials[m_tris[4 * (size_t)h.tri_id + 3]];
        }
        return hit != 0;
    }
    // Scene::traceFog, Scene.cpp:245-251: the slab test of the fog box (BBox.cpp:81-109).  A few
    // float operations on one ray: host arithmetic of the caller, not a batch.
    IntDist traceFog(const rt::Ray& ray) const {
        const float t_min = ray.t_min, t_max = ray.t_max;
        IntDist r;
        if (!m_fog_enabled) return r;
        const vec3 b[2] = {m_fog_min, m_fog_max};
        const int sx = ray.inv_d.x >= 0.0f, sy = ray.inv_d.y >= 0.0f, sz = ray.inv_d.z >= 0.0f;
        float t_entr = (b[1 - sx].x - ray.o.x) * ray.inv_d.x, t_exit = (b[sx].x - ray.o.x) * ray.inv_d.x;
        const float t_y_min = (b[1 - sy].y - ray.o.y) * ray.inv_d.y, t_y_max = (b[sy].y - ray.o.y) * ray.inv_d.y;
        if ((t_entr > t_y_max) || (t_y_min > t_exit)) return r;
        t_entr = t_entr > t_y_min ? t_entr : t_y_min;
        t_exit = t_exit < t_y_max ? t_exit : t_y_max;
        const float t_z_min = (b[1 - sz].z - ray.o.z) * ray.inv_d.z, t_z_max = (b[sz].z - ray.o.z) * ray.inv_d.z;
        if ((t_entr > t_z_max) || (t_z_min > t_exit)) return r;
        t_entr = t_entr > t_z_min ? t_entr : t_z_min;
        t_exit = t_exit < t_z_max ? t_exit : t_z_max;
        if (t_entr < t_max && t_exit > t_min) { r.entr = t_entr; r.exit = t_exit; }
        return r;
    }
    float sampleDensity(vec3 pos) const {  // DensityField::sampleDensity, on the device
        const float p[3] = {pos.x, pos.y, pos.z};
        float out = 0.0f;
        check(gigl_sample_density(m_fog, p, 1, &out));
        return out;
    }
    float sampleExtK(vec3 pos) const { return (m_abs_k + m_sca_k) * sampleDensity(pos); }  // FogVolume.cpp:32-36
    float sampleScaK(vec3 pos) const { return m_sca_k * sampleDensity(pos); }              // :38-41
    float getMajExtK() const { return m_maj_ext_k; }
    float getScaAlbedo() const { return m_sca_k / (m_abs_k + m_sca_k); }                   // :43-46
    // rt::PhotonTracer::trace as ray batches on the device (gigl_photon_trace_batch, substream mode): the
    // whole VPL set of one primary light in one call; la receives normalised, renumbered records.
    void tracePhotonsBatch(vec3 source_pos, vec3 source_intensity, vec3 shoot_dir, int max_vpl_count, uint64_t seed,
                           std::vector<uint8_t>& vpl_bytes, int& n_vpls, uint32_t& n_paths) const {
        vpl_bytes.assign((size_t)max_vpl_count * 96, 0);
        check(gigl_photon_trace_batch(m_scene, m_fog, &m_materials[0].k_d.x, (uint32_t)m_materials.size(), m_abs_k,
                                      m_sca_k, m_maj_ext_k, m_fog_enabled ? 1 : 0, &source_pos.x, &source_intensity.x,
                                      &shoot_dir.x, max_vpl_count, seed, 0, vpl_bytes.data(), &n_vpls, &n_paths));
    }
    vec3 boxMin() const { return m_box_min; }
    vec3 boxMax() const { return m_box_max; }
    bool fogEnabled() const { return m_fog_enabled; }
    const gigl_scene* handle() const { return m_scene; }
    const gigl_hit& lastHit() const { return m_last_hit; }

private:
    gigl_scene* m_scene = nullptr;
    gigl_fog* m_fog = nullptr;
    int m_device = 0;
    std::vector<uint32_t> m_tris;
    std::vector<PhongMaterial> m_materials;
    vec3 m_box_min{}, m_box_max{}, m_fog_min{}, m_fog_max{};
    float m_maj_ext_k = 0.0f, m_abs_k = 0.0f, m_sca_k = 0.0f;
    bool m_fog_enabled = false;
    mutable gigl_hit m_last_hit{};
};

namespace rt {

// rt::PhotonTracer as a host keeps calling it once the set is generated on the device.  (A host that keeps the
// reference's own serial PhotonTracer::trace routes its scene.trace / sampleExtK calls through Scene above; a
// compiling copy of that walk, used by the tests only, is tests/cpp/photon_tracer_mirror.hpp.)
struct PhotonTracer {
    // The same set as ray batches on the device, each path drawing from its own substream of `seed`
    // (include/gigl_rt.h: gigl_photon_trace_batch).  Replaces the call at Source/Common/Renderer.cpp:223.
    static void traceBatch(const Scene& scene, vec3 source_pos, vec3 source_intensity, vec3 shoot_dir,
                           int max_vpl_count, LightArray& la, uint64_t seed, uint32_t* n_paths_out = nullptr) {
        la.clear();
        std::vector<uint8_t> bytes;
        int n = 0;
        uint32_t n_paths = 0;
        scene.tracePhotonsBatch(source_pos, source_intensity, shoot_dir, max_vpl_count, seed, bytes, n, n_paths);
        la.pls.resize((size_t)n);
        if (n) std::memcpy(la.pls.data(), bytes.data(), (size_t)n * sizeof(VPL));
        if (n_paths_out) *n_paths_out = n_paths;
    }
};

}  // namespace rt
}  // namespace gigl_host
