/*
 * gigl_oracle.h — C interface of the CPU ORACLE (test infrastructure, NOT product).
 *
 * This is a CPU restatement of the reference's ray-query path (zero-radiance/GIGL,
 * Source/RT + Source/Common/BBox + Source/Fog sampling), written from the reference's
 * published algorithm with no GLM and no reference source text.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker.  The product library (gigl_amd/lib/libgigl_rt.so) never links or calls it.
 *
 * PARITY UNPINNED.  The reference cannot be built in this image without stand-ins for
 * OpenGL, Win32 and the MSVC CRT (see DESIGN.md "Oracle"), so no oracle/_ref exists, and the
 * reference holds no tests, golden vectors or fixtures for this path.  The only outside
 * evidence is the set of facts SURVEY.md Appendix B recorded from a survey-time build of the
 * reference that itself used stand-in headers (tests/golden/survey_facts.json; reproduced bit
 * for bit by tests/test_oracle_golden.py): a consistency check, not a pin.  Sub-paths those
 * facts do not even touch say so again at their definition.
 *
 * All citations are path:line under /root/reference.
 */
#ifndef GIGL_ORACLE_H
#define GIGL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

/* Source/RT/KdTree.hpp:87-132 (constructor) + :139-264 (buildTree).
 * verts: float[3*n_verts]; tris: uint32[4*n_tris] = (i0,i1,i2,mat_id) per triangle. */
ora_scene* ora_scene_create(const float* verts, uint32_t n_verts, const uint32_t* tris,
                            uint32_t n_tris, int inters_cost, int trav_cost, int max_depth,
                            uint32_t min_num_prims, float empty_bonus);
/* A scene over a tree built elsewhere, given in the reference's node encoding (below): used for
 * scenes too large to build serially here ("the build's own tree as ground truth for
 * large scenes").  Tree box = union of the triangles' boxes (KdTree.hpp:100-106). */
ora_scene* ora_scene_from_tree(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                               const uint32_t* nodes_words, uint32_t n_nodes, const uint32_t* leaf_prim_ids,
                               uint32_t n_leaf_prim_ids);
void ora_scene_destroy(ora_scene* s);

/* Statistics the reference prints (Source/RT/KdTree.hpp:126-131). */
uint32_t ora_scene_n_nodes(const ora_scene* s);
uint32_t ora_scene_n_leaf_prim_ids(const ora_scene* s);
int      ora_scene_depth(const ora_scene* s);
uint32_t ora_scene_leaf_count(const ora_scene* s);
float    ora_scene_avg_prims_per_leaf(const ora_scene* s);
/* Source/RT/KdTree.hpp:135-137; out = min xyz, max xyz. */
void     ora_scene_bbox(const ora_scene* s, float out[6]);
/* Raw tree in the reference's own 8-byte node encoding (Source/RT/KdTree.h:22-65):
 * word0 = split_pos bits | single_prim_id | prims_offset; word1 bit0 = is_leaf,
 * interior: bits1-2 axis, bits3-31 above_child; leaf: bits1-31 n_prims. */
void     ora_scene_copy_nodes(const ora_scene* s, uint32_t* out_words /* 2*n_nodes */);
void     ora_scene_copy_leaf_prim_ids(const ora_scene* s, uint32_t* out);

/* Source/RT/RTBase.h:18-29 */
float ora_rescaled_eps(float t);

/* Source/Common/BBox.cpp:81-109.  box = min xyz, max xyz.  Returns 1 on hit. */
int ora_bbox_intersect(const float box[6], const float o[3], const float d[3], float t_min,
                       float t_max, float* entr, float* exit);

/* Scene::trace (Source/Common/Scene.cpp:234-243) over a batch of FRESH rays
 * (Ray ctor: Source/RT/RTBase.cpp:111-115, then setValidRange(tmin,tmax)).
 * rays_o_tmin / rays_d_tmax: float[4*n] = (ox,oy,oz,tmin) / (dx,dy,dz,tmax).
 * Closest mode writes hits: uint32 tri_id (0xFFFFFFFF miss) + float t,u,v  (16 B/ray).
 * counters (optional, may be NULL): uint32[2*n] = nodes_visited, triangle_tests per ray.
 * n_threads <= 1: serial; otherwise OpenMP schedule(dynamic). */
void ora_trace_closest(const ora_scene* s, const float* rays_o_tmin, const float* rays_d_tmax,
                       uint64_t n, void* hits16, uint32_t* counters, int n_threads);
/* is_vis_ray = true (Source/RT/KdTree.hpp:267,328).  occluded: uint8[n] (0/1). */
void ora_trace_any(const ora_scene* s, const float* rays_o_tmin, const float* rays_d_tmax,
                   uint64_t n, uint8_t* occluded, uint32_t* counters, int n_threads);

/* Scene::trace on one ray: returns the bool; hit16 = record (or a miss record), normal_out = the
 * normal as Scene::trace leaves it (Scene.cpp:238: normalized only for closest-hit rays), range_out =
 * the ray's (t_min, t_max) as KdTree::intersect leaves them (KdTree.hpp:273-274, :310, :337-338). */
int ora_trace_one(const ora_scene* s, const float o[3], const float d[3], float t_min, float t_max,
                  int is_vis_ray, void* hit16, float normal_out[3], float range_out[2]);

/* PerspectiveCamera ctor + getPrimaryRay (Source/Common/Camera.cpp:15-33, :65-68).
 * cam_out: float[12] = pos, bottom_left, step_x, step_y. */
void ora_camera_setup(const float pos[3], const float dir[3], const float up[3], float h_fov,
                      int res_x, int res_y, float cam_out[12]);
/* Rays for pixels [first, first+n) in row-major order, centre (x+0.5, y+0.5)
 * (Source/Fog/DensityField.cpp:288). */
void ora_primary_rays(const float cam[12], int res_x, uint64_t first, uint64_t n,
                      float* rays_o_tmin, float* rays_d_tmax);

/* Shadow-ray construction (the build's definition, SURVEY.md Appendix A "Shadow-ray
 * convention"; DESIGN.md "Shadow batch").  Per (point p, light l):
 *   p = o + t*d of the primary hit, N = normalize(cross(e0,e1)) flipped to face -d,
 *   L = light position, dirv = L - p, dist = length(dirv),
 *   ray = Ray{p, normalize(dirv), N} (origin lifted by RAY_OFFSET*N), range (0, dist).
 * prim_hits16: closest-hit records of the primary rays; points with a miss emit no rays and
 * their occlusion bytes are 0xFF.  lights: float[3*n_lights]. occluded: uint8[n_pts*n_lights].
 * transm (optional): float[n_pts*n_lights] = calcTransm of the same segment (see below),
 * written for every pair with a primary hit (1.0 otherwise). */
typedef struct ora_fog {
    const uint8_t* dens;   /* res[0]*res[1]*res[2] bytes, x fastest (DensityField.cpp:59) */
    int   res[3];
    float box[6];          /* min xyz, max xyz */
    float ext_k;           /* abs_k + sca_k (Source/Fog/FogVolume.cpp:34) */
    int   n_intervals;     /* R_M_INTERVALS = 8 */
} ora_fog;
void ora_shadow_batch(const ora_scene* s, const float* prim_o_tmin, const float* prim_d_tmax,
                      const void* prim_hits16, uint64_t n_pts, const float* lights,
                      uint32_t n_lights, uint8_t* occluded, const ora_fog* fog, float* transm,
                      int n_threads);

/* The same batch in the shader's order (Source/Shaders/Surface.frag:250-256, :270-288): visibility
 * first, calcTransm only for visible lights.  transm = 0 for an occluded pair, calcTransm for a
 * visible one, 1.0 where the primary ray missed.  sums (optional): uint64[4] = pairs traced, nodes
 * visited, triangle tests, pairs marched (exact work counters). */
void ora_shadow_batch_fused(const ora_scene* s, const float* prim_o_tmin, const float* prim_d_tmax,
                            const void* prim_hits16, uint64_t n_pts, const float* lights,
                            uint32_t n_lights, uint8_t* occluded, const ora_fog* fog, float* transm,
                            uint64_t* sums, int n_threads);
/* omp_get_max_threads(): what the cpu_baseline leg of bench.py reports beside its thread count. */
int ora_max_threads(void);

/* DensityField::sampleDensity (Source/Fog/DensityField.cpp:177-211). */
float ora_sample_density(const ora_fog* fog, const float pos[3]);
/* calcTransm + intersectBBox + rayMarch (Source/Shaders/Surface.frag:139-203) with
 * sampleDensity's software trilinear filter standing in for texture().  PARITY UNPINNED:
 * GLSL cannot run here and no reference output exists for it.
 * segs: float[8*n] = (px,py,pz,_, dx,dy,dz,dist_sq). */
void ora_transmittance(const ora_fog* fog, const float* seg_p, const float* seg_d_distsq,
                       uint64_t n, float* transm, int n_threads);

/* DensityField::computePiDensity (Source/Fog/DensityField.cpp:271-327): per pixel a primary
 * closest-hit ray, then a 4*res_z-interval trapezoid march storing the running integral at every
 * i%4==2.  out: float[res_x*res_y*res_z], x fastest, then y, then z.  PARITY UNPINNED. */
void ora_compute_pi_density(const ora_scene* s, const ora_fog* fog, const float cam[12], int res_x,
                            int res_y, float* out, int n_threads);

/* DensityField noise constructor (Source/Fog/DensityField.cpp:18-76) with GLM 0.9.5.4's
 * simplex(vec3) (Source/Include/GLM/gtc/noise.inl:679-751).  out: res^3 bytes. */
void ora_make_density(const int res[3], float freq, float ampl, uint8_t* out);

/* PhotonTracer::trace (Source/RT/PhotonTracer.cpp:78-188) with the RNG as an explicit
 * float stream.  PARITY UNPINNED (no recorded VPL set).  materials: float[12*n_mat] =
 * k_d(3) pad k_s(3) n_s k_e(3) pad (Source/RT/RTBase.h:72-77).  vpls_out: 96 B records
 * (Source/VPL/PointLight.h:66-77).  Returns number of VPLs (0 on failure); *n_paths_out,
 * *n_rng_used_out report the walk. */
int ora_photon_trace(const ora_scene* s, const float* materials, uint32_t n_mat,
                     const ora_fog* fog, float abs_k, float sca_k, float maj_ext_k,
                     int fog_enabled, const float ppl_pos[3], const float ppl_intens[3],
                     const float shoot_dir[3], int max_vpl_count, const float* rng,
                     uint64_t n_rng, void* vpls_out, uint32_t* n_paths_out,
                     uint64_t* n_rng_used_out);
/* The same walk in SUBSTREAM mode, the build's own definition for the batched device walk
 * (gigl_photon_trace_batch): path p (0-based) draws u(p,k) = (splitmix64(key(p) + k*0x9E3779B97F4A7C15) >> 40) * 2^-24
 * with key(p) = splitmix64(seed ^ splitmix64(p)), and cos/sin/log/pow are the double-precision functions
 * rounded to float.  Every other operation is ora_photon_trace's.  PARITY UNPINNED. */
int ora_photon_trace_substreams(const ora_scene* s, const float* materials, uint32_t n_mat,
                                const ora_fog* fog, float abs_k, float sca_k, float maj_ext_k,
                                int fog_enabled, const float ppl_pos[3], const float ppl_intens[3],
                                const float shoot_dir[3], int max_vpl_count, uint64_t seed,
                                void* vpls_out, uint32_t* n_paths_out, uint64_t* n_draws_out);


/* Software shading stage: Surface.frag main() (Source/Shaders/Surface.frag:290-343) for one frame over
 * pixels [first, first+n_pts), and Combine.frag's normalisation + tone mapping (:72-76).  PARITY
 * UNPINNED (GLSL cannot run here; stand-ins listed at the definition).  Lights are the n_ppls
 * primary lights (ppls: 8 floats each = pos, pad, intens, pad, std140) followed by the n_vpls VPL
 * records (96 B); occluded/transm are ora_shadow_batch's outputs for that light order.
 * materials: float[12*n_mat] as ora_photon_trace.  accum_rgba: in-out float[4*n_pts];
 * fog_dist (optional): float[2*n_pts] = recordRayDist's (t_min, t_max). */
void ora_shade_surface(const ora_scene* s, const ora_fog* fog, const float cam_pos[3], int res_x, int res_y,
                       uint64_t first, const float* prim_o_tmin, const float* prim_d_tmax,
                       const void* prim_hits16, uint64_t n_pts, const float* materials, const float* ppls,
                       uint32_t n_ppls, const void* vpls, uint32_t n_vpls, const uint8_t* occluded,
                       const float* transm, const float* pi_dens, int pi_res_z, float sca_albedo,
                       int gi_enabled, int clamp_rsq, int frame_id, float* accum_rgba, float* fog_dist,
                       int n_threads);
void ora_tonemap(const float* accum_rgba, uint64_t n, int frame_id, int exposure, float* rgb);

/* Volume.frag main() (Source/Shaders/Volume.frag:268-314) in the three steps of the engine's ABI
 * (gigl_volume_samples_device, gigl_shadow_points_device, gigl_shade_volume_device).  PARITY
 * UNPINNED.  points: float[4 * (res_x/2)*(res_y/2) * n_samples], n_samples = gi_enabled ? 8 : 32;
 * occluded: uint8 per (point, light). */
void ora_halton2(float* seq, int n);   /* Source/Common/Halton.hpp, base 2 */
void ora_volume_samples(const ora_fog* fog, const float cam_pos[3], int res_x, int res_y,
                        const float* prim_o_tmin, const float* prim_d_tmax, const void* prim_hits16,
                        const float* fog_dist, const float* rnd_offsets, const float* halton,
                        const float* pi_dens, int pi_res_z, int gi_enabled, int frame_id, float* points);
void ora_shadow_points(const ora_scene* s, const float* points, uint64_t n_pts, const float* lights,
                       uint32_t n_lights, uint8_t* occluded, int n_threads);
void ora_shade_volume(const ora_fog* fog, const float cam_pos[3], int res_x, int res_y,
                      const float* prim_o_tmin, const float* prim_d_tmax, const void* prim_hits16,
                      const float* fog_dist, const float* rnd_offsets, const float* halton,
                      const float* pi_dens, int pi_res_z, const float* ppls, uint32_t n_ppls,
                      const void* vpls, uint32_t n_vpls, const uint8_t* occluded, float sca_k,
                      float sca_albedo, int gi_enabled, int clamp_rsq, int transm_opt, int frame_id,
                      float* vol_comp_rgba, int n_threads);

/* Combine.frag (Source/Shaders/Combine.frag:23-77).  PARITY UNPINNED; stand-ins at the definition.
 * depth: float[res_x*res_y] from ora_depth_buffer; accum_rgba in-out; rgb: float[3*res_x*res_y]. */
void ora_depth_buffer(const float cam_pos[3], const float cam_dir[3], float t_near, float t_far,
                      const float* prim_o_tmin, const float* prim_d_tmax, const void* prim_hits16,
                      uint64_t n, float* depth);
void ora_combine(int res_x, int res_y, const float* depth, const float* vol_comp_rgba, float ext_k,
                 int frame_id, int exposure, float* accum_rgba, float* rgb);

#ifdef __cplusplus
}
#endif
#endif
