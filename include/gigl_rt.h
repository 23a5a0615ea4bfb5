/*
 * gigl_rt.h — C ABI of the MI355X ray-query engine (libgigl_rt.so).
 *
 * Drop-in boundary for the ray-query hot path of zero-radiance/GIGL.  The reference has no FFI
 * layer: its RT module is a set of C++ member functions (Scene::trace, KdTree::intersect, ...).
 * Each entry point below names the reference interface it replaces (path:line under
 * /root/reference); INTEGRATION.md shows the few lines a maintainer of the reference would
 * change to call them.
 *
 * Conventions
 *  - Every function returns an int status (GIGL_OK = 0) and never aborts or throws across the
 *    ABI.  gigl_last_error() returns a thread-local message for the last failure.
 *    (The reference reports a miss through a bool and fatal errors through TERMINATE(),
 *    Source/Common/Utility.hpp:52-58.)
 *  - The library copies what it is given at create time; callers may free their buffers.
 *  - "_device" pointers are HIP device pointers on the scene's device; `stream` is a
 *    hipStream_t passed as void* (NULL = the default stream).  Batch calls are asynchronous
 *    on that stream.  Plain entry points take host pointers and are synchronous.
 *  - Trace calls may be issued concurrently on one handle (as KdTree::intersect is const and
 *    re-entrant, Source/Fog/DensityField.cpp:280-293); create/destroy may not.
 *  - There is no CPU fallback: without a usable HIP device every call fails with
 *    GIGL_ERR_NO_DEVICE.
 *
 * Ray layout (device friendly form of rt::Ray, Source/RT/RTBase.h:81-105): two float4 arrays,
 *   o_tmin[i] = (o.x, o.y, o.z, t_min)     d_tmax[i] = (d.x, d.y, d.z, t_max)
 * inv_d is derived as 1.0f/d per component as the Ray constructor does (RTBase.cpp:111-115).
 * Rays are FRESH: the hit slot starts at distance FLT_MAX (RTBase.cpp:133).
 */
#ifndef GIGL_RT_H
#define GIGL_RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIGL_OK             0
#define GIGL_ERR_INVALID    1   /* bad argument */
#define GIGL_ERR_NO_DEVICE  2   /* no HIP device / HIP runtime unusable */
#define GIGL_ERR_HIP        3   /* a HIP call failed */
#define GIGL_ERR_BUILD      4   /* tree build rejected the input */
#define GIGL_ERR_TOO_LARGE  5   /* size beyond a documented limit */

#define GIGL_MISS 0xFFFFFFFFu
/* device index for gigl_scene_create that builds the host tree only (bbox/stats/export work,
 * every trace call fails with GIGL_ERR_NO_DEVICE); used to check the builder without a GPU */
#define GIGL_DEVICE_NONE (-1)

typedef struct gigl_scene gigl_scene;   /* replaces Scene's RT half + rt::KdTri */
typedef struct gigl_fog   gigl_fog;     /* replaces FogVolume/DensityField's sampled data */

/* Closest-hit record.  The reference's Ray::Intersection (Source/RT/RTBase.h:95-100) holds a
 * material pointer, distance and unnormalized normal; id/u/v identify the same hit without
 * pointers (u, v are the locals of Triangle::intersect, Source/RT/RTBase.cpp:163-164). */
typedef struct gigl_hit {
    uint32_t tri_id;   /* index into the triangle array given at create; GIGL_MISS on a miss */
    float    t;        /* inters.distance; FLT_MAX on a miss */
    float    u, v;
} gigl_hit;

/* Statistics the reference prints after a build (Source/RT/KdTree.hpp:126-131). */
typedef struct gigl_tree_stats {
    uint32_t n_nodes;
    uint32_t n_leaf_prim_ids;
    uint32_t leaf_count;
    int32_t  max_depth;
    int32_t  actual_depth;
    float    avg_prims_per_leaf;
    float    build_ms;
    uint32_t degenerate_split_guard_hit;  /* 1 if a node had no valid split where the reference
                                             would index out of range (KdTree.hpp:221-232) */
    uint64_t device_bytes;
} gigl_tree_stats;

const char* gigl_last_error(void);

/* Device selection ("device-count selector"). */
int gigl_device_count(int* count);

/* Scene::loadObjects' RT half + rt::KdTree<Triangle>::KdTree
 * (Source/Common/Scene.cpp:136-158, Source/RT/KdTree.h:14-16, KdTree.hpp:87-264).
 * verts: float[3*n_verts]; tris: uint32[4*n_tris] = (i0,i1,i2,mat_id); both host pointers.
 * The tree is built on the host, flattened and uploaded once to `device`. */
int gigl_scene_create(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris,
                      int inters_cost, int trav_cost, int max_depth, uint32_t min_num_prims,
                      float empty_bonus, int device, gigl_scene** out);
int gigl_scene_destroy(gigl_scene* s);
/* KdTree::bbox (Source/RT/KdTree.h:18): out = min xyz, max xyz. */
int gigl_scene_bbox(const gigl_scene* s, float out[6]);
int gigl_scene_stats(const gigl_scene* s, gigl_tree_stats* out);
/* The device layout the scene's kernels walk (DESIGN.md "Data layout"): 0 level treelets, 1 clustered
 * lines, 2 triple records.  Option "layout" = 2 falls back to 1 when a tree exceeds the record
 * encoding, so this is how a caller (bench.py) names the kernel variant that actually ran. */
int gigl_scene_layout(const gigl_scene* s, int* layout_out);
/* The host tree in the reference's 8-byte node encoding (Source/RT/KdTree.h:22-65), for tree
 * digests: nodes_words uint32[2*n_nodes], leaf_prim_ids uint32[n_leaf_prim_ids]. */
int gigl_scene_export_tree(const gigl_scene* s, uint32_t* nodes_words, uint32_t* leaf_prim_ids);

/* Scene::trace(ray, false) over a batch (Source/Common/Scene.cpp:234-243 ->
 * KdTree::intersect, Source/RT/KdTree.hpp:266-347).  n < 2^32.  The variants without _device take
 * host pointers and return when the records are in host memory; batches of more than 2^21 rays are
 * pipelined (rays up, traversal and records down overlap on streams of the call's own), at the
 * link's rate when the arrays are page-locked. */
int gigl_trace_closest_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                              gigl_hit* hits, void* stream);
int gigl_trace_closest(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n,
                       gigl_hit* hits);
/* Scene::trace(ray, true): any-hit.  Bit i of occluded_bits (word i/64, bit i%64) is the bool
 * the reference returns; ceil(n/64) words are written. */
int gigl_trace_any_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                          uint64_t* occluded_bits, void* stream);
int gigl_trace_any(const gigl_scene* s, const float* o_tmin, const float* d_tmax, uint64_t n,
                   uint64_t* occluded_bits);
/* Several devices of one node behind one call: the scene's flattened tree and
 * triangle records are replicated on every device of the list (device to device, over xGMI between
 * two GPUs), a batch is split into contiguous ray ranges whose sizes differ by at most one ray
 * (any-hit batches: by at most one 64-ray word, so that every range owns whole words of the output),
 * each device traces its range on a stream of its own and the results are gathered by direct peer
 * copies, one point-to-point transfer per device: no ring, no reduction.
 * Extends Scene::trace over a batch (Source/Common/Scene.cpp:234-243); the reference is a single
 * CPU process and has nothing of the kind.  devices[0] must be the scene's own device (its replica
 * is the scene itself, which must outlive the handle); a device may appear more than once.
 *
 * *_device: o_tmin, d_tmax and the output are device pointers on devices[0]; `stream` is a stream of
 * devices[0] (NULL = its default stream).  The batch is ordered behind everything `stream` holds when
 * the call is made (rays still being produced on it, a pending fill of the output) and `stream` is
 * ordered behind the batch: the replicas' own streams wait for an event recorded on `stream`, and
 * `stream` waits for theirs.  gigl_multi_trace_closest_device / _any_device return when the results
 * are in place (they synchronise `stream`); the *_async forms return as soon as the work is enqueued,
 * like the single-device *_device calls.  A handle keeps two sets of replica streams and staging
 * buffers and alternates between them, so two batches issued on two caller streams overlap.
 * gigl_multi_trace_closest / _any: host pointers; each device fetches and returns its own range
 * through gigl_trace_closest / gigl_trace_any on a persistent worker thread of the handle; synchronous.
 * Results are identical to the single-device calls on the whole batch.  Calls may be issued
 * concurrently on one handle. */
typedef struct gigl_multi gigl_multi;
int gigl_multi_create(const gigl_scene* s, const int* devices, int n_devices, gigl_multi** out);
int gigl_multi_destroy(gigl_multi* m);
int gigl_multi_device_count(const gigl_multi* m, int* count);
int gigl_multi_trace_closest_device(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                    gigl_hit* hits, void* stream);
int gigl_multi_trace_closest_device_async(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                          gigl_hit* hits, void* stream);
int gigl_multi_trace_any_device(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                uint64_t* occluded_bits, void* stream);
int gigl_multi_trace_any_device_async(gigl_multi* m, const void* o_tmin, const void* d_tmax, uint64_t n,
                                      uint64_t* occluded_bits, void* stream);
int gigl_multi_trace_closest(gigl_multi* m, const float* o_tmin, const float* d_tmax, uint64_t n,
                             gigl_hit* hits);
int gigl_multi_trace_any(gigl_multi* m, const float* o_tmin, const float* d_tmax, uint64_t n,
                         uint64_t* occluded_bits);

/* The flattened scene as it lies on its device, for replication WITHOUT a rebuild: one process per GPU
 * ("uploaded to GPU 0, then ncclBroadcast").  Rank 0 builds the tree
 * (gigl_scene_create), exports the header and the device addresses of its three arrays, the caller
 * moves header and arrays with whatever it has (ncclBroadcast / torch.distributed.broadcast, peer
 * copies, a file) and every other rank adopts them.  On ranks > 0 this replaces the whole of
 * rt::KdTree<Triangle>::KdTree (Source/RT/KdTree.hpp:87-132) and the flattening.
 * gigl_scene_export_device_blob: fills *desc; ptrs_out[0..2] = device addresses of the node / record
 * array (desc->nb_nodes bytes), the separate leaf id array (nb_leaf_ids bytes, NULL when 0) and the
 * triangle table with the leaf blocks (nb_tris bytes).  They stay owned by the scene.
 * gigl_scene_create_from_device_blob: nodes, leaf_ids, tris are device pointers readable from `device`
 * (on it, or on a peer) holding desc's byte counts; they are COPIED (on `stream`, which is synchronised
 * before the call returns), so the caller may free them.  The new scene traces exactly as the source
 * does; it has no host tree (gigl_scene_export_tree and gigl_debug_check_layout fail with GIGL_ERR_INVALID). */
#define GIGL_BLOB_MAGIC 0x42474947u   /* "GIGB" */
typedef struct gigl_scene_blob {
    uint32_t magic, version;                   /* GIGL_BLOB_MAGIC, 1 */
    uint64_t nb_nodes, nb_leaf_ids, nb_tris;   /* bytes of the three device arrays */
    uint32_t layout, triples, blocks_off8, n_tris;
    uint32_t root[2], root_rec[4];
    float    box[6];                           /* tree box: min xyz, max xyz */
    gigl_tree_stats stats;
} gigl_scene_blob;
int gigl_scene_export_device_blob(const gigl_scene* s, gigl_scene_blob* desc, void* ptrs_out[3]);
int gigl_scene_create_from_device_blob(const gigl_scene_blob* desc, const void* nodes, const void* leaf_ids,
                                       const void* tris, int device, void* stream, gigl_scene** out);

/* gigl_trace_closest_device for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same inputs,
 * same outputs, bit for bit; the call allocates nothing (no hipMalloc / hipMallocAsync / hipFree, no
 * allocating library routine): its work counter and the spill area of the traversal stacks live in a
 * 24 MiB array that is part of the library's device image (one copy per GPU, held until the process
 * ends), located once per device in gigl_scene_create.  The launch is held to 256 workgroups (one
 * per CU), which that array is sized for, so large batches run slower than through
 * gigl_trace_closest_device.  The call enqueues one kernel launch and nothing else (each launch leaves the
 * array's counter words at zero for the next), so it can be captured in a HIP graph and replayed.
 * n > GIGL_NOALLOC_MAX_RAYS fails with GIGL_ERR_TOO_LARGE.  Every call on
 * a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two host threads, two
 * scenes on the same device) RACE; issue them in stream order on one stream. */
#define GIGL_NOALLOC_MAX_RAYS (1u << 24)
int gigl_trace_closest_noalloc_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                                      gigl_hit* hits, void* stream);

/* Single-ray convenience call with Scene::trace semantics (one launch; host pointers).
 * normal_out (optional) receives the normal as Scene::trace leaves it: normalized for
 * closest-hit rays, unnormalized cross(e0,e1) for visibility rays.  range_out (optional)
 * receives the ray's (t_min, t_max) as KdTree::intersect leaves them in the caller's Ray
 * (Source/RT/KdTree.hpp:273-274, :310, :337-338) — PhotonTracer's traceFog call reads them
 * (Source/RT/PhotonTracer.cpp:96-97, Source/Common/BBox.cpp:104). */
int gigl_trace_one(const gigl_scene* s, const float o[3], const float d[3], float t_min, float t_max,
                   int is_vis_ray, int* hit_out, gigl_hit* hit_rec, float normal_out[3], float range_out[2]);

/* rt::PhotonTracer::trace (Source/RT/PhotonTracer.cpp:78-188) as ray batches: VPL generation for one
 * primary light, every scene.trace of every bounce a closest-hit batch over a block of paths, every
 * Woodcock step, Russian roulette, VPL record and next ray computed on the device (one lane per path).
 * Replaces the call at Source/Common/Renderer.cpp:223.
 *
 * SUBSTREAM MODE (the build's definition; PARITY UNPINNED — the reference draws every number from one
 * global std::mt19937, which makes path p depend on paths 0..p-1 and cannot be batched): path p
 * (0-based) draws u(p,k) = (splitmix64(key(p) + k * 0x9E3779B97F4A7C15) >> 40) * 2^-24, k = 0, 1, ...,
 * with key(p) = splitmix64(seed ^ splitmix64(p)); cos, sin, log and pow are the double-precision
 * functions rounded to float.  Every other operation, the order of the draws within a path, the
 * stopping rule (the path that supplies VPL number max_vpl_count is the last one counted; at most
 * 100 * max_vpl_count paths, else no VPLs), LightArray::normalizeIntensity's renumbering and scaling
 * (Source/VPL/LightArray.hpp:46-66) are the reference's.  The serial single-stream walk remains
 * available through gigl_trace_one (gigl_amd/host/gigl_host.hpp).
 *
 * materials: host float[12 * n_mat] = (k_d, -, k_s, n_s, k_e, -) per material (rt::PhongMaterial,
 * Source/RT/RTBase.h:53-78), indexed by the material ids given to gigl_scene_create.  fog may be NULL
 * (then fog_enabled is ignored).  paths_per_batch: paths walked per block (0: the engine's choice).
 * vpls_out: host, 96 bytes per VPL (Source/VPL/PointLight.h:66-77), max_vpl_count records.
 * *n_vpls_out = max_vpl_count, or 0 when 100 * max_vpl_count paths did not supply them
 * (PhotonTracer.cpp:181-184); *n_paths_out = paths counted. */
int gigl_photon_trace_batch(const gigl_scene* s, const gigl_fog* fog, const float* materials, uint32_t n_mat,
                            float abs_k, float sca_k, float maj_ext_k, int fog_enabled,
                            const float ppl_pos[3], const float ppl_intens[3], const float shoot_dir[3],
                            int max_vpl_count, uint64_t seed, uint32_t paths_per_batch, void* vpls_out,
                            int* n_vpls_out, uint32_t* n_paths_out);

/* PerspectiveCamera (Source/Common/Camera.cpp:15-33): cam = pos, bottom_left, step_x, step_y. */
int gigl_camera_setup(const float pos[3], const float dir[3], const float up[3], float h_fov,
                      int res_x, int res_y, float cam_out[12]);
/* getPrimaryRay(x+0.5, y+0.5) for pixels [first, first+n) in row-major order
 * (Source/Common/Camera.cpp:65-68, Source/Fog/DensityField.cpp:288), generated on the device. */
int gigl_gen_primary_rays_device(const gigl_scene* s, const float cam[12], int res_x, uint64_t first,
                                 uint64_t n, void* o_tmin, void* d_tmax, void* stream);
/* Primary closest-hit with the ray generator fused into the traversal kernel. */
int gigl_trace_primary_device(const gigl_scene* s, const float cam[12], int res_x, uint64_t first,
                              uint64_t n, gigl_hit* hits, void* stream);

/* Shadow/occlusion batch: for each primary hit point x each light, one visibility ray
 * (the occlusion test Source/Shaders/Surface.frag:270-276 does with shadow maps, re-expressed
 * as Scene::trace(ray, true); ray construction in DESIGN.md "Shadow batch").
 * Pair index = point * n_lights + light; bit layout as gigl_trace_any; points whose primary
 * ray missed produce 0 bits.  lights: float[3*n_lights] device.  If fog and transm are given,
 * transm[pair] receives calcTransm (Source/Shaders/Surface.frag:190-203) of the same segment
 * (1.0 for missed points).  n_pts * n_lights < 2^32. */
int gigl_shadow_batch_device(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                             const gigl_hit* prim_hits, uint64_t n_pts, const float* lights,
                             uint32_t n_lights, uint64_t* occluded_bits, const gigl_fog* fog,
                             float* transm, void* stream);

/* The same batch in the shader's own order (Source/Shaders/Surface.frag:250-256, :270-288: the shadow
 * test first, calcTransm only for visible lights), in one call: the any-hit pass, then a march whose
 * waves compact the unoccluded pairs of their block (ballot + prefix count into LDS) so that occluded
 * pairs cost no samples and every marching lane is busy.
 * Bits as above; transm[pair] = 0 for an occluded pair (nothing arrives from that light), calcTransm
 * for a visible one, 1.0 for missed points.  Both outputs and the fog are required. */
int gigl_shadow_batch_fused_device(const gigl_scene* s, const void* prim_o_tmin, const void* prim_d_tmax,
                                   const gigl_hit* prim_hits, uint64_t n_pts, const float* lights,
                                   uint32_t n_lights, uint64_t* occluded_bits, const gigl_fog* fog,
                                   float* transm, void* stream);

/* Software shading stage ("next" row of the scope table): Surface.frag main()
 * (Source/Shaders/Surface.frag:290-343) for one frame over pixels [first, first+n_pts), consuming
 * the primary hit records, the shadow batch's occlusion bits and transmittance, and
 * computePiDensity's volume.  Lights are the n_ppls primary lights followed by the n_vpls VPLs, in
 * the order the shadow batch's `lights` array listed their positions.  All pointers are device
 * pointers except prm.  Stand-ins for the GL pipeline (G-buffer, shadow maps, texture filtering)
 * are listed in DESIGN.md "Shading stage". */
typedef struct gigl_shade_params {
    float    cam_pos[3];      /* cam_w_pos */
    int      res_x, res_y;    /* CAM_RES; pi_dens is res_x * res_y * pi_res_z */
    uint64_t first;           /* index of the batch's first pixel, row-major */
    int      n_ppls, n_vpls;  /* MAX_PPLS = 1, n_vpls <= MAX_VPLS = 150 in the reference */
    int      gi_enabled, clamp_rsq, frame_id;
    int      pi_res_z;
    float    sca_albedo;      /* FogVolume::getScaAlbedo, Source/Fog/FogVolume.cpp:44-46 */
    /* volume pass only (Volume.frag uniforms) */
    int      transm_opt;      /* shadow segments: 1 = constant medium at the sample, 0 = ray march */
    float    sca_k;           /* scattering coefficient per unit density */
} gigl_shade_params;
/* materials: float[12*n_materials] = k_d, pad, k_s, n_s, k_e, pad (Source/RT/RTBase.h:72-77).
 * ppls: float[8*n_ppls] = w_pos, pad, intens, pad (std140 PrimaryPointLight).
 * vpls: n_vpls 96-byte VPL records (Source/VPL/PointLight.h:66-77).
 * transm (optional, needs fog): per pair, as gigl_shadow_batch_device wrote it.
 * pi_dens: required when fog is given with ext_k > 0.
 * accum_rgba: float[4*n_pts], read when frame_id > 0, always written (writeToAccumBuffer).
 * fog_dist (optional): float[2*n_pts] = recordRayDist's (t_min, t_max), written when ext_k > 0. */
int gigl_shade_surface_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                              const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                              uint64_t n_pts, const float* materials, uint32_t n_materials, const float* ppls,
                              const void* vpls, const uint64_t* occluded_bits, const float* transm,
                              const float* pi_dens, float* accum_rgba, float* fog_dist, void* stream);
/* Volume.frag main() (Source/Shaders/Volume.frag:268-314): in-scattering at half resolution
 * (res_x/2 x res_y/2 pixels, n_samples = gi_enabled ? 8 : 32 samples per pixel), in three calls:
 *   1. gigl_volume_samples_device: the sample points of every pixel (Volume.frag:283-292), written
 *      as points[(pixel * n_samples + s)] = (x, y, z, valid); valid = 0 where the shader takes no
 *      sample (no fog along the ray, or camera-to-sample transmittance <= 0.001);
 *   2. gigl_shadow_points_device: shadow-map visibility re-expressed as any-hit rays from each
 *      valid point to each light, Ray{p, normalize(L - p)} over (0, |L - p|); pair index =
 *      point * n_lights + light, bit layout as gigl_trace_any, invalid points give 0 bits;
 *   3. gigl_shade_volume_device: the shading sum; vol_comp_rgba: float[4 * (res_x/2) * (res_y/2)].
 * prim_* are the full-resolution primary rays and hit records (the G-buffer: texel (2x+1, 2y+1)
 * is used), fog_dist is gigl_shade_surface_device's output for the whole image, rnd_offsets
 * float[(res_x/2)*(res_y/2)] (the reference fills it with random numbers), halton
 * float[30 * 32] (Source/Common/Halton.hpp, base 2).  prm->first is ignored.  All device pointers. */
int gigl_volume_samples_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                               const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                               const float* fog_dist, const float* rnd_offsets, const float* halton,
                               const float* pi_dens, float* points, void* stream);
int gigl_shadow_points_device(const gigl_scene* s, const float* points, uint64_t n_pts, const float* lights,
                              uint32_t n_lights, uint64_t* occluded_bits, void* stream);
int gigl_shade_volume_device(const gigl_scene* s, const gigl_fog* fog, const gigl_shade_params* prm,
                             const void* prim_o_tmin, const void* prim_d_tmax, const gigl_hit* prim_hits,
                             const float* fog_dist, const float* rnd_offsets, const float* halton,
                             const float* pi_dens, const float* ppls, const void* vpls,
                             const uint64_t* occluded_bits, float* vol_comp_rgba, void* stream);
/* Combine.frag (Source/Shaders/Combine.frag:23-77): bilateral upsampling of the half-resolution
 * volume term onto the accumulation buffer (when frame_id < 30 and ext_k > 0), then normalisation
 * and tone mapping.  depth: float[res_x*res_y] from gigl_depth_buffer_device — the GL depth buffer's
 * stand-in: lookAt + glm::perspective depth of the primary hit point (Source/Common/Camera.cpp:30-31;
 * t_near, t_far = 0.1, 2500 at Source/GIGL.cpp:41) quantised to 24 bits, 1.0 where the ray missed.
 * accum_rgba is updated in place as writeToAccumBuffer does; rgb_out: float[3*res_x*res_y]. */
int gigl_depth_buffer_device(int device, const float cam_pos[3], const float cam_dir[3], float t_near,
                             float t_far, const void* prim_o_tmin, const void* prim_d_tmax,
                             const gigl_hit* prim_hits, uint64_t n, float* depth, void* stream);
int gigl_combine_device(int device, int res_x, int res_y, const float* depth, const float* vol_comp_rgba,
                        float ext_k, int frame_id, int exposure, float* accum_rgba, float* rgb_out,
                        void* stream);
/* Combine.frag:72-76 without the volume term: rgb = 1 - exp(-exposure * accum / (1 + min(frame_id, 30))).
 * rgb_out: float[3*n]. */
int gigl_tonemap_device(int device, const float* accum_rgba, uint64_t n, int frame_id, int exposure,
                        float* rgb_out, void* stream);

/* DensityField data + FogVolume coefficients (Source/Fog/DensityField.cpp:177-211,
 * Source/Fog/FogVolume.cpp:32-36).  dens: res[0]*res[1]*res[2] bytes, x fastest; box = min, max. */
int gigl_fog_create(const uint8_t* dens, const int res[3], const float box[6], float ext_k,
                    int n_intervals, int device, gigl_fog** out);
int gigl_fog_destroy(gigl_fog* f);
/* calcTransm/rayMarch (Source/Shaders/Surface.frag:172-203) over explicit segments:
 * seg_p[i] = (p.x,p.y,p.z,unused), seg_d_distsq[i] = (dir.x,dir.y,dir.z,dist_sq). */
int gigl_transmittance_device(const gigl_fog* f, const void* seg_p, const void* seg_d_distsq, uint64_t n,
                              float* transm, void* stream);
/* DensityField::computePiDensity (Source/Fog/DensityField.cpp:271-327): for every pixel of the
 * camera a primary closest-hit ray and a 4*res_z-interval trapezoid march through the density
 * volume, storing the running integral at every i%4==2.  pi_dens: device float[res_x*res_y*res_z],
 * index x + y*res_x + z*res_x*res_y — the payload of the reference's pi_df.3dt. */
int gigl_compute_pi_density_device(const gigl_scene* s, const gigl_fog* f, const float cam[12], int res_x,
                                   int res_y, float* pi_dens, void* stream);
/* DensityField::sampleDensity (Source/Fog/DensityField.cpp:177-202) at n host positions
 * float[3*n] -> float[n]; evaluated on the device. */
int gigl_sample_density(const gigl_fog* f, const float* pos, uint64_t n, float* dens_out);

/* rt::getRescaledEPS (Source/RT/RTBase.h:18-29) evaluated on the device for n host floats;
 * exists so the parity tests can pin the device arithmetic of the traversal epsilon. */
int gigl_debug_rescaled_eps(int device, const float* t, uint64_t n, float* out);

/* Rebuilds the device layout of the scene's tree on the host (layout 0: level treelets of
 * `treelet_levels` levels, 1: clustered lines; DESIGN.md "Data layout") and checks it node by node
 * against the tree in the reference's encoding (same splits, axes, child order, leaf contents and
 * order).  Works on host-only scenes; *bytes_out (optional) = device bytes of nodes + id lists.
 * No counterpart in the reference: it pins that re-addressing changed addresses only. */
int gigl_debug_check_layout(const gigl_scene* s, int layout, int treelet_levels, uint64_t* bytes_out);

/* Scheduling counters of the last closest-hit launch made with option "stats" = 1 (instrumented
 * kernel build; diagnostic only): rounds, T-phase iterations, T lane-steps, L-phase rounds,
 * L lane-rounds, refills, refilled lanes, sum over rounds of lanes holding a ray; then, on the
 * 100 MHz real-time counter: ~(earliest wave start), latest wave end, sum of wave ends, waves. */
int gigl_debug_last_stats(uint64_t out[12]);

/* Closest-hit over an explicit device ray batch through the lean one-ray-per-lane walk that small VPL sets use
 * (gigl_walk_lean.h; scenes in triple records only), same arguments and records as gigl_trace_closest_device.
 * Diagnostic: the cross-check of that walk against the shipped kernel on batches of any size, and the measure of
 * what a lone ray costs in either form.  No counterpart in the reference. */
int gigl_debug_trace_closest_lean_device(const gigl_scene* s, const void* o_tmin, const void* d_tmax, uint64_t n,
                                         gigl_hit* hits, void* stream);

/* The same launch over time: three histograms of 1024 buckets of 81.92 us of the 100 MHz real-time counter,
 * fed by every 16th wave (bucket = (t >> 13) & 1023, so they wrap after 84 ms; gigl_debug_last_stats' earliest start says where
 * the launch begins): rays handed to lanes, rounds of all waves, lanes holding a ray summed over those
 * rounds.  Shows the ramp, the steady state and the drain of one launch (tools/timeline.py). */
int gigl_debug_last_timeline(uint64_t out[3072]);

/* Residency of the explicit-ray closest-hit kernel of this scene's layout: 256-thread workgroups per
 * CU as the occupancy query reports them (= waves per SIMD: 4 waves per workgroup over 4 SIMDs), and
 * the CU count the persistent grid is sized with.  Diagnostic; bench.py reports it. */
int gigl_debug_launch_info(const gigl_scene* s, int* blocks_per_cu, int* n_cus);

/* Launch workspaces and staging buffers come from a stream-ordered pool the library owns per device, which keeps
 * the largest blocks a process ever asked for between calls.  The pool's idle memory goes back to the device when
 * the last scene on that device is destroyed, or when this is called.  No counterpart in the reference. */
int gigl_trim_pool(int device);
/* Number of stream-ordered allocations the library has made so far in this process (diagnostic; the test of
 * gigl_trace_closest_noalloc_device asserts that a call leaves it unchanged). */
int gigl_debug_pool_allocs(uint64_t* count);

/* Tuning knobs (process-wide; read at each launch).  Names and meaning in DESIGN.md. */
int gigl_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
