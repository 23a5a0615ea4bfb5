"""Host-side Python mirror of the reference's RT interface over the C ABI (include/gigl_rt.h).

Names follow the reference (/root/reference/Source/Common/Scene.h:17-49,
Source/RT/KdTree.h:14-20): Scene.trace, Scene.bbox, FogVolume ...  Everything here calls
gigl_amd/lib/libgigl_rt.so through ctypes; there is no CPU path, and a missing library or GPU
raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GIGL_RT_LIB") or os.path.join(_HERE, "lib", "libgigl_rt.so")  # override: A/B builds
_LIB = None

GIGL_MISS = 0xFFFFFFFF
GIGL_DEVICE_NONE = -1
HIT_DTYPE = np.dtype([("tri_id", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

# every symbol include/gigl_rt.h declares
SYMBOLS = [
    "gigl_last_error", "gigl_device_count", "gigl_scene_create", "gigl_scene_destroy", "gigl_scene_bbox",
    "gigl_scene_stats", "gigl_scene_export_tree", "gigl_trace_closest_device", "gigl_trace_closest",
    "gigl_trace_any_device", "gigl_trace_any", "gigl_trace_one", "gigl_camera_setup",
    "gigl_gen_primary_rays_device", "gigl_trace_primary_device", "gigl_shadow_batch_device", "gigl_fog_create",
    "gigl_fog_destroy", "gigl_transmittance_device", "gigl_sample_density", "gigl_debug_rescaled_eps",
    "gigl_set_option", "gigl_debug_last_stats", "gigl_debug_last_timeline", "gigl_debug_trace_closest_lean_device", "gigl_compute_pi_density_device", "gigl_shade_surface_device",
    "gigl_tonemap_device", "gigl_volume_samples_device", "gigl_shadow_points_device", "gigl_shade_volume_device",
    "gigl_depth_buffer_device", "gigl_combine_device", "gigl_debug_check_layout", "gigl_scene_layout",
    "gigl_trace_closest_noalloc_device", "gigl_multi_create", "gigl_multi_destroy", "gigl_multi_device_count",
    "gigl_multi_trace_closest_device", "gigl_multi_trace_closest", "gigl_shadow_batch_fused_device",
    "gigl_debug_launch_info", "gigl_photon_trace_batch", "gigl_multi_trace_closest_device_async",
    "gigl_multi_trace_any_device", "gigl_multi_trace_any_device_async", "gigl_multi_trace_any",
    "gigl_scene_export_device_blob", "gigl_scene_create_from_device_blob", "gigl_trim_pool", "gigl_debug_pool_allocs",
]


class GiglError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gigl_rt error %d: %s" % (code, msg))
        self.code = code


class ShadeParams(C.Structure):
    """gigl_shade_params (include/gigl_rt.h)."""
    _fields_ = [("cam_pos", C.c_float * 3), ("res_x", C.c_int), ("res_y", C.c_int), ("first", C.c_uint64),
                ("n_ppls", C.c_int), ("n_vpls", C.c_int), ("gi_enabled", C.c_int), ("clamp_rsq", C.c_int),
                ("frame_id", C.c_int), ("pi_res_z", C.c_int), ("sca_albedo", C.c_float),
                ("transm_opt", C.c_int), ("sca_k", C.c_float)]


class TreeStats(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaf_prim_ids", C.c_uint32), ("leaf_count", C.c_uint32),
                ("max_depth", C.c_int32), ("actual_depth", C.c_int32), ("avg_prims_per_leaf", C.c_float),
                ("build_ms", C.c_float), ("degenerate_split_guard_hit", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class SceneBlob(C.Structure):
    """gigl_scene_blob (include/gigl_rt.h): the header of a flattened scene as it lies on its device."""
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("nb_nodes", C.c_uint64), ("nb_leaf_ids", C.c_uint64),
                ("nb_tris", C.c_uint64), ("layout", C.c_uint32), ("triples", C.c_uint32), ("blocks_off8", C.c_uint32),
                ("n_tris", C.c_uint32), ("root", C.c_uint32 * 2), ("root_rec", C.c_uint32 * 4), ("box", C.c_float * 6),
                ("stats", TreeStats)]


def lib():
    """Loads the HIP engine; raises if it was not built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP engine missing: %s (build with __graft_entry__.build() or "
                               "`make -C gigl_amd/csrc`)" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 under the
        # same sonames as /opt/rocm.  Whichever loads first serves the whole process, and torch
        # cannot initialise on top of the system copy, so in a process that uses torch for device
        # memory torch must come first.  (A C++ host never meets this: it links /opt/rocm only.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
        L.gigl_last_error.restype = C.c_char_p
        L.gigl_device_count.argtypes = [vp]
        L.gigl_scene_create.argtypes = [vp, u32, vp, u32, i, i, i, u32, f, i, vp]
        L.gigl_scene_destroy.argtypes = [vp]
        L.gigl_scene_bbox.argtypes = [vp, vp]
        L.gigl_scene_stats.argtypes = [vp, vp]
        L.gigl_scene_export_tree.argtypes = [vp, vp, vp]
        L.gigl_trace_closest_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_trace_closest.argtypes = [vp, vp, vp, u64, vp]
        L.gigl_trace_closest_noalloc_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_scene_layout.argtypes = [vp, vp]
        L.gigl_debug_launch_info.argtypes = [vp, vp, vp]
        L.gigl_multi_create.argtypes = [vp, vp, i, vp]
        L.gigl_multi_destroy.argtypes = [vp]
        L.gigl_multi_device_count.argtypes = [vp, vp]
        L.gigl_multi_trace_closest_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_multi_trace_closest_device_async.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_multi_trace_any_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_multi_trace_any_device_async.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_multi_trace_closest.argtypes = [vp, vp, vp, u64, vp]
        L.gigl_multi_trace_any.argtypes = [vp, vp, vp, u64, vp]
        L.gigl_trim_pool.argtypes = [i]
        L.gigl_debug_pool_allocs.argtypes = [vp]
        L.gigl_scene_export_device_blob.argtypes = [vp, vp, vp]
        L.gigl_scene_create_from_device_blob.argtypes = [vp, vp, vp, vp, i, vp, vp]
        L.gigl_trace_any_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_trace_any.argtypes = [vp, vp, vp, u64, vp]
        L.gigl_trace_one.argtypes = [vp, vp, vp, f, f, i, vp, vp, vp, vp]
        L.gigl_photon_trace_batch.argtypes = [vp, vp, vp, u32, f, f, f, i, vp, vp, vp, i, u64, u32, vp, vp, vp]
        L.gigl_camera_setup.argtypes = [vp, vp, vp, f, i, i, vp]
        L.gigl_gen_primary_rays_device.argtypes = [vp, vp, i, u64, u64, vp, vp, vp]
        L.gigl_trace_primary_device.argtypes = [vp, vp, i, u64, u64, vp, vp]
        L.gigl_shadow_batch_device.argtypes = [vp, vp, vp, vp, u64, vp, u32, vp, vp, vp, vp]
        L.gigl_shadow_batch_fused_device.argtypes = [vp, vp, vp, vp, u64, vp, u32, vp, vp, vp, vp]
        L.gigl_fog_create.argtypes = [vp, vp, vp, f, i, i, vp]
        L.gigl_fog_destroy.argtypes = [vp]
        L.gigl_transmittance_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_sample_density.argtypes = [vp, vp, u64, vp]
        L.gigl_debug_rescaled_eps.argtypes = [i, vp, u64, vp]
        L.gigl_set_option.argtypes = [C.c_char_p, i]
        L.gigl_debug_last_stats.argtypes = [vp]
        L.gigl_debug_last_timeline.argtypes = [vp]
        L.gigl_debug_trace_closest_lean_device.argtypes = [vp, vp, vp, u64, vp, vp]
        L.gigl_debug_check_layout.argtypes = [vp, i, i, vp]
        L.gigl_compute_pi_density_device.argtypes = [vp, vp, vp, i, i, vp, vp]
        L.gigl_shade_surface_device.argtypes = [vp, vp, vp, vp, vp, vp, u64, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.gigl_tonemap_device.argtypes = [i, vp, u64, i, i, vp, vp]
        L.gigl_volume_samples_device.argtypes = [vp] * 12
        L.gigl_shadow_points_device.argtypes = [vp, vp, u64, vp, u32, vp, vp]
        L.gigl_shade_volume_device.argtypes = [vp] * 15
        L.gigl_depth_buffer_device.argtypes = [i, vp, vp, f, f, vp, vp, vp, u64, vp, vp]
        L.gigl_combine_device.argtypes = [i, i, i, vp, vp, f, i, i, vp, vp, vp]
        for s in SYMBOLS:
            if s != "gigl_last_error":
                getattr(L, s).restype = C.c_int
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise GiglError(rc, lib().gigl_last_error().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    n = C.c_int(0)
    _check(lib().gigl_device_count(C.byref(n)))
    return n.value


def trim_pool(device=0):
    _check(lib().gigl_trim_pool(device))


def debug_pool_allocs():
    n = C.c_uint64(0)
    _check(lib().gigl_debug_pool_allocs(C.byref(n)))
    return n.value


def set_option(name, value):
    _check(lib().gigl_set_option(name.encode(), int(value)))


def debug_last_timeline():
    """(rays started, rounds, ray-holding lanes over rounds) per 81.92 us bucket of the last instrumented launch (every 16th wave),
    rotated so that bucket 0 holds the first wave's start; (3, 1024) uint64."""
    out = np.zeros(3 * 1024, np.uint64)
    _check(lib().gigl_debug_last_timeline(_p(out)))
    st = np.zeros(12, np.uint64)
    _check(lib().gigl_debug_last_stats(_p(st)))
    t0 = (~int(st[8])) & 0xFFFFFFFFFFFFFFFF
    return np.roll(out.reshape(3, 1024), -((t0 >> 13) & 1023), axis=1)


def debug_last_stats():
    out = np.zeros(12, np.uint64)
    _check(lib().gigl_debug_last_stats(_p(out)))
    names = ["rounds", "t_iters", "t_lane_steps", "l_rounds", "l_lane_rounds", "refills", "refill_lanes",
             "ray_lane_rounds"]
    d = dict(zip(names, out[:8].tolist()))
    waves = int(out[11])
    if waves:   # wave timing on the 100 MHz real-time counter -> milliseconds since the first wave started
        t0 = int(~out[8] & np.uint64(0xFFFFFFFFFFFFFFFF))
        d["waves"] = waves
        d["last_wave_end_ms"] = (int(out[9]) - t0) / 1e5
        d["mean_wave_end_ms"] = (int(out[10]) / waves - t0) / 1e5
    return d


def camera_setup(pos, direction, up, h_fov, res_x, res_y):
    out = np.zeros(12, np.float32)
    _check(lib().gigl_camera_setup(_p(_f32(pos)), _p(_f32(direction)), _p(_f32(up)),
                                   float(np.float32(h_fov)), res_x, res_y, _p(out)))
    return out


def debug_rescaled_eps(t, device=0):
    t = _f32(t).ravel()
    out = np.zeros_like(t)
    _check(lib().gigl_debug_rescaled_eps(device, _p(t), t.size, _p(out)))
    return out


def tonemap_device(accum_ptr, n, frame_id, exposure, rgb_ptr, device=0, stream=0):
    """Combine.frag:72-76 without the volume term."""
    _check(lib().gigl_tonemap_device(device, accum_ptr, n, frame_id, exposure, rgb_ptr, stream))


def depth_buffer_device(cam_pos, cam_dir, t_near, t_far, o_ptr, d_ptr, hits_ptr, n, depth_ptr, device=0, stream=0):
    """The GL depth buffer's stand-in for Combine.frag (see include/gigl_rt.h)."""
    _check(lib().gigl_depth_buffer_device(device, _p(_f32(cam_pos)), _p(_f32(cam_dir)), t_near, t_far, o_ptr, d_ptr,
                                          hits_ptr, n, depth_ptr, stream))


def combine_device(res_x, res_y, depth_ptr, vol_comp_ptr, ext_k, frame_id, exposure, accum_ptr, rgb_ptr, device=0,
                   stream=0):
    """Combine.frag (Source/Shaders/Combine.frag:23-77)."""
    _check(lib().gigl_combine_device(device, res_x, res_y, depth_ptr, vol_comp_ptr, ext_k, frame_id, exposure,
                                     accum_ptr, rgb_ptr, stream))


def unpack_bits(words, n):
    """uint64 words -> uint8[n] of 0/1 in ray order (bit i%64 of word i/64)."""
    b = np.unpackbits(np.ascontiguousarray(words, dtype="<u8").view(np.uint8), bitorder="little")
    return b[:n]


class Scene:
    """Scene's RT half: owns the k-d tree and the triangles on one GPU.

    Scene::loadObjects (RT arrays) + rt::KdTri(m_triangles, 10, 1, 30, 2):
    /root/reference/Source/Common/Scene.cpp:136-158.
    """

    def __init__(self, verts, tris, inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2,
                 empty_bonus=0.25, device=0):
        verts = _f32(verts).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 4)
        h = C.c_void_p()
        _check(lib().gigl_scene_create(_p(verts), len(verts), _p(tris), len(tris), inters_cost, trav_cost,
                                       max_depth, min_num_prims, empty_bonus, device, C.byref(h)))
        self.h = h
        self.device = device
        self.n_tris = len(tris)

    @classmethod
    def from_device_blob(cls, desc, nodes_ptr, leaf_ids_ptr, tris_ptr, device=0, stream=0):
        """A scene adopted from the flattened arrays of another one (gigl_scene_create_from_device_blob): no
        build; the arrays are copied, the caller may free them afterwards."""
        h = C.c_void_p()
        _check(lib().gigl_scene_create_from_device_blob(C.byref(desc), nodes_ptr, leaf_ids_ptr or None, tris_ptr, device,
                                                        stream, C.byref(h)))
        self = cls.__new__(cls)
        self.h, self.device, self.n_tris = h, device, int(desc.n_tris)
        return self

    def export_device_blob(self):
        """(header, [nodes_ptr, leaf_ids_ptr, tris_ptr]): gigl_scene_export_device_blob.  The arrays stay the scene's."""
        desc = SceneBlob()
        ptrs = (C.c_void_p * 3)()
        _check(lib().gigl_scene_export_device_blob(self.h, C.byref(desc), ptrs))
        return desc, [p or 0 for p in ptrs]

    def close(self):
        if getattr(self, "h", None):
            lib().gigl_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bbox(self):
        out = np.zeros(6, np.float32)
        _check(lib().gigl_scene_bbox(self.h, _p(out)))
        return out

    def stats(self):
        st = TreeStats()
        _check(lib().gigl_scene_stats(self.h, C.byref(st)))
        return st

    def layout(self):
        """The device layout in use: 0 level treelets, 1 clustered lines, 2 triple records."""
        v = C.c_int(-1)
        _check(lib().gigl_scene_layout(self.h, C.byref(v)))
        return v.value

    def launch_info(self):
        """(workgroups per CU = waves per SIMD, CUs) of the closest-hit kernel of this scene's layout."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().gigl_debug_launch_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def check_layout(self, layout=1, treelet_levels=3):
        """Rebuilds the device layout on the host and checks it against the tree; returns its bytes."""
        n = C.c_uint64(0)
        _check(lib().gigl_debug_check_layout(self.h, layout, treelet_levels, C.byref(n)))
        return n.value

    def export_tree(self):
        st = self.stats()
        nodes = np.zeros((st.n_nodes, 2), np.uint32)
        leaf = np.zeros(st.n_leaf_prim_ids, np.uint32)
        _check(lib().gigl_scene_export_tree(self.h, _p(nodes), _p(leaf)))
        return nodes, leaf

    # ---- host-buffer calls (numpy) ----
    def trace(self, o_tmin, d_tmax, is_vis_ray=False):
        """Scene::trace over a batch of fresh rays; returns hit records or 0/1 occlusion bytes."""
        ro, rd = _f32(o_tmin).reshape(-1, 4), _f32(d_tmax).reshape(-1, 4)
        n = len(ro)
        if is_vis_ray:
            words = np.zeros((n + 63) // 64, np.uint64)
            _check(lib().gigl_trace_any(self.h, _p(ro), _p(rd), n, _p(words)))
            return unpack_bits(words, n)
        hits = np.zeros(n, HIT_DTYPE)
        _check(lib().gigl_trace_closest(self.h, _p(ro), _p(rd), n, _p(hits)))
        return hits

    def trace_one(self, o, d, t_min=0.0, t_max=float(np.finfo(np.float32).max), is_vis_ray=False):
        hit = C.c_int(0)
        rec = np.zeros(1, HIT_DTYPE)
        nrm = np.zeros(3, np.float32)
        rng = np.zeros(2, np.float32)
        _check(lib().gigl_trace_one(self.h, _p(_f32(o)), _p(_f32(d)), t_min, t_max, int(is_vis_ray),
                                    C.byref(hit), _p(rec), _p(nrm), _p(rng)))
        self.last_range = rng
        return bool(hit.value), rec[0], nrm

    def photon_trace_batch(self, materials, fog, abs_k, sca_k, maj_ext_k, fog_enabled, ppl_pos, ppl_intens,
                           shoot_dir, max_vpl_count, seed, paths_per_batch=0):
        """rt::PhotonTracer::trace as ray batches, substream mode (include/gigl_rt.h):
        (n_vpls, 96-byte VPL records as a (max_vpl_count, 24) uint32 array, n_paths)."""
        materials = _f32(materials).reshape(-1, 12)
        vpls = np.zeros((max_vpl_count, 24), np.uint32)
        n_vpls, n_paths = C.c_int(0), C.c_uint32(0)
        _check(lib().gigl_photon_trace_batch(self.h, fog.h if fog is not None else None, _p(materials), len(materials),
                                             abs_k, sca_k, maj_ext_k, int(fog_enabled), _p(_f32(ppl_pos)),
                                             _p(_f32(ppl_intens)), _p(_f32(shoot_dir)), max_vpl_count,
                                             int(seed) & (2 ** 64 - 1), paths_per_batch, _p(vpls), C.byref(n_vpls),
                                             C.byref(n_paths)))
        return n_vpls.value, vpls, n_paths.value

    # ---- device-pointer calls (ints from torch.Tensor.data_ptr()) ----
    def trace_closest_device(self, o_ptr, d_ptr, n, hits_ptr, stream=0):
        _check(lib().gigl_trace_closest_device(self.h, o_ptr, d_ptr, n, hits_ptr, stream))

    def debug_trace_closest_lean_device(self, o_ptr, d_ptr, n, hits_ptr, stream=0):
        """trace_closest_device through the lean one-ray-per-lane walk (diagnostic; triple-record scenes)."""
        _check(lib().gigl_debug_trace_closest_lean_device(self.h, o_ptr, d_ptr, n, hits_ptr, stream))

    def trace_closest_noalloc_device(self, o_ptr, d_ptr, n, hits_ptr, stream=0):
        """trace_closest_device without any allocation (static device scratch; see include/gigl_rt.h)."""
        _check(lib().gigl_trace_closest_noalloc_device(self.h, o_ptr, d_ptr, n, hits_ptr, stream))

    def trace_any_device(self, o_ptr, d_ptr, n, bits_ptr, stream=0):
        _check(lib().gigl_trace_any_device(self.h, o_ptr, d_ptr, n, bits_ptr, stream))

    def gen_primary_rays_device(self, cam, res_x, first, n, o_ptr, d_ptr, stream=0):
        _check(lib().gigl_gen_primary_rays_device(self.h, _p(_f32(cam)), res_x, first, n, o_ptr, d_ptr, stream))

    def trace_primary_device(self, cam, res_x, first, n, hits_ptr, stream=0):
        _check(lib().gigl_trace_primary_device(self.h, _p(_f32(cam)), res_x, first, n, hits_ptr, stream))

    def compute_pi_density_device(self, fog, cam, res_x, res_y, out_ptr, stream=0):
        """DensityField::computePiDensity as one GPU batch (Source/Fog/DensityField.cpp:271-327)."""
        _check(lib().gigl_compute_pi_density_device(self.h, fog.h, _p(_f32(cam)), res_x, res_y, out_ptr, stream))

    def shade_surface_device(self, prm, o_ptr, d_ptr, hits_ptr, n_pts, materials_ptr, n_materials, ppls_ptr,
                             vpls_ptr, bits_ptr, accum_ptr, fog=None, transm_ptr=None, pi_dens_ptr=None,
                             fog_dist_ptr=None, stream=0):
        """Surface.frag main() as a compute kernel (Source/Shaders/Surface.frag:290-343); prm: ShadeParams."""
        _check(lib().gigl_shade_surface_device(self.h, fog.h if fog is not None else None, C.byref(prm), o_ptr,
                                               d_ptr, hits_ptr, n_pts, materials_ptr, n_materials, ppls_ptr,
                                               vpls_ptr, bits_ptr, transm_ptr, pi_dens_ptr, accum_ptr,
                                               fog_dist_ptr, stream))

    # Volume.frag main() in three steps (Source/Shaders/Volume.frag:268-314), see include/gigl_rt.h
    def volume_samples_device(self, fog, prm, o_ptr, d_ptr, hits_ptr, fog_dist_ptr, rnd_ptr, halton_ptr, pi_dens_ptr,
                              points_ptr, stream=0):
        _check(lib().gigl_volume_samples_device(self.h, fog.h, C.byref(prm), o_ptr, d_ptr, hits_ptr, fog_dist_ptr,
                                                rnd_ptr, halton_ptr, pi_dens_ptr, points_ptr, stream))

    def shadow_points_device(self, points_ptr, n_pts, lights_ptr, n_lights, bits_ptr, stream=0):
        _check(lib().gigl_shadow_points_device(self.h, points_ptr, n_pts, lights_ptr, n_lights, bits_ptr, stream))

    def shade_volume_device(self, fog, prm, o_ptr, d_ptr, hits_ptr, fog_dist_ptr, rnd_ptr, halton_ptr, pi_dens_ptr,
                            ppls_ptr, vpls_ptr, bits_ptr, vol_comp_ptr, stream=0):
        _check(lib().gigl_shade_volume_device(self.h, fog.h, C.byref(prm), o_ptr, d_ptr, hits_ptr, fog_dist_ptr,
                                              rnd_ptr, halton_ptr, pi_dens_ptr, ppls_ptr, vpls_ptr, bits_ptr,
                                              vol_comp_ptr, stream))

    def shadow_batch_device(self, o_ptr, d_ptr, hits_ptr, n_pts, lights_ptr, n_lights, bits_ptr, fog=None,
                            transm_ptr=None, stream=0):
        _check(lib().gigl_shadow_batch_device(self.h, o_ptr, d_ptr, hits_ptr, n_pts, lights_ptr, n_lights,
                                              bits_ptr, fog.h if fog is not None else None, transm_ptr, stream))


    def shadow_batch_fused_device(self, o_ptr, d_ptr, hits_ptr, n_pts, lights_ptr, n_lights, bits_ptr, fog,
                                  transm_ptr, stream=0):
        """Occlusion bits and, for the unoccluded pairs only, calcTransm, in one pass (0 where occluded)."""
        _check(lib().gigl_shadow_batch_fused_device(self.h, o_ptr, d_ptr, hits_ptr, n_pts, lights_ptr, n_lights,
                                                    bits_ptr, fog.h, transm_ptr, stream))


class MultiDeviceScene:
    """A scene replicated over a list of devices of one node (gigl_multi_*): contiguous ray ranges per
    device, hit records gathered to devices[0] (device pointers) or to the host (numpy)."""

    def __init__(self, scene, devices):
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib().gigl_multi_create(scene.h, devs, len(devices), C.byref(h)))
        self.h = h
        self.scene = scene   # replica 0 is the scene itself: keep it alive

    def close(self):
        if getattr(self, "h", None):
            lib().gigl_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self):
        n = C.c_int(0)
        _check(lib().gigl_multi_device_count(self.h, C.byref(n)))
        return n.value

    def trace(self, o_tmin, d_tmax):
        ro, rd = _f32(o_tmin).reshape(-1, 4), _f32(d_tmax).reshape(-1, 4)
        hits = np.zeros(len(ro), HIT_DTYPE)
        _check(lib().gigl_multi_trace_closest(self.h, _p(ro), _p(rd), len(ro), _p(hits)))
        return hits

    def trace_any(self, o_tmin, d_tmax):
        ro, rd = _f32(o_tmin).reshape(-1, 4), _f32(d_tmax).reshape(-1, 4)
        bits = np.zeros((len(ro) + 63) // 64, np.uint64)
        _check(lib().gigl_multi_trace_any(self.h, _p(ro), _p(rd), len(ro), _p(bits)))
        return unpack_bits(bits, len(ro))

    def trace_closest_device(self, o_ptr, d_ptr, n, hits_ptr, stream=0, wait=True):
        """Ordered behind `stream` (a stream of devices[0]) and `stream` behind it; wait=False returns once enqueued."""
        f = lib().gigl_multi_trace_closest_device if wait else lib().gigl_multi_trace_closest_device_async
        _check(f(self.h, o_ptr, d_ptr, n, hits_ptr, stream))

    def trace_any_device(self, o_ptr, d_ptr, n, bits_ptr, stream=0, wait=True):
        f = lib().gigl_multi_trace_any_device if wait else lib().gigl_multi_trace_any_device_async
        _check(f(self.h, o_ptr, d_ptr, n, bits_ptr, stream))


class FogVolume:
    """FogVolume/DensityField data on the GPU (/root/reference/Source/Fog/FogVolume.cpp:32-46)."""

    def __init__(self, dens, box, abs_k=1e-5, sca_k=1e-2, n_intervals=8, device=0):
        dens = np.ascontiguousarray(dens, dtype=np.uint8)
        assert dens.ndim == 3  # [z][y][x]
        res = (C.c_int * 3)(dens.shape[2], dens.shape[1], dens.shape[0])
        box = _f32(box).reshape(6)
        self.abs_k, self.sca_k = np.float32(abs_k), np.float32(sca_k)
        self.ext_k = float(self.abs_k + self.sca_k)             # FogVolume.cpp:34
        h = C.c_void_p()
        _check(lib().gigl_fog_create(_p(dens), res, _p(box), self.ext_k, n_intervals, device, C.byref(h)))
        self.h = h
        self.box = box

    def close(self):
        if getattr(self, "h", None):
            lib().gigl_fog_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sample_density(self, pos):
        pos = _f32(pos).reshape(-1, 3)
        out = np.zeros(len(pos), np.float32)
        _check(lib().gigl_sample_density(self.h, _p(pos), len(pos), _p(out)))
        return out

    def sample_ext_k(self, pos):
        return np.float32(self.ext_k) * self.sample_density(pos)

    def sca_albedo(self):
        return self.sca_k / (self.abs_k + self.sca_k)           # FogVolume.cpp:43-46

    def transmittance_device(self, seg_p_ptr, seg_d_ptr, n, out_ptr, stream=0):
        _check(lib().gigl_transmittance_device(self.h, seg_p_ptr, seg_d_ptr, n, out_ptr, stream))
