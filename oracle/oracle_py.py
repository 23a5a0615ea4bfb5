"""ctypes binding of the CPU ORACLE (oracle/libgigl_oracle.so).

Test infrastructure only: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gigl_amd/ may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DTYPE = np.dtype([("tri_id", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
VPL_DTYPE = np.dtype([
    ("w_pos", "<f4", 3), ("type", "<u4"), ("intens", "<f4", 3), ("sca_k", "<f4"),
    ("w_inc", "<f4", 3), ("path_id", "<u4"), ("w_norm", "<f4", 3), ("pad1", "<u4"),
    ("k_d", "<f4", 3), ("pad2", "<u4"), ("k_s", "<f4", 3), ("n_s", "<f4")])
assert VPL_DTYPE.itemsize == 96


class OraFog(C.Structure):
    _fields_ = [("dens", C.c_void_p), ("res", C.c_int * 3), ("box", C.c_float * 6),
                ("ext_k", C.c_float), ("n_intervals", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgigl_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.ora_scene_create.restype = C.c_void_p
        L.ora_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                       C.c_int, C.c_uint32, C.c_float]
        L.ora_scene_destroy.argtypes = [C.c_void_p]
        L.ora_scene_from_tree.restype = C.c_void_p
        L.ora_scene_from_tree.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_uint32]
        L.ora_trace_one.restype = C.c_int
        L.ora_trace_one.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
        for f in ("ora_scene_n_nodes", "ora_scene_n_leaf_prim_ids", "ora_scene_leaf_count"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.ora_scene_depth.restype = C.c_int
        L.ora_scene_depth.argtypes = [C.c_void_p]
        L.ora_scene_avg_prims_per_leaf.restype = C.c_float
        L.ora_scene_avg_prims_per_leaf.argtypes = [C.c_void_p]
        L.ora_scene_bbox.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_scene_copy_nodes.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_scene_copy_leaf_prim_ids.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_rescaled_eps.restype = C.c_float
        L.ora_rescaled_eps.argtypes = [C.c_float]
        L.ora_bbox_intersect.restype = C.c_int
        L.ora_bbox_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p]
        L.ora_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.c_void_p, C.c_int]
        L.ora_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_int]
        L.ora_camera_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                       C.c_void_p]
        L.ora_primary_rays.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.ora_shadow_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_shadow_batch_fused.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int]
        L.ora_max_threads.restype = C.c_int
        L.ora_sample_density.restype = C.c_float
        L.ora_sample_density.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_transmittance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.ora_compute_pi_density.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ora_make_density.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.ora_photon_trace.restype = C.c_int
        L.ora_photon_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_float,
                                       C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_photon_trace_substreams.restype = C.c_int
        L.ora_photon_trace_substreams.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float,
                                                  C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]
        L.ora_shade_surface.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int]
        L.ora_tonemap.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.ora_halton2.argtypes = [vp, ci]
        L.ora_depth_buffer.argtypes = [vp, vp, cf, cf, vp, vp, vp, C.c_uint64, vp]
        L.ora_combine.argtypes = [ci, ci, vp, vp, cf, ci, ci, vp, vp]
        L.ora_volume_samples.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
        L.ora_shadow_points.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, ci]
        L.ora_shade_volume.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, C.c_uint32, vp,
                                       C.c_uint32, vp, cf, cf, ci, ci, ci, ci, vp, ci]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Fog:
    """Host-side description of the 64^3 density volume for the oracle."""

    def __init__(self, dens, box, ext_k, n_intervals=8):
        self.dens = np.ascontiguousarray(dens, dtype=np.uint8)
        assert self.dens.ndim == 3  # [z][y][x]
        self.box = _f32(box).reshape(6)
        self.ext_k = float(np.float32(ext_k))
        self.n_intervals = int(n_intervals)
        self.c = OraFog()
        self.c.dens = self.dens.ctypes.data
        self.c.res[0], self.c.res[1], self.c.res[2] = self.dens.shape[2], self.dens.shape[1], self.dens.shape[0]
        for i in range(6):
            self.c.box[i] = float(self.box[i])
        self.c.ext_k = self.ext_k
        self.c.n_intervals = self.n_intervals

    def ref(self):
        return C.byref(self.c)


class OracleScene:
    def __init__(self, verts, tris, inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2,
                 empty_bonus=0.25, tree=None):
        """tree = (nodes uint32[n,2], leaf_prim_ids uint32[m]) in the reference's encoding: traverse that
        tree instead of building one (large scenes; see ora_scene_from_tree)."""
        self.verts = _f32(verts).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 4)
        if tree is not None:
            nodes = np.ascontiguousarray(tree[0], dtype=np.uint32).reshape(-1, 2)
            leaf = np.ascontiguousarray(tree[1], dtype=np.uint32).reshape(-1)
            self.h = lib().ora_scene_from_tree(_p(self.verts), len(self.verts), _p(self.tris), len(self.tris),
                                               _p(nodes), len(nodes), _p(leaf), len(leaf))
            return
        self.h = lib().ora_scene_create(_p(self.verts), len(self.verts), _p(self.tris), len(self.tris),
                                        inters_cost, trav_cost, max_depth, min_num_prims, empty_bonus)

    def close(self):
        if self.h:
            lib().ora_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_nodes(self):
        return lib().ora_scene_n_nodes(self.h)

    @property
    def depth(self):
        return lib().ora_scene_depth(self.h)

    @property
    def leaf_count(self):
        return lib().ora_scene_leaf_count(self.h)

    @property
    def avg_prims_per_leaf(self):
        return lib().ora_scene_avg_prims_per_leaf(self.h)

    def bbox(self):
        out = np.zeros(6, np.float32)
        lib().ora_scene_bbox(self.h, _p(out))
        return out

    def nodes(self):
        out = np.zeros((self.n_nodes, 2), np.uint32)
        lib().ora_scene_copy_nodes(self.h, _p(out))
        return out

    def leaf_prim_ids(self):
        out = np.zeros(lib().ora_scene_n_leaf_prim_ids(self.h), np.uint32)
        lib().ora_scene_copy_leaf_prim_ids(self.h, _p(out))
        return out

    def trace_closest(self, ro, rd, counters=False, n_threads=1):
        ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
        n = len(ro)
        hits = np.zeros(n, HIT_DTYPE)
        cnt = np.zeros((n, 2), np.uint32) if counters else None
        lib().ora_trace_closest(self.h, _p(ro), _p(rd), n, _p(hits), _p(cnt), n_threads)
        return (hits, cnt) if counters else hits

    def trace_one(self, o, d, t_min=0.0, t_max=float(np.finfo(np.float32).max), is_vis_ray=False):
        """Scene::trace on one ray: (hit, record, normal, (t_min, t_max) left in the ray)."""
        rec = np.zeros(1, HIT_DTYPE)
        nrm = np.zeros(3, np.float32)
        rng = np.zeros(2, np.float32)
        hit = lib().ora_trace_one(self.h, _p(_f32(o)), _p(_f32(d)), t_min, t_max, int(is_vis_ray), _p(rec), _p(nrm),
                                  _p(rng))
        return bool(hit), rec[0], nrm, rng

    def trace_any(self, ro, rd, counters=False, n_threads=1):
        ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
        n = len(ro)
        occ = np.zeros(n, np.uint8)
        cnt = np.zeros((n, 2), np.uint32) if counters else None
        lib().ora_trace_any(self.h, _p(ro), _p(rd), n, _p(occ), _p(cnt), n_threads)
        return (occ, cnt) if counters else occ

    def shadow_batch(self, ro, rd, hits, lights, fog=None, want_transm=False, n_threads=1):
        ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        lights = _f32(lights).reshape(-1, 3)
        n, nl = len(ro), len(lights)
        occ = np.zeros((n, nl), np.uint8)
        tr = np.zeros((n, nl), np.float32) if want_transm else None
        lib().ora_shadow_batch(self.h, _p(ro), _p(rd), _p(hits), n, _p(lights), nl, _p(occ),
                               fog.ref() if fog is not None else None, _p(tr), n_threads)
        return (occ, tr) if want_transm else occ

    def shadow_batch_fused(self, ro, rd, hits, lights, fog=None, n_threads=1):
        """Visibility first, calcTransm only for visible lights (0 for occluded pairs).  Returns
        (occluded, transm, sums) with sums = pairs traced, nodes visited, triangle tests, pairs marched."""
        ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        lights = _f32(lights).reshape(-1, 3)
        n, nl = len(ro), len(lights)
        occ = np.zeros((n, nl), np.uint8)
        tr = np.zeros((n, nl), np.float32)
        sums = np.zeros(4, np.uint64)
        lib().ora_shadow_batch_fused(self.h, _p(ro), _p(rd), _p(hits), n, _p(lights), nl, _p(occ),
                                     fog.ref() if fog is not None else None, _p(tr), _p(sums), n_threads)
        return occ, tr, sums


def max_threads():
    return lib().ora_max_threads()


def rescaled_eps(t):
    return lib().ora_rescaled_eps(float(t))


def bbox_intersect(box, o, d, t_min, t_max):
    box, o, d = _f32(box), _f32(o), _f32(d)
    e, x = C.c_float(), C.c_float()
    r = lib().ora_bbox_intersect(_p(box), _p(o), _p(d), t_min, t_max, C.byref(e), C.byref(x))
    return bool(r), e.value, x.value


def camera_setup(pos, direction, up, h_fov, res_x, res_y):
    out = np.zeros(12, np.float32)
    lib().ora_camera_setup(_p(_f32(pos)), _p(_f32(direction)), _p(_f32(up)), float(np.float32(h_fov)),
                           res_x, res_y, _p(out))
    return out


def primary_rays(cam, res_x, first, n):
    ro = np.zeros((n, 4), np.float32)
    rd = np.zeros((n, 4), np.float32)
    lib().ora_primary_rays(_p(_f32(cam)), res_x, first, n, _p(ro), _p(rd))
    return ro, rd


def sample_density(fog, pos):
    return lib().ora_sample_density(fog.ref(), _p(_f32(pos)))


def transmittance(fog, seg_p, seg_d_distsq, n_threads=1):
    sp, sd = _f32(seg_p).reshape(-1, 4), _f32(seg_d_distsq).reshape(-1, 4)
    out = np.zeros(len(sp), np.float32)
    lib().ora_transmittance(fog.ref(), _p(sp), _p(sd), len(sp), _p(out), n_threads)
    return out


def compute_pi_density(scene, fog, cam, res_x, res_y, n_threads=1):
    out = np.zeros((fog.dens.shape[0], res_y, res_x), np.float32)
    lib().ora_compute_pi_density(scene.h, fog.ref(), _p(_f32(cam)), res_x, res_y, _p(out), n_threads)
    return out


def make_density(res=(64, 64, 64), freq=12.0, ampl=1.0):
    r = (C.c_int * 3)(*res)
    out = np.zeros((res[2], res[1], res[0]), np.uint8)
    lib().ora_make_density(r, freq, ampl, _p(out))
    return out


def photon_trace(scene, materials, fog, abs_k, sca_k, maj_ext_k, fog_enabled, ppl_pos, ppl_intens,
                 shoot_dir, max_vpl_count, rng):
    materials = _f32(materials).reshape(-1, 12)
    rng = _f32(rng)
    vpls = np.zeros(max_vpl_count, VPL_DTYPE)
    n_paths, n_used = C.c_uint32(), C.c_uint64()
    n = lib().ora_photon_trace(scene.h, _p(materials), len(materials), fog.ref() if fog is not None else None,
                               abs_k, sca_k, maj_ext_k, int(fog_enabled), _p(_f32(ppl_pos)),
                               _p(_f32(ppl_intens)), _p(_f32(shoot_dir)), max_vpl_count, _p(rng), len(rng),
                               _p(vpls), C.byref(n_paths), C.byref(n_used))
    return n, vpls, n_paths.value, n_used.value


def photon_trace_substreams(scene, materials, fog, abs_k, sca_k, maj_ext_k, fog_enabled, ppl_pos, ppl_intens,
                            shoot_dir, max_vpl_count, seed):
    """PhotonTracer::trace in substream mode (see ora_photon_trace_substreams): (n, vpls, n_paths, n_draws)."""
    materials = _f32(materials).reshape(-1, 12)
    vpls = np.zeros(max_vpl_count, VPL_DTYPE)
    n_paths, n_draws = C.c_uint32(), C.c_uint64()
    n = lib().ora_photon_trace_substreams(scene.h, _p(materials), len(materials),
                                          fog.ref() if fog is not None else None, abs_k, sca_k, maj_ext_k,
                                          int(fog_enabled), _p(_f32(ppl_pos)), _p(_f32(ppl_intens)),
                                          _p(_f32(shoot_dir)), max_vpl_count, int(seed) & (2**64 - 1), _p(vpls),
                                          C.byref(n_paths), C.byref(n_draws))
    return n, vpls, n_paths.value, n_draws.value


def shade_surface(scene, fog, cam_pos, res_x, res_y, first, ro, rd, hits, materials, ppls, vpls, occluded,
                  transm=None, pi_dens=None, sca_albedo=0.0, gi_enabled=True, clamp_rsq=True, frame_id=0,
                  accum=None, n_threads=1):
    """Surface.frag main() for one frame (see ora_shade_surface).  ppls: (n_ppls, 8) floats; vpls:
    VPL_DTYPE records; occluded/transm: (n_pts, n_ppls + n_vpls) from OracleScene.shadow_batch.
    Returns (accum_rgba, fog_dist)."""
    ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    n = len(ro)
    materials = _f32(materials).reshape(-1, 12)
    ppls = _f32(ppls).reshape(-1, 8)
    vpls = np.ascontiguousarray(vpls, dtype=VPL_DTYPE)
    occluded = np.ascontiguousarray(occluded, dtype=np.uint8).reshape(n, len(ppls) + len(vpls))
    if transm is not None:
        transm = _f32(transm).reshape(n, len(ppls) + len(vpls))
    pi_res_z = 0
    if pi_dens is not None:
        pi_dens = _f32(pi_dens)
        pi_res_z = pi_dens.shape[0]
    accum = np.zeros((n, 4), np.float32) if accum is None else _f32(accum).reshape(n, 4).copy()
    fog_dist = np.zeros((n, 2), np.float32)
    lib().ora_shade_surface(scene.h, fog.ref() if fog is not None else None, _p(_f32(cam_pos)), res_x, res_y, first,
                            _p(ro), _p(rd), _p(hits), n, _p(materials), _p(ppls), len(ppls), _p(vpls), len(vpls),
                            _p(occluded), _p(transm), _p(pi_dens), pi_res_z, sca_albedo, int(gi_enabled),
                            int(clamp_rsq), frame_id, _p(accum), _p(fog_dist), n_threads)
    return accum, fog_dist


def tonemap(accum, frame_id, exposure):
    accum = _f32(accum).reshape(-1, 4)
    rgb = np.zeros((len(accum), 3), np.float32)
    lib().ora_tonemap(_p(accum), len(accum), frame_id, exposure, _p(rgb))
    return rgb


def halton2(n=30 * 32):
    """HaltonSG::generate<2> (Source/Common/Halton.hpp): the volume pass's sample offsets."""
    seq = np.zeros(n, np.float32)
    lib().ora_halton2(_p(seq), n)
    return seq


def volume_samples(fog, cam_pos, res_x, res_y, ro, rd, hits, fog_dist, rnd_offsets, halton, pi_dens,
                   gi_enabled=True, frame_id=0):
    """Sample points of Volume.frag (see ora_volume_samples): (n_pixels * n_samples, 4) = x, y, z, valid."""
    ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    pi_dens = _f32(pi_dens)
    n_samples = 8 if gi_enabled else 32
    pts = np.zeros(((res_x // 2) * (res_y // 2) * n_samples, 4), np.float32)
    lib().ora_volume_samples(fog.ref(), _p(_f32(cam_pos)), res_x, res_y, _p(ro), _p(rd), _p(hits),
                             _p(_f32(fog_dist)), _p(_f32(rnd_offsets)), _p(_f32(halton)), _p(pi_dens),
                             pi_dens.shape[0], int(gi_enabled), frame_id, _p(pts))
    return pts


def shadow_points(scene, points, lights, n_threads=1):
    points, lights = _f32(points).reshape(-1, 4), _f32(lights).reshape(-1, 3)
    occ = np.zeros((len(points), len(lights)), np.uint8)
    lib().ora_shadow_points(scene.h, _p(points), len(points), _p(lights), len(lights), _p(occ), n_threads)
    return occ


def shade_volume(fog, cam_pos, res_x, res_y, ro, rd, hits, fog_dist, rnd_offsets, halton, pi_dens, ppls, vpls,
                 occluded, sca_k, sca_albedo, gi_enabled=True, clamp_rsq=True, transm_opt=True, frame_id=0,
                 n_threads=1):
    """Volume.frag main() (see ora_shade_volume); returns vol_comp (n_half_res_pixels, 4)."""
    ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    pi_dens = _f32(pi_dens)
    ppls = _f32(ppls).reshape(-1, 8)
    vpls = np.ascontiguousarray(vpls, dtype=VPL_DTYPE)
    occluded = np.ascontiguousarray(occluded, dtype=np.uint8)
    out = np.zeros(((res_x // 2) * (res_y // 2), 4), np.float32)
    lib().ora_shade_volume(fog.ref(), _p(_f32(cam_pos)), res_x, res_y, _p(ro), _p(rd), _p(hits), _p(_f32(fog_dist)),
                           _p(_f32(rnd_offsets)), _p(_f32(halton)), _p(pi_dens), pi_dens.shape[0], _p(ppls),
                           len(ppls), _p(vpls), len(vpls), _p(occluded), sca_k, sca_albedo, int(gi_enabled),
                           int(clamp_rsq), int(transm_opt), frame_id, _p(out), n_threads)
    return out


def depth_buffer(cam_pos, cam_dir, t_near, t_far, ro, rd, hits):
    ro, rd = _f32(ro).reshape(-1, 4), _f32(rd).reshape(-1, 4)
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    d = np.zeros(len(ro), np.float32)
    lib().ora_depth_buffer(_p(_f32(cam_pos)), _p(_f32(cam_dir)), t_near, t_far, _p(ro), _p(rd), _p(hits), len(ro), _p(d))
    return d


def combine(res_x, res_y, depth, vol_comp, ext_k, frame_id, exposure, accum):
    """Combine.frag (see ora_combine); returns (updated accum, rgb)."""
    accum = _f32(accum).reshape(-1, 4).copy()
    rgb = np.zeros((res_x * res_y, 3), np.float32)
    lib().ora_combine(res_x, res_y, _p(_f32(depth)), _p(_f32(vol_comp)), ext_k, frame_id, exposure, _p(accum), _p(rgb))
    return accum, rgb
