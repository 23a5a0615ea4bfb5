"""CPU checks of the oracle's shading-stage restatement (ora_shade_surface, ora_tonemap): the
properties Surface.frag's structure guarantees.  PARITY UNPINNED against the reference (no GLSL
here, no reference image)."""
import os

import numpy as np

from gigl_amd import scene_io as S
from oracle import oracle_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(cornell, oracle_cornell, density, fog_box, res=48):
    cam = O.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, res, res)
    ro, rd = O.primary_rays(cam, res, 0, res * res)
    hits = oracle_cornell.trace_closest(ro, rd, n_threads=4)
    vp = np.load(os.path.join(GOLDEN, "vpl_set.npz"))
    vpls = vp["vpls"].view(O.VPL_DTYPE).reshape(-1)
    ppls = np.zeros((1, 8), np.float32)
    ppls[0, :3], ppls[0, 4:7] = S.PRIM_PL_POS, S.PRIM_PL_INTENS
    lights = np.concatenate([ppls[:, :3], vp["vpls"].view(np.float32)[:, :3]]).astype(np.float32)
    ext_k = np.float32(S.ABS_K) + np.float32(S.SCA_K)
    fog = O.Fog(density, fog_box, ext_k)
    occ, tr = oracle_cornell.shadow_batch(ro, rd, hits, lights, fog=fog, want_transm=True, n_threads=4)
    pi = O.compute_pi_density(oracle_cornell, fog, cam, res, res, n_threads=4)
    return dict(res=res, ro=ro, rd=rd, hits=hits, vpls=vpls, ppls=ppls, occ=occ, tr=tr, pi=pi, fog=fog,
                albedo=float(np.float32(S.SCA_K) / ext_k))


def test_shade_surface_properties(cornell, oracle_cornell, density, fog_box):
    c = _setup(cornell, oracle_cornell, density, fog_box)
    res, n = c["res"], c["res"] ** 2
    args = (oracle_cornell, c["fog"], S.CAM_POS, res, res, 0, c["ro"], c["rd"], c["hits"], cornell["materials"],
            c["ppls"], c["vpls"], c["occ"])
    kw = dict(transm=c["tr"], pi_dens=c["pi"], sca_albedo=c["albedo"], n_threads=4)
    full, fd = O.shade_surface(*args, **kw)
    assert full.shape == (n, 4) and (full[:, 3] == 1.0).all() and np.isfinite(full).all()
    assert (full[:, :3] >= 0).all()
    missed = c["hits"]["tri_id"] == 0xFFFFFFFF
    assert missed.any() and (full[missed, :3] == 0).all() and (fd[missed] == 0).all()
    assert (full[~missed, :3].sum(1) > 0).mean() > 0.9
    # primary-light and VPL contributions are gated separately and add up (Surface.frag:319-330)
    direct, _ = O.shade_surface(*args, gi_enabled=False, **kw)
    indirect, _ = O.shade_surface(*args, clamp_rsq=False, **kw)
    assert (direct[:, :3] <= full[:, :3] * (1 + 1e-5) + 1e-12).all()
    assert np.allclose(direct[:, :3].astype(np.float64) + indirect[:, :3], full[:, :3], rtol=1e-4, atol=1e-9)
    assert direct[:, :3].sum() > 0 and indirect[:, :3].sum() > 0
    # fog only attenuates; without fog nothing is recorded in fog_dist
    clear, fd0 = O.shade_surface(oracle_cornell, None, *args[2:], sca_albedo=c["albedo"], n_threads=4)
    assert (fd0 == 0).all() and (clear[:, :3] >= full[:, :3] * (1 - 1e-5)).all()
    assert (fd[:, 1] >= fd[:, 0]).all() and (fd[:, 1] > 0).any()
    # frame accumulation (Surface.frag:332-339) and the frame cap (:292)
    prev = np.random.default_rng(1).uniform(0, 3, (n, 4)).astype(np.float32)
    acc, _ = O.shade_surface(*args, frame_id=2, accum=prev, **kw)
    assert np.array_equal(acc[:, :3], full[:, :3] + prev[:, :3])
    frozen, _ = O.shade_surface(*args, frame_id=30, accum=prev, **kw)
    assert np.array_equal(frozen, prev)


def test_tonemap():
    a = np.array([[0, 1, 2, 1], [0.5, 0.25, 4, 1]], np.float32)
    rgb = O.tonemap(a, 3, 2)
    want = 1.0 - np.exp(-2.0 * a[:, :3].astype(np.float64) / 4.0)
    assert np.allclose(rgb, want, rtol=1e-6, atol=1e-7)
    assert np.array_equal(O.tonemap(a, 40, 2), O.tonemap(a, 30, 2))


def test_halton_base2():
    h = O.halton2(30 * 32)
    assert h.dtype == np.float32 and len(h) == 960
    assert np.array_equal(h[:7], np.array([0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875], np.float32))
    assert (h > 0).all() and (h < 1).all() and len(np.unique(h)) == 960     # radical inverse: all distinct


def test_shade_volume_properties(cornell, oracle_cornell, density, fog_box):
    c = _setup(cornell, oracle_cornell, density, fog_box)
    res, n = c["res"], c["res"] ** 2
    _, fd = O.shade_surface(oracle_cornell, c["fog"], S.CAM_POS, res, res, 0, c["ro"], c["rd"], c["hits"],
                            cornell["materials"], c["ppls"], c["vpls"], c["occ"], transm=c["tr"], pi_dens=c["pi"],
                            sca_albedo=c["albedo"], n_threads=4)
    hw = res // 2
    rnd = (np.float32(0.03) * np.random.default_rng(3).random(hw * hw, dtype=np.float32) - np.float32(0.015))
    hal = O.halton2()
    lights = np.concatenate([c["ppls"][:, :3], c["vpls"].view(np.float32).reshape(-1, 24)[:, :3]]).astype(np.float32)
    pts = O.volume_samples(c["fog"], S.CAM_POS, res, res, c["ro"], c["rd"], c["hits"], fd, rnd, hal, c["pi"])
    assert pts.shape == (hw * hw * 8, 4)
    valid = pts[:, 3] > 0
    assert 0.3 < valid.mean() <= 1.0
    box = np.asarray(fog_box)
    eps = 1e-3 * (box[3:] - box[:3])
    assert ((pts[valid, :3] >= box[:3] - eps) & (pts[valid, :3] <= box[3:] + eps)).all()   # samples lie in the fog box
    assert (pts[~valid] == 0).all()
    # pixels whose full-resolution texel (2x+1, 2y+1) recorded no fog interval take no sample
    fd_half = fd.reshape(res, res, 2)[1::2, 1::2].reshape(-1, 2)
    assert (~valid.reshape(-1, 8)[fd_half[:, 1] <= 0]).all()
    occ = O.shadow_points(oracle_cornell, pts, lights, n_threads=4)
    assert (occ[~valid] == 0xFF).all() and set(np.unique(occ[valid])) <= {0, 1}
    assert 0.02 < occ[valid].mean() < 0.98
    kw = dict(sca_k=float(np.float32(S.SCA_K)), sca_albedo=c["albedo"], n_threads=4)
    args = (c["fog"], S.CAM_POS, res, res, c["ro"], c["rd"], c["hits"], fd, rnd, hal, c["pi"], c["ppls"], c["vpls"], occ)
    vol = O.shade_volume(*args, **kw)
    assert vol.shape == (hw * hw, 4) and np.isfinite(vol).all() and (vol[:, :3] >= 0).all() and (vol[:, 3] == 1).all()
    lit = vol[:, :3].sum(1) > 0
    assert lit.any() and not lit[fd_half[:, 1] <= 0].any()
    # the constant-medium shortcut and the ray march are two estimators of the same transmittance:
    # same lit pixels, totals of the same magnitude, not the same numbers
    marched = O.shade_volume(*args, transm_opt=False, **kw)
    assert np.array_equal(marched[:, :3].sum(1) > 0, lit)
    assert 0.3 < marched[:, :3].sum() / vol[:, :3].sum() < 3.0 and not np.array_equal(marched, vol)
    # everything occluded -> nothing scattered in
    dark = O.shade_volume(*args[:-1], np.ones_like(occ), **kw)
    assert (dark[:, :3] == 0).all()
    # frame cap (Volume.frag:270)
    assert (O.shade_volume(*args, frame_id=29, **kw)[:, :3].sum() > 0)


def test_combine_properties():
    res, hw = 16, 8
    rng = np.random.default_rng(4)
    accum = rng.uniform(0, 1, (res * res, 4)).astype(np.float32)
    depth = np.full(res * res, 0.5, np.float32)
    vol = np.zeros((hw * hw, 4), np.float32)
    vol[:, :3] = [0.25, 0.5, 1.0]
    acc, rgb = O.combine(res, res, depth, vol, 0.01, 0, 3, accum)
    # a constant volume term on a constant depth is reproduced exactly away from the zero border
    inner = np.zeros((res, res), bool)
    inner[2:-2, 2:-2] = True
    inner = inner.reshape(-1)
    assert np.allclose(acc[inner, :3] - accum[inner, :3], [0.25, 0.5, 1.0], rtol=1e-5)
    assert (acc[~inner, :3] - accum[~inner, :3] <= np.array([0.25, 0.5, 1.0]) * (1 + 1e-5)).all()
    assert np.allclose(rgb, 1.0 - np.exp(-3.0 * acc[:, :3].astype(np.float64)), rtol=1e-5, atol=1e-6)
    # no fog or past the last frame: the buffer is left alone, the image is still tone-mapped
    for ext_k, frame in ((0.0, 0), (0.01, 30)):
        acc2, rgb2 = O.combine(res, res, depth, vol, ext_k, frame, 3, accum)
        assert np.array_equal(acc2, accum)
        assert np.allclose(rgb2, 1.0 - np.exp(-3.0 * accum[:, :3].astype(np.float64) / (1 + min(frame, 30))), rtol=1e-5, atol=1e-6)
    # a neighbour that lies deeper counts for less (bilateral weight, Combine.frag:41-42)
    vol2 = np.zeros((hw * hw, 4), np.float32)
    vol2.reshape(hw, hw, 4)[:, 4:, :3] = 1.0
    d2 = depth.copy().reshape(res, res)
    d2[:, 8:] = 0.6
    flat, _ = O.combine(res, res, depth, vol2, 0.01, 0, 3, np.zeros_like(accum))
    edge, _ = O.combine(res, res, d2.reshape(-1), vol2, 0.01, 0, 3, np.zeros_like(accum))
    col = 7   # last column on the near side of the depth step
    assert (edge.reshape(res, res, 4)[2:-2, col, 0] < flat.reshape(res, res, 4)[2:-2, col, 0]).all()
