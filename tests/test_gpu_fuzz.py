"""Randomised parity: triangle soups the Cornell-derived scenes do not look like (overlapping,
sliver, degenerate, axis-aligned and duplicated triangles; tiny and huge extents) under several
builder parameter sets, traced with random rays from inside, outside and on the geometry.  The
bar is the same as everywhere: tree, triangle ids, t/u/v bits and occlusion bits identical to
the oracle's."""
import numpy as np
import pytest

from gigl_amd import rt
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def soup(rng, n, scale, flavour):
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale
    e = rng.normal(0, 1, (n, 3, 3)) * scale * rng.choice([0.02, 0.1, 0.5], (n, 1, 1))
    P = (c + e).astype(np.float32)
    if flavour == "axis":          # axis-aligned rectangles' halves: many coplanar boundaries
        ax = rng.integers(0, 3, n)
        for i in range(n):
            P[i, :, ax[i]] = np.float32(np.round(P[i, 0, ax[i]] / (scale * 0.25)) * scale * 0.25)
    elif flavour == "degenerate":  # zero-area, repeated and sliver triangles mixed in
        P[::7, 2] = P[::7, 1]
        P[3::11] = P[2::11][: len(P[3::11])]
        P[5::13, 2] = P[5::13, 0] + (P[5::13, 1] - P[5::13, 0]) * np.float32(0.5)
    verts = P.reshape(-1, 3)
    tris = np.zeros((n, 4), np.uint32)
    tris[:, :3] = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    tris[:, 3] = rng.integers(0, 3, n)
    return np.ascontiguousarray(verts), tris


def rays_for(rng, verts, tris, n, scale):
    ro = np.zeros((n, 4), np.float32)
    rd = np.zeros((n, 4), np.float32)
    k = n // 4
    ro[:k, :3] = rng.uniform(-1.5, 1.5, (k, 3)) * scale                       # anywhere, also outside the box
    t = rng.integers(0, len(tris), n - k)                                     # on the geometry
    w = rng.dirichlet([1, 1, 1], n - k).astype(np.float32)
    ro[k:, :3] = (verts[tris[t, :3].astype(np.int64)] * w[:, :, None]).sum(1)
    d = rng.normal(0, 1, (n, 3))
    d[::17, rng.integers(0, 3)] = 0.0                                         # axis-parallel components (inv_d = inf)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    rd[:, :3] = d
    ro[:, 3] = rng.choice([0.0, 1e-4 * scale, 0.3 * scale], n).astype(np.float32)
    rd[:, 3] = rng.choice([np.finfo(np.float32).max, 0.7 * scale, 3.0 * scale], n).astype(np.float32)
    return ro, rd


CASES = [
    (1, 600, 1.0, "plain", dict(inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2, empty_bonus=0.25)),
    (2, 900, 300.0, "axis", dict(inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2, empty_bonus=0.25)),
    (3, 700, 1e-3, "degenerate", dict(inters_cost=10, trav_cost=1, max_depth=30, min_num_prims=2, empty_bonus=0.25)),
    (4, 1500, 5e4, "plain", dict(inters_cost=80, trav_cost=1, max_depth=18, min_num_prims=4, empty_bonus=0.5)),
    (5, 400, 10.0, "axis", dict(inters_cost=2, trav_cost=5, max_depth=8, min_num_prims=1, empty_bonus=0.0)),
    (6, 5000, 50.0, "degenerate", dict(inters_cost=10, trav_cost=1, max_depth=31, min_num_prims=2, empty_bonus=0.25)),
]


@pytest.mark.parametrize("seed,n_tris,scale,flavour,params", CASES)
def test_random_soup(seed, n_tris, scale, flavour, params):
    rng = np.random.default_rng(1000 + seed)
    verts, tris = soup(rng, n_tris, scale, flavour)
    ora = O.OracleScene(verts, tris, **params)
    gpu = rt.Scene(verts, tris, device=0, **params)
    nodes, leaf_ids = gpu.export_tree()
    assert np.array_equal(nodes, ora.nodes()) and np.array_equal(leaf_ids, ora.leaf_prim_ids())
    ro, rd = rays_for(rng, verts, tris, 30000, scale)
    want = ora.trace_closest(ro, rd, n_threads=4)
    got = gpu.trace(ro, rd)
    assert np.array_equal(got["tri_id"], want["tri_id"])
    for k in ("t", "u", "v"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert 0.02 < (want["tri_id"] != rt.GIGL_MISS).mean() < 0.999
    occ = ora.trace_any(ro, rd, n_threads=4)
    assert np.array_equal(gpu.trace(ro, rd, is_vis_ray=True), occ.astype(bool))
    gpu.close()
    # the same scene with leaf triangle blocks (the default of scenes beyond the Infinity Cache's size)
    try:
        rt.set_option("leaf_blocks", 2)
        gpu = rt.Scene(verts, tris, device=0, **params)
        got = gpu.trace(ro, rd)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(gpu.trace(ro, rd, is_vis_ray=True), occ.astype(bool))
        gpu.close()
    finally:
        rt.set_option("leaf_blocks", -1)
    ora.close()


@pytest.mark.parametrize("flavour", ["bits", "one_component", "specials"])
def test_rays_of_arbitrary_bit_patterns(cornell, flavour):
    """Rays nobody would write: random bit patterns (NaN, infinities, denormals, huge values) in all or in one
    of the eight ray words, or special values dropped into sane rays.  The traversal has to terminate and to
    agree with the oracle: ids and occlusion bits exactly, t/u/v bit for bit or NaN on both sides."""
    from gigl_amd import scene_io as S, synth
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    gpu = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    ora = O.OracleScene(V, T, **S.KD_PARAMS)
    rng = np.random.default_rng(20261111)
    n = 60_000
    ro, rd = synth.bounce_rays(V, T, n, 0x6A7BA6E)
    both = np.concatenate([ro, rd], 1)
    rnd = lambda shape: rng.integers(0, 2 ** 32, shape, dtype=np.uint64).astype(np.uint32).view(np.float32)
    if flavour == "bits":
        both = rnd((n, 8))
    elif flavour == "one_component":
        both[np.arange(n), rng.integers(0, 8, n)] = rnd(n)
    else:
        specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38, 1e-38], np.float32)
        for _ in range(2):
            both[np.arange(n), rng.integers(0, 8, n)] = specials[rng.integers(0, len(specials), n)]
    ro, rd = np.ascontiguousarray(both[:, :4]), np.ascontiguousarray(both[:, 4:])
    with np.errstate(all="ignore"):
        want = ora.trace_closest(ro, rd, n_threads=4)
        occ = ora.trace_any(ro, rd, n_threads=4)
    got = gpu.trace(ro, rd)
    assert np.array_equal(got["tri_id"], want["tri_id"])
    hit = got["tri_id"] != rt.GIGL_MISS
    assert hit.sum() > (500 if flavour == "bits" else 5000)
    for k in ("t", "u", "v"):
        a, b = got[k][hit], want[k][hit]
        assert bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()), k
    assert np.array_equal(gpu.trace(ro, rd, is_vis_ray=True), occ.astype(bool))
    gpu.close()
    ora.close()
