"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_fixtures.py).

CPU half: the oracle still reproduces them.  GPU half (marked gpu): the HIP engine, through the
C ABI, reproduces them — ids, occlusion bits and tree bit-exact; t/u/v and density bit-exact;
transmittance within 1e-5 relative."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from gigl_amd import rt, scene_io as S
from oracle import oracle_py as O


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _hits(a):
    return np.ascontiguousarray(a).view(O.HIT_DTYPE).reshape(-1)


def test_oracle_reproduces_cornell_fixture(oracle_cornell):
    g = _load("cornell_primary_subsample.npz")
    hits, cnt = oracle_cornell.trace_closest(g["o_tmin"], g["d_tmax"], counters=True)
    assert np.array_equal(hits.view(np.uint32).reshape(-1, 4), g["hits"])
    assert np.array_equal(cnt, g["counters"])
    assert np.array_equal(oracle_cornell.trace_any(g["o_tmin"], g["d_tmax"]), g["occluded"])


def test_oracle_reproduces_s6_fixture():
    g = _load("s6_bounce_rays.npz")
    ora = O.OracleScene(g["verts"], g["tris"], **S.KD_PARAMS)
    assert np.array_equal(ora.nodes(), g["tree_nodes"])
    assert np.array_equal(ora.leaf_prim_ids(), g["tree_leaf_prim_ids"])
    hits, cnt = ora.trace_closest(g["o_tmin"], g["d_tmax"], counters=True)
    assert np.array_equal(hits.view(np.uint32).reshape(-1, 4), g["hits"])
    assert np.array_equal(cnt, g["counters"])
    # host builder of the engine against the frozen tree (no GPU needed)
    ps = rt.Scene(g["verts"], g["tris"], device=-1, **S.KD_PARAMS)
    n, l = ps.export_tree()
    assert np.array_equal(n, g["tree_nodes"]) and np.array_equal(l, g["tree_leaf_prim_ids"])
    ps.close()


def test_oracle_reproduces_fog_and_vpl_fixtures(cornell, oracle_cornell, density):
    g = _load("fog_samples.npz")
    fog = O.Fog(density, g["box"], np.float32(S.ABS_K) + np.float32(S.SCA_K))
    d = np.array([O.sample_density(fog, p) for p in g["pos"]], np.float32)
    assert np.array_equal(d.view(np.uint32), g["density"].view(np.uint32))
    tr = O.transmittance(fog, g["seg_p"], g["seg_d_distsq"])
    assert np.array_equal(tr.view(np.uint32), g["transmittance"].view(np.uint32))
    v = _load("vpl_set.npz")
    n, vpls, n_paths, used = O.photon_trace(oracle_cornell, cornell["materials"], fog, S.ABS_K, S.SCA_K,
                                            S.MAJ_EXT_K, 1, S.PRIM_PL_POS, S.PRIM_PL_INTENS, [0, -1, 0],
                                            S.MAX_N_VPLS, v["rng"])
    assert n == S.MAX_N_VPLS and n_paths == int(v["n_paths"]) and used == len(v["rng"])
    assert np.array_equal(np.frombuffer(vpls.tobytes(), np.uint32).reshape(n, 24), v["vpls"])


@pytest.mark.gpu
def test_engine_reproduces_ray_fixtures(cornell):
    g = _load("cornell_primary_subsample.npz")
    sc = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    got = sc.trace(g["o_tmin"], g["d_tmax"])
    assert np.array_equal(got.view(np.uint32).reshape(-1, 4), g["hits"])
    assert np.array_equal(sc.trace(g["o_tmin"], g["d_tmax"], is_vis_ray=True), g["occluded"])
    sc.close()
    g = _load("s6_bounce_rays.npz")
    sc = rt.Scene(g["verts"], g["tris"], device=0, **S.KD_PARAMS)
    n, l = sc.export_tree()
    assert np.array_equal(n, g["tree_nodes"]) and np.array_equal(l, g["tree_leaf_prim_ids"])
    assert np.array_equal(sc.trace(g["o_tmin"], g["d_tmax"]).view(np.uint32).reshape(-1, 4), g["hits"])
    assert np.array_equal(sc.trace(g["o_tmin"], g["d_tmax"], is_vis_ray=True), g["occluded"])
    sc.close()


@pytest.mark.gpu
def test_engine_reproduces_fog_fixture(density):
    torch = pytest.importorskip("torch")
    g = _load("fog_samples.npz")
    fog = rt.FogVolume(density, g["box"], S.ABS_K, S.SCA_K)
    d = fog.sample_density(g["pos"])
    assert np.array_equal(d.view(np.uint32), g["density"].view(np.uint32))
    n = len(g["seg_p"])
    tp, td = torch.from_numpy(g["seg_p"]).cuda(), torch.from_numpy(g["seg_d_distsq"]).cuda()
    out = torch.zeros(n, dtype=torch.float32, device="cuda:0")
    fog.transmittance_device(tp.data_ptr(), td.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = g["transmittance"].astype(np.float64)
    rel = np.abs(out.cpu().numpy() - want) / np.maximum(np.abs(want), 1e-30)
    assert float(rel.max()) <= 1e-5
    fog.close()
