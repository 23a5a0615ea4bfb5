"""CPU: the oracle restatement against the reference outputs recorded at survey time
(tests/golden/survey_facts.json)."""
import hashlib

import numpy as np

from conftest import fnv1a64_hits
from gigl_amd import scene_io as S
from oracle import oracle_py as O


def test_cornell_load(cornell, facts):
    f = facts["cornell"]
    assert len(cornell["verts"]) == f["n_vertices"]
    assert len(cornell["tris"]) == f["n_triangles"]
    assert cornell["names"] == f["materials"]
    # green wall 0-1, red wall 2-3, white 4-39
    assert cornell["tris"][:, 3].tolist() == [0, 0, 1, 1] + [2] * 36


def test_cornell_tree(oracle_cornell, facts):
    f = facts["cornell"]
    assert oracle_cornell.n_nodes == f["n_nodes"]
    assert oracle_cornell.depth == f["depth"]
    assert round(float(oracle_cornell.avg_prims_per_leaf), 2) == f["avg_prims_per_leaf"]
    assert oracle_cornell.bbox().tolist() == np.array(f["bbox_min"] + f["bbox_max"], np.float32).tolist()


def test_primary_rays_digest(oracle_cornell, cornell_cam, facts):
    f = facts["primary_1024"]
    ro, rd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
    hits = oracle_cornell.trace_closest(ro, rd, n_threads=4)
    hit = hits["tri_id"] != 0xFFFFFFFF
    assert int(hit.sum()) == f["n_hits"]
    c = hits[f["centre_pixel"]["y"] * 1024 + f["centre_pixel"]["x"]]
    assert int(c["tri_id"]) == f["centre_pixel"]["tri_id"]
    assert "%.9g" % c["t"] == "%.9g" % np.float32(f["centre_pixel"]["distance"])
    assert np.bincount(hits["tri_id"][hit], minlength=40).tolist() == f["hit_histogram_by_tri_id"]
    assert fnv1a64_hits(hits) == f["fnv1a64"]["digest"]
    # any-hit returns the same bool as closest-hit (KdTree.hpp:315-332)
    occ = oracle_cornell.trace_any(ro[::7], rd[::7])
    assert np.array_equal(occ.astype(bool), hit[::7])


def test_density_volume(density, fog_box, facts):
    f = facts["fog"]
    blob = np.array([64, 64, 64], "<i4").tobytes() + density.tobytes()
    assert len(blob) == f["df_3dt"]["bytes"]
    assert hashlib.sha256(blob).hexdigest() == f["df_3dt"]["sha256"]
    fog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    ctr = (fog_box[:3] + fog_box[3:]) * np.float32(0.5)
    d = np.float32(O.sample_density(fog, ctr))
    assert "%.9g" % d == "%.9g" % f["density_at_box_centre"]
    assert "%.9g" % (np.float32(fog.ext_k) * d) == "%.9g" % f["ext_k_at_box_centre"]
    alb = np.float32(S.SCA_K) / (np.float32(S.ABS_K) + np.float32(S.SCA_K))
    assert "%.9g" % alb == "%.9g" % f["sca_albedo"]


def test_fog_box_is_tree_box_capped(oracle_cornell, facts):
    # Scene::addFog, Source/Common/Scene.cpp:167-170
    b = oracle_cornell.bbox()
    assert b[:3].tolist() == facts["fog"]["box_min"]
    mx = b[3:].copy()
    mx[1] = min(mx[1], np.float32(S.MAX_FOG_HEIGHT))
    assert mx.tolist() == np.array(facts["fog"]["box_max"], np.float32).tolist()


def test_rescaled_eps_values():
    eps = np.float32(np.finfo(np.float32).eps)
    assert O.rescaled_eps(1.0) == eps * 4          # frexp(1) -> 0.5*2^1, ++exp -> 2
    assert O.rescaled_eps(0.0) == eps * 2
    assert O.rescaled_eps(1092.0847) == eps * 4096
    assert O.rescaled_eps(-3.0) == eps * 8
    assert O.rescaled_eps(0.1) == eps / 4          # frexp -> 0.8*2^-3, ++exp -> -2
    assert O.rescaled_eps(float("inf")) == eps * 2
    assert O.rescaled_eps(2.0 ** 29) < 0           # ++exp = 31: 1 << 31 wraps negative
    assert O.rescaled_eps(2.0 ** 30) == eps        # ++exp = 32: count masked to 5 bits -> 1 << 0


def test_bbox_intersect_cases(oracle_cornell):
    box = oracle_cornell.bbox()
    hit, e, x = O.bbox_intersect(box, [278, 273, -800], [0, 0, 1], 0.0, 3.4e38)
    assert hit and e == 799.0 and x == np.float32(1359.2)
    hit, _, _ = O.bbox_intersect(box, [278, 273, -800], [0, 0, -1], 0.0, 3.4e38)
    assert not hit
    # 0 * inf = NaN when the origin lies on a slab and d is 0 on that axis: all compares false
    hit, e, x = O.bbox_intersect(box, [0, 273, -800], [0, 0, 1], 0.0, 3.4e38)
    assert isinstance(hit, bool)


def test_substream_walk_is_order_independent(cornell, oracle_cornell, density, fog_box):
    """ora_photon_trace_substreams (the mode gigl_photon_trace_batch computes): path p draws from its own
    counter-based stream, so the VPLs of a 150-record walk are the first 150 of a 400-record walk, up to
    the division by the path count (LightArray.hpp:46-66), and another seed gives another set."""
    from gigl_amd import scene_io as S
    from oracle import oracle_py as O
    fog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    args = (oracle_cornell, cornell["materials"], fog, S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS, S.PRIM_PL_INTENS,
            [0, -1, 0])
    n_a, a, paths_a, draws_a = O.photon_trace_substreams(*args, 150, 42)
    n_b, b, paths_b, draws_b = O.photon_trace_substreams(*args, 400, 42)
    assert n_a == 150 and n_b == 400 and paths_a < paths_b and draws_a < draws_b
    for k in ("w_pos", "type", "sca_k", "w_inc", "path_id", "w_norm", "k_d", "k_s", "n_s"):
        assert np.array_equal(a[k], b[k][:150]), k
    scale = np.float32(paths_b) / np.float32(paths_a)
    assert np.allclose(a["intens"], b["intens"][:150] * scale, rtol=1e-6)
    n_c, c, _, _ = O.photon_trace_substreams(*args, 150, 43)
    assert n_c == 150 and not np.array_equal(a["w_pos"], c["w_pos"])
