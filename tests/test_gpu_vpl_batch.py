"""GPU, configs[0] as ray batches: gigl_photon_trace_batch (rt::PhotonTracer::trace with one closest-hit
batch per bounce and one lane per path) against the oracle's serial walk in the same substream mode
(ora_photon_trace_substreams), all 96 bytes of every VPL record, bit for bit.

PARITY UNPINNED against the reference: the substream mode is the build's own definition (the reference
draws from one global generator, include/gigl_rt.h), and no reference VPL set was ever recorded."""
import numpy as np
import pytest

from gigl_amd import rt, scene_io as S
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["batched", "one_launch"])
def route(request):
    """Every test runs through both routes of gigl_photon_trace_batch: the chain of batched launches, and (sets of
    up to 1,024 VPLs) the one-workgroup kernel that walks a whole set in one launch.  Larger sets take the batched
    route under either setting."""
    rt.set_option("photon_small", request.param)
    yield request.param
    rt.set_option("photon_small", -1)


@pytest.fixture(scope="module")
def setup(cornell, oracle_cornell, density, fog_box):
    gpu = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    gfog = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    ofog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    yield {"gpu": gpu, "gfog": gfog, "ora": oracle_cornell, "ofog": ofog, "mats": cornell["materials"]}
    gfog.close()
    gpu.close()


def _both(su, fog_on, n_vpls, seed, per=0, mats=None, pos=S.PRIM_PL_POS):
    mats = su["mats"] if mats is None else mats
    n, want, want_paths, _ = O.photon_trace_substreams(su["ora"], mats, su["ofog"], S.ABS_K, S.SCA_K, S.MAJ_EXT_K,
                                                       fog_on, pos, S.PRIM_PL_INTENS, [0, -1, 0], n_vpls, seed)
    g_n, got, got_paths = su["gpu"].photon_trace_batch(mats, su["gfog"], S.ABS_K, S.SCA_K, S.MAJ_EXT_K, fog_on, pos,
                                                       S.PRIM_PL_INTENS, [0, -1, 0], n_vpls, seed, per)
    return (n, np.frombuffer(want.tobytes(), np.uint32).reshape(-1, 24), want_paths), (g_n, got, got_paths)


@pytest.mark.parametrize("fog_on", [1, 0])
@pytest.mark.parametrize("seed", [1, 20261110, 2 ** 63 + 12345])
def test_vpl_batch_matches_oracle(setup, fog_on, seed):
    (n, want, want_paths), (g_n, got, got_paths) = _both(setup, fog_on, S.MAX_N_VPLS, seed)
    assert n == S.MAX_N_VPLS == g_n
    assert got_paths == want_paths
    bad = np.nonzero((got != want).any(1))[0]
    assert len(bad) == 0, "first differing VPL %d:\n%s\n%s" % (bad[0], got[bad[0]], want[bad[0]])
    types = np.bincount(got[:, 3], minlength=3)
    assert types[2] > 0 and (fog_on == 0 or types[1] > 0)


def test_vpl_batch_does_not_depend_on_the_block_size(setup):
    """Paths are independent: 256 paths per block (several blocks, the record count reached inside one of
    them), one block for everything, and the default give the same records and the same path count."""
    ref = None
    for per in (256, 512, 0, 1 << 16):
        (n, want, want_paths), (g_n, got, got_paths) = _both(setup, 1, 700, 77, per)
        assert g_n == n == 700 and got_paths == want_paths and np.array_equal(got, want)
        ref = got if ref is None else ref
        assert np.array_equal(ref, got)


def test_vpl_batch_large_set(setup):
    (n, want, want_paths), (g_n, got, got_paths) = _both(setup, 1, 20000, 5)
    assert g_n == n == 20000 and got_paths == want_paths
    assert np.array_equal(got, want)
    # path ids are renumbered 0, 1, 2, ... in path order (LightArray.hpp:46-66)
    ids = got[:, 11].astype(np.int64)
    assert ids[0] == 0 and np.all(np.diff(ids) >= 0) and np.all(np.diff(ids) <= 1)


def test_vpl_batch_gives_up_like_the_reference(setup):
    """PhotonTracer.cpp:180-184: when 100 * max_vpl_count paths do not supply the VPLs the array is cleared.
    Every material emissive: every surface hit ends its path (:144-146); no fog: no volume events."""
    mats = np.array(setup["mats"], np.float32).copy()
    mats[:, 8:11] = 1.0
    (n, _, want_paths), (g_n, got, got_paths) = _both(setup, 0, 40, 3, mats=mats)
    assert n == 0 and g_n == 0
    assert got_paths == want_paths == 4000
    assert not got.any()


def test_vpl_batch_rejects_bad_arguments(setup):
    su = setup
    with pytest.raises(rt.GiglError):
        su["gpu"].photon_trace_batch(su["mats"], su["gfog"], S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS,
                                     S.PRIM_PL_INTENS, [0, -1, 0], 0, 1)
    with pytest.raises(rt.GiglError):   # a material table shorter than the ids the scene's triangles carry
        su["gpu"].photon_trace_batch(np.array(su["mats"], np.float32)[:1], su["gfog"], S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 0,
                                     S.PRIM_PL_POS, S.PRIM_PL_INTENS, [0, -1, 0], 50, 1)


@pytest.mark.parametrize("leaf_blocks", [-1, 2])
def test_vpl_batch_on_a_tessellated_scene(cornell, density, fog_box, leaf_blocks):
    """The same walk through a deep tree (S(24): 23,040 displaced triangles, stack spills, with and
    without leaf triangle blocks): hit triangles, their normals and materials come from other code
    paths than in the 40-triangle box."""
    from gigl_amd import synth
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 24)
    try:
        rt.set_option("leaf_blocks", leaf_blocks)
        gpu = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    finally:
        rt.set_option("leaf_blocks", -1)
    ora = O.OracleScene(V, T, **S.KD_PARAMS)
    gfog = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    ofog = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    su = {"gpu": gpu, "gfog": gfog, "ora": ora, "ofog": ofog, "mats": cornell["materials"]}
    try:
        for fog_on, n_vpls, seed in ((1, 600, 11), (0, 3000, 12)):
            (n, want, want_paths), (g_n, got, got_paths) = _both(su, fog_on, n_vpls, seed)
            assert n == n_vpls == g_n and got_paths == want_paths
            assert np.array_equal(got, want)
    finally:
        gfog.close()
        gpu.close()


def test_default_route_is_one_launch_for_the_reference_sized_set(setup):
    """Option photon_small = -1 (the default): a 150-VPL set asked for without a block size goes through the
    one-launch kernel, a block size keeps the batched route; records are the same, and calls of the two routes
    may alternate on one scene (they share the workspace and its work counter)."""
    rt.set_option("photon_small", -1)
    seen = []
    for per in (0, 1024, 0, 256, 0):
        (n, want, want_paths), (g_n, got, got_paths) = _both(setup, 1, S.MAX_N_VPLS, 4242, per)
        assert g_n == n == S.MAX_N_VPLS and got_paths == want_paths and np.array_equal(got, want)
        seen.append(got)
    assert all(np.array_equal(seen[0], g) for g in seen)
    # several blocks of 1,024 paths inside the one-launch kernel: 1,000 VPLs need about 3,000 paths
    rt.set_option("photon_small", 1)
    (n, want, want_paths), (g_n, got, got_paths) = _both(setup, 1, 1000, 9)
    assert g_n == n == 1000 and got_paths == want_paths > 2048 and np.array_equal(got, want)


@pytest.mark.parametrize("n_vpls", [1, 7, 1024, 1025])
def test_set_sizes_at_the_edges_of_the_one_launch_route(setup, n_vpls):
    """One record (100 paths at most, PhotonTracer.cpp:180), the largest set the one-launch kernel takes, and the first
    size beyond it (batched under either setting of the option)."""
    (n, want, want_paths), (g_n, got, got_paths) = _both(setup, 1, n_vpls, 31337 + n_vpls)
    assert g_n == n == n_vpls and got_paths == want_paths and np.array_equal(got, want)
