"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): triangle ids and occlusion bits bit-exact; distance,
barycentrics and transmittance within 1e-5 relative (in practice the float results are
bit-identical too, and the tests say so where that holds).
"""
import numpy as np
import pytest

from conftest import fnv1a64_hits
from gigl_amd import rt, scene_io as S, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

FLT_MAX = np.finfo(np.float32).max
REL_TOL = 1e-5


@pytest.fixture(scope="module")
def gpu_cornell(cornell):
    sc = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    yield sc
    sc.close()


@pytest.fixture(scope="module")
def s16(cornell):
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    o = O.OracleScene(V, T, **S.KD_PARAMS)
    yield {"V": V, "T": T, "gpu": g, "ora": o}
    g.close()


def assert_hits_equal(got, want):
    assert np.array_equal(got["tri_id"], want["tri_id"]), \
        "id mismatches: %d" % int((got["tri_id"] != want["tri_id"]).sum())
    for k in ("t", "u", "v"):
        a, b = got[k].astype(np.float64), want[k].astype(np.float64)
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
        assert float(rel.max(initial=0.0)) <= REL_TOL, k
        # stronger than the bar: same bits
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k + " bits differ"


def test_rescaled_eps_device():
    t = np.concatenate([
        np.array([0.0, -0.0, 1.0, -3.0, 0.1, 1092.0847, 1e-38, 1e-40, 1e-45, 3e38, np.inf, -np.inf, np.nan,
                  2.0 ** 29, 2.0 ** 30, 2.0 ** 31, 2.0 ** 60, 2.0 ** -30, 2.0 ** -32, 2.0 ** -33, 2.0 ** -126],
                 np.float32),
        np.random.default_rng(1).standard_normal(4096).astype(np.float32) * np.float32(1e3),
        np.exp(np.random.default_rng(2).uniform(-80, 80, 4096)).astype(np.float32)])
    got = rt.debug_rescaled_eps(t)
    want = np.array([O.rescaled_eps(x) for x in t], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_tree_matches_oracle(gpu_cornell, oracle_cornell, facts):
    nodes, leaf = gpu_cornell.export_tree()
    assert np.array_equal(nodes, oracle_cornell.nodes())
    assert np.array_equal(leaf, oracle_cornell.leaf_prim_ids())
    st = gpu_cornell.stats()
    assert st.n_nodes == facts["cornell"]["n_nodes"] and st.actual_depth == facts["cornell"]["depth"]
    assert np.array_equal(gpu_cornell.bbox(), oracle_cornell.bbox())


def test_cornell_primary_closest_digest(gpu_cornell, oracle_cornell, cornell_cam, facts):
    """Config 2, first half: 1024^2 primary rays; pinned by the reference's own digest."""
    ro, rd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
    got = gpu_cornell.trace(ro, rd)
    assert int((got["tri_id"] != rt.GIGL_MISS).sum()) == facts["primary_1024"]["n_hits"]
    assert fnv1a64_hits(got) == facts["primary_1024"]["fnv1a64"]["digest"]
    want = oracle_cornell.trace_closest(ro, rd, n_threads=4)
    assert_hits_equal(got, want)


def test_cornell_primary_any(gpu_cornell, oracle_cornell, cornell_cam):
    ro, rd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
    n = 300_001   # ragged: not a multiple of 64
    got = gpu_cornell.trace(ro[:n], rd[:n], is_vis_ray=True)
    want = oracle_cornell.trace_any(ro[:n], rd[:n], n_threads=4)
    assert np.array_equal(got, want)


def test_fused_primary_generator(gpu_cornell, oracle_cornell, cornell_cam):
    torch = pytest.importorskip("torch")
    n, first = 200_000, 333_333
    hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
    o = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
    d = torch.zeros((n, 4), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    gpu_cornell.trace_primary_device(cornell_cam, 1024, first, n, hits.data_ptr(), st)
    gpu_cornell.gen_primary_rays_device(cornell_cam, 1024, first, n, o.data_ptr(), d.data_ptr(), st)
    torch.cuda.synchronize()
    ro, rd = O.primary_rays(cornell_cam, 1024, first, n)
    assert np.array_equal(o.cpu().numpy().view(np.uint32), ro.view(np.uint32))
    assert np.array_equal(d.cpu().numpy().view(np.uint32), rd.view(np.uint32))
    got = hits.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
    assert_hits_equal(got, oracle_cornell.trace_closest(ro, rd, n_threads=4))


def test_incoherent_bounce_rays_s16(s16):
    """Config 3 at reduced tessellation: incoherent cosine-weighted bounce rays."""
    ro, rd = synth.bounce_rays(s16["V"], s16["T"], 1 << 17, synth.RAY_SEED_16M)
    got = s16["gpu"].trace(ro, rd)
    want = s16["ora"].trace_closest(ro, rd, n_threads=4)
    assert_hits_equal(got, want)
    occ = s16["gpu"].trace(ro, rd, is_vis_ray=True)
    assert np.array_equal(occ, s16["ora"].trace_any(ro, rd, n_threads=4))
    assert np.array_equal(occ.astype(bool), want["tri_id"] != rt.GIGL_MISS)


@pytest.mark.parametrize("stack_lds,refill,any_atomic,t_exit,sort_rays,sort_bits,guided,chunk",
                         [(8, 1, 1, 1, 0, 6, 0, 64), (8, 64, 0, 64, 0, 6, 64, 256), (12, 16, 1, 16, 1, 4, 1, 1024),
                          (16, 32, 0, 8, 1, 5, 4, 128), (32, 8, 1, 32, 1, 6, 8, 256), (12, 32, -1, 16, -1, 6, 4, 256)])
def test_launch_options_do_not_change_results(s16, stack_lds, refill, any_atomic, t_exit, sort_rays, sort_bits, guided,
                                              chunk):
    """LDS stack depth (8 forces the global spill path: tree depth 31), refill threshold, T-phase exit
    threshold, the any-hit output mode and ray regrouping are scheduling choices only."""
    ro, rd = synth.bounce_rays(s16["V"], s16["T"], 50_001, 0xABCD01)
    want = s16["ora"].trace_closest(ro, rd, n_threads=4)
    try:
        rt.set_option("stack_lds", stack_lds)
        rt.set_option("refill_thresh", refill)
        rt.set_option("any_atomic", any_atomic)
        rt.set_option("t_exit", t_exit)
        rt.set_option("sort_rays", sort_rays)
        rt.set_option("sort_bits", sort_bits)
        rt.set_option("guided", guided)
        rt.set_option("chunk", chunk)
        assert_hits_equal(s16["gpu"].trace(ro, rd), want)
        occ = s16["gpu"].trace(ro, rd, is_vis_ray=True)
        assert np.array_equal(occ.astype(bool), want["tri_id"] != rt.GIGL_MISS)
    finally:
        rt.set_option("stack_lds", 8)
        rt.set_option("refill_thresh", 0)
        rt.set_option("any_atomic", -1)
        rt.set_option("t_exit", 20)
        rt.set_option("sort_rays", 0)
        rt.set_option("sort_bits", 6)
        rt.set_option("guided", -1)
        rt.set_option("chunk", 0)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("stack_lds", [8, 16])
def test_device_layouts_do_not_change_results(cornell, cornell_cam, oracle_cornell, layout, stack_lds):
    """Level treelets, clustered lines and triple records (two levels per load, its own kernel) hold the
    same tree: closest-hit records and occlusion bits are the oracle's under every layout, with the
    LDS stack short enough to spill (8) and not."""
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    ora = O.OracleScene(V, T, **S.KD_PARAMS)
    try:
        rt.set_option("layout", layout)
        rt.set_option("stack_lds", stack_lds)
        g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
        gc = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
        ro, rd = synth.bounce_rays(V, T, 100_003, 0xBEEF02)
        want = ora.trace_closest(ro, rd, n_threads=4)
        assert_hits_equal(g.trace(ro, rd), want)
        for any_atomic in (0, 1):
            rt.set_option("any_atomic", any_atomic)
            assert np.array_equal(g.trace(ro, rd, is_vis_ray=True), ora.trace_any(ro, rd, n_threads=4))
        po, pd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
        sel = slice(0, 1024 * 1024, 7)
        assert_hits_equal(gc.trace(po[sel], pd[sel]), oracle_cornell.trace_closest(po[sel], pd[sel], n_threads=4))
        assert np.array_equal(gc.trace(po[sel], pd[sel], is_vis_ray=True),
                              oracle_cornell.trace_any(po[sel], pd[sel], n_threads=4))
        g.close()
        gc.close()
    finally:
        rt.set_option("layout", 2)
        rt.set_option("stack_lds", 8)
        rt.set_option("any_atomic", -1)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("params", [dict(inters_cost=10, trav_cost=1, max_depth=4, min_num_prims=2, empty_bonus=0.25),
                                    dict(inters_cost=80, trav_cost=1, max_depth=-1, min_num_prims=1, empty_bonus=0.5),
                                    dict(inters_cost=1, trav_cost=10, max_depth=12, min_num_prims=9, empty_bonus=0.0)])
def test_other_tree_shapes(cornell, layout, params):
    """Shallow trees with hundreds of prims per leaf (id arrays longer than a cache line, the overflow
    area of the layouts), deep trees from the automatic depth rule, and fat leaves: same hits as the oracle."""
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 12)
    ora = O.OracleScene(V, T, **params)
    try:
        rt.set_option("layout", layout)
        g = rt.Scene(V, T, device=0, **params)
        assert g.check_layout(layout) > 0
        ro, rd = synth.bounce_rays(V, T, 40_001, 0xF00D03)
        want = ora.trace_closest(ro, rd, n_threads=4)
        assert_hits_equal(g.trace(ro, rd), want)
        assert np.array_equal(g.trace(ro, rd, is_vis_ray=True), ora.trace_any(ro, rd, n_threads=4))
        g.close()
    finally:
        rt.set_option("layout", 2)


@pytest.mark.parametrize("block_min", [2, 4])
@pytest.mark.parametrize("params", [S.KD_PARAMS,
                                    dict(inters_cost=10, trav_cost=1, max_depth=6, min_num_prims=2, empty_bonus=0.25),
                                    dict(inters_cost=1, trav_cost=10, max_depth=14, min_num_prims=9, empty_bonus=0.0)])
def test_leaf_triangle_blocks(cornell, cornell_cam, oracle_cornell, block_min, params):
    """Triple records whose leaves read private blocks of triangle copies (three 40-byte records per
    line, in list order) instead of id lists and the shared table - what scenes beyond the Infinity
    Cache's size get by default.  Same tree, same triangles in the same order: records and bits are
    the oracle's, on deep trees, on shallow trees with hundreds of prims per leaf, and on Cornell."""
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 14)
    ora = O.OracleScene(V, T, **params)
    try:
        rt.set_option("leaf_blocks", block_min)
        g = rt.Scene(V, T, device=0, **params)
        gc = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
        assert g.check_layout(3) > g.check_layout(2)
        ro, rd = synth.bounce_rays(V, T, 80_001, 0xB10C05)
        want = ora.trace_closest(ro, rd, n_threads=4)
        assert_hits_equal(g.trace(ro, rd), want)
        for any_atomic in (0, 1):
            rt.set_option("any_atomic", any_atomic)
            assert np.array_equal(g.trace(ro, rd, is_vis_ray=True), ora.trace_any(ro, rd, n_threads=4))
        po, pd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
        sel = slice(0, 1024 * 1024, 11)
        assert_hits_equal(gc.trace(po[sel], pd[sel]), oracle_cornell.trace_closest(po[sel], pd[sel], n_threads=4))
        hit, rec, nrm = gc.trace_one([278, 273, -800], [0, 0, 1])
        w_hit, w_rec, w_nrm, _ = oracle_cornell.trace_one([278, 273, -800], [0, 0, 1])
        assert hit == w_hit and rec["tri_id"] == w_rec["tri_id"] and np.array_equal(nrm.view(np.uint32), w_nrm.view(np.uint32))
        g.close()
        gc.close()
    finally:
        rt.set_option("leaf_blocks", -1)
        rt.set_option("any_atomic", -1)


def test_edge_cases(gpu_cornell, oracle_cornell):
    # empty batch
    assert len(gpu_cornell.trace(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))) == 0
    inf = np.float32(np.inf)
    rays = [
        # o, tmin, d, tmax
        ([278, 273, -800], 0, [0, 0, 1], FLT_MAX),          # plain hit
        ([278, 273, -800], 0, [0, 0, -1], FLT_MAX),         # misses the tree box
        ([278, 273, -800], 0, [0, 0, 1], 100.0),            # range ends before the box
        ([278, 273, -800], 2000.0, [0, 0, 1], FLT_MAX),     # range starts behind everything
        ([278, 273, 100], 0, [0, 0, 1], FLT_MAX),           # origin inside the box (negative entry)
        ([0, 273, -800], 0, [0, 0, 1], FLT_MAX),            # origin on a slab with d = 0 there: 0*inf = NaN
        ([278, 0, 100], 0, [1, 0, 0], FLT_MAX),             # grazing along the floor plane
        ([278, 273, 100], 0, [0, 1, 0], FLT_MAX),           # axis aligned
        ([278, 273, 100], 0, [0, -1, 0], FLT_MAX),
        ([278, 273, 100], 0, [-0.0, -1, -0.0], FLT_MAX),    # negative zeros: inv_d = -inf
        ([278, 273, 100], 0, [np.nan, 0, 1], FLT_MAX),      # NaN direction
        ([278, 273, 100], 0, [0, 0, 0], FLT_MAX),           # zero direction
        ([278, 273, 100], 0, [0, 0, 1], inf),               # infinite range
        ([278, 273, 100], inf, [0, 0, 1], inf),             # t_min = inf: distance < t_min break
        ([130, 165, 65], 0, [0.6, 0.0, 0.8], FLT_MAX),      # starts on a vertex of the short block
        ([552.8, 0, 0], 0, [-0.70710677, 0.70710677, 0], FLT_MAX),  # from a corner along a wall
    ]
    ro = np.array([r[0] + [r[1]] for r in rays], np.float32)
    rd = np.array([r[2] + [r[3]] for r in rays], np.float32)
    want = oracle_cornell.trace_closest(ro, rd)
    got = gpu_cornell.trace(ro, rd)
    assert_hits_equal(got, want)
    assert np.array_equal(gpu_cornell.trace(ro, rd, is_vis_ray=True), oracle_cornell.trace_any(ro, rd))
    assert want["tri_id"][0] != rt.GIGL_MISS and want["tri_id"][1] == rt.GIGL_MISS


def test_shared_edge_and_vertex_ties(gpu_cornell, oracle_cornell, cornell):
    """Rays aimed exactly at shared edges / vertices of quads: the reference's tie behaviour
    (first-tested triangle wins exact distance ties, KdTree.hpp:323-332 with the strict `<` of
    RTBase.cpp:167) must come out the same."""
    V, T = cornell["verts"], cornell["tris"]
    cam = np.array(S.CAM_POS, np.float32)
    targets = []
    for q in range(len(T) // 2):
        a, b = T[2 * q], T[2 * q + 1]
        c0, c1, c2, c3 = V[a[0]], V[a[1]], V[a[2]], V[b[2]]
        targets += [c0, c1, c2, c3, (c0 + c2) * np.float32(0.5), (c0 + c1) * np.float32(0.5),
                    c0 + (c2 - c0) * np.float32(0.25)]
    targets = np.array(targets, np.float32)
    d = targets - cam
    d = d * (np.float32(1.0) / np.sqrt((d * d).sum(1, dtype=np.float32)))[:, None]
    ro = np.concatenate([np.tile(cam, (len(d), 1)), np.zeros((len(d), 1), np.float32)], 1)
    rd = np.concatenate([d, np.full((len(d), 1), FLT_MAX, np.float32)], 1)
    assert_hits_equal(gpu_cornell.trace(ro, rd), oracle_cornell.trace_closest(ro, rd))


def _cross_f32(e0, e1):
    """cross(e0, e1) in GLM's operation order (func_geometric.inl:224-227), float32."""
    f = np.float32
    return np.array([f(e0[1] * e1[2]) - f(e1[1] * e0[2]), f(e0[2] * e1[0]) - f(e1[2] * e0[0]),
                     f(e0[0] * e1[1]) - f(e1[0] * e0[1])], np.float32)


def test_trace_one(gpu_cornell, oracle_cornell, cornell, cornell_cam):
    """gigl_trace_one against Scene::trace on one ray (oracle): the bool, the record, the normal as
    Scene::trace leaves it (normalized for closest-hit rays, cross(e0,e1) for visibility rays) and the
    (t_min, t_max) KdTree::intersect leaves in the caller's ray, bit for bit."""
    V, T = cornell["verts"], cornell["tris"]
    po, pd = O.primary_rays(cornell_cam, 1024, 0, 1024 * 1024)
    rays = [(po[i, :3], pd[i, :3], 0.0, FLT_MAX) for i in range(0, 1024 * 1024, 9973)]
    rays += [([278, 273, -800], [0, 0, 1], 0.0, FLT_MAX), ([278, 273, -800], [0, 0, -1], 0.0, FLT_MAX),
             ([278, 273, -800], [0, 0, 1], 0.0, 100.0), ([278, 273, -800], [0, 0, 1], 2000.0, FLT_MAX),
             ([278, 273, 100], [0, 1, 0], 0.0, FLT_MAX), ([278, 273, 100], [0.6, 0.0, 0.8], 0.0, 50.0),
             ([0, 273, -800], [0, 0, 1], 0.0, FLT_MAX), ([278, 273, 100], [np.nan, 0, 1], 0.0, FLT_MAX)]
    n_hit = 0
    for o, d, t0, t1 in rays:
        hit, rec, nrm = gpu_cornell.trace_one(o, d, t0, t1)
        rng = gpu_cornell.last_range.copy()
        w_hit, w_rec, w_nrm, w_rng = oracle_cornell.trace_one(o, d, t0, t1)
        assert hit == w_hit and rec["tri_id"] == w_rec["tri_id"]
        for k in ("t", "u", "v"):
            assert rec[k].view(np.uint32) == w_rec[k].view(np.uint32), k
        assert np.array_equal(rng.view(np.uint32), w_rng.view(np.uint32)), "range_out"
        if hit:
            n_hit += 1
            assert np.array_equal(nrm.view(np.uint32), w_nrm.view(np.uint32)), "normal_out"
            assert abs(float(np.linalg.norm(nrm)) - 1.0) < 1e-6
        # visibility ray: the bool is the reference's; the record is the closest hit's (documented in
        # include/gigl_rt.h), so the normal is checked as cross(e0, e1) of the triangle it names
        v_hit, v_rec, v_nrm = gpu_cornell.trace_one(o, d, t0, t1, is_vis_ray=True)
        w_vhit, w_vrec, w_vnrm, _ = oracle_cornell.trace_one(o, d, t0, t1, is_vis_ray=True)
        assert v_hit == w_vhit == hit
        if v_hit:
            t = T[v_rec["tri_id"]]
            want_n = _cross_f32(V[t[1]] - V[t[0]], V[t[2]] - V[t[0]])
            assert np.array_equal(v_nrm.view(np.uint32), want_n.view(np.uint32))
            if v_rec["tri_id"] == w_vrec["tri_id"]:
                assert np.array_equal(v_nrm.view(np.uint32), w_vnrm.view(np.uint32))
    assert n_hit > 50


def test_concurrent_trace_calls_on_one_scene(s16):
    """include/gigl_rt.h: trace calls may be issued concurrently on one handle (KdTree::intersect is
    const and re-entrant, DensityField.cpp:280-293).  Two host threads trace different batches through
    the same scene at the same time; each gets what it gets alone."""
    import threading
    batches = [synth.bounce_rays(s16["V"], s16["T"], 60_001 + 1000 * k, 0x7EAD00 + k) for k in range(2)]
    alone = [s16["gpu"].trace(ro, rd) for ro, rd in batches]
    alone_any = [s16["gpu"].trace(ro, rd, is_vis_ray=True) for ro, rd in batches]
    errors = []

    def worker(k):
        try:
            ro, rd = batches[k]
            for it in range(12):
                if it % 3 == 2:
                    assert np.array_equal(s16["gpu"].trace(ro, rd, is_vis_ray=True), alone_any[k])
                else:
                    assert np.array_equal(s16["gpu"].trace(ro, rd).view(np.uint32), alone[k].view(np.uint32))
                hit, rec, _ = s16["gpu"].trace_one(ro[it, :3], rd[it, :3])
                assert rec["tri_id"] == alone[k]["tri_id"][it]
        except Exception as e:   # noqa: BLE001 - reported below, in the main thread
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_sample_density_and_transmittance(density, fog_box):
    torch = pytest.importorskip("torch")
    fog_o = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    fog_g = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    rng = np.random.default_rng(7)
    lo, hi = fog_box[:3] - 30, fog_box[3:] + 30
    pos = (lo + (hi - lo) * rng.random((20000, 3))).astype(np.float32)
    pos[:8] = [fog_box[:3], fog_box[3:], (fog_box[:3] + fog_box[3:]) / 2, fog_box[:3] - 1e-3,
               fog_box[3:] + 1e-3, [0, 0, -1], [559, 548.8, 559.2], [279.5, 274.4, 279.1]]
    got = fog_g.sample_density(pos)
    want = np.array([O.sample_density(fog_o, p) for p in pos], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # segments: random points to random "lights", some outside the fog box, some degenerate
    n = 50000
    p = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    L = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    v = L - p
    dsq = (v * v).sum(1, dtype=np.float32)
    w = v * (np.float32(1.0) / np.sqrt(dsq))[:, None]
    w[:16, 0] = 0.0   # axis-parallel components -> inf reciprocals in intersectBBox
    seg_p = np.concatenate([p, np.zeros((n, 1), np.float32)], 1)
    seg_d = np.concatenate([w, dsq[:, None]], 1).astype(np.float32)
    tp = torch.from_numpy(seg_p).cuda()
    td = torch.from_numpy(seg_d).cuda()
    out = torch.zeros(n, dtype=torch.float32, device="cuda:0")
    fog_g.transmittance_device(tp.data_ptr(), td.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = O.transmittance(fog_o, seg_p, seg_d, n_threads=4)
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-30)
    assert float(rel.max()) <= REL_TOL
    assert (want < 0.999).mean() > 0.3   # the batch does exercise the march
    fog_g.close()


def test_shadow_batch_cornell(gpu_cornell, oracle_cornell, cornell_cam, density, fog_box):
    """Config 2 second half + config 5 on a subsample: primary hit points x lights."""
    torch = pytest.importorskip("torch")
    # every 37th pixel of rows 300..700 (hits and misses both present), 150 light positions
    pix = np.arange(300 * 1024, 700 * 1024, 37, dtype=np.uint64)
    ro = np.concatenate([O.primary_rays(cornell_cam, 1024, int(p), 1)[0] for p in pix[:4000]])
    rd = np.concatenate([O.primary_rays(cornell_cam, 1024, int(p), 1)[1] for p in pix[:4000]])
    # a few guaranteed misses
    rd[:5, :3] = [0, 0, -1]
    hits = oracle_cornell.trace_closest(ro, rd)
    rng = np.random.default_rng(11)
    lights = np.stack([rng.uniform(5, 550, 150), rng.uniform(5, 545, 150), rng.uniform(5, 555, 150)], 1).astype(np.float32)
    lights[0] = S.PRIM_PL_POS
    fog_o = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    fog_g = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    want_occ, want_tr = oracle_cornell.shadow_batch(ro, rd, hits, lights, fog=fog_o, want_transm=True, n_threads=4)
    n_pts, nl = len(ro), len(lights)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_o, d_d, d_l = t(ro), t(rd), t(lights)
    d_h = torch.from_numpy(hits.view(np.int32).reshape(-1, 4).copy()).cuda()
    bits = torch.zeros((n_pts * nl + 63) // 64, dtype=torch.int64, device="cuda:0")
    tr = torch.zeros(n_pts * nl, dtype=torch.float32, device="cuda:0")
    gpu_cornell.shadow_batch_device(d_o.data_ptr(), d_d.data_ptr(), d_h.data_ptr(), n_pts, d_l.data_ptr(), nl,
                                    bits.data_ptr(), fog=fog_g, transm_ptr=tr.data_ptr(),
                                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n_pts * nl).reshape(n_pts, nl)
    missed = hits["tri_id"] == rt.GIGL_MISS
    assert missed.sum() >= 5
    want01 = np.where(want_occ == 0xFF, 0, want_occ)
    assert np.array_equal(got, want01)
    assert 0.05 < want01[~missed].mean() < 0.95
    got_tr = tr.cpu().numpy().reshape(n_pts, nl)
    rel = np.abs(got_tr.astype(np.float64) - want_tr) / np.maximum(np.abs(want_tr), 1e-30)
    assert float(rel.max()) <= REL_TOL
    # the order the pairs are traced in (light by light, the default, or pair order) and the any-hit output
    # mode are scheduling choices: the bits land at the pair's own position either way
    for order, any_atomic in ((0, -1), (1, 0), (0, 0), (1, 1)):
        try:
            rt.set_option("shadow_order", order)
            rt.set_option("any_atomic", any_atomic)
            bits.fill_(-1)
            gpu_cornell.shadow_batch_device(d_o.data_ptr(), d_d.data_ptr(), d_h.data_ptr(), n_pts, d_l.data_ptr(), nl,
                                            bits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        finally:
            rt.set_option("shadow_order", 1)
            rt.set_option("any_atomic", -1)
        got2 = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n_pts * nl).reshape(n_pts, nl)
        assert np.array_equal(got2, want01), (order, any_atomic)
    fog_g.close()


def test_shadow_batch_fused_cornell(gpu_cornell, oracle_cornell, cornell_cam, density, fog_box):
    """The shader's order (Surface.frag:250-256): visibility first, calcTransm only for the visible
    pairs, in one pass.  Bits bit-exact; transmittance 0 where occluded, <= 1e-5 relative where
    visible, 1.0 for missed points; and equal to the two-pass batch wherever the pair is visible."""
    torch = pytest.importorskip("torch")
    pix = np.arange(250 * 1024, 760 * 1024, 29, dtype=np.uint64)[:6000]
    ro = np.concatenate([O.primary_rays(cornell_cam, 1024, int(p), 1)[0] for p in pix])
    rd = np.concatenate([O.primary_rays(cornell_cam, 1024, int(p), 1)[1] for p in pix])
    rd[:7, :3] = [0, 0, -1]   # guaranteed misses
    hits = oracle_cornell.trace_closest(ro, rd)
    vp = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "vpl_set.npz"))["vpls"]
    lights = np.concatenate([[S.PRIM_PL_POS], vp.view(np.float32).reshape(len(vp), -1)[:, :3]]).astype(np.float32)
    fog_o = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    fog_g = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    w_occ, w_tr, sums = oracle_cornell.shadow_batch_fused(ro, rd, hits, lights, fog=fog_o, n_threads=4)
    n_pts, nl = len(ro), len(lights)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_o, d_d, d_l = t(ro), t(rd), t(lights)
    d_h = torch.from_numpy(hits.view(np.int32).reshape(-1, 4).copy()).cuda()
    sh = torch.cuda.current_stream().cuda_stream
    for any_atomic in (-1, 0):
        try:
            rt.set_option("any_atomic", any_atomic)
            bits = torch.zeros((n_pts * nl + 63) // 64, dtype=torch.int64, device="cuda:0")
            tr = torch.full((n_pts * nl,), -5.0, dtype=torch.float32, device="cuda:0")
            gpu_cornell.shadow_batch_fused_device(d_o.data_ptr(), d_d.data_ptr(), d_h.data_ptr(), n_pts, d_l.data_ptr(),
                                                  nl, bits.data_ptr(), fog_g, tr.data_ptr(), stream=sh)
            torch.cuda.synchronize()
        finally:
            rt.set_option("any_atomic", -1)
        got = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n_pts * nl).reshape(n_pts, nl)
        want01 = np.where(w_occ == 0xFF, 0, w_occ)
        assert np.array_equal(got, want01)
        got_tr = tr.cpu().numpy().reshape(n_pts, nl)
        assert (got_tr[want01 == 1] == 0.0).all()                       # nothing arrives from an occluded light
        assert (got_tr[hits["tri_id"] == rt.GIGL_MISS] == 1.0).all()    # missed points
        rel = np.abs(got_tr.astype(np.float64) - w_tr) / np.maximum(np.abs(w_tr), 1e-30)
        assert float(rel[want01 == 0].max()) <= REL_TOL
    assert int(sums[0]) == int((hits["tri_id"] != rt.GIGL_MISS).sum()) * nl and 0 < int(sums[3]) < int(sums[0])
    # against the two-pass batch: the same numbers wherever the pair is visible
    tr2 = torch.zeros(n_pts * nl, dtype=torch.float32, device="cuda:0")
    gpu_cornell.shadow_batch_device(d_o.data_ptr(), d_d.data_ptr(), d_h.data_ptr(), n_pts, d_l.data_ptr(), nl, None,
                                    fog=fog_g, transm_ptr=tr2.data_ptr(), stream=sh)
    torch.cuda.synchronize()
    two = tr2.cpu().numpy().reshape(n_pts, nl)
    assert np.array_equal(two[want01 == 0].view(np.uint32), got_tr[want01 == 0].view(np.uint32))
    fog_g.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_multi_device_entry_points(s16, devices):
    """gigl_multi_*: the scene replicated over a device list, contiguous ray ranges, results gathered
    to devices[0] (peer copies) or to the host.  A one-device list is the plain call; a list that
    repeats device 0 runs the replication, scatter and gather paths on a one-GPU box.  Records and
    occlusion bits equal the single-device calls' on the whole batch, bit for bit.  The batch is ordered
    behind the caller's stream: the output is filled, and the rays are produced, on that stream right
    before the call, without a synchronisation in between."""
    torch = pytest.importorskip("torch")
    n = 200_003   # ragged over 3 ranges, and not a multiple of 64
    ro, rd = synth.bounce_rays(s16["V"], s16["T"], n, 0x3D1CE5)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    sh = torch.cuda.current_stream().cuda_stream
    want = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    s16["gpu"].trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, want.data_ptr(), sh)
    want_bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda:0")
    s16["gpu"].trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n, want_bits.data_ptr(), sh)
    torch.cuda.synchronize()
    m = rt.MultiDeviceScene(s16["gpu"], devices)
    assert m.device_count() == len(devices)
    side = torch.cuda.Stream()
    for n_call in (n, 1000, n, 64, 65):   # smaller batches in between: the staging buffers are reused
        for stream in (torch.cuda.current_stream(), side):
            with torch.cuda.stream(stream):
                # rays produced and outputs filled on the caller's stream, then the call, no synchronisation
                c_o, c_d = d_o[:n_call].clone(), d_d[:n_call].clone()
                got = torch.full((n_call, 4), -3, dtype=torch.int32, device="cuda:0")
                m.trace_closest_device(c_o.data_ptr(), c_d.data_ptr(), n_call, got.data_ptr(), stream.cuda_stream)
                assert torch.equal(got, want[:n_call])
                bits = torch.full(((n_call + 63) // 64,), -1, dtype=torch.int64, device="cuda:0")
                m.trace_any_device(c_o.data_ptr(), c_d.data_ptr(), n_call, bits.data_ptr(), stream.cuda_stream)
                g = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n_call)
                assert np.array_equal(g, rt.unpack_bits(want_bits.cpu().numpy().view(np.uint64), n)[:n_call])
    # the asynchronous forms: two batches on two caller streams through the handle's two slots, results read
    # after the streams (not the device) have been waited for
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i in range(6):
        st = streams[i % 2]
        with torch.cuda.stream(st):
            n_call = n - 7 * i
            got = torch.full((n_call, 4), -3, dtype=torch.int32, device="cuda:0")
            bits = torch.zeros((n_call + 63) // 64, dtype=torch.int64, device="cuda:0")
            m.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_call, got.data_ptr(), st.cuda_stream, wait=False)
            m.trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n_call, bits.data_ptr(), st.cuda_stream, wait=False)
            outs.append((n_call, got, bits))
    for st in streams:
        st.synchronize()
    wb = rt.unpack_bits(want_bits.cpu().numpy().view(np.uint64), n)
    for n_call, got, bits in outs:
        assert torch.equal(got, want[:n_call])
        assert np.array_equal(rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n_call), wb[:n_call])
    host = m.trace(ro, rd)     # host pointers: every device fetches and returns its own range (worker threads)
    assert np.array_equal(host.view(np.uint32).reshape(-1, 4), want.cpu().numpy().view(np.uint32))
    assert np.array_equal(m.trace_any(ro, rd), wb)
    assert np.array_equal(m.trace_any(ro[:70], rd[:70]), wb[:70])
    assert len(m.trace(ro[:0], rd[:0])) == 0
    m.close()
    with pytest.raises(rt.GiglError):
        rt.MultiDeviceScene(s16["gpu"], [1_000])   # devices[0] must be the scene's device


def test_scene_adopted_from_a_device_blob(s16):
    """gigl_scene_export_device_blob / gigl_scene_create_from_device_blob: a second scene made from the header
    and copies of the three device arrays of the first (what a rank does with a broadcast tree) traces bit-equal
    to it, closest-hit and any-hit, survives the source arrays being freed, and has no host tree."""
    torch = pytest.importorskip("torch")
    from gigl_amd import shard
    src = s16["gpu"]
    desc, ptrs = src.export_device_blob()
    assert desc.magic == 0x42474947 and desc.nb_nodes > 0 and desc.nb_tris >= len(s16["T"]) * 64 and ptrs[0] and ptrs[2]
    copies = [None if not (p and nb) else torch.as_tensor(shard._DeviceBytes(p, nb), device="cuda:0").clone()
              for p, nb in zip(ptrs, (desc.nb_nodes, desc.nb_leaf_ids, desc.nb_tris))]
    torch.cuda.synchronize()
    ptr = lambda t: 0 if t is None else t.data_ptr()
    rep = rt.Scene.from_device_blob(desc, ptr(copies[0]), ptr(copies[1]), ptr(copies[2]), device=0)
    for c in copies:
        if c is not None:
            c.fill_(0xAB)   # the library copied them
    del copies
    torch.cuda.synchronize()
    n = 100_000
    ro, rd = synth.bounce_rays(s16["V"], s16["T"], n, 0xB10B)
    assert np.array_equal(rep.trace(ro, rd).view(np.uint32), src.trace(ro, rd).view(np.uint32))
    assert np.array_equal(rep.trace(ro, rd, is_vis_ray=True), src.trace(ro, rd, is_vis_ray=True))
    assert rep.layout() == src.layout() and np.array_equal(rep.bbox(), src.bbox())
    st, ss = rep.stats(), src.stats()
    assert (st.n_nodes, st.actual_depth, st.device_bytes) == (ss.n_nodes, ss.actual_depth, ss.device_bytes)
    hit, rec, nrm = rep.trace_one(ro[5, :3], rd[5, :3])
    hit2, rec2, nrm2 = src.trace_one(ro[5, :3], rd[5, :3])
    assert hit == hit2 and rec == rec2 and np.array_equal(nrm, nrm2)
    with pytest.raises(rt.GiglError):
        rep.export_tree()   # no host tree
    bad = rt.SceneBlob.from_buffer_copy(bytes(desc))
    bad.magic = 1
    with pytest.raises(rt.GiglError):
        rt.Scene.from_device_blob(bad, ptrs[0], ptrs[1], ptrs[2], device=0)
    rep.close()


@pytest.mark.parametrize("res", [(96, 64), (1024, 1024)])
def test_compute_pi_density(gpu_cornell, oracle_cornell, density, fog_box, res):
    """DensityField::computePiDensity as one GPU batch (primary closest-hit, then
    a 256-interval march per pixel), against the oracle.  (1024, 1024) is the reference's own
    configuration: the 268 MB payload of pi_df.3dt.  PARITY UNPINNED against the reference file
    itself, which was never digested."""
    torch = pytest.importorskip("torch")
    rx, ry = res
    cam = rt.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, rx, ry)
    fog_o = O.Fog(density, fog_box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    fog_g = rt.FogVolume(density, fog_box, S.ABS_K, S.SCA_K)
    out = torch.full((64, ry, rx), -1.0, dtype=torch.float32, device="cuda:0")
    gpu_cornell.compute_pi_density_device(fog_g, cam, rx, ry, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = O.compute_pi_density(oracle_cornell, fog_o, cam, rx, ry, n_threads=8)
    assert (want > 0).mean() > 0.5
    rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
    assert float(rel.max()) <= REL_TOL
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "pi density bits differ"
    # size-independent property: the running integral never decreases along a ray
    assert (np.diff(got, axis=0) >= 0).all()
    fog_g.close()


def test_full_size_scene_s158(cornell):
    """configs[2] at its full scene size (998,560 triangles, tree depth 31, 8.4 M nodes): tree equal
    to the oracle's, 2^21 incoherent rays bit-exact against the oracle, and on the full 2^24-ray batch
    the size-independent properties: any-hit bits == (closest-hit id != miss), every hit inside its
    ray's range, and a regrouped (permuted) launch returning the identical buffer."""
    torch = pytest.importorskip("torch")
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 158)
    assert len(T) == 998560
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    o = O.OracleScene(V, T, **S.KD_PARAMS)
    nodes, leaf = g.export_tree()
    assert np.array_equal(nodes, o.nodes()) and np.array_equal(leaf, o.leaf_prim_ids())
    n = 1 << 24
    ro, rd = synth.bounce_rays(V, T, n, synth.RAY_SEED_16M)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    hits = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, hits.data_ptr(), st)
    g.trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n, bits.data_ptr(), st)
    torch.cuda.synchronize()
    got = hits.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
    m = 1 << 21
    assert_hits_equal(got[:m], o.trace_closest(ro[:m], rd[:m], n_threads=8))
    hit = got["tri_id"] != rt.GIGL_MISS
    occ = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n)
    assert np.array_equal(occ.astype(bool), hit)
    assert (got["tri_id"][hit] < len(T)).all() and (got["t"][hit] > 0).all() and np.isfinite(got["t"][hit]).all()
    assert (got["t"][~hit] == FLT_MAX).all()
    try:
        rt.set_option("sort_rays", 1)
        hits2 = torch.empty_like(hits)
        g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, hits2.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(hits, hits2)
    finally:
        rt.set_option("sort_rays", 0)
    g.close()


def test_full_size_scene_s500(cornell):
    """configs[3]'s scene at full size: S(500) = 10,000,000 triangles, tree depth 31.  The builder's tree
    does not depend on its thread count (1-thread build on a host-only scene == the device scene's tree);
    the oracle TRAVERSES THE ENGINE'S EXPORTED TREE ("the build's own tree as ground truth
    for large scenes" - the serial oracle build of 10 M triangles takes minutes) on the first 2^20 rays
    of the RAY_SEED_128M batch: ids, t/u/v and any-hit bits bit-exact.  On a 2^24-ray slice of the same
    batch, the size-independent properties: any-hit bits == (closest-hit id != miss), every hit inside
    its ray's range."""
    torch = pytest.importorskip("torch")
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 500)
    assert len(T) == 10_000_000
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    st = g.stats()
    assert st.actual_depth == 31 and st.degenerate_split_guard_hit == 0
    nodes, leaf = g.export_tree()
    # thread-count independence: the same tree from a build held to 5 threads (default: min(16, cores))
    rt.set_option("build_threads", 5)
    try:
        h = rt.Scene(V, T, device=rt.GIGL_DEVICE_NONE, **S.KD_PARAMS)
    finally:
        rt.set_option("build_threads", 0)
    n2, l2 = h.export_tree()
    assert np.array_equal(nodes, n2) and np.array_equal(leaf, l2)
    h.close()
    del n2, l2
    o = O.OracleScene(V, T, tree=(nodes, leaf))
    assert np.array_equal(o.bbox(), g.bbox())
    n = 1 << 24
    ro, rd = synth.bounce_rays(V, T, n, synth.RAY_SEED_128M)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    hits = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda:0")
    sh = torch.cuda.current_stream().cuda_stream
    g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, hits.data_ptr(), sh)
    g.trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n, bits.data_ptr(), sh)
    torch.cuda.synchronize()
    got = hits.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
    occ = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n)
    m = 1 << 20
    assert_hits_equal(got[:m], o.trace_closest(ro[:m], rd[:m], n_threads=16))
    assert np.array_equal(occ[:m], o.trace_any(ro[:m], rd[:m], n_threads=16))
    hit = got["tri_id"] != rt.GIGL_MISS
    assert np.array_equal(occ.astype(bool), hit)
    assert 0.5 < hit.mean() <= 1.0
    assert (got["tri_id"][hit] < len(T)).all() and (got["t"][hit] > 0).all() and np.isfinite(got["t"][hit]).all()
    assert (got["t"][~hit] == FLT_MAX).all()
    g.close()
