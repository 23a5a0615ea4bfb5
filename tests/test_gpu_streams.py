"""Launches that overlap on one GPU: several streams tied together by events, and the host-pointer calls,
which pipeline a large batch over copy and traversal streams of their own (gigl_rt.hip, trace_host).

Both once produced wrong records: the device's default stream-ordered pool gave a launch on stream B the
workspace whose hipFreeAsync was still pending on stream A (B had waited on an earlier event of A), so two
traversals in flight shared one work counter.  The engine allocates from a pool of its own since.
"""
import ctypes as C

import numpy as np
import pytest

from gigl_amd import rt, scene_io as S, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s64(cornell):
    import torch
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 64)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    n = 3 * (1 << 21) + 12345             # four chunks of the host-pointer pipeline, the last one ragged
    ro, rd = synth.bounce_rays(V, T, n, 0x5EED57, n_threads=8)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    ref = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = ref.cpu().numpy()
    # the single-stream records themselves against the oracle (on the engine's exported tree), first 2^15 rays
    ora = O.OracleScene(V, T, tree=g.export_tree())
    m = 1 << 15
    w = ora.trace_closest(ro[:m], rd[:m], n_threads=4)
    got = want[:m].view(rt.HIT_DTYPE).reshape(-1)
    assert np.array_equal(got["tri_id"], w["tri_id"])
    hit = got["tri_id"] != rt.GIGL_MISS
    for k in ("t", "u", "v"):
        assert np.array_equal(got[k][hit].view(np.uint32), w[k][hit].view(np.uint32)), k
    yield {"gpu": g, "n": n, "ro": ro, "rd": rd, "d_o": d_o, "d_d": d_d, "want": want}
    g.close()


def test_launches_on_event_linked_streams(s64):
    """B waits on an event A recorded BEFORE A's launch, then both traverse at the same time."""
    import torch
    g, n, want = s64["gpu"], s64["n"], torch.from_numpy(s64["want"]).cuda()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    out_a = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    out_b = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for it in range(12):
        out_a.zero_(); out_b.zero_()
        torch.cuda.synchronize()
        ev = torch.cuda.Event()
        ev.record(a)
        g.trace_closest_device(s64["d_o"].data_ptr(), s64["d_d"].data_ptr(), n, out_a.data_ptr(), a.cuda_stream)
        b.wait_event(ev)
        g.trace_closest_device(s64["d_o"].data_ptr(), s64["d_d"].data_ptr(), n, out_b.data_ptr(), b.cuda_stream)
        ev2 = torch.cuda.Event()
        ev2.record(b)
        a.wait_event(ev2)                  # and the other way round for the next round
        torch.cuda.synchronize()
        assert bool((out_a == want).all()) and bool((out_b == want).all()), "round %d" % it


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pointer_batches_are_pipelined_correctly(s64, pinned):
    """gigl_trace_closest / gigl_trace_any on more rays than one chunk, pageable and page-locked buffers."""
    import torch
    g, n, want = s64["gpu"], s64["n"], s64["want"]
    L = rt.lib()
    keep = []
    if pinned:
        t_o, t_d = torch.from_numpy(s64["ro"]).pin_memory(), torch.from_numpy(s64["rd"]).pin_memory()
        t_h = torch.zeros((n, 4), dtype=torch.int32).pin_memory()
        t_b = torch.zeros((n + 63) // 64, dtype=torch.int64).pin_memory()
        keep = [t_o, t_d, t_h, t_b]
        o_ptr, d_ptr, h_ptr, b_ptr = t_o.data_ptr(), t_d.data_ptr(), t_h.data_ptr(), t_b.data_ptr()
        hits, bits = t_h.numpy(), t_b.numpy().view(np.uint64)
    else:
        hits, bits = np.zeros((n, 4), np.int32), np.zeros((n + 63) // 64, np.uint64)
        o_ptr, d_ptr, h_ptr, b_ptr = s64["ro"].ctypes.data, s64["rd"].ctypes.data, hits.ctypes.data, bits.ctypes.data
    for it in range(4):
        hits[:] = 0
        bits[:] = 0
        assert L.gigl_trace_closest(g.h, C.c_void_p(o_ptr), C.c_void_p(d_ptr), n, C.c_void_p(h_ptr)) == 0, L.gigl_last_error()
        assert np.array_equal(hits, want), "closest, call %d: %d records differ" % (it, int((hits != want).any(1).sum()))
        assert L.gigl_trace_any(g.h, C.c_void_p(o_ptr), C.c_void_p(d_ptr), n, C.c_void_p(b_ptr)) == 0, L.gigl_last_error()
        assert np.array_equal(rt.unpack_bits(bits, n), (want[:, 0] != -1).astype(np.uint8)), "any, call %d" % it
    # a batch of exactly one chunk and one just above it
    for m in (1 << 21, (1 << 21) + 1):
        assert L.gigl_trace_closest(g.h, C.c_void_p(o_ptr), C.c_void_p(d_ptr), m, C.c_void_p(h_ptr)) == 0
        assert np.array_equal(hits[:m], want[:m])
    del keep


def test_host_pointer_batches_from_two_host_threads(s64):
    """Two host threads inside pipelined host-pointer calls on one scene at the same time (ctypes drops the
    GIL): each call has streams and buffer slots of its own, the pool and the scene are shared."""
    import threading
    g, n, want = s64["gpu"], s64["n"], s64["want"].view(rt.HIT_DTYPE).reshape(-1)
    half = (n // 2) & ~63
    parts = [(0, half + 777), (half, n)]          # overlapping ranges, both longer than one chunk
    out, err = [None, None], []

    def work(k):
        try:
            lo, hi = parts[k]
            for _ in range(3):
                out[k] = g.trace(s64["ro"][lo:hi], s64["rd"][lo:hi])
                assert np.array_equal(out[k].view(np.uint32), want[lo:hi].view(np.uint32)), "thread %d" % k
                occ = g.trace(s64["ro"][lo:hi], s64["rd"][lo:hi], is_vis_ray=True)
                assert np.array_equal(occ, (want["tri_id"][lo:hi] != rt.GIGL_MISS).astype(np.uint8)), "thread %d any" % k
        except BaseException as e:   # noqa: BLE001 - reported in the main thread
            err.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err


def test_multi_device_host_pointer_ranges_are_pipelined(s64):
    """gigl_multi_trace_closest (host pointers) over a list that repeats device 0: three replicas, each inside
    its own pipelined host-pointer call at the same time, every range longer than one chunk."""
    g, want = s64["gpu"], s64["want"]
    m = rt.MultiDeviceScene(g, [0, 0, 0])
    for _ in range(2):
        got = m.trace(s64["ro"], s64["rd"])
        assert np.array_equal(got.view(np.uint32).reshape(-1, 4), want.view(np.uint32))
    m.close()


def test_pool_is_trimmed_on_request_and_with_the_last_scene(cornell):
    """The library's stream-ordered pool keeps its largest blocks between calls; gigl_trim_pool,
    and the destruction of the last scene on a device, give the idle memory back."""
    torch = pytest.importorskip("torch")
    torch.cuda.synchronize()
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 8)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    n = 1 << 20
    ro, rd = synth.bounce_rays(V, T, n, 0x7121)
    want = g.trace(ro, rd)            # host pointers: 48 MB of staging from the pool
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info(0)[0]
    rt.trim_pool(0)
    after = torch.cuda.mem_get_info(0)[0]
    assert after >= held + (40 << 20), (held, after)    # the staging went back
    assert np.array_equal(g.trace(ro, rd).view(np.uint32), want.view(np.uint32))   # and the pool still serves
    with pytest.raises(rt.GiglError):
        rt.trim_pool(1000)
    g.close()


def test_instrumented_build_counts_and_timeline(s64):
    """Option `stats`: the instrumented kernel returns the same records and its counters add up — every ray is
    handed to a lane exactly once; the timeline (every 16th wave) sees about a sixteenth of them, in order of time."""
    import torch
    g, n = s64["gpu"], s64["n"]
    out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    rt.set_option("stats", 1)
    try:
        g.trace_closest_device(s64["d_o"].data_ptr(), s64["d_d"].data_ptr(), n, out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = rt.debug_last_stats()
        tl = rt.debug_last_timeline()
    finally:
        rt.set_option("stats", 0)
    assert np.array_equal(out.cpu().numpy(), s64["want"])
    assert st["refill_lanes"] >= n and st["rounds"] > 0 and st["waves"] > 0
    assert st["t_lane_steps"] <= 64 * st["t_iters"] and st["l_lane_rounds"] <= 64 * st["l_rounds"]
    assert tl.shape == (3, 1024)
    started, rounds, lanes = (int(tl[k].sum()) for k in range(3))
    assert n // 32 < started < n // 8, (started, n)            # a sixteenth of the waves, give or take
    assert rounds > 0 and lanes <= 64 * rounds
    used = np.nonzero(tl[1])[0]
    assert used[0] <= 1 and used[-1] < 1023                    # starts with the first wave (give or take a bucket's edge), ends long before the wrap


def test_lean_walk_equals_the_shipped_kernel(s64):
    """The one-ray-per-lane walk of small VPL sets (gigl_walk_lean.h) over the module's 6.3 M rays: every record
    equal to the shipped kernel's (which the fixture pinned to the oracle)."""
    import torch
    g, n = s64["gpu"], s64["n"]
    out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    g.debug_trace_closest_lean_device(s64["d_o"].data_ptr(), s64["d_d"].data_ptr(), n, out.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), s64["want"])


def test_lean_walk_entry_point_rejects_what_it_cannot_walk(cornell):
    """gigl_debug_trace_closest_lean_device reads triple records only; an empty batch is a no-op; NULL pointers and a
    host-only scene are errors, not faults."""
    import torch
    rt.set_option("layout", 1)
    try:
        pairs = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    finally:
        rt.set_option("layout", 2)
    o = torch.zeros((64, 4), device="cuda"); d = torch.ones((64, 4), device="cuda")
    h = torch.zeros((64, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(rt.GiglError):
        pairs.debug_trace_closest_lean_device(o.data_ptr(), d.data_ptr(), 64, h.data_ptr(), 0)
    pairs.close()
    g = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    g.debug_trace_closest_lean_device(0, 0, 0, 0, 0)                      # n = 0: nothing to do
    with pytest.raises(rt.GiglError):
        g.debug_trace_closest_lean_device(0, d.data_ptr(), 64, h.data_ptr(), 0)
    g.close()
    host_only = rt.Scene(cornell["verts"], cornell["tris"], device=-1, **S.KD_PARAMS)
    with pytest.raises(rt.GiglError):
        host_only.debug_trace_closest_lean_device(o.data_ptr(), d.data_ptr(), 64, h.data_ptr(), 0)
    host_only.close()
