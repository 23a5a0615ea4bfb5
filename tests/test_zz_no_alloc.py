"""The no-allocation entry point (gigl_trace_closest_noalloc_device) against gigl_trace_closest_device.

Collected last on purpose (test_zz_*): the round's end runs `pytest -x`.  The entry point has no
counterpart in the reference; its bar is the existing entry point's output, bit for bit, on two calls
in a row, the second with a smaller input (the static scratch keeps the first call's contents).
"""
import numpy as np
import pytest

from gigl_amd import rt, scene_io as S, synth

pytestmark = pytest.mark.gpu


def _trace_both(torch, scene, ro, rd):
    n = len(ro)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    a = torch.full((n, 4), -7, dtype=torch.int32, device="cuda:0")
    b = torch.full((n, 4), -9, dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, a.data_ptr(), st)
    scene.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, b.data_ptr(), st)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_matches_allocating_entry_point(cornell):
    torch = pytest.importorskip("torch")
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    assert g.stats().actual_depth > 9          # deeper than the 8 LDS entries: the scratch's spill area is used
    # first call: more rays than 256 workgroups hold at once, ragged; second call: a smaller input
    for n, seed in ((300_007, 0xA110C01), (10_001, 0xA110C02)):
        ro, rd = synth.bounce_rays(V, T, n, seed)
        want, got = _trace_both(torch, g, ro, rd)
        assert np.array_equal(want, got), "n=%d: %d records differ" % (n, int((want != got).any(1).sum()))
        assert (want[:, 0] != -1).mean() > 0.5
    # a shallow tree (no spill) right after, on the same device's scratch
    gc = rt.Scene(cornell["verts"], cornell["tris"], device=0, **S.KD_PARAMS)
    ro, rd = synth.bounce_rays(cornell["verts"], cornell["tris"], 4_097, 0xA110C03)
    want, got = _trace_both(torch, gc, ro, rd)
    assert np.array_equal(want, got)
    # sizes beyond the stated maximum are rejected, not truncated
    with pytest.raises(rt.GiglError) as e:
        g.trace_closest_noalloc_device(1, 1, (1 << 24) + 1, 1, 0)
    assert e.value.code == 5   # GIGL_ERR_TOO_LARGE
    g.trace_closest_noalloc_device(0, 0, 0, 0, 0)   # empty batch: nothing to do
    g.close()
    gc.close()


def test_noalloc_call_allocates_nothing(cornell):
    """What the name promises: a warm call makes no stream-ordered allocation from the library's pool (its counter
    of pool allocations stays put, where the ordinary entry point moves it) and the device's free memory is what it
    was before the call."""
    torch = pytest.importorskip("torch")
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    n = 50_000
    ro, rd = synth.bounce_rays(V, T, n, 0xA110C04)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    out = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, out.data_ptr(), st)   # warm: module loaded
    torch.cuda.synchronize()
    allocs, free0 = rt.debug_pool_allocs(), torch.cuda.mem_get_info(0)[0]
    for _ in range(3):
        g.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, out.data_ptr(), st)
    torch.cuda.synchronize()
    assert rt.debug_pool_allocs() == allocs, "the no-allocation entry point went through the pool"
    assert torch.cuda.mem_get_info(0)[0] == free0
    g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, out.data_ptr(), st)   # the counter does count
    torch.cuda.synchronize()
    assert rt.debug_pool_allocs() > allocs
    g.close()


def test_noalloc_under_graph_capture(cornell):
    """The callers the entry point is for: captured once in a HIP graph, replayed several times, and once more
    after new rays were written into the same buffers.  Every replay equals gigl_trace_closest_device."""
    torch = pytest.importorskip("torch")
    V, T = synth.tessellate(cornell["verts"], cornell["tris"], 16)
    g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    n = 100_003
    ro, rd = synth.bounce_rays(V, T, n, 0xC0FFEE)
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    want = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    out = torch.zeros((n, 4), dtype=torch.int32, device="cuda:0")

    def reference():
        g.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()

    reference()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        g.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((out == 0).all()), "capturing must not run the launch"
    for it in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), "replay %d" % it
    ro2, rd2 = synth.bounce_rays(V, T, n, 0xC0FFEF)
    d_o.copy_(torch.from_numpy(ro2))
    d_d.copy_(torch.from_numpy(rd2))
    reference()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want), "replay on new rays"
    # and an ordinary call right after the replays finds the scratch as it expects it
    out.zero_()
    g.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    del graph
    g.close()
