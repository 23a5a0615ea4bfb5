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
