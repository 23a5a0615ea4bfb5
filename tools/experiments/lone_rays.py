#!/usr/bin/env python3
"""How fast does a ray step when its wave is (nearly) alone on the chip?  The longest rays of the bench batch
(build/long_rays_s<tess>.npy: indices and oracle node / triangle-test counts of the 4,096 longest of the first 2^21,
written on the CPU by the oracle walking a host-only tree)
traced as batches of 1, 64, 4096 rays: time / node visits = the per-node latency of the drain at the end of a launch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gigl_amd import rt, scene_io as S, synth
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
TESS = int(sys.argv[1]) if len(sys.argv) > 1 else 158
V, T = synth.tessellate(cv, ct, TESS)
scene = rt.Scene(V, T, device=0, **S.KD_PARAMS)
idx, nodes, tris = np.load(os.path.join(ROOT, "build", "long_rays_s%d.npy" % TESS))
ro, rd = synth.bounce_rays(V, T, 1 << 21, synth.RAY_SEED_16M if TESS == 158 else synth.RAY_SEED_128M, first=0, n_threads=16)
dev = torch.device("cuda:0")
sh = torch.cuda.current_stream().cuda_stream
for name, sel in (("the longest ray alone", idx[:1]), ("the 64 longest (one wave)", idx[:64]), ("rays 64..127", idx[64:128]),
                  ("the ray with most triangle tests alone", idx[np.argmax(tris)][None]), ("the 4,096 longest (64 waves)", idx[:4096])):
    o = torch.from_numpy(ro[sel].copy()).to(dev); d = torch.from_numpy(rd[sel].copy()).to(dev)
    hits = torch.empty((len(sel), 4), dtype=torch.int32, device=dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record(); scene.trace_closest_device(o.data_ptr(), d.data_ptr(), len(sel), hits.data_ptr(), sh); e1.record()
        torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    k = np.isin(idx, sel)
    lean = 1e9
    hits2 = torch.empty_like(hits)
    for rep in range(5):
        e0.record(); scene.debug_trace_closest_lean_device(o.data_ptr(), d.data_ptr(), len(sel), hits2.data_ptr(), sh); e1.record()
        torch.cuda.synchronize(); lean = min(lean, e0.elapsed_time(e1))
    same = bool(torch.equal(hits, hits2))
    print("%-34s %7.3f ms; longest %d nodes + %d tests -> %.3f us per node of the longest ray; all: %d nodes; lean walk %.3f ms, records %s" % (
        name, best, nodes[k].max(), tris[k][np.argmax(nodes[k])], best * 1e3 / nodes[k].max(), nodes[k].sum(), lean,
        "equal" if same else "DIFFER"), flush=True)

# the whole 2^21-ray batch through both kernels
o = torch.from_numpy(ro).to(dev); d = torch.from_numpy(rd).to(dev)
h1 = torch.empty((len(ro), 4), dtype=torch.int32, device=dev); h2 = torch.empty_like(h1)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
for rep in range(2):
    e0.record(); scene.trace_closest_device(o.data_ptr(), d.data_ptr(), len(ro), h1.data_ptr(), sh); e1.record()
    scene.debug_trace_closest_lean_device(o.data_ptr(), d.data_ptr(), len(ro), h2.data_ptr(), sh); e2.record()
    torch.cuda.synchronize()
print("all 2^21 rays: shipped kernel %.3f ms, lean one-ray-per-lane kernel %.3f ms, records %s" % (
    e0.elapsed_time(e1), e1.elapsed_time(e2), "equal" if torch.equal(h1, h2) else "DIFFER"), flush=True)
