#!/usr/bin/env python3
"""The one-launch VPL route at several block sizes (paths per launch), 150-VPL sets: ms per set.
usage: tools/experiments/vpl_lanes.py [paths_per_batch ...]   (GIGL_PHOTON_LANES in the environment picks paths per wave
in builds that read it)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gigl_amd import rt, scene_io as S
dev = torch.device("cuda:0")
ci = bench.cornell_inputs(torch, rt, S, dev)
cs, fog, mats = ci["scene"], ci["fog"], ci["mats"]
rt.set_option("photon_small", 1)
ref = None
for ppb in [int(a) for a in sys.argv[1:]] or [0, 512, 768, 1024]:
    call = lambda seed: cs.photon_trace_batch(mats, fog, S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS, S.PRIM_PL_INTENS,
                                              [0, -1, 0], 150, seed, ppb)
    for i in range(10):
        call(1000 + i)
    t = []
    for i in range(60):
        t0 = time.perf_counter(); n, got, n_paths = call(1 + i); t.append(time.perf_counter() - t0)
    n, got, n_paths = call(7)
    if ref is None:
        ref = got.copy()
    print("paths per launch %5d: median %.4f ms, min %.4f; seed 7: %d paths, records %s" % (
        ppb, np.median(t) * 1e3, np.min(t) * 1e3, n_paths, "equal" if np.array_equal(ref.view(np.uint8), got.view(np.uint8)) else "DIFFER"), flush=True)
rt.set_option("photon_small", -1)
