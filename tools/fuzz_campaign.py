#!/usr/bin/env python3
"""Extended randomised parity campaign (GPU box): tests/test_gpu_fuzz.py's soups and rays over many
seeds, random builder parameters and every device layout; closest-hit records and occlusion bits
must be the oracle's bit for bit.  usage: python tools/fuzz_campaign.py [n_scenes=60] [first_seed=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gigl_amd import rt
from oracle import oracle_py as O
from test_gpu_fuzz import rays_for, soup

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t0 = time.time()
for seed in range(first, first + n_scenes):
    rng = np.random.default_rng(77000 + seed)
    n_tris = int(rng.choice([1, 2, 3, 17, 200, 900, 4000, 20000]))
    scale = float(rng.choice([1e-3, 1.0, 37.0, 5e4]))
    flavour = str(rng.choice(["plain", "axis", "degenerate"]))
    params = dict(inters_cost=int(rng.choice([1, 10, 80])), trav_cost=int(rng.choice([1, 5])),
                  max_depth=int(rng.choice([-1, 3, 8, 18, 30, 31])), min_num_prims=int(rng.choice([0, 1, 2, 4, 9])),
                  empty_bonus=float(rng.choice([0.0, 0.25, 0.5])))
    verts, tris = soup(rng, n_tris, scale, flavour)
    ora = O.OracleScene(verts, tris, **params)
    ro, rd = rays_for(rng, verts, tris, 20001, scale)
    want = ora.trace_closest(ro, rd, n_threads=8)
    occ = ora.trace_any(ro, rd, n_threads=8).astype(bool)
    for layout in (0, 1, 2, 3):   # 3: triple records with leaf triangle blocks
        for stack_lds in (8, 16):
            rt.set_option("layout", min(layout, 2))
            rt.set_option("leaf_blocks", 2 if layout == 3 else 0)
            rt.set_option("stack_lds", stack_lds)
            gpu = rt.Scene(verts, tris, device=0, **params)
            got = gpu.trace(ro, rd)
            ok = np.array_equal(got["tri_id"], want["tri_id"])
            for k in ("t", "u", "v"):
                ok = ok and np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32))
            for mode in (0, 1):
                rt.set_option("any_atomic", mode)
                ok = ok and np.array_equal(gpu.trace(ro, rd, is_vis_ray=True), occ)
            rt.set_option("any_atomic", -1)
            if ok and layout in (2, 3) and stack_lds == 8:
                # round 3: the same scene through a device list that repeats device 0 (replication, scatter, gather;
                # host-pointer ranges on worker threads) and through a scene adopted from its flattened device arrays
                m = rt.MultiDeviceScene(gpu, [0, 0, 0])
                ok = ok and np.array_equal(m.trace(ro, rd).view(np.uint32), got.view(np.uint32))
                ok = ok and np.array_equal(m.trace_any(ro, rd).astype(bool), occ)
                m.close()
                desc, ptrs = gpu.export_device_blob()
                rep = rt.Scene.from_device_blob(desc, ptrs[0], ptrs[1], ptrs[2], device=0)
                ok = ok and np.array_equal(rep.trace(ro, rd).view(np.uint32), got.view(np.uint32))
                rep.close()
                # the lean one-ray-per-lane walk of small VPL sets (gigl_walk_lean.h) over the same rays
                import torch
                d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
                h = torch.zeros((len(ro), 4), dtype=torch.int32, device="cuda")
                gpu.debug_trace_closest_lean_device(d_o.data_ptr(), d_d.data_ptr(), len(ro), h.data_ptr(), 0)
                torch.cuda.synchronize()
                ok = ok and np.array_equal(h.cpu().numpy().view(np.uint32).reshape(-1), got.view(np.uint32).reshape(-1))
            if not ok:
                bad += 1
                print("MISMATCH seed %d layout %d stack %d tris %d %s %r" % (seed, layout, stack_lds, n_tris, flavour, params),
                      flush=True)
            gpu.close()
    ora.close()
    if seed % 10 == 9:
        print("seed %d done, %.0f s, mismatching configurations so far: %d" % (seed, time.time() - t0, bad), flush=True)
rt.set_option("layout", 2)
rt.set_option("leaf_blocks", -1)
rt.set_option("stack_lds", 8)
print("campaign: %d scenes x 4 layouts x 2 stack depths x (closest + 2 any-hit modes): %d mismatching configurations"
      % (n_scenes, bad))
sys.exit(1 if bad else 0)
