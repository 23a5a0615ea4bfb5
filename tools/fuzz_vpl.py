#!/usr/bin/env python3
"""Randomised parity campaign for VPL generation (GPU box): triangle soups with random builder parameters, materials,
light positions, fog volumes and set sizes; gigl_photon_trace_batch through both routes (the chain of batched launches
and, for sets of up to 1,024 records, the one-launch kernel with its lean walk) against the oracle's serial walk in the
same substream mode — record count, path count and all 96 bytes of every record.
usage: python tools/fuzz_vpl.py [n_scenes=40] [first_seed=0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gigl_amd import rt
from oracle import oracle_py as O
from test_gpu_fuzz import soup

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = sets = gave_up = 0
t0 = time.time()
for seed in range(first, first + n_scenes):
    rng = np.random.default_rng(991000 + seed)
    n_tris = int(rng.choice([12, 60, 400, 3000]))
    scale = float(rng.choice([1.0, 37.0, 550.0]))
    flavour = str(rng.choice(["plain", "axis", "degenerate"]))
    params = dict(inters_cost=int(rng.choice([1, 80])), trav_cost=int(rng.choice([1, 5])),
                  max_depth=int(rng.choice([-1, 3, 12, 31])), min_num_prims=int(rng.choice([0, 1, 4])),
                  empty_bonus=float(rng.choice([0.0, 0.5])))
    verts, tris = soup(rng, n_tris, scale, flavour)
    # large triangles so that paths meet something: every fourth one blown up about its first vertex
    V = verts.reshape(-1, 3, 3).copy()
    V[::4, 1:] = V[::4, :1] + (V[::4, 1:] - V[::4, :1]) * np.float32(rng.choice([3.0, 12.0]))
    verts = np.ascontiguousarray(V.reshape(-1, 3))
    mats = np.zeros((3, 12), np.float32)
    mats[:, 0:3] = rng.uniform(0.05, 0.95, (3, 3))      # k_d
    mats[:, 4:7] = rng.uniform(0.0, 0.4, (3, 3))        # k_s
    mats[:, 7] = rng.choice([1.0, 10.0, 200.0], 3)      # n_s
    if rng.random() < 0.3:
        mats[2, 8:11] = 1.0                             # an emissive material: ends the paths that meet it
    ora = O.OracleScene(verts, tris, **params)
    gpu = rt.Scene(verts, tris, device=0, **params)
    box = gpu.bbox().copy()
    dens = rng.integers(0, 256, (int(rng.choice([1, 5, 16])), int(rng.choice([2, 9])), int(rng.choice([3, 16]))), dtype=np.uint8)
    abs_k, sca_k = float(rng.choice([1e-5, 0.02])) / scale, float(rng.choice([1e-2, 0.5])) / scale
    maj = (abs_k + sca_k) * float(rng.choice([1.0, 1.7]))
    gfog = rt.FogVolume(dens, box, abs_k, sca_k)
    ofog = O.Fog(dens, box, np.float32(abs_k) + np.float32(sca_k))
    for case in range(4):
        pos = (box[:3] + (box[3:] - box[:3]) * rng.uniform(0.2, 0.8, 3)).astype(np.float32)
        intens = rng.uniform(0.5, 40.0, 3).astype(np.float32)
        n_vpls = int(rng.choice([1, 9, 150, 700, 1500]))
        fog_on = int(rng.integers(0, 2))
        s = int(rng.integers(0, 2 ** 63))
        n, want, want_paths, _ = O.photon_trace_substreams(ora, mats, ofog, abs_k, sca_k, maj, fog_on, pos, intens,
                                                           [0, -1, 0], n_vpls, s)
        want = np.frombuffer(want.tobytes(), np.uint32).reshape(-1, 24)
        for route in (0, 1):
            rt.set_option("photon_small", route)
            g_n, got, got_paths = gpu.photon_trace_batch(mats, gfog, abs_k, sca_k, maj, fog_on, pos, intens, [0, -1, 0],
                                                         n_vpls, s)
            sets += 1
            gave_up += int(g_n == 0)
            if not (g_n == n and got_paths == want_paths and np.array_equal(got[:g_n], want[:n])):
                bad += 1
                print("MISMATCH seed %d case %d route %d: %d/%d records, %d/%d paths, tris %d %s %r" % (
                    seed, case, route, g_n, n, got_paths, want_paths, n_tris, flavour, params), flush=True)
    rt.set_option("photon_small", -1)
    gfog.close(); gpu.close(); ora.close()
    if seed % 10 == 9:
        print("seed %d done, %.0f s, %d sets (%d of them given up as the reference does), mismatches so far: %d"
              % (seed, time.time() - t0, sets, gave_up, bad), flush=True)
print("VPL campaign: %d scenes, %d sets through both routes (%d given up): %d mismatching sets" % (n_scenes, sets, gave_up, bad))
sys.exit(1 if bad else 0)
