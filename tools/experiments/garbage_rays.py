#!/usr/bin/env python3
"""Rays made of random bit patterns (NaN, infinities, denormals, huge values in any component) against the
oracle on S(16): the traversal must terminate and agree (ids bit-exact; t/u/v bit-exact or NaN on both sides;
any-hit bits).  One-off robustness check; run under a time limit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from gigl_amd import rt, scene_io as S, synth
from oracle import oracle_py as O
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, 16)
g = rt.Scene(V, T, device=0, **S.KD_PARAMS)
ora = O.OracleScene(V, T, **S.KD_PARAMS)
rng = np.random.default_rng(20261111)
n = 200_000
total_bad = 0
for flavour in ("all components random bits", "one random-bit component in a sane ray", "special values in a sane ray"):
    ro, rd = synth.bounce_rays(V, T, n, 0x6A7BA6E)
    ro, rd = ro.copy(), rd.copy()
    if flavour.startswith("all"):
        ro = rng.integers(0, 2 ** 32, (n, 4), dtype=np.uint64).astype(np.uint32).view(np.float32)
        rd = rng.integers(0, 2 ** 32, (n, 4), dtype=np.uint64).astype(np.uint32).view(np.float32)
    elif flavour.startswith("one"):
        both = np.concatenate([ro, rd], 1)
        col = rng.integers(0, 8, n)
        both[np.arange(n), col] = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        ro, rd = np.ascontiguousarray(both[:, :4]), np.ascontiguousarray(both[:, 4:])
    else:
        specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38, 1e-38], np.float32)
        both = np.concatenate([ro, rd], 1)
        for _ in range(2):
            col = rng.integers(0, 8, n)
            both[np.arange(n), col] = specials[rng.integers(0, len(specials), n)]
        ro, rd = np.ascontiguousarray(both[:, :4]), np.ascontiguousarray(both[:, 4:])
    with np.errstate(all="ignore"):
        want = ora.trace_closest(ro, rd, n_threads=8)
        want_any = ora.trace_any(ro, rd, n_threads=8)
    got = g.trace(ro, rd)
    got_any = g.trace(ro, rd, is_vis_ray=True)
    bad_id = int((got["tri_id"] != want["tri_id"]).sum())
    hit = (got["tri_id"] != rt.GIGL_MISS) & (got["tri_id"] == want["tri_id"])
    bad_f = 0
    for k in ("t", "u", "v"):
        a, b = got[k][hit], want[k][hit]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        bad_f += int((~same).sum())
    bad_any = int((got_any.astype(bool) != want_any.astype(bool)).sum())
    print("%-45s hits %6d  id mismatches %d  t/u/v mismatches %d  any-hit mismatches %d" % (flavour, int(hit.sum()), bad_id, bad_f, bad_any), flush=True)
    total_bad += bad_id + bad_f + bad_any
sys.exit(1 if total_bad else 0)
