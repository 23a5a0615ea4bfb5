#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference cannot be built in this image, so these vectors come from the oracle
restatement, which is itself pinned to the reference's recorded outputs (survey_facts.json).  They
freeze the oracle's answers so that a later change to oracle or engine is caught against data:
inputs and expected outputs only.

  python3 tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gigl_amd import scene_io as S, synth  # noqa: E402
from oracle import oracle_py as O          # noqa: E402


def main():
    cv, ct, mats, _ = S.load_obj(os.path.join(HERE, "cornell_box.obj"))
    # 1. Cornell: every 257th primary ray of the 1024^2 frame (closest + any)
    ora = O.OracleScene(cv, ct, **S.KD_PARAMS)
    cam = O.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, 1024, 1024)
    ro, rd = O.primary_rays(cam, 1024, 0, 1024 * 1024)
    ro, rd = ro[::257], rd[::257]
    hits, cnt = ora.trace_closest(ro, rd, counters=True)
    np.savez_compressed(os.path.join(HERE, "cornell_primary_subsample.npz"), o_tmin=ro, d_tmax=rd,
                        hits=hits.view(np.uint32).reshape(-1, 4), occluded=ora.trace_any(ro, rd), counters=cnt)
    # 2. S(6): incoherent bounce rays on a tessellated scene (tree depth 27)
    V, T = synth.tessellate(cv, ct, 6)
    os6 = O.OracleScene(V, T, **S.KD_PARAMS)
    ro, rd = synth.bounce_rays(V, T, 3000, 0xF1C5)
    hits, cnt = os6.trace_closest(ro, rd, counters=True)
    nodes = os6.nodes()
    np.savez_compressed(os.path.join(HERE, "s6_bounce_rays.npz"), verts=V, tris=T, o_tmin=ro, d_tmax=rd,
                        hits=hits.view(np.uint32).reshape(-1, 4), occluded=os6.trace_any(ro, rd), counters=cnt,
                        tree_nodes=nodes, tree_leaf_prim_ids=os6.leaf_prim_ids())
    # 3. fog: density samples and transmittance of random segments through the reference volume
    dens = O.make_density((64, 64, 64), 12.0, 1.0)
    box = ora.bbox().copy()
    box[4] = min(box[4], np.float32(S.MAX_FOG_HEIGHT))
    fog = O.Fog(dens, box, np.float32(S.ABS_K) + np.float32(S.SCA_K))
    rng = np.random.default_rng(3)
    lo, hi = box[:3] - 30, box[3:] + 30
    pos = (lo + (hi - lo) * rng.random((2000, 3))).astype(np.float32)
    d = np.array([O.sample_density(fog, p) for p in pos], np.float32)
    p = (lo + (hi - lo) * rng.random((2000, 3))).astype(np.float32)
    L = (lo + (hi - lo) * rng.random((2000, 3))).astype(np.float32)
    v = L - p
    dsq = (v * v).sum(1, dtype=np.float32)
    w = v * (np.float32(1.0) / np.sqrt(dsq))[:, None]
    seg_p = np.concatenate([p, np.zeros((2000, 1), np.float32)], 1)
    seg_d = np.concatenate([w, dsq[:, None]], 1).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "fog_samples.npz"), box=box, pos=pos, density=d, seg_p=seg_p,
                        seg_d_distsq=seg_d, transmittance=O.transmittance(fog, seg_p, seg_d))
    # 4. VPL set for a recorded random-number stream (PARITY UNPINNED against the reference)
    stream = np.random.default_rng(20261110).random(8192, dtype=np.float32)
    n, vpls, n_paths, used = O.photon_trace(ora, mats, fog, S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS,
                                            S.PRIM_PL_INTENS, [0, -1, 0], S.MAX_N_VPLS, stream)
    assert n == S.MAX_N_VPLS
    np.savez_compressed(os.path.join(HERE, "vpl_set.npz"), rng=stream[:used], n_paths=n_paths,
                        vpls=np.frombuffer(vpls.tobytes(), np.uint32).reshape(n, 24))
    for f in sorted(os.listdir(HERE)):
        print("%-36s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
