"""Writes the bench scene S(k) and a sample of its incoherent ray batch as raw files for layout_sim.

usage: python tools/layout_sim/dump_inputs.py [k=158] [log2_rays=18] [out_dir=/tmp/layout_sim]
"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from gigl_amd import scene_io as S, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 158
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
out = sys.argv[3] if len(sys.argv) > 3 else "/tmp/layout_sim"
os.makedirs(out, exist_ok=True)
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, k)
ro, rd = synth.bounce_rays(V, T, 1 << log2n, synth.RAY_SEED_16M)
np.array([len(V), len(T), len(ro)], np.uint32).tofile(os.path.join(out, "header.bin"))
V.astype(np.float32).tofile(os.path.join(out, "verts.bin"))
T.astype(np.uint32).tofile(os.path.join(out, "tris.bin"))
ro.tofile(os.path.join(out, "ro.bin"))
rd.tofile(os.path.join(out, "rd.bin"))
print("wrote", out, len(V), len(T), len(ro))
