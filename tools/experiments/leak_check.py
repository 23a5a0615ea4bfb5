#!/usr/bin/env python3
"""Device memory before and after cycles of create / use / destroy of every handle type (scene, fog volume,
multi-device handle; traces, shadow batch, VPL batch, host-pointer pipeline in between).  The library's pool
keeps what it allocated (by design), so the first cycle may grow it; later cycles must not."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gigl_amd import rt, scene_io as S, synth
from oracle import oracle_py as O
cv, ct, mats, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, 32)
dens = O.make_density((64, 64, 64), 12.0, 1.0)
n = (1 << 21) + 5000
ro, rd = synth.bounce_rays(V, T, n, 3, n_threads=8)
torch.cuda.init(); torch.zeros(1, device="cuda")
def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20
start = free_mb()
hist = []
for cyc in range(8):
    sc = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    box = sc.bbox().copy()
    fog = rt.FogVolume(dens, box, S.ABS_K, S.SCA_K)
    h = sc.trace(ro, rd)                      # host-pointer pipeline (two chunks)
    occ = sc.trace(ro[:100000], rd[:100000], is_vis_ray=True)
    m = rt.MultiDeviceScene(sc, [0, 0])
    h2 = m.trace(ro[:300000], rd[:300000])
    m.close()
    cs = rt.Scene(cv, ct, device=0, **S.KD_PARAMS)
    materials = np.asarray(mats, np.float32).reshape(-1, 12)   # load_obj: k_d(3) pad k_s(3) n_s k_e(3) pad
    nv, vp, npth = cs.photon_trace_batch(materials, None, S.ABS_K, S.SCA_K, 0.0, False, S.PRIM_PL_POS, (1.0, 1.0, 1.0), (0, -1, 0), 2000, 42 + cyc)
    assert nv == 2000
    cs.close(); fog.close(); sc.close()
    hist.append(free_mb())
    print("cycle %d: free %.1f MiB (start %.1f)" % (cyc, hist[-1], start), flush=True)
drift = hist[1] - hist[-1]
print("kept by the pool after the first cycle: %.1f MiB; drift over the next %d cycles: %.1f MiB" % (start - hist[0], len(hist) - 1, drift))
sys.exit(1 if drift > 8.0 else 0)
