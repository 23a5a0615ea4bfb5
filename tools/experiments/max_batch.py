#!/usr/bin/env python3
"""The largest batch the boundary accepts (0xF0000000 rays, 193 GB of rays and records on one GPU): a known
2^24-ray set tiled 240 times, every record and every occlusion bit compared with the set's own records.
Index arithmetic beyond 2^31 / 2^32 bytes is what this exercises.  One-off (too large for the test suite)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gigl_amd import rt, scene_io as S, synth
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, 16)
sc = rt.Scene(V, T, device=0, **S.KD_PARAMS)
base = 1 << 24
reps = 240
n = base * reps
assert n == 0xF0000000
ro, rd = synth.bounce_rays(V, T, base, 0xB16, n_threads=16)
b_o, b_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
want = torch.empty((base, 4), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
sc.trace_closest_device(b_o.data_ptr(), b_d.data_ptr(), base, want.data_ptr(), st)
torch.cuda.synchronize()
wbits = torch.zeros(base // 64, dtype=torch.int64, device="cuda")
sc.trace_any_device(b_o.data_ptr(), b_d.data_ptr(), base, wbits.data_ptr(), st)
torch.cuda.synchronize()
print("base set: %d hits of %d" % (int((want[:, 0] != -1).sum()), base), flush=True)
big_o = torch.empty((n, 4), dtype=torch.float32, device="cuda")
big_d = torch.empty((n, 4), dtype=torch.float32, device="cuda")
for k in range(reps):
    big_o[k * base:(k + 1) * base] = b_o
    big_d[k * base:(k + 1) * base] = b_d
hits = torch.empty((n, 4), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
print("allocated %.1f GB" % (torch.cuda.memory_allocated() / 1e9), flush=True)
t = time.perf_counter()
sc.trace_closest_device(big_o.data_ptr(), big_d.data_ptr(), n, hits.data_ptr(), st)
torch.cuda.synchronize()
dt = time.perf_counter() - t
bad = 0
for k in range(reps):
    bad += int((hits[k * base:(k + 1) * base] != want).any(1).sum())
print("closest-hit: %d rays in %.2f s = %.0f Mrays/s; records differing from the tiled base set: %d" % (n, dt, n / dt / 1e6, bad), flush=True)
del hits
bad_any = {}
for mode in (0, 1):
    rt.set_option("any_atomic", mode)
    bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    t = time.perf_counter()
    sc.trace_any_device(big_o.data_ptr(), big_d.data_ptr(), n, bits.data_ptr(), st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    bad_any[mode] = int((bits.view(reps, base // 64) != wbits[None, :]).sum())
    print("any-hit (any_atomic=%d): %.2f s = %.0f Mrays/s; words differing: %d" % (mode, dt, n / dt / 1e6, bad_any[mode]), flush=True)
    del bits
try:
    sc.trace_closest_device(big_o.data_ptr(), big_d.data_ptr(), n + 1, big_o.data_ptr(), st)
    print("n + 1 accepted: WRONG")
except rt.GiglError as e:
    print("n + 1 rejected with code", e.code)
sys.exit(1 if (bad or any(bad_any.values())) else 0)
