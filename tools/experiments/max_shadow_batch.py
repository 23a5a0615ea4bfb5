#!/usr/bin/env python3
"""A shadow batch near the pair limit: Cornell 2048^2 primary hit points x 900 lights = 3.77 G pairs (index
arithmetic beyond 2^31, light-major mapping with large factors).  The two pair orders and both any-hit output
modes must agree bit for bit; 2,000 sampled pixels x all lights are checked against the oracle.  One-off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gigl_amd import rt, scene_io as S
from oracle import oracle_py as O
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
sc = rt.Scene(cv, ct, device=0, **S.KD_PARAMS)
ora = O.OracleScene(cv, ct, **S.KD_PARAMS)
res = 2048
npix = res * res
cam = rt.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, res, res)
st = torch.cuda.current_stream().cuda_stream
po = torch.empty((npix, 4), dtype=torch.float32, device="cuda")
pd = torch.empty((npix, 4), dtype=torch.float32, device="cuda")
ph = torch.empty((npix, 4), dtype=torch.int32, device="cuda")
sc.gen_primary_rays_device(cam, res, 0, npix, po.data_ptr(), pd.data_ptr(), st)
sc.trace_primary_device(cam, res, 0, npix, ph.data_ptr(), st)
torch.cuda.synchronize()
nl = 900
rng = np.random.default_rng(5)
lights = np.stack([rng.uniform(5, 550, nl), rng.uniform(5, 545, nl), rng.uniform(5, 555, nl)], 1).astype(np.float32)
dl = torch.from_numpy(lights).cuda()
npair = npix * nl
print("pairs: %d (%.2f x 2^31)" % (npair, npair / 2 ** 31), flush=True)
results = {}
for order, atomic in ((1, -1), (0, -1), (0, 0), (1, 1)):
    rt.set_option("shadow_order", order); rt.set_option("any_atomic", atomic)
    bits = torch.full(((npair + 63) // 64,), -1, dtype=torch.int64, device="cuda")
    t = time.perf_counter()
    sc.shadow_batch_device(po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, dl.data_ptr(), nl, bits.data_ptr(), stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    results[(order, atomic)] = bits
    print("shadow_order=%d any_atomic=%d: %.3f s = %.1f G pairs/s, occluded pairs: %d" % (
        order, atomic, dt, npair / dt / 1e9, int(sum(int(torch.bitwise_and(bits >> k, 1).sum()) for k in range(64)))), flush=True)
rt.set_option("shadow_order", 1); rt.set_option("any_atomic", -1)
ref = results[(1, -1)]
bad = {k: int((v != ref).sum()) for k, v in results.items()}
print("words differing from the default order:", bad, flush=True)
# oracle on sampled pixels
sel = np.arange(0, npix, 2099)[:2000]
so, sd = po[sel].cpu().numpy(), pd[sel].cpu().numpy()
sh = ph[sel].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
want_hits = ora.trace_closest(so, sd, n_threads=8)
assert np.array_equal(want_hits["tri_id"], sh["tri_id"]) and np.array_equal(want_hits["t"].view(np.uint32), sh["t"].view(np.uint32))
want_occ = ora.shadow_batch(so, sd, want_hits, lights, n_threads=8)
if isinstance(want_occ, tuple):
    want_occ = want_occ[0]
want01 = np.where(want_occ == 0xFF, 0, want_occ).astype(np.uint8)
pair = (sel[:, None].astype(np.int64) * nl + np.arange(nl)[None, :]).reshape(-1)
words = ref[torch.from_numpy(pair >> 6).cuda()].cpu().numpy().view(np.uint64)
got = ((words >> (pair & 63).astype(np.uint64)) & np.uint64(1)).astype(np.uint8).reshape(len(sel), nl)
mism = int((got != want01).sum())
print("oracle check: %d pixels x %d lights, %d pairs differ (occluded fraction %.3f)" % (len(sel), nl, mism, want01.mean()), flush=True)
sys.exit(1 if (mism or any(bad.values())) else 0)
