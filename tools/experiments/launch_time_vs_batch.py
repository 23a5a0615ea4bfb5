import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gigl_amd import rt, scene_io as S, synth
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, 158)
scene = rt.Scene(V, T, device=0, **S.KD_PARAMS)
n = 1 << 24
ro, rd = synth.bounce_rays(V, T, n, synth.RAY_SEED_16M, n_threads=16)
d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
d_h = torch.empty((n, 4), dtype=torch.int32, device="cuda")
main = torch.cuda.current_stream()
def t_launch(m, reps=8):
    scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), m, d_h.data_ptr(), main.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for k in range(reps):
        off = (k * m) % (n - m + 1) if m < n else 0
        scene.trace_closest_device(d_o.data_ptr() + off * 16, d_d.data_ptr() + off * 16, m, d_h.data_ptr() + off * 16, main.cuda_stream)
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("ideal share of the 2^24 launch: %.3f ms per 2^20 rays" % (t_launch(n, 3) / 16))
for lg in (14, 16, 18, 19, 20, 21, 22, 23):
    m = 1 << lg
    row = []
    for chunk in (0, 64, 128, 256):
        for bpc in (0,):
            for guided in (4,):
                rt.set_option("chunk", chunk); rt.set_option("blocks_per_cu", bpc); rt.set_option("guided", guided)
                row.append("c%d b%d g%d: %.3f" % (chunk, bpc, guided, t_launch(m)))
    print("2^%d rays  " % lg + "  ".join(row))
