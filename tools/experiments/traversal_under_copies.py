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
C = 1 << 21
p_in = torch.from_numpy(ro).pin_memory()
p_out = torch.zeros((n, 4), dtype=torch.int32).pin_memory()
stage = torch.empty((n, 4), dtype=torch.float32, device="cuda")
side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
def launches():
    for k in range(8):
        scene.trace_closest_device(d_o.data_ptr() + k * C * 16, d_d.data_ptr() + k * C * 16, C, d_h.data_ptr() + k * C * 16, main.cuda_stream)
def run(tag, h2d, d2h):
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        if h2d:
            with torch.cuda.stream(side):
                for k in range(16):
                    stage[k * (C // 2):(k + 1) * (C // 2)].copy_(p_in[k * (C // 2):(k + 1) * (C // 2)], non_blocking=True)
        if d2h:
            with torch.cuda.stream(side2):
                for k in range(8):
                    p_out[k * C:(k + 1) * C].copy_(d_h[k * C:(k + 1) * C], non_blocking=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main); launches(); b.record(main)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t) * 1e3
    print("%-28s kernels %.2f ms (8 x 2^21 rays), wall %.2f ms" % (tag, a.elapsed_time(b), tot))
scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr(), main.cuda_stream); torch.cuda.synchronize()
t = time.perf_counter(); scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr(), main.cuda_stream); torch.cuda.synchronize()
print("one 2^24 launch %.2f ms" % ((time.perf_counter() - t) * 1e3))
run("alone", False, False)
run("with H2D 512 MB", True, False)
run("with D2H 256 MB", False, True)
run("with both", True, True)
