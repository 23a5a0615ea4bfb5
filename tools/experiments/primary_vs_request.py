import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gigl_amd import rt, scene_io as S, synth
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
main = torch.cuda.current_stream()
for tess in (1, 158):
    V, T = (cv, ct) if tess == 1 else synth.tessellate(cv, ct, tess)
    sc = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    for res in (512, 1024, 2048):
        cam = rt.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, res, res)
        npix = res * res
        ph = torch.empty((npix, 4), dtype=torch.int32, device="cuda")
        row = []
        for chunk in (0, 64, 128, 256):
            rt.set_option("chunk", chunk)
            for _ in range(3):
                sc.trace_primary_device(cam, res, 0, npix, ph.data_ptr(), main.cuda_stream)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main)
            for _ in range(10):
                sc.trace_primary_device(cam, res, 0, npix, ph.data_ptr(), main.cuda_stream)
            b.record(main); torch.cuda.synchronize()
            row.append("c%d: %.3f" % (chunk, a.elapsed_time(b) / 10))
        print("S(%d) primary %dx%d  " % (tess, res, res) + "  ".join(row))
    sc.close()
