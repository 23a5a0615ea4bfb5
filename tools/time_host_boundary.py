#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer boundary (gigl_trace_closest / gigl_trace_any): rays and records
in host memory, pageable (numpy) and page-locked (torch pinned tensors), S(158) and the bench batch.
Not a bench line: bench.py's value is measured with inputs resident in HBM.  usage: time_host_boundary.py [log2 rays]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gigl_amd import rt, scene_io as S, synth  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = 1 << lg
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, 158)
    scene = rt.Scene(V, T, device=0, **S.KD_PARAMS)
    ro, rd = synth.bounce_rays(V, T, n, synth.RAY_SEED_16M, n_threads=16)
    L = rt.lib()
    # resident reference
    d_o, d_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    d_h = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    for _ in range(2):
        scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr(), 0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, d_h.data_ptr(), 0)
    torch.cuda.synchronize()
    res_ms = (time.perf_counter() - t) * 1e3
    want = d_h.cpu().numpy()
    print("lib %s  2^%d rays  resident: %.2f ms = %.0f Mrays/s" % (os.environ.get("GIGL_RT_LIB", "in-tree"), lg, res_ms, n / res_ms / 1e3))

    nbad_calls = [0]

    def run(tag, o_ptr, d_ptr, hits_ptr, bits_ptr, read_hits, read_bits):
        for name, fn, outp in (("closest", L.gigl_trace_closest, hits_ptr), ("any", L.gigl_trace_any, bits_ptr)):
            ms = []
            for rep in range(6):
                t = time.perf_counter()
                rc = fn(scene.h, C.c_void_p(o_ptr), C.c_void_p(d_ptr), n, C.c_void_p(outp))
                ms.append((time.perf_counter() - t) * 1e3)
                assert rc == 0, L.gigl_last_error()
                if name == "closest":
                    nb = int((read_hits() != want).any(1).sum())
                else:
                    nb = int((rt.unpack_bits(read_bits(), n) != (want[:, 0] != -1)).sum())
                if nb:
                    print("  MISMATCH %s %s call %d: %d records" % (tag, name, rep, nb))
                    nbad_calls[0] += 1
            best = min(ms[1:])
            print("  %-9s %-8s %.2f ms = %.0f Mrays/s  (%.1f GB/s over the link, both directions)"
                  % (tag, name, best, n / best / 1e3, n * (48 if name == "closest" else 32.125) / best / 1e6))
        bad = np.nonzero((read_hits() != want).any(1))[0]
        occ = rt.unpack_bits(read_bits(), n)
        bad_any = np.nonzero(occ != (want[:, 0] != -1))[0]
        for what, b in (("closest", bad), ("any", bad_any)):
            if len(b):
                print("  MISMATCH %s %s: %d records; per 2^19-ray block: %s" % (tag, what, len(b), np.bincount(b >> 19, minlength=n >> 19).tolist()))
        return len(bad) + len(bad_any)

    hits = np.zeros((n, 4), np.int32)
    bits = np.zeros((n + 63) // 64, np.uint64)
    nbad = run("pageable", ro.ctypes.data, rd.ctypes.data, hits.ctypes.data, bits.ctypes.data, lambda: hits, lambda: bits)
    p_o, p_d = torch.from_numpy(ro).pin_memory(), torch.from_numpy(rd).pin_memory()
    p_h = torch.zeros((n, 4), dtype=torch.int32).pin_memory()
    p_b = torch.zeros((n + 63) // 64, dtype=torch.int64).pin_memory()
    nbad += run("pinned", p_o.data_ptr(), p_d.data_ptr(), p_h.data_ptr(), p_b.data_ptr(), lambda: p_h.numpy(),
        lambda: p_b.numpy().view(np.uint64))
    scene.close()
    if nbad or nbad_calls[0]:
        raise SystemExit("host-pointer records differ from the device-pointer call")


if __name__ == "__main__":
    main()
