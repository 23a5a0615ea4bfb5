#!/usr/bin/env python3
"""Interleaved A/B timing of engine options on the bench workload (one process, one device).

usage: python3 tools/sweep.py [--tess 158] [--rays-log2 24] [--rounds 3] name=v,name=v ...
Each positional argument is one variant: a comma-separated list of gigl_set_option pairs, plus the
pseudo-options `sorted=1` (rays pre-sorted on the host by origin cell + direction octant) and
`any=1` (any-hit instead of closest-hit) and `layout=L` (scene created under device layout L).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def morton_sort_key(ro, rd, box, bits=10):
    lo, hi = box[:3], box[3:]
    q = np.clip(((ro[:, :3] - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)

    def spread(v):
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> b) & 1) << (3 * b)
        return r

    m = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = (rd[:, 0] < 0).astype(np.int64) | ((rd[:, 1] < 0).astype(np.int64) << 1) | ((rd[:, 2] < 0).astype(np.int64) << 2)
    return (octant << (3 * bits)) | m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tess", type=int, default=158)
    ap.add_argument("--rays-log2", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--treelet", type=int, default=0, help="treelet_levels option set before the scene is created")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch
    from gigl_amd import rt, scene_io as S, synth
    dev = torch.device("cuda", 0)
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, a.tess) if a.tess > 1 else (cv, ct)
    if a.treelet:
        rt.set_option("treelet_levels", a.treelet)
    scenes = {}

    def scene_for(layout, leaf_blocks=-1):   # layout=L, leaf_blocks=B: a scene created under those options (kd_layout.h)
        key = (layout, leaf_blocks)
        if key not in scenes:
            rt.set_option("layout", layout)
            try:
                rt.set_option("leaf_blocks", leaf_blocks)
            except rt.GiglError:
                pass   # an older library under GIGL_RT_LIB
            scenes[key] = rt.Scene(V, T, device=0, **S.KD_PARAMS)
            print("layout", layout, "leaf_blocks", leaf_blocks, "device bytes", scenes[key].stats().device_bytes, flush=True)
        return scenes[key]
    scene = scene_for(2)
    n = 1 << a.rays_log2
    ro, rd = synth.bounce_rays(V, T, n, synth.RAY_SEED_16M)
    d_o, d_d = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    sorted_rays = {}

    def presorted(bits):  # sorted=B: rays pre-sorted on the host by B bits per axis (1 = the full 10)
        bits = 10 if bits == 1 else bits
        if bits not in sorted_rays:
            order = np.argsort(morton_sort_key(ro, rd, scene.bbox(), bits), kind="stable")
            sorted_rays[bits] = (torch.from_numpy(ro[order]).to(dev), torch.from_numpy(rd[order]).to(dev))
        return sorted_rays[bits]
    hits = torch.empty((n, 4), dtype=torch.int32, device=dev)
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    sh = torch.cuda.current_stream().cuda_stream
    defaults = {"stack_lds": 8, "refill_thresh": 0, "any_atomic": -1, "blocks_per_cu": 0, "chunk": 0, "t_exit": 20, "sort_rays": 0, "sort_bits": 6, "guided": -1, "xchg_every": 1, "xchg_min": 1}
    res = {v: [] for v in a.variants}
    ref_hits = None
    for r in range(a.rounds + 1):
        for v in a.variants:
            opts = dict(defaults)
            pseudo = {"sorted": 0, "any": 0, "stats": 0, "layout": 2, "leaf_blocks": -1}
            for kv in v.split(","):
                if not kv or kv == "base":
                    continue
                k, val = kv.split("=")
                (pseudo if k in pseudo else opts)[k] = int(val)
            for k, val in opts.items():
                try:
                    rt.set_option(k, val)
                except rt.GiglError:
                    if k not in defaults:   # an older library under GIGL_RT_LIB may lack a default knob
                        raise
            rt.set_option("stats", pseudo["stats"])
            o, d = presorted(pseudo["sorted"]) if pseudo["sorted"] else (d_o, d_d)
            scene = scene_for(pseudo["layout"], pseudo["leaf_blocks"])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # the first of the two launches is untimed: a variant with a different grid makes the
            # stream-ordered pool allocate a new workspace block, which costs ~0.5 ms once
            for rep in range(2):
                if rep == 1:
                    e0.record()
                if pseudo["any"]:
                    scene.trace_any_device(o.data_ptr(), d.data_ptr(), n, bits.data_ptr(), sh)
                else:
                    scene.trace_closest_device(o.data_ptr(), d.data_ptr(), n, hits.data_ptr(), sh)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                res[v].append(e0.elapsed_time(e1))
            if pseudo["stats"] and r == a.rounds:
                st = rt.debug_last_stats()
                print(v, st, flush=True)
                print("   T-phase lane utilisation %.3f, L-phase %.3f, lanes holding a ray %.3f, T iters/round %.2f" % (
                    st["t_lane_steps"] / max(1, 64 * st["t_iters"]), st["l_lane_rounds"] / max(1, 64 * st["l_rounds"]),
                    st["ray_lane_rounds"] / max(1, 64 * st["rounds"]), st["t_iters"] / max(1, st["rounds"])), flush=True)
                tl = rt.debug_last_timeline()
                last = int(np.nonzero(tl[1])[0].max()) + 1 if tl[1].any() else 0
                print("   timeline of every 16th wave, 81.92 us buckets from the first wave's start: Mrays started (x16) | rounds | mean lanes holding a ray at a round", flush=True)
                step = max(1, last // 48)
                for b in range(0, last, step):
                    started = tl[0, b:b + step].sum(); rounds = tl[1, b:b + step].sum(); lanes = tl[2, b:b + step].sum()
                    print("   %7.2f ms  %8.3f  %8d  %5.1f  %s" % (b * 0.08192, 16 * started / 1e6, rounds, lanes / max(1, rounds),
                                                               "#" * int(round(40.0 * started / max(1, tl[0].max() * step)))), flush=True)
            if not pseudo["any"] and not pseudo["sorted"]:
                h = hits.cpu().numpy()
                if ref_hits is None:
                    ref_hits = h.copy()
                elif not np.array_equal(ref_hits, h):
                    print("!! variant %s changed the results" % v, flush=True)
    for v in a.variants:
        t = np.array(res[v])
        print("%-60s median %8.3f ms  min %8.3f ms  -> %8.1f Mrays/s" % (v, np.median(t), t.min(), n / np.median(t) / 1e3),
              flush=True)
    for sc in scenes.values():
        sc.close()


if __name__ == "__main__":
    main()
