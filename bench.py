#!/usr/bin/env python3
"""bench.py — throughput of the ray-query hot path on MI355X.

--config bounce (default; the driver's run)
  N = 1: BASELINE.json configs[2], the configuration the 1 Grays/s target is quoted on: the synthetic
  tessellated Cornell scene S(158) = 998,560 triangles and 2^24 incoherent cosine-weighted
  diffuse-bounce rays, traced closest-hit through the reference's k-d tree semantics.  A "step" is
  one closest-hit pass over the whole ray batch, rays and tree resident in HBM.
  N > 1 (one process per GPU, launched by torch.distributed.run): BASELINE.json configs[3]: S(500) =
  10,000,000 triangles, 2^27 rays in all (RAY_SEED_128M), split into contiguous ranges
  (gigl_amd.shard.shard_range), tree replicated, no data-path collective; the 16-byte hit records are
  gathered to rank 0 over RCCL behind the next step's traversal (gigl_amd.shard.ShardedTracer).
  Total work is fixed as N grows: "scaling": "strong".  `--gpus 1 --tess 500 --total-rays-log2 27`
  runs the N = 1 point of that series.
--config cornell_shadow   BASELINE.json configs[1]: Cornell box, 1024^2 primary closest-hit rays (fused
  generator) + one any-hit shadow ray per (hit pixel, VPL) for the 150 golden VPLs (~152 M pairs).
--config transmittance    BASELINE.json configs[4]: calcTransm (8-interval ray march through the 64^3
  density volume) of the same ~152 M shadow segments.
--config vpl              BASELINE.json configs[0]: Cornell box, generation of the 150-VPL set
  (rt::PhotonTracer::trace) as ray batches (gigl_photon_trace_batch, substream mode), beside the oracle's
  serial walk in the same mode; --vpls N changes the set size.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0   # same guide: aggregate L2 bandwidth, ~34.5 TB/s
# Vector-ALU issue on gfx950, measured (tools/micro/valu_issue.hip, profiles/r03_valu_issue.txt): a SIMD issues a
# wave64 instruction of the plain class (fp32 fma/mul/add, integer add, logic, move, right shift) every 2 cycles and
# every other one (compare, min/max, select on an SGPR mask, three-operand integer, packed, 64-bit, convert) every 4;
# transcendentals 8.  So a kernel's peak depends on its mix: SIMD cycles per second / its mean cost per instruction.
SIMD_GCYCLES = 256 * 4 * 2.4            # G SIMD cycles/s: 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_GINST = SIMD_GCYCLES / 4.0    # G wave-instructions/s if every instruction were of the 4-cycle class
VALU_PEAK_FAST_GINST = SIMD_GCYCLES / 2.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="bounce", choices=["bounce", "cornell_shadow", "transmittance", "vpl"])
    ap.add_argument("--vpls", type=int, default=None, help="--config vpl: VPLs per set (default MAX_N_VPLS = 150)")
    ap.add_argument("--tess", type=int, default=None,
                    help="k of scene S(k); default 158 (998,560 triangles) at N = 1, 500 (10 M) at N > 1")
    ap.add_argument("--rays-log2", type=int, default=None, help="rays per GPU = 2^this (weak scaling)")
    ap.add_argument("--total-rays-log2", type=int, default=None,
                    help="rays over all GPUs = 2^this, contiguous shards (strong scaling; default 27 at N > 1)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL gather of hit records")
    ap.add_argument("--rebuild-per-rank", action="store_true",
                    help="N > 1: every rank builds the tree itself instead of adopting rank 0's from a broadcast")
    ap.add_argument("--pipeline", type=int, default=-1,
                    help="1: consecutive steps traverse on two alternating streams; 0: one stream; -1: on at N > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time any-hit and the Cornell passes (N=1)")
    ap.add_argument("--sample-parity", dest="full_parity", action="store_false",
                    help="check only the CPU-timed sample against the oracle instead of the whole batch")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (gigl_set_option)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="N = 1 default workload: do not run the other BASELINE.json configurations after the timed "
                         "loop (any-hit on the same batch, configs[1], configs[4], configs[0]; config.other_configs)")
    return ap.parse_args()


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher (WORLD_SIZE unset): N fresh child processes of this script, one per
    rank, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them.  Called
    before anything has touched the GPU; the current process is never replaced, it waits for the children,
    passes rank 0's line through (the children inherit stdout) and exits with the first non-zero code, after
    ending the ranks that are still running."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GIGL_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
    for p in live:   # a rank failed: the others would wait for it in a collective
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def host_info(O=None):
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {"cpu_model": model, "os_cpu_count": os.cpu_count()}
    if O is not None:
        info["omp_get_max_threads"] = O.max_threads()
    return info


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


KERNEL_SOURCES = ["gigl_amd/csrc/gigl_rt.hip", "gigl_amd/csrc/gigl_trace_body.h", "gigl_amd/csrc/gigl_internal.h",
                  "gigl_amd/csrc/kd_layout.cpp", "gigl_amd/csrc/kd_layout.h"]


def kernel_sources_hash():
    """sha256 over the sources that decide what the traversal kernel executes and reads: the counter constants
    in profiles/traffic.json are quoted only while this equals the hash stored with them (tools/traffic_hash.py)."""
    import hashlib
    h = hashlib.sha256()
    try:
        for rel in KERNEL_SOURCES:
            data = open(os.path.join(ROOT, rel), "rb").read()
            if rel.endswith("gigl_rt.hip"):   # the device half only: the C-ABI host code below the marker runs no kernel code
                cut = data.find(b"\n// host side\n")
                if cut < 0:
                    return None
                data = data[:cut]
            h.update(data)
    except OSError:
        return None
    return h.hexdigest()[:16]


def traffic_json():
    """profiles/traffic.json, or None when it is missing or was taken on other kernel sources than this tree's."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None, "profiles/traffic.json missing or unreadable"
    want, have = tj.get("kernel_sources_sha256_16"), kernel_sources_hash()
    if want is None or have is None or want != have:
        return None, ("kernel sources changed since the counter passes in profiles/traffic.json were taken "
                      "(stored %s, this tree %s): not quoted" % (want, have))
    return tj, None


def traffic_from_profiles(key, layout, n_rays=1 << 24):
    """HBM-side bytes per launch from the committed PMC passes (profiles/traffic.json), with where the
    number comes from: bench.py cannot collect counters itself, so this is NOT measured by this run."""
    tj, why = traffic_json()
    if tj is None:
        return None, {"file": "profiles/traffic.json", "note": why}
    ent = tj.get(key) if isinstance(tj.get(key), dict) else None
    if ent is None and key == "bounce" and "trace_closest_hbm_bytes_per_launch" in tj:   # round-1 file
        ent = {"hbm_bytes_per_launch": tj["trace_closest_hbm_bytes_per_launch"], "layout": tj.get("layout"),
               "source": tj.get("source", "profiles/r01_v8_pmc.json"), "commit": tj.get("commit")}
    if ent is None or (ent.get("layout") is not None and ent.get("layout") != layout):
        return None, None
    src = {"file": "profiles/traffic.json", "from": ent.get("source"), "commit": ent.get("commit"),
           "kernel_sources_sha256_16": tj.get("kernel_sources_sha256_16"),
           "fetch_size_bytes": ent.get("fetch_size_bytes"), "tcc_miss_per_ray": ent.get("tcc_miss_per_ray"),
           "note": "constant from a committed rocprofv3 --pmc pass of the same command, not measured by this run"}
    b = ent.get("hbm_bytes_per_launch")
    if b is not None and n_rays != (1 << 24):   # the passes were taken on 2^24-ray launches
        b = b * n_rays / float(1 << 24)
        src["note"] += "; scaled per ray from a 2^24-ray launch to %d rays" % n_rays
    return b, src


def valu_roofline(key, units, k_ms):
    """Vector-ALU issue roofline of a kernel that is bound by it: wave instructions per unit from a committed
    counter pass (profiles/traffic.json) x the units of this run / the kernel time of this run."""
    try:
        tj, _ = traffic_json()
        ent = tj.get(key) if tj else None
        if not ent:
            return None
        ginst = ent["valu_insts_per_pair"] * units / (k_ms * 1e-3) / 1e9
        mix = ent.get("valu_static_mix") or {}
        cyc = mix.get("mean_cycles", 4.0)
        peak = SIMD_GCYCLES / cyc
        return {"achieved": round(ginst, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                "frac": round(ginst / peak, 4), "insts_per_unit": round(ent["valu_insts_per_pair"], 2),
                "mean_cycles_per_inst": cyc, "peak_2_cycle_class": VALU_PEAK_FAST_GINST, "peak_4_cycle_class": VALU_PEAK_GINST,
                "mix": mix.get("by_cycles"), "source": ent.get("source"), "commit": ent.get("commit"),
                "note": "instruction count from a committed rocprofv3 --pmc pass, not measured by this run; the peak is "
                        "2,457.6 G SIMD cycles/s / the mean issue cost of the kernel's static instruction mix "
                        "(tools/valu_mix.py; costs measured by tools/micro/valu_issue.hip), so `frac` estimates the "
                        "share of issue cycles at 2.4 GHz that the vector ALU is busy"}
    except Exception:
        return None


LINE_RATE_PEAK_G = 56.4   # L2-missing 128-byte lines per second one MI355X serves (profiles/r01_gather_lines.txt)


def line_rate_info(traffic, traffic_src, n_rays, k_ms):
    """The memory-side picture of a launch from the committed counter pass: bytes per second that cross the L2
    (reads as 128-byte lines + writes) and the missing lines per second against the measured ceiling."""
    if not traffic or not traffic_src or not traffic_src.get("tcc_miss_per_ray"):
        return None
    lines = traffic_src["tcc_miss_per_ray"] * n_rays / (k_ms * 1e-3) / 1e9
    return {"gb_per_s": round(traffic / (k_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "l2_miss_g_lines_per_s": round(lines, 2), "line_rate_ceiling": LINE_RATE_PEAK_G,
            "frac_of_line_rate": round(lines / LINE_RATE_PEAK_G, 4),
            "note": "counter-pass constants x this run's kernel time; the ceiling is tools/micro/gather_lines.hip's "
                    "(random 128-byte lines, from the Infinity Cache and from HBM alike)"}


def kernel_name(layout, closest=True, gen="explicit rays"):
    return "trace_kernel<%s, %s, %s>" % ("triples" if layout == 2 else "pairs", "closest" if closest else "any", gen)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))   # no launcher: this process becomes the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # never a silent single-GPU number under an N-GPU label, or the reverse
        raise SystemExit("WORLD_SIZE %d != --gpus %d: launch with `python -m torch.distributed.run --nproc-per-node %d "
                         "bench.py --gpus %d`, or plain `python bench.py --gpus %d` (it starts its own ranks)"
                         % (world, args.gpus, args.gpus, args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    from gigl_amd import rt, scene_io as S, shard, synth

    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("GIGL_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1 and args.rebuild_per_rank:   # the ranks of a node share its cores: each builds on its share
        rt.set_option("build_threads", max(1, min(16, (os.cpu_count() or 1) // world)))
    for kv in args.opt:
        k, v = kv.split("=")
        rt.set_option(k, int(v))
    if args.config != "bounce":
        if world > 1:
            raise SystemExit("--config %s is a single-GPU configuration" % args.config)
        out = (vpl_config if args.config == "vpl" else cornell_config)(args, torch, rt, S, device)
        print(json.dumps(out), flush=True)
        return

    # ---- workload ----
    strong = args.total_rays_log2 is not None or (world > 1 and args.rays_log2 is None)
    tess = args.tess if args.tess is not None else (158 if world == 1 and not strong else 500)
    if strong:
        n_total = 1 << (args.total_rays_log2 if args.total_rays_log2 is not None else 27)
        seed = synth.RAY_SEED_128M
    else:
        n_total = (1 << (args.rays_log2 if args.rays_log2 is not None else 24)) * world
        seed = synth.RAY_SEED_16M if world == 1 else synth.RAY_SEED_128M
    lo, hi = shard.shard_range(n_total, world, rank)
    n = hi - lo

    # ---- inputs (not timed): scene, tree, this rank's rays ----
    t0 = time.time()
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, tess)
    # N > 1: rank 0 builds and flattens the tree once, the other ranks adopt its device arrays from a broadcast
    # (§8(e): replicated tree, no rebuild per rank)
    replication = None
    if world > 1 and not args.rebuild_per_rank:
        scene = rt.Scene(V, T, device=dev_index, **S.KD_PARAMS) if rank == 0 else None
        scene, replication = shard.replicate_scene(torch, dist, rt, scene, rank, device, backend=backend)
        if scene is None:   # rank 0 could not offer its arrays: this rank builds its own tree after all
            rt.set_option("build_threads", max(1, min(16, (os.cpu_count() or 1) // world)))
            scene = rt.Scene(V, T, device=dev_index, **S.KD_PARAMS)
    else:
        scene = rt.Scene(V, T, device=dev_index, **S.KD_PARAMS)
    st = scene.stats()
    layout = scene.layout()
    gen_threads = max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    ro, rd = synth.bounce_rays(V, T, n, seed, first=lo, n_threads=gen_threads)
    d_o = torch.from_numpy(ro).to(device)
    d_d = torch.from_numpy(rd).to(device)
    setup_s = time.time() - t0
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    def trace_fn(buf):   # on torch's current stream: the tracer decides which one that is
        scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)

    # N > 1: consecutive steps traverse on two alternating streams (the drain of one launch is filled by the
    # next one's first waves), as their gathers already overlap the next traversal.  N = 1 keeps one stream:
    # the headline number is one launch after the other.
    pipeline = (world > 1) if args.pipeline < 0 else bool(args.pipeline)
    tracer = shard.ShardedTracer(torch, dist if world > 1 else None, world, rank, n_total, device, trace_fn,
                                 gather=not args.no_gather, backend=backend, pipeline=pipeline)

    def barrier():
        tracer.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tracer.prime()   # N > 1: RCCL sets up its communicator on first use; not a step, and not timed
    for _ in range(args.warmup):
        tracer.step()
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for i in range(args.steps):
        tracer.step(events=(ev0[i], ev1[i]))
    barrier()
    elapsed = time.perf_counter() - t_start
    kern_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_total * args.steps / elapsed / 1e6

    # ---- N = 1, --extras, outside the timed region: the same steps with consecutive launches on two alternating streams
    # (what a caller with a stream of batches gets; reported beside the headline, never as it) ----
    piped = None
    if world == 1 and not tracer.pipeline and args.extras:   # not by default: its overlapping launches would blur a profile of this command
        t2 = shard.ShardedTracer(torch, None, 1, 0, n_total, device, trace_fn, gather=False, pipeline=True)
        for _ in range(2):
            t2.step()
        t2.drain()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(args.steps):
            t2.step()
        t2.drain()
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        piped = {"value": round(n_total * args.steps / tp / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(tp * 1e3 / args.steps, 4),
                 "what": "the same %d steps with consecutive launches on two alternating streams: the drain of one "
                         "launch overlaps the start of the next" % args.steps}
        del t2

    # ---- N > 1, not timed: the gathered records of every rank's range against a trace of the same rays
    # on rank 0's own GPU (first 2^16 rays of each range), and rank 0's rate on its shard with the other
    # GPUs idle (the same scene on one GPU: what an efficiency has to be read against) ----
    gather_check = None
    alone = None
    if world > 1:
        full = tracer.result()
        if rank == 0 and full is not None:
            bad, checked = 0, 0
            for r, (rlo, rhi) in enumerate(tracer.ranges):
                m = min(1 << 16, rhi - rlo)
                xo, xd = synth.bounce_rays(V, T, m, seed, first=rlo)
                t_o, t_d = torch.from_numpy(xo).to(device), torch.from_numpy(xd).to(device)
                ref = torch.empty((m, 4), dtype=torch.int32, device=device)
                scene.trace_closest_device(t_o.data_ptr(), t_d.data_ptr(), m, ref.data_ptr(), sh)
                torch.cuda.synchronize()
                bad += int((ref.cpu() != full[rlo:rlo + m].cpu()).any(1).sum())
                checked += m
            gather_check = {"rays_checked": checked, "records_differing": bad,
                            "how": "first 2^16 rays of every rank's range re-traced on rank 0's GPU"}
        del full
        dist.barrier()
        if rank == 0:
            buf = tracer.bufs[0]
            trace_fn(buf)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(3):
                trace_fn(buf)
            b.record(stream)
            torch.cuda.synchronize()
            alone = {"mrays_s": round(n * 3 / a.elapsed_time(b) / 1e3, 2), "rays": int(n),
                     "what": "rank 0 tracing its own shard of the same scene while the other GPUs idle"}
        dist.barrier()

    out = None
    if rank == 0:
        hits = tracer.local_hits().cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
        # size-independent sanity at full size: every reported hit lies inside its ray's range
        hit_mask = hits["tri_id"] != rt.GIGL_MISS
        assert (hits["tri_id"][hit_mask] < len(T)).all() and (hits["t"][hit_mask] > 0).all()
        cpu = parity = bytes_per_ray = mean_nodes = mean_tris = ora = None
        counters_from = None
        hinfo = host_info()
        if not args.no_cpu_baseline:
            from oracle import oracle_py as O
            hinfo = host_info(O)
            # S(158): the oracle builds its own tree (15-20 s serial); S(500): it traverses the engine's
            # exported tree (the serial build would take minutes)
            if tess <= 200:
                ora = O.OracleScene(V, T, **S.KD_PARAMS)
                tree_src = "the oracle's own serial build"
            else:
                ora = O.OracleScene(V, T, tree=scene.export_tree())
                tree_src = "the engine's exported tree (oracle build skipped at this size)"
            cores = min(16, os.cpu_count() or 1)
            n1 = min(n, 1 << 18)
            t1 = time.perf_counter()
            ora.trace_closest(ro[:n1], rd[:n1], n_threads=1)
            cpu1 = n1 / (time.perf_counter() - t1) / 1e6
            nm = min(n, 1 << 21)
            t1 = time.perf_counter()
            want, cnt = ora.trace_closest(ro[:nm], rd[:nm], counters=True, n_threads=cores)
            cpum = nm / (time.perf_counter() - t1) / 1e6
            # parity (not timed): the batch against the oracle, in slices; the timed sample is its first slice.
            # At N > 1 only rank 0's shard is checked here (the gather check above covers the other ranks).
            full_n = n if (args.full_parity and world == 1 and n <= (1 << 27)) else nm
            id_mis, rel_max, checked = 0, {"t": 0.0, "u": 0.0, "v": 0.0}, 0
            for slo in range(0, full_n, nm):
                shi = min(full_n, slo + nm)
                w = want if slo == 0 else ora.trace_closest(ro[slo:shi], rd[slo:shi], n_threads=cores)
                got = hits[slo:shi]
                same = got["tri_id"] == w["tri_id"]
                id_mis += int((~same).sum())
                for k in rel_max:
                    a, b = got[k][same].astype(np.float64), w[k][same].astype(np.float64)
                    rel_max[k] = max(rel_max[k], float((np.abs(a - b) / np.maximum(np.abs(b), 1e-30)).max(initial=0.0)))
                checked += shi - slo
            parity = {"rays_checked": int(checked), "id_mismatches": id_mis, "max_rel_err_t": rel_max["t"],
                      "max_rel_err_u": rel_max["u"], "max_rel_err_v": rel_max["v"], "oracle_tree": tree_src}
            mean_nodes, mean_tris = float(cnt[:, 0].mean()), float(cnt[:, 1].mean())
            counters_from = "oracle counters on the first 2^%d rays of this run" % int(np.log2(nm))
            cpu = {"value": round(cpum, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": "first 2^%d rays of the same batch, OpenMP schedule(dynamic) over rays on %d threads; "
                             "single-thread on the first 2^%d rays: %.4f Mrays/s"
                             % (int(np.log2(nm)), cores, int(np.log2(n1)), cpu1),
                   "single_thread_mrays_s": round(cpu1, 4), "threads_used": cores,
                   "omp_get_max_threads": hinfo.get("omp_get_max_threads"), "cpu_model": hinfo["cpu_model"]}
        default_workload = tess == 158 and n == (1 << 24) and world == 1
        if mean_nodes is None and default_workload:
            mean_nodes, mean_tris = 61.084, 7.717
            counters_from = "profiles/r01_bench_n1.log (oracle counters of the same workload; the CPU leg was skipped)"
        if mean_nodes is not None:
            # 32 B ray in + 16 B hit out + 8 B per node visited + 56 B per triangle test
            bytes_per_ray = 32.0 + 16.0 + 8.0 * mean_nodes + 56.0 * mean_tris
        k_ms = float(np.mean(kern_ms))
        # per GPU: one launch over this rank's rays.  Pipelined steps overlap on two streams, so an event pair
        # around one launch also spans part of its neighbours: the per-launch time is then the step time.
        k_eff_ms = ms_per_step if tracer.pipeline else k_ms
        achieved = bytes_per_ray * n / (k_eff_ms * 1e-3) / 1e9 if bytes_per_ray is not None else None
        traffic, traffic_src = traffic_from_profiles("bounce" if tess == 158 else "bounce_s%d" % tess, layout, n)
        per_cu, n_cus = scene.launch_info()
        out = {
            "metric": "Mrays/s (incoherent closest-hit)", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S(%d) tessellated Cornell scene, %d triangles, k-d tree %d nodes depth %d; "
                                   "%d incoherent diffuse-bounce closest-hit rays %s%s"
                                   % (tess, len(T), st.n_nodes, st.actual_depth, n_total,
                                      "in all, contiguous shards" if world > 1 else "on one GPU",
                                      ("; hit records gathered to rank 0 over %s behind the next step's traversal" % backend)
                                      if tracer.gather else ""),
                       "baseline_config": "configs[3]" if tess == 500 else ("configs[2]" if default_workload else None),
                       "triangles": int(len(T)), "rays_total": int(n_total), "rays_rank0": int(n),
                       "tree_build_ms": round(st.build_ms, 1), "tree_device_bytes": int(st.device_bytes),
                       "device_layout": layout, "setup_s": round(setup_s, 1),
                       "tree_replication": replication,
                       "same_scene_one_gpu": alone, "gather_check": gather_check,
                       "pipelined_two_streams": piped, "commit": git_head()},
            "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src,
                         # what the memory system actually moves: the committed counter pass's bytes at this run's
                         # kernel time, and its L2-missing lines against the chip's measured random-line rate
                         "traffic_rate": line_rate_info(traffic, traffic_src, n, k_eff_ms),
                         "valu": valu_roofline("bounce" if tess == 158 else "bounce_s%d" % tess, n, k_eff_ms),
                         "kernel": kernel_name(layout), "kernel_ms": round(k_eff_ms, 4),
                         "pipelined": tracer.pipeline,
                         "kernel_ms_events": round(k_ms, 4),
                         "waves_per_simd": per_cu, "workgroups": per_cu * n_cus,
                         "algorithmic_bytes_per_ray": None if bytes_per_ray is None else round(bytes_per_ray, 2),
                         "mean_nodes_visited": mean_nodes, "mean_triangle_tests": mean_tris,
                         "counters_source": counters_from},
            "cpu_baseline": cpu,
            "parity": parity,
            "host": hinfo,
        }
        if args.extras and world == 1:
            out["extras"] = extras(torch, rt, S, scene, d_o, d_d, n, device)
        if args.other_configs and world == 1:
            out["config"]["other_configs"] = other_configs(args, torch, rt, S, device, scene, ora, d_o, d_d, ro, rd, n,
                                                           layout, hits)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()


def cornell_inputs(torch, rt, S, device):
    """Cornell box scene, camera, the golden 150-VPL set's positions and the fog volume (needs the oracle
    only for the density field, which is an INPUT: the reference's df.3dt payload)."""
    from oracle import oracle_py as O
    cv, ct, mats, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    cs = rt.Scene(cv, ct, device=device.index, **S.KD_PARAMS)
    cam = rt.camera_setup(S.CAM_POS, S.CAM_DIR, S.CAM_UP, S.CAM_HFOV, 1024, 1024)
    vp = np.load(os.path.join(ROOT, "tests", "golden", "vpl_set.npz"))["vpls"]
    lights = np.ascontiguousarray(vp.view(np.float32).reshape(len(vp), -1)[:, :3]).astype(np.float32)
    dens = O.make_density((64, 64, 64), 12.0, 1.0)
    box = cs.bbox().copy()
    box[4] = min(box[4], np.float32(S.MAX_FOG_HEIGHT))
    fog = rt.FogVolume(dens, box, S.ABS_K, S.SCA_K)
    return dict(verts=cv, tris=ct, mats=mats, scene=cs, cam=cam, vpls=vp, lights=lights, dens=dens, box=box, fog=fog)


def cornell_config(args, torch, rt, S, device, ci=None):
    """configs[1] (cornell_shadow) and configs[4] (transmittance) as driver-runnable lines.  ci: inputs shared
    with the caller's other Cornell configurations (then they stay open)."""
    from oracle import oracle_py as O
    t0 = time.time()
    own_ci = ci is None
    if own_ci:
        ci = cornell_inputs(torch, rt, S, device)
    cs, cam, lights, fog = ci["scene"], ci["cam"], ci["lights"], ci["fog"]
    nl = len(lights)
    npix = 1024 * 1024
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    po = torch.empty((npix, 4), dtype=torch.float32, device=device)
    pd = torch.empty((npix, 4), dtype=torch.float32, device=device)
    ph = torch.empty((npix, 4), dtype=torch.int32, device=device)
    dl = torch.from_numpy(lights).to(device)
    npair = npix * nl
    bits = torch.zeros((npair + 63) // 64, dtype=torch.int64, device=device)
    tr = torch.empty(npair, dtype=torch.float32, device=device) if args.config == "transmittance" else None
    cs.gen_primary_rays_device(cam, 1024, 0, npix, po.data_ptr(), pd.data_ptr(), sh)
    cs.trace_primary_device(cam, 1024, 0, npix, ph.data_ptr(), sh)
    torch.cuda.synchronize()
    nhit = int((ph[:, 0] != -1).sum().item())
    layout = cs.layout()
    setup_s = time.time() - t0
    shadow = args.config == "cornell_shadow"
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(e=None):
        if shadow:   # the whole of configs[1]: primary closest-hit, then the shadow pairs
            cs.trace_primary_device(cam, 1024, 0, npix, ph.data_ptr(), sh)
        if e:
            e[0].record(stream)
        if shadow:
            cs.shadow_batch_device(po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, dl.data_ptr(), nl, bits.data_ptr(),
                                   stream=sh)
        else:
            cs.shadow_batch_device(po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, dl.data_ptr(), nl, None, fog=fog,
                                   transm_ptr=tr.data_ptr(), stream=sh)
        if e:
            e[1].record(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    k_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))   # the dominant kernel alone
    ms_per_step = elapsed * 1e3 / args.steps
    units = (npix + nhit * nl) if shadow else nhit * nl
    value = units * args.steps / elapsed / 1e6

    # ---- oracle on a bounded sample of pixels (every 64th: 16,384 pixels x 150 lights), same inputs:
    # parity, the exact work counters the algorithmic bytes are defined on, and the CPU rate ----
    hinfo = host_info(O)
    cpu = parity = None
    sums = None
    sel = np.arange(0, npix, 64)
    if not args.no_cpu_baseline:
        ora = O.OracleScene(ci["verts"], ci["tris"], **S.KD_PARAMS)
        cores = min(16, os.cpu_count() or 1)
        so, sd = O.primary_rays(cam, 1024, 0, npix)
        so, sd = so[sel], sd[sel]
        hh = ph.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)[sel]
        fog_o = O.Fog(ci["dens"], ci["box"], fog.ext_k)
        t1 = time.perf_counter()
        if shadow:
            w_hits = ora.trace_closest(so, sd, n_threads=cores)
            occ, _, sums = ora.shadow_batch_fused(so, sd, w_hits, lights, fog=None, n_threads=cores)
            cpu_units = len(sel) + int(sums[0])
        else:
            _, w_tr = ora.shadow_batch(so, sd, hh, lights, fog=fog_o, want_transm=True, n_threads=cores)
            cpu_units = int((hh["tri_id"] != rt.GIGL_MISS).sum()) * nl
        cpu_s = time.perf_counter() - t1
        idx = (sel[:, None].astype(np.int64) * nl + np.arange(nl)[None, :]).reshape(-1)
        if shadow:
            got_bits = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), npair)[idx].reshape(len(sel), nl)
            want01 = np.where(occ == 0xFF, 0, occ)
            parity = {"pairs_checked": int(want01.size), "bit_mismatches": int((got_bits != want01).sum()),
                      "primary_id_mismatches": int((hh["tri_id"] != w_hits["tri_id"]).sum())}
        else:
            got_tr = tr.cpu().numpy()[idx].reshape(len(sel), nl)
            rel = np.abs(got_tr.astype(np.float64) - w_tr) / np.maximum(np.abs(w_tr), 1e-30)
            parity = {"pairs_checked": int(w_tr.size), "max_rel_err": float(rel.max()), "tolerance": 1e-5}
        cpu = {"value": round(cpu_units / cpu_s / 1e6, 4), "unit": "Mrays/s" if shadow else "Msegments/s",
               "cores": cores, "kind": "port", "threads_used": cores,
               "sample": "every 64th pixel of the same image (16,384 pixels) x %d lights, OpenMP over pixels on %d "
                         "threads, %.1f s" % (nl, cores, cpu_s),
               "omp_get_max_threads": hinfo.get("omp_get_max_threads"), "cpu_model": hinfo["cpu_model"]}
    per_cu, n_cus = cs.launch_info()
    if shadow:
        # SURVEY.md §8(d) on the reference's layouts, for the kernel that GENERATES its rays: nothing is read
        # per ray; per pair 1/8 B out + 8 B per node visited + 56 B per triangle test (exact oracle counters on
        # the sample), plus the primary ray and hit record of the point (48 B) once per n_lights pairs.
        if sums is not None:
            nodes_pp, tests_pp = float(sums[1]) / float(sums[0]), float(sums[2]) / float(sums[0])
            bpp = 0.125 + 8.0 * nodes_pp + 56.0 * tests_pp + 48.0 / nl
            achieved = bpp * nhit * nl / (k_ms * 1e-3) / 1e9
        else:
            nodes_pp = tests_pp = bpp = achieved = None
        # what does bound it: vector ALU issue.  Instructions per pair from the committed counter pass
        # (profiles/traffic.json), issue peak = 256 CUs x 4 SIMDs x one wave instruction per 4 cycles at 2.4 GHz
        valu_roof = valu_roofline("cornell_shadow", nhit * nl, k_ms)
        roof = {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": kernel_name(layout, False, "shadow-pair generator"), "kernel_ms": round(k_ms, 4),
                "waves_per_simd": per_cu, "algorithmic_bytes_per_pair": None if bpp is None else round(bpp, 3),
                "mean_nodes_visited": nodes_pp, "mean_triangle_tests": tests_pp,
                "counters_source": "oracle counters on the sampled pairs of this run",
                "cache_resident": True,
                "valu": valu_roof,
                "hbm_bytes_per_pair": round(0.125 + 48.0 / nl, 4),
                "l2": None if achieved is None else {"achieved": round(achieved, 1), "peak": L2_PEAK_GBS, "unit": "GB/s",
                                                     "frac": round(achieved / L2_PEAK_GBS, 4),
                                                     "what": "the same algorithmic bytes against the aggregate L2 bandwidth"},
                "note": "Cornell's whole structure (151 nodes, 40 triangles: 3.8 KB) is cache-resident: HBM sees only "
                        "the 1/8 B per pair out and the per-point inputs (hbm_bytes_per_pair), so `frac` prices the "
                        "ALGORITHMIC bytes of the reference's layouts (SURVEY.md §8(d)), not DRAM traffic, and can "
                        "exceed 1: the HBM roofline does not bound this configuration; vector ALU issue at the "
                        "lane utilisation the batch allows does (DESIGN.md §4.1, §5)"}
        metric = "Mrays/s (closest-hit + shadow any-hit)"
        wl = ("Cornell box (40 triangles, 151 nodes), 1024^2 primary closest-hit rays (fused generator) + %d any-hit "
              "shadow rays (%d hit pixels x %d golden VPLs, fused generator)" % (nhit * nl, nhit, nl))
    else:
        # HBM bytes are the 4 B out per segment (the generator reads 48 B per POINT); the volume
        # is L2-resident, so the bound that matters is L2: 9 samples x one 8-byte corner brick per segment.
        bpp = 4.0 + 48.0 / nl
        achieved = bpp * nhit * nl / (k_ms * 1e-3) / 1e9
        l2 = 72.0 * nhit * nl / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "kernel": "transm_shadow_kernel", "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes_per_segment": round(bpp, 3),
                "samples_per_s_G": round(9.0 * nhit * nl / (k_ms * 1e-3) / 1e9, 2),
                "valu": valu_roofline("transmittance", nhit * nl, k_ms),
                "l2": {"achieved": round(l2, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2 / L2_PEAK_GBS, 4),
                       "bytes_per_segment": 72, "what": "9 trilinear samples x one 8-byte corner brick (2.2 MB table, L2-resident)"},
                "note": "vector-ALU bound: ~100 instructions per trilinear sample (three IEEE divides of the "
                        "box normalisation, eight byte->unorm conversions, seven lerps with contraction off)"}
        metric = "Msegments/s (shadow-ray transmittance, 8-interval ray march)"
        wl = ("Cornell box, calcTransm (8 intervals, 9 samples, 64^3 u8 density) of %d shadow segments "
              "(%d hit pixels x %d golden VPLs, generated in the kernel)" % (nhit * nl, nhit, nl))
    out = {"metric": metric, "value": round(value, 3), "unit": "Mrays/s" if shadow else "Msegments/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": wl, "baseline_config": "configs[1]" if shadow else "configs[4]",
                      "pairs": int(nhit * nl), "hit_pixels": nhit, "n_lights": nl, "device_layout": layout,
                      "setup_s": round(setup_s, 1), "commit": git_head()},
           "roofline": roof, "cpu_baseline": cpu, "parity": parity, "host": hinfo}
    if args.extras:
        # the two passes and the fused pass (occlusion first, march only the visible pairs) side by side
        tr2 = torch.empty(npair, dtype=torch.float32, device=device)

        def timeit(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        args_ = (po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, dl.data_ptr(), nl)
        out["extras"] = {
            "shadow_any_hit_ms": round(timeit(lambda: cs.shadow_batch_device(*args_, bits.data_ptr(), stream=sh)), 3),
            "transmittance_all_pairs_ms": round(timeit(lambda: cs.shadow_batch_device(*args_, None, fog=fog, transm_ptr=tr2.data_ptr(), stream=sh)), 3),
            "fused_bits_and_visible_transmittance_ms": round(timeit(lambda: cs.shadow_batch_fused_device(*args_, bits.data_ptr(), fog, tr2.data_ptr(), stream=sh)), 3),
            "occluded_fraction": round(float(rt.unpack_bits(bits.cpu().numpy().view(np.uint64), npair).sum()) / max(nhit * nl, 1), 4)}
    if own_ci:
        fog.close()
        cs.close()
    return out


def vpl_config(args, torch, rt, S, device, ci=None):
    """configs[0]: VPL generation for the Cornell box's primary light as ray batches.  A step is one whole set
    (paths walked in blocks: one closest-hit batch per bounce, one lane per path, then the ordered gather)."""
    from oracle import oracle_py as O
    t0 = time.time()
    own_ci = ci is None
    if own_ci:
        ci = cornell_inputs(torch, rt, S, device)
    cs, fog, mats = ci["scene"], ci["fog"], ci["mats"]
    n_vpls = args.vpls or S.MAX_N_VPLS
    call = lambda seed: cs.photon_trace_batch(mats, fog, S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS,
                                              S.PRIM_PL_INTENS, [0, -1, 0], n_vpls, seed)
    setup_s = time.time() - t0
    for i in range(args.warmup):
        call(1000 + i)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    paths = 0
    for i in range(args.steps):   # every step a fresh seed, as every frame of the reference draws fresh numbers
        n, got, n_paths = call(1 + i)
        paths += n_paths
    elapsed = time.perf_counter() - t_start   # the call returns the records to the host: it synchronises itself
    ms_per_step = elapsed * 1e3 / args.steps
    hinfo = host_info(O)
    cpu = parity = None
    if not args.no_cpu_baseline:
        ora = O.OracleScene(ci["verts"], ci["tris"], **S.KD_PARAMS)
        fog_o = O.Fog(ci["dens"], ci["box"], fog.ext_k)
        walk = lambda seed: O.photon_trace_substreams(ora, mats, fog_o, S.ABS_K, S.SCA_K, S.MAJ_EXT_K, 1, S.PRIM_PL_POS,
                                                      S.PRIM_PL_INTENS, [0, -1, 0], n_vpls, seed)
        reps = max(args.steps, 20 if n_vpls <= 1000 else 3)
        per = []
        for i in range(reps):
            t1 = time.perf_counter()
            wn, want, want_paths, _ = walk(1 + (i % args.steps))
            per.append(time.perf_counter() - t1)
        cpu_s = float(np.median(per))   # per set; the median, so that one descheduled repetition does not decide it
        wn, want, want_paths, _ = walk(args.steps)
        want = np.frombuffer(want.tobytes(), np.uint32).reshape(-1, 24)
        parity = {"vpls_checked": int(n), "record_mismatches": int((got != want).any(1).sum()),
                  "n_paths": [int(n_paths), int(want_paths)], "mode": "substreams (parity unpinned against the reference)"}
        cpu = {"value": round(n_vpls / cpu_s / 1e3, 3), "unit": "kVPLs/s", "cores": 1, "kind": "port", "threads_used": 1,
               "sample": "the same sets (same seeds), serial oracle walk ora_photon_trace_substreams, median %.3f ms per set "
                         "(min %.3f, max %.3f over %d sets)" % (cpu_s * 1e3, min(per) * 1e3, max(per) * 1e3, reps),
               "omp_get_max_threads": hinfo.get("omp_get_max_threads"), "cpu_model": hinfo["cpu_model"]}
    out = {"metric": "kVPLs/s (VPL generation as ray batches)", "value": round(n_vpls * args.steps / elapsed / 1e3, 3),
           "unit": "kVPLs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "Cornell box (40 triangles), %d-VPL set from the primary light, fog on: %.0f paths per "
                                  "set, 3 bounces, one closest-hit batch per bounce" % (n_vpls, paths / args.steps),
                      "baseline_config": "configs[0]", "vpls": n_vpls, "paths_per_set": paths / args.steps,
                      "device_layout": cs.layout(), "setup_s": round(setup_s, 1), "commit": git_head()},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "kernel": ("photon_small_kernel (the whole set in one launch: one wave per 1-16 paths walks the tree "
                                   "itself and writes its records to page-locked host memory; no barrier, the last wave to "
                                   "check in raises a flag the call polls; ordered gather and scaling on the host)")
                                  if n_vpls <= 1024 else "photon_start/bounce kernels + trace_kernel per bounce",
                        "note": "latency bound at this size: a set is the dependent chain of its slowest path (three tree "
                                "walks and three path steps, some 8,000 instructions of a lone wave: 55-58 us; the average "
                                "wave takes 20) plus launch and flag latency (profiles/r03_vpl_stamps.txt); no bandwidth "
                                "figure applies"},
           "cpu_baseline": cpu, "parity": parity, "host": hinfo}
    if own_ci:
        fog.close()
        cs.close()
    return out


def any_hit_same_batch(args, torch, rt, scene, ora, d_o, d_d, ro, rd, n, device, layout, closest_hits):
    """The other half of BASELINE.json's metric on the bench batch: the same rays as any-hit (occlusion) rays,
    KdTree.hpp:328.  Timed like the headline (K launches on one stream, HIP events around each), checked against the
    oracle's any-hit walk on the CPU-timed sample and — every ray — against the closest-hit records of the timed run
    (a ray is occluded exactly when it has a closest hit: both walks accept the same first leaf)."""
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=device)
    call = lambda: scene.trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n, bits.data_ptr(), sh)
    for _ in range(max(args.warmup, 1)):
        call()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:   # the events bracket the bit array's clearing too: it is part of the call
        a.record(stream)
        call()
        b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    got = rt.unpack_bits(bits.cpu().numpy().view(np.uint64), n)
    parity = {"rays_checked_against_closest_hit_records": int(n),
              "bit_mismatches_vs_closest_hit": int((got.astype(bool) != (closest_hits["tri_id"] != rt.GIGL_MISS)).sum())}
    cpu = roof = None
    if ora is not None:
        cores = min(16, os.cpu_count() or 1)
        nm = min(n, 1 << 21)
        t1 = time.perf_counter()
        want, cnt = ora.trace_any(ro[:nm], rd[:nm], counters=True, n_threads=cores)
        cpu_s = time.perf_counter() - t1
        parity["rays_checked_against_oracle"] = int(nm)
        parity["bit_mismatches"] = int((got[:nm].astype(bool) != want.astype(bool)).sum())
        nodes, tests = float(cnt[:, 0].mean()), float(cnt[:, 1].mean())
        bpr = 32.0 + 0.125 + 8.0 * nodes + 56.0 * tests   # ray in, one bit out, nodes, triangle tests
        ach = bpr * n / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "kernel": kernel_name(layout, False),
                "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_ray": round(bpr, 2), "mean_nodes_visited": nodes,
                "mean_triangle_tests": tests, "counters_source": "oracle any-hit counters on the first 2^%d rays" % int(np.log2(nm))}
        cpu = {"value": round(nm / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": "first 2^%d rays of the same batch as any-hit rays, OpenMP over rays on %d threads" % (int(np.log2(nm)), cores)}
    return {"metric": "Mrays/s (incoherent shadow any-hit)", "value": round(n * args.steps / elapsed / 1e6, 3), "unit": "Mrays/s",
            "steps": args.steps, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "config": {"workload": "the bench batch (%d rays) traced as any-hit rays, one bit per ray" % n},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity}


def compact(j):
    """An embedded configuration's line without what the enclosing line already says."""
    drop = {"n_gpus", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "host"}
    out = {k: v for k, v in j.items() if k not in drop}
    if isinstance(out.get("cpu_baseline"), dict):
        out["cpu_baseline"] = {k: v for k, v in out["cpu_baseline"].items() if k not in ("omp_get_max_threads", "cpu_model", "threads_used")}
    if isinstance(out.get("config"), dict):
        out["config"] = {k: v for k, v in out["config"].items() if k not in ("commit", "setup_s", "device_layout")}
    return out


def other_configs(args, torch, rt, S, device, scene, ora, d_o, d_d, ro, rd, n, layout, closest_hits):
    """Outside the timed loop of the default line: the rest of BASELINE.json's metric and configurations, each with
    its own rate, kernel time, bound, cpu_baseline on a bounded sample and parity — so that the one line the driver
    runs carries the whole metric (closest-hit AND shadow any-hit) and every configuration."""
    res = {}
    t0 = time.time()
    res["any_hit_same_batch"] = any_hit_same_batch(args, torch, rt, scene, ora, d_o, d_d, ro, rd, n, device, layout, closest_hits)
    ci = cornell_inputs(torch, rt, S, device)
    sub = argparse.Namespace(steps=args.steps, warmup=max(args.warmup, 1), no_cpu_baseline=args.no_cpu_baseline,
                             extras=False, vpls=None, config=None)
    for name, key in (("cornell_shadow", "configs[1]"), ("transmittance", "configs[4]")):
        sub.config = name
        res[key] = compact(cornell_config(sub, torch, rt, S, device, ci=ci))
    sub.config = "vpl"
    res["configs[0]"] = compact(vpl_config(sub, torch, rt, S, device, ci=ci))
    ci["fog"].close()
    ci["scene"].close()
    res["seconds"] = round(time.time() - t0, 1)
    res["note"] = ("run after the timed loop of the headline configuration, on the same GPU, same --steps/--warmup; "
                   "every entry is timed like the headline (wall clock around K synchronised steps, HIP events around "
                   "the dominant kernel)")
    return res


def extras(torch, rt, S, scene, d_o, d_d, n, device):
    """Untimed-by-the-driver side measurements on the bounce batch and the Cornell passes."""
    from oracle import oracle_py as O
    res = {}
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    def timeit(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=device)
    ms = timeit(lambda: scene.trace_any_device(d_o.data_ptr(), d_d.data_ptr(), n, bits.data_ptr(), sh))
    res["any_hit_same_batch_mrays_s"] = round(n / ms / 1e3, 2)
    # the same closest-hit step issued alternately on two streams: the drain of one launch (9 % of
    # it) overlaps the start of the next, as batches of a running renderer would
    two = [torch.cuda.Stream(device=device) for _ in range(2)]
    h2 = [torch.empty((n, 4), dtype=torch.int32, device=device) for _ in range(2)]
    for reps in (2, 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            scene.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n, h2[i % 2].data_ptr(), two[i % 2].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    res["closest_hit_two_streams_mrays_s"] = round(n / dt / 1e6, 2)
    ms = timeit(lambda: scene.trace_closest_noalloc_device(d_o.data_ptr(), d_d.data_ptr(), n, h2[0].data_ptr(), sh))
    res["closest_hit_noalloc_entry_mrays_s"] = round(n / ms / 1e3, 2)
    del h2
    # Cornell passes
    ci = cornell_inputs(torch, rt, S, device)
    cs, cam, fog, vp, mats = ci["scene"], ci["cam"], ci["fog"], ci["vpls"], ci["mats"]
    npix = 1024 * 1024
    po = torch.empty((npix, 4), dtype=torch.float32, device=device)
    pd = torch.empty((npix, 4), dtype=torch.float32, device=device)
    ph = torch.empty((npix, 4), dtype=torch.int32, device=device)
    cs.gen_primary_rays_device(cam, 1024, 0, npix, po.data_ptr(), pd.data_ptr(), sh)
    ms = timeit(lambda: cs.trace_primary_device(cam, 1024, 0, npix, ph.data_ptr(), sh))
    res["cornell_primary_mrays_s"] = round(npix / ms / 1e3, 2)
    pi = torch.empty((64, 1024, 1024), dtype=torch.float32, device=device)
    ms = timeit(lambda: cs.compute_pi_density_device(fog, cam, 1024, 1024, pi.data_ptr(), sh), reps=2)
    res["compute_pi_density_1024_ms"] = round(ms, 3)   # reference: 3.1 s on 8 CPU threads
    # one frame of the software shading stage (Surface.frag + tone mapping) over the Cornell config:
    # 1 primary light + the golden 150-VPL set, visibility and transmittance from the shadow batch
    ppls = np.zeros((1, 8), np.float32)
    ppls[0, :3], ppls[0, 4:7] = S.PRIM_PL_POS, S.PRIM_PL_INTENS
    l151 = np.concatenate([ppls[:, :3], ci["lights"]]).astype(np.float32)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    d_l, d_ppl, d_vpl, d_mat = up(l151), up(ppls), up(vp.view(np.int32)), up(mats)
    npair = npix * 151
    sb = torch.zeros((npair + 63) // 64, dtype=torch.int64, device=device)
    tr = torch.empty(npair, dtype=torch.float32, device=device)
    accum = torch.zeros((npix, 4), dtype=torch.float32, device=device)
    rgb = torch.zeros((npix, 3), dtype=torch.float32, device=device)
    prm = rt.ShadeParams()
    prm.cam_pos[:] = S.CAM_POS
    prm.res_x = prm.res_y = 1024
    prm.first, prm.n_ppls, prm.n_vpls = 0, 1, 150
    prm.gi_enabled, prm.clamp_rsq, prm.frame_id, prm.pi_res_z = 1, 1, 0, 64
    prm.sca_albedo = float(np.float32(S.SCA_K) / (np.float32(S.ABS_K) + np.float32(S.SCA_K)))

    # the volume pass (Volume.frag at 512^2 x 8 samples x 151 lights) and Combine.frag
    fd = torch.zeros((npix, 2), dtype=torch.float32, device=device)
    hal = up(O.halton2())
    rnd = up(np.float32(0.03) * np.random.default_rng(7).random(512 * 512, dtype=np.float32) - np.float32(0.015))
    n_vp = 512 * 512 * 8
    pts = torch.zeros((n_vp, 4), dtype=torch.float32, device=device)
    vbits = torch.zeros((n_vp * 151 + 63) // 64, dtype=torch.int64, device=device)
    vol = torch.zeros((512 * 512, 4), dtype=torch.float32, device=device)
    depth = torch.zeros(npix, dtype=torch.float32, device=device)
    prm.transm_opt, prm.sca_k = 1, float(np.float32(S.SCA_K))
    ext_k = float(np.float32(S.ABS_K) + np.float32(S.SCA_K))

    def shade():
        cs.shade_surface_device(prm, po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, d_mat.data_ptr(), len(mats),
                                d_ppl.data_ptr(), d_vpl.data_ptr(), sb.data_ptr(), accum.data_ptr(), fog=fog,
                                transm_ptr=tr.data_ptr(), pi_dens_ptr=pi.data_ptr(), fog_dist_ptr=fd.data_ptr(), stream=sh)

    def volume():
        cs.volume_samples_device(fog, prm, po.data_ptr(), pd.data_ptr(), ph.data_ptr(), fd.data_ptr(), rnd.data_ptr(),
                                 hal.data_ptr(), pi.data_ptr(), pts.data_ptr(), sh)
        cs.shadow_points_device(pts.data_ptr(), n_vp, d_l.data_ptr(), 151, vbits.data_ptr(), sh)
        cs.shade_volume_device(fog, prm, po.data_ptr(), pd.data_ptr(), ph.data_ptr(), fd.data_ptr(), rnd.data_ptr(),
                               hal.data_ptr(), pi.data_ptr(), d_ppl.data_ptr(), d_vpl.data_ptr(), vbits.data_ptr(),
                               vol.data_ptr(), sh)

    def combine():
        rt.depth_buffer_device(S.CAM_POS, S.CAM_DIR, 0.1, 2500.0, po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix,
                               depth.data_ptr(), device=device.index, stream=sh)
        rt.combine_device(1024, 1024, depth.data_ptr(), vol.data_ptr(), ext_k, 0, 12, accum.data_ptr(), rgb.data_ptr(),
                          device=device.index, stream=sh)

    def frame():   # frame 0 of the reference's render loop, every pass
        cs.trace_primary_device(cam, 1024, 0, npix, ph.data_ptr(), sh)
        cs.shadow_batch_fused_device(po.data_ptr(), pd.data_ptr(), ph.data_ptr(), npix, d_l.data_ptr(), 151,
                                     sb.data_ptr(), fog, tr.data_ptr(), stream=sh)
        shade()
        volume()
        combine()

    frame()
    res["cornell_shade_surface_ms"] = round(timeit(shade, reps=2), 3)
    res["cornell_volume_pass_ms"] = round(timeit(volume, reps=2), 3)   # 2.1 M sample points, 317 M any-hit rays
    res["cornell_combine_ms"] = round(timeit(combine, reps=2), 3)
    res["cornell_frame_ms"] = round(timeit(frame, reps=2), 3)
    res["cornell_frame_mean_rgb"] = [round(float(x), 5) for x in rgb.mean(0).tolist()]
    fog.close()
    cs.close()
    return res


if __name__ == "__main__":
    main()
