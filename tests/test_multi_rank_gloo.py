"""CPU, world_size 2 over gloo: the N > 1 path shards the ray batch into contiguous ranges,
every rank answers its own range against a replicated tree, and the hit records are gathered
into ray order on rank 0.  No GPU here, so the injected tracer is the oracle; what is under test is
gigl_amd.shard.ShardedTracer, the class bench.py drives at N > 1: sharding, double buffering, padding
and gather order, over several steps."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


STEPS = 3


def _batch(synth, V, T, n, step, first=0, count=None):
    """Step `step` traces its own ray batch, so a stale or mixed-up hit buffer cannot pass."""
    return synth.bounce_rays(V, T, n if count is None else count, 0x51A7D + step, first=first)


def _worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gigl_amd import scene_io as S, shard, synth
    from oracle import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, 4)
    ora = O.OracleScene(V, T, **S.KD_PARAMS)          # replicated tree
    lo, hi = shard.shard_range(n, world, rank)
    step_no = [0]

    def trace_fn(buf):                                # the tracer is injected: here the oracle stands in for the GPU
        if hi > lo:
            ro, rd = _batch(synth, V, T, n, step_no[0], first=lo, count=hi - lo)   # this rank's shard only
            hits = ora.trace_closest(ro, rd)
            buf[: hi - lo] = torch.from_numpy(hits.view(np.int32).reshape(-1, 4).copy())
        step_no[0] += 1

    # the same class bench.py drives with the HIP engine and RCCL
    st = shard.ShardedTracer(torch, dist, world, rank, n, "cpu", trace_fn, gather=True, backend="gloo")
    assert (st.lo, st.hi) == (lo, hi) and not st.overlap
    st.prime()                     # the communicator set-up bench.py keeps out of its timed region
    results = []
    for _ in range(STEPS):
        st.step()
        st.drain()
        full = st.result()
        results.append(None if full is None else full.numpy().copy())
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the bench's max-over-ranks timing reduction
    assert t.item() == world
    if rank == 0:
        np.save(out_path, np.stack(results))
    else:
        assert all(r is None for r in results)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(4099, 2), (64, 2), (1003, 8), (5, 8)])
def test_sharded_trace_and_gather_equals_single_batch(tmp_path, n, world):
    """world 8 is the node bench.py is launched on: ragged ranges (1003 = 8 x 125 + 3), and fewer rays than
    ranks (three ranks with an empty range)."""
    import torch.multiprocessing as mp
    from gigl_amd import scene_io as S, synth
    from oracle import oracle_py as O
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    got = np.load(out).view(np.uint32)
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, 4)
    ora = O.OracleScene(V, T, **S.KD_PARAMS)
    for step in range(STEPS):
        ro, rd = _batch(synth, V, T, n, step)
        want = ora.trace_closest(ro, rd)
        assert np.array_equal(got[step], want.view(np.uint32).reshape(-1, 4)), step


def test_shard_ranges_cover_the_batch():
    from gigl_amd import shard
    for n in (0, 1, 7, 64, 4099, 1 << 27):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(n, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
