"""CPU, world_size 2 over gloo: the N > 1 path shards the ray batch into contiguous ranges,
every rank answers its own range against a replicated tree, and the hit records are gathered
into ray order on rank 0.  No GPU here, so each rank's tracer is the oracle; what is under test
is the sharding, padding and gather order that bench.py and a multi-GPU host use."""
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
    ro, rd = synth.bounce_rays(V, T, hi - lo, 0x51A7D, first=lo)   # this rank's shard only
    hits = ora.trace_closest(ro, rd)
    local = torch.from_numpy(hits.view(np.int32).reshape(-1, 4).copy())
    full = shard.gather_hits(local, n, world, rank, dist, dst=0)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the bench's max-over-ranks timing reduction
    assert t.item() == world
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [4099, 64])
def test_sharded_trace_and_gather_equals_single_batch(tmp_path, n):
    import torch.multiprocessing as mp
    from gigl_amd import scene_io as S, synth
    from oracle import oracle_py as O
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out).view(np.uint32)
    cv, ct, _, _ = S.load_obj(os.path.join(ROOT, "tests", "golden", "cornell_box.obj"))
    V, T = synth.tessellate(cv, ct, 4)
    ro, rd = synth.bounce_rays(V, T, n, 0x51A7D)
    want = O.OracleScene(V, T, **S.KD_PARAMS).trace_closest(ro, rd)
    assert np.array_equal(got, want.view(np.uint32).reshape(-1, 4))
