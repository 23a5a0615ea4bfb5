"""GPU: the RCCL branch of gigl_amd.shard.ShardedTracer on one GPU.  A process group of one rank over the
nccl (= RCCL) backend, the gather kept (gather_when_alone): the asynchronous gather on the second stream,
the event chain, the two hit buffers and, with pipeline=True, the two alternating traversal streams — the
code bench.py runs at N > 1 — driven by the HIP engine over several steps with a different ray batch per
step, and checked against single launches.  (More than one rank cannot share a GPU under RCCL; the
world-2 logic is covered over gloo in tests/test_multi_rank_gloo.py.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from gigl_amd import rt, scene_io as S, shard, synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
cv, ct, _, _ = S.load_obj(os.path.join(%(root)r, "tests", "golden", "cornell_box.obj"))
V, T = synth.tessellate(cv, ct, 16)
scene = rt.Scene(V, T, device=0, **S.KD_PARAMS)
n, steps = 300_001, 5
batches = [synth.bounce_rays(V, T, n, 0xC0FFEE + k) for k in range(steps)]
dev = [(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()) for o, d in batches]
want = []
for o, d in dev:
    h = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
    scene.trace_closest_device(o.data_ptr(), d.data_ptr(), n, h.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want.append(h.cpu().numpy())
for pipeline in (False, True):
    k = [0]
    def trace_fn(buf):
        o, d = dev[k[0]]
        k[0] += 1
        scene.trace_closest_device(o.data_ptr(), d.data_ptr(), n, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = shard.ShardedTracer(torch, dist, 1, 0, n, torch.device("cuda", 0), trace_fn, gather=True, backend="nccl",
                             pipeline=pipeline, gather_when_alone=True)
    assert st.overlap and st.gather and st.pipeline == pipeline
    st.prime()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):          # all steps in flight before anything is waited for
        st.step(events=ev[i])
    st.drain()
    torch.cuda.synchronize()
    assert all(a.elapsed_time(b) > 0 for a, b in ev)
    got = st.result().cpu().numpy()
    assert np.array_equal(got, want[steps - 1]), "gathered records of the last step (pipeline=%%s)" %% pipeline
    assert np.array_equal(st.local_hits().cpu().numpy(), want[steps - 1])
    # every step again, drained one by one: the gathered buffer holds that step's records
    k[0] = 0
    st2 = shard.ShardedTracer(torch, dist, 1, 0, n, torch.device("cuda", 0), trace_fn, gather=True, backend="nccl",
                              pipeline=pipeline, gather_when_alone=True)
    for i in range(steps):
        st2.step()
        st2.drain()
        torch.cuda.synchronize()
        assert np.array_equal(st2.result().cpu().numpy(), want[i]), (pipeline, i)
# the tree replication of bench.py at N > 1: header and device arrays through RCCL broadcasts straight out of the
# scene's memory, then adopted without a rebuild; the adopted scene traces bit-equal to the source
same, info = shard.replicate_scene(torch, dist, rt, scene, 0, torch.device("cuda", 0), backend="nccl", adopt_on_src=True)
assert same is scene and info["bytes"] == scene.stats().device_bytes
rep = info["adopted_on_src"]
o, d = dev[0]
h = torch.empty((n, 4), dtype=torch.int32, device="cuda:0")
rep.trace_closest_device(o.data_ptr(), d.data_ptr(), n, h.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(h.cpu().numpy(), want[0]), "adopted scene"
try:
    rep.export_tree()
    raise AssertionError("an adopted scene has no host tree")
except rt.GiglError:
    pass
rep.close()
dist.barrier()
dist.destroy_process_group()
scene.close()
print("LOOPBACK OK")
"""


def test_rccl_gather_and_pipelined_steps_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "LOOPBACK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
