"""Ray-batch sharding for one process per GPU.

Rays are independent, so the path shards with no data-path collective: the tree is replicated,
rank r traces the contiguous ray range shard_range(n, world, r), and the only exchange is the
gather of the hit buffers to rank 0 (16 B per closest-hit ray).  ShardedTracer is that step, used
as it stands by bench.py (RCCL, one rank per GPU) and by tests/test_multi_rank_gloo.py (gloo, CPU
tensors, a stand-in tracer): step i's gather runs behind step i+1's traversal (two hit buffers, a
second stream), one traversal launch per step.  With pipeline=True consecutive steps also traverse on
two alternating streams, so the drain at the end of one step's launch (the last long rays, with most of
the chip idle) is filled by the next step's first waves.
"""


def shard_range(n, world, rank):
    """Contiguous [lo, hi) of rank `rank` of `world`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(n), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedTracer:
    """One rank's side of the sharded closest-hit step.

    trace_fn(buf): traces this rank's rays [lo, hi) into buf[:hi - lo] (a [m_max, 4] int32 tensor on
    `device`), asynchronously on the current stream when the device is a GPU.  dist is
    torch.distributed (or None when world == 1); the process group must be initialised.
    gather=False: no exchange at all (each rank keeps its hits).
    On a GPU with the nccl (= RCCL) backend the gather of step i is issued on a second stream behind
    an event and overlaps the traversal of step i + 1; every other combination (CPU tensors, or GPU
    tensors with gloo: through host memory) gathers synchronously.
    pipeline=True (GPU only): step i runs on traversal stream i % 2, under which trace_fn is called (it must
    launch on torch's current stream).  gather_when_alone=True keeps the gather at world == 1 (a loopback
    that exercises the RCCL branch on one GPU).
    """

    def __init__(self, torch, dist, world, rank, n_total, device, trace_fn, gather=True, backend="nccl",
                 pipeline=False, gather_when_alone=False):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.n_total = int(n_total)
        self.device = torch.device(device)
        self.trace_fn = trace_fn
        self.ranges = [shard_range(n_total, world, r) for r in range(world)]
        self.lo, self.hi = self.ranges[rank]
        self.m = self.hi - self.lo
        self.m_max = max(hi - lo for lo, hi in self.ranges)      # the collective wants equal sizes
        self.gather = bool(gather) and (world > 1 or (gather_when_alone and dist is not None))
        self.on_gpu = self.device.type == "cuda"
        self.overlap = self.gather and self.on_gpu and backend == "nccl"
        self.pipeline = bool(pipeline) and self.on_gpu
        new = lambda: torch.zeros((self.m_max, 4), dtype=torch.int32, device=self.device)
        self.bufs = [new(), new()] if (self.gather or self.pipeline) else [new()]
        self.trav_streams = [torch.cuda.Stream(device=self.device) for _ in range(2)] if self.pipeline else None
        self.gathered = None
        if self.gather and rank == 0:
            gdev = self.device if (self.overlap or not self.on_gpu) else torch.device("cpu")
            self.gathered = [torch.zeros((self.m_max, 4), dtype=torch.int32, device=gdev) for _ in range(world)]
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.overlap else None
        self.pending = {}
        self.i = 0

    def prime(self):
        """One tiny gather outside any timed region: the backend sets up its communicator on first use (seconds
        with RCCL), which would otherwise land in the first step of a run without warm-up steps."""
        if not self.gather or self.dist is None:
            return
        dev = self.gathered[0].device if self.rank == 0 else (self.device if (self.overlap or not self.on_gpu) else self.torch.device("cpu"))
        one = self.torch.zeros(4, dtype=self.torch.int32, device=dev)
        into = [self.torch.zeros(4, dtype=self.torch.int32, device=dev) for _ in range(self.world)] if self.rank == 0 else None
        if self.overlap:
            with self.torch.cuda.stream(self.comm_stream):
                self.dist.gather(one, into, dst=0)
            self.torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        else:
            self.dist.gather(one, into, dst=0)

    def step_stream(self, i=None):
        """The stream step i (default: the next one) traverses on; None on the CPU."""
        if not self.on_gpu:
            return None
        i = self.i if i is None else i
        return self.trav_streams[i % 2] if self.pipeline else self.torch.cuda.current_stream(self.device)

    def step(self, events=None):
        """One step.  events: optional (start, end) torch.cuda.Event pair recorded around the traversal on
        the stream it runs on."""
        if self.pipeline:
            if self.i < 2:   # the first launch on each traversal stream follows whatever the caller queued before
                self.trav_streams[self.i % 2].wait_stream(self.torch.cuda.current_stream(self.device))
            with self.torch.cuda.stream(self.trav_streams[self.i % 2]):
                self._step(events)
        else:
            self._step(events)

    def _step(self, events):
        torch, dist = self.torch, self.dist
        i = self.i
        self.i += 1
        buf = self.bufs[i % len(self.bufs)]
        if (i - 2) in self.pending:
            self.pending.pop(i - 2).wait()      # the gather that last read this buffer (stream-level wait on a GPU)
        stream = torch.cuda.current_stream(self.device) if self.on_gpu else None
        if events:
            events[0].record(stream)
        self.trace_fn(buf)
        if events:
            events[1].record(stream)
        if not self.gather:
            return
        if self.overlap:
            ev = torch.cuda.Event()
            ev.record(stream)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self.pending[i] = dist.gather(buf, self.gathered if self.rank == 0 else None, dst=0, async_op=True)
        else:
            src = buf.cpu() if self.on_gpu else buf     # gloo rehearsal on a GPU: through host memory
            dist.gather(src, self.gathered if self.rank == 0 else None, dst=0)

    def drain(self):
        """Completes every traversal and gather issued so far (and orders the current stream behind them)."""
        for k in sorted(self.pending):
            self.pending.pop(k).wait()
        if self.on_gpu:
            cur = self.torch.cuda.current_stream(self.device)
            if self.overlap:
                cur.wait_stream(self.comm_stream)
            if self.pipeline:
                for t in self.trav_streams:
                    cur.wait_stream(t)

    def local_hits(self):
        """This rank's records of the last step, [m, 4]."""
        return self.bufs[(self.i - 1) % len(self.bufs)][: self.m]

    def result(self):
        """Rank 0, after drain(): the last step's records of ALL rays in ray order, [n_total, 4]
        (None on other ranks; without a gather, this rank's own records)."""
        if not self.gather:
            return self.local_hits() if self.world == 1 or self.rank == 0 else None
        if self.rank != 0:
            return None
        out = self.torch.empty((self.n_total, 4), dtype=self.torch.int32, device=self.gathered[0].device)
        for r, (lo, hi) in enumerate(self.ranges):
            out[lo:hi] = self.gathered[r][: hi - lo]
        return out


class _DeviceBytes:
    """A device address range as an object torch.as_tensor can view without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def replicate_scene(torch, dist, rt, scene, rank, device, backend="nccl", src=0, adopt_on_src=False):
    """The flattened scene of rank `src` on every rank, without rebuilding it ("uploaded to GPU 0,
    then ncclBroadcast"): the header and the three device arrays are broadcast and the other ranks adopt them
    (gigl_scene_create_from_device_blob).  `scene` is rt.Scene on rank src and None elsewhere; returns
    (scene, info).  With the nccl backend the arrays go device to device (rank src broadcasts straight out of the
    scene's own memory); with gloo (CPU rehearsal of the rank logic) through host memory.  adopt_on_src: rank src
    also adopts a copy from the arrays it broadcast (info["adopted_on_src"]; how a one-rank group tests the path)."""
    import ctypes
    import time
    device = torch.device(device)
    direct = backend == "nccl"
    t0 = time.time()
    nhdr = ctypes.sizeof(rt.SceneBlob)
    views = [None, None, None]
    hdr = torch.zeros(nhdr + 1, dtype=torch.uint8)   # last byte: rank src could offer its arrays
    if rank == src:
        try:   # nothing collective happens in here: a failure is told to the other ranks through the header
            desc, ptrs = scene.export_device_blob()
            for k, nb in enumerate((desc.nb_nodes, desc.nb_leaf_ids, desc.nb_tris)):
                if nb:
                    views[k] = torch.as_tensor(_DeviceBytes(ptrs[k], nb), device=device)
            hdr[:nhdr] = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8)
            hdr[nhdr] = 1
        except Exception as e:   # noqa: BLE001
            views = [None, None, None]
            print("replicate_scene: rank %d cannot offer its arrays (%s); every rank builds its own tree" % (src, e), flush=True)
    if direct:
        hdr = hdr.to(device)
    dist.broadcast(hdr, src)
    hdr = hdr.cpu()
    if int(hdr[nhdr]) != 1:
        return (scene if rank == src else None), {"how": "rank %d could not export its arrays: one build per rank" % src,
                                                   "bytes": 0, "seconds": round(time.time() - t0, 3)}
    desc = rt.SceneBlob.from_buffer_copy(hdr[:nhdr].numpy().tobytes())
    piece = 1 << 30   # broadcasts of at most 1 GiB: no element count near 2^31 anywhere below
    arrays = []
    for k, nb in enumerate((desc.nb_nodes, desc.nb_leaf_ids, desc.nb_tris)):
        if nb == 0:
            arrays.append(None)
            continue
        t = views[k] if rank == src else torch.empty(int(nb), dtype=torch.uint8, device=device)
        for lo in range(0, int(nb), piece):
            part = t[lo:lo + piece]
            if direct:
                dist.broadcast(part, src)
            else:
                c = part.cpu()
                dist.broadcast(c, src)
                if rank != src:
                    part.copy_(c)
        arrays.append(t)
    if rank != src:
        torch.cuda.synchronize(device)
        ptr = lambda t: 0 if t is None else t.data_ptr()
        scene = rt.Scene.from_device_blob(desc, ptr(arrays[0]), ptr(arrays[1]), ptr(arrays[2]), device=device.index,
                                          stream=torch.cuda.current_stream(device).cuda_stream)
    else:
        torch.cuda.synchronize(device)
    adopted = None
    if rank == src and adopt_on_src:
        ptr = lambda t: 0 if t is None else t.data_ptr()
        adopted = rt.Scene.from_device_blob(desc, ptr(arrays[0]), ptr(arrays[1]), ptr(arrays[2]), device=device.index,
                                            stream=torch.cuda.current_stream(device).cuda_stream)
    del arrays, views
    if rank != src:
        torch.cuda.empty_cache()   # the received copies: the scene holds its own
    info = {"how": "rank %d builds, header + 3 device arrays broadcast over %s, other ranks adopt them (no rebuild)" % (src, backend),
            "bytes": int(desc.nb_nodes + desc.nb_leaf_ids + desc.nb_tris), "seconds": round(time.time() - t0, 3)}
    if adopted is not None:
        info["adopted_on_src"] = adopted
    return scene, info
