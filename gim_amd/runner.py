"""Multi-GPU pair runner: one process per GPU, image pairs sharded with no collective on the critical path.

The reference shards ZEB pairs with Lightning's DistributedSampler (`test.py:193-197`; pads by repeating
samples and de-duplicates by identifier, `trainer/lightning.py:109,253`) and gathers *pickled* per-pair
metrics over a gloo side group once per scene (`tools/comm.py:141-176`).  Here:

  * `shard_pairs`: pair p -> rank p % world (same assignment as DistributedSampler without shuffle),
    but without padding, so nothing has to be de-duplicated;
  * `pack_matches`: matches of a batch as fp32 rows [pair_id, x0, y0, x1, y1, conf] (24 B / match);
  * `all_gather_matches`: all-gather(v) of those rows = one all_gather of counts + one of rows padded to
    the max count.  Backend "nccl" (= RCCL over xGMI) for device tensors, "gloo" on CPU (tests).
    Called once per scene / run, never per pair.
"""
import os

import torch
import torch.distributed as dist


def bind_rank_to_cores(local_rank, local_world, max_threads=8):
    """One process per GPU on one host: give every rank its own slice of the host's cores and cap torch's intra-op threads, so
    that N ranks doing host-side work (image decode, pose estimation, the match hand-out) do not oversubscribe each other --
    at ~700 pairs/s/GPU the host, not xGMI, is the scaling risk (the reference leaves this to Lightning's
    DataLoader workers, test.py:193-218).  Returns the number of cores of the slice (0: affinity not available)."""
    n = os.cpu_count() or 1
    per = max(1, n // max(1, local_world))
    cores = list(range(local_rank * per, min(n, (local_rank + 1) * per)))
    got = 0
    if cores and hasattr(os, "sched_setaffinity"):
        try:
            allowed = sorted(os.sched_getaffinity(0))
            if len(allowed) >= local_world:          # respect a cpuset narrower than the machine
                per = max(1, len(allowed) // local_world)
                cores = allowed[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, cores)
            got = len(cores)
        except OSError:
            got = 0
    torch.set_num_threads(max(1, min(max_threads, per)))
    return got


class HostPairFeeder:
    """Image batches that start in (pinned) host memory: double-buffered host-to-device staging on a COPY stream, so that the
    transfer of batch s + 1 runs on the DMA engines while the kernels of batch s run (the reference's DataLoader hands over host
    tensors and `.to(device)` them on the compute stream, trainer/lightning.py:101-110).

        feeder.put([img0_host, img1_host])           # enqueue the copy of the first batch
        for s in range(n):
            a, b = feeder.get()                      # compute stream waits for that batch's copy
            if s + 1 < n: feeder.put(next_batch)     # next copy goes out before this batch's kernels are launched
            model({... a, b ...}); feeder.done()     # slot may be overwritten once the work enqueued so far has run
    """

    def __init__(self, device, depth=2):
        self.device, self.depth = torch.device(device), depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bufs, self.ready, self.free = [None] * depth, [None] * depth, [None] * depth
        self.head = self.tail = 0
        self._last = None

    def put(self, tensors):
        k = self.head % self.depth
        assert self.head - self.tail < self.depth, "HostPairFeeder: every slot holds an unconsumed batch"
        if self.bufs[k] is None or any(b.shape != t.shape or b.dtype != t.dtype for b, t in zip(self.bufs[k], tensors)):
            self.bufs[k] = [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in tensors]
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))   # allocated on the compute stream
        if self.free[k] is not None:
            self.copy_stream.wait_event(self.free[k])
        with torch.cuda.stream(self.copy_stream):
            for b, t in zip(self.bufs[k], tensors):
                b.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[k] = ev
        self.head += 1

    def get(self):
        assert self.tail < self.head, "HostPairFeeder.get() without a pending put()"
        k = self.tail % self.depth
        torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        self.tail += 1
        self._last = k
        return self.bufs[k]

    def done(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[self._last] = ev


def shard_pairs(n_pairs, rank, world):
    """indices of the pairs this rank processes (round robin, no padding)"""
    return list(range(rank, n_pairs, world))


def pack_matches(data, pair_ids):
    """data: dict after LoFTR.forward; pair_ids: global pair index of every batch element (list / tensor), or ONE int = the id
    of batch element 0 when the batch holds consecutive pairs (no host-to-device copy then).
    Returns fp32 [M, 6] on the device of the matches.  Device tensors go through one HIP kernel (gim_pack_matches); CPU
    tensors (the gloo tests of the launch / gather protocol, which run no model) are packed with torch."""
    mk0, mk1, conf, mb = data["mkpts0_f"], data["mkpts1_f"], data["mconf"], data["m_bids"]
    if mb.is_cuda:
        from . import ops
        if isinstance(pair_ids, int):
            return ops.pack_matches(mb, mk0.float(), mk1.float(), conf.float(), None, pair_ids)
        ids = list(pair_ids) if not torch.is_tensor(pair_ids) else pair_ids.tolist()
        if all(int(v) == int(ids[0]) + k for k, v in enumerate(ids)):
            return ops.pack_matches(mb, mk0.float(), mk1.float(), conf.float(), None, int(ids[0]))
        return ops.pack_matches(mb, mk0.float(), mk1.float(), conf.float(), torch.as_tensor(ids, dtype=torch.int64, device=mb.device))
    if isinstance(pair_ids, int):
        pair_ids = [pair_ids + k for k in range(int(mb.max()) + 1 if mb.numel() else 0)]
    pid = torch.as_tensor(pair_ids, device=mb.device, dtype=torch.float32)[mb]
    return torch.cat([pid[:, None], mk0.float(), mk1.float(), conf.float()[:, None]], dim=1).contiguous()


def all_gather_matches(rows, group=None):
    """rows: fp32 [M_rank, 6].  Returns fp32 [sum M, 6] (rank order) on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    pad = torch.zeros(mx, rows.shape[1], dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
