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
import torch
import torch.distributed as dist


def shard_pairs(n_pairs, rank, world):
    """indices of the pairs this rank processes (round robin, no padding)"""
    return list(range(rank, n_pairs, world))


def pack_matches(data, pair_ids):
    """data: dict after LoFTR.forward; pair_ids: global pair index of every batch element (list/tensor).
    Returns fp32 [M, 6] on the device of the matches."""
    mk0, mk1, conf, mb = data["mkpts0_f"], data["mkpts1_f"], data["mconf"], data["m_bids"]
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
