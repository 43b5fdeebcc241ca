"""CPU, world_size 2 over gloo: the sharding + all-gather(v) of packed matches that the multi-GPU bench /
ZEB runner uses with RCCL on the GPU box."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_pairs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gim_amd.runner import all_gather_matches, pack_matches, shard_pairs
    mine = shard_pairs(n_pairs, rank, world)
    rows = []
    for p in mine:  # fake per-pair "forward": p+1 matches for pair p (ragged on purpose; pair 0 of rank 0 -> 1)
        m = p + 1
        g = torch.Generator().manual_seed(p)
        data = {"mkpts0_f": torch.rand(m, 2, generator=g), "mkpts1_f": torch.rand(m, 2, generator=g),
                "mconf": torch.rand(m, generator=g), "m_bids": torch.zeros(m, dtype=torch.int64)}
        rows.append(pack_matches(data, [p]))
    rows = torch.cat(rows) if rows else torch.zeros(0, 6)
    allrows = all_gather_matches(rows)
    if rank == 0:
        torch.save(allrows, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    from gim_amd.runner import shard_pairs
    n_pairs, world = 7, 2
    assert shard_pairs(7, 0, 2) == [0, 2, 4, 6] and shard_pairs(7, 1, 2) == [1, 3, 5]
    assert sorted(shard_pairs(7, 0, 2) + shard_pairs(7, 1, 2)) == list(range(7))  # no padding / duplicates
    out = str(tmp_path / "rows.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n_pairs, out), nprocs=world, join=True)
    rows = torch.load(out)
    assert rows.shape == (sum(p + 1 for p in range(n_pairs)), 6)
    pid = rows[:, 0].long()
    assert torch.bincount(pid, minlength=n_pairs).tolist() == [p + 1 for p in range(n_pairs)]
    # content survives the padded gather bit-exactly
    g = torch.Generator().manual_seed(3)
    mk0 = torch.rand(4, 2, generator=g)
    assert torch.equal(rows[pid == 3][:, 1:3], mk0)


def test_single_process_is_identity():
    from gim_amd.runner import all_gather_matches
    r = torch.rand(5, 6)
    assert all_gather_matches(r) is r
