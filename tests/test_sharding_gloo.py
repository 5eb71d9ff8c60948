"""The N>1 path on CPU: two gloo ranks shard the sample stream by generator subsequence, render their
shard (with the oracle -- there is no GPU here), and one reduce gives the single-process result."""

import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, threads, out_path):
    import torch
    import torch.distributed as dist

    from cudabrot_amd.sharding import reduce_histogram, shard_subsequences
    from oracle import binding as oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = shard_subsequences(rank, world, threads)
    hist, cnt = oracle.render(200, 160, 300, 20, n, 2, first_subsequence=first)
    t = torch.from_numpy(hist.view(np.int64).copy())
    reduce_histogram(t, dst=0)
    total = torch.tensor([cnt["samples"], cnt["increments"]], dtype=torch.int64)
    dist.reduce(total, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy().view(np.uint64))
        np.save(out_path + ".totals.npy", total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reduce_to_the_single_rank_histogram(oracle, tmp_path):
    import torch.multiprocessing as mp

    threads, world = 1536, 2
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(world, _free_port(), threads, out), nprocs=world, join=True)
    reduced = np.load(out)
    totals = np.load(out + ".totals.npy")
    whole, cnt = oracle.render(200, 160, 300, 20, world * threads, 2)
    assert np.array_equal(reduced, whole)
    assert int(totals[0]) == cnt["samples"] and int(totals[1]) == cnt["increments"] == int(whole.sum())


def test_shard_subsequences():
    from cudabrot_amd.sharding import shard_subsequences

    t = 512 * 512
    assert shard_subsequences(0, 8, t) == (0, t)
    assert shard_subsequences(7, 8, t) == (7 * t, t)
    # shards tile [0, N*T) without gaps or overlap
    edges = [shard_subsequences(r, 8, t) for r in range(8)]
    assert all(edges[r][0] + edges[r][1] == edges[r + 1][0] for r in range(7))
    with pytest.raises(ValueError):
        shard_subsequences(8, 8, t)


def test_reduce_is_exact_for_counts_beyond_2_pow_63():
    """u64 counters travel as int64 bit patterns; two's-complement addition keeps the bits."""
    import torch

    a = np.array([(1 << 63) + 5, 7], dtype=np.uint64)
    b = np.array([(1 << 62), 9], dtype=np.uint64)
    s = (torch.from_numpy(a.view(np.int64).copy()) + torch.from_numpy(b.view(np.int64).copy())).numpy().view(np.uint64)
    assert s[0] == np.uint64((1 << 63) + (1 << 62) + 5) and s[1] == 16
