"""N>1 path on CPU: world_size-2 gloo.  Sharding must tile the env ids exactly and the chunked rollout all-gather must
reproduce the single-rank tensor (partition + concatenate == single-rank result)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tensegrity_rl_amd.sharding import RolloutGatherer, shard_range


def test_shard_range_tiles_exactly():
    for n in (1, 7, 4096, 32768, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(r, w, n) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, D, chunk, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(rank, world, n_total)
    g = RolloutGatherer(hi - lo, D, chunk, torch.device("cpu"))
    res = None
    for t in range(chunk):
        ids = torch.arange(lo, hi, dtype=torch.float32)
        obs = ids[:, None] * 100 + t + torch.arange(D, dtype=torch.float32)[None, :] * 0.01
        rew = ids + 0.5 * t
        term = (torch.arange(lo, hi) % 3 == t % 3).to(torch.uint8)
        trunc = (torch.arange(lo, hi) % 5 == 0).to(torch.uint8)
        res = g.push(obs, rew, term, trunc)
    obs_g = RolloutGatherer.to_global(res[0]); rew_g = RolloutGatherer.to_global(res[1]); done_g = RolloutGatherer.to_global(res[2])
    if rank == 0:
        torch.save({"obs": obs_g, "rew": rew_g, "done": done_g}, out)
    dist.barrier()
    dist.destroy_process_group()


def _worker_overlap(rank, world, port, n_total, D, chunk, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(rank, world, n_total)
    g = RolloutGatherer(hi - lo, D, chunk, torch.device("cpu"), overlap=True)
    got = []
    for t in range(3 * chunk):
        ids = torch.arange(lo, hi, dtype=torch.float32)
        obs = ids[:, None] * 100 + t + torch.arange(D, dtype=torch.float32)[None, :] * 0.01
        res = g.push(obs, ids + 0.5 * t, torch.zeros(hi - lo, dtype=torch.uint8), torch.zeros(hi - lo, dtype=torch.uint8))
        if (t + 1) % chunk == 0:
            assert (res is None) == (t + 1 == chunk)      # results arrive one chunk late
        else:
            assert res is None
        if res is not None:
            got.append(RolloutGatherer.to_global(res[0]).clone())
    got.append(RolloutGatherer.to_global(g.flush()[0]).clone())
    if rank == 0:
        torch.save(torch.cat(got), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_overlapped_allgather_returns_every_chunk_once_in_order(tmp_path):
    n_total, D, chunk = 10, 3, 2
    out = str(tmp_path / "o.pt")
    mp.spawn(_worker_overlap, args=(2, _free_port(), n_total, D, chunk, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    ids = torch.arange(n_total, dtype=torch.float32)
    assert got.shape == (3 * chunk, n_total, D)
    for t in range(3 * chunk):
        assert torch.equal(got[t], ids[:, None] * 100 + t + torch.arange(D, dtype=torch.float32)[None, :] * 0.01)


@pytest.mark.timeout(120)
def test_rollout_allgather_world2_matches_single_rank(tmp_path):
    n_total, D, chunk = 64, 5, 4
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_total, D, chunk, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    ids = torch.arange(n_total, dtype=torch.float32)
    for t in range(chunk):
        obs = ids[:, None] * 100 + t + torch.arange(D, dtype=torch.float32)[None, :] * 0.01
        assert torch.equal(got["obs"][t], obs)
        assert torch.equal(got["rew"][t], ids + 0.5 * t)
        done = ((torch.arange(n_total) % 3 == t % 3) | (torch.arange(n_total) % 5 == 0)).to(torch.uint8)
        assert torch.equal(got["done"][t], done)


def _worker_ragged(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(rank, world, 11)  # 6 + 5: not a valid layout for the all-gather
    try:
        RolloutGatherer(hi - lo, 3, 2, torch.device("cpu"))
        msg = "no error"
    except ValueError as e:
        msg = str(e)
    if rank == 0:
        torch.save({"msg": msg}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ragged_shards_are_refused_on_every_rank_instead_of_hanging(tmp_path):
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker_ragged, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "equal size" in torch.load(out, weights_only=True)["msg"]
