"""Shard invariance across PROCESSES on the device (partition + concatenate == single-rank result).
Launch with torchrun from a shell (not from a process that already uses the GPU):
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/gpu_shard_check.py
Every rank steps its block of envs on cuda:0 (one-GPU box: ranks share the device and talk over gloo; on a real node use one
GPU per rank and nccl), the rollout chunks are all-gathered, and rank 0 compares them bit for bit with the same envs stepped in
one batch by a single handle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from tensegrity_rl_amd import TensegrityVecEnv
from tensegrity_rl_amd.sharding import RolloutGatherer, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
one_gpu = torch.cuda.device_count() < world
dev_index = 0 if one_gpu else int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)
dist.init_process_group("gloo" if one_gpu else "nccl")
n_total, chunk = 512, 8
lo, hi = shard_range(rank, world, n_total)
g = torch.Generator(device=dev); g.manual_seed(0)
acts = [torch.rand((n_total, 6), device=dev, generator=g) * 0.6 - 0.45 for _ in range(chunk)]   # same stream on every rank
env = TensegrityVecEnv(hi - lo, "tensegrity_env-v0", device=dev_index, seed=77, first_env_id=lo, auto_reset=False)
env.reset_tensors()
gather = RolloutGatherer(hi - lo, env.obs_dim, chunk, dev, world_size=world)
res = None
for t in range(chunk):
    o, r, te, tr, _, _ = env.step_tensors(acts[t][lo:hi].contiguous())
    res = gather.push(o, r, te, tr)
obs_g = RolloutGatherer.to_global(res[0]); rew_g = RolloutGatherer.to_global(res[1]); done_g = RolloutGatherer.to_global(res[2])
ok = True
if rank == 0:
    full = TensegrityVecEnv(n_total, "tensegrity_env-v0", device=dev_index, seed=77, first_env_id=0, auto_reset=False)
    full.reset_tensors()
    for t in range(chunk):
        o, r, te, tr, _, _ = full.step_tensors(acts[t])
        ok &= bool(torch.equal(o, obs_g[t])) and bool(torch.equal(r, rew_g[t])) and bool(torch.equal((te | tr), done_g[t]))
    print(f"shard check: world {world}, {n_total} envs, {chunk} steps: gathered rollout == single-handle rollout bit for bit: {ok}", flush=True)
    full.close()
env.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
