"""Multi-GPU sharding of the env batch: envs are independent, so rank g of G owns the contiguous block
[g*N/G, (g+1)*N/G) and no collective sits on the data path.  The only exchange is an all-gather of rollout buffers
(obs, reward, done) towards the learner, once per chunk of steps (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(rank, world_size, n_total):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RolloutGatherer:
    """Collects K steps of per-rank rollout tensors and all-gathers them in one collective per field."""

    def __init__(self, n_local, obs_dim, chunk, device, dtype=torch.float32, world_size=None, always_collective=False):
        """always_collective: issue the all-gathers even for a world of one rank (exercises RCCL on a one-GPU box)"""
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.always = bool(always_collective)
        self.chunk = chunk
        self.i = 0
        self.obs = torch.zeros((chunk, n_local, obs_dim), dtype=dtype, device=device)
        self.rew = torch.zeros((chunk, n_local), dtype=dtype, device=device)
        self.done = torch.zeros((chunk, n_local), dtype=torch.uint8, device=device)
        if self.world > 1 or self.always:
            # concatenated along dim 0 (the layout both RCCL and gloo accept for all_gather_into_tensor)
            self.g_obs = torch.zeros((self.world * chunk, n_local, obs_dim), dtype=dtype, device=device)
            self.g_rew = torch.zeros((self.world * chunk, n_local), dtype=dtype, device=device)
            self.g_done = torch.zeros((self.world * chunk, n_local), dtype=torch.uint8, device=device)

    def push(self, obs, rew, term, trunc):
        """Returns the gathered (obs, rew, done) with layout [world, chunk, n_local, ...] when a chunk completes."""
        self.obs[self.i].copy_(obs)
        self.rew[self.i].copy_(rew)
        torch.bitwise_or(term, trunc, out=self.done[self.i])
        self.i += 1
        if self.i < self.chunk:
            return None
        self.i = 0
        if self.world == 1 and not self.always:
            return self.obs.unsqueeze(0), self.rew.unsqueeze(0), self.done.unsqueeze(0)
        dist.all_gather_into_tensor(self.g_obs, self.obs)
        dist.all_gather_into_tensor(self.g_rew, self.rew)
        dist.all_gather_into_tensor(self.g_done, self.done)
        w, c = self.world, self.chunk
        return (self.g_obs.view(w, c, *self.obs.shape[1:]), self.g_rew.view(w, c, *self.rew.shape[1:]),
                self.g_done.view(w, c, *self.done.shape[1:]))

    @staticmethod
    def to_global(t):
        """[world, chunk, n_local, ...] -> [chunk, world * n_local, ...] (global env order)."""
        w, c, n = t.shape[:3]
        return t.transpose(0, 1).reshape((c, w * n) + tuple(t.shape[3:]))
