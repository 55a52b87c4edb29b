"""Multi-GPU sharding of the env batch: envs are independent, so rank g of G owns the contiguous block
[g*N/G, (g+1)*N/G) and no collective sits on the data path.  The only exchange is an all-gather of rollout buffers
(obs, reward, done) towards the learner, once per chunk of steps (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(rank, world_size, n_total):
    """Contiguous block of global env ids owned by `rank` (sizes differ by at most one; RolloutGatherer needs them equal, i.e.
    n_total divisible by world_size -- BASELINE's configs are: 4096 per GPU)."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RolloutGatherer:
    """Collects K steps of per-rank rollout tensors and all-gathers them in one collective per field.

    overlap=False: `push` returns the gathered chunk as soon as it is complete (in stream order).
    overlap=True: the collectives are issued asynchronously on RCCL's own stream and the next chunk is collected into a second
    set of buffers meanwhile, so stepping never waits for xGMI; `push` then hands out the *previous* chunk (None for the very
    first) and `flush()` the last one."""

    def __init__(self, n_local, obs_dim, chunk, device, dtype=torch.float32, world_size=None, always_collective=False, overlap=False):
        """always_collective: issue the all-gathers even for a world of one rank (exercises RCCL on a one-GPU box)"""
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.always = bool(always_collective)
        self.overlap = bool(overlap)
        self.chunk = chunk
        self.i = 0
        self.cur = 0
        self.sets = []
        coll = self.world > 1 or self.always
        if self.world > 1 and dist.is_initialized():
            # all_gather_into_tensor needs the same shard on every rank; a ragged split would hang or corrupt the collective
            lohi = torch.tensor([n_local, -n_local], dtype=torch.int64, device=device)
            dist.all_reduce(lohi, op=dist.ReduceOp.MIN)
            if int(lohi[0]) != -int(lohi[1]):
                raise ValueError(f"RolloutGatherer: shards must have equal size on every rank (min {int(lohi[0])}, max {-int(lohi[1])}); "
                                 "make n_total divisible by the world size")
        for _ in range(2 if self.overlap else 1):
            b = {"obs": torch.zeros((chunk, n_local, obs_dim), dtype=dtype, device=device),
                 "rew": torch.zeros((chunk, n_local), dtype=dtype, device=device),
                 "done": torch.zeros((chunk, n_local), dtype=torch.uint8, device=device), "work": None}
            if coll:
                # concatenated along dim 0 (the layout both RCCL and gloo accept for all_gather_into_tensor)
                b["g_obs"] = torch.zeros((self.world * chunk, n_local, obs_dim), dtype=dtype, device=device)
                b["g_rew"] = torch.zeros((self.world * chunk, n_local), dtype=dtype, device=device)
                b["g_done"] = torch.zeros((self.world * chunk, n_local), dtype=torch.uint8, device=device)
            self.sets.append(b)
        self.collectives = 0  # all-gathers issued so far
        # the synchronous interface of earlier revisions
        self.obs, self.rew, self.done = self.sets[0]["obs"], self.sets[0]["rew"], self.sets[0]["done"]

    def _views(self, b):
        w, c = self.world, self.chunk
        if "g_obs" not in b:
            return b["obs"].unsqueeze(0), b["rew"].unsqueeze(0), b["done"].unsqueeze(0)
        return (b["g_obs"].view(w, c, *b["obs"].shape[1:]), b["g_rew"].view(w, c, *b["rew"].shape[1:]),
                b["g_done"].view(w, c, *b["done"].shape[1:]))

    def _wait(self, b):
        if b["work"] is not None:
            for wk in b["work"]:
                wk.wait()
            b["work"] = None

    def push(self, obs, rew, term, trunc):
        """Returns the gathered (obs, rew, done) with layout [world, chunk, n_local, ...] when a chunk completes
        (overlap=True: the chunk before it), else None."""
        b = self.sets[self.cur]
        if self.i == 0:
            self._wait(b)  # the collective that last read these buffers (two chunks ago) has long finished
        b["obs"][self.i].copy_(obs)
        b["rew"][self.i].copy_(rew)
        torch.bitwise_or(term, trunc, out=b["done"][self.i])
        self.i += 1
        if self.i < self.chunk:
            return None
        self.i = 0
        if "g_obs" not in b:
            return self._views(b)
        if not self.overlap:
            self.collectives += 3
            dist.all_gather_into_tensor(b["g_obs"], b["obs"])
            dist.all_gather_into_tensor(b["g_rew"], b["rew"])
            dist.all_gather_into_tensor(b["g_done"], b["done"])
            return self._views(b)
        self.collectives += 3
        b["work"] = [dist.all_gather_into_tensor(b["g_obs"], b["obs"], async_op=True),
                     dist.all_gather_into_tensor(b["g_rew"], b["rew"], async_op=True),
                     dist.all_gather_into_tensor(b["g_done"], b["done"], async_op=True)]
        prev = self.sets[1 - self.cur]
        had = prev.get("filled", False)
        b["filled"] = True
        self.cur = 1 - self.cur
        if not had:
            return None
        self._wait(prev)
        return self._views(prev)

    def flush(self):
        """overlap=True: wait for and return the chunk whose collectives are still in flight (None if there is none)."""
        if not self.overlap:
            return None
        last = self.sets[1 - self.cur]
        if not last.get("filled", False):
            return None
        self._wait(last)
        return self._views(last)

    @staticmethod
    def to_global(t):
        """[world, chunk, n_local, ...] -> [chunk, world * n_local, ...] (global env order)."""
        w, c, n = t.shape[:3]
        return t.transpose(0, 1).reshape((c, w * n) + tuple(t.shape[3:]))
