"""GPU-resident rollout / replay feed.

At ~1.7 M env-steps/s the host side of Stable-Baselines3 (NumPy replay buffer, one `info` dict per env per step,
run.py:41-101) is the bottleneck, not the simulator.  These buffers keep every transition in HBM as torch tensors and take
their inputs straight from `TensegrityVecEnv.step_tensors` -- no host copy, no Python loop over envs.

Semantics follow SB3's buffers [ext: stable_baselines3/common/buffers.py] so a learner written against them sees the same
data:
* `DeviceReplayBuffer` (off-policy: SAC/TD3): ring over time of `[T, N, ...]`; `next_obs` of a finished env is its
  `terminal_observation`, not the reset observation; with `handle_timeout_termination` a time-limit truncation is stored
  as `done = 0` so the critic bootstraps through it.
* `DeviceRolloutBuffer` (on-policy: PPO/A2C): `episode_starts`, values, log-probs, GAE(lambda) returns/advantages computed
  on the device; a truncated episode's last reward is topped up with `gamma * V(terminal_observation)` by `collect()`.
* `collect()` is `collect_rollouts` for both: policy -> `step_tensors` -> buffer, device tensors only.
"""
import numpy as np


class DeviceReplayBuffer:
    def __init__(self, buffer_size, num_envs, obs_dim, act_dim=6, device="cuda", dtype=None, handle_timeout_termination=True):
        import torch
        self.torch = torch
        self.n_envs = int(num_envs)
        self.capacity = max(int(buffer_size) // self.n_envs, 1)  # SB3: buffer_size // n_envs rows
        self.device = torch.device(device)
        dt = dtype or torch.float32
        z = lambda *s, d=dt: torch.zeros(s, dtype=d, device=self.device)
        self.obs = z(self.capacity, self.n_envs, obs_dim)
        self.next_obs = z(self.capacity, self.n_envs, obs_dim)
        self.actions = z(self.capacity, self.n_envs, act_dim)
        self.rewards = z(self.capacity, self.n_envs)
        self.dones = z(self.capacity, self.n_envs)
        self.timeouts = z(self.capacity, self.n_envs)
        self.handle_timeout_termination = handle_timeout_termination
        self.pos, self.full = 0, False
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(0)

    def size(self):
        return self.capacity if self.full else self.pos

    def add(self, obs, next_obs, action, reward, terminated, truncated):
        """One time step of all envs.  `next_obs` must already hold the terminal observation for finished envs."""
        t = self.torch
        i = self.pos
        self.obs[i].copy_(obs); self.next_obs[i].copy_(next_obs); self.actions[i].copy_(action)
        self.rewards[i].copy_(reward)
        term = terminated.to(self.rewards.dtype); trunc = truncated.to(self.rewards.dtype)
        self.dones[i].copy_(t.clamp(term + trunc, max=1.0))
        self.timeouts[i].copy_(trunc * (1.0 - term))  # SB3: info["TimeLimit.truncated"] = truncated and not terminated
        self.pos += 1
        if self.pos == self.capacity:
            self.full, self.pos = True, 0

    def sample(self, batch_size):
        t = self.torch
        upper = self.capacity if self.full else self.pos
        if upper == 0:
            raise ValueError("empty buffer")
        ti = t.randint(0, upper, (batch_size,), generator=self.generator, device=self.device)
        ei = t.randint(0, self.n_envs, (batch_size,), generator=self.generator, device=self.device)
        dones = self.dones[ti, ei]
        if self.handle_timeout_termination:
            dones = dones * (1.0 - self.timeouts[ti, ei])
        return {"observations": self.obs[ti, ei], "actions": self.actions[ti, ei], "next_observations": self.next_obs[ti, ei],
                "dones": dones.unsqueeze(1), "rewards": self.rewards[ti, ei].unsqueeze(1)}


class DeviceRolloutBuffer:
    def __init__(self, n_steps, num_envs, obs_dim, act_dim=6, device="cuda", gamma=0.99, gae_lambda=0.95, dtype=None):
        import torch
        self.torch = torch
        self.n_steps, self.n_envs = int(n_steps), int(num_envs)
        self.device = torch.device(device)
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)
        dt = dtype or torch.float32
        z = lambda *s: torch.zeros(s, dtype=dt, device=self.device)
        self.obs = z(self.n_steps, self.n_envs, obs_dim)
        self.actions = z(self.n_steps, self.n_envs, act_dim)
        self.rewards = z(self.n_steps, self.n_envs)
        self.episode_starts = z(self.n_steps, self.n_envs)
        self.values = z(self.n_steps, self.n_envs)
        self.log_probs = z(self.n_steps, self.n_envs)
        self.advantages = z(self.n_steps, self.n_envs)
        self.returns = z(self.n_steps, self.n_envs)
        self.pos = 0

    def reset(self):
        self.pos = 0

    def add(self, obs, action, reward, episode_start, value, log_prob):
        i = self.pos
        self.obs[i].copy_(obs); self.actions[i].copy_(action); self.rewards[i].copy_(reward)
        self.episode_starts[i].copy_(episode_start); self.values[i].copy_(value); self.log_probs[i].copy_(log_prob)
        self.pos += 1

    def compute_returns_and_advantage(self, last_values, dones):
        """GAE(lambda), SB3 RolloutBuffer.compute_returns_and_advantage [ext]: TD(lambda) returns = advantages + values."""
        last_gae = self.torch.zeros_like(last_values)
        for step in reversed(range(self.n_steps)):
            if step == self.n_steps - 1:
                next_non_terminal = 1.0 - dones.to(last_values.dtype)
                next_values = last_values
            else:
                next_non_terminal = 1.0 - self.episode_starts[step + 1]
                next_values = self.values[step + 1]
            delta = self.rewards[step] + self.gamma * next_values * next_non_terminal - self.values[step]
            last_gae = delta + self.gamma * self.gae_lambda * next_non_terminal * last_gae
            self.advantages[step] = last_gae
        self.returns = self.advantages + self.values

    def get(self, batch_size=None):
        """Minibatches over the flattened [n_steps * n_envs] samples in a random order (one pass)."""
        t = self.torch
        total = self.n_steps * self.n_envs
        perm = t.randperm(total, device=self.device)
        flat = lambda x: x.reshape(total, *x.shape[2:])
        fields = {"observations": flat(self.obs), "actions": flat(self.actions), "old_values": flat(self.values),
                  "old_log_prob": flat(self.log_probs), "advantages": flat(self.advantages), "returns": flat(self.returns)}
        bs = batch_size or total
        for s in range(0, total, bs):
            idx = perm[s:s + bs]
            yield {k: v[idx] for k, v in fields.items()}


def collect(venv, policy, buffer, n_steps, obs=None, value_fn=None):
    """`collect_rollouts` on the device.

    policy(obs[N,D]) -> action[N,6]                              (replay buffer)
    policy(obs[N,D]) -> (action[N,6], value[N], log_prob[N])     (rollout buffer; `value_fn(obs)` bootstraps truncations)
    Returns (last_obs, stats) with stats = dict(steps, episodes, mean_return, mean_length) accumulated on the device and
    synchronised once at the end."""
    t = venv.torch
    on_policy = isinstance(buffer, DeviceRolloutBuffer)
    if obs is None:
        obs = venv.reset_tensors()
    obs = obs.clone()
    n = venv.num_envs
    ep_start = t.ones(n, device=obs.device, dtype=t.float32)
    n_ep = t.zeros((), device=obs.device); ret_sum = t.zeros((), device=obs.device, dtype=t.float64); len_sum = t.zeros((), device=obs.device, dtype=t.float64)
    done = t.zeros(n, device=obs.device, dtype=t.bool)
    for _ in range(int(n_steps)):
        out = policy(obs)
        action = out[0] if on_policy else out
        new_obs, rew, term, trunc, tobs, info = venv.step_tensors(action.to(t.float32).contiguous())
        term_b, trunc_b = term.bool(), trunc.bool()
        done = term_b | trunc_b
        rew32 = rew.to(t.float32)
        if on_policy:
            if value_fn is not None:  # SB3 on_policy_algorithm.collect_rollouts: bootstrap with V(terminal_observation) on timeouts
                to = trunc_b & ~term_b
                tv = value_fn(tobs.to(t.float32))
                rew32 = rew32 + buffer.gamma * tv * to.to(t.float32)
            buffer.add(obs.to(t.float32), action, rew32, ep_start, out[1], out[2])
            ep_start = done.to(t.float32)
        else:
            next_obs = t.where(done.unsqueeze(1), tobs, new_obs)
            buffer.add(obs.to(t.float32), next_obs.to(t.float32), action, rew32, term, trunc)
        d64 = done.to(t.float64)
        n_ep += done.sum(); ret_sum += (info[:, 14].to(t.float64) * d64).sum(); len_sum += (info[:, 15].to(t.float64) * d64).sum()
        obs = new_obs.clone()  # step_tensors returns views of the env's buffers; the next step overwrites them
    if on_policy:
        last_values = value_fn(obs.to(t.float32)) if value_fn is not None else t.zeros(n, device=obs.device)
        buffer.compute_returns_and_advantage(last_values, done)
    e = float(n_ep.item())
    stats = {"steps": int(n_steps) * n, "episodes": int(e), "mean_return": float(ret_sum.item() / e) if e else float("nan"),
             "mean_length": float(len_sum.item() / e) if e else float("nan")}
    return obs, stats
