"""Learnability check (diagnostic driver, not a test and not the product): a plain-torch PPO on top of the GPU-resident rollout
buffers (tensegrity_rl_amd/rollout.py) trains tr_env-v0 "straight" with 4096 envs on one MI355X.  Everything stays on the device:
policy -> step_tensors -> DeviceRolloutBuffer -> GAE -> minibatch updates.  Prints mean forward reward per env-step before/after.
Behavioural evidence for BASELINE configs #3 / #4 (SB3 itself is not installable here or on the GPU box): the reward a learner sees goes up.
usage: python tests/gpu_ppo_demo.py [iterations] [n_envs] [n_steps] [--uneven] [--env-id tr_env-v0|tensegrity_env-v0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from tensegrity_rl_amd import TensegrityVecEnv
from tensegrity_rl_amd.rollout import DeviceRolloutBuffer, collect

pos = [a for a in sys.argv[1:] if not a.startswith("--") and (sys.argv[sys.argv.index(a) - 1] != "--env-id")]
iters = int(pos[0]) if len(pos) > 0 else 60
n = int(pos[1]) if len(pos) > 1 else 4096
T = int(pos[2]) if len(pos) > 2 else 32
uneven = "--uneven" in sys.argv
env_id = sys.argv[sys.argv.index("--env-id") + 1] if "--env-id" in sys.argv else "tr_env-v0"
dev = torch.device("cuda")
torch.manual_seed(0)
kw = {}
if uneven:
    import numpy as np
    kw = dict(model="uneven", terrain=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_100.npy")))
env = TensegrityVecEnv(n, env_id, desired_action="straight", desired_direction=1, seed=1, **kw)
print(f"# PPO (plain torch, everything on the device) on {env_id} straight/+1, {'uneven-ground heightfield' if uneven else 'flat floor'}, {n} envs x {T} steps per iteration, {iters} iterations", flush=True)
D = env.obs_dim
lo, hi = -0.45, 0.15


class RunningNorm:
    def __init__(self, d):
        self.mean = torch.zeros(d, device=dev); self.var = torch.ones(d, device=dev); self.count = 1e-4

    def update(self, x):
        b, m, v = x.shape[0], x.mean(0), x.var(0, unbiased=False)
        tot = self.count + b
        delta = m - self.mean
        self.mean = self.mean + delta * b / tot
        self.var = (self.var * self.count + v * b + delta ** 2 * self.count * b / tot) / tot
        self.count = tot

    def __call__(self, x):
        return ((x - self.mean) / torch.sqrt(self.var + 1e-8)).clamp(-10, 10)


norm = RunningNorm(D)
pi = nn.Sequential(nn.Linear(D, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh(), nn.Linear(256, 6)).to(dev)
vf = nn.Sequential(nn.Linear(D, 256), nn.Tanh(), nn.Linear(256, 256), nn.Tanh(), nn.Linear(256, 1)).to(dev)
log_std = nn.Parameter(torch.full((6,), -0.5, device=dev))
opt = torch.optim.Adam(list(pi.parameters()) + list(vf.parameters()) + [log_std], lr=3e-4)


def to_env(u):  # squashless Gaussian in [-1, 1] units, clipped, mapped onto the action box
    return lo + 0.5 * (u.clamp(-1, 1) + 1) * (hi - lo)


@torch.no_grad()
def policy(obs):
    x = norm(obs.float())
    mu = pi(x)
    std = log_std.exp()
    u = mu + std * torch.randn_like(mu)
    logp = (-0.5 * ((u - mu) / std) ** 2 - log_std - 0.9189385).sum(1)
    policy.last_u = u
    return to_env(u), vf(x).squeeze(1), logp


@torch.no_grad()
def value(obs):
    return vf(norm(obs.float())).squeeze(1)


buf = DeviceRolloutBuffer(T, n, D, device=dev, gamma=0.99, gae_lambda=0.95)
raw_actions = torch.zeros(T, n, 6, device=dev)
obs = env.reset_tensors().clone()
first_rew = 0.0
t0 = time.time()
for it in range(iters):
    buf.reset()
    # collect() stores env-space actions; PPO needs the pre-squash sample, so record it alongside
    step_i = [0]
    def pol(o):
        a, v, lp = policy(o)
        raw_actions[step_i[0]] = policy.last_u; step_i[0] += 1
        return a, v, lp
    norm.update(obs.float())
    obs, stats = collect(env, pol, buf, T, obs=obs, value_fn=value)
    norm.update(buf.obs.reshape(-1, D))
    adv = buf.advantages.reshape(-1); ret = buf.returns.reshape(-1)
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    X = norm(buf.obs.reshape(-1, D)); U = raw_actions.reshape(-1, 6); LP0 = buf.log_probs.reshape(-1)
    N = X.shape[0]
    for ep in range(4):
        perm = torch.randperm(N, device=dev)
        for s in range(0, N, 16384):
            idx = perm[s:s + 16384]
            mu = pi(X[idx]); std = log_std.exp()
            lp = (-0.5 * ((U[idx] - mu) / std) ** 2 - log_std - 0.9189385).sum(1)
            ratio = (lp - LP0[idx]).exp()
            pl = -torch.min(ratio * adv[idx], ratio.clamp(0.8, 1.2) * adv[idx]).mean()
            vl = 0.5 * (vf(X[idx]).squeeze(1) - ret[idx]).pow(2).mean()
            loss = pl + 0.5 * vl
            opt.zero_grad(); loss.backward()
            nn.utils.clip_grad_norm_(list(pi.parameters()) + list(vf.parameters()), 0.5)
            opt.step()
    first_rew = buf.rewards.mean().item() if it == 0 else first_rew
    if it % 5 == 0 or it == iters - 1:
        print(f"iter {it:3d}  env-steps {(it + 1) * n * T / 1e6:6.2f} M  wall {time.time() - t0:6.1f} s  mean reward/step {buf.rewards.mean().item():+.4f}  "
              f"episodes {stats['episodes']}  mean len {stats['mean_length']:.0f}  std {log_std.exp().mean().item():.3f}", flush=True)
wall = time.time() - t0
print(f"# mean reward per env-step: first iteration {first_rew:+.4f} -> last {buf.rewards.mean().item():+.4f};  {iters * n * T / wall / 1e6:.2f} M env-steps/s with the learner in the loop "
      f"({iters * n * T / 1e6:.1f} M env-steps in {wall:.0f} s)", flush=True)
env.close()
