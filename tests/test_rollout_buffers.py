"""Device rollout / replay buffers: buffer logic on CPU tensors with a scripted fake env; the GPU test drives
the real simulator."""
import numpy as np
import pytest
import torch

from tensegrity_rl_amd.rollout import DeviceReplayBuffer, DeviceRolloutBuffer, collect


class FakeVenv:
    """Deterministic stand-in with the step_tensors() contract: env i terminates every (i + 2) steps, env 0 is truncated
    instead; observations encode (env, episode, t) so that terminal / reset observations can be told apart."""
    torch = torch

    def __init__(self, n, D=5):
        self.num_envs, self.obs_dim = n, D
        self.t = torch.zeros(n); self.ep = torch.zeros(n)
        self.obs = torch.zeros(n, D); self.tobs = torch.zeros(n, D); self.info = torch.zeros(n, 26)
        self.ret = torch.zeros(n)

    def _fill(self, buf, t, ep):
        buf[:, 0] = torch.arange(self.num_envs); buf[:, 1] = ep; buf[:, 2] = t

    def reset_tensors(self):
        self.t.zero_(); self._fill(self.obs, self.t, self.ep)
        return self.obs

    def step_tensors(self, a):
        n = self.num_envs
        self.t += 1
        rew = a[:, 0].clone() + 1.0
        self.ret += rew
        period = torch.arange(n) + 2.0
        done = self.t >= period
        term = done.clone(); term[0] = False
        trunc = done & ~term
        self._fill(self.tobs, self.t, self.ep)
        self.info[:, 14] = self.ret; self.info[:, 15] = self.t
        self.ep += done.float(); self.t[done] = 0; self.ret[done] = 0
        self._fill(self.obs, self.t, self.ep)
        return self.obs, rew, term.to(torch.uint8), trunc.to(torch.uint8), self.tobs, self.info


def test_replay_buffer_terminal_observation_and_timeouts():
    n = 4
    env = FakeVenv(n)
    buf = DeviceReplayBuffer(40, n, env.obs_dim, act_dim=6, device="cpu")
    assert buf.capacity == 10
    obs, stats = collect(env, lambda o: torch.zeros(o.shape[0], 6), buf, 7)
    assert buf.size() == 7 and stats["steps"] == 28
    # env 1 terminates every 3 steps: rows 2 and 5 are terminal; next_obs is the terminal observation (t = 3 of the old episode)
    assert buf.dones[2, 1] == 1 and buf.next_obs[2, 1, 2] == 3 and buf.next_obs[2, 1, 1] == 0
    assert buf.obs[3, 1, 2] == 0 and buf.obs[3, 1, 1] == 1  # the following row starts from the reset observation
    # env 0 is truncated every 2 steps: stored as a time-out, sampled with done = 0
    assert buf.timeouts[1, 0] == 1 and buf.dones[1, 0] == 1
    s = buf.sample(512)
    assert s["observations"].shape == (512, env.obs_dim) and s["dones"].shape == (512, 1) and s["rewards"].shape == (512, 1)
    e0 = s["observations"][:, 0] == 0
    assert (s["dones"][e0] == 0).all() and (s["dones"][~e0].sum() > 0)
    # episode statistics come from info[:, 14:16] of the finishing step
    assert stats["episodes"] == 3 + 2 + 1 + 1 and abs(stats["mean_return"] - stats["mean_length"]) < 1e-6
    # ring wrap
    collect(env, lambda o: torch.zeros(o.shape[0], 6), buf, 5, obs=obs)
    assert buf.full and buf.pos == 2 and buf.size() == 10


def test_rollout_buffer_gae_matches_loop():
    rng = np.random.default_rng(0)
    T, n = 12, 3
    buf = DeviceRolloutBuffer(T, n, 4, device="cpu", gamma=0.97, gae_lambda=0.9)
    rew = rng.normal(size=(T, n)).astype(np.float32); val = rng.normal(size=(T, n)).astype(np.float32)
    starts = (rng.random((T, n)) < 0.2).astype(np.float32); starts[0] = 1
    for t in range(T):
        buf.add(torch.zeros(n, 4), torch.zeros(n, 6), torch.from_numpy(rew[t]), torch.from_numpy(starts[t]),
                torch.from_numpy(val[t]), torch.zeros(n))
    last_v = rng.normal(size=n).astype(np.float32); last_done = np.array([0, 1, 0], dtype=bool)
    buf.compute_returns_and_advantage(torch.from_numpy(last_v), torch.from_numpy(last_done))
    adv = np.zeros((T, n))
    for e in range(n):
        g = 0.0
        for t in reversed(range(T)):
            nnt = (1.0 - last_done[e]) if t == T - 1 else 1.0 - starts[t + 1, e]
            nv = last_v[e] if t == T - 1 else val[t + 1, e]
            delta = rew[t, e] + 0.97 * nv * nnt - val[t, e]
            g = delta + 0.97 * 0.9 * nnt * g
            adv[t, e] = g
    np.testing.assert_allclose(buf.advantages.numpy(), adv, atol=1e-5)
    np.testing.assert_allclose(buf.returns.numpy(), adv + val, atol=1e-5)
    seen = sum(b["observations"].shape[0] for b in buf.get(batch_size=10))
    assert seen == T * n


def test_collect_on_policy_bootstraps_truncations():
    n = 3
    env = FakeVenv(n)
    buf = DeviceRolloutBuffer(4, n, env.obs_dim, device="cpu", gamma=0.5, gae_lambda=1.0)
    pol = lambda o: (torch.zeros(o.shape[0], 6), torch.zeros(o.shape[0]), torch.zeros(o.shape[0]))
    collect(env, pol, buf, 4, value_fn=lambda o: torch.full((o.shape[0],), 10.0))
    # env 0 is truncated at steps 1 and 3: reward 1 + gamma * V(terminal) = 6 ; env 1 terminates at step 2: plain reward 1
    assert buf.rewards[1, 0] == 6 and buf.rewards[3, 0] == 6 and buf.rewards[2, 1] == 1
    assert buf.episode_starts[0].tolist() == [1, 1, 1] and buf.episode_starts[2, 0] == 1 and buf.episode_starts[3, 1] == 1


@pytest.mark.gpu
def test_collect_from_the_simulator_on_device():
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 512
    venv = TensegrityVecEnv(n, "tensegrity_env-v0", seed=7)
    g = torch.Generator(device=venv.device); g.manual_seed(0)
    pol = lambda o: -0.45 + 0.6 * torch.rand((o.shape[0], 6), device=o.device, generator=g)
    buf = DeviceReplayBuffer(64 * n, n, venv.obs_dim, device=venv.device)
    obs, stats = collect(venv, pol, buf, 64)
    assert buf.size() == 64 and stats["episodes"] > 0 and 5 < stats["mean_length"] < 64
    d = buf.dones.bool()
    assert d.any()
    ti, ei = torch.nonzero(d, as_tuple=True)
    keep = ti < 63
    ti, ei = ti[keep], ei[keep]
    # after a terminal row the env continues from a *reset* observation, which differs from the stored terminal observation
    assert (buf.next_obs[ti, ei] - buf.obs[ti + 1, ei]).abs().max(dim=1).values.min() > 1e-4
    # elsewhere the chain is continuous
    ti2, ei2 = torch.nonzero(~d[:-1], as_tuple=True)
    assert (buf.next_obs[ti2, ei2] - buf.obs[ti2 + 1, ei2]).abs().max() == 0
    s = buf.sample(4096)
    assert torch.isfinite(s["observations"]).all() and s["actions"].min() >= -0.45 and s["actions"].max() <= 0.15
    venv.close()
