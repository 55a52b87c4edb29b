"""Evaluation harness on the GPU: the tracking policy shipped with the reference
(models_tracking/SAC_16525000.zip, trained against real MuJoCo on the current XML; actor weights committed as
tests/golden/policy_tracking.npz by oracle/extract_policies.py) is rolled in the batched HIP simulator through
tensegrity_rl_amd.evaluate, the way run.py:103-190 rolls it in MuJoCo."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tracking_policy_rollout_and_logs(golden_dir, tmp_path):
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    from tensegrity_rl_amd import evaluate as E
    n = 256
    venv = TensegrityVecEnv(n, "tr_env-v0", desired_action="tracking", is_test=True, auto_reset=False, seed=1234)
    actor = E.actor_from_npz(os.path.join(golden_dir, "policy_tracking.npz"), device=venv.device)
    assert actor.obs_dim == venv.obs_dim == 48
    logs = E.rollout(venv, actor, 400, deterministic=True)
    T = logs["actions"].shape[0]
    assert logs["actions"].shape == (T, n, 6) and logs["tendon_length"].shape == (T, n, 9) and logs["waypt"].shape == (T, n, 2)
    assert logs["actions"].min() >= -0.45 - 1e-6 and logs["actions"].max() <= 0.15 + 1e-6
    # info["tendon_length"] is what the observation carries when use_obs_noise is off (tr_env.py:506,575)
    assert np.abs(logs["tendon_length"] - logs["observed_tendon_slice"]).max() < 1e-5
    # run.py:144 logs obs[-9:] as "observed tendon length" whatever the layout: for the 48-dim observation that is tendons 3..8 and
    # the tracking tail, reproduced verbatim
    assert np.abs(logs["observed_tendon_length"][:, :, :6] - logs["tendon_length"][:, :, 3:]).max() < 1e-5
    assert (logs["total_bar_contact"] >= 0).all() and np.isfinite(logs["total_bar_contact"]).all()
    # behaviour: is_test puts the waypoint 3 m straight ahead (tr_env.py:833-838); the MuJoCo-trained policy closes in on it
    xy = np.stack([logs["x_pos"], logs["y_pos"]], axis=-1)
    d0 = np.linalg.norm(logs["waypt"][0] - xy[0], axis=-1)
    dmin = np.linalg.norm(logs["waypt"] - xy, axis=-1).min(axis=0)
    assert abs(np.median(d0) - 3.0) < 0.2
    progressed = np.mean(d0 - dmin > 1.0)
    reached = np.mean(dmin < 0.5)
    print("tracking policy: progressed>1m %.2f reached<0.5m %.2f" % (progressed, reached))
    assert progressed > 0.2 and reached > 0.05, (progressed, reached)
    E.save_logs(logs, str(tmp_path))
    for fn in E.LOG_FILES.values():
        assert os.path.exists(os.path.join(str(tmp_path), fn))
    ts = E.tracking_summary(logs)   # run.py --tracking_test view: waypoint on +x at 3 m, robots that walked have x > 1
    assert ts["waypt"].shape == (n, 2) and np.allclose(ts["waypt"][:, 0], 3.0, atol=0.05) and np.abs(ts["waypt"][:, 1]).max() < 1e-6
    assert np.mean(ts["xy_pos"][:, 0] > 1.0) > 0.15
    venv.close()


def test_gym_env_info_carries_reference_keys():
    from tensegrity_rl_amd import make
    env = make("tr_env-v0", desired_action="tracking")
    obs, _ = env.reset()
    obs, r, term, trunc, info = env.step(np.full(6, -0.1, dtype=np.float32))
    for k in ("reward_forward", "reward_ctrl", "reward_survive", "x_position", "y_position", "psi", "distance_from_origin",
              "x_velocity", "y_velocity", "tendon_length", "forward_reward", "waypt", "oripoint", "real_observation"):  # tr_env.py:496-511
        assert k in info, k
    assert np.array_equal(info["real_observation"], obs)  # no observation noise: the same vector (tr_env.py:484-507)
    assert info["tendon_length"].shape == (9,) and np.allclose(info["tendon_length"], obs[36:45], atol=1e-5)
    env.close()


def test_tracking_policy_whole_episode_statistics_regression(golden_dir):
    """REGRESSION test, not parity: the reference statistics were recorded by THIS simulator (tests/gpu_tracking_stats.py on MI355X), so this
    guards against drift of the HIP path's behaviour between builds -- it says nothing about MuJoCo.  What it measures:
    the MuJoCo-trained tracking policy rolled for whole episodes (`_step_num > 1000`, tr_env.py:458) on all six reset poses,
    deterministic and stochastic, against the statistics recorded at 4096 robots (tests/golden/tracking_policy_stats.json, written by
    tests/gpu_tracking_stats.py on MI355X).  Quantities: arrival within 0.5 m of the way-point 3 m ahead, progress of more than a metre,
    and how episodes end (1500 N rule / robot at rest = "unhealthy" / step limit).  This quantifies the behavioural anchor -- it does
    not pin the physics: MuJoCo's own numbers for this policy are inside a pickle the loader rules forbid reading."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_tracking_stats import measure
    ref = json.load(open(os.path.join(golden_dir, "tracking_policy_stats.json")))
    for mode, det in (("deterministic", True), ("stochastic", False)):
        got = measure(1024, det)
        for k, v in ref[mode]["all"].items():
            assert abs(got["all"][k] - v) < 0.06, (mode, k, got["all"][k], v)
        for p in range(6):
            for k in ("arrived_within_0.5m", "progressed_1m", "ended_by_step_limit"):
                assert abs(got["per_pose"][str(p)][k] - ref[mode]["per_pose"][str(p)][k]) < 0.13, (mode, p, k)
        assert abs(got["start_distance_median"] - 3.0) < 0.1
    # what the numbers say: the gait depends on the start pose (poses 0 and 3 arrive 4-5x as often as 1 and 4), the stochastic policy
    # (as in training) keeps moving to the step limit in > 80 % of the episodes, and the 1500 N rule ends < 10 %
    s = ref["stochastic"]["all"]
    assert s["ended_by_step_limit"] > 0.8 and s["ended_by_contact_force"] < 0.1



def test_on_device_ppo_learner_loop_runs_on_the_rollout_buffers():
    """BASELINE configs #3/#4's shape (a learner fed by the batched env) at the level this repository owns: stable_baselines3 is not
    installable here, so a plain-torch PPO update stands in for the learner (the same loop tests/gpu_ppo_demo.py trains with).  Two
    iterations of policy -> step_tensors -> DeviceRolloutBuffer -> GAE -> clipped-surrogate minibatch updates, everything on the device:
    the buffers stay finite, returns follow from rewards, the parameters move, the observation carried from one collect() to the next is the
    env's current one."""
    import torch
    import torch.nn as nn
    from tensegrity_rl_amd import TensegrityVecEnv
    from tensegrity_rl_amd.rollout import DeviceRolloutBuffer, collect
    dev = torch.device("cuda")
    torch.manual_seed(0)
    n, T = 1024, 16
    env = TensegrityVecEnv(n, "tr_env-v0", desired_action="straight", desired_direction=1, seed=1)
    D = env.obs_dim
    pi = nn.Sequential(nn.Linear(D, 64), nn.Tanh(), nn.Linear(64, 6)).to(dev)
    vf = nn.Sequential(nn.Linear(D, 64), nn.Tanh(), nn.Linear(64, 1)).to(dev)
    log_std = nn.Parameter(torch.full((6,), -0.5, device=dev))
    opt = torch.optim.Adam(list(pi.parameters()) + list(vf.parameters()) + [log_std], lr=3e-4)
    raw = torch.zeros(T, n, 6, device=dev)
    buf = DeviceRolloutBuffer(T, n, D, device=dev, gamma=0.99, gae_lambda=0.95)
    obs = env.reset_tensors().clone()
    w0 = pi[0].weight.detach().clone()
    for it in range(2):
        buf.reset()
        k = [0]

        @torch.no_grad()
        def pol(o):
            x = o.float(); mu = pi(x); std = log_std.exp()
            u = mu + std * torch.randn_like(mu)
            raw[k[0]] = u; k[0] += 1
            lp = (-0.5 * ((u - mu) / std) ** 2 - log_std - 0.9189385).sum(1)
            return -0.45 + 0.3 * (u.clamp(-1, 1) + 1), vf(x).squeeze(1), lp
        obs, stats = collect(env, pol, buf, T, obs=obs, value_fn=lambda o: vf(o.float()).squeeze(1).detach())
        assert torch.equal(obs, env.obs)
        assert torch.isfinite(buf.obs).all() and torch.isfinite(buf.rewards).all() and torch.isfinite(buf.advantages).all()
        adv = buf.advantages.reshape(-1); ret = buf.returns.reshape(-1)
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        X, U, LP0 = buf.obs.reshape(-1, D).float(), raw.reshape(-1, 6), buf.log_probs.reshape(-1)
        for s_ in range(0, X.shape[0], 4096):
            idx = slice(s_, s_ + 4096)
            mu = pi(X[idx]); std = log_std.exp()
            lp = (-0.5 * ((U[idx] - mu) / std) ** 2 - log_std - 0.9189385).sum(1)
            ratio = (lp - LP0[idx]).exp()
            loss = -torch.min(ratio * adv[idx], ratio.clamp(0.8, 1.2) * adv[idx]).mean() + 0.25 * (vf(X[idx]).squeeze(1) - ret[idx]).pow(2).mean()
            assert torch.isfinite(loss)
            opt.zero_grad(); loss.backward(); opt.step()
    assert (pi[0].weight.detach() - w0).abs().max() > 0
    env.close()
