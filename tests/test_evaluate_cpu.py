"""CPU-side pieces of the evaluation harness (tensegrity_rl_amd/evaluate.py): checkpoint parsing helpers, the actor restatement and the
run.py --tracking_test post-processing (run.py:343-363) on synthetic logs."""
import os

import numpy as np
import torch

from tensegrity_rl_amd import evaluate as E


def test_parse_box_reads_the_plain_text_fields_of_an_sb3_checkpoint():
    np.testing.assert_allclose(E._parse_box("[-0.45 -0.45 -0.45 -0.45 -0.45 -0.45]"), [-0.45] * 6)
    np.testing.assert_allclose(E._parse_box("[0.15 0.15 0.15 0.15 0.15 0.15]"), [0.15] * 6)
    assert np.isinf(E._parse_box("[-inf -inf]")).all()


def test_sac_actor_from_fixture_matches_a_manual_forward_pass(golden_dir):
    g = np.load(os.path.join(golden_dir, "policy_tracking.npz"))
    actor = E.actor_from_npz(os.path.join(golden_dir, "policy_tracking.npz"))
    assert actor.kind == "SAC" and actor.obs_dim == 48
    x = torch.randn(7, 48)
    h = x.numpy().astype(np.float64)
    for k in ("latent_pi_0", "latent_pi_2"):
        h = np.maximum(h @ g[k + "_weight"].astype(np.float64).T + g[k + "_bias"], 0)
    mu = h @ g["mu_weight"].astype(np.float64).T + g["mu_bias"]
    want = -0.45 + 0.5 * (np.tanh(mu) + 1) * 0.6
    np.testing.assert_allclose(actor.act(x, deterministic=True).numpy(), want, atol=1e-5)
    actor.seed(1); a1 = actor.act(x); actor.seed(1); a2 = actor.act(x)   # model.predict's default: a sample, reproducible under a seed
    assert torch.equal(a1, a2) and not torch.allclose(a1, actor.act(x, deterministic=True))
    assert a1.min() >= -0.45 - 1e-6 and a1.max() <= 0.15 + 1e-6


def test_td3_and_ppo_state_dicts_are_recognised():
    sd = {"actor.mu.0.weight": torch.randn(8, 5), "actor.mu.0.bias": torch.zeros(8), "actor.mu.2.weight": torch.randn(6, 8), "actor.mu.2.bias": torch.zeros(6)}
    a = E.actor_from_state_dict(sd, np.full(6, -1.0), np.full(6, 1.0))
    assert a.kind == "TD3" and a.act(torch.randn(3, 5)).shape == (3, 6)
    sd = {"mlp_extractor.policy_net.0.weight": torch.randn(8, 5), "mlp_extractor.policy_net.0.bias": torch.zeros(8),
          "action_net.weight": torch.randn(6, 8), "action_net.bias": torch.zeros(6), "log_std": torch.zeros(6)}
    a = E.actor_from_state_dict(sd, np.full(6, -0.45), np.full(6, 0.15))
    out = a.act(torch.randn(4, 5))
    assert a.kind == "PPO" and out.min() >= -0.45 and out.max() <= 0.15   # PPO clips to the box


def test_tracking_summary_rotates_into_the_waypoint_frame():
    T, n = 40, 3
    done = np.zeros((T, n), bool); done[10:, 1] = True                      # env 1 finishes at step 10: read 5 steps later
    ori = np.tile(np.array([[1.0, 2.0], [0.0, 0.0], [-1.0, 1.0]]), (T, 1, 1))
    way = ori + np.array([[3.0, 0.0], [0.0, 3.0], [-3.0, 0.0]])             # way-points east, north, west of the origins
    t = np.arange(T)[:, None]
    x = ori[..., 0] + 0.05 * t * np.array([1.0, 0.0, -1.0]); y = ori[..., 1] + 0.05 * t * np.array([0.0, 1.0, 0.0])
    logs = {"done": done, "oripoint": ori, "waypt": way, "x_pos": x, "y_pos": y}
    s = E.tracking_summary(logs, stop_after_done=5)
    np.testing.assert_allclose(s["waypt"], [[3, 0]] * 3, atol=1e-12)
    np.testing.assert_allclose(s["xy_pos"][:, 1], 0, atol=1e-12)             # every robot walked straight at its way-point
    np.testing.assert_allclose(s["xy_pos"][:, 0], [0.05 * 39, 0.05 * 15, 0.05 * 39], atol=1e-12)
