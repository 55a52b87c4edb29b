"""Evaluation harness: roll a shipped Stable-Baselines3 policy in the batched HIP simulator and write the logs the
reference's `run.py --test` writes (run.py:103-190), without needing stable_baselines3.

* `load_sb3_actor(zip_path)` reads `policy.pth` out of an SB3 checkpoint with `torch.load(weights_only=True)` (tensors
  only, nothing in the file is executed) and the action box out of the checkpoint's JSON `data` member (plain-text
  fields `low`/`high`; the pickled members are never touched).  Supported actors: SAC (`actor.latent_pi/mu/log_std`),
  TD3 (`actor.mu`), PPO/A2C (`mlp_extractor.policy_net` + `action_net`).
* `Actor.act(obs)` is what `model.predict(obs)` computes for those policies: SB3's default `deterministic=False`
  (run.py:136 uses the default), squashing + un-scaling to the action box for SAC/TD3, clipping for PPO/A2C.
* `rollout(venv, actor, steps)` steps all envs on the GPU (actor and simulator exchange device tensors only) and returns
  the reference's log arrays with an extra env axis; `save_logs` writes the same file names as run.py:178-188.
"""
import io
import json
import os
import re
import zipfile

import numpy as np

from . import _native as N

LOG_FILES = {"actions": "action_data.npy", "tendon_length": "tendon_data.npy",
             "observed_tendon_length": "observed_tendon_data.npy", "cap_posi": "cap_posi_data.npy",
             "observed_cap_posi": "observed_cap_posi_data.npy", "total_bar_contact": "total_bar_contact_data.npy",
             "reward_forward": "reward_forward_data.npy", "reward_ctrl": "reward_ctrl_data.npy",
             "waypt": "waypt_data.npy", "x_pos": "x_pos_data.npy", "y_pos": "y_pos_data.npy"}


def _parse_box(txt):
    return np.array([float(v) for v in re.findall(r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|inf)", txt)], dtype=np.float32)


class Actor:
    """Plain-torch restatement of an SB3 actor network (weights are data; the architecture is SB3's documented default:
    Linear/ReLU stacks for SAC/TD3, Linear/Tanh for PPO/A2C)."""

    def __init__(self, kind, layers, head, log_std, low, high, device="cpu"):
        import torch
        self.torch = torch
        self.kind = kind
        self.device = torch.device(device)
        f = lambda w: w.to(device=self.device, dtype=torch.float32)
        self.layers = [(f(w), f(b)) for w, b in layers]
        self.head = (f(head[0]), f(head[1]))
        self.log_std = None if log_std is None else (tuple(f(x) for x in log_std) if isinstance(log_std, tuple) else f(log_std))
        self.low = torch.as_tensor(low, dtype=torch.float32, device=self.device)
        self.high = torch.as_tensor(high, dtype=torch.float32, device=self.device)
        self.obs_dim = int(self.layers[0][0].shape[1]) if self.layers else int(self.head[0].shape[1])
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(0)

    def to(self, device):
        return Actor(self.kind, self.layers, self.head, self.log_std, self.low, self.high, device)

    def seed(self, s):
        self.generator.manual_seed(int(s))

    def act(self, obs, deterministic=False):
        t = self.torch
        h = obs.to(device=self.device, dtype=t.float32)
        nl = t.tanh if self.kind in ("PPO", "A2C") else t.relu
        for w, b in self.layers:
            h = nl(h @ w.T + b)
        mu = h @ self.head[0].T + self.head[1]
        if self.kind == "SAC":
            if not deterministic:
                ls = (h @ self.log_std[0].T + self.log_std[1]).clamp(-20.0, 2.0)
                mu = mu + ls.exp() * t.randn(mu.shape, generator=self.generator, device=self.device)
            a = t.tanh(mu)
            return self.low + 0.5 * (a + 1.0) * (self.high - self.low)
        if self.kind == "TD3":
            return self.low + 0.5 * (t.tanh(mu) + 1.0) * (self.high - self.low)
        if not deterministic:
            mu = mu + self.log_std.exp() * t.randn(mu.shape, generator=self.generator, device=self.device)
        return t.minimum(t.maximum(mu, self.low), self.high)


def actor_from_state_dict(sd, low, high, device="cpu"):
    keys = list(sd)
    def stack(prefix):
        idx = sorted({int(k[len(prefix):].split(".")[0]) for k in keys if k.startswith(prefix) and k.endswith(".weight")})
        return [(sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"]) for i in idx]
    if "actor.log_std.weight" in sd:
        return Actor("SAC", stack("actor.latent_pi."), (sd["actor.mu.weight"], sd["actor.mu.bias"]),
                     (sd["actor.log_std.weight"], sd["actor.log_std.bias"]), low, high, device)
    if any(k.startswith("actor.mu.") for k in keys):
        layers = stack("actor.mu.")
        return Actor("TD3", layers[:-1], layers[-1], None, low, high, device)
    if "action_net.weight" in sd:
        return Actor("PPO", stack("mlp_extractor.policy_net."), (sd["action_net.weight"], sd["action_net.bias"]),
                     sd["log_std"], low, high, device)
    raise ValueError("unrecognised SB3 policy state_dict")


def load_sb3_actor(zip_path, device="cpu", action_box=None):
    """SB3 checkpoint -> Actor.  `action_box=(low, high)` overrides the box stored in the checkpoint."""
    import torch
    with zipfile.ZipFile(zip_path) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        meta = json.loads(z.read("data"))
    if action_box is None:
        sp = meta["action_space"]
        action_box = (_parse_box(sp["low"]), _parse_box(sp["high"]))
    return actor_from_state_dict(sd, action_box[0], action_box[1], device)


def actor_from_npz(path, device="cpu"):
    """Actor from the small weight fixtures under tests/golden/policy_*.npz."""
    import torch
    g = np.load(path)
    sd = {}
    for k in g.files:
        if k in ("low", "high"):
            continue
        parts = k.split("_")          # latent_pi_0_weight / mu_weight / log_std_weight
        name = ".".join(["actor", "_".join(parts[:-2]) if parts[-2].isdigit() else "_".join(parts[:-1])] +
                        ([parts[-2]] if parts[-2].isdigit() else []) + [parts[-1]])
        sd[name] = torch.from_numpy(np.asarray(g[k], dtype=np.float32))
    if "actor.log_std.weight" not in sd:  # older fixtures carry the mean head only -> deterministic use
        sd["actor.log_std.weight"] = torch.zeros_like(sd["actor.mu.weight"])
        sd["actor.log_std.bias"] = torch.full_like(sd["actor.mu.bias"], -20.0)
    low = np.full(6, float(g["low"]), dtype=np.float32)
    high = np.full(6, float(g["high"]), dtype=np.float32)
    return actor_from_state_dict(sd, low, high, device)


def rollout(venv, actor, steps, deterministic=False, stop_after_done=500):
    """run.py:103-176 for all envs of `venv` at once.  The env must have been built with auto_reset=False to mirror the
    reference loop (which keeps stepping a finished episode for up to 500 more steps); with auto_reset=True finished envs
    restart as under SB3.  Returns {name: array[T, N, ...]} plus "done"."""
    t = venv.torch
    n, D = venv.num_envs, venv.obs_dim
    tr = venv.env_id.startswith("tr_env")
    obs = venv.reset_tensors()
    logs = {k: [] for k in LOG_FILES}
    logs["done"] = []
    logs["oripoint"] = []
    logs["observed_tendon_slice"] = []  # not a run.py file: the nine tendon entries of the observation
    extra = np.full(n, stop_after_done, dtype=np.int64)
    for _ in range(int(steps)):
        a = actor.act(obs, deterministic).contiguous()
        obs, rew, term, trunc, tobs, info = venv.step_tensors(a)
        ih = info.cpu().numpy().astype(np.float64)
        oh = obs.cpu().numpy().astype(np.float64)
        logs["actions"].append(a.cpu().numpy())
        logs["tendon_length"].append(ih[:, N.INFO_TENDON_LENGTH:N.INFO_TENDON_LENGTH + 9])
        # run.py:144-146 verbatim: obs[-9:] and obs[:18] whatever the layout (for the 48-dim tracking / aiming / vel_track observation
        # obs[-9:] is NOT the nine tendon lengths -- the reference logs it that way; the tendon slice goes to its own key)
        logs["observed_tendon_length"].append(oh[:, -9:])
        logs["observed_cap_posi"].append(oh[:, :18])
        real = venv.real_obs.cpu().numpy().astype(np.float64) if getattr(venv, "real_obs", None) is not None else oh
        logs["cap_posi"].append(real[:, :18])  # info["real_observation"][:18] (run.py:145)
        if tr:  # tr_env observation: 18 cap positions first, tendons right after the (optional) 18 cap velocities
            nt = 36 if venv.cfg.use_cap_velocity else 18
            logs["observed_tendon_slice"].append(oh[:, nt:nt + 9])
        else:
            logs["observed_tendon_slice"].append(oh[:, -9:])
        logs["total_bar_contact"].append(ih[:, N.INFO_BAR_CONTACT])
        logs["reward_forward"].append(ih[:, 0])
        logs["reward_ctrl"].append(ih[:, 1])
        logs["waypt"].append(ih[:, 10:12])
        logs["oripoint"].append(ih[:, 12:14])
        logs["x_pos"].append(ih[:, 3])
        logs["y_pos"].append(ih[:, 4])
        done = (term.cpu().numpy() | trunc.cpu().numpy()).astype(bool)
        logs["done"].append(done)
        extra -= done
        if not venv.cfg.auto_reset and (extra < 0).all():
            break
    return {k: np.stack(v) for k, v in logs.items()}


def tracking_summary(logs, stop_after_done=500):
    """run.py:310-363 (`--tracking_test`): one episode per env; the final waypoint and robot position relative to the
    episode's origin, rotated so that the waypoint lies on +x.  Each env is read at the step the reference loop would have
    stopped at (500 steps after its first `done`, or the end of the rollout)."""
    done = logs["done"]
    T, n = done.shape
    first = np.where(done.any(axis=0), done.argmax(axis=0), T)
    stop = np.minimum(first + stop_after_done, T - 1)
    idx = (stop, np.arange(n))
    ori = logs["oripoint"][idx]
    way = logs["waypt"][idx] - ori
    xy = np.stack([logs["x_pos"][idx], logs["y_pos"][idx]], axis=-1) - ori
    ang = np.arctan2(way[:, 1], way[:, 0])
    c, s_ = np.cos(ang), np.sin(ang)
    rot = lambda v: np.stack([c * v[:, 0] + s_ * v[:, 1], s_ * v[:, 0] - c * v[:, 1]], axis=-1)  # run.py:355
    return {"waypt": rot(way), "xy_pos": rot(xy), "oripoint": np.zeros_like(ori)}


def save_tracking(summary, saved_data_dir):
    os.makedirs(saved_data_dir, exist_ok=True)
    for k, fn in (("waypt", "waypt_data.npy"), ("xy_pos", "xy_pos_data.npy"), ("oripoint", "oripoint_data.npy")):
        np.save(os.path.join(saved_data_dir, fn), summary[k])


def save_logs(logs, saved_data_dir):
    """Same file names as run.py:178-188.  With one env the arrays have the reference's shapes ([T, ...])."""
    os.makedirs(saved_data_dir, exist_ok=True)
    for k, fn in LOG_FILES.items():
        a = logs[k]
        if a.shape[1] == 1:
            a = a[:, 0]
        np.save(os.path.join(saved_data_dir, fn), a)


def main(argv=None):
    import argparse
    from .vec_env import TensegrityVecEnv
    p = argparse.ArgumentParser(description="roll an SB3 checkpoint in the HIP simulator (run.py --test)")
    p.add_argument("--test", metavar="path_to_model")
    p.add_argument("--tracking_test", metavar="path_to_model", help="run.py --tracking_test: --num_envs episodes at once")
    p.add_argument("--env_id", default="tr_env-v0")
    p.add_argument("--desired_action", default="straight", choices=list(N.ACTIONS))
    p.add_argument("--desired_direction", default=1, type=int, choices=[-1, 1])
    p.add_argument("--terminate_when_unhealthy", default="yes", choices=["yes", "no"])
    p.add_argument("--saved_data_dir", default="saved_data")
    p.add_argument("--simulation_seconds", default=30, type=int)
    p.add_argument("--num_envs", default=1, type=int)
    p.add_argument("--deterministic", action="store_true")
    p.add_argument("--seed", default=1234, type=int)
    a = p.parse_args(argv)
    if not (a.test or a.tracking_test):
        p.error("one of --test / --tracking_test is required")
    if a.tracking_test:
        a.test, a.desired_action = a.tracking_test, "tracking"
    venv = TensegrityVecEnv(a.num_envs, a.env_id, desired_action=a.desired_action, desired_direction=a.desired_direction,
                            terminate_when_unhealthy=a.terminate_when_unhealthy == "yes", is_test=True, auto_reset=False,
                            seed=a.seed)
    actor = load_sb3_actor(a.test, device=venv.device)
    if actor.obs_dim != venv.obs_dim:
        raise SystemExit(f"checkpoint expects {actor.obs_dim}-dim observations, the env produces {venv.obs_dim}")
    logs = rollout(venv, actor, int(a.simulation_seconds / venv.dt), deterministic=a.deterministic)
    if a.tracking_test:
        save_tracking(tracking_summary(logs), a.saved_data_dir)
    else:
        save_logs(logs, a.saved_data_dir)
    print(json.dumps({"steps": int(logs["actions"].shape[0]), "envs": a.num_envs,
                      "final_xy_mean": [float(logs["x_pos"][-1].mean()), float(logs["y_pos"][-1].mean())],
                      "terminated_envs": int(logs["done"].any(axis=0).sum())}))


if __name__ == "__main__":
    main()
