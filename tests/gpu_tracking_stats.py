"""The one MuJoCo-trained policy on the current XML (models_tracking/SAC_16525000.zip -> tests/golden/policy_tracking.npz), rolled for
whole episodes (tr_env.py:458: `_step_num > 1000` ends a tracking episode) on N robots, all six reset poses, deterministic and
stochastic: arrival rate, time to arrive and termination causes per pose.  Writes the statistics as JSON (committed as
tests/golden/tracking_policy_stats.json; tests/test_gpu_eval.py re-measures a smaller batch against them).
usage: python tests/gpu_tracking_stats.py [N] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np


def poses_of(seed, n):
    """reset pose index (0..5) of env i in its first episode: the same philox draw the kernels make (oracle restatement of it)"""
    from oracle import oracle as O
    L = O.lib()
    pose, theta = C.c_int(0), C.c_double(0)
    t = (C.c_double * 6)(); u = (C.c_double * 2)()
    out = np.zeros(n, dtype=np.int64)
    for i in range(n):
        L.tgo_rng_draws(seed, i, 0, 1, 0.15, 0.2, -0.45, 0.15, 0.0, 2 * np.pi, C.byref(pose), C.byref(theta), t, u)
        out[i] = pose.value
    return out


def measure(n, deterministic, seed=1234, steps=1010, device=0):
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    from tensegrity_rl_amd import evaluate as E
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    venv = TensegrityVecEnv(n, "tr_env-v0", desired_action="tracking", is_test=True, auto_reset=False, seed=seed)
    actor = E.actor_from_npz(os.path.join(golden, "policy_tracking.npz"), device=venv.device)
    actor.seed(0)
    obs = venv.reset_tensors()
    way0 = None
    first_done = np.full(n, -1); cause = np.zeros(n, dtype=np.int64)  # 1 = contact force > 1500, 2 = came to rest (unhealthy), 3 = 1000-step limit
    arrive = np.full(n, -1); dmin = np.full(n, np.inf); d0 = None
    for t in range(steps):
        a = actor.act(obs, deterministic).contiguous()
        obs, rew, term, trunc, tobs, info = venv.step_tensors(a)
        ih = info.cpu().numpy().astype(np.float64)
        d = np.hypot(ih[:, 10] - ih[:, 3], ih[:, 11] - ih[:, 4])
        if d0 is None:
            d0 = d.copy()
        alive = first_done < 0
        dmin = np.where(alive, np.minimum(dmin, d), dmin)
        arrive = np.where(alive & (arrive < 0) & (d < 0.5), t + 1, arrive)
        te = term.cpu().numpy().astype(bool) & alive
        if te.any():
            slow = (np.abs(ih[:, 7]) < 1e-4) & (np.abs(ih[:, 8]) < 1e-4)
            limit = ih[:, 15] > 1000.5
            c = np.where(limit, 3, np.where(slow, 2, 1))
            cause = np.where(te, c, cause); first_done = np.where(te, t + 1, first_done)
        if (first_done >= 0).all():
            break
    venv.close()
    pose = poses_of(seed, n)
    res = {"n": n, "deterministic": bool(deterministic), "steps": steps, "start_distance_median": float(np.median(d0)), "per_pose": {}}
    for p in range(6):
        m = pose == p
        res["per_pose"][str(p)] = {
            "envs": int(m.sum()), "arrived_within_0.5m": float((arrive[m] > 0).mean()), "progressed_1m": float((d0[m] - dmin[m] > 1.0).mean()),
            "median_steps_to_arrive": float(np.median(arrive[m][arrive[m] > 0])) if (arrive[m] > 0).any() else None,
            "ended_by_contact_force": float((cause[m] == 1).mean()), "ended_at_rest": float((cause[m] == 2).mean()),
            "ended_by_step_limit": float((cause[m] == 3).mean()), "median_episode_length": float(np.median(first_done[m][first_done[m] > 0])) if (first_done[m] > 0).any() else None}
    res["all"] = {"arrived_within_0.5m": float((arrive > 0).mean()), "progressed_1m": float((d0 - dmin > 1.0).mean()),
                  "ended_by_contact_force": float((cause == 1).mean()), "ended_at_rest": float((cause == 2).mean()), "ended_by_step_limit": float((cause == 3).mean())}
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tracking_policy_stats.json"
    res = {"policy": "models_tracking/SAC_16525000.zip (actor weights: tests/golden/policy_tracking.npz)", "env": "tr_env-v0 tracking, is_test (way-point 3 m straight ahead), fp32 row kernels",
           "deterministic": measure(n, True), "stochastic": measure(n, False)}
    json.dump(res, open(out, "w"), indent=1)
    for k in ("deterministic", "stochastic"):
        print(k, json.dumps(res[k]["all"]))
        for p, v in res[k]["per_pose"].items():
            print("   pose", p, json.dumps(v))
