"""env-steps/s through the numpy-facing SB3 surface: TensegrityVecEnv.step(actions) with info_mode="minimal" -- what SB3's
collect_rollouts drives (run.py:41-101 -> VecEnv.step_async / step_wait), next to the device-tensor rate of the same env.
usage: python tests/gpu_vecenv_bench.py [n] [env-id] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv


def measure(n=4096, env_id="tensegrity_env-v0", steps=300, info_mode="minimal", donors=None):
    env = TensegrityVecEnv(n, env_id, seed=1234, info_mode=info_mode, n_donors=donors if donors is not None else 5 * n // 4)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32) for _ in range(16)]
    for i in range(30):
        env.step(acts[i % 16])
    t0 = time.perf_counter(); dones = 0
    keys = ("x_position", "y_position", "x_velocity", "reward_forward") + (("tendon_length", "real_observation") if env_id.startswith("tr_env") else ())
    sink = 0.0
    for i in range(steps):
        obs, rew, done, infos = env.step(acts[i % 16])
        dones += int(done.sum())
        if info_mode == "full":  # what run.py:139-161 does with the info of its one env, every step
            for k in keys:
                sink += float(np.sum(infos[0][k]))
    dt = time.perf_counter() - t0
    dacts = [torch.as_tensor(a, device=env.device) for a in acts]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for i in range(steps):
        env.step_tensors(dacts[i % 16])
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
    env.close()
    return {"numpy_step_env_steps_per_s": n * steps / dt, "numpy_ms_per_step": dt / steps * 1e3, "tensor_env_steps_per_s": n * steps / dt2,
            "dones_per_step": dones / steps, "info_mode": info_mode, "n": n, "env_id": env_id}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    env_id = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    for mode in ("minimal", "full"):
        print(measure(n, env_id, steps, mode), flush=True)
