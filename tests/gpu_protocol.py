"""SURVEY.md 8(d) measurement protocol on one GPU: 200 warm-up steps, T timed steps, 5 repeats, median/min/max, for both
env kinds and both precisions, plus the physics-only rate (no resets: auto_reset off, no donor lanes).
usage: python tests/gpu_protocol.py [T]   (default 2000; fp64 uses T/10)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensegrity_rl_amd import TensegrityVecEnv

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((N, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(64)]


def run(env_id, precision, steps, **kw):
    env = TensegrityVecEnv(N, env_id, precision=precision, seed=1234, **kw)
    env.reset_tensors()
    for i in range(200 if precision == "f32" else 20):
        env.step_tensors(acts[i % 64])
    torch.cuda.synchronize()
    rates = []
    s0 = env.stats()
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            env.step_tensors(acts[i % 64])
        torch.cuda.synchronize()
        rates.append(N * steps / (time.perf_counter() - t0))
    s1 = env.stats()
    resets = (s1["donor_resets"] + s1["settle_kernel_resets"] - s0["donor_resets"] - s0["settle_kernel_resets"]) / (5 * steps)
    env.close()
    rates.sort()
    return {"env": env_id, "precision": precision, "steps": steps, "kw": kw, "median": rates[2], "min": rates[0], "max": rates[-1],
            "resets_per_step": resets, "newton_iters_per_substep": s1["newton_iters_per_substep"]}


for env_id in ("tensegrity_env-v0", "tr_env-v0"):
    print(json.dumps(run(env_id, "f32", T)), flush=True)
    print(json.dumps(run(env_id, "f32", T // 4, auto_reset=False, n_donors=0)), flush=True)
print(json.dumps(run("tensegrity_env-v0", "f64", T // 10)), flush=True)
