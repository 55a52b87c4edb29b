"""soak: long random-action rollouts with auto-reset; counts non-finite terminal observations (a lane whose state blew up is
terminated and restarted by design -- this measures how often that happens) and checks the live observations stay finite"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = 4096
img = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_100.npy"))
for kind, kw in (("tensegrity_env-v0", {}), ("tr_env-v0", {}), ("tr_env-v0", dict(model="uneven", terrain=img)), ("tensegrity_env-v0", dict(model="uneven", terrain=img))):
    env = TensegrityVecEnv(n, kind, seed=3, **kw)
    env.reset_tensors()
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bad = torch.zeros((), device="cuda"); dones = torch.zeros((), device="cuda"); live_bad = torch.zeros((), device="cuda")
    t0 = time.time()
    for t in range(steps):
        a = torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45
        o, r, te, tr, tobs, info = env.step_tensors(a)
        d = (te | tr).bool()
        dones += d.sum()
        bad += (~torch.isfinite(tobs).all(dim=1) & d).sum()
        live_bad += (~torch.isfinite(o).all(dim=1)).sum() + (~torch.isfinite(r)).sum()
    torch.cuda.synchronize()
    print(f"{kind} {('uneven' if kw else 'flat'):6s}: {steps} steps x {n} envs in {time.time()-t0:.1f} s; episodes ended {int(dones.item())}, with non-finite terminal state {int(bad.item())}; "
          f"non-finite live obs/reward {int(live_bad.item())}; stats {env.stats()}", flush=True)
    env.close()
