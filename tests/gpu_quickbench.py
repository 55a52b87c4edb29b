import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
auto = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kw = {}
for a in sys.argv[5:]:
    k_, v_ = a.split('=')
    try:
        kw[k_] = int(v_)
    except ValueError:
        try:
            kw[k_] = float(v_)
        except ValueError:
            kw[k_] = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', v_)) if v_.endswith('.npy') else v_
env = TensegrityVecEnv(n, kind, auto_reset=bool(auto), **kw)
t0 = time.time(); env.reset_tensors(); torch.cuda.synchronize(); t_reset = time.time() - t0
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
for i in range(10): env.step_tensors(acts[i % 16])
torch.cuda.synchronize()
t0 = time.time(); dones = 0
for i in range(steps):
    o, r, te, tr, _, _ = env.step_tensors(acts[i % 16])
    dones += int((te | tr).sum().item()) if i % 10 == 0 else 0
torch.cuda.synchronize(); dt = time.time() - t0
# the dominant kernel alone (the launch that steps the live rows), HIP events recorded by the library on its stream: less noisy than the wall clock
env.time_kernels(128)
for i in range(128): env.step_tensors(acts[i % 16])
kt = sorted(env.kernel_times_ms(128)); env.time_kernels(0)
kern = f"live-row kernel {sum(kt)/len(kt):.4f} ms mean / {kt[len(kt)//2]:.4f} median" if kt else ""
print(f"n={n} {kind} auto={auto} {kw} reset {t_reset*1e3:.1f} ms; {steps} steps {dt*1e3:.1f} ms -> {dt/steps*1e3:.3f} ms/step, {n*steps/dt/1e6:.3f} M env-steps/s; {kern}; dones sampled {dones}; stats {env.stats()}", flush=True)
