"""development aid: life times of the wavefronts of one step launch (library built with -DTGR_WAVE_TIMES)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensegrity_rl_amd import TensegrityVecEnv, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
env = TensegrityVecEnv(n, kind, auto_reset=True)
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
for i in range(60): env.step_tensors(acts[i % 16])
torch.cuda.synchronize()
lib = N.load()
for rep in range(3):
    env.step_tensors(acts[rep]); torch.cuda.synchronize()
    m = env.n_envs + env.cfg.n_donors if hasattr(env, "n_envs") else None
    nw = 2400
    buf = np.zeros((nw, 4), dtype=np.uint64)
    lib.tgr_debug_wave_times(buf.ctypes.data_as(C.c_void_p), nw)
    t0, t1, it, act = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2].astype(np.int64), buf[:, 3].astype(np.int64)
    ok = (t1 > 0) & (act > 0)
    base = t0[ok].min()
    dur = (t1 - t0)[ok] / 100.0   # 100 MHz counter -> us
    start = (t0[ok] - base) / 100.0
    end = (t1[ok] - base) / 100.0
    print(f"rep {rep}: active waves {ok.sum()} ; kernel span {end.max():.0f} us ; wave life mean {dur.mean():.0f} us pcts(10,50,90,99,100) {np.percentile(dur,[10,50,90,99,100]).round()} ; late starters (>50us) {(start>50).sum()} start max {start.max():.0f}")
    print(f"   iterations/wave mean {it[ok].mean():.1f} pcts {np.percentile(it[ok],[10,50,90,99,100])} ; corr(dur,it) {np.corrcoef(dur, it[ok])[0,1]:.3f}; sum life / (span*2048) = {dur.sum()/(end.max()*2048):.3f}")
    live = ok.copy(); live[n // 4:] = False
    print(f"   live waves: life mean {((t1-t0)[live]/100).mean():.0f} ; donor waves active {ok[n//4:].sum()} life mean {((t1-t0)[ok & ~live]/100).mean():.0f}")
