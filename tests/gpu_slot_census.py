"""Which contact slots are in use, per environment and per WAVEFRONT (4 consecutive envs), on the headline workload -- from the device's own
sub-step trace (tgr_bind_substep_trace).  Development aid: says what the Newton loop's per-slot work (Hessian rank updates, skipped per
wavefront on a ballot) is actually spent on.  usage: gpu_slot_census.py [n] [env id] [warm-up steps] [sampled steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tensegrity_rl_amd import TensegrityVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "tensegrity_env-v0"
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 150
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
env = TensegrityVecEnv(n, kind, auto_reset=True)
env.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
for i in range(warm):
    env.step_tensors(acts[i % 16])
trace = env.bind_substep_trace()
masks, iters = [], []
for i in range(steps):
    env.step_tensors(acts[i % 16])
    w = trace.cpu().numpy().astype(np.uint32)[:, :20]
    masks.append(w & 0xffff)
    own = (w >> 22) & 0xff
    iters.append(np.diff(np.concatenate([np.zeros((n, 1), np.int64), own.astype(np.int64)], axis=1), axis=1) % 256)
m = np.stack(masks)              # [steps, n, 20]
it = np.stack(iters)
bits = ((m[..., None] >> np.arange(16)) & 1).astype(bool)   # [steps, n, 20, 16]
wave = bits.reshape(steps, n // 4, 4, 20, 16).any(axis=2)   # union over the 4 envs of a wavefront
names = [f"cap{c}" for c in range(6)] + [x for p in range(3) for x in (f"side{p}", f"pool{p}")] + [f"pool{3 + p}" for p in range(4)]
print("slot      env-level  wavefront-level (fraction of sub-steps the slot is active)")
for k in range(16):
    print(f"  {names[k]:8s}  {bits[..., k].mean():.4f}     {wave[..., k].mean():.4f}")
print("active slots per env-sub-step: mean %.2f; per wavefront (union): mean %.2f" % (bits.sum(-1).mean(), wave.sum(-1).mean()))
print("caps per env: mean %.2f, per wavefront %.2f; pair slots (6..15) per env %.3f, per wavefront %.3f" %
      (bits[..., :6].sum(-1).mean(), wave[..., :6].sum(-1).mean(), bits[..., 6:].sum(-1).mean(), wave[..., 6:].sum(-1).mean()))
for b in range(3):
    both_w = (wave[..., 2 * b] & wave[..., 2 * b + 1]).mean()
    both_e = (bits[..., 2 * b] & bits[..., 2 * b + 1]).mean()
    print(f"  bar {b}: both caps active -- env {both_e:.4f}, wavefront {both_w:.4f}")
print("wavefronts with any pair slot: %.4f; side-to-side slots (c01 %.4f c02 %.4f c12 %.4f)" % (
    wave[..., 6:].any(-1).mean(), wave[..., 6].mean(), wave[..., 8].mean(), wave[..., 10].mean()))
hist = np.bincount(wave.sum(-1).ravel(), minlength=17)
print("wavefront union size histogram:", (hist / hist.sum()).round(4).tolist())
own = it
print("Newton iterations per env-sub-step: mean %.3f; histogram %s" % (own.mean(), (np.bincount(np.minimum(own.ravel(), 12), minlength=13) / own.size).round(4).tolist()))
wmax = own.reshape(steps, n // 4, 4, 20).max(axis=2)
print("max over the 4 envs of a wavefront: mean %.3f; histogram %s" % (wmax.mean(), (np.bincount(np.minimum(wmax.ravel(), 12), minlength=13) / wmax.size).round(4).tolist()))


# what re-grouping the envs into wavefronts would buy: sum over wavefronts and sub-steps of the max over the wavefront's 4 envs, for the
# grouping as it is, for envs sorted by THIS step's iteration total (an upper bound no predictor reaches), by the previous step's total, by
# the iterations of the previous step's last 5 sub-steps, and by the number of contacts the previous step ended with
def wave_cost(order, t):
    return own[t][order].reshape(n // 4, 4, 20).max(axis=1).mean()


tot = own.sum(-1)                       # [steps, n]
last5 = own[:, :, 15:].sum(-1)
nact_end = bits[:, :, 19, :].sum(-1)    # contacts active in the last sub-step
res = {"as is": [], "sorted by this step's total (bound)": [], "by previous step's total": [], "by previous step's last 5 sub-steps": [], "by contacts at the end of the previous step": [],
       "by (contacts at end, last 5)": []}
for t in range(1, steps):
    ident = np.arange(n)
    res["as is"].append(wave_cost(ident, t))
    res["sorted by this step's total (bound)"].append(wave_cost(np.argsort(tot[t], kind="stable"), t))
    res["by previous step's total"].append(wave_cost(np.argsort(tot[t - 1], kind="stable"), t))
    res["by previous step's last 5 sub-steps"].append(wave_cost(np.argsort(last5[t - 1], kind="stable"), t))
    res["by contacts at the end of the previous step"].append(wave_cost(np.argsort(nact_end[t - 1], kind="stable"), t))
    res["by (contacts at end, last 5)"].append(wave_cost(np.lexsort((last5[t - 1], nact_end[t - 1])), t))
print("wavefront-level iterations per sub-step under a re-grouping of the envs (note: a done env restarts from a donor, which the predictors do not see):")
for k, v in res.items():
    print(f"  {k:48s} {np.mean(v):.3f}")
print("correlation of an env's iteration total with its previous step's: %.3f" % np.corrcoef(tot[1:].ravel(), tot[:-1].ravel())[0, 1])
