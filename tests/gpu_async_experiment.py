"""experiment: does settling work on a second, lower-priority HIP stream leave the live step kernel at its stand-alone rate?
live handle: 4096 envs, no auto-reset, stream A (high priority); background handle: 4352 envs (what the donor lanes of a 4096-env rollout
keep busy), no auto-reset, stream B (low priority); one launch of each per iteration, no cross-stream dependency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensegrity_rl_amd import TensegrityVecEnv
n, nb, steps = 4096, 4352, 300
live = TensegrityVecEnv(n, "tensegrity_env-v0", auto_reset=False, seed=1)
back = TensegrityVecEnv(nb, "tensegrity_env-v0", auto_reset=False, seed=2)
live.reset_tensors(); back.reset_tensors()
g = torch.Generator(device="cuda"); g.manual_seed(0)
al = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(16)]
ab = [torch.full((nb, 6), -0.2, device="cuda") for _ in range(2)]   # constant targets: settling
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
sa = torch.cuda.Stream(priority=-1); sb = torch.cuda.Stream(priority=0)
def run(with_back, same_stream=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(sa):
            live.step_tensors(al[i % 16])
        if with_back:
            with torch.cuda.stream(sa if same_stream else sb):
                back.step_tensors(ab[i % 2])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt / steps * 1e3
for rep in range(2):
    print(f"live alone: {run(False):.3f} ms/step; live + background on a second stream: {run(True):.3f} ms/step; both on one stream: {run(True, True):.3f} ms/step", flush=True)
# and how long the live launches alone take while the background runs (events on stream A)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
torch.cuda.synchronize()
for i in range(steps):
    with torch.cuda.stream(sa):
        ev[i][0].record(sa); live.step_tensors(al[i % 16]); ev[i][1].record(sa)
    with torch.cuda.stream(sb):
        back.step_tensors(ab[i % 2])
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print(f"live launch duration next to the background stream: mean {sum(ms)/len(ms):.3f} ms, median {ms[len(ms)//2]:.3f} ms")
