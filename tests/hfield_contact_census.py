"""Development aid (CPU, oracle): how often does the heightfield contact rule this restatement follows -- a cap keeps its two deepest
facets, a bar's cylinder its deepest grid edge -- leave something out that MuJoCo's one-contact-per-prism rule would report?
Rolls the uneven-ground model out on terrain_image_100's heightfield under random actions with auto-reset and counts, after every control
period, the grid triangles each cap touches and the grid edges each bar's cylinder touches.

    python tests/hfield_contact_census.py [envs] [steps] > profiles/roundN_hfield_contact_census.log"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    img = np.load(os.path.join(ROOT, "tests", "golden", "terrain_100.npy"))
    a = np.asarray(img, dtype=np.float64)
    a = (a - a.min()) / (a.max() - a.min())
    h = np.ascontiguousarray(a[::-1], dtype=np.float32)          # (vec_env.heights_from_image, without importing torch)
    O.build(); L = O.lib()
    for kind in ("tensegrity_env", "tr_env"):
        ob = O.OracleBatch(O.make_cfg(kind, "straight", heights=h, model_kind=1), n, seed=7)
        ob.reset()
        rng = np.random.default_rng(3)
        caps = np.zeros(9, dtype=np.int64); bars = np.zeros(28, dtype=np.int64)
        out = (C.c_int * 9)()
        resets = 0
        for t in range(steps):
            act = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
            _, _, term, trunc, _, _ = ob.step(act, auto_reset=True)
            resets += int(np.sum(term | trunc))
            for i in range(n):
                e = ob.env(i)
                L.tgo_hfield_census(C.byref(e.model), C.byref(e.data), out)
                for k in range(6):
                    caps[min(out[k], 8)] += 1
                for b in range(3):
                    bars[min(out[6 + b], 27)] += 1
        tot_c, tot_b = caps.sum(), bars.sum()
        print(f"{kind}, heightfield, {n} envs x {steps} control periods, random actions, {resets} resets: census after every control period")
        print("  grid triangles touched by a cap (of %d cap samples): " % tot_c + ", ".join(f"{k}: {caps[k]}" for k in range(9) if caps[k]))
        print("    -> caps touching more than two (the rule keeps two): %d = %.2e of cap samples, %.2e of touching caps"
              % (caps[3:].sum(), caps[3:].sum() / tot_c, caps[3:].sum() / max(1, caps[1:].sum())))
        print("  grid edges touched by a bar's cylinder (of %d bar samples): " % tot_b + ", ".join(f"{k}: {bars[k]}" for k in range(28) if bars[k]))
        print("    -> bars touching more than one edge (the rule keeps one): %d = %.2e of bar samples, %.2e of touching bars; bars touching the terrain at all: %.2e"
              % (bars[2:].sum(), bars[2:].sum() / tot_b, bars[2:].sum() / max(1, bars[1:].sum()), bars[1:].sum() / tot_b))
        ob.close()


if __name__ == "__main__":
    main()
