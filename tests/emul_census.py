"""Operation census of one sub-step of the row kernels from the host emulation (tests/emul/row_emul.cpp built with TGR_ROW_OPCOUNT), and
the USEFUL-LANE fraction of the instruction stream it implies: every lane-value operation of the emulation is one wave-level VALU
instruction on the GPU, executed by all 16 lanes of the row whatever the role of the phase, so

    useful_lane_frac = sum_phase ops(phase) * lanes_that_own_distinct_work(phase) / 16  /  sum_phase ops(phase)

with the per-phase lane counts below (a replicated phase -- all 16 lanes compute the same value -- counts 1 lane).  bench.py multiplies
its measured VALU rate by this number to report an honest useful lane-op rate.  TEST INFRASTRUCTURE (CPU only).
usage: python tests/emul_census.py [env-kind] [n] -> JSON on stdout (committed as profiles/round4_lane_census.json)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "emul"))

# lanes of a 16-lane row that own distinct work in each census phase (tgr_row.h TGR_PHASE / TGR_CENSUS marks)
PHASES = {0: ("kinematics (lane b: bar b)", 3), 1: ("tendons (tendon role; a bar's wrench finished on lane b)", 9),
          2: ("contact detection (slot role: 6 caps, 3 side-to-side slots; the pool's enumeration where a pair has other contacts)", 9),
          3: ("residual + cone (slot role)", 12), 4: ("wrench per slot (slot role)", 12), 12: ("bar sums (replicated)", 1), 16: ("gradient (lane b: bar b)", 3),
          5: ("Hessian rows (dof role; a side-to-side slot's two bodies in the two halves of the row)", 8),
          6: ("direction: block Cholesky + solves (dof role)", 6), 7: ("line search set-up (lane b: bar b; slot role)", 8), 10: ("line search evaluations (slot role)", 12),
          11: ("iterate update (lane b: bar b)", 3), 8: ("last sub-step readouts (replicated)", 1), 13: ("implicitfast set-up (tendon role)", 9),
          14: ("implicitfast matrix (tendon role)", 9), 15: ("implicitfast solve + correction (tendon role / replicated)", 5), 9: ("advance (lane b: bar b)", 3)}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "tensegrity_env"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    so = os.path.join(HERE, "emul", "librow_emul_census.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-DTGR_ROW_OPCOUNT", "-o", so, os.path.join(HERE, "emul", "row_emul.cpp")])
    import row_emul as RE
    from oracle import oracle as O
    RE._lib = C.CDLL(so)
    RE.lib().tgr_row_emul_set_warm_start(1)
    ob = O.OracleBatch(O.make_cfg(kind, "straight"), n, seed=1234); ob.reset()
    P = RE.params_from_oracle_model(ob.env(0).model, 5e-4, 16, 6, rtol=2e-5)
    rng = np.random.default_rng(0)
    cnt = (C.c_longlong * 48)()
    RE.lib().tgr_row_opcount(cnt, 1)
    iters = 0
    steps = 4
    for t in range(steps):
        q, v, c, w, aux = ob.get_state()
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        ctrl = c + (a.astype(np.float64) - c) * 0.02 if kind == "tr_env" else a.astype(np.float64)
        out = RE.step(P, q, v, ctrl, nsub=20, precision="mixed")[3]
        iters += out[:, 23].sum()
        ob.step(a, auto_reset=False)
    RE.lib().tgr_row_opcount(cnt, 0)
    ops = np.array(list(cnt), dtype=np.float64).reshape(2, 24)   # [fp32 | fp64][phase]
    per_sub = ops / (n * steps * 20)
    tot = per_sub.sum()
    useful = sum(per_sub[:, p].sum() * lanes / 16.0 for p, (_, lanes) in PHASES.items())
    res = {"env": kind, "envs": n, "steps": steps, "newton_iterations_per_substep_per_env": iters / (n * steps * 20),
           "ops_per_substep": tot, "fp64_share": per_sub[1].sum() / tot, "useful_lane_frac": useful / tot,
           "phases": {str(p): {"what": w, "lanes": l, "fp32_ops": per_sub[0, p], "fp64_ops": per_sub[1, p]} for p, (w, l) in sorted(PHASES.items())},
           "note": "one environment alone per emulated row: iterations are the env's own, not the maximum over the 4 rows of a wavefront"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
