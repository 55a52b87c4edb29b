"""CPU study behind the solver settings of the fp32 handles: the host emulation of the row kernels (mixed precision, exactly
the device's arithmetic) against the oracle on identical (state, action), for a list of (round fp64 pose to fp32 at the start of the step?,
tol, iteration cap, rtol), with every env-step outside 1e-4 classified on the emulation's own sub-step trace next to the oracle's
(flip / unconverged / neither) and the solver's cost (iterations an env itself needs; maximum over the 4 rows of a wavefront).
TEST INFRASTRUCTURE (CPU only).  usage: python tests/emul_tolerance_study.py [n] [steps] [flat|uneven] > profiles/round3_tolerance_study.log"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "emul"))
import numpy as np
from oracle import oracle as O
import row_emul as RE
from gpu_common import qfix, rel_env

VARIANTS = [("round 2: fp32 pose between steps, tol 1e-3, cap 8", True, 1e-3, 8, 0.0),
            ("fp64 pose in HBM, tol 1e-3, cap 8", False, 1e-3, 8, 0.0),
            ("... cap 16", False, 1e-3, 16, 0.0),
            ("... tol 5e-4", False, 5e-4, 16, 0.0),
            ("... tol 3e-4", False, 3e-4, 16, 0.0),
            ("... tol 5e-4, rtol 5e-6", False, 5e-4, 16, 5e-6),
            ("... tol 5e-4, rtol 1e-5", False, 5e-4, 16, 1e-5),
            ("... tol 5e-4, rtol 2e-5  (flat default)", False, 5e-4, 16, 2e-5),
            ("... tol 3e-4, rtol 5e-6", False, 3e-4, 16, 5e-6),
            ("... tol 1e-4, rtol 0", False, 1e-4, 16, 0.0),
            ("... tol 2e-4, rtol 5e-6", False, 2e-4, 16, 5e-6),
            ("... tol 1.5e-4, rtol 5e-6", False, 1.5e-4, 16, 5e-6),
            ("... tol 1e-4, rtol 5e-6  (heightfield default)", False, 1e-4, 16, 5e-6),
            ("... tol 1e-4, rtol 1e-5", False, 1e-4, 16, 1e-5),
            ("... tol 2e-4, rtol 1e-5", False, 2e-4, 16, 1e-5),
            ("... tol 1e-4, rtol 2e-6", False, 1e-4, 16, 2e-6),
            ("... tol 5e-5, rtol 2e-6", False, 5e-5, 16, 2e-6),
            ("... tol 5e-5, rtol 5e-6", False, 5e-5, 16, 5e-6),
            ("... tol 3e-5, rtol 1e-6", False, 3e-5, 16, 1e-6)]
if os.environ.get("TGR_STUDY_ONLY"):   # e.g. TGR_STUDY_ONLY="2e-4, rtol 5e-6|1e-4, rtol 2e-6": a subset, by substring of the name
    VARIANTS = [v for v in VARIANTS if any(k in v[0] for k in os.environ["TGR_STUDY_ONLY"].split("|"))]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    uneven = len(sys.argv) > 3 and sys.argv[3] == "uneven"
    O.build(); RE.build()
    RE.lib().tgr_row_emul_set_warm_start(1)
    H = None
    if uneven:
        from tensegrity_rl_amd.vec_env import heights_from_image
        H = heights_from_image(np.load(os.path.join(HERE, "golden", "terrain_100.npy")))
    for kind in ("tensegrity_env", "tr_env"):
        ob = O.OracleBatch(O.make_cfg(kind, "straight", **(dict(heights=H, model_kind=1) if uneven else {})), n, seed=1234); ob.reset()
        rng = np.random.default_rng(0)
        acc = {v[0]: dict(eq=[], ev=[], flip=[], unc=[], own=[]) for v in VARIANTS}
        for t in range(steps):
            q, v, c, w, aux = ob.get_state()
            act0 = ob.last_act.copy() if uneven else None
            a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
            ob.step(a, auto_reset=False)
            qo, vo = ob.get_state()[:2]
            tro = np.array([np.ctypeslib.as_array(ob.env(i).data.trace)[:20].copy() for i in range(n)], dtype=np.uint32)
            ctrl = c + (a.astype(np.float64) - c) * 0.02 if kind == "tr_env" else a.astype(np.float64)
            for name, r32, tol, cap, rt in VARIANTS:
                P = RE.params_from_oracle_model(ob.env(0).model, tol, cap, 6, rtol=rt)
                qi = q.astype(np.float32).astype(np.float64) if r32 else q
                trg = np.zeros((n, 48), dtype=np.uint32)
                q2, v2, _, out = RE.step(P, qi, v, ctrl, act=act0, nsub=20, precision="mixed", trace=trg, model=int(uneven), heights=H)
                A = acc[name]
                A["eq"].append(rel_env(qfix(q2, qo) - qo, qo, 1e-3)); A["ev"].append(rel_env(v2 - vo, vo, 1e-2))
                A["flip"].append(((trg[:, :20] & 0x3fffff) != (tro & 0x3fffff)).any(axis=1))
                A["unc"].append(((trg[:, :20] >> 30) & 1).sum(axis=1))
                A["own"].append(np.diff(np.concatenate([np.zeros((n, 1), int), ((trg[:, :20] >> 22) & 0xff).astype(int)], axis=1), axis=1) % 256)
        print(f"== {kind}, {'heightfield' if uneven else 'flat floor'}: {n} envs x {steps} control periods, mixed-precision emulation vs oracle")
        print("   variant | qpos<1e-4 | qvel<1e-4 | outside: flip / unconverged / neither | worst non-flip | unconverged lane-sub-steps | iterations per sub-step: env's own / max of 4 rows")
        for name, A in acc.items():
            eq, ev, fl, unc = [np.concatenate(A[k]) for k in ("eq", "ev", "flip", "unc")]
            own = np.concatenate(A["own"]); w4 = own.reshape(-1, 4, 20).max(axis=1)
            bad = (ev >= 1e-4) | (eq >= 1e-4); nf = bad & ~fl
            print(f"   {name:52s} | {100*(eq<1e-4).mean():6.2f} % | {100*(ev<1e-4).mean():6.2f} % | {int(bad.sum()):3d}: {int((bad&fl).sum()):3d} / {int((nf&(unc>0)).sum()):2d} / {int((nf&(unc==0)).sum()):2d} | "
                  f"{max(ev[nf].max(), eq[nf].max()) if nf.any() else 0:.1e} | {100*unc.sum()/own.size:.2f} % | {own.mean():.2f} / {w4.mean():.2f}", flush=True)
        ob.close()


if __name__ == "__main__":
    main()
