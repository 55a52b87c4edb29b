"""Which fp32 quantity of the row kernels' sub-step limits the accuracy?  CPU study behind DESIGN.md 4.1 ("Newton on e = d - d_ref").
The fp64 instantiation of the host emulation is run with ONE quantity at a time rounded to fp32 where the mixed-precision kernel stores or
forms it in fp32, and compared with the unrounded fp64 run after `nsub` sub-steps: error of the bars' spin about their own axes (the
component that decided the heightfield tr_env's outliers), share of lane-sub-steps the solver leaves unconverged at a tight tolerance.
The probes are patched into a temporary COPY of tgr_row.h / tgr_spmd.h (exact-string replacements, each asserted to match once), so the
product headers carry nothing of this.  TEST INFRASTRUCTURE (CPU only).
usage: python tests/emul_rounding_probe.py [n] [nsub] [shift|noshift] [tr_env|tensegrity_env] [uneven|flat]"""
import ctypes as C, os, shutil, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "emul"))
import numpy as np

NAMES = ["F copy of the bar axes", "contact n / rA / rB", "contact c0 / Dn", "residual at the start", "the iterate (d, or e with the shift)", "jar", "cone force",
         "bar sums", "Newton direction", "velocity store", "gradient", "slot torque", "warm start / d_ref", "I d_ref", "M d", "final d", "line-search slope"]

SPMD = [("// optional operation census of the emulation",
         "inline int& probe_mask() { static int m = 0; return m; }\n"
         "template <typename T> inline Ln<T> probe(int bit, Ln<T> x) { if ((probe_mask() >> bit) & 1) for (int i = 0; i < ROW; i++) x.v[i] = (T)(float)x.v[i]; return x; }\n"
         "// optional operation census of the emulation")]
P3 = "template <typename F> inline V3<F> probe3(int bit, V3<F> v) { v.x = probe(bit, v.x); v.y = probe(bit, v.y); v.z = probe(bit, v.z); return v; }\n"
ROWH = [
    ("namespace row {", "namespace row {\n" + P3),
    ("st3(M.fd, FEZ + 3 * b, cvt3<F>(kz));", "st3(M.fd, FEZ + 3 * b, probe3(0, cvt3<F>(kz)));"),
    ("      c.active = sel(is_floor, fhit,", "      c.n = probe3(1, c.n); c.rA = probe3(1, c.rA); c.rB = probe3(1, c.rB); c.c0 = probe(2, c.c0); c.Dn = probe(2, c.Dn);\n      c.active = sel(is_floor, fhit,"),
    ("      ja0 = yaB - yaA;", "      ja0 = yaB - yaA;\n      jl0 = probe3(3, jl0); ja0 = probe3(3, ja0);"),
    ("      st3(M.cd, CJL0, cvt3<F>((ylB + cross(yaB, cvt3<K>(c.rB))) - (ylA + cross(yaA, cvt3<K>(c.rA))) + cvt<K>(c.c0) * cvt3<K>(c.n)));",
     "      st3(M.cd, CJL0, probe3(3, cvt3<F>((ylB + cross(yaB, cvt3<K>(c.rB))) - (ylA + cross(yaA, cvt3<K>(c.rA))) + cvt<K>(c.c0) * cvt3<K>(c.n))));"),
    ("      st3(M.fd, FDL + 3 * b, ld3(M.fd, FDL + 3 * b) + alpha * pl[b]);", "      st3(M.fd, FDL + 3 * b, probe3(4, ld3(M.fd, FDL + 3 * b) + alpha * pl[b]));"),
    ("      st3(M.fd, FDA + 3 * b, ld3(M.fd, FDA + 3 * b) + alpha * pa[b]);", "      st3(M.fd, FDA + 3 * b, probe3(4, ld3(M.fd, FDA + 3 * b) + alpha * pa[b]));"),
    ("    st3(M.cd, CDLA, ld3(M.cd, CDLA) + alpha * plA); st3(M.cd, CDAA, ld3(M.cd, CDAA) + alpha * paA);",
     "    st3(M.cd, CDLA, probe3(4, ld3(M.cd, CDLA) + alpha * plA)); st3(M.cd, CDAA, probe3(4, ld3(M.cd, CDAA) + alpha * paA));"),
    ("    st3(M.cd, CDLB, ld3(M.cd, CDLB) + alpha * plB); st3(M.cd, CDAB, ld3(M.cd, CDAB) + alpha * paB);",
     "    st3(M.cd, CDLB, probe3(4, ld3(M.cd, CDLB) + alpha * plB)); st3(M.cd, CDAB, probe3(4, ld3(M.cd, CDAB) + alpha * paB));"),
    ("      dl[b] = P.warm_start ? sel3(any_contact, cl + wx * (cl - ql_), z3) : z3;", "      dl[b] = probe3(12, P.warm_start ? sel3(any_contact, cl + wx * (cl - ql_), z3) : z3);"),
    ("      da[b] = P.warm_start ? sel3(any_contact, ca + wx * (ca - qa_), z3) : z3;", "      da[b] = probe3(12, P.warm_start ? sel3(any_contact, ca + wx * (ca - qa_), z3) : z3);"),
    ("        st3(M.fd, FGA + 3 * b, cvt3<F>(PK.Ixx * kda + ((PK.Izz - PK.Ixx) * dot(kez, kda)) * kez));",
     "        st3(M.fd, FGA + 3 * b, probe3(13, cvt3<F>(PK.Ixx * kda + ((PK.Izz - PK.Ixx) * dot(kez, kda)) * kez)));"),
    ("    const V3<F> jl = ld3(M.cd, CJL0) + ((ld3(M.cd, CDLB) + cross(ld3(M.cd, CDAB), cs.rB)) - (ld3(M.cd, CDLA) + cross(ld3(M.cd, CDAA), cs.rA)));",
     "    const V3<F> jl = probe3(5, ld3(M.cd, CJL0) + ((ld3(M.cd, CDLB) + cross(ld3(M.cd, CDAB), cs.rB)) - (ld3(M.cd, CDLA) + cross(ld3(M.cd, CDAA), cs.rA))));"),
    ("    const V3<F> ja = ld3(M.cd, CJA0) + (ld3(M.cd, CDAB) - ld3(M.cd, CDAA));", "    const V3<F> ja = probe3(5, ld3(M.cd, CJA0) + (ld3(M.cd, CDAB) - ld3(M.cd, CDAA)));"),
    ("    const RCone<F> o = cone_eval(P, cs, jl, ja);", "    RCone<F> o = cone_eval(P, cs, jl, ja);\n    o.fl = probe3(6, o.fl); o.fa = probe3(6, o.fa);"),
    ("      st3(M.fd, FFCL + 3 * b, fcl); st3(M.fd, FFCA + 3 * b, fca); M.kd.st(KTAX + b, tax);",
     "      fcl = probe3(7, fcl); fca = probe3(7, fca);\n      st3(M.fd, FFCL + 3 * b, fcl); st3(M.fd, FFCA + 3 * b, fca); M.kd.st(KTAX + b, tax);"),
    ("    const F d0 = sel(conv, zero, -gHg);", "    for (int b = 0; b < 3; b++) for (int i = 0; i < 6; i++) pr[b][i] = probe(8, pr[b][i]);\n    const F d0 = sel(conv, zero, -gHg);"),
    ("    st3(M.fd, FV + 3 * b, cvt3<F>(vk));", "    st3(M.fd, FV + 3 * b, probe3(9, cvt3<F>(vk)));"),
    ("    st3(M.fd, FW + 3 * b, cvt3<F>(wk));", "    st3(M.fd, FW + 3 * b, probe3(9, cvt3<F>(wk)));"),
    ("      const V3<F> gl = P.mass * fl_ - fcl;", "      const V3<F> gl = probe3(10, P.mass * probe3(14, fl_) - fcl);"),
    ("      V3<F> ga = Ida - fca;", "      V3<F> ga = probe3(10, probe3(14, Ida) - fca);"),
    ("    V3<F> fB = o.fl, tB = cross(cs.rB, o.fl) + o.fa;", "    V3<F> fB = o.fl, tB = probe3(11, cross(cs.rB, o.fl) + o.fa);"),
    ("      st3(M.fd, FDL + 3 * b, ld3(M.fd, FRL + 3 * b) + ld3(M.fd, FDL + 3 * b));", "      st3(M.fd, FDL + 3 * b, probe3(15, ld3(M.fd, FRL + 3 * b) + ld3(M.fd, FDL + 3 * b)));"),
    ("      st3(M.fd, FDA + 3 * b, ld3(M.fd, FRA + 3 * b) + ld3(M.fd, FDA + 3 * b));", "      st3(M.fd, FDA + 3 * b, probe3(15, ld3(M.fd, FRA + 3 * b) + ld3(M.fd, FDA + 3 * b)));"),
    ("      const F d1 = g1 + alpha * g2 + rowsum(s1), d2 = g2 + rowsum(s2);",
     "      const F d1 = probe(16, probe(16, g1) + alpha * probe(16, g2) + probe(16, rowsum(s1))), d2 = g2 + rowsum(s2);"),
]
NOSHIFT = [("  constexpr bool SHIFT = MODEL != 0;", "  constexpr bool SHIFT = false;")]


def patched_library(tmp, shift):
    for rel in ("tests/emul/row_emul.cpp", "tensegrity_rl_amd/csrc/tgr_row.h", "tensegrity_rl_amd/csrc/tgr_spmd.h"):
        os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), os.path.join(tmp, rel))

    def patch(rel, reps, first_only=()):
        p = os.path.join(tmp, rel); s = open(p).read()
        for a, b in reps:
            assert s.count(a) == 1 or a in first_only, (rel, s.count(a), a)
            s = s.replace(a, b, 1)
        open(p, "w").write(s)
    patch("tensegrity_rl_amd/csrc/tgr_spmd.h", SPMD, first_only=(SPMD[0][0],))
    patch("tensegrity_rl_amd/csrc/tgr_row.h", ROWH + ([] if shift else NOSHIFT), first_only=("namespace row {",))
    patch("tests/emul/row_emul.cpp", [('extern "C" void tgr_row_emul_set_warm_start(int w) { g_warm_start = w; }',
                                       'extern "C" void tgr_row_emul_set_warm_start(int w) { g_warm_start = w; }\nextern "C" void tgr_probe_set(int m) { tgr::row::probe_mask() = m; }')])
    so = os.path.join(tmp, "librow_probe.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, os.path.join(tmp, "tests/emul/row_emul.cpp")])
    return so


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    shift = not (len(sys.argv) > 3 and sys.argv[3] == "noshift")
    kind = sys.argv[4] if len(sys.argv) > 4 else "tr_env"
    uneven = not (len(sys.argv) > 5 and sys.argv[5] == "flat")
    from oracle import oracle as O
    import row_emul as RE
    from tensegrity_rl_amd.vec_env import heights_from_image
    with tempfile.TemporaryDirectory() as tmp:
        so = patched_library(tmp, shift)
        RE.SO = so; RE.build = lambda force=False: so; RE._lib = None
        L = RE.lib()
        O.build(); L.tgr_row_emul_set_warm_start(1)
        H = heights_from_image(np.load(os.path.join(HERE, "golden", "terrain_100.npy"))) if uneven else None
        ob = O.OracleBatch(O.make_cfg(kind, "straight", **(dict(heights=H, model_kind=1) if uneven else {})), n, seed=1234); ob.reset()
        rng = np.random.default_rng(0)
        for t in range(7):
            ob.step(rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32), auto_reset=False)
        q, v, c, w, aux = ob.get_state(); act0 = ob.last_act.copy() if uneven else None
        v = v.astype(np.float32).astype(np.float64)
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        ctrl = c + (a.astype(np.float64) - c) * 0.02 if kind == "tr_env" else a.astype(np.float64)
        kw = dict(model=int(uneven), heights=H)
        L.tgr_probe_set(0)
        q3, v3, _, _ = RE.step(RE.params_from_oracle_model(ob.env(0).model, 1e-9, 40, 50, rtol=0.0), q, v, ctrl, act=act0, nsub=nsub, precision="f64", **kw)
        P = RE.params_from_oracle_model(ob.env(0).model, 1e-5, 40, 50, rtol=0.0)
        print(f"# {kind}, {'heightfield' if uneven else 'flat floor'}, {n} envs after 7 control periods of random actions, then {nsub} sub-steps in the fp64 emulation at tol 1e-5 "
              f"(cap 40), {'Newton on e = d - d_ref (SHIFT)' if shift else 'Newton on d (SHIFT off)'}; ONE quantity rounded to fp32 per line, against the unrounded run at tol 1e-9")
        print("# rounded | axial spin error rad/s: median / p90 | linear velocity error m/s: median | lane-sub-steps left unconverged in the last sub-step | own iterations so far")
        for bits in [[]] + [[b] for b in range(len(NAMES))] + [list(range(len(NAMES)))]:
            L.tgr_probe_set(sum(1 << b for b in bits))
            trg = np.zeros((n, 48), dtype=np.uint32)
            q2, v2, _, out = RE.step(P, q, v, ctrl, act=act0, nsub=nsub, precision="f64", trace=trg, **kw)
            d = np.abs(v2 - v3).reshape(n, 3, 6)
            ax = d[:, :, 5].max(axis=1); lin = d[:, :, :3].max(axis=(1, 2))
            has = (trg[:, nsub - 1] & 0x3fffff) != 0
            name = "nothing" if not bits else (NAMES[bits[0]] if len(bits) == 1 else "ALL of these")
            print(f"{name:38s} | {np.median(ax[has]):.1e} / {np.percentile(ax[has], 90):.1e} | {np.median(lin[has]):.1e} | {100 * ((trg[has, nsub - 1] >> 30) & 1).mean():5.1f} % | "
                  f"{((trg[has, nsub - 1] >> 22) & 0xff).mean():.2f}", flush=True)
        L.tgr_probe_set(0)
        ob.close()


if __name__ == "__main__":
    main()
