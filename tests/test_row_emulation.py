"""The row kernels' physics (tensegrity_rl_amd/csrc/tgr_row.h: one environment = the 16 lanes of a DPP row) checked on the CPU.

tests/emul/row_emul.cpp compiles the SAME header with TGR_ROW_EMUL -- a "lane value" becomes an array of 16 scalars, the row
collectives (row_newbcast / row_ror butterflies) become loops -- so the sub-step can be compared with the CPU checker in the build
container, where there is no GPU.  The emulation also verifies that every replicated value stays bit-identical across the 16 lanes.
-m gpu tests (tests/test_gpu_parity.py) then check the same code on the device through the C-ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_common import qfix, rel_env  # noqa: E402


def _ctrl(kind, c, a):
    return c + (a.astype(np.float64) - c) * 0.02 if kind == "tr_env" else a.astype(np.float64)  # tr_env.py:345,680-683


@pytest.mark.parametrize("kind", ["tensegrity_env", "tr_env"])
def test_fp64_row_substeps_match_the_oracle(oracle, kind):
    import row_emul as RE
    O = oracle
    n = 64
    ob = O.OracleBatch(O.make_cfg(kind, "straight"), n, seed=1234)
    ob.reset()
    P = RE.params_from_oracle_model(ob.env(0).model, 1e-9, 40)
    rng = np.random.default_rng(0)
    bar = 0.0
    for t in range(3):
        q, v, c, w, aux = ob.get_state()
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        q2, v2, _, out = RE.step(P, q, v, _ctrl(kind, c, a), nsub=20, precision="f64")
        info = ob.step(a, auto_reset=False)[5]
        qo, vo = ob.get_state()[:2]
        assert rel_env(qfix(q2, qo) - qo, qo, 1e-3).max() < 1e-8 and rel_env(v2 - vo, vo, 1e-2).max() < 1e-6
        assert np.abs(out[:, :9] - info[:, 16:25]).max() < 1e-8                                       # lagged tendon lengths
        assert (np.abs(out[:, 20] - info[:, 25]) / np.maximum(1.0, info[:, 25])).max() < 1e-4         # bar-bar contact force
        bar = max(bar, info[:, 25].max())
    if kind == "tensegrity_env":
        assert bar > 1.0  # random actions do produce bar-bar contacts to compare
    ob.close()


@pytest.mark.parametrize("kind", ["tensegrity_env", "tr_env"])
def test_mixed_precision_meets_the_stated_tolerance(oracle, kind):
    """BASELINE.json: qpos / qvel within 1e-4 relative per env-step.  The production precision of the row kernels: fp32 with pose,
    kinematics, tendon geometry and the Newton gradient in fp64, stopping rule sqrt(g^T M^-1 g) < 1e-3, state stored as fp32 between
    steps.  >= 99 % of env-steps must be inside the bound; plain fp32 (and mujoco's stopping rule) is not -- see DESIGN.md."""
    import row_emul as RE
    O = oracle
    n = 256
    ob = O.OracleBatch(O.make_cfg(kind, "straight"), n, seed=1234)
    ob.reset()
    P = RE.params_from_oracle_model(ob.env(0).model, 1e-3, 8, 6)
    rng = np.random.default_rng(0)
    eq, ev, it = [], [], []
    for t in range(4):
        q, v, c, w, aux = ob.get_state()
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        q32, v32 = q.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
        q2, v2, _, out = RE.step(P, q32, v32, _ctrl(kind, c, a), nsub=20, precision="mixed")
        ob.step(a, auto_reset=False)
        qo, vo = ob.get_state()[:2]
        eq.append(rel_env(qfix(q2, qo) - qo, qo, 1e-3)); ev.append(rel_env(v2 - vo, vo, 1e-2)); it.append(out[:, 23] / 20)
    eq, ev = np.concatenate(eq), np.concatenate(ev)
    assert (eq < 1e-4).mean() >= 0.99 and (ev < 1e-4).mean() >= 0.99, ((eq < 1e-4).mean(), (ev < 1e-4).mean())
    assert np.median(ev) < 2e-5 and np.concatenate(it).mean() < 5.0
    ob.close()


def test_what_the_pool_cannot_hold_is_counted(oracle, golden_dir):
    """The heightfield model's pool has four lanes (7, 9, 11, 15): two bar pairs crossing 9 mm deep at once have six other contacts -- the
    first four take the pool, two are dropped and COUNTED (tgr_stats dropped_contacts), the trace mask shows the full pool like the oracle's."""
    import row_emul as RE
    from tensegrity_rl_amd.vec_env import heights_from_image
    O = oracle
    h = heights_from_image(np.load(os.path.join(golden_dir, "terrain_100.npy")))
    m = O.uneven_model(h)
    P = RE.params_from_oracle_model(m, 1e-10, 60)
    q = np.stack([_crossed_bars([(0, 1), (1, 2)], 9e-3)])
    v = np.zeros((1, 18)); ctrl = np.full((1, 6), -0.2)
    trace = np.zeros((1, 48), dtype=np.uint32)
    q2, v2, _, out = RE.step(P, q, v, ctrl, nsub=1, precision="f64", model=1, heights=h, trace=trace)
    d = O.new_data(m)
    O.arr(d.qpos)[:] = q[0]; O.arr(d.qvel)[:] = v[0]; O.arr(d.ctrl)[:] = ctrl[0]
    O.lib().tgo_step(O.C.byref(m), O.C.byref(d), 1)
    otr = np.ctypeslib.as_array(d.trace)[:1]
    assert ((trace[0, :1] ^ otr) & 0x3fffff).max() == 0, ([hex(x) for x in trace[0, :1]], [hex(x) for x in otr])
    assert int(trace[0, 0]) & POOL1 == POOL1 and out[0, 21] == 2 and np.isfinite(q2).all() and np.isfinite(v2).all()


def _uneven(oracle, golden_dir, kind, n, seed):
    from tensegrity_rl_amd.vec_env import heights_from_image
    h = heights_from_image(np.load(os.path.join(golden_dir, "terrain_100.npy")))
    ob = oracle.OracleBatch(oracle.make_cfg(kind, "straight", heights=h, model_kind=1), n, seed=seed)
    ob.reset()   # drops the robots onto the terrain and settles them
    return ob, h


@pytest.mark.parametrize("kind", ["tensegrity_env", "tr_env"])
def test_uneven_ground_fp64_row_substeps_match_the_oracle(oracle, golden_dir, kind):
    """MODEL = 1 of the row kernels (heightfield floor: two terrain contacts per cap on slot pages 0 / 1, one edge contact per bar on
    lanes 12-14; filter actuators; passive tendons) against the CPU checker's uneven-ground model"""
    import row_emul as RE
    n = 48
    ob, h = _uneven(oracle, golden_dir, kind, n, 21)
    P = RE.params_from_oracle_model(ob.env(0).model, 1e-9, 40)
    rng = np.random.default_rng(2)
    ncon = nrod = ntwo = 0
    for t in range(6):
        q, v, c, w, aux = ob.get_state()
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        act0 = ob.last_act.copy()
        trace = np.zeros((n, 48), dtype=np.uint32)
        q2, v2, act, out = RE.step(P, q, v, _ctrl(kind, c, a), act=act0, nsub=20, precision="f64", model=1, heights=h, trace=trace)
        ob.step(a, auto_reset=False)
        qo, vo = ob.get_state()[:2]
        assert rel_env(qfix(q2, qo) - qo, qo, 1e-3).max() < 1e-8 and rel_env(v2 - vo, vo, 1e-2).max() < 1e-6
        # the sub-step trace: the same contact slots in every sub-step, touching the same facets and grid edges (hashed identities)
        otr = np.array([np.ctypeslib.as_array(ob.env(i).data.trace)[:20] for i in range(n)], dtype=np.uint32)
        oid = np.array([np.ctypeslib.as_array(ob.env(i).data.trace_id)[:20] for i in range(n)], dtype=np.uint32)
        assert (((trace[:, :20] ^ otr) & 0x3fffff) == 0).all() and (trace[:, 24:44] == oid).all()
        nid = locals().get("nid", 0) + int((oid != 0).sum())
        assert np.abs(act - ob.last_act).max() < 1e-12          # activations: act += h (ctrl - act) / tau
        for i in range(n):
            dd = ob.env(i).data
            ncon += dd.ncon
            floor = [dd.con[j].geom2 for j in range(dd.ncon) if dd.con[j].body1 == -1]
            nrod += sum(1 for g in floor if (g - 1) % 5 == 0)
            ntwo += len(floor) - len(set(floor))
    assert ncon > n and nrod > 0 and ntwo > 0   # caps on one and on two facets, and bars across ridges, were all exercised
    assert nid > 0
    ob.close()


def test_uneven_ground_mixed_precision_tolerance(oracle, golden_dir):
    """the production precision on the heightfield: pose in fp64, velocities in fp32, stopping rule (1e-4, 5e-6) with the iteration cap of 16
    (params_to_row's defaults for model_kind 1), Newton on e = d - d_ref (tgr_row.h SHIFT)"""
    import row_emul as RE
    n = 96
    ob, h = _uneven(oracle, golden_dir, "tr_env", n, 21)
    P = RE.params_from_oracle_model(ob.env(0).model, 1e-4, 16, 6, rtol=5e-6)
    RE.lib().tgr_row_emul_set_warm_start(1)
    rng = np.random.default_rng(2)
    eq, ev = [], []
    for t in range(5):
        q, v, c, w, aux = ob.get_state()
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        v32 = v.astype(np.float32).astype(np.float64)
        q2, v2, _, out = RE.step(P, q, v32, _ctrl("tr_env", c, a), act=ob.last_act, nsub=20, precision="mixed", model=1, heights=h)
        ob.step(a, auto_reset=False)
        qo, vo = ob.get_state()[:2]
        eq.append(rel_env(qfix(q2, qo) - qo, qo, 1e-3)); ev.append(rel_env(v2 - vo, vo, 1e-2))
    RE.lib().tgr_row_emul_set_warm_start(0)
    eq, ev = np.concatenate(eq), np.concatenate(ev)
    assert (eq < 1e-4).mean() >= 0.99 and (ev < 1e-4).mean() >= 0.99, ((eq < 1e-4).mean(), (ev < 1e-4).mean())
    assert np.median(ev) < 5e-6
    ob.close()


def _crossed_bars(pairs, pen, s0=0.4):
    """qpos of three free bars far above the floor; for every (A, B) in `pairs` bar B's axis crosses over bar A's at distance 2 r - pen, both
    at s0 from their centres -- where each carries a b-cylinder (r = 0.035 inside r = 0.0381, flat xml:77).  pen > 6.2 mm: r-r, r-b, b-r
    AND b-b touch (three contacts besides the r-cylinders side to side)"""
    R_CYL = 0.0381
    yq = np.array([np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0])          # local z -> world x
    xq = np.array([np.cos(np.pi / 4), -np.sin(np.pi / 4), 0, 0])         # local z -> world y
    q = np.zeros((3, 7))
    # bars parked far apart, then the crossing ones placed: bar A along x, bar B along y above it
    for b in range(3):
        q[b] = [3.0 * b, 0, 5.0 + 1.0 * b, *yq]
    placed = set()
    for (A, B) in pairs:
        if A not in placed:
            placed.add(A)
        ax = q[A, :3]
        # A's axis: along x if its quaternion is yq, along y if xq
        alongx = np.allclose(q[A, 3:], yq)
        if alongx:
            q[B, :3] = ax + np.array([s0, -s0, 2 * R_CYL - pen]); q[B, 3:] = xq
        else:
            q[B, :3] = ax + np.array([s0, s0, 2 * R_CYL - pen]); q[B, 3:] = yq
        placed.add(B)
    return q.reshape(21)


_PAIR_INDEX = {(0, 1): 0, (0, 2): 1, (1, 2): 2}
POOL0, POOL1 = 0xfa80, 0x8a80   # the pool lanes of the flat-floor / heightfield row kernels (tgr_row.h pool_mask)


@pytest.mark.parametrize("pairs", [[(0, 1)], [(0, 2)], [(1, 2)], [(0, 1), (1, 2)]])
def test_the_pool_carries_every_other_contact_of_the_bar_pairs(oracle, pairs):
    """A bar crossing deeper than 6.2 mm where both bars carry a b-cylinder touches four times (r-r, r-b, b-r, b-b); mujoco keeps every
    geom pair.  The r-cylinders' side-to-side contact has its fixed lane (6 + 2p); every other contact takes the next lane of the pool
    (7, 9, 11, 12..15), whose bar pair is a run-time value of the lane: fp64 row sub-steps equal the oracle's for every pair, also with two
    pairs that deep at once (six others: until round 4 the second pair's fourth contact found no slot), and nothing is dropped."""
    import row_emul as RE
    O = oracle
    m = O.default_model()
    P = RE.params_from_oracle_model(m, 1e-10, 60)
    n = 4
    pens = np.array([4e-3, 7e-3, 9e-3, 12e-3])   # the first: three contacts only (b-b needs 6.2 mm)
    q = np.stack([_crossed_bars(pairs, p_) for p_ in pens])
    rng = np.random.default_rng(0)
    v = rng.normal(0, 0.05, (n, 18))
    ctrl = np.full((n, 6), -0.2)
    trace = np.zeros((n, 48), dtype=np.uint32)
    q2, v2, _, out = RE.step(P, q, v, ctrl, nsub=3, precision="f64", trace=trace)
    for i in range(n):
        d = O.new_data(m)
        O.arr(d.qpos)[:] = q[i]; O.arr(d.qvel)[:] = v[i]; O.arr(d.ctrl)[:] = ctrl[i]
        O.lib().tgo_step(O.C.byref(m), O.C.byref(d), 3)
        qo, vo = O.arr(d.qpos).copy()[None], O.arr(d.qvel).copy()[None]
        otr = np.ctypeslib.as_array(d.trace)[:3]
        assert rel_env(qfix(q2[i:i + 1], qo) - qo, qo, 1e-3).max() < 1e-8 and rel_env(v2[i:i + 1] - vo, vo, 1e-2).max() < 1e-6, (pairs, i)
        assert ((trace[i, :3] ^ otr) & 0xffff).max() == 0, (pairs, i, [hex(x) for x in trace[i, :3]], [hex(x) for x in otr])
        assert bin(int(trace[i, 0]) & POOL0).count("1") == (2 if i == 0 else 3) * len(pairs)   # r-b, b-r (+ b-b where the crossing is deep enough) per pair
        assert all(int(trace[i, 0]) & (1 << (6 + 2 * _PAIR_INDEX[p_])) for p_ in pairs)       # the side-to-side contacts on their fixed lanes
    assert (out[:, 21] == 0).all(), out[:, 21]   # contacts that found no slot in the last sub-step


@pytest.mark.parametrize("pairs", [[(0, 1)], [(1, 2)]])
def test_the_pool_on_the_heightfield_model(oracle, golden_dir, pairs):
    """MODEL = 1: lanes 12..14 carry the bars' terrain contacts, so the pool is lanes 7, 9, 11, 15: a 4 mm crossing (r-r, r-b, b-r) uses two of
    them, a 9 mm one (+ b-b) three; nothing is dropped"""
    import row_emul as RE
    from tensegrity_rl_amd.vec_env import heights_from_image
    O = oracle
    h = heights_from_image(np.load(os.path.join(golden_dir, "terrain_100.npy")))
    m = O.uneven_model(h)
    P = RE.params_from_oracle_model(m, 1e-10, 60)
    pens = np.array([0.5e-3, 4e-3, 9e-3])   # r-r only (the b-cylinders are 3.1 mm inside) / + r-b, b-r / + b-b
    n = len(pens)
    q = np.stack([_crossed_bars(pairs, p_) for p_ in pens])
    rng = np.random.default_rng(0)
    v = rng.normal(0, 0.05, (n, 18))
    ctrl = np.full((n, 6), -0.2)
    trace = np.zeros((n, 48), dtype=np.uint32)
    # (one sub-step: this model's 1e4 N/m tendons, stretched over the metres between the parked bars, tear the crossing apart at 2 m/s)
    q2, v2, _, out = RE.step(P, q, v, ctrl, nsub=1, precision="f64", model=1, heights=h, trace=trace)
    for i in range(n):
        d = O.new_data(m)
        O.arr(d.qpos)[:] = q[i]; O.arr(d.qvel)[:] = v[i]; O.arr(d.ctrl)[:] = ctrl[i]
        O.lib().tgo_step(O.C.byref(m), O.C.byref(d), 1)
        qo, vo = O.arr(d.qpos).copy()[None], O.arr(d.qvel).copy()[None]
        otr = np.ctypeslib.as_array(d.trace)[:1]
        assert ((trace[i, :1] ^ otr) & 0x3fffff).max() == 0, (pairs, i, [hex(x) for x in trace[i, :1]], [hex(x) for x in otr])
        if True:
            assert rel_env(qfix(q2[i:i + 1], qo) - qo, qo, 1e-3).max() < 1e-8 and rel_env(v2[i:i + 1] - vo, vo, 1e-2).max() < 1e-6, (pairs, i)
    assert [bin(int(t) & POOL1).count("1") for t in trace[:, 0]] == [0, 2, 3]
    assert list(out[:, 21]) == [0, 0, 0]
