"""-m gpu: reset_model() on the device (settle kernel and donor lanes) against the oracle, and the SB3 VecEnv contract."""
import numpy as np
import pytest

from gpu_common import AUX, qfix, oracle_to_gpu_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,action", [("tensegrity_env", "straight"), ("tensegrity_env", "turn"), ("tr_env", "straight"), ("tr_env", "turn"),
                                         ("tr_env", "tracking"), ("tr_env", "aiming"), ("tr_env", "vel_track"),
                                         ("tensegrity_env", "noise"), ("tr_env", "noise")])
def test_reset_matches_oracle_fp64(oracle, kind, action):
    """same philox draws -> same start pose -> 1000 sub-steps of settling agree (fp64).  "noise": reset_noise_scale = 0.05
    (tensegrity_env.py:436-445; the oracle's handling of it is pinned by the *_reset_noise_scale-0.05 fixtures)"""
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n = 32
    kw = {}
    if action == "noise":
        action, kw = "straight", dict(reset_noise_scale=0.05)
    env = TensegrityVecEnv(n, kind + "-v0", desired_action=action, precision="f64", auto_reset=False, seed=77, first_env_id=5, **kw)
    obs = env.reset()
    ob = O.OracleBatch(O.make_cfg(kind, action, **kw), n, seed=77, first_env_id=5)
    oobs = ob.reset()
    st = env.get_state()
    q, v, c, w, aux = ob.get_state()
    # 1000 sub-steps of settling amplify the (1e-11) solver tolerance; the two solvers take different paths
    assert np.abs(qfix(st["qpos"], q) - q).max() < 2e-4
    assert np.abs(st["qvel"] - v).max() < 2e-3
    assert np.abs(st["ctrl"] - c).max() < 1e-12
    assert np.abs(st["aux"][:, AUX["reset_psi"]] - aux["reset_psi"]).max() < 1e-3
    for k_ in ("waypt_x", "waypt_y", "ori_x", "ori_y", "vcmd_x", "vcmd_y", "vcmd_w"):
        assert np.abs(st["aux"][:, AUX[k_]] - aux[k_]).max() < 2e-3, k_
    assert np.abs(obs - oobs).max() < 2e-3
    # and the first steps after the reset, including the heading ring the settle left behind (turn mode)
    rng = np.random.default_rng(3)
    for t in range(3):
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        o_obs, o_rew, o_term, _, _, _ = ob.step(a, auto_reset=False)
        g_obs, g_rew, g_term, _ = env.step(a)
        assert np.abs(g_rew - o_rew).max() < 5e-3 and np.abs(g_obs - o_obs).max() < 5e-2
    # the heading deque reset_model() leaves behind (tensegrity_env turn: 50 pushes capped at 25; tr_env turn / aiming: the prefill)
    st = env.get_state(); aux2 = ob.get_state()[4]
    assert (st["aux"][:, AUX["ring_len"]] == aux2["ring_len"]).all()
    assert np.abs(st["aux"][:, AUX["ring"]:AUX["ring"] + 32] - aux2["ring"]).max() < 5e-3
    env.close(); ob.close()


def test_reset_with_the_same_seed_reproduces_the_episode():
    """Env.reset(seed=s) twice -> the same episode twice (tgr_set_seed restarts the lanes' episode counters)"""
    from tensegrity_rl_amd.envs import make
    env = make("tr_env-v0", desired_action="tracking", precision="f64")
    o1, _ = env.reset(seed=11); q1 = env.get_state()["qpos"].copy()
    o2, _ = env.reset(seed=12)
    o3, _ = env.reset(seed=11); q3 = env.get_state()["qpos"].copy()
    assert np.array_equal(o1, o3) and np.array_equal(q1, q3) and not np.array_equal(o1, o2)
    env.close()


def test_reset_fp32_close_to_oracle(oracle):
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n = 128
    env = TensegrityVecEnv(n, "tr_env-v0", precision="f32", auto_reset=False, seed=5)
    obs = env.reset()
    ob = O.OracleBatch(O.make_cfg("tr_env", "straight"), n, seed=5)
    oobs = ob.reset()
    st = env.get_state(); q = ob.get_state()[0]
    err = np.abs(qfix(st["qpos"], q) - q).max(axis=1)
    assert np.median(err) < 1e-3 and np.mean(err < 2e-2) > 0.9  # 1000 sub-steps; settling is dissipative but not contractive
    assert np.isfinite(obs).all() and np.median(np.abs(obs - oobs).max(axis=1)) < 5e-3
    env.close(); ob.close()


def test_vecenv_contract_autoreset_truncation_and_donors(oracle):
    """SB3 VecEnv semantics [ext]: auto-reset, terminal_observation, TimeLimit.truncated, Monitor-style episode info;
    and the state an env receives from a donor lane is exactly reset_model() under that donor's RNG key."""
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n = 16
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision="f64", max_episode_steps=3, seed=11, n_donors=32, info_mode="full")
    assert env.observation_space.shape == (39,) and env.action_space.shape == (6,) and env.action_space.dtype == np.float32
    assert abs(float(env.action_space.low[0]) + 0.45) < 1e-6 and abs(float(env.action_space.high[0]) - 0.15) < 1e-6
    obs = env.reset()
    assert obs.shape == (n, 39) and obs.dtype == np.float64
    rng = np.random.default_rng(0)
    rets = np.zeros(n)
    for t in range(3):
        a = (-0.2 + 0.02 * rng.standard_normal((n, 6))).astype(np.float32)  # gentle: nobody terminates before the time limit
        obs, rew, dones, infos = env.step(a)
        assert t == 2 or not dones.any()
        rets += rew
        assert rew.dtype == np.float32 and dones.dtype == bool and len(infos) == n
        assert {"reward_forward", "reward_ctrl", "reward_survive", "x_position", "y_position", "psi", "distance_from_origin",
                "x_velocity", "y_velocity", "forward_reward"} <= set(infos[0])
    assert dones.all()  # TimeLimit at 3 steps
    for i in range(n):
        assert "terminal_observation" in infos[i] and infos[i]["terminal_observation"].shape == (39,)
        assert infos[i]["episode"]["l"] == 3 and abs(infos[i]["episode"]["r"] - rets[i]) < 1e-4
        assert isinstance(infos[i]["TimeLimit.truncated"], bool)
        assert not np.allclose(infos[i]["terminal_observation"], obs[i])  # obs row already belongs to the new episode
    st = env.get_state()
    stats = env.stats()
    assert stats["donor_resets"] == n and st["aux"][:, AUX["elapsed"]].max() == 0
    # every env now carries a donor's settled episode: reproduce it with the oracle from (key, episode)
    cfg = O.make_cfg("tensegrity_env", "straight")
    for i in range(0, n, 5):
        src = int(st["aux"][i, AUX["src"]]); ep = int(st["aux"][i, AUX["episode"]])
        assert src >= n
        key = env.lib.tgr_lane_key(env.handle, src)
        oe = O.OracleEnv(cfg)
        oobs = oe.reset_seeded(11, key, ep)
        # 1000 sub-steps of settling amplify the (1e-11) solver tolerance; the two solvers take different paths
        assert np.abs(qfix(st["qpos"][i], O.arr(oe.data.qpos)) - O.arr(oe.data.qpos)).max() < 2e-4
        assert np.abs(obs[i] - oobs).max() < 2e-3
    # Gym single-env API
    from tensegrity_rl_amd import make
    e1 = make("tr_env-v0", precision="f64", seed=3)
    o, info = e1.reset()
    assert o.shape == (45,) and info == {}
    o, r, term, trunc, info = e1.step(np.zeros(6, dtype=np.float32))
    assert isinstance(r, float) and isinstance(term, bool) and trunc is False and "waypt" in info and e1.dt == 0.02
    with pytest.raises(ValueError):
        e1.step(np.zeros(5))
    e1.close(); env.close()


def test_starved_envs_fall_back_to_settle_kernel(oracle):
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 8
    env = TensegrityVecEnv(n, "tr_env-v0", precision="f32", max_episode_steps=2, n_donors=0, seed=1)
    env.reset()
    a = np.zeros((n, 6), dtype=np.float32)
    env.step(a); obs, rew, dones, infos = env.step(a)
    assert dones.all() and env.stats()["settle_kernel_resets"] == 2 * n and env.stats()["donor_resets"] == 0
    st = env.get_state()
    assert (st["aux"][:, AUX["src"]] == np.arange(n)).all() and (st["aux"][:, AUX["episode"]] == 1).all()
    env.close()


def test_full_size_properties():
    """BASELINE config size (N = 4096): size-independent properties -- determinism, shard invariance, finiteness,
    unit quaternions, done envs restart with elapsed = 0."""
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 4096

    def rollout(first, count, steps=12, **kw):
        env = TensegrityVecEnv(count, "tensegrity_env-v0", seed=42, first_env_id=first, **(kw or dict(auto_reset=False)))
        env.reset_tensors()
        g = torch.Generator(device="cuda"); g.manual_seed(9)
        acts = [torch.rand((n, 6), device="cuda", generator=g) * 0.6 - 0.45 for _ in range(steps)]
        outs = []
        for a in acts:
            o, r, te, tr, _, _ = env.step_tensors(a[first:first + count].contiguous())
            outs.append((o.clone(), r.clone(), te.clone()))
        st = env.get_state(); env.close()
        return outs, st

    full, st_full = rollout(0, n)
    again, _ = rollout(0, n)
    for (o1, r1, t1), (o2, r2, t2) in zip(full, again):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2)  # bitwise deterministic
    part, st_part = rollout(1024, 512)
    for (o1, r1, t1), (o2, r2, t2) in zip(full, part):
        assert torch.equal(o1[1024:1536], o2) and torch.equal(r1[1024:1536], r2)  # results do not depend on the sharding
    # ... and with auto-reset on (the default: reset streams in classes of 512 envs) across two waves of resets
    full_r, _ = rollout(0, n, steps=14, max_episode_steps=6, reset_stream_envs=512)
    part_r, _ = rollout(1024, 512, steps=14, max_episode_steps=6, reset_stream_envs=512)
    for (o1, r1, t1), (o2, r2, t2) in zip(full_r, part_r):
        assert torch.equal(o1[1024:1536], o2) and torch.equal(r1[1024:1536], r2)
    o, r, t = full[-1]
    assert torch.isfinite(o).all() and torch.isfinite(r).all()
    q = st_full["qpos"].reshape(n, 3, 7)[:, :, 3:]
    assert np.abs(np.linalg.norm(q, axis=2) - 1).max() < 1e-5
    assert np.abs(o[:, :12].cpu().numpy().reshape(n, 3, 4) ** 2).sum(axis=2).max() < 1 + 1e-4
    # auto-reset at full size
    env = TensegrityVecEnv(n, "tensegrity_env-v0", seed=42, max_episode_steps=4)
    env.reset_tensors()
    for i in range(4):
        o, r, te, tr, tobs, info = env.step_tensors(torch.zeros((n, 6), device="cuda") - 0.2)
    trh = tr.cpu().numpy().astype(bool)
    assert trh.mean() > 0.9  # the rest terminated earlier and restarted their clock
    assert env.get_state()["aux"][trh, 6].max() == 0 and env.stats()["donor_resets"] >= int(trh.sum())
    assert env.stats()["settle_kernel_resets"] == 3 * n  # the initial reset of envs + donors only: nobody was starved
    env.close()


@pytest.mark.parametrize("action,capv", [("straight", 1), ("straight", 0), ("tracking", 1), ("vel_track", 1)])
def test_obs_noise_matches_oracle(oracle, action, capv):
    """tr_env use_obs_noise=True (tr_env.py:524-527,549-646,869-872): reset and step observations carry noise from philox
    draws keyed (seed, lane key, episode, observation index); the fp64 HIP path and the oracle agree on every noisy entry, and
    the noise has the reference's scale (sigma 0.05 on cap positions/velocities, 0.02 on tendon lengths)."""
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n = 32
    kw = dict(use_obs_noise=1, use_cap_velocity=capv)
    ob = O.OracleBatch(O.make_cfg("tr_env", action, **kw), n, seed=77)
    env = TensegrityVecEnv(n, "tr_env-v0", desired_action=action, precision="f64", auto_reset=False, seed=77, **kw)
    clean = TensegrityVecEnv(n, "tr_env-v0", desired_action=action, precision="f64", auto_reset=False, seed=77, use_cap_velocity=capv)
    o_obs = ob.reset()
    g_obs = env.reset()
    c_obs = clean.reset()
    np.testing.assert_allclose(g_obs, o_obs, atol=5e-4)  # 50 settling periods: solver paths differ at the 1e-4 level
    nv = 18 if capv else 0
    rng = np.random.default_rng(3)
    diffs = []
    for t in range(4):
        oracle_to_gpu_state(ob, env)
        oracle_to_gpu_state(ob, clean)
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        o_obs = ob.step(a, auto_reset=False)[0]
        g_obs = env.step(a)[0]
        c_obs = clean.step(a)[0]
        np.testing.assert_allclose(g_obs, o_obs, atol=2e-6)
        # info["real_observation"] (tr_env.py:507, read by run.py:147): the noise-free observation of the same step
        np.testing.assert_allclose(env.real_obs.cpu().numpy(), c_obs, atol=1e-12)
        diffs.append(g_obs - c_obs)
    d = np.concatenate(diffs)
    pos, ten = d[:, :18], d[:, 18 + nv:27 + nv]
    assert abs(pos.std() - 0.05) < 0.006 and abs(ten.std() - 0.02) < 0.003 and abs(pos.mean()) < 0.01
    if nv:
        assert abs(d[:, 18:36].std() - 0.05) < 0.006
    assert not np.allclose(diffs[0], diffs[1])  # fresh draws every observation
    env.info_mode = "full"
    o, r, dn, infos = env.step(a)
    assert np.abs(infos[3]["real_observation"] - o[3]).max() > 1e-3 and infos[3]["real_observation"].shape == o[3].shape
    assert clean.real_obs is None  # nothing is bound (or written) without observation noise
    env.close(); clean.close(); ob.close()


def test_edge_cases_ragged_sizes_masks_and_bad_inputs(oracle):
    """N = 1 and N not a multiple of the 64-env block, masked resets, wrong shapes, non-finite actions (the reference turns a
    non-finite state into `terminated`, tensegrity_env.py:276-288; a lane never raises)."""
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    # ragged block: 65 envs = one full block + one lane; every env must equal the oracle's env with the same key
    n = 65
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision="f64", auto_reset=False, seed=5)
    obs = env.reset()
    ob = O.OracleBatch(O.make_cfg("tensegrity_env", "straight"), n, seed=5)
    oobs = ob.reset()
    assert np.abs(obs - oobs).max() < 2e-3
    a = np.random.default_rng(1).uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
    oracle_to_gpu_state(ob, env)
    o1 = env.step(a)[0]; o2 = ob.step(a, auto_reset=False)[0]
    assert np.abs(o1 - o2).max() < 1e-5
    # masked reset: only the chosen envs restart (new episode under their key), the others keep their state
    st0 = env.get_state()
    mask = np.zeros(n, dtype=np.uint8); mask[[0, 64]] = 1
    env.reset_tensors(mask)
    st1 = env.get_state()
    keep = mask == 0
    assert np.array_equal(st0["qpos"][keep], st1["qpos"][keep]) and not np.allclose(st0["qpos"][~keep], st1["qpos"][~keep])
    assert (st1["aux"][~keep, AUX["elapsed"]] == 0).all() and (st1["aux"][~keep, AUX["episode"]] == 1).all()
    with pytest.raises(ValueError):
        env.reset_tensors(np.zeros(n + 1, dtype=np.uint8))
    with pytest.raises(ValueError):
        env.step(np.zeros((n, 5), dtype=np.float32))
    with pytest.raises(ValueError):
        env.step(np.zeros((n - 1, 6), dtype=np.float32))
    env.close(); ob.close()
    # a single env, fp32, auto-reset: non-finite actions poison the state -> terminated, restarted, finite again
    e1 = TensegrityVecEnv(1, "tensegrity_env-v0", seed=2)
    e1.reset()
    bad = np.full((1, 6), np.nan, dtype=np.float32)
    obs, rew, dones, infos = e1.step(bad)
    assert dones[0] and "terminal_observation" in infos[0] and np.isfinite(obs).all()
    obs, rew, dones, infos = e1.step(np.full((1, 6), -0.2, dtype=np.float32))
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    e1.close()
    # device-tensor API rejects tensors of the wrong shape as well
    e2 = TensegrityVecEnv(3, "tr_env-v0", seed=2)
    e2.reset_tensors()
    with pytest.raises(ValueError):
        e2.step_tensors(torch.zeros((3, 7), device="cuda"))
    out = e2.step_tensors(torch.zeros((3, 6), device="cuda", dtype=torch.float64))  # wrong dtype is converted, not rejected
    assert torch.isfinite(out[0]).all()
    e2.close()


def test_paced_donor_rows_settle_the_same_episodes(oracle, monkeypatch):
    """Paced donor rows (the donor rows beyond the first round of wave slots run HALF a control period per launch) against
    the oracle.  Which rows are paced normally depends on n and on the device's CU count; TGR_PACE_FROM (a test hook read by tgr_create)
    pins it: here donor lanes 32..47 of a 16-env fp64 handle.  With a 3-step time limit the sixteen envs finish together every third step:
    step 3 takes the (unpaced) lanes 16..31, step 6 the lanes 32..47 -- both settled by tgr_reset's settle kernel -- steps 9..51 find nothing
    ready, step 54 and 105 take 16..31 again, and step 108 takes what lanes 32..47 settled over 100 half-period launches (their episode 1) -- the I_HALF state
    machine, the mid-period store / reload of the state and the solver's cold start at sub-step 11 included -- and every one of them must
    be reset_model() under that lane's RNG key, to the tolerance the unpaced donors meet in test_vecenv_contract_autoreset_truncation_and_donors."""
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n = 16
    monkeypatch.setenv("TGR_PACE_FROM", "32")
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision="f64", max_episode_steps=3, seed=21, n_donors=32, donor_stream=1)  # (one fused launch)
    monkeypatch.delenv("TGR_PACE_FROM")
    env.reset()
    rng = np.random.default_rng(0)
    for t in range(108):
        a = (-0.2 + 0.02 * rng.standard_normal((n, 6))).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
    assert dones.all()
    st = env.get_state()
    src = st["aux"][:, AUX["src"]].astype(int); ep = st["aux"][:, AUX["episode"]].astype(int)
    assert (src >= 32).all() and (ep == 1).all(), (src, ep)      # the paced lanes' first episodes settled inside the step launches
    cfg = O.make_cfg("tensegrity_env", "straight")
    for i in range(n):
        oe = O.OracleEnv(cfg)
        oobs = oe.reset_seeded(21, env.lib.tgr_lane_key(env.handle, int(src[i])), int(ep[i]))
        assert np.abs(qfix(st["qpos"][i], O.arr(oe.data.qpos)) - O.arr(oe.data.qpos)).max() < 2e-4, i
        assert np.abs(obs[i] - oobs).max() < 2e-3, i
    env.close()


@pytest.mark.parametrize("info_mode", ["minimal", "full"])
@pytest.mark.parametrize("env_id,model", [("tensegrity_env-v0", "flat"), ("tr_env-v0", "flat"), ("tr_env-v0", "uneven")])
def test_step_and_step_tensors_return_the_same_values(golden_dir, env_id, model, info_mode):
    """The numpy-facing SB3 surface (step_async / step_wait: one packed device-to-host copy into a pinned mirror) against the device
    tensors of the same step: observations, rewards, dones, and for every finished env terminal_observation, TimeLimit.truncated and
    Monitor's episode record.  info_mode="full": every env's info is the reference's dict (tensegrity_env.py:383-398, tr_env.py:496-511),
    handed out as a lazy SNAPSHOT mapping -- its values equal the device's info block of that step, and still do after later steps."""
    import os
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    from tensegrity_rl_amd import _native as N
    n = 256
    kw = dict(model="uneven", terrain=np.load(os.path.join(golden_dir, "terrain_100.npy"))) if model == "uneven" else {}
    envs = [TensegrityVecEnv(n, env_id, seed=5, max_episode_steps=7, info_mode=info_mode, **kw) for _ in range(2)]
    o0 = envs[0].reset(); o1 = envs[1].reset_tensors().cpu().numpy()
    assert np.array_equal(o0.astype(np.float32), o1)
    rng = np.random.default_rng(1)
    finished = 0
    tr = env_id.startswith("tr_env")
    kept = []
    for t in range(16):
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        obs, rew, dones, infos = envs[0].step(a)
        tobs_, trew, tterm, ttrunc, ttobs, tinfo = [x.cpu().numpy() for x in envs[1].step_tensors(torch.as_tensor(a, device="cuda"))]
        assert obs.dtype == np.float64 and rew.dtype == np.float32 and dones.dtype == bool and len(infos) == n
        assert np.array_equal(obs.astype(np.float32), tobs_) and np.array_equal(rew, trew)
        assert np.array_equal(dones, (tterm | ttrunc).astype(bool))
        for i in range(n):
            if dones[i]:
                finished += 1
                assert np.array_equal(infos[i]["terminal_observation"].astype(np.float32), ttobs[i])
                assert infos[i]["TimeLimit.truncated"] == bool(ttrunc[i] and not tterm[i])
                assert infos[i]["episode"]["l"] == int(tinfo[i, 15]) and infos[i]["episode"]["r"] == float(tinfo[i, 14])
            elif info_mode == "minimal":
                assert infos[i] == {}
        if info_mode == "full":
            want = set(N.INFO_KEYS) | {"total_bar_contact"} | ({"waypt", "oripoint", "tendon_length", "real_observation"} if tr else set())
            for i in (0, 1, n // 2, n - 1):
                d = infos[i]
                assert want <= set(d.keys()) and (dones[i] or set(d.keys()) == want)
                for j, k in enumerate(N.INFO_KEYS):
                    assert d[k] == np.float64(tinfo[i, j]), (i, k)
                assert d["total_bar_contact"] == float(tinfo[i, N.INFO_BAR_CONTACT])
                if tr:
                    assert np.array_equal(d["waypt"], tinfo[i, 10:12].astype(np.float64)) and np.array_equal(d["oripoint"], tinfo[i, 12:14].astype(np.float64))
                    assert np.array_equal(d["tendon_length"], tinfo[i, 16:25].astype(np.float64)) and d["tendon_length"].dtype == np.float64
                    # tr_env.py:507 without observation noise: the observation of the step (of its last step for an env that just finished)
                    assert np.array_equal(d["real_observation"].astype(np.float32), ttobs[i] if dones[i] else tobs_[i])
                assert d.get("no such key") is None and dict(d)["x_position"] == d["x_position"]
            if t == 3:
                kept = [(infos[5], dict(infos[5])), (infos[-1], dict(infos[-1]))]
    for lazy, snap in kept:   # a mapping handed out at step 3 still holds step 3's values after twelve more steps
        for k, v in snap.items():
            assert np.array_equal(np.asarray(lazy[k]), np.asarray(v)) if not isinstance(v, dict) else lazy[k] == v
    assert finished >= n   # the 7-step time limit alone ends every episode twice
    for e in envs:
        e.close()


def test_reset_streams_hand_out_a_class_stream_in_order(oracle):
    """reset_stream_envs = B (results independent of the sharding): the k-th reset of a class of B consecutive global env ids
    gets episode k of the class's stream, keyed (class key, k) -- whichever donor lane settled it, or the env itself when it was not ready.
    fp64, 32 envs in two classes with 16 donor lanes each and a 3-step time limit: step 3 hands episodes 0..15 of each class to its envs in
    env order (from donors), step 6 episodes 16..31 (nothing is ready: the envs settle them themselves); both are reset_model() under
    (class key, episode) in the oracle."""
    from tensegrity_rl_amd import TensegrityVecEnv
    O = oracle
    n, B = 32, 16
    env = TensegrityVecEnv(n, "tensegrity_env-v0", precision="f64", max_episode_steps=3, seed=31, n_donors=32, reset_stream_envs=B, first_env_id=64)
    env.reset()
    cfg = O.make_cfg("tensegrity_env", "straight")
    rng = np.random.default_rng(0)
    for rnd in range(2):
        for t in range(3):
            obs, rew, dones, infos = env.step((-0.2 + 0.02 * rng.standard_normal((n, 6))).astype(np.float32))
        assert dones.all()
        st = env.get_state()
        src = st["aux"][:, AUX["src"]].astype(int); ep = st["aux"][:, AUX["episode"]].astype(int)
        assert (ep == np.tile(np.arange(B), n // B) + rnd * B).all(), ep          # in order within each class
        assert (src >= n).all() and ((src - n) // 16 == np.arange(n) // B).all()   # keyed by a lane of the env's own class
        keys = [env.lib.tgr_lane_key(env.handle, int(s)) for s in src]
        assert len(set(keys)) == n // B and keys[0] != keys[-1]
        for i in (0, 5, 15, 16, 31):
            oe = O.OracleEnv(cfg)
            oobs = oe.reset_seeded(31, keys[i], int(ep[i]))
            assert np.abs(qfix(st["qpos"][i], O.arr(oe.data.qpos)) - O.arr(oe.data.qpos)).max() < 2e-4, (rnd, i)
            assert np.abs(obs[i] - oobs).max() < 2e-3, (rnd, i)
    s = env.stats()
    assert s["donor_resets"] == n and s["settle_kernel_resets"] == n + 32 + n   # reset: envs + donors; second round: every env starved
    env.close()


@pytest.mark.parametrize("donors_per_env,limit", [(1.25, 5000), (0.25, 20), (0.0625, 5000)])
def test_reset_streams_make_auto_reset_shard_invariant(donors_per_env, limit):
    """the same 512 global envs stepped as part of a 1024-env handle and as a 512-env shard (first_env_id = 512), auto-reset on, random
    actions, 150 steps across many resets: observations, rewards and done flags are bitwise equal at every step -- with plenty of donor
    lanes; with so few (and a 20-step time limit) that most resets are settled by the envs themselves and no donor ever finishes; and with
    16 donor lanes per class under the natural episode lengths (~47 steps): under-provisioned classes whose donors DO finish episodes while
    the match kernel keeps re-arming the holders of starved episodes -- the case in which a donor's ready word and its re-arm command meet
    in one launch (readiness is decided on match-side words only: a ready stamp at or below the lane's last command tag is the replaced
    episode's)."""
    import torch
    from tensegrity_rl_amd import TensegrityVecEnv
    B, steps = 256, 150

    def rollout(first, count):
        env = TensegrityVecEnv(count, "tensegrity_env-v0", seed=42, first_env_id=first, reset_stream_envs=B, n_donors=int(donors_per_env * count),
                               max_episode_steps=limit)
        env.reset_tensors()
        g = torch.Generator(device="cuda"); g.manual_seed(9)
        outs = []
        for t in range(steps):
            a = (torch.rand((1024, 6), device="cuda", generator=g) * 0.6 - 0.45)[first:first + count].contiguous()
            o, r, te, tr, tobs, _ = env.step_tensors(a)
            outs.append((o.clone(), r.clone(), (te | tr).clone(), tobs.clone()))
        s = env.stats(); env.close()
        return outs, s

    full, s_full = rollout(0, 1024)
    part, s_part = rollout(512, 512)
    resets = 0
    for (o1, r1, d1, t1), (o2, r2, d2, t2) in zip(full, part):
        assert torch.equal(o1[512:], o2) and torch.equal(r1[512:], r2) and torch.equal(d1[512:], d2)
        assert torch.equal(t1[512:][d2.bool()], t2[d2.bool()])
        resets += int(d2.sum())
    assert resets > 512 and s_part["donor_resets"] > 0
    if donors_per_env < 1:
        assert s_part["settle_kernel_resets"] > 512 + int(donors_per_env * 512)   # starved resets did happen


def test_sb3_adapter_forwards_everything_under_a_stand_in_base_class(monkeypatch):
    """stable_baselines3 is not installable here or on the GPU box, so `TensegrityVecEnv.to_sb3()` (tensegrity_rl_amd/sb3.py) cannot run against
    the real `VecEnv` base class.  This executes it against a STAND-IN with the base class's constructor signature and its `step()` =
    `step_async()` + `step_wait()` (SB3 2.2.1 [ext]): the adapter must construct, and forward reset / step / attribute access to the
    native VecEnv unchanged.  (What SB3 itself does with the object stays untested until SB3 is importable somewhere.)"""
    import sys
    import types
    from tensegrity_rl_amd import TensegrityVecEnv

    class VecEnv:   # stand-in for stable_baselines3.common.vec_env.base_vec_env.VecEnv
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
            self.reset_infos = [{} for _ in range(num_envs)]

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()
    mods = {}
    for name in ("stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env", "stable_baselines3.common.vec_env.base_vec_env"):
        mods[name] = types.ModuleType(name)
    mods["stable_baselines3.common.vec_env.base_vec_env"].VecEnv = VecEnv
    for name, m in mods.items():
        monkeypatch.setitem(sys.modules, name, m)
    n = 64
    a_env = TensegrityVecEnv(n, "tr_env-v0", seed=3, max_episode_steps=4)
    b_env = TensegrityVecEnv(n, "tr_env-v0", seed=3, max_episode_steps=4)
    sb3 = a_env.to_sb3()
    assert isinstance(sb3, VecEnv) and sb3.num_envs == n and sb3.observation_space.shape == (45,)
    assert np.array_equal(sb3.reset(), b_env.reset())
    rng = np.random.default_rng(0)
    for t in range(5):
        act = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        o1, r1, d1, i1 = sb3.step(act)
        o2, r2, d2, i2 = b_env.step(act)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
        assert [sorted(x) for x in i1] == [sorted(x) for x in i2]
    assert d1.all() is not None and sb3.get_attr("dt")[0] == 0.02 and sb3.env_is_wrapped(object) == [False] * n
    sb3.close(); b_env.close()


def test_cap_size_noise_matches_oracle_fp64(oracle):
    """tr_env's use_cap_size_noise (tr_env.py:143,163,685-706,712-713): reset_model() first draws ONE radius ~ U(cap_size_noise_range) for the
    episode's six cap spheres (collision radius only).  The reference's own code raises on the mujoco bindings it pins (`model.geom_names`);
    oracle and HIP path build its evident intent with the draw taken from philox counter 5.  fp64: the radii agree, they cover the range,
    the settled reset states agree (robots rest on caps of different sizes: cap heights above the floor follow the radius), and steps on
    identical (state, action) agree like the other modes."""
    from tensegrity_rl_amd import TensegrityVecEnv
    from gpu_common import rel_env
    O = oracle
    n = 64
    kw = dict(use_cap_size_noise=1)
    env = TensegrityVecEnv(n, "tr_env-v0", precision="f64", auto_reset=False, seed=13, use_cap_size_noise=True)
    obs = env.reset()
    ob = O.OracleBatch(O.make_cfg("tr_env", "straight", **kw), n, seed=13)
    oobs = ob.reset()
    st = env.get_state()
    r_o = np.array([ob.env(i).model.cap_radius for i in range(n)])
    r_g = st["aux"][:, AUX["cap_radius"]]
    assert np.abs(r_g - r_o).max() < 1e-12 and r_o.min() > 0.04 and r_o.max() < 0.09 and r_o.max() - r_o.min() > 0.03
    q = ob.get_state()[0]
    assert np.abs(qfix(st["qpos"], q) - q).max() < 2e-4 and np.abs(obs - oobs).max() < 2e-3
    # the lowest cap of every robot sits one (its own) radius above the floor, minus a fraction of a millimetre of penetration
    caps = []
    for i in range(n):
        caps.append(O.arr(ob.env(i).data.cap_xpos)[:, 2].min())
    assert np.abs(np.array(caps) - r_o).max() < 1e-3
    rng = np.random.default_rng(4)
    for t in range(4):
        oracle_to_gpu_state(ob, env)
        a = rng.uniform(-0.45, 0.15, (n, 6)).astype(np.float32)
        o_obs, o_rew = ob.step(a, auto_reset=False)[:2]
        g_obs, g_rew = env.step(a)[:2]
        st = env.get_state(); q2, v2 = ob.get_state()[:2]
        assert rel_env(qfix(st["qpos"], q2) - q2, q2, 1e-3).max() < 1e-7 and rel_env(st["qvel"] - v2, v2, 1e-2).max() < 1e-5
        assert np.abs(g_obs - o_obs).max() < 1e-5 and np.abs(g_rew - o_rew).max() < 1e-6
    env.close(); ob.close()


def test_cap_size_noise_with_auto_reset_redraws_per_episode():
    """production precision, auto-reset: every episode an env receives from a donor lane carries its own radius inside the range, the
    rollout stays finite, and a handle without the option keeps the model's 0.0675 m"""
    from tensegrity_rl_amd import TensegrityVecEnv
    n = 256
    env = TensegrityVecEnv(n, "tr_env-v0", seed=2, max_episode_steps=5, use_cap_size_noise=True, cap_size_noise_range=(0.05, 0.08))
    env.reset()
    r0 = env.get_state()["aux"][:, AUX["cap_radius"]].copy()
    a = np.full((n, 6), -0.1, dtype=np.float32)
    for t in range(5):
        obs, rew, dones, infos = env.step(a)
    assert dones.all() and np.isfinite(obs).all()
    r1 = env.get_state()["aux"][:, AUX["cap_radius"]]
    assert (r1 > 0.05).all() and (r1 < 0.08).all() and np.mean(np.abs(r1 - r0) > 1e-4) > 0.9
    env.close()
    plain = TensegrityVecEnv(8, "tr_env-v0", seed=2, auto_reset=False)
    plain.reset()
    assert np.abs(plain.get_state()["aux"][:, AUX["cap_radius"]] - 0.0675).max() < 1e-9
    plain.close()
