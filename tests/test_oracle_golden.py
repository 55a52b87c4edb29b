"""The C restatement of the Python env layer (oracle/tgr_env.c) against fixtures produced by the
reference env classes themselves (oracle/gen_golden.py, reference tensegrity_env.py:291-512, tr_env.py:327-872).

Same physics feeds both, so agreement is expected to rounding (1e-9), which pins reward / obs /
termination / reset-pose maths of the oracle before the oracle is used to check the HIP path."""
import glob
import os

import numpy as np
import pytest

INFO = ["reward_forward", "reward_ctrl", "reward_survive", "x_position", "y_position", "psi",
        "distance_from_origin", "x_velocity", "y_velocity", "forward_reward"]


def qpos_close(a, b, atol):
    """qpos comparison up to the sign of each free-joint quaternion (q and -q are the same rotation;
    the reference's reset goes through scipy euler angles, which fixes an arbitrary sign)."""
    a = np.asarray(a, dtype=np.float64).reshape(3, 7).copy()
    b = np.asarray(b, dtype=np.float64).reshape(3, 7)
    for i in range(3):
        if np.dot(a[i, 3:], b[i, 3:]) < 0:
            a[i, 3:] *= -1
    np.testing.assert_allclose(a, b, rtol=0, atol=atol)


def ring_close(O, env, ring, ring_len):
    """self._heading_buffer of the reference (a deque that is never cleared: tensegrity_env.py:243) against the restatement's ring"""
    e = env.e
    assert e.ring_len == ring_len
    mine = [O.arr(e.ring)[(e.ring_head + j) % O.RING] for j in range(e.ring_len)]
    np.testing.assert_allclose(mine, ring[:ring_len], rtol=0, atol=1e-8)


def load_cases(golden_dir):
    return sorted(f for f in glob.glob(os.path.join(golden_dir, "*.npz")) if "helpers" not in f and "obs_noise" not in f and not os.path.basename(f).startswith("policy_"))


def cfg_from_fixture(O, g):
    kw = {k: v for k, v in zip(g["kwargs_keys"], g["kwargs_vals"])}
    kw = {k: (int(v) if k in ("terminate_when_unhealthy", "use_contact_forces", "use_cap_velocity") else float(v)) for k, v in kw.items()}
    if "model_kind" in g.files and int(g["model_kind"]) == 1:   # the uneven-ground XML: heightfield from the committed terrain array
        img = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_100.npy")).astype(np.float64)
        kw["heights"] = np.ascontiguousarray(((img - img.min()) / (img.max() - img.min()))[::-1], dtype=np.float32)
        kw["model_kind"] = 1
    return O.make_cfg(str(g["kind"]), str(g["action"]), desired_direction=float(g["direction"]), **kw)


def test_fixtures_present(golden_dir):
    assert len(load_cases(golden_dir)) == 14


@pytest.mark.parametrize("idx", range(14))
def test_oracle_env_matches_reference_classes(oracle, golden_dir, idx):
    O = oracle
    path = load_cases(golden_dir)[idx]
    g = np.load(path)
    env = O.OracleEnv(cfg_from_fixture(O, g))
    T = int(g["steps_per_episode"])
    k = 0
    for ep in range(len(g["reset_theta"])):
        if "reset_noise_scale" in list(g["kwargs_keys"]):  # tensegrity_env.py:436-445: the reference's own np_random draws, replayed
            assert np.abs(g["reset_dq"][ep]).max() > 1e-3 and np.abs(g["reset_dv"][ep]).max() > 1e-3
            obs0 = env.reset_noise(int(g["reset_pose"][ep]), float(g["reset_theta"][ep]), g["reset_tendons"][ep], g["reset_dq"][ep],
                                   g["reset_dv"][ep], g["reset_u_way"][ep])
        else:
            obs0 = env.reset(int(g["reset_pose"][ep]), float(g["reset_theta"][ep]), g["reset_tendons"][ep], g["reset_u_way"][ep])
        qpos_close(O.arr(env.data.qpos), g["reset_qpos"][ep], 1e-9)
        np.testing.assert_allclose(O.arr(env.data.qvel), g["reset_qvel"][ep], rtol=0, atol=1e-8)
        assert abs(env.e.reset_psi - g["reset_reset_psi"][ep]) < 1e-9
        np.testing.assert_allclose(obs0, g["reset_obs"][ep], rtol=0, atol=1e-8)
        ring_close(O, env, g["reset_ring"][ep], int(g["reset_ring_len"][ep]))
        assert env.e.step_num == int(g["reset_step_num"][ep])
        for t in range(T):
            obs, r, term, trunc, info = env.step(g["actions"][k])
            qpos_close(O.arr(env.data.qpos), g["qpos"][k], 1e-8)
            np.testing.assert_allclose(obs, g["obs"][k], rtol=0, atol=1e-7, err_msg=f"{path} step {k}")
            assert abs(r - g["reward"][k]) < 1e-7, (path, k, r, g["reward"][k])
            assert term == bool(g["terminated"][k])
            for name in INFO:
                assert abs(info[name] - g[name][k]) < 1e-7, (path, k, name)
            ring_close(O, env, g["ring"][k], int(g["ring_len"][k]))
            assert env.e.step_num == int(g["step_num"][k])
            k += 1


def test_helper_known_answers(oracle, golden_dir):
    """tr_env.py:648-683 helper outputs recorded from the reference class; also SURVEY.md 8c probes."""
    g = np.load(os.path.join(golden_dir, "tr_env_helpers.npz"))
    assert abs(g["ditch"][0] - 80.0737402916808) < 1e-9
    assert abs(g["vel_track"][0] - 1.396163535868496) < 1e-12
    assert abs(g["ang"][0] - (-2.2831853071795862)) < 1e-12
    np.testing.assert_allclose(g["filt"], 0.002, atol=1e-15)


def test_obs_noise_application_matches_reference(oracle, golden_dir):
    """tr_env.py:549-646 with use_obs_noise=True: (clean, noisy) observation pairs produced by the reference class itself and
    the standard-normal draws it consumed; the restatement applies the same draws to the clean observation."""
    import ctypes as C
    O = oracle
    g = np.load(os.path.join(golden_dir, "tr_env_obs_noise.npz"))
    for k in range(len(g["clean"])):
        cfg = O.make_cfg("tr_env", str(g["action"][k]), use_cap_velocity=int(g["use_cap_velocity"][k]), use_obs_noise=1)
        assert cfg.obs_noise_tendon_stdev == float(g["tendon_stdev"]) and cfg.obs_noise_cap_pos_stdev == float(g["cap_pos_stdev"])
        obs = g["clean"][k].copy()
        z = np.ascontiguousarray(g["z"][k])
        O.lib().tgo_apply_obs_noise(C.byref(cfg), O.dptr(z), O.dptr(obs))
        D = O.lib().tgo_env_obs_dim(C.byref(cfg))
        np.testing.assert_allclose(obs[:D], g["noisy"][k][:D], atol=1e-12)
        assert np.abs(g["noisy"][k][:D] - g["clean"][k][:D]).max() > 1e-3


def test_obs_noise_draws_are_standard_normal_and_keyed(oracle):
    import ctypes as C
    O = oracle
    z = np.zeros(48); zs = []
    for key in range(1500):
        O.lib().tgo_obs_noise_draws(C.c_uint64(7), C.c_uint64(key), C.c_uint64(2), 5, O.dptr(z)); zs.append(z[:45].copy())
    zs = np.array(zs)
    assert abs(zs.mean()) < 0.01 and abs(zs.std() - 1) < 0.01 and abs(((zs - zs.mean()) ** 4).mean() / zs.var() ** 2 - 3) < 0.1
    a = zs[3].copy()
    O.lib().tgo_obs_noise_draws(C.c_uint64(7), C.c_uint64(3), C.c_uint64(2), 5, O.dptr(z)); assert np.array_equal(z[:45], a)
    O.lib().tgo_obs_noise_draws(C.c_uint64(7), C.c_uint64(3), C.c_uint64(2), 6, O.dptr(z)); assert not np.allclose(z[:45], a)
