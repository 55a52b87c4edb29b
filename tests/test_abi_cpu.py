"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/tensegrity.h declares,
agrees with the ctypes mirror of tgr_config, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tensegrity.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tgr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tensegrity_rl_amd import _native as N
    L = N.load()
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), s
    assert set(N.SYMBOLS) <= set(syms)


def test_config_layout_and_reference_defaults():
    from tensegrity_rl_amd import _native as N
    assert N.load().tgr_sizeof_config() == C.sizeof(N.Config)
    c = N.default_config(N.ENV_TENSEGRITY, "straight")  # tensegrity_env.py:160-181
    assert (c.ctrl_cost_weight, c.tendon_reset_mean, c.tendon_reset_stdev, c.tendon_max_length, c.reward_delay_steps) == (0.001, -0.15, 0.1, -0.15, 25)
    assert (c.frame_skip, c.max_episode_steps, c.healthy_reward, c.terminate_when_unhealthy) == (20, 5000, 0.1, 1)
    c = N.default_config(N.ENV_TR, "turn")  # tr_env.py:137-173
    assert (c.ctrl_cost_weight, c.tendon_reset_mean, c.tendon_reset_stdev, c.tendon_max_length, c.reward_delay_steps) == (0.01, 0.15, 0.2, 0.15, 1)
    assert (c.contact_force_lo, c.contact_force_hi) == (-1000.0, 1000.0)
    assert list(c.ctrl_range) == [-0.45, 0.15] and list(c.solref) == [-5000, -500]


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tensegrity_rl_amd import TensegrityVecEnv, _native as N
    with pytest.raises(RuntimeError):
        TensegrityVecEnv(4)
    h = C.c_void_p()
    cfg = N.default_config(0, "straight")
    rc = N.load().tgr_create(C.byref(cfg), 4, 0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in N.load().tgr_last_error(None)


def test_make_config_kwargs_follow_reference_signatures():
    from tensegrity_rl_amd import make_config
    c = make_config("tr_env-v0", desired_action="straight", desired_direction=-1, terminate_when_unhealthy=False,
                    xml_file="x.xml", render_mode="None", is_test=False, use_cap_velocity=False)
    assert c.env_kind == 1 and c.desired_direction == -1 and c.terminate_when_unhealthy == 0 and c.use_cap_velocity == 0
    c = make_config("tensegrity_env-v0", desired_action="turn", reward_delay_seconds=0.5, contact_force_range=(-1, 1))
    assert c.reward_delay_steps == 25 and c.contact_force_hi == 1000.0
    with pytest.raises(TypeError):
        make_config("tensegrity_env-v0", bogus=1)
    with pytest.raises(ValueError):
        make_config("nope-v0")
    c = make_config("tr_env-v0", use_obs_noise=True, obs_noise_tendon_stdev=0.03)
    assert c.use_obs_noise == 1 and c.obs_noise_tendon_stdev == 0.03 and c.obs_noise_cap_pos_stdev == 0.05
    c = make_config("tr_env-v0", xml_file="/somewhere/3prism_jonathan_steady_side_uneven_ground.xml")   # run.py --env_xml
    assert c.model_kind == 1 and c.cap_radius == 0.05 and c.site_offset == 0.05 and c.gain == 15000 and list(c.bias) == [0, 0, 0]
    assert list(c.ten_stiffness) == [10000.0] * 9 and list(c.ten_springlength) == [0.5] * 6 + [0.8] * 3 and list(c.hf_size) == [50, 50, 1, 0.1]
    assert make_config("tr_env-v0", xml_file="/somewhere/3prism_jonathan_steady_side.xml").model_kind == 0
    c = make_config("tr_env-v0", use_cap_size_noise=True, cap_size_noise_range=(0.05, 0.08))   # tr_env.py:143,163
    assert c.use_cap_size_noise == 1 and list(c.cap_size_noise_range) == [0.05, 0.08]
    assert list(make_config("tr_env-v0").cap_size_noise_range) == [0.04, 0.09] and make_config("tr_env-v0").use_cap_size_noise == 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tensegrity_rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("shares no code with oracle/", ""), f


def test_spaces_fallback_box():
    from tensegrity_rl_amd.spaces import Box
    b = Box(low=-0.45, high=0.15, shape=(6,), dtype=np.float32)
    x = b.sample()
    assert x.shape == (6,) and x.dtype == np.float32 and b.contains(x)


def test_bench_refuses_to_run_without_a_gpu():
    """the HIP path is the product: bench.py must fail loudly on a box without a device instead of timing anything else"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout) and "metric" not in r.stdout
