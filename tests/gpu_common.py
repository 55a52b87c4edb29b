"""helpers shared by the -m gpu tests"""
import numpy as np

AUX = {"xy_lag_x": 0, "xy_lag_y": 1, "psi_lag": 2, "reset_psi": 3, "ep_return": 4, "step_num": 5, "elapsed": 6, "episode": 7,
       "ring_len": 8, "waypt_x": 9, "waypt_y": 10, "ori_x": 11, "ori_y": 12, "vcmd_x": 13, "vcmd_y": 14, "vcmd_w": 15, "src": 16,
       "act": 17, "ring": 23, "cap_radius": 55}  # act: 17..22, ring: 23..54 (heading deque, oldest first)
AUX_K = 56


def qfix(a, b):
    """flip quaternion signs of a to match b (q and -q are the same rotation)"""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 3, 7).copy()
    b = np.asarray(b, dtype=np.float64).reshape(-1, 3, 7)
    s = np.sign(np.sum(a[:, :, 3:] * b[:, :, 3:], axis=2, keepdims=True))
    s[s == 0] = 1
    a[:, :, 3:] *= s
    return a.reshape(-1, 21)


def oracle_to_gpu_state(ob, env):
    """copy the oracle batch's full per-env state into the HIP handle (identical (state, action) parity)"""
    qpos, qvel, ctrl, warm, aux = ob.get_state()
    n = ob.n
    A = np.zeros((n, AUX_K))
    A[:, AUX["xy_lag_x"]] = aux["xy_lag_x"]; A[:, AUX["xy_lag_y"]] = aux["xy_lag_y"]
    A[:, AUX["reset_psi"]] = aux["reset_psi"]; A[:, AUX["step_num"]] = aux["step_num"]
    A[:, AUX["elapsed"]] = aux["elapsed"]; A[:, AUX["ep_return"]] = aux["ep_return"]
    for k in ("psi_lag", "waypt_x", "waypt_y", "ori_x", "ori_y", "vcmd_x", "vcmd_y", "vcmd_w"):
        A[:, AUX[k]] = aux[k]
    A[:, AUX["act"]:AUX["act"] + 6] = ob.last_act  # actuator activations (uneven-ground model; zero otherwise)
    A[:, AUX["ring_len"]] = aux["ring_len"]; A[:, AUX["ring"]:AUX["ring"] + 32] = aux["ring"]  # self._heading_buffer
    A[:, AUX["cap_radius"]] = [ob.env(i).model.cap_radius for i in range(n)]                   # cap-size noise (tr_env.py:685-706)
    env.set_state(qpos, qvel, ctrl, warm, A)
    return qpos, qvel


def rel_env(err, ref, floor):
    """per-env relative error: max |err| over components / max(|ref| over components, floor)"""
    return np.abs(err).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), floor)
