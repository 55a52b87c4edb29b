"""The oracle's heightfield contacts against closed-form known answers (tests/hfield_known.py): a free bar along a V-groove (two facets per
cap: also MuJoCo's contact set) and a free bar balanced across a ridge (one contact on the grid edge: the documented rule).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import hfield_known as HK


def _free_model(O):
    m = O.uneven_model(HK.zigzag_heights())
    m.gain = 0.0
    for t in range(9):
        m.ten_stiffness[t] = 0.0; m.ten_damping[t] = 0.0
    assert abs(m.mass - HK.M_BAR) < 1e-12 and m.cap_radius == HK.R_CAP and m.rcyl_radius == HK.R_CYL and m.bar_half == HK.HALF
    assert tuple(m.hf_size) == (HK.RADIUS, HK.RADIUS, HK.ELEV, 0.1) and m.hf_pos[2] == HK.Z_GEOM
    assert tuple(m.solref) == (-5000.0, -500.0) and tuple(m.solimp) == (0.9, 0.95, 0.001, 0.5, 2.0)
    return m


@pytest.mark.parametrize("case", ["groove", "ridge"])
def test_oracle_free_bar_on_a_zigzag_heightfield_rests_at_the_closed_form_height(oracle, case):
    O = oracle
    L = O.lib(); m = _free_model(O)
    qpos, zr = (HK.groove_case if case == "groove" else HK.ridge_case)(4)
    for e in range(4):
        d = O.new_data(m)
        O.arr(d.qpos)[:] = qpos[e]; O.arr(d.qvel)[:] = 0
        L.tgo_step(C.byref(m), C.byref(d), 1400)           # 1.4 s = 13 time constants of the over-damped contact
        z = O.arr(d.qpos).reshape(3, 7)[:, 2]
        assert np.abs(z - zr[e]).max() < 2e-7, (z, zr[e])
        assert np.abs(O.arr(d.qvel)).max() < 1e-6
        xy = O.arr(d.qpos).reshape(3, 7)[:, :2]
        assert np.abs(xy - qpos[e].reshape(3, 7)[:, :2]).max() < 1e-6      # nothing pushes sideways
        if case == "groove":
            assert d.ncon == 12                                             # two facets under each of the six caps
            want = HK.M_BAR * HK.G / (4 * HK.COS)
            f = np.array([d.con[k].force[0] for k in range(d.ncon)])
            assert np.abs(f - want).max() < 1e-5 * want
            nz = np.array([d.con[k].frame[2] for k in range(d.ncon)]); nx = np.array([d.con[k].frame[0] for k in range(d.ncon)])
            assert np.abs(nz - HK.COS).max() < 1e-9 and np.abs(np.abs(nx) - HK.TAN * HK.COS).max() < 1e-9 and abs(nx.sum()) < 1e-9
        else:
            assert d.ncon == 3                                              # one grid-edge contact per bar, caps in the air
            f = np.array([d.con[k].force[0] for k in range(d.ncon)])
            assert np.abs(f - HK.M_BAR * HK.G).max() < 1e-5 * HK.M_BAR * HK.G
            nz = np.array([d.con[k].frame[2] for k in range(d.ncon)])
            assert np.abs(nz - 1.0).max() < 1e-9
        # cfrc_ext of a resting bar ([world, r01, r23, r45][torque3, force3]): no torque, force m g up
        cf = np.array([[d.cfrc_ext[b + 1][k] for k in range(6)] for b in range(3)])
        assert np.abs(cf[:, 5] - HK.M_BAR * HK.G).max() < 1e-4 and np.abs(cf[:, :5]).max() < 1e-4, cf
