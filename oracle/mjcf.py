"""Emit an MJCF of the 3-bar model from the oracle's constants table (TEST / VALIDATION TOOLING, not product code).

Nothing here is copied from the reference's XML: the document is generated from `tgo_default_model()` and the site/tendon
tables of oracle/tgr_physics.c.  Two uses:
* tests/test_model_constants.py compares every number in it with the reference's XML (when /root/reference is present);
* oracle/mujoco_check.py feeds it to the real MuJoCo engine *if* `import mujoco` works on the machine (it does not in the
  build container nor on the GPU box of this round), to pin the physics restatement against the real thing.
"""
import ctypes as C

import numpy as np

from . import oracle as O


def emit_mjcf(m=None, hfield_file="terrain.png"):
    """m: tgo_model (flat: O.default_model(), uneven ground: O.uneven_model(...)).  For the uneven model the heightfield asset
    refers to `hfield_file` (an image next to the document); the elevations themselves are data, not part of the model text."""
    m = m or O.default_model()
    sl = np.array(m.site_local)                      # [6][3] local site positions
    te = np.array(m.ten_ends, dtype=np.int32)        # [9][4] {bar_a, site_a, bar_b, site_b}
    uneven = m.model_kind == 1
    f = lambda *v: " ".join(repr(float(x)) for x in v)
    q0 = np.array(m.qpos0).reshape(3, 7)
    out = ['<mujoco model="tensegrity-3bar (generated from the oracle constants)">',
           '  <compiler angle="degree" coordinate="local" inertiafromgeom="true"/>',
           f'  <option timestep="{f(m.timestep)}" gravity="0 0 {f(-m.gravity)}" cone="elliptic" solver="Newton" integrator="implicitfast" '
           f'iterations="{m.newton_maxiter}" impratio="{f(m.impratio)}"><flag frictionloss="disable"/></option>',
           '  <default>',
           f'    <geom contype="1" conaffinity="1" condim="6" solref="{f(*m.solref)}" solimp="{f(*m.solimp)}" '
           f'friction="{f(m.friction[0], m.friction[2], m.friction[3])}"/>',
           (f'    <general ctrllimited="true" ctrlrange="{f(m.ctrl_lo, m.ctrl_hi)}" dyntype="filter" dynprm="{f(m.act_tau)} 0 0" '
            f'gainprm="{f(m.gain)} 0 0" biastype="none" biasprm="0 0 0"/>') if uneven else
           (f'    <general ctrllimited="true" ctrlrange="{f(m.ctrl_lo, m.ctrl_hi)}" dyntype="none" gainprm="{f(m.gain)} 0 0" biastype="affine" '
            f'biasprm="{f(m.bias0, m.bias1, m.bias2)}" forcelimited="true" forcerange="{f(m.force_lo, m.force_hi)}"/>'),
           '  </default>']
    if uneven:
        out += ['  <asset>', f'    <hfield name="terrain" file="{hfield_file}" nrow="0" ncol="0" size="{f(*m.hf_size)}"/>', '  </asset>']
    out += ['  <worldbody>',
            (f'    <geom name="floor" type="hfield" hfield="terrain" condim="3" pos="{f(*m.hf_pos)}"/>' if uneven else
             '    <geom name="floor" type="plane" size="10 10 1" pos="0 0 0"/>')]
    # geom masses behind the oracle's mass/inertia: cylinder 1.0, caps 0.5, end cylinders 1.0 (total 4.0)
    for b in range(3):
        out.append(f'    <body name="bar{b}" pos="{f(*q0[b, :3])}" quat="{f(*q0[b, 3:])}">')
        out.append(f'      <freejoint name="free{b}"/>')
        out.append(f'      <geom name="r{b}" type="cylinder" size="{f(m.rcyl_radius, m.bar_half)}" mass="1.0"/>')
        for e, sgn in enumerate((+1, -1)):
            out.append(f'      <geom name="s{2 * b + e}" type="sphere" size="{f(m.cap_radius)}" pos="0 0 {f(sgn * m.bar_half)}" mass="0.5"/>')
            out.append(f'      <geom name="b{2 * b + e}" type="cylinder" size="{f(m.bcyl_radius)}" '
                       f'fromto="0 0 {f(sgn * m.bcyl_hi)} 0 0 {f(sgn * m.bcyl_lo)}" mass="1.0"/>')
        for s in range(6):
            out.append(f'      <site name="bar{b}_site{s}" pos="{f(*sl[s])}" size="0.01"/>')
        out.append('    </body>')
    out.append('  </worldbody>')
    out.append('  <tendon>')
    for t in range(9):
        ba, sa, bb, sb = te[t]
        passive = ""
        if m.ten_stiffness[t] or m.ten_damping[t]:
            passive = f' springlength="{f(m.ten_springlen[t])}" stiffness="{f(m.ten_stiffness[t])}" damping="{f(m.ten_damping[t])}"'
        out.append(f'    <spatial name="td_{t}"{passive}><site site="bar{ba}_site{sa}"/><site site="bar{bb}_site{sb}"/></spatial>')
    out.append('  </tendon>')
    out.append('  <actuator>')
    for a in range(6):
        out.append(f'    <general name="act_{a}" tendon="td_{a}"/>')
    out.append('  </actuator>')
    out.append('</mujoco>')
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    print(emit_mjcf())
