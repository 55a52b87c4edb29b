"""Closed-form known answers on a synthetic heightfield (shared by the CPU oracle test and the -m gpu test; no oracle in the GPU test's loop).

Terrain: a zigzag of period two columns -- elevation A on odd columns, 0 on even ones -- on the reference's grid (100 x 100 vertices over
2 x 50 m, elevation 1 m, geom at z = -1: uneven xml:48,56).  Every cell is planar (its two triangles are coplanar), the walls have slope
tan(theta) = A / dx with dx = 100 / 99 m (mujoco lays ncol vertices over 2 radius_x: ncol - 1 cells [ext]); even column lines are groove
bottoms, odd ones ridge tops.

(1) A free bar lying ALONG a groove rests on its two caps, each cap on BOTH walls: two contacts per cap with normals (-+sin, 0, cos), each
    carrying N = m g / (4 cos theta), friction-free by symmetry.  MuJoCo's soft-contact law (solref (-5000, -500), default solimp; uneven
    xml:30) gives the penetration p(N) in closed form (tests/test_gpu_known_answers.py has the flat-floor version), so the bar's axis sits
    (r_cap - p) / cos(theta) above the groove line.  Both walls are touched inside a triangle, two prisms per cap: the same contact set
    MuJoCo's one-contact-per-prism rule gives, so this is a known answer of MuJoCo's model too -- it pins the cell size, the elevation
    scale, the geom offset, the facet normals and the two-facets-per-cap path.
(2) A free bar lying ACROSS a ridge, balanced on its middle (caps in the air), rests on the ridge edge with ONE contact carrying m g:
    its axis sits r_cyl - p(m g) above the ridge line.  That is the closed form of the rule this restatement follows for a bar's cylinder
    (one contact against the deepest grid edge); MuJoCo would report one contact per prism touched (two here, normals tilted
    symmetrically), so this one pins the documented rule, not MuJoCo."""
import numpy as np

M_BAR, G, R_CAP, R_CYL, HALF = 4.0, 9.81, 0.05, 0.0381, 0.688
NGRID, RADIUS, ELEV, Z_GEOM = 100, 50.0, 1.0, -1.0
DX = 2 * RADIUS / (NGRID - 1)
A_ZIG = np.float32(0.35)


def zigzag_heights():
    h = np.zeros((NGRID, NGRID), dtype=np.float32)
    h[:, 1::2] = A_ZIG
    return h


def col_x(c):
    return -RADIUS + c * DX


def row_y(r):
    return -RADIUS + r * DX


TAN = float(A_ZIG) * ELEV / DX
COS = 1.0 / np.sqrt(1.0 + TAN * TAN)


def contact_force(p, mass=M_BAR):
    d0, dmax, width, mid = 0.9, 0.95, 0.001, 0.5
    x = min(p / width, 1.0)
    y = x * x / mid if x <= mid else 1 - (1 - x) ** 2 / (1 - mid)
    imp = d0 + y * (dmax - d0)
    K = 5000.0 / dmax ** 2
    D = imp / ((1 - imp) * (1.0 / mass))
    return D * K * imp * p


def rest_penetration(load):
    lo, hi = 0.0, 1e-3
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if contact_force(mid) < load else (lo, mid)
    return 0.5 * (lo + hi)


def quat_z_to(axis):
    """unit quaternion (w, x, y, z) turning the local z axis (the bar's axis) into `axis`"""
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    c = np.cross([0, 0, 1.0], a); w = 1.0 + a[2]
    q = np.concatenate([[w], c]) if w > 1e-9 else np.array([0.0, 1.0, 0.0, 0.0])
    return q / np.linalg.norm(q)


def groove_case(n, seed=0, lift=2e-4):
    """qpos [n, 21]: the three bars of env e along grooves (even columns), caps well inside their cells; returns (qpos, z_rest [n, 3])"""
    rng = np.random.default_rng(seed)
    p = rest_penetration(M_BAR * G / (4 * COS))
    qpos = np.zeros((n, 21)); zr = np.zeros((n, 3))
    for e in range(n):
        for b in range(3):
            c = 44 + 2 * b + 6 * int(rng.integers(0, 2))            # grooves 2 m apart: the bars do not meet
            r = 30 + int(rng.integers(0, 30))
            ylo = row_y(r) + rng.uniform(0.25, 0.38)                 # lower cap 0.25..0.38 m into its cell: the upper one lands 0.61..0.74 m into its own
            sgn = 1.0 if rng.integers(0, 2) else -1.0                # the bar's axis along +y or -y
            qpos[e, 7 * b:7 * b + 3] = [col_x(c), ylo + HALF, Z_GEOM + (R_CAP - p) / COS + lift]
            qpos[e, 7 * b + 3:7 * b + 7] = quat_z_to([0.0, sgn, 0.0])
            zr[e, b] = Z_GEOM + (R_CAP - p) / COS
    return qpos, zr


def ridge_case(n, seed=1, lift=2e-4):
    """the three bars across ridges (odd columns), balanced on their middles; returns (qpos, z_rest [n, 3])"""
    rng = np.random.default_rng(seed)
    p = rest_penetration(M_BAR * G)
    top = Z_GEOM + float(A_ZIG) * ELEV
    qpos = np.zeros((n, 21)); zr = np.zeros((n, 3))
    for e in range(n):
        for b in range(3):
            c = 45 + 2 * b
            r = 30 + 3 * b + 10 * int(rng.integers(0, 3))            # three rows apart as well
            sgn = 1.0 if rng.integers(0, 2) else -1.0
            qpos[e, 7 * b:7 * b + 3] = [col_x(c), row_y(r) + rng.uniform(0.3, 0.7), top + R_CYL - p + lift]
            qpos[e, 7 * b + 3:7 * b + 7] = quat_z_to([sgn, 0.0, 0.0])
            zr[e, b] = top + R_CYL - p
    return qpos, zr
