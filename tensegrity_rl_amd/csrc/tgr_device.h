// tgr_device.h -- per-lane physics of the batched tensegrity simulator (gfx950 / CDNA4).
//
// One wavefront lane integrates one environment.  Everything a sub-step needs lives in that
// lane's VGPRs; only the tendon Jacobian pieces that must survive the contact solve are parked
// in LDS ([slot][lane] layout: conflict-free ds_read/ds_write_b32).
//
// This is NOT a port of MuJoCo and shares no code with oracle/ : the sub-step is formulated for
// this one model in world-frame twist coordinates (v, w_world) per bar, with
//   * mass matrix  diag(m I3, Ixx I3 + (Izz-Ixx) z z^T)   (bars are axially symmetric)
//   * contact rows written frame-free (normal n only; friction is isotropic in the tangent plane)
//   * the elliptic-cone Hessian as "isotropic + four rank-1" updates of an 18x18 register matrix
//   * implicitfast (M + h sum b J J^T) solved by a 9x9 Woodbury system over the nine tendons
// It reproduces what the reference computes through mujoco==2.3.7 mj_step (reference call sites
// tensegrity_env.py:297, tr_env.py:346,812; model 3prism_jonathan_steady_side.xml).
#pragma once
#include <hip/hip_runtime.h>

namespace tgr {

#define TGR_DEV __device__ __forceinline__

// ------------------------------------------------------------------ scalar helpers
template <typename T> TGR_DEV T rcp_(T x);
template <> TGR_DEV float rcp_<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> TGR_DEV double rcp_<double>(double x) { return 1.0 / x; }
template <typename T> TGR_DEV T sqrt_(T x);
template <> TGR_DEV float sqrt_<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> TGR_DEV double sqrt_<double>(double x) { return sqrt(x); }
template <typename T> TGR_DEV T rsqrt_(T x);
template <> TGR_DEV float rsqrt_<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <> TGR_DEV double rsqrt_<double>(double x) { return 1.0 / sqrt(x); }
template <typename T> TGR_DEV T abs_(T x) { return x < T(0) ? -x : x; }
template <typename T> TGR_DEV T max_(T a, T b) { return a > b ? a : b; }
template <typename T> TGR_DEV T min_(T a, T b) { return a < b ? a : b; }
template <typename T> TGR_DEV T clamp_(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <typename T> struct V3 { T x, y, z; };
template <typename T> TGR_DEV V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> TGR_DEV V3<T> operator+(V3<T> a, V3<T> b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> TGR_DEV V3<T> operator-(V3<T> a, V3<T> b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> TGR_DEV V3<T> operator-(V3<T> a) { return mk(-a.x, -a.y, -a.z); }
template <typename T> TGR_DEV V3<T> operator*(T s, V3<T> a) { return mk(s * a.x, s * a.y, s * a.z); }
template <typename T> TGR_DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> TGR_DEV V3<T> cross(V3<T> a, V3<T> b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> TGR_DEV T comp(V3<T> a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// ------------------------------------------------------------------ model parameters (host fills; kernel arg)
template <typename T> struct Params {
  T h, grav, mass, Ixx, Izz;
  T cap_r, bar_half, rcyl_r, bcyl_r, bcyl_lo, bcyl_hi, site_off, site_z;
  T ten_k[9], ten_b[9], ten_L0[9];
  T gain, bias0, bias1, bias2, force_lo, force_hi, ctrl_lo, ctrl_hi;
  T K, B;                                 // contact reference: aref = -B v - K imp dist
  T imp0, imp1, imp_width, imp_mid;       // solimp (power 2)
  T f0, f1sq, f3sq, f4sq, impratio;       // friction (slide, slide^2, torsion^2, roll^2)
  T invweight;                            // 1/m : body_invweight0 translational
  T tol, ls_tol;
  int max_iter, ls_iter, skip_clamped;
};

// static model tables (3prism_jonathan_steady_side.xml:82-87,127-164)
struct Tab {
  // site signs (x,y,z) in units of (site_off, site_off, site_z)
  static constexpr int S[6][3] = {{-1, 0, 1}, {0, 1, 1}, {1, 0, 1}, {0, 1, -1}, {1, 0, -1}, {0, -1, -1}};
  // tendon: bar_a, site_a, bar_b, site_b
  static constexpr int TEN[9][4] = {{0, 1, 2, 0}, {0, 0, 1, 1}, {1, 0, 2, 1}, {0, 5, 2, 4}, {0, 4, 1, 5},
                                    {1, 4, 2, 5}, {0, 3, 2, 2}, {0, 2, 1, 3}, {1, 2, 2, 3}};
};

// ------------------------------------------------------------------ per-lane simulation state
template <typename T> struct Sim {
  V3<T> x[3];   // bar COM
  T q[3][4];    // orientation (w,x,y,z)
  V3<T> v[3];   // linear velocity (world)
  V3<T> w[3];   // angular velocity (WORLD frame; qvel stores body frame)
  V3<T> al[3];  // warm-start acceleration, linear
  V3<T> aa[3];  // warm-start acceleration, angular (world frame)
  T ctrl[6];
};

// quantities of the forward pass of the LAST sub-step (what the reference reads after mj_step: one sub-step old)
template <typename T> struct Lag {
  V3<T> x[3];
  V3<T> z[3];
  T q[3][4];
  T tenL[9];
  T cfrc_absmax;
  int overflow;
};

template <typename T> struct Rot { V3<T> ex, ey, ez; };  // columns of the rotation matrix

template <typename T> TGR_DEV void normalize_quat(T* q) {
  T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  T s = rsqrt_(n2);
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
template <typename T> TGR_DEV Rot<T> quat_rot(const T* q) {
  T q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  T q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  T q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  Rot<T> R;
  R.ex = mk(q00 + q11 - q22 - q33, T(2) * (q12 + q03), T(2) * (q13 - q02));
  R.ey = mk(T(2) * (q12 - q03), q00 - q11 + q22 - q33, T(2) * (q23 + q01));
  R.ez = mk(T(2) * (q13 + q02), T(2) * (q23 - q01), q00 - q11 - q22 + q33);
  return R;
}
template <typename T> TGR_DEV V3<T> rot_apply(const Rot<T>& R, V3<T> l) { return l.x * R.ex + (l.y * R.ey + l.z * R.ez); }
template <typename T> TGR_DEV V3<T> rot_applyT(const Rot<T>& R, V3<T> g) { return mk(dot(R.ex, g), dot(R.ey, g), dot(R.ez, g)); }

// inverse inertia of an axially symmetric bar, world frame
template <typename T> TGR_DEV V3<T> inv_inertia(const Params<T>& P, V3<T> z, V3<T> t) {
  T iI = rcp_(P.Ixx), d = rcp_(P.Izz) - iI;
  return iI * t + (d * dot(z, t)) * z;
}
template <typename T> TGR_DEV V3<T> inertia_mul(const Params<T>& P, V3<T> z, V3<T> a) {
  return P.Ixx * a + ((P.Izz - P.Ixx) * dot(z, a)) * z;
}

// ------------------------------------------------------------------ contact slots
// slots 0..5 : floor vs cap k (bar k/2, +z end for even k)      body A = world
// slots 6..11: bar pair (0,1),(0,2),(1,2) x {r-cylinder pair, deepest other geom pair}
constexpr int NSLOT = 12;
template <typename T> struct Contact {
  V3<T> n;       // normal, from body A to body B
  V3<T> rA, rB;  // contact point relative to the COMs
  T c0;          // K * imp * dist  (<0 when penetrating)
  T Dn;          // 1/R of the normal row
  bool active;
};

template <typename T> TGR_DEV T impedance(const Params<T>& P, T dist) {
  T xx = abs_(dist) * rcp_(P.imp_width);
  T y;
  if (xx >= T(1)) y = T(1);
  else if (xx <= P.imp_mid) y = xx * xx * rcp_(P.imp_mid);
  else y = T(1) - (T(1) - xx) * (T(1) - xx) * rcp_(T(1) - P.imp_mid);
  return P.imp0 + y * (P.imp1 - P.imp0);
}
template <typename T> TGR_DEV void contact_params(const Params<T>& P, T dist, int nbody, Contact<T>& c) {
  T imp = impedance(P, dist);
  T diagApprox = P.invweight * T(nbody);
  T R0 = max_(T(1e-15), (T(1) - imp) * diagApprox * rcp_(imp));
  c.Dn = rcp_(R0);
  c.c0 = P.K * imp * dist;
}

// what the cone says at one jar: force, and the Hessian in "isotropic + rank-1" form
template <typename T> struct Cone {
  T cost;
  V3<T> fl, fa;            // efc force as world vectors (linear rows, angular rows)
  T a1, a2, b1, b2;        // C = blkdiag(a1 I + a2 nn^T, b1 I + b2 nn^T) + cg g g^T + cw w w^T
  T cg, cw;
  V3<T> gl, ga, wl, wa;
};

template <typename T, bool HESS>
TGR_DEV Cone<T> cone_eval(const Params<T>& P, const Contact<T>& c, V3<T> jl, V3<T> ja) {
  Cone<T> o;
  T mu = P.f0 * rsqrt_(P.impratio);
  T Dt = c.Dn * P.impratio;
  T rf0 = rcp_(P.f0 * P.f0);
  T Do = Dt * P.f3sq * rf0, Dr = Dt * P.f4sq * rf0;
  T jn = dot(c.n, jl);
  V3<T> jt = jl - jn * c.n;
  T jo = dot(c.n, ja);
  V3<T> jr = ja - jo * c.n;
  T tt = P.f1sq * dot(jt, jt) + P.f3sq * jo * jo + P.f4sq * dot(jr, jr);
  T Tn = sqrt_(tt);
  T N = mu * jn;
  bool top = (N >= mu * Tn) || !c.active;
  bool bottom = !top && (mu * N + Tn <= T(0));
  bool middle = !top && !bottom;
  // bottom (quadratic) zone
  T costB = T(0.5) * (c.Dn * jn * jn + Dt * dot(jt, jt) + Do * jo * jo + Dr * dot(jr, jr));
  V3<T> flB = -((c.Dn * jn) * c.n + Dt * jt);
  V3<T> faB = -((Do * jo) * c.n + Dr * jr);
  // middle (cone) zone
  T Tsafe = max_(Tn, T(1e-30));
  T rT = rcp_(Tsafe);
  T Dm = c.Dn * rcp_(mu * mu * (T(1) + mu * mu));
  T NT = N - mu * Tn;
  T costM = T(0.5) * Dm * NT * NT;
  T fn = -Dm * NT * mu;
  V3<T> wl = P.f1sq * jt;
  V3<T> wa = (P.f3sq * jo) * c.n + P.f4sq * jr;
  V3<T> flM = fn * c.n - (fn * rT) * wl;
  V3<T> faM = -(fn * rT) * wa;
  T zero = T(0);
  o.cost = bottom ? costB : (middle ? costM : zero);
  o.fl = mk(bottom ? flB.x : (middle ? flM.x : zero), bottom ? flB.y : (middle ? flM.y : zero), bottom ? flB.z : (middle ? flM.z : zero));
  o.fa = mk(bottom ? faB.x : (middle ? faM.x : zero), bottom ? faB.y : (middle ? faM.y : zero), bottom ? faB.z : (middle ? faM.z : zero));
  if (HESS) {
    T kap = -Dm * mu * NT * rT;  // >= 0 in the middle zone
    o.a1 = bottom ? Dt : (middle ? kap * P.f1sq : zero);
    o.a2 = bottom ? (c.Dn - Dt) : (middle ? -kap * P.f1sq : zero);
    o.b1 = bottom ? Dr : (middle ? kap * P.f4sq : zero);
    o.b2 = bottom ? (Do - Dr) : (middle ? kap * (P.f3sq - P.f4sq) : zero);
    o.cg = middle ? Dm : zero;
    o.cw = middle ? -kap * rT * rT : zero;
    o.gl = mu * c.n - (mu * rT) * wl;
    o.ga = -(mu * rT) * wa;
    o.wl = wl;
    o.wa = wa;
  }
  return o;
}

// ------------------------------------------------------------------ 18x18 symmetric matrix in registers (lower triangle)
constexpr __host__ __device__ int hidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <typename T, int X, int HS = 1> TGR_DEV void h_rank1_self(T* H, T c, const T* u) {
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) H[HS * hidx(6 * X + i, 6 * X + j)] += c * u[i] * u[j];
}
template <typename T, int A, int B, int HS = 1> TGR_DEV void h_rank1_cross(T* H, T c, const T* uA, const T* uB) {  // A < B
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) H[HS * hidx(6 * B + i, 6 * A + j)] += c * uB[i] * uA[j];
}
// isotropic pieces:  a1 * Jlin^T Jlin + b1 * Jang^T Jang  for (rows of body X, cols of body Y), sign s
template <typename T, int X, int Y, int HS = 1> TGR_DEV void h_iso(T* H, T s, T a1, T b1, V3<T> rX, V3<T> rY) {
  T sa = s * a1, sb = s * b1;
  // lin-lin
#pragma unroll
  for (int i = 0; i < 3; i++) H[HS * hidx(6 * X + i, 6 * Y + i)] += sa;
  // ang(X)-lin(Y): [rX]x
  const T cx[3][3] = {{T(0), -rX.z, rX.y}, {rX.z, T(0), -rX.x}, {-rX.y, rX.x, T(0)}};
  const T cy[3][3] = {{T(0), -rY.z, rY.y}, {rY.z, T(0), -rY.x}, {-rY.y, rY.x, T(0)}};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (i != j) H[HS * hidx(6 * X + 3 + i, 6 * Y + j)] += sa * cx[i][j];
      // lin(X)-ang(Y): -[rY]x   (only stored when it is in the lower triangle, i.e. X > Y)
      if (X != Y && i != j) H[HS * hidx(6 * X + i, 6 * Y + 3 + j)] += -sa * cy[i][j];
    }
  // ang-ang: (rX.rY) I - rY rX^T
  T d = dot(rX, rY);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (X == Y && j > i) continue;
      T e = -comp(rY, i) * comp(rX, j) + (i == j ? d + T(0) : T(0));
      H[HS * hidx(6 * X + 3 + i, 6 * Y + 3 + j)] += sa * e + (i == j ? sb : T(0));
    }
}

// body-space image of a contact-space vector (cl, ca) :  (cl, r x cl + ca)
template <typename T> TGR_DEV void body_vec(V3<T> cl, V3<T> ca, V3<T> r, T s, T* u) {
  V3<T> t = cross(r, cl) + ca;
  u[0] = s * cl.x; u[1] = s * cl.y; u[2] = s * cl.z; u[3] = s * t.x; u[4] = s * t.y; u[5] = s * t.z;
}

// in-place Cholesky of the packed lower triangle, n x n
template <typename T, int NN, int HS = 1> TGR_DEV void chol_factor(T* H) {
#pragma unroll
  for (int j = 0; j < NN; j++) {
    // right-looking: column j is final when we get here
    T col[NN];
    T s = max_(H[HS * hidx(j, j)], T(1e-30));
    T il = rsqrt_(s);
    H[HS * hidx(j, j)] = il;  // store 1/L_jj
#pragma unroll
    for (int i = j + 1; i < NN; i++) { col[i] = H[HS * hidx(i, j)] * il; H[HS * hidx(i, j)] = col[i]; }
#pragma unroll
    for (int i = j + 1; i < NN; i++)
#pragma unroll
      for (int k = j + 1; k <= i; k++) H[HS * hidx(i, k)] -= col[i] * col[k];
  }
}
template <typename T, int NN, int HS = 1> TGR_DEV void chol_solve(const T* L, T* x) {
#pragma unroll
  for (int i = 0; i < NN; i++) {
    T s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[HS * hidx(i, k)] * x[k];
    x[i] = s * L[HS * hidx(i, i)];
  }
#pragma unroll
  for (int i = NN - 1; i >= 0; i--) {
    T s = x[i];
#pragma unroll
    for (int k = i + 1; k < NN; k++) s -= L[HS * hidx(k, i)] * x[k];
    x[i] = s * L[HS * hidx(i, i)];
  }
}

// ------------------------------------------------------------------ contact geometry between two bars
template <typename T> struct Cand { T dist; V3<T> n, p; bool ok; };

template <typename T> TGR_DEV void keep_deeper(Cand<T>& best, int& count, T dist, V3<T> n, V3<T> p, bool valid) {
  bool hit = valid && dist < T(0);
  count += hit ? 1 : 0;
  bool take = hit && (!best.ok || dist < best.dist);
  best.dist = take ? dist : best.dist;
  best.n = mk(take ? n.x : best.n.x, take ? n.y : best.n.y, take ? n.z : best.n.z);
  best.p = mk(take ? p.x : best.p.x, take ? p.y : best.p.y, take ? p.z : best.p.z);
  best.ok = best.ok || hit;
}

// sphere (centre c, body "S") against the coaxial cylinders of the other bar (centre xo, axis zo).
// flip=false: sphere belongs to body A (normal = sphere->cyl), flip=true: sphere belongs to body B.
template <typename T>
TGR_DEV void sphere_vs_bar(const Params<T>& P, V3<T> c, V3<T> xo, V3<T> zo, bool flip, Cand<T>& best, int& count) {
  V3<T> wv = c - xo;
  T t = dot(wv, zo);
  V3<T> rad = wv - t * zo;
  T rho = sqrt_(dot(rad, rad));
  T irho = rcp_(max_(rho, T(1e-30)));
  // three cylinders: r (|t|<=half, R=rcyl), b+ (t in [lo,hi]), b- (t in [-hi,-lo])
#pragma unroll
  for (int k = 0; k < 3; k++) {
    T cen = k == 0 ? T(0) : (k == 1 ? T(0.5) * (P.bcyl_hi + P.bcyl_lo) : -T(0.5) * (P.bcyl_hi + P.bcyl_lo));
    T hh = k == 0 ? P.bar_half : T(0.5) * (P.bcyl_hi - P.bcyl_lo);
    T rc = k == 0 ? P.rcyl_r : P.bcyl_r;
    T tl = t - cen;
    T tc = clamp_(tl, -hh, hh);
    T rr = min_(rho, rc);
    bool inside = (abs_(tl) <= hh) && (rho <= rc);
    // closest point of the solid cylinder
    V3<T> q = xo + (cen + tc) * zo + (rr * irho) * rad;
    V3<T> dif = q - c;
    T dd = sqrt_(dot(dif, dif));
    T idd = rcp_(max_(dd, T(1e-30)));
    T dist = dd - P.cap_r;
    V3<T> n = idd * dif;
    // centre inside the cylinder: leave through the side wall
    T distI = rho - rc - P.cap_r;
    V3<T> nI = -irho * rad;
    dist = inside ? distI : dist;
    n = mk(inside ? nI.x : n.x, inside ? nI.y : n.y, inside ? nI.z : n.z);
    V3<T> pos = c + (P.cap_r + T(0.5) * dist) * n;
    bool valid = inside ? (rho > T(1e-12)) : (dd > T(1e-12));
    V3<T> nn = flip ? -n : n;
    keep_deeper(best, count, dist, nn, pos, valid);
  }
}

template <typename T>
TGR_DEV void bar_pair_contacts(const Params<T>& P, V3<T> xA, V3<T> zA, V3<T> xB, V3<T> zB, Contact<T>& c0, Contact<T>& c1,
                               int& dropped) {
  c0.active = false; c1.active = false;
  c0.n = mk(T(0), T(0), T(1)); c1.n = c0.n;
  c0.rA = c0.rB = c1.rA = c1.rB = mk(T(0), T(0), T(0));
  c0.c0 = c1.c0 = T(0); c0.Dn = c1.Dn = T(0);
  // line-line closest points of the two axes
  V3<T> wv = xA - xB;
  T b = dot(zA, zB), dA = dot(zA, wv), dB = dot(zB, wv);
  T den = T(1) - b * b;
  bool skew = den > T(1e-6);
  T iden = rcp_(skew ? den : T(1));
  T s = (b * dB - dA) * iden, t = (dB - b * dA) * iden;
  V3<T> pa = xA + s * zA, pb = xB + t * zB;
  V3<T> dif = pb - pa;
  T dd = sqrt_(dot(dif, dif));
  T idd = rcp_(max_(dd, T(1e-30)));
  V3<T> n = idd * dif;
  bool okdir = skew && dd > T(1e-9);
  // cheap cull: clamped segment-segment distance against the largest reach (two cap spheres)
  {
    T sc = clamp_(skew ? s : T(0), -P.bar_half, P.bar_half);
    T tc = clamp_(dB + b * sc, -P.bar_half, P.bar_half);
    sc = clamp_(b * tc - dA, -P.bar_half, P.bar_half);
    V3<T> dv = (xB + tc * zB) - (xA + sc * zA);
    if (dot(dv, dv) > (T(2) * P.cap_r + T(1e-6)) * (T(2) * P.cap_r + T(1e-6))) return;
  }
  // r-cylinder vs r-cylinder: the frequent one, own slot
  {
    bool interior = okdir && abs_(s) < P.bar_half && abs_(t) < P.bar_half;
    T dist = dd - T(2) * P.rcyl_r;
    if (interior && dist < T(0)) {
      c0.active = true;
      c0.n = n;
      V3<T> pos = pa + (P.rcyl_r + T(0.5) * dist) * n;
      c0.rA = pos - xA; c0.rB = pos - xB;
      contact_params(P, dist, 2, c0);
    }
  }
  // everything else: keep the deepest, count the rest
  Cand<T> best; best.ok = false; best.dist = T(0); best.n = c0.n; best.p = xA;
  int count = 0;
  T bc = T(0.5) * (P.bcyl_hi + P.bcyl_lo), bh = T(0.5) * (P.bcyl_hi - P.bcyl_lo);
  // cylinder pairs other than (r,r): same closest points, interior test per sub-segment
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (i == 0 && j == 0) continue;
      T ci = i == 0 ? T(0) : (i == 1 ? bc : -bc), hi_ = i == 0 ? P.bar_half : bh, ri = i == 0 ? P.rcyl_r : P.bcyl_r;
      T cj = j == 0 ? T(0) : (j == 1 ? bc : -bc), hj_ = j == 0 ? P.bar_half : bh, rj = j == 0 ? P.rcyl_r : P.bcyl_r;
      bool interior = okdir && abs_(s - ci) < hi_ && abs_(t - cj) < hj_;
      T dist = dd - ri - rj;
      V3<T> pos = pa + (ri + T(0.5) * dist) * n;
      keep_deeper(best, count, dist, n, pos, interior);
    }
  // spheres
#pragma unroll
  for (int ea = 0; ea < 2; ea++) {
    V3<T> ca = xA + (ea == 0 ? P.bar_half : -P.bar_half) * zA;
    sphere_vs_bar(P, ca, xB, zB, false, best, count);
#pragma unroll
    for (int eb = 0; eb < 2; eb++) {
      V3<T> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
      V3<T> dv = cb - ca;
      T cd = sqrt_(dot(dv, dv));
      T dist = cd - T(2) * P.cap_r;
      V3<T> nn = rcp_(max_(cd, T(1e-30))) * dv;
      V3<T> pos = ca + (P.cap_r + T(0.5) * dist) * nn;
      keep_deeper(best, count, dist, nn, pos, cd > T(1e-12));
    }
  }
#pragma unroll
  for (int eb = 0; eb < 2; eb++) {
    V3<T> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
    sphere_vs_bar(P, cb, xA, zA, true, best, count);
  }
  if (best.ok) {
    c1.active = true;
    c1.n = best.n;
    c1.rA = best.p - xA; c1.rB = best.p - xB;
    contact_params(P, best.dist, 2, c1);
  }
  dropped += count > 1 ? count - 1 : 0;
}

// ------------------------------------------------------------------ slot evaluation helpers
template <int SLOT> struct SlotBodies {
  static constexpr int A = SLOT < 6 ? -1 : (SLOT < 10 ? 0 : 1);
  static constexpr int B = SLOT < 6 ? SLOT / 2 : (SLOT < 8 ? 1 : 2);
};

template <typename T, int SLOT>
TGR_DEV void slot_jar(const Contact<T>& c, const V3<T>* yl, const V3<T>* ya, V3<T>& jl, V3<T>& ja) {
  constexpr int A = SlotBodies<SLOT>::A, B = SlotBodies<SLOT>::B;
  jl = yl[B] + cross(ya[B], c.rB);
  ja = ya[B];
  if (A >= 0) {
    jl = jl - (yl[A < 0 ? 0 : A] + cross(ya[A < 0 ? 0 : A], c.rA));
    ja = ja - ya[A < 0 ? 0 : A];
  }
}

// ------------------------------------------------------------------ the sub-step
// lds: per-lane scratch, element k of lane l at lds[k * 64 + l]; needs >= 99 slots
template <typename T>
TGR_DEV void substep(const Params<T>& P, Sim<T>& S, T* lds, int lane, bool last, Lag<T>& lag) {
  // ---- kinematics
  Rot<T> R[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    normalize_quat(S.q[b]);
    R[b] = quat_rot(S.q[b]);
  }
  if (last) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
      lag.x[b] = S.x[b];
      lag.z[b] = R[b].ez;
#pragma unroll
      for (int i = 0; i < 4; i++) lag.q[b][i] = S.q[b][i];
    }
  }
  // ---- tendons: smooth wrench per bar, Jacobian pieces parked in LDS
  V3<T> F[3], Tq[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    F[b] = mk(T(0), T(0), -P.mass * P.grav);
    // gyroscopic: -w x (I w) = -(Izz-Ixx)(z.w)(w x z)
    T zw = dot(R[b].ez, S.w[b]);
    Tq[b] = (-(P.Izz - P.Ixx) * zw) * cross(S.w[b], R[b].ez);
  }
#pragma unroll
  for (int t = 0; t < 9; t++) {
    constexpr int dummy = 0; (void)dummy;
    const int ba = Tab::TEN[t][0], sa = Tab::TEN[t][1], bb = Tab::TEN[t][2], sb = Tab::TEN[t][3];
    V3<T> ra = rot_apply(R[ba], mk(T(Tab::S[sa][0]) * P.site_off, T(Tab::S[sa][1]) * P.site_off, T(Tab::S[sa][2]) * P.site_z));
    V3<T> rb = rot_apply(R[bb], mk(T(Tab::S[sb][0]) * P.site_off, T(Tab::S[sb][1]) * P.site_off, T(Tab::S[sb][2]) * P.site_z));
    V3<T> dif = (S.x[bb] + rb) - (S.x[ba] + ra);
    T L2 = dot(dif, dif);
    T iL = rsqrt_(max_(L2, T(1e-30)));
    T L = L2 * iL;
    V3<T> u = iL * dif;
    V3<T> va = S.v[ba] + cross(S.w[ba], ra), vb = S.v[bb] + cross(S.w[bb], rb);
    T Ld = dot(u, vb - va);
    T f, damp;
    if (t < 6) {
      T cc = clamp_(S.ctrl[t], P.ctrl_lo, P.ctrl_hi);
      T fr = P.gain * cc + P.bias0 + P.bias1 * L + P.bias2 * Ld;
      f = clamp_(fr, P.force_lo, P.force_hi);
      bool clamped = (f <= P.force_lo) || (f >= P.force_hi);
      damp = -P.bias2;
      if (P.skip_clamped && clamped) damp = T(0);
      // passive part of an actuated tendon (zero in the flat model)
      f += -P.ten_k[t] * (L - P.ten_L0[t]) - P.ten_b[t] * Ld;
      damp += P.ten_b[t];
    } else {
      f = -P.ten_k[t] * (L - P.ten_L0[t]) - P.ten_b[t] * Ld;
      damp = P.ten_b[t];
    }
    if (last) lag.tenL[t] = L;
    V3<T> fu = f * u;
    V3<T> ma = cross(ra, u), mb = cross(rb, u);
    F[bb] = F[bb] + fu; Tq[bb] = Tq[bb] + f * mb;
    F[ba] = F[ba] - fu; Tq[ba] = Tq[ba] - f * ma;
    T sc = sqrt_(P.h * damp);
    T* p = lds + (t * 10) * 64 + lane;
    p[0 * 64] = u.x; p[1 * 64] = u.y; p[2 * 64] = u.z;
    p[3 * 64] = ma.x; p[4 * 64] = ma.y; p[5 * 64] = ma.z;
    p[6 * 64] = mb.x; p[7 * 64] = mb.y; p[8 * 64] = mb.z;
    p[9 * 64] = sc;
  }
  // ---- unconstrained acceleration
  V3<T> sl[3], sa_[3];
  T im = rcp_(P.mass);
#pragma unroll
  for (int b = 0; b < 3; b++) {
    sl[b] = im * F[b];
    sa_[b] = inv_inertia(P, R[b].ez, Tq[b]);
  }
  // ---- contacts
  Contact<T> C[NSLOT];
  int dropped = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int b = k / 2;
    T sg = (k % 2 == 0) ? P.bar_half : -P.bar_half;
    V3<T> arm = sg * R[b].ez;
    T cz = S.x[b].z + arm.z;
    T dist = cz - P.cap_r;
    C[k].active = dist < T(0);
    C[k].n = mk(T(0), T(0), T(1));
    C[k].rA = mk(T(0), T(0), T(0));
    C[k].rB = mk(arm.x, arm.y, arm.z - (P.cap_r + T(0.5) * dist));
    contact_params(P, dist, 1, C[k]);
  }
  bar_pair_contacts(P, S.x[0], R[0].ez, S.x[1], R[1].ez, C[6], C[7], dropped);
  bar_pair_contacts(P, S.x[0], R[0].ez, S.x[2], R[2].ez, C[8], C[9], dropped);
  bar_pair_contacts(P, S.x[1], R[1].ez, S.x[2], R[2].ez, C[10], C[11], dropped);
  if (last) lag.overflow = dropped;

  // ---- Newton on  1/2 (a-a0)^T M (a-a0) + sum_c s_c( J (a + B v) + c0 n )
  // y = a + B v is what every contact row sees
  V3<T> al[3], aa[3];
  bool any_contact = false;
#pragma unroll
  for (int k = 0; k < NSLOT; k++) any_contact = any_contact || C[k].active;

  auto total_cost = [&](const V3<T>* xl, const V3<T>* xa) -> T {
    T cost = T(0);
    V3<T> yl[3], ya[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
      V3<T> dl = xl[b] - sl[b], da = xa[b] - sa_[b];
      cost += T(0.5) * (P.mass * dot(dl, dl) + dot(da, inertia_mul(P, R[b].ez, da)));
      yl[b] = xl[b] + P.B * S.v[b];
      ya[b] = xa[b] + P.B * S.w[b];
    }
#define TGR_SLOT_COST(K)                                                \
    if (__any(C[K].active)) {                                           \
      V3<T> jl, ja; slot_jar<T, K>(C[K], yl, ya, jl, ja);               \
      jl = jl + C[K].c0 * C[K].n;                                       \
      cost += cone_eval<T, false>(P, C[K], jl, ja).cost;                \
    }
    TGR_SLOT_COST(0) TGR_SLOT_COST(1) TGR_SLOT_COST(2) TGR_SLOT_COST(3) TGR_SLOT_COST(4) TGR_SLOT_COST(5)
    TGR_SLOT_COST(6) TGR_SLOT_COST(7) TGR_SLOT_COST(8) TGR_SLOT_COST(9) TGR_SLOT_COST(10) TGR_SLOT_COST(11)
#undef TGR_SLOT_COST
    return cost;
  };

  // warm start: the cheaper of last step's solution and the unconstrained acceleration
  {
    T cw = total_cost(S.al, S.aa);
    T cs = total_cost(sl, sa_);
    bool usew = cw < cs;
#pragma unroll
    for (int b = 0; b < 3; b++) {
      al[b] = mk(usew ? S.al[b].x : sl[b].x, usew ? S.al[b].y : sl[b].y, usew ? S.al[b].z : sl[b].z);
      aa[b] = mk(usew ? S.aa[b].x : sa_[b].x, usew ? S.aa[b].y : sa_[b].y, usew ? S.aa[b].z : sa_[b].z);
    }
  }

  V3<T> fcl[3], fca[3];  // constraint wrench per bar (J^T f)
  T cfrc_max = T(0);
  const T scale = rcp_((T(3) * P.mass + T(2) * P.Ixx + P.Izz) * T(1.0 / 6.0) * T(18));
  bool conv = !any_contact;
  for (int it = 0;; it++) {
    // ---------- evaluate gradient / Hessian / forces at (al, aa)
    T* H = lds + 92 * 64 + lane;  // 18x18 lower triangle lives in LDS, [entry][lane]
#pragma unroll
    for (int i = 0; i < 171; i++) H[64 * i] = T(0);
    T g[18];
    V3<T> yl[3], ya[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
      V3<T> dl = al[b] - sl[b], da = aa[b] - sa_[b];
      V3<T> Ida = inertia_mul(P, R[b].ez, da);
      g[6 * b + 0] = P.mass * dl.x; g[6 * b + 1] = P.mass * dl.y; g[6 * b + 2] = P.mass * dl.z;
      g[6 * b + 3] = Ida.x; g[6 * b + 4] = Ida.y; g[6 * b + 5] = Ida.z;
      yl[b] = al[b] + P.B * S.v[b];
      ya[b] = aa[b] + P.B * S.w[b];
      fcl[b] = mk(T(0), T(0), T(0)); fca[b] = fcl[b];
      // mass block
      V3<T> z = R[b].ez;
      T dI = P.Izz - P.Ixx;
      H[64 * hidx(6 * b + 0, 6 * b + 0)] = P.mass; H[64 * hidx(6 * b + 1, 6 * b + 1)] = P.mass; H[64 * hidx(6 * b + 2, 6 * b + 2)] = P.mass;
      H[64 * hidx(6 * b + 3, 6 * b + 3)] = P.Ixx + dI * z.x * z.x;
      H[64 * hidx(6 * b + 4, 6 * b + 3)] = dI * z.y * z.x;
      H[64 * hidx(6 * b + 4, 6 * b + 4)] = P.Ixx + dI * z.y * z.y;
      H[64 * hidx(6 * b + 5, 6 * b + 3)] = dI * z.z * z.x;
      H[64 * hidx(6 * b + 5, 6 * b + 4)] = dI * z.z * z.y;
      H[64 * hidx(6 * b + 5, 6 * b + 5)] = P.Ixx + dI * z.z * z.z;
    }
    cfrc_max = T(0);
    V3<T> wf[3], wt[3];  // cfrc_ext accumulators
#pragma unroll
    for (int b = 0; b < 3; b++) { wf[b] = mk(T(0), T(0), T(0)); wt[b] = wf[b]; }

#define TGR_SLOT_EVAL(K)                                                                         \
    if (__any(C[K].active)) {                                                                    \
      constexpr int A_ = SlotBodies<K>::A, B_ = SlotBodies<K>::B;                                \
      constexpr int A0 = A_ < 0 ? 0 : A_;                                                        \
      V3<T> jl, ja; slot_jar<T, K>(C[K], yl, ya, jl, ja);                                        \
      jl = jl + C[K].c0 * C[K].n;                                                                \
      Cone<T> o = cone_eval<T, true>(P, C[K], jl, ja);                                           \
      V3<T> tB = cross(C[K].rB, o.fl) + o.fa;                                                    \
      fcl[B_] = fcl[B_] + o.fl; fca[B_] = fca[B_] + tB;                                          \
      wf[B_] = wf[B_] + o.fl; wt[B_] = wt[B_] + tB;                                              \
      T uB[6], uA[6];                                                                            \
      if (A_ >= 0) {                                                                             \
        V3<T> tA = cross(C[K].rA, o.fl) + o.fa;                                                  \
        fcl[A0] = fcl[A0] - o.fl; fca[A0] = fca[A0] - tA;                                        \
        wf[A0] = wf[A0] - o.fl; wt[A0] = wt[A0] - tA;                                            \
      }                                                                                          \
      h_iso<T, B_, B_, 64>(H, T(1), o.a1, o.b1, C[K].rB, C[K].rB);                                   \
      if (A_ >= 0) {                                                                             \
        h_iso<T, A0, A0, 64>(H, T(1), o.a1, o.b1, C[K].rA, C[K].rA);                                 \
        h_iso<T, B_, A0, 64>(H, T(-1), o.a1, o.b1, C[K].rB, C[K].rA);                                \
      }                                                                                          \
      V3<T> zero3 = mk(T(0), T(0), T(0));                                                        \
      const V3<T> cls[4] = {C[K].n, zero3, o.gl, o.wl};                                          \
      const V3<T> cas[4] = {zero3, C[K].n, o.ga, o.wa};                                          \
      const T cf[4] = {o.a2, o.b2, o.cg, o.cw};                                                  \
      _Pragma("unroll") for (int r = 0; r < 4; r++) {                                            \
        body_vec(cls[r], cas[r], C[K].rB, T(1), uB);                                             \
        h_rank1_self<T, B_, 64>(H, cf[r], uB);                                                       \
        if (A_ >= 0) {                                                                           \
          body_vec(cls[r], cas[r], C[K].rA, T(-1), uA);                                          \
          h_rank1_self<T, A0, 64>(H, cf[r], uA);                                                     \
          h_rank1_cross<T, A0, (A_ >= 0 ? B_ : 1), 64>(H, cf[r], uA, uB);                            \
        }                                                                                        \
      }                                                                                          \
    }
    TGR_SLOT_EVAL(0) TGR_SLOT_EVAL(1) TGR_SLOT_EVAL(2) TGR_SLOT_EVAL(3) TGR_SLOT_EVAL(4) TGR_SLOT_EVAL(5)
    TGR_SLOT_EVAL(6) TGR_SLOT_EVAL(7) TGR_SLOT_EVAL(8) TGR_SLOT_EVAL(9) TGR_SLOT_EVAL(10) TGR_SLOT_EVAL(11)
#undef TGR_SLOT_EVAL
#pragma unroll
    for (int b = 0; b < 3; b++) {
      g[6 * b + 0] -= fcl[b].x; g[6 * b + 1] -= fcl[b].y; g[6 * b + 2] -= fcl[b].z;
      g[6 * b + 3] -= fca[b].x; g[6 * b + 4] -= fca[b].y; g[6 * b + 5] -= fca[b].z;
      cfrc_max = max_(cfrc_max, max_(max_(abs_(wf[b].x), abs_(wf[b].y)), abs_(wf[b].z)));
      cfrc_max = max_(cfrc_max, max_(max_(abs_(wt[b].x), abs_(wt[b].y)), abs_(wt[b].z)));
    }
    T gn = T(0);
#pragma unroll
    for (int i = 0; i < 18; i++) gn += g[i] * g[i];
    conv = conv || !(scale * sqrt_(gn) >= P.tol);
    if (__all(conv) || it >= P.max_iter) break;

    // ---------- Newton direction
    chol_factor<T, 18, 64>(H);
    T p[18];
#pragma unroll
    for (int i = 0; i < 18; i++) p[i] = conv ? T(0) : -g[i];
    chol_solve<T, 18, 64>(H, p);
    V3<T> pl[3], pa[3];
#pragma unroll
    for (int b = 0; b < 3; b++) { pl[b] = mk(p[6 * b], p[6 * b + 1], p[6 * b + 2]); pa[b] = mk(p[6 * b + 3], p[6 * b + 4], p[6 * b + 5]); }

    // ---------- line search on the convex 1-d restriction (safeguarded Newton on phi')
    T g1 = T(0), g2 = T(0);  // Gauss part: phi_g'(alpha) = g1 + alpha g2
#pragma unroll
    for (int b = 0; b < 3; b++) {
      V3<T> dl = al[b] - sl[b], da = aa[b] - sa_[b];
      V3<T> Ipa = inertia_mul(P, R[b].ez, pa[b]);
      g1 += P.mass * dot(pl[b], dl) + dot(Ipa, da);
      g2 += P.mass * dot(pl[b], pl[b]) + dot(Ipa, pa[b]);
    }
    auto dphi = [&](T alpha, T& d1, T& d2) {
      d1 = g1 + alpha * g2; d2 = g2;
      V3<T> xl[3], xa[3];
#pragma unroll
      for (int b = 0; b < 3; b++) { xl[b] = yl[b] + alpha * pl[b]; xa[b] = ya[b] + alpha * pa[b]; }
#define TGR_SLOT_LS(K)                                                                           \
      if (__any(C[K].active)) {                                                                  \
        V3<T> jl, ja, ql, qa;                                                                    \
        slot_jar<T, K>(C[K], xl, xa, jl, ja);                                                    \
        jl = jl + C[K].c0 * C[K].n;                                                              \
        slot_jar<T, K>(C[K], pl, pa, ql, qa);                                                    \
        Cone<T> o = cone_eval<T, true>(P, C[K], jl, ja);                                         \
        d1 -= dot(o.fl, ql) + dot(o.fa, qa);                                                     \
        T nl = dot(C[K].n, ql), na = dot(C[K].n, qa);                                            \
        T gq = dot(o.gl, ql) + dot(o.ga, qa), wq = dot(o.wl, ql) + dot(o.wa, qa);                \
        d2 += o.a1 * dot(ql, ql) + o.a2 * nl * nl + o.b1 * dot(qa, qa) + o.b2 * na * na +        \
              o.cg * gq * gq + o.cw * wq * wq;                                                   \
      }
      TGR_SLOT_LS(0) TGR_SLOT_LS(1) TGR_SLOT_LS(2) TGR_SLOT_LS(3) TGR_SLOT_LS(4) TGR_SLOT_LS(5)
      TGR_SLOT_LS(6) TGR_SLOT_LS(7) TGR_SLOT_LS(8) TGR_SLOT_LS(9) TGR_SLOT_LS(10) TGR_SLOT_LS(11)
#undef TGR_SLOT_LS
    };
    T d0, h0;
    dphi(T(0), d0, h0);
    T alpha = T(1), lo = T(0), hi = T(-1);
    bool ls_done = conv || !(d0 < T(0));
    if (ls_done) alpha = T(0);
    for (int ls = 0; ls < P.ls_iter; ls++) {
      if (__all(ls_done)) break;
      T d1, d2;
      dphi(alpha, d1, d2);
      bool ok = abs_(d1) <= P.ls_tol * abs_(d0);
      if (!ls_done) {
        if (ok) ls_done = true;
        else {
          if (d1 < T(0)) lo = alpha; else hi = alpha;
          T na = d2 > T(0) ? alpha - d1 * rcp_(d2) : T(-1);
          if (hi < T(0)) { if (!(na > lo) || na > T(10) * (lo + T(1))) na = T(2) * alpha + T(1e-3); }
          else if (!(na > lo) || !(na < hi)) na = T(0.5) * (lo + hi);
          alpha = na;
        }
      }
    }
#ifdef TGR_DEBUG_ENV
    if (blockIdx.x * 64 + lane == TGR_DEBUG_ENV) printf("it %d gn %.3e d0 %.3e alpha %.6f lo %.4f hi %.4f conv %d al0 %.5f\n", it, (double)sqrt_(gn), (double)d0, (double)alpha, (double)lo, (double)hi, (int)conv, (double)al[0].x);
#endif
#pragma unroll
    for (int b = 0; b < 3; b++) { al[b] = al[b] + alpha * pl[b]; aa[b] = aa[b] + alpha * pa[b]; }
  }
  if (last) lag.cfrc_absmax = cfrc_max;
#pragma unroll
  for (int b = 0; b < 3; b++) { S.al[b] = al[b]; S.aa[b] = aa[b]; }

  // ---- implicitfast: (M + h sum_t b_t J_t J_t^T) a_new = F_smooth + F_constraint   (Woodbury over 9 tendons)
  V3<T> yl[3], ya[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    yl[b] = im * (F[b] + fcl[b]);
    ya[b] = inv_inertia(P, R[b].ez, Tq[b] + fca[b]);
  }
  {
    // tendon Jacobians back from LDS, scaled by sqrt(h b_t);  W = M^-1 J
    V3<T> u[9], ma[9], mb[9], wa_[9], wb_[9];
    T zt[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int ba = Tab::TEN[t][0], bb = Tab::TEN[t][2];
      const T* p = lds + (t * 10) * 64 + lane;
      T sc = p[9 * 64];
      u[t] = sc * mk(p[0 * 64], p[1 * 64], p[2 * 64]);
      ma[t] = sc * mk(p[3 * 64], p[4 * 64], p[5 * 64]);
      mb[t] = sc * mk(p[6 * 64], p[7 * 64], p[8 * 64]);
      wa_[t] = inv_inertia(P, R[ba].ez, ma[t]);
      wb_[t] = inv_inertia(P, R[bb].ez, mb[t]);
      // z_t = J_t^T y  with J_t = (A: -u, -ma ; B: +u, +mb)
      zt[t] = dot(u[t], yl[bb] - yl[ba]) + dot(mb[t], ya[bb]) - dot(ma[t], ya[ba]);
    }
    T Sm[45];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        const int ai = Tab::TEN[i][0], bi = Tab::TEN[i][2], aj = Tab::TEN[j][0], bj = Tab::TEN[j][2];
        T s = i == j ? T(1) : T(0);
        // contributions where tendon i and tendon j touch the same bar; sign = product of the end signs
        if (ai == aj) s += im * dot(u[i], u[j]) + dot(ma[i], wa_[j]);
        if (bi == bj) s += im * dot(u[i], u[j]) + dot(mb[i], wb_[j]);
        if (ai == bj) s -= im * dot(u[i], u[j]) + dot(ma[i], wb_[j]);
        if (bi == aj) s -= im * dot(u[i], u[j]) + dot(mb[i], wa_[j]);
        Sm[hidx(i, j)] = s;
      }
    chol_factor<T, 9>(Sm);
    chol_solve<T, 9>(Sm, zt);
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int ba = Tab::TEN[t][0], bb = Tab::TEN[t][2];
      yl[bb] = yl[bb] - (zt[t] * im) * u[t]; ya[bb] = ya[bb] - zt[t] * wb_[t];
      yl[ba] = yl[ba] + (zt[t] * im) * u[t]; ya[ba] = ya[ba] + zt[t] * wa_[t];
    }
  }
  // ---- advance: v += h a ; x += h v ; q <- exp(h w / 2) * q   (world-frame w: left multiplication)
#pragma unroll
  for (int b = 0; b < 3; b++) {
    S.v[b] = S.v[b] + P.h * yl[b];
    S.w[b] = S.w[b] + P.h * ya[b];
    S.x[b] = S.x[b] + P.h * S.v[b];
    V3<T> w = S.w[b];
    T n2 = dot(w, w);
    // half-angle x = h |w| / 2 is ~1e-3: series for sin(x)/x and cos(x) (exact to rounding for x < 1),
    // libm only for absurd spin rates
    T hh = T(0.5) * P.h;
    T x2 = hh * hh * n2;
    T sinc, cs;
    if (x2 < T(1)) {
      sinc = T(1) + x2 * (T(-1.0 / 6) + x2 * (T(1.0 / 120) + x2 * (T(-1.0 / 5040) + x2 * (T(1.0 / 362880) + x2 * T(-1.0 / 39916800)))));
      cs = T(1) + x2 * (T(-0.5) + x2 * (T(1.0 / 24) + x2 * (T(-1.0 / 720) + x2 * (T(1.0 / 40320) + x2 * (T(-1.0 / 3628800) + x2 * T(1.0 / 479001600))))));
    } else {
      double xx = sqrt(double(x2));
      sinc = T(sin(xx) / xx); cs = T(cos(xx));
    }
    T k = hh * sinc;
    T rw = cs, rx = k * w.x, ry = k * w.y, rz = k * w.z;
    T* q = S.q[b];
    T nw = rw * q[0] - rx * q[1] - ry * q[2] - rz * q[3];
    T nx = rw * q[1] + rx * q[0] + ry * q[3] - rz * q[2];
    T ny = rw * q[2] - rx * q[3] + ry * q[0] + rz * q[1];
    T nz = rw * q[3] + rx * q[2] - ry * q[1] + rz * q[0];
    q[0] = nw; q[1] = nx; q[2] = ny; q[3] = nz;
  }
}

}  // namespace tgr
