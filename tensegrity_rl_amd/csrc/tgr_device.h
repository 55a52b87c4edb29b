// tgr_device.h -- per-lane physics of the batched tensegrity simulator (gfx950 / CDNA4).
//
// One wavefront lane integrates one environment.  The rigid-body state of the lane (3 bars: pose,
// twist, warm-start) stays in VGPRs for the whole control period; everything the contact solve
// works on -- the 18x18 Hessian as six 6x6 blocks, the 12 contact slots, the solver vectors and
// the tendon Jacobians -- is staged per lane in LDS as columns  lds[slot * LW + lane]  (bank =
// lane: conflict-free ds_read/ds_write_b32).  Loops over bars / bar pairs / slots are real loops,
// so the live register set is one 6x6 block plus one contact at a time (no scratch spills).
//
// This is NOT a port of MuJoCo and shares no code with oracle/ : the sub-step is formulated for
// this one model in world-frame twist coordinates (v, w_world) per bar, with
//   * mass matrix  diag(m I3, Ixx I3 + (Izz-Ixx) z z^T)   (bars are axially symmetric)
//   * contact rows written frame-free (normal n only; friction is isotropic in the tangent plane)
//   * the elliptic-cone Hessian as "isotropic + four rank-1" updates of 6x6 blocks
//   * a blocked Cholesky that skips the off-diagonal blocks no lane of the wave couples
//   * implicitfast (M + h sum b J J^T) solved by a 9x9 Woodbury system over the nine tendons
// It reproduces what the reference computes through mujoco==2.3.7 mj_step (reference call sites
// tensegrity_env.py:297, tr_env.py:346,812; model 3prism_jonathan_steady_side.xml).
#pragma once
#include <hip/hip_runtime.h>

namespace tgr {

#define TGR_DEV __device__ __forceinline__

#ifdef TGR_STAMPS
#define TGR_STAMP(i) do { long long _t = clock64(); lag.stamp[i] += _t - lag.t_last; lag.t_last = _t; } while (0)
#else
#define TGR_STAMP(i) do { } while (0)
#endif

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
template <typename T> TGR_DEV V3<T> sel(bool c, V3<T> a, V3<T> b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

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
  T cf_lo, cf_hi;                         // contact_force_range of the env (for contact_cost)
  int max_iter, ls_iter, skip_clamped, warm_start;
  // uneven-ground model only (MODEL = 1): activation filter and heightfield floor
  T act_itau;                              // 1 / dynprm[0]
  T hf_rx, hf_ry, hf_ez, hf_x0, hf_y0, hf_z0, hf_dx, hf_dy;
  int hf_nrow, hf_ncol;
  const float* hf;                         // normalised elevations [nrow][ncol] in device memory
};

// static model tables (3prism_jonathan_steady_side.xml:82-87,127-164)
struct Tab {
  // site signs (x,y,z) in units of (site_off, site_off, site_z)
  static constexpr int S[6][3] = {{-1, 0, 1}, {0, 1, 1}, {1, 0, 1}, {0, 1, -1}, {1, 0, -1}, {0, -1, -1}};
  // tendon: bar_a, site_a, bar_b, site_b
  static constexpr int TEN[9][4] = {{0, 1, 2, 0}, {0, 0, 1, 1}, {1, 0, 2, 1}, {0, 5, 2, 4}, {0, 4, 1, 5},
                                    {1, 4, 2, 5}, {0, 3, 2, 2}, {0, 2, 1, 3}, {1, 2, 2, 3}};
  // 3prism_jonathan_steady_side_uneven_ground.xml:76-81,94-99,112-117,122-158 (MODEL = 1): same bars per tendon, other sites
  static constexpr int S1[6][3] = {{-1, 0, 1}, {1, 0, 1}, {0, -1, 1}, {0, -1, -1}, {0, 1, -1}, {1, 0, -1}};
  static constexpr int TEN1[9][4] = {{0, 0, 2, 1}, {0, 2, 1, 1}, {1, 0, 2, 2}, {0, 5, 2, 3}, {0, 4, 1, 3},
                                     {1, 5, 2, 4}, {0, 3, 2, 0}, {0, 1, 1, 4}, {1, 2, 2, 5}};
};

// ------------------------------------------------------------------ per-lane LDS columns (slot indices)
// element `slot` of this lane is lds[slot * LW]  (lds already points at this lane's column)
enum : int {
  L_TEN = 0,            // 9 tendons x {u(3), rA x u(3), rB x u(3), sqrt(h b)} = 90
  L_H = 90,             // Hessian: D0,D1,D2 (21 each, packed lower) O10,O20,O21 (36 each) = 171
  L_CF = L_H + 171,     // 6 floor slots x {rB(3), c0, Dn, active} = 36
  L_CP = L_CF + 36,     // 6 pair slots x {n(3), rA(3), rB(3), c0, Dn, active} = 72
  L_A = L_CP + 72,      // current acceleration estimate, 3 bars x (lin3, ang3 world) = 18
  L_A0 = L_A + 18,      // unconstrained acceleration
  L_BV = L_A0 + 18,     // B * twist (what the reference-acceleration adds to every contact row)
  L_G = L_BV + 18,      // gradient
  L_P = L_G + 18,       // search direction
  L_FC = L_P + 18,      // constraint wrench per bar (J^T f) == cfrc_ext
  L_EZ = L_FC + 18,     // bar axes (3 x 3)
  L_X = L_EZ + 9,       // bar COM positions (3 x 3)
  L_RED = L_X + 9,      // cross-wave partial sums (35)
  L_TOTAL = L_RED + 35  // 530
};
// MODEL = 1 only (heightfield floor).  Floor contact index k: 0..5 = deepest contact of cap k (data in L_CF, as on the plane),
// 6..8 = the cylinder of bar k-6 against the terrain, 9..14 = second contact of cap k-9
constexpr int L_FN = L_TOTAL;       // 6 x n(3)                          normals of contacts 0..5
constexpr int L_FR = L_FN + 18;     // 3 x {rB(3), c0, Dn, active}       contacts 6..8
constexpr int L_FRN = L_FR + 18;    // 3 x n(3)
constexpr int L_F2 = L_FRN + 9;     // 6 x {rB(3), c0, Dn, active}       contacts 9..14
constexpr int L_F2N = L_F2 + 36;    // 6 x n(3)
template <int MODEL> constexpr __host__ __device__ int lds_slots() { return MODEL ? L_F2N + 18 : L_TOTAL; }
template <int MODEL> constexpr __host__ __device__ int n_floor() { return MODEL ? 5 : 2; }  // floor contacts per bar wave
template <int MODEL> constexpr __host__ __device__ int num_ls() { return MODEL ? 6 : 3; }   // line-search slots per wave
// e-th floor contact of bar b: its two caps; MODEL = 1: then the bar's cylinder, then the caps' second contacts
TGR_DEV int floor_index(int b, int e) { return e < 2 ? 2 * b + e : (e == 2 ? 6 + b : 9 + 2 * b + (e - 3)); }
static_assert(L_F2N + 18 <= 640, "one block's LDS columns must fit 160 KB");
constexpr __host__ __device__ int H_D(int b) { return L_H + 21 * b; }
constexpr int H_O10 = L_H + 63, H_O20 = L_H + 99, H_O21 = L_H + 135;

// ------------------------------------------------------------------ per-lane simulation state (registers)
template <typename T> struct Sim {
  V3<T> x[3];   // bar COM
  T q[3][4];    // orientation (w,x,y,z)
  V3<T> v[3];   // linear velocity (world)
  V3<T> w[3];   // angular velocity (WORLD frame; qvel stores body frame)
  V3<T> al[3];  // warm-start acceleration, linear
  V3<T> aa[3];  // warm-start acceleration, angular (world frame)
  T ctrl[6];
  T act[6];     // actuator activations (MODEL = 1: dyntype filter)
  T capd;       // this episode's cap radius minus the model's (cap-size noise, row kernels only; 0 otherwise)
};

// quantities of the forward pass of the LAST sub-step (what the reference reads after mj_step: one sub-step old)
template <typename T> struct Lag {
  // x: the lagged bar positions RELATIVE to their centre com = (x0 + x1 + x2) / 3; the centre's x, y stay in double (what the reference calls
  // xy_position: its change per control period is the velocity that is_healthy and the rewards read, tensegrity_env.py:279,311-315)
  V3<T> x[3];
  double com[2];
  V3<T> z[3];
  T q[3][4];
  T tenL[9];
  T cfrc_absmax;
  T cfrc_clip_sq;  // sum over cfrc_ext of clip(., contact_force_range)^2
  T bar_contact;   // sum over bar-bar contacts of |linear contact force| (what run.py:149-155 logs)
  int overflow;
  int n_it, n_ls;  // solver work of this control period (wave-uniform)
  int n_unconv;    // sub-steps this lane left the solver at the iteration cap
#ifdef TGR_STAMPS
  long long stamp[32]; long long t_last;  // diagnostic build only: cycles per phase
#endif
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

// LDS column access
template <typename T, int LW> TGR_DEV V3<T> ld3(const T* lds, int slot) { return mk(lds[slot * LW], lds[(slot + 1) * LW], lds[(slot + 2) * LW]); }
template <typename T, int LW> TGR_DEV void st3(T* lds, int slot, V3<T> v) { lds[slot * LW] = v.x; lds[(slot + 1) * LW] = v.y; lds[(slot + 2) * LW] = v.z; }

// ------------------------------------------------------------------ contacts
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
// floor contact k: 0..5 = cap k; MODEL = 1 additionally 6..8 = the cylinder of bar k - 6 against the terrain
template <typename T, int LW, int MODEL = 0> TGR_DEV Contact<T> load_floor(const T* lds, int k) {
  Contact<T> c;
  const int s = !MODEL || k < 6 ? L_CF + 6 * k : (k < 9 ? L_FR + 6 * (k - 6) : L_F2 + 6 * (k - 9));
  c.n = MODEL ? ld3<T, LW>(lds, k < 6 ? L_FN + 3 * k : (k < 9 ? L_FRN + 3 * (k - 6) : L_F2N + 3 * (k - 9))) : mk(T(0), T(0), T(1));
  c.rA = mk(T(0), T(0), T(0));
  c.rB = ld3<T, LW>(lds, s);
  c.c0 = lds[(s + 3) * LW]; c.Dn = lds[(s + 4) * LW]; c.active = lds[(s + 5) * LW] != T(0);
  return c;
}
template <typename T, int LW> TGR_DEV Contact<T> load_pair(const T* lds, int k) {
  Contact<T> c;
  const int s = L_CP + 12 * k;
  c.n = ld3<T, LW>(lds, s); c.rA = ld3<T, LW>(lds, s + 3); c.rB = ld3<T, LW>(lds, s + 6);
  c.c0 = lds[(s + 9) * LW]; c.Dn = lds[(s + 10) * LW]; c.active = lds[(s + 11) * LW] != T(0);
  return c;
}
template <typename T, int LW> TGR_DEV void store_pair(T* lds, int k, const Contact<T>& c) {
  const int s = L_CP + 12 * k;
  st3<T, LW>(lds, s, c.n); st3<T, LW>(lds, s + 3, c.rA); st3<T, LW>(lds, s + 6, c.rB);
  lds[(s + 9) * LW] = c.c0; lds[(s + 10) * LW] = c.Dn; lds[(s + 11) * LW] = c.active ? T(1) : T(0);
}

// what the cone says at one jar: force, and the Hessian in "isotropic + rank-1" form
template <typename T> struct Cone {
  V3<T> fl, fa;            // efc force as world vectors (linear rows, angular rows)
  T a1, a2, b1, b2;        // C = blkdiag(a1 I + a2 nn^T, b1 I + b2 nn^T) + cg g g^T + cw w w^T
  T cg, cw;
  V3<T> gl, ga, wl, wa;
};

template <typename T>
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
  V3<T> flB = -((c.Dn * jn) * c.n + Dt * jt);
  V3<T> faB = -((Do * jo) * c.n + Dr * jr);
  // middle (cone) zone
  T Tsafe = max_(Tn, T(1e-30));
  T rT = rcp_(Tsafe);
  T Dm = c.Dn * rcp_(mu * mu * (T(1) + mu * mu));
  T NT = N - mu * Tn;
  T fn = -Dm * NT * mu;
  V3<T> wl = P.f1sq * jt;
  V3<T> wa = (P.f3sq * jo) * c.n + P.f4sq * jr;
  V3<T> flM = fn * c.n - (fn * rT) * wl;
  V3<T> faM = -(fn * rT) * wa;
  T zero = T(0);
  V3<T> z3 = mk(zero, zero, zero);
  o.fl = sel(bottom, flB, sel(middle, flM, z3));
  o.fa = sel(bottom, faB, sel(middle, faM, z3));
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
  return o;
}

// ------------------------------------------------------------------ 6x6 block algebra (registers)
constexpr __host__ __device__ int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <typename T, int LW> TGR_DEV void ld21(const T* lds, int off, T* D) {
#pragma unroll
  for (int i = 0; i < 21; i++) D[i] = lds[(off + i) * LW];
}
template <typename T, int LW> TGR_DEV void st21(T* lds, int off, const T* D) {
#pragma unroll
  for (int i = 0; i < 21; i++) lds[(off + i) * LW] = D[i];
}
template <typename T, int LW> TGR_DEV void ld36(const T* lds, int off, T* O) {
#pragma unroll
  for (int i = 0; i < 36; i++) O[i] = lds[(off + i) * LW];
}
template <typename T, int LW> TGR_DEV void st36(T* lds, int off, const T* O) {
#pragma unroll
  for (int i = 0; i < 36; i++) lds[(off + i) * LW] = O[i];
}

// body-space image of a contact-space vector (cl, ca) :  (cl, r x cl + ca)
template <typename T> TGR_DEV void body_vec(V3<T> cl, V3<T> ca, V3<T> r, T* u) {
  V3<T> t = cross(r, cl) + ca;
  u[0] = cl.x; u[1] = cl.y; u[2] = cl.z; u[3] = t.x; u[4] = t.y; u[5] = t.z;
}

// D (packed lower 6x6 of one bar) += J^T C J of a contact with arm r on this bar
template <typename T> TGR_DEV void self_block_add(T* D, const Cone<T>& o, V3<T> n, V3<T> r) {
  // isotropic: a1 [I, -[r]x ; [r]x , |r|^2 I - r r^T] + b1 [0 0; 0 I]
  D[tri(0, 0)] += o.a1; D[tri(1, 1)] += o.a1; D[tri(2, 2)] += o.a1;
  D[tri(3, 1)] += -o.a1 * r.z; D[tri(3, 2)] += o.a1 * r.y;
  D[tri(4, 0)] += o.a1 * r.z;  D[tri(4, 2)] += -o.a1 * r.x;
  D[tri(5, 0)] += -o.a1 * r.y; D[tri(5, 1)] += o.a1 * r.x;
  T rr = dot(r, r);
  D[tri(3, 3)] += o.a1 * (rr - r.x * r.x) + o.b1;
  D[tri(4, 3)] += -o.a1 * r.y * r.x;
  D[tri(4, 4)] += o.a1 * (rr - r.y * r.y) + o.b1;
  D[tri(5, 3)] += -o.a1 * r.z * r.x;
  D[tri(5, 4)] += -o.a1 * r.z * r.y;
  D[tri(5, 5)] += o.a1 * (rr - r.z * r.z) + o.b1;
  // rank-1 terms.  a2 (n, 0), b2 (0, n) always; the two cone terms (g, w) only where some lane of the wave sits on the
  // cone surface (cg = cw = 0 in the quadratic zone)
  {
    T u[6];
    body_vec(n, mk(T(0), T(0), T(0)), r, u);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) D[tri(i, j)] += o.a2 * u[i] * u[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) D[tri(3 + i, 3 + j)] += o.b2 * comp(n, i) * comp(n, j);
  }
  if (__any(o.cg != T(0))) {
    const V3<T> cls[2] = {o.gl, o.wl};
    const V3<T> cas[2] = {o.ga, o.wa};
    const T cf[2] = {o.cg, o.cw};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      T u[6];
      body_vec(cls[k], cas[k], r, u);
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) D[tri(i, j)] += cf[k] * u[i] * u[j];
    }
  }
}
// O (full 6x6, rows = body B dofs, cols = body A dofs) += J_B^T C J_A  (J_A carries the minus sign)
template <typename T> TGR_DEV void cross_block_add(T* O, const Cone<T>& o, V3<T> n, V3<T> rB, V3<T> rA) {
  T a = -o.a1, b = -o.b1;
  O[0 * 6 + 0] += a; O[1 * 6 + 1] += a; O[2 * 6 + 2] += a;
  // ang(B)-lin(A): [rB]x
  O[3 * 6 + 1] += -a * rB.z; O[3 * 6 + 2] += a * rB.y;
  O[4 * 6 + 0] += a * rB.z;  O[4 * 6 + 2] += -a * rB.x;
  O[5 * 6 + 0] += -a * rB.y; O[5 * 6 + 1] += a * rB.x;
  // lin(B)-ang(A): -[rA]x
  O[0 * 6 + 4] += a * rA.z;  O[0 * 6 + 5] += -a * rA.y;
  O[1 * 6 + 3] += -a * rA.z; O[1 * 6 + 5] += a * rA.x;
  O[2 * 6 + 3] += a * rA.y;  O[2 * 6 + 4] += -a * rA.x;
  // ang-ang: (rA.rB) I - rA rB^T
  T d = dot(rA, rB);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) O[(3 + i) * 6 + 3 + j] += a * ((i == j ? d : T(0)) - comp(rA, i) * comp(rB, j)) + (i == j ? b : T(0));
  {
    T uB[6], uA[6];
    const V3<T> z3 = mk(T(0), T(0), T(0));
    body_vec(n, z3, rB, uB);
    body_vec(n, z3, rA, uA);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) O[i * 6 + j] -= o.a2 * uB[i] * uA[j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) O[(3 + i) * 6 + 3 + j] -= o.b2 * comp(n, i) * comp(n, j);
  }
  if (__any(o.cg != T(0))) {
    const V3<T> cls[2] = {o.gl, o.wl};
    const V3<T> cas[2] = {o.ga, o.wa};
    const T cf[2] = {o.cg, o.cw};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      T uB[6], uA[6];
      body_vec(cls[k], cas[k], rB, uB);
      body_vec(cls[k], cas[k], rA, uA);
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) O[i * 6 + j] -= cf[k] * uB[i] * uA[j];
    }
  }
}

// in-place Cholesky of a packed lower 6x6; diagonal stores 1/L_jj
// ref[j]: the diagonal of the Hessian block before any Schur update.  Pivots are floored relative to it: a Schur
// complement that loses positivity to rounding must cost accuracy of one Newton direction (the line search copes),
// never produce inf.
template <typename T> TGR_DEV void diag6(const T* D, T* ref) {
#pragma unroll
  for (int j = 0; j < 6; j++) ref[j] = D[tri(j, j)];
}
template <typename T> TGR_DEV void chol6(T* D, const T* ref) {
  const T rel = sizeof(T) == 4 ? T(1e-6) : T(1e-13);
#pragma unroll
  for (int j = 0; j < 6; j++) {
    T d0 = ref[j];
    T s = D[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= D[tri(j, k)] * D[tri(j, k)];
    s = max_(s, max_(rel * abs_(d0), T(1e-30)));
    T il = rsqrt_(s);
    D[tri(j, j)] = il;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      T t = D[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= D[tri(i, k)] * D[tri(j, k)];
      D[tri(i, j)] = t * il;
    }
  }
}
template <typename T> TGR_DEV void fsolve6(const T* L, T* x) {  // x <- L^-1 x
#pragma unroll
  for (int i = 0; i < 6; i++) {
    T s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
    x[i] = s * L[tri(i, i)];
  }
}
template <typename T> TGR_DEV void bsolve6(const T* L, T* x) {  // x <- L^-T x
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    T s = x[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[tri(k, i)] * x[k];
    x[i] = s * L[tri(i, i)];
  }
}
template <typename T, int NN> TGR_DEV void chol_factor(T* H) {
#pragma unroll
  for (int j = 0; j < NN; j++) {
    T s = H[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= H[tri(j, k)] * H[tri(j, k)];
    s = max_(s, T(1e-30));
    T il = rsqrt_(s);
    H[tri(j, j)] = il;
#pragma unroll
    for (int i = j + 1; i < NN; i++) {
      T t = H[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= H[tri(i, k)] * H[tri(j, k)];
      H[tri(i, j)] = t * il;
    }
  }
}
template <typename T, int NN> TGR_DEV void chol_solve(const T* L, T* x) {
#pragma unroll
  for (int i = 0; i < NN; i++) {
    T s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
    x[i] = s * L[tri(i, i)];
  }
#pragma unroll
  for (int i = NN - 1; i >= 0; i--) {
    T s = x[i];
#pragma unroll
    for (int k = i + 1; k < NN; k++) s -= L[tri(k, i)] * x[k];
    x[i] = s * L[tri(i, i)];
  }
}

// ------------------------------------------------------------------ contact geometry between two bars
template <typename T> struct Cand { T dist; V3<T> n, p; bool ok; };

template <typename T> TGR_DEV void keep_deeper(Cand<T>& best, int& count, T dist, V3<T> n, V3<T> p, bool valid) {
  bool hit = valid && dist < T(0);
  count += hit ? 1 : 0;
  bool take = hit && (!best.ok || dist < best.dist);
  best.dist = take ? dist : best.dist;
  best.n = sel(take, n, best.n);
  best.p = sel(take, p, best.p);
  best.ok = best.ok || hit;
}

// sphere (centre c) against the coaxial cylinders of the other bar (centre xo, axis zo).
// flip=false: sphere belongs to body A (normal = sphere->cyl), flip=true: sphere belongs to body B.
template <typename T>
TGR_DEV void sphere_vs_bar(const Params<T>& P, V3<T> c, V3<T> xo, V3<T> zo, bool flip, bool enable, Cand<T>& best, int& count) {
  V3<T> wv = c - xo;
  T t = dot(wv, zo);
  V3<T> rad = wv - t * zo;
  T rho = sqrt_(dot(rad, rad));
  T irho = rcp_(max_(rho, T(1e-30)));
  // three cylinders: r (|t|<=half, R=rcyl), b+ (t in [lo,hi]), b- (t in [-hi,-lo])
#pragma unroll 1
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
    n = sel(inside, nI, n);
    V3<T> pos = c + (P.cap_r + T(0.5) * dist) * n;
    bool valid = enable && (inside ? (rho > T(1e-12)) : (dd > T(1e-12)));
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
  bool near;
  {
    T sc = clamp_(skew ? s : T(0), -P.bar_half, P.bar_half);
    T tc = clamp_(dB + b * sc, -P.bar_half, P.bar_half);
    sc = clamp_(b * tc - dA, -P.bar_half, P.bar_half);
    V3<T> dv = (xB + tc * zB) - (xA + sc * zA);
    T reach = T(2) * P.cap_r + T(1e-6);
    near = dot(dv, dv) <= reach * reach;
  }
  if (!__any(near)) return;
  // r-cylinder vs r-cylinder: the frequent one, own slot
  {
    bool interior = near && okdir && abs_(s) < P.bar_half && abs_(t) < P.bar_half;
    T dist = dd - T(2) * P.rcyl_r;
    if (interior && dist < T(0)) {
      c0.active = true;
      c0.n = n;
      V3<T> pos = pa + (P.rcyl_r + T(0.5) * dist) * n;
      c0.rA = pos - xA; c0.rB = pos - xB;
      contact_params(P, dist, 2, c0);
    }
  }
  // everything else: keep the deepest, count the rest.  Only possible when a cap sphere is within reach of the other
  // bar (end region), or the r-cylinders overlap by more than the b-cylinder clearance.
  bool deep = near && okdir && (dd - T(2) * P.rcyl_r < -(P.rcyl_r - P.bcyl_r));
  // is any of the four cap centres within two cap radii of the other bar's axis segment?
  bool ends = false;
  {
    T reach2 = (T(2) * P.cap_r + T(1e-6)) * (T(2) * P.cap_r + T(1e-6));
#pragma unroll
    for (int e = 0; e < 2; e++) {
      T sg = e == 0 ? P.bar_half : -P.bar_half;
      V3<T> ea = (xA + sg * zA) - xB;   // A's end relative to B
      T ta = clamp_(dot(ea, zB), -P.bar_half, P.bar_half);
      V3<T> da = ea - ta * zB;
      V3<T> eb = (xB + sg * zB) - xA;   // B's end relative to A
      T tb = clamp_(dot(eb, zA), -P.bar_half, P.bar_half);
      V3<T> db = eb - tb * zA;
      ends = ends || dot(da, da) <= reach2 || dot(db, db) <= reach2;
    }
    ends = ends && near;
  }
  if (!__any(deep || ends)) return;
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
      bool interior = near && okdir && abs_(s - ci) < hi_ && abs_(t - cj) < hj_;
      T dist = dd - ri - rj;
      V3<T> pos = pa + (ri + T(0.5) * dist) * n;
      keep_deeper(best, count, dist, n, pos, interior);
    }
  // spheres
#pragma unroll 1
  for (int ea = 0; ea < 2; ea++) {
    V3<T> ca = xA + (ea == 0 ? P.bar_half : -P.bar_half) * zA;
    sphere_vs_bar(P, ca, xB, zB, false, near, best, count);
#pragma unroll 1
    for (int eb = 0; eb < 2; eb++) {
      V3<T> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
      V3<T> dv = cb - ca;
      T cd = sqrt_(dot(dv, dv));
      T dist = cd - T(2) * P.cap_r;
      V3<T> nn = rcp_(max_(cd, T(1e-30))) * dv;
      V3<T> pos = ca + (P.cap_r + T(0.5) * dist) * nn;
      keep_deeper(best, count, dist, nn, pos, near && cd > T(1e-12));
    }
  }
#pragma unroll 1
  for (int eb = 0; eb < 2; eb++) {
    V3<T> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
    sphere_vs_bar(P, cb, xA, zA, true, near, best, count);
  }
  if (best.ok) {
    c1.active = true;
    c1.n = best.n;
    c1.rA = best.p - xA; c1.rB = best.p - xB;
    contact_params(P, best.dist, 2, c1);
  }
  dropped += count > 1 ? count - 1 : 0;
}

// bodies of pair p: (0,1), (0,2), (1,2)
TGR_DEV int pair_A(int p) { return p == 2 ? 1 : 0; }
TGR_DEV int pair_B(int p) { return p == 0 ? 1 : 2; }
TGR_DEV int pair_O(int p) { return p == 0 ? H_O10 : (p == 1 ? H_O20 : H_O21); }

// ------------------------------------------------------------------ Newton pieces (LDS-centric, 4 waves per block)
// The 64 envs of a block are worked on by FOUR wavefronts; lane i of every wave is env i, and the LDS columns are
// shared.  All four waves hold the lane's rigid-body state in registers (each integrates it, identically) and split the
// rest:   wave 0,1,2 = bar 0,1,2: base phase of the bar (kinematic columns, attached tendons, floor caps, one bar pair),
//                      its 6x6 diagonal Hessian block, its gradient rows, its share of the line-search slots
//         wave 3     = the off-diagonal (bar-pair) blocks and the second pair slots of the line search
//         wave 0     = additionally the dense factorisation / solve of the coupled 18x18 system
// Every decision that steers control flow is computed from shared LDS data, so it is identical in the four waves and
// the __syncthreads() below are reached uniformly.  Every LDS column has ONE writer: two template instantiations of the
// same formula may round differently, so "both write the same value" would be a race.

// which pairs touch bar b, and as which body:  pair p = (A,B): (0,1), (0,2), (1,2)
TGR_DEV int bar_pair(int b, int j) { return b == 0 ? j : (b == 1 ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2)); }
TGR_DEV bool bar_is_B(int b, int j) { return b == 0 ? false : (b == 1 ? (j == 0) : true); }

template <typename T, int LW>
TGR_DEV unsigned pair_coupling(const T* lds) {
  unsigned cpl = 0u;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    bool a = (lds[(L_CP + 12 * (2 * p) + 11) * LW] + lds[(L_CP + 12 * (2 * p + 1) + 11) * LW]) != T(0);
    if (__any(a)) cpl |= 1u << p;
  }
  return cpl;
}

// evaluate the contacts at the current A (in LDS): Hessian blocks, constraint wrench FC, gradient G.
// returns |G|^2 (all waves) ; cfrc_max = max |FC|
template <typename T, int LW, int MODEL>
TGR_DEV T newton_assemble(const Params<T>& P, T* lds, int wid, unsigned cpl, T& cfrc_max) {
  if (wid < 3) {
    const int b = wid;
    V3<T> ez = ld3<T, LW>(lds, L_EZ + 3 * b);
    T D[21];
#pragma unroll
    for (int i = 0; i < 21; i++) D[i] = T(0);
    T dI = P.Izz - P.Ixx;
    D[tri(0, 0)] = P.mass; D[tri(1, 1)] = P.mass; D[tri(2, 2)] = P.mass;
    D[tri(3, 3)] = P.Ixx + dI * ez.x * ez.x;
    D[tri(4, 3)] = dI * ez.y * ez.x; D[tri(4, 4)] = P.Ixx + dI * ez.y * ez.y;
    D[tri(5, 3)] = dI * ez.z * ez.x; D[tri(5, 4)] = dI * ez.z * ez.y; D[tri(5, 5)] = P.Ixx + dI * ez.z * ez.z;
    V3<T> al = ld3<T, LW>(lds, L_A + 6 * b), aa = ld3<T, LW>(lds, L_A + 6 * b + 3);
    V3<T> yl = al + ld3<T, LW>(lds, L_BV + 6 * b);
    V3<T> ya = aa + ld3<T, LW>(lds, L_BV + 6 * b + 3);
    V3<T> fl = mk(T(0), T(0), T(0)), fa = fl;
#pragma unroll 1
    for (int e = 0; e < n_floor<MODEL>(); e++) {
      Contact<T> c = load_floor<T, LW, MODEL>(lds, floor_index(b, e));
      if (__any(c.active)) {
        V3<T> jl = yl + cross(ya, c.rB);
        if (MODEL) jl = jl + c.c0 * c.n; else jl.z += c.c0;
        Cone<T> o = cone_eval(P, c, jl, ya);
        fl = fl + o.fl;
        fa = fa + cross(c.rB, o.fl) + o.fa;
        self_block_add(D, o, c.n, c.rB);
      }
    }
    // pair slots that touch this bar: this wave adds its own diagonal contribution and wrench
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
      const int p = bar_pair(b, j);
      if (!((cpl >> p) & 1u)) continue;
      const bool isB = bar_is_B(b, j);
      const int o_ = isB ? pair_A(p) : pair_B(p);  // the other bar
      V3<T> ylo = ld3<T, LW>(lds, L_A + 6 * o_) + ld3<T, LW>(lds, L_BV + 6 * o_);
      V3<T> yao = ld3<T, LW>(lds, L_A + 6 * o_ + 3) + ld3<T, LW>(lds, L_BV + 6 * o_ + 3);
#pragma unroll 1
      for (int s = 0; s < 2; s++) {
        Contact<T> c = load_pair<T, LW>(lds, 2 * p + s);
        if (!__any(c.active)) continue;
        V3<T> r_me = isB ? c.rB : c.rA, r_ot = isB ? c.rA : c.rB;
        V3<T> rel_l = (yl + cross(ya, r_me)) - (ylo + cross(yao, r_ot));
        V3<T> rel_a = ya - yao;
        T sg = isB ? T(1) : T(-1);
        V3<T> jl = sg * rel_l + c.c0 * c.n;
        V3<T> ja = sg * rel_a;
        Cone<T> o = cone_eval(P, c, jl, ja);
        fl = fl + sg * o.fl;
        fa = fa + sg * (cross(r_me, o.fl) + o.fa);
        self_block_add(D, o, c.n, r_me);
      }
    }
    st21<T, LW>(lds, H_D(b), D);
    st3<T, LW>(lds, L_FC + 6 * b, fl);
    st3<T, LW>(lds, L_FC + 6 * b + 3, fa);
    V3<T> dl = al - ld3<T, LW>(lds, L_A0 + 6 * b);
    V3<T> da = aa - ld3<T, LW>(lds, L_A0 + 6 * b + 3);
    V3<T> gl = P.mass * dl - fl;
    V3<T> ga = inertia_mul(P, ez, da) - fa;
    st3<T, LW>(lds, L_G + 6 * b, gl);
    st3<T, LW>(lds, L_G + 6 * b + 3, ga);
    lds[(L_RED + b) * LW] = dot(gl, gl) + dot(ga, ga);
    lds[(L_RED + 3 + b) * LW] = max_(max_(max_(abs_(fl.x), abs_(fl.y)), abs_(fl.z)), max_(max_(abs_(fa.x), abs_(fa.y)), abs_(fa.z)));
  } else {
    // wave 3: off-diagonal blocks
    T barsum = T(0);
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
      T O[36];
#pragma unroll
      for (int i = 0; i < 36; i++) O[i] = T(0);
      if (!((cpl >> p) & 1u)) {
        if (cpl != 0u) st36<T, LW>(lds, pair_O(p), O);  // the dense factorisation reads every block
        continue;
      }
      const int A = pair_A(p), B = pair_B(p);
      V3<T> ylA = ld3<T, LW>(lds, L_A + 6 * A) + ld3<T, LW>(lds, L_BV + 6 * A);
      V3<T> yaA = ld3<T, LW>(lds, L_A + 6 * A + 3) + ld3<T, LW>(lds, L_BV + 6 * A + 3);
      V3<T> ylB = ld3<T, LW>(lds, L_A + 6 * B) + ld3<T, LW>(lds, L_BV + 6 * B);
      V3<T> yaB = ld3<T, LW>(lds, L_A + 6 * B + 3) + ld3<T, LW>(lds, L_BV + 6 * B + 3);
#pragma unroll 1
      for (int s = 0; s < 2; s++) {
        Contact<T> c = load_pair<T, LW>(lds, 2 * p + s);
        if (!__any(c.active)) continue;
        V3<T> jl = (ylB + cross(yaB, c.rB)) - (ylA + cross(yaA, c.rA)) + c.c0 * c.n;
        V3<T> ja = yaB - yaA;
        Cone<T> o = cone_eval(P, c, jl, ja);
        cross_block_add(O, o, c.n, c.rB, c.rA);
        barsum += sqrt_(dot(o.fl, o.fl));
      }
      st36<T, LW>(lds, pair_O(p), O);
    }
    lds[(L_RED + 31) * LW] = barsum;
  }
  __syncthreads();
  T gn = lds[L_RED * LW] + lds[(L_RED + 1) * LW] + lds[(L_RED + 2) * LW];
  cfrc_max = max_(max_(lds[(L_RED + 3) * LW], lds[(L_RED + 4) * LW]), lds[(L_RED + 5) * LW]);
  return gn;
}

// Newton direction  P = -H^-1 G.
// The Hessian is 18x18: three 6x6 diagonal blocks (packed lower, H_D) and three 6x6 off-diagonal blocks (row = dof of
// the higher bar, H_O*).  Under contact-rich rollouts every block of 64 envs has all three bar pairs coupled in >99.9 %
// of the sub-steps (measured), so the coupled case is one dense factorisation.  A Cholesky is a dependency chain: spread
// over the four waves as a blocked pipeline it spent more time in hand-offs through LDS and barriers than in arithmetic
// (12.4 k cycles per call).  Here ONE wave does it register-resident, as row pairs so that the updates issue as
// v_pk_fma_f32 (two fp32 FMAs per lane per instruction): ~1.3 k instructions, one barrier.
// Storage: column k of the lower triangle as row pairs p = k/2 .. 8 (the entry above the diagonal inside a pair is
// don't-care).  Pivots are floored relative to the original diagonal: a pivot that loses positivity to rounding must cost
// accuracy of one Newton direction (the line search copes), never produce inf.
template <typename T> struct Pk2 { typedef T type __attribute__((ext_vector_type(2))); };
constexpr __host__ __device__ int dcoff(int k) { int o = 0; for (int c = 0; c < k; c++) o += 9 - c / 2; return o; }
constexpr __host__ __device__ int dcidx(int k, int p) { return dcoff(k) + (p - k / 2); }
constexpr int DENSE_NC = dcoff(18);  // 90 pairs
// LDS slot of Hessian entry (i, k), i >= k
constexpr __host__ __device__ int hslot(int i, int k) {
  const int bi = i / 6, bk = k / 6, r = i % 6, c = k % 6;
  return bi == bk ? H_D(bi) + tri(r, c) : (bi == 1 ? H_O10 : (bk == 0 ? H_O20 : H_O21)) + r * 6 + c;
}
template <typename T> TGR_DEV void dense_chol18(typename Pk2<T>::type* C, T* dinv) {
  typedef typename Pk2<T>::type T2;
  const T rel = sizeof(T) == 4 ? T(1e-6) : T(1e-13);
#pragma unroll
  for (int j = 0; j < 18; j++) {
    const int pj = j / 2;
    T d = (j & 1) ? C[dcidx(j, pj)].y : C[dcidx(j, pj)].x;
    d = max_(d, max_(rel * abs_(dinv[j]), T(1e-30)));
    const T il = rsqrt_(d);
    dinv[j] = il;
    const T2 ilv = {il, il};
#pragma unroll
    for (int p = pj; p < 9; p++) C[dcidx(j, p)] = C[dcidx(j, p)] * ilv;
#pragma unroll
    for (int k = j + 1; k < 18; k++) {
      const int pk = k / 2;
      const T lkj = (k & 1) ? C[dcidx(j, pk)].y : C[dcidx(j, pk)].x;
      const T2 lv = {lkj, lkj};
#pragma unroll
      for (int p = pk; p < 9; p++) C[dcidx(k, p)] = C[dcidx(k, p)] - C[dcidx(j, p)] * lv;
    }
  }
}
// x (9 row pairs) <- L^-T L^-1 x ; L = scaled columns of C, 1/L_jj in dinv
template <typename T> TGR_DEV void dense_solve18(const typename Pk2<T>::type* C, const T* dinv, typename Pk2<T>::type* x) {
  typedef typename Pk2<T>::type T2;
#pragma unroll
  for (int j = 0; j < 18; j++) {  // forward, column oriented
    const int pj = j / 2;
    const T xj = ((j & 1) ? x[pj].y : x[pj].x) * dinv[j];
    if (j & 1) x[pj].y = xj; else x[pj].x = xj;
    if (!(j & 1)) x[pj].y -= C[dcidx(j, pj)].y * xj;  // partner row of the same pair
    const T2 xv = {xj, xj};
#pragma unroll
    for (int p = pj + 1; p < 9; p++) x[p] = x[p] - C[dcidx(j, p)] * xv;
  }
#pragma unroll
  for (int i = 17; i >= 0; i--) {  // backward, row oriented
    const int pi = i / 2;
    T2 acc = {T(0), T(0)};
#pragma unroll
    for (int p = pi + 1; p < 9; p++) acc = acc + C[dcidx(i, p)] * x[p];
    T s = ((i & 1) ? x[pi].y : x[pi].x) - (acc.x + acc.y);
    if (!(i & 1)) s -= C[dcidx(i, pi)].y * x[pi].y;
    s *= dinv[i];
    if (i & 1) x[pi].y = s; else x[pi].x = s;
  }
}
template <typename T, int LW> TGR_DEV void newton_direction(T* lds, int wid, unsigned cpl, bool conv, Lag<T>& lag) {
  typedef typename Pk2<T>::type T2;
  if (cpl == 0u) {
    // no lane of the block couples two bars: three independent 6x6 solves, one per wave
    if (wid < 3) {
      T La[21], yv[6], dref[6];
      ld21<T, LW>(lds, H_D(wid), La);
      diag6(La, dref);
      chol6(La, dref);
#pragma unroll
      for (int i = 0; i < 6; i++) yv[i] = lds[(L_G + 6 * wid + i) * LW];
      fsolve6(La, yv);
      bsolve6(La, yv);
#pragma unroll
      for (int i = 0; i < 6; i++) lds[(L_P + 6 * wid + i) * LW] = conv ? T(0) : -yv[i];
    }
    TGR_STAMP(8); __syncthreads(); TGR_STAMP(9);
    return;
  }
  if (wid == 0) {
    T2 C[DENSE_NC];
    T dinv[18];
    T2 x[9];
#pragma unroll
    for (int k = 0; k < 18; k++)
#pragma unroll
      for (int p = k / 2; p < 9; p++) {
        const int i0 = 2 * p, i1 = 2 * p + 1;
        const T a = i0 >= k ? lds[hslot(i0, k) * LW] : T(0);
        C[dcidx(k, p)] = T2{a, lds[hslot(i1, k) * LW]};
      }
#pragma unroll
    for (int j = 0; j < 18; j++) dinv[j] = lds[hslot(j, j) * LW];
#pragma unroll
    for (int p = 0; p < 9; p++) x[p] = T2{lds[(L_G + 2 * p) * LW], lds[(L_G + 2 * p + 1) * LW]};
    // a lone wave cannot hide LDS latency behind other waves: have all ~100 reads in flight before the first FMA
    __builtin_amdgcn_sched_barrier(0);
    dense_chol18<T>(C, dinv);
    dense_solve18<T>(C, dinv, x);
#pragma unroll
    for (int p = 0; p < 9; p++) {
      lds[(L_P + 2 * p) * LW] = conv ? T(0) : -x[p].x;
      lds[(L_P + 2 * p + 1) * LW] = conv ? T(0) : -x[p].y;
    }
  }
  TGR_STAMP(10); __syncthreads(); TGR_STAMP(11);
}


// ---- line search.  Along a + alpha p every contact row is jar0 + alpha q, so the cone cost of a slot depends on alpha only
// through ten scalars (norms and projections of jar0 and q).  Each wave reduces its slots to those scalars once per Newton
// iteration (waves 0-2: the two caps of their bar, wave 3: the pair slots) and a line-search evaluation is ~50 instructions
// per slot plus one cross-wave sum.
template <typename T> struct LSlot { T A0, A1, A2, B0, B1, B2, jn0, qn, jo0, qo, Dn; bool act; };

template <typename T> TGR_DEV LSlot<T> ls_make(const Contact<T>& c, V3<T> jl, V3<T> ja, V3<T> ql, V3<T> qa) {
  LSlot<T> s;
  s.A0 = dot(jl, jl); s.A1 = dot(jl, ql); s.A2 = dot(ql, ql);
  s.B0 = dot(ja, ja); s.B1 = dot(ja, qa); s.B2 = dot(qa, qa);
  s.jn0 = dot(c.n, jl); s.qn = dot(c.n, ql); s.jo0 = dot(c.n, ja); s.qo = dot(c.n, qa);
  s.Dn = c.Dn; s.act = c.active;
  return s;
}
// d(cost)/d(alpha) and d2(cost)/d(alpha)2 of one slot
template <typename T> TGR_DEV void ls_slot(const Params<T>& P, const LSlot<T>& s, T alpha, T& d1, T& d2) {
  T mu = P.f0 * rsqrt_(P.impratio);
  T Dtf = s.Dn * P.impratio * rcp_(P.f0 * P.f0);  // D_j / f_j^2 of the friction rows (same for all five)
  T jn = s.jn0 + alpha * s.qn, jo = s.jo0 + alpha * s.qo;
  T jl2 = s.A0 + alpha * (T(2) * s.A1 + alpha * s.A2), ja2 = s.B0 + alpha * (T(2) * s.B1 + alpha * s.B2);
  T jlq = s.A1 + alpha * s.A2, jaq = s.B1 + alpha * s.B2;
  T tt = P.f1sq * (jl2 - jn * jn) + P.f3sq * jo * jo + P.f4sq * (ja2 - jo * jo);
  T Tn = sqrt_(max_(tt, T(0)));
  T wq = P.f1sq * (jlq - jn * s.qn) + P.f3sq * jo * s.qo + P.f4sq * (jaq - jo * s.qo);
  T qtt = P.f1sq * (s.A2 - s.qn * s.qn) + P.f3sq * s.qo * s.qo + P.f4sq * (s.B2 - s.qo * s.qo);
  T N = mu * jn;
  bool top = (N >= mu * Tn) || !s.act;
  bool bottom = !top && (mu * N + Tn <= T(0));
  // bottom: quadratic
  T b1 = s.Dn * jn * s.qn + Dtf * wq;
  T b2 = s.Dn * s.qn * s.qn + Dtf * qtt;
  // middle: 1/2 Dm (N - mu T)^2
  T rT = rcp_(max_(Tn, T(1e-30)));
  T Dm = s.Dn * rcp_(mu * mu * (T(1) + mu * mu));
  T NT = N - mu * Tn;
  T gq = mu * (s.qn - wq * rT);
  T kap = -Dm * mu * NT * rT;
  T m1 = Dm * NT * gq;
  T m2 = Dm * gq * gq + kap * (qtt - wq * wq * rT * rT);
  d1 = top ? T(0) : (bottom ? b1 : m1);
  d2 = top ? T(0) : (bottom ? b2 : m2);
}

// slot ownership in the line search: wave w < 3 takes the two caps of bar w and the first slot of pair w,
// wave 3 takes the three second pair slots (rarely active)
template <typename T, int LW>
TGR_DEV LSlot<T> ls_pair_slot(const Params<T>& P, const T* lds, int p, int s) {
  const int A = pair_A(p), B = pair_B(p);
  Contact<T> c = load_pair<T, LW>(lds, 2 * p + s);
  LSlot<T> r;
  r.act = false; r.A0 = r.A1 = r.A2 = r.B0 = r.B1 = r.B2 = T(0); r.jn0 = r.qn = r.jo0 = r.qo = T(0); r.Dn = T(0);
  if (!__any(c.active)) return r;
  V3<T> plA = ld3<T, LW>(lds, L_P + 6 * A), paA = ld3<T, LW>(lds, L_P + 6 * A + 3);
  V3<T> plB = ld3<T, LW>(lds, L_P + 6 * B), paB = ld3<T, LW>(lds, L_P + 6 * B + 3);
  V3<T> ylA = ld3<T, LW>(lds, L_A + 6 * A) + ld3<T, LW>(lds, L_BV + 6 * A);
  V3<T> yaA = ld3<T, LW>(lds, L_A + 6 * A + 3) + ld3<T, LW>(lds, L_BV + 6 * A + 3);
  V3<T> ylB = ld3<T, LW>(lds, L_A + 6 * B) + ld3<T, LW>(lds, L_BV + 6 * B);
  V3<T> yaB = ld3<T, LW>(lds, L_A + 6 * B + 3) + ld3<T, LW>(lds, L_BV + 6 * B + 3);
  V3<T> jl = (ylB + cross(yaB, c.rB)) - (ylA + cross(yaA, c.rA)) + c.c0 * c.n;
  V3<T> ja = yaB - yaA;
  V3<T> ql = (plB + cross(paB, c.rB)) - (plA + cross(paA, c.rA));
  V3<T> qa = paB - paA;
  return ls_make(c, jl, ja, ql, qa);
}

template <typename T, int LW, int MODEL>
TGR_DEV void ls_setup(const Params<T>& P, const T* lds, int wid, unsigned cpl, LSlot<T>* ls) {
#pragma unroll
  for (int k = 0; k < num_ls<MODEL>(); k++) { ls[k].act = false; ls[k].A0 = ls[k].A1 = ls[k].A2 = ls[k].B0 = ls[k].B1 = ls[k].B2 = T(0); ls[k].jn0 = ls[k].qn = ls[k].jo0 = ls[k].qo = T(0); ls[k].Dn = T(0); }
  if (wid < 3) {
    const int b = wid;
    V3<T> pl = ld3<T, LW>(lds, L_P + 6 * b), pa = ld3<T, LW>(lds, L_P + 6 * b + 3);
    V3<T> yl = ld3<T, LW>(lds, L_A + 6 * b) + ld3<T, LW>(lds, L_BV + 6 * b);
    V3<T> ya = ld3<T, LW>(lds, L_A + 6 * b + 3) + ld3<T, LW>(lds, L_BV + 6 * b + 3);
#pragma unroll
    for (int e = 0; e < n_floor<MODEL>(); e++) {
      Contact<T> c = load_floor<T, LW, MODEL>(lds, floor_index(b, e));
      V3<T> jl = yl + cross(ya, c.rB);
      if (MODEL) jl = jl + c.c0 * c.n; else jl.z += c.c0;
      V3<T> ql = pl + cross(pa, c.rB);
      ls[e < 2 ? e : e + 1] = ls_make(c, jl, ya, ql, pa);  // slot 2 is the wave's pair slot
    }
    if ((cpl >> b) & 1u) ls[2] = ls_pair_slot<T, LW>(P, lds, b, 0);
  } else {
#pragma unroll
    for (int p = 0; p < 3; p++)
      if ((cpl >> p) & 1u) ls[p] = ls_pair_slot<T, LW>(P, lds, p, 1);
  }
}

// one barrier per evaluation: partial sums go to alternating halves of the reduction area
template <typename T, int LW, int NLS>
TGR_DEV void newton_dphi(const Params<T>& P, T* lds, int wid, const LSlot<T>* ls, int parity, T alpha, T g1, T g2, T& d1, T& d2) {
  T p1 = T(0), p2 = T(0);
#pragma unroll
  for (int k = 0; k < NLS; k++) {
    if (__any(ls[k].act)) {
      T a, b;
      ls_slot(P, ls[k], alpha, a, b);
      p1 += a; p2 += b;
    }
  }
  const int base = L_RED + 6 + 8 * parity;
  lds[(base + 2 * wid) * LW] = p1;
  lds[(base + 1 + 2 * wid) * LW] = p2;
  __syncthreads();
  d1 = g1 + alpha * g2; d2 = g2;
#pragma unroll
  for (int w = 0; w < 4; w++) { d1 += lds[(base + 2 * w) * LW]; d2 += lds[(base + 1 + 2 * w) * LW]; }
}

// qacc_warmstart lives in the L_A columns between sub-steps (not in registers): park it there when a kernel starts ...
template <typename T, int LW> TGR_DEV void warm_to_lds(const Sim<T>& S, T* lds, int wid) {
#pragma unroll
  for (int b = 0; b < 3; b++)
    if (b == wid) {
      st3<T, LW>(lds, L_A + 6 * b, S.al[b]); st3<T, LW>(lds, L_A + 6 * b + 3, S.aa[b]);
      // the force warm start (warm_start == 2) reads last sub-step's constraint wrench: none before the first one
      st3<T, LW>(lds, L_FC + 6 * b, mk(T(0), T(0), T(0))); st3<T, LW>(lds, L_FC + 6 * b + 3, mk(T(0), T(0), T(0)));
    }
}
// ... and fetch it back for store_sim when the last sub-step is done
template <typename T, int LW> TGR_DEV void warm_from_lds(Sim<T>& S, const T* lds) {
#pragma unroll
  for (int b = 0; b < 3; b++) { S.al[b] = ld3<T, LW>(lds, L_A + 6 * b); S.aa[b] = ld3<T, LW>(lds, L_A + 6 * b + 3); }
}

// base phase of bar B (see substep): kinematic columns, attached tendons, a0, floor caps, bar pair B
// closest point of triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5)
template <typename T> TGR_DEV V3<T> closest_on_triangle(V3<T> p, V3<T> a, V3<T> b, V3<T> c) {
  V3<T> ab = b - a, ac = c - a, ap = p - a;
  T d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= T(0) && d2 <= T(0)) return a;
  V3<T> bp = p - b;
  T d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= T(0) && d4 <= d3) return b;
  T vc = d1 * d4 - d3 * d2;
  if (vc <= T(0) && d1 >= T(0) && d3 <= T(0)) return a + (d1 / (d1 - d3)) * ab;
  V3<T> cp = p - c;
  T d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= T(0) && d5 <= d6) return c;
  T vb = d5 * d2 - d1 * d6;
  if (vb <= T(0) && d2 >= T(0) && d6 <= T(0)) return a + (d2 / (d2 - d6)) * ac;
  T va = d3 * d6 - d5 * d4;
  if (va <= T(0) && (d4 - d3) >= T(0) && (d5 - d6) >= T(0)) return b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
  T den = T(1) / (va + vb + vc);
  return a + (vb * den) * ab + (vc * den) * ac;
}
// cap sphere (centre cw, world) against the heightfield floor: deepest triangle of the (at most 2x2) cells under the cap.
// Each grid cell is two triangles of the strip (c,r) (c,r+1) (c+1,r) (c+1,r+1).  Returns dist (>= 0: no contact), normal, and
// the surface point.
template <typename T> TGR_DEV void hfield_cap(const Params<T>& P, V3<T> cw, T* dist_out, V3<T>* n_out) {
  const T rad = P.cap_r;
  V3<T> p = mk(cw.x - P.hf_x0, cw.y - P.hf_y0, cw.z - P.hf_z0);
  dist_out[0] = dist_out[1] = T(1);
  n_out[0] = n_out[1] = mk(T(0), T(0), T(1));
  if (p.x < -P.hf_rx - rad || p.x > P.hf_rx + rad || p.y < -P.hf_ry - rad || p.y > P.hf_ry + rad || p.z > P.hf_ez + rad) return;
  const T idx = rcp_(P.hf_dx), idy = rcp_(P.hf_dy);
  int c0 = (int)floor(double((p.x - rad + P.hf_rx) * idx)), c1 = (int)floor(double((p.x + rad + P.hf_rx) * idx));
  int r0 = (int)floor(double((p.y - rad + P.hf_ry) * idy)), r1 = (int)floor(double((p.y + rad + P.hf_ry) * idy));
  c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
  c1 = c1 > P.hf_ncol - 2 ? P.hf_ncol - 2 : c1; r1 = r1 > P.hf_nrow - 2 ? P.hf_nrow - 2 : r1;
  // a cap (5 cm) spans at most 2x2 cells of the 1 m grid: fetch the 3x3 vertices in one batch (a lone wave cannot hide the
  // latency of loads issued one cell at a time), then walk the cells with predicates
  T hz[3][3];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int cc = min(c0 + i, P.hf_ncol - 1), rr = min(r0 + j, P.hf_nrow - 1);
      hz[j][i] = T(P.hf[rr * P.hf_ncol + cc]) * P.hf_ez;
    }
  // the two deepest triangles touched (mujoco: one contact per prism penetrated; two is what a cap in a fold needs)
  T bd0 = T(1), bd1 = T(1);
  V3<T> bq0 = p, bq1 = p;
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (c0 + i > c1 || r0 + j > r1) continue;
      const T x0 = T(c0 + i) * P.hf_dx - P.hf_rx, y0 = T(r0 + j) * P.hf_dy - P.hf_ry;
      const V3<T> v[4] = {mk(x0, y0, hz[j][i]), mk(x0, y0 + P.hf_dy, hz[j + 1][i]), mk(x0 + P.hf_dx, y0, hz[j][i + 1]),
                          mk(x0 + P.hf_dx, y0 + P.hf_dy, hz[j + 1][i + 1])};  // (c,r) (c,r+1) (c+1,r) (c+1,r+1)
#pragma unroll
      for (int tri = 0; tri < 2; tri++) {
        V3<T> q = closest_on_triangle(p, v[tri], v[tri + 1], v[tri + 2]);
        V3<T> df = p - q;
        T dist = sqrt_(dot(df, df)) - rad;
        if (!(dist < T(0))) continue;
        if (dist < bd0) { bd1 = bd0; bq1 = bq0; bd0 = dist; bq0 = q; }
        else if (dist < bd1) { bd1 = dist; bq1 = q; }
      }
    }
  {
    V3<T> df = p - bq0;
    T nn = sqrt_(dot(df, df));
    if (bd0 < T(0) && nn > T(1e-12)) { dist_out[0] = bd0; n_out[0] = rcp_(nn) * df; }
  }
  {
    V3<T> df = p - bq1;
    T nn = sqrt_(dot(df, df));
    if (bd1 < T(0) && nn > T(1e-12)) { dist_out[1] = bd1; n_out[1] = rcp_(nn) * df; }
  }
}
// closest points of two segments c + s a (|s| <= ha) and c2 + t a2 (|t| <= hb), unit axes (Ericson 5.1.9)
template <typename T> TGR_DEV void seg_seg_closest(V3<T> ca, V3<T> aa, T ha, V3<T> cb, V3<T> ab, T hb, T& s_out, T& t_out) {
  V3<T> w = ca - cb;
  T b = dot(aa, ab), dA = dot(aa, w), dB = dot(ab, w);
  T den = T(1) - b * b;
  T s, t;
  bool interior = false;
  if (den > T(1e-12)) {
    s = (b * dB - dA) / den;
    t = (dB - b * dA) / den;
    interior = abs_(s) < ha && abs_(t) < hb;
  } else {
    s = T(0); t = dB;
  }
  if (!interior) {
    s = clamp_(s, -ha, ha);
    t = clamp_(dB + b * s, -hb, hb);
    s = clamp_(b * t - dA, -ha, ha);
  }
  s_out = s; t_out = t;
}
// the r-cylinder of a bar (centre xc, axis az, world) against the heightfield: deepest grid edge under the bar, the bar taken
// as a capsule of the cylinder's radius (a rod over piecewise-planar terrain touches it at a convex ridge or with an end, and
// the ends belong to the cap spheres).  Returns dist (>= 0: none); n_out = normal, arm_out = contact point - bar centre.
template <typename T> TGR_DEV T hfield_rod(const Params<T>& P, V3<T> xc, V3<T> az, V3<T>& n_out, V3<T>& arm_out) {
  const T rad = P.rcyl_r, H = P.bar_half;
  n_out = mk(T(0), T(0), T(1)); arm_out = mk(T(0), T(0), T(0));
  V3<T> c = mk(xc.x - P.hf_x0, xc.y - P.hf_y0, xc.z - P.hf_z0);
  const T ex = abs_(az.x) * H + rad, ey = abs_(az.y) * H + rad;
  if (c.x + ex < -P.hf_rx || c.x - ex > P.hf_rx || c.y + ey < -P.hf_ry || c.y - ey > P.hf_ry) return T(1);
  if (c.z - abs_(az.z) * H - rad > P.hf_ez) return T(1);
  const T idx = rcp_(P.hf_dx), idy = rcp_(P.hf_dy);
  // cells under the bar's bounding box: at most 3x3 of the 1 m grid for a 1.38 m bar (the far sides of the last cells lie
  // outside the box, so the cells' own three edges -- bottom, left, diagonal -- are all that can touch)
  int c0 = (int)floor(double((c.x - ex + P.hf_rx) * idx)), c1 = (int)floor(double((c.x + ex + P.hf_rx) * idx));
  int r0 = (int)floor(double((c.y - ey + P.hf_ry) * idy)), r1 = (int)floor(double((c.y + ey + P.hf_ry) * idy));
  c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
  c1 = c1 > P.hf_ncol - 2 ? P.hf_ncol - 2 : c1; r1 = r1 > P.hf_nrow - 2 ? P.hf_nrow - 2 : r1;
  c1 = c1 > c0 + 2 ? c0 + 2 : c1; r1 = r1 > r0 + 2 ? r0 + 2 : r1;
  T hz[4][4];  // the 4x4 vertices of those cells, fetched in one batch
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int cc = min(c0 + i, P.hf_ncol - 1), rr = min(r0 + j, P.hf_nrow - 1);
      hz[j][i] = T(P.hf[rr * P.hf_ncol + cc]) * P.hf_ez;
    }
  T best = T(1);
  V3<T> bpa = c, bpb = c;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (c0 + i > c1 || r0 + j > r1) continue;
      const T x0 = T(c0 + i) * P.hf_dx - P.hf_rx, y0 = T(r0 + j) * P.hf_dy - P.hf_ry;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        // edges of cell (c, r): (c,r)-(c+1,r), (c,r)-(c,r+1), and the strip diagonal (c,r+1)-(c+1,r)
        const V3<T> va = k == 2 ? mk(x0, y0 + P.hf_dy, hz[j + 1][i]) : mk(x0, y0, hz[j][i]);
        const V3<T> vb = k == 1 ? mk(x0, y0 + P.hf_dy, hz[j + 1][i]) : mk(x0 + P.hf_dx, y0, hz[j][i + 1]);
        V3<T> ed = vb - va;
        T el = sqrt_(dot(ed, ed));
        ed = rcp_(el) * ed;
        V3<T> mid = T(0.5) * (va + vb);
        T s_, t_;
        seg_seg_closest(c, az, H, mid, ed, T(0.5) * el, s_, t_);
        V3<T> pa = c + s_ * az, pb = mid + t_ * ed;
        V3<T> df = pa - pb;
        T dist = sqrt_(dot(df, df)) - rad;
        if (dist < T(0) && dist < best) { best = dist; bpa = pa; bpb = pb; }
      }
    }
  if (best >= T(0)) return T(1);
  V3<T> df = bpa - bpb;
  T nn = sqrt_(dot(df, df));
  if (!(nn > T(1e-12))) return T(1);
  n_out = rcp_(nn) * df;
  arm_out = (bpa - c) - (rad + T(0.5) * best) * n_out;
  return best;
}

template <typename T, int LW, int B, bool last, int MODEL>
TGR_DEV void base_bar(const Params<T>& P, const Sim<T>& S, const Rot<T>* R, T* lds, Lag<T>& lag) {
  const T im = rcp_(P.mass);
  st3<T, LW>(lds, L_EZ + 3 * B, R[B].ez);
  st3<T, LW>(lds, L_X + 3 * B, S.x[B]);
  st3<T, LW>(lds, L_BV + 6 * B, P.B * S.v[B]);
  st3<T, LW>(lds, L_BV + 6 * B + 3, P.B * S.w[B]);
  V3<T> F = mk(T(0), T(0), -P.mass * P.grav);
  // gyroscopic: -w x (I w) = -(Izz-Ixx)(z.w)(w x z)
  V3<T> Tq = (-(P.Izz - P.Ixx) * dot(R[B].ez, S.w[B])) * cross(S.w[B], R[B].ez);
#pragma unroll
  for (int t = 0; t < 9; t++) {
    const int ba = Tab::TEN[t][0], bb = Tab::TEN[t][2];  // the bars of a tendon are the same in both models
    const int sa = MODEL ? Tab::TEN1[t][1] : Tab::TEN[t][1], sb = MODEL ? Tab::TEN1[t][3] : Tab::TEN[t][3];
    const bool mine = (ba == B) || (bb == B);
    // the observation and tr_env's control cost need all nine lengths: in the last sub-step also the three foreign ones
    if (!mine && !last) continue;
    V3<T> ra = rot_apply(R[ba], MODEL ? mk(T(Tab::S1[sa][0]) * P.site_off, T(Tab::S1[sa][1]) * P.site_off, T(Tab::S1[sa][2]) * P.site_z)
                                      : mk(T(Tab::S[sa][0]) * P.site_off, T(Tab::S[sa][1]) * P.site_off, T(Tab::S[sa][2]) * P.site_z));
    V3<T> rb = rot_apply(R[bb], MODEL ? mk(T(Tab::S1[sb][0]) * P.site_off, T(Tab::S1[sb][1]) * P.site_off, T(Tab::S1[sb][2]) * P.site_z)
                                      : mk(T(Tab::S[sb][0]) * P.site_off, T(Tab::S[sb][1]) * P.site_off, T(Tab::S[sb][2]) * P.site_z));
    V3<T> dif = (S.x[bb] + rb) - (S.x[ba] + ra);
    T L2 = dot(dif, dif);
    T iL = rsqrt_(max_(L2, T(1e-30)));
    T L = L2 * iL;
    if (last) lag.tenL[t] = L;
    if (!mine) continue;
    V3<T> u = iL * dif;
    V3<T> va = S.v[ba] + cross(S.w[ba], ra), vb = S.v[bb] + cross(S.w[bb], rb);
    T Ld = dot(u, vb - va);
    T f, damp;
    if (t < 6 && MODEL) {
      // dyntype filter, no bias, no force limit (uneven xml:39): the force sees the activation
      f = P.gain * S.act[t] - P.ten_k[t] * (L - P.ten_L0[t]) - P.ten_b[t] * Ld;
      damp = P.ten_b[t];
    } else if (t < 6) {
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
    V3<T> ma = cross(ra, u), mb = cross(rb, u);
    if (bb == B) { F = F + f * u; Tq = Tq + f * mb; }
    else { F = F - f * u; Tq = Tq - f * ma; }
    if (ba == B) {  // one writer per column: the two instantiations may round differently, a race would not be benign
      st3<T, LW>(lds, L_TEN + 10 * t, u);
      st3<T, LW>(lds, L_TEN + 10 * t + 3, ma);
      st3<T, LW>(lds, L_TEN + 10 * t + 6, mb);
      lds[(L_TEN + 10 * t + 9) * LW] = sqrt_(P.h * damp);
    }
  }
  // unconstrained acceleration; the Newton iteration starts from it.  The optimum is unique, so the start changes nothing
  // but the path: measured on MI355X it needs fewer line-search evaluations in the slowest lane of a block than mujoco's
  // "better of warm start and a0" (the 1.9 ms contact time constant makes last sub-step's solution a poor guess).
  V3<T> l0 = im * F, a0 = inv_inertia(P, R[B].ez, Tq);
  st3<T, LW>(lds, L_A0 + 6 * B, l0);
  st3<T, LW>(lds, L_A0 + 6 * B + 3, a0);
  if (!P.warm_start) {  // with mujoco's warm start the previous solution simply stays in L_A
    st3<T, LW>(lds, L_A + 6 * B, l0);
    st3<T, LW>(lds, L_A + 6 * B + 3, a0);
  } else if (P.warm_start == 2) {
    // force warm start: last sub-step's constraint wrench on top of THIS sub-step's unconstrained acceleration.  A contact's
    // reference acceleration moves by ~B h a every sub-step, so last sub-step's acceleration is a poor guess -- but its contact
    // forces are a good one (a cap at rest carries the same weight from one millisecond to the next).
    st3<T, LW>(lds, L_A + 6 * B, l0 + im * ld3<T, LW>(lds, L_FC + 6 * B));
    st3<T, LW>(lds, L_A + 6 * B + 3, a0 + inv_inertia(P, R[B].ez, ld3<T, LW>(lds, L_FC + 6 * B + 3)));
  }
  // floor contacts: caps 2B, 2B+1
#pragma unroll
  for (int e = 0; e < 2; e++) {
    T sg = e == 0 ? P.bar_half : -P.bar_half;
    V3<T> arm = sg * R[B].ez;
    const int s = L_CF + 6 * (2 * B + e);
    Contact<T> c;
    if (MODEL) {
      V3<T> n2[2];
      T d2[2];
      hfield_cap(P, S.x[B] + arm, d2, n2);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const bool hit = d2[j] < T(0);
        const T dist = hit ? d2[j] : T(0);
        contact_params(P, dist, 1, c);
        V3<T> rB = arm - (P.cap_r + T(0.5) * dist) * n2[j];
        const int sj = j == 0 ? s : L_F2 + 6 * (2 * B + e);
        st3<T, LW>(lds, sj, rB);
        st3<T, LW>(lds, (j == 0 ? L_FN : L_F2N) + 3 * (2 * B + e), n2[j]);
        lds[(sj + 3) * LW] = c.c0; lds[(sj + 4) * LW] = c.Dn; lds[(sj + 5) * LW] = hit ? T(1) : T(0);
      }
    } else {
      T dist = S.x[B].z + arm.z - P.cap_r;
      contact_params(P, dist, 1, c);
      lds[s * LW] = arm.x; lds[(s + 1) * LW] = arm.y; lds[(s + 2) * LW] = arm.z - (P.cap_r + T(0.5) * dist);
      lds[(s + 3) * LW] = c.c0; lds[(s + 4) * LW] = c.Dn; lds[(s + 5) * LW] = dist < T(0) ? T(1) : T(0);
    }
  }
  if (MODEL) {  // the bar's cylinder against the terrain
    V3<T> n, arm;
    T dist = hfield_rod(P, S.x[B], R[B].ez, n, arm);
    const bool hit = dist < T(0);
    if (!hit) dist = T(0);
    Contact<T> c;
    contact_params(P, dist, 1, c);
    const int s = L_FR + 6 * B;
    st3<T, LW>(lds, s, arm);
    st3<T, LW>(lds, L_FRN + 3 * B, n);
    lds[(s + 3) * LW] = c.c0; lds[(s + 4) * LW] = c.Dn; lds[(s + 5) * LW] = hit ? T(1) : T(0);
  }
  // bar pair B = (pA, pB)
  {
    constexpr int pA = B == 2 ? 1 : 0, pB = B == 0 ? 1 : 2;
    Contact<T> c0, c1;
    int dropped = 0;
    bar_pair_contacts(P, S.x[pA], R[pA].ez, S.x[pB], R[pB].ez, c0, c1, dropped);
    store_pair<T, LW>(lds, 2 * B, c0);
    store_pair<T, LW>(lds, 2 * B + 1, c1);
    lds[(L_RED + 32 + B) * LW] = T(dropped);
  }
}

// ------------------------------------------------------------------ the sub-step
// lds: this lane's column base (lds_block + lane); needs L_TOTAL slots of pitch LW
// `last` is a template parameter: what only the last sub-step records (lagged kinematics, tendon lengths, contact wrench)
// would otherwise be carried through all twenty iterations of the caller's loop as live registers
template <typename T, int LW, bool last, int MODEL>
TGR_DEV void substep(const Params<T>& P, Sim<T>& S, T* lds, int wid, Lag<T>& lag) {
  const T im = rcp_(P.mass);
  TGR_STAMP(7);
  __syncthreads();  // every wave is done with the previous sub-step's columns
  TGR_STAMP(6);
  // ---- base phase, split over the waves (every wave holds the full state S in registers, so nothing has to be
  // exchanged before the work): wave B = bar B: its kinematic columns, the six tendons attached to it (a tendon is
  // evaluated by both of its bars' waves; the wave of its first bar writes the Jacobian columns), its unconstrained acceleration, its two
  // floor caps, and the bar pair with index B.  Wave 3 has nothing to do here.
  Rot<T> R[3];
#pragma unroll
  for (int b = 0; b < 3; b++) { normalize_quat(S.q[b]); R[b] = quat_rot(S.q[b]); }
  if (last) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
      lag.z[b] = R[b].ez;
#pragma unroll
      for (int i = 0; i < 4; i++) lag.q[b][i] = S.q[b][i];
    }
    // (the centre is formed in double from the T coordinates: exact for fp64 handles, as good as the T state allows for fp32 ones)
    const double cx = ((double)S.x[0].x + (double)S.x[1].x + (double)S.x[2].x) * (1.0 / 3.0), cy = ((double)S.x[0].y + (double)S.x[1].y + (double)S.x[2].y) * (1.0 / 3.0);
    const T cz = (S.x[0].z + S.x[1].z + S.x[2].z) * T(1.0 / 3.0);
    lag.com[0] = cx; lag.com[1] = cy;
#pragma unroll
    for (int b = 0; b < 3; b++) lag.x[b] = mk(T((double)S.x[b].x - cx), T((double)S.x[b].y - cy), S.x[b].z - cz);
  }
  if (wid == 0) base_bar<T, LW, 0, last, MODEL>(P, S, R, lds, lag);
  else if (wid == 1) base_bar<T, LW, 1, last, MODEL>(P, S, R, lds, lag);
  else if (wid == 2) base_bar<T, LW, 2, last, MODEL>(P, S, R, lds, lag);
  TGR_STAMP(0);
  __syncthreads();  // the columns of this sub-step are complete
  T nact = T(0);  // the flags are 0/1: one batch of reads and a sum (a short-circuit `||` would expose twelve LDS round trips)
#pragma unroll
  for (int k = 0; k < 6; k++) nact += lds[(L_CF + 6 * k + 5) * LW] + lds[(L_CP + 12 * k + 11) * LW];
  if (MODEL) {
    nact += lds[(L_FR + 5) * LW] + lds[(L_FR + 11) * LW] + lds[(L_FR + 17) * LW];
#pragma unroll
    for (int k = 0; k < 6; k++) nact += lds[(L_F2 + 6 * k + 5) * LW];
  }
  const bool any_contact = nact != T(0);
  if (last) lag.overflow = (int)(lds[(L_RED + 32) * LW] + lds[(L_RED + 33) * LW] + lds[(L_RED + 34) * LW]);
  if (P.warm_start && wid < 3 && !any_contact) {
    // mujoco's warm start (optional): L_A still holds last sub-step's solution.  A lane without contacts must start (and
    // stay) at a0; nobody else looks at its rows of L_A (all its contact slots are inactive), so no barrier is needed.
    const int b = wid;
    st3<T, LW>(lds, L_A + 6 * b, ld3<T, LW>(lds, L_A0 + 6 * b));
    st3<T, LW>(lds, L_A + 6 * b + 3, ld3<T, LW>(lds, L_A0 + 6 * b + 3));
  }
  const T scale = rcp_((T(3) * P.mass + T(2) * P.Ixx + P.Izz) * T(1.0 / 6.0) * T(18));
  bool conv = !any_contact;
  T cfrc_max = T(0);
  const unsigned cpl = pair_coupling<T, LW>(lds);
#ifdef TGR_STAMPS
  int n_lane = 0;
#endif
  int n_it = 0, n_ls = 0;
#pragma unroll 1
  for (int it = 0;; it++) {
    TGR_STAMP(6);
    T gn = newton_assemble<T, LW, MODEL>(P, lds, wid, cpl, cfrc_max);
    TGR_STAMP(1);
    conv = conv || !(scale * sqrt_(gn) >= P.tol);
    if (__all(conv) || it >= P.max_iter) break;
    n_it++;
#ifdef TGR_STAMPS
    lag.stamp[22] += conv ? 0 : 1000000;  // iterations this lane itself still needed (diagnostic build)
    lag.stamp[23] += 1000000;             // iterations the block ran
    n_lane += conv ? 0 : 1;
#endif
    newton_direction<T, LW>(lds, wid, cpl, conv, lag);
    TGR_STAMP(2);
    // Gauss part of the line derivative  phi_g'(alpha) = g1 + alpha g2 ;  d0 = G.P  (per-bar partials, then summed)
    if (wid < 3) {
      const int b = wid;
      V3<T> ez = ld3<T, LW>(lds, L_EZ + 3 * b);
      V3<T> pl = ld3<T, LW>(lds, L_P + 6 * b), pa = ld3<T, LW>(lds, L_P + 6 * b + 3);
      V3<T> dl = ld3<T, LW>(lds, L_A + 6 * b) - ld3<T, LW>(lds, L_A0 + 6 * b);
      V3<T> da = ld3<T, LW>(lds, L_A + 6 * b + 3) - ld3<T, LW>(lds, L_A0 + 6 * b + 3);
      V3<T> Ipa = inertia_mul(P, ez, pa);
      lds[(L_RED + 22 + 3 * b) * LW] = P.mass * dot(pl, dl) + dot(Ipa, da);
      lds[(L_RED + 23 + 3 * b) * LW] = P.mass * dot(pl, pl) + dot(Ipa, pa);
      lds[(L_RED + 24 + 3 * b) * LW] = dot(pl, ld3<T, LW>(lds, L_G + 6 * b)) + dot(pa, ld3<T, LW>(lds, L_G + 6 * b + 3));
    }
    // the line-search slots, and their contribution at the first trial point alpha = 1, ride on the same barrier
    LSlot<T> lsl[num_ls<MODEL>()];
    ls_setup<T, LW, MODEL>(P, lds, wid, cpl, lsl);
    n_ls++;
    {
      T p1 = T(0), p2 = T(0);
#pragma unroll
      for (int k = 0; k < num_ls<MODEL>(); k++) {
        if (__any(lsl[k].act)) {
          T a, b;
          ls_slot(P, lsl[k], T(1), a, b);
          p1 += a; p2 += b;
        }
      }
      const int base = L_RED + 6 + 8 * (n_ls & 1);
      lds[(base + 2 * wid) * LW] = p1;
      lds[(base + 1 + 2 * wid) * LW] = p2;
    }
    TGR_STAMP(3);
    __syncthreads();
    T g1 = T(0), g2 = T(0), d0 = T(0);
#pragma unroll
    for (int b = 0; b < 3; b++) { g1 += lds[(L_RED + 22 + 3 * b) * LW]; g2 += lds[(L_RED + 23 + 3 * b) * LW]; d0 += lds[(L_RED + 24 + 3 * b) * LW]; }
    T d1 = g1 + g2, d2 = g2;
    {
      const int base = L_RED + 6 + 8 * (n_ls & 1);
#pragma unroll
      for (int w = 0; w < 4; w++) { d1 += lds[(base + 2 * w) * LW]; d2 += lds[(base + 1 + 2 * w) * LW]; }
    }
    // mujoco's second stopping rule: expected improvement of this step (|d0| / 2) below tolerance
    conv = conv || !(scale * T(0.5) * abs_(d0) >= P.tol);
    conv = conv || !(d0 < T(0));  // no descent left at working precision
    // exact line search: safeguarded Newton on phi'; (d1, d2) is phi', phi'' at the current alpha
    T alpha = T(1), lo = T(0), hi = T(-1);
    bool ls_done = conv;
    if (ls_done) alpha = T(0);
    const T gtol = P.ls_tol * abs_(d0);
#pragma unroll 1
    for (int ls = 0;; ls++) {
      if (!ls_done) {
        if (abs_(d1) <= gtol) ls_done = true;
        else {
          if (d1 < T(0)) lo = alpha; else hi = alpha;
          T na = d2 > T(0) ? alpha - d1 * rcp_(d2) : T(-1);
          if (hi < T(0)) { if (!(na > lo) || na > T(10) * (lo + T(1))) na = T(2) * alpha + T(1e-3); }
          else if (!(na > lo) || !(na < hi)) na = T(0.5) * (lo + hi);
          if (hi > T(0) && (hi - lo) <= T(1e-6) * hi) ls_done = true;
          alpha = na;
        }
      }
      if (__all(ls_done) || ls + 1 >= P.ls_iter) break;
      n_ls++;
      newton_dphi<T, LW, num_ls<MODEL>()>(P, lds, wid, lsl, n_ls & 1, alpha, g1, g2, d1, d2);
    }
#ifdef TGR_DEBUG_ENV
    if (wid == 0 && (int)(blockIdx.x * LW + (threadIdx.x & 63)) == TGR_DEBUG_ENV)
      printf("it %d cpl %u gn %.4e d0 %.4e g1 %.3e g2 %.3e alpha %.5f conv %d P0 %.4e P6 %.4e P12 %.4e\n", it, cpl, (double)gn, (double)d0, (double)g1, (double)g2,
             (double)alpha, (int)conv, (double)lds[L_P * LW], (double)lds[(L_P + 6) * LW], (double)lds[(L_P + 12) * LW]);
#endif
    TGR_STAMP(4);
    if (wid < 3) {
#pragma unroll
      for (int i = 0; i < 6; i++) { T pi = lds[(L_P + 6 * wid + i) * LW]; if (alpha != T(0)) lds[(L_A + 6 * wid + i) * LW] += alpha * pi; }
    }
    __syncthreads();
  }
  TGR_STAMP(6);
#ifdef TGR_STAMPS
  lag.stamp[12 + (n_lane > 9 ? 9 : n_lane)] += 1000000;  // histogram: iterations this lane needed
  lag.stamp[24 + (n_it > 7 ? 7 : n_it)] += 1000000;      // histogram: iterations the block ran (7 = 7 or 8)
#endif
  if (last) {
    lag.cfrc_absmax = cfrc_max;
    T sq = T(0);
#pragma unroll
    for (int i = 0; i < 18; i++) { T v = clamp_(lds[(L_FC + i) * LW], P.cf_lo, P.cf_hi); sq += v * v; }
    lag.cfrc_clip_sq = sq;
    lag.bar_contact = lds[(L_RED + 31) * LW];
  }
  lag.n_it += n_it; lag.n_ls += n_ls; lag.n_unconv += conv ? 0 : 1;
  // (the solution stays in L_A: it is next sub-step's warm start and, after the last sub-step, qacc_warmstart)

  // ---- implicitfast: (M + h sum_t b_t J_t J_t^T) a_new = F_smooth + F_constraint   (Woodbury over 9 tendons)
  V3<T> yl[3], ya[3], ez[3];
#pragma unroll
  for (int b = 0; b < 3; b++) {
    ez[b] = ld3<T, LW>(lds, L_EZ + 3 * b);
    yl[b] = ld3<T, LW>(lds, L_A0 + 6 * b) + im * ld3<T, LW>(lds, L_FC + 6 * b);
    ya[b] = ld3<T, LW>(lds, L_A0 + 6 * b + 3) + inv_inertia(P, ez[b], ld3<T, LW>(lds, L_FC + 6 * b + 3));
  }
  {
    // tendon Jacobians back from LDS, scaled by sqrt(h b_t);  W = M^-1 J
    V3<T> u[9], ma[9], mb[9], wa_[9], wb_[9];
    T zt[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int ba = Tab::TEN[t][0], bb = Tab::TEN[t][2];
      T sc = lds[(L_TEN + 10 * t + 9) * LW];
      u[t] = sc * ld3<T, LW>(lds, L_TEN + 10 * t);
      ma[t] = sc * ld3<T, LW>(lds, L_TEN + 10 * t + 3);
      mb[t] = sc * ld3<T, LW>(lds, L_TEN + 10 * t + 6);
      wa_[t] = inv_inertia(P, ez[ba], ma[t]);
      wb_[t] = inv_inertia(P, ez[bb], mb[t]);
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
        Sm[tri(i, j)] = s;
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
  if (MODEL) {  // mj_advance: explicit Euler on the activations, act_dot = (ctrl - act) / tau evaluated with this sub-step's act
#pragma unroll
    for (int a = 0; a < 6; a++) S.act[a] += P.h * (clamp_(S.ctrl[a], P.ctrl_lo, P.ctrl_hi) - S.act[a]) * P.act_itau;
  }
}

}  // namespace tgr
