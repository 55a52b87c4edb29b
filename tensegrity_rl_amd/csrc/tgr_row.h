// tgr_row.h -- one physics sub-step of the 3-bar tensegrity robot in the "one environment = one DPP row" layout
// (tgr_spmd.h): 16 lanes per environment, 4 environments per wavefront, no barriers and no cross-wave traffic inside the sub-step.
//
// What it computes is what the reference gets from mujoco==2.3.7 mj_step for 3prism_jonathan_steady_side.xml (call sites
// tensegrity_env.py:297, tr_env.py:346,812): tendon/actuator forces, soft contacts with elliptic cones solved by Newton's
// method, implicitfast integration -- the same convex problem and the same formulation (world-frame twists, frame-free contact
// rows, "isotropic + rank-1" cone Hessian) as the block kernels of tgr_device.h, which remain the cross-check.  What differs is
// the mapping onto the machine:
//   replicated in the row : rigid-body state, Newton iterate d (18), unconstrained acceleration a0, step p, all scalars -- kept in LDS slabs
//                            between uses, not in registers
//   bar role     (lane b < 3)  : what concerns ONE bar at a time -- kinematics, its tendon and constraint wrench, gradient, its share of the
//                                 norms and of the line search's Gauss part, iterate update, advance: lane b works on bar b and stores the
//                                 result in the slab (lanes 3.. repeat bar 0)
//   tendon role  (lane t < 9)  : geometry, force and Jacobian of tendon t; row t of the 9x9 implicitfast (Woodbury) system
//   slot role    (lane c < 16) : contact slot c -- cap c on the floor (c < 6), the r-cylinders of bar pair (c-6)>>1 side to side (c = 6, 8, 10),
//                                 MODEL 1: bar c-12 on the terrain (c = 12..14), and the POOL (every other contact between two bars, the pair a
//                                 run-time value of the lane): detection, cone evaluation, line-search coefficients
//   dof role     (lane i < 6)  : row i of every 6x6 block of the 18x18 Newton Hessian (repeated on lanes 8..13: a pair slot's two bodies are
//                                 worked on side by side); block Cholesky with row_newbcast
// A wavefront iterates until its four environments have converged (the block kernels wait for 64).
#pragma once
#include "tgr_spmd.h"

namespace tgr {
namespace row {

#if defined(TGR_OTHERS_HIST) && !defined(TGR_ROW_EMUL)
__device__ unsigned long long g_others_hist[64];
#endif

template <typename S> struct RParams {
  S h, grav, mass, Ixx, Izz;
  S cap_r, bar_half, rcyl_r, bcyl_r, bcyl_lo, bcyl_hi, site_off, site_z;
  S ten_k[9], ten_b[9], ten_L0[9];
  S gain, bias0, bias1, bias2, force_lo, force_hi, ctrl_lo, ctrl_hi;
  S K, B;
  S imp0, imp1, imp_width, imp_mid;
  S f0, f1sq, f3sq, f4sq, impratio;
  S invweight;
  S tol, rtol, ls_tol;  // stopping rule: sqrt(g^T M^-1 g) < max(tol, rtol * sqrt(d^T M d)) -- tgr_sim.hip params_to_row
  S cf_lo, cf_hi;
  int max_iter, ls_iter, skip_clamped, warm_start;
  S act_itau;
  S hf_rx, hf_ry, hf_ez, hf_x0, hf_y0, hf_z0, hf_dx, hf_dy;
  int hf_nrow, hf_ncol;
  const float* hf;
  // derived on the host (derive()): reciprocals and products of the above.  Formed inside the kernel they are divisions of uniform values --
  // an IEEE division is ten instructions, twice that in fp64 -- repeated wherever the compiler does not keep the quotient in an SGPR
  S imass, iIxx, iIzz, iimp_width, iimp_mid, iimp_1m, mu, rf0, dmf, dtf, ihf_dx, ihf_dy;
  void derive() {
    imass = S(1) / mass; iIxx = S(1) / Ixx; iIzz = S(1) / Izz;
    iimp_width = S(1) / imp_width; iimp_mid = S(1) / imp_mid; iimp_1m = S(1) / (S(1) - imp_mid);
    mu = f0 / std::sqrt(impratio); rf0 = S(1) / (f0 * f0);
    dmf = S(1) / (mu * mu * (S(1) + mu * mu)); dtf = impratio / (f0 * f0);
    ihf_dx = S(1) / hf_dx; ihf_dy = S(1) / hf_dy;
  }
};

// static model tables (3prism_jonathan_steady_side.xml:82-87,127-164; ..._uneven_ground.xml:76-81,122-158)
struct RTab {
  static constexpr int S0[6][3] = {{-1, 0, 1}, {0, 1, 1}, {1, 0, 1}, {0, 1, -1}, {1, 0, -1}, {0, -1, -1}};
  static constexpr int S1[6][3] = {{-1, 0, 1}, {1, 0, 1}, {0, -1, 1}, {0, -1, -1}, {0, 1, -1}, {1, 0, -1}};
  static constexpr int TEN0[9][4] = {{0, 1, 2, 0}, {0, 0, 1, 1}, {1, 0, 2, 1}, {0, 5, 2, 4}, {0, 4, 1, 5}, {1, 4, 2, 5}, {0, 3, 2, 2}, {0, 2, 1, 3}, {1, 2, 2, 3}};
  static constexpr int TEN1[9][4] = {{0, 0, 2, 1}, {0, 2, 1, 1}, {1, 0, 2, 2}, {0, 5, 2, 3}, {0, 4, 1, 3}, {1, 5, 2, 4}, {0, 3, 2, 0}, {0, 1, 1, 4}, {1, 2, 2, 5}};
};
constexpr int ten_ba(int t) { return RTab::TEN0[t][0]; }  // the bars of a tendon are the same in both models
constexpr int ten_bb(int t) { return RTab::TEN0[t][2]; }
constexpr int pair_a(int p) { return p == 2 ? 1 : 0; }    // bar pairs (A,B): (0,1), (0,2), (1,2)
constexpr int pair_b(int p) { return p == 0 ? 1 : 2; }
// contact slot c of page 0: 0..5 cap c on the floor; 6, 8, 10: the r-cylinders of bar pair (c-6)>>1 side to side; MODEL 1: 12..14 rod of bar
// c-12 on the terrain; the POOL -- MODEL 0: lanes 7, 9, 11, 12..15, MODEL 1: 7, 9, 11, 15 -- every OTHER contact between two bars (a crossing
// deeper than 3.1 mm touches the b-cylinders inside the r-cylinders as well, bar ends meet cap to cap or cap to cylinder: mujoco keeps
// every geom pair), whichever pair it belongs to: the pair a pool lane serves (pl) is a run-time value of the lane, and its bars follow it
// (lane_roles); page 1 (MODEL 1 only): 0..5 second terrain contact of cap c
template <int MODEL> constexpr unsigned pool_mask() { return MODEL ? 0x8a80u : 0xfa80u; }
template <int MODEL> constexpr int pool_size() { return MODEL ? 4 : 7; }
template <int MODEL> constexpr int slot_bar_a(int page, int c) {
  return page != 0 ? -1 : ((c == 6 || c == 8 || c == 10) ? pair_a((c - 6) >> 1) : -1);
}
template <int MODEL> constexpr int slot_bar_b(int page, int c) {
  return c < 6 ? (c >> 1) : (page != 0 ? -1 : ((c == 6 || c == 8 || c == 10) ? pair_b((c - 6) >> 1) : ((MODEL && c >= 12 && c < 15) ? c - 12 : -1)));
}
template <int MODEL> constexpr int n_pages() { return MODEL ? 2 : 1; }

// ------------------------------------------------------------------ replicated state of one environment
// K: the lane type of pose, kinematics, tendon geometry and smooth forces (double in the mixed-precision production kernels: a tendon
// length must be good to ~1e-9 m for the stated 1e-4 relative bound on velocities of a few cm/s); F: everything else
// what the reference reads after mj_step: quantities of the forward pass of the LAST sub-step (one sub-step old)
template <typename K, typename F> struct RLag {
  // (the pose and the tendon lengths at the start of the last sub-step -- what the observation is built from -- are parked in the lp
  // slab, the bar axes stay in kd: thirty values that would otherwise sit in registers through the whole last sub-step)
  F cfrc_absmax, cfrc_clip_sq, bar_contact;
  F overflow;   // dropped bar-bar contacts in the last sub-step (a count, kept as a real)
  F n_unconv;   // sub-steps this environment left the solver at the iteration cap
  F n_own;      // Newton iterations this environment itself still needed (the wavefront runs until its slowest one is done)
  int n_it, n_ls;
#ifdef TGR_ROW_STAMPS  // diagnostic build: shader cycles per phase of the sub-step (s_memtime), summed over the wavefront's sub-steps
  unsigned long long stamp[24], t_last;
#endif
};
#if defined(TGR_ROW_STAMPS) && !defined(TGR_ROW_EMUL)
#define TGR_RSTAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); lag.stamp[i] += _t - lag.t_last; lag.t_last = _t; } while (0)
#else
#define TGR_RSTAMP(i) ((void)0)
#endif
template <typename F> struct Rot { V3<F> ex, ey, ez; };

template <typename F> TGR_RDEV void normalize_quat(F* q) {
  F s = rsqrt_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] = q[0] * s; q[1] = q[1] * s; q[2] = q[2] * s; q[3] = q[3] * s;
}
template <typename F> TGR_RDEV Rot<F> quat_rot(const F* q) {
  typedef typename LaneTraits<F>::S S;
  F q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  F q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  F q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  Rot<F> R;
  R.ex = mk<F>(q00 + q11 - q22 - q33, S(2) * (q12 + q03), S(2) * (q13 - q02));
  R.ey = mk<F>(S(2) * (q12 - q03), q00 - q11 + q22 - q33, S(2) * (q23 + q01));
  R.ez = mk<F>(S(2) * (q13 + q02), S(2) * (q23 - q01), q00 - q11 - q22 + q33);
  return R;
}
template <typename F> TGR_RDEV V3<F> rot_apply(const Rot<F>& R, V3<F> l) { return l.x * R.ex + (l.y * R.ey + l.z * R.ez); }
template <typename F> TGR_RDEV V3<F> rot_applyT(const Rot<F>& R, V3<F> g) { return mk<F>(dot(R.ex, g), dot(R.ey, g), dot(R.ez, g)); }
template <typename F, typename S> TGR_RDEV V3<F> inv_inertia(const RParams<S>& P, V3<F> z, V3<F> t) {
  const S iI = P.iIxx, d = P.iIzz - iI;
  return iI * t + (d * dot(z, t)) * z;
}
template <typename F, typename S> TGR_RDEV V3<F> inertia_mul(const RParams<S>& P, V3<F> z, V3<F> a) {
  return P.Ixx * a + ((P.Izz - P.Ixx) * dot(z, a)) * z;
}
// lane-dependent choice among the three bars' replicated vectors
template <bool USEK, typename SK, typename S> TGR_RDEV const auto& pick_params(const RParams<SK>& PK, const RParams<S>& P) {
  if constexpr (USEK) return PK; else return P;
}
template <typename F, typename I> TGR_RDEV V3<F> pick_bar(I b, const V3<F>* a) { return sel3(b == 0, a[0], sel3(b == 1, a[1], a[2])); }
template <typename F, typename I> TGR_RDEV V3<F> pick_bar0(I b, const V3<F>* a) {  // b < 0: the world (zero)
  typedef typename LaneTraits<F>::S S;
  const V3<F> z = mk<F>(F(S(0)), F(S(0)), F(S(0)));
  return sel3(b == 0, a[0], sel3(b == 1, a[1], sel3(b == 2, a[2], z)));
}
template <typename F, typename I> TGR_RDEV F pick_comp(I k, V3<F> a) { return sel(k == 0, a.x, sel(k == 1, a.y, a.z)); }

// ------------------------------------------------------------------ per-lane role constants (set up once per kernel)
template <typename F> struct Roles {
  typedef typename LaneTraits<F>::I I;
  typedef typename LaneTraits<F>::B B;
  I r;
  // tendon role (its K constants -- site coordinates, stiffness, damping, spring length -- live in the role slab)
  B is_ten; I ba, bb; B actuated;
  // slot role (page 0)
  B is_cap, is_side, is_pool, is_rod, pooled; I sA, sB; F cap_sign;  // pooled: a pool lane that carries a contact
  // dof role
  B is_dof, is_ang; I k;
};
// the role constants that are not worth a register: tendon t's site coordinates, stiffness, damping, spring length -> role slab
template <typename K, typename F, typename SK, int MODEL> TGR_RDEV void store_role_consts(const RParams<SK>& P, const Slab<K>& rd) {
  typedef typename LaneTraits<F>::I I;
  const I r = lane_id();
  I sia = I(0), sib = I(0);
  K tk = K(SK(0)), tb = K(SK(0)), tL0 = K(SK(0));
  static_for<9>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const auto m = r == t;
    sia = sel(m, I(MODEL ? RTab::TEN1[t][1] : RTab::TEN0[t][1]), sia); sib = sel(m, I(MODEL ? RTab::TEN1[t][3] : RTab::TEN0[t][3]), sib);
    tk = sel(m, K(P.ten_k[t]), tk); tb = sel(m, K(P.ten_b[t]), tb); tL0 = sel(m, K(P.ten_L0[t]), tL0);
  });
  rd.st(6, tk); rd.st(7, tb); rd.st(8, tL0);  // RTK, RTB, RTL0
  V3<K> sa = mk<K>(K(SK(0)), K(SK(0)), K(SK(0))), sb = sa;
  static_for<6>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    const V3<K> loc = MODEL ? mk<K>(K(SK(RTab::S1[s][0]) * P.site_off), K(SK(RTab::S1[s][1]) * P.site_off), K(SK(RTab::S1[s][2]) * P.site_z))
                            : mk<K>(K(SK(RTab::S0[s][0]) * P.site_off), K(SK(RTab::S0[s][1]) * P.site_off), K(SK(RTab::S0[s][2]) * P.site_z));
    sa = sel3(sia == s, loc, sa); sb = sel3(sib == s, loc, sb);
  });
  st3(rd, 0, sa); st3(rd, 3, sb);  // RSA, RSB
}
// what lane r does: pure arithmetic on the lane number, recomputed where it is needed (a role kept in a register for the whole kernel
// costs more than the few instructions that rebuild it)
// pl: the bar pair a POOL lane's contact belongs to in this sub-step (a lane value; < 0: none)
template <typename F, int MODEL> TGR_RDEV Roles<F> lane_roles(typename LaneTraits<F>::I pl = typename LaneTraits<F>::I(-1)) {
  typedef typename LaneTraits<F>::I I;
  typedef typename LaneTraits<F>::S S;
  Roles<F> R;
  I r = lane_id();
  TGR_LAUNDER(r);
  R.r = r;
  R.is_ten = r < 9;
  // bars of tendon r (both models): TEN0[t][0], TEN0[t][2]
  R.ba = sel((r == 2) | (r == 5) | (r == 8), I(1), I(0));
  R.bb = sel((r == 1) | (r == 4) | (r == 7), I(1), I(2));
  R.actuated = r < 6;
  R.is_cap = r < 6;
  R.is_side = (r == 6) | (r == 8) | (r == 10);
  R.is_rod = MODEL ? ((r >= 12) & (r < 15)) : (r < 0);
  R.is_pool = (r == 7) | (r == 9) | (r == 11) | (MODEL ? (r == 15) : (r >= 12));
  R.pooled = R.is_pool & (pl >= 0);
  const I pr = sel(R.pooled, pl, (r - 6) >> 1);
  const auto anypair = R.is_side | R.pooled;
  R.sA = sel(anypair, sel(pr == 2, I(1), I(0)), I(-1));
  R.sB = sel(R.is_cap, r >> 1, sel(anypair, sel(pr == 0, I(1), I(2)), sel(R.is_rod, r - 12, I(-1))));
  R.cap_sign = sel((r & 1) == 0, F(S(1)), F(S(-1)));
  R.is_dof = r < 6;
  R.is_ang = (r >= 3) & (r < 6);
  R.k = sel(r < 3, r, r - 3);
  return R;
}
static_assert(RTab::TEN0[2][0] == 1 && RTab::TEN0[5][0] == 1 && RTab::TEN0[8][0] == 1 && RTab::TEN0[0][0] == 0 && RTab::TEN0[1][0] == 0 &&
              RTab::TEN0[3][0] == 0 && RTab::TEN0[4][0] == 0 && RTab::TEN0[6][0] == 0 && RTab::TEN0[7][0] == 0, "lane_roles: bar a of a tendon");
static_assert(RTab::TEN0[1][2] == 1 && RTab::TEN0[4][2] == 1 && RTab::TEN0[7][2] == 1 && RTab::TEN0[0][2] == 2 && RTab::TEN0[2][2] == 2 &&
              RTab::TEN0[3][2] == 2 && RTab::TEN0[5][2] == 2 && RTab::TEN0[6][2] == 2 && RTab::TEN0[8][2] == 2, "lane_roles: bar b of a tendon");

// ------------------------------------------------------------------ contacts
template <typename F> struct RContact {
  typedef typename LaneTraits<F>::B B;
  V3<F> n, rA, rB;  // normal A -> B, contact point relative to the COMs
  F c0, Dn;         // K imp dist (< 0 when penetrating), 1/R of the normal row
  B active;
};
template <typename F, typename S> TGR_RDEV F impedance(const RParams<S>& P, F dist) {
  F xx = abs_(dist) * P.iimp_width;
  F y = sel(xx >= S(1), F(S(1)), sel(xx <= P.imp_mid, xx * xx * P.iimp_mid, S(1) - (S(1) - xx) * (S(1) - xx) * P.iimp_1m));
  return P.imp0 + y * (P.imp1 - P.imp0);
}
template <typename F, typename S> TGR_RDEV void contact_params(const RParams<S>& P, F dist, S nbody, F& c0, F& Dn) {
  F imp = impedance(P, dist);
  F R0 = max_((S(1) - imp) * (P.invweight * nbody) * rcp_(imp), F(S(1e-15)));
  Dn = rcp_(R0);
  c0 = P.K * imp * dist;
}

template <typename F> struct RCone {
  V3<F> fl, fa;
  F a1, a2, b1, b2, cg, cw;
  V3<F> gl, ga, wl, wa;
};
// cone force and Hessian ("isotropic + rank-1" form) of one contact at jar = (jl, ja); same algebra as tgr_device.h cone_eval
template <typename F, typename S> TGR_RDEV RCone<F> cone_eval(const RParams<S>& P, const RContact<F>& c, V3<F> jl, V3<F> ja) {
  RCone<F> o;
  const S mu = P.mu;
  const S rf0 = P.rf0;
  F Dt = c.Dn * P.impratio;
  F Do = Dt * (P.f3sq * rf0), Dr = Dt * (P.f4sq * rf0);
  F jn = dot(c.n, jl);
  V3<F> jt = jl - jn * c.n;
  F jo = dot(c.n, ja);
  V3<F> jr = ja - jo * c.n;
  F tt = P.f1sq * dot(jt, jt) + P.f3sq * jo * jo + P.f4sq * dot(jr, jr);
  // |T| and 1 / |T| from ONE transcendental (rsq): sqrt followed by rcp of its result is two, one waiting for the other
  const F rT = rsqrt_(max_(tt, F(S(1e-36))));
  F Tn = tt * rT;
  F N = mu * jn;
  auto top = (N >= mu * Tn) | !c.active;
  auto bottom = !top & (mu * N + Tn <= S(0));
  auto middle = !top & !bottom;
  V3<F> flB = -((c.Dn * jn) * c.n + Dt * jt);
  V3<F> faB = -((Do * jo) * c.n + Dr * jr);
  F Dm = c.Dn * P.dmf;
  F NT = N - mu * Tn;
  F fn = -Dm * NT * mu;
  V3<F> wl = P.f1sq * jt;
  V3<F> wa = (P.f3sq * jo) * c.n + P.f4sq * jr;
  V3<F> flM = fn * c.n - (fn * rT) * wl;
  V3<F> faM = -(fn * rT) * wa;
  const F zero = F(S(0));
  const V3<F> z3 = mk<F>(zero, zero, zero);
  o.fl = sel3(bottom, flB, sel3(middle, flM, z3));
  o.fa = sel3(bottom, faB, sel3(middle, faM, z3));
  F kap = -Dm * mu * NT * rT;
  o.a1 = sel(bottom, Dt, sel(middle, kap * P.f1sq, zero));
  o.a2 = sel(bottom, c.Dn - Dt, sel(middle, -kap * P.f1sq, zero));
  o.b1 = sel(bottom, Dr, sel(middle, kap * P.f4sq, zero));
  o.b2 = sel(bottom, Do - Dr, sel(middle, kap * (P.f3sq - P.f4sq), zero));
  o.cg = sel(middle, Dm, zero);
  o.cw = sel(middle, -kap * rT * rT, zero);
  o.gl = mu * c.n - (mu * rT) * wl;
  o.ga = -(mu * rT) * wa;
  o.wl = wl;
  o.wa = wa;
  return o;
}

// ---- contact geometry between two bars (slot role); same rules as tgr_device.h bar_pair_contacts.
// A pair of bars has ONE side-to-side contact of its r-cylinders (a fixed slot: lanes 6, 8, 10) and any number of OTHER contacts -- cap
// against cap, cap against a cylinder, b-cylinders inside the r-cylinders of a deep crossing: 24 candidate geometries -- which mujoco keeps
// all of.  The others go to a POOL of lanes that carry the contact of a run-time pair (MODEL 0: lanes 7, 9, 11, 12..15; MODEL 1: 7, 9, 11,
// 15), handed out in the order (rank by depth, pair).  The candidates are enumerated once per rank -- pick the next one in the order
// (distance, candidate number) after the previous pick -- instead of keeping the k deepest in registers: k is whatever the pool holds, the
// enumeration is only entered by the wavefronts that have such a contact at all (one sub-step in a hundred), and a pick is routed to its pool
// lane the moment it is made.  (Until round 4: three fixed ranks per pair + one roving lane; a bar crossing deeper than 6 mm near the caps
// has four to six others -- profiles/round4_other_contacts_histogram.log -- and the rest was dropped and counted.)
template <typename F> struct RPairGeom {
  typedef typename LaneTraits<F>::B B;
  V3<F> xA, zA, xB, zB, pa, n;
  F s, t, dd, capr;
  B near_, okdir, more;  // more: the pair may have other contacts than the side-to-side one (a deep crossing, or bar ends within reach)
};
template <typename F, typename S, typename B, typename V>
TGR_RDEV void sphere_vs_bar(const RParams<S>& P, F capr, V3<F> c, V3<F> xo, V3<F> zo, bool flip, B enable, V&& visit) {
  V3<F> wv = c - xo;
  F t = dot(wv, zo);
  V3<F> rad = wv - t * zo;
  F rho = sqrt_(dot(rad, rad));
  F irho = rcp_(max_(rho, F(S(1e-30))));
  for (int k = 0; k < 3; k++) {
    const S cen = k == 0 ? S(0) : (k == 1 ? S(0.5) * (P.bcyl_hi + P.bcyl_lo) : -S(0.5) * (P.bcyl_hi + P.bcyl_lo));
    const S hh = k == 0 ? P.bar_half : S(0.5) * (P.bcyl_hi - P.bcyl_lo);
    const S rc = k == 0 ? P.rcyl_r : P.bcyl_r;
    F tl = t - cen;
    F tc = clamp_(tl, F(-hh), F(hh));
    F rr = min_(rho, F(rc));
    auto inside = (abs_(tl) <= hh) & (rho <= rc);
    V3<F> q = xo + (cen + tc) * zo + (rr * irho) * rad;
    V3<F> dif = q - c;
    F dd = sqrt_(dot(dif, dif));
    F idd = rcp_(max_(dd, F(S(1e-30))));
    F dist = sel(inside, rho - rc - capr, dd - capr);
    V3<F> n = sel3(inside, -irho * rad, idd * dif);
    V3<F> pos = c + (capr + S(0.5) * dist) * n;
    auto valid = enable & sel(inside, rho > S(1e-12), dd > S(1e-12));
    V3<F> nn = flip ? -n : n;
    visit(dist, nn, pos, valid);
  }
}
// the geometry the two bars share, and c0: r-cylinder against r-cylinder side to side (kept as a candidate up to near0 above touching: the
// caller decides on a distance formed in K).  lanes: the lanes whose outcome is used (they alone steer the wavefront-uniform early exits)
template <typename F, typename S>
TGR_RDEV RPairGeom<F> pair_side_contact(const RParams<S>& P, F capr, V3<F> xA, V3<F> zA, V3<F> xB, V3<F> zB, typename LaneTraits<F>::B lanes,
                                        RContact<F>& c0, S near0 = S(0)) {
  typedef typename LaneTraits<F>::B B;
  const F zero = F(S(0));
  RPairGeom<F> G;
  G.xA = xA; G.zA = zA; G.xB = xB; G.zB = zB; G.capr = capr;
  c0.active = B(false);
  c0.n = mk<F>(zero, zero, F(S(1)));
  c0.rA = mk<F>(zero, zero, zero); c0.rB = c0.rA;
  c0.c0 = zero; c0.Dn = zero;
  V3<F> wv = xA - xB;
  F b = dot(zA, zB), dA = dot(zA, wv), dB = dot(zB, wv);
  F den = S(1) - b * b;
  auto skew = den > S(1e-6);
  F iden = rcp_(sel(skew, den, F(S(1))));
  F s = (b * dB - dA) * iden, t = (dB - b * dA) * iden;
  V3<F> pa = xA + s * zA, pb = xB + t * zB;
  V3<F> dif = pb - pa;
  F dd = sqrt_(dot(dif, dif));
  F idd = rcp_(max_(dd, F(S(1e-30))));
  V3<F> n = idd * dif;
  auto okdir = skew & (dd > S(1e-9));
  G.s = s; G.t = t; G.dd = dd; G.pa = pa; G.n = n; G.okdir = okdir; G.more = B(false);
  {
    F sc = clamp_(sel(skew, s, zero), F(-P.bar_half), F(P.bar_half));
    F tc = clamp_(dB + b * sc, F(-P.bar_half), F(P.bar_half));
    sc = clamp_(b * tc - dA, F(-P.bar_half), F(P.bar_half));
    V3<F> dv = (xB + tc * zB) - (xA + sc * zA);
    const F reach = S(2) * capr + S(1e-6);
    G.near_ = lanes & (dot(dv, dv) <= reach * reach);
  }
  if (!any_lane(G.near_)) return G;
  {
    auto interior = G.near_ & okdir & (abs_(s) < P.bar_half) & (abs_(t) < P.bar_half);
    F dist = dd - S(2) * P.rcyl_r;
    auto hit = interior & (dist < near0);
    c0.active = hit;
    c0.n = sel3(hit, n, c0.n);
    V3<F> pos = pa + (P.rcyl_r + S(0.5) * dist) * n;
    c0.rA = sel3(hit, pos - xA, c0.rA); c0.rB = sel3(hit, pos - xB, c0.rB);
    F cc, dn;
    contact_params(P, sel(hit, dist, zero), S(2), cc, dn);
    c0.c0 = sel(hit, cc, zero); c0.Dn = sel(hit, dn, zero);
  }
  auto deep = G.near_ & okdir & (dd - S(2) * P.rcyl_r < -(P.rcyl_r - P.bcyl_r));
  B ends = B(false);
  {
    const F reach2 = (S(2) * capr + S(1e-6)) * (S(2) * capr + S(1e-6));
    for (int e = 0; e < 2; e++) {
      const S sg = e == 0 ? P.bar_half : -P.bar_half;
      V3<F> ea = (xA + sg * zA) - xB;
      F ta = clamp_(dot(ea, zB), F(-P.bar_half), F(P.bar_half));
      V3<F> da = ea - ta * zB;
      V3<F> eb = (xB + sg * zB) - xA;
      F tb = clamp_(dot(eb, zA), F(-P.bar_half), F(P.bar_half));
      V3<F> db = eb - tb * zA;
      ends = ends | (dot(da, da) <= reach2) | (dot(db, db) <= reach2);
    }
    ends = ends & G.near_;
  }
  G.more = deep | ends;
  return G;
}
// every OTHER candidate contact of the pair, in a fixed order: visit(distance, normal A -> B, position, valid)
template <typename F, typename S, typename V> TGR_RDEV void pair_other_candidates(const RParams<S>& P, const RPairGeom<F>& G, V&& visit) {
  const V3<F> xA = G.xA, zA = G.zA, xB = G.xB, zB = G.zB;
  const S bc = S(0.5) * (P.bcyl_hi + P.bcyl_lo), bh = S(0.5) * (P.bcyl_hi - P.bcyl_lo);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      if (i == 0 && j == 0) continue;
      const S ci = i == 0 ? S(0) : (i == 1 ? bc : -bc), hi_ = i == 0 ? P.bar_half : bh, ri = i == 0 ? P.rcyl_r : P.bcyl_r;
      const S cj = j == 0 ? S(0) : (j == 1 ? bc : -bc), hj_ = j == 0 ? P.bar_half : bh, rj = j == 0 ? P.rcyl_r : P.bcyl_r;
      auto interior = G.near_ & G.okdir & (abs_(G.s - ci) < hi_) & (abs_(G.t - cj) < hj_);
      F dist = G.dd - (ri + rj);
      V3<F> pos = G.pa + (ri + S(0.5) * dist) * G.n;
      visit(dist, G.n, pos, interior);
    }
  for (int ea = 0; ea < 2; ea++) {
    V3<F> ca = xA + (ea == 0 ? P.bar_half : -P.bar_half) * zA;
    sphere_vs_bar(P, G.capr, ca, xB, zB, false, G.near_, visit);
    for (int eb = 0; eb < 2; eb++) {
      V3<F> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
      V3<F> dv = cb - ca;
      F cd = sqrt_(dot(dv, dv));
      F dist = cd - S(2) * G.capr;
      V3<F> nn = rcp_(max_(cd, F(S(1e-30)))) * dv;
      V3<F> pos = ca + (G.capr + S(0.5) * dist) * nn;
      visit(dist, nn, pos, G.near_ & (cd > S(1e-12)));
    }
  }
  for (int eb = 0; eb < 2; eb++) {
    V3<F> cb = xB + (eb == 0 ? P.bar_half : -P.bar_half) * zB;
    sphere_vs_bar(P, G.capr, cb, xA, zA, true, G.near_, visit);
  }
}
// the next candidate in the order (distance, candidate number) after (prev_d, prev_i); count (first call only: prev_i < 0): how many there are
template <typename F> struct RPick { F dist; V3<F> n, p; typename LaneTraits<F>::I idx; typename LaneTraits<F>::B ok; F count; };
template <typename F, typename S>
TGR_RDEV RPick<F> pair_next_other(const RParams<S>& P, const RPairGeom<F>& G, F prev_d, typename LaneTraits<F>::I prev_i) {
  typedef typename LaneTraits<F>::I I;
  typedef typename LaneTraits<F>::B B;
  RPick<F> k;
  k.ok = B(false); k.dist = F(S(0)); k.n = G.n; k.p = G.xA; k.idx = I(-1); k.count = F(S(0));
  int idx = 0;
  pair_other_candidates(P, G, [&](F dist, V3<F> n, V3<F> pos, B valid) {
    const auto hit = valid & (dist < S(0));
    k.count = k.count + sel(hit, F(S(1)), F(S(0)));
    const auto after = (dist > prev_d) | ((dist == prev_d) & (prev_i < idx));
    const auto better = hit & after & (!k.ok | (dist < k.dist));  // (of equally deep ones the first in the enumeration)
    k.dist = sel(better, dist, k.dist); k.n = sel3(better, n, k.n); k.p = sel3(better, pos, k.p); k.idx = sel(better, I(idx), k.idx);
    k.ok = k.ok | better;
    idx++;
  });
  return k;
}

// ------------------------------------------------------------------ heightfield floor (MODEL 1), slot role, select form
// closest point of triangle (a, b, c) to p (Ericson, Real-Time Collision Detection 5.1.5): every region evaluated, the first that applies
// taken -- lanes of a row sit in different regions, so there is nothing to branch on
template <typename F> TGR_RDEV V3<F> closest_on_triangle(V3<F> p, V3<F> a, V3<F> b, V3<F> c, typename LaneTraits<F>::B& interior) {
  typedef typename LaneTraits<F>::S S;
  const V3<F> ab = b - a, ac = c - a, ap = p - a, bp = p - b, cp = p - c;
  const F d1 = dot(ab, ap), d2 = dot(ac, ap), d3 = dot(ab, bp), d4 = dot(ac, bp), d5 = dot(ab, cp), d6 = dot(ac, cp);
  const F vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  const auto rA = (d1 <= S(0)) & (d2 <= S(0));
  const auto rB = (d3 >= S(0)) & (d4 <= d3);
  const auto rAB = (vc <= S(0)) & (d1 >= S(0)) & (d3 <= S(0));
  const auto rC = (d6 >= S(0)) & (d5 <= d6);
  const auto rAC = (vb <= S(0)) & (d2 >= S(0)) & (d6 <= S(0));
  const auto rBC = (va <= S(0)) & ((d4 - d3) >= S(0)) & ((d5 - d6) >= S(0));
  const V3<F> qAB = a + (d1 / (d1 - d3)) * ab;
  const V3<F> qAC = a + (d2 / (d2 - d6)) * ac;
  const V3<F> qBC = b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
  const F den = S(1) / (va + vb + vc);
  const V3<F> qI = a + (vb * den) * ab + (vc * den) * ac;
  interior = !(rA | rB | rAB | rC | rAC | rBC);
  return sel3(rA, a, sel3(rB, b, sel3(rAB, qAB, sel3(rC, c, sel3(rAC, qAC, sel3(rBC, qBC, qI))))));
}
// normalised elevation at grid vertex (cc, rr), clamped to the grid
template <typename F, typename I, typename S> TGR_RDEV F hf_at(const RParams<S>& P, I cc, I rr) {
  const I c = clampi(cc, 0, P.hf_ncol - 1), r = clampi(rr, 0, P.hf_nrow - 1);
  return cvt<F>(gather_f(P.hf, r * P.hf_ncol + c)) * P.hf_ez;
}
// cap sphere (centre given relative to the heightfield's origin, K) against the terrain: the two deepest triangles of the (at most 2x2)
// cells under it.  dist >= 0: no contact.  Same rules as tgr_device.h hfield_cap; coordinates are taken relative to the corner of the first
// cell before they become F, so that a robot 30 m from the origin resolves its contact distance as well as one at the origin.
template <typename K, typename F, typename SK, typename S>
TGR_RDEV void hfield_cap(const RParams<SK>& PK, const RParams<S>& P, F rad, V3<K> pk, F* dist_out, V3<F>* n_out, typename LaneTraits<F>::I* id_out) {
  typedef typename LaneTraits<F>::I I;
  const F zero = F(S(0));
  const F px = cvt<F>(pk.x), py = cvt<F>(pk.y), pz = cvt<F>(pk.z);
  const auto inb = !((px < -P.hf_rx - rad) | (px > P.hf_rx + rad) | (py < -P.hf_ry - rad) | (py > P.hf_ry + rad) | (pz > P.hf_ez + rad));
  const S idx = P.ihf_dx, idy = P.ihf_dy;
  I c0 = to_int(floor_((px - rad + P.hf_rx) * idx)), c1 = to_int(floor_((px + rad + P.hf_rx) * idx));
  I r0 = to_int(floor_((py - rad + P.hf_ry) * idy)), r1 = to_int(floor_((py + rad + P.hf_ry) * idy));
  c0 = clampi(c0, 0, P.hf_ncol - 2); r0 = clampi(r0, 0, P.hf_nrow - 2);   // (an out-of-range cap is masked by inb)
  c1 = clampi(c1, 0, P.hf_ncol - 2); r1 = clampi(r1, 0, P.hf_nrow - 2);
  F hz[3][3];
  for (int j = 0; j < 3; j++)
    for (int i = 0; i < 3; i++) hz[j][i] = hf_at<F>(P, c0 + i, r0 + j);
  // position relative to the corner vertex (c0, r0), formed in K
  const K x0 = cvt<K>(from_int(c0, F())) * PK.hf_dx - PK.hf_rx, y0 = cvt<K>(from_int(r0, F())) * PK.hf_dy - PK.hf_ry;
  const K pxk = pk.x - x0, pyk = pk.y - y0;
  const V3<F> p = mk<F>(cvt<F>(pxk), cvt<F>(pyk), pz);
  F bd0 = F(S(1)), bd1 = F(S(1));
  V3<F> bq0 = p, bq1 = p;
  I bi0 = I(0), bi1 = I(0);  // WHICH facets: (col * 128 + row) * 2 + tri (for the sub-step trace)
  for (int j = 0; j < 2; j++)
    for (int i = 0; i < 2; i++) {
      const auto cell = inb & ((c0 + i) <= c1) & ((r0 + j) <= r1);
      const S xa = S(i) * P.hf_dx, ya = S(j) * P.hf_dy;
      const V3<F> v[4] = {mk<F>(F(xa), F(ya), hz[j][i]), mk<F>(F(xa), F(ya + P.hf_dy), hz[j + 1][i]), mk<F>(F(xa + P.hf_dx), F(ya), hz[j][i + 1]),
                          mk<F>(F(xa + P.hf_dx), F(ya + P.hf_dy), hz[j + 1][i + 1])};  // (c,r) (c,r+1) (c+1,r) (c+1,r+1)
      for (int tri = 0; tri < 2; tri++) {
        typename LaneTraits<F>::B interior;
        const V3<F> q = closest_on_triangle(p, v[tri], v[tri + 1], v[tri + 2], interior);
        const V3<F> df = p - q;
        F dist = sqrt_(dot(df, df)) - rad;
        {
          // A cap that rests on the face of a facet (the usual case): its distance is the distance to the facet's plane, formed in K like the
          // floor distance of the flat model -- in F the metre-sized cell coordinates leave 6e-8 m of noise in it, enough to move a cap
          // across dist = 0 in 0.3 % of env-steps.  Edges of the triangles: tri 0 = (c,r) (c,r+1) (c+1,r), tri 1 = (c,r+1) (c+1,r) (c+1,r+1);
          // both have the normal (-dy (h_e - h_0), -dx (h_n - h_0), dx dy) with (h_0, h_e, h_n) = heights at the corner the two axis-parallel
          // edges meet in, along x and along y from it.
          const K h0 = cvt<K>(tri == 0 ? hz[j][i] : hz[j + 1][i + 1]), he = cvt<K>(tri == 0 ? hz[j][i + 1] : hz[j + 1][i]),
                  hn = cvt<K>(tri == 0 ? hz[j + 1][i] : hz[j][i + 1]);
          const SK sg = tri == 0 ? SK(1) : SK(-1);  // tri 1: the corner is (c+1, r+1), its edges run towards -x and -y
          const K nx = -(PK.hf_dy * (he - h0)) * sg, ny = -(PK.hf_dx * (hn - h0)) * sg, nz = K(PK.hf_dx * PK.hf_dy);
          const K rx = pxk - SK(tri == 0 ? i : i + 1) * PK.hf_dx, ry = pyk - SK(tri == 0 ? j : j + 1) * PK.hf_dy, rz = pk.z - h0;
          const K dpl = (nx * rx + ny * ry + nz * rz) * rsqrt_(nx * nx + ny * ny + nz * nz) - cvt<K>(rad);
          dist = sel(interior, cvt<F>(dpl), dist);
        }
        const auto hit = cell & (dist < S(0));
        const auto first = hit & (dist < bd0), second = hit & !(dist < bd0) & (dist < bd1);
        bd1 = sel(first, bd0, sel(second, dist, bd1)); bq1 = sel3(first, bq0, sel3(second, q, bq1));
        bd0 = sel(first, dist, bd0); bq0 = sel3(first, q, bq0);
        const I fid = ((c0 + i) * 128 + (r0 + j)) * 2 + tri;
        bi1 = sel(first, bi0, sel(second, fid, bi1)); bi0 = sel(first, fid, bi0);
      }
    }
  {
    const V3<F> df = p - bq0;
    const F nn = sqrt_(dot(df, df));
    const auto ok = (bd0 < S(0)) & (nn > S(1e-12));
    dist_out[0] = sel(ok, bd0, F(S(1))); n_out[0] = sel3(ok, rcp_(max_(nn, F(S(1e-30)))) * df, mk<F>(zero, zero, F(S(1))));
  }
  {
    const V3<F> df = p - bq1;
    const F nn = sqrt_(dot(df, df));
    const auto ok = (bd1 < S(0)) & (nn > S(1e-12));
    dist_out[1] = sel(ok, bd1, F(S(1))); n_out[1] = sel3(ok, rcp_(max_(nn, F(S(1e-30)))) * df, mk<F>(zero, zero, F(S(1))));
  }
  id_out[0] = bi0; id_out[1] = bi1;
}
// closest points of two segments c + s a (|s| <= ha) and c2 + t a2 (|t| <= hb), unit axes (Ericson 5.1.9), select form
template <typename F, typename S> TGR_RDEV void seg_seg_closest(V3<F> ca, V3<F> aa, S ha, V3<F> cb, V3<F> ab, F hb, F& s_out, F& t_out) {
  const V3<F> w = ca - cb;
  const F b = dot(aa, ab), dA = dot(aa, w), dB = dot(ab, w);
  const F den = S(1) - b * b;
  const auto skew = den > S(1e-12);
  const F iden = rcp_(sel(skew, den, F(S(1))));
  const F s = sel(skew, (b * dB - dA) * iden, F(S(0))), t = sel(skew, (dB - b * dA) * iden, dB);
  const auto interior = skew & (abs_(s) < ha) & (abs_(t) < hb);
  const F s2 = clamp_(s, F(-ha), F(ha));
  const F t2 = clamp_(dB + b * s2, -hb, hb);
  const F s3 = clamp_(b * t2 - dA, F(-ha), F(ha));
  s_out = sel(interior, s, s3); t_out = sel(interior, t, t2);
}
// the r-cylinder of a bar (centre relative to the heightfield's origin in K, axis az; both REPLICATED) against the terrain: deepest grid
// edge under the bar, the bar taken as a capsule of the cylinder's radius.  Same rules as tgr_device.h hfield_rod (3 x 3 cells x 3 edges in
// the order j, i, k; the first of equally deep edges wins), but the 27 edge tests are dealt out over the 16 lanes of the row (lane l takes
// edges l and l + 16) and the winner comes back through the row: 2 tests per lane instead of 27.  Returns dist (>= 0: none), normal,
// arm -- replicated.  (azk: the bar's axis in K.)
template <typename K, typename F, typename SK, typename S>
TGR_RDEV F hfield_rod_row(const RParams<SK>& PK, const RParams<S>& P, V3<K> ck, V3<K> azk, V3<F>& n_out, V3<F>& arm_out, V3<F>& off_out,
                          typename LaneTraits<F>::I& id_out) {
  typedef typename LaneTraits<F>::I I;
  const S rad = P.rcyl_r, H = P.bar_half, near = S(2e-5);
  const V3<F> az = cvt3<F>(azk);
  const F zero = F(S(0));
  const V3<F> z3 = mk<F>(zero, zero, zero);
  const F cx = cvt<F>(ck.x), cy = cvt<F>(ck.y), cz = cvt<F>(ck.z);
  const F ex = abs_(az.x) * H + rad, ey = abs_(az.y) * H + rad;
  auto inb = !((cx + ex < -P.hf_rx) | (cx - ex > P.hf_rx) | (cy + ey < -P.hf_ry) | (cy - ey > P.hf_ry));
  inb = inb & !(cz - abs_(az.z) * H - rad > P.hf_ez);
  const S idx = P.ihf_dx, idy = P.ihf_dy;
  I c0 = to_int(floor_((cx - ex + P.hf_rx) * idx)), c1 = to_int(floor_((cx + ex + P.hf_rx) * idx));
  I r0 = to_int(floor_((cy - ey + P.hf_ry) * idy)), r1 = to_int(floor_((cy + ey + P.hf_ry) * idy));
  c0 = clampi(c0, 0, P.hf_ncol - 2); r0 = clampi(r0, 0, P.hf_nrow - 2);
  c1 = clampi(c1, 0, P.hf_ncol - 2); r1 = clampi(r1, 0, P.hf_nrow - 2);
  c1 = sel(c1 > c0 + 2, c0 + 2, c1); r1 = sel(r1 > r0 + 2, r0 + 2, r1);
  const K x0 = cvt<K>(from_int(c0, F())) * PK.hf_dx - PK.hf_rx, y0 = cvt<K>(from_int(r0, F())) * PK.hf_dy - PK.hf_ry;
  const V3<F> c = mk<F>(cvt<F>(ck.x - x0), cvt<F>(ck.y - y0), cz);
  F dh[2], th[2];  // my two edges: distance (1: none) and index
  const I l = lane_id();
  for (int h = 0; h < 2; h++) {
    // my edge t = (j * 3 + i) * 3 + k of cell (c0 + i, r0 + j): k = 0: (c,r)-(c+1,r), 1: (c,r)-(c,r+1), 2: the strip diagonal (c,r+1)-(c+1,r)
    const I t = l + 16 * h;
    const I j = (t * 57) >> 9, rem = t - j * 9, i = (rem * 43) >> 7, k = rem - i * 3;
    const auto cell = inb & (t < 27) & ((c0 + i) <= c1) & ((r0 + j) <= r1);
    const I ia = i, ja = j + sel(k == 2, I(1), I(0)), ib = i + sel(k == 1, I(0), I(1)), jb = j + sel(k == 1, I(1), I(0));
    const V3<F> va = mk<F>(from_int(ia, F()) * P.hf_dx, from_int(ja, F()) * P.hf_dy, hf_at<F>(P, c0 + ia, r0 + ja));
    const V3<F> vb = mk<F>(from_int(ib, F()) * P.hf_dx, from_int(jb, F()) * P.hf_dy, hf_at<F>(P, c0 + ib, r0 + jb));
    V3<F> ed = vb - va;
    const F el = sqrt_(dot(ed, ed));
    ed = rcp_(el) * ed;
    const V3<F> mid = S(0.5) * (va + vb);
    F s_, t_;
    seg_seg_closest(c, az, H, mid, ed, S(0.5) * el, s_, t_);
    const V3<F> pa = c + s_ * az, pb = mid + t_ * ed;
    const V3<F> df = pa - pb;
    const F dist = sqrt_(dot(df, df)) - rad;
    const auto take = cell & (dist < near);
    dh[h] = sel(take, dist, F(S(1))); th[h] = sel(take, from_int(t, F()), F(S(99)));
  }
  // the deepest edge of the row (of equally deep ones the first in the order of the plain loop) -- and the runner-up
  const auto swap = dh[1] < dh[0];
  const F best = sel(swap, dh[1], dh[0]), bt = sel(swap, th[1], th[0]), worse = sel(swap, dh[0], dh[1]), wt = sel(swap, th[0], th[1]);
  const F dmin = -rowmax(-best);
  const auto cand = (best == dmin) & (best < near);
  const F tmin = -rowmax(-sel(cand, bt, F(S(99))));
  const auto mine = bt == tmin;   // (the lane that holds the winner offers its other edge as runner-up)
  const F dx = sel(mine, worse, best), tx = sel(mine, wt, bt);
  const F d2 = -rowmax(-dx);
  const F t2 = -rowmax(-sel((dx == d2) & (dx < near), tx, F(S(99))));
  // The search above is F and only NAMES the edge; whether the bar touches it, how deep, and where, is formed again in K for that one edge
  // (every lane the same): over the metre-sized cells an F distance carries 2e-7 m of rounding, which moved a bar across dist = 0 in one
  // env-step in a thousand (profiles/round3_accuracy_classified.log: most of the uneven model's contact flips were these).  Candidates
  // are therefore kept up to `near` above the surface, and the K distance decides.
  // WHICH edge, too, is decided in K when the two deepest are closer than F resolves (a bar lying across two ridges): round 4's identity
  // trace (word 24 + s of the sub-step trace) showed the remaining non-flip outliers of size 0.1 .. 1 on the heightfield to be exactly these
  // -- the same mask of active slots, another grid edge under the bar (profiles/round4_accuracy_classified_4096x16.log)
  const auto some = tmin < S(27);
  struct EdgeK { K dist, nn; V3<K> df, arm; I id; };
  const V3<K> cK = mk<K>(ck.x - x0, ck.y - y0, ck.z);
  auto evalK = [&](I tw) {
    EdgeK e;
    const I j = (tw * 57) >> 9, rem = tw - j * 9, i = (rem * 43) >> 7, k = rem - i * 3;
    const I ia = i, ja = j + sel(k == 2, I(1), I(0)), ib = i + sel(k == 1, I(0), I(1)), jb = j + sel(k == 1, I(1), I(0));
    e.id = ((c0 + i) * 128 + (r0 + j)) * 3 + k;  // WHICH grid edge: the k-th edge starting at vertex (col, row) (for the sub-step trace)
    const V3<K> vaK = mk<K>(cvt<K>(from_int(ia, F())) * PK.hf_dx, cvt<K>(from_int(ja, F())) * PK.hf_dy, cvt<K>(hf_at<F>(P, c0 + ia, r0 + ja)));
    const V3<K> vbK = mk<K>(cvt<K>(from_int(ib, F())) * PK.hf_dx, cvt<K>(from_int(jb, F())) * PK.hf_dy, cvt<K>(hf_at<F>(P, c0 + ib, r0 + jb)));
    V3<K> edK = vbK - vaK;
    const K elK = sqrt_(dot(edK, edK));
    edK = (K(SK(1)) / elK) * edK;
    const V3<K> midK = SK(0.5) * (vaK + vbK);
    K sK, tK;
    seg_seg_closest(cK, azk, PK.bar_half, midK, edK, SK(0.5) * elK, sK, tK);
    e.arm = sK * azk;
    e.df = (cK + e.arm) - (midK + tK * edK);
    e.nn = sqrt_(dot(e.df, e.df));
    e.dist = e.nn - PK.rcyl_r;
    return e;
  };
  EdgeK e = evalK(sel(some, to_int(tmin), I(0)));
  const auto close = some & (t2 < S(27)) & (d2 - dmin < S(2e-6));
  if (any_lane(close)) {
    const EdgeK e2 = evalK(sel(close, to_int(t2), I(0)));
    const auto second = close & ((e2.dist < e.dist) | ((e2.dist == e.dist) & (t2 < tmin)));
    e.dist = sel(second, e2.dist, e.dist); e.nn = sel(second, e2.nn, e.nn); e.df = sel3(second, e2.df, e.df); e.arm = sel3(second, e2.arm, e.arm);
    e.id = sel(second, e2.id, e.id);
  }
  id_out = e.id;
  const auto ok = some & (e.dist < SK(0)) & (e.nn > SK(1e-12));
  const F dist = cvt<F>(e.dist);
  n_out = sel3(ok, cvt3<F>((K(SK(1)) / max_(e.nn, K(SK(1e-30)))) * e.df), mk<F>(zero, zero, F(S(1))));
  off_out = sel3(ok, -(rad + S(0.5) * dist) * n_out, z3);  // from the point of the axis the contact sits under to the contact
  arm_out = sel3(ok, cvt3<F>(e.arm) + off_out, z3);
  return sel(ok, dist, F(S(1)));
}

// ------------------------------------------------------------------ multiply-add with a broadcast operand in ONE instruction
// acc + bcast<L>(x) * m is what the factorisations below are made of.  The compiler emits it as v_mov_b32_dpp + v_fma_f32: its DPP combiner
// folds row_newbcast into v_mul / v_add but not into v_fmac, and an fma with a negated operand is VOP3, which has no DPP form on this
// machine.  v_fmac_f32 with a DPP source 0 and a NEGATED source 1 exists (VOP2 DPP carries source modifiers), so the multiply-adds go
// out as inline assembly, several to a block.  A VALU write of a register needs two wait states before a DPP read of it, and the hazard
// recogniser does not look into inline assembly: every block opens with s_nop 1 (which covers whatever was written before it), and inside a
// block no DPP source is a register the block writes.  fp32 on the device only; everything else takes the plain form.
#if !defined(TGR_ROW_EMUL) && !defined(TGR_NO_DPP_ASM)
#define TGR_DPP_ASM 1
#define TGR_FM(acc, src, mul, lane) "v_fmac_f32_dpp " acc ", " src ", " mul " row_newbcast:" lane " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// a[k] += bcast<L>(a[k]) * m for up to five registers (column update of the Cholesky factorisation: m = -L_ij / L_jj)
template <int L> TGR_RDEV void fm_self(float m, float& a0) { asm("s_nop 1\n\t" TGR_FM("%0", "%0", "%1", "%2") : "+v"(a0) : "v"(m), "n"(L)); }
template <int L> TGR_RDEV void fm_self(float m, float& a0, float& a1) {
  asm("s_nop 1\n\t" TGR_FM("%0", "%0", "%2", "%3") TGR_FM("%1", "%1", "%2", "%3") : "+v"(a0), "+v"(a1) : "v"(m), "n"(L));
}
template <int L> TGR_RDEV void fm_self(float m, float& a0, float& a1, float& a2) {
  asm("s_nop 1\n\t" TGR_FM("%0", "%0", "%3", "%4") TGR_FM("%1", "%1", "%3", "%4") TGR_FM("%2", "%2", "%3", "%4") : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(m), "n"(L));
}
template <int L> TGR_RDEV void fm_self(float m, float& a0, float& a1, float& a2, float& a3) {
  asm("s_nop 1\n\t" TGR_FM("%0", "%0", "%4", "%5") TGR_FM("%1", "%1", "%4", "%5") TGR_FM("%2", "%2", "%4", "%5") TGR_FM("%3", "%3", "%4", "%5")
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "n"(L));
}
template <int L> TGR_RDEV void fm_self(float m, float& a0, float& a1, float& a2, float& a3, float& a4) {
  asm("s_nop 1\n\t" TGR_FM("%0", "%0", "%5", "%6") TGR_FM("%1", "%1", "%5", "%6") TGR_FM("%2", "%2", "%5", "%6") TGR_FM("%3", "%3", "%5", "%6") TGR_FM("%4", "%4", "%5", "%6")
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(m), "n"(L));
}
// ... for N registers a[0..N), in blocks of up to five
template <int L, int N> TGR_RDEV void fm_self_n(float m, float* a) {
  if constexpr (N >= 5) { fm_self<L>(m, a[0], a[1], a[2], a[3], a[4]); if constexpr (N > 5) fm_self_n<L, N - 5>(m, a + 5); }
  else if constexpr (N == 4) fm_self<L>(m, a[0], a[1], a[2], a[3]);
  else if constexpr (N == 3) fm_self<L>(m, a[0], a[1], a[2]);
  else if constexpr (N == 2) fm_self<L>(m, a[0], a[1]);
  else if constexpr (N == 1) fm_self<L>(m, a[0]);
}
// acc -= sum_{i < N} x[i] * bcast<L>(y[i])  (one accumulator, N <= 6 terms: a row of C - X Y^T, a step of a triangular solve)
template <int L, int N> TGR_RDEV void fm_dot(float& acc, const float* y, const float* x) {
  static_assert(N >= 1 && N <= 6, "fm_dot");
  if constexpr (N == 1) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%2", "%3") : "+v"(acc) : "v"(y[0]), "v"(x[0]), "n"(L));
  else if constexpr (N == 2) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%3", "%5") TGR_FM("%0", "%2", "-%4", "%5") : "+v"(acc) : "v"(y[0]), "v"(y[1]), "v"(x[0]), "v"(x[1]), "n"(L));
  else if constexpr (N == 3) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%4", "%7") TGR_FM("%0", "%2", "-%5", "%7") TGR_FM("%0", "%3", "-%6", "%7")
                                 : "+v"(acc) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "n"(L));
  else if constexpr (N == 4) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%5", "%9") TGR_FM("%0", "%2", "-%6", "%9") TGR_FM("%0", "%3", "-%7", "%9") TGR_FM("%0", "%4", "-%8", "%9")
                                 : "+v"(acc) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "n"(L));
  else if constexpr (N == 5) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%6", "%11") TGR_FM("%0", "%2", "-%7", "%11") TGR_FM("%0", "%3", "-%8", "%11") TGR_FM("%0", "%4", "-%9", "%11")
                                 TGR_FM("%0", "%5", "-%10", "%11")
                                 : "+v"(acc) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "n"(L));
  else asm("s_nop 1\n\t" TGR_FM("%0", "%1", "-%7", "%13") TGR_FM("%0", "%2", "-%8", "%13") TGR_FM("%0", "%3", "-%9", "%13") TGR_FM("%0", "%4", "-%10", "%13")
           TGR_FM("%0", "%5", "-%11", "%13") TGR_FM("%0", "%6", "-%12", "%13")
           : "+v"(acc) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "n"(L));
}
// two such accumulators side by side (their chains interleave): acc0 -= sum x[i] bcast<L0>(y[i]), acc1 -= sum x[i] bcast<L1>(y[i]), six terms each
template <int L0, int L1> TGR_RDEV void fm_dot6x2(float& acc0, float& acc1, const float* y, const float* x) {
  asm("s_nop 1\n\t" TGR_FM("%0", "%2", "-%8", "%14") TGR_FM("%1", "%2", "-%8", "%15") TGR_FM("%0", "%3", "-%9", "%14") TGR_FM("%1", "%3", "-%9", "%15")
      TGR_FM("%0", "%4", "-%10", "%14") TGR_FM("%1", "%4", "-%10", "%15") TGR_FM("%0", "%5", "-%11", "%14") TGR_FM("%1", "%5", "-%11", "%15")
      TGR_FM("%0", "%6", "-%12", "%14") TGR_FM("%1", "%6", "-%12", "%15") TGR_FM("%0", "%7", "-%13", "%14") TGR_FM("%1", "%7", "-%13", "%15")
      : "+v"(acc0), "+v"(acc1)
      : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "n"(L0), "n"(L1));
}
// acc[j] += bcast<L>(src[j]) * m for j < N <= 5 (independent accumulators, one shared multiplier: a column of the replicated back-substitution)
template <int L, int N> TGR_RDEV void fm_cols(float* acc, const float* src, float m) {
  static_assert(N >= 1 && N <= 5, "fm_cols");
  if constexpr (N == 1) asm("s_nop 1\n\t" TGR_FM("%0", "%1", "%2", "%3") : "+v"(acc[0]) : "v"(src[0]), "v"(m), "n"(L));
  else if constexpr (N == 2) asm("s_nop 1\n\t" TGR_FM("%0", "%2", "%4", "%5") TGR_FM("%1", "%3", "%4", "%5") : "+v"(acc[0]), "+v"(acc[1]) : "v"(src[0]), "v"(src[1]), "v"(m), "n"(L));
  else if constexpr (N == 3) asm("s_nop 1\n\t" TGR_FM("%0", "%3", "%6", "%7") TGR_FM("%1", "%4", "%6", "%7") TGR_FM("%2", "%5", "%6", "%7")
                                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(m), "n"(L));
  else if constexpr (N == 4) asm("s_nop 1\n\t" TGR_FM("%0", "%4", "%8", "%9") TGR_FM("%1", "%5", "%8", "%9") TGR_FM("%2", "%6", "%8", "%9") TGR_FM("%3", "%7", "%8", "%9")
                                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(m), "n"(L));
  else asm("s_nop 1\n\t" TGR_FM("%0", "%5", "%10", "%11") TGR_FM("%1", "%6", "%10", "%11") TGR_FM("%2", "%7", "%10", "%11") TGR_FM("%3", "%8", "%10", "%11") TGR_FM("%4", "%9", "%10", "%11")
           : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(m), "n"(L));
}
template <int L, int N> TGR_RDEV void fm_cols_n(float* acc, const float* src, float m) {
  if constexpr (N > 5) { fm_cols<L, 5>(acc, src, m); fm_cols_n<L, N - 5>(acc + 5, src + 5, m); }
  else if constexpr (N >= 1) fm_cols<L, N>(acc, src, m);
}
#else
#define TGR_DPP_ASM 0
#endif

// ------------------------------------------------------------------ row-distributed dense algebra (lane i holds row i)
// In-place Cholesky of an N x N SPD matrix whose row i lives in A[0..N) of lane i (full rows, both triangles).  Afterwards the
// lower part of the rows holds L (A[j], j < i) and il[j] = 1 / L_jj is replicated.  Pivots are floored: a Schur complement that loses
// positivity to rounding must cost accuracy of one Newton direction, never produce inf.  The floor of pivot j is a bound the CALLER knows
// (floor_of(j), a uniform scalar): every matrix factored here is "something positive definite plus positive semi-definite terms" -- the mass
// matrix plus contact terms, the identity plus the tendon dampers -- and a Cholesky pivot of A + B, B >= 0, is never below the same pivot of
// A.  (Until round 4 the floor was 1e-6 of the original diagonal entry: six broadcasts, six multiplies and six maxima per factorisation for a
// test that the known bound does in one maximum per pivot.)
template <int N, typename F, typename FL> TGR_RDEV void rchol(F* A, F* il, FL floor_of) {
  typedef typename LaneTraits<F>::S S;
  static_for<N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    F d;
#if TGR_DPP_ASM
    if constexpr (std::is_same<F, float>::value) {
      // broadcast and floor in ONE instruction (v_max_f32 with a DPP source; the floor has to sit in a VGPR).  From the compiler the pair comes
      // as three: the broadcast travels through an integer bit cast, after which fmaxf() canonicalises its operand first (a v_max of it with itself)
      const float fl = floor_of(j);
      asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(A[j]), "v"(fl), "n"(j));
    } else
#endif
    d = max_(bcast<j>(A[j]), F(S(floor_of(j))));
    const F ilj = rsqrt_(d);
    il[j] = ilj;
    const F t = A[j] * (ilj * ilj);  // L_ij / L_jj
#if TGR_DPP_ASM
    if constexpr (std::is_same<F, float>::value) {
      fm_self_n<j, N - 1 - j>(-t, A + j + 1);
    } else
#endif
    static_for<N - 1 - j>([&](auto kc) {
      constexpr int k = j + 1 + decltype(kc)::value;
      A[k] = A[k] - t * bcast<j>(A[k]);
    });
    A[j] = A[j] * ilj;  // L_ij (lane j: L_jj)
  });
}
// X <- X L^-T for the row X of every lane (L: row-distributed factor of rchol, N x N): solves X_new L^T = X
template <int N, typename F> TGR_RDEV void rtrsm(const F* Lr, const F* il, F* X) {
  static_for<N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    F s = X[j];
#if TGR_DPP_ASM
    if constexpr (N == 6 && std::is_same<F, float>::value) {
      if constexpr (j > 0) fm_dot<j, j>(s, Lr, X);
    } else
#endif
    static_for<j>([&](auto kc) { constexpr int k = decltype(kc)::value; s = s - X[k] * bcast<j>(Lr[k]); });
    X[j] = s * il[j];
  });
}
// C <- C - X Y^T  (row i of C: C[k] -= sum_m X_i[m] Y_k[m]); rows of X, Y, C distributed, N columns
template <int N, typename F> TGR_RDEV void rsyrk(F* C, const F* X, const F* Y) {
#if TGR_DPP_ASM
  if constexpr (N == 6 && std::is_same<F, float>::value) {
    fm_dot6x2<0, 1>(C[0], C[1], Y, X); fm_dot6x2<2, 3>(C[2], C[3], Y, X); fm_dot6x2<4, 5>(C[4], C[5], Y, X);
    return;
  }
#endif
  static_for<N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    F s = C[k];
    static_for<N>([&](auto mc) { constexpr int m = decltype(mc)::value; s = s - X[m] * bcast<k>(Y[m]); });
    C[k] = s;
  });
}
// forward substitution L x = y: y distributed (lane i holds y_i; destroyed), the solution comes out REPLICATED in xr[N].
// (lanes r <= j hold rounding dust in Lr[j] after rchol; their y is no longer read once x_r has been taken)
template <int N, typename F> TGR_RDEV void rfwd(const F* Lr, const F* il, F& y, F* xr) {
  static_for<N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const F xj = bcast<j>(y) * il[j];
    xr[j] = xj;
    y = y - Lr[j] * xj;
  });
}
// y <- y - X z  with z replicated (row i: y_i -= sum_m X_i[m] z[m])
template <int N, typename F> TGR_RDEV void rgemv_rep(const F* X, const F* z, F& y) {
  static_for<N>([&](auto mc) { constexpr int m = decltype(mc)::value; y = y - X[m] * z[m]; });
}
// y <- y - X z  with z distributed over the N lanes (row i: y_i -= sum_m X_i[m] z_m)
template <int N, typename F> TGR_RDEV void rgemv(const F* X, F z, F& y) {
  static_for<N>([&](auto mc) { constexpr int m = decltype(mc)::value; y = y - X[m] * bcast<m>(z); });
}
// backward substitution: solves L^T x = y for y given REPLICATED (yr[N]); returns x distributed (xd, lane i holds x_i, other
// lanes zero) and replicated (xr[N])
template <int N, typename F> TGR_RDEV void rbwd(const F* Lr, const F* il, const F* yr, F& xd, F* xr) {
  typedef typename LaneTraits<F>::S S;
  const auto r = lane_id();
  xd = F(S(0));
  static_for<N>([&](auto ic) {
    constexpr int i = N - 1 - decltype(ic)::value;
    const F s = rowsum(Lr[i] * xd);  // lanes k > i hold their final x_k, the others zero
    const F xi = (yr[i] - s) * il[i];
    xr[i] = xi;
    xd = sel(r == i, xi, xd);
  });
}
// the same solve with everything replicated: after x_i is known, column i of L^T -- row i of L, which lane i holds -- is broadcast and
// taken out of the right-hand sides above it.  No row sums: the dependent chain is one multiply and one fma per column (the row-sum form
// chains four DPP adds per column), and the broadcasts do not wait for x.  yr is consumed.
template <int N, typename F> TGR_RDEV void rbwd_rep(const F* Lr, const F* il, F* yr, F* xr) {
  static_for<N>([&](auto ic) {
    constexpr int i = N - 1 - decltype(ic)::value;
    const F xi = yr[i] * il[i];
    xr[i] = xi;
#if TGR_DPP_ASM
    if constexpr (std::is_same<F, float>::value) {
      fm_cols_n<i, i>(yr, Lr, -xi);
    } else
#endif
    static_for<i>([&](auto jc) { constexpr int j = decltype(jc)::value; yr[j] = yr[j] - bcast<i>(Lr[j]) * xi; });
  });
}
// lane i <- xr[i] (i < N), the other lanes zero
template <int N, typename F> TGR_RDEV F rscatter(const F* xr) {
  typedef typename LaneTraits<F>::S S;
  const auto r = lane_id();
  F xd = F(S(0));
  static_for<N>([&](auto ic) { constexpr int i = decltype(ic)::value; xd = sel(r == i, xr[i], xd); });
  return xd;
}
template <int N, typename F> TGR_RDEV void rgather(F yd, F* yr) {
  static_for<N>([&](auto ic) { constexpr int i = decltype(ic)::value; yr[i] = bcast<i>(yd); });
}
// t_i = sum_k X_k[i] z_k  (X^T z) for i < N, replicated results
template <int N, typename F> TGR_RDEV void rgemvT(const F* X, F zd, F* out) {
  static_for<N>([&](auto ic) { constexpr int i = decltype(ic)::value; out[i] = rowsum(X[i] * zd); });
}

// ------------------------------------------------------------------ line search pieces (slot role)
template <typename F> struct RLSlot { F A0, A1, A2, B0, B1, B2, jn0, qn, jo0, qo, Dn; typename LaneTraits<F>::B act; };
template <typename F> TGR_RDEV RLSlot<F> ls_make(const RContact<F>& c, V3<F> jl, V3<F> ja, V3<F> ql, V3<F> qa) {
  RLSlot<F> s;
  s.A0 = dot(jl, jl); s.A1 = dot(jl, ql); s.A2 = dot(ql, ql);
  s.B0 = dot(ja, ja); s.B1 = dot(ja, qa); s.B2 = dot(qa, qa);
  s.jn0 = dot(c.n, jl); s.qn = dot(c.n, ql); s.jo0 = dot(c.n, ja); s.qo = dot(c.n, qa);
  s.Dn = c.Dn; s.act = c.active;
  return s;
}
template <typename F, typename S> TGR_RDEV void ls_slot(const RParams<S>& P, const RLSlot<F>& s, F alpha, F& d1, F& d2) {
  const S mu = P.mu;
  F Dtf = s.Dn * P.dtf;
  F jn = s.jn0 + alpha * s.qn, jo = s.jo0 + alpha * s.qo;
  F jl2 = s.A0 + alpha * (S(2) * s.A1 + alpha * s.A2), ja2 = s.B0 + alpha * (S(2) * s.B1 + alpha * s.B2);
  F jlq = s.A1 + alpha * s.A2, jaq = s.B1 + alpha * s.B2;
  F tt = P.f1sq * (jl2 - jn * jn) + P.f3sq * jo * jo + P.f4sq * (ja2 - jo * jo);
  const F rT = rsqrt_(max_(tt, F(S(1e-36))));
  F Tn = max_(tt, F(S(0))) * rT;
  F wq = P.f1sq * (jlq - jn * s.qn) + P.f3sq * jo * s.qo + P.f4sq * (jaq - jo * s.qo);
  F qtt = P.f1sq * (s.A2 - s.qn * s.qn) + P.f3sq * s.qo * s.qo + P.f4sq * (s.B2 - s.qo * s.qo);
  F N = mu * jn;
  auto top = (N >= mu * Tn) | !s.act;
  auto bottom = !top & (mu * N + Tn <= S(0));
  F b1 = s.Dn * jn * s.qn + Dtf * wq;
  F b2 = s.Dn * s.qn * s.qn + Dtf * qtt;
  F Dm = s.Dn * P.dmf;
  F NT = N - mu * Tn;
  F gq = mu * (s.qn - wq * rT);
  F kap = -Dm * mu * NT * rT;
  F m1 = Dm * NT * gq;
  F m2 = Dm * gq * gq + kap * (qtt - wq * wq * rT * rT);
  const F zero = F(S(0));
  d1 = sel(top, zero, sel(bottom, b1, m1));
  d2 = sel(top, zero, sel(bottom, b2, m2));
}

// ------------------------------------------------------------------ Newton Hessian, dof role
// per-lane dof constants: e = unit vector of my component, ang = am I an angular row
template <typename F> struct DofC { V3<F> e; typename LaneTraits<F>::B ang, dof, upper; };  // upper: lanes 8.. (the dof constants repeat there)
// column r of J = [[I, -[arm]x], [0, I]] for my dof r:  (u_lin, u_ang)
template <typename F> TGR_RDEV void dof_col(const DofC<F>& d, V3<F> arm, V3<F>& ul, V3<F>& ua) {
  typedef typename LaneTraits<F>::S S;
  const V3<F> z = mk<F>(F(S(0)), F(S(0)), F(S(0)));
  ul = sel3(d.ang, cross(d.e, arm), d.e);
  ua = sel3(d.ang, d.e, z);
}
// slot data after the broadcast from its slot lane (replicated in the row)
template <typename F> struct SlotB { V3<F> n, rA, rB; F a1, a2, b1, b2, cg, cw; V3<F> gl, ga, wl, wa; };
// w = C u  for the cone Hessian C of the broadcast slot
template <typename F> TGR_RDEV void cone_mul(const SlotB<F>& s, bool cone, V3<F> ul, V3<F> ua, V3<F>& wl, V3<F>& wa) {
  wl = s.a1 * ul + (s.a2 * dot(s.n, ul)) * s.n;
  wa = s.b1 * ua + (s.b2 * dot(s.n, ua)) * s.n;
  if (cone) {
    F gu = s.cg * (dot(s.gl, ul) + dot(s.ga, ua)), wu = s.cw * (dot(s.wl, ul) + dot(s.wa, ua));
    wl = wl + gu * s.gl + wu * s.wl;
    wa = wa + gu * s.ga + wu * s.wa;
  }
}
// row += sign * (wl, arm x wl + wa)   (my row of J_row^T C J_col, J_col built on `arm`)
template <typename F> TGR_RDEV void row_acc(F* R, V3<F> wl, V3<F> wa, V3<F> arm, bool neg) {
  V3<F> t = cross(arm, wl) + wa;
  if (neg) { R[0] = R[0] - wl.x; R[1] = R[1] - wl.y; R[2] = R[2] - wl.z; R[3] = R[3] - t.x; R[4] = R[4] - t.y; R[5] = R[5] - t.z; }
  else { R[0] = R[0] + wl.x; R[1] = R[1] + wl.y; R[2] = R[2] + wl.z; R[3] = R[3] + t.x; R[4] = R[4] + t.y; R[5] = R[5] + t.z; }
}
// the 18x18 Hessian as 6 blocks, row r of each in lane r:  [D0 ; O10 D1 ; O20 O21 D2]
template <typename F> struct RHess { F D[3][6]; F O10[6], O20[6], O21[6]; };
template <typename F> TGR_RDEV F* hess_off(RHess<F>& H, int a, int b) { return (a == 0 && b == 1) ? H.O10 : ((a == 0 && b == 2) ? H.O20 : H.O21); }

// am / cm: the wavefront's ballots of (slot active) and (slot in the middle zone of its cone), taken once per iteration
template <int MODEL, int PAGE, int C, typename F>
TGR_RDEV void hess_add_slot(RHess<F>& H, const DofC<F>& d, const RContact<F>& c, const RCone<F>& o, BMask am, BMask cm) {
  typedef typename LaneTraits<F>::S S;
  if (!slot_in<C>(am)) return;
  SlotB<F> s;
  s.n = bcast3<C>(c.n); s.rA = bcast3<C>(c.rA); s.rB = bcast3<C>(c.rB);
  s.a1 = bcast<C>(o.a1); s.a2 = bcast<C>(o.a2); s.b1 = bcast<C>(o.b1); s.b2 = bcast<C>(o.b2);
  const bool cone = slot_in<C>(cm);
  if (cone) { s.cg = bcast<C>(o.cg); s.cw = bcast<C>(o.cw); s.gl = bcast3<C>(o.gl); s.ga = bcast3<C>(o.ga); s.wl = bcast3<C>(o.wl); s.wa = bcast3<C>(o.wa); }
  constexpr int A = slot_bar_a<MODEL>(PAGE, C), Bb = slot_bar_b<MODEL>(PAGE, C);
  static_assert(Bb >= 0, "slot without a body");
  V3<F> ul, ua, wl, wa;
  if constexpr (A < 0) {
    dof_col(d, s.rB, ul, ua);
    cone_mul(s, cone, ul, ua, wl, wa);
    row_acc(H.D[Bb], wl, wa, s.rB, false);
  } else {
    // A slot between two bars: the lower half of the row (lanes 0..5) works on body B's columns, the upper half (lanes 8..13, where the dof
    // constants d repeat) on body A's -- the two differ only in the lever the Jacobian column is built on -- and one shift brings A's diagonal
    // block down.  (One by one the two bodies cost a second column, a second cone product and a third accumulation: 122 operations, 84 so.
    // The bar pairs' side-to-side slots are in use in 40 % of the wavefronts of the headline workload each: profiles/round4_slot_census.log.)
    const V3<F> arm = sel3(d.upper, s.rA, s.rB);
    dof_col(d, arm, ul, ua);
    cone_mul(s, cone, ul, ua, wl, wa);
    const V3<F> t = cross(arm, wl) + wa;
    F* RB = H.D[Bb]; F* RA = H.D[A];
    RB[0] = RB[0] + wl.x; RB[1] = RB[1] + wl.y; RB[2] = RB[2] + wl.z; RB[3] = RB[3] + t.x; RB[4] = RB[4] + t.y; RB[5] = RB[5] + t.z;  // (lower half: B's block)
    RA[0] = RA[0] + upper8(wl.x); RA[1] = RA[1] + upper8(wl.y); RA[2] = RA[2] + upper8(wl.z);
    RA[3] = RA[3] + upper8(t.x); RA[4] = RA[4] + upper8(t.y); RA[5] = RA[5] + upper8(t.z);
    row_acc(hess_off(H, A, Bb), wl, wa, s.rA, true);  // rows: B dofs, columns: A dofs (lower half)
  }
}

// a POOL lane's slot: its bar pair is a run-time value of the environment (plC: lane C's pl, broadcast), so each of the three pairs is handled
// under a mask (environments whose slot serves another pair add zero), and a pair nobody in the wavefront routes to this lane is skipped
template <typename F> TGR_RDEV void row_acc_masked(F* R, V3<F> wl, V3<F> wa, V3<F> arm, F m) {  // m: +1, -1 or 0
  V3<F> t = cross(arm, wl) + wa;
  R[0] = R[0] + m * wl.x; R[1] = R[1] + m * wl.y; R[2] = R[2] + m * wl.z; R[3] = R[3] + m * t.x; R[4] = R[4] + m * t.y; R[5] = R[5] + m * t.z;
}
template <int C, typename F, typename I>
TGR_RDEV void hess_add_pool(RHess<F>& H, const DofC<F>& d, const RContact<F>& c, const RCone<F>& o, I pl, BMask am, BMask cm) {
  typedef typename LaneTraits<F>::S S;
  if (!slot_in<C>(am)) return;
  const I plC = bcast<C>(pl);
  SlotB<F> s;
  s.n = bcast3<C>(c.n); s.rA = bcast3<C>(c.rA); s.rB = bcast3<C>(c.rB);
  s.a1 = bcast<C>(o.a1); s.a2 = bcast<C>(o.a2); s.b1 = bcast<C>(o.b1); s.b2 = bcast<C>(o.b2);
  const bool cone = slot_in<C>(cm);
  if (cone) { s.cg = bcast<C>(o.cg); s.cw = bcast<C>(o.cw); s.gl = bcast3<C>(o.gl); s.ga = bcast3<C>(o.ga); s.wl = bcast3<C>(o.wl); s.wa = bcast3<C>(o.wa); }
  V3<F> ulB, uaB, wlB, waB, ulA, uaA, wlA, waA;
  dof_col(d, s.rB, ulB, uaB);
  cone_mul(s, cone, ulB, uaB, wlB, waB);
  dof_col(d, s.rA, ulA, uaA);
  cone_mul(s, cone, ulA, uaA, wlA, waA);
  static_for<3>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int A = pair_a(p), Bb = pair_b(p);
    const auto mine = plC == p;
    if (!any_lane(mine)) return;
    const F m = sel(mine, F(S(1)), F(S(0)));
    row_acc_masked(H.D[Bb], wlB, waB, s.rB, m);
    row_acc_masked(hess_off(H, A, Bb), wlB, waB, s.rA, -m);  // rows: B dofs, columns: A dofs
    row_acc_masked(H.D[A], wlA, waA, s.rA, m);
  });
}

// p = -H^-1 g.  g: per bar distributed (lane r: component r of the bar's 6-vector); returns p replicated, pr[b][0..6)
// c01 / c02 / c12: does ANY environment of the wavefront couple bars (0,1) / (0,2) / (1,2)?  (uniform flags; an off-diagonal block
// nobody fills is zero and its part of the factorisation is skipped.  O21 also fills in through O20 O10^T.)
// fl_lin / fl_ang: floors of the pivots of a bar's linear / angular rows (half of what the mass matrix alone would give: m, Izz)
template <typename F, typename S_> TGR_RDEV void newton_direction(RHess<F>& H, const F* gd, F pr[3][6], F& gHg, bool c01, bool c02, bool c12, S_ fl_lin, S_ fl_ang) {
  typedef typename LaneTraits<F>::S S;
  auto floor_of = [&](int j) { return j < 3 ? fl_lin : fl_ang; };
  const bool n21 = c12 || (c01 && c02);
  F il0[6], il1[6], il2[6];
  rchol<6>(H.D[0], il0, floor_of);
  if (c01) rtrsm<6>(H.D[0], il0, H.O10);
  if (c02) rtrsm<6>(H.D[0], il0, H.O20);
  if (c01) rsyrk<6>(H.D[1], H.O10, H.O10);
  if (c01 && c02) rsyrk<6>(H.O21, H.O20, H.O10);
  rchol<6>(H.D[1], il1, floor_of);
  if (n21) rtrsm<6>(H.D[1], il1, H.O21);
  if (c02) rsyrk<6>(H.D[2], H.O20, H.O20);
  if (n21) rsyrk<6>(H.D[2], H.O21, H.O21);
  rchol<6>(H.D[2], il2, floor_of);
  // forward: L y = -g  (the y of each block comes out replicated; the sign goes in here, three negations, instead of onto the eighteen
  // components of the solution)
  F y0 = -gd[0], y1 = -gd[1], y2 = -gd[2], yr0[6], yr1[6], yr2[6];
  rfwd<6>(H.D[0], il0, y0, yr0);
  if (c01) rgemv_rep<6>(H.O10, yr0, y1);
  if (c02) rgemv_rep<6>(H.O20, yr0, y2);
  rfwd<6>(H.D[1], il1, y1, yr1);
  if (n21) rgemv_rep<6>(H.O21, yr1, y2);
  rfwd<6>(H.D[2], il2, y2, yr2);
  // g^T H^-1 g = |L^-1 g|^2: the Newton decrement falls out of the forward solve (replicated, no row sum)
  gHg = F(S(0));
  for (int i = 0; i < 6; i++) gHg = gHg + yr0[i] * yr0[i] + yr1[i] * yr1[i] + yr2[i] * yr2[i];
  // backward: L^T x = y, replicated (rbwd_rep); a coupled block needs the solution above it spread over the lanes once
  F t[6];
  rbwd_rep<6>(H.D[2], il2, yr2, pr[2]);
  const F x2 = (n21 || c02) ? rscatter<6>(pr[2]) : F(S(0));
  if (n21) {
    rgemvT<6>(H.O21, x2, t);
    for (int i = 0; i < 6; i++) yr1[i] = yr1[i] - t[i];
  }
  rbwd_rep<6>(H.D[1], il1, yr1, pr[1]);
  if (c01) {
    const F x1 = rscatter<6>(pr[1]);
    rgemvT<6>(H.O10, x1, t);
    for (int i = 0; i < 6; i++) yr0[i] = yr0[i] - t[i];
  }
  if (c02) {
    rgemvT<6>(H.O20, x2, t);
    for (int i = 0; i < 6; i++) yr0[i] = yr0[i] - t[i];
  }
  rbwd_rep<6>(H.D[0], il0, yr0, pr[0]);
}

// constraint wrench on bar b.  The caps of bar b sit on lanes 2b, 2b+1: PfB / PtB / PkB are their neighbour sums (pairsum); the side-to-side
// slot of pair (A, B) = (0,1), (0,2), (1,2) on lane 6, 8, 10 pushes bar B with +f and bar A with -f; MODEL 1: bar b's cylinder slot on lane
// 12 + b; a pool lane in use (am: the wavefront's ballot of active slots) serves the pair its pl names.  (An idle slot carries zero force.)
template <int L, typename F> TGR_RDEV V3<F> bcast3v(V3<F> a) { return mk<F>(bcast<L>(a.x), bcast<L>(a.y), bcast<L>(a.z)); }
template <int MODEL, typename K, typename F, typename I>
TGR_RDEV void bar_sum(int b, V3<F> PfB, V3<F> PtB, K PkB, V3<F> fB, V3<F> tB, V3<F> tA, K tauB, K tauA, BMask am, I pl, V3<F>& f, V3<F>& t, K& tax) {
  if (b == 0) {
    f = (bcast3v<0>(PfB) - bcast3v<6>(fB)) - bcast3v<8>(fB);
    t = (bcast3v<0>(PtB) - bcast3v<6>(tA)) - bcast3v<8>(tA);
    tax = (bcast<0>(PkB) - bcast<6>(tauA)) - bcast<8>(tauA);
    if (MODEL) { f = f + bcast3v<12>(fB); t = t + bcast3v<12>(tB); tax = tax + bcast<12>(tauB); }
  } else if (b == 1) {
    f = (bcast3v<2>(PfB) + bcast3v<6>(fB)) - bcast3v<10>(fB);
    t = (bcast3v<2>(PtB) + bcast3v<6>(tB)) - bcast3v<10>(tA);
    tax = (bcast<2>(PkB) + bcast<6>(tauB)) - bcast<10>(tauA);
    if (MODEL) { f = f + bcast3v<13>(fB); t = t + bcast3v<13>(tB); tax = tax + bcast<13>(tauB); }
  } else {
    f = (bcast3v<4>(PfB) + bcast3v<8>(fB)) + bcast3v<10>(fB);
    t = (bcast3v<4>(PtB) + bcast3v<8>(tB)) + bcast3v<10>(tB);
    tax = (bcast<4>(PkB) + bcast<8>(tauB)) + bcast<10>(tauB);
    if (MODEL) { f = f + bcast3v<14>(fB); t = t + bcast3v<14>(tB); tax = tax + bcast<14>(tauB); }
  }
  if (slots_in<pool_mask<MODEL>()>(am)) {
    typedef typename LaneTraits<F>::S S;
    typedef typename LaneTraits<K>::S SK;
    const V3<F> z3 = mk<F>(F(S(0)), F(S(0)), F(S(0)));
    auto add = [&](auto lc) {
      constexpr int L = decltype(lc)::value;
      if (!slot_in<L>(am)) return;
      // pair plL = (A, B): bar A gets -f (torque about A's centre: tA), bar B gets +f.  (0,1), (0,2), (1,2)
      const I plL = bcast<L>(pl);
      const auto asA = b == 0 ? ((plL == 0) | (plL == 1)) : (b == 1 ? (plL == 2) : (plL == 99));
      const auto asB = b == 0 ? (plL == 99) : (b == 1 ? (plL == 0) : ((plL == 1) | (plL == 2)));
      f = (f + sel3(asB, bcast3v<L>(fB), z3)) - sel3(asA, bcast3v<L>(fB), z3);
      t = (t + sel3(asB, bcast3v<L>(tB), z3)) - sel3(asA, bcast3v<L>(tA), z3);
      tax = (tax + sel(asB, bcast<L>(tauB), K(SK(0)))) - sel(asA, bcast<L>(tauA), K(SK(0)));
    };
    add(std::integral_constant<int, 7>{}); add(std::integral_constant<int, 9>{}); add(std::integral_constant<int, 11>{});
    if (!MODEL) { add(std::integral_constant<int, 12>{}); add(std::integral_constant<int, 13>{}); add(std::integral_constant<int, 14>{}); }
    add(std::integral_constant<int, 15>{});
  }
}

// ------------------------------------------------------------------ the sub-step
// Between its uses the environment's state lives in slabs (LDS on the GPU), not in registers:
//   kd (K, replicated): pose x, q; unconstrained acceleration l0, a0; bar axes ez       fd (F, replicated): v, w, ctrl, act
//   td (F, per lane):   tendon-role results u, ra x u, rb x u, damping                  rd (K, per lane, per block): tendon-role constants
constexpr int KX = 0, KQ = 9, KL0 = 21, KA0 = 30, KEZ = 39, KTAX = 48, KACT = 51, K_SLAB = 57;
constexpr int FV = 0, FW = 9, FCTRL = 18, FDL = 30, FDA = 39, FEZ = 48, FFCL = 57, FFCA = 66, FCAPD = 75, FG = 76, F_SLAB = 94;  // FDL / FDA: d = a - a0; FCAPD: cap radius minus the model's
// FG: the gradient, bar by bar (6 each) -- lane b forms bar b's, the dof lanes read it back transposed (lane r: component r of every bar)
// MODEL 1 iterates on e = d - d_ref (see substep): d_ref (FRL / FRA) and I d_ref (FGA) sit behind the common part of the slab
constexpr int FRL = 94, FRA = 103, FGA = 112;
template <int MODEL> constexpr int f_slab() { return MODEL ? 121 : F_SLAB; }
constexpr int TU = 0, TMA = 3, TMB = 6, TDAMP = 9, T_SLAB = 10;
// cd (F, per lane): my contact slot.  (The iterate of my slot's two bars is NOT kept here: a lane reads it from the replicated iterate in fd
// with its own bar numbers as index -- copies per slot cost twelve loads, fmas and stores per lane and iteration to keep up to date.)
constexpr int CN = 0, CRA = 3, CRB = 6, CC0 = 9, CDN = 10, CACT = 11, CJL0 = 12, CJA0 = 15, C_SLAB = 18;
constexpr int AMB = 0, AMA = 3, AZB = 6, AZA = 9, A_SLAB = 12;  // ad (K, per lane): my slot's axial levers z x r and axes z of its two bars
constexpr int RSA = 0, RSB = 3, RTK = 6, RTB = 7, RTL0 = 8, R_SLAB = 9;
// hd (F, replicated): the solver's history -- the iterate d the sub-step before last ended on (18) and whether the last two sub-steps
// had contacts (2).  On the GPU it lives in columns 9 and 10 of the tendon slab (lanes 9.. own no tendon): hidx() is its addressing.
constexpr int HD = 0, HPC = 18, HPPC = 19, H_SLAB = 20;
// lp (F, replicated): the lagged pose of the last sub-step, x (9), q (12), and its tendon lengths (9) -- columns 11..13 of the tendon slab.
constexpr int LPX = 0, LPQ = 9, LPT = 21, LP_SLAB = 30;
// lc (K, replicated): the lagged centre of the three bars, (x0 + x1 + x2) / 3 -- what the reference calls xy_position (tensegrity_env.py:293-294,
// 311-315).  The lagged bar positions in lp are kept RELATIVE to it: everything the env layer derives from them (headings, cap positions
// about the centroid) is a difference of positions, and the one absolute quantity -- the centre itself, whose change per control period is
// the x / y velocity the health test and the reward read -- stays in K.  A robot 30 m from the origin resolves its fp32 coordinates to
// 2e-6 m, i.e. 1e-4 m/s of velocity: the size of the is_healthy threshold (tensegrity_env.py:279).  On the GPU: columns 14, 15 of the tendon slab.
constexpr int LCX = 0, LC_SLAB = 3;
// xs (F, replicated): small per-environment scalars (none at present: the pool lanes' pairs live in a register).  On
// the GPU: rows 3.. of column 14 of the tendon slab.
constexpr int XS_SLAB = 1;
#ifdef TGR_ROW_EMUL
constexpr int hidx(int i) { return i; }
constexpr int lidx(int i) { return i; }
#else
constexpr int hidx(int i) { return (i % 10) * 16 + i / 10; }
constexpr int lidx(int i) { return (i % 10) * 16 + i / 10; }
#endif
// dbg: optional sub-step trace of this environment (null in production; tgr_bind_substep_trace): word s = contact slots active in sub-step s
// (bits 0..15 page 0, 16..21 page 1) | (Newton iterations the environment itself needed so far, mod 256) << 22 | (left at the cap or stalled) << 30 | (stalled) << 31
// words TRACE_SUB + s (MODEL 1): which facets / grid edges the terrain contacts of sub-step s touch (see the contact stage)
constexpr int TRACE_SUB = 24, TRACE_WORDS = 48;
template <typename K, typename F> struct RMem { Slab<K> kd; Slab<F> fd; Slab<F> td; Slab<K> rd; Slab<K> ad; Slab<F> cd; Slab<F> hd; Slab<F> lp; Slab<K> lc; Slab<F> xs; unsigned* dbg; };

#ifndef TGR_IMPLICIT_K
#define TGR_IMPLICIT_K 1
#endif
// `LAST`: what only the last sub-step of a control period records (lagged kinematics, tendon lengths, contact wrench).
// K / PK: lane and scalar type of the pose, the kinematics, the tendon geometry and the smooth forces; F / P: of the rest.
// argPK / argP: where the model constants (K and F copies) sit in the kernel's argument block.  TGR_REARG() re-reads them through a fresh
// pointer at the head of the stages OUTSIDE the Newton loop (and once before it), so that a stage holds only its own constants in SGPRs:
// 0 spilled SGPRs (196 before: parked in VGPR lanes, fetched back with v_readlane -- 5 % of the instruction stream), 0 spilled VGPRs, 238
// VGPRs.  Worth +1 % (profiles/round4_ab_sgpr.log) -- and only once every quotient of model constants (1 / mass, 1 / Izz in fp64 ...) came
// precomputed from the host (RParams::derive): formed in the kernel from re-read values they were IEEE divisions repeated per stage,
// which made the first attempt 4 % slower.
template <typename K, typename F, typename SK, typename S, int MODEL, bool LAST>
TGR_RDEV void substep(const ArgRef<RParams<SK>> argPK, const ArgRef<RParams<S>> argP, const RMem<K, F>& M, RLag<K, F>& lag, int sidx = 0) {
  typedef typename LaneTraits<F>::I I;
  typedef typename LaneTraits<F>::B B;
  const RParams<SK>* PKp = argPK.ptr_same();
  const RParams<S>* Pp = argP.ptr_same();
#define PK (*PKp)
#define P (*Pp)
#ifndef TGR_NO_REARG
#define TGR_REARG() do { PKp = argPK.ptr(); Pp = argP.ptr(); } while (0)
#else
#define TGR_REARG() ((void)0)
#endif
  const F zero = F(S(0));
  const K kzero = K(SK(0));
  const V3<F> z3 = mk<F>(zero, zero, zero);
  const V3<K> kz3 = mk<K>(kzero, kzero, kzero);
#define im (P.imass)
#define imk (PK.imass)
  typename LaneTraits<F>::B any1 = typename LaneTraits<F>::B(false);  // MODEL 1: my page-1 slot (a cap's second terrain contact) is active
  RContact<F> c;
  V3<F> jl0, ja0;  // my slot's residual at the unconstrained acceleration: J (a0 + B v) + c0 n
  F dropped;
  I pl;            // the bar pair my POOL lane's contact belongs to (-1: none); held in a register through the Newton loop
  // MODEL 1, mixed precision: my tendon's Jacobian pieces stay in K registers for the implicitfast stage (see there); otherwise td holds them
  constexpr bool TEN_K = (MODEL != 0) && TGR_IMPLICIT_K && !std::is_same<K, F>::value;
  // MODEL 1: Newton iterates on e = d - d_ref, d_ref = the warm start.  The uneven-ground model's tendons (1e4 N/m, uneven xml:38) and cap
  // friction give the bars explicit accelerations of 100 .. 1000 rad/s^2 that the implicit stage takes back; an F iterate of that size is
  // known to 6e-5 rad/s^2, which times a lever and a contact's 1 / R is 1e-2 N of noise in the gradient -- the solver then stalls far above
  // the tolerance (measured in the emulation by rounding single quantities of the fp64 instantiation to fp32: the iterate alone left 72 % of
  // lane-sub-steps unconverged at tol 1e-5, everything else together 5 %).  e is small, and J d_ref is folded into the residual once, in K.
  constexpr bool SHIFT = MODEL != 0;
  V3<K> ten_u = kz3, ten_ma = kz3, ten_mb = kz3;
  TGR_RSTAMP(15);  // (between sub-steps: loop overhead)
  {
    TGR_PHASE(0);
    const Roles<F> Ro = lane_roles<F, MODEL>();
    // ---- kinematics: normalised orientation and bar axis of every bar, back into the slabs.  Lane b works on bar b (lanes 3.. repeat bar 0)
    // and stores ITS bar: nobody needs the result in registers -- every later stage reads the slabs
    {
      const I bi = sel(Ro.r == 1, I(1), sel(Ro.r == 2, I(2), I(0)));
      K q[4] = {M.kd.ld(KQ + 4 * bi), M.kd.ld(KQ + 4 * bi + 1), M.kd.ld(KQ + 4 * bi + 2), M.kd.ld(KQ + 4 * bi + 3)};
      normalize_quat(q);
      for (int i = 0; i < 4; i++) M.kd.st_rep(KQ + 4 * bi + i, q[i]);
      const V3<K> kz = mk<K>(SK(2) * (q[1] * q[3] + q[0] * q[2]), SK(2) * (q[2] * q[3] - q[0] * q[1]), q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]);
      st3_rep(M.kd, KEZ + 3 * bi, kz);
      st3_rep(M.fd, FEZ + 3 * bi, cvt3<F>(kz));
    }
    if (LAST) {  // the lagged orientations (what the observation is built from), and
      for (int i = 0; i < 12; i++) M.lp.st(lidx(LPQ + i), cvt<F>(M.kd.ld(KQ + i)));
    }
    if (LAST) {  // the lagged bar positions, relative to their centre (K); the centre itself stays in K (lc)
      const V3<K> x0 = ld3(M.kd, KX), x1 = ld3(M.kd, KX + 3), x2 = ld3(M.kd, KX + 6);
      const V3<K> com = SK(1.0 / 3.0) * ((x0 + x1) + x2);
      st3(M.lc, LCX, com);
      const V3<F> l0 = cvt3<F>(x0 - com), l1 = cvt3<F>(x1 - com), l2 = cvt3<F>(x2 - com);
      M.lp.st(lidx(LPX), l0.x); M.lp.st(lidx(LPX + 1), l0.y); M.lp.st(lidx(LPX + 2), l0.z);
      M.lp.st(lidx(LPX + 3), l1.x); M.lp.st(lidx(LPX + 4), l1.y); M.lp.st(lidx(LPX + 5), l1.z);
      M.lp.st(lidx(LPX + 6), l2.x); M.lp.st(lidx(LPX + 7), l2.y); M.lp.st(lidx(LPX + 8), l2.z);
    }
    TGR_RSTAMP(0);
    TGR_PHASE(1); TGR_REARG();
    // ---- tendons (tendon role): geometry, force, Jacobian pieces; smooth wrench per bar (replicated).  Each lane reads the pose of
    // ITS two bars straight from the slab (a lane-dependent LDS address), so no lane ever holds all three bars
    {
      const I ia = Ro.ba, ib = Ro.bb;
      auto site = [&](I bar, V3<K> s) {  // world offset of a site: rotate its local coordinates by the bar's quaternion
        const K w = M.kd.ld(KQ + 4 * bar);
        const V3<K> qv = ld3(M.kd, KQ + 4 * bar + 1);
        const V3<K> t = SK(2) * cross(qv, s);
        return s + w * t + cross(qv, t);
      };
      const V3<K> ra = site(ia, ld3(M.rd, RSA));
      TGR_SCHED();
      const V3<K> rb = site(ib, ld3(M.rd, RSB));
      TGR_SCHED();
      const V3<K> dif = (ld3(M.kd, KX + 3 * ib) + rb) - (ld3(M.kd, KX + 3 * ia) + ra);
      const K L2 = dot(dif, dif);
      const K iL = rsqrt_(max_(L2, K(SK(1e-30))));
      const K L = L2 * iL;
      if (LAST) static_for<9>([&](auto tc) { constexpr int t = decltype(tc)::value; M.lp.st(lidx(LPT + t), cvt<F>(bcast<t>(L))); });
      const V3<K> u = iL * dif;
      TGR_SCHED();
      const V3<K> va = cvt3<K>(ld3(M.fd, FV + 3 * ia)) + cross(cvt3<K>(ld3(M.fd, FW + 3 * ia)), ra);
      const V3<K> vb = cvt3<K>(ld3(M.fd, FV + 3 * ib)) + cross(cvt3<K>(ld3(M.fd, FW + 3 * ib)), rb);
      const K Ld = dot(u, vb - va);
      TGR_SCHED();
      const I ti = sel(Ro.actuated, Ro.r, I(0));  // my actuator (lanes >= 6: unused)
      const K tk = M.rd.ld(RTK), tb = M.rd.ld(RTB), tL0 = M.rd.ld(RTL0);
      const K passive = -tk * (L - tL0) - tb * Ld;
      K f, damp;
      if (MODEL) {  // dyntype filter, no bias, no force limit (uneven xml:39)
        f = sel(Ro.actuated, PK.gain * M.kd.ld(KACT + ti), kzero) + passive;
        damp = tb;
      } else {
        const K cc = clamp_nan_(cvt<K>(M.fd.ld(FCTRL + ti)), K(PK.ctrl_lo), K(PK.ctrl_hi));
        const K fr = PK.gain * cc + PK.bias0 + PK.bias1 * L + PK.bias2 * Ld;
        const K fa = clamp_nan_(fr, K(PK.force_lo), K(PK.force_hi));  // (NaN-propagating: a non-finite action must reach the state)
        const auto clamped = (fa <= PK.force_lo) | (fa >= PK.force_hi);
        K dact = K(-PK.bias2);
        if (PK.skip_clamped) dact = sel(clamped, kzero, dact);
        f = sel(Ro.actuated, fa, kzero) + passive;
        damp = sel(Ro.actuated, dact, kzero) + tb;
      }
      f = sel(Ro.is_ten, f, kzero);
      TGR_SCHED();
      const V3<K> ma = cross(ra, u), mb = cross(rb, u);
      if (TEN_K) { ten_u = u; ten_ma = ma; ten_mb = mb; }
#ifndef TGR_ROW_EMUL
      if (lane_id() < 9)  // (the other lanes' columns of this slab hold the solver's history, hd)
#endif
      {
        st3(M.td, TU, cvt3<F>(u)); st3(M.td, TMA, cvt3<F>(ma)); st3(M.td, TMB, cvt3<F>(mb));
        M.td.st(TDAMP, cvt<F>(damp));
      }
      // Tendon t pulls bars (A, B) = (0,2), (0,1), (1,2) for t mod 3 = 0, 1, 2 (static_assert at lane_roles): the sums per class land on
      // lanes 0, 1, 2 (trisum) and a bar's wrench is two broadcasts of them -- no masked sum over the row.
      // the force sum only feeds linear accelerations (fp32 resolves them); the torque sum decides the spin about the bar's own axis
      const K ften = sel(Ro.is_ten, f, kzero);
      // (uneven-ground model: tendon forces of 1e3 N that cancel to a few N per bar -- their sum, too, is formed in K)
      typedef typename std::conditional<MODEL != 0, K, F>::type FS;
      const V3<FS> Sfu = mk<FS>(trisum(cvt<FS>(ften * u.x)), trisum(cvt<FS>(ften * u.y)), trisum(cvt<FS>(ften * u.z)));
      const V3<K> Sma = mk<K>(trisum(ften * ma.x), trisum(ften * ma.y), trisum(ften * ma.z));
      const V3<K> Smb = mk<K>(trisum(ften * mb.x), trisum(ften * mb.y), trisum(ften * mb.z));
      // the three bars' tendon wrenches from the class sums (replicated), then lane b finishes bar b -- gravity, the gyroscopic torque and the
      // inverse inertia are fp64 work that three lanes do side by side instead of every lane three times -- and stores it
      V3<K> Fb[3], Tb[3];
      {
        const V3<FS> s0 = bcast3v<0>(Sfu), s1 = bcast3v<1>(Sfu), s2 = bcast3v<2>(Sfu);
        Fb[0] = cvt3<K>(-(s0 + s1)); Fb[1] = cvt3<K>(s1 - s2); Fb[2] = cvt3<K>(s0 + s2);
        const V3<K> a0 = bcast3v<0>(Sma), a1 = bcast3v<1>(Sma), a2 = bcast3v<2>(Sma), b0 = bcast3v<0>(Smb), b1 = bcast3v<1>(Smb), b2 = bcast3v<2>(Smb);
        Tb[0] = -(a0 + a1); Tb[1] = b1 - a2; Tb[2] = b0 + b2;
      }
      {
        const I bi = sel(Ro.r == 1, I(1), sel(Ro.r == 2, I(2), I(0)));
        V3<K> Fm = pick_bar(bi, Fb);
        const V3<K> Tm = pick_bar(bi, Tb);
        Fm.z = Fm.z - PK.mass * PK.grav;
        const V3<K> wk = cvt3<K>(ld3(M.fd, FW + 3 * bi)), kz = ld3(M.kd, KEZ + 3 * bi);
        const V3<K> Tq = Tm + (-(PK.Izz - PK.Ixx) * dot(kz, wk)) * cross(wk, kz);  // gyroscopic
        st3_rep(M.kd, KL0 + 3 * bi, imk * Fm); st3_rep(M.kd, KA0 + 3 * bi, inv_inertia(PK, kz, Tq));
        TGR_SCHED();
      }
    }
    TGR_RSTAMP(1);
    TGR_PHASE(2); TGR_REARG();
    // ---- contacts (slot role)
    I jb = sel(Ro.sB >= 0, Ro.sB, I(0)), ja_ = sel(Ro.sA >= 0, Ro.sA, I(0));  // my slot's bars (a missing one reads bar 0 and is masked)
    auto hasA = Ro.sA >= 0;
    {
      // caps on the floor: the distance is a small difference of pose quantities -> K
      const V3<K> kzc = ld3(M.kd, KEZ + 3 * jb), kxc = ld3(M.kd, KX + 3 * jb);
      const V3<F> arm = (Ro.cap_sign * P.bar_half) * cvt3<F>(kzc);
      // this environment's cap radius: the model's, plus the episode's cap-size noise (tr_env.py:685-706; zero unless use_cap_size_noise)
      const F capr = P.cap_r + M.fd.ld(FCAPD);
      F dist, cc, dn;
      V3<F> fn = mk<F>(zero, zero, F(S(1))), frB;  // normal and arm of my floor contact (cap, or MODEL 1: rod)
      V3<F> foff = z3;                              // MODEL 1: the same arm without its part along the bar's axis
      auto fhit = Ro.is_cap;
      if (MODEL) {
        // heightfield floor (uneven xml:48,56): caps get their two deepest triangles (page 0 / page 1), a bar's r-cylinder one edge
        const V3<K> org = mk<K>(K(PK.hf_x0), K(PK.hf_y0), K(PK.hf_z0));
        F d2[2]; V3<F> n2[2];
        I fid[2], eid = I(0);
        V3<F> rn, rarm, roff = z3;
        hfield_cap<K, F>(PK, P, capr, (kxc + (cvt<K>(Ro.cap_sign) * PK.bar_half) * kzc) - org, d2, n2, fid);
        F rd = F(S(1));
        rn = mk<F>(zero, zero, F(S(1))); rarm = z3;
        static_for<3>([&](auto bc_) {  // (every bar's cylinder: the whole row works on it, lane 12 + b keeps the outcome)
          constexpr int b = decltype(bc_)::value;
          V3<F> bn, barm, boff;
          I be;
          const F bd = hfield_rod_row<K, F>(PK, P, ld3(M.kd, KX + 3 * b) - org, ld3(M.kd, KEZ + 3 * b), bn, barm, boff, be);
          const auto mine = Ro.r == 12 + b;
          rd = sel(mine, bd, rd); rn = sel3(mine, bn, rn); rarm = sel3(mine, barm, rarm); roff = sel3(mine, boff, roff); eid = sel(mine, be, eid);
        });
        const auto hit0 = sel(Ro.is_cap, d2[0] < S(0), sel(Ro.is_rod, rd < S(0), false));
        dist = sel(hit0, sel(Ro.is_cap, d2[0], rd), zero);
        fn = sel3(Ro.is_cap, n2[0], rn);
        foff = sel3(Ro.is_cap, -(capr + S(0.5) * dist) * n2[0], roff);
        frB = sel3(Ro.is_cap, arm + foff, rarm);
        fhit = hit0;
        // page 1: the cap's second terrain contact
        const auto hit1 = Ro.is_cap & (d2[1] < S(0));
        const F dist1 = sel(hit1, d2[1], zero);
        F cc1, dn1;
        contact_params(P, dist1, S(1), cc1, dn1);
        const V3<F> off1 = -(capr + S(0.5) * dist1) * n2[1];
        const V3<F> rB1 = arm + off1;
        st3(M.cd, C_SLAB + CN, n2[1]); st3(M.cd, C_SLAB + CRA, z3); st3(M.cd, C_SLAB + CRB, rB1);
        M.cd.st(C_SLAB + CC0, cc1); M.cd.st(C_SLAB + CDN, dn1); M.cd.st(C_SLAB + CACT, sel(hit1, F(S(1)), zero));
        st3(M.ad, A_SLAB + AMB, cross(kzc, cvt3<K>(off1))); st3(M.ad, A_SLAB + AZB, kzc);
        const V3<F> ylB1 = cvt3<F>(ld3(M.kd, KL0 + 3 * jb)) + P.B * ld3(M.fd, FV + 3 * jb), yaB1 = cvt3<F>(ld3(M.kd, KA0 + 3 * jb)) + P.B * ld3(M.fd, FW + 3 * jb);
        st3(M.cd, C_SLAB + CJL0, (ylB1 + cross(yaB1, rB1)) + cc1 * n2[1]); st3(M.cd, C_SLAB + CJA0, yaB1);
        any1 = hit1;
        if (M.dbg && sidx < TRACE_SUB) {
          // trace, second word of the sub-step: WHICH facets / grid edges the terrain contacts touch (the mask of active slots does not
          // say) -- low 16 bits: the caps (a cap's two facets enter symmetrically), high 16 bits: the bars' cylinders; the CPU checker hashes its
          // own contacts the same way (tgo_feature_hash)
          auto mix = [&](I id) { return xshr(wmul(id + 1, I((int)2654435761u)), 15); };
          const I zi = I(0);
          const I capterm = sel(Ro.is_cap, wmul(Ro.r + 1, wadd(sel(hit0, mix(fid[0]), zi), sel(hit1, mix(fid[1]), zi))), zi);
          const I rodterm = sel(Ro.is_rod & hit0, wmul(Ro.r - 11, mix(eid)), zi);
          const I caps = rowsum_w(capterm), rods = rowsum_w(rodterm);
          M.dbg[TRACE_SUB + sidx] = ((unsigned)lane0(xshr(caps, 16)) & 0xffffu) | (((unsigned)lane0(xshr(rods, 16)) & 0xffffu) << 16);
        }
      } else {
        const K kdist = kxc.z + (cvt<K>(Ro.cap_sign) * PK.bar_half) * kzc.z - (PK.cap_r + cvt<K>(M.fd.ld(FCAPD)));
        dist = cvt<F>(kdist);
        frB = mk<F>(arm.x, arm.y, arm.z - (capr + S(0.5) * dist));
        fhit = Ro.is_cap & (dist < S(0));
      }
      contact_params(P, dist, S(1), cc, dn);
      const auto is_floor = MODEL ? (Ro.is_cap | Ro.is_rod) : Ro.is_cap;
      // bar pairs: every lane runs the routine on the bars of its role; the side lanes (6, 8, 10: pairs (0,1), (0,2), (1,2)) own the outcome
      // (positions relative to bar A's centre, the difference formed in K: a robot that has walked 30 m resolves a 0.5 mm bar-bar
      // penetration as well as one at the origin -- in absolute F coordinates it would carry 2e-6 m of rounding)
      const V3<F> pxA = z3, pzA = ld3(M.fd, FEZ + 3 * ja_), pxB = cvt3<F>(kxc - ld3(M.kd, KX + 3 * ja_)), pzB = cvt3<F>(kzc);
      RContact<F> p0;
      TGR_SCHED();
      const RPairGeom<F> G = pair_side_contact(P, capr, pxA, pzA, pxB, pzB, Ro.is_side, p0, S(2e-5));
      TGR_SCHED();
      c.n = sel3(is_floor, fn, p0.n);
      c.rA = sel3(is_floor, z3, p0.rA);
      c.rB = sel3(is_floor, frB, p0.rB);
      c.c0 = sel(is_floor, cc, p0.c0);
      c.Dn = sel(is_floor, dn, p0.Dn);
      c.active = (is_floor & fhit) | (!is_floor & Ro.is_side & p0.active);
      pl = I(-1);
      dropped = zero;
      if (any_lane(G.more)) {
        // The OTHER contacts of the pairs, rank by rank (depth), each routed to the next free pool lane the moment it is picked.  Rare -- a bar
        // end near another bar, or a crossing deeper than 3.1 mm: one (env, sub-step) in three hundred -- so everything sits behind
        // wavefront-uniform tests.  What the pool cannot hold is counted (never seen with seven lanes; MODEL 1 has four).
        F prev_d = F(S(-1e30)), npool = zero, total = zero;
        I prev_i = I(-1);
        for (int rank = 0; rank < 24; rank++) {
          const RPick<F> pk = pair_next_other(P, G, prev_d, prev_i);
          const auto got = Ro.is_side & G.more & pk.ok;
          if (rank == 0) {
            const F cnt = sel(Ro.is_side & G.more, pk.count, zero);
            total = (bcast<6>(cnt) + bcast<8>(cnt)) + bcast<10>(cnt);
          }
          if (!any_lane(got)) break;
          prev_d = sel(pk.ok, pk.dist, prev_d); prev_i = sel(pk.ok, pk.idx, prev_i);
          RContact<F> q;
          q.n = pk.n; q.rA = pk.p - G.xA; q.rB = pk.p - G.xB;
          contact_params(P, sel(got, pk.dist, zero), S(2), q.c0, q.Dn);
          const F gotf = sel(got, F(S(1)), zero);
          static_for<3>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            const auto okp = bcast<6 + 2 * p>(gotf) != S(0);  // (replicated: does pair p of THIS environment have a contact of this rank)
            if (!any_lane(okp)) return;
            // position npool of the pool: lanes 7, 9, 11, then 12.. (MODEL 1: then 15)
            const I pos = to_int(npool);
            const I L = sel(pos < 3, 7 + 2 * pos, MODEL ? I(15) : 9 + pos);
            const auto here = okp & (npool < S(pool_size<MODEL>())) & (Ro.r == L);
            c.n = sel3(here, bcast3<6 + 2 * p>(q.n), c.n); c.rA = sel3(here, bcast3<6 + 2 * p>(q.rA), c.rA); c.rB = sel3(here, bcast3<6 + 2 * p>(q.rB), c.rB);
            c.c0 = sel(here, bcast<6 + 2 * p>(q.c0), c.c0); c.Dn = sel(here, bcast<6 + 2 * p>(q.Dn), c.Dn);
            c.active = c.active | here;
            pl = sel(here, I(p), pl);
            npool = npool + sel(okp, F(S(1)), zero);
          });
        }
        dropped = sel((Ro.r == 0) & (total > S(pool_size<MODEL>())), total - S(pool_size<MODEL>()), zero);
#if defined(TGR_OTHERS_HIST) && !defined(TGR_ROW_EMUL)
        // development aid: how many "other" contacts the bar pairs of an environment have in all, per (env, sub-step) that has any
        if ((Ro.r == 0) & (total > S(0))) atomicAdd(&g_others_hist[(int)min_(total, F(S(31)))], 1ull);
#endif
      }
      // my slot's bars, now that a pool lane knows whose contact it carries
      const auto pooled = pl >= 0;
      jb = sel(pooled, sel(pl == 0, I(1), I(2)), jb); ja_ = sel(pooled, sel(pl == 2, I(1), I(0)), ja_);
      hasA = hasA | pooled;
      V3<K> kzcL = kzc;
      if (any_lane(pooled)) kzcL = ld3(M.kd, KEZ + 3 * jb);
      {
        // Two bars side to side: whether they touch, and how deep, is decided on the distance of the two axes formed again in K (the F
        // routine above names the candidate): over relative coordinates of 0.3 m an F distance carries 1e-7 m, ten times what the pose
        // itself is uncertain by -- a third of the flat floor's contact flips started at such a contact.
        const auto side = Ro.is_side & c.active;
        if (any_lane(side)) {
          const V3<K> zAk = ld3(M.kd, KEZ + 3 * ja_), wv = ld3(M.kd, KX + 3 * ja_) - kxc;
          const K b = dot(zAk, kzc), dA = dot(zAk, wv), dB = dot(kzc, wv);
          const K iden = K(SK(1)) / max_(K(SK(1)) - b * b, K(SK(1e-12)));
          const K s_ = (b * dB - dA) * iden, t_ = (dB - b * dA) * iden;
          const V3<K> dif = (t_ * kzc - s_ * zAk) - wv;
          const K distK = sqrt_(dot(dif, dif)) - SK(2) * PK.rcyl_r;
          const auto hitK = side & (distK < SK(0));
          F cc0, dn0;
          contact_params(P, sel(hitK, cvt<F>(distK), zero), S(2), cc0, dn0);
          c.active = (side & hitK) | (!side & c.active);
          c.c0 = sel(side, sel(hitK, cc0, zero), c.c0); c.Dn = sel(side, sel(hitK, dn0, zero), c.Dn);
        }
      }
      // The torque of my contact force about the OWN AXIS of each of its bars is the one place where fp32 is not enough (1e-6 N m inside
      // 10 N m cross products; the bar's inertia about that axis is 0.0038 kg m^2):  z . (r x f + t) = f . (z x r) + t . z, with the
      // levers z x r in K.  For a cap the axial part of r drops out of z x r exactly, so only its small offset from the bar end is used.
      const V3<K> kzA = sel3(hasA, ld3(M.kd, KEZ + 3 * ja_), kz3);
      // (On the heightfield the normal force has a lever about the axis too -- on the flat floor z x (0, 0, r) . (0, 0, f) is exactly 0 --
      // so there the offset is taken as it was built, -(r + dist / 2) n, not back out of the rounded sum arm + offset: that round trip
      // leaves 1e-7 m in a random direction, times a normal force of 10 N, on an inertia of 0.0038 kg m^2.)
      const V3<K> offB = cvt3<K>(MODEL ? sel3(is_floor, foff, c.rB) : sel3(Ro.is_cap, c.rB - arm, c.rB));
      st3(M.ad, AMB, cross(kzcL, offB)); st3(M.ad, AMA, cross(kzA, cvt3<K>(c.rA)));
      st3(M.ad, AZB, kzcL); st3(M.ad, AZA, kzA);
    }
    // my slot's residual at the unconstrained acceleration
    {
      const V3<F> ylB = cvt3<F>(ld3(M.kd, KL0 + 3 * jb)) + P.B * ld3(M.fd, FV + 3 * jb), yaB = cvt3<F>(ld3(M.kd, KA0 + 3 * jb)) + P.B * ld3(M.fd, FW + 3 * jb);
      const V3<F> ylA = sel3(hasA, cvt3<F>(ld3(M.kd, KL0 + 3 * ja_)) + P.B * ld3(M.fd, FV + 3 * ja_), z3);
      const V3<F> yaA = sel3(hasA, cvt3<F>(ld3(M.kd, KA0 + 3 * ja_)) + P.B * ld3(M.fd, FW + 3 * ja_), z3);
      jl0 = (ylB + cross(yaB, c.rB)) - (ylA + cross(yaA, c.rA)) + c.c0 * c.n;
      ja0 = yaB - yaA;
    }
  }
  TGR_RSTAMP(2);
  if (LAST) lag.overflow = rowsum(dropped);
  const auto any_contact = rowsum(sel(c.active | any1, F(S(1)), zero)) > S(0);
  // MODEL 1: does any cap of the wavefront rest on two facets?  (uniform; the whole second page is skipped otherwise -- the common case)
  const bool page1 = MODEL && any_lane(any1);
  // ---- Newton on d = a - a0 (replicated).  Stopping rule: sqrt(g^T M^-1 g) < tol, in sqrt(kg) m/s^2.  mujoco's rule -- the plain norm
  // of g over (mean inertia x nv) -- lets a torque residual about a bar's own axis (inertia 0.0038 kg m^2) pass that still spins the bar
  // up by 1e-2 rad/s^2; weighting by M^-1 bounds the acceleration error itself (H >= M), which is what the tolerance on qvel is about.
  // Everything the iterations share lives in the slabs, not in registers: the iterate d (fd: FDL / FDA, and picked for my slot's two
  // bars in cd), my contact slot (cd), the bar axes (fd: FEZ), the constraint wrench of the last assembly (fd: FFCL / FFCA, kd: KTAX).
  // Start: last sub-step's constraint acceleration (P.warm_start) -- a contact's reference acceleration moves by ~B h a every sub-step,
  // so last sub-step's ACCELERATION is a poor guess, but the part of it the contacts produced is a good one -- or zero.
  {
    const Roles<F> Ro = lane_roles<F, MODEL>(pl);
    V3<F> dl[3], da[3];
    const F pc = M.hd.ld(hidx(HPC)), ppc = M.hd.ld(hidx(HPPC));
    #ifndef TGR_WX
#define TGR_WX 0.75
#endif
    const F wx = S(TGR_WX) * pc * ppc;
    M.hd.st(hidx(HPPC), pc); M.hd.st(hidx(HPC), sel(any_contact, F(S(1)), zero));
    for (int b = 0; b < 3; b++) {
      // ... extrapolated by 3/4 of its last change when the two sub-steps before this one both had contacts (-9 % iterations)
      const V3<F> cl = ld3(M.fd, FDL + 3 * b), ca = ld3(M.fd, FDA + 3 * b);
      const V3<F> ql_ = mk<F>(M.hd.ld(hidx(HD + 3 * b)), M.hd.ld(hidx(HD + 3 * b + 1)), M.hd.ld(hidx(HD + 3 * b + 2)));
      const V3<F> qa_ = mk<F>(M.hd.ld(hidx(HD + 9 + 3 * b)), M.hd.ld(hidx(HD + 9 + 3 * b + 1)), M.hd.ld(hidx(HD + 9 + 3 * b + 2)));
      M.hd.st(hidx(HD + 3 * b), cl.x); M.hd.st(hidx(HD + 3 * b + 1), cl.y); M.hd.st(hidx(HD + 3 * b + 2), cl.z);
      M.hd.st(hidx(HD + 9 + 3 * b), ca.x); M.hd.st(hidx(HD + 9 + 3 * b + 1), ca.y); M.hd.st(hidx(HD + 9 + 3 * b + 2), ca.z);
      dl[b] = P.warm_start ? sel3(any_contact, cl + wx * (cl - ql_), z3) : z3;
      da[b] = P.warm_start ? sel3(any_contact, ca + wx * (ca - qa_), z3) : z3;
      if (SHIFT) {
        // the start becomes the reference d_ref; the iterate e = d - d_ref starts at zero.  I d_ref is formed once, in K
        const V3<K> kez = ld3(M.kd, KEZ + 3 * b), kda = cvt3<K>(da[b]);
        st3(M.fd, FRL + 3 * b, dl[b]); st3(M.fd, FRA + 3 * b, da[b]);
        st3(M.fd, FGA + 3 * b, cvt3<F>(PK.Ixx * kda + ((PK.Izz - PK.Ixx) * dot(kez, kda)) * kez));
        st3(M.fd, FDL + 3 * b, z3); st3(M.fd, FDA + 3 * b, z3);
      } else {
        st3(M.fd, FDL + 3 * b, dl[b]); st3(M.fd, FDA + 3 * b, da[b]);
      }
    }
    st3(M.cd, CN, c.n); st3(M.cd, CRA, c.rA); st3(M.cd, CRB, c.rB);
    M.cd.st(CC0, c.c0); M.cd.st(CDN, c.Dn); M.cd.st(CACT, sel(c.active, F(S(1)), zero));
    if (SHIFT) {
      // my slot's residual AT d_ref, in K: J (a0 + B v) + c0 n and J d_ref are 10 m/s^2 each (100..1000 rad/s^2 times levers) and cancel to
      // the 0.1 m/s^2 the contact really sees; the iterations then only add J e, which is small
      const I jb = sel(Ro.sB >= 0, Ro.sB, I(0)), ja_ = sel(Ro.sA >= 0, Ro.sA, I(0));
      const auto hasA = Ro.sA >= 0;
      const V3<K> rl = cvt3<K>(pick_bar0(Ro.sB, dl)), ra = cvt3<K>(pick_bar0(Ro.sB, da));
      const V3<K> ylB = (ld3(M.kd, KL0 + 3 * jb) + PK.B * cvt3<K>(ld3(M.fd, FV + 3 * jb))) + rl;
      const V3<K> yaB = (ld3(M.kd, KA0 + 3 * jb) + PK.B * cvt3<K>(ld3(M.fd, FW + 3 * jb))) + ra;
      const V3<K> ylA = sel3(hasA, (ld3(M.kd, KL0 + 3 * ja_) + PK.B * cvt3<K>(ld3(M.fd, FV + 3 * ja_))) + cvt3<K>(pick_bar0(Ro.sA, dl)), kz3);
      const V3<K> yaA = sel3(hasA, (ld3(M.kd, KA0 + 3 * ja_) + PK.B * cvt3<K>(ld3(M.fd, FW + 3 * ja_))) + cvt3<K>(pick_bar0(Ro.sA, da)), kz3);
      st3(M.cd, CJL0, cvt3<F>((ylB + cross(yaB, cvt3<K>(c.rB))) - (ylA + cross(yaA, cvt3<K>(c.rA))) + cvt<K>(c.c0) * cvt3<K>(c.n)));
      st3(M.cd, CJA0, cvt3<F>(yaB - yaA));
      // page 1 (a cap's second facet: same bar, no body A)
      if (MODEL) {
        const V3<K> rB1 = cvt3<K>(ld3(M.cd, C_SLAB + CRB)), n1 = cvt3<K>(ld3(M.cd, C_SLAB + CN));
        st3(M.cd, C_SLAB + CJL0, cvt3<F>((ylB + cross(yaB, rB1)) + cvt<K>(M.cd.ld(C_SLAB + CC0)) * n1));
        st3(M.cd, C_SLAB + CJA0, cvt3<F>(yaB));
      }
    } else {
      st3(M.cd, CJL0, jl0); st3(M.cd, CJA0, ja0);
    }
  }
  auto conv = !any_contact;
  auto stall = conv & !conv;  // rows whose line search found no descent at working precision
  F bar_force = zero;
  int n_it = 0, n_ls = 0;
  TGR_FENCE();
  TGR_RSTAMP(3);  // Newton set-up (warm start, slot into the slab)
  TGR_REARG();
  for (int it = 0;; it++) {
    TGR_PHASE(3);
    const Roles<F> Ro = lane_roles<F, MODEL>(pl);
    DofC<F> dc;  // (repeated in the upper half of the row, lanes 8..13: hess_add_slot works on a pair slot's two bodies side by side)
    {
      const I r8 = Ro.r & 7;
      const I k8 = sel(r8 < 3, r8, r8 - 3);
      dc.dof = r8 < 6; dc.ang = (r8 >= 3) & (r8 < 6); dc.upper = Ro.r >= 8;
      dc.e = mk<F>(sel(dc.dof & (k8 == 0), F(S(1)), zero), sel(dc.dof & (k8 == 1), F(S(1)), zero), sel(dc.dof & (k8 == 2), F(S(1)), zero));
    }
    // ---- my slot's residual jar at the current iterate, cone force
    // (my slot: in registers since the contact stage -- it is read in every stage of the iteration, so holding it across the back edge costs
    // nothing at the loop's peak, and re-reading it from the slab at the head of every iteration was twelve loads the cone evaluation waited for)
    const RContact<F>& cs = c;
    // (the iterate at my slot's two bars, straight from the replicated iterate: a lane-dependent LDS address; a slot without a bar A -- floor
    // contacts -- takes zero there, an idle lane reads bar 0 and is masked by its inactive slot)
    const I jbi = sel(Ro.sB >= 0, Ro.sB, I(0)), jai = sel(Ro.sA >= 0, Ro.sA, I(0));
    const auto hasAi = Ro.sA >= 0;
    const V3<F> dLB = ld3(M.fd, FDL + 3 * jbi), dAB = ld3(M.fd, FDA + 3 * jbi);
    const V3<F> dLA = sel3(hasAi, ld3(M.fd, FDL + 3 * jai), z3), dAA = sel3(hasAi, ld3(M.fd, FDA + 3 * jai), z3);
    // (the residual at the start of the iteration: likewise in registers -- jl0, ja0 -- where the stage that forms it leaves it there; MODEL 1
    // forms it in K at the reference iterate and parks it in the slab)
    const V3<F> jl = (SHIFT ? ld3(M.cd, CJL0) : jl0) + ((dLB + cross(dAB, cs.rB)) - (dLA + cross(dAA, cs.rA)));
    const V3<F> ja = (SHIFT ? ld3(M.cd, CJA0) : ja0) + (dAB - dAA);
    const RCone<F> o = cone_eval(P, cs, jl, ja);
    // MODEL 1, page 1: a cap's second terrain contact (same bar, no body A)
    RContact<F> cs1;
    V3<F> jl1 = z3, ja1 = z3;
    RCone<F> o1;
    cs1.active = typename LaneTraits<F>::B(false);
    o1.fl = z3; o1.fa = z3;
    if (page1) {
      cs1.n = ld3(M.cd, C_SLAB + CN); cs1.rA = z3; cs1.rB = ld3(M.cd, C_SLAB + CRB);
      cs1.c0 = M.cd.ld(C_SLAB + CC0); cs1.Dn = M.cd.ld(C_SLAB + CDN); cs1.active = M.cd.ld(C_SLAB + CACT) != S(0);
      jl1 = ld3(M.cd, C_SLAB + CJL0) + (dLB + cross(dAB, cs1.rB));
      ja1 = ld3(M.cd, C_SLAB + CJA0) + dAB;
      o1 = cone_eval(P, cs1, jl1, ja1);
    }
    TGR_RSTAMP(4);  // residual + cone
    TGR_PHASE(4);
    // ---- constraint wrench per bar, gradient (dof role: component r of each bar's residual)
    V3<F> fB = o.fl, tB = cross(cs.rB, o.fl) + o.fa;
    const V3<F> tA = cross(cs.rA, o.fl) + o.fa;
    K tauB, tauA;
    {
      const V3<K> kfl = cvt3<K>(o.fl), kfa = cvt3<K>(o.fa);
      tauB = dot(kfl, ld3(M.ad, AMB)) + dot(kfa, ld3(M.ad, AZB));
      tauA = dot(kfl, ld3(M.ad, AMA)) + dot(kfa, ld3(M.ad, AZA));
    }
    if (page1) {
      fB = fB + o1.fl; tB = tB + cross(cs1.rB, o1.fl) + o1.fa;
      tauB = tauB + dot(cvt3<K>(o1.fl), ld3(M.ad, A_SLAB + AMB)) + dot(cvt3<K>(o1.fa), ld3(M.ad, A_SLAB + AZB));
    }
    TGR_CENSUS(12);
    // The slots of a bar sit on fixed lanes (caps 2b, 2b+1; pair (a, b) on two neighbouring lanes), so its wrench is not a masked
    // sum over the row: neighbour sums first (one DPP add), then three broadcasts per bar (bar_sum below).  Lanes that own no slot are
    // never read.
    const V3<F> PfB = mk<F>(pairsum(fB.x), pairsum(fB.y), pairsum(fB.z)), PtB = mk<F>(pairsum(tB.x), pairsum(tB.y), pairsum(tB.z));
    const K PkB = pairsum(tauB);
    const BMask am = ballot_of(cs.active);  // which slots are in use anywhere in the wavefront (once per iteration)
    const bool pool = slots_in<pool_mask<MODEL>()>(am);
    K gn;   // g^T M^-1 g
    F dn;   // d^T M d: the scale of the forces the gradient is the difference of
    F gd[3];
    V3<F> ez3[3];
    {
      // Every lane forms the three bars' sums (the broadcasts they are made of are the same instruction for the whole row); from there
      // on LANE b WORKS ON BAR b (lanes 3.. repeat bar 0): it stores the bar's wrench, forms the bar's gradient and its share of the two norms,
      // and leaves the gradient in the slab, where the dof lanes pick it up transposed.  A third of the arithmetic of doing every bar on
      // every lane, most of the slab traffic gone.
      const I bi = sel(Ro.r == 1, I(1), sel(Ro.r == 2, I(2), I(0)));
      // (the slab reads of my bar go out before the sums, whose DPP chains cover their latency)
      const V3<F> dl = ld3(M.fd, FDL + 3 * bi), da = ld3(M.fd, FDA + 3 * bi), ezb = ld3(M.fd, FEZ + 3 * bi);
      const V3<K> kez = ld3(M.kd, KEZ + 3 * bi);
      // (... and so do the three bar axes the Hessian's mass-matrix rows are made of: asked for at the head of the Hessian stage they were a
      // round trip through LDS with nothing to cover it)
      for (int b = 0; b < 3; b++) ez3[b] = ld3(M.fd, FEZ + 3 * b);
      TGR_SCHED();
      V3<F> fcl, fca;
      K tax;
      bar_sum<MODEL>(0, PfB, PtB, PkB, fB, tB, tA, tauB, tauA, am, pl, fcl, fca, tax);
      static_for<2>([&](auto bc) {  // (a bar's sums are taken up at once: three sets of them alive together cost registers the kernel does not have)
        constexpr int b = 1 + decltype(bc)::value;
        V3<F> f_, t_;
        K x_;
        bar_sum<MODEL>(b, PfB, PtB, PkB, fB, tB, tA, tauB, tauA, am, pl, f_, t_, x_);
        const auto mine = bi == b;
        fcl = sel3(mine, f_, fcl); fca = sel3(mine, t_, fca); tax = sel(mine, x_, tax);
        TGR_SCHED();
      });
      TGR_CENSUS(16);
      st3_rep(M.fd, FFCL + 3 * bi, fcl); st3_rep(M.fd, FFCA + 3 * bi, fca); M.kd.st_rep(KTAX + bi, tax);
      // (SHIFT: dl, da are the iterate e; the full d = d_ref + e)
      const V3<F> rl = SHIFT ? ld3(M.fd, FRL + 3 * bi) : z3, ra = SHIFT ? ld3(M.fd, FRA + 3 * bi) : z3;
      const V3<F> fl_ = SHIFT ? rl + dl : dl, fa_ = SHIFT ? ra + da : da;
      const V3<F> gl = P.mass * fl_ - fcl;
      const V3<F> Ida = SHIFT ? ld3(M.fd, FGA + 3 * bi) + inertia_mul(P, ezb, da) : inertia_mul(P, ezb, da);
      V3<F> ga = Ida - fca;
      const F dnb = P.mass * dot(fl_, fl_) + dot(Ida, fa_);
      // M-norm of the residual: force and transverse torque in F, the axial torque from K
      const K gax = PK.Izz * dot(kez, SHIFT ? cvt3<K>(ra) + cvt3<K>(da) : cvt3<K>(da)) - tax;
      const F gz = dot(ezb, ga);
      const K gnb = cvt<K>(im * dot(gl, gl) + (dot(ga, ga) - gz * gz) * P.iIxx) + gax * gax * PK.iIzz;
      ga = ga + (cvt<F>(gax) - gz) * ezb;  // the direction is solved for the same axial residual the test looks at
      st3_rep(M.fd, FG + 6 * bi, gl); st3_rep(M.fd, FG + 6 * bi + 3, ga);
      gn = (bcast<0>(gnb) + bcast<1>(gnb)) + bcast<2>(gnb);
      dn = (bcast<0>(dnb) + bcast<1>(dnb)) + bcast<2>(dnb);
      // (lanes past the sixth read component 0 and are never looked at: the solves below take their right-hand sides from lanes 0..5 by
      // broadcast.  No select on the loaded value -- its first use is in the forward solve, after the factorisation.)
      const I ri = sel(Ro.is_dof, Ro.r, I(0));
      for (int b = 0; b < 3; b++) gd[b] = M.fd.ld(FG + 6 * b + ri);
    }
    if (LAST) bar_force = rowsum(sel(Ro.is_side | Ro.pooled, sqrt_(dot(o.fl, o.fl)), zero));
    // converged: the M^-1-norm of the gradient is below the tolerance -- or below what the arithmetic resolves of it: the gradient is the
    // difference of inertial and contact forces of size sqrt(d^T M d) (a hard contact onset: 1e3 sqrt(kg) m/s^2), each known to a few ulp
    conv = conv | !(gn >= PK.tol * PK.tol + (PK.rtol * PK.rtol) * cvt<K>(dn));
#ifdef TGR_ROW_TRACE
    std::fprintf(stderr, "it %d gn %.3e conv %d\n", it, (double)gn.v[0], (int)conv.v[0]);
#endif
    TGR_RSTAMP(5);  // wrench, bar sums, gradient, convergence test
    if (!any_lane(!conv) || it >= P.max_iter) break;
    n_it++;
    lag.n_own = lag.n_own + sel(conv, zero, F(S(1)));
    TGR_PHASE(5);
    // ---- Hessian rows (dof role)
    RHess<F> H;
    {
      // my row of the mass matrix: m e for a linear dof, Ixx e + (Izz - Ixx) (e . z) z for an angular one -- the parts that do not depend on the
      // bar are formed once
      const S dI = P.Izz - P.Ixx;
      const V3<F> lin = sel3(dc.ang, z3, P.mass * dc.e), ixe = sel3(dc.ang, P.Ixx * dc.e, z3);
      for (int b = 0; b < 3; b++) {
        const V3<F> ezb = ez3[b];
        const F t = sel(dc.ang, dI * dot(dc.e, ezb), zero);
        H.D[b][0] = lin.x; H.D[b][1] = lin.y; H.D[b][2] = lin.z; H.D[b][3] = ixe.x + t * ezb.x; H.D[b][4] = ixe.y + t * ezb.y; H.D[b][5] = ixe.z + t * ezb.z;
      }
    }
    for (int i = 0; i < 6; i++) { H.O10[i] = zero; H.O20[i] = zero; H.O21[i] = zero; }
    const BMask cm = ballot_of(o.cg != S(0));
    // (the caps and the bar pairs' side-to-side slots one by one; the pool -- a pair's other contacts: 0.3 % of wavefronts per lane,
    // profiles/round4_slot_census.log -- behind ONE test of the ballot: a skipped slot is a scalar test and a taken branch)
    static_for<6>([&](auto cc) { hess_add_slot<MODEL, 0, decltype(cc)::value>(H, dc, cs, o, am, cm); });
    static_for<3>([&](auto cc) { hess_add_slot<MODEL, 0, 6 + 2 * decltype(cc)::value>(H, dc, cs, o, am, cm); });
    if constexpr (MODEL != 0) static_for<3>([&](auto cc) { hess_add_slot<MODEL, 0, 12 + decltype(cc)::value>(H, dc, cs, o, am, cm); });
    if (pool) {
      static_for<3>([&](auto cc) { hess_add_pool<7 + 2 * decltype(cc)::value>(H, dc, cs, o, pl, am, cm); });
      if constexpr (MODEL == 0) static_for<3>([&](auto cc) { hess_add_pool<12 + decltype(cc)::value>(H, dc, cs, o, pl, am, cm); });
      hess_add_pool<15>(H, dc, cs, o, pl, am, cm);
    }
    if (page1) {
      const BMask am1 = ballot_of(cs1.active), cm1 = ballot_of(o1.cg != S(0));
      static_for<6>([&](auto cc) { hess_add_slot<MODEL, 1, decltype(cc)::value>(H, dc, cs1, o1, am1, cm1); });
    }
    // which bar pairs does any environment of the wavefront couple?  (uniform: lets the factorisation skip zero blocks)
    bool cp0 = slot_in<6>(am), cp1 = slot_in<8>(am), cp2 = slot_in<10>(am);
    if (pool) { cp0 = cp0 || any_lane(Ro.pooled & cs.active & (pl == 0)); cp1 = cp1 || any_lane(Ro.pooled & cs.active & (pl == 1)); cp2 = cp2 || any_lane(Ro.pooled & cs.active & (pl == 2)); }
    TGR_RSTAMP(6);  // Hessian rows
    TGR_PHASE(6);
    // ---- direction p = -H^-1 g
    F pr[3][6], gHg;
    // (what the line search and the update read of my bar -- lane b: bar b -- is asked for here, before the factorisation that covers the trip)
    const I bi_ = sel(Ro.r == 1, I(1), sel(Ro.r == 2, I(2), I(0)));
    const V3<F> ezm = ld3(M.fd, FEZ + 3 * bi_), dlm = ld3(M.fd, FDL + 3 * bi_), dam = ld3(M.fd, FDA + 3 * bi_);
    newton_direction(H, gd, pr, gHg, cp0, cp1, cp2, S(0.5) * P.mass, S(0.5) * P.Izz);
    const F d0 = sel(conv, zero, -gHg);  // p . g = -g^T H^-1 g
    TGR_RSTAMP(7);  // direction
    TGR_PHASE(7);
    // ---- line search: Gauss part (replicated), cone part (slot role)
    // The Gauss part's two coefficients, bar by bar: lane b takes bar b (lanes 3.. repeat bar 0), three broadcasts bring the shares together.
    // (Rows that have converged carry a zero step in the update below; what the line search computes for them in between is never used --
    // alpha stays zero, ls_done is set from the start -- so only the step of my own bar is masked.)
    V3<F> pl[3], pa[3];
    for (int b = 0; b < 3; b++) { pl[b] = mk<F>(pr[b][0], pr[b][1], pr[b][2]); pa[b] = mk<F>(pr[b][3], pr[b][4], pr[b][5]); }
    const V3<F> plb = sel3(conv, z3, pick_bar(bi_, pl)), pab = sel3(conv, z3, pick_bar(bi_, pa));
    F g1, g2;
    {
      const V3<F> Ipa = inertia_mul(P, ezm, pab);
      F g1b;
      if (SHIFT) g1b = P.mass * dot(plb, ld3(M.fd, FRL + 3 * bi_) + dlm) + (dot(pab, ld3(M.fd, FGA + 3 * bi_)) + dot(Ipa, dam));
      else g1b = P.mass * dot(plb, dlm) + dot(Ipa, dam);
      const F g2b = P.mass * dot(plb, plb) + dot(Ipa, pab);
      g1 = (bcast<0>(g1b) + bcast<1>(g1b)) + bcast<2>(g1b);
      g2 = (bcast<0>(g2b) + bcast<1>(g2b)) + bcast<2>(g2b);
    }
    // (a slot's body B always exists where the slot can be active; idle lanes -- sB < 0 -- take bar 2 and are masked by their inactive slot)
    const V3<F> plA = pick_bar0(Ro.sA, pl), paA = pick_bar0(Ro.sA, pa), plB = pick_bar(Ro.sB, pl), paB = pick_bar(Ro.sB, pa);
    const V3<F> ql = (plB + cross(paB, cs.rB)) - (plA + cross(paA, cs.rA));
    const V3<F> qa = paB - paA;
    const RLSlot<F> ls = ls_make(cs, jl, ja, ql, qa);
    RLSlot<F> ls1;
    if (page1) ls1 = ls_make(cs1, jl1, ja1, plB + cross(paB, cs1.rB), paB);
    conv = conv | !(abs_(d0) >= P.tol * P.tol);  // Newton decrement g^T H^-1 g (<= g^T M^-1 g) below tolerance
    conv = conv | !(d0 < S(0));                  // no descent left at working precision
    TGR_RSTAMP(8);  // line search set-up
    if (!any_lane(!conv)) break;                 // (the iterate stays where the wrench above was taken)
    TGR_CENSUS(10);
    F alpha = sel(conv, zero, F(S(1))), lo = zero, hi = F(S(-1));
    auto ls_done = conv;
    const F gtol = P.ls_tol * abs_(d0);
    for (int k = 0;; k++) {
      F s1, s2;
      ls_slot(P, ls, alpha, s1, s2);
      if (page1) {
        F t1, t2;
        ls_slot(P, ls1, alpha, t1, t2);
        s1 = s1 + t1; s2 = s2 + t2;
      }
      n_ls++;
      const F d1 = g1 + alpha * g2 + rowsum(s1), d2 = g2 + rowsum(s2);
      const auto small = abs_(d1) <= gtol;
      const auto go = !ls_done & !small;
      ls_done = ls_done | small;
      lo = sel(go & (d1 < S(0)), alpha, lo);
      hi = sel(go & !(d1 < S(0)), alpha, hi);
      F na = sel(d2 > S(0), alpha - d1 * rcp_(d2), F(S(-1)));
      const auto nohi = hi < S(0);
      na = sel(nohi & (!(na > lo) | (na > S(10) * (lo + S(1)))), S(2) * alpha + S(1e-3), na);
      na = sel(!nohi & (!(na > lo) | !(na < hi)), S(0.5) * (lo + hi), na);
      ls_done = ls_done | (go & (hi > S(0)) & ((hi - lo) <= S(1e-6) * hi));
      alpha = sel(go, na, alpha);
      if (!any_lane(!ls_done) || k + 1 >= P.ls_iter) break;
    }
    TGR_RSTAMP(9);  // line search evaluations
    TGR_CENSUS(11);
    {
      // every trial step pointed uphill although the Newton decrement promises descent: the slope at the iterate is below what the
      // arithmetic resolves (hard contact onsets: residuals of 1e2..1e3 m/s^2 in fp32).  The iterate stays; the row is done, and counted
      // with the unconverged ones.
      const auto stalled = !ls_done & (hi > S(0)) & !(lo > S(0));
      alpha = sel(stalled, zero, alpha);
      stall = stall | stalled;
      conv = conv | stalled;
    }
#ifdef TGR_ROW_TRACE
    std::fprintf(stderr, "      d0 %.3e alpha %.6f lo %.4f hi %.4f g1 %.3e g2 %.3e ls_done %d n_ls %d\n", (double)d0.v[0], (double)alpha.v[0], (double)lo.v[0], (double)hi.v[0], (double)g1.v[0], (double)g2.v[0], (int)ls_done.v[0], n_ls);
#endif
    {
      // the iterate moves: lane b updates bar b (lanes 3.. repeat bar 0) and stores it -- every reader takes it from the slab
      const V3<F> nl = dlm + alpha * plb, na_ = dam + alpha * pab;
      st3_rep(M.fd, FDL + 3 * bi_, nl); st3_rep(M.fd, FDA + 3 * bi_, na_);
    }
    TGR_FENCE();
    TGR_RSTAMP(10);  // iterate update
    if (!any_lane(!conv)) break;  // (only stalled rows were left: their iterate has not moved, the wrench above is the one of the iterate)
  }
  TGR_RSTAMP(11);  // (leaving the loop)
  TGR_PHASE(8); TGR_REARG();
  TGR_FENCE();
  if (SHIFT)
    for (int b = 0; b < 3; b++) {  // back to d = d_ref + e: what the next sub-step starts from and what the export calls qacc - a0
      st3(M.fd, FDL + 3 * b, ld3(M.fd, FRL + 3 * b) + ld3(M.fd, FDL + 3 * b));
      st3(M.fd, FDA + 3 * b, ld3(M.fd, FRA + 3 * b) + ld3(M.fd, FDA + 3 * b));
    }
  lag.n_it += n_it; lag.n_ls += n_ls;
  lag.n_unconv = lag.n_unconv + sel(conv & !stall, zero, F(S(1)));
  if (M.dbg && sidx < TRACE_SUB) {  // (every lane of the row stores the same word)
    unsigned w = row_bits(M.cd.ld(CACT) != S(0));
    if (MODEL) w |= (row_bits(M.cd.ld(C_SLAB + CACT) != S(0)) & 0x3fu) << 16;
    w |= ((unsigned)(int)lane0(lag.n_own) & 0xffu) << 22;
    w |= lane0(conv & !stall) ? 0u : 1u << 30;
    w |= lane0(stall) ? 1u << 31 : 0u;
    M.dbg[sidx] = w;
  }
  if (LAST) {
    F sq = zero, cfrc_max = zero;
    for (int b = 0; b < 3; b++) {
      const V3<F> fl = ld3(M.fd, FFCL + 3 * b), fa = ld3(M.fd, FFCA + 3 * b);
      cfrc_max = max_(cfrc_max, max_(max_(max_(abs_(fl.x), abs_(fl.y)), abs_(fl.z)), max_(max_(abs_(fa.x), abs_(fa.y)), abs_(fa.z))));
      const V3<F> a = mk<F>(clamp_(fl.x, F(P.cf_lo), F(P.cf_hi)), clamp_(fl.y, F(P.cf_lo), F(P.cf_hi)), clamp_(fl.z, F(P.cf_lo), F(P.cf_hi)));
      const V3<F> t = mk<F>(clamp_(fa.x, F(P.cf_lo), F(P.cf_hi)), clamp_(fa.y, F(P.cf_lo), F(P.cf_hi)), clamp_(fa.z, F(P.cf_lo), F(P.cf_hi)));
      sq = sq + dot(a, a) + dot(t, t);
    }
    lag.cfrc_clip_sq = sq;
    lag.cfrc_absmax = cfrc_max;
    lag.bar_contact = bar_force;
    // (qacc of the last forward pass -- what mujoco keeps as qacc_warmstart -- is a0 + d: both stay in the slabs, the export adds them)
  }
  // ---- implicitfast: (M + h sum_t b_t J_t J_t^T) a_new = F_smooth + F_constraint  (Woodbury over the nine tendons, tendon role).
  // W: the arithmetic of this stage.  F for the flat-floor model; K for the uneven-ground model, whose nine tendons carry 1e4 N/m and
  // 1e3 N s/m (uneven xml:38): accelerations of 1e3 m/s^2 that cancel to the few m/s^2 the robot really does -- in F that cancellation left
  // 1e-4 m/s^2 of rounding in every sub-step, 0.4 % of tr_env env-steps outside the 1e-4 bound on qvel with no contact flip to blame.
  TGR_RSTAMP(12);  // trace, last sub-step read-outs
  constexpr bool WK = (MODEL != 0) && TGR_IMPLICIT_K;
  typedef typename std::conditional<WK, K, F>::type W;
  typedef typename std::conditional<WK, SK, S>::type SW;
  const RParams<SW>& PW = pick_params<WK>(PK, P);
  const SW imw = PW.imass;
  const W wzero = W(SW(0));
  const V3<W> wz3 = mk<W>(wzero, wzero, wzero);
  V3<W> yl[3], ya[3];    // the damping correction
  {
    const Roles<F> Ro = lane_roles<F, MODEL>();
    // acceleration before the damping correction, at my tendon's two bars (lane-indexed slab reads)
    auto accel = [&](I bar, V3<W>& al_, V3<W>& aa_) {
      const V3<K> kz = ld3(M.kd, KEZ + 3 * bar);
      const V3<K> kf = cvt3<K>(ld3(M.fd, FFCA + 3 * bar));
      const V3<K> yak = PK.iIxx * kf + (M.kd.ld(KTAX + bar) * PK.iIzz - dot(kz, kf) * PK.iIxx) * kz;
      if (WK) al_ = cvt3<W>(ld3(M.kd, KL0 + 3 * bar) + imk * cvt3<K>(ld3(M.fd, FFCL + 3 * bar)));
      else al_ = cvt3<W>(cvt3<F>(ld3(M.kd, KL0 + 3 * bar)) + im * ld3(M.fd, FFCL + 3 * bar));
      aa_ = cvt3<W>(ld3(M.kd, KA0 + 3 * bar) + yak);
    };
    V3<W> flA, faA, flB, faB;
    accel(Ro.ba, flA, faA);
    TGR_SCHED();
    accel(Ro.bb, flB, faB);
    TGR_SCHED();
    // (lanes 9.. own no tendon: what they read here is the solver's history or never written -- zero, not garbage, enters the row sums)
    const W sc = sqrt_(PW.h * cvt<W>(sel(Ro.is_ten, M.td.ld(TDAMP), zero)));
    // (the Jacobian of my tendon: from the K registers where the stage is K -- a moment arm z x r of 0.7 m rounded to F is off by 4e-8 m, which
    // is 6e-7 of the 0.0675 m lever the tendon has about the bar's own axis; with tendon dampers that slave that spin to the tendon rates
    // (uneven xml:38: 1e3 N s/m on an axial inertia of 0.0038 kg m^2) it showed as 1e-6 rad/s of noise per sub-step)
    V3<W> us, mas, mbs;
    if (TEN_K) {
      const auto kt = Ro.is_ten;
      us = sc * cvt3<W>(sel3(kt, ten_u, kz3)); mas = sc * cvt3<W>(sel3(kt, ten_ma, kz3)); mbs = sc * cvt3<W>(sel3(kt, ten_mb, kz3));
    } else {
      us = sc * cvt3<W>(sel3(Ro.is_ten, ld3(M.td, TU), z3)); mas = sc * cvt3<W>(sel3(Ro.is_ten, ld3(M.td, TMA), z3));
      mbs = sc * cvt3<W>(sel3(Ro.is_ten, ld3(M.td, TMB), z3));
    }
    auto ezw = [&](I bar) { return WK ? cvt3<W>(ld3(M.kd, KEZ + 3 * bar)) : cvt3<W>(ld3(M.fd, FEZ + 3 * bar)); };
    const V3<W> was = inv_inertia(PW, ezw(Ro.ba), mas), wbs = inv_inertia(PW, ezw(Ro.bb), mbs);
    W zt = dot(us, flB - flA) + dot(mbs, faB) - dot(mas, faA);
    TGR_SCHED();
    TGR_CENSUS(13);
    W Sr[9], il9[9];
    static_for<9>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      constexpr int as = ten_ba(s), bs = ten_bb(s);
      const V3<W> u2 = bcast3<s>(us), wa2 = bcast3<s>(was), wb2 = bcast3<s>(wbs);
      const W uu = imw * dot(us, u2);
      W v = sel(Ro.r == s, W(SW(1)), wzero);
      v = v + sel(Ro.ba == as, uu + dot(mas, wa2), wzero);
      v = v + sel(Ro.bb == bs, uu + dot(mbs, wb2), wzero);
      v = v - sel(Ro.ba == bs, uu + dot(mas, wb2), wzero);
      v = v - sel(Ro.bb == as, uu + dot(mbs, wa2), wzero);
      Sr[s] = sel(Ro.is_ten, v, wzero);
    });
    TGR_SCHED();
    TGR_CENSUS(14);
    rchol<9>(Sr, il9, [](int) { return SW(0.5); });  // (the identity plus the dampers' positive semi-definite part)
    TGR_SCHED();
    W zr[9];
    rfwd<9>(Sr, il9, zt, zr);
    rbwd_rep<9>(Sr, il9, zr, zr);
    const W zd = rscatter<9>(zr);  // lane t: its tendon's multiplier (zero on the lanes without a tendon)
    TGR_CENSUS(15);
    {
      // (zd is zero on the lanes without a tendon; classes t mod 3 as in the tendon stage: bar A of a tendon gets +, bar B -)
      const W zi = zd * imw;
      const V3<W> Su = mk<W>(trisum(zi * us.x), trisum(zi * us.y), trisum(zi * us.z));
      const V3<W> Swa = mk<W>(trisum(zd * was.x), trisum(zd * was.y), trisum(zd * was.z));
      const V3<W> Swb = mk<W>(trisum(zd * wbs.x), trisum(zd * wbs.y), trisum(zd * wbs.z));
      yl[0] = bcast3v<0>(Su) + bcast3v<1>(Su);     ya[0] = bcast3v<0>(Swa) + bcast3v<1>(Swa);
      yl[1] = bcast3v<2>(Su) - bcast3v<1>(Su);     ya[1] = bcast3v<2>(Swa) - bcast3v<1>(Swb);
      yl[2] = -(bcast3v<0>(Su) + bcast3v<2>(Su));  ya[2] = -(bcast3v<0>(Swb) + bcast3v<2>(Swb));
    }
    (void)wz3;
  }
  TGR_RSTAMP(13);  // implicitfast
  TGR_PHASE(9); TGR_REARG();
  // ---- advance: v += h a ; x += h v ; q <- exp(h w / 2) * q   (world-frame w: left multiplication).  Lane b advances bar b (lanes 3..
  // repeat bar 0) and stores it: a third of the instructions of doing all three bars on every lane, most of them fp64
  {
    const Roles<F> Ro = lane_roles<F, MODEL>();
    const I bi = sel(Ro.r == 1, I(1), sel(Ro.r == 2, I(2), I(0)));
    // acceleration = unconstrained (K) + by the constraint wrench (force and transverse torque from the F sums, axial torque from K) + damping
    const V3<K> kz = ld3(M.kd, KEZ + 3 * bi);
    const V3<K> kf = cvt3<K>(ld3(M.fd, FFCA + 3 * bi));
    const V3<K> yak = PK.iIxx * kf + (M.kd.ld(KTAX + bi) * PK.iIzz - dot(kz, kf) * PK.iIxx) * kz;
    const V3<W> ylb = pick_bar(bi, yl), yab = pick_bar(bi, ya);
    const V3<K> vk = cvt3<K>(ld3(M.fd, FV + 3 * bi)) + PK.h * (WK ? (ld3(M.kd, KL0 + 3 * bi) + imk * cvt3<K>(ld3(M.fd, FFCL + 3 * bi))) + cvt3<K>(ylb)
                                                                       : ld3(M.kd, KL0 + 3 * bi) + cvt3<K>(cvt3<W>(im * ld3(M.fd, FFCL + 3 * bi)) + ylb));
    const V3<K> wk = cvt3<K>(ld3(M.fd, FW + 3 * bi)) + PK.h * ((ld3(M.kd, KA0 + 3 * bi) + yak) + cvt3<K>(yab));
    const V3<K> xk = ld3(M.kd, KX + 3 * bi) + PK.h * vk;
    const V3<K> w = wk;
    const SK hh = SK(0.5) * PK.h;
    const K x2 = (hh * hh) * dot(w, w);
    // half angle ~1e-3: series for sin(x)/x and cos(x) (exact to rounding for x < 1; spins beyond that end the episode anyway)
    const K sinc = SK(1) + x2 * (SK(-1.0 / 6) + x2 * (SK(1.0 / 120) + x2 * (SK(-1.0 / 5040) + x2 * (SK(1.0 / 362880) + x2 * SK(-1.0 / 39916800)))));
    const K cs = SK(1) + x2 * (SK(-0.5) + x2 * (SK(1.0 / 24) + x2 * (SK(-1.0 / 720) + x2 * (SK(1.0 / 40320) + x2 * (SK(-1.0 / 3628800) + x2 * SK(1.0 / 479001600))))));
    const K k = hh * sinc;
    const K rw = cs, rx = k * w.x, ry = k * w.y, rz = k * w.z;
    const K q[4] = {M.kd.ld(KQ + 4 * bi), M.kd.ld(KQ + 4 * bi + 1), M.kd.ld(KQ + 4 * bi + 2), M.kd.ld(KQ + 4 * bi + 3)};  // (normalised at the start)
    TGR_SCHED();  // (every read of the old state is above, every store below: lanes 3.. write bar 0's words a second time)
    st3_rep(M.fd, FV + 3 * bi, cvt3<F>(vk));
    st3_rep(M.fd, FW + 3 * bi, cvt3<F>(wk));
    st3_rep(M.kd, KX + 3 * bi, xk);
    M.kd.st_rep(KQ + 4 * bi, rw * q[0] - rx * q[1] - ry * q[2] - rz * q[3]);
    M.kd.st_rep(KQ + 4 * bi + 1, rw * q[1] + rx * q[0] + ry * q[3] - rz * q[2]);
    M.kd.st_rep(KQ + 4 * bi + 2, rw * q[2] - rx * q[3] + ry * q[0] + rz * q[1]);
    M.kd.st_rep(KQ + 4 * bi + 3, rw * q[3] + rx * q[2] - ry * q[1] + rz * q[0]);
    if (MODEL) {
      // actuator a on lane a (in K: the actuator gain is 15000 N per unit of activation, and twenty fp32 roundings of the activation are 1e-3 N)
      const I ai = sel(Ro.r < 6, Ro.r, I(0));
      const K ac = M.kd.ld(KACT + ai);
      M.kd.st_rep(KACT + ai, ac + PK.h * (cvt<K>(clamp_nan_(M.fd.ld(FCTRL + ai), F(P.ctrl_lo), F(P.ctrl_hi))) - ac) * PK.act_itau);
    }
  }
  TGR_RSTAMP(14);  // advance
#undef PK
#undef P
#undef im
#undef imk
#undef TGR_REARG
}
}  // namespace row
}  // namespace tgr
