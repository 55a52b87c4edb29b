// row_emul.cpp -- HOST EMULATION of the row kernels' physics (tensegrity_rl_amd/csrc/tgr_row.h compiled with TGR_ROW_EMUL: one
// "lane value" = the 16 lanes of a DPP row as an array).  TEST INFRASTRUCTURE: lets tests/test_row_emulation.py check the
// sub-step against the CPU oracle in the build container, where there is no GPU.  Nothing in the product links this.
#define TGR_ROW_EMUL 1
#include "../../tensegrity_rl_amd/csrc/tgr_row.h"

#include <cstdio>

using namespace tgr::row;
static int g_warm_start = 0;
extern "C" void tgr_row_emul_set_warm_start(int w) { g_warm_start = w; }

template <typename T> static void fill_params(const double* p, RParams<T>& P, const float* hf, int nrow, int ncol) {
  int i = 0;
  P.h = T(p[i++]); P.grav = T(p[i++]); P.mass = T(p[i++]); P.Ixx = T(p[i++]); P.Izz = T(p[i++]);
  P.cap_r = T(p[i++]); P.bar_half = T(p[i++]); P.rcyl_r = T(p[i++]); P.bcyl_r = T(p[i++]); P.bcyl_lo = T(p[i++]); P.bcyl_hi = T(p[i++]);
  P.site_off = T(p[i++]); P.site_z = T(p[i++]);
  for (int t = 0; t < 9; t++) P.ten_k[t] = T(p[i++]);
  for (int t = 0; t < 9; t++) P.ten_b[t] = T(p[i++]);
  for (int t = 0; t < 9; t++) P.ten_L0[t] = T(p[i++]);
  P.gain = T(p[i++]); P.bias0 = T(p[i++]); P.bias1 = T(p[i++]); P.bias2 = T(p[i++]); P.force_lo = T(p[i++]); P.force_hi = T(p[i++]);
  P.ctrl_lo = T(p[i++]); P.ctrl_hi = T(p[i++]); P.K = T(p[i++]); P.B = T(p[i++]);
  P.imp0 = T(p[i++]); P.imp1 = T(p[i++]); P.imp_width = T(p[i++]); P.imp_mid = T(p[i++]);
  P.f0 = T(p[i++]); P.f1sq = T(p[i++]); P.f3sq = T(p[i++]); P.f4sq = T(p[i++]); P.impratio = T(p[i++]); P.invweight = T(p[i++]);
  P.tol = T(p[i++]); P.rtol = T(p[i++]); P.ls_tol = T(p[i++]); P.cf_lo = T(p[i++]); P.cf_hi = T(p[i++]);
  P.max_iter = (int)p[i++]; P.ls_iter = (int)p[i++]; P.skip_clamped = (int)p[i++]; P.act_itau = T(p[i++]);
  P.warm_start = 0;
  P.hf_rx = T(p[i++]); P.hf_ry = T(p[i++]); P.hf_ez = T(p[i++]); P.hf_x0 = T(p[i++]); P.hf_y0 = T(p[i++]); P.hf_z0 = T(p[i++]);
  P.hf = hf; P.hf_nrow = nrow; P.hf_ncol = ncol;
  P.hf_dx = T(ncol > 1 ? 2 * p[i - 6] / (ncol - 1) : 1.0); P.hf_dy = T(nrow > 1 ? 2 * p[i - 5] / (nrow - 1) : 1.0);
  P.derive();
}

// every lane of the row must hold the same bits of a replicated value
template <typename T> static int check_rep(const Ln<T>& x, const char* what, int env) {
  for (int i = 1; i < ROW; i++)
    if (std::memcmp(&x.v[i], &x.v[0], sizeof(T)) != 0 && !(x.v[i] != x.v[i] && x.v[0] != x.v[0])) {
      std::fprintf(stderr, "row_emul: replicated value %s differs across lanes (env %d lane %d: %.17g vs %.17g)\n", what, env, i, (double)x.v[i], (double)x.v[0]);
      return 1;
    }
  return 0;
}

template <typename TK, typename T, int MODEL>
static int run(const double* params, const float* hf, int nrow, int ncol, int n, int nsub, double* qpos, double* qvel, const double* ctrl, double* act,
               double* out /* [n][32] */, unsigned* trace /* [n][TRACE_WORDS] or null */, const double* capd /* [n] or null */) {
  typedef Ln<T> F;
  typedef Ln<TK> K;
  RParams<T> P;
  RParams<TK> PK;
  fill_params<T>(params, P, hf, nrow, ncol);
  fill_params<TK>(params, PK, hf, nrow, ncol);
  P.warm_start = PK.warm_start = g_warm_start;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : bad)
  for (int e = 0; e < n; e++) {
    K kd[K_SLAB], rd[R_SLAB], ad[A_SLAB * n_pages<MODEL>()];
    F hd[H_SLAB], lp[LP_SLAB];
    K lc[LC_SLAB];
    F xs[XS_SLAB];
    F fd[f_slab<MODEL>()], td[T_SLAB], cd[C_SLAB * n_pages<MODEL>()];
    RMem<K, F> M;
    M.kd = Slab<K>{kd, 1, 0}; M.fd = Slab<F>{fd, 1, 0}; M.td = Slab<F>{td, 1, 0}; M.rd = Slab<K>{rd, 1, 0}; M.ad = Slab<K>{ad, 1, 0}; M.cd = Slab<F>{cd, 1, 0}; M.hd = Slab<F>{hd, 1, 0}; M.lp = Slab<F>{lp, 1, 0}; M.lc = Slab<K>{lc, 1, 0}; M.xs = Slab<F>{xs, 1, 0};
    M.dbg = trace ? trace + (size_t)TRACE_WORDS * e : nullptr;
    store_role_consts<K, F, TK, MODEL>(PK, M.rd);
    RLag<K, F> lag;
    for (int b = 0; b < 3; b++) {
      const double* q = qpos + 21 * e + 7 * b;
      K qq[4];
      for (int i = 0; i < 3; i++) kd[KX + 3 * b + i] = K(TK(q[i]));
      for (int i = 0; i < 4; i++) qq[i] = K(TK(q[3 + i]));
      normalize_quat(qq);
      for (int i = 0; i < 4; i++) kd[KQ + 4 * b + i] = qq[i];
      const Rot<K> R = quat_rot(qq);
      const double* v = qvel + 18 * e + 6 * b;
      for (int i = 0; i < 3; i++) fd[FV + 3 * b + i] = F(T(v[i]));
      st3(M.fd, FW + 3 * b, cvt3<F>(rot_apply(R, mk<K>(K(TK(v[3])), K(TK(v[4])), K(TK(v[5]))))));
    }
    for (int i = 0; i < 6; i++) { fd[FCTRL + i] = F(T(ctrl[6 * e + i])); kd[KACT + i] = K(TK(act[6 * e + i])); }
    for (int i = 0; i < 18; i++) fd[FDL + i] = F(T(0));
    fd[FCAPD] = F(T(capd ? capd[e] : 0.0));
    for (int i = 0; i < H_SLAB; i++) hd[i] = F(T(0));
    lag.n_it = 0; lag.n_ls = 0; lag.n_unconv = F(T(0)); lag.n_own = F(T(0)); lag.cfrc_absmax = F(T(0)); lag.cfrc_clip_sq = F(T(0)); lag.bar_contact = F(T(0)); lag.overflow = F(T(0));
    const ArgRef<RParams<TK>> pkr{&PK};
    const ArgRef<RParams<T>> pr{&P};
    for (int s = 0; s + 1 < nsub; s++) substep<K, F, TK, T, MODEL, false>(pkr, pr, M, lag, s);
    substep<K, F, TK, T, MODEL, true>(pkr, pr, M, lag, nsub - 1);
    for (int b = 0; b < 3; b++) {
      K qq[4] = {kd[KQ + 4 * b], kd[KQ + 4 * b + 1], kd[KQ + 4 * b + 2], kd[KQ + 4 * b + 3]};
      normalize_quat(qq);
      const Rot<K> R = quat_rot(qq);
      const V3<K> wb = rot_applyT(R, cvt3<K>(ld3(M.fd, FW + 3 * b)));
      double* q = qpos + 21 * e + 7 * b;
      double* v = qvel + 18 * e + 6 * b;
      bad += check_rep(kd[KX + 3 * b], "x", e) + check_rep(qq[0], "q", e) + check_rep(fd[FV + 3 * b + 2], "v", e) + check_rep(wb.x, "w", e);
      for (int i = 0; i < 3; i++) q[i] = kd[KX + 3 * b + i].v[0];
      for (int i = 0; i < 4; i++) q[3 + i] = qq[i].v[0];
      for (int i = 0; i < 3; i++) v[i] = fd[FV + 3 * b + i].v[0];
      v[3] = wb.x.v[0]; v[4] = wb.y.v[0]; v[5] = wb.z.v[0];
    }
    for (int i = 0; i < 6; i++) act[6 * e + i] = kd[KACT + i].v[0];
    double* o = out + 32 * e;
    for (int t = 0; t < 9; t++) { o[t] = lp[LPT + t].v[0]; bad += check_rep(lp[LPT + t], "tenL", e); }
    for (int i = 0; i < 9; i++) o[9 + i] = (double)lp[LPX + i].v[0] + (double)lc[LCX + i % 3].v[0];  // (lp holds the lagged positions relative to their centre lc)
    for (int i = 0; i < 3; i++) { o[26 + i] = (double)lc[LCX + i].v[0]; bad += check_rep(lc[LCX + i], "lagged centre", e); }
    o[18] = lag.cfrc_absmax.v[0]; o[19] = lag.cfrc_clip_sq.v[0]; o[20] = lag.bar_contact.v[0]; o[21] = lag.overflow.v[0];
    o[22] = lag.n_unconv.v[0]; o[23] = lag.n_it; o[24] = lag.n_ls; o[25] = lag.n_own.v[0];
    bad += check_rep(lag.cfrc_absmax, "cfrc", e) + check_rep(lag.bar_contact, "bar_contact", e) + check_rep(lag.n_unconv, "n_unconv", e);
  }
  return bad;
}

#ifdef TGR_ROW_OPCOUNT
extern "C" void tgr_row_opcount(long long* out /* [2][24] */, int reset) {
  OpCount& c = opcount();
  for (int p = 0; p < 2; p++) for (int i = 0; i < 24; i++) { out[p * 24 + i] = c.n[p][i]; if (reset) c.n[p][i] = 0; }
}
#endif
// precision: 0 = fp32 throughout, 1 = fp64 throughout, 2 = mixed (pose / kinematics / tendons fp64, the rest fp32)
extern "C" int tgr_row_emul(int precision, int model, const double* params, const float* hf, int nrow, int ncol, int n, int nsub, double* qpos,
                            double* qvel, const double* ctrl, double* act, double* out, unsigned* trace, const double* capd) {
  if (model == 0) {
    if (precision == 1) return run<double, double, 0>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
    if (precision == 2) return run<double, float, 0>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
    return run<float, float, 0>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
  }
  if (precision == 1) return run<double, double, 1>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
  if (precision == 2) return run<double, float, 1>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
  return run<float, float, 1>(params, hf, nrow, ncol, n, nsub, qpos, qvel, ctrl, act, out, trace, capd);
}
