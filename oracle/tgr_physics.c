/*
 * tgr_physics.c -- fp64 CPU restatement of one MuJoCo 2.3.7 sub-step for the
 * 3-bar / 9-tendon tensegrity model (3prism_jonathan_steady_side.xml).
 *
 * TEST INFRASTRUCTURE (see tgr_oracle.h).  Deliberately written the plain way:
 * dense 18-dof vectors, dense 6x18 contact Jacobians, dense 18x18 Cholesky --
 * it shares no code with the HIP kernels it checks.
 *
 * Pipeline restated (mujoco==2.3.7, requirements.txt:5; call sites
 * tensegrity_env.py:297,448,484 / tr_env.py:346,744,800,812):
 *   mj_step = mj_forward ; implicitfast velocity solve ; semi-implicit position update
 *   mj_forward = kinematics -> tendon length/Jacobian -> collision -> constraint rows
 *              -> tendon velocity, passive, bias, aref -> actuation -> qacc_smooth
 *              -> Newton solve of the convex contact problem (elliptic cones)
 *   mj_rnePostConstraint -> cfrc_ext
 */
#include "tgr_oracle.h"

#include <math.h>
#include <string.h>

#define MINVAL 1e-15

/* ------------------------------------------------------------------ small vector helpers */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static void quat2mat(double* R, const double* q) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
static void mulquat(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void normquat(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* ------------------------------------------------------------------ model tables
 * sites per bar (xml:82-87,100-105,118-123): identical local layout on the three bars */
static const double SITE_SGN[6][3] = {
    {-1, 0, +1}, /* h02 h24 h40 */
    {0, +1, +1}, /* h04 h20 h42 */
    {+1, 0, +1}, /* h03 h25 h41 */
    {0, +1, -1}, /* h14 h30 h52 */
    {+1, 0, -1}, /* h13 h35 h51 */
    {0, -1, -1}, /* h15 h31 h53 */
};
/* tendons (xml:127-164): {bar_a, site_a, bar_b, site_b} */
static const int TEN[TGO_NTEN][4] = {
    {0, 1, 2, 0}, /* td_0 h04-h40 */
    {0, 0, 1, 1}, /* td_1 h02-h20 */
    {1, 0, 2, 1}, /* td_2 h24-h42 */
    {0, 5, 2, 4}, /* td_3 h15-h51 */
    {0, 4, 1, 5}, /* td_4 h13-h31 */
    {1, 4, 2, 5}, /* td_5 h35-h53 */
    {0, 3, 2, 2}, /* td_6 h14-h41 */
    {0, 2, 1, 3}, /* td_7 h03-h30 */
    {1, 2, 2, 3}, /* td_8 h25-h52 */
};

/* table access for tooling (oracle/mjcf.py emits an MJCF from these constants) */
void tgo_site_sign(int s, double out[3]) { for (int i = 0; i < 3; i++) out[i] = SITE_SGN[s][i]; }
void tgo_tendon_ends(int t, int out[4]) { for (int i = 0; i < 4; i++) out[i] = TEN[t][i]; }

void tgo_default_model(tgo_model* m) {
  memset(m, 0, sizeof(*m));
  m->timestep = 0.001;
  m->gravity = 9.81;
  m->cap_radius = 0.0675;
  m->bar_half = 0.688;
  m->rcyl_radius = 0.0381;
  m->bcyl_radius = 0.035;
  m->bcyl_lo = 0.238;
  m->bcyl_hi = 0.688;
  m->site_off = 0.0675;
  m->site_z = 0.668;
  /* inertiafromgeom (xml:9) with explicit geom masses: r01 1.0, s0/s1 0.5, b0/b1 1.0 (xml:72-78) */
  {
    double mr = 1.0, ms = 0.5, mb = 1.0;
    double L = 2 * m->bar_half, r = m->rcyl_radius;
    double Ixx = mr * (3 * r * r + L * L) / 12.0, Izz = mr * r * r / 2.0;
    double rs = m->cap_radius;
    double Is = 0.4 * ms * rs * rs;
    Ixx += 2 * (Is + ms * m->bar_half * m->bar_half);
    Izz += 2 * Is;
    double Lb = m->bcyl_hi - m->bcyl_lo, rb = m->bcyl_radius, cb = 0.5 * (m->bcyl_hi + m->bcyl_lo);
    Ixx += 2 * (mb * (3 * rb * rb + Lb * Lb) / 12.0 + mb * cb * cb);
    Izz += 2 * (mb * rb * rb / 2.0);
    m->mass = mr + 2 * ms + 2 * mb;
    m->inertia[0] = Ixx; m->inertia[1] = Ixx; m->inertia[2] = Izz;
    m->invweight_tran = 1.0 / m->mass; /* body_invweight0 of a free body */
  }
  for (int t = 0; t < TGO_NTEN; t++) {
    m->ten_stiffness[t] = t >= 6 ? 700.0 : 0.0; /* xml:152-163 */
    m->ten_damping[t] = t >= 6 ? 100.0 : 0.0;
    m->ten_springlen[t] = 0.8;
  }
  m->gain = 6667.0; m->bias0 = 3290.0; m->bias1 = -6666.0; m->bias2 = -133.0; /* xml:39 */
  m->force_lo = -267.0; m->force_hi = 0.0; m->ctrl_lo = -0.45; m->ctrl_hi = 0.15;
  m->solref[0] = -5000; m->solref[1] = -500; /* xml:30 */
  m->solimp[0] = 0.9; m->solimp[1] = 0.95; m->solimp[2] = 0.001; m->solimp[3] = 0.5; m->solimp[4] = 2;
  /* contact friction = max of the two geoms, expanded to (slide,slide,torsion,roll,roll) */
  m->friction[0] = 1; m->friction[1] = 1; m->friction[2] = 0.005; m->friction[3] = 1e-4; m->friction[4] = 1e-4;
  m->impratio = 1;
  m->actuator_deriv_skip_clamped = 0;
  m->newton_maxiter = 100;
  m->model_kind = 0;
  for (int s_ = 0; s_ < 6; s_++) {
    m->site_local[s_][0] = SITE_SGN[s_][0] * m->site_off; m->site_local[s_][1] = SITE_SGN[s_][1] * m->site_off;
    m->site_local[s_][2] = SITE_SGN[s_][2] * m->site_z;
  }
  memcpy(m->ten_ends, TEN, sizeof(TEN));
  m->act_filter = 0; m->act_tau = 1.0;
  m->hf_data = NULL; m->hf_nrow = m->hf_ncol = 0;
  {
    static const double q0[TGO_NQ] = {/* xml:71,90,108 */
        0.08369179, -0.28792231, 0.24830847, -0.49145555, 0.7539914, -0.27511722, -0.33805166,
        0.14497616, -0.19291743, 0.35052097, -0.84766041, 0.27950622, 0.45085889, 0.00862359,
        0.04557825, -0.29876206, 0.39531985, -0.35798606, -0.47531391, 0.72471075, 0.34744352};
    memcpy(m->qpos0, q0, sizeof(q0));
  }
}

/* 3prism_jonathan_steady_side_uneven_ground.xml.  Differences to the flat model: caps r = 0.05 (xml:69-70),
 * tendon sites at 0.05 from the axis in a different arrangement (xml:76-81,94-99,112-117), all nine tendons passive
 * spring-dampers k = 1e4, b = 1e3 with rest length 0.5 / 0.8 (xml:38,122-158), actuators dyntype=filter (tau 1 s) with
 * gain 15000, no bias, no force limit (xml:39), other initial pose (xml:65,84,102), heightfield floor of radius 50 m,
 * elevation 1 m, placed at z = -1 (xml:48,56). */
void tgo_uneven_model(tgo_model* m, const float* heights, int nrow, int ncol) {
  tgo_default_model(m);
  m->model_kind = 1;
  m->cap_radius = 0.05;
  m->site_off = 0.05;
  { /* inertia with the smaller caps */
    double mr = 1.0, ms = 0.5, mb = 1.0;
    double L = 2 * m->bar_half, r = m->rcyl_radius, rs = m->cap_radius;
    double Ixx = mr * (3 * r * r + L * L) / 12.0, Izz = mr * r * r / 2.0, Is = 0.4 * ms * rs * rs;
    Ixx += 2 * (Is + ms * m->bar_half * m->bar_half); Izz += 2 * Is;
    double Lb = m->bcyl_hi - m->bcyl_lo, rb = m->bcyl_radius, cb = 0.5 * (m->bcyl_hi + m->bcyl_lo);
    Ixx += 2 * (mb * (3 * rb * rb + Lb * Lb) / 12.0 + mb * cb * cb); Izz += 2 * (mb * rb * rb / 2.0);
    m->inertia[0] = Ixx; m->inertia[1] = Ixx; m->inertia[2] = Izz;
  }
  /* the six site positions of a bar, in the order used by ten_ends below: (-x,+z) (+x,+z) (-y,+z) (-y,-z) (+y,-z) (+x,-z) */
  static const double sg[6][3] = {{-1, 0, 1}, {1, 0, 1}, {0, -1, 1}, {0, -1, -1}, {0, 1, -1}, {1, 0, -1}};
  for (int s_ = 0; s_ < 6; s_++) {
    m->site_local[s_][0] = sg[s_][0] * m->site_off; m->site_local[s_][1] = sg[s_][1] * m->site_off; m->site_local[s_][2] = sg[s_][2] * m->site_z;
  }
  /* site names -> index per bar: r01: h04 h03 h02 h14 h13 h15 ; r23: h24 h20 h25 h31 h30 h35 ; r45: h41 h40 h42 h51 h53 h52 */
  static const int te[TGO_NTEN][4] = {
      {0, 0, 2, 1}, /* td_0 h04-h40 */ {0, 2, 1, 1}, /* td_1 h02-h20 */ {1, 0, 2, 2}, /* td_2 h24-h42 */
      {0, 5, 2, 3}, /* td_3 h15-h51 */ {0, 4, 1, 3}, /* td_4 h13-h31 */ {1, 5, 2, 4}, /* td_5 h35-h53 */
      {0, 3, 2, 0}, /* td_6 h14-h41 */ {0, 1, 1, 4}, /* td_7 h03-h30 */ {1, 2, 2, 5}, /* td_8 h25-h52 */
  };
  memcpy(m->ten_ends, te, sizeof(te));
  for (int t = 0; t < TGO_NTEN; t++) {
    m->ten_stiffness[t] = 10000.0; m->ten_damping[t] = 1000.0; m->ten_springlen[t] = t < 6 ? 0.5 : 0.8;
  }
  m->act_filter = 1; m->act_tau = 1.0;
  m->gain = 15000.0; m->bias0 = m->bias1 = m->bias2 = 0.0;
  m->force_lo = -1e300; m->force_hi = 1e300; /* forcelimited not set */
  {
    static const double q0[TGO_NQ] = {/* uneven xml:65,84,102 */
        0.25551711, -0.00069342, 0.22404039, -0.49720971, 0.24315431, 0.75327284, -0.35530059,
        0.14409445, 0.0654207, 0.33662589, 0.42572066, 0.01379464, -0.53972521, 0.72613244,
        0.28544944, -0.04883333, 0.38591159, 0.137357, 0.06898275, -0.85996553, 0.48665565};
    memcpy(m->qpos0, q0, sizeof(q0));
  }
  m->hf_data = heights; m->hf_nrow = nrow; m->hf_ncol = ncol;
  m->hf_size[0] = 50; m->hf_size[1] = 50; m->hf_size[2] = 1; m->hf_size[3] = 0.1;
  m->hf_pos[0] = 0; m->hf_pos[1] = 0; m->hf_pos[2] = -1.0;
}

void tgo_reset_data(const tgo_model* m, tgo_data* d) {
  memset(d, 0, sizeof(*d));
  memcpy(d->qpos, m->qpos0, sizeof(d->qpos));
}

/* ------------------------------------------------------------------ kinematics + tendons */
static void kinematics(const tgo_model* m, tgo_data* d) {
  for (int b = 0; b < TGO_NBAR; b++) {
    double* q = d->qpos + 7 * b;
    normquat(q + 3); /* mj_kinematics normalises the free-joint quaternion */
    memcpy(d->xpos[b], q, 3 * sizeof(double));
    quat2mat(d->xmat[b], q + 3);
    const double* R = d->xmat[b];
    for (int s = 0; s < 6; s++) {
      const double* l = m->site_local[s];
      for (int i = 0; i < 3; i++) d->site_xpos[b][s][i] = q[i] + R[3 * i] * l[0] + R[3 * i + 1] * l[1] + R[3 * i + 2] * l[2];
    }
    for (int e = 0; e < 2; e++) { /* caps s(2b) at +z, s(2b+1) at -z */
      double z = e == 0 ? m->bar_half : -m->bar_half;
      for (int i = 0; i < 3; i++) d->cap_xpos[2 * b + e][i] = q[i] + R[3 * i + 2] * z;
    }
  }
}

/* translational Jacobian-transpose action: generalised force of world force f applied at world point p of bar b */
static void add_point_force(const tgo_data* d, int b, const double* p, const double* f, double scale, double* qfrc) {
  const double* R = d->xmat[b];
  double r[3] = {p[0] - d->xpos[b][0], p[1] - d->xpos[b][1], p[2] - d->xpos[b][2]};
  double t[3];
  cross3(t, r, f);
  for (int i = 0; i < 3; i++) qfrc[6 * b + i] += scale * f[i];
  for (int k = 0; k < 3; k++) qfrc[6 * b + 3 + k] += scale * (R[k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]);
}

static void tendons(const tgo_model* m, tgo_data* d) {
  memset(d->ten_J, 0, sizeof(d->ten_J));
  for (int t = 0; t < TGO_NTEN; t++) {
    int ba = m->ten_ends[t][0], bb = m->ten_ends[t][2];
    const double* pa = d->site_xpos[ba][m->ten_ends[t][1]];
    const double* pb = d->site_xpos[bb][m->ten_ends[t][3]];
    double dif[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    double len = norm3(dif);
    d->ten_length[t] = len;
    if (len < MINVAL) continue;
    for (int i = 0; i < 3; i++) dif[i] /= len;
    /* J = (jac_b - jac_a)^T dif : same operator as a unit force along dif at b and -dif at a */
    add_point_force(d, bb, pb, dif, +1.0, d->ten_J[t]);
    add_point_force(d, ba, pa, dif, -1.0, d->ten_J[t]);
  }
}

/* ------------------------------------------------------------------ collision */
static void make_frame(double* f) {
  /* mju_makeFrame: f[0:3] normal given; build tangents */
  double n = norm3(f);
  for (int i = 0; i < 3; i++) f[i] /= n;
  f[3] = f[4] = f[5] = 0;
  if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  double dd = dot3(f, f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] -= dd * f[i];
  n = norm3(f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] /= n;
  cross3(f + 6, f, f + 3);
}

static void add_contact(tgo_data* d, const double* pos, const double* normal, double dist, int b1, int b2, int g1, int g2) {
  if (d->ncon >= TGO_MAXCON) { d->flags |= TGO_FLAG_CON_OVERFLOW; return; }
  tgo_contact* c = &d->con[d->ncon++];
  memset(c, 0, sizeof(*c));
  memcpy(c->pos, pos, 3 * sizeof(double));
  memcpy(c->frame, normal, 3 * sizeof(double));
  make_frame(c->frame);
  c->dist = dist;
  c->body1 = b1; c->body2 = b2; c->geom1 = g1; c->geom2 = g2;
}

/* closest points of two segments c+s*a (|s|<=ha), c2+t*a2 (|t|<=hb); returns whether both params are interior */
static int seg_seg(const double* ca, const double* aa, double ha, const double* cb, const double* ab, double hb,
                   double* s_out, double* t_out) {
  double w[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
  double b = dot3(aa, ab), dA = dot3(aa, w), dB = dot3(ab, w);
  double den = 1.0 - b * b;
  double s, t;
  int interior = 0;
  if (den > 1e-12) {
    s = (b * dB - dA) / den;
    t = (dB - b * dA) / den;
    if (fabs(s) < ha && fabs(t) < hb) interior = 1;
  } else {
    s = 0; t = dB;
  }
  if (!interior) {
    /* clamp s, recompute t, clamp t, recompute s (Ericson 5.1.9) */
    if (s > ha) s = ha; else if (s < -ha) s = -ha;
    t = dB + b * s;
    if (t > hb) t = hb; else if (t < -hb) t = -hb;
    s = b * t - dA;
    if (s > ha) s = ha; else if (s < -ha) s = -ha;
  }
  *s_out = s; *t_out = t;
  return interior;
}

typedef struct { int type; /* 0 sphere 1 cylinder */ double c[3]; double a[3]; double r, h; int geom; } bgeom;

static void bar_geoms(const tgo_model* m, const tgo_data* d, int b, bgeom* g) {
  const double* R = d->xmat[b];
  double a[3] = {R[2], R[5], R[8]};
  double off[5] = {0, m->bar_half, -m->bar_half, 0.5 * (m->bcyl_hi + m->bcyl_lo), -0.5 * (m->bcyl_hi + m->bcyl_lo)};
  for (int k = 0; k < 5; k++) {
    g[k].geom = 1 + 5 * b + k;
    for (int i = 0; i < 3; i++) { g[k].c[i] = d->xpos[b][i] + off[k] * a[i]; g[k].a[i] = a[i]; }
  }
  g[0].type = 1; g[0].r = m->rcyl_radius; g[0].h = m->bar_half;
  g[1].type = 0; g[1].r = m->cap_radius; g[1].h = 0;
  g[2].type = 0; g[2].r = m->cap_radius; g[2].h = 0;
  g[3].type = 1; g[3].r = m->bcyl_radius; g[3].h = 0.5 * (m->bcyl_hi - m->bcyl_lo);
  g[4].type = 1; g[4].r = m->bcyl_radius; g[4].h = 0.5 * (m->bcyl_hi - m->bcyl_lo);
}

static void sphere_sphere(tgo_data* d, const bgeom* s1, int b1, const bgeom* s2, int b2) {
  double dif[3] = {s2->c[0] - s1->c[0], s2->c[1] - s1->c[1], s2->c[2] - s1->c[2]};
  double cd = norm3(dif);
  if (cd > s1->r + s2->r) return;
  double dist = cd - s1->r - s2->r;
  double n[3] = {1, 0, 0};
  if (cd > MINVAL) { n[0] = dif[0] / cd; n[1] = dif[1] / cd; n[2] = dif[2] / cd; } else d->flags |= TGO_FLAG_DEEP;
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = s1->c[i] + n[i] * (s1->r + 0.5 * dist);
  if (dist < 0) add_contact(d, pos, n, dist, b1, b2, s1->geom, s2->geom);
}

/* sphere (geom1) vs solid finite cylinder (geom2): exact closest feature for a centre outside the cylinder */
static void sphere_cyl(tgo_data* d, const bgeom* s, int bs, const bgeom* c, int bc) {
  double w[3] = {s->c[0] - c->c[0], s->c[1] - c->c[1], s->c[2] - c->c[2]};
  double t = dot3(w, c->a);
  double rad[3] = {w[0] - t * c->a[0], w[1] - t * c->a[1], w[2] - t * c->a[2]};
  double rho = norm3(rad);
  double tc = t > c->h ? c->h : (t < -c->h ? -c->h : t);
  double rc = rho > c->r ? c->r : rho;
  double q[3];
  if (fabs(t) <= c->h && rho <= c->r) {
    /* centre inside the cylinder: leave through the side wall */
    d->flags |= TGO_FLAG_DEEP;
    if (rho < MINVAL) return;
    double dist = rho - c->r - s->r;
    double n[3] = {-rad[0] / rho, -rad[1] / rho, -rad[2] / rho};
    double pos[3];
    for (int i = 0; i < 3; i++) pos[i] = s->c[i] + n[i] * (s->r + 0.5 * dist);
    add_contact(d, pos, n, dist, bs, bc, s->geom, c->geom);
    return;
  }
  for (int i = 0; i < 3; i++) q[i] = c->c[i] + tc * c->a[i] + (rho > MINVAL ? rad[i] / rho * rc : 0.0);
  double dif[3] = {q[0] - s->c[0], q[1] - s->c[1], q[2] - s->c[2]};
  double dd = norm3(dif);
  double dist = dd - s->r;
  if (dist >= 0 || dd < MINVAL) return;
  double n[3] = {dif[0] / dd, dif[1] / dd, dif[2] / dd};
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = s->c[i] + n[i] * (s->r + 0.5 * dist);
  add_contact(d, pos, n, dist, bs, bc, s->geom, c->geom);
}

/* cylinder vs cylinder: side-side only (closest points interior to both axes).  Outer ends of the r-cylinders are
 * enveloped by the cap spheres and the b-cylinders lie radially inside the r-cylinder, so rim contacts need
 * centimetre-deep penetration; they are not generated (DESIGN.md "bar-bar contact rule"). */
static void cyl_cyl(tgo_data* d, const bgeom* A, int bA, const bgeom* B, int bB) {
  double s, t;
  if (!seg_seg(A->c, A->a, A->h, B->c, B->a, B->h, &s, &t)) return;
  double pa[3], pb[3], dif[3];
  for (int i = 0; i < 3; i++) { pa[i] = A->c[i] + s * A->a[i]; pb[i] = B->c[i] + t * B->a[i]; dif[i] = pb[i] - pa[i]; }
  double dd = norm3(dif);
  double dist = dd - A->r - B->r;
  if (dist >= 0) return;
  if (dd < 1e-9) { d->flags |= TGO_FLAG_DEEP; return; }
  double n[3] = {dif[0] / dd, dif[1] / dd, dif[2] / dd};
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = pa[i] + n[i] * (A->r + 0.5 * dist);
  add_contact(d, pos, n, dist, bA, bB, A->geom, B->geom);
}

/* closest point of triangle (a,b,c) to p (Ericson, Real-Time Collision Detection 5.1.5) */
static void closest_on_triangle(const double* p, const double* a, const double* b, const double* c, double* out) {
  double ab[3], ac[3], ap[3], bp[3], cp[3];
  for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(out, a, 24); return; }
  for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(out, b, 24); return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i]; return; }
  for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(out, c, 24); return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i]; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
    return;
  }
  double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
  for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
}

/* Cap sphere against the heightfield floor.  MuJoCo (mjc_ConvexHField [ext]) walks the grid cells under the geom's bounding
 * box, turns every grid triangle into a prism and runs its convex penetration test on each; for a sphere that is shallowly
 * in contact this yields, per triangle, the closest point of the triangle, the direction from it to the centre as normal and
 * radius - distance as depth.  Restated analytically with two simplifications that DESIGN.md records: only the two
 * deepest triangles of a cap are kept, and the cell diagonal runs from (c, r+1) to (c+1, r) (the order in which mujoco
 * strips the vertices of a grid row).  Cylinders are not tested against the terrain. */
static void hfield_caps(const tgo_model* m, tgo_data* d) {
  const int nr = m->hf_nrow, nc = m->hf_ncol;
  const double rx = m->hf_size[0], ry = m->hf_size[1], ez = m->hf_size[2];
  const double dx = 2 * rx / (nc - 1), dy = 2 * ry / (nr - 1), rad = m->cap_radius;
  for (int k = 0; k < 6; k++) {
    const double* cw = d->cap_xpos[k];
    double p[3] = {cw[0] - m->hf_pos[0], cw[1] - m->hf_pos[1], cw[2] - m->hf_pos[2]};
    if (p[0] < -rx - rad || p[0] > rx + rad || p[1] < -ry - rad || p[1] > ry + rad || p[2] > ez + rad) continue;
    int c0 = (int)floor((p[0] - rad + rx) / dx), c1 = (int)floor((p[0] + rad + rx) / dx);
    int r0 = (int)floor((p[1] - rad + ry) / dy), r1 = (int)floor((p[1] + rad + ry) / dy);
    if (c0 < 0) c0 = 0;
    if (r0 < 0) r0 = 0;
    if (c1 > nc - 2) c1 = nc - 2;
    if (r1 > nr - 2) r1 = nr - 2;
    /* the two deepest triangles touched (MuJoCo emits one contact per prism the sphere penetrates; two is what a cap resting
     * in a fold of the terrain needs, and what it gets on the shared diagonal of a flat cell) */
    double bd[2] = {0, 0}, bq[2][3] = {{0, 0, 0}, {0, 0, 0}}; int have = 0;
    int bf[2] = {0, 0}; /* which facet: (col * 128 + row) * 2 + tri */
    for (int r = r0; r <= r1; r++)
      for (int c = c0; c <= c1; c++) {
        double v[4][3]; /* (c,r) (c,r+1) (c+1,r) (c+1,r+1) */
        for (int j = 0; j < 4; j++) {
          int cc = c + (j >> 1), rr = r + (j & 1);
          v[j][0] = cc * dx - rx; v[j][1] = rr * dy - ry; v[j][2] = (double)m->hf_data[rr * nc + cc] * ez;
        }
        for (int tri = 0; tri < 2; tri++) {
          double q[3];
          closest_on_triangle(p, v[tri], v[tri + 1], v[tri + 2], q);
          double dif[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
          double dist = norm3(dif) - rad;
          if (!(dist < 0)) continue;
          const int fid = (c * 128 + r) * 2 + tri;
          if (have < 1 || dist < bd[0]) {
            bd[1] = bd[0]; memcpy(bq[1], bq[0], 24); bf[1] = bf[0];
            bd[0] = dist; memcpy(bq[0], q, 24); bf[0] = fid;
            have = have < 2 ? have + 1 : 2;
          } else if (have < 2 || dist < bd[1]) {
            bd[1] = dist; memcpy(bq[1], q, 24); bf[1] = fid;
            have = 2;
          }
        }
      }
    for (int j = 0; j < have; j++) {
      double n[3] = {p[0] - bq[j][0], p[1] - bq[j][1], p[2] - bq[j][2]};
      double nn = norm3(n);
      if (nn < MINVAL) { d->flags |= TGO_FLAG_DEEP; continue; }
      for (int i = 0; i < 3; i++) n[i] /= nn;
      double pos[3];
      for (int i = 0; i < 3; i++) pos[i] = cw[i] - n[i] * (rad + 0.5 * bd[j]);
      add_contact(d, pos, n, bd[j], -1, k / 2, 0, 1 + 5 * (k / 2) + 1 + (k % 2));
      d->con[d->ncon - 1].feature = bf[j];
    }
  }
}

/* The r-cylinder of a bar against the heightfield.  A thin rod over piecewise-planar terrain touches it either with an end --
 * which the cap sphere around that end reaches first -- or where it crosses a convex ridge, i.e. a grid edge.  So the bar is
 * tested, as a capsule of the cylinder's radius around its axis, against the grid edges under its bounding box (cell sides and
 * the strip diagonal); the deepest one is kept (one contact per bar).  MuJoCo runs its convex test on every prism under the
 * geom instead; the b-cylinders (radially inside the r-cylinder) are not tested. */
static void hfield_bars(const tgo_model* m, tgo_data* d) {
  const int nr = m->hf_nrow, nc = m->hf_ncol;
  const double rx = m->hf_size[0], ry = m->hf_size[1], ez = m->hf_size[2];
  const double dx = 2 * rx / (nc - 1), dy = 2 * ry / (nr - 1), rad = m->rcyl_radius, H = m->bar_half;
  for (int b = 0; b < TGO_NBAR; b++) {
    const double* R = d->xmat[b];
    double a[3] = {R[2], R[5], R[8]};
    double c[3] = {d->xpos[b][0] - m->hf_pos[0], d->xpos[b][1] - m->hf_pos[1], d->xpos[b][2] - m->hf_pos[2]};
    double ex = fabs(a[0]) * H + rad, ey = fabs(a[1]) * H + rad;
    if (c[0] + ex < -rx || c[0] - ex > rx || c[1] + ey < -ry || c[1] - ey > ry) continue;
    if (c[2] - fabs(a[2]) * H - rad > ez) continue;
    int c0 = (int)floor((c[0] - ex + rx) / dx), c1 = (int)floor((c[0] + ex + rx) / dx) + 1;
    int r0 = (int)floor((c[1] - ey + ry) / dy), r1 = (int)floor((c[1] + ey + ry) / dy) + 1;
    if (c0 < 0) c0 = 0;
    if (r0 < 0) r0 = 0;
    if (c1 > nc - 1) c1 = nc - 1;
    if (r1 > nr - 1) r1 = nr - 1;
    double best = 0, bpa[3] = {0, 0, 0}, bpb[3] = {0, 0, 0}; int have = 0, bedge = 0;
    for (int r = r0; r <= r1; r++)
      for (int cc = c0; cc <= c1; cc++) {
        /* edges starting at vertex (cc, r): to (cc+1, r), to (cc, r+1), and the strip diagonal (cc, r+1)-(cc+1, r) */
        static const int E[3][4] = {{0, 0, 1, 0}, {0, 0, 0, 1}, {0, 1, 1, 0}};
        for (int k = 0; k < 3; k++) {
          int ca = cc + E[k][0], ra = r + E[k][1], cb = cc + E[k][2], rb = r + E[k][3];
          if (ca > nc - 1 || cb > nc - 1 || ra > nr - 1 || rb > nr - 1) continue;
          double va[3] = {ca * dx - rx, ra * dy - ry, (double)m->hf_data[ra * nc + ca] * ez};
          double vb[3] = {cb * dx - rx, rb * dy - ry, (double)m->hf_data[rb * nc + cb] * ez};
          double mid[3], ed[3];
          for (int i = 0; i < 3; i++) { mid[i] = 0.5 * (va[i] + vb[i]); ed[i] = vb[i] - va[i]; }
          double el = norm3(ed);
          if (el < MINVAL) continue;
          for (int i = 0; i < 3; i++) ed[i] /= el;
          double s_, t_;
          seg_seg(c, a, H, mid, ed, 0.5 * el, &s_, &t_);
          double pa[3], pb[3], df[3];
          for (int i = 0; i < 3; i++) { pa[i] = c[i] + s_ * a[i]; pb[i] = mid[i] + t_ * ed[i]; df[i] = pa[i] - pb[i]; }
          double dist = norm3(df) - rad;
          if (dist < 0 && (!have || dist < best)) { have = 1; best = dist; memcpy(bpa, pa, 24); memcpy(bpb, pb, 24); bedge = (cc * 128 + r) * 3 + k; }
        }
      }
    if (!have) continue;
    double n[3] = {bpa[0] - bpb[0], bpa[1] - bpb[1], bpa[2] - bpb[2]};
    double nn = norm3(n);
    if (nn < MINVAL) { d->flags |= TGO_FLAG_DEEP; continue; }
    for (int i = 0; i < 3; i++) n[i] /= nn;
    double pos[3];
    for (int i = 0; i < 3; i++) pos[i] = bpa[i] + m->hf_pos[i] - n[i] * (rad + 0.5 * best);
    add_contact(d, pos, n, best, -1, b, 0, 1 + 5 * b);
    d->con[d->ncon - 1].feature = bedge;
  }
}

/* Census for DESIGN.md's heightfield assumptions (development aid; not on the path of tgo_step): with the kinematics d holds, how many
 * grid triangles does each cap touch (MuJoCo: one contact per prism touched; hfield_caps keeps the two deepest) and how many grid edges
 * does each bar's cylinder touch (hfield_bars keeps the deepest).  out[0..5] = triangles per cap, out[6..8] = edges per bar. */
void tgo_hfield_census(const tgo_model* m, const tgo_data* d, int* out) {
  for (int i = 0; i < 9; i++) out[i] = 0;
  if (!m->hf_data) return;
  const int nr = m->hf_nrow, nc = m->hf_ncol;
  const double rx = m->hf_size[0], ry = m->hf_size[1], ez = m->hf_size[2];
  const double dx = 2 * rx / (nc - 1), dy = 2 * ry / (nr - 1);
  for (int k = 0; k < 6; k++) {
    const double rad = m->cap_radius;
    const double* cw = d->cap_xpos[k];
    double p[3] = {cw[0] - m->hf_pos[0], cw[1] - m->hf_pos[1], cw[2] - m->hf_pos[2]};
    if (p[0] < -rx - rad || p[0] > rx + rad || p[1] < -ry - rad || p[1] > ry + rad || p[2] > ez + rad) continue;
    int c0 = (int)floor((p[0] - rad + rx) / dx), c1 = (int)floor((p[0] + rad + rx) / dx);
    int r0 = (int)floor((p[1] - rad + ry) / dy), r1 = (int)floor((p[1] + rad + ry) / dy);
    if (c0 < 0) c0 = 0;
    if (r0 < 0) r0 = 0;
    if (c1 > nc - 2) c1 = nc - 2;
    if (r1 > nr - 2) r1 = nr - 2;
    for (int r = r0; r <= r1; r++)
      for (int c = c0; c <= c1; c++) {
        double v[4][3];
        for (int j = 0; j < 4; j++) {
          int cc = c + (j >> 1), rr = r + (j & 1);
          v[j][0] = cc * dx - rx; v[j][1] = rr * dy - ry; v[j][2] = (double)m->hf_data[rr * nc + cc] * ez;
        }
        for (int tri = 0; tri < 2; tri++) {
          double q[3];
          closest_on_triangle(p, v[tri], v[tri + 1], v[tri + 2], q);
          double dif[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
          if (norm3(dif) - rad < 0) out[k]++;
        }
      }
  }
  for (int b = 0; b < TGO_NBAR; b++) {
    const double rad = m->rcyl_radius, H = m->bar_half;
    const double* R = d->xmat[b];
    double a[3] = {R[2], R[5], R[8]};
    double c[3] = {d->xpos[b][0] - m->hf_pos[0], d->xpos[b][1] - m->hf_pos[1], d->xpos[b][2] - m->hf_pos[2]};
    double ex = fabs(a[0]) * H + rad, ey = fabs(a[1]) * H + rad;
    if (c[0] + ex < -rx || c[0] - ex > rx || c[1] + ey < -ry || c[1] - ey > ry) continue;
    if (c[2] - fabs(a[2]) * H - rad > ez) continue;
    int c0 = (int)floor((c[0] - ex + rx) / dx), c1 = (int)floor((c[0] + ex + rx) / dx) + 1;
    int r0 = (int)floor((c[1] - ey + ry) / dy), r1 = (int)floor((c[1] + ey + ry) / dy) + 1;
    if (c0 < 0) c0 = 0;
    if (r0 < 0) r0 = 0;
    if (c1 > nc - 1) c1 = nc - 1;
    if (r1 > nr - 1) r1 = nr - 1;
    for (int r = r0; r <= r1; r++)
      for (int cc = c0; cc <= c1; cc++) {
        static const int E[3][4] = {{0, 0, 1, 0}, {0, 0, 0, 1}, {0, 1, 1, 0}};
        for (int k = 0; k < 3; k++) {
          int ca = cc + E[k][0], ra = r + E[k][1], cb = cc + E[k][2], rb = r + E[k][3];
          if (ca > nc - 1 || cb > nc - 1 || ra > nr - 1 || rb > nr - 1) continue;
          double va[3] = {ca * dx - rx, ra * dy - ry, (double)m->hf_data[ra * nc + ca] * ez};
          double vb[3] = {cb * dx - rx, rb * dy - ry, (double)m->hf_data[rb * nc + cb] * ez};
          double mid[3], ed[3];
          for (int i = 0; i < 3; i++) { mid[i] = 0.5 * (va[i] + vb[i]); ed[i] = vb[i] - va[i]; }
          double el = norm3(ed);
          if (el < MINVAL) continue;
          for (int i = 0; i < 3; i++) ed[i] /= el;
          double s_, t_, df[3];
          seg_seg(c, a, H, mid, ed, 0.5 * el, &s_, &t_);
          for (int i = 0; i < 3; i++) df[i] = c[i] + s_ * a[i] - (mid[i] + t_ * ed[i]);
          if (norm3(df) - rad < 0) out[6 + b]++;
        }
      }
  }
}

static void collision(const tgo_model* m, tgo_data* d) {
  d->ncon = 0;
  if (m->hf_data) { hfield_caps(m, d); hfield_bars(m, d); }
  /* floor plane (geom 0, z=0, normal +z) vs the 6 cap spheres: mjc_PlaneSphere */
  for (int b = 0; b < TGO_NBAR && !m->hf_data; b++) {
    for (int e = 0; e < 2; e++) {
      const double* c = d->cap_xpos[2 * b + e];
      double cdist = c[2];
      if (cdist > m->cap_radius) continue;
      double dist = cdist - m->cap_radius;
      if (dist >= 0) continue; /* margin 0: the constraint is only instantiated for dist < 0 */
      double n[3] = {0, 0, 1};
      double pos[3] = {c[0], c[1], c[2] - (m->cap_radius + 0.5 * dist)};
      add_contact(d, pos, n, dist, -1, b, 0, 1 + 5 * b + 1 + e);
    }
    /* plane-cylinder cannot occur before a cap is ~29 mm inside the floor; report it if it would */
    {
      const double* R = d->xmat[b];
      double az = R[8];
      double rim = m->rcyl_radius * sqrt(fmax(0.0, 1.0 - az * az));
      if (d->xpos[b][2] - fabs(az) * m->bar_half - rim < 0) d->flags |= TGO_FLAG_PLANE_CYL;
    }
  }
  /* bar-bar */
  bgeom g[TGO_NBAR][5];
  for (int b = 0; b < TGO_NBAR; b++) bar_geoms(m, d, b, g[b]);
  for (int A = 0; A < TGO_NBAR; A++)
    for (int B = A + 1; B < TGO_NBAR; B++) {
      double s, t;
      seg_seg(g[A][0].c, g[A][0].a, m->bar_half, g[B][0].c, g[B][0].a, m->bar_half, &s, &t);
      double dif[3];
      for (int i = 0; i < 3; i++) dif[i] = (g[B][0].c[i] + t * g[B][0].a[i]) - (g[A][0].c[i] + s * g[A][0].a[i]);
      if (norm3(dif) > 2 * m->cap_radius + 1e-9) continue; /* nothing of A can touch anything of B */
      for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
          const bgeom* ga = &g[A][i];
          const bgeom* gb = &g[B][j];
          if (ga->type == 0 && gb->type == 0) sphere_sphere(d, ga, A, gb, B);
          else if (ga->type == 0 && gb->type == 1) sphere_cyl(d, ga, A, gb, B);
          else if (ga->type == 1 && gb->type == 0) sphere_cyl(d, gb, B, ga, A);
          else cyl_cyl(d, ga, A, gb, B);
        }
    }
}

/* ------------------------------------------------------------------ constraint rows */
static double impedance(const double* solimp, double pos) {
  /* getimpedance(): margin 0 */
  if (solimp[0] == solimp[1] || solimp[2] <= MINVAL) return 0.5 * (solimp[0] + solimp[1]);
  double x = fabs(pos / solimp[2]);
  if (x >= 1) return solimp[1];
  if (x <= 0) return solimp[0];
  double y, mid = solimp[3], pw = solimp[4];
  if (pw == 1) y = x;
  else if (x <= mid) y = pow(x, pw) / pow(mid, pw - 1);
  else y = 1 - pow(1 - x, pw) / pow(1 - mid, pw - 1);
  return solimp[0] + y * (solimp[1] - solimp[0]);
}

static void make_constraint(const tgo_model* m, tgo_data* d) {
  for (int ci = 0; ci < d->ncon; ci++) {
    tgo_contact* c = &d->con[ci];
    memset(c->J, 0, sizeof(c->J));
    for (int side = 0; side < 2; side++) {
      int b = side == 0 ? c->body1 : c->body2;
      double sg = side == 0 ? -1.0 : 1.0;
      if (b < 0) continue;
      const double* R = d->xmat[b];
      double r[3] = {c->pos[0] - d->xpos[b][0], c->pos[1] - d->xpos[b][1], c->pos[2] - d->xpos[b][2]};
      for (int row = 0; row < 3; row++) {
        const double* f = c->frame + 3 * row;
        /* translational rows: f . (v + w_world x r) ; w_world = R w_body */
        for (int i = 0; i < 3; i++) c->J[row][6 * b + i] += sg * f[i];
        double rxf[3];
        cross3(rxf, r, f); /* f.(w x r) = w.(r x f) */
        for (int k = 0; k < 3; k++) {
          c->J[row][6 * b + 3 + k] += sg * (R[k] * rxf[0] + R[3 + k] * rxf[1] + R[6 + k] * rxf[2]);
          /* rotational rows: f . w_world */
          c->J[3 + row][6 * b + 3 + k] += sg * (R[k] * f[0] + R[3 + k] * f[1] + R[6 + k] * f[2]);
        }
      }
    }
    /* impedance, R, D (mj_makeImpedance) */
    double imp = impedance(m->solimp, c->dist);
    double diagApprox = m->invweight_tran * ((c->body1 >= 0) + (c->body2 >= 0));
    double R0 = fmax(MINVAL, (1 - imp) * diagApprox / imp);
    c->R[0] = R0;
    c->R[1] = R0 / m->impratio;
    c->mu = m->friction[0] * sqrt(c->R[1] / c->R[0]);
    for (int j = 1; j < 5; j++) c->R[j + 1] = c->R[1] * m->friction[0] * m->friction[0] / (m->friction[j] * m->friction[j]);
    for (int j = 0; j < 6; j++) c->D[j] = 1.0 / c->R[j];
  }
}

static void reference_constraint(const tgo_model* m, tgo_data* d) {
  double dmax = m->solimp[1];
  double K, B;
  if (m->solref[0] > 0) { /* standard (timeconst, dampratio) */
    K = 1.0 / fmax(MINVAL, dmax * dmax * m->solref[0] * m->solref[0] * m->solref[1] * m->solref[1]);
    B = 2.0 / fmax(MINVAL, dmax * m->solref[0]);
  } else { /* direct (-stiffness, -damping) */
    K = -m->solref[0] / fmax(MINVAL, dmax * dmax);
    B = -m->solref[1] / fmax(MINVAL, dmax);
  }
  for (int ci = 0; ci < d->ncon; ci++) {
    tgo_contact* c = &d->con[ci];
    double imp = impedance(m->solimp, c->dist);
    for (int j = 0; j < 6; j++) {
      double vel = 0;
      for (int k = 0; k < TGO_NV; k++) vel += c->J[j][k] * d->qvel[k];
      double pos = j == 0 ? c->dist : 0.0;
      c->aref[j] = -B * vel - K * imp * pos;
    }
  }
}

/* ------------------------------------------------------------------ smooth forces */
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void smooth_forces(const tgo_model* m, tgo_data* d) {
  memset(d->qfrc_passive, 0, sizeof(d->qfrc_passive));
  memset(d->qfrc_actuator, 0, sizeof(d->qfrc_actuator));
  for (int t = 0; t < TGO_NTEN; t++) {
    double v = 0;
    for (int k = 0; k < TGO_NV; k++) v += d->ten_J[t][k] * d->qvel[k];
    d->ten_velocity[t] = v;
    double f = -m->ten_stiffness[t] * (d->ten_length[t] - m->ten_springlen[t]) - m->ten_damping[t] * v;
    if (m->ten_stiffness[t] == 0 && m->ten_damping[t] == 0) f = 0;
    for (int k = 0; k < TGO_NV; k++) d->qfrc_passive[k] += d->ten_J[t][k] * f;
  }
  for (int a = 0; a < TGO_NU; a++) {
    double ctrl = clampd(d->ctrl[a], m->ctrl_lo, m->ctrl_hi);
    double in = ctrl;
    if (m->act_filter) { /* dyntype filter: act_dot = (ctrl - act) / tau, the force sees the activation (mj_fwdActuation [ext]) */
      d->act_dot[a] = (ctrl - d->act[a]) / fmax(MINVAL, m->act_tau);
      in = d->act[a];
    }
    double f = m->gain * in + m->bias0 + m->bias1 * d->ten_length[a] + m->bias2 * d->ten_velocity[a];
    f = clampd(f, m->force_lo, m->force_hi);
    d->actuator_force[a] = f;
    for (int k = 0; k < TGO_NV; k++) d->qfrc_actuator[k] += d->ten_J[a][k] * f;
  }
  /* bias (mj_rne, flg_acc=0): gravity on the linear dofs, w x Iw on the body-frame angular dofs */
  for (int b = 0; b < TGO_NBAR; b++) {
    const double* w = d->qvel + 6 * b + 3;
    double Iw[3] = {m->inertia[0] * w[0], m->inertia[1] * w[1], m->inertia[2] * w[2]};
    double g[3];
    cross3(g, w, Iw);
    d->qfrc_bias[6 * b + 0] = 0; d->qfrc_bias[6 * b + 1] = 0; d->qfrc_bias[6 * b + 2] = m->mass * m->gravity;
    d->qfrc_bias[6 * b + 3] = g[0]; d->qfrc_bias[6 * b + 4] = g[1]; d->qfrc_bias[6 * b + 5] = g[2];
  }
  for (int k = 0; k < TGO_NV; k++) {
    d->qfrc_smooth[k] = d->qfrc_passive[k] - d->qfrc_bias[k] + d->qfrc_actuator[k];
    int j = k % 6;
    double Mk = j < 3 ? m->mass : m->inertia[j - 3];
    d->qacc_smooth[k] = d->qfrc_smooth[k] / Mk;
  }
}

/* ------------------------------------------------------------------ elliptic cone cost (PrimalUpdateConstraint) */
static double cone_eval(const tgo_model* m, tgo_contact* c, const double* jar, double* force, double* C /*36 or NULL*/) {
  const double* fr = m->friction;
  double mu = c->mu;
  double U[6];
  U[0] = jar[0] * mu;
  for (int j = 1; j < 6; j++) U[j] = jar[j] * fr[j - 1];
  double N = U[0], T = 0;
  for (int j = 1; j < 6; j++) T += U[j] * U[j];
  T = sqrt(T);
  if (C) memset(C, 0, 36 * sizeof(double));
  for (int j = 0; j < 6; j++) force[j] = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) { c->zone = 0; return 0.0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    c->zone = 2;
    double cost = 0;
    for (int j = 0; j < 6; j++) {
      cost += 0.5 * c->D[j] * jar[j] * jar[j];
      force[j] = -c->D[j] * jar[j];
      if (C) C[7 * j] = c->D[j];
    }
    return cost;
  }
  c->zone = 1;
  double Dm = c->D[0] / (mu * mu * (1 + mu * mu));
  double NT = N - mu * T;
  double cost = 0.5 * Dm * NT * NT;
  force[0] = -Dm * NT * mu;
  for (int j = 1; j < 6; j++) force[j] = -force[0] / T * U[j] * fr[j - 1];
  if (C) {
    /* C = Dm ( g g^T + NT * hess(NT) ),  g0 = mu, gj = -mu fr_j U_j / T,
       hess_jk = -mu fr_j fr_k (delta_jk / T - U_j U_k / T^3),  j,k >= 1 */
    double g[6];
    g[0] = mu;
    for (int j = 1; j < 6; j++) g[j] = -mu * fr[j - 1] * U[j] / T;
    for (int j = 0; j < 6; j++)
      for (int k = 0; k < 6; k++) {
        double h = 0;
        if (j >= 1 && k >= 1) h = -mu * fr[j - 1] * fr[k - 1] * ((j == k ? 1.0 / T : 0.0) - U[j] * U[k] / (T * T * T));
        C[6 * j + k] = Dm * (g[j] * g[k] + NT * h);
      }
  }
  return cost;
}

static int cholesky(double* A, int n) { /* in place lower, returns 0 on success */
  for (int j = 0; j < n; j++) {
    double s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (s <= 0) return -1;
    double l = sqrt(s);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / l;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

static double Mdiag(const tgo_model* m, int k) { int j = k % 6; return j < 3 ? m->mass : m->inertia[j - 3]; }

/* total cost at qacc; optionally gradient and Hessian */
static double total_cost(const tgo_model* m, tgo_data* d, const double* qacc, double* grad, double* H) {
  double cost = 0;
  for (int k = 0; k < TGO_NV; k++) {
    double e = qacc[k] - d->qacc_smooth[k];
    cost += 0.5 * Mdiag(m, k) * e * e;
    if (grad) grad[k] = Mdiag(m, k) * e;
  }
  if (H) {
    memset(H, 0, TGO_NV * TGO_NV * sizeof(double));
    for (int k = 0; k < TGO_NV; k++) H[k * TGO_NV + k] = Mdiag(m, k);
  }
  for (int ci = 0; ci < d->ncon; ci++) {
    tgo_contact* c = &d->con[ci];
    double jar[6], C[36];
    for (int j = 0; j < 6; j++) {
      double s = -c->aref[j];
      for (int k = 0; k < TGO_NV; k++) s += c->J[j][k] * qacc[k];
      jar[j] = s;
    }
    cost += cone_eval(m, c, jar, c->force, H ? C : NULL);
    if (grad)
      for (int j = 0; j < 6; j++)
        for (int k = 0; k < TGO_NV; k++) grad[k] -= c->J[j][k] * c->force[j];
    if (H && c->zone != 0) {
      double CJ[6][TGO_NV];
      for (int j = 0; j < 6; j++)
        for (int k = 0; k < TGO_NV; k++) {
          double s = 0;
          for (int l = 0; l < 6; l++) s += C[6 * j + l] * c->J[l][k];
          CJ[j][k] = s;
        }
      for (int a = 0; a < TGO_NV; a++)
        for (int b = 0; b < TGO_NV; b++) {
          double s = 0;
          for (int j = 0; j < 6; j++) s += c->J[j][a] * CJ[j][b];
          H[a * TGO_NV + b] += s;
        }
    }
  }
  return cost;
}

/* derivative of the cost along qacc + alpha p : phi'(alpha), phi''(alpha) */
static void line_eval(const tgo_model* m, tgo_data* d, const double* qacc, const double* p, double alpha, double* d1, double* d2) {
  double x[TGO_NV];
  double g1 = 0, g2 = 0;
  for (int k = 0; k < TGO_NV; k++) {
    x[k] = qacc[k] + alpha * p[k];
    g1 += p[k] * Mdiag(m, k) * (x[k] - d->qacc_smooth[k]);
    g2 += p[k] * Mdiag(m, k) * p[k];
  }
  for (int ci = 0; ci < d->ncon; ci++) {
    tgo_contact* c = &d->con[ci];
    double jar[6], jp[6], C[36], f[6];
    for (int j = 0; j < 6; j++) {
      double s = -c->aref[j], sp = 0;
      for (int k = 0; k < TGO_NV; k++) { s += c->J[j][k] * x[k]; sp += c->J[j][k] * p[k]; }
      jar[j] = s; jp[j] = sp;
    }
    int zone = c->zone;
    cone_eval(m, c, jar, f, C);
    c->zone = zone;
    for (int j = 0; j < 6; j++) {
      g1 -= f[j] * jp[j];
      for (int k = 0; k < 6; k++) g2 += jp[j] * C[6 * j + k] * jp[k];
    }
  }
  *d1 = g1; *d2 = g2;
}

double tgo_debug_ls_tol = 1e-13; /* test hook: relative line-search tolerance */
int tgo_debug_start = 0; /* test hook: 0 = better of warm/smooth (mujoco), 1 = always warm, 2 = always smooth */
double tgo_debug_tol = 1e-12; /* test hook: gradient tolerance (scaled) */
int tgo_debug_trace = 0;
#include <stdio.h>
static void newton_solve(const tgo_model* m, tgo_data* d) {
  double qacc[TGO_NV], grad[TGO_NV], p[TGO_NV], H[TGO_NV * TGO_NV];
  d->solver_iter = 0;
  if (d->ncon == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(d->qacc));
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(d->qacc));
    memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
    return;
  }
  /* warmstart(): the cheaper of qacc_warmstart and qacc_smooth */
  {
    double cw = total_cost(m, d, d->qacc_warmstart, NULL, NULL);
    double cs = total_cost(m, d, d->qacc_smooth, NULL, NULL);
    int usew = cw < cs;
    if (tgo_debug_start == 1) usew = 1;
    if (tgo_debug_start == 2) usew = 0;
    memcpy(qacc, usew ? d->qacc_warmstart : d->qacc_smooth, sizeof(qacc));
  }
  /* scale as mujoco: 1/(meaninertia * nv); we iterate far below its 1e-8 tolerance so the
     restatement sits at the (unique) optimum the reference solver approaches */
  double meaninertia = (3 * m->mass + m->inertia[0] + m->inertia[1] + m->inertia[2]) / 6.0;
  double scale = 1.0 / (meaninertia * TGO_NV);
  int it;
  for (it = 0; it < m->newton_maxiter; it++) {
    total_cost(m, d, qacc, grad, H);
    double gn = 0;
    for (int k = 0; k < TGO_NV; k++) gn += grad[k] * grad[k];
    gn = sqrt(gn);
    if (scale * gn < tgo_debug_tol) break;
    if (cholesky(H, TGO_NV) != 0) { d->flags |= TGO_FLAG_NOCONV; break; }
    for (int k = 0; k < TGO_NV; k++) p[k] = -grad[k];
    chol_solve(H, TGO_NV, p);
    /* exact line search on the convex 1-d restriction: safeguarded Newton on phi' */
    double d0, h0;
    line_eval(m, d, qacc, p, 0.0, &d0, &h0);
    if (d0 >= 0) break;
    double lo = 0, hi = -1, alpha = 1.0, d1, d2;
    int ok = 0;
    for (int ls = 0; ls < 100; ls++) {
      line_eval(m, d, qacc, p, alpha, &d1, &d2);
      if (fabs(d1) <= tgo_debug_ls_tol * fabs(d0)) { ok = 1; break; }
      if (d1 < 0) lo = alpha; else hi = alpha;
      double na = d2 > 0 ? alpha - d1 / d2 : -1;
      if (hi < 0) { if (na <= lo || na > 10 * (lo + 1)) na = 2 * alpha + 1e-3; }
      else if (na <= lo || na >= hi) na = 0.5 * (lo + hi);
      if (hi > 0 && hi - lo < 1e-16 * (1 + hi)) { alpha = na; ok = 1; break; }
      alpha = na;
    }
    (void)ok;
    if (tgo_debug_trace) { int z[8]={0}; for (int ci=0;ci<d->ncon&&ci<8;ci++) z[ci]=d->con[ci].zone; printf("it %d gn %.3e d0 %.3e alpha %.6f zones %d%d%d%d qacc0 %.5f\n", it, gn, d0, alpha, z[0],z[1],z[2],z[3], qacc[0]); }
    double step = 0, mag = 0;
    for (int k = 0; k < TGO_NV; k++) { qacc[k] += alpha * p[k]; step += fabs(alpha * p[k]); mag += fabs(qacc[k]); }
    if (step <= 1e-15 * (1.0 + mag)) break; /* converged to rounding */
  }
  if (it >= m->newton_maxiter) d->flags |= TGO_FLAG_NOCONV;
  d->solver_iter = it;
  total_cost(m, d, qacc, NULL, NULL); /* final forces / zones */
  memcpy(d->qacc, qacc, sizeof(qacc));
  memcpy(d->qacc_warmstart, qacc, sizeof(qacc));
  memset(d->qfrc_constraint, 0, sizeof(d->qfrc_constraint));
  for (int ci = 0; ci < d->ncon; ci++) {
    tgo_contact* c = &d->con[ci];
    for (int j = 0; j < 6; j++)
      for (int k = 0; k < TGO_NV; k++) d->qfrc_constraint[k] += c->J[j][k] * c->force[j];
  }
}

/* ------------------------------------------------------------------ mj_rnePostConstraint: cfrc_ext */
static void contact_wrench(tgo_data* d) {
  memset(d->cfrc_ext, 0, sizeof(d->cfrc_ext));
  for (int ci = 0; ci < d->ncon; ci++) {
    const tgo_contact* c = &d->con[ci];
    double f[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < 3; i++) { f[i] += c->frame[3 * r + i] * c->force[r]; tq[i] += c->frame[3 * r + i] * c->force[3 + r]; }
    for (int side = 0; side < 2; side++) {
      int b = side == 0 ? c->body1 : c->body2;
      double sg = side == 0 ? -1.0 : 1.0;
      if (b < 0) continue; /* world row is skipped: if ((k = geom_bodyid)) */
      double r[3] = {c->pos[0] - d->xpos[b][0], c->pos[1] - d->xpos[b][1], c->pos[2] - d->xpos[b][2]};
      double rxf[3];
      cross3(rxf, r, f);
      for (int i = 0; i < 3; i++) {
        d->cfrc_ext[b + 1][i] += sg * (tq[i] + rxf[i]);
        d->cfrc_ext[b + 1][3 + i] += sg * f[i];
      }
    }
  }
}

/* ------------------------------------------------------------------ public */
void tgo_forward(const tgo_model* m, tgo_data* d) {
  d->flags = 0;
  kinematics(m, d);
  tendons(m, d);
  collision(m, d);
  make_constraint(m, d);
  smooth_forces(m, d);
  reference_constraint(m, d);
  newton_solve(m, d);
  contact_wrench(d);
}

static void implicitfast_advance(const tgo_model* m, tgo_data* d) {
  double h = m->timestep;
  double A[TGO_NV * TGO_NV], rhs[TGO_NV];
  memset(A, 0, sizeof(A));
  for (int k = 0; k < TGO_NV; k++) { A[k * TGO_NV + k] = Mdiag(m, k); rhs[k] = d->qfrc_smooth[k] + d->qfrc_constraint[k]; }
  /* qDeriv = d(passive+actuator)/dv = -sum_t b_t J_t^T J_t  (mjd_smooth_vel, flg_bias = 0) */
  for (int t = 0; t < TGO_NTEN; t++) {
    double b = m->ten_damping[t];
    if (t < TGO_NU) {
      double f = d->actuator_force[t];
      int clamped = (f <= m->force_lo) || (f >= m->force_hi);
      if (!(m->actuator_deriv_skip_clamped && clamped)) b += -m->bias2;
    }
    if (b == 0) continue;
    for (int i = 0; i < TGO_NV; i++)
      for (int j = 0; j < TGO_NV; j++) A[i * TGO_NV + j] += h * b * d->ten_J[t][i] * d->ten_J[t][j];
  }
  cholesky(A, TGO_NV);
  chol_solve(A, TGO_NV, rhs);
  /* mj_advance */
  for (int k = 0; k < TGO_NV; k++) d->qvel[k] += h * rhs[k];
  for (int b = 0; b < TGO_NBAR; b++) {
    double* q = d->qpos + 7 * b;
    const double* v = d->qvel + 6 * b;
    for (int i = 0; i < 3; i++) q[i] += h * v[i];
    /* mju_quatIntegrate: q <- normalize(q) * exp(h w / 2), w in the body frame */
    double w[3] = {v[3], v[4], v[5]};
    double n = norm3(w);
    double qr[4] = {1, 0, 0, 0};
    if (n >= MINVAL) {
      double ang = h * n;
      double s = sin(0.5 * ang) / n;
      qr[0] = cos(0.5 * ang); qr[1] = w[0] * s; qr[2] = w[1] * s; qr[3] = w[2] * s;
    }
    normquat(q + 3);
    double qn[4];
    mulquat(qn, q + 3, qr);
    memcpy(q + 3, qn, sizeof(qn));
  }
  if (m->act_filter)
    for (int a = 0; a < TGO_NU; a++) d->act[a] += h * d->act_dot[a]; /* mj_advance: explicit Euler on the activations */
  d->time += h;
}

uint32_t tgo_contact_mask(const tgo_model* m, const tgo_data* d) {
  uint32_t mask = 0;
  int floor_hits[6] = {0, 0, 0, 0, 0, 0}, others[3] = {0, 0, 0};
  for (int ci = 0; ci < d->ncon; ci++) {
    const tgo_contact* c = &d->con[ci];
    if (c->body1 < 0) { /* floor (plane or heightfield) against geom2 of bar body2 */
      const int g = c->geom2 - 1 - 5 * c->body2; /* 0 r-cylinder, 1 cap +z, 2 cap -z */
      if (g == 0) mask |= 1u << (12 + c->body2);
      else if (g == 1 || g == 2) {
        const int cap = 2 * c->body2 + (g - 1);
        mask |= (floor_hits[cap]++ == 0) ? (1u << cap) : (1u << (16 + cap));
      }
    } else {
      const int A = c->body1 < c->body2 ? c->body1 : c->body2, B = c->body1 < c->body2 ? c->body2 : c->body1;
      const int p = A == 0 ? (B == 1 ? 0 : 1) : 2;
      const int g1 = c->geom1 - 1 - 5 * c->body1, g2 = c->geom2 - 1 - 5 * c->body2;
      if (g1 == 0 && g2 == 0) mask |= 1u << (6 + 2 * p);
      else others[p]++;
    }
  }
  /* every OTHER contact between two bars takes the next lane of the row kernels' POOL, whichever pair it belongs to (tgr_row.h: flat floor
   * lanes 7, 9, 11, 12..15; heightfield, whose lanes 12..14 carry the bars' terrain contacts: 7, 9, 11, 15) */
  {
    static const int pool0[7] = {7, 9, 11, 12, 13, 14, 15}, pool1[4] = {7, 9, 11, 15};
    const int np = m->hf_data ? 4 : 7, total = others[0] + others[1] + others[2];
    for (int i = 0; i < total && i < np; i++) mask |= 1u << (m->hf_data ? pool1[i] : pool0[i]);
  }
  return mask;
}

/* which facets / grid edges the heightfield contacts touch, as the row kernels hash them (tgr_row.h, sub-step trace): low 16 bits the caps
 * (a cap's facets enter symmetrically: which of two is "the first" does not matter), high 16 bits the bars' cylinders */
static uint32_t mix_id(uint32_t id) { uint32_t x = (id + 1u) * 2654435761u; return x ^ (x >> 15); }
uint32_t tgo_feature_hash(const tgo_model* m, const tgo_data* d) {
  if (!m->hf_data) return 0;
  uint32_t caps = 0, rods = 0;
  for (int ci = 0; ci < d->ncon; ci++) {
    const tgo_contact* c = &d->con[ci];
    if (c->body1 >= 0) continue;
    const int g = c->geom2 - 1 - 5 * c->body2; /* 0 r-cylinder, 1 cap +z, 2 cap -z */
    if (g == 0) rods += (uint32_t)(c->body2 + 1) * mix_id((uint32_t)c->feature);
    else if (g == 1 || g == 2) caps += (uint32_t)(2 * c->body2 + (g - 1) + 1) * mix_id((uint32_t)c->feature);
  }
  return ((caps ^ (caps >> 16)) & 0xffffu) | (((rods ^ (rods >> 16)) & 0xffffu) << 16);
}

void tgo_step(const tgo_model* m, tgo_data* d, int nstep) {
  d->trace_n = 0;
  for (int i = 0; i < nstep; i++) {
    tgo_forward(m, d);
    if (d->trace_n < TGO_TRACE) d->trace_id[d->trace_n] = tgo_feature_hash(m, d);
    if (d->trace_n < TGO_TRACE)
      d->trace[d->trace_n++] = tgo_contact_mask(m, d) | ((uint32_t)(d->solver_iter & 0xff) << 22) | ((d->flags & TGO_FLAG_NOCONV) ? 1u << 30 : 0u);
    implicitfast_advance(m, d);
  }
}
