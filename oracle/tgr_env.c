/*
 * tgr_env.c -- fp64 restatement of the reference's Python env layer on top of
 * tgr_physics.c.  TEST INFRASTRUCTURE (see tgr_oracle.h).
 *
 * Follows, line by line where it matters:
 *   tensegrity_env.py:245-289  healthy_reward / control_cost / contact_cost / is_healthy / terminated
 *   tensegrity_env.py:291-410  step
 *   tensegrity_env.py:412-430  _get_obs
 *   tensegrity_env.py:433-512  reset_model
 *   tr_env.py:280-325,327-527,529-646,648-683,709-872
 * Pinned against the reference classes themselves by tests/golden/ (oracle/gen_golden.py).
 */
#include "tgr_oracle.h"

#include <math.h>
#include <string.h>

#define PI 3.14159265358979323846

const double tgo_rolling_qpos[6][TGO_NQ] = {/* tr_env.py:723-728 */
    {0.07900689, -0.32670045, 0.23079722, 0.49365198, -0.74001353, 0.26668361, 0.37090101, 0.13713385, -0.24342633, 0.32722167, 0.82936968, -0.31256817, -0.46189217, -0.03320677, 0.04903377, -0.3421725, 0.36675097, 0.33407281, 0.43794432, -0.72515863, -0.41321313},
    {0.15521685, -0.20651043, 0.38922255, 0.85639289, -0.26723449, -0.44110818, -0.02450564, 0.02999107, -0.33576412, 0.43868814, 0.33839518, 0.48544838, -0.73094128, -0.33993149, 0.08083394, -0.31942006, 0.25783949, 0.51726058, -0.74281033, 0.29432583, 0.30667022},
    {0.02985312, -0.33588999, 0.43866597, 0.33840617, 0.48522953, -0.73107566, -0.33994403, 0.08072907, -0.31942136, 0.25766037, 0.51740763, -0.74276722, 0.29421311, 0.30663471, 0.15537661, -0.20664637, 0.38923648, 0.85640002, -0.26722239, -0.44110397, -0.02446392},
    {0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798, 0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067, 0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277},
    {0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067, 0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277, 0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798},
    {0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277, 0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798, 0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067}};

/* ------------------------------------------------------------------ philox4x32-10 */
void tgo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double u53(uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; }

void tgo_rng_draws(uint64_t seed, uint64_t env_id, uint64_t episode, int env_kind, double mean, double stdev,
                   double lo, double hi, double hmin, double hmax, int* pose_idx, double* theta, double tendons[6],
                   double u_way[2]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)episode, 0};
  uint32_t o[4];
  (void)env_kind;
  tgo_philox4x32(ctr, key, o);
  *pose_idx = (int)(o[0] % 6u);
  *theta = hmin + u53(o[1], o[2]) * (hmax - hmin);
  for (int k = 0; k < 3; k++) {
    ctr[3] = 1 + k;
    tgo_philox4x32(ctr, key, o);
    double u1 = 1.0 - u53(o[0], o[1]), u2 = u53(o[2], o[3]);
    double r = sqrt(-2.0 * log(u1));
    double z0 = r * cos(2 * PI * u2), z1 = r * sin(2 * PI * u2);
    double t0 = z0 * stdev + mean, t1 = z1 * stdev + mean;
    tendons[2 * k] = t0 > hi ? hi : (t0 < lo ? lo : t0);
    tendons[2 * k + 1] = t1 > hi ? hi : (t1 < lo ? lo : t1);
  }
  ctr[3] = 4;
  tgo_philox4x32(ctr, key, o);
  u_way[0] = u53(o[0], o[1]);
  u_way[1] = u53(o[2], o[3]);
}

/* ------------------------------------------------------------------ observation noise */
void tgo_obs_noise_draws(uint64_t seed, uint64_t key, uint64_t episode, int elapsed, double z[48]) {
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int j = 0; j < 12; j++) {
    uint32_t ctr[4] = {(uint32_t)key, (uint32_t)(key >> 32), (uint32_t)episode, 0x80000000u + ((uint32_t)elapsed << 4) + (uint32_t)j};
    uint32_t o[4];
    tgo_philox4x32(ctr, k, o);
    for (int h = 0; h < 2; h++) { /* Box-Muller on 32-bit uniforms: u1 in (0,1], u2 in [0,1) */
      double u1 = ((double)o[2 * h] + 1.0) / 4294967296.0, u2 = (double)o[2 * h + 1] / 4294967296.0;
      double r = sqrt(-2.0 * log(u1));
      z[4 * j + 2 * h] = r * cos(2 * PI * u2);
      z[4 * j + 2 * h + 1] = r * sin(2 * PI * u2);
    }
  }
}

void tgo_apply_obs_noise(const tgo_env_cfg* c, const double z[45], double* obs) {
  /* layout [cap pos 18][cap vel 18 if use_cap_velocity][tendons 9][tail]; draws: pos, tendons, vel (tr_env.py:549-610) */
  int nv = c->use_cap_velocity ? 18 : 0;
  double sx = 0, sy = 0;
  for (int i = 0; i < 18; i++) {
    obs[i] = z[i] * c->obs_noise_cap_pos_stdev + obs[i];
    if (i % 3 == 0) sx += obs[i];
    if (i % 3 == 1) sy += obs[i];
  }
  for (int t = 0; t < 9; t++) obs[18 + nv + t] = z[18 + t] * c->obs_noise_tendon_stdev + obs[18 + nv + t];
  for (int i = 0; i < nv; i++) obs[18 + i] = z[27 + i] * c->obs_noise_cap_pos_stdev + obs[18 + i]; /* cap_pos stdev, tr_env.py:604 */
  if (c->desired_action == TGO_ACT_TRACKING || c->desired_action == TGO_ACT_AIMING) { /* tr_env.py:626-639 */
    int k = 18 + nv + 9;
    double tx = obs[k] - sx / 6, ty = obs[k + 1] - sy / 6;
    double n = sqrt(tx * tx + ty * ty);
    obs[k] = tx; obs[k + 1] = ty; obs[k + 2] = atan2(ty / n, tx / n);
  }
}

/* ------------------------------------------------------------------ config */
void tgo_env_default_cfg(tgo_env_cfg* c, int env_kind, int desired_action) {
  memset(c, 0, sizeof(*c));
  c->env_kind = env_kind;
  c->desired_action = desired_action;
  c->desired_direction = 1;
  c->terminate_when_unhealthy = 1;
  c->use_contact_forces = 0;
  c->use_cap_velocity = 1;
  c->use_obs_noise = 0;
  c->use_cap_size_noise = 0; c->cap_size_noise_range[0] = 0.04; c->cap_size_noise_range[1] = 0.09; /* tr_env.py:143,163 */
  c->obs_noise_tendon_stdev = 0.02;
  c->obs_noise_cap_pos_stdev = 0.05;
  c->is_test = 0;
  c->contact_cost_weight = 5e-4;
  c->healthy_reward = 0.1;
  c->contact_force_lo = -1.0; c->contact_force_hi = 1.0;
  if (desired_action == TGO_ACT_TURN) { c->contact_force_lo = -1000.0; c->contact_force_hi = 1000.0; }
  c->min_reset_heading = 0.0; c->max_reset_heading = 2 * PI;
  c->tendon_min_length = -0.45;
  c->frame_skip = 20;
  c->max_episode_steps = 5000;
  double dt = 0.001 * 20;
  if (env_kind == TGO_ENV_TENSEGRITY) { /* tensegrity_env.py:160-181 */
    c->ctrl_cost_weight = 0.001;
    c->tendon_reset_mean = -0.15; c->tendon_reset_stdev = 0.1; c->tendon_max_length = -0.15;
    c->reward_delay_steps = (int)(0.5 / dt);
  } else { /* tr_env.py:137-173 */
    c->ctrl_cost_weight = 0.01;
    c->tendon_reset_mean = 0.15; c->tendon_reset_stdev = 0.2; c->tendon_max_length = 0.15;
    c->reward_delay_steps = (int)(0.02 / dt);
  }
  c->waypt_range[0] = 2.5; c->waypt_range[1] = 3.5;
  c->waypt_angle_range[0] = -PI / 6; c->waypt_angle_range[1] = PI / 6;
  c->ditch_reward_max = 300; c->ditch_reward_stdev = 0.15;
  c->waypt_reward_amplitude = 100; c->waypt_reward_stdev = 0.10;
  c->yaw_reward_weight = 1;
}

int tgo_env_obs_dim(const tgo_env_cfg* c) {
  if (c->env_kind == TGO_ENV_TENSEGRITY) return 39;
  int n = 27;
  if (c->use_cap_velocity) n += 18;
  if (c->desired_action == TGO_ACT_TRACKING || c->desired_action == TGO_ACT_AIMING || c->desired_action == TGO_ACT_VEL_TRACK) n += 3;
  return n;
}

void tgo_env_init(tgo_env* e, const tgo_env_cfg* c) {
  memset(e, 0, sizeof(*e));
  e->cfg = *c;
  if (c->model_kind == 1) tgo_uneven_model(&e->model, c->hf_data, c->hf_nrow, c->hf_ncol);
  else tgo_default_model(&e->model);
  tgo_reset_data(&e->model, &e->data);
  tgo_forward(&e->model, &e->data);
  e->x_velocity = 1; e->y_velocity = 1; /* tensegrity_env.py:204-205 */
  e->obs_dim = tgo_env_obs_dim(c);
}

/* ------------------------------------------------------------------ helpers */
static double env_dt(const tgo_env* e) { return e->model.timestep * e->cfg.frame_skip; }

static void com_xy(const tgo_data* d, double* xy) {
  xy[0] = (d->xpos[0][0] + d->xpos[1][0] + d->xpos[2][0]) / 3;
  xy[1] = (d->xpos[0][1] + d->xpos[1][1] + d->xpos[2][1]) / 3;
}
static void left_right(const tgo_data* d, double* left, double* right) {
  for (int i = 0; i < 3; i++) {
    left[i] = (d->cap_xpos[0][i] + d->cap_xpos[2][i] + d->cap_xpos[4][i]) / 3;
    right[i] = (d->cap_xpos[1][i] + d->cap_xpos[3][i] + d->cap_xpos[5][i]) / 3;
  }
}
static double heading_straight(const tgo_data* d) {
  double l[3], r[3];
  left_right(d, l, r);
  return atan2(-(l[0] - r[0]), l[1] - r[1]);
}
static double angle_normalize(double th) { /* tr_env.py:648-654 */
  while (th > PI) th -= 2 * PI;
  while (th <= -PI) th += 2 * PI;
  return th;
}
static void ring_push(tgo_env* e, double v) {
  e->ring[(e->ring_head + e->ring_len) % TGO_RING] = v;
  e->ring_len++;
}
static double ring_pop(tgo_env* e) {
  double v = e->ring[e->ring_head];
  e->ring_head = (e->ring_head + 1) % TGO_RING;
  e->ring_len--;
  return v;
}
/* scipy Rotation.from_matrix(M).as_quat() -> (x,y,z,w) */
static void mat2quat_scipy(const double* M, double* q) {
  double dec[4] = {M[0], M[4], M[8], M[0] + M[4] + M[8]};
  int ch = 0;
  for (int i = 1; i < 4; i++) if (dec[i] > dec[ch]) ch = i;
  if (ch != 3) {
    int i = ch, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1 - dec[3] + 2 * M[3 * i + i];
    q[j] = M[3 * j + i] + M[3 * i + j];
    q[k] = M[3 * k + i] + M[3 * i + k];
    q[3] = M[3 * k + j] - M[3 * j + k];
  } else {
    q[0] = M[7] - M[5]; q[1] = M[2] - M[6]; q[2] = M[3] - M[1]; q[3] = 1 + dec[3];
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static int is_healthy(const tgo_env* e) {
  const tgo_data* d = &e->data;
  int finite = 1;
  for (int i = 0; i < TGO_NQ; i++) finite &= isfinite(d->qpos[i]) != 0;
  for (int i = 0; i < TGO_NV; i++) finite &= isfinite(d->qvel[i]) != 0;
  int act = e->cfg.desired_action;
  int speed_rule = act == TGO_ACT_TURN || (e->cfg.env_kind == TGO_ENV_TR && act == TGO_ACT_AIMING);
  if (speed_rule) {
    int any = 0;
    for (int i = 0; i < TGO_NV; i++) any |= fabs(d->qvel[i]) > 0.1;
    return finite && any;
  }
  double mv = 0.0001;
  return finite && ((e->x_velocity > mv || e->x_velocity < -mv) || (e->y_velocity > mv || e->y_velocity < -mv));
}
static double ditch_reward(const tgo_env* e, const double* xy) { /* tr_env.py:656-667 */
  const tgo_env_cfg* c = &e->cfg;
  double pv[2] = {e->waypt[0] - e->oripoint[0], e->waypt[1] - e->oripoint[1]};
  double dp = sqrt(pv[0] * pv[0] + pv[1] * pv[1]);
  double pn[2] = {pv[0] / dp, pv[1] / dp};
  double tv[2] = {e->waypt[0] - xy[0], e->waypt[1] - xy[1]};
  double along = tv[0] * pn[0] + tv[1] * pn[1];
  double bx = tv[0] - along * pn[0], by = tv[1] - along * pn[1];
  double bias = sqrt(bx * bx + by * by);
  double ditch = c->ditch_reward_max * (1.0 - fabs(along) / dp) * exp(-bias * bias / (2 * c->ditch_reward_stdev * c->ditch_reward_stdev));
  double dx = xy[0] - e->waypt[0], dy = xy[1] - e->waypt[1];
  double way = c->waypt_reward_amplitude * exp(-(dx * dx + dy * dy) / (2 * c->waypt_reward_stdev * c->waypt_reward_stdev));
  return ditch + way;
}

/* ------------------------------------------------------------------ observations */
void tgo_env_get_obs(const tgo_env* e, double* obs) {
  const tgo_data* d = &e->data;
  if (e->cfg.env_kind == TGO_ENV_TENSEGRITY) { /* tensegrity_env.py:412-430 */
    for (int b = 0; b < 3; b++) {
      /* geom r01 frame = body frame * quat(0 0 0 1): columns 0,1 negated (xml:72) */
      double M[9];
      for (int i = 0; i < 3; i++) { M[3 * i] = -d->xmat[b][3 * i]; M[3 * i + 1] = -d->xmat[b][3 * i + 1]; M[3 * i + 2] = d->xmat[b][3 * i + 2]; }
      mat2quat_scipy(M, obs + 4 * b);
    }
    memcpy(obs + 12, d->qvel, 18 * sizeof(double));
    memcpy(obs + 30, d->ten_length, 9 * sizeof(double));
    return;
  }
  /* tr_env.py:529-646 (noise-free branch) */
  double center[3] = {0, 0, 0};
  for (int k = 0; k < 6; k++) for (int i = 0; i < 3; i++) center[i] += d->cap_xpos[k][i];
  for (int i = 0; i < 3; i++) center[i] /= 6;
  int o = 0;
  for (int k = 0; k < 6; k++) for (int i = 0; i < 3; i++) obs[o++] = d->cap_xpos[k][i] - center[i];
  if (e->cfg.use_cap_velocity) {
    for (int k = 0; k < 6; k++) {
      int b = k / 2;
      const double* v = d->qvel + 6 * b;
      double r[3] = {d->cap_xpos[k][0] - d->xpos[b][0], d->cap_xpos[k][1] - d->xpos[b][1], d->cap_xpos[k][2] - d->xpos[b][2]};
      /* vel_lin + cross(vel_ang, r): qvel[3:6] used as if world-frame (tr_env.py:596-601) */
      obs[o++] = v[0] + (v[4] * r[2] - v[5] * r[1]);
      obs[o++] = v[1] + (v[5] * r[0] - v[3] * r[2]);
      obs[o++] = v[2] + (v[3] * r[1] - v[4] * r[0]);
    }
  }
  for (int t = 0; t < 9; t++) obs[o++] = d->ten_length[t];
  int act = e->cfg.desired_action;
  if (act == TGO_ACT_TRACKING || act == TGO_ACT_AIMING) {
    double tv[2] = {e->waypt[0] - center[0], e->waypt[1] - center[1]};
    double n = sqrt(tv[0] * tv[0] + tv[1] * tv[1]);
    obs[o++] = tv[0]; obs[o++] = tv[1];
    obs[o++] = atan2(tv[1] / n, tv[0] / n);
  } else if (act == TGO_ACT_VEL_TRACK) {
    obs[o++] = e->lin_vel_cmd[0]; obs[o++] = e->lin_vel_cmd[1]; obs[o++] = e->ang_vel_cmd;
  }
  if (e->cfg.use_obs_noise) { /* tr_env.py:524-527,869-872: step() and reset() return observation_with_noise */
    double z[48];
    tgo_obs_noise_draws(e->rng_seed, e->rng_key, e->rng_episode, e->elapsed, z);
    tgo_apply_obs_noise(&e->cfg, z, obs);
  }
}

/* ------------------------------------------------------------------ step */
static void env_step_inner(tgo_env* e, const double* action, double* reward, int* terminated, double* info) {
  tgo_data* d = &e->data;
  const tgo_env_cfg* c = &e->cfg;
  double dt = env_dt(e);
  int act = c->desired_action;
  double xy_before[2], xy_after[2];
  com_xy(d, xy_before);
  double psi_before = heading_straight(d);
  if (c->env_kind == TGO_ENV_TR) { /* _action_filter, tr_env.py:345,680-683 */
    for (int i = 0; i < 6; i++) d->ctrl[i] = d->ctrl[i] + 1 * (action[i] - d->ctrl[i]) * dt;
  } else {
    for (int i = 0; i < 6; i++) d->ctrl[i] = action[i];
  }
  tgo_step(&e->model, d, c->frame_skip);
  com_xy(d, xy_after);
  e->x_velocity = (xy_after[0] - xy_before[0]) / dt;
  e->y_velocity = (xy_after[1] - xy_before[1]) / dt;
  double left[3], right[3];
  left_right(d, left, right);
  double psi_after = atan2(-(left[0] - right[0]), left[1] - right[1]);
  if (c->env_kind == TGO_ENV_TENSEGRITY && act == TGO_ACT_TURN) /* tensegrity_env.py:320-322 */
    psi_after = atan2(right[1] - left[1], right[0] - left[0]);

  double ctrl_cost_full;
  if (c->env_kind == TGO_ENV_TENSEGRITY) {
    double s = 0;
    for (int i = 0; i < 6; i++) s += action[i] * action[i];
    ctrl_cost_full = c->ctrl_cost_weight * s;
  } else {
    double s = 0;
    for (int i = 0; i < 6; i++) { double v = action[i] + 0.5 - d->ten_length[i]; s += v * v; }
    ctrl_cost_full = c->ctrl_cost_weight * s;
  }

  double forward_reward = 0, ctrl_cost = 0, healthy_reward = 0;
  int term = 0;
  double hr_prop = ((is_healthy(e) || c->terminate_when_unhealthy) ? 1.0 : 0.0) * c->healthy_reward;
  int term_prop;
  if (act == TGO_ACT_TURN) {
    ring_push(e, psi_after);
    if (e->ring_len > c->reward_delay_steps) {
      double old = ring_pop(e);
      if (psi_after < -PI / 2 && old > PI / 2) psi_after = 2 * PI + psi_after;
      else if (psi_after > PI / 2 && old < -PI / 2) psi_after = -2 * PI + psi_after;
      double delta = (psi_after - old) / (dt * c->reward_delay_steps);
      forward_reward = delta * c->desired_direction;
      ctrl_cost = ctrl_cost_full;
    }
    healthy_reward = c->terminate_when_unhealthy ? hr_prop : 0;
  } else if (act == TGO_ACT_STRAIGHT) {
    double dx = xy_after[0] - xy_before[0], dy = xy_after[1] - xy_before[1];
    double psi_move = atan2(dy, dx);
    double psi_diff = fabs(psi_move - e->reset_psi);
    forward_reward = c->desired_direction * (sqrt(dx * dx + dy * dy) * cos(psi_diff) / dt);
    ctrl_cost = ctrl_cost_full;
    healthy_reward = c->terminate_when_unhealthy ? hr_prop : 0;
  } else if (act == TGO_ACT_AIMING) {
    double td[2] = {e->waypt[0] - xy_before[0], e->waypt[1] - xy_before[1]};
    double n = sqrt(td[0] * td[0] + td[1] * td[1]);
    double target_psi = atan2(td[1] / n, td[0] / n);
    double np_ = angle_normalize(target_psi - psi_after);
    ring_push(e, np_);
    if (e->ring_len > c->reward_delay_steps) {
      double old = ring_pop(e);
      double delta = -(fabs(np_) - fabs(old)) / (dt * c->reward_delay_steps);
      forward_reward = delta * c->yaw_reward_weight;
    }
    ctrl_cost = ctrl_cost_full;
    healthy_reward = 0;
  } else if (act == TGO_ACT_TRACKING) {
    forward_reward = ditch_reward(e, xy_after) - ditch_reward(e, xy_before);
    ctrl_cost = ctrl_cost_full;
    healthy_reward = 0;
  } else { /* vel_track */
    double ang = angle_normalize(psi_after - psi_before) / dt;
    double ex = e->x_velocity - e->lin_vel_cmd[0], ey = e->y_velocity - e->lin_vel_cmd[1];
    double lin_err2 = ex * ex + ey * ey;
    double ang_err = ang - e->ang_vel_cmd;
    forward_reward = 1.0 * exp(-5.0 * lin_err2) + 0.5 * exp(-7.0 * ang_err * ang_err);
    ctrl_cost = ctrl_cost_full;
    healthy_reward = c->terminate_when_unhealthy ? hr_prop : 0;
  }
  term_prop = c->terminate_when_unhealthy ? !is_healthy(e) : 0;
  term = term_prop;
  if (c->env_kind == TGO_ENV_TR && (act == TGO_ACT_AIMING || act == TGO_ACT_TRACKING) && e->step_num > 1000) term = 1;
  for (int b = 0; b < 4; b++)
    for (int i = 0; i < 6; i++)
      if (d->cfrc_ext[b][i] > 1500 || d->cfrc_ext[b][i] < -1500) term = 1;
  double costs = ctrl_cost;
  double rewards = forward_reward + healthy_reward;
  if (info) {
    memset(info, 0, TGO_INFO_K * sizeof(double));
    info[TGO_INFO_REWARD_FORWARD] = forward_reward;
    info[TGO_INFO_REWARD_CTRL] = -ctrl_cost;
    info[TGO_INFO_REWARD_SURVIVE] = healthy_reward;
    info[TGO_INFO_X_POSITION] = xy_after[0];
    info[TGO_INFO_Y_POSITION] = xy_after[1];
    info[TGO_INFO_PSI] = psi_after;
    info[TGO_INFO_DISTANCE] = sqrt(xy_after[0] * xy_after[0] + xy_after[1] * xy_after[1]);
    info[TGO_INFO_X_VELOCITY] = e->x_velocity;
    info[TGO_INFO_Y_VELOCITY] = e->y_velocity;
    info[TGO_INFO_FORWARD_REWARD] = forward_reward;
    info[TGO_INFO_WAYPT_X] = e->waypt[0]; info[TGO_INFO_WAYPT_Y] = e->waypt[1];
    info[TGO_INFO_ORI_X] = e->oripoint[0]; info[TGO_INFO_ORI_Y] = e->oripoint[1];
    for (int t = 0; t < TGO_NTEN; t++) info[TGO_INFO_TENDON_LENGTH + t] = d->ten_length[t]; /* tr_env.py:506 */
    double bar = 0; /* run.py:149-155: contacts with neither geom the floor, |mj_contactForce[0:3]| summed */
    for (int ci = 0; ci < d->ncon; ci++) {
      const tgo_contact* k = &d->con[ci];
      if (k->body1 >= 0 && k->body2 >= 0) bar += sqrt(k->force[0] * k->force[0] + k->force[1] * k->force[1] + k->force[2] * k->force[2]);
    }
    info[TGO_INFO_BAR_CONTACT] = bar;
  }
  if (c->use_contact_forces) {
    double s = 0;
    for (int b = 0; b < 4; b++)
      for (int i = 0; i < 6; i++) {
        double v = d->cfrc_ext[b][i];
        v = v < c->contact_force_lo ? c->contact_force_lo : (v > c->contact_force_hi ? c->contact_force_hi : v);
        s += v * v;
      }
    double contact_cost = c->contact_cost_weight * s;
    costs += contact_cost;
    if (info) info[TGO_INFO_REWARD_CTRL] = -contact_cost; /* tensegrity_env.py:404 overwrite */
  }
  *reward = rewards - costs;
  *terminated = term;
  if (c->env_kind == TGO_ENV_TR) e->step_num += 1;
}

void tgo_env_step(tgo_env* e, const float action[6], double* obs, double* reward, int* terminated, int* truncated, double* info) {
  double a[6];
  for (int i = 0; i < 6; i++) a[i] = (double)action[i];
  env_step_inner(e, a, reward, terminated, info);
  e->elapsed += 1;
  tgo_env_get_obs(e, obs); /* observation k of the episode (0 = reset) keys the observation noise */
  *truncated = e->elapsed >= e->cfg.max_episode_steps; /* gym TimeLimit */
  e->ep_return += *reward;
  if (info) { info[TGO_INFO_EP_RETURN] = e->ep_return; info[TGO_INFO_EP_LENGTH] = e->elapsed; }
}

/* ------------------------------------------------------------------ reset */
void tgo_yaw_pose(const double qpos_in[TGO_NQ], double theta, double qpos_out[TGO_NQ]) {
  /* tensegrity_env.py:450-483: p <- Rz(theta) p ; q <- from_euler(euler(q) + [0,0,theta]) == qz(theta) * q */
  double c = cos(theta), s = sin(theta), ch = cos(0.5 * theta), sh = sin(0.5 * theta);
  for (int b = 0; b < 3; b++) {
    const double* q = qpos_in + 7 * b;
    double* o = qpos_out + 7 * b;
    o[0] = c * q[0] - s * q[1]; o[1] = s * q[0] + c * q[1]; o[2] = q[2];
    double w = q[3], x = q[4], y = q[5], z = q[6];
    o[3] = ch * w - sh * z; o[4] = ch * x - sh * y; o[5] = ch * y + sh * x; o[6] = ch * z + sh * w;
  }
}

void tgo_reset_noise_draws(uint64_t seed, uint64_t env_id, uint64_t episode, double scale, double dq[TGO_NQ], double dv[TGO_NV]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)episode, 0};
  uint32_t o[4];
  for (int j = 0; j < 11; j++) {
    ctr[3] = 16 + j;
    tgo_philox4x32(ctr, key, o);
    dq[2 * j] = -scale + u53(o[0], o[1]) * 2.0 * scale;
    if (2 * j + 1 < TGO_NQ) dq[2 * j + 1] = -scale + u53(o[2], o[3]) * 2.0 * scale;
  }
  for (int j = 0; j < 9; j++) {
    ctr[3] = 32 + j;
    tgo_philox4x32(ctr, key, o);
    double u1 = 1.0 - u53(o[0], o[1]), u2 = u53(o[2], o[3]);
    double r = sqrt(-2.0 * log(u1));
    dv[2 * j] = scale * r * cos(2 * PI * u2);
    dv[2 * j + 1] = scale * r * sin(2 * PI * u2);
  }
}

void tgo_env_reset(tgo_env* e, int pose_idx, double theta, const double tendons[6], const double u_way[2], double* obs) {
  tgo_env_reset_noise(e, pose_idx, theta, tendons, u_way, NULL, NULL, obs);
}

void tgo_env_reset_noise(tgo_env* e, int pose_idx, double theta, const double tendons[6], const double u_way[2], const double dq[TGO_NQ],
                         const double dv[TGO_NV], double* obs) {
  const tgo_env_cfg* c = &e->cfg;
  tgo_data* d = &e->data;
  int act = c->desired_action;
  /* MujocoEnv.reset -> mj_resetData */
  tgo_reset_data(&e->model, d);
  double qpos[TGO_NQ];
  if (c->env_kind == TGO_ENV_TR) memcpy(qpos, tgo_rolling_qpos[pose_idx], sizeof(qpos));
  else memcpy(qpos, e->model.qpos0, sizeof(qpos));
  if (dq) { /* tensegrity_env.py:439-441: the perturbed quaternion is re-normalised by scipy's Rotation.from_quat (:451-456) */
    for (int i = 0; i < TGO_NQ; i++) qpos[i] += dq[i];
    for (int b = 0; b < 3; b++) {
      double* q = qpos + 7 * b + 3;
      double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int i = 0; i < 4; i++) q[i] /= nn;
    }
  }
  tgo_yaw_pose(qpos, theta, d->qpos);
  if (dv) memcpy(d->qvel, dv, sizeof(double) * TGO_NV); /* qvel = init_qvel + s * N(0,1): not rotated (tensegrity_env.py:442-445,484) */
  tgo_forward(&e->model, d); /* set_state -> mj_forward */
  if (c->env_kind == TGO_ENV_TENSEGRITY) {
    double r; int t;
    for (int i = 0; i < 50; i++) env_step_inner(e, tendons, &r, &t, NULL); /* tensegrity_env.py:495-496 */
  } else {
    for (int i = 0; i < 6; i++) d->ctrl[i] = tendons[i];
    for (int i = 0; i < 50; i++) tgo_step(&e->model, d, c->frame_skip); /* tr_env.py:811-812 */
  }
  double left[3], right[3];
  left_right(d, left, right);
  e->reset_psi = atan2(-(left[0] - right[0]), left[1] - right[1]);
  if (c->env_kind == TGO_ENV_TR) {
    if (act == TGO_ACT_TRACKING) { /* tr_env.py:827-839 */
      e->oripoint[0] = (left[0] + right[0]) / 2; e->oripoint[1] = (left[1] + right[1]) / 2;
      double len = c->waypt_range[0] + u_way[0] * (c->waypt_range[1] - c->waypt_range[0]);
      double yaw = c->waypt_angle_range[0] + u_way[1] * (c->waypt_angle_range[1] - c->waypt_angle_range[0]) + e->reset_psi;
      if (c->is_test) {
        len = 0.5 * c->waypt_range[1] + 0.5 * c->waypt_range[0];
        yaw = (0.5 * c->waypt_angle_range[1] + 0.5 * c->waypt_angle_range[0]) + e->reset_psi;
      }
      e->waypt[0] = e->oripoint[0] + len * cos(yaw); e->waypt[1] = e->oripoint[1] + len * sin(yaw);
    } else if (act == TGO_ACT_AIMING) { /* tr_env.py:842-855, typo in oripoint[0] kept */
      e->oripoint[0] = left[0] + right[0] / 2; e->oripoint[1] = (left[1] + right[1]) / 2;
      double len = c->waypt_range[0] + u_way[0] * (c->waypt_range[1] - c->waypt_range[0]);
      double yaw = -PI + u_way[1] * (2 * PI) + e->reset_psi;
      if (c->is_test) {
        len = 0.5 * c->waypt_range[1] + 0.5 * c->waypt_range[0];
        yaw = (0.75 * PI + 0.25 * (-PI)) + e->reset_psi;
      }
      e->waypt[0] = e->oripoint[0] + len * cos(yaw); e->waypt[1] = e->oripoint[1] + len * sin(yaw);
      if (c->is_test) { e->waypt[0] = 0; e->waypt[1] = 0; }
    } else if (act == TGO_ACT_VEL_TRACK) {
      e->lin_vel_cmd[0] = 0.5 * cos(e->reset_psi); e->lin_vel_cmd[1] = 0.5 * sin(e->reset_psi);
      e->ang_vel_cmd = 0.0;
    }
    e->step_num = 0;
    if (act == TGO_ACT_TURN || act == TGO_ACT_AIMING) { /* tr_env.py:864-866 */
      double r; int t;
      for (int i = 0; i < c->reward_delay_steps; i++) env_step_inner(e, tendons, &r, &t, NULL);
    }
  }
  e->elapsed = 0;
  e->ep_return = 0;
  if (obs) tgo_env_get_obs(e, obs);
}

double tgo_cap_size_draw(uint64_t seed, uint64_t env_id, uint64_t episode, double lo, double hi) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)episode, 5};
  uint32_t o[4];
  tgo_philox4x32(ctr, key, o);
  return lo + u53(o[0], o[1]) * (hi - lo);
}

void tgo_env_reset_seeded(tgo_env* e, uint64_t seed, uint64_t env_id, uint64_t episode, double* obs) {
  const tgo_env_cfg* c = &e->cfg;
  int pose; double theta, tendons[6], uw[2];
  /* tr_env.py:712-713: the cap size is drawn first, before the pose */
  if (c->use_cap_size_noise) e->model.cap_radius = tgo_cap_size_draw(seed, env_id, episode, c->cap_size_noise_range[0], c->cap_size_noise_range[1]);
  tgo_rng_draws(seed, env_id, episode, c->env_kind, c->tendon_reset_mean, c->tendon_reset_stdev, c->tendon_min_length,
                c->tendon_max_length, c->min_reset_heading, c->max_reset_heading, &pose, &theta, tendons, uw);
  e->rng_seed = seed; e->rng_key = env_id; e->rng_episode = episode;
  if (c->reset_noise_scale != 0.0) {
    double dq[TGO_NQ], dv[TGO_NV];
    tgo_reset_noise_draws(seed, env_id, episode, c->reset_noise_scale, dq, dv);
    tgo_env_reset_noise(e, pose, theta, tendons, uw, dq, dv, obs);
  } else {
    tgo_env_reset(e, pose, theta, tendons, uw, obs);
  }
}
