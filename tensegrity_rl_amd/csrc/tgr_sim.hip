// tgr_sim.hip -- kernels + C-ABI of the batched tensegrity simulator for MI355X (gfx950).
//
// Layout in HBM: structure-of-arrays, field-major:  st[field][env]  so that the 64 lanes of a
// wave (64 consecutive envs) read/write 256 contiguous bytes per field.  One env-step = ONE
// kernel launch: state is read once, all 20 sub-steps run out of registers/LDS, state + outputs
// are written once.  Observations leave through an LDS transpose so the [N][D] row-major
// buffer SB3 wants is still written with coalesced stores.
//
// Reference behaviour reproduced: tensegrity_env.py:291-512 and tr_env.py:327-872 on top of
// mujoco 2.3.7 mj_step (see include/tensegrity.h for the entry-point mapping).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tensegrity.h"
#include "tgr_device.h"
#include "tgr_row.h"

#ifndef TGR_STAMP_BLOCKS
#define TGR_STAMP_BLOCKS (blockIdx.x < 64)  // diagnostic build: which blocks report their phase cycles
#endif
namespace tgr {

// ------------------------------------------------------------------ state layout
enum : int {
  F_QPOS = 0, F_QVEL = 21, F_WARM = 39, F_CTRL = 57, F_XYLAG = 63, F_PSILAG = 65, F_RESETPSI = 66,
  F_EPRET = 67, F_WAYPT = 68, F_ORI = 70, F_VCMD = 72, F_ACT = 75, F_CAPD = 81, NF = 82
};
enum : int { I_STEPNUM = 0, I_ELAPSED, I_RINGLEN, I_RINGHEAD, I_EPISODE, I_PHASE, I_SRC, I_SRCEP, I_HALF, NI };  // I_HALF: row kernels, paced donor rows
constexpr int RING = 32;
// rows of the fp64 side array hp[HP_ROWS][m] (fp32 handles on the row kernels): qpos, and what the env layer keeps of ABSOLUTE positions --
// the lagged centre (xy_position_before), tr_env's way-point and origin.  In fp32 a robot 30 m from the origin would resolve them to 2e-6 m:
// 1e-4 m/s of x / y velocity, the size of the is_healthy threshold (tensegrity_env.py:279).  st keeps rounded copies.
enum : int { HP_QPOS = 0, HP_XYLAG = 21, HP_WAYPT = 23, HP_ORI = 25, HP_ROWS = 27 };
static_assert(HP_WAYPT == HP_XYLAG + 2 && HP_ORI == HP_XYLAG + 4 && HP_ROWS == HP_XYLAG + 6, "set_state's hp rows");
static_assert(RING == TGR_RING_CAPACITY && TGR_AUX_K == TGR_AUX_RING + RING + 1, "aux layout of tgr_get_state / tgr_set_state");
constexpr double PI = 3.14159265358979323846;

template <typename T> struct EnvCfg {
  int env_kind, action, terminate_when_unhealthy, use_contact_forces, use_cap_velocity, is_test;
  int reward_delay_steps, frame_skip, max_episode_steps, obs_dim, use_obs_noise, model_kind;
  T noise_tendon, noise_cap;
  T direction, ctrl_cost_weight, contact_cost_weight, healthy_reward, contact_lo, contact_hi, dt;
  T ditch_max, ditch_std, waypt_amp, waypt_std, yaw_weight;
  double waypt_range[2], waypt_angle_range[2];
  double min_heading, max_heading, ten_mean, ten_std, ten_max, ten_min, reset_noise;
  unsigned long long seed, first_env_id;
  // reset streams (tgr_config.reset_stream_envs): envs per class (0: one pool per handle), donor lanes per class
  int stream_envs, stream_donors;
  // cap-size noise (tr_env.py:143,163,685-706): per episode one radius ~ U(cap_lo, cap_hi) for the six cap spheres; cap_nominal = the model's
  int cap_noise; double cap_lo, cap_hi, cap_nominal;
};

// counters[STAMP_SLOT + 2 * (launch & 1) + {0, 1}]: when the live-row launch of that step started / when its last wavefront ended
// (launch index << 40 | 100 MHz clock) -- what the donor launches stagger themselves by (KArgs::stagger_pct)
constexpr int STAMP_SLOT = 150;
#ifndef TGR_DONOR_STAGGER_PCT
#define TGR_DONOR_STAGGER_PCT 42
#endif
template <typename T> struct KArgs {
  Params<T> P;
  EnvCfg<T> E;
  int n, m;  // live envs, total lanes (live + donors)
  int slow_e;  // row kernels: donor rows from this index on (a multiple of 4) settle in half quanta -- see step_kernel_row; >= m: none
  T* st; int* ist; T* ring;
  double* hp;  // fp32 handles stepped by the row kernels: qpos (and the env layer's absolute positions) in fp64, [HP_ROWS][m] (the pose is fp64
               // inside the kernels; rounding it to fp32 between env-steps alone moved a cap across dist = 0 in 0.1..0.4 % of env-steps).
               // null: the pose lives in st only
  const float* actions;
  T* obs; T* reward; uint8_t* terminated; uint8_t* truncated; T* terminal_obs; T* info;
  int* list_count; int* list; int* rlist;  // envs to (re)settle / ready donors
  int* list_ep;   // reset streams: the stream episode a starved env is to be settled with (parallel to list), or null
  int* cls;       // reset streams: per class {next episode to start settling, next episode to hand out}
  int* holder;    // reset streams: per class, which donor lane holds episode (ep % stream_table)
  int stream_table;
  // donor stream (tgr_config.donor_stream): the donor lanes are stepped by their own launches on a second, lower-priority HIP stream, one
  // per tgr_step, concurrently with the live launch -- no launch boundary is shared.  The two sides never write the same word:
  //   dready[lane]   (donor side)  launch index at which the lane's episode became ready, 0 while it is settling
  //   dapplied[lane] (donor side)  tag of the last command the lane has carried out
  //   dcmd[lane]     (match side)  episode << 32 | tag: "start a new episode" (tag = index of the step whose match kernel issued it)
  //   consumed[lane] (match side)  the dready value of the episode that was handed out last
  // Step i's match kernel waits (HIP event) for donor launch i - 1 and only trusts dready <= i - 1; donor launch j waits for the match kernel of
  // step j - 1 and only carries out commands with tag <= j - 1.  What either side sees is thus a function of launch indices alone -- results
  // do not depend on how the two streams happen to overlap.
  int async, launch_idx, lane_base, lane_limit;
  // STAGGER of the two streams.  A step is the live-row launch + 45 us of serial work on the caller's stream (match kernel, launch gaps) next
  // to the donor-row launch on the donor stream.  Started TOGETHER -- which is what happens whenever the donor launch is the shorter of the two:
  // it starts the moment the previous step's match kernel ends -- both launches' tails coincide and nothing covers the serial section (0.94 ms
  // per step); staggered, either stream's tail is covered by the other's bulk (0.87 ms).  So: the live launch stamps its start and
  // every live wavefront its end (counters[STAMP_SLOT..]); a donor wavefront that finds itself starting within stagger_pct / 4 percent of the
  // previous live launch's duration after this step's live launch sleeps (s_sleep: it holds a wave slot, not the ALU) until stagger_pct percent
  // of that duration have passed, for at most 0.6 ms; one that starts later already is staggered and goes ahead.  Timing only: nothing a
  // kernel computes depends on it.  0: off.  (fp32 flat-floor handles: the instantiation with two wavefronts per SIMD.)
  int stagger_pct;
  unsigned long long* dcmd; int* dapplied; int* dready; int* consumed;
  uint8_t* doneflag;
  T* dobs;  // reset observations of the donors [m-n][D]
  unsigned long long* counters;
  int auto_reset;
  T* obs_reset;  // where the settle kernel writes reset observations ([N][D]) or null
  T* obs_clean;  // info["real_observation"] of tr_env.py:507: the noise-free observation of the step ([N][D]) or null
  unsigned* dbg;  // row kernels: sub-step trace of the live envs ([N][row::TRACE_WORDS], tgr_bind_substep_trace) or null
};

// an absolute position of the env layer: from the fp64 side array where the handle has one, else from st; stores keep both up to date
template <typename T> TGR_DEV double aux_ld(const KArgs<T>& k, int hp_row, int st_field, int e) {
  return k.hp ? k.hp[(size_t)hp_row * k.m + e] : (double)k.st[(size_t)st_field * k.m + e];
}
template <typename T> TGR_DEV void aux_st(const KArgs<T>& k, int hp_row, int st_field, int e, double v) {
  if (k.hp) k.hp[(size_t)hp_row * k.m + e] = v;
  k.st[(size_t)st_field * k.m + e] = T(v);
}

__constant__ double c_qpos0[21] = {0.08369179, -0.28792231, 0.24830847, -0.49145555, 0.7539914, -0.27511722, -0.33805166,
                                   0.14497616, -0.19291743, 0.35052097, -0.84766041, 0.27950622, 0.45085889, 0.00862359,
                                   0.04557825, -0.29876206, 0.39531985, -0.35798606, -0.47531391, 0.72471075, 0.34744352};
// 3prism_jonathan_steady_side_uneven_ground.xml:65,84,102
__constant__ double c_qpos0_uneven[21] = {0.25551711, -0.00069342, 0.22404039, -0.49720971, 0.24315431, 0.75327284, -0.35530059,
                                          0.14409445, 0.0654207, 0.33662589, 0.42572066, 0.01379464, -0.53972521, 0.72613244,
                                          0.28544944, -0.04883333, 0.38591159, 0.137357, 0.06898275, -0.85996553, 0.48665565};
// tr_env.py:723-728
__constant__ double c_rolling[6][21] = {
    {0.07900689, -0.32670045, 0.23079722, 0.49365198, -0.74001353, 0.26668361, 0.37090101, 0.13713385, -0.24342633, 0.32722167, 0.82936968, -0.31256817, -0.46189217, -0.03320677, 0.04903377, -0.3421725, 0.36675097, 0.33407281, 0.43794432, -0.72515863, -0.41321313},
    {0.15521685, -0.20651043, 0.38922255, 0.85639289, -0.26723449, -0.44110818, -0.02450564, 0.02999107, -0.33576412, 0.43868814, 0.33839518, 0.48544838, -0.73094128, -0.33993149, 0.08083394, -0.31942006, 0.25783949, 0.51726058, -0.74281033, 0.29432583, 0.30667022},
    {0.02985312, -0.33588999, 0.43866597, 0.33840617, 0.48522953, -0.73107566, -0.33994403, 0.08072907, -0.31942136, 0.25766037, 0.51740763, -0.74276722, 0.29421311, 0.30663471, 0.15537661, -0.20664637, 0.38923648, 0.85640002, -0.26722239, -0.44110397, -0.02446392},
    {0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798, 0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067, 0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277},
    {0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067, 0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277, 0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798},
    {0.27475478, 0.2682452, 0.4387596, 0.47235593, 0.87732918, -0.01675131, 0.08302277, 0.24191878, 0.30939576, 0.25838614, 0.04211683, -0.66689235, -0.44050762, 0.59952798, 0.1105878, 0.33967509, 0.38925944, 0.50825334, 0.20884794, -0.4715363, 0.68972067}};

// ------------------------------------------------------------------ counter-based RNG (philox4x32-10)
__device__ inline void philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* o) {
  for (int r = 0; r < 10; r++) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ inline double u53(unsigned a, unsigned b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; }

struct Draws { int pose; double theta; double tendons[6]; double u_way[2]; double u_cap; };
template <typename T> __device__ inline Draws reset_draws(const EnvCfg<T>& E, unsigned long long env_id, unsigned long long episode) {
  Draws d;
  unsigned k0 = (unsigned)E.seed, k1 = (unsigned)(E.seed >> 32);
  unsigned c0 = (unsigned)env_id, c1 = (unsigned)(env_id >> 32), c2 = (unsigned)episode;
  unsigned o[4];
  philox(c0, c1, c2, 0, k0, k1, o);
  d.pose = (int)(o[0] % 6u);
  d.theta = E.min_heading + u53(o[1], o[2]) * (E.max_heading - E.min_heading);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    philox(c0, c1, c2, 1 + k, k0, k1, o);
    double u1 = 1.0 - u53(o[0], o[1]), u2 = u53(o[2], o[3]);
    double r = sqrt(-2.0 * log(u1));
    double z0 = r * cos(2 * PI * u2), z1 = r * sin(2 * PI * u2);
    double t0 = z0 * E.ten_std + E.ten_mean, t1 = z1 * E.ten_std + E.ten_mean;
    d.tendons[2 * k] = t0 > E.ten_max ? E.ten_max : (t0 < E.ten_min ? E.ten_min : t0);
    d.tendons[2 * k + 1] = t1 > E.ten_max ? E.ten_max : (t1 < E.ten_min ? E.ten_min : t1);
  }
  philox(c0, c1, c2, 4, k0, k1, o);
  d.u_way[0] = u53(o[0], o[1]);
  d.u_way[1] = u53(o[2], o[3]);
  d.u_cap = 0.0;
  if (E.cap_noise) {
    philox(c0, c1, c2, 5, k0, k1, o);
    d.u_cap = u53(o[0], o[1]);
  }
  return d;
}

// ------------------------------------------------------------------ state <-> registers
template <typename T> TGR_DEV void load_sim(const T* st, int n, int e, Sim<T>& S) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    S.x[b] = mk(st[(F_QPOS + 7 * b + 0) * n + e], st[(F_QPOS + 7 * b + 1) * n + e], st[(F_QPOS + 7 * b + 2) * n + e]);
#pragma unroll
    for (int i = 0; i < 4; i++) S.q[b][i] = st[(F_QPOS + 7 * b + 3 + i) * n + e];
    normalize_quat(S.q[b]);
    Rot<T> R = quat_rot(S.q[b]);
    S.v[b] = mk(st[(F_QVEL + 6 * b + 0) * n + e], st[(F_QVEL + 6 * b + 1) * n + e], st[(F_QVEL + 6 * b + 2) * n + e]);
    V3<T> wb = mk(st[(F_QVEL + 6 * b + 3) * n + e], st[(F_QVEL + 6 * b + 4) * n + e], st[(F_QVEL + 6 * b + 5) * n + e]);
    S.w[b] = rot_apply(R, wb);
    S.al[b] = mk(st[(F_WARM + 6 * b + 0) * n + e], st[(F_WARM + 6 * b + 1) * n + e], st[(F_WARM + 6 * b + 2) * n + e]);
    V3<T> ab = mk(st[(F_WARM + 6 * b + 3) * n + e], st[(F_WARM + 6 * b + 4) * n + e], st[(F_WARM + 6 * b + 5) * n + e]);
    S.aa[b] = rot_apply(R, ab);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { S.ctrl[i] = st[(F_CTRL + i) * n + e]; S.act[i] = st[(F_ACT + i) * n + e]; }
  S.capd = st[F_CAPD * n + e];
}
// wb_out: body-frame angular velocity (what qvel stores), needed by the tr_env observation
template <typename T> TGR_DEV void store_sim(T* st, int n, int e, Sim<T>& S, const Lag<T>& lag, V3<T>* wb_out) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    normalize_quat(S.q[b]);
    Rot<T> R = quat_rot(S.q[b]);
    st[(F_QPOS + 7 * b + 0) * n + e] = S.x[b].x; st[(F_QPOS + 7 * b + 1) * n + e] = S.x[b].y; st[(F_QPOS + 7 * b + 2) * n + e] = S.x[b].z;
#pragma unroll
    for (int i = 0; i < 4; i++) st[(F_QPOS + 7 * b + 3 + i) * n + e] = S.q[b][i];
    V3<T> wb = rot_applyT(R, S.w[b]);
    wb_out[b] = wb;
    st[(F_QVEL + 6 * b + 0) * n + e] = S.v[b].x; st[(F_QVEL + 6 * b + 1) * n + e] = S.v[b].y; st[(F_QVEL + 6 * b + 2) * n + e] = S.v[b].z;
    st[(F_QVEL + 6 * b + 3) * n + e] = wb.x; st[(F_QVEL + 6 * b + 4) * n + e] = wb.y; st[(F_QVEL + 6 * b + 5) * n + e] = wb.z;
    // warm start is kept the way mujoco keeps it: body-frame angular part of the last forward pass
    Rot<T> Rl = quat_rot(lag.q[b]);
    V3<T> ab = rot_applyT(Rl, S.aa[b]);
    st[(F_WARM + 6 * b + 0) * n + e] = S.al[b].x; st[(F_WARM + 6 * b + 1) * n + e] = S.al[b].y; st[(F_WARM + 6 * b + 2) * n + e] = S.al[b].z;
    st[(F_WARM + 6 * b + 3) * n + e] = ab.x; st[(F_WARM + 6 * b + 4) * n + e] = ab.y; st[(F_WARM + 6 * b + 5) * n + e] = ab.z;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { st[(F_CTRL + i) * n + e] = S.ctrl[i]; st[(F_ACT + i) * n + e] = S.act[i]; }
  st[F_CAPD * n + e] = S.capd;
}

template <typename T> TGR_DEV T atan2_(T y, T x) { return sizeof(T) == 4 ? T(atan2f(float(y), float(x))) : T(atan2(double(y), double(x))); }
template <typename T> TGR_DEV T cos_(T x) { return sizeof(T) == 4 ? T(cosf(float(x))) : T(cos(double(x))); }

template <typename T> TGR_DEV void caps_of(const Params<T>& P, const Lag<T>& lag, V3<T>* cap) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    cap[2 * b] = lag.x[b] + P.bar_half * lag.z[b];
    cap[2 * b + 1] = lag.x[b] - P.bar_half * lag.z[b];
  }
}
template <typename T> TGR_DEV void left_right(const V3<T>* cap, V3<T>& l, V3<T>& r) {
  T third = T(1.0 / 3.0);
  l = third * (cap[0] + cap[2] + cap[4]);
  r = third * (cap[1] + cap[3] + cap[5]);
}

template <typename T> struct ModeState;
// observation of the env into o[0..D)
template <typename T>
TGR_DEV void make_obs(const Params<T>& P, const EnvCfg<T>& E, const Sim<T>& S, const Lag<T>& lag, const V3<T>* wb, const ModeState<T>& ms, T* o) {
  // o points into LDS (this lane's row of the block's observation tile): runtime offsets are plain address arithmetic there
  if (E.env_kind == TGR_ENV_TENSEGRITY) {
    // tensegrity_env.py:412-430: scipy quaternion (x,y,z,w) of the r-cylinder geom frame = body * quat(0,0,0,1)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      T w = lag.q[b][0], x = lag.q[b][1], y = lag.q[b][2], z = lag.q[b][3];
      T g[4] = {y, -x, w, -z};
      T m = g[0] * g[0]; T sg = g[0];
#pragma unroll
      for (int i = 1; i < 4; i++) { T a = g[i] * g[i]; bool big = a > m; m = big ? a : m; sg = big ? g[i] : sg; }
      T s = sg < T(0) ? T(-1) : T(1);
#pragma unroll
      for (int i = 0; i < 4; i++) o[4 * b + i] = s * g[i];
      o[12 + 6 * b + 0] = S.v[b].x; o[12 + 6 * b + 1] = S.v[b].y; o[12 + 6 * b + 2] = S.v[b].z;
      o[12 + 6 * b + 3] = wb[b].x; o[12 + 6 * b + 4] = wb[b].y; o[12 + 6 * b + 5] = wb[b].z;
    }
#pragma unroll
    for (int t = 0; t < 9; t++) o[30 + t] = lag.tenL[t];
    return;
  }
  // tr_env.py:529-646
  V3<T> cap[6];
  caps_of(P, lag, cap);
  V3<T> c = T(1.0 / 6.0) * (((cap[0] + cap[1]) + (cap[2] + cap[3])) + (cap[4] + cap[5]));
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) { V3<T> r = cap[i] - c; o[k++] = r.x; o[k++] = r.y; o[k++] = r.z; }
  if (E.use_cap_velocity) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int b = i / 2;
      V3<T> r = cap[i] - lag.x[b];
      V3<T> vv = S.v[b] + cross(wb[b], r);  // body-frame w used as if world (tr_env.py:596-601)
      o[k++] = vv.x; o[k++] = vv.y; o[k++] = vv.z;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; t++) o[k++] = lag.tenL[t];
  if (E.action == TGR_ACT_TRACKING || E.action == TGR_ACT_AIMING) {  // tr_env.py:626-639
    // (c is relative to the lagged centre: the way-point is taken relative to it in double first)
    T tx = T((ms.waypt[0] - lag.com[0]) - (double)c.x), ty = T((ms.waypt[1] - lag.com[1]) - (double)c.y);
    T in = rsqrt_(tx * tx + ty * ty);
    o[k++] = tx; o[k++] = ty; o[k++] = atan2_(ty * in, tx * in);
  } else if (E.action == TGR_ACT_VEL_TRACK) {
    o[k++] = ms.vcmd[0]; o[k++] = ms.vcmd[1]; o[k++] = ms.vcmd[2];
  }
}

// tr_env.py:549-646, use_obs_noise: what step()/reset() return is observation_with_noise.  o: this lane's observation row
// (noise-free, in LDS).  45 standard normals in the reference's order -- cap positions (18), tendon lengths (9), cap velocities
// (18) -- from philox keyed (seed; key, episode, 0x80000000 + 16 * k + j), k = index of the observation within the episode.
// ZEXT: the 48 normal draws are kept at zext (the row kernels hand in idle LDS: as a local array indexed by the loop counter they would
// live in scratch memory)
template <typename T, bool ZEXT = false>
TGR_DEV void apply_obs_noise(const EnvCfg<T>& E, unsigned long long key, unsigned episode, int kobs, T* o, T* zext = nullptr) {
  T zl[ZEXT ? 1 : 48];
  T* const z = ZEXT ? zext : zl;
  const unsigned k0 = (unsigned)E.seed, k1 = (unsigned)(E.seed >> 32);
#pragma unroll 1
  for (int j = 0; j < 12; j++) {
    unsigned w[4];
    philox((unsigned)key, (unsigned)(key >> 32), episode, 0x80000000u + ((unsigned)kobs << 4) + (unsigned)j, k0, k1, w);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // Box-Muller on 32-bit uniforms: u1 in (0,1], u2 in [0,1)
      if (sizeof(T) == 4) {
        float u1 = ((float)w[2 * h] + 1.0f) * 2.3283064365386963e-10f, u2 = (float)w[2 * h + 1] * 2.3283064365386963e-10f;
        float r = sqrtf(-2.0f * logf(u1)), sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[4 * j + 2 * h] = T(r * cs); z[4 * j + 2 * h + 1] = T(r * sn);
      } else {
        double u1 = ((double)w[2 * h] + 1.0) / 4294967296.0, u2 = (double)w[2 * h + 1] / 4294967296.0;
        double r = sqrt(-2.0 * log(u1));
        z[4 * j + 2 * h] = T(r * cos(2 * PI * u2)); z[4 * j + 2 * h + 1] = T(r * sin(2 * PI * u2));
      }
    }
  }
  const int nv = E.use_cap_velocity ? 18 : 0;
  T sx = T(0), sy = T(0);
#pragma unroll
  for (int i = 0; i < 18; i++) {
    T v = z[i] * E.noise_cap + o[i];
    o[i] = v;
    if (i % 3 == 0) sx += v;
    if (i % 3 == 1) sy += v;
  }
#pragma unroll
  for (int t = 0; t < 9; t++) o[18 + nv + t] = z[18 + t] * E.noise_tendon + o[18 + nv + t];
  if (nv) {
#pragma unroll
    for (int i = 0; i < 18; i++) o[18 + i] = z[27 + i] * E.noise_cap + o[18 + i];  // cap_pos stdev (tr_env.py:604)
  }
  if (E.action == TGR_ACT_TRACKING || E.action == TGR_ACT_AIMING) {  // tr_env.py:626-639
    const int k = 18 + nv + 9;
    T tx = o[k] - sx * T(1.0 / 6.0), ty = o[k + 1] - sy * T(1.0 / 6.0);
    T in = rsqrt_(tx * tx + ty * ty);
    o[k] = tx; o[k + 1] = ty; o[k + 2] = atan2_(ty * in, tx * in);
  }
}

template <typename T> struct StepOut { T reward; bool terminated; T info[10]; T psi_straight; double xy[2]; };

// per-env members of tr_env's command modes (tr_env.py:229-240): way-point, origin, velocity command
template <typename T> struct ModeState { double waypt[2], ori[2]; T vcmd[3]; T psi_before; int step_num; };  // (absolute positions: double)
template <typename T> TGR_DEV T exp_(T x) { return sizeof(T) == 4 ? T(expf(float(x))) : T(exp(double(x))); }
template <typename T> TGR_DEV T sin_(T x) { return sizeof(T) == 4 ? T(sinf(float(x))) : T(sin(double(x))); }
template <typename T> TGR_DEV T angle_normalize(T th) {  // tr_env.py:648-654 -> (-pi, pi]
  while (th > T(PI)) th -= T(2 * PI);
  while (th <= T(-PI)) th += T(2 * PI);
  return th;
}
// (in double whatever the handle's precision: the tracking reward is the DIFFERENCE of two of these, each up to ditch_max + waypt_amp = 400,
// a control period apart -- evaluated in fp32 the difference would carry 5e-5 of rounding)
template <typename T> TGR_DEV double ditch_reward(const EnvCfg<T>& E, const ModeState<T>& ms, double x, double y) {  // tr_env.py:656-667
  const double pvx = ms.waypt[0] - ms.ori[0], pvy = ms.waypt[1] - ms.ori[1];
  const double dp = sqrt(pvx * pvx + pvy * pvy);
  const double idp = 1.0 / dp;
  const double pnx = pvx * idp, pny = pvy * idp;
  const double tvx = ms.waypt[0] - x, tvy = ms.waypt[1] - y;
  const double along = tvx * pnx + tvy * pny;
  const double bx = tvx - along * pnx, by = tvy - along * pny;
  const double bias2 = bx * bx + by * by;
  const double dstd = (double)E.ditch_std, wstd = (double)E.waypt_std;
  const double ditch = (double)E.ditch_max * (1.0 - fabs(along) * idp) * exp(-bias2 / (2.0 * dstd * dstd));
  const double way = (double)E.waypt_amp * exp(-(tvx * tvx + tvy * tvy) / (2.0 * wstd * wstd));
  return ditch + way;
}

// ring buffer of headings (collections.deque that is never cleared: tensegrity_env.py:243)
template <typename T> struct RingRef { T* ring; int n, e; int len, head; };
template <typename T> TGR_DEV void ring_push(RingRef<T>& r, T v) { r.ring[((r.head + r.len) % RING) * r.n + r.e] = v; r.len++; }
template <typename T> TGR_DEV T ring_pop(RingRef<T>& r) { T v = r.ring[r.head * r.n + r.e]; r.head = (r.head + 1) % RING; r.len--; return v; }

// reward / termination of one env step (tensegrity_env.py:291-410, tr_env.py:327-527; straight and turn)
template <typename T>
TGR_DEV StepOut<T> env_epilogue(const Params<T>& P, const EnvCfg<T>& E, const Sim<T>& S, const Lag<T>& lag, const V3<T>* wb,
                                const T* action, double xy_bx, double xy_by, T reset_psi, RingRef<T>& ring, const ModeState<T>& ms) {
  StepOut<T> o;
  // xy_position_after / _before (tensegrity_env.py:293-294,311-315): the centre of the three bars, in double; its change per control period
  // is formed in double and only then rounded
  const double xa = lag.com[0], ya = lag.com[1];
  T idt = rcp_(E.dt);
  const T dxy[2] = {T(xa - xy_bx), T(ya - xy_by)};
  T xv = dxy[0] * idt, yv = dxy[1] * idt;
  V3<T> cap[6], l, r;
  caps_of(P, lag, cap);
  left_right(cap, l, r);
  T psi_s = atan2_(-(l.x - r.x), l.y - r.y);
  T psi = psi_s;
  if (E.env_kind == TGR_ENV_TENSEGRITY && E.action == TGR_ACT_TURN) psi = atan2_(r.y - l.y, r.x - l.x);
  T cc = T(0);
  if (E.env_kind == TGR_ENV_TENSEGRITY) {
#pragma unroll
    for (int i = 0; i < 6; i++) cc += action[i] * action[i];
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) { T v = action[i] + T(0.5) - lag.tenL[i]; cc += v * v; }
  }
  cc *= E.ctrl_cost_weight;
  bool finite = true;
#pragma unroll
  for (int b = 0; b < 3; b++) {
    finite = finite && isfinite(S.x[b].x + S.x[b].y + S.x[b].z + S.q[b][0] + S.q[b][1] + S.q[b][2] + S.q[b][3]);
    finite = finite && isfinite(S.v[b].x + S.v[b].y + S.v[b].z + wb[b].x + wb[b].y + wb[b].z);
  }
  bool healthy;
  if (E.action == TGR_ACT_TURN || (E.env_kind == TGR_ENV_TR && E.action == TGR_ACT_AIMING)) {
    bool any = false;
#pragma unroll
    for (int b = 0; b < 3; b++)
      any = any || abs_(S.v[b].x) > T(0.1) || abs_(S.v[b].y) > T(0.1) || abs_(S.v[b].z) > T(0.1) || abs_(wb[b].x) > T(0.1) ||
            abs_(wb[b].y) > T(0.1) || abs_(wb[b].z) > T(0.1);
    healthy = finite && any;
  } else {
    T mv = T(0.0001);
    healthy = finite && ((xv > mv || xv < -mv) || (yv > mv || yv < -mv));
  }
  T fwd = T(0), ctrl_cost = T(0);
  if (E.action == TGR_ACT_TURN) {
    ring_push(ring, psi);
    if (ring.len > E.reward_delay_steps) {
      T old = ring_pop(ring);
      T hp = T(PI / 2);
      if (psi < -hp && old > hp) psi = T(2 * PI) + psi;
      else if (psi > hp && old < -hp) psi = T(-2 * PI) + psi;
      T delta = (psi - old) * rcp_(E.dt * T(E.reward_delay_steps));
      fwd = delta * E.direction;
      ctrl_cost = cc;
    }
  } else if (E.action == TGR_ACT_STRAIGHT) {
    T dx = dxy[0], dy = dxy[1];
    T pm = atan2_(dy, dx);
    T pd = abs_(pm - reset_psi);
    fwd = E.direction * (sqrt_(dx * dx + dy * dy) * cos_(pd) * idt);
    ctrl_cost = cc;
  } else if (E.action == TGR_ACT_AIMING) {  // tr_env.py:425-446
    T tdx = T(ms.waypt[0] - xy_bx), tdy = T(ms.waypt[1] - xy_by);
    T in = rsqrt_(tdx * tdx + tdy * tdy);
    T target_psi = atan2_(tdy * in, tdx * in);
    T np_ = angle_normalize(target_psi - psi);
    ring_push(ring, np_);
    if (ring.len > E.reward_delay_steps) {
      T old = ring_pop(ring);
      T delta = -(abs_(np_) - abs_(old)) * rcp_(E.dt * T(E.reward_delay_steps));
      fwd = delta * E.yaw_weight;
    }
    ctrl_cost = cc;
  } else if (E.action == TGR_ACT_TRACKING) {  // tr_env.py:448-460
    fwd = T(ditch_reward(E, ms, xa, ya) - ditch_reward(E, ms, xy_bx, xy_by));
    ctrl_cost = cc;
  } else {  // vel_track, tr_env.py:462-474
    T ang = angle_normalize(psi - ms.psi_before) * idt;
    T ex = xv - ms.vcmd[0], ey = yv - ms.vcmd[1], ew = ang - ms.vcmd[2];
    fwd = exp_(T(-5) * (ex * ex + ey * ey)) + T(0.5) * exp_(T(-7) * ew * ew);
    ctrl_cost = cc;
  }
  T hr = E.terminate_when_unhealthy ? ((healthy || E.terminate_when_unhealthy) ? E.healthy_reward : T(0)) : T(0);
  if (E.action == TGR_ACT_AIMING || E.action == TGR_ACT_TRACKING) hr = T(0);
  bool term = E.terminate_when_unhealthy ? !healthy : false;
  if ((E.action == TGR_ACT_AIMING || E.action == TGR_ACT_TRACKING) && ms.step_num > 1000) term = true;
  if (lag.cfrc_absmax > T(1500)) term = true;
  if (!finite) term = true;  // a non-finite lane never survives the step
  T costs = ctrl_cost;
  o.info[TGR_INFO_REWARD_FORWARD] = fwd;
  o.info[TGR_INFO_REWARD_CTRL] = -ctrl_cost;
  if (E.use_contact_forces) {  // tensegrity_env.py:401-404 (the info overwrite is the reference's)
    T contact_cost = E.contact_cost_weight * lag.cfrc_clip_sq;
    costs += contact_cost;
    o.info[TGR_INFO_REWARD_CTRL] = -contact_cost;
  }
  o.info[TGR_INFO_REWARD_SURVIVE] = hr;
  o.info[TGR_INFO_X_POSITION] = T(xa);
  o.info[TGR_INFO_Y_POSITION] = T(ya);
  o.info[TGR_INFO_PSI] = psi;
  o.info[TGR_INFO_DISTANCE] = T(sqrt(xa * xa + ya * ya));
  o.info[TGR_INFO_X_VELOCITY] = xv;
  o.info[TGR_INFO_Y_VELOCITY] = yv;
  o.info[TGR_INFO_FORWARD_REWARD] = fwd;
  o.reward = fwd + hr - costs;
  o.terminated = term;
  o.psi_straight = psi_s;
  o.xy[0] = xa; o.xy[1] = ya;
  return o;
}

// ------------------------------------------------------------------ lanes, roles, keys
// lanes [0, n)  : live environments (slot i == lane i)
// lanes [n, m)  : donors.  A donor settles a fresh episode (reset_model's 50 control periods) inside the SAME step
//                 launches that advance the live lanes; when a live env finishes, match_kernel hands it a ready
//                 donor state and the donor starts over.  Nothing ever waits for a settle.
enum : int { PH_LIVE = -1, PH_READY = 0, PH_INIT = -2 };  // > 0: settle periods still to run

// RNG key of a lane's episodes.  Live lane: the env's global id.  Donor lane: its own key -- or, with reset streams, the key of its
// CLASS (a block of stream_envs consecutive global env ids and the donor lanes that serve it): the episodes of a class form one stream
// (class key, 0), (class key, 1), ... handed out in order, so that what an env restarts with depends on its class alone, not on the
// sharding, the donor lane that happened to settle the episode, or the timing.
template <typename T> TGR_DEV unsigned long long lane_key(const EnvCfg<T>& E, int n, int lane) {
  if (lane < n) return E.first_env_id + (unsigned long long)lane;
  if (E.stream_envs > 0) return (1ull << 62) + (1ull << 40) + E.first_env_id / (unsigned long long)E.stream_envs + (unsigned long long)((lane - n) / E.stream_donors);
  return (1ull << 62) + E.first_env_id * 8ull + (unsigned long long)(lane - n);
}

// start pose of an episode: yaw-randomised table pose, zero velocity, random tendon targets
// tmp: 39 doubles of work space for the perturbed pose (indexed by run-time loop counters: as a local array it would live in scratch
// memory -- the row kernels hand in a piece of LDS that is idle at that point)
template <typename T> TGR_DEV void init_episode(const EnvCfg<T>& E, unsigned long long key, unsigned long long episode, Sim<T>& S, double* tmp) {
  Draws dr = reset_draws(E, key, episode);
  double c = cos(dr.theta), s = sin(dr.theta), ch = cos(0.5 * dr.theta), sh = sin(0.5 * dr.theta);
  const double* q0 = E.env_kind == TGR_ENV_TR ? c_rolling[dr.pose] : (E.model_kind ? c_qpos0_uneven : c_qpos0);
  double* const qn = tmp;
  double* const vn = tmp + 21;
#pragma unroll
  for (int i = 0; i < 21; i++) qn[i] = q0[i];
#pragma unroll
  for (int i = 0; i < 18; i++) vn[i] = 0.0;
  if (E.reset_noise != 0.0) {
    // reset_noise_scale (tensegrity_env.py:436-445, tr_env.py:734-743): qpos += U(-s, s), qvel = s N(0,1), before the yaw transform;
    // scipy's Rotation.from_quat re-normalises the perturbed quaternion (tensegrity_env.py:451-456)
    const unsigned k0 = (unsigned)E.seed, k1 = (unsigned)(E.seed >> 32);
    unsigned o[4];
#pragma unroll 1
    for (int j = 0; j < 11; j++) {
      philox((unsigned)key, (unsigned)(key >> 32), (unsigned)episode, 16 + j, k0, k1, o);
      qn[2 * j] += -E.reset_noise + u53(o[0], o[1]) * 2.0 * E.reset_noise;
      if (2 * j + 1 < 21) qn[2 * j + 1] += -E.reset_noise + u53(o[2], o[3]) * 2.0 * E.reset_noise;
    }
#pragma unroll 1
    for (int j = 0; j < 9; j++) {
      philox((unsigned)key, (unsigned)(key >> 32), (unsigned)episode, 32 + j, k0, k1, o);
      double u1 = 1.0 - u53(o[0], o[1]), u2 = u53(o[2], o[3]);
      double r = sqrt(-2.0 * log(u1));
      vn[2 * j] = E.reset_noise * r * cos(2 * PI * u2); vn[2 * j + 1] = E.reset_noise * r * sin(2 * PI * u2);
    }
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double* q = qn + 7 * b + 3;
      double in = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      q[0] *= in; q[1] *= in; q[2] *= in; q[3] *= in;
    }
  }
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const double* q = qn + 7 * b;
    S.x[b] = mk(T(c * q[0] - s * q[1]), T(s * q[0] + c * q[1]), T(q[2]));
    double w = q[3], x = q[4], y = q[5], z = q[6];
    S.q[b][0] = T(ch * w - sh * z); S.q[b][1] = T(ch * x - sh * y); S.q[b][2] = T(ch * y + sh * x); S.q[b][3] = T(ch * z + sh * w);
    // qvel is (v_world, w_body) and is not rotated (tensegrity_env.py:484); Sim keeps w in the world frame
    S.v[b] = mk(T(vn[6 * b]), T(vn[6 * b + 1]), T(vn[6 * b + 2]));
    S.w[b] = rot_apply(quat_rot(S.q[b]), mk(T(vn[6 * b + 3]), T(vn[6 * b + 4]), T(vn[6 * b + 5])));
    S.al[b] = mk(T(0), T(0), T(0)); S.aa[b] = S.al[b];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { S.ctrl[i] = T(dr.tendons[i]); S.act[i] = T(0); }  // mj_resetData zeroes the activations
  S.capd = E.cap_noise ? T(E.cap_lo + dr.u_cap * (E.cap_hi - E.cap_lo) - E.cap_nominal) : T(0);  // _reset_cap_size (tr_env.py:685-706,712-713)
}

template <typename T> TGR_DEV int settle_total(const EnvCfg<T>& E) {
  // tr_env turn: reward_delay_steps extra full env steps after the 50 settle periods (tr_env.py:864-866)
  return 50 + ((E.env_kind == TGR_ENV_TR && (E.action == TGR_ACT_TURN || E.action == TGR_ACT_AIMING)) ? E.reward_delay_steps : 0);
}

// bookkeeping after settle period `st` (0-based) of reset_model() (tensegrity_env.py:495-510, tr_env.py:811-866)
template <typename T>
TGR_DEV void settle_period_done(const Params<T>& P, const EnvCfg<T>& E, const Lag<T>& lag, int st, RingRef<T>& ring, int& step_num,
                                T& reset_psi, ModeState<T>& ms, double* xy_prev, unsigned long long key, unsigned long long episode) {
  V3<T> cap[6], l, r;
  caps_of(P, lag, cap);
  left_right(cap, l, r);
  T psi_s = atan2_(-(l.x - r.x), l.y - r.y);
  const bool tr = E.env_kind == TGR_ENV_TR;
  if (st == 49) {
    reset_psi = psi_s;
    if (tr && (E.action == TGR_ACT_TRACKING || E.action == TGR_ACT_AIMING)) {
      Draws dr = reset_draws(E, key, episode);
      double len = E.waypt_range[0] + dr.u_way[0] * (E.waypt_range[1] - E.waypt_range[0]);
      double yaw;
      if (E.action == TGR_ACT_TRACKING) {  // tr_env.py:827-839
        // (l, r are relative to the lagged centre)
        ms.ori[0] = lag.com[0] + (double)((l.x + r.x) * T(0.5)); ms.ori[1] = lag.com[1] + (double)((l.y + r.y) * T(0.5));
        yaw = E.waypt_angle_range[0] + dr.u_way[1] * (E.waypt_angle_range[1] - E.waypt_angle_range[0]) + (double)reset_psi;
        if (E.is_test) { len = 0.5 * E.waypt_range[1] + 0.5 * E.waypt_range[0]; yaw = 0.5 * E.waypt_angle_range[1] + 0.5 * E.waypt_angle_range[0] + (double)reset_psi; }
      } else {  // aiming, tr_env.py:842-855 (the reference's oripoint[0] = lx + rx/2 is kept)
        ms.ori[0] = 1.5 * lag.com[0] + (double)(l.x + r.x * T(0.5)); ms.ori[1] = lag.com[1] + (double)((l.y + r.y) * T(0.5));
        yaw = -PI + dr.u_way[1] * 2 * PI + (double)reset_psi;
        if (E.is_test) { len = 0.5 * E.waypt_range[1] + 0.5 * E.waypt_range[0]; yaw = 0.75 * PI + 0.25 * (-PI) + (double)reset_psi; }
      }
      ms.waypt[0] = ms.ori[0] + len * cos(yaw); ms.waypt[1] = ms.ori[1] + len * sin(yaw);
      if (E.action == TGR_ACT_AIMING && E.is_test) { ms.waypt[0] = 0.0; ms.waypt[1] = 0.0; }
    } else if (tr && E.action == TGR_ACT_VEL_TRACK) {
      ms.vcmd[0] = T(0.5) * cos_(reset_psi); ms.vcmd[1] = T(0.5) * sin_(reset_psi); ms.vcmd[2] = T(0);
    }
  }
  const bool ring_mode = E.action == TGR_ACT_TURN || (tr && E.action == TGR_ACT_AIMING);
  bool env_step = (!tr) || st >= 50;  // settling through self.step() vs do_simulation()
  if (env_step && ring_mode) {
    T v;
    if (!tr) v = atan2_(r.y - l.y, r.x - l.x);
    else if (E.action == TGR_ACT_TURN) v = psi_s;
    else {
      T tdx = T(ms.waypt[0] - xy_prev[0]), tdy = T(ms.waypt[1] - xy_prev[1]);
      T in = rsqrt_(tdx * tdx + tdy * tdy);
      v = angle_normalize(atan2_(tdy * in, tdx * in) - psi_s);
    }
    ring_push(ring, v);
    if (ring.len > E.reward_delay_steps) (void)ring_pop(ring);
  }
  if (st >= 50) step_num += 1;
  xy_prev[0] = lag.com[0];
  xy_prev[1] = lag.com[1];
}

// what reset_model() leaves behind besides qpos/qvel
template <typename T>
TGR_DEV void store_modes(const KArgs<T>& k, int e, const ModeState<T>& ms) {
  T* const st = k.st; const int m = k.m;
  aux_st(k, HP_WAYPT, F_WAYPT, e, ms.waypt[0]); aux_st(k, HP_WAYPT + 1, F_WAYPT + 1, e, ms.waypt[1]);
  aux_st(k, HP_ORI, F_ORI, e, ms.ori[0]); aux_st(k, HP_ORI + 1, F_ORI + 1, e, ms.ori[1]);
  st[(F_VCMD + 0) * m + e] = ms.vcmd[0]; st[(F_VCMD + 1) * m + e] = ms.vcmd[1]; st[(F_VCMD + 2) * m + e] = ms.vcmd[2];
}
template <typename T> TGR_DEV ModeState<T> load_modes(const KArgs<T>& k, int e) {
  const T* const st = k.st; const int* const ist = k.ist; const int m = k.m;
  ModeState<T> ms;
  ms.waypt[0] = aux_ld(k, HP_WAYPT, F_WAYPT, e); ms.waypt[1] = aux_ld(k, HP_WAYPT + 1, F_WAYPT + 1, e);
  ms.ori[0] = aux_ld(k, HP_ORI, F_ORI, e); ms.ori[1] = aux_ld(k, HP_ORI + 1, F_ORI + 1, e);
  ms.vcmd[0] = st[(F_VCMD + 0) * m + e]; ms.vcmd[1] = st[(F_VCMD + 1) * m + e]; ms.vcmd[2] = st[(F_VCMD + 2) * m + e];
  ms.psi_before = st[F_PSILAG * m + e];
  ms.step_num = ist[I_STEPNUM * m + e];
  return ms;
}
template <typename T>
TGR_DEV void store_reset_aux(const KArgs<T>& k, int e, const Lag<T>& lag, T reset_psi, int step_num, const RingRef<T>& ring) {
  const Params<T>& P = k.P;
  T* const st = k.st; int* const ist = k.ist; const int m = k.m;
  aux_st(k, HP_XYLAG, F_XYLAG, e, lag.com[0]);
  aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, e, lag.com[1]);
  V3<T> cap[6], l, r;
  caps_of(P, lag, cap);
  left_right(cap, l, r);
  st[F_PSILAG * m + e] = atan2_(-(l.x - r.x), l.y - r.y);
  st[F_RESETPSI * m + e] = reset_psi;
  st[F_EPRET * m + e] = T(0);
  ist[I_STEPNUM * m + e] = step_num;
  ist[I_ELAPSED * m + e] = 0;
  ist[I_RINGLEN * m + e] = ring.len; ist[I_RINGHEAD * m + e] = ring.head;
}

// ------------------------------------------------------------------ after the twenty sub-steps: bookkeeping of one env (shared by the
// block kernels and the row kernels).  `w0`: this lane owns the env's global stores; orow: the env's row of the observation tile
template <typename T, bool ZEXT = false>
TGR_DEV void finish_lane(const KArgs<T>& k, Sim<T>& S, const Lag<T>& lag, int e, bool valid, bool live, bool active, bool w0, int phase,
                         int total, T* orow, T* zext = nullptr) {
  const Params<T>& P = k.P;
  const EnvCfg<T>& E = k.E;
  const int n = k.n, m = k.m;
  const int D = E.obs_dim;
  // everything the epilogue needs and the physics did not is (re)read only now: a value held across the twenty sub-steps
  // costs a register there, or a scratch round trip
  T action[6];
  if (live) {
#pragma unroll
    for (int i = 0; i < 6; i++) action[i] = T(k.actions[e * 6 + i]);
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) action[i] = S.ctrl[i];  // settling: constant tendon targets
  }
  const double xy_bx = aux_ld(k, HP_XYLAG, F_XYLAG, e), xy_by = aux_ld(k, HP_XYLAG + 1, F_XYLAG + 1, e);
  T reset_psi = k.st[F_RESETPSI * m + e];
  T ep_ret = k.st[F_EPRET * m + e];
  int step_num = k.ist[I_STEPNUM * m + e], elapsed = k.ist[I_ELAPSED * m + e];
  RingRef<T> ring; ring.ring = k.ring; ring.n = m; ring.e = e;
  ring.len = k.ist[I_RINGLEN * m + e]; ring.head = k.ist[I_RINGHEAD * m + e];
  ModeState<T> ms = load_modes(k, e);
  V3<T> wb[3];
  if (active && w0) store_sim(k.st, m, e, S, lag, wb);
  else if (active) {
#pragma unroll
    for (int b = 0; b < 3; b++) { normalize_quat(S.q[b]); wb[b] = rot_applyT(quat_rot(S.q[b]), S.w[b]); }
  } else {
#pragma unroll
    for (int b = 0; b < 3; b++) wb[b] = mk(T(0), T(0), T(0));
  }
  if (!live) {
    // ---- donor lane: one more settle period done
    if (active && w0) {
      const int stp = total - phase;  // 0-based index of the period just run
      if (stp == 0) { step_num = 0; }
      double xy_prev[2] = {xy_bx, xy_by};
      settle_period_done(P, E, lag, stp, ring, step_num, reset_psi, ms, xy_prev, lane_key(E, n, e),
                         (unsigned long long)(unsigned)k.ist[I_SRCEP * m + e]);
      aux_st(k, HP_XYLAG, F_XYLAG, e, xy_prev[0]); aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, e, xy_prev[1]);
      store_modes(k, e, ms);
      ms.step_num = step_num;
      phase -= 1;
      k.st[F_RESETPSI * m + e] = reset_psi;
      k.ist[I_STEPNUM * m + e] = step_num;
      k.ist[I_RINGLEN * m + e] = ring.len; k.ist[I_RINGHEAD * m + e] = ring.head;
      k.ist[I_PHASE * m + e] = phase;
      if (lag.overflow) atomicAdd(&k.counters[0], (unsigned long long)lag.overflow);
      if (phase == PH_READY) {
        store_reset_aux(k, e, lag, reset_psi, step_num, ring);
        make_obs(P, E, S, lag, wb, ms, orow);
        if (E.use_obs_noise) apply_obs_noise<T, ZEXT>(E, lane_key(E, n, e), (unsigned)k.ist[I_SRCEP * m + e], 0, orow, zext);
        T* op = k.dobs + (size_t)(e - n) * D;
        for (int i = 0; i < D; i++) op[i] = orow[i];
        if (k.async) k.dready[e] = k.launch_idx;  // (donor stream: ready as of this donor launch)
      }
    }
  }
  StepOut<T> out;
  bool truncated = false, done = false;
  if (live) {
    out = env_epilogue(P, E, S, lag, wb, action, xy_bx, xy_by, reset_psi, ring, ms);
    if (E.env_kind == TGR_ENV_TR) step_num += 1;
    elapsed += 1;
    truncated = elapsed >= E.max_episode_steps;
    ep_ret += out.reward;
    done = out.terminated || truncated;
    if (valid && w0) {
      make_obs(P, E, S, lag, wb, ms, orow);
      if (k.obs_clean) {
        T* cp = k.obs_clean + (size_t)e * D;
        for (int i = 0; i < D; i++) cp[i] = orow[i];
      }
      if (E.use_obs_noise) apply_obs_noise<T, ZEXT>(E, lane_key(E, n, k.ist[I_SRC * m + e]), (unsigned)k.ist[I_SRCEP * m + e], elapsed, orow, zext);
      aux_st(k, HP_XYLAG, F_XYLAG, e, out.xy[0]); aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, e, out.xy[1]);
      k.st[F_PSILAG * m + e] = out.psi_straight;
      k.st[F_EPRET * m + e] = ep_ret;
      k.ist[I_STEPNUM * m + e] = step_num; k.ist[I_ELAPSED * m + e] = elapsed;
      k.ist[I_RINGLEN * m + e] = ring.len; k.ist[I_RINGHEAD * m + e] = ring.head;
      k.reward[e] = out.reward;
      k.terminated[e] = out.terminated ? 1 : 0;
      k.truncated[e] = truncated ? 1 : 0;
      k.doneflag[e] = (done && k.auto_reset) ? 1 : 0;
      if (k.info) {
        T* ip = k.info + (size_t)e * TGR_INFO_K;
#pragma unroll
        for (int i = 0; i < 10; i++) ip[i] = out.info[i];
        ip[TGR_INFO_WAYPT_X] = T(ms.waypt[0]); ip[TGR_INFO_WAYPT_Y] = T(ms.waypt[1]);
        ip[TGR_INFO_ORI_X] = T(ms.ori[0]); ip[TGR_INFO_ORI_Y] = T(ms.ori[1]);
        ip[TGR_INFO_EP_RETURN] = ep_ret; ip[TGR_INFO_EP_LENGTH] = T(elapsed);
#pragma unroll
        for (int i = 0; i < 9; i++) ip[TGR_INFO_TENDON_LENGTH + i] = lag.tenL[i];
        ip[TGR_INFO_BAR_CONTACT] = lag.bar_contact;
      }
      if (lag.overflow) atomicAdd(&k.counters[0], (unsigned long long)lag.overflow);
      if (done && k.terminal_obs) {
        T* tp = k.terminal_obs + (size_t)e * D;
        for (int i = 0; i < D; i++) tp[i] = orow[i];
      }
    }
  }
}

// ------------------------------------------------------------------ step kernel
template <typename T, int LW, int MODEL> __global__ __launch_bounds__(256) void step_kernel(KArgs<T> k) {
  __shared__ T lds_all[lds_slots<MODEL>() * LW];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 4 waves share the block's LW envs
  if (lane >= LW) return;
  const int gid = blockIdx.x * LW + lane;
  const int n = k.n, m = k.m;
  const bool valid = gid < m;
  const int e = valid ? gid : m - 1;
  const Params<T>& P = k.P;
  const EnvCfg<T>& E = k.E;
  T* lds = lds_all + lane;

  const bool live = e < n;
  int phase = k.ist[I_PHASE * m + e];
  const bool donor_run = !live && (phase > 0 || phase == PH_INIT);
  const bool active = valid && (live || donor_run);
  if (!__any(active)) return;  // a wave of idle donors

  Sim<T> S;
  int episode = k.ist[I_EPISODE * m + e];
  const int total = settle_total(E);
  if (!live && phase == PH_INIT) {
    double ie_tmp[39];
    init_episode(E, lane_key(E, n, e), (unsigned long long)(unsigned)episode, S, ie_tmp);
    phase = total;
    if (valid) { k.ist[I_SRC * m + e] = e; k.ist[I_SRCEP * m + e] = episode; k.ist[I_EPISODE * m + e] = episode + 1; }
  } else {
    load_sim(k.st, m, e, S);
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const T a = T(k.actions[e * 6 + i]);
      S.ctrl[i] = E.env_kind == TGR_ENV_TR ? S.ctrl[i] + (a - S.ctrl[i]) * E.dt : a;  // tr_env: _action_filter, k_FILTER = 1
    }
  }

  Lag<T> lag;
  lag.cfrc_absmax = T(0); lag.cfrc_clip_sq = T(0); lag.overflow = 0; lag.n_it = 0; lag.n_ls = 0; lag.n_unconv = 0;
  lag.bar_contact = T(0);
#pragma unroll
  for (int i = 0; i < 9; i++) lag.tenL[i] = T(0);  // wave 3 never evaluates tendons
#ifdef TGR_STAMPS
  for (int i = 0; i < 32; i++) lag.stamp[i] = 0;
  lag.t_last = clock64();
#endif
  warm_to_lds<T, LW>(S, lds, wid);
#pragma unroll 1
  for (int s = 0; s + 1 < E.frame_skip; s++) substep<T, LW, false, MODEL>(P, S, lds, wid, lag);
  substep<T, LW, true, MODEL>(P, S, lds, wid, lag);

#ifdef TGR_STAMPS
  { long long _t = clock64(); lag.stamp[5] += _t - lag.t_last; lag.t_last = _t; }
  if (lane == 0 && TGR_STAMP_BLOCKS) for (int i = 0; i < 32; i++) atomicAdd(&k.counters[8 + wid * 32 + i], (unsigned long long)lag.stamp[i]);
#endif
  const bool w0 = wid == 0;  // one wave owns every global store
  if (lane == 0 && w0) {
    atomicAdd(&k.counters[3], (unsigned long long)lag.n_it);
    atomicAdd(&k.counters[4], (unsigned long long)lag.n_ls);
    atomicAdd(&k.counters[5], (unsigned long long)E.frame_skip);
  }
  if (active && w0 && lag.n_unconv) atomicAdd(&k.counters[6], (unsigned long long)lag.n_unconv);
  if (w0) {  // lane-sub-steps run: what counters[6] is a fraction of
    const int nact = __popcll(__ballot(active));
    if (lane == 0) atomicAdd(&k.counters[7], (unsigned long long)(nact * E.frame_skip));
  }
  if (active && w0) warm_from_lds<T, LW>(S, lds);
  const int D = E.obs_dim;
  const int Dp = (D & 1) ? D : D + 1;  // odd row pitch of the observation tile: conflict-free
  T* orow = lds_all + lane * Dp;       // this lane's row (the physics columns are dead after the barrier below)
  __syncthreads();
  finish_lane(k, S, lag, e, valid, live, active, w0, phase, total, orow);
  // observations of the live lanes of this block leave through the LDS tile: coalesced stores into the [N][D] buffer
  const int base = blockIdx.x * LW;
  const int rows = min(LW, n - base);
  if (rows > 0) {
    __syncthreads();
    for (int f = wid * LW + lane; f < rows * D; f += 4 * LW) {
      int r = f / D, c = f - r * D;
      k.obs[(size_t)base * D + f] = lds_all[r * Dp + c];
    }
  }
}

// ------------------------------------------------------------------ row kernels: one environment = one DPP row (tgr_row.h)
// A wavefront = 4 environments; the 16 lanes of a row hold the environment's state replicated and split the work by role; between
// stages the state is parked in LDS slabs only the row itself reads and writes: no barrier anywhere.
// T: type of the state arrays / outputs; TK: type of pose, kinematics and tendon geometry inside the kernel (double in production).
// Environments per block (16 lanes each).  One wavefront per block: the blocks of a launch finish at very different times (Newton
// iterations) and a 4096-env rollout with its donor rows is just over one round of the 2048 wave slots of the chip -- with 4-wave blocks
// the last 64 blocks waited for whole blocks to drain and the launch took two wave life times; single waves slot in as soon as any
// wave retires.  (20 KB of LDS per wave for the fp32 flat-floor kernels: 8 blocks per CU = 2 waves per SIMD.)
#ifndef TGR_ROW_ENVS
#define TGR_ROW_ENVS 4
#endif
template <typename T, int MODEL> constexpr int row_envs() { return TGR_ROW_ENVS; }
static_assert(TGR_ROW_ENVS % 4 == 0 && row::ROW == 16, "a row (16 lanes) must lie inside one wavefront: its LDS slabs are not guarded by barriers");
// the flat-floor model keeps no activations: its K slab ends before KACT
template <int MODEL> constexpr int k_slab() { return MODEL ? row::K_SLAB : row::KACT; }
// wavefronts per SIMD the row kernels are compiled for (register budget 512 / waves): two for fp32 handles -- the donor rows of a
// 4096-env random-action rollout bring the grid to ~2300 wavefronts on 1024 SIMDs -- one for the fp64 parity instantiation
#ifndef TGR_ROW_WAVES
#define TGR_ROW_WAVES(T, MODEL) ((sizeof(T) == 4 && !(MODEL)) ? 2 : 1)
#endif

template <typename T, typename TK> struct RowArgs { row::RParams<T> P; row::RParams<TK> PK; };
// the ONE argument of the row kernels: its members' byte offsets in the kernarg segment are what the kernels re-read their arguments by
// (row::kernarg_again: nothing loaded at kernel entry is held in SGPRs across the sub-steps)
template <typename T, typename TK> struct RowKArgs { KArgs<T> k; RowArgs<T, TK> ra; };
template <typename T, typename TK> TGR_DEV row::ArgRef<row::RParams<T>> arg_P() {
  typedef RowKArgs<T, TK> A; typedef RowArgs<T, TK> R;
  return row::ArgRef<row::RParams<T>>{(unsigned)(__builtin_offsetof(A, ra) + __builtin_offsetof(R, P))};
}
template <typename T, typename TK> TGR_DEV row::ArgRef<row::RParams<TK>> arg_PK() {
  typedef RowKArgs<T, TK> A; typedef RowArgs<T, TK> R;
  return row::ArgRef<row::RParams<TK>>{(unsigned)(__builtin_offsetof(A, ra) + __builtin_offsetof(R, PK))};
}

// the row's state slabs in shared memory (tgr_row.h RMem): written and read by the 16 lanes of the row alone, so no barrier guards them
template <typename T, typename TK, int MODEL> struct RowShared {
  static constexpr int NE = row_envs<T, MODEL>(), PAGES = row::n_pages<MODEL>();
  TK kd[NE * k_slab<MODEL>()];
  TK rd[row::R_SLAB * 16];
  TK ad[NE * row::A_SLAB * PAGES * 16];
  T fd[NE * row::f_slab<MODEL>()];
  T td[NE * row::T_SLAB * 16];
  T cd[NE * row::C_SLAB * PAGES * 16];  // (after the last sub-step: the row's observation line, <= 49 values)
};
template <typename T, typename TK, int MODEL> TGR_DEV row::RMem<TK, T> row_mem(RowShared<T, TK, MODEL>& sh, int slot, int r, unsigned* dbg = nullptr) {
  constexpr int PAGES = row::n_pages<MODEL>();
  row::RMem<TK, T> M;
  M.dbg = dbg;
  M.kd = row::Slab<TK>{sh.kd + slot * k_slab<MODEL>(), 1, 0};
  M.fd = row::Slab<T>{sh.fd + slot * row::f_slab<MODEL>(), 1, 0};
  M.td = row::Slab<T>{sh.td + slot * row::T_SLAB * 16, 16, r};
  M.hd = row::Slab<T>{sh.td + slot * row::T_SLAB * 16 + 9, 1, 0};   // columns 9, 10 of the tendon slab (tgr_row.h hidx)
  M.lp = row::Slab<T>{sh.td + slot * row::T_SLAB * 16 + 11, 1, 0};  // columns 11..13 (lidx)
  // columns 14, 15: rows 0..2 the lagged centre (K: one double spans the two columns of an fp32 handle's row); row 9 the donor-stream word
  M.lc = row::Slab<TK>{reinterpret_cast<TK*>(sh.td + slot * row::T_SLAB * 16 + 14), (int)(16 * sizeof(T) / sizeof(TK)), 0};
  M.xs = row::Slab<T>{sh.td + slot * row::T_SLAB * 16 + 14 + 16 * row::LC_SLAB, 16, 0};  // rows 3.. of column 14: small replicated scalars
  static_assert(row::LC_SLAB + row::XS_SLAB <= 9, "row 9 of column 14 is the donor-stream word");
  M.rd = row::Slab<TK>{sh.rd, 16, r};
  M.ad = row::Slab<TK>{sh.ad + slot * row::A_SLAB * PAGES * 16, 16, r};
  M.cd = row::Slab<T>{sh.cd + slot * row::C_SLAB * PAGES * 16, 16, r};
  return M;
}
// state of the block kernels' layout -> the row's slabs
template <typename T, typename TK, int MODEL> TGR_DEV void row_import(const Sim<T>& S, const row::RMem<TK, T>& M) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    M.kd.st(row::KX + 3 * b, TK(S.x[b].x)); M.kd.st(row::KX + 3 * b + 1, TK(S.x[b].y)); M.kd.st(row::KX + 3 * b + 2, TK(S.x[b].z));
#pragma unroll
    for (int i = 0; i < 4; i++) M.kd.st(row::KQ + 4 * b + i, TK(S.q[b][i]));
    M.fd.st(row::FV + 3 * b, S.v[b].x); M.fd.st(row::FV + 3 * b + 1, S.v[b].y); M.fd.st(row::FV + 3 * b + 2, S.v[b].z);
    M.fd.st(row::FW + 3 * b, S.w[b].x); M.fd.st(row::FW + 3 * b + 1, S.w[b].y); M.fd.st(row::FW + 3 * b + 2, S.w[b].z);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { M.fd.st(row::FCTRL + i, S.ctrl[i]); if (MODEL) M.kd.st(row::KACT + i, TK(S.act[i])); }
  M.fd.st(row::FCAPD, S.capd);
#pragma unroll
  for (int i = 0; i < 18; i++) M.fd.st(row::FDL + i, T(0));
#pragma unroll
  for (int i = 0; i < row::H_SLAB; i++) M.hd.st(row::hidx(i), T(0));
}
template <typename T, typename TK, int MODEL> TGR_DEV void row_load(const T* st, const double* hp, int m, int e, const row::RMem<TK, T>& M) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
#pragma unroll
    for (int i = 0; i < 3; i++) M.kd.st(row::KX + 3 * b + i, hp ? TK(hp[(7 * b + i) * m + e]) : TK(st[(F_QPOS + 7 * b + i) * m + e]));
    TK q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = hp ? TK(hp[(7 * b + 3 + i) * m + e]) : TK(st[(F_QPOS + 7 * b + 3 + i) * m + e]);
    row::normalize_quat(q);
#pragma unroll
    for (int i = 0; i < 4; i++) M.kd.st(row::KQ + 4 * b + i, q[i]);
    const row::Rot<TK> R = row::quat_rot(q);
#pragma unroll
    for (int i = 0; i < 3; i++) M.fd.st(row::FV + 3 * b + i, st[(F_QVEL + 6 * b + i) * m + e]);
    const row::V3<TK> wb = row::mk<TK>(TK(st[(F_QVEL + 6 * b + 3) * m + e]), TK(st[(F_QVEL + 6 * b + 4) * m + e]), TK(st[(F_QVEL + 6 * b + 5) * m + e]));
    row::st3(M.fd, row::FW + 3 * b, row::cvt3<T>(row::rot_apply(R, wb)));
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { M.fd.st(row::FCTRL + i, st[(F_CTRL + i) * m + e]); if (MODEL) M.kd.st(row::KACT + i, TK(st[(F_ACT + i) * m + e])); }
  M.fd.st(row::FCAPD, st[F_CAPD * m + e]);
#pragma unroll
  for (int i = 0; i < 18; i++) M.fd.st(row::FDL + i, T(0));
#pragma unroll
  for (int i = 0; i < row::H_SLAB; i++) M.hd.st(row::hidx(i), T(0));
}
// the env layer (finish_lane, settle_period_done, make_obs ...) works on the block kernels' structs
template <typename T, typename TK, int MODEL>
TGR_DEV void row_export(const row::RMem<TK, T>& M, const row::RLag<TK, T>& rl, Sim<T>& S, Lag<T>& lag) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    S.x[b] = mk(T(M.kd.ld(row::KX + 3 * b)), T(M.kd.ld(row::KX + 3 * b + 1)), T(M.kd.ld(row::KX + 3 * b + 2)));
    S.v[b] = mk(M.fd.ld(row::FV + 3 * b), M.fd.ld(row::FV + 3 * b + 1), M.fd.ld(row::FV + 3 * b + 2));
    S.w[b] = mk(M.fd.ld(row::FW + 3 * b), M.fd.ld(row::FW + 3 * b + 1), M.fd.ld(row::FW + 3 * b + 2));
    // qacc of the last forward pass = unconstrained acceleration + the solver's iterate (angular part in the world frame)
    S.al[b] = mk(T(M.kd.ld(row::KL0 + 3 * b)) + M.fd.ld(row::FDL + 3 * b), T(M.kd.ld(row::KL0 + 3 * b + 1)) + M.fd.ld(row::FDL + 3 * b + 1),
                 T(M.kd.ld(row::KL0 + 3 * b + 2)) + M.fd.ld(row::FDL + 3 * b + 2));
    S.aa[b] = mk(T(M.kd.ld(row::KA0 + 3 * b)) + M.fd.ld(row::FDA + 3 * b), T(M.kd.ld(row::KA0 + 3 * b + 1)) + M.fd.ld(row::FDA + 3 * b + 1),
                 T(M.kd.ld(row::KA0 + 3 * b + 2)) + M.fd.ld(row::FDA + 3 * b + 2));
    lag.x[b] = mk(M.lp.ld(row::lidx(row::LPX + 3 * b)), M.lp.ld(row::lidx(row::LPX + 3 * b + 1)), M.lp.ld(row::lidx(row::LPX + 3 * b + 2)));
    lag.z[b] = mk(T(M.kd.ld(row::KEZ + 3 * b)), T(M.kd.ld(row::KEZ + 3 * b + 1)), T(M.kd.ld(row::KEZ + 3 * b + 2)));
#pragma unroll
    for (int i = 0; i < 4; i++) { S.q[b][i] = T(M.kd.ld(row::KQ + 4 * b + i)); lag.q[b][i] = M.lp.ld(row::lidx(row::LPQ + 4 * b + i)); }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { S.ctrl[i] = M.fd.ld(row::FCTRL + i); S.act[i] = MODEL ? T(M.kd.ld(row::KACT + i)) : T(0); }
  S.capd = M.fd.ld(row::FCAPD);
#pragma unroll
  for (int i = 0; i < 9; i++) lag.tenL[i] = M.lp.ld(row::lidx(row::LPT + i));
  lag.com[0] = (double)M.lc.ld(row::LCX); lag.com[1] = (double)M.lc.ld(row::LCX + 1);  // (lp's positions are relative to it)
  lag.cfrc_absmax = rl.cfrc_absmax; lag.cfrc_clip_sq = rl.cfrc_clip_sq; lag.bar_contact = rl.bar_contact;
  lag.overflow = (int)rl.overflow; lag.n_it = rl.n_it; lag.n_ls = rl.n_ls; lag.n_unconv = (int)rl.n_unconv;
}
// the pose of the row's environment at full precision -> hp (called by the lane that owns the env's global stores, next to store_sim)
template <typename T, typename TK> TGR_DEV void row_store_pose(const row::RMem<TK, T>& M, double* hp, int m, int e) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
#pragma unroll
    for (int i = 0; i < 3; i++) hp[(7 * b + i) * m + e] = (double)M.kd.ld(row::KX + 3 * b + i);
    TK q[4] = {M.kd.ld(row::KQ + 4 * b), M.kd.ld(row::KQ + 4 * b + 1), M.kd.ld(row::KQ + 4 * b + 2), M.kd.ld(row::KQ + 4 * b + 3)};
    row::normalize_quat(q);
#pragma unroll
    for (int i = 0; i < 4; i++) hp[(7 * b + 3 + i) * m + e] = (double)q[i];
  }
}
template <typename T, typename TK> TGR_DEV void row_lag_init(row::RLag<TK, T>& rl) {
  // (cfrc_absmax, cfrc_clip_sq, bar_contact, overflow are assigned by the last sub-step, not accumulated: nothing to initialise)
  rl.n_unconv = T(0); rl.n_own = T(0); rl.n_it = 0; rl.n_ls = 0;
#ifdef TGR_ROW_STAMPS
  for (int i = 0; i < 24; i++) rl.stamp[i] = 0ull;
  rl.t_last = __builtin_amdgcn_s_memtime();
#endif
}

#ifdef TGR_WAVE_TIMES  // development aid: start / end time, iterations of every wavefront of the last step launch
__device__ unsigned long long g_wave_times[4 * 16384];
#endif
template <typename T, typename TK, int MODEL> __global__ __launch_bounds__((16 * row_envs<T, MODEL>()), (TGR_ROW_WAVES(T, MODEL))) void step_kernel_row(RowKArgs<T, TK> args) {
  typedef RowKArgs<T, TK> RowKArgsT;
  const KArgs<T>& k = args.k;
  const RowArgs<T, TK>& ra = args.ra;
  constexpr int ROW_ENVS = row_envs<T, MODEL>();
#ifdef TGR_WAVE_TIMES
  const unsigned long long wt0 = wall_clock64();
#endif
  __shared__ RowShared<T, TK, MODEL> sh;
  const int r = (int)(threadIdx.x & 15u);
  const int slot = (int)(threadIdx.x >> 4);  // environment of this row within the block
  // XCD-aware order: blocks are dealt round-robin over the 8 XCDs, each with its own L2.  Within a chunk of 64 blocks the (8 x 8) index is
  // transposed, so that the 8 blocks an XCD gets from the chunk are 8 CONSECUTIVE groups of environments: the 128-byte line of a state
  // field they share is fetched into one L2 instead of eight (19 -> 5 MB fetched per launch).  The launch pads the grid to a multiple of 64.
  const int bx = (int)blockIdx.x;
  const int lb = (bx & ~63) | ((bx & 7) << 3) | ((bx >> 3) & 7);
  const int gid = k.lane_base + lb * ROW_ENVS + slot;
  const int n = k.n, m = k.m;
  const bool valid = gid < k.lane_limit;
  const int e = valid ? gid : m - 1;
  const EnvCfg<T>& E = k.E;
  // reset streams: the classes of this step's match kernel append their starved envs to list[] -- the count starts from zero (the settle
  // kernel of the previous step, the list's last reader, precedes this launch on the stream)
  if (E.stream_envs > 0 && k.lane_base == 0 && bx == 0 && threadIdx.x == 0) *k.list_count = 0;
  const bool live = valid && e < n;  // (rows past the end do nothing at all)
  int phase = k.ist[I_PHASE * m + e];
  unsigned long long cmd = 0ull;
  int cmd_tag0 = 0;
  if (k.async && valid && e >= n) {  // donor stream: a command of an earlier step's match kernel that this lane has not carried out yet
    cmd = k.dcmd[e];
    const int tag = (int)(unsigned)(cmd & 0xffffffffull);
    if (tag != 0 && tag != k.dapplied[e] && tag < k.launch_idx) { phase = PH_INIT; cmd_tag0 = tag; }
  }
  // dcmd is read ONCE per launch: this step's match kernel runs concurrently and may overwrite the word (a starved class re-arms the holder of
  // an episode: match_kernel_stream) -- a second read after the sub-steps could see a different command than the one carried out above.
  // The decision is parked in the row's LDS (row 9 of column 14 of the tendon slab: lanes >= 9 own no tendon) and read back below.
  *reinterpret_cast<int*>(sh.td + slot * (row::T_SLAB * 16) + 9 * 16 + 14) = cmd_tag0;
  const bool donor_run = valid && e >= n && (phase > 0 || phase == PH_INIT);
  const bool active = live || donor_run;
  const int total = settle_total(E);
  // Paced donor rows.  A launch whose busy wavefronts slightly exceed the wave slots of the chip (4096 envs + ~4300 settling donors =
  // 2110 wavefronts for 2048 slots) ends with the wavefronts that had to wait for a slot: they start when the first ones retire and
  // then run a whole period.  Those are the highest-indexed busy donors.  The donor rows from slow_e on therefore advance HALF a
  // control period per launch (same sub-steps in the same order, over two launches; I_HALF says which half is due): half as long,
  // they end before the longest first-round wavefront does.  Wavefront-uniform: slow_e and the rows of a wavefront are multiples of 4.
  const bool slow = k.lane_base + lb * ROW_ENVS >= k.slow_e;
  const int nsub = slow ? E.frame_skip / 2 : E.frame_skip;
  const unsigned long long mask40 = (1ull << 40) - 1ull;
  if (k.stagger_pct > 0) {  // (donor stream only: KArgs::stagger_pct)
    unsigned long long* const mine = k.counters + STAMP_SLOT + 2 * (k.launch_idx & 1);
    if (k.lane_base == 0) {
      if (bx == 0 && threadIdx.x == 0)
        __hip_atomic_store(mine, ((unsigned long long)(unsigned)k.launch_idx << 40) | (wall_clock64() & mask40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (__any(active)) {
      auto rd = [](unsigned long long* p) {
        const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffull));
      };
      // how long the previous live launch took (its stamps carry its index; anything else: no stagger this time)
      unsigned long long* const prev = k.counters + STAMP_SLOT + 2 * ((k.launch_idx - 1) & 1);
      const unsigned long long ps = rd(prev), pe = rd(prev + 1);
      const bool have = (unsigned)(ps >> 40) == (unsigned)(k.launch_idx - 1) && (unsigned)(pe >> 40) == (unsigned)(k.launch_idx - 1);
      unsigned long long dur = have ? ((pe - ps) & mask40) : 0ull;
      dur = dur > 400000ull ? 400000ull : dur;  // (4 ms)
      const unsigned long long need = dur * (unsigned long long)k.stagger_pct / 100ull, early = need / 4ull;
      const unsigned long long t0 = wall_clock64();
      bool first = true;
      for (; need > 0ull;) {  // (every wavefront on its own, every path out bounded by the clock)
        const unsigned long long w = rd(mine), now = wall_clock64();
        const unsigned idx = (unsigned)(w >> 40);
        const unsigned long long age = (now - w) & mask40;
        const bool late = idx > (unsigned)k.launch_idx || (idx == (unsigned)k.launch_idx && age >= (first ? early : need));
        if (late || now - t0 >= 60000ull) break;
        first = false;
        __builtin_amdgcn_s_sleep(64);
      }
    }
  }
  row::RLag<TK, T> rl;
  row_lag_init<T, TK>(rl);
  if (__any(active)) {  // (a wavefront of idle donor rows has nothing to do)
    const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, slot, r, (k.dbg && live) ? k.dbg + (size_t)e * row::TRACE_WORDS : nullptr);
    int episode = k.ist[I_EPISODE * m + e];
    if (k.async && E.stream_envs > 0 && cmd != 0ull) episode = (int)(unsigned)(cmd >> 32);  // reset streams: the command names the episode
    if (donor_run && phase == PH_INIT) {
      Sim<T> S0;
      init_episode(E, lane_key(E, n, e), (unsigned long long)(unsigned)episode, S0, (double*)(sh.cd + slot * (row::C_SLAB * row::n_pages<MODEL>() * 16)));
      row_import<T, TK, MODEL>(S0, M);
      phase = total;
      if (valid && r == 0) { k.ist[I_SRC * m + e] = e; k.ist[I_SRCEP * m + e] = episode; k.ist[I_EPISODE * m + e] = episode + 1; }
    } else {
      row_load<T, TK, MODEL>(k.st, k.hp, m, e, M);
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const T a = T(k.actions[e * 6 + i]);
        const T c0 = M.fd.ld(row::FCTRL + i);
        M.fd.st(row::FCTRL + i, E.env_kind == TGR_ENV_TR ? c0 + (a - c0) * E.dt : a);  // tr_env: _action_filter, k_FILTER = 1
      }
    }
    row::store_role_consts<TK, T, TK, MODEL>(ra.PK, M.rd);
#pragma unroll 1
    for (int s = 0; s + 1 < nsub; s++) row::substep<TK, T, TK, T, MODEL, false>(arg_PK<T, TK>(), arg_P<T, TK>(), M, rl, s);
    row::substep<TK, T, TK, T, MODEL, true>(arg_PK<T, TK>(), arg_P<T, TK>(), M, rl, nsub - 1);
#ifdef TGR_ROW_STAMPS
    if ((threadIdx.x & 63u) == 0 && live) {  // live-row wavefronts only: phase cycles, and the wavefront count in slot 23
      for (int i = 0; i < 16; i++) atomicAdd(&k.counters[8 + i], rl.stamp[i]);
      atomicAdd(&k.counters[8 + 23], 1ull);
    }
#endif
  }
  // Who this row is gets worked out a second time, from laundered thread and block numbers: carried through the twenty sub-steps in
  // registers it was spilled (a dozen dwords per lane: 6 MB of scratch stores per launch), recomputed it is ten instructions.
  // (TGR_ROW_ENVS == 4: the block is one wavefront and the thread number is the lane number, which mbcnt gives without keeping v0 alive)
  unsigned tx2 = TGR_ROW_ENVS == 4 ? __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : threadIdx.x, bx2 = blockIdx.x;
  asm volatile("" : "+v"(tx2));
  asm volatile("" : "+s"(bx2));
  {
    // ... and so are the kernel's arguments: k is read AGAIN from the kernarg segment through a pointer the compiler cannot connect with the
    // loads at the top of the kernel.  Held in SGPRs across the sub-steps for the sake of this epilogue, the two dozen pointers and the env
    // configuration of KArgs were most of ~200 spilled SGPRs -- parked in VGPR lanes and fetched back with v_readlane (5 % of the
    // instruction stream, and four VGPRs of a budget that has none to spare).
    const KArgs<T>& k = *row::kernarg_again<KArgs<T>>((unsigned)__builtin_offsetof(RowKArgsT, k));
    const int r = (int)(tx2 & 15u), slot = (int)(tx2 >> 4);
    const int lb = (int)((bx2 & ~63u) | ((bx2 & 7u) << 3) | ((bx2 >> 3) & 7u));
    const int gid = k.lane_base + lb * ROW_ENVS + slot;
    const int n = k.n, m = k.m;
    const bool valid = gid < k.lane_limit;
    const int e = valid ? gid : m - 1;
    const EnvCfg<T>& E = k.E;
    const bool live = valid && e < n;
    int phase = k.ist[I_PHASE * m + e];  // (only written below)
    const int cmd_tag = *reinterpret_cast<const int*>(sh.td + slot * (row::T_SLAB * 16) + 9 * 16 + 14);  // (the command carried out at the top, or 0)
    if (cmd_tag) phase = PH_INIT;
    const bool donor_run = valid && e >= n && (phase > 0 || phase == PH_INIT);
    const bool active = live || donor_run;
    const int total = settle_total(E);
    const int D = E.obs_dim;
    if (!__any(active)) return;
    const bool slow = k.lane_base + lb * ROW_ENVS >= k.slow_e;
    const int nsub = slow ? E.frame_skip / 2 : E.frame_skip;
    if (cmd_tag && (tx2 & 15u) == 0) { k.dapplied[e] = cmd_tag; k.dready[e] = 0; }  // (the command is carried out; the episode is not ready)
    // a paced row that has just been set up, or whose half flag is clear, has run the FIRST half of its period
    const bool first_half = slow && donor_run && (phase == PH_INIT || k.ist[I_HALF * m + e] == 0);
    if (donor_run && phase == PH_INIT) phase = total;  // (the episode was set up at the top)
    const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, slot, r);
    Sim<T> S;
    Lag<T> lag;
    row_export<T, TK, MODEL>(M, rl, S, lag);
    const bool w0 = r == 0;  // one lane of the row owns the env's global stores
    if ((threadIdx.x & 63u) == 0) {
      atomicAdd(&k.counters[3], (unsigned long long)lag.n_it);
      atomicAdd(&k.counters[4], (unsigned long long)lag.n_ls);
      atomicAdd(&k.counters[5], (unsigned long long)nsub);
    }
    if (active && w0 && lag.n_unconv) atomicAdd(&k.counters[6], (unsigned long long)lag.n_unconv);
    {
      const int nact = __popcll(__ballot(active && w0));
      if ((threadIdx.x & 63u) == 0) atomicAdd(&k.counters[7], (unsigned long long)(nact * nsub));
    }
    if (slow) {
      // (every busy row of a paced wavefront is a donor row; a wavefront is never half live)
      if (first_half) {
        // mid-period: the state goes back as it is, nothing of the period's end happens yet
        if (w0) {
          V3<T> wb[3];
          store_sim(k.st, m, e, S, lag, wb);
          if (k.hp) row_store_pose<T, TK>(M, k.hp, m, e);
          k.ist[I_PHASE * m + e] = phase;  // (a freshly set up row leaves PH_INIT here)
          k.ist[I_HALF * m + e] = 1;
          if (lag.overflow) atomicAdd(&k.counters[0], (unsigned long long)lag.overflow);
        }
        return;
      }
      if (donor_run && w0) k.ist[I_HALF * m + e] = 0;
    }
    T* const orow = sh.cd + slot * (row::C_SLAB * row::n_pages<MODEL>() * 16);  // the contact slots are dead now
    if (k.hp && active && w0) row_store_pose<T, TK>(M, k.hp, m, e);  // (before the observation line takes over the contact slab; kd stays)
    finish_lane<T, true>(k, S, lag, e, valid, live, active, w0, phase, total, orow, orow + 64);  // (orow: <= 49 values of the row's 480)
#ifdef TGR_WAVE_TIMES
    if ((threadIdx.x & 63u) == 0) {
      const int wv = blockIdx.x * (ROW_ENVS / 4) + (threadIdx.x >> 6);
      if (wv < 16384) { g_wave_times[4 * wv] = wt0; g_wave_times[4 * wv + 1] = wall_clock64(); g_wave_times[4 * wv + 2] = (unsigned long long)lag.n_it; g_wave_times[4 * wv + 3] = __popcll(__ballot(active && w0)); }
    }
#endif
    // the observation of a live env leaves through the row's tile line: lane 0 wrote it, the 16 lanes store it.  A row never spans two
    // wavefronts (static_assert below), and LDS operations of one wavefront complete in order; the fence keeps the compiler from moving
    // the loads above lane 0's stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live)
      for (int c = r; c < D; c += 16) k.obs[(size_t)e * D + c] = orow[c];
    if (k.stagger_pct > 0 && k.lane_base == 0 && (tx2 & 63u) == 0)  // (the live launch's end: the latest of its wavefronts' -- KArgs::stagger_pct)
      atomicMax(k.counters + STAMP_SLOT + 2 * (k.launch_idx & 1) + 1, ((unsigned long long)(unsigned)k.launch_idx << 40) | (wall_clock64() & ((1ull << 40) - 1ull)));
  }
}

// reset_model() in one go, row layout: one row per entry of list[] (tgr_reset, and live envs that found no ready donor).
// Nothing of an environment's bookkeeping lives in registers across the twenty sub-steps of a settle period: who the row is gets worked out
// again after them (as in step_kernel_row), and what reset_model() carries from period to period -- heading deque, _step_num, reset_psi,
// way-point, the lagged centre -- goes through the env's own words in global memory, which the reset rewrites anyway.  (Carried in
// registers through the 50 x 20 sub-steps it cost 111 spilled VGPRs, 324 B of scratch per lane.)
template <typename T> struct SettleRow { int e, klane, episode; bool valid; unsigned long long key; };
template <typename T> TGR_DEV SettleRow<T> settle_row_of(const KArgs<T>& k, int li, int count) {
  SettleRow<T> w;
  w.valid = li < count;
  const int idx = w.valid ? li : count - 1;
  w.e = k.list[idx];
  // reset streams: a live env that found its stream episode not settled yet settles it here itself -- same (class key, episode), same state
  const int sep = k.list_ep ? k.list_ep[idx] : -1;
  w.klane = sep >= 0 ? k.n + (w.e / k.E.stream_envs) * k.E.stream_donors : w.e;  // the lane whose key the episode is drawn under
  w.episode = sep >= 0 ? sep : k.ist[I_EPISODE * k.m + w.e];
  w.key = lane_key(k.E, k.n, w.klane);
  return w;
}
template <typename T, typename TK, int MODEL> __global__ __launch_bounds__((16 * row_envs<T, MODEL>()), (TGR_ROW_WAVES(T, MODEL))) void settle_kernel_row(RowKArgs<T, TK> args) {
  const KArgs<T>& k = args.k;
  const RowArgs<T, TK>& ra = args.ra;
  constexpr int ROW_ENVS = row_envs<T, MODEL>();
  __shared__ RowShared<T, TK, MODEL> sh;
  const int count = *k.list_count;
  if (blockIdx.x * ROW_ENVS >= count) return;
  if (!__any((int)(blockIdx.x * ROW_ENVS + (threadIdx.x >> 4)) < count)) return;  // (no barrier follows in this kernel)
  const int total = settle_total(k.E);
  {
    const int r = (int)(threadIdx.x & 15u), slot = (int)(threadIdx.x >> 4);
    const SettleRow<T> w = settle_row_of(k, blockIdx.x * ROW_ENVS + slot, count);
    Sim<T> S;
    init_episode(k.E, w.key, (unsigned long long)(unsigned)w.episode, S, (double*)(sh.cd + slot * (row::C_SLAB * row::n_pages<MODEL>() * 16)));
    const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, slot, r);
    row_import<T, TK, MODEL>(S, M);
    row::store_role_consts<TK, T, TK, MODEL>(ra.PK, M.rd);
    if (w.valid && r == 0) {  // what the periods below read back
      k.ist[I_STEPNUM * k.m + w.e] = 0;
      k.st[F_RESETPSI * k.m + w.e] = T(0);
      aux_st(k, HP_XYLAG, F_XYLAG, w.e, 0.0); aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, w.e, 0.0);
    }
  }
  row::RLag<TK, T> rl;
  row_lag_init<T, TK>(rl);
#pragma unroll 1
  for (int st = 0; st < total; st++) {
    {
      const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, (int)(threadIdx.x >> 4), (int)(threadIdx.x & 15u));
#pragma unroll 1
      for (int s = 0; s + 1 < k.E.frame_skip; s++) row::substep<TK, T, TK, T, MODEL, false>(arg_PK<T, TK>(), arg_P<T, TK>(), M, rl);
      row::substep<TK, T, TK, T, MODEL, true>(arg_PK<T, TK>(), arg_P<T, TK>(), M, rl);
    }
    // the period's bookkeeping (tensegrity_env.py:495-510, tr_env.py:811-866): one lane of the row, through global memory
    unsigned tx2 = TGR_ROW_ENVS == 4 ? __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : threadIdx.x, bx2 = blockIdx.x;
    asm volatile("" : "+v"(tx2));
    asm volatile("" : "+s"(bx2));
    const int r = (int)(tx2 & 15u), slot = (int)(tx2 >> 4);
    const SettleRow<T> w = settle_row_of(k, (int)bx2 * ROW_ENVS + slot, *k.list_count);
    if (w.valid && r == 0) {
      const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, slot, r);
      Sim<T> S;
      Lag<T> lag;
      row_export<T, TK, MODEL>(M, rl, S, lag);
      const int m = k.m, e = w.e;
      RingRef<T> ring; ring.ring = k.ring; ring.n = m; ring.e = e;
      ring.len = k.ist[I_RINGLEN * m + e]; ring.head = k.ist[I_RINGHEAD * m + e];
      int step_num = k.ist[I_STEPNUM * m + e];
      T reset_psi = k.st[F_RESETPSI * m + e];
      ModeState<T> ms = load_modes(k, e);
      double xy_prev[2] = {aux_ld(k, HP_XYLAG, F_XYLAG, e), aux_ld(k, HP_XYLAG + 1, F_XYLAG + 1, e)};
      settle_period_done(k.P, k.E, lag, st, ring, step_num, reset_psi, ms, xy_prev, w.key, (unsigned long long)(unsigned)w.episode);
      k.ist[I_RINGLEN * m + e] = ring.len; k.ist[I_RINGHEAD * m + e] = ring.head;
      k.ist[I_STEPNUM * m + e] = step_num;
      k.st[F_RESETPSI * m + e] = reset_psi;
      store_modes(k, e, ms);
      aux_st(k, HP_XYLAG, F_XYLAG, e, xy_prev[0]); aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, e, xy_prev[1]);
    }
  }
  unsigned tx2 = TGR_ROW_ENVS == 4 ? __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : threadIdx.x, bx2 = blockIdx.x;
  asm volatile("" : "+v"(tx2));
  asm volatile("" : "+s"(bx2));
  const int r = (int)(tx2 & 15u), slot = (int)(tx2 >> 4);
  const SettleRow<T> w = settle_row_of(k, (int)bx2 * ROW_ENVS + slot, *k.list_count);
  if (!(w.valid && r == 0)) return;
  const int n = k.n, m = k.m, e = w.e;
  const row::RMem<TK, T> M = row_mem<T, TK, MODEL>(sh, slot, r);
  Sim<T> S;
  Lag<T> lag;
  row_export<T, TK, MODEL>(M, rl, S, lag);
  V3<T> wb[3];
  store_sim(k.st, m, e, S, lag, wb);
  if (k.hp) row_store_pose<T, TK>(M, k.hp, m, e);
  RingRef<T> ring; ring.ring = k.ring; ring.n = m; ring.e = e;
  ring.len = k.ist[I_RINGLEN * m + e]; ring.head = k.ist[I_RINGHEAD * m + e];
  const int step_num = k.ist[I_STEPNUM * m + e];
  store_reset_aux(k, e, lag, k.st[F_RESETPSI * m + e], step_num, ring);
  ModeState<T> ms = load_modes(k, e);
  ms.step_num = step_num;
  k.ist[I_SRC * m + e] = w.klane;
  k.ist[I_SRCEP * m + e] = w.episode;
  if (w.klane == e) k.ist[I_EPISODE * m + e] = w.episode + 1;  // (a stream episode does not advance the env's own counter of explicit resets)
  k.ist[I_PHASE * m + e] = e < n ? PH_LIVE : PH_READY;
  if (k.async && e >= n) { k.dready[e] = k.launch_idx; k.consumed[e] = 0; k.dcmd[e] = 0ull; k.dapplied[e] = 0; }  // (full reset: the streams are idle)
  atomicAdd(&k.counters[1], 1ull);
  if (lag.overflow) atomicAdd(&k.counters[0], (unsigned long long)lag.overflow);
  const int D = k.E.obs_dim;
  T* orow = sh.cd + slot * (row::C_SLAB * row::n_pages<MODEL>() * 16);
  make_obs(k.P, k.E, S, lag, wb, ms, orow);
  if (k.E.use_obs_noise) apply_obs_noise<T, true>(k.E, w.key, (unsigned)w.episode, 0, orow, orow + 64);
  T* op = e < n ? (k.obs_reset ? k.obs_reset + (size_t)e * D : nullptr) : k.dobs + (size_t)(e - n) * D;
  if (op)
    for (int i = 0; i < D; i++) op[i] = orow[i];
}

// ------------------------------------------------------------------ settle kernel: reset_model() in one go
// one lane per entry of list[]; used by tgr_reset() and for live envs that found no ready donor.
// reproduces tensegrity_env.py:433-512 / tr_env.py:709-872.
template <typename T, int LW, int MODEL> __global__ __launch_bounds__(256) void settle_kernel(KArgs<T> k) {
  __shared__ T lds_all[lds_slots<MODEL>() * LW];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (lane >= LW) return;
  const int count = *k.list_count;
  if (blockIdx.x * LW >= count) return;
  const int li = blockIdx.x * LW + lane;
  const bool valid = li < count;
  const int e = k.list[valid ? li : count - 1];
  const int n = k.n, m = k.m;
  const Params<T>& P = k.P;
  const EnvCfg<T>& E = k.E;
  T* lds = lds_all + lane;

  int episode = k.ist[I_EPISODE * m + e];
  Sim<T> S;
  double ie_tmp[39];
  init_episode(E, lane_key(E, n, e), (unsigned long long)(unsigned)episode, S, ie_tmp);
  RingRef<T> ring; ring.ring = k.ring; ring.n = m; ring.e = e;
  ring.len = k.ist[I_RINGLEN * m + e]; ring.head = k.ist[I_RINGHEAD * m + e];
  int step_num = 0;
  const int total = settle_total(E);
  Lag<T> lag;
  lag.cfrc_absmax = T(0); lag.cfrc_clip_sq = T(0); lag.overflow = 0; lag.n_it = 0; lag.n_ls = 0; lag.n_unconv = 0;
  lag.bar_contact = T(0);
#pragma unroll
  for (int i = 0; i < 9; i++) lag.tenL[i] = T(0);  // wave 3 never evaluates tendons
  T reset_psi = T(0);
  ModeState<T> ms = load_modes(k, e);
  double xy_prev[2] = {0.0, 0.0};
  warm_to_lds<T, LW>(S, lds, wid);
#pragma unroll 1
  for (int st = 0; st < total; st++) {
#pragma unroll 1
    for (int s = 0; s + 1 < E.frame_skip; s++) substep<T, LW, false, MODEL>(P, S, lds, wid, lag);
    substep<T, LW, true, MODEL>(P, S, lds, wid, lag);
    if (valid) settle_period_done(P, E, lag, st, ring, step_num, reset_psi, ms, xy_prev, lane_key(E, n, e), (unsigned long long)(unsigned)episode);
  }
  if (!valid || wid != 0) return;
  V3<T> wb[3];
  warm_from_lds<T, LW>(S, lds);
  store_sim(k.st, m, e, S, lag, wb);
  store_reset_aux(k, e, lag, reset_psi, step_num, ring);
  store_modes(k, e, ms);
  ms.step_num = step_num;
  k.ist[I_SRC * m + e] = e;
  k.ist[I_SRCEP * m + e] = episode;
  k.ist[I_EPISODE * m + e] = episode + 1;
  k.ist[I_PHASE * m + e] = e < n ? PH_LIVE : PH_READY;
  atomicAdd(&k.counters[1], 1ull);
  if (lag.overflow) atomicAdd(&k.counters[0], (unsigned long long)lag.overflow);
  const int D = E.obs_dim;
  T* orow = lds_all + lane * ((D & 1) ? D : D + 1);  // wave 0 only from here on; the columns are dead
  make_obs(P, E, S, lag, wb, ms, orow);
  if (E.use_obs_noise) apply_obs_noise(E, lane_key(E, n, e), (unsigned)episode, 0, orow);
  T* op = e < n ? (k.obs_reset ? k.obs_reset + (size_t)e * D : nullptr) : k.dobs + (size_t)(e - n) * D;
  if (op)
    for (int i = 0; i < D; i++) op[i] = orow[i];
}

// ------------------------------------------------------------------ match kernel (one block)
// pairs the k-th finished live env (ascending index) with the k-th ready donor (ascending index), copies the donor's
// settled episode into the env's lane, re-arms the donor, and lists the envs that found no donor for settle_kernel.
template <typename T> __global__ __launch_bounds__(1024) void match_kernel(KArgs<T> k) {
  __shared__ int cnt[1024];
  __shared__ int nd_s, nr_s;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int n = k.n, m = k.m, nd_lanes = m - n;
  // ordered compaction of done slots -> list (temporarily) ; ready donors -> rlist
  auto compact = [&](int total, auto pred, int* out, int* total_out) {
    int chunk = (total + NT - 1) / NT;
    int lo = min(tid * chunk, total), hi = min(lo + chunk, total);
    int c = 0;
    for (int i = lo; i < hi; i++) c += pred(i) ? 1 : 0;
    cnt[tid] = c;
    __syncthreads();
    // exclusive scan (Hillis-Steele on NT entries)
    for (int off = 1; off < NT; off <<= 1) {
      int v = tid >= off ? cnt[tid - off] : 0;
      __syncthreads();
      cnt[tid] += v;
      __syncthreads();
    }
    int start = cnt[tid] - c;
    for (int i = lo; i < hi; i++)
      if (pred(i)) out[start++] = i;
    if (tid == NT - 1) *total_out = cnt[tid];
    __syncthreads();
  };
  // a donor lane whose settled episode can be handed out: phase READY -- or, with the donor stream, ready as of a donor launch that has
  // completed (dready < this step's index) and not handed out yet
  auto donor_ready = [&](int lane) {
    if (!k.async) return k.ist[I_PHASE * m + lane] == PH_READY;
    const int rd = k.dready[lane];
    // (rd > tag of the lane's last command: an episode commanded at step t cannot be ready before donor launch t + 1 -- a dready at or
    // below it belongs to the episode the command replaced, whatever the donor side has or has not cleared yet)
    return rd != 0 && rd < k.launch_idx && rd != k.consumed[lane] && rd > (int)(unsigned)(k.dcmd[lane] & 0xffffffffull);
  };
  compact(n, [&](int i) { return k.doneflag[i] != 0; }, k.list, &nd_s);
  compact(nd_lanes, [&](int i) { return donor_ready(n + i); }, k.rlist, &nr_s);
  const int nd = nd_s, nr = nr_s;
  const int pairs = min(nd, nr);
  const int D = k.E.obs_dim;
  // copy donor -> slot : state fields, int fields, heading ring, reset observation
  const int NH = k.hp ? HP_ROWS : 0;
  const int per = NF + RING + NH + D;
  for (int w = tid; w < pairs * per; w += NT) {
    int p = w / per, f = w - p * per;
    int slot = k.list[p], donor = n + k.rlist[p];
    if (f < NF) k.st[f * m + slot] = k.st[f * m + donor];
    else if (f < NF + RING) k.ring[(f - NF) * m + slot] = k.ring[(f - NF) * m + donor];
    else if (f < NF + RING + NH) k.hp[(f - NF - RING) * m + slot] = k.hp[(f - NF - RING) * m + donor];
    else if (k.obs_reset) k.obs_reset[(size_t)slot * D + (f - NF - RING - NH)] = k.dobs[(size_t)(donor - n) * D + (f - NF - RING - NH)];
  }
  __syncthreads();
  for (int p = tid; p < pairs; p += NT) {
    int slot = k.list[p], donor = n + k.rlist[p];
    k.ist[I_STEPNUM * m + slot] = k.ist[I_STEPNUM * m + donor];
    k.ist[I_ELAPSED * m + slot] = 0;
    k.ist[I_RINGLEN * m + slot] = k.ist[I_RINGLEN * m + donor];
    k.ist[I_RINGHEAD * m + slot] = k.ist[I_RINGHEAD * m + donor];
    k.ist[I_SRC * m + slot] = donor;
    k.ist[I_SRCEP * m + slot] = k.ist[I_SRCEP * m + donor];
    if (k.async) { k.consumed[donor] = k.dready[donor]; k.dcmd[donor] = (unsigned long long)(unsigned)k.launch_idx; }
    else { k.ist[I_PHASE * m + donor] = PH_INIT; k.ist[I_HALF * m + donor] = 0; }
    k.doneflag[slot] = 0;
  }
  __syncthreads();
  // starved envs: shift to the front of list[] for the synchronous settle
  int starved = nd - pairs;
  for (int p = tid; p < starved; p += NT) k.rlist[p] = k.list[pairs + p];
  __syncthreads();
  for (int p = tid; p < starved; p += NT) { k.list[p] = k.rlist[p]; k.doneflag[k.rlist[p]] = 0; }
  if (tid == 0) {
    *k.list_count = starved;
    if (pairs) atomicAdd(&k.counters[2], (unsigned long long)pairs);
  }
}

// ------------------------------------------------------------------ match kernel, reset streams (one block per class)
// A class = stream_envs consecutive envs + stream_donors donor lanes.  Its donors settle the episodes (class key, 0), (class key, 1), ...
// of ONE stream; finished envs of the class take them strictly in order -- the k-th reset of the class (by step, then by env index) gets
// episode k, whichever donor holds it.  Every donor needs the same number of launches (no paced rows in this mode), so episodes become
// ready in the order they were started and the ready ones are the contiguous range [next_hand, next_hand + nr).  An env whose episode is
// not ready settles it itself (settle kernel, same key and episode -> same state) and the donor working on it starts a new one.  What an
// env restarts with is therefore a function of its class's history alone: identical for every sharding whose shards are whole classes.
template <typename T> __global__ __launch_bounds__(256) void match_kernel_stream(KArgs<T> k) {
  __shared__ int cnt[256];
  __shared__ int nd_s, nr_s;
  const int tid = threadIdx.x, NT = blockDim.x, c = blockIdx.x;
  const int n = k.n, m = k.m, B = k.E.stream_envs, Dc = k.E.stream_donors, TB = k.stream_table;
  const int e0 = c * B, d0 = n + c * Dc;
  int* const list = k.rlist + m + e0;  // this class's finished envs, ascending (scratch: the second half of rlist)
  int* const rl = k.rlist + c * Dc;    // donor lane holding episode next_hand + i, or -1
  int* const hold = k.holder + c * TB;
  const int next_start = k.cls[2 * c], next_hand = k.cls[2 * c + 1];
  {  // ordered compaction of the class's done flags
    const int chunk = (B + NT - 1) / NT;
    const int lo = min(tid * chunk, B), hi = min(lo + chunk, B);
    int cc = 0;
    for (int i = lo; i < hi; i++) cc += k.doneflag[e0 + i] != 0 ? 1 : 0;
    cnt[tid] = cc;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
      int v = tid >= off ? cnt[tid - off] : 0;
      __syncthreads();
      cnt[tid] += v;
      __syncthreads();
    }
    int start = cnt[tid] - cc;
    for (int i = lo; i < hi; i++)
      if (k.doneflag[e0 + i] != 0) list[start++] = e0 + i;
    if (tid == NT - 1) nd_s = cnt[tid];
  }
  if (tid == 0) nr_s = 0;
  for (int j = tid; j < Dc; j += NT) rl[j] = -1;
  __syncthreads();
  for (int j = tid; j < Dc; j += NT) {
    bool ready;
    // (rd > tag of the lane's last command: an episode commanded at step t cannot be ready before donor launch t + 1.  A dready at or below
    // it is the REPLACED episode's -- the starved path below re-arms a holder without touching consumed[], and the donor side clears dready
    // only at the end of the launch that carries the command out, while I_SRCEP already names the new episode)
    if (k.async) { const int rd = k.dready[d0 + j]; ready = rd != 0 && rd < k.launch_idx && rd != k.consumed[d0 + j] && rd > (int)(unsigned)(k.dcmd[d0 + j] & 0xffffffffull); }
    else ready = k.ist[I_PHASE * m + d0 + j] == PH_READY;
    if (ready) {
      const int off = k.ist[I_SRCEP * m + d0 + j] - next_hand;
      if (off >= 0 && off < Dc) { rl[off] = d0 + j; atomicAdd(&nr_s, 1); }
    }
  }
  __syncthreads();
  const int nd = nd_s;
  if (nd == 0) return;
  // the ready prefix: episodes next_hand, next_hand + 1, ... as long as their holders are ready
  int nr = 0;
  if (tid == 0) {
    while (nr < nr_s && nr < nd && rl[nr] >= 0) nr++;
    nr_s = nr;
  }
  __syncthreads();
  nr = nr_s;
  const int pairs = min(nd, nr);
  const int D = k.E.obs_dim;
  const int NH = k.hp ? HP_ROWS : 0;
  const int per = NF + RING + NH + D;
  for (int w = tid; w < pairs * per; w += NT) {
    const int p = w / per, f = w - p * per;
    const int slot = list[p], donor = rl[p];
    if (f < NF) k.st[f * m + slot] = k.st[f * m + donor];
    else if (f < NF + RING) k.ring[(f - NF) * m + slot] = k.ring[(f - NF) * m + donor];
    else if (f < NF + RING + NH) k.hp[(f - NF - RING) * m + slot] = k.hp[(f - NF - RING) * m + donor];
    else if (k.obs_reset) k.obs_reset[(size_t)slot * D + (f - NF - RING - NH)] = k.dobs[(size_t)(donor - n) * D + (f - NF - RING - NH)];
  }
  __syncthreads();
  for (int p = tid; p < pairs; p += NT) {
    const int slot = list[p], donor = rl[p];
    k.ist[I_STEPNUM * m + slot] = k.ist[I_STEPNUM * m + donor];
    k.ist[I_ELAPSED * m + slot] = 0;
    k.ist[I_RINGLEN * m + slot] = k.ist[I_RINGLEN * m + donor];
    k.ist[I_RINGHEAD * m + slot] = k.ist[I_RINGHEAD * m + donor];
    k.ist[I_SRC * m + slot] = donor;
    k.ist[I_SRCEP * m + slot] = k.ist[I_SRCEP * m + donor];
    // the donor starts the next episode nobody has started yet
    if (k.async) {
      k.consumed[donor] = k.dready[donor];
      k.dcmd[donor] = ((unsigned long long)(unsigned)(next_start + p) << 32) | (unsigned long long)(unsigned)k.launch_idx;
    } else {
      k.ist[I_PHASE * m + donor] = PH_INIT;
      k.ist[I_EPISODE * m + donor] = next_start + p;
      k.ist[I_HALF * m + donor] = 0;
    }
    hold[(next_start + p) & (TB - 1)] = donor;
    k.doneflag[slot] = 0;
  }
  __syncthreads();
  // envs whose episode is not ready: they settle it themselves; a donor that is working on it takes a new one
  if (tid == 0) {
    const int starved = nd - pairs;
    int fresh = next_start + pairs;  // next episode index nobody holds
    if (starved > 0) {
      const int base = atomicAdd(k.list_count, starved);
      for (int s_ = 0; s_ < starved; s_++) {
        const int env = list[pairs + s_], ep = next_hand + pairs + s_;
        k.list_ep[base + s_] = ep;
        k.list[base + s_] = env;
        k.doneflag[env] = 0;
        if (ep < next_start) {  // somebody is settling it (or holds it behind a gap): re-arm that donor
          const int donor = hold[ep & (TB - 1)];
          if (k.async) k.dcmd[donor] = ((unsigned long long)(unsigned)fresh << 32) | (unsigned long long)(unsigned)k.launch_idx;
          else { k.ist[I_PHASE * m + donor] = PH_INIT; k.ist[I_EPISODE * m + donor] = fresh; k.ist[I_HALF * m + donor] = 0; }
          hold[fresh & (TB - 1)] = donor;
          fresh++;
        }
      }
    }
    const int hand = next_hand + nd;
    k.cls[2 * c] = fresh > hand ? fresh : hand;
    k.cls[2 * c + 1] = hand;
    if (pairs) atomicAdd(&k.counters[2], (unsigned long long)pairs);
  }
}
// set_state: recompute the derived (lagged) fields from qpos the way mj_forward would
template <typename T> __global__ void refresh_lag_kernel(KArgs<T> k) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= k.n) return;
  const int m = k.m;
  Sim<T> S;
  load_sim(k.st, m, e, S);
  Lag<T> lag;
  // the centre in double, from the fp64 pose where the handle keeps one; the bar positions relative to it (the headings below are
  // differences of positions)
  double px[3][3], c[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const T xs[3] = {S.x[b].x, S.x[b].y, S.x[b].z};
#pragma unroll
    for (int i = 0; i < 3; i++) { px[b][i] = k.hp ? k.hp[(size_t)(HP_QPOS + 7 * b + i) * m + e] : (double)xs[i]; c[i] += px[b][i] * (1.0 / 3.0); }
  }
#pragma unroll
  for (int b = 0; b < 3; b++) {
    lag.x[b] = mk(T(px[b][0] - c[0]), T(px[b][1] - c[1]), T(px[b][2] - c[2]));
    lag.z[b] = quat_rot(S.q[b]).ez;
  }
  lag.com[0] = c[0]; lag.com[1] = c[1];
  aux_st(k, HP_XYLAG, F_XYLAG, e, c[0]);
  aux_st(k, HP_XYLAG + 1, F_XYLAG + 1, e, c[1]);
  V3<T> cap[6], l, r;
  caps_of(k.P, lag, cap);
  left_right(cap, l, r);
  k.st[F_PSILAG * m + e] = atan2_(-(l.x - r.x), l.y - r.y);
}

}  // namespace tgr

// =================================================================== host side / C-ABI
using namespace tgr;

struct tgr_handle {
  tgr_config cfg;
  int n = 0, m = 0, device = 0, obs_dim = 0;
  int n_cu = 256;  // compute units of the device (how many rows one round of the row kernels holds)
  int stagger_pct = TGR_DONOR_STAGGER_PCT;  // TGR_DONOR_STAGGER_PCT=<percent of a live-row launch>: stagger of the donor launch behind the live one (0: off)
  int pace_from = -1;  // TGR_PACE_FROM=<lane>: test hook -- pace the donor rows from this lane on, whatever the device's size
  void* st = nullptr;   // T[NF][m]
  int* ist = nullptr;   // int[NI][m]
  double* hp = nullptr; // double[HP_ROWS][m]: qpos and the env layer's absolute positions at full precision (fp32 handles on the row kernels), else null
  void* ring = nullptr; // T[RING][m]
  void* dobs = nullptr; // T[m-n][D]
  int* list_count = nullptr;
  int* list = nullptr;
  int* rlist = nullptr;
  int* list_ep = nullptr;  // reset streams (tgr_config.reset_stream_envs > 0): see KArgs
  int* cls = nullptr;
  int* holder = nullptr;
  int n_classes = 0, stream_envs = 0, stream_donors = 0, stream_table = 0;
  // donor stream (see KArgs)
  bool async = false;
  int launch = 1;  // index of the last tgr_step (donor launches and match kernels are numbered by it)
  bool seed_dirty = false;  // tgr_set_seed on a handle with donor lanes: they hold episodes settled under the old key until a full tgr_reset
  hipStream_t sb = nullptr;
  hipEvent_t ev_donor[2] = {nullptr, nullptr}, ev_match = nullptr;
  unsigned long long* dcmd = nullptr; int* dapplied = nullptr; int* dready = nullptr; int* consumed = nullptr;
  // tgr_time_kernels: HIP events around the launch that steps the live rows (the dominant kernel), a ring of tm_n pairs
  std::vector<hipEvent_t> tm_start, tm_stop;
  int tm_n = 0, tm_every = 1; long long tm_count = 0, tm_step = 0;  // (tm_every: only every tm_every-th step is timed)
  uint8_t* doneflag = nullptr;
  unsigned long long* counters = nullptr;
  float* hf = nullptr;  // heightfield elevations (model_kind 1)
  void* obs_clean = nullptr;  // caller-owned [N][D] buffer bound with tgr_bind_real_observation, or null
  unsigned* trace = nullptr;  // caller-owned [N][TGR_TRACE_WORDS] buffer bound with tgr_bind_substep_trace, or null
  int hf_nrow = 0, hf_ncol = 0;
  std::string err;
};

static thread_local std::string g_err;

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                              \
      return TGR_ERR_HIP;                                                                        \
    }                                                                                            \
  } while (0)

static void free_handle(tgr_handle* h) {
  (void)hipFree(h->hf); (void)hipFree(h->hp);
  (void)hipFree(h->st); (void)hipFree(h->ist); (void)hipFree(h->ring); (void)hipFree(h->dobs); (void)hipFree(h->list_count);
  (void)hipFree(h->list); (void)hipFree(h->rlist); (void)hipFree(h->doneflag); (void)hipFree(h->counters);
  (void)hipFree(h->list_ep); (void)hipFree(h->cls); (void)hipFree(h->holder);
  (void)hipFree(h->dcmd); (void)hipFree(h->dapplied); (void)hipFree(h->dready); (void)hipFree(h->consumed);
  if (h->sb) (void)hipStreamDestroy(h->sb);
  for (int i = 0; i < 2; i++) if (h->ev_donor[i]) (void)hipEventDestroy(h->ev_donor[i]);
  if (h->ev_match) (void)hipEventDestroy(h->ev_match);
  for (hipEvent_t e : h->tm_start) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->tm_stop) (void)hipEventDestroy(e);
  delete h;
}

#ifndef TGR_LW_F32
#define TGR_LW_F32 64
#endif
#ifndef TGR_LW_F64
#define TGR_LW_F64 32
#endif
template <typename T> struct LaneW { static constexpr int value = sizeof(T) == 4 ? TGR_LW_F32 : TGR_LW_F64; };  // lanes (envs) per block

extern "C" const char* tgr_version(void) { return "tensegrity-mi355x 0.2 (gfx950)"; }
extern "C" int tgr_sizeof_config(void) { return (int)sizeof(tgr_config); }

extern "C" int tgr_default_config(tgr_config* c, int env_kind, int desired_action) {
  if (!c) return TGR_ERR_INVALID;
  std::memset(c, 0, sizeof(*c));
  c->env_kind = env_kind;
  c->desired_action = desired_action;
  c->desired_direction = 1;
  c->terminate_when_unhealthy = 1;
  c->use_contact_forces = 0;
  c->use_cap_velocity = 1;
  c->is_test = 0;
  c->contact_cost_weight = 5e-4;
  c->healthy_reward = 0.1;
  c->contact_force_lo = -1.0; c->contact_force_hi = 1.0;
  if (desired_action == TGR_ACT_TURN) { c->contact_force_lo = -1000.0; c->contact_force_hi = 1000.0; }
  c->min_reset_heading = 0.0; c->max_reset_heading = 2 * PI;
  c->tendon_min_length = -0.45;
  c->frame_skip = 20;
  c->max_episode_steps = 5000;
  c->timestep = 0.001;
  double dt = c->timestep * c->frame_skip;
  if (env_kind == TGR_ENV_TENSEGRITY) {  // tensegrity_env.py:160-181
    c->ctrl_cost_weight = 0.001;
    c->tendon_reset_mean = -0.15; c->tendon_reset_stdev = 0.1; c->tendon_max_length = -0.15;
    c->reward_delay_steps = (int)(0.5 / dt);
  } else {  // tr_env.py:137-173
    c->ctrl_cost_weight = 0.01;
    c->tendon_reset_mean = 0.15; c->tendon_reset_stdev = 0.2; c->tendon_max_length = 0.15;
    c->reward_delay_steps = (int)(0.02 / dt);
  }
  c->gravity = 9.81;
  c->cap_radius = 0.0675;
  c->solref[0] = -5000; c->solref[1] = -500;
  c->solimp[0] = 0.9; c->solimp[1] = 0.95; c->solimp[2] = 0.001; c->solimp[3] = 0.5; c->solimp[4] = 2;
  c->friction[0] = 1; c->friction[1] = 0.005; c->friction[2] = 0.0001;
  c->impratio = 1;
  c->gain = 6667; c->bias[0] = 3290; c->bias[1] = -6666; c->bias[2] = -133;
  c->force_range[0] = -267; c->force_range[1] = 0;
  c->ctrl_range[0] = -0.45; c->ctrl_range[1] = 0.15;
  for (int t = 0; t < 9; t++) {
    c->ten_stiffness[t] = t >= 6 ? 700.0 : 0.0;
    c->ten_damping[t] = t >= 6 ? 100.0 : 0.0;
    c->ten_springlength[t] = 0.8;
  }
  c->actuator_deriv_skip_clamped = 0;
  c->precision = TGR_PRECISION_F32;
  c->auto_reset = 1;
  c->n_donors = -1;
  c->seed = 1234;
  c->first_env_id = 0;
  c->solver_tolerance = 0;
  c->solver_iterations = 0;
  c->waypt_range[0] = 2.5; c->waypt_range[1] = 3.5;  // tr_env.py:166-172
  c->waypt_angle_range[0] = -PI / 6; c->waypt_angle_range[1] = PI / 6;
  c->ditch_reward_max = 300; c->ditch_reward_stdev = 0.15; c->waypt_reward_amplitude = 100; c->waypt_reward_stdev = 0.10;
  c->yaw_reward_weight = 1;
  c->solver_warm_start = 0;
  c->use_obs_noise = 0; c->obs_noise_tendon_stdev = 0.02; c->obs_noise_cap_pos_stdev = 0.05;
  c->model_kind = 0; c->site_offset = 0.0675; c->act_tau = 1.0;
  c->hf_size[0] = 50; c->hf_size[1] = 50; c->hf_size[2] = 1; c->hf_size[3] = 0.1;
  c->hf_pos[0] = 0; c->hf_pos[1] = 0; c->hf_pos[2] = -1.0;
  c->reset_noise_scale = 0.0;
  c->kernel_layout = TGR_KERNEL_AUTO;
  c->solver_rel_tolerance = 0;
  c->use_cap_size_noise = 0; c->cap_size_noise_range[0] = 0.04; c->cap_size_noise_range[1] = 0.09;  // tr_env.py:143,163
  return TGR_OK;
}

// 3prism_jonathan_steady_side_uneven_ground.xml vs the flat model: xml:38-39 (tendon and actuator defaults), :48,56 (heightfield),
// :69-70 (caps), :76-81 (sites), :122-158 (tendons)
extern "C" int tgr_uneven_config(tgr_config* c) {
  if (!c) return TGR_ERR_INVALID;
  c->model_kind = 1;
  c->cap_radius = 0.05;
  c->site_offset = 0.05;
  for (int t = 0; t < 9; t++) { c->ten_stiffness[t] = 10000.0; c->ten_damping[t] = 1000.0; c->ten_springlength[t] = t < 6 ? 0.5 : 0.8; }
  c->gain = 15000.0; c->bias[0] = c->bias[1] = c->bias[2] = 0.0;
  c->force_range[0] = -1e30; c->force_range[1] = 1e30;  // forcelimited is not set
  c->act_tau = 1.0;
  c->hf_size[0] = 50; c->hf_size[1] = 50; c->hf_size[2] = 1; c->hf_size[3] = 0.1;
  c->hf_pos[0] = 0; c->hf_pos[1] = 0; c->hf_pos[2] = -1.0;
  return TGR_OK;
}

static int obs_dim_of(const tgr_config& c) {
  if (c.env_kind == TGR_ENV_TENSEGRITY) return 39;
  int d = 27;
  if (c.use_cap_velocity) d += 18;
  if (c.desired_action == TGR_ACT_TRACKING || c.desired_action == TGR_ACT_AIMING || c.desired_action == TGR_ACT_VEL_TRACK) d += 3;
  return d;
}

static bool use_row_kernels(const tgr_handle* h);
template <typename T> static void fill_args(const tgr_handle* h, KArgs<T>& k) {
  const tgr_config& c = h->cfg;
  Params<T>& P = k.P;
  P.h = T(c.timestep); P.grav = T(c.gravity);
  // inertiafromgeom with explicit geom masses (xml:72-78): cylinder 1.0, two spheres 0.5, two cylinders 1.0
  const double half = 0.688, rr = 0.0381, rs = c.cap_radius, rb = 0.035, blo = 0.238, bhi = 0.688;
  double L = 2 * half;
  double Ixx = 1.0 * (3 * rr * rr + L * L) / 12.0, Izz = 1.0 * rr * rr / 2.0;
  double Is = 0.4 * 0.5 * rs * rs;
  Ixx += 2 * (Is + 0.5 * half * half); Izz += 2 * Is;
  double Lb = bhi - blo, cb = 0.5 * (bhi + blo);
  Ixx += 2 * (1.0 * (3 * rb * rb + Lb * Lb) / 12.0 + 1.0 * cb * cb); Izz += 2 * (1.0 * rb * rb / 2.0);
  P.mass = T(4.0); P.Ixx = T(Ixx); P.Izz = T(Izz);
  P.cap_r = T(c.cap_radius); P.bar_half = T(half); P.rcyl_r = T(rr); P.bcyl_r = T(rb); P.bcyl_lo = T(blo); P.bcyl_hi = T(bhi);
  P.site_off = T(c.site_offset); P.site_z = T(0.668);
  for (int t = 0; t < 9; t++) { P.ten_k[t] = T(c.ten_stiffness[t]); P.ten_b[t] = T(c.ten_damping[t]); P.ten_L0[t] = T(c.ten_springlength[t]); }
  P.gain = T(c.gain); P.bias0 = T(c.bias[0]); P.bias1 = T(c.bias[1]); P.bias2 = T(c.bias[2]);
  P.force_lo = T(c.force_range[0]); P.force_hi = T(c.force_range[1]); P.ctrl_lo = T(c.ctrl_range[0]); P.ctrl_hi = T(c.ctrl_range[1]);
  double dmax = c.solimp[1];
  P.K = T(-c.solref[0] / (dmax * dmax)); P.B = T(-c.solref[1] / dmax);  // direct (negative) solref
  P.imp0 = T(c.solimp[0]); P.imp1 = T(c.solimp[1]); P.imp_width = T(c.solimp[2]); P.imp_mid = T(c.solimp[3]);
  P.f0 = T(c.friction[0]); P.f1sq = T(c.friction[0] * c.friction[0]); P.f3sq = T(c.friction[1] * c.friction[1]);
  P.f4sq = T(c.friction[2] * c.friction[2]); P.impratio = T(c.impratio);
  P.invweight = T(0.25);
  double tol = c.solver_tolerance > 0 ? c.solver_tolerance : (sizeof(T) == 4 ? 3e-6 : 1e-11);  // fp32: accuracy/speed trade-off measured in DESIGN.md section 4
  P.tol = T(tol); P.ls_tol = T(0.01);
  if (const char* v = getenv("TGR_LS_TOL")) P.ls_tol = T(atof(v));  // (A/B measurements of the line-search tolerance)
  P.max_iter = c.solver_iterations > 0 ? c.solver_iterations : (sizeof(T) == 4 ? 8 : 40);
  P.ls_iter = sizeof(T) == 4 ? 6 : 50;
  P.skip_clamped = c.actuator_deriv_skip_clamped;
  P.warm_start = c.solver_warm_start;
  P.act_itau = T(1.0 / (c.act_tau > 1e-12 ? c.act_tau : 1e-12));
  P.hf = h->hf; P.hf_nrow = h->hf_nrow; P.hf_ncol = h->hf_ncol;
  P.hf_rx = T(c.hf_size[0]); P.hf_ry = T(c.hf_size[1]); P.hf_ez = T(c.hf_size[2]);
  P.hf_x0 = T(c.hf_pos[0]); P.hf_y0 = T(c.hf_pos[1]); P.hf_z0 = T(c.hf_pos[2]);
  P.hf_dx = T(h->hf_ncol > 1 ? 2 * c.hf_size[0] / (h->hf_ncol - 1) : 1.0); P.hf_dy = T(h->hf_nrow > 1 ? 2 * c.hf_size[1] / (h->hf_nrow - 1) : 1.0);
  EnvCfg<T>& E = k.E;
  E.env_kind = c.env_kind; E.action = c.desired_action; E.terminate_when_unhealthy = c.terminate_when_unhealthy;
  E.use_contact_forces = c.use_contact_forces; E.use_cap_velocity = c.use_cap_velocity; E.is_test = c.is_test;
  E.reward_delay_steps = c.reward_delay_steps; E.frame_skip = c.frame_skip; E.max_episode_steps = c.max_episode_steps;
  E.obs_dim = h->obs_dim;
  E.use_obs_noise = (c.use_obs_noise && c.env_kind == TGR_ENV_TR) ? 1 : 0;
  E.model_kind = c.model_kind;
  E.noise_tendon = T(c.obs_noise_tendon_stdev); E.noise_cap = T(c.obs_noise_cap_pos_stdev);
  E.direction = T(c.desired_direction); E.ctrl_cost_weight = T(c.ctrl_cost_weight); E.contact_cost_weight = T(c.contact_cost_weight);
  E.healthy_reward = T(c.healthy_reward); E.contact_lo = T(c.contact_force_lo); E.contact_hi = T(c.contact_force_hi);
  E.dt = T(c.timestep * c.frame_skip);
  E.ditch_max = T(c.ditch_reward_max); E.ditch_std = T(c.ditch_reward_stdev); E.waypt_amp = T(c.waypt_reward_amplitude);
  E.waypt_std = T(c.waypt_reward_stdev); E.yaw_weight = T(c.yaw_reward_weight);
  E.waypt_range[0] = c.waypt_range[0]; E.waypt_range[1] = c.waypt_range[1];
  E.waypt_angle_range[0] = c.waypt_angle_range[0]; E.waypt_angle_range[1] = c.waypt_angle_range[1];
  P.cf_lo = T(c.contact_force_lo); P.cf_hi = T(c.contact_force_hi);
  E.min_heading = c.min_reset_heading; E.max_heading = c.max_reset_heading;
  E.ten_mean = c.tendon_reset_mean; E.ten_std = c.tendon_reset_stdev; E.ten_max = c.tendon_max_length; E.ten_min = c.tendon_min_length;
  E.seed = c.seed; E.first_env_id = c.first_env_id; E.reset_noise = c.reset_noise_scale;
  E.stream_envs = h->n_classes > 0 ? h->stream_envs : 0; E.stream_donors = h->stream_donors;
  E.cap_noise = (c.use_cap_size_noise && c.env_kind == TGR_ENV_TR) ? 1 : 0; E.cap_lo = c.cap_size_noise_range[0]; E.cap_hi = c.cap_size_noise_range[1];
  E.cap_nominal = c.cap_radius;
  k.n = h->n; k.m = h->m;
  {
    // Paced donor rows (step_kernel_row): the rows that do not fit the first round of wave slots -- one round holds 4 rows per wavefront,
    // 8 wavefronts per CU for the fp32 flat-floor kernels (4 otherwise) -- but never more than the upper half of the donor pool, and only
    // when the live rows alone leave room in that round (otherwise a launch is several rounds anyway and pacing would only slow resets).
    const int donors = h->m - h->n;
    const int round_rows = h->n_cu * ((c.precision != TGR_PRECISION_F64 && c.model_kind == 0) ? 8 : 4) * 4;
    int from = round_rows > h->n + donors / 2 ? round_rows : h->n + donors / 2;
    from = ((from + 3) / 4) * 4;
    // (reset streams hand episodes out in the order they were started: every donor must need the same number of launches)
    bool pace = c.auto_reset && donors >= 64 && (c.frame_skip % 2) == 0 && h->n < round_rows && from < h->m && h->n_classes == 0 && !h->async;
    if (h->pace_from >= 0) {  // test hook (tests/test_gpu_reset_vecenv.py): which rows are paced no longer depends on n or the CU count
      from = ((h->pace_from > h->n ? h->pace_from : h->n) + 3) / 4 * 4;
      pace = c.auto_reset && donors > 0 && (c.frame_skip % 2) == 0 && from < h->m && h->n_classes == 0 && !h->async;
    }
    k.slow_e = pace ? from : h->m + 4;
  }
  k.st = (T*)h->st; k.ist = h->ist; k.ring = (T*)h->ring; k.dobs = (T*)h->dobs; k.hp = h->hp;
  k.list_count = h->list_count; k.list = h->list; k.rlist = h->rlist; k.doneflag = h->doneflag; k.counters = h->counters;
  k.list_ep = h->list_ep; k.cls = h->cls; k.holder = h->holder; k.stream_table = h->stream_table;
  k.async = h->async ? 1 : 0; k.launch_idx = h->launch; k.lane_base = 0; k.lane_limit = h->m; k.stagger_pct = (h->async && c.precision != TGR_PRECISION_F64 && c.model_kind == 0) ? h->stagger_pct : 0;
  k.dcmd = h->dcmd; k.dapplied = h->dapplied; k.dready = h->dready; k.consumed = h->consumed;
  k.auto_reset = c.auto_reset;
  k.actions = nullptr; k.obs = nullptr; k.reward = nullptr; k.terminated = nullptr; k.truncated = nullptr;
  k.terminal_obs = nullptr; k.info = nullptr; k.obs_reset = nullptr; k.obs_clean = nullptr; k.dbg = nullptr;
}

static int alloc_handle(tgr_handle* h, int n_envs, int donors) {
  HIP_TRY(h, hipSetDevice(h->device));
  size_t ts = h->cfg.precision == TGR_PRECISION_F64 ? 8 : 4;
  size_t m = (size_t)h->m;
  HIP_TRY(h, hipMalloc(&h->st, ts * NF * m));
  HIP_TRY(h, hipMemset(h->st, 0, ts * NF * m));
  HIP_TRY(h, hipMalloc((void**)&h->ist, sizeof(int) * NI * m));
  {
    std::vector<int> ist(NI * m, 0);
    for (size_t e = 0; e < m; e++) ist[I_PHASE * m + e] = e < (size_t)n_envs ? PH_LIVE : PH_INIT;
    HIP_TRY(h, hipMemcpy(h->ist, ist.data(), sizeof(int) * NI * m, hipMemcpyHostToDevice));
  }
  if (h->cfg.precision != TGR_PRECISION_F64 && use_row_kernels(h)) {
    HIP_TRY(h, hipMalloc((void**)&h->hp, sizeof(double) * HP_ROWS * m));
    HIP_TRY(h, hipMemset(h->hp, 0, sizeof(double) * HP_ROWS * m));
  }
  HIP_TRY(h, hipMalloc(&h->ring, ts * RING * m));
  HIP_TRY(h, hipMemset(h->ring, 0, ts * RING * m));
  HIP_TRY(h, hipMalloc(&h->dobs, ts * (size_t)h->obs_dim * (size_t)(donors > 0 ? donors : 1)));
  HIP_TRY(h, hipMalloc((void**)&h->list_count, sizeof(int)));
  HIP_TRY(h, hipMemset(h->list_count, 0, sizeof(int)));
  HIP_TRY(h, hipMalloc((void**)&h->list, sizeof(int) * m));
  HIP_TRY(h, hipMalloc((void**)&h->rlist, sizeof(int) * 2 * m));
  if (h->async) {
    int least = 0, greatest = 0;
    HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(h, hipStreamCreateWithPriority(&h->sb, hipStreamNonBlocking, least));
    for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_donor[i], hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_match, hipEventDisableTiming));
    HIP_TRY(h, hipMalloc((void**)&h->dcmd, sizeof(unsigned long long) * m)); HIP_TRY(h, hipMemset(h->dcmd, 0, sizeof(unsigned long long) * m));
    HIP_TRY(h, hipMalloc((void**)&h->dapplied, sizeof(int) * m)); HIP_TRY(h, hipMemset(h->dapplied, 0, sizeof(int) * m));
    HIP_TRY(h, hipMalloc((void**)&h->dready, sizeof(int) * m)); HIP_TRY(h, hipMemset(h->dready, 0, sizeof(int) * m));
    HIP_TRY(h, hipMalloc((void**)&h->consumed, sizeof(int) * m)); HIP_TRY(h, hipMemset(h->consumed, 0, sizeof(int) * m));
  }
  if (h->n_classes > 0) {
    HIP_TRY(h, hipMalloc((void**)&h->list_ep, sizeof(int) * m));
    HIP_TRY(h, hipMemset(h->list_ep, 0xff, sizeof(int) * m));
    HIP_TRY(h, hipMalloc((void**)&h->cls, sizeof(int) * 2 * h->n_classes));
    HIP_TRY(h, hipMemset(h->cls, 0, sizeof(int) * 2 * h->n_classes));
    HIP_TRY(h, hipMalloc((void**)&h->holder, sizeof(int) * (size_t)h->n_classes * h->stream_table));
    HIP_TRY(h, hipMemset(h->holder, 0, sizeof(int) * (size_t)h->n_classes * h->stream_table));
  }
  HIP_TRY(h, hipMalloc((void**)&h->doneflag, m));
  HIP_TRY(h, hipMemset(h->doneflag, 0, m));
  HIP_TRY(h, hipMalloc((void**)&h->counters, sizeof(unsigned long long) * 160));
  HIP_TRY(h, hipMemset(h->counters, 0, sizeof(unsigned long long) * 160));
  return TGR_OK;
}

// parameters of the row kernels in both precisions (tgr_row.h).  The row kernels' stopping rule is sqrt(g^T M^-1 g) < tol
// [sqrt(kg) m/s^2], not mujoco's scaled plain norm: solver_tolerance of the config means that for them.
template <typename T> static void params_to_row(const Params<T>& P, const tgr_config& c, row::RParams<T>& R) {
  R.h = P.h; R.grav = P.grav; R.mass = P.mass; R.Ixx = P.Ixx; R.Izz = P.Izz;
  R.cap_r = P.cap_r; R.bar_half = P.bar_half; R.rcyl_r = P.rcyl_r; R.bcyl_r = P.bcyl_r; R.bcyl_lo = P.bcyl_lo; R.bcyl_hi = P.bcyl_hi;
  R.site_off = P.site_off; R.site_z = P.site_z;
  for (int t = 0; t < 9; t++) { R.ten_k[t] = P.ten_k[t]; R.ten_b[t] = P.ten_b[t]; R.ten_L0[t] = P.ten_L0[t]; }
  R.gain = P.gain; R.bias0 = P.bias0; R.bias1 = P.bias1; R.bias2 = P.bias2; R.force_lo = P.force_lo; R.force_hi = P.force_hi;
  R.ctrl_lo = P.ctrl_lo; R.ctrl_hi = P.ctrl_hi; R.K = P.K; R.B = P.B;
  R.imp0 = P.imp0; R.imp1 = P.imp1; R.imp_width = P.imp_width; R.imp_mid = P.imp_mid;
  R.f0 = P.f0; R.f1sq = P.f1sq; R.f3sq = P.f3sq; R.f4sq = P.f4sq; R.impratio = P.impratio; R.invweight = P.invweight;
  R.ls_tol = P.ls_tol; R.cf_lo = P.cf_lo; R.cf_hi = P.cf_hi;
  R.skip_clamped = P.skip_clamped; R.act_itau = P.act_itau;
  R.warm_start = c.solver_warm_start >= 0 ? 1 : 0;  // row kernels: every solve starts from last sub-step's CONSTRAINT acceleration unless -1
  R.hf_rx = P.hf_rx; R.hf_ry = P.hf_ry; R.hf_ez = P.hf_ez; R.hf_x0 = P.hf_x0; R.hf_y0 = P.hf_y0; R.hf_z0 = P.hf_z0; R.hf_dx = P.hf_dx; R.hf_dy = P.hf_dy;
  R.hf_nrow = P.hf_nrow; R.hf_ncol = P.hf_ncol; R.hf = P.hf;
  const bool f32 = c.precision != TGR_PRECISION_F64;
  // (uneven-ground model: 5 cm caps on tilted facets put more contact torque about the bars' own axes -- a tighter default keeps qvel inside 1e-4)
  // fp32 handles: g^T M^-1 g < tol^2 + rtol^2 d^T M d with (tol, rtol) = (5e-4, 2e-5) on the flat floor, (7e-5, 3.5e-6) on the heightfield --
  // the relative term is what fp32 resolves of a gradient that is the difference of forces of that size (DESIGN.md 4.1,
  // profiles/round3_tolerance_study.log); iteration cap 16, and a row whose line search finds no descent at working precision leaves at
  // once (counted with the unconverged ones)
  // (heightfield, round 4: (1e-4, 5e-6) -> (7e-5, 3.5e-6): at 4096 envs x 16 control periods the quasi-static tr_env robots went from 99.889 % to
  // 99.924 % of env-steps inside 1e-4 on qvel for +4 % iterations -- profiles/round4_uneven_tolerance_4096x16.log)
  R.tol = T(c.solver_tolerance > 0 ? c.solver_tolerance : (f32 ? (c.model_kind ? 7e-5 : 5e-4) : 1e-8));
  R.rtol = T(c.solver_rel_tolerance > 0 ? c.solver_rel_tolerance : (c.solver_rel_tolerance < 0 ? 0.0 : (f32 ? (c.model_kind ? 3.5e-6 : 2e-5) : 0.0)));
  R.max_iter = c.solver_iterations > 0 ? c.solver_iterations : (f32 ? 16 : 40);
  R.ls_iter = f32 ? 6 : 50;
  R.derive();
}
template <typename T> static void fill_row_args(const tgr_handle* h, const KArgs<T>& k, RowArgs<T, double>& ra) {
  params_to_row<T>(k.P, h->cfg, ra.P);
  KArgs<double> kd;
  fill_args<double>(h, kd);
  params_to_row<double>(kd.P, h->cfg, ra.PK);
}
// which kernels step this handle: the row kernels unless the block kernels (the cross-check layout) are asked for
static bool use_row_kernels(const tgr_handle* h) { return h->cfg.kernel_layout != TGR_KERNEL_BLOCK; }

extern "C" int tgr_create(const tgr_config* cfg, int n_envs, int device, tgr_handle** out) {
  if (!cfg || !out || n_envs <= 0) { g_err = "tgr_create: bad arguments"; return TGR_ERR_INVALID; }
  if (cfg->desired_action < TGR_ACT_STRAIGHT || cfg->desired_action > TGR_ACT_VEL_TRACK ||
      (cfg->env_kind == TGR_ENV_TENSEGRITY && cfg->desired_action > TGR_ACT_TURN)) {
    g_err = "tgr_create: desired_action not supported by this env class (tensegrity_env: straight|turn; tr_env: + aiming|tracking|vel_track)";
    return TGR_ERR_UNSUPPORTED;
  }
  if (cfg->model_kind < 0 || cfg->model_kind > 1) { g_err = "tgr_create: model_kind must be 0 (flat) or 1 (uneven ground)"; return TGR_ERR_UNSUPPORTED; }
  if (cfg->solimp[4] != 2.0 || cfg->solref[0] >= 0) { g_err = "tgr_create: only direct solref and power-2 solimp"; return TGR_ERR_UNSUPPORTED; }
  if (cfg->use_cap_size_noise && cfg->env_kind != TGR_ENV_TR) {  // (the reference's tensegrity_env has no such kwarg: TypeError there)
    g_err = "tgr_create: use_cap_size_noise is a tr_env option"; return TGR_ERR_UNSUPPORTED;
  }
  if (cfg->use_cap_size_noise && cfg->env_kind == TGR_ENV_TR && cfg->kernel_layout == TGR_KERNEL_BLOCK) {
    g_err = "tgr_create: use_cap_size_noise needs the row kernels"; return TGR_ERR_UNSUPPORTED;
  }
  if (cfg->n_donors > 8 * n_envs) { g_err = "tgr_create: n_donors must be <= 8 * n_envs"; return TGR_ERR_INVALID; }
  if (cfg->reward_delay_steps >= RING - 1 || cfg->reward_delay_steps < 0) { g_err = "tgr_create: reward_delay_steps too large"; return TGR_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "tgr_create: no HIP device (the HIP path is required; there is no CPU fallback)"; return TGR_ERR_HIP; }
  if (device < 0 || device >= ndev) { g_err = "tgr_create: bad device index"; return TGR_ERR_INVALID; }
  tgr_handle* h = new (std::nothrow) tgr_handle();
  if (!h) return TGR_ERR_INVALID;
  h->cfg = *cfg; h->n = n_envs; h->device = device; h->obs_dim = obs_dim_of(*cfg);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
  }
  if (const char* v = getenv("TGR_PACE_FROM")) h->pace_from = atoi(v);
  if (const char* v = getenv("TGR_DONOR_STAGGER_PCT")) h->stagger_pct = atoi(v);
  int donors = cfg->n_donors < 0 ? 2 * n_envs : cfg->n_donors;
  if (!cfg->auto_reset) donors = 0;
  // donor stream: 0 = auto (on whenever the handle has donor lanes, is stepped by the row kernels and n_envs is a multiple of the four rows
  // of a wavefront), 1 = off (one launch steps live and donor rows, paced donor rows), 2 = required
  {
    const bool can = donors > 0 && use_row_kernels(h) && n_envs % 4 == 0;
    if (cfg->donor_stream == 2 && !can) { g_err = "tgr_create: donor_stream needs donor lanes, the row kernels and n_envs % 4 == 0"; delete h; return TGR_ERR_INVALID; }
    h->async = cfg->donor_stream != 1 && can;
    if (const char* v = getenv("TGR_DONOR_STREAM")) { if (atoi(v) == 0) h->async = false; }  // (A/B measurements)
  }
  // reset streams: 0 = auto (classes of 512 envs when the donor stream is on and the sizes divide), < 0 = off, B > 0 = classes of B envs
  {
    int B = cfg->reset_stream_envs;
    auto fits = [&](int b) { return donors > 0 && use_row_kernels(h) && n_envs % b == 0 && cfg->first_env_id % (uint64_t)b == 0 && donors % (n_envs / b) == 0; };
    if (B == 0) B = (h->async && fits(512)) ? 512 : -1;
    if (B > 0 && donors > 0) {
      if (!fits(B)) {
        g_err = "tgr_create: reset_stream_envs needs the row kernels, n_envs and first_env_id multiples of it, and n_donors a multiple of the number of classes";
        delete h;
        return TGR_ERR_INVALID;
      }
      h->stream_envs = B;
      h->n_classes = n_envs / B;
      h->stream_donors = donors / h->n_classes;
      h->stream_table = 1;
      while (h->stream_table < 2 * h->stream_donors) h->stream_table <<= 1;
    }
  }
  h->m = n_envs + donors;
  *out = nullptr;
  const int rc = alloc_handle(h, n_envs, donors);
  if (rc != TGR_OK) {  // nothing half-built leaves this function: the message moves to the thread's slot, the handle is freed
    g_err = "tgr_create: " + h->err;
    free_handle(h);
    return rc;
  }
  *out = h;
  return TGR_OK;
}

#ifdef TGR_OTHERS_HIST
extern "C" int tgr_debug_others_hist(unsigned long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(tgr::row::g_others_hist), sizeof(unsigned long long) * 64);
  return 0;
}
#endif
#ifdef TGR_WAVE_TIMES
extern "C" int tgr_debug_wave_times(unsigned long long* out, int n_waves) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(tgr::g_wave_times), sizeof(unsigned long long) * 4 * n_waves);
  return 0;
}
#endif
extern "C" int tgr_set_hfield(tgr_handle* h, const float* heights, int nrow, int ncol) {
  if (!h) return TGR_ERR_INVALID;
  if (!heights || nrow < 2 || ncol < 2) { h->err = "tgr_set_hfield: need a [nrow >= 2][ncol >= 2] array"; return TGR_ERR_INVALID; }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  if (h->hf) { (void)hipFree(h->hf); h->hf = nullptr; }
  HIP_TRY(h, hipMalloc((void**)&h->hf, sizeof(float) * (size_t)nrow * ncol));
  HIP_TRY(h, hipMemcpy(h->hf, heights, sizeof(float) * (size_t)nrow * ncol, hipMemcpyHostToDevice));
  h->hf_nrow = nrow; h->hf_ncol = ncol;
  return TGR_OK;
}

extern "C" int tgr_destroy(tgr_handle* h) {
  if (!h) return TGR_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_handle(h);
  return TGR_OK;
}
extern "C" int tgr_num_envs(const tgr_handle* h) { return h ? h->n : 0; }
extern "C" int tgr_obs_dim(const tgr_handle* h) { return h ? h->obs_dim : 0; }
extern "C" int tgr_precision(const tgr_handle* h) { return h ? h->cfg.precision : 0; }
extern "C" const char* tgr_last_error(const tgr_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }
extern "C" int tgr_synchronize(tgr_handle* h) {
  if (!h) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  return TGR_OK;
}
extern "C" uint64_t tgr_lane_key(const tgr_handle* h, int lane) {
  if (!h) return 0;
  if (lane < h->n) return h->cfg.first_env_id + (uint64_t)lane;
  if (h->n_classes > 0)  // reset streams: the key of the donor lane's class (tgr_sim.hip lane_key)
    return (1ull << 62) + (1ull << 40) + h->cfg.first_env_id / (uint64_t)h->stream_envs + (uint64_t)((lane - h->n) / h->stream_donors);
  return (1ull << 62) + h->cfg.first_env_id * 8ull + (uint64_t)(lane - h->n);
}
extern "C" int tgr_set_seed(tgr_handle* h, uint64_t seed) {
  if (!h) return TGR_ERR_INVALID;
  h->cfg.seed = seed;
  // Env.reset(seed=s) twice must give the same episode twice: every lane's episode counter starts over
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemset(h->ist + (size_t)I_EPISODE * h->m, 0, sizeof(int) * (size_t)h->m));
  if (h->n_classes > 0) HIP_TRY(h, hipMemset(h->cls, 0, sizeof(int) * 2 * h->n_classes));  // (the full reset that follows deals the streams out afresh)
  // the donor lanes still hold episodes settled under the old key (and, with the donor stream, stay marked ready): a full tgr_reset must
  // follow before the handle is stepped again -- tgr_step refuses until then rather than hand out old-seed episodes silently
  if (h->m > h->n) h->seed_dirty = true;
  return TGR_OK;
}
extern "C" int tgr_bind_real_observation(tgr_handle* h, void* real_obs_dev) {
  if (!h) return TGR_ERR_INVALID;
  h->obs_clean = real_obs_dev;
  return TGR_OK;
}
static_assert(TGR_TRACE_WORDS == row::TRACE_WORDS, "sub-step trace layout");
extern "C" int tgr_bind_substep_trace(tgr_handle* h, uint32_t* trace_dev) {
  if (!h) return TGR_ERR_INVALID;
  h->trace = trace_dev;
  return TGR_OK;
}
extern "C" int tgr_time_kernels_every(tgr_handle* h, int n_slots, int every);
extern "C" int tgr_time_kernels(tgr_handle* h, int n_slots) { return tgr_time_kernels_every(h, n_slots, 1); }
extern "C" int tgr_time_kernels_every(tgr_handle* h, int n_slots, int every) {
  if (!h || n_slots < 0 || every < 1) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  for (hipEvent_t e : h->tm_start) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->tm_stop) (void)hipEventDestroy(e);
  h->tm_start.assign(n_slots, nullptr); h->tm_stop.assign(n_slots, nullptr);
  h->tm_n = 0; h->tm_count = 0; h->tm_step = 0; h->tm_every = every;
  for (int i = 0; i < n_slots; i++) { HIP_TRY(h, hipEventCreate(&h->tm_start[i])); HIP_TRY(h, hipEventCreate(&h->tm_stop[i])); }
  h->tm_n = n_slots;
  return TGR_OK;
}
extern "C" int tgr_kernel_times_ms(tgr_handle* h, float* out_ms, int n) {
  if (!h || !out_ms || n < 0) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  const long long have = h->tm_count < h->tm_n ? h->tm_count : h->tm_n;
  const int cnt = (int)(have < n ? have : n);
  for (int i = 0; i < cnt; i++) {
    const int slot = (int)((h->tm_count - cnt + i) % h->tm_n);
    HIP_TRY(h, hipEventElapsedTime(&out_ms[i], h->tm_start[slot], h->tm_stop[slot]));
  }
  return cnt;
}
extern "C" int tgr_stats(tgr_handle* h, uint64_t out[8]) {
  if (!h || !out) return TGR_ERR_INVALID;
  unsigned long long tmp[8];
#ifdef TGR_ROW_STAMPS
  { unsigned long long t2[160]; (void)hipDeviceSynchronize(); (void)hipMemcpy(t2, h->counters, sizeof(t2), hipMemcpyDeviceToHost);
    const char* nm[16] = {"kinematics", "tendons", "contacts", "newton set-up", "residual+cone", "wrench+sums+gradient", "hessian", "direction", "ls set-up", "ls evals",
                          "iterate update", "loop exit", "read-outs", "implicitfast", "advance", "between"};
    double tot = 0; for (int i = 0; i < 16; i++) tot += (double)t2[8 + i];
    printf("row stamps: %llu live wavefront-steps, %.0f cycles each\n", t2[8 + 23], tot / (double)(t2[8 + 23] ? t2[8 + 23] : 1));
    for (int i = 0; i < 16; i++) printf("  %-22s %6.2f %%  %9.0f cycles per wavefront-step\n", nm[i], 100.0 * (double)t2[8 + i] / tot, (double)t2[8 + i] / (double)(t2[8 + 23] ? t2[8 + 23] : 1)); }
#endif
#ifdef TGR_STAMPS
  { unsigned long long t2[160]; (void)hipDeviceSynchronize(); (void)hipMemcpy(t2, h->counters, sizeof(t2), hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; w++) { printf("wave %d Mcycles:", w); for (int i = 0; i < 32; i++) printf(" %6llu", t2[8 + w * 32 + i] / 1000000ull); printf("\n"); } }
#endif
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(tmp, h->counters, sizeof(tmp), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; i++) out[i] = tmp[i];
  return TGR_OK;
}

// settle_kernel_row over (at most) cnt list entries
template <typename T> static void launch_settle_row(const tgr_handle* h, int cnt, hipStream_t s, const KArgs<T>& k, const RowArgs<T, double>& ra) {
  if (h->cfg.model_kind) {
    constexpr int NE = row_envs<T, 1>();
    hipLaunchKernelGGL((settle_kernel_row<T, double, 1>), dim3((cnt + NE - 1) / NE), dim3(16 * NE), 0, s, RowKArgs<T, double>{k, ra});
  } else {
    constexpr int NE = row_envs<T, 0>();
    hipLaunchKernelGGL((settle_kernel_row<T, double, 0>), dim3((cnt + NE - 1) / NE), dim3(16 * NE), 0, s, RowKArgs<T, double>{k, ra});
  }
}

template <typename T> static int reset_impl(tgr_handle* h, const uint8_t* mask, void* obs, hipStream_t s) {
  if (h->cfg.model_kind == 1 && !h->hf) { h->err = "tgr_reset: the uneven-ground model needs tgr_set_hfield first"; return TGR_ERR_INVALID; }
  std::vector<int> idx;
  idx.reserve(h->m);
  for (int i = 0; i < h->n; i++)
    if (!mask || mask[i]) idx.push_back(i);
  // a full reset also settles every donor, so the pool starts out ready
  if (!mask)
    for (int i = h->n; i < h->m; i++) idx.push_back(i);
  int cnt = (int)idx.size();
  if (cnt == 0) return TGR_OK;
  if (!mask) h->seed_dirty = false;
  if (h->async && !mask) {
    HIP_TRY(h, hipStreamSynchronize(h->sb));  // (a full reset rewrites the donor lanes: their stream must be idle)
    h->launch = 1;  // launch indices start over: the settle kernel below rewrites dready / consumed / dcmd / dapplied of every donor lane
  }
  if (h->n_classes > 0) {
    // reset streams: explicit resets of live envs run under the env's own key; a full reset also deals each class's donors the next
    // stream_donors episodes of its stream (donor j of the class: episode next_start + j), ready to be handed out in that order
    HIP_TRY(h, hipStreamSynchronize(s));
    HIP_TRY(h, hipMemset(h->list_ep, 0xff, sizeof(int) * (size_t)h->m));
    if (!mask) {
      const int C_ = h->n_classes, Dc = h->stream_donors, TB = h->stream_table;
      std::vector<int> cls(2 * C_), ep((size_t)C_ * Dc), hold((size_t)C_ * TB, 0);
      HIP_TRY(h, hipMemcpy(cls.data(), h->cls, sizeof(int) * 2 * C_, hipMemcpyDeviceToHost));
      for (int c = 0; c < C_; c++) {
        const int base = cls[2 * c];
        for (int j = 0; j < Dc; j++) { ep[(size_t)c * Dc + j] = base + j; hold[(size_t)c * TB + ((base + j) & (TB - 1))] = h->n + c * Dc + j; }
        cls[2 * c] = base + Dc; cls[2 * c + 1] = base;
      }
      HIP_TRY(h, hipMemcpy(h->cls, cls.data(), sizeof(int) * 2 * C_, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->holder, hold.data(), sizeof(int) * hold.size(), hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(h->ist + (size_t)I_EPISODE * h->m + h->n, ep.data(), sizeof(int) * ep.size(), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(h, hipMemcpyAsync(h->list, idx.data(), sizeof(int) * cnt, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(h->list_count, &cnt, sizeof(int), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipStreamSynchronize(s));  // idx / cnt are host temporaries
  KArgs<T> k;
  fill_args<T>(h, k);
  k.obs_reset = (T*)obs;
  constexpr int LW = LaneW<T>::value;
  if (use_row_kernels(h)) {
    RowArgs<T, double> ra;
    fill_row_args<T>(h, k, ra);
    launch_settle_row<T>(h, cnt, s, k, ra);
  } else {
#ifndef TGR_FAST_BUILD
    int blocks = (cnt + LW - 1) / LW;
    if (h->cfg.model_kind) hipLaunchKernelGGL((settle_kernel<T, LW, 1>), dim3(blocks), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((settle_kernel<T, LW, 0>), dim3(blocks), dim3(256), 0, s, k);
#else
    h->err = "development build (TGR_FAST_BUILD): block kernels not compiled"; return TGR_ERR_UNSUPPORTED;
#endif
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemsetAsync(h->list_count, 0, sizeof(int), s));
  if (h->async && !mask) {
    // the next donor launch waits for this settle kernel; the next match kernel finds "the previous donor launch" complete
    HIP_TRY(h, hipEventRecord(h->ev_match, s));
    for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventRecord(h->ev_donor[i], h->sb));
  }
  return TGR_OK;
}

extern "C" int tgr_reset(tgr_handle* h, const uint8_t* mask_host, void* obs_dev, void* hip_stream) {
  if (!h) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)hip_stream;
#ifdef TGR_FAST_BUILD  // development build: fp32 row kernels only (compiles in a fraction of the time)
  if (h->cfg.precision == TGR_PRECISION_F64) { h->err = "development build (TGR_FAST_BUILD): fp32 only"; return TGR_ERR_UNSUPPORTED; }
  return reset_impl<float>(h, mask_host, obs_dev, s);
#else
  return h->cfg.precision == TGR_PRECISION_F64 ? reset_impl<double>(h, mask_host, obs_dev, s) : reset_impl<float>(h, mask_host, obs_dev, s);
#endif
}

template <typename T>
static int step_impl(tgr_handle* h, const float* actions, T* obs, T* reward, uint8_t* terminated, uint8_t* truncated,
                     T* terminal_obs, T* info, hipStream_t s) {
  if (!actions || !obs || !reward || !terminated || !truncated) { h->err = "tgr_step: null buffer"; return TGR_ERR_INVALID; }
  if (h->cfg.model_kind == 1 && !h->hf) { h->err = "tgr_step: the uneven-ground model needs tgr_set_hfield first"; return TGR_ERR_INVALID; }
  if (h->seed_dirty) { h->err = "tgr_step: tgr_set_seed must be followed by a full tgr_reset (mask = NULL): the donor lanes hold episodes settled under the old seed"; return TGR_ERR_INVALID; }
  KArgs<T> k;
  fill_args<T>(h, k);
  k.actions = actions; k.obs = obs; k.reward = reward; k.terminated = terminated; k.truncated = truncated;
  k.terminal_obs = terminal_obs; k.info = info;
  k.obs_reset = obs;
  k.obs_clean = (T*)h->obs_clean;
  k.dbg = h->trace;
  constexpr int LW = LaneW<T>::value;
  const bool rowk = use_row_kernels(h);
  RowArgs<T, double> ra;
  if (h->async) {
    // (32-bit launch indices: ~4 weeks of continuous stepping at 900 steps/s; a full tgr_reset starts them over)
    if (h->launch >= 0x7ffffff0) { h->err = "tgr_step: launch counter exhausted -- call tgr_reset (all envs) to start it over"; return TGR_ERR_INVALID; }
    h->launch += 1;  // this step's index: its donor launch and its match kernel carry it
  }
  k.launch_idx = h->launch;
  if (rowk) {
    fill_row_args<T>(h, k, ra);
    auto launch_rows = [&](const KArgs<T>& kk, int lanes, hipStream_t st) {
      if (h->cfg.model_kind) {
        constexpr int NE = row_envs<T, 1>();
        hipLaunchKernelGGL((step_kernel_row<T, double, 1>), dim3((((lanes + NE - 1) / NE + 63) / 64) * 64), dim3(16 * NE), 0, st, RowKArgs<T, double>{kk, ra});
      } else {
        constexpr int NE = row_envs<T, 0>();
        hipLaunchKernelGGL((step_kernel_row<T, double, 0>), dim3((((lanes + NE - 1) / NE + 63) / 64) * 64), dim3(16 * NE), 0, st, RowKArgs<T, double>{kk, ra});
      }
    };
    if (h->async) {
      // live rows on the caller's stream; the donor rows advance one control period on their own low-priority stream, concurrently --
      // after the match kernel of the previous step (whose commands this launch carries out)
      k.lane_base = 0; k.lane_limit = h->n;
      const int ti = (h->tm_n && (h->tm_step++ % h->tm_every) == 0) ? (int)(h->tm_count++ % h->tm_n) : -1;
      if (ti >= 0) HIP_TRY(h, hipEventRecord(h->tm_start[ti], s));
      launch_rows(k, h->n, s);
      if (ti >= 0) HIP_TRY(h, hipEventRecord(h->tm_stop[ti], s));
      HIP_TRY(h, hipGetLastError());
      KArgs<T> kd = k;
      kd.lane_base = h->n; kd.lane_limit = h->m; kd.dbg = nullptr;
      HIP_TRY(h, hipStreamWaitEvent(h->sb, h->ev_match, 0));
      launch_rows(kd, h->m - h->n, h->sb);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipEventRecord(h->ev_donor[h->launch & 1], h->sb));
      // this step's match kernel reads the donor lanes as the PREVIOUS donor launch left them
      HIP_TRY(h, hipStreamWaitEvent(s, h->ev_donor[(h->launch - 1) & 1], 0));
      k.lane_base = 0; k.lane_limit = h->m;
    } else {
      const int ti = (h->tm_n && (h->tm_step++ % h->tm_every) == 0) ? (int)(h->tm_count++ % h->tm_n) : -1;
      if (ti >= 0) HIP_TRY(h, hipEventRecord(h->tm_start[ti], s));
      launch_rows(k, h->m, s);
      if (ti >= 0) HIP_TRY(h, hipEventRecord(h->tm_stop[ti], s));
    }
  } else {
#ifndef TGR_FAST_BUILD
    int blocks = (h->m + LW - 1) / LW;
    if (h->cfg.model_kind) hipLaunchKernelGGL((step_kernel<T, LW, 1>), dim3(blocks), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((step_kernel<T, LW, 0>), dim3(blocks), dim3(256), 0, s, k);
#else
    h->err = "development build (TGR_FAST_BUILD): block kernels not compiled"; return TGR_ERR_UNSUPPORTED;
#endif
  }
  HIP_TRY(h, hipGetLastError());
  if (h->cfg.auto_reset) {
    if (h->n_classes > 0) {
      hipLaunchKernelGGL((match_kernel_stream<T>), dim3(h->n_classes), dim3(256), 0, s, k);  // (list_count was zeroed by the step kernel)
    } else {
      hipLaunchKernelGGL((match_kernel<T>), dim3(1), dim3(1024), 0, s, k);
    }
    HIP_TRY(h, hipGetLastError());
    // envs that found no ready donor are settled here and now (grid sized for the worst case, exits on an empty list)
    if (rowk) {
      launch_settle_row<T>(h, h->n, s, k, ra);
    } else {
#ifndef TGR_FAST_BUILD
      int sblocks = (h->n + LW - 1) / LW;
      if (h->cfg.model_kind) hipLaunchKernelGGL((settle_kernel<T, LW, 1>), dim3(sblocks), dim3(256), 0, s, k);
      else hipLaunchKernelGGL((settle_kernel<T, LW, 0>), dim3(sblocks), dim3(256), 0, s, k);
#endif
    }
    HIP_TRY(h, hipGetLastError());
    if (h->async) HIP_TRY(h, hipEventRecord(h->ev_match, s));
  }
  return TGR_OK;
}

extern "C" int tgr_step(tgr_handle* h, const float* actions, float* obs, float* reward, uint8_t* terminated,
                        uint8_t* truncated, float* terminal_obs, float* info, void* hip_stream) {
  if (!h) return TGR_ERR_INVALID;
  if (h->cfg.precision != TGR_PRECISION_F32) { h->err = "tgr_step: handle is fp64, use tgr_step_f64"; return TGR_ERR_INVALID; }
  HIP_TRY(h, hipSetDevice(h->device));
  return step_impl<float>(h, actions, obs, reward, terminated, truncated, terminal_obs, info, (hipStream_t)hip_stream);
}
extern "C" int tgr_step_f64(tgr_handle* h, const float* actions, double* obs, double* reward, uint8_t* terminated,
                            uint8_t* truncated, double* terminal_obs, double* info, void* hip_stream) {
  if (!h) return TGR_ERR_INVALID;
  if (h->cfg.precision != TGR_PRECISION_F64) { h->err = "tgr_step_f64: handle is fp32, use tgr_step"; return TGR_ERR_INVALID; }
  HIP_TRY(h, hipSetDevice(h->device));
#ifdef TGR_FAST_BUILD
  h->err = "development build (TGR_FAST_BUILD): fp32 only"; return TGR_ERR_UNSUPPORTED;
#else
  return step_impl<double>(h, actions, obs, reward, terminated, truncated, terminal_obs, info, (hipStream_t)hip_stream);
#endif
}

// ---- state exchange (host arrays, double)
template <typename T> static int get_state_impl(tgr_handle* h, double* qpos, double* qvel, double* ctrl, double* warm, double* aux) {
  size_t n = h->m, nlive = h->n;
  std::vector<T> st(NF * n);
  std::vector<int> ist(NI * n);
  std::vector<T> ring;
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(st.data(), h->st, sizeof(T) * NF * n, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(ist.data(), h->ist, sizeof(int) * NI * n, hipMemcpyDeviceToHost));
  if (aux) {
    ring.resize(RING * n);
    HIP_TRY(h, hipMemcpy(ring.data(), h->ring, sizeof(T) * RING * n, hipMemcpyDeviceToHost));
  }
  std::vector<double> hp;
  if ((qpos || aux) && h->hp) {
    hp.resize(HP_ROWS * n);
    HIP_TRY(h, hipMemcpy(hp.data(), h->hp, sizeof(double) * HP_ROWS * n, hipMemcpyDeviceToHost));
  }
  // an absolute position of the env layer: from the fp64 side array where the handle has one
  auto absq = [&](int hp_row, int st_field, size_t e) { return h->hp ? hp[(size_t)hp_row * n + e] : (double)st[(size_t)st_field * n + e]; };
  for (size_t e = 0; e < nlive; e++) {
    if (qpos) for (int i = 0; i < 21; i++) qpos[e * 21 + i] = h->hp ? hp[i * n + e] : (double)st[(F_QPOS + i) * n + e];
    if (qvel) for (int i = 0; i < 18; i++) qvel[e * 18 + i] = (double)st[(F_QVEL + i) * n + e];
    if (ctrl) for (int i = 0; i < 6; i++) ctrl[e * 6 + i] = (double)st[(F_CTRL + i) * n + e];
    if (warm) for (int i = 0; i < 18; i++) warm[e * 18 + i] = (double)st[(F_WARM + i) * n + e];
    if (aux) {
      double* a = aux + e * TGR_AUX_K;
      a[TGR_AUX_XY_LAG_X] = absq(HP_XYLAG, F_XYLAG, e); a[TGR_AUX_XY_LAG_Y] = absq(HP_XYLAG + 1, F_XYLAG + 1, e);
      a[TGR_AUX_PSI_LAG] = (double)st[F_PSILAG * n + e]; a[TGR_AUX_RESET_PSI] = (double)st[F_RESETPSI * n + e];
      a[TGR_AUX_EP_RETURN] = (double)st[F_EPRET * n + e];
      a[TGR_AUX_STEP_NUM] = ist[I_STEPNUM * n + e]; a[TGR_AUX_ELAPSED] = ist[I_ELAPSED * n + e];
      a[TGR_AUX_EPISODE] = ist[I_SRCEP * n + e]; a[TGR_AUX_RING_LEN] = ist[I_RINGLEN * n + e]; a[TGR_AUX_SRC] = ist[I_SRC * n + e];
      a[TGR_AUX_WAYPT_X] = absq(HP_WAYPT, F_WAYPT, e); a[TGR_AUX_WAYPT_Y] = absq(HP_WAYPT + 1, F_WAYPT + 1, e);
      a[TGR_AUX_ORI_X] = absq(HP_ORI, F_ORI, e); a[TGR_AUX_ORI_Y] = absq(HP_ORI + 1, F_ORI + 1, e);
      a[TGR_AUX_VCMD_X] = (double)st[(F_VCMD + 0) * n + e]; a[TGR_AUX_VCMD_Y] = (double)st[(F_VCMD + 1) * n + e];
      a[TGR_AUX_VCMD_W] = (double)st[(F_VCMD + 2) * n + e];
      for (int i = 0; i < 6; i++) a[TGR_AUX_ACT + i] = (double)st[(F_ACT + i) * n + e];
      a[TGR_AUX_CAP_RADIUS] = h->cfg.cap_radius + (double)st[F_CAPD * n + e];
      const int len = ist[I_RINGLEN * n + e], head = ist[I_RINGHEAD * n + e];
      for (int i = 0; i < RING; i++) a[TGR_AUX_RING + i] = i < len ? (double)ring[(size_t)((head + i) % RING) * n + e] : 0.0;
    }
  }
  return TGR_OK;
}
template <typename T>
static int set_state_impl(tgr_handle* h, const double* qpos, const double* qvel, const double* ctrl, const double* warm, const double* aux) {
  size_t n = h->m, nlive = h->n;
  std::vector<T> st(NF * n);
  std::vector<int> ist(NI * n);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(st.data(), h->st, sizeof(T) * NF * n, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(ist.data(), h->ist, sizeof(int) * NI * n, hipMemcpyDeviceToHost));
  for (size_t e = 0; e < nlive; e++) {
    if (qpos) for (int i = 0; i < 21; i++) st[(F_QPOS + i) * n + e] = (T)qpos[e * 21 + i];
    if (qvel) for (int i = 0; i < 18; i++) st[(F_QVEL + i) * n + e] = (T)qvel[e * 18 + i];
    if (ctrl) for (int i = 0; i < 6; i++) st[(F_CTRL + i) * n + e] = (T)ctrl[e * 6 + i];
    if (warm) for (int i = 0; i < 18; i++) st[(F_WARM + i) * n + e] = (T)warm[e * 18 + i];
  }
  HIP_TRY(h, hipMemcpy(h->st, st.data(), sizeof(T) * NF * n, hipMemcpyHostToDevice));
  if (qpos && h->hp) {
    std::vector<double> hp(21 * n);
    HIP_TRY(h, hipMemcpy(hp.data(), h->hp, sizeof(double) * 21 * n, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < nlive; e++)
      for (int i = 0; i < 21; i++) hp[i * n + e] = qpos[e * 21 + i];
    HIP_TRY(h, hipMemcpy(h->hp, hp.data(), sizeof(double) * 21 * n, hipMemcpyHostToDevice));  // (rows HP_QPOS ..: the first 21 rows)
  }
  if (qpos) {
    KArgs<T> k;
    fill_args<T>(h, k);
    hipLaunchKernelGGL((refresh_lag_kernel<T>), dim3((h->n + 63) / 64), dim3(64), 0, 0, k);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipDeviceSynchronize());
  }
  if (aux) {
    HIP_TRY(h, hipMemcpy(st.data(), h->st, sizeof(T) * NF * n, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < nlive; e++) {
      const double* a = aux + e * TGR_AUX_K;
      st[(F_XYLAG + 0) * n + e] = (T)a[TGR_AUX_XY_LAG_X]; st[(F_XYLAG + 1) * n + e] = (T)a[TGR_AUX_XY_LAG_Y];
      st[F_PSILAG * n + e] = (T)a[TGR_AUX_PSI_LAG]; st[F_RESETPSI * n + e] = (T)a[TGR_AUX_RESET_PSI];
      st[F_EPRET * n + e] = (T)a[TGR_AUX_EP_RETURN];
      ist[I_STEPNUM * n + e] = (int)a[TGR_AUX_STEP_NUM]; ist[I_ELAPSED * n + e] = (int)a[TGR_AUX_ELAPSED];
      st[(F_WAYPT + 0) * n + e] = (T)a[TGR_AUX_WAYPT_X]; st[(F_WAYPT + 1) * n + e] = (T)a[TGR_AUX_WAYPT_Y];
      st[(F_ORI + 0) * n + e] = (T)a[TGR_AUX_ORI_X]; st[(F_ORI + 1) * n + e] = (T)a[TGR_AUX_ORI_Y];
      st[(F_VCMD + 0) * n + e] = (T)a[TGR_AUX_VCMD_X]; st[(F_VCMD + 1) * n + e] = (T)a[TGR_AUX_VCMD_Y];
      st[(F_VCMD + 2) * n + e] = (T)a[TGR_AUX_VCMD_W];
      for (int i = 0; i < 6; i++) st[(F_ACT + i) * n + e] = (T)a[TGR_AUX_ACT + i];
      if (a[TGR_AUX_CAP_RADIUS] > 0.0) st[F_CAPD * n + e] = (T)(a[TGR_AUX_CAP_RADIUS] - h->cfg.cap_radius);  // (0: keep the lane's radius)
    }
    // the heading deque: replaced by the caller's (oldest first)
    std::vector<T> ring(RING * n);
    HIP_TRY(h, hipMemcpy(ring.data(), h->ring, sizeof(T) * RING * n, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < nlive; e++) {
      const double* a = aux + e * TGR_AUX_K;
      int len = (int)a[TGR_AUX_RING_LEN];
      if (len < 0 || len >= RING) { h->err = "tgr_set_state: aux ring length out of range"; return TGR_ERR_INVALID; }
      ist[I_RINGLEN * n + e] = len; ist[I_RINGHEAD * n + e] = 0;
      for (int i = 0; i < len; i++) ring[(size_t)i * n + e] = (T)a[TGR_AUX_RING + i];
    }
    HIP_TRY(h, hipMemcpy(h->ring, ring.data(), sizeof(T) * RING * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->st, st.data(), sizeof(T) * NF * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->ist, ist.data(), sizeof(int) * NI * n, hipMemcpyHostToDevice));
    if (h->hp) {  // the env layer's absolute positions at full precision (rows HP_XYLAG .. HP_ROWS)
      const size_t nr = HP_ROWS - HP_XYLAG;
      std::vector<double> hx(nr * n);
      double* const dst = h->hp + (size_t)HP_XYLAG * n;
      HIP_TRY(h, hipMemcpy(hx.data(), dst, sizeof(double) * nr * n, hipMemcpyDeviceToHost));
      for (size_t e = 0; e < nlive; e++) {
        const double* a = aux + e * TGR_AUX_K;
        hx[0 * n + e] = a[TGR_AUX_XY_LAG_X]; hx[1 * n + e] = a[TGR_AUX_XY_LAG_Y];
        hx[2 * n + e] = a[TGR_AUX_WAYPT_X]; hx[3 * n + e] = a[TGR_AUX_WAYPT_Y];
        hx[4 * n + e] = a[TGR_AUX_ORI_X]; hx[5 * n + e] = a[TGR_AUX_ORI_Y];
      }
      HIP_TRY(h, hipMemcpy(dst, hx.data(), sizeof(double) * nr * n, hipMemcpyHostToDevice));
    }
  }
  return TGR_OK;
}

extern "C" int tgr_get_state(tgr_handle* h, double* qpos, double* qvel, double* ctrl, double* warm, double* aux) {
  if (!h) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  return h->cfg.precision == TGR_PRECISION_F64 ? get_state_impl<double>(h, qpos, qvel, ctrl, warm, aux)
                                                : get_state_impl<float>(h, qpos, qvel, ctrl, warm, aux);
}
extern "C" int tgr_set_state(tgr_handle* h, const double* qpos, const double* qvel, const double* ctrl, const double* warm,
                             const double* aux) {
  if (!h) return TGR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  return h->cfg.precision == TGR_PRECISION_F64 ? set_state_impl<double>(h, qpos, qvel, ctrl, warm, aux)
                                                : set_state_impl<float>(h, qpos, qvel, ctrl, warm, aux);
}
