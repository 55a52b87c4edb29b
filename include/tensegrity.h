/*
 * tensegrity.h -- C-ABI of the MI355X batched tensegrity simulator (libtensegrity.so).
 *
 * Drop-in boundary for the reference's hot path.  Each entry point names the reference
 * interface it replaces (paths relative to drsteinkauz/tensegrity-RL):
 *
 *   tgr_create      gym.make('tensegrity_env-v0' | 'tr_env-v0', **kwargs) -> env.__init__
 *                   (tensegrity_env/tensegrity_env/envs/tensegrity_env.py:160-243,
 *                    tr_env/tr_env/envs/tr_env.py:137-278, run.py:413-418) which loads the model
 *                   through gym MujocoEnv.__init__ -> mujoco.MjModel.from_xml_path / MjData.
 *   tgr_reset       Env.reset() -> MujocoEnv.reset -> mj_resetData + reset_model()
 *                   (tensegrity_env.py:433-512, tr_env.py:709-872).
 *   tgr_step[_f64]  Env.step(action) (tensegrity_env.py:291-410, tr_env.py:327-527) =
 *                   MujocoEnv.do_simulation -> mujoco.mj_step(nstep=20) + mj_rnePostConstraint,
 *                   then _get_obs / reward / termination; batched over N envs with SB3 VecEnv
 *                   auto-reset semantics (DummyVecEnv.step_wait: terminal_observation, reset obs).
 *   tgr_get_state / tgr_set_state
 *                   env.data.qpos/qvel/ctrl/qacc_warmstart reads and MujocoEnv.set_state
 *                   (tensegrity_env.py:448,484) -- used for "identical (state, action)" parity.
 *
 * Conventions: plain C types only.  All array arguments of tgr_reset/tgr_step are DEVICE
 * pointers owned by the caller (e.g. torch tensors), valid for the duration of the call's
 * stream work; get/set_state take HOST pointers and synchronise.  Every function returns 0 on
 * success and a negative code on failure; tgr_last_error() gives the message.  One caller
 * thread per handle.  Kernels are enqueued on the HIP stream passed in (NULL = default stream).
 */
#ifndef TENSEGRITY_H
#define TENSEGRITY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGR_OK 0
#define TGR_ERR_INVALID (-1)
#define TGR_ERR_HIP (-2)
#define TGR_ERR_UNSUPPORTED (-3)

#define TGR_ENV_TENSEGRITY 0 /* id 'tensegrity_env-v0' */
#define TGR_ENV_TR 1         /* id 'tr_env-v0' */

#define TGR_ACT_STRAIGHT 0
#define TGR_ACT_TURN 1
#define TGR_ACT_AIMING 2
#define TGR_ACT_TRACKING 3
#define TGR_ACT_VEL_TRACK 4

#define TGR_PRECISION_F32 0
#define TGR_PRECISION_F64 1

#define TGR_KERNEL_AUTO 0
#define TGR_KERNEL_BLOCK 1
#define TGR_KERNEL_ROW 2

#define TGR_NQ 21
#define TGR_NV 18
#define TGR_NU 6
#define TGR_INFO_K 26
/* info row layout (float/double [N][TGR_INFO_K]); the first ten are the reference's info dict
 * (tensegrity_env.py:388-400), then tr_env's waypt/oripoint (tr_env.py:509-510), Monitor-style episode stats,
 * tr_env's info["tendon_length"] (tr_env.py:506) and the bar-bar contact force the reference's test harness logs
 * (run.py:149-155: sum over contacts between two bars of |mj_contactForce[0:3]|) */
enum {
  TGR_INFO_REWARD_FORWARD = 0, TGR_INFO_REWARD_CTRL, TGR_INFO_REWARD_SURVIVE, TGR_INFO_X_POSITION,
  TGR_INFO_Y_POSITION, TGR_INFO_PSI, TGR_INFO_DISTANCE, TGR_INFO_X_VELOCITY, TGR_INFO_Y_VELOCITY,
  TGR_INFO_FORWARD_REWARD, TGR_INFO_WAYPT_X, TGR_INFO_WAYPT_Y, TGR_INFO_ORI_X, TGR_INFO_ORI_Y,
  TGR_INFO_EP_RETURN, TGR_INFO_EP_LENGTH,
  TGR_INFO_TENDON_LENGTH = 16, /* ..24 */
  TGR_INFO_BAR_CONTACT = 25
};

typedef struct tgr_config {
  /* --- env semantics: same names / defaults as the reference constructors */
  int32_t env_kind;       /* TGR_ENV_* */
  int32_t desired_action; /* TGR_ACT_* */
  double desired_direction;
  int32_t terminate_when_unhealthy;
  int32_t use_contact_forces;
  int32_t use_cap_velocity; /* tr_env */
  int32_t is_test;          /* tr_env */
  double ctrl_cost_weight, contact_cost_weight, healthy_reward;
  double contact_force_lo, contact_force_hi;
  double min_reset_heading, max_reset_heading;
  double tendon_reset_mean, tendon_reset_stdev, tendon_max_length, tendon_min_length;
  int32_t reward_delay_steps;
  int32_t frame_skip;        /* 20 */
  int32_t max_episode_steps; /* 5000, gym TimeLimit of the registration */
  /* --- model (3prism_jonathan_steady_side.xml) */
  double timestep, gravity, cap_radius;
  double solref[2], solimp[5], friction[3], impratio;
  double gain, bias[3], force_range[2], ctrl_range[2];
  double ten_stiffness[9], ten_damping[9], ten_springlength[9];
  int32_t actuator_deriv_skip_clamped;
  /* --- execution */
  int32_t precision;    /* TGR_PRECISION_* */
  int32_t auto_reset;   /* 1 = VecEnv semantics, 0 = plain Gym Env semantics */
  int32_t n_donors;     /* extra lanes that pre-settle fresh episodes inside the step launches; -1 = 2 * n_envs, 0 = none */
  uint64_t seed;        /* reset RNG: philox4x32-10 keyed (seed, first_env_id + i, episode) */
  uint64_t first_env_id;
  double solver_tolerance; /* <=0: default for the precision */
  int32_t solver_iterations;
  /* --- tr_env command modes (tr_env.py:164-172) */
  double waypt_range[2], waypt_angle_range[2];
  double ditch_reward_max, ditch_reward_stdev, waypt_reward_amplitude, waypt_reward_stdev, yaw_reward_weight;
  /* start of the Newton solve of a sub-step.  Block kernels: 0 (default) the unconstrained acceleration a0, 1 the previous solution
   * (mujoco's qacc_warmstart), 2 a0 + M^-1 (last sub-step's constraint wrench).  Row kernels: >= 0 (default) a0 + last sub-step's
   * constraint acceleration (measured on MI355X: 20 % faster on tensegrity_env-v0 than the cold start), -1 a0.  The optimum is unique:
   * the start changes the path, not the result (up to the solver tolerance) */
  int32_t solver_warm_start;
  /* tr_env observation noise (tr_env.py:142,161-162,549-646): step()/reset() return observation_with_noise.  The
   * reference draws from an unseeded generator; here the standard normals are philox draws keyed (seed; lane key, episode,
   * observation index within the episode), i.e. same distribution, reproducible */
  int32_t use_obs_noise;
  double obs_noise_tendon_stdev, obs_noise_cap_pos_stdev; /* 0.02, 0.05 */
  /* --- model variant (the reference selects it with xml_file): 0 = 3prism_jonathan_steady_side.xml (flat floor),
   * 1 = 3prism_jonathan_steady_side_uneven_ground.xml (tgr_uneven_config fills the fields that differ: caps 0.05, tendon site
   * layout, nine passive tendons, dyntype=filter actuators, initial pose, heightfield floor -- upload the elevations with
   * tgr_set_hfield before the first reset) */
  int32_t model_kind;
  double site_offset;     /* radial offset of the tendon sites (0.0675 / 0.05) */
  double act_tau;         /* dynprm[0] of the filter actuators */
  double hf_size[4];      /* heightfield: radius_x, radius_y, elevation_z, base_z (uneven xml:48) */
  double hf_pos[3];       /* position of the heightfield geom (uneven xml:56) */
  /* reset_noise_scale (tensegrity_env.py:169,436-445; tr_env.py:152,734-743): before the yaw transform qpos += U(-s, s)^21 and
   * qvel = s * N(0, 1)^18.  The reference draws from self.np_random; here philox keyed (seed; lane key, episode), counters 16..26
   * (uniforms) and 32..40 (Box-Muller normals) */
  double reset_noise_scale;
  /* which kernels step the handle.  TGR_KERNEL_ROW: one environment per 16-lane DPP row, 4 per wavefront, state parked in LDS slabs
   * (tgr_row.h); pose, kinematics, tendon geometry and axial torques in fp64 inside the kernel when the handle is fp32 ("mixed
   * precision"); solver_tolerance then bounds sqrt(g^T M^-1 g).  TGR_KERNEL_BLOCK: 64 environments x 4 wavefronts with per-lane LDS
   * columns (tgr_device.h), all arithmetic in the handle's precision, mujoco's stopping rule: kept as a cross-check.  AUTO = ROW */
  int32_t kernel_layout;
  /* row kernels: the solve of a sub-step also stops when sqrt(g^T M^-1 g) < solver_rel_tolerance * sqrt(d^T M d), d the constraint
   * acceleration: what the arithmetic resolves of the gradient when contact forces are large.  0: default for the precision (fp32 handles
   * 2e-5 on the flat floor and 5e-6 on the heightfield, fp64 handles off), < 0: off */
  double solver_rel_tolerance;
  /* auto-reset that does not depend on the sharding.  < 0: one donor pool per handle, a finished env takes the
   * lowest-indexed ready donor -- which episode that is depends on the other envs of the handle.  0 (default): B = 512 when the donor
   * stream is on and n_envs, first_env_id and n_donors divide accordingly, else one pool.  B > 0: envs are grouped in classes of B
   * consecutive GLOBAL env ids, each with its own n_donors / (n_envs / B) donor lanes and ONE stream of episodes keyed (class, 0), (class, 1),
   * ... handed out strictly in order: the k-th reset of a class gets episode k.  Any sharding whose shards are whole classes (n_envs and
   * first_env_id multiples of B) then gives every env the same trajectory, resets included.  (Without the donor stream it costs the paced
   * donor rows) */
  int32_t reset_stream_envs;
  /* donor stream: 0 (default) = auto, 1 = off, 2 = required.  On: the donor lanes (episodes being settled for the next resets) are stepped
   * by their own launches on a second, lower-priority HIP stream the handle owns, one per tgr_step, concurrently with the launch that steps
   * the live envs on the caller's stream; hand-over is ordered by HIP events and launch indices, so results do not depend on how the two
   * streams overlap.  Off: one launch steps live and donor rows together (round-2 behaviour, paced donor rows).  Auto = on
   * whenever the handle has donor lanes, is stepped by the row kernels and n_envs is a multiple of 4 */
  int32_t donor_stream;
  /* tr_env.py:143,163,685-706,712-713: use_cap_size_noise -- reset_model() starts with _reset_cap_size(cap_size_noise_range): one radius ~
   * U(range) for the six cap spheres of the episode (collision radius only: masses, inertias, tendon sites stay).  The reference's own code
   * path raises on the mujoco 2.3.7 bindings it pins (model.geom_names / geom_name2id are mujoco-py names); this builds its evident intent.
   * The uniform comes from philox counter 5 of (seed; lane key, episode).  tr_env + row kernels only */
  int32_t use_cap_size_noise;
  double cap_size_noise_range[2];
} tgr_config;

typedef struct tgr_handle tgr_handle;

/* fills *cfg with the reference defaults of the given env class / desired_action.
 * replaces: the constructor defaults of tensegrity_env.__init__ (tensegrity_env/tensegrity_env/envs/tensegrity_env.py:160-181) and
 * tr_env.__init__ (tr_env/tr_env/envs/tr_env.py:137-173), and the model constants of 3prism_jonathan_steady_side.xml:13-40,56-164 */
int tgr_default_config(tgr_config* cfg, int env_kind, int desired_action);
/* switches an already filled *cfg to the uneven-ground model.
 * replaces: xml_file=3prism_jonathan_steady_side_uneven_ground.xml (run.py:377 --env_xml, tr_env.py:139) -> MjModel.from_xml_path */
int tgr_uneven_config(tgr_config* cfg);
/* heightfield of model_kind 1: normalised elevations in [0,1], host pointer, [nrow][ncol], row 0 at y = -radius_y, column 0 at
 * x = -radius_x (mujoco's hfield_data layout [ext]); copied to the device */
int tgr_set_hfield(tgr_handle* h, const float* heights_host, int nrow, int ncol);
/* N envs on `device`.  replaces: gym.make(id, ...) (run.py:413-418) -> tensegrity_env/tr_env.__init__ -> MujocoEnv.__init__
 * (MjModel.from_xml_path + MjData), N times, and SB3's DummyVecEnv around them (run.py:41-85) */
int tgr_create(const tgr_config* cfg, int n_envs, int device, tgr_handle** out);
int tgr_destroy(tgr_handle* h);
int tgr_num_envs(const tgr_handle* h);
int tgr_obs_dim(const tgr_handle* h);
int tgr_precision(const tgr_handle* h);

/* reset envs with mask[i] != 0 (mask: HOST uint8[N] or NULL = all); obs: DEVICE [N][obs_dim]
 * of the handle's precision (float or double), rows of reset envs are written; may be NULL.
 * replaces: Env.reset() -> mj_resetData + reset_model() (tensegrity_env.py:433-512, tr_env.py:709-872: yaw-randomised pose, random
 * tendon targets, 50 control periods of settling, way-point sampling) */
int tgr_reset(tgr_handle* h, const uint8_t* mask_host, void* obs_dev, void* hip_stream);

/* one control period for all envs (fp32 handle).
 * replaces: Env.step() (tensegrity_env.py:291-410, tr_env.py:327-527) = do_simulation -> mj_step(nstep=20) + mj_rnePostConstraint,
 * _get_obs (tensegrity_env.py:412-430, tr_env.py:529-646), reward / termination (tensegrity_env.py:245-289,308-372), and the
 * auto-reset of SB3's DummyVecEnv.step_wait (terminal_obs = infos[i]["terminal_observation"], truncated = TimeLimit(5000)) */
int tgr_step(tgr_handle* h, const float* actions /*[N][6]*/, float* obs /*[N][D]*/, float* reward /*[N]*/,
             uint8_t* terminated /*[N]*/, uint8_t* truncated /*[N]*/, float* terminal_obs /*[N][D] or NULL*/,
             float* info /*[N][TGR_INFO_K] or NULL*/, void* hip_stream);
/* same for an fp64 handle (parity checking) */
int tgr_step_f64(tgr_handle* h, const float* actions, double* obs, double* reward, uint8_t* terminated,
                 uint8_t* truncated, double* terminal_obs, double* info, void* hip_stream);

/* HOST arrays, any may be NULL: qpos[N][21], qvel[N][18] (MuJoCo convention: world linear,
 * body-frame angular), ctrl[N][6], qacc_warmstart[N][18], aux[N][TGR_AUX_K].
 * replaces: reads of env.data.qpos/qvel/ctrl/qacc_warmstart and MujocoEnv.set_state (tensegrity_env.py:448,484; tr_env.py:744,763,800) */
#define TGR_AUX_K 56
enum {
  TGR_AUX_XY_LAG_X = 0, TGR_AUX_XY_LAG_Y, TGR_AUX_PSI_LAG, TGR_AUX_RESET_PSI, TGR_AUX_EP_RETURN,
  TGR_AUX_STEP_NUM, TGR_AUX_ELAPSED, TGR_AUX_EPISODE, TGR_AUX_RING_LEN, TGR_AUX_WAYPT_X, TGR_AUX_WAYPT_Y,
  TGR_AUX_ORI_X, TGR_AUX_ORI_Y, TGR_AUX_VCMD_X, TGR_AUX_VCMD_Y, TGR_AUX_VCMD_W,
  TGR_AUX_SRC, /* lane whose RNG key produced the current episode: the env itself or a donor lane; TGR_AUX_EPISODE is the episode index under that key */
  TGR_AUX_ACT = 17, /* ..22: actuator activations (model_kind 1) */
  /* 23..54: the heading deque self._heading_buffer (tensegrity_env.py:242-243,326-345; tr_env.py:278,376-378,430-432), oldest entry
   * first, TGR_AUX_RING_LEN valid entries.  tgr_set_state with aux replaces the deque (TGR_AUX_RING_LEN entries) */
  TGR_AUX_RING = 23,
  /* 55: the radius of the env's six cap spheres (tr_env.py:685-706 use_cap_size_noise: redrawn per episode; else the model's).  tgr_set_state:
   * a value > 0 replaces it, 0 keeps the lane's */
  TGR_AUX_CAP_RADIUS = 55
};
#define TGR_RING_CAPACITY 32
int tgr_get_state(tgr_handle* h, double* qpos, double* qvel, double* ctrl, double* qacc_warmstart, double* aux);
/* set_state: like MujocoEnv.set_state it also recomputes the derived (lagged) kinematics from qpos;
 * aux (if given) then overrides the stored aux fields */
int tgr_set_state(tgr_handle* h, const double* qpos, const double* qvel, const double* ctrl,
                  const double* qacc_warmstart, const double* aux);

/* counters since creation: [0] dropped bar-bar contacts (slot overflow), [1] episodes settled by the settle kernel
 * (tgr_reset, or a finished env that found no ready donor), [2] envs reset from a donor lane, [3] Newton iterations,
 * [4] line-search evaluations, [5] sub-steps -- [3..5] summed over wavefronts (wave-uniform loop counts),
 * [6] lane-sub-steps that left the Newton solver at its iteration cap, [7] lane-sub-steps run (live and settling lanes) */
int tgr_stats(tgr_handle* h, uint64_t out[8]);
/* re-key the reset RNG and restart every lane's episode counter, so that reset(seed=s) twice gives the same episode twice; takes
 * effect for subsequent resets.  On a handle with donor lanes a full tgr_reset (mask = NULL) MUST follow: the donor lanes hold episodes
 * settled under the old key, and tgr_step returns TGR_ERR_INVALID until they have been settled afresh -- it never hands them out silently.
 * (The Python VecEnv's seed() therefore only re-keys; the reset() SB3 issues next is the one full reset.)
 * replaces: Env.reset(seed=...) (gym API; the reference's own draws at tensegrity_env.py:460,486 / tr_env.py:730,775,802,831 are unseeded) */
int tgr_set_seed(tgr_handle* h, uint64_t seed);
/* tr_env with use_obs_noise: step() returns observation_with_noise and puts the noise-free one into info["real_observation"]
 * (tr_env.py:484-507, read by run.py:147).  Bind a DEVICE buffer [N][obs_dim] of the handle's precision (caller-owned, must stay
 * valid until unbound with NULL or the handle is destroyed); every tgr_step then also writes the noise-free observation of the step
 * (for an env that finished: of its last step, like terminal_obs) there.  Without observation noise it equals obs before auto-reset. */
int tgr_bind_real_observation(tgr_handle* h, void* real_obs_dev);
/* diagnostic (accuracy classification, tests/gpu_accuracy_report.py): bind a DEVICE buffer uint32 [N][TGR_TRACE_WORDS] (caller-owned, valid
 * until unbound with NULL); every tgr_step of a row-kernel handle then records, per env and sub-step s < frame_skip, word s =
 * contact slots instantiated in the forward pass of that sub-step (bit c < 6: cap c on the floor; 6 + 2p: the r-cylinders of bar pair p
 * side to side; every OTHER contact between two bars, whichever pair: the next bit of the pool 7, 9, 11, 12, 13, 14, 15 (heightfield:
 * 7, 9, 11, 15); 12 + b (heightfield): bar b's cylinder on the terrain; 16 + c: cap c on a second facet)
 * | (Newton iterations the env itself needed so far this step, mod 256) << 22 | (the solver left this sub-step at its iteration cap) << 30
 * | (its line search stalled at working precision) << 31;
 * and (heightfield model) word 24 + s = WHICH facets and grid edges those terrain contacts touch, hashed: low 16 bits the caps' facets
 * ((col * 128 + row) * 2 + triangle of the cell), high 16 bits the grid edges under the bars' cylinders ((col * 128 + row) * 3 + k) -- a
 * change of facet under a cap leaves the mask of word s unchanged; the CPU checker hashes its own contacts the same way.
 * replaces: reads of env.data.contact / data.solver_iter after mj_step (run.py:154-161 reads data.contact the same way) */
#define TGR_TRACE_WORDS 48
int tgr_bind_substep_trace(tgr_handle* h, uint32_t* trace_dev);
/* measurement (bench.py's roofline): keep HIP events around the launch of the dominant kernel -- the step kernel over the live rows (with
 * the donor stream) or over all rows (fused launch) -- in a ring of n_slots pairs, recorded on the stream the kernel is launched on;
 * n_slots = 0 switches it off.  tgr_kernel_times_ms synchronises and returns the durations [ms] of the last min(n, recorded) such launches,
 * oldest first (its return value is their number, or a negative error code).  No reference counterpart */
int tgr_time_kernels(tgr_handle* h, int n_slots);
/* the same, timing only every `every`-th step from now on (an event pair is two packets on the stream: sampled sparsely the timed run is not
 * slowed by its own measurement -- bench.py times one step in four INSIDE its timed region) */
int tgr_time_kernels_every(tgr_handle* h, int n_slots, int every);
int tgr_kernel_times_ms(tgr_handle* h, float* out_ms, int n);
int tgr_synchronize(tgr_handle* h);
const char* tgr_last_error(const tgr_handle* h);
const char* tgr_version(void);
int tgr_sizeof_config(void);
/* RNG key of lane (lane < n: the env itself, else donor lane - n): (seed, key, episode) feeds philox4x32-10 */
uint64_t tgr_lane_key(const tgr_handle* h, int lane);

#ifdef __cplusplus
}
#endif
#endif
