/*
 * tgr_batch.c -- batched driver around the scalar oracle env (OpenMP over envs).
 * TEST INFRASTRUCTURE: used by tests as the checker and by bench.py's cpu_baseline leg.
 */
#include "tgr_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct tgo_batch {
  int n;
  tgo_env* envs;
  uint64_t seed;
  uint64_t first_env_id;
  uint64_t* episode;
} tgo_batch;

tgo_batch* tgo_batch_create(const tgo_env_cfg* cfg, int n, uint64_t seed, uint64_t first_env_id) {
  tgo_batch* b = (tgo_batch*)calloc(1, sizeof(tgo_batch));
  b->n = n; b->seed = seed; b->first_env_id = first_env_id;
  b->envs = (tgo_env*)calloc((size_t)n, sizeof(tgo_env));
  b->episode = (uint64_t*)calloc((size_t)n, sizeof(uint64_t));
  for (int i = 0; i < n; i++) tgo_env_init(&b->envs[i], cfg);
  return b;
}
void tgo_batch_destroy(tgo_batch* b) {
  if (!b) return;
  free(b->envs); free(b->episode); free(b);
}
tgo_env* tgo_batch_env(tgo_batch* b, int i) { return &b->envs[i]; }
int tgo_batch_obs_dim(const tgo_batch* b) { return b->envs[0].obs_dim; }

void tgo_batch_reset(tgo_batch* b, const uint8_t* mask, double* obs, int threads) {
  int D = b->envs[0].obs_dim;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int i = 0; i < b->n; i++) {
    if (mask && !mask[i]) continue;
    tgo_env_reset_seeded(&b->envs[i], b->seed, b->first_env_id + (uint64_t)i, b->episode[i], obs ? obs + (size_t)i * D : NULL);
    b->episode[i] += 1;
  }
}

/* VecEnv semantics: done envs are auto-reset; obs holds the reset obs, terminal_obs the last one */
void tgo_batch_step(tgo_batch* b, const float* actions, double* obs, double* reward, uint8_t* terminated, uint8_t* truncated,
                    double* terminal_obs, double* info, int auto_reset, int threads) {
  int D = b->envs[0].obs_dim;
  (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int i = 0; i < b->n; i++) {
    int term, trunc;
    double linfo[TGO_INFO_K];
    tgo_env_step(&b->envs[i], actions + 6 * (size_t)i, obs + (size_t)i * D, reward + i, &term, &trunc, linfo);
    terminated[i] = (uint8_t)term; truncated[i] = (uint8_t)trunc;
    if (info) memcpy(info + (size_t)i * TGO_INFO_K, linfo, sizeof(linfo));
    if (auto_reset && (term || trunc)) {
      if (terminal_obs) memcpy(terminal_obs + (size_t)i * D, obs + (size_t)i * D, (size_t)D * sizeof(double));
      tgo_env_reset_seeded(&b->envs[i], b->seed, b->first_env_id + (uint64_t)i, b->episode[i], obs + (size_t)i * D);
      b->episode[i] += 1;
    }
  }
}

/* state exchange used by parity tests: aux = {xy_lag[2], psi_lag, reset_psi, x_vel, y_vel, step_num, elapsed, ep_return} */
int tgo_sizeof_env(void) { return (int)sizeof(tgo_env); }
int tgo_sizeof_data(void) { return (int)sizeof(tgo_data); }
int tgo_sizeof_model(void) { return (int)sizeof(tgo_model); }
int tgo_sizeof_cfg(void) { return (int)sizeof(tgo_env_cfg); }
