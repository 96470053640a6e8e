// pds_gae.hip -- batched GAE-lambda / value targets / discounted returns over a lockstep rollout
// [T, N] (the caller of the hot path: SURVEY.md section 8f rank 1).
//
// Replaces core.Buffer.finish_path + calculate_adv_and_value_targets of the reference trainer
// (algs/core.py:461-533), which runs scipy.signal.lfilter once per finished path on the host
// (discount_cumsum, algs/core.py:105-119).  Here one thread owns one env column and scans it
// backwards in time; at step t all lanes of a wave read consecutive envs, so every access is a
// coalesced 4 B/lane stream.  HBM-bound: ~26 B per (t, env).
//
// Path boundaries: an env that terminated at step t bootstraps with 0, one that was truncated
// (TimeLimit) with V(final_obs) -- ALSO when it terminated on that very step: the reference tests
// `if truncated or epoch_ended: v = V(o)` first (algs/iwpg/iwpg.py:374-379; tests/golden/rollout.npz holds nine such
// paths); the last step of the rollout bootstraps with V(o_T) unless the env finished exactly there.  Reward scaling (use_reward_scaling,
// algs/core.py:523-529): the rewards that enter the TD residuals are divided by the running std of
// the discounted returns and clipped to +-10; the discounted returns themselves use raw rewards.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"

namespace pds_gae_detail {  // named (not anonymous) so that profiler kernel names are readable

__global__ __launch_bounds__(256) void gae_kernel(const float *__restrict__ rew, const float *__restrict__ val,
                                                  const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc,
                                                  const float *__restrict__ final_val, const float *__restrict__ last_val,
                                                  float gamma, float lam, float rew_scale, float rew_clip,
                                                  long long T, long long N, float *__restrict__ adv,
                                                  float *__restrict__ target_v, float *__restrict__ disc_ret) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float next_val = last_val[n];  // V(o_T): epoch cut-off bootstrap
  float next_ret = next_val;     // discount_cumsum(rews + [last_val]) carries the bootstrap as a reward
  float next_adv = 0.f;
  const float gl = gamma * lam;
  for (long long t = T - 1; t >= 0; --t) {
    const long long i = t * N + n;
    const float r = rew[i], v = val[i];
    const bool te = term[i] != 0, tr = trunc[i] != 0;
    if (te || tr) {  // a path ends at t: finish_path(last_val)
      const float b = (tr && final_val != nullptr) ? final_val[i] : 0.f;  // (cut wins over terminated: iwpg.py:374-379)
      next_val = b;
      next_ret = b;
      next_adv = 0.f;
    }
    float rs = r;
    if (rew_scale > 0.f) rs = fminf(fmaxf(r * rew_scale, -rew_clip), rew_clip);
    const float delta = rs + gamma * next_val - v;  // algs/core.py:467
    const float a = delta + gl * next_adv;          // discount_cumsum(deltas, gamma * lam)
    const float g = r + gamma * next_ret;           // discount_cumsum(rews, gamma)[:-1]
    adv[i] = a;
    target_v[i] = a + v;                            // algs/core.py:469
    disc_ret[i] = g;
    next_val = v;
    next_adv = a;
    next_ret = g;
  }
}

}  // namespace pds_gae_detail
using namespace pds_gae_detail;

extern "C" int pds_gae(const float *d_rew, const float *d_val, const uint8_t *d_terminated,
                       const uint8_t *d_truncated, const float *d_final_val, const float *d_last_val,
                       float gamma, float lam, float rew_scale, float rew_clip, int64_t T, int64_t N,
                       float *d_adv, float *d_target_v, float *d_disc_ret, void *stream) {
  if (!d_rew || !d_val || !d_terminated || !d_truncated || !d_last_val || !d_adv || !d_target_v || !d_disc_ret ||
      T < 1 || N < 1)
    return PDS_EINVAL;
  const dim3 grid((unsigned)((N + 255) / 256));
  hipLaunchKernelGGL(gae_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_rew, d_val, d_terminated, d_truncated,
                     d_final_val, d_last_val, gamma, lam, rew_scale, rew_clip, (long long)T, (long long)N, d_adv,
                     d_target_v, d_disc_ret);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}
