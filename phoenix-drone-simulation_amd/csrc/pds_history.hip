// pds_history.hip -- observation histories other than the default of 2 (observation_history_size = H,
// envs/base.py:44, 303-319, 417-431; experiments/04_* use 1, 2, 4, 6, 8).
//
// The step kernel produces the reference's default row [o(k-1), u(k-2), o(k), u(k-1)] (two "halves" of
// |o| + 4 floats).  For another H the env keeps the last H halves of every env, [N, H, half], and one
// launch per step advances them:
//   env still running:   hist' = [hist[1:], newest half]
//   env finished (auto-reset): final = [hist[1:], newest half of its LAST observation (final_obs)]
//                              hist' = [H - 1 copies of the reset row's first half, its second half]
//                              (DroneBaseEnv.reset fills the deques with the first reset observation and
//                               compute_history() appends the next one, envs/base.py:417-431)
// HBM-bound
// (reads H * half + up to 2 * half floats per env, writes H * half, + H * half for a finished env).
// Until round 2 this was a chain of torch.cat / torch.where calls (5 - 25 x the step kernel's own time).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"

namespace pds_history_detail {

constexpr int kEnvsPerBlock = 64;

// One block per 64 envs: their H * half floats each are one contiguous piece of hist' (coalesced 4-B streams);
// (env, slot, float) of an element come from two float multiplications by reciprocals (exact for these
// small integers: the nearest integer boundary is >= 0.5 / (H * half) away), not from integer divisions.
// (Staging the piece through LDS for 16-byte accesses was slower: 403 vs 337 us per step at 2^20 envs, H = 4.)
__global__ __launch_bounds__(256) void history_kernel(long long n, int half, int H, const float *__restrict__ obs2,
                                                      const uint8_t *__restrict__ term, const uint8_t *__restrict__ trunc,
                                                      const float *__restrict__ final_obs2, int auto_reset,
                                                      const float *__restrict__ hist_in, float *__restrict__ hist_out,
                                                      float *__restrict__ final_hist) {
  const int row = H * half;
  const float inv_row = 1.0f / (float)row, inv_half = 1.0f / (float)half;
  const long long env0 = (long long)blockIdx.x * kEnvsPerBlock;
  const int envs = (int)min((long long)kEnvsPerBlock, n - env0);
  const long long base = env0 * row;
  for (int k = threadIdx.x; k < envs * row; k += 256) {
    const int el = (int)(((float)k + 0.5f) * inv_row);
    const int r = k - el * row;
    const int j = (int)(((float)r + 0.5f) * inv_half), c = r - j * half;  // history slot, float within the half
    const long long env = env0 + el;
    const bool done = (term[env] | trunc[env]) != 0;
    const float *o2 = obs2 + env * (2ll * half);
    // the env's history as it stands after this step, before any reset
    float shifted;
    if (j < H - 1) {
      shifted = hist_in[base + k + half];
    } else {
      shifted = (auto_reset && done) ? final_obs2[env * (2ll * half) + half + c] : o2[half + c];
    }
    float out = shifted;
    if (auto_reset && done) {
      if (final_hist != nullptr) final_hist[base + k] = shifted;
      out = (j < H - 1) ? o2[c] : o2[half + c];
    }
    hist_out[base + k] = out;
  }
}

}  // namespace pds_history_detail

extern "C" int pds_history_advance(int64_t n, int half, int history, const float *d_obs2, const uint8_t *d_terminated,
                                   const uint8_t *d_truncated, const float *d_final_obs2, int auto_reset,
                                   const float *d_hist_in, float *d_hist_out, float *d_final_hist, void *stream) {
  if (n < 1 || half < 1 || history < 1 || !d_obs2 || !d_terminated || !d_truncated || !d_hist_in || !d_hist_out ||
      d_hist_in == d_hist_out || (auto_reset && !d_final_obs2))
    return PDS_EINVAL;
  if ((long long)history * half > 4096) return PDS_EINVAL;  // (the kernel's index arithmetic is exact up to here)
  const unsigned grid = (unsigned)((n + pds_history_detail::kEnvsPerBlock - 1) / pds_history_detail::kEnvsPerBlock);
  hipLaunchKernelGGL(pds_history_detail::history_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long long)n, half, history,
                     d_obs2, d_terminated, d_truncated, d_final_obs2, auto_reset, d_hist_in, d_hist_out, d_final_hist);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}
