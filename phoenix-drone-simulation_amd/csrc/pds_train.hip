// pds_train.hip -- the small per-step kernels of the on-device PPO rollout / update.
// Each one replaces a chain of 8-12 elementwise PyTorch launches that made the rollout launch-bound
// (30 launches, ~210 us per env-step at 8192 envs against a 6 us step kernel):
//   pds_gaussian_sample   ActorCritic.step's dist.sample() + log_prob().sum(-1)      algs/core.py:370-393
//   pds_rollout_record    buf.store + the ep_ret / ep_len bookkeeping of roll_out     algs/iwpg/iwpg.py:350-385
//   pds_adam_step         torch.optim.Adam.step on the 6 tensors of one MLP           algs/iwpg/iwpg.py:94-104
//   pds_permutation       the mini-batch shuffle of update_value_net                   algs/iwpg/iwpg.py:487-522
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pds.h"
#include "pds_device.h"

namespace pds_train_detail {  // named (not anonymous) so that profiler kernel names are readable

// a = mu + sigma z, z ~ N(0,1) from Philox4x32-10 keyed by (seed, call counter); one thread per env
// (d_out <= 8: at most 2 blocks).  logp = -sum(0.5 z^2 + log sigma + 0.5 log 2 pi).
__global__ __launch_bounds__(256) void sample_kernel(const float *mu, const float *log_std, long long n, int d,
                                                     uint64_t seed, uint64_t call, const unsigned long long *call_base,
                                                     unsigned long long id_base, int deterministic, float *act,
                                                     float *logp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (call_base != nullptr) call += *call_base;  // device-side part of the call counter (hipGraph replays)
  float z[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = 0.f;
  if (!deterministic) {
    const unsigned long long gid = id_base + (unsigned long long)i;
    for (int b = 0; b * 4 < d; ++b) {
      // counter = (sample id lo, sample id hi << 8 | block, call lo, call hi), key = seed
      const pds::U4 r = pds::philox4x32_10((uint32_t)gid, ((uint32_t)(gid >> 32) << 8) | (uint32_t)b, (uint32_t)call,
                                           (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
      pds::box_muller(r.x, r.y, z[4 * b], z[4 * b + 1]);
      pds::box_muller(r.z, r.w, z[4 * b + 2], z[4 * b + 3]);
    }
  }
  float lp = 0.f;
  for (int j = 0; j < d; ++j) {
    const float ls = log_std[j];
    act[i * d + j] = fmaf(expf(ls), z[j], mu[i * d + j]);
    lp += -0.5f * z[j] * z[j] - ls - 0.91893853320467274178f;
  }
  logp[i] = lp;
}

// stores the step's reward / flags into the [T, N] buffers and keeps the running episode return
// and length; finished episodes add (return, length, 1) to stats[0..2] (one atomic triple per block).
__global__ __launch_bounds__(256) void record_kernel(const float *rew, const uint8_t *term, const uint8_t *trunc,
                                                     long long n, float *rew_buf, uint8_t *term_buf,
                                                     uint8_t *trunc_buf, float *ep_ret, float *ep_len, float *stats) {
  __shared__ float red[3][4];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (i < n) {
    const float r = rew[i];
    const uint8_t te = term[i], tr = trunc[i];
    rew_buf[i] = r; term_buf[i] = te; trunc_buf[i] = tr;
    const float er = ep_ret[i] + r, el = ep_len[i] + 1.f;
    const bool done = (te | tr) != 0;
    if (done) { s0 = er; s1 = el; s2 = 1.f; }
    ep_ret[i] = done ? 0.f : er;
    ep_len[i] = done ? 0.f : el;
  }
  for (int d = 32; d >= 1; d >>= 1) { s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (t != 0.f) atomicAdd(stats + threadIdx.x, t);
  }
}

// torch.optim.Adam (no weight decay, no amsgrad) on the flat gradient of one MLP:
// m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(pds_mlp m, const float *g, float *em, float *ev, int total,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2s) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int o_b1 = m.h1 * m.d_in, o_w2 = o_b1 + m.h1, o_b2 = o_w2 + m.h2 * m.h1, o_w3 = o_b2 + m.h2,
            o_b3 = o_w3 + m.d_out * m.h2;
  float *dst;
  if (p < o_b1) dst = const_cast<float *>(m.w1) + p;
  else if (p < o_w2) dst = const_cast<float *>(m.b1) + (p - o_b1);
  else if (p < o_b2) dst = const_cast<float *>(m.w2) + (p - o_w2);
  else if (p < o_w3) dst = const_cast<float *>(m.b2) + (p - o_b2);
  else if (p < o_b3) dst = const_cast<float *>(m.w3) + (p - o_w3);
  else dst = const_cast<float *>(m.b3) + (p - o_b3);
  const float gr = g[p];
  const float mm = b1 * em[p] + (1.f - b1) * gr;
  const float vv = b2 * ev[p] + (1.f - b2) * gr * gr;
  em[p] = mm; ev[p] = vv;
  const float denom = sqrtf(vv) / bc2s + eps;
  *dst = *dst - (lr / bc1) * (mm / denom);
}

// *c += inc by one thread (the device-side call counter of a captured rollout)
__global__ void counter_add_kernel(unsigned long long *c, unsigned long long inc) { *c += inc; }

// out[i] = p(i) for a pseudo-random bijection p of [0, n): a 6-round Feistel network over the smallest power of two
// with an even number of bits >= n (a bijection of that domain whatever the round function is), walked along its
// cycle until it lands in [0, n) again (which restricts a bijection of the larger set to one of the smaller; the
// domain is < 4 n, so < 4 evaluations on average).  Round keys: Philox4x32-10(seed, call).  One elementwise launch in
// place of torch.randperm's radix sort + merge passes (~160 us at 2^19 elements) for the value net's mini-batch
// shuffles (IWPGAlgorithm.update_value_net, algs/iwpg/iwpg.py:487-522: np.random.shuffle of the indices).
__device__ __forceinline__ uint32_t feistel_round(uint32_t r, uint32_t key) {
  uint32_t h = r * 0x9E3779B1u + key;  // murmur3's finaliser on the keyed half
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__global__ __launch_bounds__(256) void permutation_kernel(long long *out, long long n, int half_bits, uint64_t seed,
                                                          uint64_t call) {
  // (the round keys are the same for every thread: grid-stride loop, at most 2^22 threads evaluate the two blocks)
  const pds::U4 k0 = pds::philox4x32_10(0u, 0x7065726du, (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
  const pds::U4 k1 = pds::philox4x32_10(1u, 0x7065726du, (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t key[6] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y};
  const uint32_t mask = half_bits >= 32 ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long x = (unsigned long long)i;
    do {
      uint32_t L = (uint32_t)(x >> half_bits), R = (uint32_t)x & mask;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const uint32_t t = L ^ (feistel_round(R, key[r]) & mask);
        L = R; R = t;
      }
      x = ((unsigned long long)L << half_bits) | R;
    } while (x >= (unsigned long long)n);
    out[i] = (long long)x;
  }
}

}  // namespace pds_train_detail
using namespace pds_train_detail;


extern "C" int pds_gaussian_sample_dev(const float *d_mu, const float *d_log_std, int64_t n, int d_out, uint64_t seed,
                                       const uint64_t *d_call_base, uint64_t call_offset, uint64_t id_base,
                                       int deterministic, float *d_act, float *d_logp, void *stream) {
  if (!d_mu || !d_log_std || !d_act || !d_logp || n < 1 || d_out < 1 || d_out > 8) return PDS_EINVAL;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_mu, d_log_std,
                     (long long)n, d_out, seed, call_offset, reinterpret_cast<const unsigned long long *>(d_call_base),
                     (unsigned long long)id_base, deterministic, d_act, d_logp);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_gaussian_sample(const float *d_mu, const float *d_log_std, int64_t n, int d_out, uint64_t seed,
                                   uint64_t call, uint64_t id_base, int deterministic, float *d_act, float *d_logp,
                                   void *stream) {
  return pds_gaussian_sample_dev(d_mu, d_log_std, n, d_out, seed, nullptr, call, id_base, deterministic, d_act, d_logp, stream);
}

extern "C" int pds_permutation(int64_t *d_out, int64_t n, uint64_t seed, uint64_t call, void *stream) {
  if (!d_out || n < 1 || n > (1ll << 62)) return PDS_EINVAL;
  int bits = 1;
  while (bits < 63 && (1ll << bits) < n) ++bits;
  const int half_bits = (bits + 1) / 2;  // the Feistel domain: 2^(2 half_bits) >= n, < 4 n
  long long blocks = (n + 255) / 256;
  if (blocks > (1ll << 14)) blocks = 1ll << 14;  // grid-stride beyond 2^22 elements (no 32-bit grid overflow for any n)
  hipLaunchKernelGGL(pds_train_detail::permutation_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (long long *)d_out, (long long)n, half_bits, seed, call);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_counter_add(uint64_t *d_counter, uint64_t inc, void *stream) {
  if (!d_counter) return PDS_EINVAL;
  hipLaunchKernelGGL(pds_train_detail::counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long *>(d_counter), (unsigned long long)inc);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_rollout_record(const float *d_rew, const uint8_t *d_term, const uint8_t *d_trunc, int64_t n,
                                  float *d_rew_buf, uint8_t *d_term_buf, uint8_t *d_trunc_buf, float *d_ep_ret,
                                  float *d_ep_len, float *d_stats, void *stream) {
  if (!d_rew || !d_term || !d_trunc || !d_rew_buf || !d_term_buf || !d_trunc_buf || !d_ep_ret || !d_ep_len || !d_stats ||
      n < 1)
    return PDS_EINVAL;
  hipLaunchKernelGGL(record_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rew, d_term,
                     d_trunc, (long long)n, d_rew_buf, d_term_buf, d_trunc_buf, d_ep_ret, d_ep_len, d_stats);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}

extern "C" int pds_adam_step(const pds_mlp *m, const float *d_grads, float *d_exp_avg, float *d_exp_avg_sq,
                             int64_t step, float lr, float beta1, float beta2, float eps, void *stream) {
  if (!m || !d_grads || !d_exp_avg || !d_exp_avg_sq || step < 1) return PDS_EINVAL;
  const int total = pds_mlp_param_count(m);
  if (total < 0) return PDS_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, *m, d_grads, d_exp_avg,
                     d_exp_avg_sq, total, lr, beta1, beta2, eps, bc1, bc2s);
  return hipGetLastError() == hipSuccess ? PDS_OK : PDS_EHIP;
}
