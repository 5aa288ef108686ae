// pf_ppo.hip — PPO rollout / GAE / minibatch update kernels + the ppo_* half of the C ABI.
//
// Replaces rsl_rl v1.0.2 (un-vendored dependency of the reference; call sites legged_gym/utils/task_registry.py:147,
// legged_gym/scripts/train.py:44): ActorCritic.act/evaluate, RolloutStorage.add_transitions/compute_returns/
// mini_batch_generator, PPO.update (SURVEY.md Appendix B lists the semantics restated here).
// Dense layers run on the fp32 MFMA GEMM of pf_gemm.h; everything else is small fused elementwise/reduction kernels.
// All reductions are two-stage and atomic-free, so an update is bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pointfoot_hip.h"
#include "../../include/pointfoot_hip_dev.h"
#include "pf_gemm.h"
#include "pf_gemm3.h"
#include "pf_gemm3f.h"

void pf_set_error(const std::string& s);

#define PPO_HIP_OK(expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) { pf_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return 1; } \
  } while (0)

// scalar slots in d_scalars
enum { SC_LR = 0, SC_KL = 1, SC_GNORM = 2, SC_VLOSS = 3, SC_SLOSS = 4, SC_STEP = 5, SC_ADV_MEAN = 6, SC_ADV_STD = 7,
       SC_CLIPCOEF = 8, SC_NUPD = 9, SC_ADV_SUM = 10, SC_ADV_SUMSQ = 11, SC_ADV_CNT = 12, SC_ENTROPY = 13 };

namespace {

struct Layer { int in, out; size_t w, b; };   // offsets into the flat parameter buffer

struct Net {
  int nl = 0;
  Layer l[PPO_MAX_LAYERS];
};

constexpr int RED_BLOCKS = 240;     // stage-1 blocks of every two-stage reduction
constexpr int PPO_SQ_PARTIALS = 8192;   // capacity of the gradient norm's per-block partials (ppo_minibatch_step)

// ------------------------------------------------------------------------------------- philox / normal
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// ------------------------------------------------------------------------------------- rollout
// Normal(mu, std).sample(), log_prob, store the transition (PPO.act + RolloutStorage.add_transitions)
__global__ void k_sample(int N, int A, int O, int P, const float* __restrict__ obs, const float* __restrict__ cobs,
                         const float* __restrict__ mean, const float* __restrict__ value, const float* __restrict__ stdp,
                         const float* __restrict__ inject_normal, uint32_t k0, uint32_t k1, uint32_t counter, uint32_t env_base,
                         float* __restrict__ actions_out, float* __restrict__ s_obs, float* __restrict__ s_cobs,
                         float* __restrict__ s_act, float* __restrict__ s_val, float* __restrict__ s_logp,
                         float* __restrict__ s_mu, float* __restrict__ s_sigma) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // transition copy (coalesced grid-stride), then one thread per env for the sample
  for (size_t j = i; j < (size_t)N * O; j += (size_t)gridDim.x * blockDim.x) s_obs[j] = obs[j];
  for (size_t j = i; j < (size_t)N * P; j += (size_t)gridDim.x * blockDim.x) s_cobs[j] = cobs[j];
  if (i >= N) return;
  float lp = 0.0f;
  for (int k = 0; k < A; ++k) {
    float z;
    if (inject_normal) z = inject_normal[(size_t)i * A + k];
    else {
      uint32_t r[4];
      philox4(env_base + (uint32_t)i, counter, 1u, (uint32_t)k, k0, k1, r);
      float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
      float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
      z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    float mu = mean[(size_t)i * A + k], sg = stdp[k];
    float a = mu + sg * z;
    float d = (a - mu) / sg;
    lp += -0.5f * d * d - logf(sg) - 0.91893853320467274178f;
    actions_out[(size_t)i * A + k] = a;
    s_act[(size_t)i * A + k] = a; s_mu[(size_t)i * A + k] = mu; s_sigma[(size_t)i * A + k] = sg;
  }
  s_val[i] = value[i]; s_logp[i] = lp;
}

// PPO.process_env_step for one env: bootstrap on time-outs, store reward / done, the runner's episode bookkeeping.  Runs either as
// its own launch (k_process) or, for a step whose env outputs stay in place until then (ppo_process_env_step_deferred), in the tail
// of the NEXT transition's k_act_heads -- one launch fewer per policy step.
struct ProcArgs {
  const float* rew; const uint8_t* dones; const uint8_t* time_outs; const float* s_val;
  float *s_rew, *s_done, *ep_ret, *ep_len, *ep_stats, *fin_ret, *fin_len;
  float gamma;
};
// in two halves, so that a host kernel can issue the loads long before it needs them
struct ProcIn { float r, er, el, v; bool done, to; };
__device__ __forceinline__ ProcIn process_load(const ProcArgs& p, int i) {
  ProcIn x;
  x.r = p.rew[i]; x.done = p.dones[i] != 0;
  x.er = p.ep_ret ? p.ep_ret[i] : 0.0f; x.el = p.ep_ret ? p.ep_len[i] : 0.0f;
  x.to = p.time_outs ? p.time_outs[i] != 0 : false; x.v = p.time_outs ? p.s_val[i] : 0.0f;
  return x;
}
__device__ __forceinline__ void process_store(const ProcArgs& p, int i, const ProcIn& x) {
  float r = x.r;
  if (p.ep_ret) {     // OnPolicyRunner bookkeeping (cur_reward_sum / cur_episode_length / rewbuffer), kept on the device
    float cr = x.er + r, cl = x.el + 1.0f;
    if (x.done) {
      // with the finished-episode arrays bound, k_episode_ring sums them into ep_stats once per rollout (in a fixed order); ~100
      // same-address atomics per step otherwise -- they were most of this kernel's 4.9 us
      if (p.fin_ret) { p.fin_ret[i] = cr; p.fin_len[i] = cl; }
      else { atomicAdd(p.ep_stats, cr); atomicAdd(p.ep_stats + 1, cl); atomicAdd(p.ep_stats + 2, 1.0f); }
      cr = 0.0f; cl = 0.0f;
    }
    p.ep_ret[i] = cr; p.ep_len[i] = cl;
  }
  if (p.time_outs) r += p.gamma * (x.v * (x.to ? 1.0f : 0.0f));
  p.s_rew[i] = r; p.s_done[i] = x.done ? 1.0f : 0.0f;
}
__device__ __forceinline__ void process_body(const ProcArgs& p, int i) { process_store(p, i, process_load(p, i)); }

// k_thin_fwd (both heads) + k_sample in one launch for the rollout: 8 lanes per env compute the action means and the
// value from the last hidden activations, then lane k of the group draws action dimension k (same Philox keys as k_sample).
__global__ __launch_bounds__(256) void k_act_heads(int N, int A, int O, int P, int Ka, int Kc, const float* __restrict__ obs,
                                                   const float* __restrict__ cobs, const float* __restrict__ Ha,
                                                   const float* __restrict__ Hc, const float* __restrict__ Wa,
                                                   const float* __restrict__ ba, const float* __restrict__ Wc,
                                                   const float* __restrict__ bc, const float* __restrict__ stdp,
                                                   const float* __restrict__ inject_normal, uint32_t k0, uint32_t k1, uint32_t counter,
                                                   uint32_t env_base, float* __restrict__ actions_out, float* __restrict__ s_obs,
                                                   float* __restrict__ s_cobs, float* __restrict__ s_act, float* __restrict__ s_val,
                                                   float* __restrict__ s_logp, float* __restrict__ s_mu, float* __restrict__ s_sigma, ProcArgs prev) {
  extern __shared__ __attribute__((aligned(16))) float asm_[];
  float* sWa = asm_;               // [A][Ka]
  float* sWc = sWa + A * Ka;       // [Kc]
  const int lane8 = threadIdx.x & 7;
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
  // Every global load of the kernel goes out before the first wait: the hidden rows of this lane's env (up to 128 columns per
  // network in registers), the deferred process_env_step's inputs, the head weights for LDS and the transition copy's first
  // pieces -- one memory round trip for the launch instead of three dependent ones (weights -> barrier -> rows; a load -> store
  // pair per copy iteration).
  const bool rows4 = ((Ka | Kc) & 3) == 0, pre = rows4 && Ka <= 128 && Kc <= 128 && i < N;
  float4 xa[4], xc[4];
  if (pre) {
    const float4* xr = reinterpret_cast<const float4*>(Ha + (size_t)i * Ka);
    const float4* cr = reinterpret_cast<const float4*>(Hc + (size_t)i * Kc);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xa[u] = (lane8 + 8 * u) < (Ka >> 2) ? xr[lane8 + 8 * u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      xc[u] = (lane8 + 8 * u) < (Kc >> 2) ? cr[lane8 + 8 * u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  }
  ProcIn pin{};
  if (prev.rew && lane8 == 0 && i < N) pin = process_load(prev, i);      // the previous transition's deferred process_env_step: loads now, rest at the end
  for (int k = threadIdx.x; k < A * Ka; k += 256) sWa[k] = Wa[k];
  for (int k = threadIdx.x; k < Kc; k += 256) sWc[k] = Wc[k];
  {
    // transition copy: 16 bytes per lane where the two arrays allow it, four pieces in flight per lane
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x, gs = (size_t)gridDim.x * 256;
    const size_t no = (size_t)N * O, np_ = (size_t)N * P;
    const bool v4 = ((no | np_) & 3) == 0 && ((reinterpret_cast<uintptr_t>(obs) | reinterpret_cast<uintptr_t>(cobs) |
                                                reinterpret_cast<uintptr_t>(s_obs) | reinterpret_cast<uintptr_t>(s_cobs)) & 15) == 0;
    if (v4) {
      const float4 *o4 = reinterpret_cast<const float4*>(obs), *c4 = reinterpret_cast<const float4*>(cobs);
      float4 *so4 = reinterpret_cast<float4*>(s_obs), *sc4 = reinterpret_cast<float4*>(s_cobs);
      const size_t n1 = no >> 2, n2 = np_ >> 2;
      for (size_t j = g; j < n1 + n2; j += 4 * gs) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t q = j + u * gs; if (q < n1 + n2) v[u] = q < n1 ? o4[q] : c4[q - n1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t q = j + u * gs; if (q < n1 + n2) { if (q < n1) so4[q] = v[u]; else sc4[q - n1] = v[u]; } }
      }
    } else {
      for (size_t j = g; j < no; j += gs) s_obs[j] = obs[j];
      for (size_t j = g; j < np_; j += gs) s_cobs[j] = cobs[j];
    }
  }
  __syncthreads();
  if (i >= N) return;                 // whole 8-lane groups leave together (shuffles below stay inside a group)
  float mu[16], val = 0.0f;
#pragma unroll
  for (int o = 0; o < 16; ++o) mu[o] = 0.0f;
  if (pre) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k4 = lane8 + 8 * u;
      if (k4 < (Ka >> 2)) {
        const float4 x = xa[u];
#pragma unroll
        for (int o = 0; o < 16; ++o) if (o < A) {
          float4 w = *reinterpret_cast<const float4*>(&sWa[o * Ka + 4 * k4]);
          mu[o] += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
        }
      }
      if (k4 < (Kc >> 2)) {
        const float4 x = xc[u], w = *reinterpret_cast<const float4*>(&sWc[4 * k4]);
        val += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
      }
    }
  } else if (rows4) {
    const float4* xr = reinterpret_cast<const float4*>(Ha + (size_t)i * Ka);
    for (int k4 = lane8; k4 < (Ka >> 2); k4 += 8) {
      float4 x = xr[k4];
#pragma unroll
      for (int o = 0; o < 16; ++o) if (o < A) {
        float4 w = *reinterpret_cast<const float4*>(&sWa[o * Ka + 4 * k4]);
        mu[o] += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
      }
    }
    const float4* cr = reinterpret_cast<const float4*>(Hc + (size_t)i * Kc);
    for (int k4 = lane8; k4 < (Kc >> 2); k4 += 8) {
      float4 x = cr[k4], w = *reinterpret_cast<const float4*>(&sWc[4 * k4]);
      val += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
    }
  } else {
    for (int k = lane8; k < Ka; k += 8) {
      float x = Ha[(size_t)i * Ka + k];
#pragma unroll
      for (int o = 0; o < 16; ++o) if (o < A) mu[o] += x * sWa[o * Ka + k];
    }
    for (int k = lane8; k < Kc; k += 8) val += Hc[(size_t)i * Kc + k] * sWc[k];
  }
#pragma unroll
  for (int o = 0; o < 16; ++o) if (o < A) {
    float v = mu[o];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    mu[o] = v + ba[o];
  }
  val += __shfl_xor(val, 1, 64); val += __shfl_xor(val, 2, 64); val += __shfl_xor(val, 4, 64);
  val += bc[0];
  float lp = 0.0f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int k = lane8 + 8 * kk;
    if (k < A) {
      float z;
      if (inject_normal) z = inject_normal[(size_t)i * A + k];
      else {
        uint32_t r[4];
        philox4(env_base + (uint32_t)i, counter, 1u, (uint32_t)k, k0, k1, r);
        float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
        float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
        z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
      }
      float m = 0.0f;
#pragma unroll
      for (int o = 0; o < 16; ++o) if (o == k) m = mu[o];     // register select (mu[] is indexed by constants only)
      float sg = stdp[k];
      float a = m + sg * z;
      float d = (a - m) / sg;
      lp += -0.5f * d * d - logf(sg) - 0.91893853320467274178f;
      actions_out[(size_t)i * A + k] = a;
      s_act[(size_t)i * A + k] = a; s_mu[(size_t)i * A + k] = m; s_sigma[(size_t)i * A + k] = sg;
    }
  }
  lp += __shfl_xor(lp, 1, 64); lp += __shfl_xor(lp, 2, 64); lp += __shfl_xor(lp, 4, 64);
  if (lane8 == 0) {
    s_val[i] = val; s_logp[i] = lp;
    if (prev.rew) process_store(prev, i, pin);
  }
}

__global__ void k_process(int N, ProcArgs p) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) process_body(p, i);
}

// rsl_rl OnPolicyRunner.learn: `rewbuffer.extend(cur_reward_sum[new_ids])`, `lenbuffer.extend(cur_episode_length[new_ids])` with
// deque(maxlen=100) -- the last 100 FINISHED EPISODES in (step, env index) order.  One launch per rollout, one workgroup: a scan over
// the rollout's T*N done flags in that order appends the finished episodes to a ring of PPO_EP_RING entries in device memory; no
// host synchronisation, the runner reads the ring only when it logs.
// Two launches per rollout.  k_episode_chunks: the T*N flags in chunks of EP_CHUNK-multiples, one workgroup per chunk -- count of
// finished episodes, sum of their returns and of their lengths (the runner's ep_stats) into part[4 b ..].  k_episode_ring: one
// workgroup of sixteen wavefronts; every wavefront takes one of the LAST chunks (its number of episodes "behind" comes from the chunk
// counts), reads it backwards 512 flags at a time with eight independent coalesced loads and writes the episodes that are among the
// last PPO_EP_RING of the rollout -- of ~2000 finished episodes per rollout only the last 100 can stay, so ~5 000 of the 98 304
// flags are read a second time.  (The first versions walked all flags in one workgroup with one dependent load per iteration: 90 us.)
constexpr int EP_CHUNK = 512;
__device__ __forceinline__ float block_sum(float v, float* sm);
__host__ __device__ inline long long ep_chunk_len(long long n) { return (long long)EP_CHUNK * ((n + (long long)EP_CHUNK * 1024 - 1) / ((long long)EP_CHUNK * 1024)); }
__global__ void __launch_bounds__(256) k_episode_chunks(long long n, long long chunk, const float* __restrict__ s_done, const float* __restrict__ fin,
                                                        float* __restrict__ part) {
  __shared__ float sm[8];
  const long long lo = (long long)blockIdx.x * chunk;
  float c = 0.0f, sr = 0.0f, sl = 0.0f;
  for (long long k = threadIdx.x; k < chunk; k += 256) {
    const long long i = lo + k;
    if (i < n && s_done[i] != 0.0f) { c += 1.0f; sr += fin[i]; sl += fin[n + i]; }
  }
  c = block_sum(c, sm); sr = block_sum(sr, sm); sl = block_sum(sl, sm);
  if (threadIdx.x == 0) { part[4 * blockIdx.x] = c; part[4 * blockIdx.x + 1] = sr; part[4 * blockIdx.x + 2] = sl; }
}
__global__ void __launch_bounds__(1024) k_episode_ring(int N, int T, long long chunk, int nchunks, const float* __restrict__ s_done,
                                                       const float* __restrict__ fin, const float* __restrict__ part, int* __restrict__ ring,
                                                       float* __restrict__ ep_stats) {
  __shared__ int cnt[1024];                   // episodes per chunk (nchunks <= 1024), then their inclusive suffix sums
  __shared__ float sums[2][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long n = (long long)N * T;
  cnt[tid] = tid < nchunks ? (int)part[4 * tid] : 0;
  float sr = tid < nchunks ? part[4 * tid + 1] : 0.0f, sl = tid < nchunks ? part[4 * tid + 2] : 0.0f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); sl += __shfl_xor(sl, o, 64); }
  if (lane == 0) { sums[0][w] = sr; sums[1][w] = sl; }
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {        // cnt[b] := episodes in chunks b .. nchunks-1
    const int v = tid + d < 1024 ? cnt[tid + d] : 0;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  const int tot = cnt[0];
  const int head = ring[0], count = ring[1], seq = ring[2], step0 = ring[3];
  __syncthreads();                            // every wavefront has the header before thread 0 rewrites it
  for (int b = nchunks - 1 - w; b >= 0; b -= 16) {
    int behind = b + 1 < nchunks ? cnt[b + 1] : 0;
    if (behind >= PPO_EP_RING) break;         // (suffix counts only grow towards the front: nothing further ahead can qualify)
    if (cnt[b] == behind) continue;           // no episode ended in this chunk
    for (long long hi = chunk; hi > 0 && behind < PPO_EP_RING; hi -= EP_CHUNK) {
      const long long base = (long long)b * chunk + hi - EP_CHUNK;
      float dn[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const long long i = base + 64 * k + lane; dn[k] = i < n ? s_done[i] : 0.0f; }
#pragma unroll
      for (int k = 7; k >= 0; --k) {
        const long long i = base + 64 * k + lane;
        const bool d = dn[k] != 0.0f;
        const unsigned long long m = __ballot(d);
        if (d) {
          const int rev = behind + __popcll(m >> lane) - 1;     // finished episodes behind this one in (step, env) order
          if (rev < PPO_EP_RING) {
            const int p = tot - 1 - rev;
            int* e = ring + 4 + 4 * ((head + p) % PPO_EP_RING);
            e[0] = __float_as_int(fin[i]); e[1] = __float_as_int(fin[n + i]); e[2] = step0 + (int)(i / N); e[3] = seq + p;
          }
        }
        behind += __popcll(m);
      }
    }
  }
  if (tid == 0) {
    ring[0] = (head + tot) % PPO_EP_RING; ring[1] = count + tot < PPO_EP_RING ? count + tot : PPO_EP_RING;
    ring[2] = seq + tot; ring[3] = step0 + T;
    float tr = 0.0f, tl = 0.0f;
    for (int k = 0; k < 16; ++k) { tr += sums[0][k]; tl += sums[1][k]; }
    ep_stats[0] += tr; ep_stats[1] += tl; ep_stats[2] += (float)tot;
  }
}

// RolloutStorage.compute_returns: GAE reverse scan, one thread per env; advantages = returns - values
__global__ void k_gae(int N, int T, float gamma, float lam, const float* __restrict__ last_values, const float* __restrict__ values,
                      const float* __restrict__ rewards, const float* __restrict__ dones, float* __restrict__ returns,
                      float* __restrict__ adv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float a = 0.0f;
  for (int t = T - 1; t >= 0; --t) {
    float nv = t == T - 1 ? last_values[i] : values[(size_t)(t + 1) * N + i];
    float nnt = 1.0f - dones[(size_t)t * N + i];
    float v = values[(size_t)t * N + i];
    float delta = rewards[(size_t)t * N + i] + nnt * gamma * nv - v;
    a = delta + nnt * gamma * lam * a;
    float ret = a + v;
    returns[(size_t)t * N + i] = ret;
    adv[(size_t)t * N + i] = ret - v;
  }
}

// ------------------------------------------------------------------------------------- reductions
__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float s = 0.0f;
  for (int i = 0; i < nw; ++i) s += sm[i];
  return s;
}

// stage 1: partial[b][0..1] = sum, sumsq over a strided slice
__global__ void k_moments1(size_t n, const float* __restrict__ x, float* __restrict__ partial) {
  __shared__ float sm[8];
  float s = 0.0f, q = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float v = x[i]; s += v; q += v * v; }
  s = block_sum(s, sm); q = block_sum(q, sm);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = q; }
}
__global__ void k_moments2(int nb, size_t n, const float* __restrict__ partial, float* __restrict__ scalars) {
  // one wavefront: lane-strided double sums, then a butterfly (a fixed order; one thread walking the 240 partials took 14 us)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nb; b += 64) { s += partial[2 * b]; q += partial[2 * b + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (threadIdx.x == 0) { scalars[SC_ADV_SUM] = (float)s; scalars[SC_ADV_SUMSQ] = (float)q; scalars[SC_ADV_CNT] = (float)n; }
}
// advantages = (adv - mean) / (std + 1e-8), torch.std is unbiased; sums may have been all-reduced by the host
__global__ void k_normalize(size_t n, float* __restrict__ x, float* __restrict__ scalars) {
  float cnt = scalars[SC_ADV_CNT], s = scalars[SC_ADV_SUM], q = scalars[SC_ADV_SUMSQ];
  float mean = s / cnt;
  float var = fmaxf((q - cnt * mean * mean) / (cnt - 1.0f), 0.0f);
  float sd = sqrtf(var);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = (x[i] - mean) / (sd + 1e-8f);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scalars[SC_ADV_MEAN] = mean; scalars[SC_ADV_STD] = sd; }
}

// out[i] = sum_s slabs[s][i].  Block = 64 consecutive outputs x 4 slab lanes (each lane sums every 4th slab with two
// independent accumulators), combined through LDS in a fixed order: deterministic, 4x the parallelism of one thread
// per output and short dependent-load chains.
// (every body returns the square of what this thread wrote -- the gradient norm's first stage, ppo_minibatch_step)
__device__ __forceinline__ float reduce_body(int n, int S, const float* __restrict__ slabs, float* __restrict__ out, int bid) {
  __shared__ float sm[4][64];
  int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  int i = bid * 64 + col;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.0f;
  if (i < n) {
    // eight independent loads in flight per thread: the narrow layers have few columns and many slabs
    int k = sl;
    for (; k + 28 < S; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slabs[(size_t)(k + 4 * u) * n + i];
    }
    for (; k < S; k += 4) a[0] += slabs[(size_t)k * n + i];
  }
  sm[sl][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  float q = 0.0f;
  if (sl == 0 && i < n) { const float r = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]); out[i] = r; q = r * r; }
  return q;
}
// same reduction, four columns per thread (n % 4 == 0, 16-byte aligned slabs and output): dwordx4 loads
__device__ __forceinline__ float reduce4_body(int n4, int S, const float4* __restrict__ slabs, float4* __restrict__ out, int bid) {
  __shared__ float4 sm[4][64];
  int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  int i = bid * 64 + col;
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [](float4& x, const float4& y) { x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; };
  if (i < n4) {
    int k = sl;
    for (; k + 12 < S; k += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) add(a[u], slabs[(size_t)(k + 4 * u) * n4 + i]);
    }
    for (; k < S; k += 4) add(a[0], slabs[(size_t)k * n4 + i]);
  }
  add(a[0], a[1]); add(a[2], a[3]); add(a[0], a[2]);
  sm[sl][col] = a[0];
  __syncthreads();
  float qq = 0.0f;
  if (sl == 0 && i < n4) {
    float4 r = sm[0][col]; add(r, sm[1][col]); float4 q = sm[2][col]; add(q, sm[3][col]); add(r, q);
    out[i] = r;
    qq = (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
  }
  return qq;
}
// few columns, many slabs (the thin heads: n ~ 800, S ~ 800): 16 columns x 16 slab lanes per block
__device__ __forceinline__ float reduce_tall_body(int n, int S, const float* __restrict__ slabs, float* __restrict__ out, int bid) {
  __shared__ float sm[16][17];
  int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  int i = bid * 16 + col;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.0f;
  if (i < n) {
    int k = sl;
    for (; k + 112 < S; k += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slabs[(size_t)(k + 16 * u) * n + i];
    }
    for (; k < S; k += 16) a[0] += slabs[(size_t)k * n + i];
  }
  sm[sl][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  float q = 0.0f;
  if (sl == 0 && i < n) {
    float t = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += sm[j][col];
    out[i] = t;
    q = t * t;
  }
  return q;
}
// one split-K slab reduction: out[i] = sum_s slabs[s][i], fixed order.  kind 0: 64 columns per block; 1: 256 columns per
// block as float4; 2: 16 columns x 16 slab lanes (few columns, many slabs)
struct ReduceJob { int n, S, kind, blocks; const float* slabs; float* out; };
__device__ __forceinline__ float reduce_job(const ReduceJob& j, int bid) {
  if (j.kind == 1) return reduce4_body(j.n / 4, j.S, (const float4*)j.slabs, (float4*)j.out, bid);
  if (j.kind == 2) return reduce_tall_body(j.n, j.S, j.slabs, j.out, bid);
  return reduce_body(j.n, j.S, j.slabs, j.out, bid);
}
__global__ __launch_bounds__(256) void k_reduce_slabs(ReduceJob j) { reduce_job(j, (int)blockIdx.x); }
// two reductions in one launch (the actor's and the critic's layer i)
__global__ __launch_bounds__(256) void k_reduce_slabs2(ReduceJob j0, ReduceJob j1) {
  if ((int)blockIdx.x < j0.blocks) reduce_job(j0, (int)blockIdx.x); else reduce_job(j1, (int)blockIdx.x - j0.blocks);
}
static inline ReduceJob reduce_plan(int n, int S, const float* slabs, float* out) {
  ReduceJob j{n, S, 0, (n + 63) / 64, slabs, out};
  if (n < 4096 && S >= 128) { j.kind = 2; j.blocks = (n + 15) / 16; }
  else if ((n & 3) == 0 && ((((size_t)slabs) | ((size_t)out)) & 15) == 0) { j.kind = 1; j.blocks = (n / 4 + 63) / 64; }
  return j;
}
static inline void reduce_slabs(int n, int S, const float* slabs, float* out, hipStream_t s) {
  ReduceJob j = reduce_plan(n, S, slabs, out);
  hipLaunchKernelGGL(k_reduce_slabs, dim3(j.blocks), dim3(256), 0, s, j);
}
// all slab reductions of one backward pass (both networks, every layer) in ONE launch; its last block finishes the loss
// reductions (dL/dstd, KL sum) when `lf.partial` is set -- the arithmetic of k_loss_finish without a launch of its own
// (6.2 us saved for 2.5 us more in this kernel)
struct LossFinish { int nb, A, B; float ecoef; const float* partial; const float* stdp; float* dstd; float* grads_tail; float* scalars; };
__device__ float loss_finish_body(const LossFinish& f);
// sq_partial (ppo_minibatch_step, one process): every block leaves the sum of the squares of the gradient entries it wrote --
// together the blocks write every entry exactly once -- so the gradient norm needs no pass of its own over the gradients
struct ReduceJobs { ReduceJob j[2 * (PPO_MAX_LAYERS + 1)]; int n; LossFinish lf; float* sq_partial; };
__global__ __launch_bounds__(256) void k_reduce_slabs_n(ReduceJobs js) {
  __shared__ float smq[8];
  int b = (int)blockIdx.x;
  float q = 0.0f;
  bool done = false;
  for (int i = 0; i < js.n && !done; ++i) {
    if (b < js.j[i].blocks) { q = reduce_job(js.j[i], b); done = true; }
    else b -= js.j[i].blocks;
  }
  if (!done && js.lf.partial) q = loss_finish_body(js.lf);
  if (js.sq_partial) {
    q = block_sum(q, smq);
    if (threadIdx.x == 0) js.sq_partial[blockIdx.x] = q;
  }
}
static inline int reduce_slabs_n_blocks(const ReduceJobs& js) {
  int blocks = js.lf.partial ? 1 : 0;
  for (int i = 0; i < js.n; ++i) blocks += js.j[i].blocks;
  return blocks;
}
static inline void reduce_slabs_n(const ReduceJobs& js, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_slabs_n, dim3(reduce_slabs_n_blocks(js)), dim3(256), 0, s, js);
}
static inline void reduce_slabs2(const ReduceJob& j0, const ReduceJob& j1, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_slabs2, dim3(j0.blocks + j1.blocks), dim3(256), 0, s, j0, j1);
}

// ------------------------------------------------------------------------------------- minibatch gather
struct GatherJob { const float* src; float* dst; int cols, dst_ld; };      // dst rows are dst_ld >= cols wide, zero padded
struct GatherArgs { GatherJob job[9]; int njobs; };
// rsl_rl mini_batch_generator: rows perm[b] of every storage array -> contiguous minibatch arrays (one launch)
__global__ void k_gather(int B, const int* __restrict__ perm, GatherArgs ga) {
  const GatherJob j = ga.job[blockIdx.y];
  // 32-bit index arithmetic (the host checks B * cols < 2^31): a 64-bit divide per element made this kernel ALU-bound
  const unsigned stride = gridDim.x * blockDim.x;
  if (j.dst_ld == j.cols && (j.cols & 3) == 0 && (((size_t)j.src | (size_t)j.dst) & 15) == 0) {      // rows are whole float4s (critic obs: 148 = 37 x 4)
    const unsigned c4n = (unsigned)j.cols >> 2, n4 = (unsigned)B * c4n;
    const float4* src = reinterpret_cast<const float4*>(j.src);
    float4* dst = reinterpret_cast<float4*>(j.dst);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      unsigned r = i / c4n, c = i - r * c4n;
      dst[i] = src[(size_t)perm[r] * c4n + c];
    }
    return;
  }
  const unsigned cols = (unsigned)j.cols, ld = (unsigned)j.dst_ld, n = (unsigned)B * ld;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned r = i / ld, c = i - r * ld;
    j.dst[i] = c < cols ? j.src[(size_t)perm[r] * cols + c] : 0.0f;
  }
}

// ------------------------------------------------------------------------------------- thin output layers (out <= 8)
// Y[M, NO] = X[M, K] W[NO, K]^T + b     (actor mean / critic value heads)
template <int NO_MAX>
__global__ void k_thin_fwd(int M, int K, int NO, const float* __restrict__ X, const float* __restrict__ W,
                           const float* __restrict__ b, float* __restrict__ Y) {
  extern __shared__ float sw[];   // NO*K
  for (int i = threadIdx.x; i < NO * K; i += blockDim.x) sw[i] = W[i];
  __syncthreads();
  // 8 lanes cooperate on one row so the row read is one contiguous burst (16 B per lane when K % 4 == 0)
  int lane8 = threadIdx.x & 7;
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  if (row >= M) return;
  float acc[NO_MAX];
#pragma unroll
  for (int o = 0; o < NO_MAX; ++o) acc[o] = 0.0f;
  if ((K & 3) == 0) {
    const float4* xr = reinterpret_cast<const float4*>(X + (size_t)row * K);
    for (int k4 = lane8; k4 < (K >> 2); k4 += 8) {
      float4 x = xr[k4];
#pragma unroll
      for (int o = 0; o < NO_MAX; ++o) if (o < NO) {
        float4 w = *reinterpret_cast<const float4*>(&sw[o * K + 4 * k4]);
        acc[o] += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
      }
    }
  } else {
    for (int k = lane8; k < K; k += 8) {
      float x = X[(size_t)row * K + k];
#pragma unroll
      for (int o = 0; o < NO_MAX; ++o) if (o < NO) acc[o] += x * sw[o * K + k];
    }
  }
#pragma unroll
  for (int o = 0; o < NO_MAX; ++o) {
    float v = acc[o];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    if (o < NO && lane8 == 0) Y[(size_t)row * NO + o] = v + b[o];
  }
}
// backward of the thin head: Gprev[M,K] = (dY[M,NO] W[NO,K]) * ELU'(H[M,K]);  slabs: dW[NO,K] and db[NO] partials
template <int NO_MAX>
__global__ void k_thin_bwd(int M, int K, int NO, int rows_per_block, const float* __restrict__ dY, const float* __restrict__ H,
                           const float* __restrict__ W, float* __restrict__ Gprev, float* __restrict__ slabs) {
  float* slabsW = slabs + (size_t)blockIdx.x * ((size_t)NO * K + NO);
  float* slabsB = slabsW + (size_t)NO * K;
  int k = threadIdx.x;              // blockDim.x >= K
  int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float w[NO_MAX], dw[NO_MAX], db[NO_MAX];
#pragma unroll
  for (int o = 0; o < NO_MAX; ++o) { w[o] = (o < NO && k < K) ? W[o * K + k] : 0.0f; dw[o] = 0.0f; db[o] = 0.0f; }
  int r = r0;
  for (; r + 4 <= r1; r += 4) {       // four rows in flight per thread
    float h[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = k < K ? H[(size_t)(r + u) * K + k] : 0.0f; g[u] = 0.0f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int o = 0; o < NO_MAX; ++o) if (o < NO) { float d = dY[(size_t)(r + u) * NO + o]; g[u] += d * w[o]; dw[o] += d * h[u]; db[o] += d; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (k < K) Gprev[(size_t)(r + u) * K + k] = g[u] * (h[u] > 0.0f ? 1.0f : h[u] + 1.0f);
  }
  for (; r < r1; ++r) {
    float h = k < K ? H[(size_t)r * K + k] : 0.0f, g = 0.0f;
#pragma unroll
    for (int o = 0; o < NO_MAX; ++o) if (o < NO) { float d = dY[(size_t)r * NO + o]; g += d * w[o]; dw[o] += d * h; db[o] += d; }
    if (k < K) Gprev[(size_t)r * K + k] = g * (h > 0.0f ? 1.0f : h + 1.0f);
  }
#pragma unroll
  for (int o = 0; o < NO_MAX; ++o) if (o < NO) {
    if (k < K) slabsW[(size_t)o * K + k] = dw[o];
    if (k == 0) slabsB[o] = db[o];
  }
}

// ------------------------------------------------------------------------------------- PPO loss (forward + gradient wrt heads)
// Per sample: surrogate / clipped value / entropy losses and KL (rsl_rl PPO.update); writes dL/dmean, dL/dvalue
// (already scaled by 1/B and the loss coefficients) and per-block partials: [0..A) dL/dstd, A: kl, A+1: value loss,
// A+2: surrogate loss, A+3: entropy.
__global__ void k_loss(int B, int A, float clip, float vcoef, float ecoef, int clipped_value,
                       const float* __restrict__ mean, const float* __restrict__ value, const float* __restrict__ stdp,
                       const float* __restrict__ act, const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                       const float* __restrict__ old_sigma, const float* __restrict__ adv, const float* __restrict__ ret,
                       const float* __restrict__ old_val, float* __restrict__ dmean, float* __restrict__ dvalue,
                       float* __restrict__ partial) {
  __shared__ float sm[8];
  float dstd[16], kl = 0.0f, vl = 0.0f, sl = 0.0f, en = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) dstd[k] = 0.0f;
  const float invB = 1.0f / (float)B;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
    float lp = 0.0f, kls = 0.0f, ens = 0.0f;
    float dmu[16], dsg[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < A) {
      float mu = mean[(size_t)i * A + k], sg = stdp[k], a = act[(size_t)i * A + k];
      float d = a - mu;
      lp += -0.5f * (d * d) / (sg * sg) - logf(sg) - 0.91893853320467274178f;
      dmu[k] = d / (sg * sg);
      dsg[k] = (d * d) / (sg * sg * sg) - 1.0f / sg;
      float om = old_mu[(size_t)i * A + k], os = old_sigma[(size_t)i * A + k];
      kls += logf(sg / os + 1e-5f) + (os * os + (om - mu) * (om - mu)) / (2.0f * sg * sg) - 0.5f;
      ens += 0.5f + 0.91893853320467274178f + logf(sg);
    }
    float ad = adv[i];
    float ratio = expf(lp - old_logp[i]);
    float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
    float g_lp = (s1 >= s2) ? (-ad * ratio) * invB : 0.0f;     // d(mean surrogate)/d(log_prob)
    sl += fmaxf(s1, s2);
    float v = value[i], r = ret[i], gv;
    if (clipped_value) {
      float ov = old_val[i];
      float vc = ov + fminf(fmaxf(v - ov, -clip), clip);
      float e1 = (v - r) * (v - r), e2 = (vc - r) * (vc - r);
      bool inside = fabsf(v - ov) < clip;          // v_clipped == v: both branches identical
      gv = (e1 >= e2 || inside) ? 2.0f * (v - r) : 0.0f;
      vl += fmaxf(e1, e2);
    } else { gv = 2.0f * (v - r); vl += (r - v) * (r - v); }
    dvalue[i] = vcoef * gv * invB;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < A) {
      dmean[(size_t)i * A + k] = g_lp * dmu[k];
      dstd[k] += g_lp * dsg[k];
    }
    kl += kls; en += ens;
  }
  float* out = partial + (size_t)blockIdx.x * (A + 4);
#pragma unroll
  for (int k = 0; k < 16; ++k) if (k < A) { float s = block_sum(dstd[k], sm); if (threadIdx.x == 0) out[k] = s; }
  float s;
  s = block_sum(kl, sm); if (threadIdx.x == 0) out[A] = s;
  s = block_sum(vl, sm); if (threadIdx.x == 0) out[A + 1] = s;
  s = block_sum(sl, sm); if (threadIdx.x == 0) out[A + 2] = s;
  s = block_sum(en, sm); if (threadIdx.x == 0) out[A + 3] = s;
}
// Fused minibatch head: action-mean / value heads forward, the PPO loss of k_loss, and the heads' backward in ONE kernel
// (replaces k_thin_fwd x2, k_loss, k_thin_bwd x2 and their launch gaps; 32 rows per block).
//   phase 1 (8 lanes per row): mean = Ha Wa^T + ba, value = Hc Wc^T + bc, loss terms, dL/dmean, dL/dvalue -> LDS
//   phase 2 (thread per hidden unit; threads 0..127 actor, 128..255 critic): Ga = (dmean Wa) * ELU'(Ha), Gc likewise,
//            head weight / bias gradient partials of the block's rows -> slabs (reduced by k_reduce_slabs*)
// partial[block][0..A+4) as k_loss.  Same formulas and summation structure as the separate kernels.
constexpr int HEAD_ROWS = 32;
// AT: compile-time number of actions (6: biped, 12: quadrupeds) or 16 = any A <= 16 with run-time masks; with the exact count
// the per-action loops carry no dead iterations (the kernel is VALU-bound: 2.1 k VALU instructions per wave with AT = 16).
template <int AT>
__global__ __launch_bounds__(256) void k_heads(int B, int A, int Ka, int Kc, float clip, float vcoef, float ecoef, int clipped_value,
                                               const float* __restrict__ Ha, const float* __restrict__ Hc,
                                               const float* __restrict__ Wa, const float* __restrict__ ba,
                                               const float* __restrict__ Wc, const float* __restrict__ bc,
                                               const float* __restrict__ stdp, const float* __restrict__ act,
                                               const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                                               const float* __restrict__ old_sigma, const float* __restrict__ adv,
                                               const float* __restrict__ ret, const float* __restrict__ old_val,
                                               float* __restrict__ Ga, float* __restrict__ Gc, float* __restrict__ slabsA,
                                               float* __restrict__ slabsC, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  float* sWa = hsm;                       // [A][Ka]
  float* sWc = sWa + A * Ka;              // [Kc]
  float* sD = sWc + Kc;                   // [HEAD_ROWS][A + 1]: dL/dmean, dL/dvalue of the block's rows
  float* sP = sD + HEAD_ROWS * (A + 1);   // [4 waves][A + 4] loss partials
  for (int i = threadIdx.x; i < A * Ka; i += 256) sWa[i] = Wa[i];
  for (int i = threadIdx.x; i < Kc; i += 256) sWc[i] = Wc[i];
  __syncthreads();
  const int lane8 = threadIdx.x & 7, lr = threadIdx.x >> 3;
  const int row = blockIdx.x * HEAD_ROWS + lr;
  const bool rv = row < B;
  const float invB = 1.0f / (float)B;
  // ---- phase 1: heads forward (8 lanes per row; dwordx4 row reads when the widths allow)
  float mu[AT], val = 0.0f;
#pragma unroll
  for (int o = 0; o < AT; ++o) mu[o] = 0.0f;
  if (rv) {
    if (((Ka | Kc) & 3) == 0) {
      const float4* xr = reinterpret_cast<const float4*>(Ha + (size_t)row * Ka);
      for (int k4 = lane8; k4 < (Ka >> 2); k4 += 8) {
        float4 x = xr[k4];
#pragma unroll
        for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) {
          float4 w = *reinterpret_cast<const float4*>(&sWa[o * Ka + 4 * k4]);
          mu[o] += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
        }
      }
      const float4* cr = reinterpret_cast<const float4*>(Hc + (size_t)row * Kc);
      for (int k4 = lane8; k4 < (Kc >> 2); k4 += 8) {
        float4 x = cr[k4], w = *reinterpret_cast<const float4*>(&sWc[4 * k4]);
        val += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
      }
    } else {
      for (int k = lane8; k < Ka; k += 8) {
        float x = Ha[(size_t)row * Ka + k];
#pragma unroll
        for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) mu[o] += x * sWa[o * Ka + k];
      }
      for (int k = lane8; k < Kc; k += 8) val += Hc[(size_t)row * Kc + k] * sWc[k];
    }
  }
#pragma unroll
  for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) {
    float v = mu[o];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    mu[o] = v + ba[o];
  }
  val += __shfl_xor(val, 1, 64); val += __shfl_xor(val, 2, 64); val += __shfl_xor(val, 4, 64);
  val += bc[0];
  // ---- loss (k_loss formulas): lane j of a row's 8-lane group does action dimensions j and j + 8 (the divisions and
  //      logs are the bulk of the instructions), the per-sample sums come back by a 3-step shuffle inside the group
  float dstd_own[2] = {0.0f, 0.0f}, kl = 0.0f, vl = 0.0f, sl = 0.0f, en = 0.0f;
  {
    const int i = rv ? row : 0;
    float lp = 0.0f, kls = 0.0f, ens = 0.0f, dmu_own[2] = {0.0f, 0.0f}, dsg_own[2] = {0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < (AT > 8 ? 2 : 1); ++kk) {
      const int k = lane8 + 8 * kk;
      if (rv && k < A) {
        float m = 0.0f;
#pragma unroll
        for (int o = 0; o < AT; ++o) if (o == k) m = mu[o];       // register select (mu[] is indexed by constants only)
        const float sg = stdp[k], a = act[(size_t)i * A + k];
        const float d = a - m;
        lp += -0.5f * (d * d) / (sg * sg) - logf(sg) - 0.91893853320467274178f;
        dmu_own[kk] = d / (sg * sg);
        dsg_own[kk] = (d * d) / (sg * sg * sg) - 1.0f / sg;
        const float om = old_mu[(size_t)i * A + k], os = old_sigma[(size_t)i * A + k];
        kls += logf(sg / os + 1e-5f) + (os * os + (om - m) * (om - m)) / (2.0f * sg * sg) - 0.5f;
        ens += 0.5f + 0.91893853320467274178f + logf(sg);
      }
    }
    lp += __shfl_xor(lp, 1, 64); lp += __shfl_xor(lp, 2, 64); lp += __shfl_xor(lp, 4, 64);
    kls += __shfl_xor(kls, 1, 64); kls += __shfl_xor(kls, 2, 64); kls += __shfl_xor(kls, 4, 64);
    ens += __shfl_xor(ens, 1, 64); ens += __shfl_xor(ens, 2, 64); ens += __shfl_xor(ens, 4, 64);
    float g_lp = 0.0f;
    if (rv) {
      const float ad = adv[i];
      const float ratio = expf(lp - old_logp[i]);
      const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
      g_lp = (s1 >= s2) ? (-ad * ratio) * invB : 0.0f;
      if (lane8 == 0) {
        sl = fmaxf(s1, s2);
        const float v = val, r = ret[i];
        float gv;
        if (clipped_value) {
          const float ov = old_val[i];
          const float vc = ov + fminf(fmaxf(v - ov, -clip), clip);
          const float e1 = (v - r) * (v - r), e2 = (vc - r) * (vc - r);
          const bool inside = fabsf(v - ov) < clip;
          gv = (e1 >= e2 || inside) ? 2.0f * (v - r) : 0.0f;
          vl = fmaxf(e1, e2);
        } else { gv = 2.0f * (v - r); vl = (r - v) * (r - v); }
        sD[lr * (A + 1) + A] = vcoef * gv * invB;
        kl = kls; en = ens;
      }
    } else if (lane8 == 0) sD[lr * (A + 1) + A] = 0.0f;            // rows past the batch contribute nothing
#pragma unroll
    for (int kk = 0; kk < (AT > 8 ? 2 : 1); ++kk) {
      const int k = lane8 + 8 * kk;
      if (k < A) { sD[lr * (A + 1) + k] = g_lp * dmu_own[kk]; dstd_own[kk] = g_lp * dsg_own[kk]; }
    }
  }
  // block partials: wave sums (fixed shuffle tree) -> LDS -> thread 0 adds the four waves in order
  {
    const int w = threadIdx.x >> 6;
    auto wsum = [](float v) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      return v;
    };
    // dL/dstd[k]: lanes with lane8 == k (mod 8) of the wave's 8 rows -> shuffle over the row bits only
#pragma unroll
    for (int kk = 0; kk < (AT > 8 ? 2 : 1); ++kk) {
      float t = dstd_own[kk];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      const int k = lane8 + 8 * kk;
      if ((threadIdx.x & 63) < 8 && k < A) sP[w * (A + 4) + k] = t;
    }
    float t;
    t = wsum(kl); if ((threadIdx.x & 63) == 0) sP[w * (A + 4) + A] = t;
    t = wsum(vl); if ((threadIdx.x & 63) == 0) sP[w * (A + 4) + A + 1] = t;
    t = wsum(sl); if ((threadIdx.x & 63) == 0) sP[w * (A + 4) + A + 2] = t;
    t = wsum(en); if ((threadIdx.x & 63) == 0) sP[w * (A + 4) + A + 3] = t;
  }
  __syncthreads();
  if (threadIdx.x < A + 4) {
    const int c = threadIdx.x;
    partial[(size_t)blockIdx.x * (A + 4) + c] = (sP[c] + sP[(A + 4) + c]) + (sP[2 * (A + 4) + c] + sP[3 * (A + 4) + c]);
  }
  // ---- phase 2: heads backward, thread per hidden unit
  const int r0 = blockIdx.x * HEAD_ROWS, nr = min(HEAD_ROWS, B - r0);
  if (threadIdx.x < 128) {
    float* sl_out = slabsA + (size_t)blockIdx.x * ((size_t)A * Ka + A);
    for (int k = threadIdx.x; k < Ka; k += 128) {
      float w[AT], dw[AT];
#pragma unroll
      for (int o = 0; o < AT; ++o) { w[o] = (AT < 16 || o < A) ? sWa[o * Ka + k] : 0.0f; dw[o] = 0.0f; }
      int r = 0;
      for (; r + 8 <= nr; r += 8) {
        float h[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { h[u] = Ha[(size_t)(r0 + r + u) * Ka + k]; g[u] = 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) { float d = sD[(r + u) * (A + 1) + o]; g[u] += d * w[o]; dw[o] += d * h[u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) Ga[(size_t)(r0 + r + u) * Ka + k] = g[u] * (h[u] > 0.0f ? 1.0f : h[u] + 1.0f);
      }
      for (; r < nr; ++r) {
        float h = Ha[(size_t)(r0 + r) * Ka + k], g = 0.0f;
#pragma unroll
        for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) { float d = sD[r * (A + 1) + o]; g += d * w[o]; dw[o] += d * h; }
        Ga[(size_t)(r0 + r) * Ka + k] = g * (h > 0.0f ? 1.0f : h + 1.0f);
      }
#pragma unroll
      for (int o = 0; o < AT; ++o) if (AT < 16 || o < A) sl_out[(size_t)o * Ka + k] = dw[o];
    }
    if (threadIdx.x < A) {        // bias gradient = column sums of dL/dmean over the block's rows
      float db = 0.0f;
      for (int r = 0; r < nr; ++r) db += sD[r * (A + 1) + threadIdx.x];
      sl_out[(size_t)A * Ka + threadIdx.x] = db;
    }
  } else {
    const int t = threadIdx.x - 128;
    float* sl_out = slabsC + (size_t)blockIdx.x * ((size_t)Kc + 1);
    for (int k = t; k < Kc; k += 128) {
      const float w = sWc[k];
      float dw = 0.0f;
      int r = 0;
      for (; r + 8 <= nr; r += 8) {
        float h[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = Hc[(size_t)(r0 + r + u) * Kc + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float d = sD[(r + u) * (A + 1) + A];
          dw += d * h[u];
          Gc[(size_t)(r0 + r + u) * Kc + k] = d * w * (h[u] > 0.0f ? 1.0f : h[u] + 1.0f);
        }
      }
      for (; r < nr; ++r) {
        float h = Hc[(size_t)(r0 + r) * Kc + k], d = sD[r * (A + 1) + A];
        dw += d * h;
        Gc[(size_t)(r0 + r) * Kc + k] = d * w * (h > 0.0f ? 1.0f : h + 1.0f);
      }
      sl_out[k] = dw;
    }
    if (t == 0) {
      float db = 0.0f;
      for (int r = 0; r < nr; ++r) db += sD[r * (A + 1) + A];
      sl_out[Kc] = db;
    }
  }
}
// finish the loss reductions: dL/dstd (incl. the entropy bonus), KL sum + count for the adaptive LR
__device__ float loss_finish_body(const LossFinish& f) {      // returns the square of the std gradient entry this thread wrote (else 0)
  const int nb = f.nb, A = f.A, B = f.B;
  const float ecoef = f.ecoef;
  const float* __restrict__ partial = f.partial; const float* __restrict__ stdp = f.stdp;
  float* __restrict__ dstd = f.dstd; float* __restrict__ grads_tail = f.grads_tail; float* __restrict__ scalars = f.scalars;
  // 32 columns x 8 row-lanes: lane (c, r) sums rows r, r+8, ... of column c; fixed-order combine in LDS
  __shared__ float sm[8][33];
  int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  float s = 0.0f;
  if (c < A + 4) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = 0.0f;
    int b = r;
    for (; b + 56 < nb; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += partial[(size_t)(b + 8 * u) * (A + 4) + c];
    }
    for (; b < nb; b += 8) t[0] += partial[(size_t)b * (A + 4) + c];
    s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  sm[r][c] = s;
  __syncthreads();
  float wrote = 0.0f;
  if (r == 0 && c < A + 4) {
    float t = 0.0f;
    for (int i = 0; i < 8; ++i) t += sm[i][c];
    if (c < A) { const float g = t - ecoef / stdp[c]; dstd[c] = g; wrote = g * g; }
    else if (c == A) { grads_tail[0] = t; grads_tail[1] = (float)B; }
    else if (c == A + 1) scalars[SC_VLOSS] += t / (float)B;
    else if (c == A + 2) scalars[SC_SLOSS] += t / (float)B;
    else scalars[SC_ENTROPY] = t / (float)B;
  }
  return wrote;
}
__global__ __launch_bounds__(256) void k_loss_finish(LossFinish f) { (void)loss_finish_body(f); }

// ------------------------------------------------------------------------------------- optimiser
__global__ void k_sumsq1(size_t n, const float* __restrict__ g, float scale, float* __restrict__ partial) {
  __shared__ float sm[8];
  float q = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float v = g[i] * scale; q += v * v; }
  q = block_sum(q, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = q;
}
// single thread: adaptive-KL learning rate (rsl_rl PPO.update), clip coefficient, Adam step counter
__global__ void k_prepare_step(int nb, const float* __restrict__ partial, const float* __restrict__ grads_tail, int adaptive,
                               float desired_kl, float max_grad_norm, float* __restrict__ scalars) {
  float q = 0.0f;
  for (int b = threadIdx.x; b < nb; b += 64) q += partial[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float norm = sqrtf(q);
  scalars[SC_GNORM] = norm;
  float coef = max_grad_norm / (norm + 1e-6f);
  scalars[SC_CLIPCOEF] = coef < 1.0f ? coef : 1.0f;
  float kl = grads_tail[0] / grads_tail[1];
  scalars[SC_KL] = kl;
  if (adaptive && desired_kl > 0.0f) {
    float lr = scalars[SC_LR];
    if (kl > desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
    else if (kl < desired_kl / 2.0f && kl > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
    scalars[SC_LR] = lr;
  }
  scalars[SC_STEP] += 1.0f;
  scalars[SC_NUPD] += 1.0f;
}
__global__ void k_adam(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       float gscale, float b1, float b2, float eps, const float* __restrict__ scalars) {
  float lr = scalars[SC_LR], coef = scalars[SC_CLIPCOEF] * gscale, t = scalars[SC_STEP];
  float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  float step = lr / bc1, ibc2s = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) * ibc2s + eps);
  }
}

// One process (no gradient all-reduce between the backward pass and the optimiser): grad-norm clip, adaptive-KL learning rate and
// Adam in ONE launch.  Every block adds up the norm's first stage (k_reduce_slabs_n's per-block squares, fixed order: every block
// gets the same bits), takes the learning-rate decision for itself from the scalars of the PREVIOUS step, and steps its share of
// the parameters; the block that finishes last (ticket) publishes the new scalars -- by then every block has read the old ones.
__global__ __launch_bounds__(256) void k_adam_fused(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    int nsq, const float* __restrict__ sq_partial, const float* __restrict__ grads_tail, int adaptive,
                                                    float desired_kl, float max_grad_norm, float b1, float b2, float eps, float* __restrict__ scalars,
                                                    unsigned* __restrict__ ticket) {
  __shared__ float sm[8];
  __shared__ bool last;
  float q = 0.0f;
  for (int b = threadIdx.x; b < nsq; b += blockDim.x) q += sq_partial[b];
  q = block_sum(q, sm);
  const float norm = sqrtf(q);
  float coef = max_grad_norm / (norm + 1e-6f);
  coef = coef < 1.0f ? coef : 1.0f;
  const float kl = grads_tail[0] / grads_tail[1];
  float lr = scalars[SC_LR];
  if (adaptive && desired_kl > 0.0f) {
    if (kl > desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
    else if (kl < desired_kl / 2.0f && kl > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
  }
  const float t = scalars[SC_STEP] + 1.0f, nupd = scalars[SC_NUPD] + 1.0f;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step = lr / bc1, ibc2s = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) * ibc2s + eps);
  }
  __syncthreads();      // every thread of this block has read the old scalars (and is done with them)
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    scalars[SC_GNORM] = norm; scalars[SC_CLIPCOEF] = coef; scalars[SC_KL] = kl; scalars[SC_LR] = lr; scalars[SC_STEP] = t; scalars[SC_NUPD] = nupd;
    *ticket = 0u;
  }
}

}  // namespace

// ============================================================================================ host side
struct PpoAlg {
  PpoConfig cfg;
  PpoBuffers b;
  Net actor, critic;
  size_t std_off = 0, nparams = 0, early_off = 0;     // early_off: first float of the EARLY gradient group (see ppo_plan)
  int device = 0;
  bool bound = false;
  int batch = 0;           // minibatch rows
  // workspace carve (floats)
  size_t w_xa, w_xc, w_act, w_oldlogp, w_oldmu, w_oldsig, w_adv, w_ret, w_oldval;
  int ld_xa = 0, ld_xc = 0;   // row strides of the minibatch observation copies: widths rounded up to 4 floats (16-byte rows: dwordx4 loads)
  size_t w_perm[9];        // the nine storage arrays of an update in minibatch order (all T*N rows): ppo_begin_update
  bool planes_valid = false;   // the bf16 weight planes match d_params as this handle last saw them (cleared by bind / optimiser step)
  bool permuted = false;   // w_perm holds the current permutation (set by ppo_begin_update, dropped by the next rollout)
  size_t w_ha[PPO_MAX_LAYERS], w_hc[PPO_MAX_LAYERS];   // activations per layer output (last = head output)
  size_t w_wp[2][PPO_MAX_LAYERS], w_wtp[2][PPO_MAX_LAYERS];      // bf16 planes of W (and W^T) per hidden layer, in floats of workspace
  size_t w_dmean, w_dvalue, w_g[2][PPO_MAX_LAYERS], w_slabs[2], w_slabl[2][PPO_MAX_LAYERS + 1], w_partial, w_sqpart, w_lastval, w_total;
  int sq_blocks = 0;       // blocks of the last whole-pass slab reduction that left gradient-norm partials (0: none)
  bool ticket_zeroed = false;   // [0] actor / [1] critic scratch (run concurrently)
  hipStream_t aux = nullptr;        // second stream: the critic network runs beside the actor network
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool dual = true, fused_heads = true;
  bool paired = true;      // actor + critic layer i in one launch (needs fused_heads and equal depth); else two streams
  bool dma = true;         // bf16x6 forward / data-gradient GEMMs on the LDS-DMA kernel (pf_gemm3f.h) where shapes allow
  bool split3 = true;      // dense layers on the bf16 matrix pipes by exact three-way operand split (pf_gemm3.h); false: fp32 MFMA
  int max_rows = 0, max_width = 0, splits = 0;
  bool pair_rollout = true;   // ppo_act: actor and critic layer i in one launch (false: two chains on two streams; PF_PPO_PAIRED_ROLLOUT=0, diagnostic)
  bool l0_fp32 = false;    // rollout-sized launches (M <= 8192) of the input layers on the fp32 MFMA kernels (PF_PPO_L0_FP32=1: diagnostic)
  bool pend = false;       // a deferred process_env_step waits for the next ppo_act / ppo_compute_returns
  ProcArgs pend_args{};
};

// layers [lo, hi) of one network into the flat buffer
static void net_layout(Net& n, const int* dims, int nl, size_t& off, int lo, int hi) {
  n.nl = nl;
  for (int i = lo; i < hi; ++i) {
    n.l[i].in = dims[i]; n.l[i].out = dims[i + 1];
    // every tensor starts on a 16-byte boundary: the GEMMs read weight rows with dwordx4 loads
    off = (off + 3) & ~(size_t)3; n.l[i].w = off; off += (size_t)dims[i] * dims[i + 1];
    off = (off + 3) & ~(size_t)3; n.l[i].b = off; off += dims[i + 1];
  }
}

static int ppo_check_cfg(const PpoConfig* c) {
  if (!c) { pf_set_error("ppo: null config"); return 1; }
  if (c->n_actor_layers < 2 || c->n_actor_layers > PPO_MAX_LAYERS || c->n_critic_layers < 2 || c->n_critic_layers > PPO_MAX_LAYERS) {
    pf_set_error("ppo: 1..3 hidden layers supported"); return 1;
  }
  if (c->num_actions < 1 || c->num_actions > 16) { pf_set_error("ppo: num_actions must be in 1..16"); return 1; }
  if (c->actor_dims[0] != c->num_obs || c->critic_dims[0] != c->num_critic_obs || c->actor_dims[c->n_actor_layers] != c->num_actions ||
      c->critic_dims[c->n_critic_layers] != 1) { pf_set_error("ppo: layer dims inconsistent with obs/action sizes"); return 1; }
  // rsl_rl floors the minibatch size (mini_batch_size = batch_size // num_mini_batches): the last (N*T) % num_mini_batches
  // entries of the permutation are not used in that update
  if (c->num_mini_batches < 1 || (long long)c->num_envs * c->num_steps < c->num_mini_batches) { pf_set_error("ppo: num_mini_batches must be in 1..num_envs*num_steps"); return 1; }
  {
    long long rows = (long long)c->num_envs * c->num_steps / c->num_mini_batches;
    long long widest = c->num_obs > c->num_critic_obs ? c->num_obs : c->num_critic_obs;
    if (rows * widest >= (1ll << 31)) { pf_set_error("ppo: minibatch rows * obs width must stay below 2^31"); return 1; }
  }
  for (int i = 1; i < c->n_actor_layers; ++i) if (c->actor_dims[i] & 3) { pf_set_error("ppo: hidden widths must be multiples of 4"); return 1; }
  for (int i = 1; i < c->n_critic_layers; ++i) if (c->critic_dims[i] & 3) { pf_set_error("ppo: hidden widths must be multiples of 4"); return 1; }
  for (int i = 1; i < c->n_actor_layers; ++i) if (c->actor_dims[i] > 1024) { pf_set_error("ppo: hidden width > 1024"); return 1; }
  for (int i = 1; i < c->n_critic_layers; ++i) if (c->critic_dims[i] > 1024) { pf_set_error("ppo: hidden width > 1024"); return 1; }
  return 0;
}

static void ppo_plan(PpoAlg* a) {
  const PpoConfig& c = a->cfg;
  // Flat parameter / gradient layout, in the order the backward pass FINISHES gradients, last first: the input layers of
  // both networks (their gradients come last: the LATE all-reduce group), then every other layer, the action std and -- in the
  // gradient buffer -- the 4-float KL tail (the EARLY group, complete before the last two GEMM launches of a backward pass).
  // Each group is one contiguous range (ppo_grad_groups), so the multi-GPU path needs two all-reduce calls per minibatch.
  size_t off = 0;
  net_layout(a->actor, c.actor_dims, c.n_actor_layers, off, 0, 1);
  net_layout(a->critic, c.critic_dims, c.n_critic_layers, off, 0, 1);
  off = (off + 3) & ~(size_t)3;
  a->early_off = off;
  net_layout(a->actor, c.actor_dims, c.n_actor_layers, off, 1, c.n_actor_layers);
  net_layout(a->critic, c.critic_dims, c.n_critic_layers, off, 1, c.n_critic_layers);
  off = (off + 3) & ~(size_t)3;
  a->std_off = off; off += c.num_actions;
  a->nparams = (off + 3) & ~(size_t)3;
  a->batch = c.num_envs * c.num_steps / c.num_mini_batches;
  a->max_rows = a->batch > c.num_envs ? a->batch : c.num_envs;
  int mw = 0; size_t maxw = 0;
  for (int i = 0; i < a->actor.nl; ++i) { mw = std::max(mw, a->actor.l[i].out); maxw = std::max(maxw, (size_t)a->actor.l[i].in * a->actor.l[i].out); }
  for (int i = 0; i < a->critic.nl; ++i) { mw = std::max(mw, a->critic.l[i].out); maxw = std::max(maxw, (size_t)a->critic.l[i].in * a->critic.l[i].out); }
  a->max_width = mw;
  a->splits = 128;
  size_t R = a->max_rows, w = 0;
  auto take = [&](size_t n) { size_t at = w; w += (n + 63) & ~(size_t)63; return at; };
  a->ld_xa = (c.num_obs + 3) & ~3; a->ld_xc = (c.num_critic_obs + 3) & ~3;
  a->w_xa = take(R * a->ld_xa); a->w_xc = take(R * a->ld_xc); a->w_act = take(R * c.num_actions);
  a->w_oldlogp = take(R); a->w_oldmu = take(R * c.num_actions); a->w_oldsig = take(R * c.num_actions);
  a->w_adv = take(R); a->w_ret = take(R); a->w_oldval = take(R);
  {
    const size_t TN = (size_t)c.num_envs * c.num_steps;
    const int cols[9] = {a->ld_xa, a->ld_xc, c.num_actions, 1, c.num_actions, c.num_actions, 1, 1, 1};
    for (int i = 0; i < 9; ++i) a->w_perm[i] = take(TN * cols[i]);
  }
  for (int i = 0; i < a->actor.nl; ++i) a->w_ha[i] = take(R * a->actor.l[i].out);
  for (int i = 0; i < a->critic.nl; ++i) a->w_hc[i] = take(R * a->critic.l[i].out);
  a->w_dmean = take(R * c.num_actions); a->w_dvalue = take(R);
  for (int k = 0; k < 2; ++k) for (int l = 0; l < PPO_MAX_LAYERS; ++l) a->w_g[k][l] = take(R * mw);
  size_t thin_blocks = (R + 31) / 32;
  for (int k = 0; k < 2; ++k) a->w_slabs[k] = take(std::max<size_t>((size_t)a->splits * (maxw + mw), thin_blocks * (16 * 1024 + 16)));
  // paired mode: every layer keeps its own slabs so that all reductions of a backward pass run as one launch at its end
  for (int k = 0; k < 2; ++k) {
    const Net& n = k == 0 ? a->actor : a->critic;
    for (int i = 0; i < n.nl; ++i) {
      const Layer& L = n.l[i];
      size_t nw = (size_t)L.out * L.in + L.out;
      if (i == n.nl - 1) a->w_slabl[k][i] = take(thin_blocks * nw);
      else {
        int tile, S, S3;
        pf_pick_split(L.out, L.in, (int)R, a->splits, &tile, &S);
        pf_gemm3_pick_split(L.out, L.in, (int)R, a->splits, &tile, &S3);
        a->w_slabl[k][i] = take((size_t)(std::max(S, S3) + 1) * nw);
      }
    }
  }
  for (int k = 0; k < 2; ++k) {
    const Net& n = k == 0 ? a->actor : a->critic;
    for (int i = 0; i + 1 < n.nl; ++i) {
      const Layer& L = n.l[i];
      const size_t inp = (size_t)((L.in + 7) & ~7), outp = (size_t)((L.out + 7) & ~7);
      a->w_wp[k][i] = take((3 * (size_t)L.out * inp + 1) / 2 + 8);       // bf16 elements -> floats
      a->w_wtp[k][i] = take((3 * (size_t)L.in * outp + 1) / 2 + 8);
    }
  }
  a->w_partial = take(std::max<size_t>((size_t)RED_BLOCKS * 24, thin_blocks * 20));
  a->w_sqpart = take(PPO_SQ_PARTIALS + 4);      // per-block squares of the slab-reduction launch (+ the fused optimiser's ticket word)
  a->w_lastval = take(c.num_envs);
  a->w_total = w;
}

// bf16 planes of every hidden layer's W and W^T (pf_gemm3.h: the B operand of the forward and data-gradient GEMMs is staged by
// plain copies instead of being split again by every block).  Re-done at the head of every entry point that runs a dense layer,
// so the planes can never be stale whoever wrote the parameters (optimiser step, checkpoint load, broadcast): ~4 us per launch.
static void split_weights(PpoAlg* a, hipStream_t s) {
  if (!a->split3) return;
  SplitJobs js; js.n = 0;
  int blocks = 0;
  for (int k = 0; k < 2; ++k) {
    const Net& n = k == 0 ? a->actor : a->critic;
    for (int i = 0; i + 1 < n.nl; ++i) {
      const Layer& L = n.l[i];
      const int inp = (L.in + 7) & ~7, outp = (L.out + 7) & ~7;
      SplitJob w{a->b.d_params + L.w, reinterpret_cast<unsigned short*>(a->b.d_workspace + a->w_wp[k][i]), L.out, L.in, L.in, inp, 0, (long long)L.out * inp, 0};
      w.blocks = (int)(((long long)w.rows * w.ldp + 255) / 256);
      js.j[js.n++] = w; blocks += w.blocks;
      if (i > 0) {
        SplitJob t{a->b.d_params + L.w, reinterpret_cast<unsigned short*>(a->b.d_workspace + a->w_wtp[k][i]), L.in, L.out, L.in, outp, 1, (long long)L.in * outp, 0};
        t.blocks = (int)(((long long)t.rows * t.ldp + 255) / 256);
        js.j[js.n++] = t; blocks += t.blocks;
      }
    }
  }
  hipLaunchKernelGGL(k_split_weights, dim3(blocks), dim3(256), 0, s, js);
  a->planes_valid = true;
}
// B operand of layer i as planes: the forward GEMM reads W [out][in], the data-gradient GEMM W^T [in][out]
static void use_planes(PpoAlg* a, GemmArgs& g, int net, int layer, bool transposed) {
  if (!a->split3) return;
  const Layer& L = (net == 0 ? a->actor : a->critic).l[layer];
  const int inp = (L.in + 7) & ~7, outp = (L.out + 7) & ~7;
  if (!transposed) { g.Bp = reinterpret_cast<const unsigned short*>(a->b.d_workspace + a->w_wp[net][layer]); g.ldbp = inp; g.bplane = (long long)L.out * inp; }
  else { g.Bp = reinterpret_cast<const unsigned short*>(a->b.d_workspace + a->w_wtp[net][layer]); g.ldbp = outp; g.bplane = (long long)L.in * outp; }
}

// First layer on a zero-padded input copy (row stride ldx, a multiple of 4): with the weights as bf16 planes (rows zero padded
// to 8) the reduction simply runs over the padded width and every A load is a dwordx4; the fp32 B operand has no padding, so
// that path keeps K and only takes the stride.
static void pad_input(GemmArgs& g, int ldx) {
  g.lda = ldx;
  if (g.Bp && ldx <= g.ldbp) g.K = ldx;
}
// forward through one MLP: X[M, in0] -> activations; hidden layers on the MFMA GEMM (bias + ELU fused), head on the thin kernel
// ldx: row stride of X when it is a zero-padded copy (0: rows are exactly in0 wide)
static void mlp_forward(PpoAlg* a, const Net& n, const float* X, int M, const size_t* w_h, hipStream_t s, bool with_head = true, int ldx = 0) {
  float* ws = a->b.d_workspace; const float* P = a->b.d_params;
  const float* in = X;
  for (int i = 0; i < n.nl; ++i) {
    const Layer& L = n.l[i];
    float* out = ws + w_h[i];
    if (i == n.nl - 1) {
      if (!with_head) break;          // the minibatch path runs both heads inside k_heads
      int threads = 256, rows_per_block = threads / 8;
      hipLaunchKernelGGL((k_thin_fwd<16>), dim3((M + rows_per_block - 1) / rows_per_block), dim3(threads), sizeof(float) * L.out * L.in, s,
                         M, L.in, L.out, in, P + L.w, P + L.b, out);
    } else {
      GemmArgs g{};
      g.A = in; g.B = P + L.w; g.C = out; g.M = M; g.N = L.out; g.K = L.in; g.lda = L.in; g.ldb = L.in; g.ldc = L.out; g.bias = P + L.b;
      use_planes(a, g, &n == &a->actor ? 0 : 1, i, false);
      if (i == 0 && ldx > 0) pad_input(g, ldx);
      if (a->split3 && a->dma && pf_gemm3f_ok<GEMM_EPI_BIAS_ELU>(g)) pf_gemm3f_launch<GEMM_EPI_BIAS_ELU>(g, s);
      else if (a->split3) pf_gemm3_launch<true, true, GEMM_EPI_BIAS_ELU>(g, 0, s);
      else pf_gemm_auto<true, true, GEMM_EPI_BIAS_ELU>(g, 0, s);
    }
    in = out;
  }
}

// backward through one MLP given dHead[M, out_last]; writes weight/bias gradients into d_grads
// (dHead == nullptr: k_heads already wrote g[last-1] and the head's gradient slabs; only their reduction is left)
static void mlp_backward(PpoAlg* a, const Net& n, const float* X, int M, const size_t* w_h, const float* dHead, hipStream_t s, int set, int ldx = 0) {
  float* ws = a->b.d_workspace; const float* P = a->b.d_params; float* G = a->b.d_grads;
  float* slabs = ws + a->w_slabs[set];
  const int last = n.nl - 1;
  // Every layer has its own gradient buffer g[i] = dL/dz_i.  (Putting the weight-gradient GEMMs on further side streams
  // was measured and adds nothing: 20.5 vs 20.1 ms per update at the time.)
  auto g_of = [&](int i) { return ws + a->w_g[set][i]; };
  // thin head: g[last-1] = dZ_{last-1}, dW_last, db_last (one kernel), then its slab reduce
  {
    const Layer& L = n.l[last];
    int rows_per_block = 32, nblk = (M + rows_per_block - 1) / rows_per_block;
    int threads = (L.in + 63) / 64 * 64;
    float* hslabs = slabs;      // head slabs share the buffer: the stream runs in order
    if (dHead)
      hipLaunchKernelGGL((k_thin_bwd<16>), dim3(nblk), dim3(threads), 0, s, M, L.in, L.out, rows_per_block, dHead, ws + w_h[last - 1], P + L.w,
                         g_of(last - 1), hslabs);
    int nw = L.out * L.in + L.out;       // W then b, contiguous in the flat gradient buffer
    reduce_slabs(nw, nblk, hslabs, G + L.w, s);
  }
  for (int i = last - 1; i >= 0; --i) {
    const Layer& L = n.l[i];
    const float* in = i == 0 ? X : ws + w_h[i - 1];
    const float* gcur = g_of(i);
    if (i > 0) {
      // dZ_{i-1} = (gcur[M,out] * W_i[out,in]) * ELU'(h_{i-1})
      GemmArgs g{};
      g.A = gcur; g.B = P + L.w; g.C = g_of(i - 1); g.M = M; g.N = L.in; g.K = L.out; g.lda = L.out; g.ldb = L.in; g.ldc = L.in;
      g.H = ws + w_h[i - 1]; g.ldh = L.in;
      if (a->split3) {      // B = W^T planes, k (= out) contiguous
        use_planes(a, g, &n == &a->actor ? 0 : 1, i, true);
        if (a->dma && pf_gemm3f_ok<GEMM_EPI_MUL_DELU>(g)) pf_gemm3f_launch<GEMM_EPI_MUL_DELU>(g, s);
        else pf_gemm3_launch<true, true, GEMM_EPI_MUL_DELU>(g, 0, s);
      }
      else pf_gemm_auto<true, false, GEMM_EPI_MUL_DELU>(g, 0, s);
    }
    // dW_i = gcur[M, out]^T * in[M, in] and db_i = column sums of gcur, both as split-K slabs [S][out*in + out]
    // (deterministic slab reduce straight into the flat gradient: W then b)
    {
      GemmArgs g{};
      g.A = gcur; g.B = in; g.C = slabs; g.M = L.out; g.N = L.in; g.K = M; g.lda = L.out; g.ldb = (i == 0 && ldx > 0) ? ldx : L.in; g.ldc = L.in;
      int tile, want;
      if (a->split3) pf_gemm3_pick_split(L.out, L.in, M, a->splits, &tile, &want);
      else pf_pick_split(L.out, L.in, M, a->splits, &tile, &want);
      int per = ((M + want - 1) / want + 31) / 32 * 32;
      int S = (M + per - 1) / per;
      g.k_per_split = per;
      if (a->split3) pf_gemm3_launch<false, false, GEMM_EPI_NONE, true>(g, S, s, tile);
      else if (tile == 0) pf_gemm_launch<128, 128, false, false, GEMM_EPI_NONE, true>(g, S, s);
      else if (tile == 1) pf_gemm_launch<128, 64, false, false, GEMM_EPI_NONE, true>(g, S, s);
      else pf_gemm_launch<64, 64, false, false, GEMM_EPI_NONE, true>(g, S, s);
      int nw = L.out * L.in + L.out;
      reduce_slabs(nw, S, slabs, G + L.w, s);
    }
  }
}

// kernels launch on the calling thread's current device: it has to be the one the handle was created for
static bool ppo_wrong_device(const PpoAlg* a) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == a->device) return false;
  pf_set_error("ppo: the calling thread's current HIP device is not the algorithm's device (torch.cuda.set_device)");
  return true;
}

// ---- paired execution: layer i of the actor and of the critic go out as ONE launch (pf_gemm_kernel2 / k_reduce_slabs2).
// Same overlap as two streams -- one problem's tail blocks are filled by the other's -- without cross-stream events
// (each costs ~10 us on the critical path) and with half the launches.  Needs the same number of layers in both nets.
static GemmArgs fwd_args(PpoAlg* a, const Net& n, int i, const float* in, const size_t* w_h, int M) {
  const Layer& L = n.l[i];
  GemmArgs g{};
  g.A = in; g.B = a->b.d_params + L.w; g.C = a->b.d_workspace + w_h[i]; g.M = M; g.N = L.out; g.K = L.in; g.lda = L.in; g.ldb = L.in; g.ldc = L.out;
  g.bias = a->b.d_params + L.b;
  return g;
}
static void mlp_forward_pair(PpoAlg* a, const float* Xa, const float* Xc, int M, hipStream_t s, int ldxa = 0, int ldxc = 0) {
  float* ws = a->b.d_workspace;
  const float *ina = Xa, *inc = Xc;
  for (int i = 0; i + 1 < a->actor.nl; ++i) {       // hidden layers; the heads run in k_heads / k_act_heads
    GemmArgs ga = fwd_args(a, a->actor, i, ina, a->w_ha, M), gc = fwd_args(a, a->critic, i, inc, a->w_hc, M);
    use_planes(a, ga, 0, i, false); use_planes(a, gc, 1, i, false);
    if (i == 0 && ldxa > 0) pad_input(ga, ldxa);
    if (i == 0 && ldxc > 0) pad_input(gc, ldxc);
    if (i == 0 && a->l0_fp32 && M <= 8192) { ga.Bp = gc.Bp = nullptr; pf_gemm_auto2<true, true, GEMM_EPI_BIAS_ELU>(gc, 0, ga, 0, s); }
    else if (a->split3 && a->dma && pf_gemm3f_ok<GEMM_EPI_BIAS_ELU>(gc) && pf_gemm3f_ok<GEMM_EPI_BIAS_ELU>(ga)) pf_gemm3f_launch2<GEMM_EPI_BIAS_ELU>(gc, ga, s);
    else if (a->split3) pf_gemm3_launch2<true, true, GEMM_EPI_BIAS_ELU>(gc, 0, ga, 0, s);
    else pf_gemm_auto2<true, true, GEMM_EPI_BIAS_ELU>(gc, 0, ga, 0, s);     // the larger problem (critic) first
    ina = ws + a->w_ha[i]; inc = ws + a->w_hc[i];
  }
}
// part: -1 = the whole pass with ONE slab-reduction launch at its end (single GPU).  0 / 1 = the pass in two parts for the
// multi-GPU path, each ending with the reduction of the gradients it has finished, so that the all-reduce of the first
// group runs under the kernels of the second (ppo_minibatch_backward_part):
//   part 0: data gradients down to g_1, weight gradients of every layer but the input layer (heads included) -> EARLY group
//           [actor / critic layers >= 1, std, KL tail]; the loss scalars are finished here;
//   part 1: data gradient into layer 0's output, the input layers' weight gradients -> LATE group [actor L0, critic L0].
// The flat parameter / gradient buffer is laid out group by group (ppo_plan), so each group is ONE contiguous range.
static void mlp_backward_pair(PpoAlg* a, const float* Xa, const float* Xc, int M, hipStream_t s, int ldxa = 0, int ldxc = 0, const LossFinish* lf = nullptr, int part = -1, bool want_sq = false) {
  const int ldx[2] = {ldxa, ldxc};
  float* ws = a->b.d_workspace; const float* P = a->b.d_params; float* G = a->b.d_grads;
  const Net* net[2] = {&a->actor, &a->critic};
  const size_t* w_h[2] = {a->w_ha, a->w_hc};
  const float* X[2] = {Xa, Xc};
  const int last = a->actor.nl - 1;
  auto g_of = [&](int set, int i) { return ws + a->w_g[set][i]; };
  ReduceJobs js; js.n = 0; js.lf = LossFinish{}; js.sq_partial = nullptr;
  auto bwd_x = [&](int i) {      // dZ_{i-1} = (g_i W_i) * ELU'(h_{i-1})
    GemmArgs g[2];
    for (int k = 0; k < 2; ++k) {
      const Layer& L = net[k]->l[i];
      g[k] = GemmArgs{};
      g[k].A = g_of(k, i); g[k].B = P + L.w; g[k].C = g_of(k, i - 1); g[k].M = M; g[k].N = L.in; g[k].K = L.out;
      g[k].lda = L.out; g[k].ldb = L.in; g[k].ldc = L.in; g[k].H = ws + w_h[k][i - 1]; g[k].ldh = L.in;
    }
    if (a->split3) {      // B = W^T planes, k (= out) contiguous
      use_planes(a, g[0], 0, i, true); use_planes(a, g[1], 1, i, true);
      if (a->dma && pf_gemm3f_ok<GEMM_EPI_MUL_DELU>(g[1]) && pf_gemm3f_ok<GEMM_EPI_MUL_DELU>(g[0])) pf_gemm3f_launch2<GEMM_EPI_MUL_DELU>(g[1], g[0], s);
      else pf_gemm3_launch2<true, true, GEMM_EPI_MUL_DELU>(g[1], 0, g[0], 0, s);
    }
    else pf_gemm_auto2<true, false, GEMM_EPI_MUL_DELU>(g[1], 0, g[0], 0, s);
  };
  auto bwd_w = [&](int i) {
    // dW_i, db_i as split-K slabs, then the deterministic slab reduce straight into the flat gradient
    GemmArgs g[2]; int S[2], tiles3[2] = {0, 0}, want[2]; ReduceJob j[2];
    // the two networks' layer-i gradients have the same shape except at the input layer (27 vs 148 columns): both go out in
    // ONE launch either way.  Equal shapes split the reduction half as often; unequal ones take a common tile (128 x 64, or
    // 64 x 64 for narrow layers) and share ~512 workgroups in proportion to their tile counts.
    const bool same = net[0]->l[i].out == net[1]->l[i].out && net[0]->l[i].in == net[1]->l[i].in;
    for (int k = 0; k < 2; ++k) {
      const Layer& L = net[k]->l[i];
      if (a->split3) pf_gemm3_pick_split_shared(L.out, L.in, M, a->splits, same ? 2 : 1, &tiles3[k], &want[k]);
      else pf_pick_split(L.out, L.in, M, a->splits, &tiles3[k], &want[k]);
    }
    if (a->split3 && tiles3[0] != tiles3[1]) {
      const int tc = std::min(net[0]->l[i].out, net[1]->l[i].out) > 64 ? 1 : 3, bm = tc == 1 ? 128 : 64, bn = 64;
      int tot = 0;
      for (int k = 0; k < 2; ++k) tot += ((net[k]->l[i].out + bm - 1) / bm) * ((net[k]->l[i].in + bn - 1) / bn);
      int Sc = std::max(1, std::min(std::min(want[0], want[1]), (512 + tot - 1) / tot));      // never more slabs than either layer's buffer holds
      while (Sc > 1 && (M + Sc - 1) / Sc < 128) --Sc;
      tiles3[0] = tiles3[1] = tc; want[0] = want[1] = Sc;
    }
    for (int k = 0; k < 2; ++k) {
      const Layer& L = net[k]->l[i];
      g[k] = GemmArgs{};
      g[k].A = g_of(k, i); g[k].B = i == 0 ? X[k] : ws + w_h[k][i - 1]; g[k].C = ws + a->w_slabl[k][i];
      g[k].M = L.out; g[k].N = L.in; g[k].K = M; g[k].lda = L.out; g[k].ldb = (i == 0 && ldx[k] > 0) ? ldx[k] : L.in; g[k].ldc = L.in;
      int per = ((M + want[k] - 1) / want[k] + 31) / 32 * 32;
      S[k] = (M + per - 1) / per;
      g[k].k_per_split = per;
      j[k] = reduce_plan(L.out * L.in + L.out, S[k], ws + a->w_slabl[k][i], G + L.w);
    }
    if (a->split3) {
      if (tiles3[0] == tiles3[1]) pf_gemm3_tn_launch2(g[1], S[1], g[0], S[0], tiles3[0], s);
      else {
        pf_gemm3_launch<false, false, GEMM_EPI_NONE, true>(g[1], S[1], s, tiles3[1]);
        pf_gemm3_launch<false, false, GEMM_EPI_NONE, true>(g[0], S[0], s, tiles3[0]);
      }
    } else pf_gemm_launch2<false, false, GEMM_EPI_NONE, true>(g[1], S[1], g[0], S[0], s);
    js.j[js.n++] = j[1]; js.j[js.n++] = j[0];
  };
  auto head_jobs = [&]() {      // head gradient slabs written by k_heads; the loss scalars ride in the same launch
    const int nblk = (M + HEAD_ROWS - 1) / HEAD_ROWS;
    for (int k = 0; k < 2; ++k) { const Layer& L = net[k]->l[last]; js.j[js.n++] = reduce_plan(L.out * L.in + L.out, nblk, ws + a->w_slabl[k][last], G + L.w); }
    if (lf) js.lf = *lf;
  };
  if (part < 0) {
    head_jobs();
    for (int i = last - 1; i >= 0; --i) { if (i > 0) bwd_x(i); bwd_w(i); }
    a->sq_blocks = 0;
    if (want_sq && reduce_slabs_n_blocks(js) <= PPO_SQ_PARTIALS) { js.sq_partial = ws + a->w_sqpart; a->sq_blocks = reduce_slabs_n_blocks(js); }
    reduce_slabs_n(js, s);      // every layer has its own slabs: one reduction launch for the whole pass
  } else if (part == 0) {
    head_jobs();
    for (int i = last - 1; i >= 2; --i) { bwd_x(i); bwd_w(i); }
    if (last - 1 >= 1) bwd_w(1);      // g_1 exists (from bwd_x(2), or from k_heads when there are two hidden layers)
    reduce_slabs_n(js, s);
  } else {
    if (last - 1 >= 1) bwd_x(1);
    bwd_w(0);
    reduce_slabs_n(js, s);
  }
}

// The actor and the critic are independent networks: run the critic on a second stream so one network's launch tails,
// prologues and epilogues overlap the other's main loops (both are matrix-pipe bound; the gain is in the gaps).
static hipStream_t ppo_fork(PpoAlg* a, hipStream_t s) {
  if (!a->dual || !a->aux) return s;
  (void)hipEventRecord(a->ev_fork, s);
  (void)hipStreamWaitEvent(a->aux, a->ev_fork, 0);
  return a->aux;
}
static void ppo_join(PpoAlg* a, hipStream_t s, hipStream_t other) {
  if (other == s) return;
  (void)hipEventRecord(a->ev_join, other);
  (void)hipStreamWaitEvent(s, a->ev_join, 0);
}

extern "C" {

#if defined(G3_TN_STAMPS)
/* diagnostic build only (tools/gemm_tn_stamps.py; not declared in any header): read and clear the weight-gradient kernel's phase counters */
int pf_debug_gemm_tn_stamps(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g3_tn_stamp_acc), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g3_tn_stamp_acc), z, sizeof(z)) != hipSuccess;
}
#endif

/* Development / test hook (not part of the drop-in boundary: declared in include/pointfoot_hip_dev.h): run one GEMM of the PPO
 * kernel family.  mode 0: C[M,N] = A[M,K] B[N,K]^T + bias, ELU; 1: C[M,N] = A[M,K] B[K,N] * ELU'(H); 2: C[M,N] = A[K,M]^T B[K,N]
 * (split-K slabs into `ws`, then reduced into C with the column sums of A in C[M*N .. M*N+M)).  tile: -1 auto, 0/1/2. */
int pf_gemm_test(int mode, int M, int N, int K, const float* A, const float* B, float* Cm, const float* bias, const float* H,
                 float* ws, int splits, int tile, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemmArgs g{};
  g.A = A; g.B = B; g.C = Cm; g.M = M; g.N = N; g.K = K; g.bias = bias; g.H = H; g.ldh = N; g.ldc = N;
  auto run = [&](auto kc_a, auto kc_b, auto epi, auto cs, int S) {
    constexpr bool AK = decltype(kc_a)::value, BK_ = decltype(kc_b)::value, CS = decltype(cs)::value;
    constexpr int E = decltype(epi)::value;
    int t = tile >= 0 ? tile : pf_pick_tile(M, N, K, S);
    if (t == 0) pf_gemm_launch<128, 128, AK, BK_, E, CS>(g, S, s);
    else if (t == 1) pf_gemm_launch<128, 64, AK, BK_, E, CS>(g, S, s);
    else pf_gemm_launch<64, 64, AK, BK_, E, CS>(g, S, s);
  };
  using T = std::true_type; using F = std::false_type;
  if (mode >= 30) {      // bf16x6 with LDS-DMA staging (pf_gemm3f.h); tile: -1 auto, 0 = 128 columns per block, 1 = 64
    const int Kp = (K + 7) & ~7;
    SplitJobs js; js.n = 1;
    js.j[0] = SplitJob{B, reinterpret_cast<unsigned short*>(ws), N, K, mode == 30 ? K : N, Kp, mode == 30 ? 0 : 1, (long long)N * Kp, 0};
    js.j[0].blocks = (int)(((long long)N * Kp + 255) / 256);
    hipLaunchKernelGGL(k_split_weights, dim3(js.j[0].blocks), dim3(256), 0, s, js);
    g.lda = K; g.ldb = K; g.Bp = reinterpret_cast<const unsigned short*>(ws); g.ldbp = Kp; g.bplane = (long long)N * Kp;
    const int bn = tile >= 0 ? tile : -1;      // variant: 0 = 128 columns per block, 1 = 64; -1 auto
    if (mode == 30) {
      if (!pf_gemm3f_ok<GEMM_EPI_BIAS_ELU>(g)) { pf_set_error("pf_gemm_test: shape / alignment not eligible for the LDS-DMA kernel"); return 2; }
      pf_gemm3f_launch<GEMM_EPI_BIAS_ELU>(g, s, bn);
    } else {
      if (!pf_gemm3f_ok<GEMM_EPI_MUL_DELU>(g)) { pf_set_error("pf_gemm_test: shape / alignment not eligible for the LDS-DMA kernel"); return 2; }
      pf_gemm3f_launch<GEMM_EPI_MUL_DELU>(g, s, bn);
    }
    PPO_HIP_OK(hipGetLastError());
    return 0;
  }
  if (mode >= 20) {      // bf16x6 with B (the weights) pre-split into bf16 planes in `ws`, as the PPO path runs the dense layers
    const int Kp = (K + 7) & ~7;
    SplitJobs js; js.n = 1;
    // mode 20: B = W [N][K];  mode 21: B = W [K][N], planes of W^T [N][K]
    js.j[0] = SplitJob{B, reinterpret_cast<unsigned short*>(ws), N, K, mode == 20 ? K : N, Kp, mode == 20 ? 0 : 1, (long long)N * Kp, 0};
    js.j[0].blocks = (int)(((long long)N * Kp + 255) / 256);
    hipLaunchKernelGGL(k_split_weights, dim3(js.j[0].blocks), dim3(256), 0, s, js);
    g.lda = K; g.ldb = K; g.Bp = reinterpret_cast<const unsigned short*>(ws); g.ldbp = Kp; g.bplane = (long long)N * Kp;
    if (mode == 20) pf_gemm3_launch<true, true, GEMM_EPI_BIAS_ELU>(g, 0, s, tile);
    else pf_gemm3_launch<true, true, GEMM_EPI_MUL_DELU>(g, 0, s, tile);
    PPO_HIP_OK(hipGetLastError());
    return 0;
  }
  if (mode >= 10) {      // the same three arrangements on the bf16x6 path (tile: -1 auto, 0 128x128, 1 128x64, 2 64x128, 3 64x64)
    if (mode == 10) { g.lda = K; g.ldb = K; pf_gemm3_launch<true, true, GEMM_EPI_BIAS_ELU>(g, 0, s, tile); }
    else if (mode == 11) { g.lda = K; g.ldb = N; pf_gemm3_launch<true, false, GEMM_EPI_MUL_DELU>(g, 0, s, tile); }
    else {
      g.lda = M; g.ldb = N; g.C = ws;
      int want = splits, t3 = tile;
      if (tile < 0) pf_gemm3_pick_split(M, N, K, splits, &t3, &want);
      int per = ((K + want - 1) / want + 31) / 32 * 32, S = (K + per - 1) / per;
      g.k_per_split = per;
      pf_gemm3_launch<false, false, GEMM_EPI_NONE, true>(g, S, s, t3);
      reduce_slabs(M * N + M, S, ws, Cm, s);
    }
    PPO_HIP_OK(hipGetLastError());
    return 0;
  }
  if (mode == 0) { g.lda = K; g.ldb = K; run(T{}, T{}, std::integral_constant<int, GEMM_EPI_BIAS_ELU>{}, F{}, 0); }
  else if (mode == 1) { g.lda = K; g.ldb = N; run(T{}, F{}, std::integral_constant<int, GEMM_EPI_MUL_DELU>{}, F{}, 0); }
  else {
    g.lda = M; g.ldb = N; g.C = ws;
    int want = splits;
    if (tile < 0) pf_pick_split(M, N, K, splits, &tile, &want);
    int per = ((K + want - 1) / want + 31) / 32 * 32, S = (K + per - 1) / per;
    g.k_per_split = per;
    run(F{}, F{}, std::integral_constant<int, GEMM_EPI_NONE>{}, T{}, S);
    int nw = M * N + M;
    reduce_slabs(nw, S, ws, Cm, s);
  }
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

int64_t ppo_num_params(const PpoConfig* cfg) {
  if (ppo_check_cfg(cfg)) return -1;
  PpoAlg a; a.cfg = *cfg; ppo_plan(&a);
  return (int64_t)a.nparams;
}
int32_t ppo_param_offsets(const PpoConfig* cfg, int64_t* offsets, int32_t max_entries) {
  if (ppo_check_cfg(cfg)) return -1;
  PpoAlg a; a.cfg = *cfg; ppo_plan(&a);
  int n = 0;
  auto push = [&](size_t v) { if (offsets && n < max_entries) offsets[n] = (int64_t)v; ++n; };
  for (int i = 0; i < a.actor.nl; ++i) { push(a.actor.l[i].w); push(a.actor.l[i].b); }
  for (int i = 0; i < a.critic.nl; ++i) { push(a.critic.l[i].w); push(a.critic.l[i].b); }
  push(a.std_off);
  return n;
}
int64_t ppo_workspace_floats(const PpoConfig* cfg) {
  if (ppo_check_cfg(cfg)) return -1;
  PpoAlg a; a.cfg = *cfg; ppo_plan(&a);
  return (int64_t)a.w_total;
}

PpoAlg* ppo_create(const PpoConfig* cfg, int32_t device) {
  if (ppo_check_cfg(cfg)) return nullptr;
  PpoAlg* a = new PpoAlg();
  a->cfg = *cfg; a->device = device;
  ppo_plan(a);
  {
    // gemm_mode 0: pick by size.  The bf16-split kernels win on the 512/256/128-wide networks (pointfoot_rough, the
    // quadrupeds: 11.8 vs 13.2 ms per update); on pointfoot_flat's 128/64/32 layers the plain fp32 MFMA kernels do
    // (2.9 vs 3.5 ms): there is too little arithmetic per tile to pay for splitting the operands.
    int widest = 0;
    for (int i = 1; i < cfg->n_actor_layers; ++i) widest = std::max(widest, cfg->actor_dims[i]);
    for (int i = 1; i < cfg->n_critic_layers; ++i) widest = std::max(widest, cfg->critic_dims[i]);
    a->split3 = cfg->gemm_mode == 2 || (cfg->gemm_mode == 0 && widest >= 256);
  }
  if (const char* gm = getenv("PF_PPO_GEMM")) a->split3 = atoi(gm) == 2;       // diagnostics: 1 = fp32 MFMA, 2 = bf16x6
  if (const char* dm = getenv("PF_PPO_DMA")) a->dma = atoi(dm) != 0;           // diagnostics: 0 = register-staged bf16x6 kernels everywhere
  if (const char* fh = getenv("PF_PPO_FUSED_HEADS")) a->fused_heads = atoi(fh) != 0;
  if (const char* pr = getenv("PF_PPO_PAIRED")) a->paired = atoi(pr) != 0;
  if (const char* l0 = getenv("PF_PPO_L0_FP32")) a->l0_fp32 = atoi(l0) != 0;
  if (const char* pr = getenv("PF_PPO_PAIRED_ROLLOUT")) a->pair_rollout = atoi(pr) != 0;
  a->paired = a->paired && a->fused_heads && a->actor.nl == a->critic.nl;
  const char* e = getenv("PF_PPO_DUAL_STREAM");
  a->dual = (e ? atoi(e) != 0 : true) && (!a->paired || !a->pair_rollout);      // the second stream is only used by the unpaired paths
  if (a->dual) {
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&a->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming) != hipSuccess) {
      a->aux = nullptr; a->dual = false;     // no GPU visible (host-only use of ppo_num_params etc.): single stream
      (void)hipGetLastError();
    }
  }
  return a;
}
void ppo_destroy(PpoAlg* a) {
  if (!a) return;
  if (a->aux) (void)hipStreamDestroy(a->aux);
  if (a->ev_fork) (void)hipEventDestroy(a->ev_fork);
  if (a->ev_join) (void)hipEventDestroy(a->ev_join);
  delete a;
}

int ppo_bind_buffers(PpoAlg* a, const PpoBuffers* b) {
  if (!a || !b) { pf_set_error("ppo_bind_buffers: null argument"); return 1; }
  const void* req[] = {b->d_params, b->d_grads, b->d_adam_m, b->d_adam_v, b->d_obs, b->d_critic_obs, b->d_actions, b->d_rewards, b->d_dones,
                       b->d_values, b->d_returns, b->d_advantages, b->d_log_prob, b->d_mu, b->d_sigma, b->d_perm, b->d_workspace, b->d_scalars};
  for (const void* p : req) if (!p) { pf_set_error("ppo_bind_buffers: a required buffer is NULL"); return 1; }
  if (b->workspace_floats < (int64_t)a->w_total) { pf_set_error("ppo_bind_buffers: workspace too small (see ppo_workspace_floats)"); return 1; }
  a->b = *b; a->bound = true; a->planes_valid = false; a->pend = false;
  return 0;
}

static void ppo_flush_process(PpoAlg* a, hipStream_t s);

int ppo_act(PpoAlg* a, int32_t t, const float* d_obs, const float* d_critic_obs, float* d_actions_out, int64_t sample_counter,
            const float* d_inject_normal, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_act: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  if (t < 0 || t >= c.num_steps) { pf_set_error("ppo_act: transition index out of range (rollout buffer overflow)"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  const int N = c.num_envs, A = c.num_actions, O = c.num_obs, P = c.num_critic_obs;
  float* ws = a->b.d_workspace;
  // Parameters change between rollouts (optimiser step, checkpoint load, broadcast): the planes are rebuilt at the head of every
  // rollout, and whenever this handle knows them stale (first dense call after create / bind, an optimiser step since).  A new
  // rollout also invalidates the permuted copy of the storage.
  if (t == 0 || !a->planes_valid) split_weights(a, s);
  if (t == 0) a->permuted = false;
  const bool pair = a->paired && a->pair_rollout;
  hipStream_t s2 = pair ? s : ppo_fork(a, s);
  size_t tn = (size_t)t * N;
  if (a->fused_heads) {
    if (pair) mlp_forward_pair(a, d_obs, d_critic_obs, N, s);
    else {
      mlp_forward(a, a->actor, d_obs, N, a->w_ha, s, false);
      mlp_forward(a, a->critic, d_critic_obs, N, a->w_hc, s2, false);
      ppo_join(a, s, s2);
    }
    const Layer& La = a->actor.l[a->actor.nl - 1];
    const Layer& Lc = a->critic.l[a->critic.nl - 1];
    hipLaunchKernelGGL(k_act_heads, dim3(std::max((N + 31) / 32, 512)), dim3(256), sizeof(float) * ((size_t)A * La.in + Lc.in), s, N, A, O, P,
                       La.in, Lc.in, d_obs, d_critic_obs, ws + a->w_ha[a->actor.nl - 2], ws + a->w_hc[a->critic.nl - 2],
                       a->b.d_params + La.w, a->b.d_params + La.b, a->b.d_params + Lc.w, a->b.d_params + Lc.b,
                       a->b.d_params + a->std_off, d_inject_normal, c.seed_lo, c.seed_hi, (uint32_t)sample_counter, 0u, d_actions_out,
                       a->b.d_obs + tn * O, a->b.d_critic_obs + tn * P, a->b.d_actions + tn * A, a->b.d_values + tn, a->b.d_log_prob + tn,
                       a->b.d_mu + tn * A, a->b.d_sigma + tn * A, a->pend ? a->pend_args : ProcArgs{});
    a->pend = false;
  } else {
    ppo_flush_process(a, s);
    mlp_forward(a, a->actor, d_obs, N, a->w_ha, s);
    mlp_forward(a, a->critic, d_critic_obs, N, a->w_hc, s2);
    ppo_join(a, s, s2);
    hipLaunchKernelGGL(k_sample, dim3(std::max((N + 255) / 256, 512)), dim3(256), 0, s, N, A, O, P, d_obs, d_critic_obs, ws + a->w_ha[a->actor.nl - 1],
                       ws + a->w_hc[a->critic.nl - 1], a->b.d_params + a->std_off, d_inject_normal, c.seed_lo, c.seed_hi, (uint32_t)sample_counter,
                       0u, d_actions_out, a->b.d_obs + tn * O, a->b.d_critic_obs + tn * P, a->b.d_actions + tn * A, a->b.d_values + tn,
                       a->b.d_log_prob + tn, a->b.d_mu + tn * A, a->b.d_sigma + tn * A);
  }
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

// PPO.act up to the last hidden layer; the heads, the sampling, the transition store and a pending deferred process_env_step are
// handed to the env step (pf_step_policy) as a PfPolicyHeads -- exactly the arguments k_act_heads would have been launched with.
int ppo_act_hidden(PpoAlg* a, int32_t t, const float* d_obs, const float* d_critic_obs, float* d_actions_out, int64_t sample_counter,
                   const float* d_inject_normal, PfPolicyHeads* h, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_act_hidden: buffers not bound"); return 1; }
  if (!h || !d_actions_out) { pf_set_error("ppo_act_hidden: null argument"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  if (t < 0 || t >= c.num_steps) { pf_set_error("ppo_act_hidden: transition index out of range (rollout buffer overflow)"); return 1; }
  const Layer& La = a->actor.l[a->actor.nl - 1];
  const Layer& Lc = a->critic.l[a->critic.nl - 1];
  const int A = c.num_actions;
  if (!a->fused_heads || A > 16 || La.in > 128 || Lc.in > 128 || ((La.in | Lc.in) & 3) || a->actor.nl < 2 || a->critic.nl < 2) return 2;
  hipStream_t s = (hipStream_t)stream;
  const int N = c.num_envs, O = c.num_obs, P = c.num_critic_obs;
  float* ws = a->b.d_workspace;
  if (t == 0 || !a->planes_valid) split_weights(a, s);
  if (t == 0) a->permuted = false;
  const bool pair = a->paired && a->pair_rollout;
  if (pair) mlp_forward_pair(a, d_obs, d_critic_obs, N, s);
  else {
    hipStream_t s2 = ppo_fork(a, s);
    mlp_forward(a, a->actor, d_obs, N, a->w_ha, s, false);
    mlp_forward(a, a->critic, d_critic_obs, N, a->w_hc, s2, false);
    ppo_join(a, s, s2);
  }
  PPO_HIP_OK(hipGetLastError());
  const size_t tn = (size_t)t * N;
  std::memset(h, 0, sizeof(*h));
  h->d_ha = ws + a->w_ha[a->actor.nl - 2]; h->d_hc = ws + a->w_hc[a->critic.nl - 2];
  h->d_wa = a->b.d_params + La.w; h->d_ba = a->b.d_params + La.b; h->d_wc = a->b.d_params + Lc.w; h->d_bc = a->b.d_params + Lc.b;
  h->d_std = a->b.d_params + a->std_off; h->d_inject_normal = d_inject_normal;
  h->d_obs = d_obs; h->d_critic_obs = d_critic_obs; h->d_actions_out = d_actions_out;
  h->s_obs = a->b.d_obs + tn * O; h->s_critic_obs = a->b.d_critic_obs + tn * P; h->s_actions = a->b.d_actions + tn * A;
  h->s_values = a->b.d_values + tn; h->s_log_prob = a->b.d_log_prob + tn; h->s_mu = a->b.d_mu + tn * A; h->s_sigma = a->b.d_sigma + tn * A;
  if (a->pend) {
    const ProcArgs& q = a->pend_args;
    h->p_rew = q.rew; h->p_dones = q.dones; h->p_time_outs = q.time_outs; h->p_values = q.s_val; h->p_s_rew = q.s_rew; h->p_s_done = q.s_done;
    h->p_ep_ret = q.ep_ret; h->p_ep_len = q.ep_len; h->p_ep_stats = q.ep_stats; h->p_fin_ret = q.fin_ret; h->p_fin_len = q.fin_len; h->p_gamma = q.gamma;
    a->pend = false;      // carried out by the step launch the caller enqueues next on this stream
  }
  h->ka = La.in; h->kc = Lc.in; h->num_obs = O; h->num_critic_obs = P; h->num_actions = A;
  h->seed_lo = c.seed_lo; h->seed_hi = c.seed_hi; h->counter = (uint32_t)sample_counter; h->env_base = 0u;
  return 0;
}

int ppo_act_inference(PpoAlg* a, const float* d_obs, int32_t n, float* d_actions_out, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_act_inference: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  if (n < 1 || n > a->max_rows) { pf_set_error("ppo_act_inference: batch larger than the workspace"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  split_weights(a, s);
  mlp_forward(a, a->actor, d_obs, n, a->w_ha, s);
  PPO_HIP_OK(hipMemcpyAsync(d_actions_out, a->b.d_workspace + a->w_ha[a->actor.nl - 1], sizeof(float) * n * a->cfg.num_actions,
                            hipMemcpyDeviceToDevice, s));
  return 0;
}

static ProcArgs proc_args(PpoAlg* a, int32_t t, const float* d_rew, const uint8_t* d_dones, const uint8_t* d_time_outs) {
  const PpoConfig& c = a->cfg;
  const size_t tn = (size_t)t * c.num_envs;
  ProcArgs p;
  p.rew = d_rew; p.dones = d_dones; p.time_outs = d_time_outs; p.s_val = a->b.d_values + tn;
  p.s_rew = a->b.d_rewards + tn; p.s_done = a->b.d_dones + tn;
  p.ep_ret = a->b.d_ep_stats ? a->b.d_ep_return : nullptr; p.ep_len = a->b.d_ep_length; p.ep_stats = a->b.d_ep_stats;
  const bool ring = a->b.d_ep_fin && a->b.d_ep_ring && a->b.d_ep_stats;      // then k_episode_ring also accumulates d_ep_stats
  p.fin_ret = ring ? a->b.d_ep_fin + tn : nullptr;
  p.fin_len = ring ? a->b.d_ep_fin + (size_t)c.num_steps * c.num_envs + tn : nullptr;
  p.gamma = c.gamma;
  return p;
}
// a deferred step that no k_act_heads picked up (end of the rollout, a non-fused head path, an immediate step behind it)
static void ppo_flush_process(PpoAlg* a, hipStream_t s) {
  if (!a->pend) return;
  hipLaunchKernelGGL(k_process, dim3((a->cfg.num_envs + 255) / 256), dim3(256), 0, s, a->cfg.num_envs, a->pend_args);
  a->pend = false;
}

int ppo_process_env_step(PpoAlg* a, int32_t t, const float* d_rew, const uint8_t* d_dones, const uint8_t* d_time_outs, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_process_env_step: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  if (t < 0 || t >= c.num_steps) { pf_set_error("ppo_process_env_step: transition index out of range"); return 1; }
  ppo_flush_process(a, (hipStream_t)stream);
  hipLaunchKernelGGL(k_process, dim3((c.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, c.num_envs,
                     proc_args(a, t, d_rew, d_dones, d_time_outs));
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

int ppo_process_env_step_deferred(PpoAlg* a, int32_t t, const float* d_rew, const uint8_t* d_dones, const uint8_t* d_time_outs, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_process_env_step_deferred: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  if (t < 0 || t >= a->cfg.num_steps) { pf_set_error("ppo_process_env_step_deferred: transition index out of range"); return 1; }
  ppo_flush_process(a, (hipStream_t)stream);
  a->pend_args = proc_args(a, t, d_rew, d_dones, d_time_outs);
  a->pend = true;
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

int ppo_compute_returns(PpoAlg* a, const float* d_last_critic_obs, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_compute_returns: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* ws = a->b.d_workspace;
  const int N = c.num_envs, T = c.num_steps;
  ppo_flush_process(a, s);
  split_weights(a, s);
  mlp_forward(a, a->critic, d_last_critic_obs, N, a->w_hc, s);
  PPO_HIP_OK(hipMemcpyAsync(ws + a->w_lastval, ws + a->w_hc[a->critic.nl - 1], sizeof(float) * N, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_gae, dim3((N + 255) / 256), dim3(256), 0, s, N, T, c.gamma, c.lam, ws + a->w_lastval, a->b.d_values, a->b.d_rewards,
                     a->b.d_dones, a->b.d_returns, a->b.d_advantages);
  size_t n = (size_t)N * T;
  hipLaunchKernelGGL(k_moments1, dim3(RED_BLOCKS), dim3(256), 0, s, n, a->b.d_advantages, ws + a->w_partial);
  hipLaunchKernelGGL(k_moments2, dim3(1), dim3(64), 0, s, RED_BLOCKS, n, ws + a->w_partial, a->b.d_scalars);
  if (a->b.d_ep_ring && a->b.d_ep_fin && a->b.d_ep_stats)      // the runner's deque(maxlen=100) of finished episodes, on the device
  {
    const long long chunk = ep_chunk_len((long long)n);
    const int nchunks = (int)(((long long)n + chunk - 1) / chunk);
    hipLaunchKernelGGL(k_episode_chunks, dim3(nchunks), dim3(256), 0, s, (long long)n, chunk, a->b.d_dones, a->b.d_ep_fin, ws + a->w_partial);
    hipLaunchKernelGGL(k_episode_ring, dim3(1), dim3(1024), 0, s, N, T, chunk, nchunks, a->b.d_dones, a->b.d_ep_fin, ws + a->w_partial,
                       a->b.d_ep_ring, a->b.d_ep_stats);
  }
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

int ppo_normalize_advantages(PpoAlg* a, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_normalize_advantages: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  size_t n = (size_t)a->cfg.num_envs * a->cfg.num_steps;
  a->permuted = false;      // the permuted copy holds the advantages as they were
  hipLaunchKernelGGL(k_normalize, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, n, a->b.d_advantages, a->b.d_scalars);
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

static void ppo_gather(PpoAlg* a, int rows, const int* perm, const size_t* dst_off, hipStream_t s) {
  const PpoConfig& c = a->cfg;
  const int A = c.num_actions, O = c.num_obs, P = c.num_critic_obs;
  GatherArgs ga;
  const float* srcs[9] = {a->b.d_obs, a->b.d_critic_obs, a->b.d_actions, a->b.d_log_prob, a->b.d_mu, a->b.d_sigma, a->b.d_advantages,
                          a->b.d_returns, a->b.d_values};
  const int cols[9] = {O, P, A, 1, A, A, 1, 1, 1};
  const int lds[9] = {a->ld_xa, a->ld_xc, A, 1, A, A, 1, 1, 1};
  for (int i = 0; i < 9; ++i) ga.job[i] = GatherJob{srcs[i], a->b.d_workspace + dst_off[i], cols[i], lds[i]};
  ga.njobs = 9;
  hipLaunchKernelGGL(k_gather, dim3(1024, 9), dim3(256), 0, s, rows, perm, ga);
}

/* rsl_rl's mini_batch_generator draws ONE permutation per update and reuses it in every epoch: the storage is brought into
 * that order once, and every minibatch of every epoch is then a contiguous slice (instead of 20 gathers per update). */
int ppo_begin_update(PpoAlg* a, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_begin_update: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const long long TN = (long long)a->cfg.num_envs * a->cfg.num_steps;
  const long long widest = std::max(a->ld_xa, a->ld_xc);
  if (getenv("PF_PPO_PERMUTE") && atoi(getenv("PF_PPO_PERMUTE")) == 0) { a->permuted = false; return 0; }
  if (TN * widest >= (1ll << 31)) { a->permuted = false; return 0; }      // 32-bit gather indices: fall back to per-minibatch gathers
  ppo_gather(a, (int)TN, a->b.d_perm, a->w_perm, (hipStream_t)stream);
  a->permuted = true;
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"

static int minibatch_backward(PpoAlg* a, int32_t mb, void* stream, int part, bool want_sq = false) {
  if (a) a->sq_blocks = 0;
  if (!a || !a->bound) { pf_set_error("ppo_minibatch_backward: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  if (mb < 0 || mb >= c.num_mini_batches) { pf_set_error("ppo_minibatch_backward: minibatch index out of range"); return 1; }
  if (part >= 0 && !(a->fused_heads && a->paired)) { pf_set_error("ppo_minibatch_backward_part: needs the paired execution mode (equal depth, fused heads)"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  float* ws = a->b.d_workspace;
  const int B = a->batch, A = c.num_actions, O = c.num_obs, P = c.num_critic_obs;
  const int* perm = a->b.d_perm + (size_t)mb * B;
  if (part == 1) {      // second half of a pass begun with part 0: the minibatch arrays and activations are in place
    const size_t o_xa = a->permuted ? a->w_perm[0] + (size_t)mb * B * a->ld_xa : a->w_xa, o_xc = a->permuted ? a->w_perm[1] + (size_t)mb * B * a->ld_xc : a->w_xc;
    mlp_backward_pair(a, ws + o_xa, ws + o_xc, B, s, a->ld_xa, a->ld_xc, nullptr, 1);
    PPO_HIP_OK(hipGetLastError());
    return 0;
  }
  split_weights(a, s);
  // minibatch arrays: slices of the permuted storage (ppo_begin_update) or, without it, gathered now
  size_t mbo[9];
  {
    const size_t direct[9] = {a->w_xa, a->w_xc, a->w_act, a->w_oldlogp, a->w_oldmu, a->w_oldsig, a->w_adv, a->w_ret, a->w_oldval};
    const int cols[9] = {a->ld_xa, a->ld_xc, A, 1, A, A, 1, 1, 1};
    if (a->permuted) { for (int i = 0; i < 9; ++i) mbo[i] = a->w_perm[i] + (size_t)mb * B * cols[i]; }
    else { for (int i = 0; i < 9; ++i) mbo[i] = direct[i]; ppo_gather(a, B, perm, direct, s); }
  }
  const size_t o_xa = mbo[0], o_xc = mbo[1], o_act = mbo[2], o_logp = mbo[3], o_mu = mbo[4], o_sig = mbo[5], o_adv = mbo[6], o_ret = mbo[7], o_val = mbo[8];
  hipStream_t s2 = a->paired ? s : ppo_fork(a, s);
  const float* stdp = a->b.d_params + a->std_off;
  const Layer& La = a->actor.l[a->actor.nl - 1];
  const Layer& Lc = a->critic.l[a->critic.nl - 1];
  if (a->fused_heads) {
    // hidden layers on the GEMM, then both heads + loss + head backward in one kernel
    if (a->paired) mlp_forward_pair(a, ws + o_xa, ws + o_xc, B, s, a->ld_xa, a->ld_xc);
    else {
      mlp_forward(a, a->actor, ws + o_xa, B, a->w_ha, s, false, a->ld_xa);
      mlp_forward(a, a->critic, ws + o_xc, B, a->w_hc, s2, false, a->ld_xc);
      ppo_join(a, s, s2);
    }
    const int nblk = (B + HEAD_ROWS - 1) / HEAD_ROWS;
    const size_t lds = sizeof(float) * ((size_t)A * La.in + Lc.in + HEAD_ROWS * (A + 1) + 4 * (A + 4));
    auto kh = A == 6 ? k_heads<6> : A == 12 ? k_heads<12> : k_heads<16>;
    hipLaunchKernelGGL(kh, dim3(nblk), dim3(256), lds, s, B, A, La.in, Lc.in, c.clip_param, c.value_loss_coef, c.entropy_coef,
                       c.use_clipped_value_loss, ws + a->w_ha[a->actor.nl - 2], ws + a->w_hc[a->critic.nl - 2], a->b.d_params + La.w,
                       a->b.d_params + La.b, a->b.d_params + Lc.w, a->b.d_params + Lc.b, stdp, ws + o_act, ws + o_logp,
                       ws + o_mu, ws + o_sig, ws + o_adv, ws + o_ret, ws + o_val,
                       ws + a->w_g[0][a->actor.nl - 2], ws + a->w_g[1][a->critic.nl - 2],
                       ws + (a->paired ? a->w_slabl[0][a->actor.nl - 1] : a->w_slabs[0]), ws + (a->paired ? a->w_slabl[1][a->critic.nl - 1] : a->w_slabs[1]),
                       ws + a->w_partial);
    const LossFinish lf{nblk, A, B, c.entropy_coef, ws + a->w_partial, stdp, a->b.d_grads + a->std_off, a->b.d_grads + a->nparams, a->b.d_scalars};
    if (a->paired) mlp_backward_pair(a, ws + o_xa, ws + o_xc, B, s, a->ld_xa, a->ld_xc, &lf, part, want_sq);      // finished inside the slab reduction launch
    else {
      hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, s, lf);
      s2 = ppo_fork(a, s);
      mlp_backward(a, a->actor, ws + o_xa, B, a->w_ha, nullptr, s, 0, a->ld_xa);
      mlp_backward(a, a->critic, ws + o_xc, B, a->w_hc, nullptr, s2, 1, a->ld_xc);
    }
  } else {
    mlp_forward(a, a->actor, ws + o_xa, B, a->w_ha, s, true, a->ld_xa);
    mlp_forward(a, a->critic, ws + o_xc, B, a->w_hc, s2, true, a->ld_xc);
    ppo_join(a, s, s2);
    hipLaunchKernelGGL(k_loss, dim3(RED_BLOCKS), dim3(256), 0, s, B, A, c.clip_param, c.value_loss_coef, c.entropy_coef, c.use_clipped_value_loss,
                       ws + a->w_ha[a->actor.nl - 1], ws + a->w_hc[a->critic.nl - 1], stdp, ws + o_act, ws + o_logp, ws + o_mu,
                       ws + o_sig, ws + o_adv, ws + o_ret, ws + o_val, ws + a->w_dmean, ws + a->w_dvalue, ws + a->w_partial);
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, s,
                       LossFinish{RED_BLOCKS, A, B, c.entropy_coef, ws + a->w_partial, stdp, a->b.d_grads + a->std_off, a->b.d_grads + a->nparams, a->b.d_scalars});
    s2 = ppo_fork(a, s);
    mlp_backward(a, a->actor, ws + o_xa, B, a->w_ha, ws + a->w_dmean, s, 0, a->ld_xa);
    mlp_backward(a, a->critic, ws + o_xc, B, a->w_hc, ws + a->w_dvalue, s2, 1, a->ld_xc);
  }
  ppo_join(a, s, s2);
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

extern "C" {

int ppo_minibatch_backward(PpoAlg* a, int32_t mb, void* stream) { return minibatch_backward(a, mb, stream, -1); }

int ppo_minibatch_backward_part(PpoAlg* a, int32_t mb, int32_t part, void* stream) {
  if (part != 0 && part != 1) { pf_set_error("ppo_minibatch_backward_part: part must be 0 or 1"); return 1; }
  return minibatch_backward(a, mb, stream, part);
}

int ppo_minibatch_step(PpoAlg* a, int32_t mb, void* stream) {
  if (a && a->cfg.world_size > 1) { pf_set_error("ppo_minibatch_step: one process only (the gradient all-reduce sits between backward and apply)"); return 1; }
  if (int rc = minibatch_backward(a, mb, stream, -1, true)) return rc;
  if (a->sq_blocks <= 0) return ppo_minibatch_apply(a, stream);      // (unpaired execution modes: no partials were left)
  const PpoConfig& c = a->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* ws = a->b.d_workspace;
  unsigned* ticket = reinterpret_cast<unsigned*>(ws + a->w_sqpart + PPO_SQ_PARTIALS);
  if (!a->ticket_zeroed) { PPO_HIP_OK(hipMemsetAsync(ticket, 0, sizeof(unsigned), s)); a->ticket_zeroed = true; }
  hipLaunchKernelGGL(k_adam_fused, dim3(RED_BLOCKS), dim3(256), 0, s, a->nparams, a->b.d_params, a->b.d_grads, a->b.d_adam_m, a->b.d_adam_v,
                     a->sq_blocks, ws + a->w_sqpart, a->b.d_grads + a->nparams, c.adaptive_schedule, c.desired_kl, c.max_grad_norm,
                     c.adam_beta1, c.adam_beta2, c.adam_eps, a->b.d_scalars, ticket);
  a->planes_valid = false;
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

int32_t ppo_supports_parts(const PpoAlg* a) { return a && a->fused_heads && a->paired ? 1 : 0; }

int32_t ppo_grad_groups(const PpoConfig* cfg, int64_t* ranges) {
  if (ppo_check_cfg(cfg) || !ranges) return -1;
  PpoAlg a; a.cfg = *cfg; ppo_plan(&a);
  ranges[0] = (int64_t)a.early_off; ranges[1] = (int64_t)(a.nparams + 4 - a.early_off);      // EARLY: layers >= 1, std, KL tail
  ranges[2] = 0; ranges[3] = (int64_t)a.early_off;                                          // LATE: the two input layers
  return 0;
}

int ppo_minibatch_apply(PpoAlg* a, void* stream) {
  if (!a || !a->bound) { pf_set_error("ppo_minibatch_apply: buffers not bound"); return 1; }
  if (ppo_wrong_device(a)) return 1;
  const PpoConfig& c = a->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* ws = a->b.d_workspace;
  float gscale = 1.0f / (float)(c.world_size > 0 ? c.world_size : 1);
  hipLaunchKernelGGL(k_sumsq1, dim3(RED_BLOCKS), dim3(256), 0, s, a->nparams, a->b.d_grads, gscale, ws + a->w_partial);
  hipLaunchKernelGGL(k_prepare_step, dim3(1), dim3(64), 0, s, RED_BLOCKS, ws + a->w_partial, a->b.d_grads + a->nparams, c.adaptive_schedule,
                     c.desired_kl, c.max_grad_norm, a->b.d_scalars);
  hipLaunchKernelGGL(k_adam, dim3(RED_BLOCKS), dim3(256), 0, s, a->nparams, a->b.d_params, a->b.d_grads, a->b.d_adam_m, a->b.d_adam_v, gscale,
                     c.adam_beta1, c.adam_beta2, c.adam_eps, a->b.d_scalars);
  a->planes_valid = false;
  PPO_HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
