/* TEST INFRASTRUCTURE (oracle) — Philox4x32-10 counter-based generator (Salmon et al., SC'11),
 * restated from the published algorithm.  The reference draws from torch's unseeded global CPU
 * generator (legged_gym/utils/helpers.py:74 is commented out), so bit-parity of random draws with
 * the reference is undefined; this generator only has to agree between the oracle and the HIP path. */
#ifndef PFO_PHILOX_H
#define PFO_PHILOX_H
#include <stdint.h>

static inline void pfo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in [0,1) with 24 random bits (same granularity as torch.rand for float32) */
static inline float pfo_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* slot j of the virtual random row of (env, step, stream) */
static inline float pfo_rand_slot(uint32_t seed_lo, uint32_t seed_hi, uint32_t env, uint32_t step,
                                  uint32_t stream, uint32_t j) {
  uint32_t o[4];
  pfo_philox4x32(env, step, stream, j >> 2, seed_lo, seed_hi, o);
  return pfo_u01(o[j & 3]);
}
#endif
