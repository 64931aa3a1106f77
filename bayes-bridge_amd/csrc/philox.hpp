// Counter-based Philox4x32-10 (Salmon et al., SC'11) for the on-device draws.
// Every random quantity of a chain is addressed by (seed, stream id, element
// index, draw counter), so results do not depend on grid shape or on how many
// chains share a GPU.  Device RNG gives distribution parity with the
// reference's NumPy streams, never bit parity ("RNG").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbx {

struct Philox {
  uint32_t key[2];
  uint32_t ctr[4];
  uint32_t out[4];
  int have;  // unread 32-bit words in out

  // `trial` selects one of 4096 independent sub-streams of an element (the
  // speculative rejection samplers give every candidate proposal its own).
  __host__ __device__ Philox(uint64_t seed, uint64_t stream, uint64_t index,
                             uint32_t trial = 0) {
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    ctr[0] = trial << 20;  // low 20 bits: draw counter
    ctr[1] = (uint32_t)stream;
    ctr[2] = (uint32_t)index;
    ctr[3] = (uint32_t)(index >> 32) ^ ((uint32_t)(stream >> 32) << 16);
    have = 0;
  }

  __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b,
                                                 uint32_t& hi, uint32_t& lo) {
    const uint64_t prod = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(prod >> 32);
    lo = (uint32_t)prod;
  }

  __host__ __device__ inline void refill() {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
    for (int round = 0; round < 10; ++round) {
      uint32_t hi0, lo0, hi1, lo1;
      mulhilo(0xD2511F53u, c0, hi0, lo0);
      mulhilo(0xCD9E8D57u, c2, hi1, lo1);
      const uint32_t n0 = hi1 ^ c1 ^ k0;
      const uint32_t n1 = lo1;
      const uint32_t n2 = hi0 ^ c3 ^ k1;
      const uint32_t n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    have = 4;
    ctr[0] += 1;  // 2^20 blocks of 4 words per (seed, stream, index, trial)
  }

  __host__ __device__ inline uint32_t next_u32() {
    if (have == 0) refill();
    return out[--have];
  }

  // Uniform on (0, 1): 53 random bits, never exactly 0 or 1.
  __host__ __device__ inline double uniform() {
    const uint64_t hi = next_u32();
    const uint64_t lo = next_u32();
    const uint64_t bits = ((hi << 32) | lo) >> 11;  // 53 bits
    return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
  }

  // Standard normal by Box-Muller (one of the pair is discarded: the callers
  // are rejection samplers with data-dependent consumption).
  __device__ inline double normal() {
    const double u1 = uniform();
    const double u2 = uniform();
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
  }
};

// Stream ids (second counter word) of the draws inside one Gibbs iteration.
enum PhiloxStream : uint64_t {
  STREAM_ETA1 = 1,   // cg_sampler.py:61
  STREAM_ETA2 = 2,   // cg_sampler.py:62
  STREAM_PG = 3,     // bayesbridge.py:406
  STREAM_GSCALE = 4, // bayesbridge.py:438
  STREAM_LSCALE = 5, // bayesbridge.py:463
  STREAM_OBSVAR = 6  // bayesbridge.py:403
};

}  // namespace bbx
