/* igmc_rng.h -- counter-based hashes shared by the HIP kernels and the C oracle.
 *
 * The reference draws per-hop samples with CPython random.sample over a set
 * (util_functions.py:222-229) and dropout masks with torch RNG streams; neither
 * is reproducible across devices/worker counts.  The engine uses
 * stateless hashes instead, keyed by what identifies the draw:
 *
 *   per-hop sampling : uniform k-subset = the k candidates with the smallest
 *                      igmc_sample_key(salt, id); the key is a BIJECTION of the
 *                      32-bit id for a fixed salt, so keys never tie and the
 *                      selection is an exact uniform k-subset w.r.t. the salt.
 *   dropout          : keep iff igmc_u01(hash) >= p.
 */
#ifndef IGMC_RNG_H
#define IGMC_RNG_H
#include <stdint.h>

#if defined(__HIPCC__) && !defined(IGMC_HIPEMU)
#define IGMC_HD __host__ __device__ static inline
#else
#define IGMC_HD static inline
#endif

IGMC_HD uint64_t igmc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

IGMC_HD uint32_t igmc_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

/* salt for one (dataset seed, epoch, link position, hop distance, side) draw */
IGMC_HD uint64_t igmc_sample_salt(uint64_t seed, uint64_t epoch, uint64_t link, uint32_t dist, uint32_t side) {
  uint64_t s = igmc_splitmix64(seed ^ 0x49474D43ull);
  s = igmc_splitmix64(s ^ epoch);
  s = igmc_splitmix64(s ^ link);
  s = igmc_splitmix64(s ^ (((uint64_t)dist << 1) | side));
  return s;
}

/* bijective in `id` for fixed salt */
IGMC_HD uint32_t igmc_sample_key(uint64_t salt, uint32_t id) {
  return igmc_fmix32(igmc_fmix32(id + (uint32_t)salt) ^ (uint32_t)(salt >> 32));
}

/* edge dropout: dir = 0 user->item, 1 item->user, 2 undirected (force_undirected) */
IGMC_HD uint32_t igmc_edge_hash(uint64_t seed, uint64_t step, uint32_t graph, uint32_t u, uint32_t v, uint32_t dir) {
  uint64_t s = igmc_splitmix64(seed ^ 0x45444745ull);
  s = igmc_splitmix64(s ^ step);
  s = igmc_splitmix64(s ^ (((uint64_t)graph << 2) | dir));
  s = igmc_splitmix64(s ^ (((uint64_t)u << 32) | v));
  return (uint32_t)(s >> 32);
}

/* MLP dropout(0.5) on hidden unit j of graph g */
IGMC_HD uint32_t igmc_unit_hash(uint64_t seed, uint64_t step, uint32_t graph, uint32_t j) {
  uint64_t s = igmc_splitmix64(seed ^ 0x4D4C5044ull);
  s = igmc_splitmix64(s ^ step);
  s = igmc_splitmix64(s ^ (((uint64_t)graph << 32) | j));
  return (uint32_t)(s >> 32);
}

/* uniform in [0,1) with 24 bits */
IGMC_HD float igmc_u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

#endif /* IGMC_RNG_H */
