// g2_image.h -- layout of the staged weight images (internal): shared by their two producers, k_g2_compose
// (graphstep2.hip: from the current parameters) and k_finalize_ts (model.hip: from the parameters it has just updated).
#pragma once
#include "common.h"

#define G2_NR 5                   // relations the fragments are built for (R <= G2_NR)
#define G2_NT 3                   // bf16 terms of an f32 value
// One staged weight image = the B operand of a layer's dense transform as bf16 terms, in MFMA fragment order:
// [term (hi, mid, lo)][block (relation 0..4, 5 = root)][16-column tile nt][lane] x 8 bf16; lane (li = column n & 15,
// kq) holds rows k(kq, e) = (e < 4 ? 4 kq + e : 16 + 4 kq + e - 4) of its block -- the order in which the gather's
// accumulators hold the input features of a row.
#define G2_WIMG (G2_NT * (G2_NR + 1) * 2 * 64 * 4)   // 4-byte words of one staged image (36 KB)
// Models with more than G2_NR relations take them in GROUPS of G2_NR (R <= G2_NR * G2_NG_MAX): one staged image per group --
// block G2_NR of group 0 = root, of the other groups = zero -- and a layer-0 table of 64 rows.  A model whose layer-0 table
// [R L relation-label rows | L root rows | bias] has more than 32 rows (two hops: L = 6) takes the same two-group layout
// (its second group holds no relation and is skipped at run time): the layout is chosen by g2_groups(R, L).
// g2_w: [3 layers][forward image, transposed image][group] then the layer-0 table [32 | 64][32] f32
#define G2_NG_MAX 2
__host__ __device__ static inline int g2_rel_groups(int R) { return (R + G2_NR - 1) / G2_NR; }      // groups that hold relations
__host__ __device__ static inline int g2_groups(int R, int L) {
  const int ng = g2_rel_groups(R);
  return (ng == 1 && R * L + L + 1 > 32) ? 2 : ng;
}
__host__ __device__ static inline int g2_t0_rows(int R, int L) { return g2_groups(R, L) == 1 ? 32 : 64; }
__host__ __device__ static inline size_t g2_img_off(int ng, int l, int trans, int grp) {
  return (size_t)(((l - 1) * 2 + trans) * ng + grp) * G2_WIMG;
}
__host__ __device__ static inline size_t g2_t0_off(int ng) { return (size_t)6 * ng * G2_WIMG; }
__host__ __device__ static inline size_t g2_w_words(int R, int L) { return g2_t0_off(g2_groups(R, L)) + (size_t)g2_t0_rows(R, L) * 32; }

#ifndef IGMC_HIPEMU
typedef __bf16 g2_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g2_f32x2 __attribute__((ext_vector_type(2)));
#endif

// {bf16(x) | bf16(y) << 16}, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t g2_pk_bf16(float x, float y) {
#ifdef IGMC_HIPEMU
  return hipemu_f32_to_bf16_rne(x) | (hipemu_f32_to_bf16_rne(y) << 16);
#else
  g2_f32x2 v = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g2_bf16x2));
#endif
}
// the three bf16 terms of two f32 values: hi + mid + lo == x to 24 bits (each residual is exact in f32)
__device__ __forceinline__ void g2_split2(float x, float y, uint32_t& h, uint32_t& mi, uint32_t& lo) {
  h = g2_pk_bf16(x, y);
  const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xFFFF0000u);
  mi = g2_pk_bf16(rx, ry);
  const float sx = rx - __uint_as_float(mi << 16), sy = ry - __uint_as_float(mi & 0xFFFF0000u);
  lo = g2_pk_bf16(sx, sy);
}

// W_r[c][f] = sum_b att[r,b] basis_b[c][f]: ONE order of operations for both producers (explicit fmas: the images of a
// step do not depend on which kernel formed them)
__device__ __forceinline__ float g2_wsum(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
  return fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, a0 * b0)));
}

// halfword index of element (k, n) of block r, term t, inside one image
__device__ __forceinline__ int g2_img_index(int t, int r, int k, int n) {
  return ((((t * (G2_NR + 1) + r) * 2 + (n >> 4)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 8) + 4 * (k >> 4) + (k & 3);
}
