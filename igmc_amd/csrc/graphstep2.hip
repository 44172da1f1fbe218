// graphstep2.hip -- forward + loss + backward of one enclosing subgraph by a cluster of workgroups, with the
// relational message passing on the MATRIX CORES (gfx950 / CDNA4).
//
// Capped extraction leaves the induced block of every subgraph as a dense byte matrix relm[user][item] = relation + 1
// (extract.hip: k_relm).  With at most 128 nodes per side the per-relation aggregate of a 16-row bundle
//     T_r[i] = sum_{j : rel(i,j) = r, edge j -> i kept} x[j]          (reference models.py:199-202 -> PyG RGCNConv, aggr = add)
// is the dense product  T_r = A_r X  with A_r the 16 x K 0/1 block of relation r:  A_r is EXACT in bf16 and X splits
// into three bf16 terms (hi + mid + lo = x to 24 bits; every product 1.0 * bf16 is exact and the sums are f32), so the
// gather runs as v_mfma_f32_16x16x32_bf16 at f32 accuracy: 5 relations x 4 k-steps x 3 terms x 2 feature tiles = 120
// MFMAs (~2 k cycles) per bundle and layer pass instead of ~11 k cycles of LDS row gathers (graphstep.hip).
//
//  * the A fragments of a wave's bundle are built ONCE per launch from relm (8 bytes -> 8 bf16 per relation with a
//    handful of bit tricks) and stay in registers for all six layer passes;
//  * the gather is computed TRANSPOSED, T^T = X^T A^T (A operand = 16-byte reads of the bf16 planes [term][feature]
//    [node], B operand = the A_r fragments): its accumulators (lane = row, 4 consecutive features) are then directly
//    the A operand of the dense transform  out = [T_0..T_4 | x] @ [W_0; ..; W_4; root]  (f32 MFMA, k permuted the
//    same way on the weight side) -- no tile round trip through LDS in the forward;
//  * the transform's output layout (lane = feature, 4 consecutive rows) is node-contiguous: the bf16 terms of the
//    next layer's planes are formed in registers and published as bf16 TERM PLANES [term][feature][node] -- the very
//    image the gather reads from LDS --, four nodes = one 8-byte store per term;
//  * bundles are SIDE-PURE (16 consecutive user rows or 16 consecutive item rows) and statically owned by a PAIR of waves of
//    the cluster for the whole launch (one 16-column feature tile each): user bundles only ever need the item planes and
//    vice versa, a wave re-reads only rows it produced itself (h_l for the backward), and no degree ranking / schedule /
//    run lists exist any more -- every bundle costs the same.
//
// Cluster exchange (round 5; g2_prims.h): the members of a cluster sit on ONE XCD and hand their rows over through that
// XCD's L2 -- ORDINARY stores of the planes into the exchange region of (exchange, subgraph, side), a wait for their
// acknowledgement, ONE flag word per wave; the consumer polls the flags of the opposite side's bundles (sc1 loads) and copies
// the planes global -> LDS directly (global_load_lds).  No tagged data words, no sc1 stores, no tag wrap.  dPre_3 is non-zero
// on the two target rows only and is rebuilt locally from the head's d feat: 5 exchanges per launch (h_0, h_1, h_2, dPre_2,
// dPre_1) + the 256-float centre-node readout (8-byte {f32, tag} words, polled).
//
// This translation unit in four files (round 6): graphstep2.hip (this one: the shared gather / transform steps),
// g2_subgraph.h (k_graph_step2), g2_compose.h (k_g2_compose, layout, eligibility and launch of the subgraph kernel),
// dl_kernels.h (the dense-layer kernels k_dl_* for slots of up to 256 nodes a side and up to ten relations).
//
// Eligibility (else the per-layer kernels run): dense block present, R <= 5, layer-0 table <= 32 rows,
// no side features, both sides <= 16 * (2 * cluster size) <= 128 rows.
#include "launch.h"
#include "g2_image.h"
#include "head_sub.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#define G2_THREADS 512            // k_graph_step2: eight waves = two per SIMD
#define G2_NB 4                   // 16-row bundles of a workgroup; a bundle is worked by a PAIR of waves (2 b, 2 b + 1), one
                                  // per 16-column feature tile: gather, half of the transform's K, epilogue of that tile
#define G2C_THREADS 256           // k_g2_compose
#define G2_KS 4                   // k-steps of 32 opposite-side nodes (K <= 128)
#define G2_TP (G2_NR * 32 + 4)    // pitch of a wave's 16-row T' tile (backward)
#define G2_XP 36                  // pitch of a wave's 16-row x / dPre / h tile
#define G2_FXTAG 7                // exchange index of the centre-node readout
// (layout of the staged weight images: g2_image.h)

// phase clocks (debug aid, IGMC_GS_TIMING=1; igmc_debug_g2_clocks): thread 0 of workgroup 0 (member 0, user side) -> slots
// 0..39 (fine stamps of its wave 0: 40..63), thread 0 of member 2 of the same subgraph (item side) -> slots 64..103
__device__ unsigned long long g_g2_clk[128];
// ... and (same switch) start / end of EVERY workgroup on the constant-rate wall clock + its XCC id: [wg][0..2]
__device__ unsigned long long g_g2_wg[1024][3];
#include "g2_prims.h"      // the gfx950 primitives of this file (and their stand-ins for the CPU emulation build)

// ---- the relation-space aggregate of one bundle on the matrix cores, ONE 16-column feature tile of it (the wave's half
//      hf of the pair): acc[r] (lane = row, regs = features 16 hf + 4 (lane >> 4) + 0..3) = sum over the opposite side's
//      nodes of A_r[row][node] * x[node][feature].  All G2_NR relations always run (fragments of absent relations are
//      zero); the three plane fragments of k-step s + 1 are requested before the 15 MFMAs of k-step s are issued.
//      frag(r, s) yields the A_r fragment of k-step s (resident registers, or expanded from the block bytes).
template <typename Frag>
__device__ __forceinline__ void g2_gather_h(const uint32_t* pl, int kp, int nks, Frag frag, int li, int kq, int hf, f32x4 (&acc)[G2_NR]) {
#pragma unroll
  for (int r = 0; r < G2_NR; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tstride = 32 * kp >> 1;            // dwords per term
  const uint32_t* base = pl + ((16 * hf + li) * kp >> 1) + 4 * kq;
  u32x4 pf[2][G2_NT];
  auto request = [&](int s, int buf) {
#pragma unroll
    for (int sp = 0; sp < G2_NT; ++sp) pf[buf][sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
  };
  request(0, 0);
#pragma unroll
  for (int s = 0; s < G2_KS; ++s) {
    if (s < nks) {
      if (s + 1 < G2_KS) request(s + 1, (s + 1) & 1);      // unconditional (a k-step past the side reads bytes that are
      G2_SCHED_BARRIER();                                  // never used): a guarded request is sunk below the MFMAs
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) {
        const u32x4 fr = frag(r, s);         // formed ONCE per (relation, k-step): 8 VALU ops for the three terms' MFMAs
#pragma unroll
        for (int q = 0; q < G2_NT; ++q) acc[r] = g2_mfma_bf16(pf[s & 1][q], fr, acc[r]);
      }
      G2_SCHED_BARRIER();
    }
  }
}

// This wave's HALF of the dense transform  out = [T_0..T_4 | x] @ [W_0; ..; W_4; root]: the K slots that belong to its
// feature tile -- input features 16 hf + 0..15 of every relation block and of the bundle's own rows x -- against BOTH
// output tiles; the pair's two partial outputs are summed through LDS by the caller.  Same arithmetic as g2_transform
// (both operands as three bf16 terms, six products), one 16x16x32 step = TWO blocks' halves: (T_0, T_1), (T_2, T_3),
// (T_4, x).  The gather's accumulators are the A operand (lane (row li, kq): features 16 hf + 4 kq + 0..3 of a block = four
// k-slots); the B operand takes the matching 8-byte half (dwords 2 hf, 2 hf + 1) of the lane's 16-byte fragment of each of
// the step's two blocks in the staged image (g2_image.h: elements 0..3 = rows 4 kq + e, 4..7 = rows 16 + 4 kq + e - 4).
__device__ __forceinline__ void g2_transform_h(const f32x4 (&acc)[G2_NR], const float* xrows, const uint32_t* sW, int li, int kq, int hf,
                                               f32x4 (&o)[2]) {
  f32x4 o0a = (f32x4){0.f, 0.f, 0.f, 0.f}, o0b = o0a, o1a = o0a, o1b = o0a;
  const float4 xv = *(const float4*)(xrows + li * G2_XP + 16 * hf + 4 * kq);
  const uint2* wf = (const uint2*)sW + 2 * (kq * 16 + li) + hf;     // the lane's half fragment inside a [64]-lane group
  uint2 bf[2][2 * G2_NT][2];                                        // [buffer][2 term + nt][block of the step]
  auto request = [&](int st, int buf) {
#pragma unroll
    for (int t = 0; t < G2_NT; ++t)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) bf[buf][2 * t + nt][b2] = wf[((t * (G2_NR + 1) + 2 * st + b2) * 2 + nt) * 128];
  };
  request(0, 0);
#pragma unroll
  for (int st = 0; st < (G2_NR + 1) / 2; ++st) {
    if (st + 1 < (G2_NR + 1) / 2) request(st + 1, (st + 1) & 1);
    float v[8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      v[rr] = acc[2 * st][rr];
      v[4 + rr] = (2 * st + 1 < G2_NR) ? acc[(2 * st + 1 < G2_NR) ? 2 * st + 1 : 0][rr] : (rr == 0 ? xv.x : rr == 1 ? xv.y : rr == 2 ? xv.z : xv.w);
    }
    u32x4 ah, am, al;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t h, mi, lo;
      g2_split2(v[2 * q], v[2 * q + 1], h, mi, lo);
      ah[q] = h;
      am[q] = mi;
      al[q] = lo;
    }
    G2_SCHED_BARRIER();
    u32x4 b[2 * G2_NT];
#pragma unroll
    for (int q = 0; q < 2 * G2_NT; ++q) b[q] = (u32x4){bf[st & 1][q][0].x, bf[st & 1][q][0].y, bf[st & 1][q][1].x, bf[st & 1][q][1].y};
    o0a = g2_mfma_bf16(ah, b[0], o0a);
    o1a = g2_mfma_bf16(ah, b[1], o1a);
    o0b = g2_mfma_bf16(ah, b[2], o0b);
    o1b = g2_mfma_bf16(ah, b[3], o1b);
    o0a = g2_mfma_bf16(am, b[0], o0a);
    o1a = g2_mfma_bf16(am, b[1], o1a);
    o0b = g2_mfma_bf16(ah, b[4], o0b);
    o1b = g2_mfma_bf16(ah, b[5], o1b);
    o0a = g2_mfma_bf16(al, b[0], o0a);
    o1a = g2_mfma_bf16(al, b[1], o1a);
    o0b = g2_mfma_bf16(am, b[2], o0b);
    o1b = g2_mfma_bf16(am, b[3], o1b);
    G2_SCHED_BARRIER();
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    o[0][rr] = o0a[rr] + o0b[rr];
    o[1][rr] = o1a[rr] + o1b[rr];
  }
}

// out (lane = output feature 16 nt + li, regs = rows 4 kq + 0..3) = [T_0..T_4 | x] @ [W_0; ..; W_4; root] on the bf16 matrix
// cores at f32 accuracy: both operands as three bf16 terms, the six products down to 2^-24 (hi*hi, hi*mid, mid*hi, hi*lo,
// lo*hi, mid*mid; f32 MFMA runs at 1/16 of this rate: 96 x 32 cycles per bundle against 72 x 17 here).  The gather's
// accumulators ARE the A operand: lane (row li, kq) holds the input features 4 kq + 0..3 and 16 + 4 kq + 0..3 of every
// relation block = the 8 k-slots of one 16x16x32 step; the staged image (k_g2_compose) orders the weights the same way.
// Block G2_NR = the layer's own rows x, read from the bundle's LDS tile.  The six B fragments of block g + 1 are
// requested before the 12 MFMAs of block g.
__device__ __forceinline__ void g2_transform(const f32x4 (&acc)[G2_NR][2], const float* xrows, const uint32_t* sW, int li, int kq,
                                             f32x4 (&o)[2]) {
  f32x4 o0a = (f32x4){0.f, 0.f, 0.f, 0.f}, o0b = o0a, o1a = o0a, o1b = o0a;
  const float4 x0 = *(const float4*)(xrows + li * G2_XP + 4 * kq), x1 = *(const float4*)(xrows + li * G2_XP + 16 + 4 * kq);
  const u32x4* wf = (const u32x4*)sW + (kq * 16 + li);             // lane's fragment inside a [64]-lane group
  u32x4 bf[2][2 * G2_NT];
  auto request = [&](int g, int buf) {
#pragma unroll
    for (int t = 0; t < G2_NT; ++t) {
      bf[buf][2 * t] = wf[((t * (G2_NR + 1) + g) * 2 + 0) * 64];
      bf[buf][2 * t + 1] = wf[((t * (G2_NR + 1) + g) * 2 + 1) * 64];
    }
  };
  request(0, 0);
#pragma unroll
  for (int g = 0; g <= G2_NR; ++g) {
    if (g < G2_NR) request(g + 1, (g + 1) & 1);
    float v[8];
    if (g < G2_NR) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        v[rr] = acc[g][0][rr];
        v[4 + rr] = acc[g][1][rr];
      }
    } else {
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
      v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
    u32x4 ah, am, al;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t h, mi, lo;
      g2_split2(v[2 * q], v[2 * q + 1], h, mi, lo);
      ah[q] = h;
      am[q] = mi;
      al[q] = lo;
    }
    G2_SCHED_BARRIER();
    const u32x4 (&b)[2 * G2_NT] = bf[g & 1];        // [2 term + nt]
    o0a = g2_mfma_bf16(ah, b[0], o0a);
    o1a = g2_mfma_bf16(ah, b[1], o1a);
    o0b = g2_mfma_bf16(ah, b[2], o0b);
    o1b = g2_mfma_bf16(ah, b[3], o1b);
    o0a = g2_mfma_bf16(am, b[0], o0a);
    o1a = g2_mfma_bf16(am, b[1], o1a);
    o0b = g2_mfma_bf16(ah, b[4], o0b);
    o1b = g2_mfma_bf16(ah, b[5], o1b);
    o0a = g2_mfma_bf16(al, b[0], o0a);
    o1a = g2_mfma_bf16(al, b[1], o1a);
    o0b = g2_mfma_bf16(am, b[2], o0b);
    o1b = g2_mfma_bf16(am, b[3], o1b);
    G2_SCHED_BARRIER();
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    o[0][rr] = o0a[rr] + o0b[rr];
    o[1][rr] = o1a[rr] + o1b[rr];
  }
}

#include "g2_subgraph.h"
#include "g2_compose.h"
#include "dl_kernels.h"
