// model.h -- device-side view of the IGMC model geometry + workspace (internal).
#pragma once
#include "common.h"

#define IGMC_KCAT 160     // (num_bases + 1) * 32 : [basis-space aggregate | self] width
#ifndef IGMC_WG_BLOCKS
#define IGMC_WG_BLOCKS 64    // grid.x of the weight-gradient kernel (per-block partials, per layer); 128: ml_100k +1.7 %, flixster -2 %; 32: +9 us in the products
#endif
#define IGMC_STASH_LAYER 672  // floats per conv layer of the weights-only stash (fin_stash; layout: model.hip)
#define IGMC_TS_BLOCKS 256   // partial slots of the relation-space tables (one per workgroup of k_graph_step)
#define IGMC_GATHER_BLOCKS 4096   // max grid of the row-walker kernels (4 rows = 4 waves per block)
#define IGMC_L0_BLOCKS 256
#define IGMC_HG 8            // graphs per workgroup in the head kernels

struct ModelDev {
  int R, Bs, L, S, D;           // relations, bases(=4), node labels (2h+2), side features, lin1 fan-in
  int64_t off_basis[4], off_root[4], off_bias[4], off_att[4];
  int64_t off_l1w, off_l1b, off_l2w, off_l2b, n_params;
  int node_cap, edge_cap, graph_cap;
  // activations / scratch (all fp32 unless noted)
  float* h[4];        // [Ncap,32]   layer outputs tanh(conv_l(...))      reference models.py:199-202
  float* agg;         // [Ncap,128]  basis-space aggregate of the forward pass (scratch, reused per layer)
  float* gagg[3];     // [Ncap,128]  backward: basis-space aggregate of dPre for layers 1..3
  float* Y[3];        // [Ncap,128]  h_{l-1} @ [basis_0|..|basis_3] for layers 1..3 (att gradient)
  float* dpre[4];     // [Ncap,32]   dLoss/d(pre-activation of layer l)
  float* feat;        // [Bcap,D]    centre-node readout          reference models.py:205-209
  float* a1;          // [Bcap,128]  relu(lin1(feat))
  uint8_t* lmask;     // [Bcap,128]  keep mask of the 0.5 dropout
  float* dz;          // [Bcap,128]
  float* gfeat;       // [Bcap,D]
  float* err;         // [Bcap]      out - y
  uint16_t* cnt0;     // [Ncap,R*L]  per-node histogram of kept in-edge codes (layer-0 weight gradient)
  // gradient partials
  float* wg_part;     // [4][IGMC_WG_BLOCKS][32*160+32]  (slice 3 = layer-0 table when it has <= 32 rows)
  float* gatt_part;   // [3][IGMC_GATHER_BLOCKS][R*4]
  float* l0_part;     // [IGMC_L0_BLOCKS][(R*L+L+1)*32]
  float* graw;        // [3][32*160+32] + [3][R*4] + l0 rows: reduced partials
  float* ts_part;     // [4][IGMC_TS_BLOCKS][ts_stride] relation-space tables [W_r rows | root rows | bias] per layer (or NULL)
  float* ts_raw;      // [4][ts_stride] their sum over the workgroups
  int ts_stride;      // (R*32 + 33) * 32
  float* fin_stash;   // [4][IGMC_STASH_LAYER] per conv layer: Gram of the bases [0..15], ARR matrix M [16..31], att moments [32..160), att copy [160..160+R*4);
                      // then [16] Adam scalars of the step -- written by k_tail_ts, read by k_finalize_ts (or NULL)
  float* datt_part;   // [4*ts_stride/32][4] partial <dW_r, basis_b> products of 32 table elements (k_tail_ts)
  int* gs_bar;        // k_graph_step clusters: [0] workgroups that finished the launch, [1] launch sequence number
  int* gs_err;        // [1] set when a cluster exchange timed out
  unsigned long long* gs_ts;   // [4] device-side launch clock of k_graph_step (igmc_profile_enable(2)): [0] earliest workgroup
                               // start of the running launch (wall clock ticks), [1] sum of launch durations, [2] launches,
                               // [3] workgroups that finished the running launch
  unsigned long long* g2_ex;   // graphstep2: [5 exchanges][g2_graphs][2 sides][32 features][128 nodes] {hi, mid, lo, tag} words
  size_t g2_ex_stride;         // words per exchange
  unsigned long long* g2_fx;   // [g2_graphs][256] {f32, tag} words of the centre-node readout
  unsigned char* g2_px;        // k_graph_step2's plane exchange: [5 exchanges][g2_graphs][2 sides][32 KB] (g2_prims.h); null: not allocated
  size_t g2_px_stride;         // bytes per exchange
  int g2_graphs;               // subgraph slots of the two buffers above (0: not allocated)
  float* g2_w;                 // [6 images of (5*32+32) x 20 float2 | 1024] composed weights of the step (k_g2_compose)
  int ex_nodes;                // nodes a side the exchange regions of g2_ex hold: 128 (k_graph_step2) or 256 (k_dl_fwd)
  int img_current;             // host side, per call: g2_w holds the images of the call's parameters (compose is skipped)
  const float* adam_m1;        // per call (fused step with Adam): the moments, for the stash of att's old moments
  const float* adam_m2;
  float* arr_part;    // [4] ARR regulariser per layer
  const float* dcat[3];   // sort-pool readout (sortpool.hip): dense d loss / d h_l [Ncap,32] of layers 0..2 added to EVERY row
                          // in the conv backward; NULL = centre-node readout (gfeat on the two target rows)
  const float* side;  // [B,S] borrowed side features or NULL
  const int64_t* ctrl;  // optional device-side step control (igmc_hip.h) or NULL
};

static inline int igmc_wg_stride() { return 32 * IGMC_KCAT + 32; }

// ---- small device-side reductions shared by model.hip / graphstep.hip
__device__ __forceinline__ float igmc_wave_sum_f(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// sum over the 16 lanes of a DPP row (= one 16-lane group), result in every lane of the row.
// On gfx950 this is four VALU adds with the row_ror DPP modifier (no LDS-crossbar round trips).
__device__ __forceinline__ float igmc_group16_sum_f(float v) {
#ifdef IGMC_HIPEMU
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 16);
  return v;
#else
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
#endif
}
__device__ __forceinline__ float igmc_block_sum_f(float v, float* sm) {
  v = igmc_wave_sum_f(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float tot = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) tot += sm[w];
  __syncthreads();
  return tot;
}

