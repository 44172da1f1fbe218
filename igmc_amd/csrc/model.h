// model.h -- device-side view of the IGMC model geometry + workspace (internal).
#pragma once
#include "common.h"

#define IGMC_KCAT 160     // (num_bases + 1) * 32 : [basis-space aggregate | self] width
#define IGMC_WG_BLOCKS 128   // grid.x of the weight-gradient kernel (per-block partials, per layer)
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
  float* arr_part;    // [4] ARR regulariser per layer
  const float* side;  // [B,S] borrowed side features or NULL
  const int64_t* ctrl;  // optional device-side step control (igmc_hip.h) or NULL
};

static inline int igmc_wg_stride() { return 32 * IGMC_KCAT + 32; }
