// sortpool.h -- device view + launchers of the sort-pool readout (sortpool.hip; reference models.py:63-167)
#pragma once
#include "model.h"

struct SpDev {
  int k, Q1, Q2, dense;     // pooled rows, rows after MaxPool1d(2,2), after Conv1d(16,32,5), flattened width 32 * Q2
  int P, nmax;              // sort width (power of two >= nmax), largest node count of a subgraph (slot capacity)
  // TRUE parameter layout (one flat buffer; layers 0..2 as in the engine layout, layer 3 with ONE output column)
  int64_t t_basis3, t_root3, t_bias3, t_att3, t_conv_end;
  int64_t t_c1w, t_c2w, t_l1w, t_l1b, t_l2w, t_l2b, n_params;     // conv biases follow their weights
  // per-batch scratch
  int32_t* sel;             // [Bcap][k]      pooled node (batch-wide index) or -1
  float* y1;                // [Bcap][16][k]  relu(conv1)
  float* flat;              // [Bcap][dense]  relu(conv2), flattened [32][Q2]
  float* a1;                // [Bcap][128]    relu(lin1)
  uint8_t* lmask;           // [Bcap][128]
  float* dz;                // [Bcap][128]
  float* dflat;             // [Bcap][dense]  d loss / d flat through conv2's ReLU (k_sp_dflat)
  float* lin_part;          // [8][Bcap]      lin2 partial dot products of the 16-unit tiles (k_sp_lin_fwd)
  int* lin_ctr;             // [Bcap/16 + 1]  arrivals per row tile (zero between launches)
  float* dout;              // [Bcap]
  float* part_c1;           // [Bcap][16*97+16]
  float* part_c2;           // [Bcap][32*16*5+32]
  float* dcat[3];           // [Ncap,32] d loss / d h_l, l = 0..2
};

int igmc_sp_lds_ok(const SpDev& sp);
int igmc_sp_prepare(const SpDev& sp);
void igmc_launch_sp_pack(const ModelDev& m, const SpDev& sp, const float* Pd, float* pe, void* stream);
void igmc_launch_sp_forward(const ModelDev& m, const SpDev& sp, const BatchDev& b, const float* Pd, int B, int training,
                            const uint8_t* inj_mask, uint64_t seed, uint64_t step, float* out, void* stream);
void igmc_launch_sp_backward(const ModelDev& m, const SpDev& sp, const BatchDev& b, const float* Pd, int B, float grad_scale,
                             void* stream);
void igmc_launch_sp_wgrad(const ModelDev& m, const SpDev& sp, const BatchDev& b, int B, const float* ge, float* Gd,
                          void* stream);
