// sortpool.hip -- the sort-pool readout family (reference models.py:63-167, DGCNN / DGCNN_RS):
//
//     concat = [h_0 | h_1 | h_2 | h_3]            (97 channels: latent_dim = [32, 32, 32, 1])
//     x = global_sort_pool(concat, batch, k)      (PyG 1.4.2: per graph, nodes sorted by the LAST channel, descending,
//                                                  the first k rows, zero rows where the graph has fewer nodes)
//     x = relu(Conv1d(1, 16, 97, 97)(x))          = one 97 -> 16 linear map per pooled node row
//     x = MaxPool1d(2, 2)(x)
//     x = relu(Conv1d(16, 32, 5, 1)(x))
//     x = relu(lin1(flatten(x)));  dropout(0.5);  out = lin2(x)[:, 0]
//
// on top of the four R-GCN layers, which are the per-layer kernels of model.hip unchanged: the fourth layer (32 -> 1)
// runs as a 32 -> 32 layer whose weights are zero beyond output column 0 (k_sp_pack), so h_3[:, 0] is the reference's
// h_3 and the other columns are tanh(0) = 0; their parameter gradients are exactly zero (nothing reads those columns).
//
// One workgroup per subgraph for the readout forward (k_sp_fwd) and for its backward (k_sp_bwd); the weight gradients
// that sum over the batch are formed by k_sp_wgrad in a fixed order (per-graph partials of the two small convolutions,
// a batched product for lin1) -- no float atomics, bit-reproducible.  The readout's gradient w.r.t. the node states is
// DENSE (up to k nodes per graph, all 97 channels): it is written as three [N, 32] arrays + dPre_3 and added to every row
// by the conv backward (ModelDev::dcat).
#include "launch.h"
#include "sortpool.h"
#include <stdio.h>

#define SP_THREADS 256
#define SP_C 97          // channels of a pooled row
#define SP_C1 16         // conv1 output channels
#define SP_C2 32         // conv2 output channels
#define SP_KW 5          // conv2 kernel width

// ---------------------------------------------------------------------------------------------- parameter layouts
// engine layout (model.h: every layer 32 wide) <- true layout (layer 3 has ONE output column)
__global__ __launch_bounds__(SP_THREADS) void k_sp_pack(ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                         float* __restrict__ pe) {
  const int64_t n = m.off_l1w;                       // conv parameters of the engine layout
  const int64_t o3 = m.off_basis[3];                 // layers 0..2 sit at identical offsets in both layouts
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    float v;
    if (i < o3) v = Pd[i];
    else if (i < m.off_root[3]) {                    // basis_3[b][k][o]
      const int64_t e = i - o3;
      v = ((e & 31) == 0) ? Pd[sp.t_basis3 + (e >> 5)] : 0.f;
    } else if (i < m.off_bias[3]) {                  // root_3[k][o]
      const int64_t e = i - m.off_root[3];
      v = ((e & 31) == 0) ? Pd[sp.t_root3 + (e >> 5)] : 0.f;
    } else if (i < m.off_att[3]) {                   // bias_3[o]
      v = (i == m.off_bias[3]) ? Pd[sp.t_bias3] : 0.f;
    } else v = Pd[sp.t_att3 + (i - m.off_att[3])];
    pe[i] = v;
  }
}

// true-layout gradient of the conv parameters <- engine-layout gradient
__global__ __launch_bounds__(SP_THREADS) void k_sp_unpack(ModelDev m, SpDev sp, const float* __restrict__ ge,
                                                           float* __restrict__ Gd) {
  const int64_t n = sp.t_conv_end;
  const int64_t o3 = m.off_basis[3];
  for (int64_t i = (int64_t)blockIdx.x * SP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * SP_THREADS) {
    float v;
    if (i < o3) v = ge[i];
    else if (i < sp.t_root3) v = ge[m.off_basis[3] + (i - sp.t_basis3) * 32];
    else if (i < sp.t_bias3) v = ge[m.off_root[3] + (i - sp.t_root3) * 32];
    else if (i < sp.t_att3) v = ge[m.off_bias[3]];
    else v = ge[m.off_att[3] + (i - sp.t_att3)];
    Gd[i] = v;
  }
}

// one channel of the concatenated state of node i
__device__ __forceinline__ float sp_cat(const ModelDev& m, int i, int j) {
  return (j < 96) ? m.h[j >> 5][(size_t)i * 32 + (j & 31)] : m.h[3][(size_t)i * 32];
}

// descending by key, ties by ascending node index (torch.sort(stable=True, descending=True) order)
__device__ __forceinline__ bool sp_before(float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia < ib); }

// the k first nodes of graph g in sort-pool order -> sel[0..k) (batch-wide node index, -1 = padding); LDS: keys / idx [P]
__device__ void sp_select(const BatchDev& b, const ModelDev& m, const SpDev& sp, int g, float* keys, int* idx, int* sel_out) {
  const int tid = threadIdx.x;
  const int n0 = b.node_off[g], n = b.node_off[g + 1] - n0;
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += SP_THREADS) {
    keys[i] = (i < n) ? m.h[3][(size_t)(n0 + i) * 32] : -3.0e38f;
    idx[i] = (i < n) ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += SP_THREADS) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);                  // this sub-sequence is sorted "before-first"
        const float ka = keys[lo], kb = keys[hi];
        const int ia = idx[lo], ib = idx[hi];
        const bool swap = up ? sp_before(kb, ib, ka, ia) : sp_before(ka, ia, kb, ib);
        if (swap) {
          keys[lo] = kb; keys[hi] = ka;
          idx[lo] = ib; idx[hi] = ia;
        }
      }
      __syncthreads();
    }
  }
  for (int p = tid; p < sp.k; p += SP_THREADS) sel_out[p] = (p < n) ? n0 + idx[p] : -1;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- forward
// dynamic LDS: keys[P] | idx[P] | sel[k] | w1[16*97+16] | w2[32*80+32] | y1[16*k] | z[16*Q1] | y2[32*Q2] | a1[128] | red[8]
template <bool TRAIN>
__global__ __launch_bounds__(SP_THREADS) void k_sp_fwd(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                        const uint8_t* __restrict__ inj_mask, uint64_t seed,
                                                        uint64_t step_arg, float* __restrict__ out) {
  IGMC_DYN_SMEM(smem);
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = sp.k, Q1 = sp.Q1, Q2 = sp.Q2, dense = sp.dense, P = sp.P;
  float* keys = (float*)smem;
  int* idx = (int*)(keys + P);
  int* sel = idx + P;
  float* w1 = (float*)(sel + k);
  float* w2 = w1 + SP_C1 * SP_C + SP_C1;
  float* y1 = w2 + SP_C2 * SP_C1 * SP_KW + SP_C2;
  float* z = y1 + SP_C1 * k;
  float* y2 = z + SP_C1 * Q1;
  float* a1s = y2 + SP_C2 * Q2;
  float* red = a1s + 128;
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  for (int i = tid; i < SP_C1 * SP_C + SP_C1; i += SP_THREADS) w1[i] = Pd[sp.t_c1w + i];      // weight [16][97] then bias [16]
  for (int i = tid; i < SP_C2 * SP_C1 * SP_KW + SP_C2; i += SP_THREADS) w2[i] = Pd[sp.t_c2w + i];
  sp_select(b, m, sp, g, keys, idx, sel);
  for (int p = tid; p < k; p += SP_THREADS) sp.sel[(size_t)g * k + p] = sel[p];
  // conv1 (one pooled row per thread: its 97 channels are read once) + ReLU
  for (int p = tid; p < k; p += SP_THREADS) {
    const int i = sel[p];
    float acc[SP_C1];
#pragma unroll
    for (int oc = 0; oc < SP_C1; ++oc) acc[oc] = w1[SP_C1 * SP_C + oc];
    if (i >= 0) {
      for (int j = 0; j < SP_C; ++j) {
        const float x = sp_cat(m, i, j);
#pragma unroll
        for (int oc = 0; oc < SP_C1; ++oc) acc[oc] += w1[oc * SP_C + j] * x;
      }
    }
#pragma unroll
    for (int oc = 0; oc < SP_C1; ++oc) {
      const float v = acc[oc] > 0.f ? acc[oc] : 0.f;
      y1[oc * k + p] = v;
      if (TRAIN) sp.y1[((size_t)g * SP_C1 + oc) * k + p] = v;
    }
  }
  __syncthreads();
  for (int i = tid; i < SP_C1 * Q1; i += SP_THREADS) {            // MaxPool1d(2, 2)
    const int ic = i / Q1, q = i - ic * Q1;
    const float a = y1[ic * k + 2 * q], c = y1[ic * k + 2 * q + 1];
    z[i] = a >= c ? a : c;
  }
  __syncthreads();
  for (int i = tid; i < SP_C2 * Q2; i += SP_THREADS) {            // conv2 + ReLU; flatten index = oc2 * Q2 + q
    const int oc = i / Q2, q = i - oc * Q2;
    float acc = w2[SP_C2 * SP_C1 * SP_KW + oc];
    for (int ic = 0; ic < SP_C1; ++ic)
#pragma unroll
      for (int t = 0; t < SP_KW; ++t) acc += w2[(oc * SP_C1 + ic) * SP_KW + t] * z[ic * Q1 + q + t];
    const float v = acc > 0.f ? acc : 0.f;
    y2[i] = v;
    if (TRAIN) sp.flat[(size_t)g * dense + i] = v;
  }
  __syncthreads();
  // lin1 (dense -> 128): wave w takes units w, w + 4, ..; lanes stride over the fan-in (coalesced rows)
  for (int j = wave; j < 128; j += SP_THREADS / 64) {
    const float* wrow = Pd + sp.t_l1w + (size_t)j * dense;
    float s = 0.f;
    for (int i = lane; i < dense; i += 64) s += wrow[i] * y2[i];
    s = igmc_wave_sum_f(s);
    if (lane == 0) {
      float av = s + Pd[sp.t_l1b + j];
      av = av > 0.f ? av : 0.f;
      int keep = 1;
      if (TRAIN) {
        keep = inj_mask ? (int)inj_mask[g * 128 + j]
                        : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)j)) >= 0.5f);
        sp.a1[g * 128 + j] = av;
        sp.lmask[g * 128 + j] = (uint8_t)keep;
      }
      a1s[j] = (TRAIN ? (keep ? av * 2.f : 0.f) : av) * Pd[sp.t_l2w + j];        // F.dropout(p = 0.5): kept * 2
    }
  }
  __syncthreads();
  if (wave == 0) {
    float s = a1s[lane] + a1s[lane + 64];
    s = igmc_wave_sum_f(s);
    if (lane == 0) {
      const float o = s + Pd[sp.t_l2b];
      out[g] = o;
      m.err[g] = o - b.y[g];
    }
  }
  (void)red;
}

// ---------------------------------------------------------------------------------------------- backward
// dynamic LDS: sel[k] | rank[nmax] | w1[16*97] | w2[32*80] | y1[16*k] | z[16*Q1] | dy2[32*Q2] | dzp[16*Q1] | dy1[16*k] | dz[128]
__global__ __launch_bounds__(SP_THREADS) void k_sp_bwd(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                        float grad_scale) {
  IGMC_DYN_SMEM(smem);
  const int g = blockIdx.x, tid = threadIdx.x;
  const int k = sp.k, Q1 = sp.Q1, Q2 = sp.Q2, dense = sp.dense;
  const int n0 = b.node_off[g], n = b.node_off[g + 1] - n0;
  int* sel = (int*)smem;
  int* rank = sel + k;
  float* w1 = (float*)(rank + sp.nmax);
  float* w2 = w1 + SP_C1 * SP_C;
  float* y1 = w2 + SP_C2 * SP_C1 * SP_KW;
  float* z = y1 + SP_C1 * k;
  float* dy2 = z + SP_C1 * Q1;
  float* dzp = dy2 + SP_C2 * Q2;
  float* dy1 = dzp + SP_C1 * Q1;
  float* dzs = dy1 + SP_C1 * k;
  for (int i = tid; i < SP_C1 * SP_C; i += SP_THREADS) w1[i] = Pd[sp.t_c1w + i];
  for (int i = tid; i < SP_C2 * SP_C1 * SP_KW; i += SP_THREADS) w2[i] = Pd[sp.t_c2w + i];
  for (int i = tid; i < n; i += SP_THREADS) rank[i] = -1;
  for (int p = tid; p < k; p += SP_THREADS) sel[p] = sp.sel[(size_t)g * k + p];
  for (int i = tid; i < SP_C1 * k; i += SP_THREADS) y1[i] = sp.y1[(size_t)g * SP_C1 * k + i];
  const float dout = 2.f * m.err[g] * grad_scale;                  // d (mean squared error) / d out
  if (tid == 0) sp.dout[g] = dout;
  if (tid < 128) {
    const float av = sp.a1[g * 128 + tid];
    const float dzv = (av > 0.f && sp.lmask[g * 128 + tid]) ? dout * Pd[sp.t_l2w + tid] * 2.f : 0.f;
    dzs[tid] = dzv;
    sp.dz[g * 128 + tid] = dzv;
  }
  __syncthreads();
  for (int p = tid; p < k; p += SP_THREADS)
    if (sel[p] >= 0) rank[sel[p] - n0] = p;
  for (int i = tid; i < SP_C1 * Q1; i += SP_THREADS) {
    const int ic = i / Q1, q = i - ic * Q1;
    const float a = y1[ic * k + 2 * q], c = y1[ic * k + 2 * q + 1];
    z[i] = a >= c ? a : c;
  }
  // d flat = dz @ lin1.weight, through conv2's ReLU
  for (int i = tid; i < dense; i += SP_THREADS) {
    float s = 0.f;
    for (int j = 0; j < 128; ++j) {
      const float dzv = dzs[j];
      if (dzv != 0.f) s += dzv * Pd[sp.t_l1w + (size_t)j * dense + i];
    }
    dy2[i] = (sp.flat[(size_t)g * dense + i] > 0.f) ? s : 0.f;
  }
  __syncthreads();
  // conv2: weight / bias gradient of this graph, gradient w.r.t. the pooled sequence
  float* pc2 = sp.part_c2 + (size_t)g * (SP_C2 * SP_C1 * SP_KW + SP_C2);
  for (int i = tid; i < SP_C2 * SP_C1 * SP_KW; i += SP_THREADS) {
    const int oc = i / (SP_C1 * SP_KW), rem = i - oc * (SP_C1 * SP_KW), ic = rem / SP_KW, t = rem - ic * SP_KW;
    float s = 0.f;
    for (int q = 0; q < Q2; ++q) s += dy2[oc * Q2 + q] * z[ic * Q1 + q + t];
    pc2[i] = s;
  }
  for (int oc = tid; oc < SP_C2; oc += SP_THREADS) {
    float s = 0.f;
    for (int q = 0; q < Q2; ++q) s += dy2[oc * Q2 + q];
    pc2[SP_C2 * SP_C1 * SP_KW + oc] = s;
  }
  for (int i = tid; i < SP_C1 * Q1; i += SP_THREADS) {
    const int ic = i / Q1, qp = i - ic * Q1;
    float s = 0.f;
    for (int oc = 0; oc < SP_C2; ++oc)
#pragma unroll
      for (int t = 0; t < SP_KW; ++t) {
        const int q = qp - t;
        if (q >= 0 && q < Q2) s += w2[(oc * SP_C1 + ic) * SP_KW + t] * dy2[oc * Q2 + q];
      }
    dzp[i] = s;
  }
  __syncthreads();
  // max-pool (the first maximum takes the gradient, like torch) and conv1's ReLU
  for (int i = tid; i < SP_C1 * k; i += SP_THREADS) {
    const int oc = i / k, p = i - oc * k, q = p >> 1;
    float d = 0.f;
    if (q < Q1) {
      const float a = y1[oc * k + 2 * q], c = y1[oc * k + 2 * q + 1];
      const bool mine = (p & 1) ? (c > a) : (a >= c);
      if (mine && y1[i] > 0.f) d = dzp[oc * Q1 + q];
    }
    dy1[i] = d;
  }
  __syncthreads();
  // conv1: weight / bias gradient of this graph
  float* pc1 = sp.part_c1 + (size_t)g * (SP_C1 * SP_C + SP_C1);
  for (int i = tid; i < SP_C1 * SP_C; i += SP_THREADS) {
    const int oc = i / SP_C, j = i - oc * SP_C;
    float s = 0.f;
    for (int p = 0; p < k; ++p) {
      const float d = dy1[oc * k + p];
      if (d != 0.f && sel[p] >= 0) s += d * sp_cat(m, sel[p], j);
    }
    pc1[i] = s;
  }
  for (int oc = tid; oc < SP_C1; oc += SP_THREADS) {
    float s = 0.f;
    for (int p = 0; p < k; ++p) s += dy1[oc * k + p];
    pc1[SP_C1 * SP_C + oc] = s;
  }
  // gradient w.r.t. the node states: every node of the graph is written exactly once (zeros when it was not pooled)
  for (int i = tid; i < n * 128; i += SP_THREADS) {
    const int node = i >> 7, c = i & 127;             // c < 96: channel of h_0..h_2; c >= 96: column c - 96 of dPre_3
    const int p = rank[node];
    const size_t row = (size_t)(n0 + node) * 32;
    float v = 0.f;
    if (c < 96) {
      if (p >= 0)
        for (int oc = 0; oc < SP_C1; ++oc) v += dy1[oc * k + p] * w1[oc * SP_C + c];
      sp.dcat[c >> 5][row + (c & 31)] = v;
    } else {
      if (p >= 0 && c == 96) {
        for (int oc = 0; oc < SP_C1; ++oc) v += dy1[oc * k + p] * w1[oc * SP_C + 96];
        const float hv = m.h[3][row];
        v *= 1.f - hv * hv;
      }
      m.dpre[3][row + (c - 96)] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradients over the batch
// blocks [0, nb1): lin1.weight tiles (one element per thread: sum over the graphs of dz[g][j] * flat[g][i]);
// then one block each for: conv1 partial sums, conv2 partial sums, lin1.bias + lin2
__global__ __launch_bounds__(SP_THREADS) void k_sp_wgrad(BatchDev b, SpDev sp, int B, int nb1, float* __restrict__ Gd) {
  const int tid = threadIdx.x;
  const int dense = sp.dense;
  if ((int)blockIdx.x < nb1) {
    const int64_t e = (int64_t)blockIdx.x * SP_THREADS + tid;
    if (e < (int64_t)128 * dense) {
      const int j = (int)(e / dense), i = (int)(e - (int64_t)j * dense);
      float s = 0.f;
      for (int g = 0; g < B; ++g) {
        const float dzv = sp.dz[g * 128 + j];
        if (dzv != 0.f) s += dzv * sp.flat[(size_t)g * dense + i];
      }
      Gd[sp.t_l1w + e] = s;
    }
    return;
  }
  const int role = blockIdx.x - nb1;
  if (role == 0) {
    const int n1 = SP_C1 * SP_C + SP_C1;
    for (int i = tid; i < n1; i += SP_THREADS) {
      float s = 0.f;
      for (int g = 0; g < B; ++g) s += sp.part_c1[(size_t)g * n1 + i];
      Gd[sp.t_c1w + i] = s;
    }
  } else if (role == 1) {
    const int n2 = SP_C2 * SP_C1 * SP_KW + SP_C2;
    for (int i = tid; i < n2; i += SP_THREADS) {
      float s = 0.f;
      for (int g = 0; g < B; ++g) s += sp.part_c2[(size_t)g * n2 + i];
      Gd[sp.t_c2w + i] = s;
    }
  } else {
    if (tid < 128) {
      float sb = 0.f, sw = 0.f;
      for (int g = 0; g < B; ++g) {
        sb += sp.dz[g * 128 + tid];
        const float av = sp.a1[g * 128 + tid];
        sw += sp.dout[g] * (sp.lmask[g * 128 + tid] ? av * 2.f : 0.f);
      }
      Gd[sp.t_l1b + tid] = sb;
      Gd[sp.t_l2w + tid] = sw;
    }
    if (tid == 128) {
      float s = 0.f;
      for (int g = 0; g < B; ++g) s += sp.dout[g];
      Gd[sp.t_l2b] = s;
    }
  }
  (void)b;
}

// ---------------------------------------------------------------------------------------------- host side
static size_t sp_fwd_lds(const SpDev& sp) {
  return (size_t)(2 * sp.P + sp.k + SP_C1 * SP_C + SP_C1 + SP_C2 * SP_C1 * SP_KW + SP_C2 + SP_C1 * sp.k + SP_C1 * sp.Q1 +
                  SP_C2 * sp.Q2 + 128 + 8) * 4;
}
static size_t sp_bwd_lds(const SpDev& sp) {
  return (size_t)(sp.k + sp.nmax + SP_C1 * SP_C + SP_C2 * SP_C1 * SP_KW + SP_C1 * sp.k + SP_C1 * sp.Q1 + SP_C2 * sp.Q2 +
                  SP_C1 * sp.Q1 + SP_C1 * sp.k + 128) * 4;
}

int igmc_sp_lds_ok(const SpDev& sp) { return sp_fwd_lds(sp) <= 160 * 1024 && sp_bwd_lds(sp) <= 160 * 1024; }

int igmc_sp_prepare(const SpDev& sp) {
#ifndef IGMC_HIPEMU
  const int fw = (int)sp_fwd_lds(sp), bw = (int)sp_bwd_lds(sp);
  if (hipFuncSetAttribute((const void*)k_sp_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, fw) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_sp_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, fw) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_sp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, bw) != hipSuccess) return 1;
#else
  (void)sp;
#endif
  return 0;
}

void igmc_launch_sp_pack(const ModelDev& m, const SpDev& sp, const float* Pd, float* pe, void* stream) {
  IGMC_PLAUNCH("k_sp_pack", k_sp_pack, 64, SP_THREADS, 0, stream, m, sp, Pd, pe);
}

void igmc_launch_sp_forward(const ModelDev& m, const SpDev& sp, const BatchDev& b, const float* Pd, int B, int training,
                            const uint8_t* inj_mask, uint64_t seed, uint64_t step, float* out, void* stream) {
  const size_t sm = sp_fwd_lds(sp);
  if (training) IGMC_PLAUNCH("k_sp_fwd", (k_sp_fwd<true>), B, SP_THREADS, sm, stream, b, m, sp, Pd, inj_mask, seed, step, out);
  else IGMC_PLAUNCH("k_sp_fwd", (k_sp_fwd<false>), B, SP_THREADS, sm, stream, b, m, sp, Pd, inj_mask, seed, step, out);
}

void igmc_launch_sp_backward(const ModelDev& m, const SpDev& sp, const BatchDev& b, const float* Pd, int B, float grad_scale,
                             void* stream) {
  IGMC_PLAUNCH("k_sp_bwd", k_sp_bwd, B, SP_THREADS, sp_bwd_lds(sp), stream, b, m, sp, Pd, grad_scale);
}

void igmc_launch_sp_wgrad(const ModelDev& m, const SpDev& sp, const BatchDev& b, int B, const float* ge, float* Gd,
                          void* stream) {
  const int nb1 = (int)(((int64_t)128 * sp.dense + SP_THREADS - 1) / SP_THREADS);
  IGMC_PLAUNCH("k_sp_wgrad", k_sp_wgrad, nb1 + 3, SP_THREADS, 0, stream, b, sp, B, nb1, Gd);
  IGMC_PLAUNCH("k_sp_unpack", k_sp_unpack, 64, SP_THREADS, 0, stream, m, sp, ge, Gd);
}
