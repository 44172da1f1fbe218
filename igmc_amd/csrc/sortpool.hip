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
#define SP_WG 1024       // threads of the per-subgraph kernels (k_sp_fwd, k_sp_bwd): their loops over LDS are latency-bound
                         // chains -- four waves per SIMD hide what one wave per SIMD exposes
#define SP_C 97          // channels of a pooled row
#define SP_C1 16         // conv1 output channels
#define SP_C2 32         // conv2 output channels
#define SP_KW 5          // conv2 kernel width

// ---------------------------------------------------------------------------------------------- parameter layouts
// engine layout (model.h: every layer 32 wide) <- true layout (layer 3 has ONE output column)
__global__ __launch_bounds__(SP_THREADS) void k_sp_pack(ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                         float* __restrict__ pe) {
  igmc_kernarg_warm<sizeof(ModelDev) + sizeof(SpDev) + 32>();
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
__device__ __forceinline__ void sp_unpack_body(const ModelDev& m, const SpDev& sp, const float* __restrict__ ge,
                                               float* __restrict__ Gd, int blk, int nblk) {
  const int64_t n = sp.t_conv_end;
  const int64_t o3 = m.off_basis[3];
  for (int64_t i = (int64_t)blk * SP_THREADS + threadIdx.x; i < n; i += (int64_t)nblk * SP_THREADS) {
    float v;
    if (i < o3) v = ge[i];
    else if (i < sp.t_root3) v = ge[m.off_basis[3] + (i - sp.t_basis3) * 32];
    else if (i < sp.t_bias3) v = ge[m.off_root[3] + (i - sp.t_root3) * 32];
    else if (i < sp.t_att3) v = ge[m.off_bias[3]];
    else v = ge[m.off_att[3] + (i - sp.t_att3)];
    Gd[i] = v;
  }
}
__global__ __launch_bounds__(SP_THREADS) void k_sp_unpack(ModelDev m, SpDev sp, const float* __restrict__ ge,
                                                           float* __restrict__ Gd) {
  igmc_kernarg_warm<sizeof(ModelDev) + sizeof(SpDev) + 32>();
  sp_unpack_body(m, sp, ge, Gd, blockIdx.x, gridDim.x);
}

// one channel of the concatenated state of node i
__device__ __forceinline__ float sp_cat(const ModelDev& m, int i, int j) {
  return (j < 96) ? m.h[j >> 5][(size_t)i * 32 + (j & 31)] : m.h[3][(size_t)i * 32];
}

// descending by key, ties by ascending node index (torch.sort(stable=True, descending=True) order)
__device__ __forceinline__ bool sp_before(float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia < ib); }

// the k first nodes of graph g in sort-pool order -> sel[0..k) (batch-wide node index, -1 = padding); LDS: keys / idx [P].
// RANK sort: the position of node i is the number of nodes that come before it (the order is total: ties go to the lower
// index), counted by four threads per node over a quarter of the nodes each -- every lane of a wave reads the same key at
// the same time (an LDS broadcast) -- and summed with integer LDS atomics.  n^2 / 1024 = 64 compare steps a thread at
// n = 256, one barrier; the bitonic network it replaces was 36 barrier-separated stages with 128 busy threads.
__device__ void sp_select(const BatchDev& b, const ModelDev& m, const SpDev& sp, int g, float* keys, int* idx, int* sel_out) {
  const int tid = threadIdx.x;
  const int n0 = b.node_off[g], n = b.node_off[g + 1] - n0;
  for (int i = tid; i < n; i += SP_WG) {
    keys[i] = m.h[3][(size_t)(n0 + i) * 32];
    idx[i] = 0;
  }
  for (int p = tid; p < sp.k; p += SP_WG) sel_out[p] = -1;
  __syncthreads();
  const int nr = (n + 63) & ~63, q4 = (n + 3) >> 2;
  for (int e = tid; e < 4 * nr; e += SP_WG) {
    const int part = e / nr, i = e - part * nr;
    if (i < n) {
      const float ki = keys[i];
      const int j0 = part * q4, j1 = (j0 + q4 < n) ? j0 + q4 : n;
      int cnt = 0;
      for (int j = j0; j < j1; ++j) cnt += sp_before(keys[j], j, ki, i) ? 1 : 0;
      if (cnt) atomicAdd(&idx[i], cnt);
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += SP_WG)
    if (idx[i] < sp.k) sel_out[idx[i]] = n0 + i;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- forward
// dynamic LDS: keys[P] | idx[P] | sel[k] | w1[16*97+16] | w2[32*80+32] | y1[16*k] | z[16*Q1] | y2[32*Q2] | a1[128] | red[8] | xs[k*98]
template <bool TRAIN>
__global__ __launch_bounds__(SP_WG) void k_sp_fwd(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                        const uint8_t* __restrict__ inj_mask, uint64_t seed,
                                                        uint64_t step_arg, float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + sizeof(SpDev) + 32>();
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
  for (int i = tid; i < SP_C1 * SP_C + SP_C1; i += SP_WG) w1[i] = Pd[sp.t_c1w + i];      // weight [16][97] then bias [16]
  for (int i = tid; i < SP_C2 * SP_C1 * SP_KW + SP_C2; i += SP_WG) w2[i] = Pd[sp.t_c2w + i];
  sp_select(b, m, sp, g, keys, idx, sel);
  for (int p = tid; p < k; p += SP_WG) sp.sel[(size_t)g * k + p] = sel[p];
  // conv1 + ReLU.  The pooled rows are staged in LDS once (coalesced), then one (row, output channel) per thread: lanes =
  // 16 channels x 4 rows (the row's value is a broadcast, the 16 weights sit in 16 banks).  One pooled row per thread with
  // its 97 channels read from HBM inside the loop left 60 of 1024 threads walking 97 serial round trips.
  float* xs = red + 8;                               // [k][SP_C + 1]
  for (int i = tid; i < k * 128; i += SP_WG) {
    const int p = i >> 7, c = i & 127;
    if (c < SP_C) xs[p * (SP_C + 1) + c] = (sel[p] >= 0) ? sp_cat(m, sel[p], c) : 0.f;
  }
  __syncthreads();
  {   // y1[oc][p] = relu(bias[oc] + sum_j x[p][j] w1[oc][j]): k x 16, K = 97 on the f32 matrix cores, a 16-row tile per wave
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    for (int mt = wave; 16 * mt < k; mt += SP_WG / 64) {
      const int p = 16 * mt + li;                    // A: rows = pooled positions, B: columns = output channels
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int j0 = 0; j0 < SP_C; j0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + 4 * u + kq;
          av[u] = (p < k && j < SP_C) ? xs[p * (SP_C + 1) + j] : 0.f;
          bv[u] = (j < SP_C) ? w1[li * SP_C + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      const float bias = w1[SP_C1 * SP_C + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {                   // D[row 4 kq + r][column li] = position 16 mt + 4 kq + r, channel li
        const int pp = 16 * mt + 4 * kq + r;
        if (pp < k) {
          const float a = acc[r] + bias;
          const float v = a > 0.f ? a : 0.f;
          y1[li * k + pp] = v;
          if (TRAIN) sp.y1[((size_t)g * SP_C1 + li) * k + pp] = v;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < SP_C1 * Q1; i += SP_WG) {            // MaxPool1d(2, 2)
    const int ic = i / Q1, q = i - ic * Q1;
    const float a = y1[ic * k + 2 * q], c = y1[ic * k + 2 * q + 1];
    z[i] = a >= c ? a : c;
  }
  __syncthreads();
  {   // conv2 + ReLU: y2[oc][q] = relu(bias[oc] + sum_(ic, t) w2[oc][ic][t] z[ic][q + t]): 32 x Q2, K = 80; flatten = oc * Q2 + q
    const int lane2 = tid & 63, wave2 = tid >> 6, li = lane2 & 15, kq = lane2 >> 4;
    const int ntq = (Q2 + 15) >> 4;
    for (int tile = wave2; tile < 2 * ntq; tile += SP_WG / 64) {
      const int mt = tile / ntq, nt = tile - mt * ntq;
      const int q = 16 * nt + li;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < SP_C1 * SP_KW; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = k0 + 4 * u + kq, ic = kk / SP_KW, t = kk - ic * SP_KW;
          av[u] = w2[((16 * mt + li) * SP_C1 + ic) * SP_KW + t];
          bv[u] = (q < Q2) ? z[ic * Q1 + q + t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      if (q < Q2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oc = 16 * mt + 4 * kq + r;
          const float a = acc[r] + w2[SP_C2 * SP_C1 * SP_KW + oc];
          const float v = a > 0.f ? a : 0.f;
          y2[oc * Q2 + q] = v;
          sp.flat[(size_t)g * dense + oc * Q2 + q] = v;             // lin1 / lin2 run batched over the subgraphs (k_sp_lin_fwd)
        }
      }
    }
  }
  (void)a1s; (void)red; (void)lane; (void)wave; (void)inj_mask; (void)seed; (void)step; (void)out;
}

// lin1 (dense -> 128) + ReLU + dropout + lin2, batched over the subgraphs on the f32 matrix cores: workgroup = (16 subgraphs)
// x (16 hidden units), its four waves split the fan-in (one round of <= 13 16-wide chunks each: every load of a wave is in
// flight at once), partial accumulators meet in LDS.  lin2 needs all 128 units of a subgraph: every workgroup leaves its
// 16-unit partial dot product, the LAST of the eight of a row tile to finish adds them in index order (a fixed order whoever
// is last).  One workgroup per subgraph streamed the 426 KB of lin1.weight through each: 208 of the 247 us of the old
// k_sp_fwd; four workgroups of 8 waves walking the whole fan-in: 60 us.
template <bool TRAIN>
__global__ __launch_bounds__(SP_THREADS) void k_sp_lin_fwd(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                             const uint8_t* __restrict__ inj_mask, uint64_t seed,
                                                             uint64_t step_arg, float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + sizeof(SpDev) + 32>();
  __shared__ float sacc[4][64][4];
  __shared__ int s_last;
  const int B = b.totals[3], dense = sp.dense;
  const int mt = blockIdx.x, nt = blockIdx.y;
  const int row0 = mt * 16;
  if (row0 >= B) return;
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ga = (row0 + li < B) ? row0 + li : B - 1;
  const int n = nt * 16 + li;
  const float* arow = sp.flat + (size_t)ga * dense + 4 * kq;
  const float* wrow = Pd + sp.t_l1w + (size_t)n * dense + 4 * kq;
  const float b1 = Pd[sp.t_l1b + n], w2 = Pd[sp.t_l2w + n], l2b = Pd[sp.t_l2b];
  const int nch = dense / 16, per = (nch + 3) / 4;
  f32x4 acc4[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  for (int c0 = wave * per; c0 < nch && c0 < (wave + 1) * per; c0 += 13) {
    float4 a4[13], b4[13];
    const int cend = (wave + 1) * per < nch ? (wave + 1) * per : nch;
#pragma unroll
    for (int u = 0; u < 13; ++u) {
      const int c = (c0 + u < cend) ? c0 + u : cend - 1;
      a4[u] = *(const float4*)(arow + 16 * c);
      b4[u] = *(const float4*)(wrow + 16 * c);
    }
#pragma unroll
    for (int u = 0; u < 13; ++u) {
      if (c0 + u >= cend) continue;
      f32x4& acc = acc4[u & 1];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].x, b4[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].y, b4[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].z, b4[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].w, b4[u].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) sacc[wave][lane][rr] = acc4[0][rr] + acc4[1][rr];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = kq * 4 + rr, g = row0 + r;
      float a = ((sacc[0][lane][rr] + sacc[1][lane][rr]) + (sacc[2][lane][rr] + sacc[3][lane][rr])) + b1;
      a = a > 0.f ? a : 0.f;
      int keep = 1;
      if (TRAIN && g < B) {
        keep = inj_mask ? (int)inj_mask[g * 128 + n]
                        : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)n)) >= 0.5f);
        sp.a1[g * 128 + n] = a;
        sp.lmask[g * 128 + n] = (uint8_t)keep;
      }
      const float p = igmc_group16_sum_f((TRAIN ? (keep ? a * 2.f : 0.f) : a) * w2);      // F.dropout(p = 0.5): kept * 2
      if (li == 0 && g < B) sp.lin_part[(size_t)nt * m.graph_cap + g] = p;
    }
  }
  // the last workgroup of the row tile to get here has every partial in sight
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    s_last = (atomicAdd(sp.lin_ctr + mt, 1) == (int)gridDim.y - 1);
  }
  __syncthreads();
  if (s_last) {
    __threadfence();
    if (tid < 16 && row0 + tid < B) {
      const int g = row0 + tid;
      float s2 = 0.f;
      for (int t = 0; t < (int)gridDim.y; ++t) s2 += sp.lin_part[(size_t)t * m.graph_cap + g];
      const float o = s2 + l2b;
      out[g] = o;
      m.err[g] = o - b.y[g];
    }
    if (tid == 0) sp.lin_ctr[mt] = 0;
  }
}

// d flat = dz @ lin1.weight through conv2's ReLU, batched over the subgraphs on the f32 matrix cores: workgroup = (16
// subgraphs) x (four 16-column tiles of the 32 Q2 columns), K = the 128 hidden units; dz is formed on the fly from the
// forward's a1 / mask / residual (the first column tile also stores it, and d out, for the weight-gradient kernel)
__global__ __launch_bounds__(SP_THREADS) void k_sp_dflat(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                          float grad_scale) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + sizeof(SpDev) + 32>();
  const int B = b.totals[3], dense = sp.dense;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int nt = blockIdx.y * 4 + wave;
  if (row0 >= B || nt * 16 >= dense) return;
  const int ga = (row0 + li < B) ? row0 + li : B - 1;
  const int i0 = nt * 16;
  const float dout = 2.f * m.err[ga] * grad_scale;                  // d (mean squared error) / d out
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float4 dz4[8];
  float bw[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c) {                     // hidden units 16 c + 4 kq .. + 3
    const int j = 16 * c + 4 * kq;
    const float4 av = *(const float4*)(sp.a1 + (size_t)ga * 128 + j);
    const uint32_t mk = *(const uint32_t*)(sp.lmask + (size_t)ga * 128 + j);
    const float4 w2 = *(const float4*)(Pd + sp.t_l2w + j);
    dz4[c].x = (row0 + li < B && av.x > 0.f && (mk & 0xFFu)) ? dout * w2.x * 2.f : 0.f;
    dz4[c].y = (row0 + li < B && av.y > 0.f && (mk & 0xFF00u)) ? dout * w2.y * 2.f : 0.f;
    dz4[c].z = (row0 + li < B && av.z > 0.f && (mk & 0xFF0000u)) ? dout * w2.z * 2.f : 0.f;
    dz4[c].w = (row0 + li < B && av.w > 0.f && (mk & 0xFF000000u)) ? dout * w2.w * 2.f : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) bw[c][t] = Pd[sp.t_l1w + (size_t)(j + t) * dense + i0 + li];
  }
  if (nt == 0 && row0 + li < B) {
#pragma unroll
    for (int c = 0; c < 8; ++c) *(float4*)(sp.dz + (size_t)ga * 128 + 16 * c + 4 * kq) = dz4[c];
    if (kq == 0) sp.dout[ga] = dout;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f32x4& acc = (c & 1) ? acc1 : acc0;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz4[c].x, bw[c][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz4[c].y, bw[c][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz4[c].z, bw[c][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz4[c].w, bw[c][3], acc, 0, 0, 0);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int g = row0 + 4 * kq + rr;
    if (g >= B) continue;
    const size_t at = (size_t)g * dense + i0 + li;
    sp.dflat[at] = (sp.flat[at] > 0.f) ? acc0[rr] + acc1[rr] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- backward
// debug aid (IGMC_SP_TIMING=1; igmc_debug_sp_clocks): shader-clock stamps of workgroup 0 of the last k_sp_bwd launch
__device__ unsigned long long g_sp_clk[16];
#ifdef IGMC_HIPEMU
#define SP_STAMP(k) do { } while (0)
#else
#define SP_STAMP(k) do { if (timing && blockIdx.x == 0 && threadIdx.x == 0) g_sp_clk[k] = __builtin_readcyclecounter(); } while (0)
#endif
extern "C" int igmc_debug_sp_clocks(unsigned long long* out, int n) {
#ifndef IGMC_HIPEMU
  if (n > 16) n = 16;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_clk), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return 1;
#else
  for (int i = 0; i < n; ++i) out[i] = 0;
#endif
  return 0;
}
// dynamic LDS: sel[k] | rank[nmax] | w1[16*97] | w2[32*80] | y1[16*k] | z[16*Q1] | dy2[32*Q2] | dzp[16*Q1] | dy1[16*k] | dz[128] | xs[k*98]
__global__ __launch_bounds__(SP_WG) void k_sp_bwd(BatchDev b, ModelDev m, SpDev sp, const float* __restrict__ Pd,
                                                        float grad_scale, int timing) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + sizeof(SpDev) + 32>();
  IGMC_DYN_SMEM(smem);
  SP_STAMP(0);
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
  for (int i = tid; i < SP_C1 * SP_C; i += SP_WG) w1[i] = Pd[sp.t_c1w + i];
  for (int i = tid; i < SP_C2 * SP_C1 * SP_KW; i += SP_WG) w2[i] = Pd[sp.t_c2w + i];
  for (int i = tid; i < n; i += SP_WG) rank[i] = -1;
  for (int p = tid; p < k; p += SP_WG) sel[p] = sp.sel[(size_t)g * k + p];
  for (int i = tid; i < SP_C1 * k; i += SP_WG) y1[i] = sp.y1[(size_t)g * SP_C1 * k + i];
  // (dz, d out and d flat: k_sp_dflat, batched over the subgraphs)
  for (int i = tid; i < dense; i += SP_WG) dy2[i] = sp.dflat[(size_t)g * dense + i];
  __syncthreads();
  SP_STAMP(1);
  for (int p = tid; p < k; p += SP_WG)
    if (sel[p] >= 0) rank[sel[p] - n0] = p;
  for (int i = tid; i < SP_C1 * Q1; i += SP_WG) {
    const int ic = i / Q1, q = i - ic * Q1;
    const float a = y1[ic * k + 2 * q], c = y1[ic * k + 2 * q + 1];
    z[i] = a >= c ? a : c;
  }
  __syncthreads();
  SP_STAMP(2);
  // conv2: weight / bias gradient of this graph and the gradient w.r.t. the pooled sequence -- three small matrix products
  // (operands read straight from the LDS arrays, no im2col copy) on v_mfma_f32_16x16x4_f32, one 16 x 16 output tile per wave:
  //   d w2[oc][(ic, t)] = sum_q dy2[oc][q] z[ic][q + t]              32 x 80, K = Q2:        waves 0..9
  //   dzp[ic][qp]       = sum_(oc, t) w2[oc][ic][t] dy2[oc][qp - t]   16 x Q1, K = 32 * 5:    waves 10..
  // (the scalar loops they replace read two LDS words per product: 13 + 9 k cycles of this kernel)
  float* pc2 = sp.part_c2 + (size_t)g * (SP_C2 * SP_C1 * SP_KW + SP_C2);
  {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    if (wave < 10) {
      const int mt = wave / 5, nt = wave - 5 * mt;
      const int oc = 16 * mt + li, nn = 16 * nt + li, ic = nn / SP_KW, t = nn - ic * SP_KW;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int q0 = 0; q0 < Q2; q0 += 16) {           // (the operands of four k-steps are requested together)
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + 4 * u + kq;
          av[u] = (q < Q2) ? dy2[oc * Q2 + q] : 0.f;
          bv[u] = (q < Q2) ? z[ic * Q1 + q + t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) pc2[(16 * mt + 4 * kq + r) * (SP_C1 * SP_KW) + nn] = acc[r];
    } else {
      const int ntiles = (Q1 + 15) >> 4;
      for (int nt = wave - 10; nt < ntiles; nt += SP_WG / 64 - 10) {
        const int qp = 16 * nt + li;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < SP_C2 * SP_KW; k0 += 16) {
          float av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int kk = k0 + 4 * u + kq, oc = kk / SP_KW, t = kk - oc * SP_KW, q = qp - t;
            av[u] = w2[(oc * SP_C1 + li) * SP_KW + t];
            bv[u] = (q >= 0 && q < Q2 && qp < Q1) ? dy2[oc * Q2 + q] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        if (qp < Q1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dzp[(4 * kq + r) * Q1 + qp] = acc[r];
        }
      }
    }
  }
  for (int oc = tid; oc < SP_C2; oc += SP_WG) {
    float s = 0.f;
    for (int q = 0; q < Q2; ++q) s += dy2[oc * Q2 + q];
    pc2[SP_C2 * SP_C1 * SP_KW + oc] = s;
  }
  SP_STAMP(3);
  SP_STAMP(4);
  __syncthreads();
  // max-pool (the first maximum takes the gradient, like torch) and conv1's ReLU
  for (int i = tid; i < SP_C1 * k; i += SP_WG) {
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
  SP_STAMP(5);
  // conv1: weight / bias gradient of this graph.  The pooled rows come from LDS (staged once, coalesced): read from global memory inside the p loop they were 360 serial round trips per thread, 100 us of this kernel.
  float* xs = dzs + 128;                             // [k][SP_C + 1]
  for (int i = tid; i < k * 128; i += SP_WG) {
    const int p = i >> 7, c = i & 127;
    if (c < SP_C) xs[p * (SP_C + 1) + c] = (sel[p] >= 0) ? sp_cat(m, sel[p], c) : 0.f;
  }
  __syncthreads();
  SP_STAMP(6);
  float* pc1 = sp.part_c1 + (size_t)g * (SP_C1 * SP_C + SP_C1);
  {   // d w1[oc][j] = sum_p dy1[oc][p] x[p][j]: 16 x 97, K = k, on the f32 matrix cores, waves 0..6 one 16-column tile each
      // (rows of padding positions are zero in xs; the scalar loop -- 2 k LDS words per output -- was 43 k cycles of 97 k)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    if (wave < (SP_C + 15) / 16) {
      const int j = 16 * wave + li;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int p0 = 0; p0 < k; p0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + 4 * u + kq;
          av[u] = (p < k) ? dy1[li * k + p] : 0.f;
          bv[u] = (p < k && j < SP_C) ? xs[p * (SP_C + 1) + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      if (j < SP_C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pc1[(4 * kq + r) * SP_C + j] = acc[r];
      }
    }
  }
  for (int oc = tid; oc < SP_C1; oc += SP_WG) {
    float s = 0.f;
    for (int p = 0; p < k; ++p) s += dy1[oc * k + p];
    pc1[SP_C1 * SP_C + oc] = s;
  }
  SP_STAMP(7);
  // gradient w.r.t. the node states.  First per POOLED position on the f32 matrix cores, G[p][c] = sum_oc dy1[oc][p] w1[oc][c]
  // (k x 97, K = 16: 49 tiles over the 16 waves) into the rows' LDS array (the pooled rows are no longer needed), ...
  __syncthreads();                                   // (conv1's weight gradient has read xs)
  {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int mtiles = (k + 15) >> 4, ntiles = (SP_C + 15) >> 4;
    for (int tile = wave; tile < mtiles * ntiles; tile += SP_WG / 64) {
      const int mt = tile / ntiles, nt = tile - mt * ntiles;
      const int p = 16 * mt + li, c = 16 * nt + li;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int oc = 4 * u + kq;
        av[u] = (p < k) ? dy1[oc * k + p] : 0.f;
        bv[u] = (c < SP_C) ? w1[oc * SP_C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      if (c < SP_C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = 16 * mt + 4 * kq + r;
          if (pp < k) xs[pp * (SP_C + 1) + c] = acc[r];
        }
      }
    }
  }
  __syncthreads();
  // ... then every node of the graph is written exactly once (zeros when it was not pooled)
  for (int i = tid; i < n * 128; i += SP_WG) {
    const int node = i >> 7, c = i & 127;             // c < 96: channel of h_0..h_2; c >= 96: column c - 96 of dPre_3
    const int p = rank[node];
    const size_t row = (size_t)(n0 + node) * 32;
    if (c < 96) {
      sp.dcat[c >> 5][row + (c & 31)] = (p >= 0) ? xs[p * (SP_C + 1) + c] : 0.f;
    } else {
      float v = 0.f;
      if (p >= 0 && c == 96) {
        const float hv = m.h[3][row];
        v = xs[p * (SP_C + 1) + 96] * (1.f - hv * hv);
      }
      m.dpre[3][row + (c - 96)] = v;
    }
  }
  SP_STAMP(8);
  (void)dzs; (void)grad_scale;
}

// ---------------------------------------------------------------------------------------------- weight gradients over the batch
// blocks [0, nb1): lin1.weight tiles (one element per thread: sum over the graphs of dz[g][j] * flat[g][i]);
// then the conv1 / conv2 partial sums (one output per thread) and one block for lin1.bias + lin2
// ... and (nun > 0) nun more blocks that bring the conv parameters' gradient from the engine layout into the true one (what
// k_sp_unpack does as a launch of its own): disjoint elements of Gd, nothing in this launch reads them
__global__ __launch_bounds__(SP_THREADS) void k_sp_wgrad(BatchDev b, SpDev sp, int B, int nb1, float* __restrict__ Gd, ModelDev m,
                                                          const float* __restrict__ ge, int nun) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(SpDev) + sizeof(ModelDev) + 32>();
  const int tid = threadIdx.x;
  const int dense = sp.dense;
  if ((int)blockIdx.x >= (int)gridDim.x - nun) {
    sp_unpack_body(m, sp, ge, Gd, (int)blockIdx.x - ((int)gridDim.x - nun), nun);
    return;
  }
  if ((int)blockIdx.x < nb1) {
    // d lin1.weight[j][i] = sum_g dz[g][j] flat[g][i]: 8 x (dense / 16) output tiles, one per wave, K = the subgraphs
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x * 4 + wave, ntn = dense / 16;
    if (tile >= 8 * ntn) return;
    const int j0 = (tile / ntn) * 16, i0 = (tile % ntn) * 16;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int g0 = 0; g0 < B; g0 += 32) {              // 8 MFMA steps (32 subgraphs) of loads in flight
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + 4 * u + kq, gc = g < B ? g : B - 1;
        av[u] = sp.dz[(size_t)gc * 128 + j0 + li];
        bv[u] = sp.flat[(size_t)gc * dense + i0 + li];
        if (g >= B) av[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4& acc = (u & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      Gd[sp.t_l1w + (size_t)(j0 + 4 * kq + rr) * dense + i0 + li] = acc0[rr] + acc1[rr];
    return;
  }
  const int role = blockIdx.x - nb1;
  const int n1 = SP_C1 * SP_C + SP_C1, n2 = SP_C2 * SP_C1 * SP_KW + SP_C2;
  const int nbc1 = (n1 + SP_THREADS - 1) / SP_THREADS, nbc2 = (n2 + SP_THREADS - 1) / SP_THREADS;
  if (role < nbc1 + nbc2) {
    // conv1 / conv2: one output per thread, the subgraphs' partials summed in index order (16 loads in flight per round;
    // ONE workgroup looping over all outputs and subgraphs was 550 serial loads per thread, 120 us)
    const bool c1 = role < nbc1;
    const int i = (c1 ? role : role - nbc1) * SP_THREADS + tid, nn = c1 ? n1 : n2;
    if (i >= nn) return;
    const float* part = (c1 ? sp.part_c1 : sp.part_c2) + i;
    float s = 0.f;
    for (int g0 = 0; g0 < B; g0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(g0 + u < B ? g0 + u : B - 1) * nn];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (g0 + u < B) s += v[u];
    }
    Gd[(c1 ? sp.t_c1w : sp.t_c2w) + i] = s;
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
                  SP_C2 * sp.Q2 + 128 + 8 + sp.k * (SP_C + 1)) * 4;
}
static size_t sp_bwd_lds(const SpDev& sp) {
  return (size_t)(sp.k + sp.nmax + SP_C1 * SP_C + SP_C2 * SP_C1 * SP_KW + SP_C1 * sp.k + SP_C1 * sp.Q1 + SP_C2 * sp.Q2 +
                  SP_C1 * sp.Q1 + SP_C1 * sp.k + 128 + sp.k * (SP_C + 1)) * 4;
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
  if (training) {
    IGMC_PLAUNCH("k_sp_fwd", (k_sp_fwd<true>), B, SP_WG, sm, stream, b, m, sp, Pd, inj_mask, seed, step, out);
    IGMC_PLAUNCH("k_sp_lin_fwd", (k_sp_lin_fwd<true>), dim3((B + 15) / 16, 8), SP_THREADS, 0, stream, b, m, sp, Pd, inj_mask, seed, step, out);
  } else {
    IGMC_PLAUNCH("k_sp_fwd", (k_sp_fwd<false>), B, SP_WG, sm, stream, b, m, sp, Pd, inj_mask, seed, step, out);
    IGMC_PLAUNCH("k_sp_lin_fwd", (k_sp_lin_fwd<false>), dim3((B + 15) / 16, 8), SP_THREADS, 0, stream, b, m, sp, Pd, inj_mask, seed, step, out);
  }
}

void igmc_launch_sp_backward(const ModelDev& m, const SpDev& sp, const BatchDev& b, const float* Pd, int B, float grad_scale,
                             void* stream) {
  IGMC_PLAUNCH("k_sp_dflat", k_sp_dflat, dim3((B + 15) / 16, (sp.dense / 16 + 3) / 4), SP_THREADS, 0, stream, b, m, sp, Pd,
               grad_scale);
  IGMC_PLAUNCH("k_sp_bwd", k_sp_bwd, B, SP_WG, sp_bwd_lds(sp), stream, b, m, sp, Pd, grad_scale, getenv("IGMC_SP_TIMING") ? 1 : 0);
}

void igmc_launch_sp_wgrad(const ModelDev& m, const SpDev& sp, const BatchDev& b, int B, const float* ge, float* Gd,
                          void* stream) {
  const int nb1 = (8 * (sp.dense / 16) + 3) / 4;        // d lin1.weight: one 16 x 16 output tile per wave
  const int nbc = (SP_C1 * SP_C + SP_C1 + SP_THREADS - 1) / SP_THREADS + (SP_C2 * SP_C1 * SP_KW + SP_C2 + SP_THREADS - 1) / SP_THREADS;
  const int nun = 32;
  IGMC_PLAUNCH("k_sp_wgrad", k_sp_wgrad, nb1 + nbc + 1 + nun, SP_THREADS, 0, stream, b, sp, B, nb1, Gd, m, ge, nun);
}
