// model.hip -- IGMC forward / backward / optimiser kernels for gfx950 (CDNA4).
//
// Math (reference models.py:190-217, PyG-1.4.2 RGCNConv):
//   conv_l(x)[i] = sum_{(j->i, r)} x[j] W_l[r] + x[i] root_l + bias_l,  W_l[r] = sum_b att_l[r,b] basis_l[b]
//   h_l = tanh(conv_l(h_{l-1}));  readout = [h_0..h_3](target user) | [h_0..h_3](target item)
//   out = lin2(dropout(relu(lin1(readout)))) * multiply_by
//
// MI355X formulation (NOT the reference's per-edge [E,32,32] weight materialisation):
//   * aggregate in BASIS space:  A[i][b] = sum_e att[rel_e, b] * x[src_e]   (4 accumulators per
//     feature, independent of the number of relations R), then ONE dense transform
//     [A | x] (N x 160) @ [basis ; root] (160 x 32) on f32 MFMA (16x16x4), + bias, tanh fused.
//   * the gather walks a dst-sorted CSR (atomic-free, bit-reproducible): one wave per row.
//   * layer 0 has one-hot inputs: it degenerates to a table lookup W0[rel*L + label] per edge.
//   * backward reuses the same gather on the symmetric CSR (keep-bit 1 = transposed edge):
//       G[j][b]   = sum_{e: src=j} att[rel_e,b] dPre[dst_e]
//       dX        = [G | dPre] @ [basis^T ; root^T]
//       d basis_b = X^T G_b,  d root = X^T dPre,  d bias = sum dPre      (MFMA, per-block partials)
//       d att[r,b]= sum_j < (x_j basis_b), sum_{e: src=j, rel=r} dPre[dst_e] >   (rows are sorted
//                   by relation, so this is one dot product per relation RUN, not per edge)
//   * every serialized kernel costs >= 4.7 us on this part even when trivial (rocprofv3), so the step is
//     organised for FEW launches: derived weight layouts are produced while staging into LDS (no prep
//     kernel), the three Y products and the three weight-gradient products are batched into one launch
//     each (blockIdx.y = layer), and loss / epoch-total / control-block tick ride in the Adam kernel.
#include "model.h"
#include "g2_image.h"
#include <stdlib.h>

// ---- XCD affinity -------------------------------------------------------------------------------------
// MI355X has 8 XCDs with private, mutually non-coherent 4 MiB L2s, and workgroup b of a launch runs on XCD
// b % 8 (observed placement; used for SPEED only, never for correctness).  Subgraphs are independent: a row of
// graph g only ever reads rows of graph g.  So the graphs of a batch are split into 8 contiguous segments and
// every row-walking kernel lets XCD x work on segment x: what layer l wrote stays in the L2 that layer l+1
// reads it from, instead of being re-fetched from the other XCDs / Infinity Cache by all eight L2s
// (rocprofv3: TCC hit rate 50 % and FETCH_SIZE = 8 x the feature matrix without this).
struct XcdSeg {
  int lo, hi;      // row range [lo, hi) of this workgroup's XCD segment
  int j, nj;       // index of the workgroup within its XCD, workgroups per XCD
};
__device__ __forceinline__ XcdSeg igmc_xcd_segment(const BatchDev& b) {
  XcdSeg s;
  const int B = b.totals[3];
  const int x = blockIdx.x & 7;
  s.j = blockIdx.x >> 3;
  s.nj = (gridDim.x + 7 - x) >> 3;          // workgroups with this residue
  s.lo = b.node_off[(x * B) >> 3];
  s.hi = b.node_off[((x + 1) * B) >> 3];
  return s;
}

// =================================================================== row walkers
// One WAVE per destination row.  The row's CSR entries are consumed in chunks of 16: 16-lane group
// `grp` of the wave owns chunks grp, grp+4, ...; lane t of the group loads entry c0+t (one coalesced
// 64-byte index load per chunk) and the group then broadcasts entry k with a width-16 shuffle, so
// 4 x 4 feature-row loads (128 B each, float2 per lane) are in flight per wave instruction.  The four
// partial sums are combined with two xor-shuffles at the end of the row.

// =================================================================== layer 0 forward (one-hot input)
// h0[i] = tanh( sum_e W0[rel_e*L + label(src_e)] + root0[label_i] + bias0 ),  W0[r][c] = sum_b att0[r,b] basis0[b][c]
// STORE (training): also emits cnt0[i][code] = number of kept incoming edges with that code, so that the
// layer-0 weight gradient is a dense [codes x N] @ [N x 32] product that never touches the edges again.
template <bool FLAGS, bool STORE>
__global__ __launch_bounds__(IGMC_BLOCK) void k_l0_fwd(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                         float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const int RL = m.R * m.L, LF = m.L * 32;
  float* sW0 = (float*)smem;                 // [R*L][32]
  float* sroot = sW0 + RL * 32;              // [L][32]
  float* sbias = sroot + LF;                 // [32]
  int* shist = (int*)(sbias + 32);           // [4 waves][R*L]   (STORE only)
  for (int i = threadIdx.x; i < RL * 32; i += IGMC_BLOCK) {
    const int r = i / LF, cf = i % LF;
    float s = 0.f;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) s += P[m.off_att[0] + r * 4 + bb] * P[m.off_basis[0] + bb * LF + cf];
    sW0[i] = s;
  }
  for (int i = threadIdx.x; i < LF; i += IGMC_BLOCK) sroot[i] = P[m.off_root[0] + i];
  if (threadIdx.x < 32) sbias[threadIdx.x] = P[m.off_bias[0] + threadIdx.x];
  __syncthreads();
  const int N = b.totals[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, t = lane & 15;
  int* hist = shist + wave * RL;
  const XcdSeg seg = igmc_xcd_segment(b);
  (void)N;
  for (int i = seg.lo + seg.j * 4 + wave; i < seg.hi; i += seg.nj * 4) {
    float ax = 0.f, ay = 0.f;
    if (STORE) {
      for (int c = lane; c < RL; c += 64) hist[c] = 0;
      IGMC_WAVE_SYNC();
    }
    const int beg = b.row_ptr[i], end = b.row_ptr[i + 1];
    for (int c0 = beg + grp * 16; c0 < end; c0 += 64) {
      const int e = c0 + t;
      int w = -1;                              // code, or -1 = skip
      if (e < end && (!FLAGS || (b.eflag[e] & 1))) w = b.ecode[e];
      if (STORE && w >= 0) atomicAdd(&hist[w], 1);
      const int n = (end - c0 < 16) ? end - c0 : 16;
      for (int k = 0; k < n; ++k) {
        const int code = __shfl(w, k, 16);
        if (code >= 0) {
          ax += sW0[code * 32 + 2 * t];
          ay += sW0[code * 32 + 2 * t + 1];
        }
      }
    }
    ax += __shfl_xor(ax, 16, 64);
    ay += __shfl_xor(ay, 16, 64);
    ax += __shfl_xor(ax, 32, 64);
    ay += __shfl_xor(ay, 32, 64);
    if (grp == 0) {
      const int lab = b.node_label[i];
      float2 o;
      o.x = tanhf(ax + sbias[2 * t] + sroot[lab * 32 + 2 * t]);
      o.y = tanhf(ay + sbias[2 * t + 1] + sroot[lab * 32 + 2 * t + 1]);
      *(float2*)(out + (size_t)i * 32 + 2 * t) = o;
    }
    if (STORE) {
      IGMC_WAVE_SYNC();
      for (int c = lane; c < RL; c += 64) m.cnt0[(size_t)i * RL + c] = (uint16_t)hist[c];
      IGMC_WAVE_SYNC();
    }
  }
}

// Basis-space aggregate of ONE row (see "row walkers" above) by NG 16-lane groups: NG = 4 -> the whole wave
// works on row i and lanes of group 0 hold the result; NG = 1 -> every group walks its OWN row i (i differs
// between the groups of a wave) and holds its own result.  ax[bb], ay[bb] = features 2t, 2t+1 of A[i][bb].
template <bool FLAGS, bool TRANS, bool ATTG, int NG = 4>
__device__ __forceinline__ void gather_row(const BatchDev& b, const float* __restrict__ in, const float* s_att,
                                           float* my_gatt, const float* __restrict__ Y, int i, int beg, int end,
                                           int lane, float (&ax)[4], float (&ay)[4]) {
  const int grp = lane >> 4, t = lane & 15;
  const int kbit = TRANS ? 1 : 0;
  (void)grp;
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) { ax[bb] = 0.f; ay[bb] = 0.f; }
  {
    float yx[4], yy[4];
    float tx = 0.f, ty = 0.f;
    int cur = -1;
    if (ATTG) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 v = *(const float2*)(Y + (size_t)i * 128 + q * 32 + 2 * t);
        yx[q] = v.x;
        yy[q] = v.y;
      }
    }
    int c0 = beg + (NG == 4 ? grp * 16 : 0);
    uint32_t w_next = 0u;
    int kp_next = 0;
    {
      const int e = c0 + t;
      if (e < end) {
        w_next = b.ecr[e];
        kp_next = FLAGS ? (b.eflag[e] >> kbit) & 1 : 1;
      }
    }
    for (; c0 < end; c0 += 16 * NG) {
      const uint32_t w = w_next;
      const int kp = kp_next;
      {   // prefetch the NEXT chunk's entries: the load overlaps with this chunk's feature-row loads
        const int e1 = c0 + 16 * NG + t;
        w_next = 0u;
        kp_next = 0;
        if (e1 < end) {
          w_next = b.ecr[e1];
          kp_next = FLAGS ? (b.eflag[e1] >> kbit) & 1 : 1;
        }
      }
      // broadcast the 16 entries of the chunk and issue ALL their feature-row loads before any use:
      // one memory round trip per chunk instead of four
      uint32_t wk[16];
      float2 x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        wk[q] = __shfl(w, q, 16);
        const int kk = __shfl(kp, q, 16);          // 0 for dropped edges and for lanes past the row end
        wk[q] = kk ? wk[q] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (wk[q] != 0xFFFFFFFFu) x[q] = *(const float2*)(in + (size_t)(wk[q] & 0xFFFFFFu) * 32 + 2 * t);
        else { x[q].x = 0.f; x[q].y = 0.f; }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (wk[q] == 0xFFFFFFFFu) continue;
        const int r = (int)(wk[q] >> 24);
        if (!ATTG) {
          const float* a = s_att + r * 4;
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            ax[bb] += a[bb] * x[q].x;
            ay[bb] += a[bb] * x[q].y;
          }
        } else {
          if (r != cur) {
            if (cur >= 0) {   // flush the finished relation run
              const float* a = s_att + cur * 4;
#pragma unroll
              for (int bb = 0; bb < 4; ++bb) {
                ax[bb] += a[bb] * tx;
                ay[bb] += a[bb] * ty;
                const float p = igmc_group16_sum_f(yx[bb] * tx + yy[bb] * ty);
                if (t == 0) my_gatt[cur * 4 + bb] += p;
              }
            }
            cur = r;
            tx = 0.f;
            ty = 0.f;
          }
          tx += x[q].x;
          ty += x[q].y;
        }
      }
    }
    if (ATTG && cur >= 0) {
      const float* a = s_att + cur * 4;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        ax[bb] += a[bb] * tx;
        ay[bb] += a[bb] * ty;
        const float p = igmc_group16_sum_f(yx[bb] * tx + yy[bb] * ty);
        if (t == 0) my_gatt[cur * 4 + bb] += p;
      }
    }
    if (NG == 4) {
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        ax[bb] += __shfl_xor(ax[bb], 16, 64);
        ay[bb] += __shfl_xor(ay[bb], 16, 64);
        ax[bb] += __shfl_xor(ax[bb], 32, 64);
        ay[bb] += __shfl_xor(ay[bb], 32, 64);
      }
    }
  }
}

// =================================================================== dense row transform on f32 MFMA
// OUT[N,NO] = epilogue([A1[N,K1] | A2[N,K2]] @ W[K1+K2, NO]).  One wave = 16 rows x NO columns,
// v_mfma_f32_16x16x4_f32; W is staged once per block in LDS (pitch NO+4: conflict-free b32 reads).
// WMODE selects how W is derived from the layer's parameters WHILE staging (no prep kernel):
//   W_PLAIN : [basis ; root] as stored (contiguous in the flat buffer)           -> forward
//   W_BWD_T : [basis_b^T ; root^T]   sW[b*32+fo][f] = basis[b][f][fo]            -> dLoss/dx
//   W_YCAT  : [basis_0|..|basis_3]   sW[f][b*32+fo] = basis[b][f][fo]            -> Y (att gradient)
enum { EPI_NONE = 0, EPI_BIAS_TANH = 1, EPI_BWD = 2 };
enum { W_PLAIN = 0, W_BWD_T = 1, W_YCAT = 2 };

template <int K1, int K2, int NO, int EPI, int WMODE, int NTH = IGMC_BLOCK>
__device__ __forceinline__ void dense_body(const BatchDev& b, const float* __restrict__ A1,
                                           const float* __restrict__ A2, const float* __restrict__ basis,
                                           const float* __restrict__ root, const float* __restrict__ bias,
                                           float* __restrict__ out, const float* __restrict__ xin,
                                           const float* __restrict__ gfeat, int D, int rlayer,
                                           float* __restrict__ zero_out, float* sW) {
  constexpr int K = K1 + K2, PITCH = NO + 4, NT = NO / 16;
  if (WMODE == W_PLAIN) {
    for (int idx = threadIdx.x; idx < K * NO; idx += NTH) sW[(idx / NO) * PITCH + (idx % NO)] = basis[idx];
  } else {
    for (int idx = threadIdx.x; idx < 4096; idx += NTH) {     // coalesced read of basis[b][f][fo]
      const int bb = idx >> 10, f = (idx >> 5) & 31, fo = idx & 31;
      if (WMODE == W_BWD_T) sW[(bb * 32 + fo) * PITCH + f] = basis[idx];
      else sW[f * PITCH + bb * 32 + fo] = basis[idx];
    }
    if (WMODE == W_BWD_T)
      for (int idx = threadIdx.x; idx < 1024; idx += NTH) sW[(128 + (idx & 31)) * PITCH + (idx >> 5)] = root[idx];
  }
  __syncthreads();
  const int N = b.totals[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  constexpr int ROWS = NTH / 4;      // 16 rows per wave
  for (int tile = blockIdx.x; tile * ROWS < N; tile += gridDim.x) {
    const int row0 = tile * ROWS + wave * 16;
    if (row0 >= N) continue;
    const int row = (row0 + li < N) ? row0 + li : N - 1;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const float* src = (s * 16 < K1) ? (A1 + (size_t)row * K1 + s * 16) : (A2 + (size_t)row * K2 + (s * 16 - K1));
      const float4 a4 = *(const float4*)(src + 4 * kq);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) {
        const int k = s * 16 + 4 * kq + mm;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv = sW[k * PITCH + nt * 16 + li];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mm], bv, acc[nt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int orow = row0 + kq * 4 + rr;
        if (orow >= N) continue;
        const int n = nt * 16 + li;
        float v = acc[nt][rr];
        if (EPI == EPI_BIAS_TANH) {
          v = tanhf(v + bias[n]);
          if (zero_out) zero_out[(size_t)orow * 32 + n] = 0.f;
        }
        if (EPI == EPI_BWD) {
          const int lab = b.node_label[orow];
          if (lab < 2) v += gfeat[(size_t)b.node_graph[orow] * D + lab * 128 + rlayer * 32 + n];
          const float xv = xin[(size_t)orow * 32 + n];
          v *= (1.f - xv * xv);
        }
        out[(size_t)orow * NO + n] = v;
      }
    }
  }
}

// Y_l = h_{l-1} @ [basis_0|..|basis_3] for l = 1..3 in ONE launch (blockIdx.y = l-1)
__global__ __launch_bounds__(IGMC_BLOCK) void k_dense_y_all(BatchDev b, ModelDev m, const float* __restrict__ P) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const int li = blockIdx.y;
  dense_body<0, 32, 128, EPI_NONE, W_YCAT>(b, nullptr, m.h[li], P + m.off_basis[li + 1], nullptr, nullptr, m.Y[li],
                                           nullptr, nullptr, 0, 0, nullptr, (float*)smem);
}

// =================================================================== fused R-GCN layer (gather + dense)
// One 256-thread workgroup per tile of 16 rows: wave w walks rows 4w..4w+3 of the tile in basis space straight into an LDS
// tile [16][A(128) | self(32)], ONE 16-lane group per row (all 2400 waves of an ml_1m batch are resident at once), then
// the 4 waves multiply the tile with the layer's [basis ; root] operand on f32 MFMA (2 column tiles x 2 K-slices of 80,
// 20 MFMAs each) and the epilogue (bias + tanh, or the tanh' / readout-gradient backward epilogue) writes the 128-byte
// output rows.  Forward never materialises the 512-byte basis-space rows in HBM; backward writes them once (the weight
// gradient needs them).  (Rounds 1-2 kept three more formulations selectable -- separate gather + dense kernels,
// 16-wave tiles, tiles of row segments: all measured slower, removed in round 3.)
#define IGMC_TP 164
template <bool FLAGS, bool BWD>
__global__ __launch_bounds__(IGMC_BLOCK) void k_rgcn_layer4(BatchDev b, ModelDev m, const float* __restrict__ P, int l,
                                                             float* __restrict__ zero_out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const int R = m.R;
  float* tile = (float*)smem;               // [16][IGMC_TP]
  float* red = tile + 16 * IGMC_TP;         // [2 k-slices][2 col tiles][64 lanes * 4]
  float* s_att = red + 1024;                // [R][4]
  float* s_gatt = s_att + R * 4;            // BWD: [16 groups][R*4]
  const float* __restrict__ in = BWD ? m.dpre[l] : m.h[l - 1];
  const float* __restrict__ Yl = BWD ? m.Y[l - 1] : nullptr;
  const float* att = P + m.off_att[l];
  const float* basis = P + m.off_basis[l];
  const float* root = P + m.off_root[l];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, t = lane & 15;
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < R * 4; i += IGMC_BLOCK) s_att[i] = att[i];
  if (BWD)
    for (int i = tid; i < 16 * R * 4; i += IGMC_BLOCK) s_gatt[i] = 0.f;
  const int nt = wave & 1, ks = wave >> 1;
  float bw[20];
#pragma unroll
  for (int j = 0; j < 20; ++j) {
    const int k = ks * 80 + 4 * j + kq, n = nt * 16 + li;
    if (!BWD) bw[j] = basis[k * 32 + n];
    else bw[j] = (k < 128) ? basis[((k >> 5) * 32 + n) * 32 + (k & 31)] : root[n * 32 + (k - 128)];
  }
  __syncthreads();
  const int N = b.totals[0];
  const int trow = wave * 4 + grp;            // row of the tile owned by this 16-lane group
  float* my_gatt = s_gatt + trow * R * 4;
  const XcdSeg seg = igmc_xcd_segment(b);
  (void)N;
  for (int tl = seg.j; seg.lo + tl * 16 < seg.hi; tl += seg.nj) {
    const int trow0 = seg.lo + tl * 16;       // first row of the tile (tiles are cut per XCD segment)
    const int i = trow0 + trow;
    // everything that does not depend on the gather is requested FIRST, so that it is in flight during the
    // index / feature-row round trips: row bounds, the self row, and the epilogue operands of this thread
    int beg = 0, end = 0;
    float2 xs;
    xs.x = 0.f;
    xs.y = 0.f;
    if (i < seg.hi) {
      beg = b.row_ptr[i];
      end = b.row_ptr[i + 1];
      xs = *(const float2*)(in + (size_t)i * 32 + 2 * t);
    }
    float epi_x[2] = {0.f, 0.f}, epi_b[2] = {0.f, 0.f};
    int epi_lab[2] = {9, 9}, epi_g[2] = {0, 0};
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int o = tid + h2 * IGMC_BLOCK;
      const int ol = (o & 255) >> 2, rr = o & 3;
      const int orow = trow0 + (ol >> 4) * 4 + rr, n = (o >> 8) * 16 + (ol & 15);
      if (!BWD) {
        epi_b[h2] = P[m.off_bias[l] + n];
      } else if (orow < seg.hi) {
        epi_lab[h2] = b.node_label[orow];
        epi_g[h2] = b.node_graph[orow];
        epi_x[h2] = m.h[l - 1][(size_t)orow * 32 + n];
      }
    }
    if (i < seg.hi) {
      float ax[4], ay[4];
#ifdef IGMC_EXP_CAPROW
      if (end > beg + IGMC_EXP_CAPROW) end = beg + IGMC_EXP_CAPROW;      // timing experiment only (wrong results)
#endif
      gather_row<FLAGS, BWD, BWD, 1>(b, in, s_att, my_gatt, Yl, i, beg, end, lane, ax, ay);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        tile[trow * IGMC_TP + bb * 32 + 2 * t] = ax[bb];
        tile[trow * IGMC_TP + bb * 32 + 2 * t + 1] = ay[bb];
        if (BWD) {
          float2 o;
          o.x = ax[bb];
          o.y = ay[bb];
          *(float2*)(m.gagg[l - 1] + (size_t)i * 128 + bb * 32 + 2 * t) = o;
        }
      }
      tile[trow * IGMC_TP + 128 + 2 * t] = xs.x;
      tile[trow * IGMC_TP + 128 + 2 * t + 1] = xs.y;
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        tile[trow * IGMC_TP + q * 32 + 2 * t] = 0.f;
        tile[trow * IGMC_TP + q * 32 + 2 * t + 1] = 0.f;
      }
    }
    __syncthreads();
    {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 20; ++j) {
        const float a = tile[li * IGMC_TP + ks * 80 + 4 * j + kq];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[j], acc, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) red[(ks * 2 + nt) * 256 + lane * 4 + rr] = acc[rr];
    }
    __syncthreads();
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int o = tid + h2 * IGMC_BLOCK;
      const int ont = o >> 8, idx = o & 255, ol = idx >> 2, rr = idx & 3;
      const float v0 = red[(0 * 2 + ont) * 256 + idx] + red[(1 * 2 + ont) * 256 + idx];
      const int orow = trow0 + (ol >> 4) * 4 + rr, n = ont * 16 + (ol & 15);
      if (orow < seg.hi) {
        float v = v0;
        if (!BWD) {
          v = tanhf(v + epi_b[h2]);
          m.h[l][(size_t)orow * 32 + n] = v;
          if (zero_out) zero_out[(size_t)orow * 32 + n] = 0.f;
        } else {
          const int lab = epi_lab[h2];
          if (m.dcat[l - 1]) v += m.dcat[l - 1][(size_t)orow * 32 + n];
          else if (lab < 2) v += m.gfeat[(size_t)epi_g[h2] * m.D + lab * 128 + (l - 1) * 32 + n];
          const float xv = epi_x[h2];
          m.dpre[l - 1][(size_t)orow * 32 + n] = v * (1.f - xv * xv);
        }
      }
    }
    __syncthreads();
  }
  if (BWD) {
    float* gp = m.gatt_part + ((size_t)(l - 1) * IGMC_GATHER_BLOCKS + blockIdx.x) * R * 4;
    for (int i = tid; i < R * 4; i += IGMC_BLOCK) {
      float sacc = 0.f;
      for (int g = 0; g < 16; ++g) sacc += s_gatt[g * R * 4 + i];
      gp[i] = sacc;
    }
  }
}

// =================================================================== weight gradients  G = X^T [D1 | D2]
// X = h_{l-1} [N,32], D1 = G_l [N,128], D2 = dPre_l [N,32]  ->  per-block partial [32][160] (+ column sums of
// D2 = d bias).  Layer slices ly = ly_base + blockIdx.y (0..2 = conv layers 1..3, 3 = layer 0) in ONE launch.
// (Running the slices as parallel graph branches on auxiliary streams was measured SLOWER: every cross-stream
// edge of the hipGraph costs several microseconds on this part -- more than the kernels it would hide.)
__device__ __forceinline__ void wgrad_body(const BatchDev& b, const ModelDev& m, int ly) {
  __shared__ float sacc[32 * IGMC_KCAT + 32];
  // ly == 3: layer 0, whose "X" is synthesised from the per-node code histogram cnt0 plus the one-hot
  // columns for d root0[label] / d bias0 (codes < 32 only; larger tables use k_l0_bwd)
  const bool is_l0 = ly == 3;
  const int RL = m.R * m.L;
  const float* __restrict__ X = m.h[is_l0 ? 0 : ly];
  const float* __restrict__ D1 = m.gagg[is_l0 ? 0 : ly];
  const float* __restrict__ D2 = m.dpre[is_l0 ? 0 : ly + 1];
  float* part = m.wg_part + ((size_t)ly * IGMC_WG_BLOCKS + blockIdx.x) * (32 * IGMC_KCAT + 32);
  const int N = b.totals[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  f32x4 acc[2][10];
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc[m2][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const int ntile = (N + 15) >> 4;
  // A tile = 16 rows = four row-groups (row 4 rg + kq feeds k-slot kq of MFMA step rg).  All operands of a tile are
  // requested at once and the next tile's before this tile's 80 MFMAs, as 16-byte loads: lane li takes the B columns
  // 8 li .. 8 li + 7 of G (tile nt <-> column 8 li + nt) and 2 li, 2 li + 1 of dPre -- the column order inside the
  // product is free, the copy-out below undoes it (4 loads per row-group instead of 11 four-byte ones).
  struct WgTile {
    float2 a2[4];
    float4 ga[4], gb[4];
    float2 d2[4];
  };
  auto load_tile = [&](int tile, WgTile& t) {
    const int row0 = tile * 16;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = row0 + 4 * rg + kq;
      const bool ok = row < N;
      const int rs = ok ? row : 0;
      t.ga[rg] = make_float4(0.f, 0.f, 0.f, 0.f);
      t.gb[rg] = t.ga[rg];
      if (!is_l0) {
        t.a2[rg] = *(const float2*)(X + (size_t)rs * 32 + 2 * li);
        if (ok) {
          t.ga[rg] = *(const float4*)(D1 + (size_t)rs * 128 + 8 * li);
          t.gb[rg] = *(const float4*)(D1 + (size_t)rs * 128 + 8 * li + 4);
        }
      } else {
        const int lab = b.node_label[rs];
        const int c0 = 2 * li, c1 = 2 * li + 1;
        t.a2[rg].x = (c0 < RL) ? (float)m.cnt0[(size_t)rs * RL + c0] : ((c0 == RL + lab || c0 == RL + m.L) ? 1.f : 0.f);
        t.a2[rg].y = (c1 < RL) ? (float)m.cnt0[(size_t)rs * RL + c1] : ((c1 == RL + lab || c1 == RL + m.L) ? 1.f : 0.f);
      }
      if (!ok) { t.a2[rg].x = 0.f; t.a2[rg].y = 0.f; }
      t.d2[rg] = ok ? *(const float2*)(D2 + (size_t)rs * 32 + 2 * li) : make_float2(0.f, 0.f);
    }
  };
  const int tstep = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  WgTile cur;
  if (tile < ntile) load_tile(tile, cur);
  for (; tile < ntile; tile += tstep) {
    WgTile nxt;
    const bool more = tile + tstep < ntile;
    if (more) load_tile(tile + tstep, nxt);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float bv[10] = {cur.ga[rg].x, cur.ga[rg].y, cur.ga[rg].z, cur.ga[rg].w, cur.gb[rg].x,
                            cur.gb[rg].y, cur.gb[rg].z, cur.gb[rg].w, cur.d2[rg].x, cur.d2[rg].y};
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) {
        if (is_l0 && nt < 8) continue;
        if (nt >= 8) bsum[nt - 8] += bv[nt];
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a2[rg].x, bv[nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a2[rg].y, bv[nt], acc[1][nt], 0, 0, 0);
      }
    }
    if (more) cur = nxt;
  }
  // d bias: reduce the 4 row-slots (kq) of each column
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    bsum[q] += __shfl_xor(bsum[q], 16, 64);
    bsum[q] += __shfl_xor(bsum[q], 32, 64);
  }
  // deterministic cross-wave reduction through LDS, wave 0 first (LDS keeps the product's column order: tile-major)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int nt = 0; nt < 10; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int f = 2 * (kq * 4 + rr) + m2, n = nt * 16 + li;
            const float v = acc[m2][nt][rr];
            if (w == 0) sacc[f * IGMC_KCAT + n] = v;
            else sacc[f * IGMC_KCAT + n] += v;
          }
      if (kq == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (w == 0) sacc[32 * IGMC_KCAT + q * 16 + li] = bsum[q];
          else sacc[32 * IGMC_KCAT + q * 16 + li] += bsum[q];
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 32 * IGMC_KCAT + 32; i += IGMC_BLOCK) {
    int src;
    if (i < 32 * IGMC_KCAT) {
      const int f = i / IGMC_KCAT, n = i - f * IGMC_KCAT;
      const int np = (n < 128) ? (n & 7) * 16 + (n >> 3) : (8 + ((n - 128) & 1)) * 16 + ((n - 128) >> 1);
      src = f * IGMC_KCAT + np;
    } else {
      const int c = i - 32 * IGMC_KCAT;
      src = 32 * IGMC_KCAT + (c & 1) * 16 + (c >> 1);
    }
    part[i] = sacc[src];
  }
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_wgrad(BatchDev b, ModelDev m, int ly_base) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  wgrad_body(b, m, ly_base + blockIdx.y);
}

// =================================================================== layer 0 backward (dense, no edges)
// T[c][f] = sum_i weight(i,c) * dPre0[i][f],  c < R*L: cnt0[i][c] (kept in-edges with that code);
// c = R*L + label_i: d root0;  c = R*L + L: d bias0.   Thread (cg = tid>>5, f = tid&31) owns codes cg + 8k.
// Each block reduces a contiguous node range into its own partial (fixed order -> reproducible).
template <int KMAX>
__global__ __launch_bounds__(IGMC_BLOCK) void k_l0_bwd(BatchDev b, ModelDev m, const float* __restrict__ dpre,
                                                         float* __restrict__ part) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  const int N = b.totals[0];
  const int RL = m.R * m.L, rows = RL + m.L + 1;
  const int f = threadIdx.x & 31, cg = threadIdx.x >> 5;
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  const int chunk = (N + (int)gridDim.x - 1) / (int)gridDim.x;
  const int lo = blockIdx.x * chunk, hi = (lo + chunk < N) ? lo + chunk : N;
  for (int i0 = lo; i0 < hi; i0 += 4) {          // 4 nodes in flight: the loads of a group are independent
    float d[4];
    int lab[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (i0 + u < hi) ? i0 + u : hi - 1;
      d[u] = (i0 + u < hi) ? dpre[(size_t)i * 32 + f] : 0.f;
      lab[u] = b.node_label[i];
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int c = cg + 8 * k;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (i0 + u < hi) ? i0 + u : hi - 1;
        float w = 0.f;
        if (c < RL) w = (float)m.cnt0[(size_t)i * RL + c];
        else if (c == RL + lab[u] || c == RL + m.L) w = 1.f;
        acc[k] += w * d[u];
      }
    }
  }
  float* dst = part + (size_t)blockIdx.x * rows * 32;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int c = cg + 8 * k;
    if (c < rows) dst[c * 32 + f] = acc[k];
  }
}

// =================================================================== head: readout + MLP (+loss residual)
// reference models.py:203-215.  IGMC_HG graphs per workgroup; lin1.weight [128, D] is read in its native
// layout: one wave per output unit j, lanes across the fan-in (coalesced), wave reduction per (graph, j).
__device__ __forceinline__ float head_feat(const BatchDev& b, const ModelDev& m, int g, int k) {
  if (k < 256) {
    const int side = k >> 7, l = (k >> 5) & 3, f = k & 31;
    const int nu = b.node_off[g], nv = nu + b.n_users[g];
    return m.h[l][(size_t)(side ? nv : nu) * 32 + f];
  }
  return m.side[(size_t)g * m.S + (k - 256)];
}

template <bool FEAT_LDS>
__global__ __launch_bounds__(512) void k_head_fwd(BatchDev b, ModelDev m, const float* __restrict__ P, int training,
                                                    const uint8_t* __restrict__ inj_mask, uint64_t seed,
                                                    uint64_t step_arg, float mult, float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  __shared__ float sa[IGMC_HG][128];
  __shared__ float red[2][IGMC_HG];
  float* sfeat = (float*)smem;      // [HG][D]  (FEAT_LDS)
  const int B = b.totals[3];
  const int g0 = blockIdx.x * IGMC_HG, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int D = m.D;
  for (int idx = tid; idx < IGMC_HG * D; idx += 512) {
    const int gg = idx / D, k = idx % D, g = g0 + gg;
    const float v = (g < B) ? head_feat(b, m, g, k) : 0.f;
    if (g < B) m.feat[(size_t)g * D + k] = v;
    if (FEAT_LDS) sfeat[idx] = v;
  }
  __syncthreads();
  for (int j = wave; j < 128; j += 8) {
    float acc[IGMC_HG];
#pragma unroll
    for (int gg = 0; gg < IGMC_HG; ++gg) acc[gg] = 0.f;
    const float* wrow = P + m.off_l1w + (int64_t)j * D;
    for (int k = lane; k < D; k += 64) {
      const float w = wrow[k];
#pragma unroll
      for (int gg = 0; gg < IGMC_HG; ++gg) {
        const float fv = FEAT_LDS ? sfeat[gg * D + k] : ((g0 + gg < B) ? m.feat[(size_t)(g0 + gg) * D + k] : 0.f);
        acc[gg] += w * fv;
      }
    }
#pragma unroll
    for (int gg = 0; gg < IGMC_HG; ++gg) {
      const float s = igmc_wave_sum_f(acc[gg]);
      if (lane == 0) sa[gg][j] = s + P[m.off_l1b + j];
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int j = tid;
    const float w2 = P[m.off_l2w + j];
#pragma unroll
    for (int gg = 0; gg < IGMC_HG; ++gg) {
      const int g = g0 + gg;
      float a = sa[gg][j] > 0.f ? sa[gg][j] : 0.f;
      if (training && g < B) {
        m.a1[g * 128 + j] = a;
        const int keep = inj_mask ? (int)inj_mask[g * 128 + j]
                                  : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)j)) >= 0.5f);
        m.lmask[g * 128 + j] = (uint8_t)keep;
        a = keep ? a * 2.f : 0.f;    // F.dropout(p=0.5): kept units scaled by 1/(1-p)
      }
      const float p = igmc_wave_sum_f(a * w2);
      if ((j & 63) == 0) red[j >> 6][gg] = p;
    }
  }
  __syncthreads();
  if (tid < IGMC_HG && g0 + tid < B) {
    const float o = (red[0][tid] + red[1][tid] + P[m.off_l2b]) * mult;
    out[g0 + tid] = o;
    m.err[g0 + tid] = o - b.y[g0 + tid];
  }
}

// backward A: dz, d feat, and dPre of the top layer on the two target rows of each graph.
// 1024 threads = 4 slices of the 128 hidden units x 256 fan-in columns.
__global__ __launch_bounds__(1024) void k_head_bwd_a(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                       const float* __restrict__ gout, int from_err,
                                                       float grad_scale, float mult, float drop_scale,
                                                       float* __restrict__ dpre_top) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  __shared__ float sdz[IGMC_HG][128];
  __shared__ float sred[3][IGMC_HG][256];
  const int B = b.totals[3];
  const int g0 = blockIdx.x * IGMC_HG, tid = threadIdx.x;
  const int D = m.D;
  for (int idx = tid; idx < IGMC_HG * 128; idx += 1024) {
    const int gg = idx >> 7, j = idx & 127, g = g0 + gg;
    float dzv = 0.f;
    if (g < B) {
      const float dp = (from_err ? 2.f * m.err[g] * grad_scale : gout[g]) * mult;
      const float a = m.a1[g * 128 + j];
      dzv = (a > 0.f && m.lmask[g * 128 + j]) ? dp * P[m.off_l2w + j] * drop_scale : 0.f;
      m.dz[g * 128 + j] = dzv;
    }
    sdz[gg][j] = dzv;
  }
  __syncthreads();
  const int kk = tid & 255, js = tid >> 8;
  for (int kc = 0; kc < D; kc += 256) {
    const int k = kc + kk;
    float acc[IGMC_HG];
#pragma unroll
    for (int gg = 0; gg < IGMC_HG; ++gg) acc[gg] = 0.f;
    if (k < D) {
#pragma unroll 8
      for (int jj = js * 32; jj < js * 32 + 32; ++jj) {
        const float w = P[m.off_l1w + (int64_t)jj * D + k];
#pragma unroll
        for (int gg = 0; gg < IGMC_HG; ++gg) acc[gg] += sdz[gg][jj] * w;
      }
    }
    if (js > 0) {
#pragma unroll
      for (int gg = 0; gg < IGMC_HG; ++gg) sred[js - 1][gg][kk] = acc[gg];
    }
    __syncthreads();
    if (js == 0 && k < D) {
#pragma unroll
      for (int gg = 0; gg < IGMC_HG; ++gg) {
        const int g = g0 + gg;
        if (g >= B) continue;
        const float v = ((acc[gg] + sred[0][gg][kk]) + sred[1][gg][kk]) + sred[2][gg][kk];
        m.gfeat[(size_t)g * D + k] = v;
        if (k < 256 && ((k >> 5) & 3) == 3) {
          const int side = k >> 7, f = k & 31;
          const int nu = b.node_off[g], nv = nu + b.n_users[g];
          const size_t node = (size_t)(side ? nv : nu);
          const float hv = m.h[3][node * 32 + f];
          dpre_top[node * 32 + f] = v * (1.f - hv * hv);
        }
      }
    }
    __syncthreads();
  }
}

// backward B: d lin1.weight [128, D] = dz^T @ feat.  Block (jt, kc): 8 hidden units x 256 fan-in columns;
// feat / dz are staged through LDS in chunks of 32 graphs.  One extra block row does the bias/lin2 grads.
#define IGMC_HW_G 32
__global__ __launch_bounds__(IGMC_BLOCK) void k_head_bwd_w(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                             const float* __restrict__ gout, int from_err,
                                                             float grad_scale, float mult, float drop_scale,
                                                             float* __restrict__ grad) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  __shared__ float sfe[IGMC_HW_G][256];
  __shared__ float sdz[IGMC_HW_G][8];
  const int B = b.totals[3];
  const int D = m.D, tid = threadIdx.x;
  if (blockIdx.x == 16) {                 // d lin1.bias, d lin2.weight, d lin2.bias (blockIdx.y == 0 only)
    if (blockIdx.y != 0) return;
    if (tid < 128) {
      float s = 0.f;
      for (int g = 0; g < B; ++g) s += m.dz[g * 128 + tid];
      grad[m.off_l1b + tid] = s;
    } else {
      const int j = tid - 128;
      float s = 0.f, s2 = 0.f;
      for (int g = 0; g < B; ++g) {
        const float dp = (from_err ? 2.f * m.err[g] * grad_scale : gout[g]) * mult;
        const float a = m.lmask[g * 128 + j] ? m.a1[g * 128 + j] * drop_scale : 0.f;
        s += dp * a;
        s2 += dp;
      }
      grad[m.off_l2w + j] = s;
      if (j == 0) grad[m.off_l2b] = s2;
    }
    return;
  }
  const int j0 = blockIdx.x * 8, k = blockIdx.y * 256 + tid;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.f;
  for (int gc = 0; gc < B; gc += IGMC_HW_G) {
    const int ng = (B - gc < IGMC_HW_G) ? B - gc : IGMC_HW_G;
    for (int gg = 0; gg < ng; ++gg) sfe[gg][tid] = (k < D) ? m.feat[(size_t)(gc + gg) * D + k] : 0.f;
    if (tid < ng * 8) sdz[tid >> 3][tid & 7] = m.dz[(gc + (tid >> 3)) * 128 + j0 + (tid & 7)];
    __syncthreads();
    for (int gg = 0; gg < ng; ++gg) {
      const float fv = sfe[gg][tid];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += sdz[gg][q] * fv;
    }
    __syncthreads();
  }
  if (k < D) {
#pragma unroll
    for (int q = 0; q < 8; ++q) grad[m.off_l1w + (int64_t)(j0 + q) * D + k] = acc[q];
  }
}

// =================================================================== head on f32 MFMA (D % 16 == 0)
// The three head products are tiny GEMMs ([B,D]x[D,128], [B,128]x[128,D], [128,B]x[B,D]); both MFMA operands
// are loaded straight from global memory as float4 / 64-byte row segments (no LDS staging), so each kernel is
// a handful of MFMAs per wave and finishes in the launch floor.
__device__ __forceinline__ const float* head_feat_ptr(const BatchDev& b, const ModelDev& m, int g, int k) {
  if (k < 256) {
    const int side = k >> 7, l = (k >> 5) & 3, f = k & 31;
    const int nu = b.node_off[g], nv = nu + b.n_users[g];
    return m.h[l] + (size_t)(side ? nv : nu) * 32 + f;
  }
  return m.side + (size_t)g * m.S + (k - 256);
}

// lin1 (+ReLU, dropout) + lin2 for 16 graphs per workgroup; wave w owns hidden units [16w, 16w+16)
__global__ __launch_bounds__(512) void k_head_fwd_mfma(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                         int training, const uint8_t* __restrict__ inj_mask,
                                                         uint64_t seed, uint64_t step_arg, float mult,
                                                         float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  __shared__ float spart[8][16];
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  const int B = b.totals[3], D = m.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 16, n0 = wave * 16;
  const int ga = (row0 + li < B) ? row0 + li : B - 1;        // graph feeding the A fragment of this lane
  const float* wrow = P + m.off_l1w + (int64_t)(n0 + li) * D;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nch = D / 16;
  for (int s0 = 0; s0 < nch; s0 += 8) {          // 8 chunks (16 float4 loads) in flight before the MFMAs
    float4 a4[8], b4[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k0 = ((s0 + u < nch) ? s0 + u : nch - 1) * 16 + 4 * kq;
      a4[u] = *(const float4*)head_feat_ptr(b, m, ga, k0);
      b4[u] = *(const float4*)(wrow + k0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u >= nch) continue;
      const int k0 = (s0 + u) * 16 + 4 * kq;
      if (training && wave == 0 && row0 + li < B) *(float4*)(m.feat + (size_t)ga * D + k0) = a4[u];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].x, b4[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].y, b4[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].z, b4[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].w, b4[u].w, acc, 0, 0, 0);
    }
  }
  const int n = n0 + li;
  const float b1 = P[m.off_l1b + n], w2 = P[m.off_l2w + n];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = kq * 4 + rr, g = row0 + r;
    float a = acc[rr] + b1;
    a = a > 0.f ? a : 0.f;
    if (training && g < B) {
      m.a1[g * 128 + n] = a;
      const int keep = inj_mask ? (int)inj_mask[g * 128 + n]
                                : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)n)) >= 0.5f);
      m.lmask[g * 128 + n] = (uint8_t)keep;
      a = keep ? a * 2.f : 0.f;    // F.dropout(p=0.5): kept units scaled by 1/(1-p)
    }
    const float p = igmc_group16_sum_f(a * w2);
    if (li == 0) spart[wave][r] = p;
  }
  __syncthreads();
  if (threadIdx.x < 16 && row0 + (int)threadIdx.x < B) {
    const int g = row0 + threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += spart[w][threadIdx.x];
    const float o = (s + P[m.off_l2b]) * mult;
    out[g] = o;
    m.err[g] = o - b.y[g];
  }
}

// Training head in ONE launch (fused-step path): blockIdx.y == 0 -> forward of 16 graphs, their residual, and
// immediately the backward down to d feat / dPre of the top layer (dz never leaves the workgroup except for
// the copy the lin1 weight gradient needs); blockIdx.y = 1..3 -> the Y products of the conv layers.
__global__ __launch_bounds__(512) void k_head_train(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                      const uint8_t* __restrict__ inj_mask, uint64_t seed,
                                                      uint64_t step_arg, float mult, float grad_scale,
                                                      float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  if (blockIdx.y > 0) {
    const int ly = blockIdx.y - 1;
    if ((int)blockIdx.x * 128 >= b.totals[0]) return;       // sized by the arena's node capacity: no tile, no staging
    dense_body<0, 32, 128, EPI_NONE, W_YCAT, 512>(b, nullptr, m.h[ly], P + m.off_basis[ly + 1], nullptr, nullptr,
                                                  m.Y[ly], nullptr, nullptr, 0, 0, nullptr, (float*)smem);
    return;
  }
  const int B = b.totals[3], D = m.D;
  const int row0 = blockIdx.x * 16;
  if (row0 >= B) return;
  __shared__ float spart[8][16];
  __shared__ float serr[16];
  __shared__ float sdz[16][132];
  // Four workgroups carry this role at batch 50, so its duration is the number of DEPENDENT memory round trips, not a
  // throughput: everything a later stage reads from memory (labels, lin2, the lin1 columns of the d feat product, the
  // top-layer rows of the target nodes) is requested up front, next to the operands of lin1.
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int n0 = wave * 16;
  const int ga = (row0 + li < B) ? row0 + li : B - 1;
  const int n = n0 + li;
  const float* wrow = P + m.off_l1w + (int64_t)n * D;
  const int nch = D / 16;
  // targets' node rows of this lane's graphs (A rows of lin1: graph ga; epilogue of d feat: graphs row0 + 4 kq + rr)
  const int nu_a = b.node_off[ga], nv_a = nu_a + b.n_users[ga];
  int nu_e[4], nv_e[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int g = row0 + kq * 4 + rr, gs = g < B ? g : B - 1;
    nu_e[rr] = b.node_off[gs];
    nv_e[rr] = nu_e[rr] + b.n_users[gs];
  }
  auto feat_ptr = [&](int k0) -> const float* {
    if (k0 < 256) return m.h[(k0 >> 5) & 3] + (size_t)((k0 >> 7) ? nv_a : nu_a) * 32 + (k0 & 31);
    return m.side + (size_t)ga * m.S + (k0 - 256);
  };
  // addresses as (uniform base) + (32-bit byte offset of the lane): one offset register serves every load of a kind and
  // the bases live in scalar registers -- the role is VALU-bound, not memory-bound, once its loads are batched (8 waves
  // on each of 4 CUs; address arithmetic was a third of its instructions)
  auto ld4 = [](const float* base, uint32_t byte) { return *(const float4*)((const char*)base + byte); };
  auto ld1 = [](const float* base, uint32_t byte) { return *(const float*)((const char*)base + byte); };
  const uint32_t offU = ((uint32_t)nu_a * 32u + 4u * kq) * 4u, offV = ((uint32_t)nv_a * 32u + 4u * kq) * 4u;
  const uint32_t offW = ((uint32_t)n * (uint32_t)D + 4u * kq) * 4u;
  f32x4 acc4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc4[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a4[16], b4[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {                    // chunks 0..15 = the 256 conv features (D >= 256): k0 = 16 u + 4 kq
    a4[u] = ld4(m.h[(u >> 1) & 3] + (u & 1) * 16, (u >> 3) ? offV : offU);
    b4[u] = ld4(P + m.off_l1w + u * 16, offW);
  }
  const float b1 = P[m.off_l1b + n], w2 = P[m.off_l2w + n];
  const float l2b = P[m.off_l2b];
  const float yv = (threadIdx.x < 16 && row0 + (int)threadIdx.x < B) ? b.y[row0 + threadIdx.x] : 0.f;
  // d feat = dz @ lin1.weight: column tile nt = wave + 8 it (< 16: inside the conv features); its 32 x 4 weight values
  // and (tiles 6, 7 of a side = the top layer's slice of the concatenation) h_3 of the target rows
  float bvp[2][8][4];
  float hvp[2][4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c0 = (wave + 8 * it) * 16;
    const uint32_t offB = ((uint32_t)(4 * kq) * (uint32_t)D + (uint32_t)(c0 + li)) * 4u;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) bvp[it][u][mm] = ld1(P + m.off_l1w + (size_t)(u * 16 + mm) * D, offB);
    const int k = c0 + li;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      hvp[it][rr] = 0.f;
      if (((k >> 5) & 3) == 3) hvp[it][rr] = ld1(m.h[3], ((uint32_t)((k >> 7) ? nv_e[rr] : nu_e[rr]) * 32u + (uint32_t)(k & 31)) * 4u);
    }
  }
  for (int s0 = 0; s0 < nch; s0 += 16) {
    if (s0 > 0) {                                   // side features: D > 256
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k0 = ((s0 + u < nch) ? s0 + u : nch - 1) * 16 + 4 * kq;
        a4[u] = *(const float4*)feat_ptr(k0);
        b4[u] = *(const float4*)(wrow + k0);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (s0 + u >= nch) continue;
      const int k0 = (s0 + u) * 16 + 4 * kq;
      if (wave == 0 && row0 + li < B) *(float4*)(m.feat + (size_t)ga * D + k0) = a4[u];
      f32x4& acc = acc4[u & 3];                     // four independent chains instead of 64 dependent MFMAs
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].x, b4[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].y, b4[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].z, b4[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].w, b4[u].w, acc, 0, 0, 0);
    }
  }
  float av[4];
  int kp[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = kq * 4 + rr, g = row0 + r;
    float a = ((acc4[0][rr] + acc4[1][rr]) + (acc4[2][rr] + acc4[3][rr])) + b1;
    a = a > 0.f ? a : 0.f;
    av[rr] = a;
    kp[rr] = 0;
    if (g < B) {
      m.a1[g * 128 + n] = a;
      kp[rr] = inj_mask ? (int)inj_mask[g * 128 + n]
                        : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)n)) >= 0.5f);
      m.lmask[g * 128 + n] = (uint8_t)kp[rr];
    }
    const float p = igmc_group16_sum_f((kp[rr] ? a * 2.f : 0.f) * w2);
    if (li == 0) spart[wave][r] = p;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int g = row0 + threadIdx.x;
    float e = 0.f;
    if (g < B) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += spart[w][threadIdx.x];
      const float o = (s + l2b) * mult;
      out[g] = o;
      e = o - yv;
      m.err[g] = e;
    }
    serr[threadIdx.x] = e;
  }
  __syncthreads();
  // ---- backward through lin2 / dropout / relu: dz (kept in LDS for the next product, copied out for d lin1.weight)
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = kq * 4 + rr, g = row0 + r;
    const float dp = 2.f * serr[r] * grad_scale * mult;
    const float dzv = (g < B && av[rr] > 0.f && kp[rr]) ? dp * w2 * 2.f : 0.f;
    sdz[r][n] = dzv;
    if (g < B) m.dz[g * 128 + n] = dzv;
  }
  __syncthreads();
  // ---- d feat = dz @ lin1.weight  (16 x 128 @ 128 x D), column tiles strided over the 8 waves
  for (int nt = wave, it = 0; nt * 16 < D; nt += 8, ++it) {
    const int c0 = nt * 16;
    f32x4 g4a = (f32x4){0.f, 0.f, 0.f, 0.f}, g4b = g4a;
    if (it < 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          f32x4& g4 = (u & 1) ? g4b : g4a;
          g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(sdz[li][u * 16 + 4 * kq + mm], bvp[it][u][mm], g4, 0, 0, 0);
        }
    } else {
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm)
          g4a = __builtin_amdgcn_mfma_f32_16x16x4f32(sdz[li][u * 16 + 4 * kq + mm],
                                                     P[m.off_l1w + (int64_t)(u * 16 + 4 * kq + mm) * D + c0 + li], g4a, 0, 0, 0);
    }
    const int k = c0 + li;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int g = row0 + kq * 4 + rr;
      if (g >= B) continue;
      const float v = g4a[rr] + g4b[rr];
      m.gfeat[(size_t)g * D + k] = v;
      if (k < 256 && ((k >> 5) & 3) == 3) {
        const size_t node = (size_t)((k >> 7) ? nv_e[rr] : nu_e[rr]);
        const float hv = hvp[it & 1][rr];
        m.dpre[3][node * 32 + (k & 31)] = v * (1.f - hv * hv);
      }
    }
  }
}

// d feat = dz @ lin1.weight, dz formed on the fly; wave -> 16 fan-in columns; also dPre of the top layer
__global__ __launch_bounds__(512) void k_head_bwd_a_mfma(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                           const float* __restrict__ gout, int from_err,
                                                           float grad_scale, float mult, float drop_scale,
                                                           float* __restrict__ dpre_top) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  const int B = b.totals[3], D = m.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int nt = blockIdx.y * 8 + wave;
  if (nt * 16 >= D) return;
  const int n0 = nt * 16;
  const int ga = (row0 + li < B) ? row0 + li : B - 1;
  const bool arow_ok = row0 + li < B;
  const float dp = (from_err ? 2.f * m.err[ga] * grad_scale : gout[ga]) * mult;
  const float* w1 = P + m.off_l1w;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int j0 = s * 16 + 4 * kq;
    const float4 a4 = *(const float4*)(m.a1 + ga * 128 + j0);
    const uint32_t mk = *(const uint32_t*)(m.lmask + ga * 128 + j0);
    const float4 w4 = *(const float4*)(P + m.off_l2w + j0);
    float dzv[4];
    dzv[0] = (arow_ok && a4.x > 0.f && (mk & 0xFFu)) ? dp * w4.x * drop_scale : 0.f;
    dzv[1] = (arow_ok && a4.y > 0.f && ((mk >> 8) & 0xFFu)) ? dp * w4.y * drop_scale : 0.f;
    dzv[2] = (arow_ok && a4.z > 0.f && ((mk >> 16) & 0xFFu)) ? dp * w4.z * drop_scale : 0.f;
    dzv[3] = (arow_ok && a4.w > 0.f && ((mk >> 24) & 0xFFu)) ? dp * w4.w * drop_scale : 0.f;
    if (nt == 0 && arow_ok) {
      float4 o;
      o.x = dzv[0]; o.y = dzv[1]; o.z = dzv[2]; o.w = dzv[3];
      *(float4*)(m.dz + ga * 128 + j0) = o;
    }
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      const float bv = w1[(int64_t)(j0 + mm) * D + n0 + li];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv[mm], bv, acc, 0, 0, 0);
    }
  }
  const int k = n0 + li;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int g = row0 + kq * 4 + rr;
    if (g >= B) continue;
    const float v = acc[rr];
    m.gfeat[(size_t)g * D + k] = v;
    if (k < 256 && ((k >> 5) & 3) == 3) {
      const int side = k >> 7, f = k & 31;
      const int nu = b.node_off[g], nv = nu + b.n_users[g];
      const size_t node = (size_t)(side ? nv : nu);
      const float hv = m.h[3][node * 32 + f];
      dpre_top[node * 32 + f] = v * (1.f - hv * hv);
    }
  }
}

// d lin1.weight = dz^T @ feat (reduction over the B graphs, 4 per MFMA); block x = 16 hidden units, wave -> 16
// fan-in columns.  Wave 0 of the y == 0 blocks also forms d lin1.bias = dz^T 1, d lin2.weight = adrop^T dp and
// d lin2.bias = 1^T dp with three more MFMAs per step (B operand = 1 / dp_g), so there is no serial tail.
__device__ __forceinline__ void head_bwd_w_body(const BatchDev& b, const ModelDev& m, const float* __restrict__ P,
                                                const float* __restrict__ gout, int from_err, float grad_scale,
                                                float mult, float drop_scale, float* __restrict__ grad, int bx, int by,
                                                int B_known = -1) {
  // (the batch size: a caller that has it in its arguments saves the dependent round trip to the arena's totals)
  const int B = B_known >= 0 ? B_known : b.totals[3], D = m.D, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int j0 = bx * 16;
  const int nt = by * 4 + wave;
  if (nt * 16 >= D) return;
  const int n0 = nt * 16;
  const bool extra = nt == 0;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 accb1 = acc, accw2 = acc, accb2 = acc;
  for (int g0 = 0; g0 < B; g0 += 32) {            // 8 MFMA steps (32 graphs) of loads in flight
    // every operand of the 8 steps is requested before the first use, at clamped addresses (a predicated load is a branch
    // and a wait; lmask -> a1 was a dependent pair): the wave with the extra products sits on the critical path of the
    // launch (k_tail_ts / k_wgrad_head), 32 serial round trips before
    float av[8], bv[8], ad[8], dpv[8];
    uint32_t mk[8];                     // (one register each: packed bytes would need every value right after its load)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = g0 + 4 * u + kq;
      const int gs = g < B ? g : B - 1;
      av[u] = m.dz[gs * 128 + j0 + li];
      bv[u] = m.feat[(size_t)gs * D + n0 + li];
      dpv[u] = 0.f;
      ad[u] = 0.f;
      mk[u] = 0;
      if (extra) {
        dpv[u] = from_err ? m.err[gs] : gout[gs];
        ad[u] = m.a1[gs * 128 + j0 + li];
        mk[u] = m.lmask[gs * 128 + j0 + li];
      }
    }
#ifndef IGMC_HIPEMU
    __builtin_amdgcn_sched_barrier(0);        // (keeps the uses below from being hoisted between the loads)
    asm volatile("" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]), "+v"(mk[4]), "+v"(mk[5]), "+v"(mk[6]), "+v"(mk[7]));
#endif
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = g0 + 4 * u + kq < B;
      if (!ok) { av[u] = 0.f; bv[u] = 0.f; }
      if (extra) {
        dpv[u] = ok ? (from_err ? 2.f * dpv[u] * grad_scale : dpv[u]) * mult : 0.f;
        ad[u] = (ok && mk[u]) ? ad[u] * drop_scale : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      if (extra) {
        accb1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], (g0 + 4 * u + kq < B) ? 1.f : 0.f, accb1, 0, 0, 0);
        accw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[u], dpv[u], accw2, 0, 0, 0);
        accb2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, dpv[u], accb2, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) grad[m.off_l1w + (int64_t)(j0 + kq * 4 + rr) * D + n0 + li] = acc[rr];
  if (extra && li == 0) {       // every column of the extra accumulators holds the same sums
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      grad[m.off_l1b + j0 + kq * 4 + rr] = accb1[rr];
      grad[m.off_l2w + j0 + kq * 4 + rr] = accw2[rr];
    }
    if (bx == 0 && kq == 0) grad[m.off_l2b] = accb2[0];
  }
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_head_bwd_w_mfma(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                                  const float* __restrict__ gout, int from_err,
                                                                  float grad_scale, float mult, float drop_scale,
                                                                  float* __restrict__ grad) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  head_bwd_w_body(b, m, P, gout, from_err, grad_scale, mult, drop_scale, grad, blockIdx.x, blockIdx.y);
}

// ONE launch for every weight-gradient product of the step: blockIdx.y < nsl -> conv weight-gradient slice,
// blockIdx.y == nsl -> d lin1 / d lin2 (linear block id = (j tile, column tile)).
__global__ __launch_bounds__(IGMC_BLOCK) void k_wgrad_head(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                             const float* __restrict__ gout, int from_err,
                                                             float grad_scale, float mult, float drop_scale,
                                                             float* __restrict__ grad, int nsl) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  if ((int)blockIdx.y < nsl) {
    wgrad_body(b, m, blockIdx.y);
  } else {
    const int ny = (m.D / 16 + 3) / 4;
    if ((int)blockIdx.x < 8 * ny)
      head_bwd_w_body(b, m, P, gout, from_err, grad_scale, mult, drop_scale, grad, blockIdx.x & 7, blockIdx.x >> 3);
  }
}

// =================================================================== partial reduction + finalize
// graw layout: [3][5152] conv1..3 (32x160 + 32) | [3][R*4] d att | [(R*L+L+1)*32] layer-0 table
// Sections A (weight-gradient partials) and C (layer-0 tables): 64 outputs x 4 partial-slices per block;
// section B (d att, few outputs x many partials): one wave per output.  Fixed summation order.
__device__ __forceinline__ void fin_stash_body(const ModelDev& m, const float* __restrict__ P, int l, const int64_t* ctrl);

// nstash = 4: four extra workgroups stash the weights-only quantities of the conv layers for k_finalize_ts (basis-space
// mode), like the stash role of k_tail_ts.
__global__ __launch_bounds__(IGMC_BLOCK) void k_reduce_partials(ModelDev m, int n_gatt_parts, int l0_mfma,
                                                                  int n_wg_parts, const float* __restrict__ P,
                                                                  const int64_t* ctrl, int nstash) {
  igmc_kernarg_warm<sizeof(ModelDev) + 32>();
  __shared__ float sred[4][64];
  if (nstash && (int)blockIdx.x >= (int)gridDim.x - nstash) {
    fin_stash_body(m, P, (int)blockIdx.x - ((int)gridDim.x - nstash), ctrl);
    return;
  }
  const int wgs = 32 * IGMC_KCAT + 32, na = m.R * 4, rows0 = m.R * m.L + m.L + 1, n0 = rows0 * 32;
  const int nlay = l0_mfma ? 4 : 3;
  const int nblkA = (nlay * wgs + 63) / 64, nblkB = (3 * na + 3) / 4, nblkC = l0_mfma ? 0 : (n0 + 63) / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int blk = blockIdx.x;
  if (blk < nblkA + nblkC) {
    const bool secA = blk < nblkA;
    const int o = (secA ? blk : blk - nblkA) * 64 + lane;
    const int nout = secA ? nlay * wgs : n0;
    float s = 0.f;
    if (o < nout) {
      if (secA) {
        const int l = o / wgs, i = o % wgs;
        // 16 partials of a wave per round trip (clamped addresses; same order of additions)
        const float* p = m.wg_part + (size_t)l * IGMC_WG_BLOCKS * wgs + i;
        for (int k0 = wave; k0 < n_wg_parts; k0 += 64) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u;
            v[u] = p[(size_t)(k < n_wg_parts ? k : n_wg_parts - 1) * wgs];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (k0 + 4 * u < n_wg_parts) s += v[u];
        }
      } else {
        const float* p = m.l0_part + o;
        for (int k0 = wave; k0 < IGMC_L0_BLOCKS; k0 += 64) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int k = k0 + 4 * u;
            v[u] = p[(size_t)(k < IGMC_L0_BLOCKS ? k : IGMC_L0_BLOCKS - 1) * n0];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (k0 + 4 * u < IGMC_L0_BLOCKS) s += v[u];
        }
      }
    }
    sred[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && o < nout) {
      const float tot = (sred[0][lane] + sred[1][lane]) + (sred[2][lane] + sred[3][lane]);
      if (secA && o >= 3 * wgs) {
        // layer-0 slice of k_wgrad_all: partial[c][128 + f]  ->  table T0[c][f]
        const int i = o - 3 * wgs, c = i / IGMC_KCAT, col = i % IGMC_KCAT;
        if (i < 32 * IGMC_KCAT && c < rows0 && col >= 128) m.graw[3 * wgs + 3 * na + c * 32 + (col - 128)] = tot;
      } else {
        m.graw[secA ? o : 3 * wgs + 3 * na + o] = tot;
      }
    }
  } else {
    blk -= nblkA + nblkC;
    const int o = blk * 4 + wave;
    if (o < 3 * na && blk < nblkB) {
      const int l = o / na, i = o % na;
      const float* p = m.gatt_part + (size_t)l * IGMC_GATHER_BLOCKS * na + i;
      float s = 0.f;
      for (int k0 = lane; k0 < n_gatt_parts; k0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 64 * u;
          v[u] = p[(size_t)(k < n_gatt_parts ? k : n_gatt_parts - 1) * na];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + 64 * u < n_gatt_parts) s += v[u];
      }
      s = igmc_wave_sum_f(s);
      if (lane == 0) m.graw[3 * wgs + o] = s;
    }
  }
}

// relation-space tables of graphstep.hip: ts_raw[l][i] = sum over the workgroups' partials (fixed order);
// 64 outputs per block, the 4 waves split the partial slices
// Pold != NULL: wave 0 also forms, per 32 consecutive outputs (always inside one dW_r block: ts_stride and fin*32 are
// multiples of 32), the partial dot products <dW_r, basis_b> (b = 0..3) with the CURRENT parameters -- d att[r,b] is
// the sum of the fin partials of its block (k_finalize_ts), so that kernel never reads a basis element it does not own.
__device__ __forceinline__ void reduce_ts_body(const ModelDev& m, int nparts, int stride, int B, int blk,
                                               const float* __restrict__ Pold = nullptr) {
  // 64 outputs a workgroup as 16 float4 columns x 16 partial groups: thread (column o4 = lane & 15, group pg = 4 wave +
  // (lane >> 4)) sums the partial slots pg, pg + 16, .. of its four outputs (<= 14 sixteen-byte loads, all requested before
  // the first add: ONE round trip; a wave's load instruction covers four 256-byte runs), the 16 groups are then added in
  // index order through LDS -- a fixed order, so the result does not depend on the launch.  ~90 registers a thread: every
  // workgroup of the launch is resident at once (the 56-loads-a-thread form needed 292 and took two residency rounds).
  __shared__ float4 sred4[16][16];
  const int ts = m.ts_stride, rows0 = m.R * m.L + m.L + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int o4 = lane & 15, pg = wave * 4 + (lane >> 4);
  const int o0 = blk * 64 + 4 * o4;                     // first of the thread's four outputs (ts is a multiple of 32: the four share a layer)
  const int l = o0 / ts, i0 = o0 % ts;
  const bool ok4 = l < 4 && (l > 0 || i0 < rows0 * 32);
  // wave 0's second role (Pold): the four basis values of ITS output blk * 64 + lane, requested together with the partials
  float pbv[4] = {0.f, 0.f, 0.f, 0.f};
  if (Pold && wave == 0) {
    const int o = blk * 64 + lane, lo_ = o / ts, i = o % ts;
    if (lo_ < 4) {
      const int nE = ((lo_ == 0) ? m.L : 32) * 32;
      if (i < m.R * nE) {
        const float* basis = Pold + m.off_basis[lo_] + i % nE;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) pbv[bb] = basis[bb * nE];
      }
    }
  }
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok4) {
    // slot of member c of subgraph g = g + c * stride (one-workgroup launches: stride = IGMC_TS_BLOCKS, cs = 1); the valid
    // slots in the order (c, g) are numbered q = c * ng + g
    const float* p = m.ts_part + (size_t)l * IGMC_TS_BLOCKS * ts + i0;
    const int ng = (nparts < stride) ? nparts : ((B < stride) ? B : stride), cs = (nparts + stride - 1) / stride;
    const int nq = cs * ng;
    for (int q0 = pg; q0 < nq; q0 += 16 * 14) {
      float4 v[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        const int q = q0 + 16 * u, qc = q < nq ? q : nq - 1;
        const int c = qc / ng, g = qc - c * ng;
        v[u] = *(const float4*)(p + (size_t)(c * stride + g) * ts);
      }
#pragma unroll
      for (int u = 0; u < 14; ++u)
        if (q0 + 16 * u < nq) {
          s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w;
        }
    }
  }
  sred4[pg][o4] = s4;
  __syncthreads();
  if (wave == 0) {
    // lane -> output blk * 64 + lane: component (lane & 3) of column (lane >> 2), the 16 groups in index order
    const float* sr = (const float*)sred4;
    float tot = 0.f;
#pragma unroll
    for (int g16 = 0; g16 < 16; ++g16) tot += sr[(g16 * 16 + (lane >> 2)) * 4 + (lane & 3)];
    const int o = blk * 64 + lane;
    const int lo_ = o / ts, i = o % ts;
    const bool ok = lo_ < 4 && (lo_ > 0 || i < rows0 * 32);
    if (!ok) tot = 0.f;
    if (ok) m.ts_raw[o] = tot;
    if (Pold) {
      float pb[4];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) pb[bb] = tot * pbv[bb];      // (pbv is zero outside the dW_r blocks)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) pb[bb] += __shfl_xor(pb[bb], d, 64);     // stays inside a 32-lane half
      if ((lane & 31) == 0 && lo_ < 4) *(float4*)(m.datt_part + (size_t)(o >> 5) * 4) = make_float4(pb[0], pb[1], pb[2], pb[3]);
    }
  }
}

// Weights-only quantities of conv layer l for k_finalize_ts, formed while the tables are being reduced: Gram matrix of
// the bases, ARR matrix M[b][b'] = sum_r att[r,b] c[r,b'], the ARR value, a copy of att (k_finalize_ts updates att in
// place while other workgroups still need the old values) and, from the control block, the Adam scalars of the step.
#define IGMC_STASH_G 0
#define IGMC_STASH_M 16
#define IGMC_STASH_ATTM1 32       // Adam moments of att before the step (R <= 16; k_finalize_ts with img: every workgroup of
#define IGMC_STASH_ATTM2 96       // the layer forms the new att, while the owner updates the moments in place)
#define IGMC_STASH_ATT 160        // copy of att: R <= 128
// (IGMC_STASH_LAYER floats per layer: model.h)
#define IGMC_STASH_SCAL (4 * IGMC_STASH_LAYER)
__device__ __forceinline__ void fin_stash_body(const ModelDev& m, const float* __restrict__ P, int l,
                                               const int64_t* ctrl) {
  __shared__ float sg10[4][10];
  __shared__ float sG[16];
  __shared__ float s_att[128];                       // att[r][b] of the layer (R <= 32: else read from HBM where needed)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nE = ((l == 0) ? m.L : 32) * 32, R = m.R, na = R * 4;
  const float* basis = P + m.off_basis[l];
  const float* attg = P + m.off_att[l];
  float* st = m.fin_stash + l * IGMC_STASH_LAYER;
  // Every load of the role is requested HERE, before the first use: att, the thread's (<= 4) elements of the four bases, att's
  // Adam moments, the step's scalars.  Left inside the loops below they were a dozen dependent round trips -- the longest
  // chain of the whole k_tail_ts launch.
  const bool small = na <= 128;
  const float attv = (small && tid < na) ? attg[tid] : 0.f;
  float bq[4][4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + it * IGMC_BLOCK, ec = e < nE ? e : nE - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[it][q] = (e < nE) ? basis[q * nE + ec] : 0.f;
  }
  float m1v = 0.f, m2v = 0.f;
  const bool mom = m.adam_m1 && na <= 64 && tid >= 128 && tid < 128 + na;
  if (mom) {
    m1v = m.adam_m1[m.off_att[l] + tid - 128];
    m2v = m.adam_m2[m.off_att[l] + tid - 128];
  }
  double scal = 0.0;
  const bool sc = l == 0 && ctrl && tid >= 192 && tid < 198;
  if (sc) {
    const double* d = (const double*)ctrl;
    const int k = tid - 192;
    const int src = (k == 0) ? IGMC_CTRL_STEP_SIZE : (k == 1) ? IGMC_CTRL_INV_SQRT_BC2 : (k == 2) ? IGMC_CTRL_BETA1
                  : (k == 3) ? IGMC_CTRL_BETA2 : (k == 4) ? IGMC_CTRL_EPS : IGMC_CTRL_WD;
    scal = d[src];
  }
  if (small && tid < na) s_att[tid] = attv;
  float gp[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) gp[q] = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {                     // (element order per thread as before: e = tid, tid + 256, ..)
    if (tid + it * IGMC_BLOCK < nE) {
      const float b0 = bq[it][0], b1 = bq[it][1], b2 = bq[it][2], b3 = bq[it][3];
      gp[0] += b0 * b0; gp[1] += b0 * b1; gp[2] += b0 * b2; gp[3] += b0 * b3;
      gp[4] += b1 * b1; gp[5] += b1 * b2; gp[6] += b1 * b3;
      gp[7] += b2 * b2; gp[8] += b2 * b3; gp[9] += b3 * b3;
    }
  }
  for (int e = tid + 4 * IGMC_BLOCK; e < nE; e += IGMC_BLOCK) {      // (nE <= 1024: never)
    const float b0 = basis[e], b1 = basis[nE + e], b2 = basis[2 * nE + e], b3 = basis[3 * nE + e];
    gp[0] += b0 * b0; gp[1] += b0 * b1; gp[2] += b0 * b2; gp[3] += b0 * b3;
    gp[4] += b1 * b1; gp[5] += b1 * b2; gp[6] += b1 * b3;
    gp[7] += b2 * b2; gp[8] += b2 * b3; gp[9] += b3 * b3;
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) gp[q] = igmc_wave_sum_f(gp[q]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 10; ++q) sg10[wave][q] = gp[q];
  }
  __syncthreads();                                   // s_att, sg10
  const float* att = small ? (const float*)s_att : attg;
  if (tid >= 64 && tid < 80) {           // M[b][b'] = sum_r att[r,b] c[r,b'],  c[r] = 2 (d[r-1] - d[r]),  d[r] = att[r+1]-att[r]
    const int bb = (tid - 64) >> 2, bp = tid & 3;
    float sacc = 0.f;
    for (int r = 0; r < R; ++r) {
      const float dm = (r > 0) ? att[r * 4 + bp] - att[(r - 1) * 4 + bp] : 0.f;
      const float dn = (r + 1 < R) ? att[(r + 1) * 4 + bp] - att[r * 4 + bp] : 0.f;
      sacc += att[r * 4 + bb] * 2.f * (dm - dn);
    }
    st[IGMC_STASH_M + tid - 64] = sacc;
  }
  if (tid >= 128 && na <= IGMC_STASH_LAYER - IGMC_STASH_ATT) {
    for (int i = tid - 128; i < na; i += IGMC_BLOCK - 128) {
      st[IGMC_STASH_ATT + i] = att[i];
      if (m.adam_m1 && na <= 64) {
        st[IGMC_STASH_ATTM1 + i] = mom && i == tid - 128 ? m1v : m.adam_m1[m.off_att[l] + i];
        st[IGMC_STASH_ATTM2 + i] = mom && i == tid - 128 ? m2v : m.adam_m2[m.off_att[l] + i];
      }
    }
  }
  if (sc) m.fin_stash[IGMC_STASH_SCAL + tid - 192] = (float)scal;
  if (tid == 0) {
    const int ij[10][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}, {1, 2}, {1, 3}, {2, 2}, {2, 3}, {3, 3}};
    for (int q = 0; q < 10; ++q) {
      const float v = (sg10[0][q] + sg10[1][q]) + (sg10[2][q] + sg10[3][q]);
      sG[ij[q][0] * 4 + ij[q][1]] = v;
      sG[ij[q][1] * 4 + ij[q][0]] = v;
    }
    float reg = 0.f;                       // reg = sum_r d[r]^T Gm d[r]   (reference train_eval.py:167-174)
    for (int r = 0; r + 1 < R; ++r) {
      float d[4];
      for (int q = 0; q < 4; ++q) d[q] = att[(r + 1) * 4 + q] - att[r * 4 + q];
      for (int p1 = 0; p1 < 4; ++p1)
        for (int p2 = 0; p2 < 4; ++p2) reg += d[p1] * sG[p1 * 4 + p2] * d[p2];
    }
    m.arr_part[l] = reg;
    for (int q = 0; q < 16; ++q) st[IGMC_STASH_G + q] = sG[q];
  }
}

// Both consumers of k_graph_step's outputs in ONE launch (they are independent of each other): the first `nlin`
// workgroups form d lin1 / d lin2 (batched product over the subgraphs), the others sum the relation-space tables.
// nstash = 4: the last four workgroups stash the weights-only quantities k_finalize_ts needs and the reduction also
// forms the d att partial products (0: the tables only, for k_finalize).
__global__ __launch_bounds__(IGMC_BLOCK) void k_tail_ts(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                          float grad_scale, float mult, float drop_scale,
                                                          float* __restrict__ grad, int nlin, int nparts, int stride,
                                                          int B, int nstash, const int64_t* ctrl, int bump_seq) {
  __builtin_amdgcn_s_setprio(3);      // (step chain: ahead of the extraction chain's waves wherever the two share a SIMD)
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 64>();
  const int nred = (int)gridDim.x - nlin - nstash;
  if (bump_seq && blockIdx.x == 0 && threadIdx.x == 0) m.gs_bar[1] += 1;      // (every workgroup of k_graph_step2 is done)
  if ((int)blockIdx.x < nlin)
    head_bwd_w_body(b, m, P, nullptr, 1, grad_scale, mult, drop_scale, grad, blockIdx.x & 7, blockIdx.x >> 3, B);
  else if ((int)blockIdx.x < nlin + nred)
    reduce_ts_body(m, nparts, stride, B, blockIdx.x - nlin, nstash ? P : nullptr);
  else
    fin_stash_body(m, P, blockIdx.x - nlin - nred, ctrl);
}

struct FinishArgs {
  int enabled;
  int use_flags;      // the step ran under edge dropout (the tick then also checks the arena's dropout stamp)
  BatchDev b;
  ModelDev m;
  float ARR;
  float* loss;
  double* total;
};

// End of a step (igmc_hip.h, device-side step control): counters, Adam bias corrections, and -- at the end of a GROUP of M
// steps -- the cursor of the group's parity moves on by two groups and the parity flips.  Only the cursor of the group that
// just finished is written: the one a concurrent prefetch of the next group reads never changes while it may be read.
__device__ __forceinline__ void ctrl_advance(int64_t* ctrl) {
  double* d = (double*)ctrl;
  const int64_t M = ctrl[IGMC_CTRL_GROUP] > 0 ? ctrl[IGMC_CTRL_GROUP] : 1;
  const int64_t gq = ctrl[IGMC_CTRL_GQ] & 1, gk = ctrl[IGMC_CTRL_GK] + 1;
  ctrl[IGMC_CTRL_STEP] += 1;
  ctrl[IGMC_CTRL_K] += 1;
  if (gk >= M) {
    ctrl[gq ? IGMC_CTRL_FIRST_ODD : IGMC_CTRL_FIRST] += 2 * M * ctrl[IGMC_CTRL_BATCH];
    ctrl[IGMC_CTRL_GK] = 0;
    ctrl[IGMC_CTRL_GQ] = gq ^ 1;
  } else {
    ctrl[IGMC_CTRL_GK] = gk;
  }
  ctrl[IGMC_CTRL_ADAM_T] += 1;
  const double t = (double)ctrl[IGMC_CTRL_ADAM_T];
  d[IGMC_CTRL_STEP_SIZE] = d[IGMC_CTRL_LR] / (1.0 - pow(d[IGMC_CTRL_BETA1], t));
  d[IGMC_CTRL_INV_SQRT_BC2] = 1.0 / sqrt(1.0 - pow(d[IGMC_CTRL_BETA2], t));
}
// ... after checking that the arena the step consumed held the batch of its cursor (and, under edge dropout, that batch's
// draws): every extraction stamps its arena with the `first` it resolved.  A mismatch means a batch extracted from a stale
// cursor or an arena overwritten too early; it sets sync_err instead of training on silently.
__device__ __forceinline__ void ctrl_check_and_advance(int64_t* ctrl, const BatchDev& b, int use_flags) {
  const int64_t gq = ctrl[IGMC_CTRL_GQ] & 1;
  const int64_t want = ctrl[gq ? IGMC_CTRL_FIRST_ODD : IGMC_CTRL_FIRST] + ctrl[IGMC_CTRL_GK] * ctrl[IGMC_CTRL_BATCH];
  const int64_t got = b.stamp[0];
  int bad = 0;
  if (got >= 0 && got != want) bad |= 2;
  if (use_flags && got >= 0 && b.stamp[1] >= 0 && (uint64_t)b.stamp[1] != igmc_ctrl_drop_key(ctrl, (int)want)) bad |= 4;
  if (bad) ctrl[IGMC_CTRL_SYNC_ERR] |= bad;
  ctrl_advance(ctrl);
}

// loss[0] = mean_g err^2 + ARR * sum_l reg_l   (reference train_eval.py:162-174); loss[1] = sum err^2
__device__ __forceinline__ void loss_body(const BatchDev& b, const ModelDev& m, float ARR, float* loss,
                                          double* total, float* smf) {
  const int B = b.totals[3];
  float s = 0.f;
  for (int g = threadIdx.x; g < B; g += IGMC_BLOCK) s += m.err[g] * m.err[g];
  s = igmc_block_sum_f(s, smf);
  if (threadIdx.x == 0) {
    const float reg = m.arr_part[0] + m.arr_part[1] + m.arr_part[2] + m.arr_part[3];
    const float l0 = s / (float)B + ARR * reg;
    loss[0] = l0;
    loss[1] = s;
    if (total) total[0] += (double)l0 * (double)B;     // epoch total of loss * num_graphs (ref :176)
  }
}

// one Adam update (torch.optim.Adam semantics)
__device__ __forceinline__ void adam_elem(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                                          float* __restrict__ m2, int64_t i, float step_size, float inv_sqrt_bc2,
                                          float beta1, float beta2, float eps, float wd) {
  float gi = g[i];
  const float pi = p[i];
  if (wd != 0.f) gi += wd * pi;
  const float a = beta1 * m1[i] + (1.f - beta1) * gi;
  const float v = beta2 * m2[i] + (1.f - beta2) * gi * gi;
  m1[i] = a;
  m2[i] = v;
  p[i] = pi - step_size * a / (sqrtf(v) * inv_sqrt_bc2 + eps);
}

// Adam on [lo, hi) by one workgroup: NB elements per thread and round, all 4 NB loads requested before the first use
// (the element-by-element loop pays one memory round trip per element: the stores of element k keep the compiler from
// requesting element k + 1 early)
template <int NB>
__device__ __forceinline__ void adam_range(float* p, const float* g, float* m1, float* m2, int64_t lo, int64_t hi,
                                           float step_size, float inv_sqrt_bc2, float beta1, float beta2, float eps,
                                           float wd) {
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)NB * IGMC_BLOCK) {
    float gv[NB], pv[NB], av[NB], vv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int64_t i = i0 + (int64_t)k * IGMC_BLOCK;
      const int64_t ic = i < hi ? i : hi - 1;      // clamped, not predicated: the 4 NB loads go out back to back
      gv[k] = g[ic];
      pv[k] = p[ic];
      av[k] = m1[ic];
      vv[k] = m2[ic];
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int64_t i = i0 + (int64_t)k * IGMC_BLOCK;
      if (i < hi) {
        float gi = gv[k];
        if (wd != 0.f) gi += wd * pv[k];
        const float a = beta1 * av[k] + (1.f - beta1) * gi;
        const float v = beta2 * vv[k] + (1.f - beta2) * gi * gi;
        m1[i] = a;
        m2[i] = v;
        p[i] = pv[k] - step_size * a / (sqrtf(v) * inv_sqrt_bc2 + eps);
      }
    }
  }
}

// optional optimiser tail of k_finalize (single-GPU fused step): Adam on the parameters whose gradient the
// workgroup just produced (conv layers) / on a slice of the lin parameters (extra workgroups), then the LAST
// workgroup to finish emits loss / epoch total and advances the control block.
struct AdamTail {
  int enabled;
  float* p;
  float* m1;
  float* m2;
  float step_size, inv_sqrt_bc2, beta1, beta2, eps, wd;
  int64_t* ctrl;
  int* done;
  BatchDev b;
  float ARR;
  float* loss;
  double* total;
  int use_flags;      // the step ran under edge dropout (the tick then also checks the arena's dropout stamp)
  const int* skip;    // data-parallel step: set by the gradient exchange when its sums did not arrive (StepExchange::failed);
                      // the whole launch then returns at once -- no parameter, moment or counter is touched
};

// IGMC_FIN_NB workgroups per conv layer: turn the reduced partials into the flat gradient, add the
// adjacent-rating-regulariser gradient (reference train_eval.py:167-174), emit the ARR value, and (AdamTail) update
// the parameters -- every element in ONE pass (gradient, ARR term, store, Adam), elements interleaved over the
// layer's workgroups, the small per-layer quantities (Gram matrix of the bases) recomputed by each of them.
// ARR through the 4x4 Gram matrix of the bases:  W[r] = sum_b att[r,b] basis[b]  =>
//   D[r] = W[r+1]-W[r] = sum_b d[r,b] basis[b],  d[r] = att[r+1]-att[r]
//   reg = sum_r d[r]^T Gm d[r],  dreg/dW[r] = 2(D[r-1]-D[r]) = sum_b c[r,b] basis[b],  c[r] = 2(d[r-1]-d[r])
//   d att[r,b] += ARR * sum_b' c[r,b'] Gm[b',b];   d basis[b] += ARR * sum_b' (sum_r att[r,b] c[r,b']) basis[b']
// ts_mode: every layer's gradient arrives as a relation-space table [R*fin + fin + 1][32] (graphstep.hip; rows
// r*fin + c = d W[r][c], then d root[c], then d bias) instead of basis-space partials (fin = L for layer 0, else 32).
#define IGMC_FIN_NB 8
__global__ __launch_bounds__(IGMC_BLOCK) void k_finalize(ModelDev m, const float* P, float* __restrict__ grad,
                                                           float arr_coef, AdamTail at, int ts_mode) {
  igmc_kernarg_warm<sizeof(ModelDev) + sizeof(AdamTail) + 32>();
  if (at.skip && *at.skip) return;      // (uniform over the launch: written by the kernel in front of it)
  __shared__ float smf[8];
  __shared__ float sG[16], sM[16];
  __shared__ int s_lastl;
  __shared__ float sg10[4][10];
  __shared__ int s_last;
  if (at.enabled && at.ctrl) {      // hipGraph replay: the Adam scalars live in HBM
    const double* d = (const double*)at.ctrl;
    at.step_size = (float)d[IGMC_CTRL_STEP_SIZE];
    at.inv_sqrt_bc2 = (float)d[IGMC_CTRL_INV_SQRT_BC2];
    at.beta1 = (float)d[IGMC_CTRL_BETA1];
    at.beta2 = (float)d[IGMC_CTRL_BETA2];
    at.eps = (float)d[IGMC_CTRL_EPS];
    at.wd = (float)d[IGMC_CTRL_WD];
  }
  auto finish = [&](int64_t i, float g) { grad[i] = g; };
  if (blockIdx.x >= 4 * IGMC_FIN_NB) {   // extra workgroups: Adam on lin1 / lin2 (their gradients are final already)
    const int64_t n_lin = m.n_params - m.off_l1w;
    const int nb = gridDim.x - 4 * IGMC_FIN_NB;
    const int64_t chunk = (n_lin + nb - 1) / nb;
    const int64_t lo = m.off_l1w + (int64_t)(blockIdx.x - 4 * IGMC_FIN_NB) * chunk;
    const int64_t hi = (lo + chunk < m.n_params) ? lo + chunk : m.n_params;
    adam_range<8>(at.p, grad, at.m1, at.m2, lo, hi, at.step_size, at.inv_sqrt_bc2, at.beta1, at.beta2, at.eps, at.wd);
  } else {
    const int l = blockIdx.x / IGMC_FIN_NB, part = blockIdx.x % IGMC_FIN_NB, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fin = (l == 0) ? m.L : 32;
    const int nE = fin * 32;                 // elements of one W[r] / basis[b]
    const int wgs = 32 * IGMC_KCAT + 32, na = m.R * 4, R = m.R;
    const float* basis = P + m.off_basis[l];
    const float* att = P + m.off_att[l];
    const int e0 = part * IGMC_BLOCK + tid, estep = IGMC_FIN_NB * IGMC_BLOCK;     // this thread's elements
    // ---- Gram matrix of the bases (10 unique entries), recomputed by every workgroup of the layer
    float gp[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) gp[q] = 0.f;
    for (int e = tid; e < nE; e += IGMC_BLOCK) {
      const float b0 = basis[e], b1 = basis[nE + e], b2 = basis[2 * nE + e], b3 = basis[3 * nE + e];
      gp[0] += b0 * b0; gp[1] += b0 * b1; gp[2] += b0 * b2; gp[3] += b0 * b3;
      gp[4] += b1 * b1; gp[5] += b1 * b2; gp[6] += b1 * b3;
      gp[7] += b2 * b2; gp[8] += b2 * b3; gp[9] += b3 * b3;
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) gp[q] = igmc_wave_sum_f(gp[q]);
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 10; ++q) sg10[wave][q] = gp[q];
    }
    __syncthreads();
    if (tid == 0) {
      const int ij[10][2] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}, {1, 2}, {1, 3}, {2, 2}, {2, 3}, {3, 3}};
      for (int q = 0; q < 10; ++q) {
        const float v = (sg10[0][q] + sg10[1][q]) + (sg10[2][q] + sg10[3][q]);
        sG[ij[q][0] * 4 + ij[q][1]] = v;
        sG[ij[q][1] * 4 + ij[q][0]] = v;
      }
    }
    // c[r][b] = 2 (d[r-1][b] - d[r][b]),  d[r] = att[r+1]-att[r]  (d[-1] = d[R-1] = 0)
    if (tid >= 64 && tid < 80) {           // M[b][b'] = sum_r att[r,b] c[r,b']
      const int bb = (tid - 64) >> 2, bp = tid & 3;
      float sacc = 0.f;
      for (int r = 0; r < R; ++r) {
        const float dm = (r > 0) ? att[r * 4 + bp] - att[(r - 1) * 4 + bp] : 0.f;
        const float dn = (r + 1 < R) ? att[(r + 1) * 4 + bp] - att[r * 4 + bp] : 0.f;
        sacc += att[r * 4 + bb] * 2.f * (dm - dn);
      }
      sM[tid - 64] = sacc;
    }
    __syncthreads();
    if (tid == 0 && part == 0) {           // reg = sum_r d[r]^T Gm d[r]
      float reg = 0.f;
      for (int r = 0; r + 1 < R; ++r) {
        float d[4];
        for (int q = 0; q < 4; ++q) d[q] = att[(r + 1) * 4 + q] - att[r * 4 + q];
        for (int p1 = 0; p1 < 4; ++p1)
          for (int p2 = 0; p2 < 4; ++p2) reg += d[p1] * sG[p1 * 4 + p2] * d[p2];
      }
      m.arr_part[l] = reg;
    }
    const bool table = (l == 0) || ts_mode;
    const float* raw = m.graw + (size_t)(l >= 1 ? l - 1 : 0) * wgs;                 // basis-space partial sums (l >= 1)
    const float* t0 = ts_mode ? m.ts_raw + (size_t)l * m.ts_stride : m.graw + 3 * wgs + 3 * na;
    // ---- d basis (+ ARR):  element e = (b, c, f)
    for (int e = e0; e < 4 * nE; e += estep) {
      const int bb = e / nE, cf = e % nE;
      float g;
      if (!table) {
        g = raw[(cf >> 5) * IGMC_KCAT + bb * 32 + (cf & 31)];
      } else {
        g = 0.f;
        for (int r = 0; r < R; ++r) g += att[r * 4 + bb] * t0[(size_t)r * nE + cf];
      }
      if (arr_coef != 0.f) {
        const float b0 = basis[cf], b1 = basis[nE + cf], b2 = basis[2 * nE + cf], b3 = basis[3 * nE + cf];
        g += arr_coef * (sM[bb * 4 + 0] * b0 + sM[bb * 4 + 1] * b1 + sM[bb * 4 + 2] * b2 + sM[bb * 4 + 3] * b3);
      }
      finish(m.off_basis[l] + e, g);
    }
    // ---- d root, d bias
    for (int e = e0; e < nE; e += estep)
      finish(m.off_root[l] + e, table ? t0[(size_t)R * nE + e] : raw[(e >> 5) * IGMC_KCAT + 128 + (e & 31)]);
    if (part == 0 && tid < 32)
      finish(m.off_bias[l] + tid, table ? t0[(size_t)(R * fin + fin) * 32 + tid] : raw[32 * IGMC_KCAT + tid]);
    // ---- d att (+ ARR): one wave per entry; table form: d att[r,b] = <dW[r], basis[b]>
    for (int rb = part * 4 + wave; rb < na; rb += IGMC_FIN_NB * 4) {
      const int r = rb >> 2, bb = rb & 3;
      float g;
      if (!table) {
        g = m.graw[3 * wgs + (size_t)(l - 1) * na + rb];
      } else {
        float sacc = 0.f;
        {   // nE <= 1024: at most 16 (table, basis) pairs per lane, all requested before the sum
          float tv[16], bv[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int e = lane + 64 * k;
            tv[k] = (e < nE) ? t0[(size_t)r * nE + e] : 0.f;
            bv[k] = (e < nE) ? basis[bb * nE + e] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) sacc += tv[k] * bv[k];
        }
        g = igmc_wave_sum_f(sacc);
      }
      if (arr_coef != 0.f) {
        float sacc = 0.f;
        for (int bp = 0; bp < 4; ++bp) {
          const float dm = (r > 0) ? att[r * 4 + bp] - att[(r - 1) * 4 + bp] : 0.f;
          const float dn = (r + 1 < R) ? att[(r + 1) * 4 + bp] - att[r * 4 + bp] : 0.f;
          sacc += 2.f * (dm - dn) * sG[bp * 4 + bb];
        }
        g += arr_coef * sacc;
      }
      if (lane == 0) finish(m.off_att[l] + rb, g);
    }
    if (at.enabled) {
      // Adam on this layer's parameters (basis, root, bias, att are contiguous) by the LAST of the layer's
      // workgroups to finish: by then every workgroup of the layer has read the old parameters and stored its
      // share of the gradient (no spinning; at.done[1 + l] counts the arrivals)
      __syncthreads();
      if (tid == 0) {
        __threadfence();
        s_lastl = (atomicAdd(at.done + 1 + l, 1) == IGMC_FIN_NB - 1);
      }
      __syncthreads();
      if (s_lastl) {
        __threadfence();
        adam_range<12>(at.p, grad, at.m1, at.m2, m.off_basis[l], m.off_att[l] + na, at.step_size, at.inv_sqrt_bc2,
                       at.beta1, at.beta2, at.eps, at.wd);
        if (tid == 0) at.done[1 + l] = 0;
      }
    }
  }   // conv-layer workgroups
  if (at.enabled) {
    // the last workgroup to arrive has every arr_part / err in sight: loss, epoch total, control-block tick
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = (atomicAdd(at.done, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      loss_body(at.b, m, at.ARR, at.loss, at.total, smf);
      if (threadIdx.x == 0) {
        *at.done = 0;
        if (at.ctrl) ctrl_check_and_advance(at.ctrl, at.b, at.use_flags);
      }
    }
  }
}

// Relation-space tables -> flat gradient (+ ARR) -> Adam, with NO hand-off between workgroups: k_tail_ts has already
// formed everything that needs parameters a thread does not own (Gram / ARR matrices, the ARR value, a copy of att,
// the <dW_r, basis_b> partials, the Adam scalars), so a thread reads the table rows of ITS column (c, f), produces the
// gradient of basis[0..3][c][f] and root[c][f], and updates exactly those parameters in the same pass; bias and att
// are taken by the layer's first workgroup.  Workgroups 4*IGMC_FTS_NB.. : Adam on slices of lin1 / lin2, the last one
// loss + epoch total + control-block tick (nobody else reads the control block in this launch).
// (k_finalize needs two in-kernel hand-offs for the same work -- layer-wide before Adam, grid-wide before the tick --
//  each an agent-scope fence pair + an atomic round trip.)
#define IGMC_FTS_NB 4
// The basis-space mode of this tail is correct up to 128 relations (the stash holds them) but only pays up to 32: its per-relation
// loops (ARR matrix and value in the stash role, layer 0's d att) are serial in R -- yahoo_music's 71 relations measured
// k_reduce_partials 15 -> 30 us and the gradient / Adam launch 57 -> 64 us against k_finalize (profiles/r04_experiments).
#define IGMC_FBS_MAX_R 32
// gradient g of parameter i -> flat gradient, Adam moments, parameter; returns the parameter's value after the step.
// store = false: the value only (a workgroup that needs a neighbour's updated parameter forms it itself; the owner stores)
__device__ __forceinline__ float fts_emit(float* __restrict__ grad, const AdamTail& at, int64_t i, float g, float pold,
                                          float m1old, float m2old, bool store = true) {
  if (store) grad[i] = g;
  if (!at.enabled) return pold;
  float gi = g;
  if (at.wd != 0.f) gi += at.wd * pold;
  const float a = at.beta1 * m1old + (1.f - at.beta1) * gi;
  const float v = at.beta2 * m2old + (1.f - at.beta2) * gi * gi;
  const float pnew = pold - at.step_size * a / (sqrtf(v) * at.inv_sqrt_bc2 + at.eps);
  if (store) {
    at.m1[i] = a;
    at.m2[i] = v;
    at.p[i] = pnew;
  }
  return pnew;
}

// img != 0 (with Adam): the weight images of the UPDATED parameters are written too -- what k_g2_compose would form from
// them for the next step's subgraph / dense-layer kernels (g2_image.h): a thread holds the new basis_0..3[c][f] and
// root[c][f] of its column; the layer's new att (20 values) is formed by EVERY workgroup of the layer (the owner stores it),
// so W_r[c][f] = sum_b att[r,b] basis_b[c][f] needs nothing from another workgroup.  The next step then starts with the
// subgraph kernel: one launch and one round trip to the weights less per step.
__global__ __launch_bounds__(IGMC_BLOCK) void k_finalize_ts(ModelDev m, const float* P, float* __restrict__ grad,
                                                              float arr_coef, AdamTail at, int nlin, int bs, int img) {
  __builtin_amdgcn_s_setprio(3);      // (step chain: ahead of the extraction chain's waves wherever the two share a SIMD)
  igmc_kernarg_warm<sizeof(ModelDev) + sizeof(AdamTail) + 48>();
  // bs != 0: the per-layer path's sources -- conv layers 1..3 in BASIS space (graw: d basis_b, d root, d bias straight
  // from the weight-gradient kernel, d att from the layer kernels' partials), layer 0 as its relation-space table in graw;
  // the stash comes from k_reduce_partials.  bs == 0: the relation-space tables of the subgraph kernels (ts_raw).
  if (at.skip && *at.skip) return;      // (uniform over the launch: written by the kernel in front of it)
  __shared__ float smf[8];
  const int tid = threadIdx.x;
  const float* stash = m.fin_stash;
  if (at.enabled && at.ctrl) {      // hipGraph replay: the Adam scalars of the step, stashed by k_tail_ts
    at.step_size = stash[IGMC_STASH_SCAL + 0];
    at.inv_sqrt_bc2 = stash[IGMC_STASH_SCAL + 1];
    at.beta1 = stash[IGMC_STASH_SCAL + 2];
    at.beta2 = stash[IGMC_STASH_SCAL + 3];
    at.eps = stash[IGMC_STASH_SCAL + 4];
    at.wd = stash[IGMC_STASH_SCAL + 5];
  }
  if ((int)blockIdx.x < 4 * IGMC_FTS_NB) {
    const int l = blockIdx.x / IGMC_FTS_NB, part = blockIdx.x % IGMC_FTS_NB;
    const int fin = (l == 0) ? m.L : 32, nE = fin * 32, R = m.R, na = R * 4;
    const float* stg = stash + l * IGMC_STASH_LAYER;
    __shared__ float s_st[IGMC_STASH_LAYER];      // the layer's stash: requested first, read from LDS after the barrier below
    float stq[(IGMC_STASH_LAYER + IGMC_BLOCK - 1) / IGMC_BLOCK];
#pragma unroll
    for (int u = 0; u < (IGMC_STASH_LAYER + IGMC_BLOCK - 1) / IGMC_BLOCK; ++u) {
      const int i = tid + u * IGMC_BLOCK;
      stq[u] = stg[i < IGMC_STASH_LAYER ? i : IGMC_STASH_LAYER - 1];
    }
    const float* st = s_st;
    const int wgs = 32 * IGMC_KCAT + 32;
    const bool table = !bs || l == 0;
    const float* t0 = bs ? m.graw + 3 * wgs + 3 * na : m.ts_raw + (size_t)l * m.ts_stride;
    const float* raw = m.graw + (size_t)(l >= 1 ? l - 1 : 0) * wgs;
    // basis-space mode, layer 0: d att[r,b] = <dW_r, basis_b> needs every basis element of the layer BEFORE its owner
    // (a thread of this very workgroup: nE <= 256) updates it -> formed first, then a barrier
    __shared__ float s_gatt0[512];
    if (bs && l == 0) {
      if (part == 0) {
        // a thread owns element e = tid (strided by the workgroup for fin > 8) of every product: the R table values and
        // the 4 basis values of e requested together, the na = 4 R wave sums taken side by side (a round trip and six
        // crossbar steps in all; entry after entry it was 5 dependent rounds of both), waves combined in fixed order
        __shared__ float s_gpart[IGMC_BLOCK / 64][32];
        const int lane = tid & 63, wave = tid >> 6;
        if (R <= 8) {
          float acc[32];
#pragma unroll
          for (int i = 0; i < 32; ++i) acc[i] = 0.f;
          for (int e0 = 0; e0 < nE; e0 += IGMC_BLOCK) {
            const int e = e0 + tid, ec = e < nE ? e : nE - 1;
            float tv[8], bq[4];
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = t0[(size_t)(r < R ? r : R - 1) * nE + ec];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) bq[bb] = P[m.off_basis[0] + (int64_t)bb * nE + ec];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
              for (int bb = 0; bb < 4; ++bb)
                if (r < R && e < nE) acc[r * 4 + bb] += tv[r] * bq[bb];
          }
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) {
            float t[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) t[i] = __shfl_xor(acc[i], d, 64);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] += t[i];
          }
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) s_gpart[wave][i] = acc[i];
          }
          __syncthreads();
          if (tid < na) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < IGMC_BLOCK / 64; ++w) sacc += s_gpart[w][tid];
            s_gatt0[tid] = sacc;
          }
        } else {                  // more than 8 relations: one wave per entry, lanes over the nE elements
          for (int rb = wave; rb < na; rb += IGMC_BLOCK / 64) {
            const int r = rb >> 2, bb = rb & 3;
            float sacc = 0.f;
            for (int e = lane; e < nE; e += 64) sacc += t0[(size_t)r * nE + e] * P[m.off_basis[0] + (int64_t)bb * nE + e];
            sacc = igmc_wave_sum_f(sacc);
            if (lane == 0) s_gatt0[rb] = sacc;
          }
        }
      }
      __syncthreads();
    }
    __shared__ float s_attn[64];                // the layer's att after the step (img)
    float pn[5] = {0.f, 0.f, 0.f, 0.f, 0.f};    // this thread's basis_0..3[c][f], root[c][f] after the step (img)
    int e_img = -1;
    const bool emit = img && at.enabled && m.g2_w && R <= G2_NR * G2_NG_MAX && fin <= 32;
    const int ng = g2_groups(R, m.L);
    // ---- every load of the workgroup is requested here, before the first store: the thread's column of the main pass
    // (nE <= 1024: ONE element a thread), the bias role's four values, the att role's first entry (its fin partial sums or
    // the basis-space gradient, the parameter and its moments).  The stores of the main pass used to come first: the two
    // roles then paid a second round trip, and the weight images wait for the att role.
    const int e = part * IGMC_BLOCK + tid;
    const bool has = e < nE;
    int64_t idx[5];
    float pv[5], m1v[5], m2v[5], g[5], tv[8];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      idx[q] = (q < 4) ? m.off_basis[l] + (int64_t)q * nE + (has ? e : 0) : m.off_root[l] + (has ? e : 0);
      pv[q] = has ? P[idx[q]] : 0.f;
      m1v[q] = (has && at.enabled) ? at.m1[idx[q]] : 0.f;
      m2v[q] = (has && at.enabled) ? at.m2[idx[q]] : 0.f;
      g[q] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) tv[r] = 0.f;
    if (has) {
      if (table) {
#pragma unroll
        for (int r = 0; r < 8; ++r) tv[r] = (r < R) ? t0[(size_t)r * nE + e] : 0.f;
        g[4] = t0[(size_t)R * nE + e];
      } else {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) g[bb] = raw[(e >> 5) * IGMC_KCAT + bb * 32 + (e & 31)];
        g[4] = raw[(e >> 5) * IGMC_KCAT + 128 + (e & 31)];
      }
    }
    const bool roles = part == 0 || emit;
    const bool bias_role = roles && tid < 32 && part == 0;
    const bool att_role = roles && !bias_role && tid >= 64 && tid - 64 < na;
    float bq[4] = {0.f, 0.f, 0.f, 0.f};          // bias role: gradient source, parameter, moments
    if (bias_role) {
      const int64_t i = m.off_bias[l] + tid;
      bq[0] = table ? t0[(size_t)(R * fin + fin) * 32 + tid] : raw[32 * IGMC_KCAT + tid];
      bq[1] = P[i];
      bq[2] = at.enabled ? at.m1[i] : 0.f;
      bq[3] = at.enabled ? at.m2[i] : 0.f;
    }
    float dpre[32], am1 = 0.f, am2 = 0.f, ag = 0.f;      // att role, entry tid - 64
#pragma unroll
    for (int k = 0; k < 32; ++k) dpre[k] = 0.f;
    if (att_role) {
      const int rb = tid - 64, r = rb >> 2, bb = rb & 3;
      const int64_t i = m.off_att[l] + rb;
      if (at.enabled && !emit) {
        am1 = at.m1[i];
        am2 = at.m2[i];
      }
      if (bs) {
        if (l > 0) ag = m.graw[3 * wgs + (size_t)(l - 1) * na + rb];
      } else {
        const float* dp = m.datt_part + ((size_t)l * m.ts_stride + (size_t)r * nE) / 32 * 4 + bb;
#pragma unroll
        for (int k = 0; k < 32; ++k) dpre[k] = (k < fin) ? dp[(size_t)k * 4] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < (IGMC_STASH_LAYER + IGMC_BLOCK - 1) / IGMC_BLOCK; ++u) {
      const int i = tid + u * IGMC_BLOCK;
      if (i < IGMC_STASH_LAYER) s_st[i] = stq[u];
    }
    __syncthreads();
    // (the bias / att roles FIRST: the weight images wait for the layer's new att, formed by wave 1 -- behind its share of the
    //  main pass it kept every other wave of the workgroup waiting at the barrier below for 2.6 k cycles)
    if (roles) {
      if (bias_role) {                             // d bias
        const int64_t i = m.off_bias[l] + tid;
        const float bnew = fts_emit(grad, at, i, bq[0], bq[1], bq[2], bq[3]);
        if (emit && l == 0) m.g2_w[g2_t0_off(ng) + (R * fin + fin) * 32 + tid] = bnew;       // layer-0 table: bias row
      } else if (tid >= 64) {                      // d att[r,b] = <dW_r, basis_b> (+ ARR): fin partials, fixed order
       for (int rb = tid - 64; rb < na; rb += IGMC_BLOCK - 64) {      // (one entry a thread up to 48 relations)
        const int r = rb >> 2, bb = rb & 3;
        const bool first = rb == tid - 64;         // (its operands were requested above)
        const int64_t i = m.off_att[l] + rb;
        const float pold = st[IGMC_STASH_ATT + rb];
        // (img: the moments before the step come from the stash -- the owner workgroup updates them in place meanwhile)
        const float m1o = !at.enabled ? 0.f : emit ? st[IGMC_STASH_ATTM1 + rb] : first ? am1 : at.m1[i];
        const float m2o = !at.enabled ? 0.f : emit ? st[IGMC_STASH_ATTM2 + rb] : first ? am2 : at.m2[i];
        float g = 0.f;
        if (bs) {
          g = (l == 0) ? s_gatt0[rb] : first ? ag : m.graw[3 * wgs + (size_t)(l - 1) * na + rb];
        } else if (first && fin <= 32) {
#pragma unroll
          for (int k = 0; k < 32; ++k) g += dpre[k];
        } else {
          const float* dp = m.datt_part + ((size_t)l * m.ts_stride + (size_t)r * nE) / 32 * 4 + bb;
          for (int k0 = 0; k0 < fin; k0 += 32) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = (k0 + k < fin) ? dp[(size_t)(k0 + k) * 4] : 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) g += v[k];
          }
        }
        if (arr_coef != 0.f) {
          float sacc = 0.f;
          for (int bp = 0; bp < 4; ++bp) {
            const float a0 = st[IGMC_STASH_ATT + r * 4 + bp];
            const float dm = (r > 0) ? a0 - st[IGMC_STASH_ATT + (r - 1) * 4 + bp] : 0.f;
            const float dn = (r + 1 < R) ? st[IGMC_STASH_ATT + (r + 1) * 4 + bp] - a0 : 0.f;
            sacc += 2.f * (dm - dn) * st[IGMC_STASH_G + bp * 4 + bb];
          }
          g += arr_coef * sacc;
        }
        const float anew = fts_emit(grad, at, i, g, pold, m1o, m2o, part == 0);
        if (emit) s_attn[rb] = anew;
       }
      }
    }
    if (has) {       // one round for fin <= 32
      if (table) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r < R) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) g[bb] += st[IGMC_STASH_ATT + r * 4 + bb] * tv[r];
          }
        for (int r = 8; r < R; ++r) {
          const float tvr = t0[(size_t)r * nE + e];
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) g[bb] += st[IGMC_STASH_ATT + r * 4 + bb] * tvr;
        }
      }
      if (arr_coef != 0.f) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
          g[bb] += arr_coef * (st[IGMC_STASH_M + bb * 4 + 0] * pv[0] + st[IGMC_STASH_M + bb * 4 + 1] * pv[1] +
                               st[IGMC_STASH_M + bb * 4 + 2] * pv[2] + st[IGMC_STASH_M + bb * 4 + 3] * pv[3]);
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) pn[q] = fts_emit(grad, at, idx[q], g[q], pv[q], m1v[q], m2v[q]);
      e_img = e;
    }
    if (emit) {
      __syncthreads();
      if (e_img >= 0) {
        const int c = e_img >> 5, f = e_img & 31;
        if (l == 0) {       // layer-0 table [W_0[r * L + c] | root_0[c] | bias_0] (rows past R L + L stay zero: first compose)
          float* t0w = m.g2_w + g2_t0_off(ng);
          for (int r = 0; r < R; ++r)
            t0w[(r * fin + c) * 32 + f] = g2_wsum(s_attn[r * 4], s_attn[r * 4 + 1], s_attn[r * 4 + 2], s_attn[r * 4 + 3],
                                                  pn[0], pn[1], pn[2], pn[3]);
          t0w[(R * fin + c) * 32 + f] = pn[4];
        } else {            // forward image: element (k = c, n = f); transposed image: element (k = f, n = c)
          // relation r -> block r % G2_NR of group r / G2_NR; root -> block G2_NR of group 0
#pragma unroll 3
          for (int r = 0; r <= R; ++r) {           // (blocks of relations the model does not have stay zero: first compose)
            const float v = (r == R) ? pn[4]
                          : g2_wsum(s_attn[r * 4], s_attn[r * 4 + 1], s_attn[r * 4 + 2], s_attn[r * 4 + 3], pn[0], pn[1], pn[2], pn[3]);
            const int grp = (r == R) ? 0 : r / G2_NR, blk = (r == R) ? G2_NR : r % G2_NR;
            uint16_t* imf = (uint16_t*)(m.g2_w + g2_img_off(ng, l, 0, grp));
            uint16_t* imt = (uint16_t*)(m.g2_w + g2_img_off(ng, l, 1, grp));
            uint32_t h, mi, lo;
            g2_split2(v, 0.f, h, mi, lo);
            const uint32_t t3[3] = {h, mi, lo};
#pragma unroll
            for (int t = 0; t < G2_NT; ++t) {
              imf[g2_img_index(t, blk, c, f)] = (uint16_t)t3[t];
              imt[g2_img_index(t, blk, f, c)] = (uint16_t)t3[t];
            }
          }
        }
      }
    }
  } else if ((int)blockIdx.x < 4 * IGMC_FTS_NB + nlin) {      // Adam on lin1 / lin2 (their gradients are final already)
    const int64_t n_lin = m.n_params - m.off_l1w;
    const int64_t chunk = (n_lin + nlin - 1) / nlin;
    const int64_t lo = m.off_l1w + (int64_t)(blockIdx.x - 4 * IGMC_FTS_NB) * chunk;
    const int64_t hi = (lo + chunk < m.n_params) ? lo + chunk : m.n_params;
    adam_range<8>(at.p, grad, at.m1, at.m2, lo, hi, at.step_size, at.inv_sqrt_bc2, at.beta1, at.beta2, at.eps, at.wd);
  } else {                                                     // loss, epoch total, control-block tick
    loss_body(at.b, m, at.ARR, at.loss, at.total, smf);
    if (tid == 0 && at.ctrl) ctrl_check_and_advance(at.ctrl, at.b, at.use_flags);
  }
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_loss(BatchDev b, ModelDev m, float ARR, float* __restrict__ loss) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  __shared__ float smf[8];
  loss_body(b, m, ARR, loss, nullptr, smf);
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_sse_acc(BatchDev b, const float* __restrict__ out, double* acc, int64_t* ctrl) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  __shared__ float smf[8];
  const int B = b.totals[3];
  float s = 0.f;
  for (int g = threadIdx.x; g < B; g += IGMC_BLOCK) {
    const float d = out[g] - b.y[g];
    s += d * d;
  }
  s = igmc_block_sum_f(s, smf);
  if (threadIdx.x == 0) {
    acc[0] += (double)s;
    acc[1] += (double)B;
    if (ctrl) ctrl_advance(ctrl);      // the evaluation step's tick in the same launch (igmc_sse_accumulate_tick)
  }
}

// =================================================================== fused Adam (flat buffer) + step epilogue
// With `fin` set, block 0 also produces the step's loss / epoch total, and the LAST block to finish advances
// the device-side step control (igmc_hip.h) for the next replay of the step graph.
__global__ void k_tick(int64_t* ctrl) {
  if (threadIdx.x == 0 && blockIdx.x == 0) ctrl_advance(ctrl);
}
// a new group of M steps starts at the current position (igmc_ctrl_regroup)
__global__ void k_regroup(int64_t* ctrl, int M, int64_t first_cur, int64_t first_next) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ctrl[IGMC_CTRL_GROUP] = M > 0 ? M : 1;
  ctrl[IGMC_CTRL_GK] = 0;
  ctrl[IGMC_CTRL_GQ] = 0;
  ctrl[IGMC_CTRL_FIRST] = first_cur;
  ctrl[IGMC_CTRL_FIRST_ODD] = first_next;
}

// pacing gate of the extraction chain (igmc_ctrl_gate): one wave that returns once `gk_min` steps of the group of parity q
// are done -- or that group is over, or `timeout_ticks` of the 100 MHz wall clock have passed -- and then, if it had to wait
// for that step (or `delay_always`), `delay_ticks` later: the step that has just begun gets its workgroups onto the chip first.
// The kernels queued behind the gate on its stream start then.  A HINT, never a dependency: whatever follows the gate is
// correct at any time (the extraction of the NEXT group touches nothing the running group reads), so a gate that gives up
// only costs the pacing.
__global__ void k_step_gate(int64_t* ctrl, int q, int gk_min, long long delay_ticks, int delay_always,
                            long long timeout_ticks) {
#ifndef IGMC_HIPEMU
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long t0 = wall_clock64();
  bool waited = false;
  for (;;) {
    if ((igmc_ctrl_ld(ctrl + IGMC_CTRL_GQ) & 1) != (int64_t)(q & 1)) return;
    if (igmc_ctrl_ld(ctrl + IGMC_CTRL_GK) >= (int64_t)gk_min) break;
    if (wall_clock64() - t0 > timeout_ticks) {      // (counted: a caller whose gates keep giving up paces by edges instead)
      atomicAdd((unsigned long long*)(ctrl + IGMC_CTRL_GATE_TIMEOUTS), 1ull);
      return;
    }
    waited = true;
    __builtin_amdgcn_s_sleep(16);
  }
  if (waited || delay_always) {
    const long long t1 = wall_clock64();
    while (wall_clock64() - t1 < delay_ticks) __builtin_amdgcn_s_sleep(8);
  }
#endif
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m1, float* __restrict__ m2, int64_t n,
                                                       float step_size, float inv_sqrt_bc2, float beta1, float beta2,
                                                       float eps, float wd, int64_t* ctrl, int tick, FinishArgs fin) {
  __shared__ float smf[8];
  if (ctrl) {      // hipGraph replay: the scalars live in HBM
    const double* d = (const double*)ctrl;
    step_size = (float)d[IGMC_CTRL_STEP_SIZE];
    inv_sqrt_bc2 = (float)d[IGMC_CTRL_INV_SQRT_BC2];
    beta1 = (float)d[IGMC_CTRL_BETA1];
    beta2 = (float)d[IGMC_CTRL_BETA2];
    eps = (float)d[IGMC_CTRL_EPS];
    wd = (float)d[IGMC_CTRL_WD];
  }
  {   // a contiguous slice per workgroup, eight elements a thread and round with all their loads in flight (adam_range)
    const int64_t chunk = ((n + gridDim.x - 1) / gridDim.x + IGMC_BLOCK - 1) / IGMC_BLOCK * IGMC_BLOCK;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    if (lo < n) adam_range<8>(p, g, m1, m2, lo, hi, step_size, inv_sqrt_bc2, beta1, beta2, eps, wd);
  }
  if (fin.enabled && blockIdx.x == 0) loss_body(fin.b, fin.m, fin.ARR, fin.loss, fin.total, smf);
  if (ctrl && tick) {
    // every block has read its scalars above; the last one to get here advances the control block
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      int* done = (int*)&ctrl[IGMC_CTRL_DONE];
      if (atomicAdd(done, 1) == (int)gridDim.x - 1) {
        *done = 0;
        if (fin.enabled) ctrl_check_and_advance(ctrl, fin.b, fin.use_flags);
        else ctrl_advance(ctrl);
      }
    }
  }
}

// =================================================================== host launch sequences
#include "launch.h"

// grid for row-parallel kernels: enough workgroups for the capacity, capped, and a multiple of 8 (>= 8) so
// that every XCD residue owns workgroups (XCD-affine segments, see igmc_xcd_segment)
static inline int igmc_rows_grid(int cap_rows, int rows_per_block, int max_blocks) {
  int g = (cap_rows + rows_per_block - 1) / rows_per_block;
  if (g > max_blocks) g = max_blocks;
  g = (g + 7) & ~7;
  return g < 8 ? 8 : g;
}
// grid for the XCD-segmented kernels: 8 x (workgroups needed by the LARGEST segment = ceil(B/8) graphs of at
// most `slot` rows each), so that no XCD has to take a second round
static inline int igmc_xcd_grid(const ModelDev& m, int B, int rows_per_block, int max_blocks) {
  const int slot = (m.node_cap + m.graph_cap - 1) / m.graph_cap;
  const long rows = (long)((B + 7) / 8) * slot;
  long per = (rows + rows_per_block - 1) / rows_per_block;
  if (per < 1) per = 1;
  long g = 8 * per;
  if (g > max_blocks) g = max_blocks & ~7;
  return (int)(g < 8 ? 8 : g);
}

// grid of k_l0_fwd: every workgroup composes the layer-0 table W0[r * L + c] (R L x 32 values, 4 fmas + 5 loads each) before
// it walks rows; with many relations (yahoo_music: 284 table rows) thousands of 4-row workgroups spend their time on that
// -- there a workgroup takes 8 rows per table row of work it has to amortise
static inline int igmc_l0_grid(const ModelDev& m, int B) {
  int rows = 4;
  if (m.R * m.L > 32) rows = (m.R * m.L) / 8;
  return igmc_xcd_grid(m, B, rows, IGMC_GATHER_BLOCKS);
}

void igmc_launch_forward(const ModelDev& m, const ModelAux& ax, const BatchDev& b, const float* P, int B, int training,
                         int use_flags, const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult,
                         float* out, void* stream) {
  {
    G2Layout lay2;
    int cs2 = 1;
    if (!training && m.R * m.L + m.L + 1 <= 32 && igmc_g2_eligible(m, b, B, &lay2, &cs2)) {
      igmc_launch_graph_step2(m, b, P, B, 0, use_flags, lay2, cs2, nullptr, seed, step, mult, 0.f, out, stream);
      return;
    }
  }
  igmc_launch_conv_forward(m, b, P, B, training, use_flags, stream);
  const int hgrid = (B + IGMC_HG - 1) / IGMC_HG;
  const size_t fs = (size_t)IGMC_HG * m.D * sizeof(float);
  if (m.D % 16 == 0)
    IGMC_PLAUNCH("k_head_fwd", k_head_fwd_mfma, (B + 15) / 16, 512, 0, stream, b, m, P, training, inj_mask, seed, step,
                 mult, out);
  else if (fs <= 48 * 1024)
    IGMC_PLAUNCH("k_head_fwd", (k_head_fwd<true>), hgrid, 512, fs, stream, b, m, P, training, inj_mask, seed, step,
                 mult, out);
  else
    IGMC_PLAUNCH("k_head_fwd", (k_head_fwd<false>), hgrid, 512, 0, stream, b, m, P, training, inj_mask, seed, step,
                 mult, out);
  (void)ax;
}

static inline int igmc_fin_mode() {
  const char* fe = getenv("IGMC_FIN_MODE");        // 0: the hand-off version of the gradient / Adam tail (k_finalize); read
  return fe ? atoi(fe) : 1;                        // on every call: tests switch it per case
}

// 1 = the conv backward of a dense readout gradient (sort-pool family) takes the one-launch form with relation-space tables
// (igmc_launch_conv_backward); the forward then need not leave the Y products behind
static int igmc_conv_bwd_tables(const ModelDev& m, const BatchDev& b, int B) {
  const int fts = igmc_fin_mode() && m.fin_stash && m.datt_part && m.R <= 8;
  return fts && m.R * m.L + m.L + 1 <= 32 && m.dcat[0] && igmc_dl_eligible(m, b, B) && igmc_dl_fwd_eligible(m, b, B) &&
         igmc_dl_bwd_eligible(m, b, B);
}

// The four conv layers alone (h_0..h_3 left in HBM): the per-layer kernels, whatever the readout that follows
// (centre-node readout of IGMC: igmc_launch_forward; sort-pool readout of DGCNN_RS: sortpool.hip)
void igmc_launch_conv_forward(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                              void* stream) {
  const size_t l0s = (size_t)(m.R * m.L * 32 + m.L * 32 + 32) * sizeof(float) + (size_t)4 * m.R * m.L * sizeof(int);
  const int g16 = igmc_xcd_grid(m, B, 4, IGMC_GATHER_BLOCKS);   // one wave per row, 4 rows per block
  const int g64 = igmc_rows_grid(m.node_cap, 64, 512);
  // slots of 129..256 nodes a side with a dense block: every conv layer on the matrix cores (graphstep2.hip, k_dl_layer0 /
  // k_dl_layer) from the blocks alone -- no edge list is read
  // (more than five relations: the one-launch forward takes them in groups, igmc_dl_wide)
  const int wide = igmc_dl_wide(m, b, B);
  const int dl = wide || igmc_dl_eligible(m, b, B);
  const int dlf = wide || (dl && igmc_dl_fwd_eligible(m, b, B));
  if (dlf) {                                              // all four layers in ONE launch (k_dl_fwd); nothing follows that
    igmc_launch_g2_compose(m, P, stream);                 // would advance its exchange tags: its last workgroup does
    igmc_launch_dl_fwd(m, b, P, B, training, use_flags, training ? m.dpre[3] : nullptr, 1, stream);
  } else if (dl) {
    igmc_launch_g2_compose(m, P, stream);                 // the step's weight images + layer-0 table
    igmc_launch_dl_layer0(m, b, B, training, use_flags, stream);
  } else if (training) {
    if (use_flags) IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<true, true>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, P, m.h[0]);
    else IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<false, true>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, P, m.h[0]);
  } else {
    if (use_flags) IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<true, false>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, P, m.h[0]);
    else IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<false, false>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, P, m.h[0]);
  }
  const size_t ysz = (size_t)32 * (128 + 4) * sizeof(float);
  const int gt = igmc_xcd_grid(m, B, 16, 2048);                        // fused layer: 16 rows per workgroup
  const size_t fsm4 = (size_t)(16 * IGMC_TP + 1024 + m.R * 4) * sizeof(float);
  for (int l = 1; l < 4 && !dlf; ++l) {
    // the top layer's launch also clears dPre_3 (only its target rows are written by the head backward)
    float* zo = (training && l == 3) ? m.dpre[3] : nullptr;
    if (dl) {
      igmc_launch_dl_layer(m, b, P, B, l, 0, use_flags, zo, stream);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_rgcn_layer_fwd", (k_rgcn_layer4<true, false>), gt, IGMC_BLOCK, fsm4, stream, b, m, P, l, zo);
      else IGMC_PLAUNCH("k_rgcn_layer_fwd", (k_rgcn_layer4<false, false>), gt, IGMC_BLOCK, fsm4, stream, b, m, P, l, zo);
    }
  }
  // training: the products Y_l = h_{l-1} @ [basis_0 | .. | basis_3] the backward's att gradient reads
  if (training && !igmc_conv_bwd_tables(m, b, B)) IGMC_PLAUNCH("k_dense_y_all", k_dense_y_all, dim3(g64, 3), IGMC_BLOCK, ysz, stream, b, m, P);
}

void igmc_launch_backward(const ModelDev& m, const ModelAux& ax, const BatchDev& b, const float* P, int B, int use_flags,
                          const float* gout, int from_err, float grad_scale, float mult, float drop_scale,
                          float arr_coef, float* grad, void* stream) {
  const int g16 = igmc_xcd_grid(m, B, 4, IGMC_GATHER_BLOCKS);
  const int g64 = igmc_rows_grid(m.node_cap, 64, 512);
  const int hgrid = (B + IGMC_HG - 1) / IGMC_HG;
  const int na = m.R * 4;
  const int rows0 = m.R * m.L + m.L + 1;
  const int l0_mfma = rows0 <= 32;       // layer-0 table gradient rides in the MFMA weight-gradient kernel
  void* s2 = stream;      // single stream: see the note at k_wgrad
  (void)ax;
  (void)g16; (void)g64; (void)na; (void)l0_mfma;
  if (m.D % 16 == 0)
    IGMC_PLAUNCH("k_head_bwd_a", k_head_bwd_a_mfma, dim3((B + 15) / 16, (m.D / 16 + 7) / 8), 512, 0, stream, b, m, P,
                 gout, from_err, grad_scale, mult, drop_scale, m.dpre[3]);
  else
    IGMC_PLAUNCH("k_head_bwd_a", k_head_bwd_a, hgrid, 1024, 0, stream, b, m, P, gout, from_err, grad_scale, mult,
                 drop_scale, m.dpre[3]);
  if (m.D % 16 == 0)
    IGMC_PLAUNCH("k_head_bwd_w", k_head_bwd_w_mfma, dim3(8, (m.D / 16 + 3) / 4), IGMC_BLOCK, 0, s2, b, m, P, gout,
                 from_err, grad_scale, mult, drop_scale, grad);
  else
    IGMC_PLAUNCH("k_head_bwd_w", k_head_bwd_w, dim3(17, (m.D + 255) / 256), IGMC_BLOCK, 0, s2, b, m, P, gout, from_err,
                 grad_scale, mult, drop_scale, grad);
  igmc_launch_conv_backward(m, b, P, B, use_flags, arr_coef, grad, stream);
}

// Backward of the four conv layers from dPre_3 (in m.dpre[3]) and the readout's gradient w.r.t. h_0..h_2 (gfeat on the
// target rows, or the dense m.dcat[l]): dPre_2..0, weight-gradient partials, their reduction, gradient (+ ARR) of the conv
// parameters into `grad` (k_finalize without Adam).
void igmc_launch_conv_backward(const ModelDev& m, const BatchDev& b, const float* P, int B, int use_flags, float arr_coef,
                               float* grad, void* stream) {
  const int g16 = igmc_xcd_grid(m, B, 4, IGMC_GATHER_BLOCKS);
  const int g64 = igmc_rows_grid(m.node_cap, 64, 512);
  const int na = m.R * 4;
  const int rows0 = m.R * m.L + m.L + 1;
  const int l0_mfma = rows0 <= 32;
  const int gt = igmc_xcd_grid(m, B, 16, 2048);
  const size_t bsm4 = (size_t)(16 * IGMC_TP + 1024 + m.R * 4 + 16 * m.R * 4) * sizeof(float);
  const int dl = igmc_dl_eligible(m, b, B);     // (the images of this step were composed by the forward)
  // Dense readout gradient (sort-pool family) on the one-launch backward of the dense layers: dPre_3 of every row from
  // m.dpre[3], the readout gradient of layers 0..2 added per row from m.dcat, relation-space tables -> k_tail_ts (no lin1 / lin2
  // role) -> k_finalize_ts: three launches instead of the three layer passes, the Y products' consumers (k_wgrad), the
  // partials' reduction and k_finalize.  (IGMC_DL_TS=0 / IGMC_DL_FUSED=0|1: the per-layer form below.)
  {
    if (igmc_conv_bwd_tables(m, b, B)) {
      igmc_launch_dl_bwd(m, b, B, use_flags, stream, nullptr, 1);
      const int gstride = (B + 7) & ~7, gg = igmc_dl_grid(b, B) / B * gstride;
      IGMC_PLAUNCH("k_tail_ts", k_tail_ts, (4 * m.ts_stride + 63) / 64 + 4, IGMC_BLOCK, 0, stream, b, m, P, 0.f, 1.f, 2.f, grad,
                   0, gg, gstride, B, 4, (const int64_t*)nullptr, 1);
      AdamTail none;
      memset(&none, 0, sizeof(none));
      IGMC_PLAUNCH("k_finalize", k_finalize_ts, 4 * IGMC_FTS_NB, IGMC_BLOCK, 0, stream, m, P, grad, arr_coef, none, 0, 0, 0);
      return;
    }
  }
  for (int l = 3; l >= 1; --l) {
    // transposed gather of dPre_l (+ d att partials), then [G | dPre_l] @ [basis^T ; root^T] + backward epilogue
    if (dl) {
      igmc_launch_dl_layer(m, b, P, B, l, 1, use_flags, nullptr, stream);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_rgcn_layer_bwd", (k_rgcn_layer4<true, true>), gt, IGMC_BLOCK, bsm4, stream, b, m, P, l, (float*)nullptr);
      else IGMC_PLAUNCH("k_rgcn_layer_bwd", (k_rgcn_layer4<false, true>), gt, IGMC_BLOCK, bsm4, stream, b, m, P, l, (float*)nullptr);
    }
  }
  const float* d0 = m.dpre[0];
  IGMC_PLAUNCH("k_wgrad", k_wgrad, dim3(IGMC_WG_BLOCKS, l0_mfma ? 4 : 3), IGMC_BLOCK, 0, stream, b, m, 0);
  if (l0_mfma) {
  } else if (rows0 <= 64) IGMC_PLAUNCH("k_l0_bwd", (k_l0_bwd<8>), IGMC_L0_BLOCKS, IGMC_BLOCK, 0, stream, b, m, d0, m.l0_part);
  else IGMC_PLAUNCH("k_l0_bwd", (k_l0_bwd<40>), IGMC_L0_BLOCKS, IGMC_BLOCK, 0, stream, b, m, d0, m.l0_part);
  {
    const int wgs2 = igmc_wg_stride(), n0 = rows0 * 32;
    const int nblk = ((l0_mfma ? 4 : 3) * wgs2 + 63) / 64 + (l0_mfma ? 0 : (n0 + 63) / 64) + (3 * na + 3) / 4;
    IGMC_PLAUNCH("k_reduce_partials", k_reduce_partials, nblk, IGMC_BLOCK, 0, stream, m,
                 dl ? igmc_dl_grid(b, B) : gt, l0_mfma,
                 IGMC_WG_BLOCKS, (const float*)nullptr, (const int64_t*)nullptr, 0);
  }
  {
    AdamTail none;
    memset(&none, 0, sizeof(none));
    IGMC_PLAUNCH("k_finalize", k_finalize, 4 * IGMC_FIN_NB, IGMC_BLOCK, 0, stream, m, P, grad, arr_coef, none, 0);
  }
}

// Fused-step sequence (loss + gradients [+ Adam]) with the multi-role launches:
//   l0_fwd, 3 x layer_fwd, {head fwd+bwd | Y}, 3 x layer_bwd, {weight grads | lin grads}, reduce, finalize[+Adam]

int igmc_step_exchange_inside(const ModelDev& m, const BatchDev& b, int B) {
  const int ny = (m.D / 16 + 3) / 4;
  if (!((m.D % 16 == 0) && 8 * ny <= IGMC_WG_BLOCKS)) return 0;          // generic sequence: flat gradient only
  G2Layout lay2;
  int cs2 = 1;
  if (m.R * m.L + m.L + 1 <= 32 && igmc_g2_eligible(m, b, B, &lay2, &cs2))
    return igmc_fin_mode() && m.fin_stash && m.datt_part && m.R <= 8;
  return igmc_fin_mode() && m.fin_stash && m.R <= IGMC_FBS_MAX_R;
}

int igmc_launch_loss_grad(const ModelDev& m_in, const ModelAux& ax, const BatchDev& b, float* P, int B, int use_flags,
                          const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult, float ARR,
                          float grad_scale, float arr_scale, float* out, float* grad, float* loss, const AdamTail* adam,
                          void* stream, const StepExchange* xch, int* img_emitted) {
  ModelDev m = m_in;
  m.adam_m1 = adam ? adam->m1 : nullptr;       // (the stash roles keep att's moments of before the step)
  m.adam_m2 = adam ? adam->m2 : nullptr;
  if (img_emitted) *img_emitted = 0;
  const int rows0 = m.R * m.L + m.L + 1;
  // the gradient / Adam kernel also leaves the weight images of the updated parameters
  const int img = adam && m.g2_w && m.R <= G2_NR * G2_NG_MAX && rows0 <= (g2_t0_rows(m.R, m.L) == 32 ? 32 : 48);
  const int l0_mfma = rows0 <= 32;
  const int gy = igmc_rows_grid(m.node_cap, 128, 512);
  const int hb = (B + 15) / 16;
  const int ny = (m.D / 16 + 3) / 4;
  const bool fast_head = (m.D % 16 == 0) && 8 * ny <= IGMC_WG_BLOCKS;
  const int64_t n_lin = m.n_params - m.off_l1w;      // lin1 / lin2 are the tail of the flat parameter vector
  AdamTail at;
  memset(&at, 0, sizeof(at));
  if (adam) at = *adam;
  at.use_flags = use_flags;
  at.skip = xch ? xch->failed : nullptr;
  if (!fast_head) {      // generic sequence
    igmc_launch_forward(m, ax, b, P, B, 1, use_flags, inj_mask, seed, step, mult, out, stream);
    igmc_launch_backward(m, ax, b, P, B, use_flags, nullptr, 1, grad_scale, mult, 2.f, ARR * arr_scale, grad, stream);
    if (adam) {
      igmc_launch_finish(m, b, at.p, grad, at.m1, at.m2, at.step_size, at.inv_sqrt_bc2, at.beta1, at.beta2, at.eps, at.wd,
                         at.ctrl, ARR, at.loss, at.total, use_flags, stream);
    } else if (loss) {
      igmc_launch_loss(m, b, ARR, loss, stream);
    }
    return 0;
  }
  G2Layout lay2;
  int cs2 = 1;
  if (l0_mfma && igmc_g2_eligible(m, b, B, &lay2, &cs2)) {
    // one workgroup (cluster) per subgraph: forward, residual and backward down to the per-workgroup gradient partials
    const int cs = igmc_gs_cluster(B);
    const int gstride = (cs > 1) ? ((B + 7) & ~7) : IGMC_TS_BLOCKS;
    const int gg = (cs > 1) ? cs * gstride : igmc_gs_grid(B);
    int bump_seq = 0;       // 1: k_tail_ts advances the launch sequence number of the subgraph kernel's exchange tags
    bump_seq = igmc_launch_graph_step2(m, b, P, B, 1, use_flags, lay2, cs2, inj_mask, seed, step, mult, grad_scale, out, stream);
    // IGMC_FIN_MODE=0: the hand-off version of the gradient / Adam tail (k_finalize) instead of k_finalize_ts
    const int fts = igmc_fin_mode() && m.fin_stash && m.datt_part && m.R <= 8;
    IGMC_PLAUNCH("k_tail_ts", k_tail_ts, 8 * ny + (4 * m.ts_stride + 63) / 64 + (fts ? 4 : 0), IGMC_BLOCK, 0, stream, b, m,
                 (const float*)P, grad_scale, mult, 2.f, grad, 8 * ny, gg, gstride, B, fts ? 4 : 0,
                 (const int64_t*)(adam ? at.ctrl : nullptr), bump_seq);
    if (xch && fts) {      // tables + d att partials (one allocation) and the lin gradients, summed over the ranks
      const int rc = xch->sum(xch->user, m.ts_raw, (int64_t)4 * m.ts_stride + (int64_t)4 * m.ts_stride / 32 * 4,
                              grad + m.off_l1w, n_lin, stream);
      if (rc) return rc;
    }
    if (adam) {
      at.enabled = 1;
      at.b = b;
      at.ARR = ARR;
      if (fts) {
        IGMC_PLAUNCH("k_finalize_adam", k_finalize_ts, 4 * IGMC_FTS_NB + 32 + 1, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 32, 0, img);
        if (img_emitted) *img_emitted = img;
      } else IGMC_PLAUNCH("k_finalize_adam", k_finalize, 4 * IGMC_FIN_NB + 32, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 1);
    } else {
      if (fts) IGMC_PLAUNCH("k_finalize", k_finalize_ts, 4 * IGMC_FTS_NB, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 0, 0, 0);
      else IGMC_PLAUNCH("k_finalize", k_finalize, 4 * IGMC_FIN_NB, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 1);
      if (loss) igmc_launch_loss(m, b, ARR, loss, stream);
    }
    return 0;
  }
  const size_t l0s = (size_t)(m.R * m.L * 32 + m.L * 32 + 32) * sizeof(float) + (size_t)4 * m.R * m.L * sizeof(int);
  const int g16 = igmc_xcd_grid(m, B, 4, IGMC_GATHER_BLOCKS);
  const int gt = igmc_xcd_grid(m, B, 16, 2048);
  // slots of 129..256 nodes a side with a dense block: every conv layer on the matrix cores (graphstep2.hip, k_dl_layer0 /
  // k_dl_layer) from the blocks alone -- no edge list is read
  // ... and all four of them as ONE launch where the members of a subgraph can hand h_l to each other (k_dl_fwd); the launch
  // sequence number of its exchange tags is advanced by k_tail_ts (tables path) -- else by the launch's last workgroup
  const int fts_pre = igmc_fin_mode() && m.fin_stash && m.datt_part && m.R <= G2_NR * G2_NG_MAX;
  // more than five relations (igmc_dl_wide): the one-launch forward / backward in relation groups + the tables' tail, or nothing
  const int wide = fts_pre && igmc_dl_wide(m, b, B);
  const int dl = wide || igmc_dl_eligible(m, b, B);
  const int dlts = wide || (dl && l0_mfma && fts_pre && m.R <= 8 && igmc_dl_ts_eligible(m, b, B));
  const int dlf = wide || (dl && igmc_dl_fwd_eligible(m, b, B));
  if (dlf) {
    igmc_launch_g2_compose(m, (const float*)P, stream);
    igmc_launch_dl_fwd(m, b, (const float*)P, B, 1, use_flags, m.dpre[3], dlts ? 0 : 1, stream);
  } else if (dl) {
    igmc_launch_g2_compose(m, (const float*)P, stream);
    igmc_launch_dl_layer0(m, b, B, 1, use_flags, stream);
  } else if (use_flags) IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<true, true>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, (const float*)P, m.h[0]);
  else IGMC_PLAUNCH("k_l0_fwd", (k_l0_fwd<false, true>), igmc_l0_grid(m, B), IGMC_BLOCK, l0s, stream, b, m, (const float*)P, m.h[0]);
  const size_t fsm4 = (size_t)(16 * IGMC_TP + 1024 + m.R * 4) * sizeof(float);
  const size_t bsm4 = (size_t)(16 * IGMC_TP + 1024 + m.R * 4 + 16 * m.R * 4) * sizeof(float);
  const int gl = dl ? igmc_dl_grid(b, B) : gt;      // grid of the layer kernels
  for (int l = 1; l < 4 && !dlf; ++l) {
    float* zo = (l == 3) ? m.dpre[3] : nullptr;
    if (dl) {
      igmc_launch_dl_layer(m, b, (const float*)P, B, l, 0, use_flags, zo, stream);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_rgcn_layer_fwd", (k_rgcn_layer4<true, false>), gt, IGMC_BLOCK, fsm4, stream, b, m, (const float*)P, l, zo);
      else IGMC_PLAUNCH("k_rgcn_layer_fwd", (k_rgcn_layer4<false, false>), gt, IGMC_BLOCK, fsm4, stream, b, m, (const float*)P, l, zo);
    }
  }
  const size_t ysz = (size_t)32 * (128 + 4) * sizeof(float);
  // dense layers whose backward passes leave relation-space tables (igmc_dl_ts_eligible): the tail of the subgraph kernel
  // -- k_tail_ts sums the workgroups' tables and forms d lin1 / d lin2, k_finalize_ts turns them into gradients (+ Adam) --
  // replaces the Y products, G, the weight-gradient products and their reduction
  if (dlts) {
    if (wide || (dlf && igmc_dl_bwd_eligible(m, b, B))) {
      // the three backward layers as ONE launch, the loss head of each subgraph (side features included) in its set-up
      // (IGMC_DL_HEAD=0: the head as a launch of its own in front of it)
      const char* eh = getenv("IGMC_DL_HEAD");
      DlHead hd;
      hd.P = (const float*)P; hd.inj_mask = inj_mask; hd.seed = seed; hd.step = step; hd.mult = mult; hd.grad_scale = grad_scale;
      hd.out = out;
      const bool inside = !(eh && atoi(eh) == 0);
      if (!inside) igmc_launch_head_sub(m, b, (const float*)P, B, inj_mask, seed, step, mult, grad_scale, out, stream);
      igmc_launch_dl_bwd(m, b, B, use_flags, stream, inside ? &hd : nullptr);
    } else {
      // the head: one workgroup per subgraph, then one launch per backward layer
      igmc_launch_head_sub(m, b, (const float*)P, B, inj_mask, seed, step, mult, grad_scale, out, stream);
      for (int l = 3; l >= 1; --l) igmc_launch_dl_layer(m, b, (const float*)P, B, l, 1, use_flags, nullptr, stream, 1);
    }
    const int gstride = (B + 7) & ~7, gg = igmc_dl_grid(b, B) / B * gstride;
    IGMC_PLAUNCH("k_tail_ts", k_tail_ts, 8 * ny + (4 * m.ts_stride + 63) / 64 + 4, IGMC_BLOCK, 0, stream, b, m,
                 (const float*)P, grad_scale, mult, 2.f, grad, 8 * ny, gg, gstride, B, 4,
                 (const int64_t*)(adam ? at.ctrl : nullptr), dlf ? 1 : 0);
    if (xch) {
      const int rc = xch->sum(xch->user, m.ts_raw, (int64_t)4 * m.ts_stride + (int64_t)4 * m.ts_stride / 32 * 4,
                              grad + m.off_l1w, n_lin, stream);
      if (rc) return rc;
    }
    if (adam) {
      at.enabled = 1;
      at.b = b;
      at.ARR = ARR;
      IGMC_PLAUNCH("k_finalize_adam", k_finalize_ts, 4 * IGMC_FTS_NB + 32 + 1, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 32, 0, img);
      if (img_emitted) *img_emitted = img;
    } else {
      IGMC_PLAUNCH("k_finalize", k_finalize_ts, 4 * IGMC_FTS_NB, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 0, 0, 0);
      if (loss) igmc_launch_loss(m, b, ARR, loss, stream);
    }
    return 0;
  }
  IGMC_PLAUNCH("k_head_train", k_head_train, dim3(hb > gy ? hb : gy, 4), 512, ysz, stream, b, m, (const float*)P, inj_mask,
               seed, step, mult, grad_scale, out);
  const int dlb = dl;
  for (int l = 3; l >= 1; --l) {
    if (dlb) {
      igmc_launch_dl_layer(m, b, (const float*)P, B, l, 1, use_flags, nullptr, stream);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_rgcn_layer_bwd", (k_rgcn_layer4<true, true>), gt, IGMC_BLOCK, bsm4, stream, b, m, (const float*)P, l, (float*)nullptr);
      else IGMC_PLAUNCH("k_rgcn_layer_bwd", (k_rgcn_layer4<false, true>), gt, IGMC_BLOCK, bsm4, stream, b, m, (const float*)P, l, (float*)nullptr);
    }
  }
  const int nsl = l0_mfma ? 4 : 3;
  IGMC_PLAUNCH("k_wgrad_head", k_wgrad_head, dim3(IGMC_WG_BLOCKS, nsl + 1), IGMC_BLOCK, 0, stream, b, m, (const float*)P,
               (const float*)nullptr, 1, grad_scale, mult, 2.f, grad, nsl);
  if (!l0_mfma) {
    const float* d0 = m.dpre[0];
    if (rows0 <= 64) IGMC_PLAUNCH("k_l0_bwd", (k_l0_bwd<8>), IGMC_L0_BLOCKS, IGMC_BLOCK, 0, stream, b, m, d0, m.l0_part);
    else IGMC_PLAUNCH("k_l0_bwd", (k_l0_bwd<40>), IGMC_L0_BLOCKS, IGMC_BLOCK, 0, stream, b, m, d0, m.l0_part);
  }
  {
    const int wgs2 = igmc_wg_stride(), n0 = rows0 * 32, na = m.R * 4;
    const int nblk = (nsl * wgs2 + 63) / 64 + (l0_mfma ? 0 : (n0 + 63) / 64) + (3 * na + 3) / 4;
    // gradient / Adam tail without hand-offs (k_finalize_ts in basis-space mode) for R <= IGMC_FBS_MAX_R (the
    // layer-0 table comes from the MFMA weight-gradient kernel or from k_l0_bwd's partials: same place, same layout);
    // IGMC_FIN_MODE=0: the hand-off version (k_finalize)
    const int fbs = igmc_fin_mode() && m.fin_stash && m.R <= IGMC_FBS_MAX_R;
    IGMC_PLAUNCH("k_reduce_partials", k_reduce_partials, nblk + (fbs ? 4 : 0), IGMC_BLOCK, 0, stream, m, gl, l0_mfma,
                 IGMC_WG_BLOCKS, (const float*)P, (const int64_t*)(adam ? at.ctrl : nullptr), fbs ? 4 : 0);
    if (xch && fbs) {      // the reduced basis-space sums (+ layer-0 table, d att) and the lin gradients, over the ranks
      const int rc = xch->sum(xch->user, m.graw, (int64_t)3 * wgs2 + 3 * na + n0, grad + m.off_l1w, n_lin, stream);
      if (rc) return rc;
    }
    if (fbs) {
      if (adam) {
        at.enabled = 1;
        at.b = b;
        at.ARR = ARR;
        IGMC_PLAUNCH("k_finalize_adam", k_finalize_ts, 4 * IGMC_FTS_NB + 32 + 1, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 32, 1, img);
        if (img_emitted) *img_emitted = img;
      } else {
        IGMC_PLAUNCH("k_finalize", k_finalize_ts, 4 * IGMC_FTS_NB, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 0, 1, 0);
        if (loss) igmc_launch_loss(m, b, ARR, loss, stream);
      }
      return 0;
    }
  }
  if (adam) {
    at.enabled = 1;
    at.b = b;
    at.ARR = ARR;
    IGMC_PLAUNCH("k_finalize_adam", k_finalize, 4 * IGMC_FIN_NB + 32, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 0);
  } else {
    IGMC_PLAUNCH("k_finalize", k_finalize, 4 * IGMC_FIN_NB, IGMC_BLOCK, 0, stream, m, (const float*)P, grad, ARR * arr_scale, at, 0);
    if (loss) igmc_launch_loss(m, b, ARR, loss, stream);
  }
  return 0;
}

int igmc_launch_train_step(const ModelDev& m, const ModelAux& ax, const BatchDev& b, float* P, int B, int use_flags,
                           const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult, float ARR, float* out,
                           float* grad, float* m1, float* m2, float step_size, float inv_sqrt_bc2, float beta1,
                           float beta2, float eps, float wd, int64_t* ctrl, int* done, float* loss, double* total,
                           void* stream, float grad_scale, const StepExchange* xch, int* img_emitted) {
  AdamTail at;
  memset(&at, 0, sizeof(at));
  at.p = P; at.m1 = m1; at.m2 = m2;
  at.step_size = step_size; at.inv_sqrt_bc2 = inv_sqrt_bc2; at.beta1 = beta1; at.beta2 = beta2; at.eps = eps; at.wd = wd;
  at.ctrl = ctrl; at.done = done; at.loss = loss; at.total = total;
  return igmc_launch_loss_grad(m, ax, b, P, B, use_flags, inj_mask, seed, step, mult, ARR,
                               grad_scale != 0.f ? grad_scale : 1.0f / (float)B, 1.0f, out, grad, nullptr, &at, stream, xch,
                               img_emitted);
}

void igmc_launch_loss(const ModelDev& m, const BatchDev& b, float ARR, float* loss, void* stream) {
  IGMC_PLAUNCH("k_loss", k_loss, 1, IGMC_BLOCK, 0, stream, b, m, ARR, loss);
}

void igmc_launch_sse(const BatchDev& b, const float* out, double* acc, int64_t* ctrl, void* stream) {
  IGMC_PLAUNCH("k_sse_acc", k_sse_acc, 1, IGMC_BLOCK, 0, stream, b, out, acc, ctrl);
}

void igmc_launch_tick(int64_t* ctrl, void* stream) { IGMC_PLAUNCH("k_tick", k_tick, 1, 64, 0, stream, ctrl); }
void igmc_launch_regroup(int64_t* ctrl, int M, int64_t first_cur, int64_t first_next, void* stream) {
  IGMC_PLAUNCH("k_regroup", k_regroup, 1, 64, 0, stream, ctrl, M, first_cur, first_next);
}
void igmc_launch_gate(int64_t* ctrl, int q, int gk_min, long long delay_ticks, int delay_always, long long timeout_ticks,
                      void* stream) {
  IGMC_PLAUNCH("k_step_gate", k_step_gate, 1, 64, 0, stream, ctrl, q, gk_min, delay_ticks, delay_always, timeout_ticks);
}

// (one workgroup per 2048 elements, at most 256: the tick of the step's last workgroup is an atomic round trip per workgroup on
//  ONE counter -- 550 single-round workgroups spent 20 us of a 28 us launch queueing on it)
static int adam_grid(int64_t n) {
  int grid = (int)((n + 8 * IGMC_BLOCK - 1) / (8 * IGMC_BLOCK));
  if (grid > 256) grid = 256;
  return grid < 1 ? 1 : grid;
}

void igmc_launch_adam(float* p, const float* g, float* m1, float* m2, int64_t n, float step_size,
                      float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd, int64_t* ctrl, int tick,
                      void* stream) {
  FinishArgs fin;
  memset(&fin, 0, sizeof(fin));
  IGMC_PLAUNCH("k_adam", k_adam, adam_grid(n), IGMC_BLOCK, 0, stream, p, g, m1, m2, n, step_size, inv_sqrt_bc2, beta1,
               beta2, eps, wd, ctrl, tick, fin);
}

// Adam + loss + epoch total (+ control-block tick) in ONE launch
void igmc_launch_finish(const ModelDev& m, const BatchDev& b, float* p, const float* g, float* m1, float* m2,
                        float step_size, float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd,
                        int64_t* ctrl, float ARR, float* loss, double* total, int use_flags, void* stream) {
  FinishArgs fin;
  fin.enabled = 1;
  fin.use_flags = use_flags;
  fin.b = b;
  fin.m = m;
  fin.ARR = ARR;
  fin.loss = loss;
  fin.total = total;
  IGMC_PLAUNCH("k_step_finish", k_adam, adam_grid(m.n_params), IGMC_BLOCK, 0, stream, p, g, m1, m2,
               (int64_t)m.n_params, step_size, inv_sqrt_bc2, beta1, beta2, eps, wd, ctrl, ctrl ? 1 : 0, fin);
}

int igmc_model_prepare(const ModelDev& m) {
#ifndef IGMC_HIPEMU
  // dynamic LDS above 48 KB needs an explicit opt-in on HIP (many relations -> large d-att tables in the backward)
  const int bsm4 = (int)((size_t)(16 * IGMC_TP + 1024 + m.R * 4 + 16 * m.R * 4) * sizeof(float));
  if (bsm4 > 48 * 1024) {
    if (bsm4 > 150 * 1024) return 1;
    if (hipFuncSetAttribute((const void*)k_rgcn_layer4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bsm4) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_rgcn_layer4<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bsm4) != hipSuccess) return 1;
  }
#endif
  (void)m;
  return igmc_gs_prepare();
}
