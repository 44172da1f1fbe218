// dl_kernels.h -- the dense-layer kernels (k_dl_layer0 / k_dl_layer / k_dl_fwd / k_dl_bwd, k_head_sub) and their host side
// (included once, by graphstep2.hip: they share its gather / transform steps and exchange primitives)
#pragma once

// =====================================================================================================================
// Dense per-layer kernels ("denselayer" path): slots of 129..256 nodes a side (BASELINE config 2: ml_100k, cap 200).
// Too large for the one-launch subgraph kernel above (fragments / planes of K <= 128 fill its registers and LDS), but the
// dense induced block still turns the relational aggregation of a layer into MFMA products:
//   one workgroup = 64 consecutive rows (4 bundles) of ONE side of ONE subgraph, one LAYER PASS per launch;
//   the opposite side's input rows (h_{l-1} forward, dPre_l backward; <= 256 x 32 f32 from HBM / L2) are split into the
//   three bf16 planes in LDS, the bundle's 16 relm rows sit in LDS as bytes and the A_r fragments are expanded from them
//   per k-step (g2_expand4) -- no fragment residency, K loops at run time;
//   gather T_r = A_r X and transform [T_r | x] @ [W_r; root] are g2's (same plane / image layouts, images of k_g2_compose).
// They REPLACE k_rgcn_layer4 forward / backward inside the per-layer sequence of model.hip and keep its contract: forward
// writes h_l; backward writes dPre_{l-1}, the basis-space aggregate G = sum_r att[r,b] T'_r (what the weight-gradient
// kernel multiplies with X^T) and the per-workgroup d att partials <Y_b, T'_r>.  Item-side workgroups read the
// transposed block relmT the extraction keeps for such arenas.
struct DlArgs {
  const int32_t* n_users;
  const int32_t* n_items;
  const int32_t* node_off;
  const uint8_t* relm;
  const uint8_t* relmT;
  int cap_u, cap_v, relm_ld, relmT_ld, nqu, nqv, R, D, l, kp;
  const float* in;         // forward: h_{l-1}; backward: dPre_l                       [N, 32]
  const float* hprev;      // backward: h_{l-1}
  float* out;              // forward: h_l; backward: dPre_{l-1}
  float* zero_out;         // forward, top layer of a training step: dPre_3 rows cleared (or NULL)
  float* gagg;             // backward: G [N, 128]
  const float* Y;          // backward: h_{l-1} @ [basis_0 | .. | basis_3]             [N, 128]
  float* gatt_part;        // backward: [grid][R * 4] partial <Y_b, T'_r>
  const float* gfeat;      // backward: readout gradient on the target rows [B, D] ...
  const float* dcat;       // ... or dense [N, 32] (sort-pool readout)
  const float* img;        // the layer's weight image (forward) / transposed image (backward)
  const float* bias;       // forward
  const float* att;        // backward: [R, 4]
  // backward with relation-space tables (TS): the workgroup's partial table h_{l-1}^T [T'_r | dPre_l] (+ d bias_l) instead
  // of G / the d att partials -- what k_tail_ts sums and k_finalize_ts turns into gradients, as after k_graph_step2
  float* ts_part;          // [4][IGMC_TS_BLOCKS][ts_stride]
  int ts_stride, slot_stride, L;
  const uint16_t* cnt0;    // l == 1: [N, R * L] neighbour-label histograms of layer 0 (k_dl_layer0) ...
  const uint8_t* node_label;   // ... and the nodes' own labels: the layer-0 table gradient is formed here too
};

// acc += a * b as ONE scalar VALU fma.  The d att partials of k_dl_layer<BWD> are 160 of these per lane; left to the
// compiler they become v_pk_fma_f32 / v_pk_mul_f32 chains whose operand pairs are assembled with v_mov and op_sel, and on
// gfx950 that code gave run-to-run different sums on identical inputs (same launch repeated: G and dPre bit-identical, a
// few partials off by 1e-2 relative; always the low halves of the packed chains).  Scalar fmas are reproducible.
// (DL_FMAC: g2_prims.h)
#define DL_NW 8                   // waves (= 16-row bundles) per workgroup of the dense layer kernel
#define DL_THREADS (64 * DL_NW)
// The rows of a side are split EVENLY over its nq workgroups, in whole 16-row bundles: workgroup q takes bundles [bpw q, bpw (q + 1))
// with bpw = ceil(bundles of the side / nq) <= DL_NW.  (First-fill -- 128 rows to workgroup 0, the rest to workgroup 1 -- left a
// flixster launch waiting for its one 8-bundle workgroup while half of the workgroups had no rows at all.)
struct DlRows {
  int base, nact;                 // first row of the workgroup, bundles of it that hold rows (0: nothing of the side here)
};
__device__ __forceinline__ DlRows dl_rows(int n_own, int nq, int q) {
  const int nb = (n_own + 15) >> 4, bpw = (nb + nq - 1) / nq, left = nb - bpw * q;
  DlRows r;
  r.base = 16 * bpw * q;
  r.nact = left < 0 ? 0 : (left < bpw ? left : bpw);
  return r;
}
// Workgroups of a subgraph: at least one per 128 rows of a side's slot capacity, then -- while the whole launch still fits one
// workgroup per CU (224) and the partial-table slots -- one more for the side whose workgroups would hold the most bundles, as
// long as that is more than two: the bundles of a side are split evenly over its workgroups (dl_rows), so a 101 + 101-row
// slot runs as (4 + 3) + (4 + 3) bundles on four CUs, flixster's 50 + 155 rows as 4 + (4 + 3 + 3).
struct DlSplit {
  int nqu, nqv;                   // workgroups of the user side / of the item side
};
static DlSplit dl_split(int cap_u, int cap_v, int B) {
  DlSplit sp;
  sp.nqu = (cap_u + 16 * DL_NW - 1) / (16 * DL_NW);
  sp.nqv = (cap_v + 16 * DL_NW - 1) / (16 * DL_NW);
  const int stride = (B + 7) & ~7;
  int pmax = B > 0 ? 224 / B : 0;
  if (IGMC_TS_BLOCKS / stride < pmax) pmax = IGMC_TS_BLOCKS / stride;
  const int nbu = (cap_u + 15) >> 4, nbv = (cap_v + 15) >> 4;
  while (sp.nqu + sp.nqv < pmax) {
    const int pu = (nbu + sp.nqu - 1) / sp.nqu, pv = (nbv + sp.nqv - 1) / sp.nqv;
    if ((pu > pv ? pu : pv) <= 2) break;
    if (pu >= pv) ++sp.nqu;
    else ++sp.nqv;
  }
  return sp;
}
// (workgroup index -> subgraph, side, workgroup of the side)
#define DL_DECODE(a, bid, g, rem, side, q, nqs)                                      \
  const int g = (bid) / ((a).nqu + (a).nqv), rem = (bid) - g * ((a).nqu + (a).nqv);  \
  const int side = rem >= (a).nqu ? 1 : 0, q = side ? rem - (a).nqu : rem, nqs = side ? (a).nqv : (a).nqu
// ... of the ONE-LAUNCH kernels (k_dl_fwd / k_dl_bwd), whose members exchange rows through the L2 of one XCD: workgroups are
// dealt to the eight XCDs round robin by index, so the members rem = 0 .. nmem - 1 of subgraph 8 j + x are the workgroups
// 8 nmem j + 8 rem + x (the launch is padded to whole blocks of 8 nmem workgroups; g >= B: nothing to do)
#define DLX_DECODE(a, bid, g, rem, side, q, nqs)                                                              \
  const int g = ((bid) / (8 * ((a).nqu + (a).nqv))) * 8 + ((bid) & 7), rem = ((bid) >> 3) % ((a).nqu + (a).nqv); \
  const int side = rem >= (a).nqu ? 1 : 0, q = side ? rem - (a).nqu : rem, nqs = side ? (a).nqv : (a).nqu
#define DL_PIT 2                  // plane-staging items per thread: 128 * k-steps / DL_THREADS, k-steps <= 8
#define DL_RIT 17                 // block-row dwords per lane: 16 rows x (32 * k-steps + 8) / 4 / 64, k-steps <= 8
// LDS plan (4-byte words): [own rows XOA][TS: h_{l-1} rows HSA, d bias scratch][planes | block rows | weight image][sums]
// -- with TS the T' tiles of the table product ALIAS planes / block rows / image (all dead after the transform)
__host__ __device__ static inline int dl_words_x() { return DL_NW * 16 * G2_XP; }
__host__ __device__ static inline int dl_words_front(bool ts) { return dl_words_x() * (ts ? 2 : 1) + (ts ? 16 * 32 : 0); }
__host__ __device__ static inline int dl_words_mid(int kp, bool ts) {
  const int w = (G2_NT * 32 * kp >> 1) + DL_NW * 4 * kp + G2_WIMG, t = DL_NW * 16 * G2_TP;
  return (ts && t > w) ? t : w;
}
template <bool FLAGS, bool BWD, bool TS>
__global__ __launch_bounds__(DL_THREADS) void k_dl_layer(DlArgs a) {
  igmc_kernarg_warm<sizeof(DlArgs) + 32>();
  static_assert(!TS || BWD, "tables are a product of the backward pass");
  IGMC_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  DL_DECODE(a, bid, g, rem, side, q, nqs);
  const int cu = a.n_users[g], cv = a.n_items[g];
  const int n_own = side ? cv : cu, n_opp = side ? cu : cv;
  const int R = a.R;
  const int ts = a.ts_stride;
  // partial-table slot of this workgroup: member c = side * nq + q of subgraph g -> g + c * stride (k_tail_ts's order)
  float* wpart = TS ? a.ts_part + ((size_t)a.l * IGMC_TS_BLOCKS + g + (size_t)rem * a.slot_stride) * ts : nullptr;
  float* part0 = TS ? a.ts_part + ((size_t)g + (size_t)rem * a.slot_stride) * ts : nullptr;
  const int rows0 = R * a.L + a.L + 1;
  const DlRows dr = dl_rows(n_own, nqs, q);
  if (dr.nact == 0) {                            // nothing of this side in the workgroup's rows (uniform)
    if (BWD && !TS && tid < R * 4) a.gatt_part[(size_t)bid * R * 4 + tid] = 0.f;
    if (TS) {                                    // an all-zero partial table (the reduction reads every slot)
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = tid; i < (ts >> 2); i += DL_THREADS) ((float4*)wpart)[i] = z4;
      if (a.l == 1)
        for (int i = tid; i < rows0 * 8; i += DL_THREADS) ((float4*)part0)[i] = z4;
    }
    return;
  }
  const int nb = a.node_off[g];
  const int own0 = nb + (side ? cu : 0), opp0 = nb + (side ? 0 : cu);
  const int kp = a.kp, rmp = kp;                 // plane pitch (bf16) = relm row pitch (bytes) = 32 * max k-steps + 8
  const int nks = (n_opp + 31) >> 5;
  float* XOA = (float*)smem;                                         // [DL_NW][16][G2_XP]
  float* HSA = XOA + dl_words_x();                                   // TS: [DL_NW][16][G2_XP] h_{l-1} rows of the bundles
  float* sbias = HSA + dl_words_x();                                 // TS: [16][32] d bias partial sums
  uint32_t* PLN = (uint32_t*)(XOA + dl_words_front(TS));             // [3][32][kp] bf16
  unsigned char* RMW = (unsigned char*)(PLN + (G2_NT * 32 * kp >> 1));   // [DL_NW waves][16 rows][rmp] bytes
  float2* sW2 = (float2*)(RMW + DL_NW * 16 * rmp);                   // [G2_WIMG words]
  float* TIL = (float*)PLN;                                          // TS: [DL_NW][16][G2_TP] T' tiles (aliases the above)
  float* sred = (float*)PLN + dl_words_mid(kp, TS);                  // [DL_NW][32] + att [32]
  float* s_att = sred + DL_NW * 32;
  const int row0 = dr.base + 16 * wave;
  const bool active = wave < dr.nact;              // (=> row0 < n_own; the waves past the workgroup's bundles idle)
  // ---- staging.  A workgroup is one residency round of the launch, so its duration is its chain of memory round trips:
  // EVERY load of the staging work is requested here, before the first use (clamped addresses instead of predicates: no
  // branches, values zeroed afterwards), and the scheduling barrier keeps the compiler from sinking them to their uses.
  constexpr int NWQ = (G2_WIMG / 4 + DL_THREADS - 1) / DL_THREADS;
  f32x4 wq[NWQ];                                   // weight image (5 x 16 bytes per thread; a native vector type: a
#pragma unroll                                     // float4 struct copy is a memcpy the optimiser leaves in scratch)
  for (int u = 0; u < NWQ; ++u) {
    const int i = tid + u * DL_THREADS;
    wq[u] = ((const f32x4*)a.img)[i < G2_WIMG / 4 ? i : G2_WIMG / 4 - 1];
  }
  const int npair = 16 * nks;                      // node pairs covered by the k-steps (<= 16 * 8)
  float4 x0[DL_PIT], x1[DL_PIT];                   // opposite side's rows: a thread takes two nodes x four features
#pragma unroll
  for (int u = 0; u < DL_PIT; ++u) {
    const int i = tid + u * DL_THREADS, jp = i >> 3, fq = i & 7;
    const int j0 = 2 * jp < n_opp ? 2 * jp : n_opp - 1, j1 = 2 * jp + 1 < n_opp ? 2 * jp + 1 : n_opp - 1;
    x0[u] = *(const float4*)(a.in + (size_t)(opp0 + j0) * 32 + 4 * fq);
    x1[u] = *(const float4*)(a.in + (size_t)(opp0 + j1) * 32 + 4 * fq);
  }
  const int ldb = side ? a.relmT_ld : a.relm_ld, ldw = ldb >> 2;
  const uint8_t* rsrc = side ? a.relmT + (size_t)g * a.cap_v * a.relmT_ld : a.relm + (size_t)g * a.cap_u * a.relm_ld;
  const int rw = rmp >> 2;
  // (r = i / rw for i < 17 * 64 without a division per element: the quotient by a 20-bit reciprocal is exact there)
  const uint32_t rw_magic = ((1u << 20) + (uint32_t)rw - 1u) / (uint32_t)rw;
  uint32_t rmq[DL_RIT];                            // this wave's 16 rows of the dense block, dwords lane + 64 u
#pragma unroll
  for (int u = 0; u < DL_RIT; ++u) {
    const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
    const int rc = row0 + r < n_own ? row0 + r : n_own - 1, cc = c < ldw ? c : ldw - 1;
    rmq[u] = ((const uint32_t*)(rsrc + (size_t)rc * ldb))[cc];
  }
  float4 xq[2];                                    // own rows of the layer input (root / self term of the transform)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = lane + 64 * u, r = i >> 3, c4 = i & 7;
    const int rc = row0 + r < n_own ? row0 + r : n_own - 1;
    xq[u] = *(const float4*)(a.in + (size_t)(own0 + rc) * 32 + 4 * c4);
  }
  float biasv[2] = {0.f, 0.f};
  if (!BWD) {
    biasv[0] = a.bias[li];
    biasv[1] = a.bias[16 + li];
  }
  if (BWD && !TS && tid < 32) s_att[tid] = (tid < R * 4) ? a.att[tid] : 0.f;
  // TS, layer 1: the rows' layer-0 inputs [neighbour-label histogram | own label | 1] for the layer-0 table gradient
  const int RL = R * a.L;
  uint16_t c0q[5];                                 // histogram entries lane + 64 u of the wave's 16 rows (R L <= 20)
  int own_lab = 0;
  if (TS && a.l == 1) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = lane + 64 * u, r = i / RL, c = i - r * RL;
      const int rc = (r < 16 && row0 + r < n_own) ? row0 + r : n_own - 1;
      c0q[u] = a.cnt0[(size_t)(own0 + rc) * RL + (r < 16 ? c : 0)];
    }
    own_lab = (int)a.node_label[own0 + (row0 + li < n_own ? row0 + li : n_own - 1)];
  }
  G2_SCHED_BARRIER();
  // ---- planes: three bf16 terms of two nodes' features per word
  {
    const int tstride = 32 * kp >> 1;
#pragma unroll
    for (int u = 0; u < DL_PIT; ++u) {
      const int i = tid + u * DL_THREADS, jp = i >> 3, fq = i & 7;
      if (i >= npair * 8) continue;
      const bool k0 = 2 * jp < n_opp, k1 = 2 * jp + 1 < n_opp;
      const float v0[4] = {x0[u].x, x0[u].y, x0[u].z, x0[u].w}, v1[4] = {x1[u].x, x1[u].y, x1[u].z, x1[u].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t h, mi, lo;
        g2_split2(k0 ? v0[c] : 0.f, k1 ? v1[c] : 0.f, h, mi, lo);
        uint32_t* p = PLN + ((4 * fq + c) * kp >> 1) + jp;
        p[0] = h;
        p[tstride] = mi;
        p[2 * tstride] = lo;
      }
    }
  }
  // ---- the wave's block rows (bytes; rows past the side and columns past the block are zero)
  {
    uint32_t* dst = (uint32_t*)(RMW + (size_t)wave * 16 * rmp);
#pragma unroll
    for (int u = 0; u < DL_RIT; ++u) {
      const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
      if (i < 16 * rw) dst[i] = (row0 + r < n_own && c < ldw && 4 * c < 32 * nks) ? rmq[u] : 0u;
    }
  }
  {
    float* XO = XOA + wave * 16 * G2_XP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = lane + 64 * u, r = i >> 3, c4 = i & 7;
      *(float4*)(XO + r * G2_XP + 4 * c4) = (active && row0 + r < n_own) ? xq[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int u = 0; u < NWQ; ++u) {
    const int i = tid + u * DL_THREADS;
    if (i < G2_WIMG / 4) ((f32x4*)sW2)[i] = wq[u];
  }
  __syncthreads();
  if (TS) {       // d bias_l = column sums of dPre_l over the workgroup's rows (fixed order): 16 partial sums per column ...
    const int n = tid & 31, part = tid >> 5;
    float sb = 0.f;
    for (int row = part; row < DL_NW * 16; row += DL_THREADS / 32) sb += XOA[(row >> 4) * 16 * G2_XP + (row & 15) * G2_XP + n];
    sbias[part * 32 + n] = sb;
  }
  float gsum[TS ? 1 : G2_NR * 4];
#pragma unroll
  for (int i = 0; i < (TS ? 1 : G2_NR * 4); ++i) gsum[i] = 0.f;
  f32x4 acc[G2_NR][2];                             // T_r (forward) / T'_r (backward) of the bundle: lane = row li
#pragma unroll
  for (int r = 0; r < G2_NR; ++r) {
    acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float xprev[2][4], dv[2][4];                     // backward: h_{l-1} / dPre_{l-1} of rows 4 kq + rr, feature 16 nt + li
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      xprev[nt][rr] = 0.f;
      dv[nt][rr] = 0.f;
    }
  if (active) {
    // ---- T_r^T = X^T A_r^T over the k-steps of the opposite side; fragments expanded per k-step from the row's bytes
    const int row = row0 + li;                       // this lane's row in the gather's accumulators
    // backward: everything the epilogues read from HBM / L2 is requested BEFORE the gather (Y rows of the lane's row for
    // the d att partials, h_{l-1} of the transform's output rows for tanh'): eight + eight dependent round trips otherwise
    float4 ypre[TS ? 1 : 4][2];
    float addv[2][4];
    if (BWD) {
      const int rowc = row < n_own ? row : n_own - 1;
      if (!TS) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            ypre[bb][t] = *(const float4*)(a.Y + (size_t)(own0 + rowc) * 128 + bb * 32 + 16 * t + 4 * kq);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int rw = row0 + 4 * kq + rr, rwc = rw < n_own ? rw : n_own - 1;
          xprev[nt][rr] = a.hprev[(size_t)(own0 + rwc) * 32 + 16 * nt + li];
          // what is added to the transform's output: the readout gradient (dense: sort-pool; else the target row only)
          addv[nt][rr] = 0.f;
          if (a.dcat) addv[nt][rr] = a.dcat[(size_t)(own0 + rwc) * 32 + 16 * nt + li];
          else if (rw == 0 && a.gfeat) addv[nt][rr] = a.gfeat[(size_t)g * a.D + side * 128 + (a.l - 1) * 32 + 16 * nt + li];
        }
    }
    const unsigned char* rmo = RMW + (size_t)(wave * 16 + li) * rmp + 8 * kq;
    const int tstride = 32 * kp >> 1, toff = 16 * kp >> 1;
    const uint32_t* base = PLN + (li * kp >> 1) + 4 * kq;
    // keep bit of the direction this pass walks: forward = edge opposite -> own, backward = own -> opposite
    const int kbit = BWD ? (side ? IGMC_RELM_KF : IGMC_RELM_KT) : (side ? IGMC_RELM_KT : IGMC_RELM_KF);
#pragma unroll 1
    for (int s = 0; s < nks; ++s) {
      const uint2 w = *(const uint2*)(rmo + 32 * s);
      u32x4 pf[2 * G2_NT];
#pragma unroll
      for (int sp = 0; sp < G2_NT; ++sp) {
        pf[2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
        pf[2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s);
      }
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) {
        u32x4 af;
        uint32_t a0, a1, a2, a3;
        g2_expand4<FLAGS>(w.x, (uint32_t)(r + 1), kbit, a0, a1);
        g2_expand4<FLAGS>(w.y, (uint32_t)(r + 1), kbit, a2, a3);
        af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
#pragma unroll
        for (int qq = 0; qq < 2 * G2_NT; ++qq) acc[r][qq & 1] = g2_mfma_bf16(pf[qq], af, acc[r][qq & 1]);
      }
    }
    if (BWD && !TS) {
      // ---- basis-space aggregate G (what the weight-gradient kernel multiplies with X^T) and the d att partials
      if (row < n_own) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 y4 = ypre[bb][t];
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
            float gv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < G2_NR; ++r) {
              const float at = s_att[r * 4 + bb];
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                gv[rr] += at * acc[r][t][rr];
                DL_FMAC(gsum[r * 4 + bb], yv[rr], acc[r][t][rr]);
              }
            }
            *(float4*)(a.gagg + (size_t)(own0 + row) * 128 + bb * 32 + 16 * t + 4 * kq) = make_float4(gv[0], gv[1], gv[2], gv[3]);
          }
      }
    }
    // ---- dense transform + epilogue (lane = output feature 16 nt + li, registers = rows 4 kq + rr)
    f32x4 o[2];
    g2_transform(acc, XOA + wave * 16 * G2_XP, (const uint32_t*)sW2, li, kq, o);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int f = 16 * nt + li;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int rw = row0 + 4 * kq + rr;
        if (rw < n_own) {
          const size_t at = (size_t)(own0 + rw) * 32 + f;
          if (!BWD) {
            a.out[at] = g2_tanh(o[nt][rr] + biasv[nt]);
            if (a.zero_out) a.zero_out[at] = 0.f;
          } else {
            const float d = o[nt][rr] + addv[nt][rr];
            const float x = xprev[nt][rr];
            dv[nt][rr] = d * (1.f - x * x);
            a.out[at] = dv[nt][rr];
          }
        } else if (BWD) {
          xprev[nt][rr] = 0.f;                       // (rows past the side: zero K entries of the table product)
        }
      }
    }
  }
  if (TS) {
    // ---- weight-gradient table h_{l-1}^T [T'_0 .. T'_4 | dPre_l] over the workgroup's rows, as in k_graph_step2: the
    // bundles' T' tiles and h rows go to LDS (the tiles over planes / block rows / weight image: every wave is done with
    // them at the barrier), then the 2 x 12 output tiles are split over the 8 waves (3 each), K = the active bundles' rows
    // -- no cross-wave reduction, one plain store per value (one subgraph per workgroup)
    __syncthreads();
    if (tid < 32) {                                // ... (d bias) summed in a fixed order
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < DL_THREADS / 32; ++p) s += sbias[p * 32 + tid];
      wpart[(R * 32 + 32) * 32 + tid] = s;
    }
    float* T = TIL + wave * 16 * G2_TP;
    float* HS = HSA + wave * 16 * G2_XP;
    if (active) {
#pragma unroll
      for (int r = 0; r < G2_NR; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *(float4*)(T + li * G2_TP + r * 32 + 16 * t + 4 * kq) = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) HS[(4 * kq + rr) * G2_XP + 16 * nt + li] = xprev[nt][rr];
    }
    __syncthreads();
    const int nact = dr.nact;                      // bundles of this workgroup that hold rows
    {
      f32x4 w3[3];
#pragma unroll
      for (int i3 = 0; i3 < 3; ++i3) w3[i3] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int m2w = wave >> 2, nt0 = 3 * (wave & 3);       // in-feature half, first of the wave's three column tiles
#pragma unroll 1
      for (int wb = 0; wb < nact; ++wb) {
        const float* Tb = TIL + wb * 16 * G2_TP;
        const float* Hb = HSA + wb * 16 * G2_XP;
        const float* Db = XOA + wb * 16 * G2_XP;
        float av[4], bw[4][3];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          av[s4] = Hb[(4 * s4 + kq) * G2_XP + m2w * 16 + li];
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) {
            const int nt = nt0 + i3;                 // column tile: 0..9 = T' of relation nt >> 1, 10..11 = dPre_l (d root)
            bw[s4][i3] = (nt < 2 * G2_NR) ? Tb[(4 * s4 + kq) * G2_TP + nt * 16 + li]
                                          : Db[(4 * s4 + kq) * G2_XP + (nt - 2 * G2_NR) * 16 + li];
          }
        }
        G2_SCHED_BARRIER();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) w3[i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bw[s4][i3], w3[i3], 0, 0, 0);
      }
#pragma unroll
      for (int i3 = 0; i3 < 3; ++i3) {
        const int nt = nt0 + i3, r = nt >> 1;        // 32-column block: relation, or G2_NR = root
        if (r >= R && r < G2_NR) continue;
        float* pp = wpart + (kq * 4) * 32 + li + (r < R ? r : R) * 1024 + m2w * 512 + (nt & 1) * 16;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = w3[i3][rr];
      }
    }
    if (a.l == 1) {
      // ---- layer-0 table gradient T0'[c][f] = sum_i [hist | onehot | 1](i, c) dPre_0[i][f] over the workgroup's rows:
      // the rows' inputs and dPre_0 (this launch's output, still in registers) as tiles over the T' tiles (dead now)
      __syncthreads();
      float* HI = TIL + wave * 16 * G2_XP;                              // [DL_NW][16][G2_XP] inputs
      float* D0 = TIL + (DL_NW + wave) * 16 * G2_XP;                    // [DL_NW][16][G2_XP] dPre_0
      if (active) {
        for (int i = lane; i < 16 * G2_XP; i += 64) HI[i] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) D0[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
      }
      __syncthreads();                               // (HI zero fill before the scattered writes below)
      if (active) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = lane + 64 * u, r = i / RL, c = i - r * RL;
          if (r < 16 && row0 + r < n_own) HI[r * G2_XP + c] = (float)c0q[u];
        }
        if (kq == 0 && row0 + li < n_own) {
          HI[li * G2_XP + RL + own_lab] = 1.f;
          HI[li * G2_XP + RL + a.L] = 1.f;
        }
      }
      __syncthreads();
      if (wave < 4) {
        const int m2 = wave >> 1, wn = wave & 1;     // code half, feature half
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int wb = 0; wb < nact; ++wb) {
          const float* Hb = TIL + wb * 16 * G2_XP;
          const float* Db = TIL + (DL_NW + wb) * 16 * G2_XP;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Hb[(4 * s4 + kq) * G2_XP + m2 * 16 + li],
                                                        Db[(4 * s4 + kq) * G2_XP + wn * 16 + li], acc0, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int c = m2 * 16 + kq * 4 + rr;
          if (c < rows0) part0[c * 32 + wn * 16 + li] = acc0[rr];
        }
      }
    }
  }
  if (BWD && !TS) {
    // d att partial of the workgroup: lanes -> wave (fixed order), waves -> workgroup
    // the 20 butterflies side by side: one LDS-crossbar round trip per step for all of them (one after the other they are
    // 120 dependent round trips, ~4 us of this kernel); same order of additions per value
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      float t[G2_NR * 4];
#pragma unroll
      for (int i = 0; i < G2_NR * 4; ++i) t[i] = __shfl_xor(gsum[i], d, 64);
#pragma unroll
      for (int i = 0; i < G2_NR * 4; ++i) gsum[i] += t[i];
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < G2_NR * 4; ++i) sred[wave * 32 + i] = gsum[i];
    }
    __syncthreads();
    if (tid < R * 4) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < DL_NW; ++w) s += sred[w * 32 + tid];
      a.gatt_part[(size_t)bid * R * 4 + tid] = s;
    }
  }
}

// =====================================================================================================================
// The forward of the dense per-layer path as ONE launch: layer 0 and the three conv layers of k_dl_layer0 / k_dl_layer with
// the members of a subgraph (2 nq workgroups: row blocks of 128 x two sides) handing h_l to each other through tagged
// exchange words, as the members of k_graph_step2's clusters do.  A per-layer launch spends 12 of its 16 us outside the
// matrix cores (launch, staging round trips, the launch's tail); here the block rows are staged once, a layer's weight
// image is requested a layer ahead, the opposite side's rows arrive as bf16 terms already in plane order, and a layer
// boundary is one poll of the exchange.  Exchange regions: [exchange x][subgraph][side][32 features][DLX_K nodes]
// (DLX_K = 256: g2_prims.h).
struct DlfArgs {
  const int32_t* n_users;
  const int32_t* n_items;
  const int32_t* node_off;
  const uint8_t* node_label;
  const uint8_t* s_lab;          // the arena's slot-based labels [graph][slot] (node_label is its collated copy)
  int slot;
  const uint8_t* relm;
  const uint8_t* relmT;
  int cap_u, cap_v, relm_ld, relmT_ld, nqu, nqv, R, L, kp;
  float* h[4];
  float* zero_out;               // training: dPre_3 rows cleared (or NULL)
  uint16_t* cnt0;                // training: [N, R * L] (or NULL)
  const float* g2_w;             // forward / transposed images of layers 1..3, then the layer-0 table
  const float* P;
  int off_bias[4];
  unsigned long long* ex;
  size_t ex_stride;              // words per exchange
  int* gs_bar;
  int* gs_err;
  int self_seq;                  // 1: the last workgroup advances the launch sequence number (no kernel follows that would)
  int timing;                    // debug aid: workgroup + 1 whose phase clocks are recorded (DL_STAMP)
  int B;                         // subgraphs of the batch (the launch is padded to XCD-aligned blocks of workgroups)
};

// NG = relation groups (g2_image.h): NG > 1 takes the relations five at a time -- gather of group g, then the transform with
// group g's image accumulating into the same output; one image is staged at a time (the next one is requested while the
// current group's matrix work runs) -- and lays the 64-row layer-0 table behind the image.
#define DLF_HP2 52                // pitch of a row's layer-0 input [hist | onehot | 1] with NG > 1 (R L + L + 1 <= 48)
__host__ __device__ static inline int dlf_words(int kp, int ng = 1) {
  return 2 * DL_NW * 16 * G2_XP + (G2_NT * 32 * kp >> 1) + DL_NW * 4 * kp + G2_WIMG + (ng > 1 ? 64 * 32 : 0);
}
// GS (group split, two relation groups, at most DL_NW / 2 bundles a workgroup): waves 0..3 take the bundles' first relation group,
// waves 4..7 the second one, both groups' images resident -- half as many bundle slots (row tiles, block rows), one tile per bundle
// for the second group's partial output, two images
#define DL_GB (DL_NW / 2)
__host__ __device__ static inline int dlf_words_gs(int kp) {
  return 3 * DL_GB * 16 * G2_XP + (G2_NT * 32 * kp >> 1) + DL_GB * 4 * kp + 2 * G2_WIMG + 64 * 32;
}

template <bool FLAGS, bool STORE, int NG, bool GS = false>
__global__ __launch_bounds__(DL_THREADS) void k_dl_fwd(DlfArgs a) {
  __builtin_amdgcn_s_setprio(3);      // (step chain: ahead of the extraction chain's waves wherever the two share a SIMD)
  igmc_kernarg_warm<sizeof(DlfArgs) + 32>();
  static_assert(!GS || NG == 2, "the group split is for two relation groups");
  constexpr int HP = (NG == 1) ? G2_XP : DLF_HP2;
  constexpr int NB = GS ? DL_GB : DL_NW;           // bundle slots of the workgroup
  IGMC_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bw = GS ? (wave & (DL_GB - 1)) : wave;      // bundle of the wave, and (GS) the relation group it takes
  const int gw = GS ? wave / DL_GB : 0;
  const int li = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  DLX_DECODE(a, bid, g, rem, side, q, nqs);
  if (g >= a.B) {                                // padding of the launch
    dlx_seq_done(a.gs_bar, a.self_seq);
    return;
  }
  const int cu = a.n_users[g], cv = a.n_items[g];
  const int n_own = side ? cv : cu, n_opp = side ? cu : cv;
  const uint32_t seq = g2_ld_seq(a.gs_bar);
  const uint32_t tag0 = seq * 8u + 1u;
  auto xtag = [&](int x) { return tag0 + (uint32_t)x; };      // flag value of exchange x of this launch (never 0)
  const DlRows dr = dl_rows(n_own, nqs, q);
  if (dr.nact == 0) {                            // nothing of this side in the workgroup's rows: nobody waits for it
    dlx_seq_done(a.gs_bar, a.self_seq);
    return;
  }
  DL_STAMP(0);
  const int R = a.R, L = a.L, RL = R * L;
  const int ngr = (NG == 1) ? 1 : g2_rel_groups(R);      // groups that hold relations (a two-hop model's second group: none)
  const int nb = a.node_off[g];
  const int own0 = nb + (side ? cu : 0), opp0 = nb + (side ? 0 : cu);
  const int kp = a.kp, rmp = kp;
  const int nks = (n_opp + 31) >> 5;
  const int npad_opp = ((n_opp + 15) >> 4) << 4;
  float* XO0 = (float*)smem;                                              // [NB][16][G2_XP] ping
  float* XO1 = XO0 + NB * 16 * G2_XP;                                     // pong
  float* PXP = XO1 + NB * 16 * G2_XP;                                     // GS: [NB][16][G2_XP] the second group's partial outputs
  uint32_t* PLN = (uint32_t*)(XO1 + (GS ? 2 : 1) * NB * 16 * G2_XP);      // [3][32][kp] bf16
  unsigned char* RMW = (unsigned char*)(PLN + (G2_NT * 32 * kp >> 1));    // [NB][16][rmp] bytes
  float2* sW2 = (float2*)(RMW + NB * 16 * rmp);                           // [G2_WIMG words]
  // layer 0 only, inside the image's space: one-hot label planes, the rows' inputs, the layer-0 table
  uint32_t* OHP = (uint32_t*)sW2;                                         // [8 labels][kp] bf16
  float* HIA = (float*)(OHP + (8 * kp >> 1));                             // [NB][16][HP]
  float* sT0 = (NG == 1) ? HIA + DL_NW * 16 * G2_XP : (float*)sW2 + G2_WIMG;      // [32][32] / behind the image: [64][32]
  float2* sW2b = (float2*)(sT0 + 64 * 32);                                // GS: the second group's image
  const int row0 = dr.base + 16 * bw;
  const bool active = bw < dr.nact;
  const bool lead = !GS || gw == 0;                // the wave that owns the bundle's rows (set-up, layer 0, epilogues)
  // plane exchange regions (g2_prims.h: bf16 term planes [term][feature][kp] + a flag per bundle, through the XCD's L2)
  const size_t exs = a.ex_stride;
  auto px_of = [&](int x, int sd) { return (unsigned char*)(a.ex + (size_t)x * exs + ((size_t)g * 2 + sd) * (32 * DLX_K)); };
  static_assert(32 * DLX_K * 8 == DLX_PX_BYTES, "an exchange region of the dense-layer kernels is 64 KB");

  // ---- staging: every global load of the set-up is requested before the first use
  const int ldb = side ? a.relmT_ld : a.relm_ld, ldw = ldb >> 2;
  const uint8_t* rsrc = side ? a.relmT + (size_t)g * a.cap_v * a.relmT_ld : a.relm + (size_t)g * a.cap_u * a.relm_ld;
  const int rw = rmp >> 2;
  // (r = i / rw for i < 17 * 64 without a division per element: the quotient by a 20-bit reciprocal is exact there)
  const uint32_t rw_magic = ((1u << 20) + (uint32_t)rw - 1u) / (uint32_t)rw;
  uint32_t rmq[DL_RIT];
#pragma unroll
  for (int u = 0; u < DL_RIT; ++u) {
    const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
    const int rc = row0 + r < n_own ? row0 + r : n_own - 1, cc = c < ldw ? c : ldw - 1;
    rmq[u] = lead ? ((const uint32_t*)(rsrc + (size_t)rc * ldb))[cc] : 0u;
  }
  // (labels from the arena's slot-based array, of which the collated node_label is a copy: no wait for the subgraph's node
  //  offset in front of these loads)
  const uint8_t* slab_own = a.s_lab + (size_t)g * a.slot + (side ? a.cap_u : 0);
  const uint8_t* slab_opp = a.s_lab + (size_t)g * a.slot + (side ? 0 : a.cap_u);
  const int own_lab = (int)slab_own[row0 + li < n_own ? row0 + li : n_own - 1];
  int l0 = 255, l1 = 255;                         // labels of the opposite side's node pair tid (< 16 nks <= 128)
  if (tid < 16 * nks) {
    l0 = (2 * tid < n_opp) ? (int)slab_opp[2 * tid] : 255;
    l1 = (2 * tid + 1 < n_opp) ? (int)slab_opp[2 * tid + 1] : 255;
  }
  // layer-0 table: 1024 (NG = 1: eight bytes a thread) / 2048 floats
  const float2 t0v = ((const float2*)(a.g2_w + g2_t0_off(NG)))[tid];
  const float2 t0w = (NG > 1) ? ((const float2*)(a.g2_w + g2_t0_off(NG)))[DL_THREADS + tid] : make_float2(0.f, 0.f);
  constexpr int NWQ = (G2_WIMG / 4 + DL_THREADS - 1) / DL_THREADS;
  f32x4 wq[NWQ];                                  // a layer's weight image, requested a layer ahead
  auto wpre = [&](int l, int grp) {
    const f32x4* src = (const f32x4*)(a.g2_w + g2_img_off(NG, l, 0, grp));
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
      const int i = tid + u * DL_THREADS;
      wq[u] = src[i < G2_WIMG / 4 ? i : G2_WIMG / 4 - 1];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
      const int i = tid + u * DL_THREADS;
      if (i < G2_WIMG / 4) ((f32x4*)sW2)[i] = wq[u];
    }
  };
  // GS: the image of layer l for THIS wave's relation group, global -> LDS directly (no registers in between), 1 KB pieces
  // dealt to the four waves of the group; requested when its space is free, landed by the barrier in front of its first use
  auto wload = [&](int l) {
    const float4* src = (const float4*)(a.g2_w + g2_img_off(NG, l, 0, gw));
    float4* dst = (float4*)(gw ? sW2b : sW2);
#pragma unroll
    for (int j = 0; j < (G2_WIMG / 256 + DL_GB - 1) / DL_GB; ++j) {
      const int c = bw + j * DL_GB;
      if (c < G2_WIMG / 256) g2_glds16(src + c * 64, dst + c * 64, lane);
    }
  };
  G2_SCHED_BARRIER();
  {   // zero fills under the loads' latency: planes (k-steps past the published rows must read zeros), both row tiles, inputs
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < (G2_NT * 32 * kp >> 3); i += DL_THREADS) ((float4*)PLN)[i] = z4;
    for (int i = tid; i < 2 * NB * 16 * G2_XP / 4; i += DL_THREADS) ((float4*)XO0)[i] = z4;
    for (int i = tid; i < NB * 16 * HP / 4; i += DL_THREADS) ((float4*)HIA)[i] = z4;
  }
  if (tid < 16 * nks) {
#pragma unroll
    for (int lb = 0; lb < 8; ++lb) OHP[(lb * kp >> 1) + tid] = ((l0 == lb) ? 0x3F80u : 0u) | ((l1 == lb) ? 0x3F800000u : 0u);
  }
  DL_STAMP(31);
  if (lead) {
    uint32_t* dst = (uint32_t*)(RMW + (size_t)bw * 16 * rmp);
#pragma unroll
    for (int u = 0; u < DL_RIT; ++u) {
      const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
      if (i < 16 * rw) dst[i] = (row0 + r < n_own && c < ldw && 4 * c < 32 * nks) ? rmq[u] : 0u;
    }
  }
  DL_STAMP(32);
  ((float2*)sT0)[tid] = t0v;
  if (NG > 1) ((float2*)sT0)[DL_THREADS + tid] = t0w;
  if (!GS) wpre(1, 0);
  __syncthreads();
  DL_STAMP(1);

  const unsigned char* rmo = RMW + (size_t)(bw * 16 + li) * rmp + 8 * kq;
  const int kbit = side ? IGMC_RELM_KT : IGMC_RELM_KF;                   // keep bit of the edge opposite -> own
  // epilogue of a layer: the bundle's rows -> LDS tile (next layer's own rows), h_l, exchange x = l (bf16 terms)
  auto fwd_out = [&](int l, const float (&v)[2][4], float* XO) {
    float* hrow = a.h[l] + (size_t)(own0 + row0 + 4 * kq) * 32 + li;
    float w[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) w[nt][rr] = (row0 + 4 * kq + rr < n_own) ? v[nt][rr] : 0.f;
    // the planes and the bundle's flag first (what the other side waits for), the wave's own copies behind them
    if (l < 3) {
      g2_publish_planes(px_of(l, side), kp, li, row0 + 4 * kq, w[0]);
      g2_publish_planes(px_of(l, side), kp, 16 + li, row0 + 4 * kq, w[1]);
      g2_flag_raise(px_of(l, side), row0 >> 4, xtag(l), lane, DLX_PX_FLAGS);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        XO[(4 * kq + rr) * G2_XP + 16 * nt + li] = w[nt][rr];
        if (row0 + 4 * kq + rr < n_own) {
          hrow[rr * 32 + 16 * nt] = w[nt][rr];
          if (l == 3 && a.zero_out) a.zero_out[(size_t)(own0 + row0 + 4 * kq + rr) * 32 + 16 * nt + li] = 0.f;
        }
      }
  };
  // the opposite side's planes of exchange x: wait for the flags of its bundles, then global -> LDS (landed by the barrier)
  auto fetch = [&](int x) {
    g2_flags_wait(px_of(x, 1 - side), (n_opp + 15) >> 4, xtag(x), lane, a.gs_err, DLX_PX_FLAGS);
    g2_planes_load_exact(PLN, px_of(x, 1 - side), 192 * kp, wave, lane, DL_NW);
  };

  // ================================================================ layer 0: h_0 = tanh([hist | onehot(label) | 1] @ T0)
  // (GS: the two waves of a bundle count one relation group each into the bundle's input tile)
  if (active) {
    const uint32_t* ohp = OHP + ((li & 7) * kp >> 1) + 4 * kq;
    float* hi = HIA + bw * 16 * HP;
    const int row = row0 + li;
    const int g_lo = GS ? gw : 0, g_hi = GS ? gw + 1 : ngr;
#pragma unroll 1
    for (int grp = g_lo; grp < g_hi; ++grp) {
      const uint32_t rb = (uint32_t)(G2_NR * grp);
      f32x4 hacc[G2_NR];
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) hacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int s = 0; s < nks; ++s) {
        const uint2 w = *(const uint2*)(rmo + 32 * s);
        u32x4 pfh = *(const u32x4*)(ohp + 16 * s);
        if (li >= 8) pfh = (u32x4){0u, 0u, 0u, 0u};        // (label rows 8..15 of the 16-row operand do not exist)
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          u32x4 af;
          uint32_t a0, a1, a2, a3;
          g2_expand4<FLAGS>(w.x, rb + (uint32_t)(r + 1), kbit, a0, a1);
          g2_expand4<FLAGS>(w.y, rb + (uint32_t)(r + 1), kbit, a2, a3);
          af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
          hacc[r] = g2_mfma_bf16(pfh, af, hacc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < G2_NR; ++r)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int c = 4 * kq + rr, rg = (int)rb + r;
          if (rg < R && c < L) {
            hi[li * HP + rg * L + c] = hacc[r][rr];
            if (STORE && row < n_own) a.cnt0[(size_t)(own0 + row) * RL + rg * L + c] = (uint16_t)(int)(hacc[r][rr] + 0.5f);
          }
        }
    }
    if (lead && kq == 0 && row < n_own) {
      hi[li * HP + RL + own_lab] = 1.f;
      hi[li * HP + RL + L] = 1.f;
    }
  }
  if (GS) __syncthreads();                           // (both halves of the bundles' input tiles)
  if (active && lead) {
    float* hi = HIA + bw * 16 * HP;
    if (!GS) IGMC_WAVE_SYNC();                       // (the tile is this wave's own: no workgroup barrier)
    // [hist | onehot | 1] (16 rows x 32 / 48 table rows) @ T0 on the f32 matrix cores: the accumulators (lane = feature
    // 16 nt + li, registers = rows 4 kq + rr) are the epilogue's layout
    f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
    constexpr int NJ = (NG == 1) ? 8 : 12;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float av = hi[li * HP + 4 * j + kq];
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT0[(4 * j + kq) * 32 + li], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT0[(4 * j + kq) * 32 + 16 + li], o1, 0, 0, 0);
    }
    float v[2][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      v[0][rr] = g2_tanh(o0[rr]);
      v[1][rr] = g2_tanh(o1[rr]);
    }
    fwd_out(0, v, XO0 + bw * 16 * G2_XP);
  }
  __syncthreads();                                   // the image's space (one-hot planes, inputs, table) is free
  DL_STAMP(2);
  if (GS) wload(1);                                  // (requested here, not earlier: a pending one would be waited for at every barrier above)

  // ================================================================ conv layers 1..3
#pragma unroll 1
  for (int l = 1; l < 4; ++l) {
    float* XOc = ((l & 1) ? XO0 : XO1) + bw * 16 * G2_XP;        // h_{l-1} of the bundle's rows
    float* XOn = ((l & 1) ? XO1 : XO0) + bw * 16 * G2_XP;        // h_l
    if constexpr (GS) {
      // both groups at once: waves 0..3 gather and transform the bundles' first relation group, waves 4..7 the second one (its
      // image has no root block: the own rows it multiplies are the first group's tile, times zeros); the second group's
      // output goes to the bundle's partial tile, the first group's wave adds it (0 + first + second, the order of the
      // group-after-group form) and runs the epilogue
      const float bias0 = a.P[a.off_bias[l] + li], bias1 = a.P[a.off_bias[l] + 16 + li];
      DL_STAMP(3 + (l - 1) * 9);
      fetch(l - 1);
      __syncthreads();                               // planes and both images have landed
      DL_STAMP(4 + (l - 1) * 9);
      f32x4 og[2];
      og[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      og[1] = og[0];
      if (active) {
        const uint32_t rb = (uint32_t)(G2_NR * gw);
        f32x4 acc[G2_NR][2];
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int tstride = 32 * kp >> 1, toff = 16 * kp >> 1;
        const uint32_t* base = PLN + (li * kp >> 1) + 4 * kq;
#pragma unroll 1
        for (int s = 0; s < nks; ++s) {
          const uint2 w = *(const uint2*)(rmo + 32 * s);
          u32x4 pf[2 * G2_NT];
#pragma unroll
          for (int sp = 0; sp < G2_NT; ++sp) {
            pf[2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
            pf[2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s);
          }
#pragma unroll
          for (int r = 0; r < G2_NR; ++r) {
            u32x4 af;
            uint32_t a0, a1, a2, a3;
            g2_expand4<FLAGS>(w.x, rb + (uint32_t)(r + 1), kbit, a0, a1);
            g2_expand4<FLAGS>(w.y, rb + (uint32_t)(r + 1), kbit, a2, a3);
            af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
#pragma unroll
            for (int qq = 0; qq < 2 * G2_NT; ++qq) acc[r][qq & 1] = g2_mfma_bf16(pf[qq], af, acc[r][qq & 1]);
          }
        }
        DL_STAMP(6 + (l - 1) * 9);
        g2_transform(acc, XOc, (const uint32_t*)(gw ? sW2b : sW2), li, kq, og);
        DL_STAMP(7 + (l - 1) * 9);
        if (gw == 1) {
          f32x4* px = (f32x4*)(PXP + bw * 16 * G2_XP) + 2 * lane;
          px[0] = og[0];
          px[1] = og[1];
        }
      }
      __syncthreads();                               // partial outputs in place; both images are free
      if (l < 3) wload(l + 1);
      if (active && gw == 0) {
        const f32x4* px = (const f32x4*)(PXP + bw * 16 * G2_XP) + 2 * lane;
        const f32x4 p0 = px[0], p1 = px[1];
        float v[2][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          v[0][rr] = g2_tanh(((0.f + og[0][rr]) + p0[rr]) + bias0);
          v[1][rr] = g2_tanh(((0.f + og[1][rr]) + p1[rr]) + bias1);
        }
        fwd_out(l, v, XOn);
      }
      DL_STAMP(11 + (l - 1) * 9);
      __syncthreads();                               // planes may be overwritten
      continue;
    }
    stage();
    const float bias0 = a.P[a.off_bias[l] + li], bias1 = a.P[a.off_bias[l] + 16 + li];
    DL_STAMP(3 + (l - 1) * 9);
    fetch(l - 1);
    __syncthreads();
    DL_STAMP(4 + (l - 1) * 9);
    f32x4 o[2];
    o[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o[1] = o[0];
#pragma unroll 1
    for (int grp = 0; grp < ngr; ++grp) {
      const uint32_t rb = (uint32_t)(G2_NR * grp);
      if (grp > 0) {                                 // the next group's image takes the place of the last one
        __syncthreads();
        stage();
        __syncthreads();
      }
      if (grp + 1 < ngr) wpre(l, grp + 1);
      else if (l < 3) wpre(l + 1, 0);
      DL_STAMP(5 + (l - 1) * 9 + 3 * grp);
      if (active) {
        f32x4 acc[G2_NR][2];
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int tstride = 32 * kp >> 1, toff = 16 * kp >> 1;
        const uint32_t* base = PLN + (li * kp >> 1) + 4 * kq;
#pragma unroll 1
        for (int s = 0; s < nks; ++s) {
          const uint2 w = *(const uint2*)(rmo + 32 * s);
          u32x4 pf[2 * G2_NT];
#pragma unroll
          for (int sp = 0; sp < G2_NT; ++sp) {
            pf[2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
            pf[2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s);
          }
#pragma unroll
          for (int r = 0; r < G2_NR; ++r) {
            u32x4 af;
            uint32_t a0, a1, a2, a3;
            g2_expand4<FLAGS>(w.x, rb + (uint32_t)(r + 1), kbit, a0, a1);
            g2_expand4<FLAGS>(w.y, rb + (uint32_t)(r + 1), kbit, a2, a3);
            af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
#pragma unroll
            for (int qq = 0; qq < 2 * G2_NT; ++qq) acc[r][qq & 1] = g2_mfma_bf16(pf[qq], af, acc[r][qq & 1]);
          }
        }
        DL_STAMP(6 + (l - 1) * 9 + 3 * grp);
        f32x4 og[2];
        g2_transform(acc, XOc, (const uint32_t*)sW2, li, kq, og);      // (group > 0: block G2_NR of the image is zero)
        DL_STAMP(7 + (l - 1) * 9 + 3 * grp);
        if (NG == 1) {
          o[0] = og[0];
          o[1] = og[1];
        } else {
          o[0] += og[0];
          o[1] += og[1];
        }
      }
    }
    if (active) {
      float v[2][4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        v[0][rr] = g2_tanh(o[0][rr] + bias0);
        v[1][rr] = g2_tanh(o[1][rr] + bias1);
      }
      fwd_out(l, v, XOn);
    }
    DL_STAMP(11 + (l - 1) * 9);
    __syncthreads();                                 // planes / image may be overwritten
  }
  DL_STAMP(30);
  dlx_seq_done(a.gs_bar, a.self_seq);
}

// ... and the backward of the three conv layers as ONE launch (relation-space tables: see k_dl_layer<*, true, true>): dPre_l
// travels between the members through exchanges 3 and 4, the block rows are staged once, dPre_3 (non-zero on the two target
// rows only) is built in place from the head's output, and nothing but the tables leaves the launch.
struct DlbArgs {
  const int32_t* n_users;
  const int32_t* n_items;
  const int32_t* node_off;
  const uint8_t* node_label;
  const uint8_t* s_lab;          // the arena's slot-based labels [graph][slot]
  int slot;
  const uint8_t* relm;
  const uint8_t* relmT;
  int cap_u, cap_v, relm_ld, relmT_ld, nqu, nqv, R, L, D, kp;
  const float* h[3];             // h_0 .. h_2
  const float* dpre3;            // [N, 32]: the head's dPre_3 (target rows; dense3: every row)
  const float* gfeat;            // [B, D] readout gradient on the target rows
  // dense3 != 0 (sort-pool readout, reference models.py:123-167): dPre_3 is dense -- planes and own rows of layer 3 are staged from
  // dpre3's rows -- and the readout gradient of layers 0..2 arrives per row in dcat[l] [N, 32] instead of gfeat on the target rows
  int dense3;
  const float* dcat[3];
  const float* g2_w;
  const uint16_t* cnt0;
  float* ts_part;
  int ts_stride, slot_stride;
  unsigned long long* ex;
  size_t ex_stride;
  int* gs_bar;
  int* gs_err;
  int timing;
  int B;                         // subgraphs of the batch (the launch is padded to XCD-aligned blocks of workgroups)
  // head != 0: the launch also runs the subgraph's readout + MLP head (head_sub.h) in its set-up -- every workgroup of a subgraph
  // for itself, under the latency of its block-row loads -- instead of a k_head_sub launch in front of it
  int head;
  BatchDev hb;
  ModelDev hm;
  const float* P;
  const uint8_t* inj_mask;
  uint64_t seed, step;
  float mult, grad_scale;
  float* out;
};

// (ng > 1: the T' tiles of FOUR bundles at a time, over the image alone -- the planes stay in place for the next group)
__host__ __device__ static inline int dlb_words(int kp, int ng = 1) {
  const int mid = (G2_NT * 32 * kp >> 1) + G2_WIMG, til = DL_NW * 16 * G2_TP;
  if (ng > 1) return 2 * DL_NW * 16 * G2_XP + DL_NW * 4 * kp + (G2_NT * 32 * kp >> 1) + (DL_NW / 2) * 16 * G2_TP;
  return 2 * DL_NW * 16 * G2_XP + DL_NW * 4 * kp + (mid > til ? mid : til);
}

// NG > 1 (relation groups, g2_image.h): a layer runs group after group -- gather, transform (accumulating dX), T' tiles, the
// group's blocks of the table.  The next group's gather needs the planes again, so here the tiles take the place of the
// IMAGE only: four bundles' tiles at a time (two rounds of the table product for a workgroup with more than four bundles).
// DENSE3: the sort-pool family's dense readout gradient (DlbArgs::dense3; a compile-time switch: as a run-time one it cost the
// centre-node variants 2 us).
// (GS: row / h / partial tiles of DL_GB bundles, their block rows, the planes, both groups' transposed images -- whose space
//  the T' tiles of one group at a time take over)
__host__ __device__ static inline int dlb_words_gs(int kp) {
  const int mid = (G2_NT * 32 * kp >> 1) + 2 * G2_WIMG, til = 2 * DL_GB * 16 * G2_TP;      // (the T' tiles of both groups alias them)
  return 3 * DL_GB * 16 * G2_XP + DL_GB * 4 * kp + (mid > til ? mid : til);
}

template <bool FLAGS, int NG, bool DENSE3, bool GS = false>
__global__ __launch_bounds__(DL_THREADS) void k_dl_bwd(DlbArgs a) {
  __builtin_amdgcn_s_setprio(3);      // (step chain: ahead of the extraction chain's waves wherever the two share a SIMD)
  igmc_kernarg_warm<sizeof(DlbArgs) + 32>();
  static_assert(!GS || (NG == 2 && !DENSE3), "the group split is for two relation groups with the centre-node readout");
  constexpr int HP = (NG == 1) ? G2_XP : DLF_HP2;           // pitch of a row's layer-0 input
  constexpr int C0N = (NG == 1) ? 5 : 11;                   // histogram entries a lane holds: 16 R L / 64
  constexpr int NB = GS ? DL_GB : DL_NW;                    // bundle slots of the workgroup
  IGMC_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  DLX_DECODE(a, bid, g, rem, side, q, nqs);
  if (g >= a.B) return;                          // padding of the launch
  const int cu = a.n_users[g], cv = a.n_items[g];
  const int n_own = side ? cv : cu, n_opp = side ? cu : cv;
  const int R = a.R, L = a.L, RL = R * L, rows0 = RL + L + 1;
  const int ngr = (NG == 1) ? 1 : g2_rel_groups(R);      // groups that hold relations
  const int ts = a.ts_stride;
  const size_t slot = (size_t)g + (size_t)rem * a.slot_stride;
  float* part0 = a.ts_part + slot * ts;
  if (a.timing && tid == 0 && bid < 1024) {        // (debug aid: start / end of every workgroup on the wall clock)
    g_g2_wg[bid][0] = g2_wall_clock();
    g_g2_wg[bid][1] = 0ull;
    g_g2_wg[bid][2] = ((unsigned long long)n_own << 32) | (unsigned long long)n_opp;
  }
  const DlRows dr = dl_rows(n_own, nqs, q);
  if (dr.nact == 0) {                            // nothing of this side in the workgroup's rows: all-zero partial tables
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 1; l < 4; ++l) {
      float* wp = a.ts_part + ((size_t)l * IGMC_TS_BLOCKS + slot) * ts;
      for (int i = tid; i < (ts >> 2); i += DL_THREADS) ((float4*)wp)[i] = z4;
    }
    for (int i = tid; i < rows0 * 8; i += DL_THREADS) ((float4*)part0)[i] = z4;
    return;
  }
  DL_STAMP(40);
  const uint32_t seq = g2_ld_seq(a.gs_bar);
  const uint32_t tag0 = seq * 8u + 1u;
  auto xtag = [&](int x) { return tag0 + (uint32_t)x; };      // flag value of exchange x of this launch (never 0)
  const int nb = a.node_off[g];
  const int own0 = nb + (side ? cu : 0), opp0 = nb + (side ? 0 : cu);
  const int kp = a.kp, rmp = kp;
  const int nks = (n_opp + 31) >> 5;
  const int npad_opp = ((n_opp + 15) >> 4) << 4;
  float* XOA = (float*)smem;                                              // [NB][16][G2_XP] dPre_l of the rows
  float* HSA = XOA + NB * 16 * G2_XP;                                     // [NB][16][G2_XP] h_{l-1} of the rows
  float* sbias = HSA;                                                     // (d bias scratch: dead before the h rows land)
  float* PXP = HSA + NB * 16 * G2_XP;                                     // GS: [NB][16][G2_XP] the second group's partial dX
  unsigned char* RMW = (unsigned char*)(HSA + (GS ? 2 : 1) * NB * 16 * G2_XP);      // [NB][16][rmp] bytes
  uint32_t* PLN = (uint32_t*)(RMW + NB * 16 * rmp);                       // [3][32][kp] bf16
  float2* sW2 = (float2*)(PLN + (G2_NT * 32 * kp >> 1));                  // [G2_WIMG words]
  float2* sW2b = sW2 + G2_WIMG / 2;                                       // GS: the second group's image
  // T' tiles: [DL_NW][16][G2_TP] over planes + image (NG = 1) / [DL_NW / 2][16][G2_TP] over the image (NG > 1; GS: both images)
  float* TIL = (NG == 1) ? (float*)PLN : (float*)sW2;
  const int bw0 = GS ? (wave & (DL_GB - 1)) : wave;       // bundle of the wave, and (GS) the relation group it takes
  const int gw0 = GS ? wave / DL_GB : 0;
  const bool lead0 = !GS || gw0 == 0;
  const int row0 = dr.base + 16 * bw0;
  const bool active = bw0 < dr.nact;
  const size_t exs = a.ex_stride;
  // plane exchange regions (g2_prims.h: bf16 term planes + a flag per bundle, through the XCD's L2)
  auto px_of = [&](int x, int sd) { return (unsigned char*)(a.ex + (size_t)x * a.ex_stride + ((size_t)g * 2 + sd) * (32 * DLX_K)); };

  // the loss head's loads (features of the target rows, the wave's lin1 rows) go out first: over by the time the head starts
  HeadPre hpre;
  if (!DENSE3 && a.head) head_sub_prefetch(hpre, a.hb, a.hm, a.P, g, tid, nb, nb + cu);
  // ---- staging: block rows, the target rows of dPre_3, the layer-0 inputs of the rows (layer-0 table gradient), image 3
  const int ldb = side ? a.relmT_ld : a.relm_ld, ldw = ldb >> 2;
  const uint8_t* rsrc = side ? a.relmT + (size_t)g * a.cap_v * a.relmT_ld : a.relm + (size_t)g * a.cap_u * a.relm_ld;
  const int rw = rmp >> 2;
  // (r = i / rw for i < 17 * 64 without a division per element: the quotient by a 20-bit reciprocal is exact there)
  const uint32_t rw_magic = ((1u << 20) + (uint32_t)rw - 1u) / (uint32_t)rw;
  uint32_t rmq[DL_RIT];
#pragma unroll
  for (int u = 0; u < DL_RIT; ++u) {
    const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
    const int rc = row0 + r < n_own ? row0 + r : n_own - 1, cc = c < ldw ? c : ldw - 1;
    rmq[u] = lead0 ? ((const uint32_t*)(rsrc + (size_t)rc * ldb))[cc] : 0u;
  }
  // (the rows' neighbour-label histograms -- the layer-0 table's inputs -- are requested under the LAST table product: held
  //  from here they cost registers, and spills, through all three layers)
  uint16_t c0q[C0N];
#pragma unroll
  for (int u = 0; u < C0N; ++u) c0q[u] = 0;
  const int own_lab = (int)a.s_lab[(size_t)g * a.slot + (side ? a.cap_u : 0) + (row0 + li < n_own ? row0 + li : n_own - 1)];
  constexpr int NWQ = (G2_WIMG / 4 + DL_THREADS - 1) / DL_THREADS;
  f32x4 wq[NWQ];                                  // a layer's transposed weight image: requested at the top of the layer,
  auto wpre = [&](int l, int grp) {               // stored behind the exchange poll (held across a layer it costs 38 spills)
    const f32x4* src = (const f32x4*)(a.g2_w + g2_img_off(NG, l, 1, grp));
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
      const int i = tid + u * DL_THREADS;
      wq[u] = src[i < G2_WIMG / 4 ? i : G2_WIMG / 4 - 1];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < NWQ; ++u) {
      const int i = tid + u * DL_THREADS;
      if (i < G2_WIMG / 4) ((f32x4*)sW2)[i] = wq[u];
    }
  };
  // GS: the transposed image of layer l for THIS wave's relation group, global -> LDS directly, 1 KB pieces dealt to the four
  // waves of the group; requested when the images' space is free, landed by the barrier in front of its first use
  auto wload = [&](int l) {
    const float4* src = (const float4*)(a.g2_w + g2_img_off(NG, l, 1, gw0));
    float4* dst = (float4*)(gw0 ? sW2b : sW2);
#pragma unroll
    for (int j = 0; j < (G2_WIMG / 256 + DL_GB - 1) / DL_GB; ++j) {
      const int c = bw0 + j * DL_GB;
      if (c < G2_WIMG / 256) g2_glds16(src + c * 64, dst + c * 64, lane);
    }
  };
  G2_SCHED_BARRIER();
  {   // planes (dPre_3: node 0 of the opposite side only; k-steps past the published rows read zeros later) and row tiles
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < (G2_NT * 32 * kp >> 3); i += DL_THREADS) ((float4*)PLN)[i] = z4;
    for (int i = tid; i < 2 * NB * 16 * G2_XP / 4; i += DL_THREADS) ((float4*)XOA)[i] = z4;
  }
  if (lead0) {
    uint32_t* dst = (uint32_t*)(RMW + (size_t)bw0 * 16 * rmp);
#pragma unroll
    for (int u = 0; u < DL_RIT; ++u) {
      const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
      if (i < 16 * rw) dst[i] = (row0 + r < n_own && c < ldw && 4 * c < 32 * nks) ? rmq[u] : 0u;
    }
  }
  __syncthreads();
  if (!DENSE3 && a.head) {   // loss head of the subgraph in the image's space (free until the first layer stages its image); what it leaves
                  // in HBM -- dPre_3 of the target rows, d feat -- is read back below by this very workgroup (barrier in between)
    const uint64_t hstep = a.hm.ctrl ? (uint64_t)a.hm.ctrl[IGMC_CTRL_STEP] : a.step;
    head_sub_compute<true>(hpre, a.hb, a.hm, a.P, g, tid, nb, nb + cu, (float*)sW2, a.inj_mask, a.seed, hstep, a.mult, a.grad_scale, a.out);
    __syncthreads();
  }
  if (DENSE3) {
    // dPre_3 of every row: the opposite side's rows as bf16 planes (a thread: two nodes x four features, as k_dl_layer stages
    // its input), the bundles' own rows into their tiles
    const int tstride = 32 * kp >> 1;
#pragma unroll
    for (int u = 0; u < DL_PIT; ++u) {
      const int i = tid + u * DL_THREADS, jp = i >> 3, fq = i & 7;
      if (i >= 16 * nks * 8) continue;
      const bool k0 = 2 * jp < n_opp, k1 = 2 * jp + 1 < n_opp;
      const int j0 = k0 ? 2 * jp : n_opp - 1, j1 = k1 ? 2 * jp + 1 : n_opp - 1;
      const float4 x0 = *(const float4*)(a.dpre3 + (size_t)(opp0 + j0) * 32 + 4 * fq);
      const float4 x1 = *(const float4*)(a.dpre3 + (size_t)(opp0 + j1) * 32 + 4 * fq);
      const float v0[4] = {x0.x, x0.y, x0.z, x0.w}, v1[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t h, mi, lo;
        g2_split2(k0 ? v0[c] : 0.f, k1 ? v1[c] : 0.f, h, mi, lo);
        uint32_t* pp = PLN + ((4 * fq + c) * kp >> 1) + jp;
        pp[0] = h;
        pp[tstride] = mi;
        pp[2 * tstride] = lo;
      }
    }
    if (active) {
      float* XOw = XOA + wave * 16 * G2_XP;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = lane + 64 * u, r = i >> 3, c4 = i & 7;
        if (row0 + r < n_own) *(float4*)(XOw + r * G2_XP + 4 * c4) = *(const float4*)(a.dpre3 + (size_t)(own0 + row0 + r) * 32 + 4 * c4);
      }
    }
  }
  const float d3 = (!DENSE3 && tid < 64) ? a.dpre3[(size_t)((tid >> 5) ? own0 : opp0) * 32 + (tid & 31)] : 0.f;   // opposite | own target row
  if (DENSE3) {
  } else if (tid < 32) {                           // dPre_3 of the opposite side's target node: the three terms of node 0
    uint32_t h, mi, lo;
    g2_split2(d3, 0.f, h, mi, lo);
    uint32_t* p2 = PLN + (tid * kp >> 1);
    p2[0] = h & 0xFFFFu;
    p2[32 * kp >> 1] = mi & 0xFFFFu;
    p2[2 * (32 * kp >> 1)] = lo & 0xFFFFu;
  } else if (tid < 64 && q == 0) {
    XOA[tid & 31] = d3;                            // own target row = row 0 of the side's first bundle (workgroup 0's)
  }
  // (no barrier: the layer loop starts with one)

  const unsigned char* rmo = RMW + (size_t)(bw0 * 16 + li) * rmp + 8 * kq;
  const int kbit = side ? IGMC_RELM_KF : IGMC_RELM_KT;                   // keep bit of the edge own -> opposite
  const int nact = dr.nact;                        // bundles of this workgroup that hold rows
  float* XO = XOA + bw0 * 16 * G2_XP;
  float* T = TIL + ((NG == 1) ? wave : (wave & 3)) * 16 * G2_TP;
  float* HS = HSA + bw0 * 16 * G2_XP;
  float dv[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dv[nt][rr] = 0.f;

#pragma unroll 1
  for (int l = 3; l >= 1; --l) {
    float* wpart = a.ts_part + ((size_t)l * IGMC_TS_BLOCKS + slot) * ts;
    float xprev[2][4], addv[2][4];
    f32x4 o[2];
    o[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o[1] = o[0];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        xprev[nt][rr] = 0.f;
        addv[nt][rr] = 0.f;
      }
    if (active && lead0) {   // h_{l-1} of the rows and the readout gradient: requested ahead of the exchange poll / the image
      int lane_ = lane;
      G2_OPAQUE(lane_);
      const int li_ = lane_ & 15, kq_ = lane_ >> 4;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int rw2 = row0 + 4 * kq_ + rr, rwc = rw2 < n_own ? rw2 : n_own - 1;
          const float hv = a.h[l - 1][(size_t)(own0 + rwc) * 32 + 16 * nt + li_];
          xprev[nt][rr] = (rw2 < n_own) ? hv : 0.f;      // (rows past the side: zero K entries of the table product)
          if (DENSE3) addv[nt][rr] = (rw2 < n_own) ? a.dcat[l - 1][(size_t)(own0 + rwc) * 32 + 16 * nt + li_] : 0.f;
          else addv[nt][rr] = (rw2 == 0) ? a.gfeat[(size_t)g * a.D + side * 128 + (l - 1) * 32 + 16 * nt + li_] : 0.f;
        }
    }
    if constexpr (GS) {
      // ---- both relation groups at once: waves 0..3 gather and transform the bundles' first group, waves 4..7 the second one
      //      (its image has no root block: the dPre_l rows it multiplies meet zeros); the second group's partial dX goes to the
      //      bundle's partial tile, the first group's wave adds it (0 + first + second: the order of the group-after-group
      //      form) and runs the epilogue.  The table products follow group after group on all eight waves -- the second
      //      group's first: its waves lay down their T' tiles while the leaders are in their epilogue.
      int tid_g = threadIdx.x;
      G2_OPAQUE(tid_g);
      const int tid = tid_g, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
      const int bw = wave & (DL_GB - 1), gw = wave / DL_GB;
      const int row0 = dr.base + 16 * bw;
      const bool active = bw < nact;
      const unsigned char* rmo = RMW + (size_t)(bw * 16 + li) * rmp + 8 * kq;
      float* XO = XOA + bw * 16 * G2_XP;
      float* HS = HSA + bw * 16 * G2_XP;
      const int sk = 42 + (3 - l) * NG * 7;          // (phase clocks)
      DL_STAMP(sk);
      wload(l);
      if (l < 3) {       // the opposite side's dPre_l: flags of its bundles, then global -> LDS (landed by the barrier)
        g2_flags_wait(px_of(5 - l, 1 - side), (n_opp + 15) >> 4, xtag(5 - l), lane, a.gs_err, DLX_PX_FLAGS);
        g2_planes_load_exact(PLN, px_of(5 - l, 1 - side), 192 * kp, wave, lane, DL_NW);
      }
      __syncthreads();                               // planes, both images, dPre_l of the rows are in place
      DL_STAMP(sk + 1);
      {   // d bias_l = column sums of dPre_l over the workgroup's rows (fixed order)
        {
          const int n = tid & 31, part = tid >> 5;
          float sb = 0.f;
          for (int row = part; row < NB * 16; row += DL_THREADS / 32) sb += XOA[(row >> 4) * 16 * G2_XP + (row & 15) * G2_XP + n];
          sbias[part * 32 + n] = sb;
        }
        __syncthreads();
        if (tid < 32) {
          float s2 = 0.f;
#pragma unroll
          for (int p2 = 0; p2 < DL_THREADS / 32; ++p2) s2 += sbias[p2 * 32 + tid];
          wpart[(R * 32 + 32) * 32 + tid] = s2;
        }
      }
      DL_STAMP(sk + 2);
      f32x4 acc[G2_NR][2];
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) {
        acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      f32x4 og[2];
      og[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      og[1] = og[0];
      if (active) {
        const uint32_t rb = (uint32_t)(G2_NR * gw);
        const int tstride = 32 * kp >> 1, toff = 16 * kp >> 1;
        const uint32_t* base = PLN + (li * kp >> 1) + 4 * kq;
        const int nke = (l == 3) ? 1 : nks;          // dPre_3 of the centre-node readout lives on node 0: one k-step
#pragma unroll 1
        for (int s2 = 0; s2 < nke; ++s2) {
          const uint2 w = *(const uint2*)(rmo + 32 * s2);
          u32x4 pf[2 * G2_NT];
#pragma unroll
          for (int sp = 0; sp < G2_NT; ++sp) {
            pf[2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s2);
            pf[2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s2);
          }
#pragma unroll
          for (int r = 0; r < G2_NR; ++r) {
            u32x4 af;
            uint32_t a0, a1, a2, a3;
            g2_expand4<FLAGS>(w.x, rb + (uint32_t)(r + 1), kbit, a0, a1);
            g2_expand4<FLAGS>(w.y, rb + (uint32_t)(r + 1), kbit, a2, a3);
            af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
#pragma unroll
            for (int qq = 0; qq < 2 * G2_NT; ++qq) acc[r][qq & 1] = g2_mfma_bf16(pf[qq], af, acc[r][qq & 1]);
          }
        }
        DL_STAMP(sk + 3);
        g2_transform(acc, XO, (const uint32_t*)(gw ? sW2b : sW2), li, kq, og);
        if (gw == 1) {
          f32x4* px = (f32x4*)(PXP + bw * 16 * G2_XP) + 2 * lane;
          px[0] = og[0];
          px[1] = og[1];
        }
      }
      DL_STAMP(sk + 4);
      __syncthreads();                               // partial dX in place; every wave is done with planes / images
      DL_STAMP(sk + 5);
      float* T8 = (float*)PLN + (gw * NB + bw) * 16 * G2_TP;      // the T' tile of (group, bundle): over planes + images
      if (active && gw == 1) {                       // (laid down while the leaders are in their epilogue)
#pragma unroll
        for (int r = 0; r < G2_NR; ++r)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *(float4*)(T8 + li * G2_TP + r * 32 + 16 * t + 4 * kq) = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
      }
      if (active && gw == 0) {
        const f32x4* px = (const f32x4*)(PXP + bw * 16 * G2_XP) + 2 * lane;
        const f32x4 p0 = px[0], p1 = px[1];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const bool ok = row0 + 4 * kq + rr < n_own;
            const float x = xprev[nt][rr];
            const float ov = (0.f + og[nt][rr]) + (nt ? p1[rr] : p0[rr]);
            dv[nt][rr] = ok ? (ov + addv[nt][rr]) * (1.f - x * x) : 0.f;
          }
          if (l > 1) g2_publish_planes(px_of(6 - l, side), kp, 16 * nt + li, row0 + 4 * kq, dv[nt]);
        }
        if (l > 1) g2_flag_raise(px_of(6 - l, side), row0 >> 4, xtag(6 - l), lane, DLX_PX_FLAGS);      // the bundle's dPre_{l-1} is in the L2
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) HS[(4 * kq + rr) * G2_XP + 16 * nt + li] = xprev[nt][rr];
        if (l == 1) {
          int lane_ = lane;                          // (opaque: the address arithmetic stays here, not above the layer loop)
          G2_OPAQUE(lane_);
#pragma unroll
          for (int u = 0; u < C0N; ++u) {
            const int i = lane_ + 64 * u, r = i / RL, c = i - r * RL;
            const int rc = (r < 16 && row0 + r < n_own) ? row0 + r : n_own - 1;
            c0q[u] = a.cnt0[(size_t)(own0 + rc) * RL + (r < 16 ? c : 0)];
          }
        }
      }
      // ---- the tables h_{l-1}^T [T'_0 .. T'_4 | dPre_l] of BOTH groups in one pass: the T' tiles of the two groups x four
      //      bundles take the place of planes + images (all dead: the next layer's fetch rewrites the whole plane image);
      //      wave = (row half, three column tiles) of either group's table -- six independent accumulators
      if (active && gw == 0) {
#pragma unroll
        for (int r = 0; r < G2_NR; ++r)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *(float4*)(T8 + li * G2_TP + r * 32 + 16 * t + 4 * kq) = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
      }
      __syncthreads();                               // tiles and h rows are in place
      {
        f32x4 w3[2][3];
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) w3[gq][i3] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int m2w = wave >> 2, nt0 = 3 * (wave & 3);
#pragma unroll 1
        for (int wb = 0; wb < nact; ++wb) {
          const float* Tb0 = (const float*)PLN + wb * 16 * G2_TP;
          const float* Tb1 = (const float*)PLN + (NB + wb) * 16 * G2_TP;
          const float* Hb = HSA + wb * 16 * G2_XP;
          const float* Db = XOA + wb * 16 * G2_XP;
          float av[4], bwv[2][4][3];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            av[s4] = Hb[(4 * s4 + kq) * G2_XP + m2w * 16 + li];
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) {
              const int nt = nt0 + i3;
              bwv[0][s4][i3] = (nt < 2 * G2_NR) ? Tb0[(4 * s4 + kq) * G2_TP + nt * 16 + li]
                                                : Db[(4 * s4 + kq) * G2_XP + (nt - 2 * G2_NR) * 16 + li];
              bwv[1][s4][i3] = (nt < 2 * G2_NR) ? Tb1[(4 * s4 + kq) * G2_TP + nt * 16 + li] : 0.f;
            }
          }
          G2_SCHED_BARRIER();
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) {
              w3[0][i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bwv[0][s4][i3], w3[0][i3], 0, 0, 0);
              if (nt0 + i3 < 2 * G2_NR) w3[1][i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bwv[1][s4][i3], w3[1][i3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) {
            const int nt = nt0 + i3, r = nt >> 1;       // 32-column block: relation 5 gq + r, or G2_NR = dPre_l (d root: group 0)
            const int rg = G2_NR * gq + r;
            if (r < G2_NR ? rg >= R : gq > 0) continue;
            float* pp = wpart + (kq * 4) * 32 + li + (r < G2_NR ? rg : R) * 1024 + m2w * 512 + (nt & 1) * 16;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = w3[gq][i3][rr];
          }
      }
      __syncthreads();                               // tiles, h rows and dPre_l are consumed
      DL_STAMP(sk + 6);
      if (l > 1 && active && gw == 0) {              // dPre_{l-1} of the rows becomes the next layer's own rows (the d root
#pragma unroll                                       // block of group 0's product has read dPre_l)
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) XO[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
      }
      continue;
    }
#pragma unroll 1
    for (int grp = 0; grp < ngr; ++grp) {
      const uint32_t rb = (uint32_t)(G2_NR * grp);
      const int sk = 42 + ((3 - l) * NG + grp) * 7;      // (phase clocks)
      // per-lane indices re-derived from an opaque copy of the thread index: what is computed from them stays inside the
      // group's pass (hoisted above the layer loop, the loop-invariant addresses occupy -- and spill -- registers throughout)
      int tid_g = threadIdx.x;
      G2_OPAQUE(tid_g);
      const int tid = tid_g, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
      const int row0 = dr.base + 16 * wave;
      const bool active = wave < nact;
      const unsigned char* rmo = RMW + (size_t)(wave * 16 + li) * rmp + 8 * kq;
      float* XO = XOA + wave * 16 * G2_XP;
      float* T = TIL + ((NG == 1) ? wave : (wave & 3)) * 16 * G2_TP;
      float* HS = HSA + wave * 16 * G2_XP;
      DL_STAMP(sk);
      wpre(l, grp);
      if (l < 3 && grp == 0) {       // the opposite side's dPre_l: flags of its bundles, then global -> LDS (landed by the barrier)
        g2_flags_wait(px_of(5 - l, 1 - side), (n_opp + 15) >> 4, xtag(5 - l), lane, a.gs_err, DLX_PX_FLAGS);
        g2_planes_load_exact(PLN, px_of(5 - l, 1 - side), 192 * kp, wave, lane, DL_NW);
      }
      stage();
      __syncthreads();                               // planes, image, dPre_l of the rows are in place
      DL_STAMP(sk + 1);
      if (grp == 0) {   // d bias_l = column sums of dPre_l over the workgroup's rows (fixed order)
        {
          const int n = tid & 31, part = tid >> 5;
          float sb = 0.f;
          for (int row = part; row < DL_NW * 16; row += DL_THREADS / 32) sb += XOA[(row >> 4) * 16 * G2_XP + (row & 15) * G2_XP + n];
          sbias[part * 32 + n] = sb;
        }
        __syncthreads();
        if (tid < 32) {
          float s2 = 0.f;
#pragma unroll
          for (int p = 0; p < DL_THREADS / 32; ++p) s2 += sbias[p * 32 + tid];
          wpart[(R * 32 + 32) * 32 + tid] = s2;
        }
      }
      DL_STAMP(sk + 2);
      f32x4 acc[G2_NR][2];
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) {
        acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (active) {
        const int tstride = 32 * kp >> 1, toff = 16 * kp >> 1;
        const uint32_t* base = PLN + (li * kp >> 1) + 4 * kq;
        const int nke = (l == 3 && !DENSE3) ? 1 : nks;      // dPre_3 of the centre-node readout lives on node 0: one k-step
#pragma unroll 1
        for (int s = 0; s < nke; ++s) {
          const uint2 w = *(const uint2*)(rmo + 32 * s);
          u32x4 pf[2 * G2_NT];
#pragma unroll
          for (int sp = 0; sp < G2_NT; ++sp) {
            pf[2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
            pf[2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s);
          }
#pragma unroll
          for (int r = 0; r < G2_NR; ++r) {
            u32x4 af;
            uint32_t a0, a1, a2, a3;
            g2_expand4<FLAGS>(w.x, rb + (uint32_t)(r + 1), kbit, a0, a1);
            g2_expand4<FLAGS>(w.y, rb + (uint32_t)(r + 1), kbit, a2, a3);
            af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
#pragma unroll
            for (int qq = 0; qq < 2 * G2_NT; ++qq) acc[r][qq & 1] = g2_mfma_bf16(pf[qq], af, acc[r][qq & 1]);
          }
        }
        DL_STAMP(sk + 3);
        f32x4 og[2];
        g2_transform(acc, XO, (const uint32_t*)sW2, li, kq, og);       // (group > 0: block G2_NR of the image is zero)
        if (NG == 1) {
          o[0] = og[0];
          o[1] = og[1];
        } else {
          o[0] += og[0];
          o[1] += og[1];
        }
        if (grp == ngr - 1) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const bool ok = row0 + 4 * kq + rr < n_own;
              const float x = xprev[nt][rr];
              dv[nt][rr] = ok ? (o[nt][rr] + addv[nt][rr]) * (1.f - x * x) : 0.f;
            }
            if (l > 1) g2_publish_planes(px_of(6 - l, side), kp, 16 * nt + li, row0 + 4 * kq, dv[nt]);
            if (NG > 1 && grp > 0 && l > 1) {        // dPre_{l-1} of the rows becomes the next layer's own rows: nobody reads
#pragma unroll                                       // this tile any more (the d root block belongs to group 0's product)
              for (int rr = 0; rr < 4; ++rr) XO[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
            }
          }
          if (l > 1) g2_flag_raise(px_of(6 - l, side), row0 >> 4, xtag(6 - l), lane, DLX_PX_FLAGS);      // the bundle's dPre_{l-1} is in the L2
        }
      }
      DL_STAMP(sk + 4);
      __syncthreads();                               // every wave is done with planes / image: the T' tiles take their space
      DL_STAMP(sk + 5);
      if (l == 1 && grp == ngr - 1) {
        int lane_ = lane;                            // (opaque: the address arithmetic stays here, not above the layer loop)
        G2_OPAQUE(lane_);
#pragma unroll
        for (int u = 0; u < C0N; ++u) {
          const int i = lane_ + 64 * u, r = i / RL, c = i - r * RL;
          const int rc = (r < 16 && row0 + r < n_own) ? row0 + r : n_own - 1;
          c0q[u] = a.cnt0[(size_t)(own0 + rc) * RL + (r < 16 ? c : 0)];
        }
      }
      if (active && grp == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) HS[(4 * kq + rr) * G2_XP + 16 * nt + li] = xprev[nt][rr];
      }
      {   // table h_{l-1}^T [T'_0 .. T'_4 | dPre_l] of the group: 2 x 12 output tiles over the 8 waves, K = the active bundles' rows
        f32x4 w3[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) w3[i3] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int m2w = wave >> 2, nt0 = 3 * (wave & 3);
        constexpr int TB = (NG == 1) ? DL_NW : DL_NW / 2;      // bundles whose tiles are in LDS at a time
#pragma unroll 1
        for (int wb0 = 0; wb0 < nact; wb0 += TB) {
          if (wb0 > 0) __syncthreads();              // (the last round's tiles are consumed)
          if (active && wave >= wb0 && wave < wb0 + TB) {
#pragma unroll
            for (int r = 0; r < G2_NR; ++r)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                *(float4*)(T + li * G2_TP + r * 32 + 16 * t + 4 * kq) = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
          }
          __syncthreads();
          const int wb1 = (wb0 + TB < nact) ? wb0 + TB : nact;
#pragma unroll 1
          for (int wb = wb0; wb < wb1; ++wb) {
            const float* Tb = TIL + (wb - wb0) * 16 * G2_TP;
            const float* Hb = HSA + wb * 16 * G2_XP;
            const float* Db = XOA + wb * 16 * G2_XP;
            float av[4], bw[4][3];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              av[s4] = Hb[(4 * s4 + kq) * G2_XP + m2w * 16 + li];
#pragma unroll
              for (int i3 = 0; i3 < 3; ++i3) {
                const int nt = nt0 + i3;
                bw[s4][i3] = (nt < 2 * G2_NR) ? Tb[(4 * s4 + kq) * G2_TP + nt * 16 + li]
                                              : Db[(4 * s4 + kq) * G2_XP + (nt - 2 * G2_NR) * 16 + li];
              }
            }
            G2_SCHED_BARRIER();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
              for (int i3 = 0; i3 < 3; ++i3) w3[i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bw[s4][i3], w3[i3], 0, 0, 0);
          }
        }
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
          const int nt = nt0 + i3, r = nt >> 1;         // 32-column block: relation rb + r, or G2_NR = dPre_l (d root: group 0)
          const int rg = (int)rb + r;
          if (r < G2_NR ? rg >= R : grp > 0) continue;
          float* pp = wpart + (kq * 4) * 32 + li + (r < G2_NR ? rg : R) * 1024 + m2w * 512 + (nt & 1) * 16;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = w3[i3][rr];
        }
      }
      DL_STAMP(sk + 6);
      __syncthreads();                               // tiles, h rows and dPre_l are consumed
      if (NG == 1) {
        if (l > 1) {
          // dPre_{l-1} of the rows becomes the next layer's own rows.  (The planes' space held the tiles: the next layer's
          // fetch copies the WHOLE plane image of its exchange region -- 192 kp bytes, g2_planes_load_exact -- over it; a zero
          // fill + barrier here, from the days of row-by-row reloads, cost 2.5 k cycles a layer.)
          if (active) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) XO[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
          }
        }
      } else if (ngr == 1 && l > 1) {                // (one group that holds relations: its product has just read dPre_l)
        if (active) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) XO[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
        }
      }                                              // (else: planes untouched, the rows' dPre_{l-1} written by the epilogue)
    }
  }
  DL_STAMP(41);
  // ---- layer-0 table gradient T0'[c][f] = sum_i [hist | onehot | 1](i, c) dPre_0[i][f] over the workgroup's rows
  {
    float* L0 = (float*)PLN;                         // (planes and image are dead: the rows' inputs and dPre_0 as tiles)
    float* HI = L0 + bw0 * 16 * HP;
    float* D0 = L0 + NB * 16 * HP + bw0 * 16 * G2_XP;
    const bool active = bw0 < nact && lead0;
    if (active) {
      for (int i = lane; i < 16 * HP; i += 64) HI[i] = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) D0[(4 * kq + rr) * G2_XP + 16 * nt + li] = dv[nt][rr];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int u = 0; u < C0N; ++u) {
        const int i = lane + 64 * u, r = i / RL, c = i - r * RL;
        if (r < 16 && row0 + r < n_own) HI[r * HP + c] = (float)c0q[u];
      }
      if (kq == 0 && row0 + li < n_own) {
        HI[li * HP + RL + own_lab] = 1.f;
        HI[li * HP + RL + L] = 1.f;
      }
    }
    __syncthreads();
    if (wave < 4 * NG) {                             // (code rows m2 * 16 .., feature half wn): 32 / 64 table rows
      const int m2 = wave >> 1, wn = wave & 1;
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int wb = 0; wb < nact; ++wb) {
        const float* Hb = L0 + wb * 16 * HP;
        const float* Db = L0 + NB * 16 * HP + wb * 16 * G2_XP;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Hb[(4 * s4 + kq) * HP + m2 * 16 + li],
                                                      Db[(4 * s4 + kq) * G2_XP + wn * 16 + li], acc0, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int c = m2 * 16 + kq * 4 + rr;
        if (c < rows0) part0[c * 32 + wn * 16 + li] = acc0[rr];
      }
    }
  }
  DL_STAMP(126);
  if (a.timing && tid == 0 && bid < 1024) g_g2_wg[bid][1] = g2_wall_clock();
}

// Training head of the dense per-layer path, ONE workgroup per subgraph (k_graph_step2's head as a launch of its own): the
// 256 conv features of the two target rows -> lin1 / ReLU / dropout / lin2 / residual -> dz, d feat and dPre_3 on the
// target rows.  (k_head_train's head role takes 16 subgraphs per workgroup on the f32 matrix cores: four workgroups at
// batch 50, 19 us of dependent round trips; 50 workgroups of one subgraph each are done in a third of that.)  Side features
// (--use-features, reference models.py:208-209) ride along: D = 256 + S, the body is head_sub.h.
__global__ __launch_bounds__(512) void k_head_sub(BatchDev b, ModelDev m, const float* __restrict__ P,
                                                    const uint8_t* __restrict__ inj_mask, uint64_t seed, uint64_t step_arg,
                                                    float mult, float grad_scale, float* __restrict__ out) {
  igmc_kernarg_warm<sizeof(BatchDev) + sizeof(ModelDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const int g = blockIdx.x;
  if (g >= b.totals[3]) return;
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : step_arg;
  const int nu = b.node_off[g], nv = nu + b.n_users[g];
  HeadPre hp;
  head_sub_prefetch(hp, b, m, P, g, (int)threadIdx.x, nu, nv);
  head_sub_compute<true>(hp, b, m, P, g, (int)threadIdx.x, nu, nv, (float*)smem, inj_mask, seed, step, mult, grad_scale, out);
}

void igmc_launch_head_sub(const ModelDev& m, const BatchDev& b, const float* P, int B, const uint8_t* inj_mask, uint64_t seed,
                          uint64_t step, float mult, float grad_scale, float* out, void* stream) {
  IGMC_PLAUNCH("k_head_sub", k_head_sub, B, 512, (size_t)igmc_head_sub_lds_floats(m.D) * sizeof(float), stream, b, m, P, inj_mask,
               seed, step, mult, grad_scale, out);
}

// Layer 0 of the dense per-layer path (one-hot input): per row the histogram of (relation, label of the neighbour) over
// its kept in-edges -- one-hot label planes x the relation masks on the matrix cores, as in k_graph_step2 -- then
// h_0 = tanh([hist | onehot(own label) | 1] @ T0) with the composed layer-0 table (k_g2_compose).  STORE (training): the
// histogram also goes to cnt0[node][R * L] (uint16), what the layer-0 weight gradient is formed from.  No edge list.
struct Dl0Args {
  const int32_t* n_users;
  const int32_t* n_items;
  const int32_t* node_off;
  const uint8_t* node_label;
  const uint8_t* relm;
  const uint8_t* relmT;
  int cap_u, cap_v, relm_ld, relmT_ld, nqu, nqv, R, L, kp;
  const float* t0;         // composed layer-0 table [32][32]
  float* out;              // h_0 [N, 32]
  uint16_t* cnt0;          // [N, R * L] or NULL
};

template <bool FLAGS, bool STORE>
__global__ __launch_bounds__(DL_THREADS) void k_dl_layer0(Dl0Args a) {
  igmc_kernarg_warm<sizeof(Dl0Args) + 32>();
  IGMC_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  DL_DECODE(a, bid, g, rem, side, q, nqs);
  const int cu = a.n_users[g], cv = a.n_items[g];
  const int n_own = side ? cv : cu, n_opp = side ? cu : cv;
  const DlRows dr = dl_rows(n_own, nqs, q);
  if (dr.nact == 0) return;
  const int R = a.R, L = a.L, RL = R * L;
  const int nb = a.node_off[g];
  const int own0 = nb + (side ? cu : 0), opp0 = nb + (side ? 0 : cu);
  const int kp = a.kp, rmp = kp;
  const int nks = (n_opp + 31) >> 5;
  uint32_t* OHP = (uint32_t*)smem;                                       // [8 labels][kp] bf16 one-hot planes
  unsigned char* RMW = (unsigned char*)(OHP + (8 * kp >> 1));            // [DL_NW][16][rmp] bytes
  float* HI = (float*)(RMW + DL_NW * 16 * rmp);                          // [DL_NW][16][G2_XP] input rows [hist | onehot | 1]
  float* sT0 = HI + DL_NW * 16 * G2_XP;                                  // [32][32]
  const int row0 = dr.base + 16 * wave;
  const bool active = wave < dr.nact;
  for (int i = tid; i < 256; i += DL_THREADS) ((float4*)sT0)[i] = ((const float4*)a.t0)[i];
  // one thread per node pair of the opposite side (<= 160): its two labels once, the eight plane words from them
  const int own_lab = (row0 + li < n_own) ? (int)a.node_label[own0 + row0 + li] : 0;      // (used after the gather)
  for (int jp = tid; jp < 16 * nks; jp += DL_THREADS) {
    const int l0 = (2 * jp < n_opp) ? (int)a.node_label[opp0 + 2 * jp] : 255;
    const int l1 = (2 * jp + 1 < n_opp) ? (int)a.node_label[opp0 + 2 * jp + 1] : 255;
#pragma unroll
    for (int lb = 0; lb < 8; ++lb) OHP[(lb * kp >> 1) + jp] = ((l0 == lb) ? 0x3F80u : 0u) | ((l1 == lb) ? 0x3F800000u : 0u);
  }
  {
    const int ldb = side ? a.relmT_ld : a.relm_ld, ldw = ldb >> 2;
    const uint8_t* src = side ? a.relmT + (size_t)g * a.cap_v * a.relmT_ld : a.relm + (size_t)g * a.cap_u * a.relm_ld;
    uint32_t* dst = (uint32_t*)(RMW + (size_t)wave * 16 * rmp);
    const int rw = rmp >> 2;
  // (r = i / rw for i < 17 * 64 without a division per element: the quotient by a 20-bit reciprocal is exact there)
  const uint32_t rw_magic = ((1u << 20) + (uint32_t)rw - 1u) / (uint32_t)rw;
    uint32_t rmq[DL_RIT];                          // all requested before the first use (see k_dl_layer)
#pragma unroll
    for (int u = 0; u < DL_RIT; ++u) {
      const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
      const int rc = row0 + r < n_own ? row0 + r : n_own - 1, cc = c < ldw ? c : ldw - 1;
      rmq[u] = ((const uint32_t*)(src + (size_t)rc * ldb))[cc];
    }
    G2_SCHED_BARRIER();
#pragma unroll
    for (int u = 0; u < DL_RIT; ++u) {
      const int i = lane + 64 * u, r = (int)(((uint32_t)i * rw_magic) >> 20), c = i - r * rw;
      if (i < 16 * rw) dst[i] = (row0 + r < n_own && c < ldw && 4 * c < 32 * nks) ? rmq[u] : 0u;
    }
  }
  float* hi = HI + wave * 16 * G2_XP;
  for (int i = lane; i < 16 * G2_XP; i += 64) hi[i] = 0.f;
  __syncthreads();
  if (active) {
    f32x4 hacc[G2_NR];
#pragma unroll
    for (int r = 0; r < G2_NR; ++r) hacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* rmo = RMW + (size_t)(wave * 16 + li) * rmp + 8 * kq;
    const uint32_t* ohp = OHP + ((li & 7) * kp >> 1) + 4 * kq;
    const int kbit = side ? IGMC_RELM_KT : IGMC_RELM_KF;                       // keep bit of the edge opposite -> own
#pragma unroll 1
    for (int s = 0; s < nks; ++s) {
      const uint2 w = *(const uint2*)(rmo + 32 * s);
      u32x4 pfh = *(const u32x4*)(ohp + 16 * s);
      if (li >= 8) pfh = (u32x4){0u, 0u, 0u, 0u};        // (label rows 8..15 of the 16-row operand do not exist)
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) {
        u32x4 af;
        uint32_t a0, a1, a2, a3;
        g2_expand4<FLAGS>(w.x, (uint32_t)(r + 1), kbit, a0, a1);
        g2_expand4<FLAGS>(w.y, (uint32_t)(r + 1), kbit, a2, a3);
        af[0] = a0; af[1] = a1; af[2] = a2; af[3] = a3;
        hacc[r] = g2_mfma_bf16(pfh, af, hacc[r]);
      }
    }
    // lane (row li, kq): counts of the neighbour labels 4 kq + rr, per relation
    const int row = row0 + li;
#pragma unroll
    for (int r = 0; r < G2_NR; ++r)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int c = 4 * kq + rr;
        if (r < R && c < L) {
          hi[li * G2_XP + r * L + c] = hacc[r][rr];
          if (STORE && row < n_own) a.cnt0[(size_t)(own0 + row) * RL + r * L + c] = (uint16_t)(int)(hacc[r][rr] + 0.5f);
        }
      }
    if (kq == 0 && row < n_own) {
      hi[li * G2_XP + RL + own_lab] = 1.f;
      hi[li * G2_XP + RL + L] = 1.f;
    }
  }
  __syncthreads();
  if (active) {
    // h_0 = tanh([hist | onehot | 1] @ T0) on the f32 matrix cores (RL + L + 1 <= 32 table rows), the order of sums of k_dl_fwd:
    // lane = feature 16 nt + li, registers = rows 4 kq + rr
    f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float av = hi[li * G2_XP + 4 * j + kq];
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT0[(4 * j + kq) * 32 + li], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT0[(4 * j + kq) * 32 + 16 + li], o1, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = row0 + 4 * kq + rr;
      if (row < n_own) {
        a.out[(size_t)(own0 + row) * 32 + li] = g2_tanh(o0[rr]);
        a.out[(size_t)(own0 + row) * 32 + 16 + li] = g2_tanh(o1[rr]);
      }
    }
  }
}

static size_t dl0_lds(int kp) {
  return ((size_t)(8 * kp >> 1) + (size_t)DL_NW * 4 * kp + (size_t)DL_NW * 16 * G2_XP + 1024) * 4;
}

void igmc_launch_dl_layer0(const ModelDev& m, const BatchDev& b, int B, int training, int use_flags, void* stream) {
  Dl0Args a;
  memset(&a, 0, sizeof(a));
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  a.n_users = b.n_users; a.n_items = b.n_items; a.node_off = b.node_off; a.node_label = b.node_label;
  a.relm = b.relm; a.relmT = b.relmT;
  a.cap_u = b.cap_u; a.cap_v = b.cap_v; a.relm_ld = b.relm_ld; a.relmT_ld = b.relmT_ld;
  { const DlSplit sq = dl_split(b.cap_u, b.cap_v, B); a.nqu = sq.nqu; a.nqv = sq.nqv; } a.R = m.R; a.L = m.L; a.kp = 32 * ((cmax + 31) >> 5) + 8;
  a.t0 = m.g2_w + 6 * G2_WIMG;
  a.out = m.h[0];
  a.cnt0 = training ? m.cnt0 : nullptr;
  const int grid = B * (a.nqu + a.nqv);
  const size_t sm = dl0_lds(a.kp);
  if (training) {
    if (use_flags) IGMC_PLAUNCH("k_dl_layer0", (k_dl_layer0<true, true>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_layer0", (k_dl_layer0<false, true>), grid, DL_THREADS, sm, stream, a);
  } else {
    if (use_flags) IGMC_PLAUNCH("k_dl_layer0", (k_dl_layer0<true, false>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_layer0", (k_dl_layer0<false, false>), grid, DL_THREADS, sm, stream, a);
  }
}

static size_t dl_lds(int kp, bool ts = false) {
  return ((size_t)dl_words_front(ts) + (size_t)dl_words_mid(kp, ts) + DL_NW * 32 + 32) * 4;
}

// 1 = the dense per-layer kernels take the conv layers of this arena (IGMC_DL=0 switches them off)
static int dl_base_ok(const ModelDev& m, const BatchDev& b, int B, int wide) {
  const char* e = getenv("IGMC_DL");
  if (e && atoi(e) == 0) return 0;
  if (!b.relm || !b.relmT || !m.g2_w || m.L > 8) return 0;
  const int rows0 = m.R * m.L + m.L + 1;
  // wide: the two-group layout -- six to ten relations, or a layer-0 table of 33..48 rows (two hops)
  if (wide ? (g2_groups(m.R, m.L) == 1 || m.R > G2_NR * G2_NG_MAX || rows0 > 48) : (m.R > G2_NR || rows0 > 32)) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  return cmax <= 256 && B * (sq.nqu + sq.nqv) <= IGMC_GATHER_BLOCKS;
}
int igmc_dl_eligible(const ModelDev& m, const BatchDev& b, int B) {
  if (!dl_base_ok(m, b, B, 0)) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  return dl_lds(32 * ((cmax + 31) >> 5) + 8) <= (size_t)160 * 1024;
}

// 1 = the backward passes of this arena can leave relation-space tables (k_dl_layer<*, true, true>): the tail of the
// subgraph kernel (k_tail_ts -> k_finalize_ts) then replaces G / Y / the weight-gradient products (IGMC_DL_TS=0: never)
int igmc_dl_ts_eligible(const ModelDev& m, const BatchDev& b, int B) {
  const char* e = getenv("IGMC_DL_TS");
  if (e && atoi(e) == 0) return 0;
  if (!igmc_dl_eligible(m, b, B) || !m.ts_part || !m.cnt0) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  const int stride = (B + 7) & ~7;
  if ((sq.nqu + sq.nqv) * stride > IGMC_TS_BLOCKS || m.R * m.L > 20) return 0;
  return dl_lds(32 * ((cmax + 31) >> 5) + 8, true) <= (size_t)160 * 1024;
}

int igmc_dl_grid(const BatchDev& b, int B) {
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  return B * (sq.nqu + sq.nqv);
}

void igmc_launch_g2_compose(const ModelDev& m, const float* P, void* stream) {
  if (m.img_current) return;      // (igmc_model_weights_unchanged: the images of these parameters are in place)
  ++g_igmc_compose_count;
  IGMC_PLAUNCH("k_g2_compose", k_g2_compose, 2 * 3 * g2_groups(m.R, m.L) * (G2_NR + 1) + g2_t0_rows(m.R, m.L) / 32, G2C_THREADS, 0, stream, m, P, m.g2_w);
}

// one conv layer pass: forward (bwd = 0: h_{l-1} -> h_l) or backward (dPre_l -> dPre_{l-1}, G, d att partials)
void igmc_launch_dl_layer(const ModelDev& m, const BatchDev& b, const float* P, int B, int l, int bwd, int use_flags,
                          float* zero_out, void* stream, int tables) {
  DlArgs a;
  memset(&a, 0, sizeof(a));
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  a.n_users = b.n_users; a.n_items = b.n_items; a.node_off = b.node_off; a.relm = b.relm; a.relmT = b.relmT;
  a.cap_u = b.cap_u; a.cap_v = b.cap_v; a.relm_ld = b.relm_ld; a.relmT_ld = b.relmT_ld;
  { const DlSplit sq = dl_split(b.cap_u, b.cap_v, B); a.nqu = sq.nqu; a.nqv = sq.nqv; } a.R = m.R; a.D = m.D; a.l = l; a.kp = 32 * ((cmax + 31) >> 5) + 8;
  a.in = bwd ? m.dpre[l] : m.h[l - 1];
  a.hprev = m.h[l - 1];
  a.out = bwd ? m.dpre[l - 1] : m.h[l];
  a.zero_out = zero_out;
  a.gagg = bwd ? m.gagg[l - 1] : nullptr;
  a.Y = bwd ? m.Y[l - 1] : nullptr;
  a.gatt_part = bwd ? m.gatt_part + (size_t)(l - 1) * IGMC_GATHER_BLOCKS * m.R * 4 : nullptr;
  a.gfeat = m.gfeat; a.dcat = bwd ? m.dcat[l - 1] : nullptr;
  a.img = m.g2_w + (size_t)((l - 1) * 2 + (bwd ? 1 : 0)) * G2_WIMG;
  a.bias = P + m.off_bias[l]; a.att = P + m.off_att[l];
  a.L = m.L;
  const int grid = B * (a.nqu + a.nqv);
  if (bwd && tables) {       // relation-space tables instead of G / d att partials (igmc_dl_ts_eligible)
    a.ts_part = m.ts_part; a.ts_stride = m.ts_stride; a.slot_stride = (B + 7) & ~7;
    a.cnt0 = m.cnt0; a.node_label = b.node_label;
    a.gagg = nullptr; a.Y = nullptr; a.gatt_part = nullptr;
    const size_t smt = dl_lds(a.kp, true);
    if (use_flags) IGMC_PLAUNCH("k_dl_layer_bwd", (k_dl_layer<true, true, true>), grid, DL_THREADS, smt, stream, a);
    else IGMC_PLAUNCH("k_dl_layer_bwd", (k_dl_layer<false, true, true>), grid, DL_THREADS, smt, stream, a);
    return;
  }
  const size_t sm = dl_lds(a.kp);
  if (bwd) {
    if (use_flags) IGMC_PLAUNCH("k_dl_layer_bwd", (k_dl_layer<true, true, false>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_layer_bwd", (k_dl_layer<false, true, false>), grid, DL_THREADS, sm, stream, a);
  } else {
    if (use_flags) IGMC_PLAUNCH("k_dl_layer_fwd", (k_dl_layer<true, false, false>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_layer_fwd", (k_dl_layer<false, false, false>), grid, DL_THREADS, sm, stream, a);
  }
}

// 1 = the forward of this arena's dense layers runs as ONE launch (k_dl_fwd): exchange regions for 256 nodes a side, every
// workgroup of the launch resident at once (IGMC_DL_FUSED=0: the per-layer launches)
int igmc_dl_fwd_eligible(const ModelDev& m, const BatchDev& b, int B) {
  if (!igmc_g2_xcd_ok()) return 0;      // (the members' exchange goes through the L2 of one XCD)
  const char* e = getenv("IGMC_DL_FUSED");
  if (e && atoi(e) == 0) return 0;
  if (!igmc_dl_eligible(m, b, B) || !m.g2_ex || m.ex_nodes < DLX_K || b.graph_cap > m.g2_graphs) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  if (B * (sq.nqu + sq.nqv) > 224) return 0;                  // (one workgroup per CU, all of them resident: the members wait for each other)
  return (size_t)dlf_words(32 * ((cmax + 31) >> 5) + 8) * 4 <= (size_t)160 * 1024;
}

// 1 = the group-split forms of k_dl_fwd / k_dl_bwd take this arena: two relation groups, no workgroup with more than DL_NW / 2
// bundles (dl_split / dl_rows over the slot capacities), both images beside the planes in LDS.  IGMC_DL_GSPLIT=0: the
// group-after-group form (test hook).
static int dl_gsplit(const ModelDev& m, const BatchDev& b, int B) {
  if (g2_groups(m.R, m.L) != 2 || g2_rel_groups(m.R) != 2) return 0;
  const char* e = getenv("IGMC_DL_GSPLIT");
  if (e && atoi(e) == 0) return 0;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  const int nbu = (b.cap_u + 15) >> 4, nbv = (b.cap_v + 15) >> 4;
  if ((nbu + sq.nqu - 1) / sq.nqu > DL_GB || (nbv + sq.nqv - 1) / sq.nqv > DL_GB) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v, kp = 32 * ((cmax + 31) >> 5) + 8;
  return (size_t)dlf_words_gs(kp) * 4 <= 160 * 1024 && (size_t)dlb_words_gs(kp) * 4 <= 160 * 1024;
}

void igmc_launch_dl_fwd(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                        float* zero_out, int self_seq, void* stream) {
  DlfArgs a;
  memset(&a, 0, sizeof(a));
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  a.n_users = b.n_users; a.n_items = b.n_items; a.node_off = b.node_off; a.node_label = b.node_label;
  a.relm = b.relm; a.relmT = b.relmT;
  a.cap_u = b.cap_u; a.cap_v = b.cap_v; a.relm_ld = b.relm_ld; a.relmT_ld = b.relmT_ld;
  { const DlSplit sq = dl_split(b.cap_u, b.cap_v, B); a.nqu = sq.nqu; a.nqv = sq.nqv; } a.R = m.R; a.L = m.L; a.kp = 32 * ((cmax + 31) >> 5) + 8;
  a.s_lab = b.s_lab; a.slot = b.slot;
  for (int l = 0; l < 4; ++l) {
    a.h[l] = m.h[l];
    a.off_bias[l] = (int)m.off_bias[l];
  }
  a.zero_out = zero_out;
  a.cnt0 = training ? m.cnt0 : nullptr;
  a.g2_w = m.g2_w; a.P = P;
  a.ex = m.g2_ex; a.ex_stride = m.g2_ex_stride;
  a.gs_bar = m.gs_bar; a.gs_err = m.gs_err;
  a.self_seq = self_seq;
  a.timing = getenv("IGMC_DL_TIMING") ? atoi(getenv("IGMC_DL_TIMING")) : 0;
  a.B = B;
  const int grid = 8 * ((B + 7) / 8) * (a.nqu + a.nqv);      // (XCD-aligned blocks of 8 * members workgroups: DLX_DECODE)
  const int ng = g2_groups(m.R, m.L);
  const int gs = dl_gsplit(m, b, B);
  const size_t sm = (size_t)(gs ? dlf_words_gs(a.kp) : dlf_words(a.kp, ng)) * 4;
#ifdef IGMC_HIPEMU
  hipemu::rt().co_cs = a.nqu + a.nqv;                   // the members of a subgraph run together:
  hipemu::rt().co_stride = 8;                           // workgroups 8 nmem j + x + 8 rem
  hipemu::rt().co_block = 8 * (a.nqu + a.nqv);
#endif
  if (ng == 1) {
    if (training) {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, true, 1>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, true, 1>), grid, DL_THREADS, sm, stream, a);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, false, 1>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, false, 1>), grid, DL_THREADS, sm, stream, a);
    }
  } else if (gs) {
    if (training) {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, true, 2, true>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, true, 2, true>), grid, DL_THREADS, sm, stream, a);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, false, 2, true>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, false, 2, true>), grid, DL_THREADS, sm, stream, a);
    }
  } else {
    if (training) {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, true, 2>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, true, 2>), grid, DL_THREADS, sm, stream, a);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<true, false, 2>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_fwd", (k_dl_fwd<false, false, 2>), grid, DL_THREADS, sm, stream, a);
    }
  }
}

// (k_dl_bwd: same conditions as k_dl_fwd -- whose launch precedes it and maintains the exchange regions -- plus the tables')
// 1 = more than G2_NR relations (<= G2_NR * G2_NG_MAX, layer-0 table <= 48 rows) on the one-launch dense kernels, which take the
// relations in groups: k_dl_fwd / k_head_sub / k_dl_bwd<*, NG> with the relation-space tables behind them -- all of it or
// nothing (the per-layer kernels k_dl_layer0 / k_dl_layer stop at G2_NR relations)
// 1 = ... and in the group-split form (dl_gsplit): both relation groups at once on the halves of a workgroup
int igmc_dl_wide_gsplit(const ModelDev& m, const BatchDev& b, int B) { return igmc_dl_wide(m, b, B) && dl_gsplit(m, b, B); }

int igmc_dl_wide(const ModelDev& m, const BatchDev& b, int B) {
  if (!igmc_g2_xcd_ok()) return 0;      // (the members' exchange goes through the L2 of one XCD)
  if (!dl_base_ok(m, b, B, 1)) return 0;
  const char* e = getenv("IGMC_DL_FUSED");
  if (e && atoi(e) != 2) return 0;
  const char* et = getenv("IGMC_DL_TS");
  if (et && atoi(et) == 0) return 0;
  // (the same predicate as the launch sequence's `fts_pre`, model.hip: with the tables' tail switched off -- IGMC_FIN_MODE=0 --
  //  or its stash missing the step does NOT go wide, and the arena must then carry the CSR the row walkers read)
  const char* ef = getenv("IGMC_FIN_MODE");
  if ((ef && atoi(ef) == 0) || !m.fin_stash || !m.datt_part) return 0;
  if (!m.g2_ex || m.ex_nodes < DLX_K || b.graph_cap > m.g2_graphs || !m.ts_part || !m.cnt0) return 0;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  const DlSplit sq = dl_split(b.cap_u, b.cap_v, B);
  const int stride = (B + 7) & ~7, kp = 32 * ((cmax + 31) >> 5) + 8;
  if (B * (sq.nqu + sq.nqv) > 224 || (sq.nqu + sq.nqv) * stride > IGMC_TS_BLOCKS) return 0;
  const int ng = g2_groups(m.R, m.L);
  return (size_t)dlf_words(kp, ng) * 4 <= (size_t)160 * 1024 && (size_t)dlb_words(kp, ng) * 4 <= (size_t)160 * 1024;
}

int igmc_dl_bwd_eligible(const ModelDev& m, const BatchDev& b, int B) {
  if (!igmc_g2_xcd_ok()) return 0;      // (the members' exchange goes through the L2 of one XCD)
  if (!igmc_dl_fwd_eligible(m, b, B) || !igmc_dl_ts_eligible(m, b, B)) return 0;
  const char* e = getenv("IGMC_DL_FUSED");
  if (e && atoi(e) == 1) return 0;                 // (1: the forward only)
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  return (size_t)dlb_words(32 * ((cmax + 31) >> 5) + 8) * 4 <= (size_t)160 * 1024;
}

void igmc_launch_dl_bwd(const ModelDev& m, const BatchDev& b, int B, int use_flags, void* stream, const DlHead* head, int dense3) {
  DlbArgs a;
  memset(&a, 0, sizeof(a));
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  a.n_users = b.n_users; a.n_items = b.n_items; a.node_off = b.node_off; a.node_label = b.node_label;
  a.relm = b.relm; a.relmT = b.relmT;
  a.cap_u = b.cap_u; a.cap_v = b.cap_v; a.relm_ld = b.relm_ld; a.relmT_ld = b.relmT_ld;
  { const DlSplit sq = dl_split(b.cap_u, b.cap_v, B); a.nqu = sq.nqu; a.nqv = sq.nqv; } a.R = m.R; a.L = m.L; a.D = m.D; a.kp = 32 * ((cmax + 31) >> 5) + 8;
  a.s_lab = b.s_lab; a.slot = b.slot;
  for (int l = 0; l < 3; ++l) a.h[l] = m.h[l];
  a.dpre3 = m.dpre[3]; a.gfeat = m.gfeat; a.g2_w = m.g2_w; a.cnt0 = m.cnt0;
  a.ts_part = m.ts_part; a.ts_stride = m.ts_stride; a.slot_stride = (B + 7) & ~7;
  a.ex = m.g2_ex; a.ex_stride = m.g2_ex_stride;
  a.gs_bar = m.gs_bar; a.gs_err = m.gs_err;
  a.timing = getenv("IGMC_DL_TIMING") ? atoi(getenv("IGMC_DL_TIMING")) : 0;
  if (dense3) {
    a.dense3 = 1;
    for (int l = 0; l < 3; ++l) a.dcat[l] = m.dcat[l];
  }
  if (head) {
    a.head = 1; a.hb = b; a.hm = m; a.P = head->P; a.inj_mask = head->inj_mask; a.seed = head->seed; a.step = head->step;
    a.mult = head->mult; a.grad_scale = head->grad_scale; a.out = head->out;
  }
  a.B = B;
  const int grid = 8 * ((B + 7) / 8) * (a.nqu + a.nqv);      // (XCD-aligned blocks of 8 * members workgroups: DLX_DECODE)
  const int gs = !dense3 && dl_gsplit(m, b, B);
  const size_t sm = (size_t)(gs ? dlb_words_gs(a.kp) : dlb_words(a.kp, g2_groups(m.R, m.L))) * 4;
#ifdef IGMC_HIPEMU
  hipemu::rt().co_cs = a.nqu + a.nqv;
  hipemu::rt().co_stride = 8;
  hipemu::rt().co_block = 8 * (a.nqu + a.nqv);
#endif
  if (g2_groups(m.R, m.L) == 1) {
    if (dense3) {
      if (use_flags) IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<true, 1, true>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<false, 1, true>), grid, DL_THREADS, sm, stream, a);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<true, 1, false>), grid, DL_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<false, 1, false>), grid, DL_THREADS, sm, stream, a);
    }
  } else if (gs) {
    if (use_flags) IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<true, 2, false, true>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<false, 2, false, true>), grid, DL_THREADS, sm, stream, a);
  } else {        // (relation groups: centre-node readout only -- igmc_conv_bwd_tables asks for the one-group layout)
    if (use_flags) IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<true, 2, false>), grid, DL_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_dl_bwd", (k_dl_bwd<false, 2, false>), grid, DL_THREADS, sm, stream, a);
  }
}

int igmc_dl_prepare() {
#ifndef IGMC_HIPEMU
  const int mx = 160 * 1024;
  if (hipFuncSetAttribute((const void*)k_dl_layer<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
#define DL_MAXLDS(k) if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1
  DL_MAXLDS((k_dl_bwd<true, 1, false>)); DL_MAXLDS((k_dl_bwd<false, 1, false>)); DL_MAXLDS((k_dl_bwd<true, 2, false>)); DL_MAXLDS((k_dl_bwd<false, 2, false>));
  DL_MAXLDS((k_dl_bwd<true, 1, true>)); DL_MAXLDS((k_dl_bwd<false, 1, true>));
  DL_MAXLDS((k_dl_bwd<true, 2, false, true>)); DL_MAXLDS((k_dl_bwd<false, 2, false, true>));
  DL_MAXLDS((k_dl_fwd<true, true, 1>)); DL_MAXLDS((k_dl_fwd<false, true, 1>)); DL_MAXLDS((k_dl_fwd<true, false, 1>)); DL_MAXLDS((k_dl_fwd<false, false, 1>));
  DL_MAXLDS((k_dl_fwd<true, true, 2>)); DL_MAXLDS((k_dl_fwd<false, true, 2>)); DL_MAXLDS((k_dl_fwd<true, false, 2>)); DL_MAXLDS((k_dl_fwd<false, false, 2>));
  DL_MAXLDS((k_dl_fwd<true, true, 2, true>)); DL_MAXLDS((k_dl_fwd<false, true, 2, true>)); DL_MAXLDS((k_dl_fwd<true, false, 2, true>)); DL_MAXLDS((k_dl_fwd<false, false, 2, true>));
#undef DL_MAXLDS
  if (hipFuncSetAttribute((const void*)k_dl_layer0<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer0<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer0<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_dl_layer0<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
#endif
  return 0;
}
