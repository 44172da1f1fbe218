// g2_subgraph.h -- k_graph_step2: forward + loss + backward of one enclosing subgraph by a cluster of workgroups
// (included once, by graphstep2.hip, behind the shared gather / transform steps; design notes at the top of that file)
#pragma once

// Everything the kernel reads, in ONE compact argument block (the full BatchDev / ModelDev / GsArgs views are ~1.3 KB of
// kernel arguments: loading and address-forming from them cost ~1200 scalar instructions before the first useful load).
struct G2Args {
  // batch
  const int32_t* n_users;
  const int32_t* n_items;
  int B;
  const uint8_t* s_lab;
  const uint8_t* relm;
  const float* y;
  int cap_u, cap_v, slot, relm_ld, graph_cap;
  // model
  int R, L, D, ts_stride;
  float* h[4];
  float* ts_part;
  unsigned char* g2_px;        // plane exchange regions [5 exchanges][graphs][2 sides][G2_PX_BYTES] (g2_prims.h)
  unsigned long long* g2_fx;
  size_t g2_px_stride;         // bytes per exchange
  const float* g2_w;
  int* gs_bar;
  int* gs_err;
  float* a1;
  float* dz;
  float* feat;
  float* gfeat;
  float* err;
  uint8_t* lmask;
  const int64_t* ctrl;
  int off_bias[4];
  int off_l1w, off_l1b, off_l2w, off_l2b;
  // call
  const float* P;
  const uint8_t* inj_mask;
  uint64_t seed, step;
  float mult, grad_scale;
  float* out;
  unsigned long long* ts;
  int timing, cs, stride, self_seq;
  G2Layout lay2;
};

// ONCE: a workgroup takes exactly one subgraph (clusters, cs > 1: every product launch) -- the body is then STRAIGHT-LINE code.
// Written as a loop over subgraphs, the compiler hoisted every subgraph-invariant scalar -- some 120 LDS section bases,
// strides and argument fields -- into the loop's preheader and parked them in VGPR lanes: 430 scalar instructions (~3.4 k
// cycles, profiles/r06_g2_phase_clocks.txt "loop start -> loads issued") before the first vector load of the launch left.
template <bool FLAGS, bool TRAIN, bool ONCE>
__global__ __launch_bounds__(G2_THREADS) void k_graph_step2(G2Args a) {
  __builtin_amdgcn_s_setprio(3);      // (step chain: ahead of the extraction chain's waves wherever the two share a SIMD)
  const float* P = a.P;
  IGMC_DYN_SMEM(smem);
  float* S = (float*)smem;
  const G2Layout lay = a.lay2;
  const int kp = lay.kp, nsides = lay.nsides;
  uint32_t* PLN = (uint32_t*)(S + lay.planes);          // [nsides][3 terms][32 features][kp] bf16: gather source
  uint32_t* OHP = (uint32_t*)(S + lay.ohp);             // [nsides][8 labels][kp] bf16 one-hot label planes (layer 0)
  unsigned char* slab = (unsigned char*)(S + lay.lab);  // [2][128] node labels of both sides
  float* XOA = S + lay.xo;                              // [2][4 bundles][16][G2_XP]: the bundle's own rows of x / dPre
  float* HSS = S + lay.hs;                              // [4][16][G2_XP] h_{l-1} rows of the bundle (backward)
  float* TILES = S + lay.tile;                          // [4][16][G2_TP] T' rows of the bundle (backward)
  float* HIST = S + lay.hist;                           // [4][16][G2_XP] layer-0 input [code histogram | onehot | 1]
  float* PXA = S + lay.px;                              // [4 bundles][2 waves][64 lanes] x 4: the pair's partial outputs
  float2* sW2 = (float2*)(S + lay.wreg);                // [G2_WIMG words] B operand of the layer as bf16 term fragments
  float* sT0 = S + lay.t0;                              // [32][32] layer-0 table
  float* sfeat = S + lay.head;            // [256] centre-node readout
  float* sgf = sfeat + 256;               // [256] d feat
  float* sa1 = sgf + 256;                 // [128]
  float* skeep = sa1 + 128;               // [128]
  float* sdz = skeep + 128;               // [128]
  float* sred = sdz + 128;                // [512]
  float* misc = sred + 512;               // [16]
  const int R = a.R, L = a.L, RL = R * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int B = a.B;
  const int ts = a.ts_stride;
  const int cs = a.cs;
  // The members of a cluster are CONSECUTIVE workgroups (subgraph = blockIdx / cs): workgroups are handed to the CUs in
  // index order, so with any number >= cs of free CUs the first clusters are complete, finish and make room for the
  // next -- a partly resident chip (another kernel holding CUs) slows the launch down but cannot leave every resident
  // member waiting for a non-resident one.  (Members need not share an XCD: sc1 words are served from the coherent
  // level wherever they were written.)  The bounded polls stay as the backstop.
  // Workgroups are dealt to the eight XCDs round robin by index: the members c = 0 .. cs - 1 of cluster (j, x) are the
  // workgroups 8 cs j + 8 c + x -- one XCD, one L2, which is where they exchange their rows (g2_prims.h); an XCD takes the
  // members of its clusters one after the other, so the residency argument above holds per XCD.
  const int cm = (cs > 1) ? (int)((blockIdx.x >> 3) % cs) : 0;
  const int half = 2 * cs;                              // waves of the cluster per side
  const uint32_t seq = g2_ld_seq(a.gs_bar);
  const uint32_t tag0 = seq * 8u + 1u;
  const uint64_t step = a.ctrl ? (uint64_t)a.ctrl[IGMC_CTRL_STEP] : a.step;
  if (a.ts && tid == 0) g2_clock_open(a.ts);
  auto xtag = [&](int x) { return tag0 + (uint32_t)x; };      // flag value of exchange x of this launch (never 0)
  G2_STAMP(0);
  if (a.timing && tid == 0 && blockIdx.x < 1024) {
    g_g2_wg[blockIdx.x][0] = g2_wall_clock();
    g_g2_wg[blockIdx.x][2] = g2_xcc_id();
  }

  // ---- the first subgraph's extents are requested before anything else (two dependent round trips overlap with
  //      the staging of the layer-0 table, which k_g2_compose formed from the current weights)
  const int g_first = (cs > 1) ? (int)(blockIdx.x / (8 * cs)) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
  const int g_pre = (g_first < a.graph_cap) ? g_first : a.graph_cap - 1;      // (a padding workgroup: any valid slot)
  const int pre_cu = a.n_users[g_pre], pre_cv = a.n_items[g_pre];
  // ... and so are the set-up's global loads, which depend on the subgraph slot only (labels from the per-graph scratch
  // slots, the block bytes of the lane's fragments): ONE round trip, under the kernel's scalar prologue
  const int ld = a.relm_ld;
  auto load_label = [&](int g2) {
    // (capacity of the lane's side by arithmetic on the two VALUES: a per-lane select between the two kernel-argument
    //  fields compiles to a vector load from the argument segment + a vmcnt(0) wait in front of every other load)
    const int hi = (tid >> 7) & 1, t7 = tid & 127;              // (threads 256.. repeat the first 256: their value is unused)
    const int capx = a.cap_u + hi * (a.cap_v - a.cap_u);
    return (int)a.s_lab[(size_t)g2 * a.slot + hi * a.cap_u + ((t7 < capx) ? t7 : 0)];
  };
  int labv_raw = 0;
  // layer-0 table (4 KB): requested here, global -> LDS directly, landed by the set-up's barriers
  if (tid < 256) g2_glds16((const float4*)(a.g2_w + 6 * G2_WIMG) + (tid & ~63), (float4*)sT0 + (tid & ~63), tid & 63);
  // partial-table slot of this workgroup: member c of subgraph g -> g + c * stride (what k_tail_ts sums)
  const int tslot = (cs > 1) ? g_first + cm * a.stride : (int)blockIdx.x;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};        // layer-0 table gradient tile (code half, feature half) of this wave
  bool first_graph = true;
  G2_STAMP(1);

  int g = g_first;
  if (g < B) do {
    // per-lane indices are re-derived from an opaque copy of the thread index INSIDE the subgraph loop: everything
    // computed from them then stays inside it (hoisted out of the loop, hundreds of loop-invariant addresses occupy --
    // and spill -- registers for the whole kernel)
    int tid_g = threadIdx.x;
    G2_OPAQUE(tid_g);
    const int tid = tid_g, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int bw = wave >> 1, hf = wave & 1;              // bundle of the workgroup, feature tile of the pair (waves 2 b and
                                                          // 2 b + 1 sit on different SIMDs; each SIMD holds two bundles' waves)
    const int gw = cm * G2_NB + bw;
    const int side = gw / half, bi = gw - side * half;    // this wave's side (0 users, 1 items) and bundle of that side
    labv_raw = load_label(g);
    // The block bytes of THIS LANE's fragments -- the 8 opposite-side nodes 32 s + 8 kq .. + 7 of its row (users: 8 consecutive
    // bytes of the row; items: the column, one byte of 8 consecutive rows) -- straight from the arena's row-major block into
    // registers (round 6).  They went global -> registers -> an LDS image -> registers before: a 35 KB zero fill, the image's
    // stores, a gather of it and two of the set-up's three barriers, all in front of layer 0.  Rows of the slot past the
    // subgraph's users hold an earlier subgraph's bytes (the extraction clears cu rows): masked below, once the extents are in.
    // No bounds checks: the arena's block allocation carries a tail pad of 128 rows (capi.hip), so a lane of an inactive row
    // or k-step reads bytes that exist and are then masked -- by `active` / the row's validity, and by the count of valid
    // opposite-side nodes among the lane's eight (the per-load selects of a guarded form cost the item side 5 k cycles).
    uint32_t bw0[G2_KS], bw1[G2_KS];
    {
      const unsigned char* rmg = a.relm + (size_t)g * a.cap_u * ld;      // (uniform: scalar base, 32-bit lane offsets)
      const uint32_t rrow = (uint32_t)(16 * bi + li);
#pragma unroll
      for (int s = 0; s < G2_KS; ++s) {
        const uint32_t c0 = (uint32_t)(32 * s + 8 * kq);
        if (side == 0) {
          const uint32_t* p4 = (const uint32_t*)(rmg + (rrow * (uint32_t)ld + c0));      // (ld % 4 == 0)
          bw0[s] = p4[0];
          bw1[s] = p4[1];
        } else {
          const unsigned char* pc = rmg + (c0 * (uint32_t)ld + rrow);
          bw0[s] = (uint32_t)pc[0] | ((uint32_t)pc[ld] << 8) | ((uint32_t)pc[2 * ld] << 16) | ((uint32_t)pc[3 * ld] << 24);
          bw1[s] = (uint32_t)pc[4 * ld] | ((uint32_t)pc[5 * ld] << 8) | ((uint32_t)pc[6 * ld] << 16) | ((uint32_t)pc[7 * ld] << 24);
        }
      }
    }
    // ---- the LDS zero fills (16-byte stores) run under the latency of the set-up's global loads: placed in front of
    //      everything that needs the subgraph's extents (a wait for THOSE in front of the fills is a round trip of idling)
    G2_STAMP(48);
    {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      // (the plane image is NOT cleared: its first reader is layer 1's gather, behind a fetch that copies the whole image
      //  from the exchange region -- planes_load -- and dPre_3's set-up clears what it needs itself; there is no block image)
      for (int i = tid; i < 2 * G2_NB * 16 * G2_XP / 4; i += G2_THREADS) ((float4*)XOA)[i] = z4;
      for (int i = tid; i < G2_NB * 16 * G2_XP / 4; i += G2_THREADS) ((float4*)HIST)[i] = z4;
    }
    G2_STAMP(49);
    const int cu = first_graph ? pre_cu : a.n_users[g], cv = first_graph ? pre_cv : a.n_items[g];
    const int n_own = side ? cv : cu, n_opp = side ? cu : cv;
    const int nbs = g * a.slot + (side ? a.cap_u : 0);       // first row of this wave's side in the h_l scratch (slot-based:
                                                             // the kernel never reads the collated node offsets)
    const int nbun = (n_own + 15) >> 4;
    const bool active = bi < nbun;
    const int row0 = 16 * bi;
    const int nks = (n_opp + 31) >> 5;
    const int sx = (nsides == 2) ? side : 0, so = (nsides == 2) ? 1 - side : 0;    // LDS images of own / opposite side
    uint32_t* pl = PLN + so * lay.pside;
    uint32_t* ohp = OHP + so * (8 * kp >> 1);
    float* XO0 = XOA + bw * 16 * G2_XP;                       // ping
    float* XO1 = XOA + (G2_NB + bw) * 16 * G2_XP;             // pong
    float* HS = HSS + bw * 16 * G2_XP;
    float* T = TILES + bw * 16 * G2_TP;
    float* HI = HIST + bw * 16 * G2_XP;
    f32x4* PXo = (f32x4*)PXA + (bw * 2 + hf) * 64 + lane;           // this wave's partial of the PARTNER's tile
    const f32x4* PXi = (const f32x4*)PXA + (bw * 2 + (1 - hf)) * 64 + lane;    // the partner's partial of this wave's tile
    // plane exchange regions of this subgraph: [exchange x][g][side]
    const size_t exs = a.g2_px_stride;
    auto px_of = [&](int x, int sd) { return a.g2_px + (size_t)x * exs + ((size_t)g * 2 + sd) * G2_PX_BYTES; };
    unsigned long long* fx = a.g2_fx + (size_t)g * 256;
    const int nbun_opp = (n_opp + 15) >> 4;

    // ---- set-up: labels, the block image (rows = users), one-hot label planes
    if (tid < 256) slab[tid] = (unsigned char)(((tid & 127) < ((tid >> 7) ? cv : cu)) ? labv_raw : 255);
    G2_STAMP(50);
    __syncthreads();
    G2_STAMP(2);
    {
      // one-hot planes of the labels of the opposite side(s): plane[label][node] = 1.0 (bf16)
      for (int i = tid; i < nsides * 8 * (kp >> 1); i += G2_THREADS) {
        const int s2 = i / (8 * (kp >> 1)), rem = i - s2 * (8 * (kp >> 1));
        const int lb = rem / (kp >> 1), q = rem - lb * (kp >> 1);
        const int sd = (nsides == 2) ? s2 : 1 - side;        // the side whose labels this image holds
        const int l0 = (2 * q < 128) ? slab[sd * 128 + 2 * q] : 255, l1 = (2 * q + 1 < 128) ? slab[sd * 128 + 2 * q + 1] : 255;
        OHP[i] = ((l0 == lb) ? 0x3F80u : 0u) | ((l1 == lb) ? 0x3F800000u : 0u);
      }
    }
    __syncthreads();
    G2_STAMP(3);
    // ---- A fragments of this wave's bundle: A[r][s] = the 16 x 32 block (rows of the bundle) x (opposite nodes 32 s ..)
    //      of relation r as the MFMA B operand of the transposed gather; forward and (with edge dropout) backward masks
    //      (both waves of a pair hold the bundle's fragments).  Two waves per SIMD leave a wave 256 registers: the fragments
    //      stay resident as BYTE masks -- AM[r][s][h] = 0xFF in the bytes of the four nodes 4 (2 kq + h) .. + 3 of k-step s
    //      whose block byte is relation r (40 registers instead of 80) -- and become bf16 1.0 / 0.0 pairs by a byte
    //      permute + and per dword right in front of their MFMAs (VALU work that runs under the partner wave's matrix
    //      work).  With edge dropout the BACKWARD masks (the other keep bit) are derived from the eight block bytes per
    //      k-step inside the backward gather.
    uint32_t AM[G2_NR][G2_KS][2];
    uint32_t RB[FLAGS ? G2_KS : 1][2];
    const int kb = side ? IGMC_RELM_KF : IGMC_RELM_KT;      // keep bit of the edge  own -> opposite
    {
      const int kf = side ? IGMC_RELM_KT : IGMC_RELM_KF;    // keep bit of the edge  opposite -> own
#pragma unroll
      for (int s = 0; s < G2_KS; ++s) {
        uint32_t w0 = 0u, w1 = 0u;
        if (active && s < nks && row0 + li < n_own) {
          w0 = bw0[s];
          w1 = bw1[s];
          {   // the lane's 8 bytes are 8 opposite-side nodes: past that side's extent they are stale rows (items' lanes) or
              // the next row's bytes (users' lanes, where the row pitch ends inside the k-step)
            const int keep = n_opp - (32 * s + 8 * kq);
            if (keep < 8) {
              w1 = (keep <= 4) ? 0u : (w1 & (0xFFFFFFFFu >> (8 * (8 - keep))));
              if (keep < 4) w0 = (keep <= 0) ? 0u : (w0 & (0xFFFFFFFFu >> (8 * (4 - keep))));
            }
          }
        }
        if constexpr (FLAGS) {
          RB[s][0] = w0;
          RB[s][1] = w1;
        }
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          AM[r][s][0] = g2_bytemask<FLAGS>(w0, (uint32_t)(r + 1), kf);
          AM[r][s][1] = g2_bytemask<FLAGS>(w1, (uint32_t)(r + 1), kf);
        }
      }
    }
    // (opaque copies: the compiler would otherwise form all 20 fragments once and keep them -- the 80 registers again)
    auto frag_f = [&](int r, int s) {
      uint32_t m0 = AM[r][s][0], m1 = AM[r][s][1];
      G2_OPAQUE(m0);
      G2_OPAQUE(m1);
      return g2_mask_frag(m0, m1);
    };
    auto frag_b = [&](int r, int s) {
      uint32_t m0 = FLAGS ? RB[FLAGS ? s : 0][0] : AM[r][s][0], m1 = FLAGS ? RB[FLAGS ? s : 0][1] : AM[r][s][1];
      G2_OPAQUE(m0);
      G2_OPAQUE(m1);
      if constexpr (FLAGS) return g2_mask_frag(g2_bytemask<true>(m0, (uint32_t)(r + 1), kb), g2_bytemask<true>(m1, (uint32_t)(r + 1), kb));
      else return g2_mask_frag(m0, m1);
    };
    G2_STAMP(4);                        // (no barrier: the masks came from registers; the one-hot planes were landed above)

    // B operand of the next conv layer ([W_0; ..; W_4; root] or the transposes, composed once per step by
    // k_g2_compose): requested a phase ahead, written to LDS by stage()
    // It travels global -> LDS directly (global_load_lds, 1 KB per wave instruction; no staging registers, no ds_write
    // pass), requested as soon as the previous image is dead -- right behind the barrier that ends a layer's matrix work --
    // and landed by the barrier in front of the next layer's (the waves drain their vector counter there anyway).
    static_assert(G2_WIMG % 256 == 0, "an image is a whole number of 1 KB pieces");
    auto wload = [&](int l, int trans) {
      const float4* src = (const float4*)(a.g2_w + (size_t)((l - 1) * 2 + trans) * G2_WIMG);
#pragma unroll
      for (int j = 0; j < (G2_WIMG / 256 + G2_THREADS / 64 - 1) / (G2_THREADS / 64); ++j) {
        const int c = wave + j * (G2_THREADS / 64);
        if (c < G2_WIMG / 256) g2_glds16(src + c * 64, (float4*)sW2 + c * 64, lane);
      }
    };
    wload(1, 0);
    // epilogue of a forward layer: tanh, own rows -> LDS tile + h_l (this wave re-reads them in the backward),
    // bf16 terms -> exchange x (l < 3), centre rows -> readout
    //      -- of this wave's feature tile (output features 16 hf + li)
    auto fwd_out = [&](int l, const f32x4& o, float bias, float* XO) {
      const int nt = hf;
      float* hrow = a.h[l] + (size_t)(nbs + row0 + 4 * kq) * 32 + li;                 // rows 4 kq + rr, feature 16 nt + li
      float* xo = XO + 4 * kq * G2_XP + li;
      float v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float tv = g2_tanh(o[rr] + bias);
        v[rr] = (row0 + 4 * kq + rr < n_own) ? tv : 0.f;
      }
      // the planes and their flag FIRST (what the other side waits for: the flag's wait covers three 8-byte stores only);
      // the wave's own copies -- LDS tile, h_l rows for the backward, the readout word -- go out behind it
      if (l < 3) {
        g2_publish_planes(px_of(l, side), kp, 16 * nt + li, row0 + 4 * kq, v);
        g2_flag_raise(px_of(l, side), 2 * bi + hf, xtag(l), lane);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        xo[rr * G2_XP + 16 * nt] = v[rr];
        if (TRAIN && row0 + 4 * kq + rr < n_own) hrow[rr * 32 + 16 * nt] = v[rr];
      }
      if (bi == 0 && kq == 0) g2_pub_f32(fx + side * 128 + l * 32 + 16 * nt + li, v[0], tag0 + G2_FXTAG);
    };
    // this wave's rows of exchange x are in the L2: its flag (every wave of an active bundle, after its epilogue)
    auto raise = [&](int x) { g2_flag_raise(px_of(x, side), 2 * bi + hf, xtag(x), lane); };
    // the opposite side's planes of exchange x: wait for the flags of its bundles' waves, then global -> LDS
    auto fetch = [&](int x) {
      if (nsides == 1) {
        g2_flags_wait(px_of(x, 1 - side), 2 * nbun_opp, xtag(x), lane, a.gs_err);
        g2_planes_load(PLN, px_of(x, 1 - side), kp, wave, lane, G2_THREADS / 64);
      } else {
        for (int s2 = 0; s2 < 2; ++s2) {
          g2_flags_wait(px_of(x, s2), 2 * (((s2 ? cv : cu) + 15) >> 4), xtag(x), lane, a.gs_err);
          g2_planes_load(PLN + s2 * lay.pside, px_of(x, s2), kp, wave, lane, G2_THREADS / 64);
        }
      }
    };

    // ================================================================ layer 0: h0 = tanh([hist | onehot(label) | 1] @ T0)
    if (active) {
      // code histogram of the bundle's rows on the matrix cores: hist_r^T (labels x rows) = onehot^T A_r^T
      f32x4 hacc[G2_NR];
#pragma unroll
      for (int r = 0; r < G2_NR; ++r) hacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 pfh[G2_KS];
#pragma unroll
      for (int s = 0; s < G2_KS; ++s) {
        pfh[s] = *(const u32x4*)(ohp + ((li & 7) * kp >> 1) + 16 * s + 4 * kq);
        if (li >= 8 || s >= nks) pfh[s] = (u32x4){0u, 0u, 0u, 0u};      // (never feed bytes from beyond the image)
      }
      G2_SCHED_BARRIER();
#pragma unroll
      for (int s = 0; s < G2_KS; ++s) {
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          hacc[r] = g2_mfma_bf16(pfh[s], frag_f(r, s), hacc[r]);
        }
      }
      // (both waves of the pair form the whole histogram and write the same values: no hand-off between them in layer 0)
      G2_STAMP(51);
      // lane (row li, kq): counts of labels 4 kq + rr -> the row's input vector [hist | onehot(own label) | 1]
#pragma unroll
      for (int r = 0; r < G2_NR; ++r)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (r < R && 4 * kq + rr < L) HI[li * G2_XP + r * L + 4 * kq + rr] = hacc[r][rr];
      if (kq == 0 && row0 + li < n_own) {
        HI[li * G2_XP + RL + slab[side * 128 + row0 + li]] = 1.f;
        HI[li * G2_XP + RL + L] = 1.f;
      }
      IGMC_WAVE_SYNC();
      f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float av = HI[li * G2_XP + 4 * j + kq];
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sT0[(4 * j + kq) * 32 + 16 * hf + li], o, 0, 0, 0);
      }
      G2_STAMP(52);
      fwd_out(0, o, 0.f, XO0);
    }
    G2_STAMP(5);

    // ================================================================ conv layers 1..3, forward
#pragma unroll
    for (int l = 1; l < 4; ++l) {
      float* XOc = (l & 1) ? XO0 : XO1;             // x of the bundle's own rows (h_{l-1})
      float* XOn = (l & 1) ? XO1 : XO0;             // h_l
      G2_STAMP(6 + 3 * (l - 1));
      const float bias0 = P[a.off_bias[l] + 16 * hf + li];
      // the opposite side's h_{l-1} as bf16 planes, and the layer's weight image (the previous one died at the previous
      // layer's barrier): both global -> LDS, landed by the barrier below
      fetch(l - 1);
      if (l > 1) wload(l, 0);
      __syncthreads();
      G2_STAMP(7 + 3 * (l - 1));
      float bias0_ = bias0;                       // landed: no wait for it is left inside the epilogue (a wait there
      G2_OPAQUE(bias0_);                          // would also drain the epilogue's own stores, one round trip each)
      f32x4 o[2];
      if (active) {
        int lane_ = lane;
        G2_OPAQUE(lane_);
        const int li_ = lane_ & 15, kq_ = lane_ >> 4;
        f32x4 acc[G2_NR];
        g2_gather_h(pl, kp, nks, frag_f, li_, kq_, hf, acc);
        if (l == 2) G2_STAMP(40);
        g2_transform_h(acc, XOc, (const uint32_t*)sW2, li_, kq_, hf, o);
        if (l == 2) G2_STAMP(41);
        *PXo = hf ? o[0] : o[1];                    // the partner's tile: this wave's half of its K
      }
      __syncthreads();                              // the pair's partials are exchanged (planes / sW2 are dead as well)
      if (l == 2) G2_STAMP(42);
      if (active) {
        const f32x4 po = *PXi;
        f32x4 of;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) of[rr] = hf ? po[rr] + o[1][rr] : o[0][rr] + po[rr];      // (K half 0 + K half 1)
        fwd_out(l, of, bias0_, XOn);
      }
      G2_STAMP(36 + (l - 1));
      // (no barrier here: planes / sW2 were dead at the barrier above; the pair's two halves of h_l in XOn and the reuse of
      //  PX are ordered by the next phase's barrier -- behind the next layer's reload, or the readout poll)
      G2_STAMP(8 + 3 * (l - 1));
    }

    // ================================================================ head: lin1 / ReLU / dropout / lin2 / residual
    // the head's small operands -- this lane's lin1 bias and lin2 weight, lin2's bias, the target rating, the lin2 weight of
    // the dz pass -- are requested HERE, in front of the readout poll: left at their uses they were three dependent round trips
    // (bias / weight behind the lin1 butterfly, lin2 bias / y behind the next barrier, the dz pass's weight behind the one
    // after) inside a phase that is nothing but latency
    const int ju_h = 16 * wave + ((lane >> 1) & 15);
    float hd_l1b = P[a.off_l1b + ju_h], hd_l2w = P[a.off_l2w + ju_h], hd_l2b = P[a.off_l2b], hd_y = a.y[g];
    float hd_l2wt = TRAIN ? P[a.off_l2w + (tid & 127)] : 0.f;
    if (tid < 256) sfeat[tid] = g2_poll_f32(fx + tid, tag0 + G2_FXTAG, a.gs_err);
    __syncthreads();
    G2_OPAQUE(hd_l1b); G2_OPAQUE(hd_l2w); G2_OPAQUE(hd_l2b); G2_OPAQUE(hd_y); G2_OPAQUE(hd_l2wt);      // (landed with the poll)
    G2_STAMP(15);
    // (the wave's sixteen lin1 rows stay in registers: d feat = dz @ lin1.weight below needs exactly these rows and columns
    //  again -- a second round trip to the weights, behind a data-dependent row selection, was 2 k cycles of the chain)
    float4 w4[16];
    {
      // lin1 (256 -> 128): wave w takes hidden units 16 w .. 16 w + 15.  One weight row (1 KB, 8 cache lines) per load
      // instruction, lane = 4 consecutive fan-in columns; the 16 per-lane partial dot products are then reduced over the
      // 64 lanes by a transposing butterfly (each step halves the values a lane holds) over lane bits 4..1 and two plain
      // steps over bits 0 and 5: lanes with (lane >> 1 & 15) = j end up with unit 16 w + j.  (A lane per row-half -- 64
      // cache lines per load instruction -- kept the head at ~10 k cycles.)
      const int ju = 16 * wave + ((lane >> 1) & 15), part = (lane & 1) | (lane >> 5);     // hidden unit of this lane; part 0 stores
      const float4 f4 = *(const float4*)(sfeat + 4 * lane);
      const float* wrow = P + a.off_l1w + (int64_t)(16 * wave) * 256 + 4 * lane;
      float v[16];
      {                                      // 16 rows: all 16 requests leave before the first use
#pragma unroll
        for (int q = 0; q < 16; ++q) w4[q] = *(const float4*)(wrow + q * 256);
        G2_SCHED_BARRIER();
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (w4[q].x * f4.x + w4[q].y * f4.y) + (w4[q].z * f4.z + w4[q].w * f4.w);
        G2_SCHED_BARRIER();
      }
      // (one literal stage per halving: a loop over the stages is not unrolled and turns v[] into select chains)
#define G2_BFLY(H)                                                                   \
      {                                                                              \
        const bool up = (lane & (2 * (H))) != 0;                                     \
        _Pragma("unroll") for (int i = 0; i < (H); ++i) {                            \
          const float send = up ? v[i] : v[i + (H)], keep = up ? v[i + (H)] : v[i];  \
          v[i] = keep + __shfl_xor(send, 2 * (H));                                   \
        }                                                                            \
      }
      G2_BFLY(8) G2_BFLY(4) G2_BFLY(2) G2_BFLY(1)
#undef G2_BFLY
      float s = v[0] + __shfl_xor(v[0], 1);
      s += __shfl_xor(s, 32);
      G2_STAMP(54);
      if (part == 0) {
        float av = s + hd_l1b;
        av = av > 0.f ? av : 0.f;
        int keep = 1;
        if (TRAIN) {
          keep = a.inj_mask ? (int)a.inj_mask[g * 128 + ju]
                            : (int)(igmc_u01(igmc_unit_hash(a.seed, step, (uint32_t)g, (uint32_t)ju)) >= 0.5f);
          if (cm == 0) {
            a.a1[g * 128 + ju] = av;
            a.lmask[g * 128 + ju] = (uint8_t)keep;
          }
          sa1[ju] = av;
          skeep[ju] = keep ? 1.f : 0.f;
        }
        sred[ju] = (TRAIN ? (keep ? av * 2.f : 0.f) : av) * hd_l2w;      // F.dropout(p = 0.5): kept * 2
      }
    }
    __syncthreads();
    if (wave == 0) {
      float s = sred[lane] + sred[lane + 64];
      s = igmc_wave_sum_f(s);
      if (lane == 0) {
        const float o = (s + hd_l2b) * a.mult;
        const float e = o - hd_y;
        if (cm == 0) {
          a.out[g] = o;
          a.err[g] = e;
        }
        misc[0] = e;
      }
    }
    if (!TRAIN) {
      first_graph = false;      // (a looping workgroup must not reuse the first subgraph's prefetched extents)
      __syncthreads();
      continue;
    }
    if (TRAIN) {
      __syncthreads();
      G2_STAMP(16);
      if (tid < 128) {
        const float dp = 2.f * misc[0] * a.grad_scale * a.mult;
        const float dzv = (sa1[tid] > 0.f && skeep[tid] != 0.f) ? dp * hd_l2wt * 2.f : 0.f;
        sdz[tid] = dzv;
        if (cm == 0) a.dz[g * 128 + tid] = dzv;
      }
      if (cm == 0 && tid < 256) a.feat[(size_t)g * a.D + tid] = sfeat[tid];
      __syncthreads();
      {   // d feat = dz @ lin1.weight: wave w takes hidden units 16 w .. 16 w + 15, lane -> 4 fan-in columns, from the rows
          // the forward left in registers (rows with dz == 0 -- ReLU / dropout: ~3/4 of them -- add exact zeros)
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* dz4 = (const float4*)(sdz + 16 * wave);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 d4 = dz4[q4];
          const float dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 wv = w4[4 * q4 + u];
            s4.x += dq[u] * wv.x; s4.y += dq[u] * wv.y; s4.z += dq[u] * wv.z; s4.w += dq[u] * wv.w;
          }
        }
        *(float4*)(TILES + wave * 256 + 4 * lane) = s4;
      }
      wload(3, 1);            // the first backward layer's image: lands under the dPre_3 set-up and the first backward gather
      __syncthreads();
      if (tid < 256) {
        const float v = ((TILES[tid] + TILES[256 + tid]) + (TILES[512 + tid] + TILES[768 + tid])) +
                        ((TILES[1024 + tid] + TILES[1280 + tid]) + (TILES[1536 + tid] + TILES[1792 + tid]));
        sgf[tid] = v;
        if (cm == 0) a.gfeat[(size_t)g * a.D + tid] = v;
      }
      __syncthreads();
      G2_STAMP(17);
      // ---- dPre_3: non-zero on the two centre rows only.  Own rows -> XO0 (h_3 sits in XO1), the opposite side's
      //      planes are rebuilt locally (node 0 of every feature; everything else zero): no exchange
      //      (only the first k-step -- nodes 0..31 of every term / feature row -- is read by layer 3's gather; the rest of
      //      the planes still holds h_2: finite values that the next exchange overwrites)
      for (int i = tid; i < nsides * G2_NT * 32 * 16; i += G2_THREADS) {
        const int s2 = i / (G2_NT * 32 * 16), r2 = i - s2 * (G2_NT * 32 * 16);
        PLN[s2 * lay.pside + (r2 >> 4) * (kp >> 1) + (r2 & 15)] = 0u;
      }
      for (int i = lane; i < 16 * G2_XP; i += 64) XO0[i] = 0.f;
      __syncthreads();
      if (tid < 32 * nsides) {
        const int s2 = tid >> 5, f = tid & 31;
        const int sd = (nsides == 2) ? s2 : 1 - side;
        const float hv = sfeat[sd * 128 + 96 + f];
        const float d = sgf[sd * 128 + 96 + f] * (1.f - hv * hv);
        uint32_t h, mi, lo;
        g2_split2(d, 0.f, h, mi, lo);
        uint32_t* p2 = PLN + s2 * lay.pside + (f * kp >> 1);
        p2[0] = h & 0xFFFFu;
        p2[32 * kp >> 1] = mi & 0xFFFFu;
        p2[2 * (32 * kp >> 1)] = lo & 0xFFFFu;
      }
      if (active && bi == 0 && lane < 32) {
        const float hv = sfeat[side * 128 + 96 + lane];
        XO0[lane] = sgf[side * 128 + 96 + lane] * (1.f - hv * hv);
      }
      __syncthreads();
      G2_STAMP(18);

      // ============================================================== conv layers 3..1, backward
#pragma unroll
      for (int l = 3; l >= 1; --l) {
        float* XOc = (l & 1) ? XO0 : XO1;            // dPre_l of the bundle's own rows
        float* XOn = (l & 1) ? XO1 : XO0;            // dPre_{l-1}
        float* wpart = a.ts_part + ((size_t)l * IGMC_TS_BLOCKS + tslot) * ts;
        {   // d bias_l = column sums of dPre_l over this workgroup's rows (fixed order)
          const int n = tid & 31, part = tid >> 5;
          float sb = 0.f;
          for (int row = part; row < G2_NB * 16; row += G2_THREADS / 32)
            sb += XOA[(((l & 1) ? 0 : G2_NB) + (row >> 4)) * 16 * G2_XP + (row & 15) * G2_XP + n];
          sred[part * 32 + n] = sb;
        }
        __syncthreads();
        if (tid < 32) {
          float s = 0.f;
          for (int p = 0; p < G2_THREADS / 32; ++p) s += sred[p * 32 + tid];
          if (first_graph) wpart[(R * 32 + 32) * 32 + tid] = s;
          else wpart[(R * 32 + 32) * 32 + tid] += s;
        }
        G2_STAMP(19 + 5 * (3 - l));
        float hreg[4];                               // h_{l-1}: rows 4 kq + rr, feature 16 hf + li
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) hreg[rr] = 0.f;
        f32x4 o[2];
        if (active) {
          // h_{l-1} of the bundle's rows (written by this very wave in the forward): tanh' and the table product
          {
            const float* hrow = a.h[l - 1] + (size_t)(nbs + row0 + 4 * kq) * 32 + 16 * hf + li;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (row0 + 4 * kq + rr < n_own) hreg[rr] = hrow[rr * 32];
          }
          int lane_ = lane;
          G2_OPAQUE(lane_);
          const int li_ = lane_ & 15, kq_ = lane_ >> 4;
          f32x4 acc[G2_NR];
          if (l == 2) G2_STAMP(43);
          g2_gather_h(pl, kp, (l == 3) ? 1 : nks, frag_b, li_, kq_, hf, acc);
          if (l == 2) G2_STAMP(44);
          // T' rows of the bundle -> LDS (B operand of the weight-gradient table): lane = row, 4 consecutive features of
          // this wave's tile
#pragma unroll
          for (int r = 0; r < G2_NR; ++r)
            *(float4*)(T + li * G2_TP + r * 32 + 16 * hf + 4 * kq) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) HS[(4 * kq + rr) * G2_XP + 16 * hf + li] = hreg[rr];
          // dX = [T' | dPre_l] @ [W_r^T ; root^T]: this wave's half of K, both output tiles
          if (l == 2) G2_STAMP(45);
          g2_transform_h(acc, XOc, (const uint32_t*)sW2, li_, kq_, hf, o);
          if (l == 2) G2_STAMP(46);
          *PXo = hf ? o[0] : o[1];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) G2_OPAQUE(hreg[rr]);      // landed here: behind the image request below a wait for
                                                                   // them would wait for the whole image as well
        } else if (hf == 0) {
          // idle bundle: its tile / h rows are K entries of the workgroup's table product
          for (int i = lane; i < 16 * G2_TP; i += 64) T[i] = 0.f;
          for (int i = lane; i < 16 * G2_XP; i += 64) HS[i] = 0.f;
        }
        __syncthreads();                             // the pair's partials are exchanged; tiles / h chunks complete
        if (l == 2) G2_STAMP(47);
        if (active) {
          // + readout gradient on the centre row, * tanh'(h_{l-1}): output features 16 hf + li of rows 4 kq + rr
          const f32x4 po = *PXi;
          const int f = 16 * hf + li;
          float v[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * kq + rr;
            float d = hf ? po[rr] + o[1][rr] : o[0][rr] + po[rr];
            if (bi == 0 && row == 0) d += sgf[side * 128 + (l - 1) * 32 + f];
            const float x = hreg[rr];
            v[rr] = (row0 + row < n_own) ? d * (1.f - x * x) : 0.f;
          }
          if (l > 1) {
            g2_publish_planes(px_of(6 - l, side), kp, f, row0 + 4 * kq, v);
            raise(6 - l);
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) XOn[(4 * kq + rr) * G2_XP + f] = v[rr];
        }
        G2_STAMP(20 + 5 * (3 - l));
        // (no barrier: the table product reads T' / h_{l-1} / dPre_l, complete at the barrier above; the dPre_{l-1} halves
        //  just written are ordered by the barrier at the end of the layer)
        G2_STAMP(21 + 5 * (3 - l));
        if (l > 1) wload(l - 1, 1);                  // next image: lands under the table product (LDS + matrix work only)
        {
          // weight-gradient table h_{l-1}^T [T' | dPre_l], split by OUTPUT tile: wave w computes 3 of the 2 x 12 tiles
          // (row half m2 = in-features, column tile nt: 0..9 = T' of relation nt >> 1, 10..11 = dPre -> d root) over
          // K = the 64 rows of the workgroup's four bundles -- no cross-wave reduction
          f32x4 w3[3];
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) w3[i3] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const int m2w = wave >> 2, wo = wave & 3;       // tiles 3 wo .. 3 wo + 2 of the row half
#pragma unroll 1
          for (int wb = 0; wb < G2_NB; ++wb) {
            const float* Tb = TILES + wb * 16 * G2_TP;
            const float* Hb = HSS + wb * 16 * G2_XP;
            const float* Db = XOA + (((l & 1) ? 0 : G2_NB) + wb) * 16 * G2_XP;
            float av[4], bw3[4][3];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              av[s4] = Hb[(4 * s4 + kq) * G2_XP + m2w * 16 + li];
              const float* tb = Tb + (4 * s4 + kq) * G2_TP + li;
              const float* db = Db + (4 * s4 + kq) * G2_XP + li;
#pragma unroll
              for (int i3 = 0; i3 < 3; ++i3) {
                const int nt = 3 * wo + i3;              // (wave-uniform: column tiles 10, 11 come from the dPre tile)
                bw3[s4][i3] = (nt < 2 * G2_NR) ? tb[nt * 16] : db[(nt - 2 * G2_NR) * 16];
              }
            }
            G2_SCHED_BARRIER();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
              for (int i3 = 0; i3 < 3; ++i3)
                w3[i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bw3[s4][i3], w3[i3], 0, 0, 0);
          }
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3) {
            const int m2 = m2w, nt = 3 * wo + i3;
            const int r = nt >> 1;                      // 32-column block: relation, or G2_NR = root
            if (r >= R && r < G2_NR) continue;
            float* pp = wpart + (kq * 4) * 32 + li + (r < R ? r : R) * 1024 + m2 * 512 + (nt & 1) * 16;
            if (first_graph) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = w3[i3][rr];
            } else {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) pp[rr * 32] += w3[i3][rr];
            }
          }
        }
        G2_STAMP(22 + 5 * (3 - l));
        // the opposite side's dPre_{l-1}: its waves raised their flags before their own table products (the exchange ran
        // under this one); global -> LDS, landed by the barrier below
        if (l > 1) fetch(6 - l);
        __syncthreads();
        G2_STAMP(23 + 5 * (3 - l));
      }

      // ============================================================== layer-0 table gradient (dPre_0 is in XO1)
      // T0'[c][f] = sum_i [hist | onehot | 1](i, c) dPre_0[i][f] over this workgroup's rows; wave = (code half, feature half)
      if (wave < 4) {
        const int m2 = wave >> 1, wn = wave & 1;
        for (int wb = 0; wb < G2_NB; ++wb) {
          const float* Hb = HIST + wb * 16 * G2_XP;
          const float* Db = XOA + (G2_NB + wb) * 16 * G2_XP;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Hb[(4 * s4 + kq) * G2_XP + m2 * 16 + li],
                                                        Db[(4 * s4 + kq) * G2_XP + wn * 16 + li], acc0, 0, 0, 0);
        }
      }
      first_graph = false;
      __syncthreads();
      G2_STAMP(34);
    }
  } while (!ONCE && (g += (int)gridDim.x) < B);

  if (TRAIN && wave < 4) {
    float* part0 = a.ts_part + (size_t)tslot * ts;             // slice 0 of [4][IGMC_TS_BLOCKS][ts]
    const int m2 = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int c = m2 * 16 + kq * 4 + rr;
      if (c < RL + L + 1) part0[c * 32 + wn * 16 + li] = acc0[rr];
    }
  }
  // The launch sequence number (exchange tags) advances once per launch, after every workgroup has read it: a training
  // launch leaves that to the next kernel on the stream (k_tail_ts, one store) unless a.self_seq says otherwise; else the
  // workgroup that finishes LAST does it here (an atomic round trip at the end of every workgroup).
  if (tid == 0 && a.self_seq) {
    const unsigned long long t1 = a.ts ? g2_wall_clock() : 0ull;
    if (g2_last_workgroup_advances(a.gs_bar) && a.ts) g2_clock_close(a.ts, t1);
  }
  G2_STAMP(35);
  if (a.timing && tid == 0 && blockIdx.x < 1024) g_g2_wg[blockIdx.x][1] = g2_wall_clock();
}

// debug aid: start / end (wall clock, 100 MHz) and XCC of every workgroup of the last k_graph_step2 launched with IGMC_GS_TIMING
extern "C" int igmc_debug_g2_wg_clocks(unsigned long long* out, int n_wg) {
#ifndef IGMC_HIPEMU
  if (n_wg > 1024) n_wg = 1024;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g2_wg), (size_t)n_wg * 3 * sizeof(unsigned long long)) != hipSuccess) return 1;
#else
  for (int i = 0; i < 3 * n_wg; ++i) out[i] = 0;
#endif
  return 0;
}

// debug aid: phase clocks (shader cycles) of the last k_graph_step2 launched with IGMC_GS_TIMING set
extern "C" int igmc_debug_g2_clocks(unsigned long long* out, int n) {
#ifndef IGMC_HIPEMU
  if (n > 128) n = 128;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g2_clk), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return 1;
#else
  for (int i = 0; i < n; ++i) out[i] = 0;
#endif
  return 0;
}
