// g2_compose.h -- k_g2_compose (the step's weight images and layer-0 table) and the host side of the subgraph kernel: LDS
// layout, cluster size, XCD probe, eligibility, launch (included once, by graphstep2.hip)
#pragma once

// Weights-only part of the step, formed ONCE per launch instead of by every workgroup and layer pass: the B operands
// [W_0; ..; W_4; root] of the three conv layers (W_r = sum_b att[r,b] basis_b) in the LDS image of the forward
// (element (k, n) at [k][n & 15].{x: n < 16, y: n >= 16}, rows padded to G2_WP float2) and of the backward (their
// transposes), and the layer-0 table [W0[r*L + c] | root0[c] | bias0].  37 small workgroups: the launch is as long as one
// round trip to the weights plus three 8-byte stores per thread.
__global__ __launch_bounds__(G2C_THREADS) void k_g2_compose(ModelDev m, const float* P, float* w) {
  __shared__ float s_att[4 * G2_NR * G2_NG_MAX];
  const int tid = threadIdx.x, R = m.R, L = m.L, RL = R * L, LF = L * 32;
  // blockIdx.x: 2 * (3 layers x relation groups x 6 matrices) image blocks (bit 0 = transposed), then the layer-0 table blocks
  // (32 rows each)
  const int ng = g2_groups(R, L);
  const int tableb = 2 * 3 * ng * (G2_NR + 1);
  const int mi = (int)(blockIdx.x >> 1);                   // matrix of the images: (layer, group, block)
  const int l = ((int)blockIdx.x >= tableb) ? 0 : 1 + mi / (ng * (G2_NR + 1)), trans = blockIdx.x & 1;
  // every global load of the block is requested before the first use (one round trip)
  if ((int)blockIdx.x >= tableb) {
    const int c0 = 32 * ((int)blockIdx.x - tableb);        // first table row of this block
    const float attv = (tid < R * 4) ? P[m.off_att[0] + tid] : 0.f;
    float bv[4][4], rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + q * G2C_THREADS, c = c0 + (i >> 5), f = i & 31;
      const int cf = (c < RL) ? (c % L) * 32 + f : 0;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) bv[q][bb] = (c < RL) ? P[m.off_basis[0] + bb * LF + cf] : 0.f;
      rv[q] = (c >= RL && c < RL + L) ? P[m.off_root[0] + (c - RL) * 32 + f] : ((c == RL + L) ? P[m.off_bias[0] + f] : 0.f);
    }
    if (tid < 4 * G2_NR * G2_NG_MAX) s_att[tid] = attv;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + q * G2C_THREADS, c = c0 + (i >> 5);
      float sacc = rv[q];
      if (c < RL) {
        const int r = c / L;
        sacc = g2_wsum(s_att[r * 4], s_att[r * 4 + 1], s_att[r * 4 + 2], s_att[r * 4 + 3], bv[q][0], bv[q][1], bv[q][2], bv[q][3]);
      }
      w[g2_t0_off(ng) + (size_t)c0 * 32 + i] = sacc;
    }
    return;
  }
  // image blocks: one relation (G2_NR = the root matrix) of one image per workgroup; a thread takes the four
  // consecutive k of one column n, which are four consecutive bf16 of one lane's fragment: one 8-byte store per term
  const int grp = (mi / (G2_NR + 1)) % ng, blk = mi % (G2_NR + 1);
  uint16_t* img = (uint16_t*)(w + g2_img_off(ng, l, trans, grp));
  // block blk of group grp: relation G2_NR grp + blk; block G2_NR = root (group 0) / zero (the other groups)
  const int r = (blk == G2_NR) ? ((grp == 0) ? -1 : R) : G2_NR * grp + blk;
  const int n = tid & 31, kg = tid >> 5;                  // element (k = 4 kg + q, n) of the block's B operand
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (r < 0) {
    const float* root = P + m.off_root[l];
    if (trans) {
      const float4 r4 = *(const float4*)(root + n * 32 + 4 * kg);
      v[0] = r4.x; v[1] = r4.y; v[2] = r4.z; v[3] = r4.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = root[(4 * kg + q) * 32 + n];
    }
  } else if (r < R) {
    const float* basis = P + m.off_basis[l];
    float bq[4][4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      if (trans) {
        const float4 b4 = *(const float4*)(basis + bb * 1024 + n * 32 + 4 * kg);
        bq[bb][0] = b4.x; bq[bb][1] = b4.y; bq[bb][2] = b4.z; bq[bb][3] = b4.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[bb][q] = basis[bb * 1024 + (4 * kg + q) * 32 + n];
      }
    }
    float at[4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) at[bb] = P[m.off_att[l] + r * 4 + bb];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = g2_wsum(at[0], at[1], at[2], at[3], bq[0][q], bq[1][q], bq[2][q], bq[3][q]);
  }
  uint32_t h01, m01, l01, h23, m23, l23;
  g2_split2(v[0], v[1], h01, m01, l01);
  g2_split2(v[2], v[3], h23, m23, l23);
  const int lane = (kg & 3) * 16 + (n & 15), nt = n >> 4, e0 = 4 * (kg >> 2);
  const uint32_t t3[3][2] = {{h01, h23}, {m01, m23}, {l01, l23}};
#pragma unroll
  for (int t = 0; t < G2_NT; ++t)
    *(uint2*)(img + ((size_t)(((t * (G2_NR + 1) + blk) * 2 + nt) * 64 + lane)) * 8 + e0) = make_uint2(t3[t][0], t3[t][1]);
}

// ------------------------------------------------------------------------------------------------ host side
// workgroups per subgraph: 4 (2) when 4 (2) x the padded batch still fits one workgroup per CU with a margin
int igmc_gs_cluster(int B) {
#ifdef IGMC_HIPEMU
  // the emulator runs workgroups one after the other unless a test asks for clusters (their members then run
  // together: hipemu::Runtime::co_cs)
  const char* ee = getenv("IGMC_GS_CLUSTER");
  const int want_e = ee ? atoi(ee) : 1;
  const int stride_e = (B + 7) & ~7;
  if (want_e >= 4 && 4 * stride_e <= 224) return 4;
  if (want_e >= 2 && 2 * stride_e <= 224) return 2;
  return 1;
#else
  static int cus = -1;
  if (cus < 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 0;
  }
  int want = (cus >= 240) ? 4 : 1;        // the clustered launch needs (almost) every CU of an MI355X to itself
  const char* e = getenv("IGMC_GS_CLUSTER");
  if (e) want = atoi(e);
  const int stride = (B + 7) & ~7;
  if (want >= 4 && 4 * stride <= 224) return 4;
  if (want >= 2 && 2 * stride <= 224) return 2;
  return 1;
#endif
}

int igmc_gs_grid(int B) {
  int cap = IGMC_WG_BLOCKS;
  const char* e = getenv("IGMC_GS_GRID");      // test hook: fewer workgroups than graphs (accumulating partials)
  if (e && atoi(e) > 0 && atoi(e) < cap) cap = atoi(e);
  return B < cap ? B : cap;
}

// LDS plan + eligibility for a batch arena / cluster size
int igmc_g2_layout(const ModelDev& m, const BatchDev& b, int cs, G2Layout* lay) {
  const int RL = m.R * m.L;
  if (m.S != 0 || m.D != 256 || m.R > G2_NR || m.L > 8 || RL + m.L + 1 > 32 || !m.ts_part || !m.g2_px || !m.g2_fx || !m.g2_w || !b.relm) return 0;
  const int half = 2 * cs;
  const int cmax = b.cap_u > b.cap_v ? b.cap_u : b.cap_v;
  if (cmax > 16 * half || cmax > 128) return 0;
  if (b.graph_cap > m.g2_graphs) return 0;
  const int kmax = ((cmax + 31) >> 5) << 5;
  lay->kp = kmax + 8;
  lay->nsides = (cs == 1) ? 2 : 1;
  lay->rmr = ((cmax + 15) >> 4) << 4;
  lay->rmc = kmax;
  int o = 0;
  lay->pside = ((G2_NT * 32 * lay->kp * 2 + 1023) & ~1023) >> 2;      // words of one side's planes, padded to 1 KB pieces
  lay->planes = o; o += lay->nsides * lay->pside;
  lay->ohp = o; o += lay->nsides * (8 * lay->kp >> 1);
  lay->lab = o; o += 64;
  lay->xo = o; o += 2 * G2_NB * 16 * G2_XP;
  lay->hs = o; o += G2_NB * 16 * G2_XP;
  int tw = G2_NB * 16 * G2_TP;
  if (tw < (G2_THREADS / 64) * 256) tw = (G2_THREADS / 64) * 256;      // (d feat partials of the eight waves)
  const int rw = 2 * lay->rmr * (lay->rmc + 8) / 4;
  if (rw > tw) tw = rw;
  if (tw < 1024) tw = 1024;
  lay->tile = o; o += tw;
  lay->hist = o; o += G2_NB * 16 * G2_XP;
  lay->px = o; o += G2_NB * 2 * 64 * 4;
  lay->wreg = o; o += G2_WIMG;
  lay->t0 = o; o += 1024;
  lay->att = o; o += 64;
  lay->head = o; o += 256 + 256 + 3 * 128 + 512 + 16;
  lay->words = o;
  return (size_t)o * 4 <= 160 * 1024;
}

// The plane exchange of k_graph_step2 goes through the L2 of ONE XCD: workgroups b and b + 8 of a launch must sit on the
// same XCD (round-robin dispatch over the eight XCDs of an MI355X in SPX mode; trivially true on a one-XCD partition).
// Checked once per DEVICE on the device itself -- the hardware XCC id of every workgroup of a 64-workgroup launch -- and
// neither the subgraph kernel nor the one-launch dense-layer kernels (k_dl_fwd / k_dl_bwd: the same exchange since round 5)
// are used where it does not hold: the per-layer kernels run.
__global__ void k_g2_xcc_probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(g2_xcc_id() & 15ull);
}
int igmc_g2_xcd_ok() {
#ifdef IGMC_HIPEMU
  return 1;
#else
  static int oks[64];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 64; ++i) oks[i] = -1;
    init = true;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int& ok = oks[dev];
  if (ok >= 0) return ok;
  ok = 0;
  int* d = nullptr;
  int h[64];
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return ok;
  hipLaunchKernelGGL(k_g2_xcc_probe, dim3(64), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
    ok = 1;
    for (int b = 0; b + 8 < 64; ++b)
      if (h[b] != h[b + 8]) ok = 0;
  }
  (void)hipFree(d);
  if (!ok) fprintf(stderr, "[igmc] device %d: workgroups b and b + 8 of a launch do not share an XCD: the subgraph kernel and the one-launch dense layers are not used\n", dev);
  return ok;
#endif
}

// 1 = the matrix-core subgraph kernel takes this batch configuration (IGMC_GRAPH_STEP=0 forces the per-layer kernels)
int igmc_g2_eligible(const ModelDev& m, const BatchDev& b, int B, G2Layout* lay, int* cs_out) {
  const char* en = getenv("IGMC_GRAPH_STEP");      // read on every call: tests switch it per case
  if (en && atoi(en) == 0) return 0;
  if (!igmc_g2_xcd_ok()) return 0;
  const int cs = igmc_gs_cluster(B);
  if (!igmc_g2_layout(m, b, cs, lay)) return 0;
  *cs_out = cs;
  return 1;
}

// returns 1 when the launch leaves the advance of the launch sequence number to the caller's next kernel (k_tail_ts)
int g_igmc_compose_count = 0;      // launches of k_g2_compose so far (capi.hip: did a call refresh the weight images)

int igmc_launch_graph_step2(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                            const G2Layout& lay, int cs, const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult,
                            float grad_scale, float* out, void* stream) {
  G2Args a;
  memset(&a, 0, sizeof(a));
  a.n_users = b.n_users; a.n_items = b.n_items; a.B = B; a.s_lab = b.s_lab; a.relm = b.relm; a.y = b.y;
  a.cap_u = b.cap_u; a.cap_v = b.cap_v; a.slot = b.slot; a.relm_ld = b.relm_ld; a.graph_cap = b.graph_cap;
  a.R = m.R; a.L = m.L; a.D = m.D; a.ts_stride = m.ts_stride;
  for (int l = 0; l < 4; ++l) {
    a.h[l] = m.h[l];
    a.off_bias[l] = (int)m.off_bias[l];
  }
  a.ts_part = m.ts_part; a.g2_px = m.g2_px; a.g2_fx = m.g2_fx; a.g2_px_stride = m.g2_px_stride; a.g2_w = m.g2_w;
  a.gs_bar = m.gs_bar; a.gs_err = m.gs_err; a.a1 = m.a1; a.dz = m.dz; a.feat = m.feat; a.gfeat = m.gfeat; a.err = m.err;
  a.lmask = m.lmask; a.ctrl = m.ctrl;
  a.off_l1w = (int)m.off_l1w; a.off_l1b = (int)m.off_l1b; a.off_l2w = (int)m.off_l2w; a.off_l2b = (int)m.off_l2b;
  a.P = P;
  a.inj_mask = inj_mask;
  a.seed = seed;
  a.step = step;
  a.mult = mult;
  a.grad_scale = grad_scale;
  a.out = out;
  a.lay2 = lay;
  a.timing = getenv("IGMC_GS_TIMING") ? 1 : 0;
  a.ts = (g_igmc_prof_on == 2) ? m.gs_ts : nullptr;
  // (the device-side launch clock rides in the last workgroup's counter; evaluation launches have no following kernel)
  a.self_seq = (!training || a.ts) ? 1 : 0;
  a.cs = cs;
  a.stride = (cs > 1) ? ((B + 7) & ~7) : 1;
  const int grid = (cs > 1) ? cs * 8 * ((B + 7) / 8) : igmc_gs_grid(B);      // (clusters in XCD-aligned blocks of 8 cs workgroups)
  const size_t sm = (size_t)lay.words * 4;
  if (!m.img_current) {
    IGMC_PLAUNCH("k_g2_compose", k_g2_compose, 2 * 3 * g2_groups(m.R, m.L) * (G2_NR + 1) + g2_t0_rows(m.R, m.L) / 32, G2C_THREADS, 0, stream, m, P, m.g2_w);
    ++g_igmc_compose_count;
  }
#ifdef IGMC_HIPEMU
  if (cs > 1) {        // (after the launch above: a launch consumes the co-residency request)
    hipemu::rt().co_cs = cs;
    hipemu::rt().co_stride = 8;        // members of cluster (j, x) = workgroups 8 cs j + x + 8 c
    hipemu::rt().co_block = 8 * cs;
  }
#endif
  if (getenv("IGMC_GS_TRACE")) fprintf(stderr, "[igmc] k_graph_step B=%d train=%d flags=%d v2 kp=%d lds=%zu cluster=%d grid=%d\n", B, training, use_flags, lay.kp, sm, cs, grid);
  if (cs > 1) {        // one subgraph per workgroup: the straight-line variants
    if (training) {
      if (use_flags) IGMC_PLAUNCH("k_graph_step", (k_graph_step2<true, true, true>), grid, G2_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_graph_step", (k_graph_step2<false, true, true>), grid, G2_THREADS, sm, stream, a);
    } else {
      if (use_flags) IGMC_PLAUNCH("k_graph_fwd", (k_graph_step2<true, false, true>), grid, G2_THREADS, sm, stream, a);
      else IGMC_PLAUNCH("k_graph_fwd", (k_graph_step2<false, false, true>), grid, G2_THREADS, sm, stream, a);
    }
  } else if (training) {
    if (use_flags) IGMC_PLAUNCH("k_graph_step", (k_graph_step2<true, true, false>), grid, G2_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_graph_step", (k_graph_step2<false, true, false>), grid, G2_THREADS, sm, stream, a);
  } else {
    if (use_flags) IGMC_PLAUNCH("k_graph_fwd", (k_graph_step2<true, false, false>), grid, G2_THREADS, sm, stream, a);
    else IGMC_PLAUNCH("k_graph_fwd", (k_graph_step2<false, false, false>), grid, G2_THREADS, sm, stream, a);
  }
  return a.self_seq ? 0 : 1;
}

int igmc_dl_prepare();
int igmc_gs_prepare() { return igmc_g2_prepare(); }
int igmc_g2_prepare() {
  if (igmc_dl_prepare()) return 1;
  (void)igmc_g2_xcd_ok();      // (probed here, at model creation: never inside a stream capture)
#ifndef IGMC_HIPEMU
  const int mx = 160 * 1024;
  const void* fns[8] = {(const void*)k_graph_step2<true, true, true>,   (const void*)k_graph_step2<false, true, true>,
                        (const void*)k_graph_step2<true, false, true>,  (const void*)k_graph_step2<false, false, true>,
                        (const void*)k_graph_step2<true, true, false>,  (const void*)k_graph_step2<false, true, false>,
                        (const void*)k_graph_step2<true, false, false>, (const void*)k_graph_step2<false, false, false>};
  for (const void* fn : fns)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
#endif
  return 0;
}
