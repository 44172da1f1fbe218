// extract.hip -- enclosing-subgraph extraction, labelling and collation on gfx950.
//
// Replaces, for a whole batch and entirely in HBM/LDS:
//   subgraph_extraction_labeling   reference util_functions.py:208-277
//   neighbors                      reference util_functions.py:300-304
//   construct_pyg_graph            reference util_functions.py:280-297
//   PyG Batch.from_data_list       call site reference train_eval.py:44-51
//
// One 256-thread workgroup per (user,item) link.  The rating graph (CSR + CSC,
// ~9 MB for ml_1m) is L2 / Infinity-Cache resident, so this stage is integer work
// bounded by latency and LDS, not by HBM: all per-subgraph state lives in LDS
// bitmaps over the user / item id spaces --
//   vis  visited set          (reference :218-221, updated BEFORE sampling)
//   new  this hop's fringe    (reference :217)
//   sel  nodes kept so far    (u_nodes / v_nodes minus the targets)
//   pre  per-word exclusive popcount of sel  -> local index = 1 + rank (target = 0)
// Neighbour rows are read with 256 coalesced lanes; membership of an id in the
// selected set is one LDS word + one popcount; there are no atomics on floats and
// no order-dependent writes, so the output is bit-reproducible.
//
// Kernels:  k_extract_nodes (BFS + per-hop sampling -> node lists; one workgroup per link)
//           k_relm + k_emit (capped extraction: dense induced block from the user rows, then CSR emission)
//           k_count         (uncapped fallback: induced degree of every selected node; S workgroups per link)
//           k_fill          (batch offsets + dst-sorted CSR of the batch, labels, PyG `batch` vector; S per link)
//           k_edge_flags    (edge dropout keep flags, reference models.py:193-198)
#include "launch.h"

struct ExtractArgs {
  GraphDev g;
  BatchDev b;
  const int32_t* link_u;
  const int32_t* link_v;
  const float* link_y;
  const int32_t* link_idx;
  int first, B, replay;
  int split;             // 1: grid (B, 2), one workgroup per SIDE of a link (hop 1: the two fringes are independent)
  double sample_ratio;
  uint64_t seed, epoch;
  const int64_t* ctrl;   // optional device-side step control (first / epoch), see igmc_hip.h
};

// ---------------------------------------------------------------- bitmap helpers
__device__ __forceinline__ void bm_clear(uint32_t* a, int W) {
  for (int w = threadIdx.x; w < W; w += IGMC_BLOCK) a[w] = 0u;
}
__device__ __forceinline__ bool bm_test(const uint32_t* a, int j) { return (a[j >> 5] >> (j & 31)) & 1u; }
__device__ __forceinline__ int bm_rank(const uint32_t* sel, const uint32_t* pre, int j) {
  return (int)pre[j >> 5] + __popc(sel[j >> 5] & ((1u << (j & 31)) - 1u));
}

// mark the neighbours of list[lo,hi) in `out` (reference neighbors(), :300-304)
__device__ void expand_fringe(const int32_t* list, int lo, int hi, const int32_t* ptr,
                              const int32_t* idx, uint32_t* out) {
  for (int f = lo; f < hi; ++f) {
    const int n = list[f];
    const int beg = ptr[n], end = ptr[n + 1];
    for (int p = beg + (int)threadIdx.x; p < end; p += IGMC_BLOCK) {
      const int j = idx[p];
      atomicOr(&out[j >> 5], 1u << (j & 31));
    }
  }
}

// exclusive per-word popcount prefix of a bitmap; returns the total
__device__ int bm_prefix(const uint32_t* bm, uint32_t* pre, int W, int* sm) {
  int running = 0;
  for (int base = 0; base < W; base += IGMC_BLOCK) {
    const int w = base + threadIdx.x;
    const int c = (w < W) ? __popc(bm[w]) : 0;
    int tot;
    const int ex = igmc_block_scan_excl(c, &tot, sm);
    if (w < W) pre[w] = (uint32_t)(running + ex);
    running += tot;
  }
  return running;
}

// k-th smallest (1-based) sampling key among the set bits of bm.  Keys are a bijection of
// the id, so exactly k candidates have key <= the returned threshold.
__device__ uint32_t radix_select(const uint32_t* bm, int W, int k, uint64_t salt, int* hist, int* sm) {
  uint32_t prefix = 0u, mask = 0u;
  int remaining = k;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += IGMC_BLOCK) {
      uint32_t bits = bm[w];
      while (bits) {
        const int bp = __ffs((int)bits) - 1;
        bits &= bits - 1u;
        const uint32_t key = igmc_sample_key(salt, (uint32_t)(w * 32 + bp));
        if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    const int c = hist[threadIdx.x];
    int tot;
    const int ex = igmc_block_scan_excl(c, &tot, sm);
    if (c > 0 && ex < remaining && remaining <= ex + c) {
      sm[8] = (int)threadIdx.x;
      sm[9] = remaining - ex;
    }
    __syncthreads();
    prefix |= ((uint32_t)sm[8]) << shift;
    mask |= 0xFFu << shift;
    remaining = sm[9];
    __syncthreads();
  }
  return prefix;
}

#ifdef IGMC_HIPEMU
// (CPU emulation only: the tests lower the bound through the environment to drive the four-pass path)
static inline int igmc_sample_park() { const char* e = getenv("IGMC_SAMPLE_PARK"); return e ? atoi(e) : IGMC_BLOCK; }
#define IGMC_SAMPLE_PARK igmc_sample_park()
#else
#define IGMC_SAMPLE_PARK IGMC_BLOCK
#endif
// keep only the k candidates with the smallest keys (uniform k-subset, reference :222-229).  Two walks over the candidate
// bits: (1) histogram of the keys' top byte -> the byte b that holds the k-th smallest key and how many (r) of its keys are
// wanted; (2) keep every key below b outright and park the keys OF b (cnt / 256 of them on average) in a short list, whose
// r smallest are then found by counting -- each parked key against the others -- and set again.  A list that would not
// fit (more than 256 keys in one byte of a hash: not seen; tests force it) goes through radix_select's four passes instead;
// the kept set is the same either way: the k smallest keys, which never tie (igmc_rng.h).
__device__ void sample_fringe(uint32_t* bm, int W, int cnt, int k, uint64_t salt, int* hist, int* sm) {
  if (k >= cnt) return;
  if (k <= 0) {
    bm_clear(bm, W);
    __syncthreads();
    return;
  }
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (int w = threadIdx.x; w < W; w += IGMC_BLOCK) {
    uint32_t bits = bm[w];
    while (bits) {
      const int bp = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      atomicAdd(&hist[igmc_sample_key(salt, (uint32_t)(w * 32 + bp)) >> 24], 1);
    }
  }
  __syncthreads();
  const int c = hist[threadIdx.x];
  int tot;
  const int ex = igmc_block_scan_excl(c, &tot, sm);
  if (c > 0 && ex < k && k <= ex + c) {
    sm[8] = (int)threadIdx.x;
    sm[9] = k - ex;
    sm[10] = c;
  }
  if (threadIdx.x == 0) sm[11] = 0;
  __syncthreads();
  const uint32_t b = (uint32_t)sm[8];
  const int r = sm[9], cb = sm[10];
  if (cb > IGMC_SAMPLE_PARK) {
    __syncthreads();
    const uint32_t T = radix_select(bm, W, k, salt, hist, sm);
    for (int w = threadIdx.x; w < W; w += IGMC_BLOCK) {
      uint32_t bits = bm[w], keep = 0u;
      while (bits) {
        const int bp = __ffs((int)bits) - 1;
        bits &= bits - 1u;
        if (igmc_sample_key(salt, (uint32_t)(w * 32 + bp)) <= T) keep |= 1u << bp;
      }
      bm[w] = keep;
    }
    __syncthreads();
    return;
  }
  uint32_t* ckey = (uint32_t*)hist;          // (every count was read before the scan's barriers)
  int* cid = hist + IGMC_BLOCK;
  for (int w = threadIdx.x; w < W; w += IGMC_BLOCK) {
    uint32_t bits = bm[w], keep = 0u;
    while (bits) {
      const int bp = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      const uint32_t key = igmc_sample_key(salt, (uint32_t)(w * 32 + bp));
      if ((key >> 24) < b) keep |= 1u << bp;
      else if ((key >> 24) == b) {
        const int s = atomicAdd(&sm[11], 1);
        ckey[s] = key;
        cid[s] = w * 32 + bp;
      }
    }
    bm[w] = keep;
  }
  __syncthreads();
  if ((int)threadIdx.x < cb) {
    const uint32_t mine = ckey[threadIdx.x];
    int below = 0;
    for (int j = 0; j < cb; ++j) below += ckey[j] < mine;
    const int id = cid[threadIdx.x];
    if (below < r) atomicOr(&bm[id >> 5], 1u << (id & 31));
  }
  __syncthreads();
}

// append the set bits of bm (ascending id) to list[cnt..], tag them with `dist`, OR into sel.  SLOTS (one-hop split
// launch): the fringe is the whole selection of this side and arrives in ascending id, so position in the list IS the
// local index -- the node slot (s_gid, s_lab) is written here and the rank pass after the hops has nothing left to do.
template <bool SLOTS>
__device__ int append_fringe(const uint32_t* bm, uint32_t* sel, int W, int32_t* list, uint8_t* dists,
                             int cnt, int dist, int* sm, int32_t* sg = nullptr, uint8_t* sl = nullptr, int side = 0) {
  int running = cnt;
  for (int base = 0; base < W; base += IGMC_BLOCK) {
    const int w = base + threadIdx.x;
    uint32_t bits = (w < W) ? bm[w] : 0u;
    int tot;
    int o = running + igmc_block_scan_excl(__popc(bits), &tot, sm);
    if (!SLOTS && w < W) sel[w] |= bits;
    while (bits) {
      const int bp = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      list[o] = w * 32 + bp;
      dists[o] = (uint8_t)dist;
      if (SLOTS) {
        sg[o] = w * 32 + bp;
        sl[o] = (uint8_t)(2 * dist + side);        // reference :245
      }
      ++o;
    }
    running += tot;
  }
  return running;
}

// ---------------------------------------------------------------- kernel 1
__device__ __forceinline__ void extract_nodes_body(const ExtractArgs& a) {
  IGMC_DYN_SMEM(smem);
  __shared__ int hist[2 * IGMC_BLOCK];      // sample_fringe: 256 counts, then keys | ids of the deciding byte
  __shared__ int sm[16];
  const int g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wu = (a.g.n_users + 31) >> 5, Wv = (a.g.n_items + 31) >> 5;
  uint32_t* vis_u = (uint32_t*)smem;
  uint32_t* new_u = vis_u + Wu;
  uint32_t* sel_u = new_u + Wu;
  uint32_t* pre_u = sel_u + Wu;
  uint32_t* vis_v = pre_u + Wu;
  uint32_t* new_v = vis_v + Wv;
  uint32_t* sel_v = new_v + Wv;
  uint32_t* pre_v = sel_v + Wv;

  const int cap_u = a.b.cap_u;
  const size_t so = (size_t)g * a.b.slot;
  int32_t* tl = a.b.t_list + so;
  uint8_t* td = a.b.t_dist + so;
  int32_t* sg = a.b.s_gid + so;
  uint8_t* sl = a.b.s_lab + so;
  int32_t* sd = a.b.s_deg + so;

  // hop 1 (a.split): the user side of a link (raters of the target item) and its item side (items of the target user)
  // never meet before the dense block is formed, so each gets its own workgroup -- half the dependent chain
  const bool do_u = !a.split || blockIdx.y == 0, do_v = !a.split || blockIdx.y == 1;
  int cu, cv, u0, v0;
  if (do_u) bm_clear(sel_u, Wu);
  if (do_v) bm_clear(sel_v, Wv);
  if (!a.replay) {
    // with a control block attached the host `first` is a selector q | (i << 1): batch i of the group of parity q (see
    // igmc_hip.h) -- a prefetch of the next group on another stream reads the cursor the concurrent steps never write
    const int first = a.ctrl ? igmc_ctrl_first(a.ctrl, a.first) : a.first;
    const uint64_t epoch = a.ctrl ? (uint64_t)igmc_ctrl_ld(a.ctrl + IGMC_CTRL_EPOCH) : a.epoch;
    const int pos = a.link_idx ? a.link_idx[first + g] : first + g;
    if (g == 0 && tid == 0 && do_u) {      // stamp of the batch in this arena (checked by the tick of the consuming step)
      a.b.stamp[0] = a.ctrl ? (int64_t)first : -1;      // (no control block: nothing to compare with)
      a.b.stamp[1] = -1;
    }
    u0 = a.link_u[pos];
    v0 = a.link_v[pos];
    if (do_u) bm_clear(vis_u, Wu);
    if (do_v) bm_clear(vis_v, Wv);
    __syncthreads();
    if (tid == 0) {
      if (do_u) {
        vis_u[u0 >> 5] |= 1u << (u0 & 31);
        tl[0] = u0;
        td[0] = 0;
        a.b.y[g] = a.link_y[pos];
        if (a.split) { sg[0] = u0; sl[0] = 0; }
      }
      if (do_v) {
        vis_v[v0 >> 5] |= 1u << (v0 & 31);
        tl[cap_u] = v0;
        td[cap_u] = 0;
        if (a.split) { sg[cap_u] = v0; sl[cap_u] = 1; }
      }
    }
    cu = 1;
    cv = 1;
    int fu_lo = 0, fu_hi = 1, fv_lo = 0, fv_hi = 1;
    __syncthreads();
    for (int dist = 1; dist <= a.b.hop; ++dist) {
      if (do_u) bm_clear(new_u, Wu);
      if (do_v) bm_clear(new_v, Wv);
      __syncthreads();
      // simultaneous swap (reference :217): users' rows give item candidates and vice versa.  (Split launch, hop 1: the
      // fringes are the targets themselves, read from the link arrays -- the other side's list belongs to another workgroup.)
      if (a.split) {
        if (do_v) expand_fringe(&u0, 0, 1, a.g.u_ptr, a.g.u_idx, new_v);
        if (do_u) expand_fringe(&v0, 0, 1, a.g.v_ptr, a.g.v_idx, new_u);
      } else {
        expand_fringe(tl, fu_lo, fu_hi, a.g.u_ptr, a.g.u_idx, new_v);
        expand_fringe(tl + cap_u, fv_lo, fv_hi, a.g.v_ptr, a.g.v_idx, new_u);
      }
      __syncthreads();
      int c = 0;
      int cnt_u = 0, cnt_v = 0;
      if (do_u) {
        for (int w = tid; w < Wu; w += IGMC_BLOCK) {
          const uint32_t x = new_u[w] & ~vis_u[w];
          new_u[w] = x;
          vis_u[w] |= x;   // visited updated before sampling (reference :220-221)
          c += __popc(x);
        }
        cnt_u = igmc_block_sum_i(c, sm);
      }
      c = 0;
      if (do_v) {
        for (int w = tid; w < Wv; w += IGMC_BLOCK) {
          const uint32_t x = new_v[w] & ~vis_v[w];
          new_v[w] = x;
          vis_v[w] |= x;
          c += __popc(x);
        }
        cnt_v = igmc_block_sum_i(c, sm);
      }
      int ku = cnt_u, kv = cnt_v;
      if (a.sample_ratio < 1.0) {   // int(sample_ratio*len), reference :222-224
        ku = (int)(a.sample_ratio * (double)cnt_u);
        kv = (int)(a.sample_ratio * (double)cnt_v);
      }
      if (a.b.max_nodes_per_hop >= 0) {   // reference :225-229
        if (a.b.max_nodes_per_hop < ku) ku = a.b.max_nodes_per_hop;
        if (a.b.max_nodes_per_hop < kv) kv = a.b.max_nodes_per_hop;
      }
      const uint64_t link_uid = (uint64_t)(uint32_t)pos;
      if (do_u) sample_fringe(new_u, Wu, cnt_u, ku, igmc_sample_salt(a.seed, epoch, link_uid, dist, 0), hist, sm);
      if (do_v) sample_fringe(new_v, Wv, cnt_v, kv, igmc_sample_salt(a.seed, epoch, link_uid, dist, 1), hist, sm);
      if (ku == 0 && kv == 0) break;   // reference :230-231 (a split launch has one hop: nothing follows either way)
      int ncu = cu, ncv = cv;
      if (a.split) {
        if (do_u) ncu = append_fringe<true>(new_u, sel_u, Wu, tl, td, cu, dist, sm, sg, sl, 0);
        if (do_v) ncv = append_fringe<true>(new_v, sel_v, Wv, tl + cap_u, td + cap_u, cv, dist, sm, sg + cap_u, sl + cap_u, 1);
      } else {
        ncu = append_fringe<false>(new_u, sel_u, Wu, tl, td, cu, dist, sm);
        ncv = append_fringe<false>(new_v, sel_v, Wv, tl + cap_u, td + cap_u, cv, dist, sm);
      }
      fu_lo = cu; fu_hi = ncu; cu = ncu;
      fv_lo = cv; fv_hi = ncv; cv = ncv;
      __syncthreads();
    }
  } else {
    cu = a.b.n_users[g];
    cv = a.b.n_items[g];
    u0 = tl[0];
    v0 = tl[cap_u];
    __syncthreads();
    for (int i = 1 + tid; i < cu; i += IGMC_BLOCK) atomicOr(&sel_u[tl[i] >> 5], 1u << (tl[i] & 31));
    for (int i = 1 + tid; i < cv; i += IGMC_BLOCK) atomicOr(&sel_v[tl[cap_u + i] >> 5], 1u << (tl[cap_u + i] & 31));
  }
  __syncthreads();

  // local index = 1 + rank among the selected ids (ascending); targets are local 0.  (A free-running split launch wrote
  // its slots while appending: one hop, one fringe, already in id order.)
  if (!a.split || a.replay) {
    if (do_u) bm_prefix(sel_u, pre_u, Wu, sm);
    if (do_v) bm_prefix(sel_v, pre_v, Wv, sm);
    __syncthreads();
    if (do_u)
      for (int i = tid; i < cu; i += IGMC_BLOCK) {
        const int id = tl[i];
        const int li = (i == 0) ? 0 : 1 + bm_rank(sel_u, pre_u, id);
        sg[li] = id;
        sl[li] = (uint8_t)(2 * td[i]);             // reference :245
      }
    if (do_v)
      for (int i = tid; i < cv; i += IGMC_BLOCK) {
        const int id = tl[cap_u + i];
        const int li = (i == 0) ? 0 : 1 + bm_rank(sel_v, pre_v, id);
        sg[cap_u + li] = id;
        sl[cap_u + li] = (uint8_t)(2 * td[cap_u + i] + 1);
      }
  }
  if (tid == 0) {
    if (do_u) {
      a.b.n_users[g] = cu;
      a.b.edge_cnt[g] = 0;
      if (g == 0 && a.b.relm) a.b.totals[3] = a.B;      // consumers of a lean batch (no CSR emission) read the batch size here
    }
    if (do_v) a.b.n_items[g] = cv;
  }
  if (a.b.relmT && do_v) {     // ... and of the transposed copy the cv x cap_u part
    uint32_t* rt = (uint32_t*)(a.b.relmT + (size_t)g * a.b.cap_v * a.b.relmT_ld);
    const int nw = (cv * a.b.relmT_ld) >> 2;
    for (int i = tid; i < nw; i += IGMC_BLOCK) rt[i] = 0u;
  }
  if (a.b.relm && do_u) {      // clear this link's dense (user, item) -> relation block (only the cu x cap_v part is used)
    uint32_t* rm = (uint32_t*)(a.b.relm + (size_t)g * a.b.cap_u * a.b.relm_ld);
    const int nw = (cu * a.b.relm_ld) >> 2;
    for (int i = tid; i < nw; i += IGMC_BLOCK) rm[i] = 0u;
  }
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_extract_nodes(ExtractArgs a) { igmc_kernarg_warm<sizeof(ExtractArgs) + 32>(); extract_nodes_body(a); }
// ... for a whole GROUP of batches in one launch (igmc_extract_group): blockIdx.z = batch i of the group, its arena =
// set[i], its selector = a.first + 2 i (selector q | (i << 1), igmc_hip.h).  Extraction is a dependent chain per workgroup,
// so its throughput is the number of workgroups in flight: M batches in one launch take about as long as two or three.
__global__ __launch_bounds__(IGMC_BLOCK) void k_extract_nodes_set(ExtractArgs a, const BatchDev* __restrict__ set) {
  igmc_kernarg_warm<sizeof(ExtractArgs) + 32>();
  a.b = set[blockIdx.z];
  a.first += 2 * (int)blockIdx.z;
  extract_nodes_body(a);
}


// ---------------------------------------------------------------- shared by k_count / k_fill
// Rebuild the membership bitmaps of graph g from its node slot (ids by local index).
__device__ void rebuild_sel(const int32_t* sg, int cap_u, int cu, int cv, uint32_t* sel_u, int Wu,
                            uint32_t* sel_v, int Wv) {
  bm_clear(sel_u, Wu);
  bm_clear(sel_v, Wv);
  __syncthreads();
  for (int i = 1 + threadIdx.x; i < cu; i += IGMC_BLOCK) atomicOr(&sel_u[sg[i] >> 5], 1u << (sg[i] & 31));
  for (int i = 1 + threadIdx.x; i < cv; i += IGMC_BLOCK) atomicOr(&sel_v[sg[cap_u + i] >> 5], 1u << (sg[cap_u + i] & 31));
  __syncthreads();
}

// ---------------------------------------------------------------- kernel 2: induced degrees
// Arow[u_nodes][:, v_nodes] with the target entry removed (reference :236-238).  Row r of graph g
// (users first, then items) is owned by exactly one wave: r = (blockIdx.y*4 + wave) mod (4*S).
__global__ __launch_bounds__(IGMC_BLOCK) void k_count(GraphDev G, BatchDev b) {
  igmc_kernarg_warm<sizeof(GraphDev) + sizeof(BatchDev) + 32>();
  IGMC_DYN_SMEM(smem);
  const int g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wu = (G.n_users + 31) >> 5, Wv = (G.n_items + 31) >> 5;
  uint32_t* sel_u = (uint32_t*)smem;
  uint32_t* sel_v = sel_u + Wu;
  const int cap_u = b.cap_u;
  const size_t so = (size_t)g * b.slot;
  const int32_t* sg = b.s_gid + so;
  int32_t* sd = b.s_deg + so;
  const int cu = b.n_users[g], cv = b.n_items[g];
  const int u0 = sg[0], v0 = sg[cap_u];
  rebuild_sel(sg, cap_u, cu, cv, sel_u, Wu, sel_v, Wv);
  const int stride = gridDim.y * (IGMC_BLOCK / 64);
  int etot = 0;
  for (int r = blockIdx.y * (IGMC_BLOCK / 64) + wave; r < cu + cv; r += stride) {
    const bool is_u = r < cu;
    const int li = is_u ? r : r - cu;
    const int id = sg[is_u ? li : cap_u + li];
    const int32_t* ptr = is_u ? G.u_ptr : G.v_ptr;
    const int32_t* idx = is_u ? G.u_idx : G.v_idx;
    const uint32_t* sel = is_u ? sel_v : sel_u;
    const int t0 = is_u ? v0 : u0;
    const int beg = ptr[id], end = ptr[id + 1];
    int c = 0;
    for (int p0 = beg; p0 < end; p0 += 256) {
      int j[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = p0 + q * 64 + lane;
        j[q] = (p < end) ? idx[p] : -1;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j[q] >= 0) c += (j[q] == t0) ? (li != 0) : (int)bm_test(sel, j[q]);
    }
    c = igmc_wave_sum_i(c);
    if (lane == 0) {
      sd[is_u ? li : cap_u + li] = c;
      etot += c;
    }
  }
  if (lane == 0 && etot) atomicAdd(&b.edge_cnt[g], etot);   // integer: order-independent
}

// ---------------------------------------------------------------- kernel 4: CSR fill
// Entry layout: ecr = source node (24 bits) | relation << 24 ;  ecode = relation*L + label(source).
__global__ __launch_bounds__(IGMC_BLOCK) void k_fill(GraphDev G, BatchDev b) {
  igmc_kernarg_warm<sizeof(GraphDev) + sizeof(BatchDev) + 32>();
  IGMC_DYN_SMEM(smem);
  __shared__ int sm[16];
  const int g = blockIdx.x, B = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // batch-wide node / edge offsets: every workgroup derives them from the per-graph counts (B is small), the
  // y == 0 workgroup of each graph publishes them -- no separate scan launch.
  int nb = 0, eb = 0, totn = 0, tote = 0;
  for (int base = 0; base < B; base += IGMC_BLOCK) {
    const int i = base + tid;
    const int n = (i < B) ? b.n_users[i] + b.n_items[i] : 0;
    const int e = (i < B) ? b.edge_cnt[i] : 0;
    nb += igmc_block_sum_i(i < g ? n : 0, sm);
    eb += igmc_block_sum_i(i < g ? e : 0, sm);
    totn += igmc_block_sum_i(n, sm);
    tote += igmc_block_sum_i(e, sm);
  }
  const int ovf = (totn > b.node_cap) || (tote > b.edge_cap);
  if (blockIdx.y == 0 && tid == 0) {
    b.node_off[g] = nb;
    b.edge_off[g] = eb;
    if (g == 0) {
      b.node_off[B] = totn;
      b.edge_off[B] = tote;
      b.totals[0] = ovf ? 0 : totn;
      b.totals[1] = ovf ? 0 : tote;
      b.totals[2] = ovf;
      b.totals[3] = B;
      b.totals[4] = totn;
      b.totals[5] = tote;
      if (!ovf) b.row_ptr[totn] = tote;
    }
  }
  if (ovf) return;
  const int Wu = (G.n_users + 31) >> 5, Wv = (G.n_items + 31) >> 5;
  uint32_t* sel_u = (uint32_t*)smem;
  uint32_t* pre_u = sel_u + Wu;
  uint32_t* sel_v = pre_u + Wu;
  uint32_t* pre_v = sel_v + Wv;
  const int cap_u = b.cap_u;
  const size_t so = (size_t)g * b.slot;
  const int32_t* sg = b.s_gid + so;
  const uint8_t* sl = b.s_lab + so;
  const int32_t* sd = b.s_deg + so;
  const int cu = b.n_users[g], cv = b.n_items[g];
  const int u0 = sg[0], v0 = sg[cap_u];
  const int L = b.num_labels;

  rebuild_sel(sg, cap_u, cu, cv, sel_u, Wu, sel_v, Wv);
  bm_prefix(sel_u, pre_u, Wu, sm);
  bm_prefix(sel_v, pre_v, Wv, sm);

  // CSR row pointers (every block of the graph recomputes the scan; block y == 0 publishes it together
  // with the node arrays).  Row starts of THIS block's rows are kept in registers via the same scan.
  const int nn = cu + cv;
  const int stride = gridDim.y * (IGMC_BLOCK / 64);
  const int my_first = blockIdx.y * (IGMC_BLOCK / 64) + wave;
  int running = 0;
  for (int base = 0; base < nn; base += IGMC_BLOCK) {
    const int n = base + tid;
    const int s = (n < cu) ? n : cap_u + (n - cu);
    const int deg = (n < nn) ? sd[s] : 0;
    int tot;
    const int ex = igmc_block_scan_excl(deg, &tot, sm);
    if (n < nn) {
      // rows handled by this block need their start: publish through global row_ptr (all blocks write the
      // same value -> benign), node arrays only once
      b.row_ptr[nb + n] = eb + running + ex;
      if (blockIdx.y == 0) {
        b.node_label[nb + n] = sl[s];
        b.node_gid[nb + n] = sg[s];
        b.node_graph[nb + n] = g;
      }
    }
    running += tot;
  }
  __syncthreads();

  for (int r = my_first; r < nn; r += stride) {
    const bool is_u = r < cu;            // user row: CSR row of the user  intersected with selected items
    const int li = is_u ? r : r - cu;    // item row: CSC column of the item intersected with selected users
    const int id = sg[is_u ? li : cap_u + li];
    const int32_t* ptr = is_u ? G.u_ptr : G.v_ptr;
    const int32_t* idx = is_u ? G.u_idx : G.v_idx;
    const uint8_t* rel = is_u ? G.u_rel : G.v_rel;
    const uint32_t* sel = is_u ? sel_v : sel_u;
    const uint32_t* pre = is_u ? pre_v : pre_u;
    const int t0 = is_u ? v0 : u0;
    const int nbase = is_u ? nb + cu : nb;         // node index base of the neighbour side
    const int sbase = is_u ? cap_u : 0;            // slot base of the neighbour side (labels)
    const int beg = ptr[id], end = ptr[id + 1];
    int o = b.row_ptr[nb + r];
    for (int p0 = beg; p0 < end; p0 += 256) {
      int j[4], rl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = p0 + q * 64 + lane;
        const bool valid = p < end;
        j[q] = valid ? idx[p] : -1;
        rl[q] = valid ? (int)rel[p] : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool m = (j[q] >= 0) && ((j[q] == t0) ? (li != 0) : bm_test(sel, j[q]));
        const unsigned long long bal = __ballot(m);
        if (m) {
          const int pos = o + __popcll(bal & ((1ull << lane) - 1ull));
          const int ln = (j[q] == t0) ? 0 : 1 + bm_rank(sel, pre, j[q]);
          b.ecr[pos] = (uint32_t)(nbase + ln) | ((uint32_t)rl[q] << 24);
          b.ecode[pos] = (uint16_t)(rl[q] * L + (int)sl[sbase + ln]);
          b.edst[pos] = (uint16_t)r;
          b.eflag[pos] = 3;
        }
        o += __popcll(bal);
      }
    }
  }
}

// ================================================================ dense-block path (capped extraction)
// With a per-hop cap the induced block Arow[u_nodes][:, v_nodes] is at most ~(1+h*cap)^2 entries (101 x 101 for
// ml_1m): it is materialised ONCE as a dense byte matrix relm[u_local][v_local] = relation+1 from the selected
// USERS' CSR rows only.  Degrees, item rows and the relation-sorted order all come from that matrix, so the
// long CSC columns of popular items (thousands of entries, ~4/5 of the traversal work of the generic kernels)
// are never scanned.  Everything stays atomic-free on data => bit-reproducible.

// kernel 2d: fill relm from the selected users' rows; counts the (undirected) edges of the link.
// Work is balanced by ENTRIES, not rows: the CSR rows of the link's selected users are viewed as one concatenated
// stream (row starts = an exclusive scan of the degrees, in LDS), block y of the link's S blocks takes the y-th of S
// equal slices of it and every thread takes entries at stride 256 inside the slice, locating its row by a binary
// search over the <= 256 row starts.  (One wave per ROW, as before, made the kernel as long as the longest row: an
// active user's 2 400 ratings = 10 dependent 256-entry rounds = 35-50 us for ~8 MB of traffic.)
#ifndef RELM_Q
#define RELM_Q 4          // entries a thread has in flight per round
#endif
__device__ __forceinline__ void relm_body(const GraphDev& G, const BatchDev& b) {
  IGMC_DYN_SMEM(smem);
  __shared__ int sm[16];
  const int g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int Wv = (G.n_items + 31) >> 5;
  uint32_t* sel_v = (uint32_t*)smem;
  uint32_t* pre_v = sel_v + Wv;
  int* rstart = (int*)(pre_v + Wv);          // [cap_u + 1] first stream position of every selected user's row
  int* rbase = rstart + b.cap_u + 1;         // [cap_u]     CSR position of the row's first entry
  const int cap_u = b.cap_u;
  const size_t so = (size_t)g * b.slot;
  const int32_t* sg = b.s_gid + so;
  const int cu = b.n_users[g], cv = b.n_items[g];
  const int v0 = sg[cap_u];
  const int ld = b.relm_ld;
  uint8_t* rm = b.relm + (size_t)g * cap_u * ld;
  // row extents first (two dependent loads), the item bitmap under their latency
  int deg = 0, base = 0;
  if (tid < cu) {
    const int id = sg[tid];
    base = G.u_ptr[id];
    deg = G.u_ptr[id + 1] - base;
  }
  bm_clear(sel_v, Wv);
  __syncthreads();
  for (int i = 1 + tid; i < cv; i += IGMC_BLOCK) atomicOr(&sel_v[sg[cap_u + i] >> 5], 1u << (sg[cap_u + i] & 31));
  __syncthreads();
  bm_prefix(sel_v, pre_v, Wv, sm);
  int total;
  const int ex = igmc_block_scan_excl(deg, &total, sm);        // cu <= 256 = one pass (dense path: cap_u <= 256)
  if (tid < cu) {
    rstart[tid] = ex;
    rbase[tid] = base;
  }
  if (tid == 0) rstart[cu] = total;
  __syncthreads();
  const int S = gridDim.y;
  const int lo = (int)((long long)total * blockIdx.y / S), hi = (int)((long long)total * (blockIdx.y + 1) / S);
  int c = 0;
  // A thread's entries e = lo + tid, + 256, + 512, .. only grow: ONE binary search for its first entry, then the row index
  // walks forward (rows average a few hundred entries, the stride is 256: less than one step per entry -- the per-entry
  // binary search was eight dependent LDS reads in front of every global load); RELM_Q entries per round are in flight.
  int arow = 0;
  {
    const int e = lo + tid;
    if (e < hi) {
      int a = 0, z = cu;                       // largest row with rstart[row] <= e
      while (z - a > 1) {
        const int mid = (a + z) >> 1;
        if (rstart[mid] <= e) a = mid;
        else z = mid;
      }
      arow = a;
    }
  }
  for (int e0 = lo; e0 < hi; e0 += RELM_Q * IGMC_BLOCK) {
    int j[RELM_Q], rl[RELM_Q], row[RELM_Q];
#pragma unroll
    for (int q = 0; q < RELM_Q; ++q) {
      const int e = e0 + q * IGMC_BLOCK + tid;
      j[q] = -1;
      rl[q] = 0;
      row[q] = 0;
      if (e < hi) {
        while (rstart[arow + 1] <= e) ++arow;  // (rstart[cu] = total > e: ends; empty rows are stepped over)
        row[q] = arow;
        const int p = rbase[arow] + (e - rstart[arow]);
        j[q] = G.u_idx[p];
        rl[q] = (int)G.u_rel[p];
      }
    }
#pragma unroll
    for (int q = 0; q < RELM_Q; ++q) {
      if (j[q] < 0) continue;
      const bool mt = (j[q] == v0) ? (row[q] != 0) : bm_test(sel_v, j[q]);
      if (mt) {
        const int lv = (j[q] == v0) ? 0 : 1 + bm_rank(sel_v, pre_v, j[q]);
        rm[(size_t)row[q] * ld + lv] = (uint8_t)((rl[q] + 1) | IGMC_RELM_KEEP);      // relation + 1, both directions kept
        if (b.relmT) b.relmT[((size_t)g * b.cap_v + lv) * b.relmT_ld + row[q]] = (uint8_t)((rl[q] + 1) | IGMC_RELM_KEEP);
        ++c;
      }
    }
  }
  c = igmc_wave_sum_i(c);
  if (lane == 0 && c) atomicAdd(&b.edge_cnt[g], 2 * c);   // directed edges; integer => order-independent
}
__global__ __launch_bounds__(IGMC_BLOCK) void k_relm(GraphDev G, BatchDev b) { igmc_kernarg_warm<sizeof(GraphDev) + sizeof(BatchDev) + 32>(); relm_body(G, b); }
__global__ __launch_bounds__(IGMC_BLOCK) void k_relm_set(GraphDev G, const BatchDev* __restrict__ set) { igmc_kernarg_warm<sizeof(GraphDev) + 32>(); relm_body(G, set[blockIdx.z]); }


// kernel 3d: batch offsets, degrees, row pointers and the relation-sorted CSR, all from relm
__global__ __launch_bounds__(IGMC_BLOCK) void k_emit(BatchDev b) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  IGMC_DYN_SMEM(smem);
  __shared__ int sm[16];
  const int g = blockIdx.x, B = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int nb = 0, eb = 0, totn = 0, tote = 0;
  for (int base = 0; base < B; base += IGMC_BLOCK) {
    const int i = base + tid;
    const int n = (i < B) ? b.n_users[i] + b.n_items[i] : 0;
    const int e = (i < B) ? b.edge_cnt[i] : 0;
    nb += igmc_block_sum_i(i < g ? n : 0, sm);
    eb += igmc_block_sum_i(i < g ? e : 0, sm);
    totn += igmc_block_sum_i(n, sm);
    tote += igmc_block_sum_i(e, sm);
  }
  const int ovf = (totn > b.node_cap) || (tote > b.edge_cap);
  if (blockIdx.y == 0 && tid == 0) {
    b.node_off[g] = nb;
    b.edge_off[g] = eb;
    if (g == 0) {
      b.node_off[B] = totn;
      b.edge_off[B] = tote;
      b.totals[0] = ovf ? 0 : totn;
      b.totals[1] = ovf ? 0 : tote;
      b.totals[2] = ovf;
      b.totals[3] = B;
      b.totals[4] = totn;
      b.totals[5] = tote;
      if (!ovf) b.row_ptr[totn] = tote;
    }
  }
  if (ovf) return;
  const int cap_u = b.cap_u;
  const size_t so = (size_t)g * b.slot;
  const int32_t* sg = b.s_gid + so;
  const uint8_t* sl = b.s_lab + so;
  const int cu = b.n_users[g], cv = b.n_items[g];
  const int ld = b.relm_ld, ldw = ld >> 2;
  const int L = b.num_labels, nn = cu + cv;
  int* rstart = (int*)smem;                              // [slot] CSR row starts of this graph
  uint32_t* rmw = (uint32_t*)(rstart + b.slot);          // [cu * ld / 4] the dense block, staged in LDS
  const uint8_t* rm = (const uint8_t*)rmw;
  {
    const uint32_t* src = (const uint32_t*)(b.relm + (size_t)g * cap_u * ld);
    const int nw = cu * ldw;
    for (int i = tid; i < nw; i += IGMC_BLOCK) rmw[i] = src[i];
  }
  __syncthreads();
  // ---- degrees (one thread per row / column), row pointers, slots of the graph
  int running = 0;
  for (int base = 0; base < nn; base += IGMC_BLOCK) {
    const int n = base + tid;
    int deg = 0;
    if (n < cu) {
      for (int k = 0; k < ldw; ++k) {                    // bytes beyond cv are zero
        const uint32_t w = rmw[n * ldw + k];
        deg += ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w >> 24) != 0);
      }
    } else if (n < nn) {
      const int vl = n - cu;
      for (int u = 0; u < cu; ++u) deg += rm[u * ld + vl] != 0;
    }
    int tot;
    const int ex = igmc_block_scan_excl(deg, &tot, sm);
    if (n < nn) {
      const int rs = eb + running + ex;
      rstart[n] = rs;
      if (blockIdx.y == 0) {
        const int s = (n < cu) ? n : cap_u + (n - cu);
        b.row_ptr[nb + n] = rs;
        b.node_label[nb + n] = sl[s];
        b.node_gid[nb + n] = sg[s];
        b.node_graph[nb + n] = g;
      }
    }
    running += tot;
  }
  __syncthreads();
  // ---- emission: a row's bytes are read once (<= 4 per lane), then one ballot-compaction pass per relation
  const int stride = gridDim.y * (IGMC_BLOCK / 64);
  const int R = b.max_rel + 1;
  for (int r = blockIdx.y * (IGMC_BLOCK / 64) + wave; r < nn; r += stride) {
    const bool is_u = r < cu;
    const int len = is_u ? cv : cu;                 // neighbours live on the other side (len <= 256, see launcher)
    const int nbase = is_u ? nb + cu : nb;
    const int sbase = is_u ? cap_u : 0;
    int o = rstart[r];
    int val[4], lab[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = q * 64 + lane;
      val[q] = 0;
      lab[q] = 0;
      if (k < len) {
        val[q] = is_u ? rm[r * ld + k] : rm[k * ld + (r - cu)];
        if (val[q]) lab[q] = sl[sbase + k];
      }
    }
    for (int rel = 0; rel < R; ++rel) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q * 64 >= len) break;
        const bool mt = (val[q] & IGMC_RELM_CODE) == rel + 1;
        const unsigned long long bal = __ballot(mt);
        if (mt) {
          const int pos = o + __popcll(bal & ((1ull << lane) - 1ull));
          b.ecr[pos] = (uint32_t)(nbase + q * 64 + lane) | ((uint32_t)rel << 24);
          b.ecode[pos] = (uint16_t)(rel * L + lab[q]);
          b.edst[pos] = (uint16_t)r;
          // keep flags of the entry (bit 0: column -> row, bit 1: row -> column) from the block's two keep bits, which are
          // those of the USER row; an item row sees the two directions swapped.  All kept unless a dense edge dropout
          // (k_relm_dropout, lean arenas) ran before this emission.
          const int fl = (val[q] >> IGMC_RELM_KF) & 3;
          b.eflag[pos] = (uint8_t)(is_u ? fl : ((fl >> 1) | ((fl & 1) << 1)));
        }
        o += __popcll(bal);
      }
    }
  }
}

// Node part of the emission alone (batch offsets, per-node label / id / graph, totals): what the model kernels that work
// on the dense blocks need of the collated batch (dense per-layer path: no edge list is read anywhere in the step).
__device__ __forceinline__ void emit_nodes_body(const BatchDev& b) {
  __shared__ int sm[16];
  const int g = blockIdx.x, B = gridDim.x;
  const int tid = threadIdx.x;
  int nb = 0, eb = 0, totn = 0, tote = 0;
  for (int base = 0; base < B; base += IGMC_BLOCK) {
    const int i = base + tid;
    const int n = (i < B) ? b.n_users[i] + b.n_items[i] : 0;
    const int e = (i < B) ? b.edge_cnt[i] : 0;
    nb += igmc_block_sum_i(i < g ? n : 0, sm);
    eb += igmc_block_sum_i(i < g ? e : 0, sm);
    totn += igmc_block_sum_i(n, sm);
    tote += igmc_block_sum_i(e, sm);
  }
  const int ovf = (totn > b.node_cap) || (tote > b.edge_cap);
  if (tid == 0) {
    b.node_off[g] = nb;
    b.edge_off[g] = eb;
    if (g == 0) {
      b.node_off[B] = totn;
      b.edge_off[B] = tote;
      b.totals[0] = ovf ? 0 : totn;
      b.totals[1] = ovf ? 0 : tote;
      b.totals[2] = ovf;
      b.totals[3] = B;
      b.totals[4] = totn;
      b.totals[5] = tote;
    }
  }
  if (ovf) return;
  const size_t so = (size_t)g * b.slot;
  const int cu = b.n_users[g], nn = cu + b.n_items[g];
  for (int n = tid; n < nn; n += IGMC_BLOCK) {
    const int s = (n < cu) ? n : b.cap_u + (n - cu);
    b.node_label[nb + n] = b.s_lab[so + s];
    b.node_gid[nb + n] = b.s_gid[so + s];
    b.node_graph[nb + n] = g;
  }
}
__global__ __launch_bounds__(IGMC_BLOCK) void k_emit_nodes(BatchDev b) { igmc_kernarg_warm<sizeof(BatchDev) + 32>(); emit_nodes_body(b); }
__global__ __launch_bounds__(IGMC_BLOCK) void k_emit_nodes_set(const BatchDev* __restrict__ set) { emit_nodes_body(set[blockIdx.z]); }


void igmc_launch_emit_nodes(const BatchDev& b, int B, void* stream) {
  IGMC_PLAUNCH("k_emit_nodes", k_emit_nodes, B, IGMC_BLOCK, 0, stream, b);
}

// ---------------------------------------------------------------- edge dropout flags
// reference models.py:193-198 -> PyG dropout_adj: independent Bernoulli(1-p) per DIRECTED edge
// (shared by the two directions when force_undirected).  16 lanes per CSR row.
__global__ __launch_bounds__(IGMC_BLOCK) void k_edge_flags(BatchDev b, float p, int force_undirected,
                                                            uint64_t seed, uint64_t step_arg, const int64_t* ctrl) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  // control block: key by (epoch, batch index) of the selected batch -- race-free under prefetching
  const uint64_t step = ctrl ? igmc_ctrl_drop_key(ctrl, igmc_ctrl_first(ctrl, (int)step_arg)) : step_arg;
  if (blockIdx.x == 0 && threadIdx.x == 0) b.stamp[1] = ctrl ? (int64_t)step : -1;
  const int N = b.totals[0];
  const int grp = (blockIdx.x * IGMC_BLOCK + threadIdx.x) >> 4, t = threadIdx.x & 15;
  const int ngrp = (gridDim.x * IGMC_BLOCK) >> 4;
  for (int i = grp; i < N; i += ngrp) {
    const int beg = b.row_ptr[i], end = b.row_ptr[i + 1];
    const bool row_user = (b.node_label[i] & 1) == 0;
    const uint32_t gi = (uint32_t)b.node_gid[i], gr = (uint32_t)b.node_graph[i];
    for (int e = beg + t; e < end; e += 16) {
      const uint32_t gc = (uint32_t)b.node_gid[b.ecr[e] & 0xFFFFFFu];
      const uint32_t u = row_user ? gi : gc, v = row_user ? gc : gi;
      const uint32_t dirF = force_undirected ? 2u : (row_user ? 1u : 0u);   // col -> row
      const uint32_t dirT = force_undirected ? 2u : (row_user ? 0u : 1u);   // row -> col
      const int kf = igmc_u01(igmc_edge_hash(seed, step, gr, u, v, dirF)) >= p;
      const int kt = igmc_u01(igmc_edge_hash(seed, step, gr, u, v, dirT)) >= p;
      b.eflag[e] = (uint8_t)(kf | (kt << 1));
      if (b.relm && row_user) {      // the dense block carries the same two bits (graphstep2.hip builds its masks from it)
        const int nb = b.node_off[gr];
        uint8_t* q = b.relm + ((size_t)gr * b.cap_u + (size_t)(i - nb)) * b.relm_ld + ((int)(b.ecr[e] & 0xFFFFFFu) - nb - b.n_users[gr]);
        *q = (uint8_t)((*q & IGMC_RELM_CODE) | (kf << IGMC_RELM_KF) | (kt << IGMC_RELM_KT));
        if (b.relmT) b.relmT[((size_t)gr * b.cap_v + ((int)(b.ecr[e] & 0xFFFFFFu) - nb - b.n_users[gr])) * b.relmT_ld + (i - nb)] = *q;
      }
    }
  }
}

// keep bits of the dense block from the per-entry flags (after flags were injected / cleared through the C ABI)
__global__ __launch_bounds__(IGMC_BLOCK) void k_relm_flags(BatchDev b) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  const int N = b.totals[0];
  const int grp = (blockIdx.x * IGMC_BLOCK + threadIdx.x) >> 4, t = threadIdx.x & 15;
  const int ngrp = (gridDim.x * IGMC_BLOCK) >> 4;
  for (int i = grp; i < N; i += ngrp) {
    if (b.node_label[i] & 1) continue;              // user rows only: they hold both directions of every edge
    const int gr = b.node_graph[i], nb = b.node_off[gr];
    uint8_t* row = b.relm + ((size_t)gr * b.cap_u + (size_t)(i - nb)) * b.relm_ld - nb - b.n_users[gr];
    for (int e = b.row_ptr[i] + t; e < b.row_ptr[i + 1]; e += 16) {
      uint8_t* q = row + (int)(b.ecr[e] & 0xFFFFFFu);
      *q = (uint8_t)((*q & IGMC_RELM_CODE) | ((b.eflag[e] & 3) << IGMC_RELM_KF));
      if (b.relmT) b.relmT[((size_t)gr * b.cap_v + ((int)(b.ecr[e] & 0xFFFFFFu) - nb - b.n_users[gr])) * b.relmT_ld + (i - nb)] = *q;
    }
  }
}

// edge dropout of a lean arena: the Bernoulli draws of k_edge_flags (same key: graph, user id, item id, direction), taken
// straight from the dense blocks -- no CSR needed.  grid (B, 4): a workgroup takes every 4th dword column group.
__device__ __forceinline__ void relm_dropout_body(const BatchDev& b, float p, int force_undirected, uint64_t seed,
                                                  uint64_t step_arg, const int64_t* ctrl) {
  const uint64_t step = ctrl ? igmc_ctrl_drop_key(ctrl, igmc_ctrl_first(ctrl, (int)step_arg)) : step_arg;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) b.stamp[1] = ctrl ? (int64_t)step : -1;
  const int g = blockIdx.x;
  const int cu = b.n_users[g], cv = b.n_items[g];
  const int ld = b.relm_ld, ldw = ld >> 2, cw = (cv + 3) >> 2;
  const int32_t* sg = b.s_gid + (size_t)g * b.slot;
  uint32_t* rm = (uint32_t*)(b.relm + (size_t)g * b.cap_u * ld);
  const int n = cu * cw;
  for (int i = blockIdx.y * IGMC_BLOCK + threadIdx.x; i < n; i += gridDim.y * IGMC_BLOCK) {
    const int row = i / cw, k = i - row * cw;
    uint32_t w = rm[row * ldw + k];
    if (!w) continue;
    const uint32_t u = (uint32_t)sg[row];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t by = (w >> (8 * q)) & 0xFFu;
      if (!(by & IGMC_RELM_CODE)) continue;
      const uint32_t v = (uint32_t)sg[b.cap_u + 4 * k + q];
      // user row: column -> row is item -> user (direction 1), row -> column is user -> item (direction 0)
      const uint32_t kf = igmc_u01(igmc_edge_hash(seed, step, (uint32_t)g, u, v, force_undirected ? 2u : 1u)) >= p;
      const uint32_t kt = igmc_u01(igmc_edge_hash(seed, step, (uint32_t)g, u, v, force_undirected ? 2u : 0u)) >= p;
      w = (w & ~(IGMC_RELM_KEEP << (8 * q))) | (((kf << IGMC_RELM_KF) | (kt << IGMC_RELM_KT)) << (8 * q));
      if (b.relmT) b.relmT[((size_t)g * b.cap_v + 4 * k + q) * b.relmT_ld + row] = (uint8_t)((w >> (8 * q)) & 0xFFu);
    }
    rm[row * ldw + k] = w;
  }
}
__global__ __launch_bounds__(IGMC_BLOCK) void k_relm_dropout(BatchDev b, float p, int force_undirected, uint64_t seed,
                                                              uint64_t step_arg, const int64_t* ctrl) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  relm_dropout_body(b, p, force_undirected, seed, step_arg, ctrl);
}
__global__ __launch_bounds__(IGMC_BLOCK) void k_relm_dropout_set(const BatchDev* __restrict__ set, float p, int force_undirected,
                                                                  uint64_t seed, uint64_t sel0, const int64_t* ctrl) {
  relm_dropout_body(set[blockIdx.z], p, force_undirected, seed, sel0 + 2 * blockIdx.z, ctrl);
}


void igmc_launch_relm_dropout(const BatchDev& b, int B, float p, int force_undirected, uint64_t seed, uint64_t step,
                              const int64_t* ctrl, void* stream) {
  IGMC_PLAUNCH("k_relm_dropout", k_relm_dropout, dim3(B, 4), IGMC_BLOCK, 0, stream, b, p, force_undirected, seed, step, ctrl);
}

void igmc_launch_relm_flags(const BatchDev& b, void* stream) {
  if (b.relm) IGMC_PLAUNCH("k_relm_flags", k_relm_flags, 256, IGMC_BLOCK, 0, stream, b);
}

// ---------------------------------------------------------------- static (cached) subgraphs
// reference MyDataset (util_functions.py:69-110): the enclosing subgraphs of a dataset are extracted ONCE and kept.  The
// native cache keeps, per link, the node sets with their hop distances (packed arrays resident in HBM, also saved under
// <root>/processed/); a batch loads the lists of its links into the arena's per-graph slots and the replay stages
// (induced edges, labels, collation / dense blocks) run as for any other batch -- asynchronous, capturable.
__global__ __launch_bounds__(IGMC_BLOCK) void k_load_nodes(BatchDev b, const int64_t* uoff, const int32_t* unodes,
                                                            const uint8_t* udist, const int64_t* voff, const int32_t* vnodes,
                                                            const uint8_t* vdist, const float* link_y, const int32_t* link_idx,
                                                            int first_arg, const int64_t* ctrl) {
  igmc_kernarg_warm<sizeof(BatchDev) + 32>();
  const int g = blockIdx.x, tid = threadIdx.x;
  const int first = ctrl ? igmc_ctrl_first(ctrl, first_arg) : first_arg;
  const int pos = link_idx ? link_idx[first + g] : first + g;
  if (g == 0 && tid == 0) {
    b.stamp[0] = ctrl ? (int64_t)first : -1;
    b.stamp[1] = -1;
  }
  const int64_t u0 = uoff[pos], v0 = voff[pos];
  int cu = (int)(uoff[pos + 1] - u0), cv = (int)(voff[pos + 1] - v0);
  cu = cu < b.cap_u ? cu : b.cap_u;        // (a cache built for this arena geometry never exceeds it)
  cv = cv < b.cap_v ? cv : b.cap_v;
  int32_t* tl = b.t_list + (size_t)g * b.slot;
  uint8_t* td = b.t_dist + (size_t)g * b.slot;
  for (int i = tid; i < cu; i += IGMC_BLOCK) {
    tl[i] = unodes[u0 + i];
    td[i] = udist[u0 + i];
  }
  for (int i = tid; i < cv; i += IGMC_BLOCK) {
    tl[b.cap_u + i] = vnodes[v0 + i];
    td[b.cap_u + i] = vdist[v0 + i];
  }
  if (tid == 0) {
    b.n_users[g] = cu;
    b.n_items[g] = cv;
    b.y[g] = link_y[pos];
  }
}

void igmc_launch_load_nodes(const BatchDev& b, const int64_t* uoff, const int32_t* unodes, const uint8_t* udist,
                            const int64_t* voff, const int32_t* vnodes, const uint8_t* vdist, const float* link_y,
                            const int32_t* link_idx, int first, int B, const int64_t* ctrl, void* stream) {
  IGMC_PLAUNCH("k_load_nodes", k_load_nodes, B, IGMC_BLOCK, 0, stream, b, uoff, unodes, udist, voff, vnodes, vdist, link_y,
               link_idx, first, ctrl);
}

// ---------------------------------------------------------------- side features of the target nodes
// reference util_functions.py:250-253, :272-275: a subgraph carries the feature rows of its two TARGET nodes only.
// The dataset keeps one [n_links, S] matrix (row k = [u_features[link_u[k]] | v_features[link_v[k]]]) in HBM; the rows
// of the batch's links are gathered here, in the extraction branch, through the same (control block, permutation)
// indexing as k_extract_nodes -- so the fused / captured training step needs no host-side index_select.
__global__ __launch_bounds__(IGMC_BLOCK) void k_side_gather(const float* src, int S, const int32_t* link_idx, int first_arg,
                                                             int B, const int64_t* ctrl, float* dst) {
  const int first = ctrl ? igmc_ctrl_first(ctrl, first_arg) : first_arg;
  for (int i = blockIdx.x * IGMC_BLOCK + threadIdx.x; i < B * S; i += gridDim.x * IGMC_BLOCK) {
    const int g = i / S, f = i - g * S;
    const int pos = link_idx ? link_idx[first + g] : first + g;
    dst[i] = src[(size_t)pos * S + f];
  }
}

void igmc_launch_side_gather(const float* src, int S, const int32_t* link_idx, int first, int B, const int64_t* ctrl,
                             float* dst, void* stream) {
  int grid = (B * S + IGMC_BLOCK - 1) / IGMC_BLOCK;
  grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
  IGMC_PLAUNCH("k_side_gather", k_side_gather, grid, IGMC_BLOCK, 0, stream, src, S, link_idx, first, B, ctrl, dst);
}

__global__ __launch_bounds__(IGMC_BLOCK) void k_fill_u8(uint8_t* p, int64_t n, uint8_t v) {
  for (int64_t i = (int64_t)blockIdx.x * IGMC_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IGMC_BLOCK) p[i] = v;
}

// ---------------------------------------------------------------- host launchers
size_t igmc_extract_smem_bytes(const GraphDev& g) {
  const size_t Wu = (g.n_users + 31) >> 5, Wv = (g.n_items + 31) >> 5;
  return 4 * (Wu + Wv) * sizeof(uint32_t);
}

// CSR emission of a capped batch from its dense blocks (idempotent: everything it reads is final after k_relm)
void igmc_launch_emit(const BatchDev& b, int B, void* stream) {
  int S = 2048 / (B > 0 ? B : 1);
  S = S < 1 ? 1 : (S > 16 ? 16 : S);
  IGMC_PLAUNCH("k_emit", k_emit, dim3(B, S), IGMC_BLOCK, (size_t)b.slot * sizeof(int) + (size_t)b.cap_u * b.relm_ld, stream, b);
}

// lean != 0 (capped arenas only): stop after the dense blocks -- node sets, labels, relm, y, the batch size -- which is
// all the matrix-core subgraph kernel reads; the collated CSR is emitted on demand (igmc_launch_emit)
void igmc_launch_extract(const GraphDev& g, const BatchDev& b, const int32_t* link_u, const int32_t* link_v,
                         const float* link_y, const int32_t* link_idx, int first, int B, int replay,
                         double sample_ratio, uint64_t seed, uint64_t epoch, const int64_t* ctrl, int lean, void* stream) {
  ExtractArgs a;
  a.ctrl = ctrl;
  a.g = g; a.b = b;
  a.link_u = link_u; a.link_v = link_v; a.link_y = link_y; a.link_idx = link_idx;
  a.first = first; a.B = B; a.replay = replay;
  a.sample_ratio = sample_ratio; a.seed = seed; a.epoch = epoch;
  const size_t smem = igmc_extract_smem_bytes(g);
  // S workgroups per link for the row passes: fill the chip even at batch 50
  int S = 2048 / (B > 0 ? B : 1);
  S = S < 1 ? 1 : (S > 16 ? 16 : S);
  a.split = (b.hop == 1 && !replay) ? 1 : 0;
  IGMC_PLAUNCH("k_extract_nodes", k_extract_nodes, dim3(B, a.split ? 2 : 1), IGMC_BLOCK, smem, stream, a);
  if (b.relm) {      // capped extraction (igmc_batch_create decides)
    const size_t Wu = (g.n_users + 31) >> 5, Wv = (g.n_items + 31) >> 5;
    int Sr = 400 / (B > 0 ? B : 1);        // entry-balanced slices: ~400 workgroups in all (one residency round)
    Sr = Sr < 1 ? 1 : (Sr > 16 ? 16 : Sr);
    IGMC_PLAUNCH("k_relm", k_relm, dim3(B, Sr), IGMC_BLOCK, (2 * Wv + 2 * (size_t)b.cap_u + 2) * sizeof(uint32_t), stream, g, b);
    if (!lean) igmc_launch_emit(b, B, stream);
    else if (b.relmT) igmc_launch_emit_nodes(b, B, stream);      // dense per-layer path: node arrays only, in this branch
  } else {
    IGMC_PLAUNCH("k_count", k_count, dim3(B, S), IGMC_BLOCK, smem / 4, stream, g, b);
    IGMC_PLAUNCH("k_fill", k_fill, dim3(B, S), IGMC_BLOCK, smem / 2, stream, g, b);
  }
}

// Extraction (+ edge dropout on the dense blocks) of `count` batches of one group in ONE launch per stage: arenas set[0 ..
// count) (a device array of identical geometry `b0`: lean, dense blocks, control block attached), selectors sel0 + 2 i.
void igmc_launch_extract_set(const GraphDev& g, const BatchDev* d_set, const BatchDev& b0, int count, const int32_t* link_u,
                             const int32_t* link_v, const float* link_y, const int32_t* link_idx, int sel0, int B,
                             double sample_ratio, uint64_t seed, const int64_t* ctrl, float drop_p, int force_undirected,
                             uint64_t drop_seed, void* stream) {
  ExtractArgs a;
  a.ctrl = ctrl;
  a.g = g; a.b = b0;
  a.link_u = link_u; a.link_v = link_v; a.link_y = link_y; a.link_idx = link_idx;
  a.first = sel0; a.B = B; a.replay = 0;
  a.sample_ratio = sample_ratio; a.seed = seed; a.epoch = 0;
  const size_t smem = igmc_extract_smem_bytes(g);
  a.split = (b0.hop == 1) ? 1 : 0;
  IGMC_PLAUNCH("k_extract_nodes", k_extract_nodes_set, dim3(B, a.split ? 2 : 1, count), IGMC_BLOCK, smem, stream, a, d_set);
  const size_t Wv = (g.n_items + 31) >> 5;
  int Sr = 400 / (B > 0 ? B : 1);
  Sr = Sr < 1 ? 1 : (Sr > 16 ? 16 : Sr);
  IGMC_PLAUNCH("k_relm", k_relm_set, dim3(B, Sr, count), IGMC_BLOCK, (2 * Wv + 2 * (size_t)b0.cap_u + 2) * sizeof(uint32_t), stream, g, d_set);
  if (b0.relmT) IGMC_PLAUNCH("k_emit_nodes", k_emit_nodes_set, dim3(B, 1, count), IGMC_BLOCK, 0, stream, d_set);
  if (drop_p > 0.f)
    IGMC_PLAUNCH("k_relm_dropout", k_relm_dropout_set, dim3(B, 4, count), IGMC_BLOCK, 0, stream, d_set, drop_p, force_undirected,
                 drop_seed, (uint64_t)sel0, ctrl);
}

void igmc_launch_edge_flags(const BatchDev& b, float p, int force_undirected, uint64_t seed, uint64_t step,
                            const int64_t* ctrl, void* stream) {
  IGMC_PLAUNCH("k_edge_flags", k_edge_flags, 512, IGMC_BLOCK, 0, stream, b, p, force_undirected, seed, step, ctrl);
}

void igmc_launch_fill_u8(uint8_t* p, int64_t n, uint8_t v, void* stream) {
  IGMC_PLAUNCH("k_fill_u8", k_fill_u8, 256, IGMC_BLOCK, 0, stream, p, n, v);
}

// dynamic LDS above 64 KB needs an explicit opt-in on HIP
int igmc_extract_prepare(size_t smem) {
#ifndef IGMC_HIPEMU
  if (smem > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)k_extract_nodes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem / 2)) != hipSuccess) return 1;
  }
#endif
  (void)smem;
  return 0;
}

