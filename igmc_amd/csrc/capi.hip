// capi.hip -- C-ABI host layer of libigmc_hip.so (see include/igmc_hip.h for the contract).
#include "launch.h"
#include "sortpool.h"
#include "g2_image.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static thread_local std::string g_err;
int g_igmc_prof_on = 0;

#define IGMC_FAIL(msg)                                                      \
  do {                                                                      \
    g_err = std::string(__func__) + ": " + (msg);                           \
    return 1;                                                               \
  } while (0)
#define HIPCHECK(expr)                                                      \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) {                                                 \
      g_err = std::string(__func__) + ": " #expr " -> " + hipGetErrorString(e_); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

// Buffers of one handle are carved out of a few large slabs (256-byte aligned) instead of one hipMalloc each: the
// ~40 arrays a kernel chain touches then share a handful of large pages (fewer address-translation misses on the
// first touch of every hop) and creation does 1-2 driver calls instead of 40.
struct Allocs {
  std::vector<void*> ptrs;
  size_t bytes = 0;
  char* slab = nullptr;
  size_t slab_left = 0;
  static constexpr size_t SLAB = (size_t)32 << 20;
  template <typename T> int get(T** out, size_t n) {
    const size_t b = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    bytes += b;
    if (b > SLAB / 2) {                 // large arrays get their own allocation
      void* p = nullptr;
      if (hipMalloc(&p, b) != hipSuccess) return 1;
      ptrs.push_back(p);
      *out = (T*)p;
      return 0;
    }
    if (b > slab_left) {
      void* p = nullptr;
      if (hipMalloc(&p, SLAB) != hipSuccess) return 1;
      ptrs.push_back(p);
      slab = (char*)p;
      slab_left = SLAB;
    }
    *out = (T*)slab;
    slab += b;
    slab_left -= b;
    return 0;
  }
  void release() {
    for (void* p : ptrs) hipFree(p);
    ptrs.clear();
    slab = nullptr;
    slab_left = 0;
  }
};

struct igmc_graph {
  int device;
  GraphDev d;
  int64_t nnz;
  int max_rel;
  int max_deg_u, max_deg_v;     // longest user row / item column (bounds the hop-1 subgraph sides)
  Allocs mem;
};

struct igmc_batch {
  const igmc_graph* g;
  BatchDev d;
  int last_B;
  const float* side;
  int n_side;
  const float* side_src;    // dataset-wide [n_links, n_side] matrix (igmc_batch_bind_side_source); rows gathered per batch
  float* side_buf;          // [graph_cap, n_side] owned by the arena
  int side_buf_cols;
  int lean;                 // igmc_batch_set_lean: extraction stops after the dense blocks (capped arenas)
  const int64_t* ctrl;
  Allocs mem;
};

struct igmc_model {
  int device;
  ModelDev d;
  ModelAux ax;
  int* done_ctr;
  int last_B, last_training, last_flags;
  // weight images (g2_w) of the subgraph / dense-layer kernels: whose parameters they hold, and the caller's one-shot
  // assertion that those parameters have not changed since the previous call (igmc_model_weights_unchanged)
  int img_valid, img_hint;
  const float* img_params;
  Allocs mem;
};

extern int g_igmc_compose_count;      // graphstep2.hip: launches of k_g2_compose so far

// One call's view of the images: the compose launch is skipped when the caller asserted unchanged parameters AND the
// library knows the images in place are those of the previous call's (possibly updated) parameters.
struct ImgScope {
  igmc_model* m;
  const float* p;
  int cur, before;
  ImgScope(igmc_model* m_, const float* p_) : m(m_), p(p_) {
    cur = m->img_hint && m->img_valid && m->img_params == p;
    m->img_hint = 0;
    m->d.img_current = cur;
    before = g_igmc_compose_count;
  }
  void done_unchanged() {       // the call left the parameters as they were
    m->img_valid = cur || g_igmc_compose_count != before;
    m->img_params = p;
    m->d.img_current = 0;
  }
  void done_updated(int emitted) {      // the call updated the parameters (images: only if its last kernel wrote them)
    m->img_valid = emitted;
    m->img_params = p;
    m->d.img_current = 0;
  }
};

extern "C" int igmc_model_weights_unchanged(igmc_model* m, int on) {
  if (!m) IGMC_FAIL("null model");
  m->img_hint = on ? 1 : 0;
  return 0;
}

extern "C" const char* igmc_last_error(void) { return g_err.c_str(); }
extern "C" int igmc_version(void) { return 100; }

// ------------------------------------------------------------------ profiling
struct ProfRec {
  std::string name;
  hipEvent_t a, b;
};
static std::vector<ProfRec> g_prof;

void igmc_prof_begin(const char* name, void* stream) {
  ProfRec r;
  r.name = name;
  hipEventCreate(&r.a);
  hipEventCreate(&r.b);
  hipEventRecord(r.a, (hipStream_t)stream);
  g_prof.push_back(r);
}
void igmc_prof_end(void* stream) { hipEventRecord(g_prof.back().b, (hipStream_t)stream); }

// on = 1: HIP events around every kernel launch (eager launches only; igmc_profile_fetch).
// on = 2: no events (legal inside hipGraph capture); k_graph_step launches enqueued / captured from now on clock
//         themselves on the device (igmc_profile_gs_clock).
extern "C" int igmc_profile_enable(int on) {
  g_igmc_prof_on = on;
  return 0;
}

// Launch clock of k_graph_step since the last reset: launches, mean duration in microseconds (device wall clock,
// earliest workgroup start -> last workgroup end).  Synchronises the device.
extern "C" int igmc_profile_gs_clock(const igmc_model* m, int64_t* launches, double* mean_us, int reset) {
  if (!m) IGMC_FAIL("null model");
  unsigned long long ts[4];
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(ts, m->d.gs_ts, sizeof(ts), hipMemcpyDeviceToHost));
  int khz = 0;
#ifndef IGMC_HIPEMU
  HIPCHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m->device));
#endif
  if (launches) *launches = (int64_t)ts[2];
  if (mean_us) *mean_us = (ts[2] && khz > 0) ? (double)ts[1] / (double)ts[2] * 1e3 / (double)khz : 0.0;
  if (reset) {
    const unsigned long long ts0[4] = {~0ull, 0ull, 0ull, 0ull};
    HIPCHECK(hipMemcpy(m->d.gs_ts, ts0, sizeof(ts0), hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int igmc_profile_fetch(char names[][48], float* ms, int* calls, int cap) {
  std::vector<std::string> order;
  std::map<std::string, std::pair<double, int>> agg;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.b);
    float t = 0.f;
    hipEventElapsedTime(&t, r.a, r.b);
    if (!agg.count(r.name)) order.push_back(r.name);
    agg[r.name].first += t;
    agg[r.name].second += 1;
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  g_prof.clear();
  int n = 0;
  for (auto& nm : order) {
    if (n >= cap) break;
    snprintf(names[n], 48, "%s", nm.c_str());
    ms[n] = (float)agg[nm].first;
    if (calls) calls[n] = agg[nm].second;
    ++n;
  }
  return n;
}

// ------------------------------------------------------------------ rating graph
extern "C" int igmc_graph_create(int n_users, int n_items, int64_t nnz, const int32_t* h_indptr,
                                 const int32_t* h_indices, const uint8_t* h_rating, int device,
                                 igmc_graph** out) {
  if (n_users <= 0 || n_items <= 0 || nnz < 0 || !h_indptr || !out) IGMC_FAIL("bad arguments");
  if (nnz >= INT_MAX) IGMC_FAIL("nnz must fit int32");
  HIPCHECK(hipSetDevice(device));
  // drop explicit zeros, convert label+1 -> relation id, sort rows by (relation, item)
  std::vector<int32_t> uptr(n_users + 1, 0), uidx, vptr(n_items + 1, 0), vidx;
  std::vector<uint8_t> urel, vrel;
  uidx.reserve(nnz);
  urel.reserve(nnz);
  int max_rel = 0;
  std::vector<std::pair<int32_t, int32_t>> row;   // (rel, item)
  for (int u = 0; u < n_users; ++u) {
    row.clear();
    for (int32_t p = h_indptr[u]; p < h_indptr[u + 1]; ++p) {
      if (h_rating[p] == 0) continue;
      const int it = h_indices[p];
      if (it < 0 || it >= n_items) IGMC_FAIL("column index out of range");
      row.emplace_back((int32_t)h_rating[p] - 1, it);
    }
    std::sort(row.begin(), row.end());
    for (size_t k = 1; k < row.size(); ++k)
      if (row[k].second == row[k - 1].second && row[k].first == row[k - 1].first) IGMC_FAIL("duplicate entries in CSR row");
    for (auto& e : row) {
      urel.push_back((uint8_t)e.first);
      uidx.push_back(e.second);
      max_rel = std::max(max_rel, (int)e.first);
      vptr[e.second + 1]++;
    }
    uptr[u + 1] = (int32_t)uidx.size();
  }
  const int64_t nz = (int64_t)uidx.size();
  for (int v = 0; v < n_items; ++v) vptr[v + 1] += vptr[v];
  vidx.assign(nz, 0);
  vrel.assign(nz, 0);
  {
    // counting sort by (item, relation, user): pass over relations keeps columns relation-sorted
    std::vector<int32_t> cur(vptr.begin(), vptr.end() - 1);
    for (int r = 0; r <= max_rel; ++r)
      for (int u = 0; u < n_users; ++u)
        for (int32_t p = uptr[u]; p < uptr[u + 1]; ++p)
          if (urel[p] == r) {
            const int32_t q = cur[uidx[p]]++;
            vidx[q] = u;
            vrel[q] = (uint8_t)r;
          }
  }
  igmc_graph* g = new igmc_graph();
  g->device = device;
  g->nnz = nz;
  g->max_rel = max_rel;
  g->max_deg_u = g->max_deg_v = 0;
  for (int u = 0; u < n_users; ++u) g->max_deg_u = std::max(g->max_deg_u, uptr[u + 1] - uptr[u]);
  for (int v = 0; v < n_items; ++v) g->max_deg_v = std::max(g->max_deg_v, vptr[v + 1] - vptr[v]);
  int32_t *d_uptr, *d_uidx, *d_vptr, *d_vidx;
  uint8_t *d_urel, *d_vrel;
  if (g->mem.get(&d_uptr, n_users + 1) || g->mem.get(&d_uidx, nz) || g->mem.get(&d_urel, nz) ||
      g->mem.get(&d_vptr, n_items + 1) || g->mem.get(&d_vidx, nz) || g->mem.get(&d_vrel, nz)) {
    g->mem.release();
    delete g;
    IGMC_FAIL("hipMalloc failed");
  }
  HIPCHECK(hipMemcpy(d_uptr, uptr.data(), (n_users + 1) * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_vptr, vptr.data(), (n_items + 1) * 4, hipMemcpyHostToDevice));
  if (nz) {
    HIPCHECK(hipMemcpy(d_uidx, uidx.data(), nz * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_urel, urel.data(), nz, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_vidx, vidx.data(), nz * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_vrel, vrel.data(), nz, hipMemcpyHostToDevice));
  }
  g->d.n_users = n_users;
  g->d.n_items = n_items;
  g->d.u_ptr = d_uptr; g->d.u_idx = d_uidx; g->d.u_rel = d_urel;
  g->d.v_ptr = d_vptr; g->d.v_idx = d_vidx; g->d.v_rel = d_vrel;
  *out = g;
  return 0;
}

extern "C" void igmc_graph_destroy(igmc_graph* g) {
  if (!g) return;
  g->mem.release();
  delete g;
}
extern "C" int64_t igmc_graph_hbm_bytes(const igmc_graph* g) { return g ? (int64_t)g->mem.bytes : 0; }

// ------------------------------------------------------------------ batch arena
extern "C" int igmc_batch_create(const igmc_graph* g, int max_graphs, int hop, int max_nodes_per_hop,
                                 igmc_batch** out) {
  if (!g || !out || max_graphs <= 0 || hop < 1) IGMC_FAIL("bad arguments");
  if (2 * hop + 2 > 250) IGMC_FAIL("hop too large");
  HIPCHECK(hipSetDevice(g->device));
  const size_t smem = igmc_extract_smem_bytes(g->d);
  if (smem > 150 * 1024) IGMC_FAIL("graph too large for the LDS bitmaps (users+items must be < ~300k)");
  if (igmc_extract_prepare(smem)) IGMC_FAIL("hipFuncSetAttribute failed");
  // slot capacity of one side: the target plus, per hop, at most max_nodes_per_hop new nodes; the hop-1 fringe of a
  // side is the neighbourhood of ONE node of the other side, so the longest row / column of the rating matrix bounds
  // it as well (an uncapped run on a sparse graph -- the Monti datasets -- then still gets slots small enough for the
  // dense induced blocks and the subgraph kernel)
  auto side_cap = [&](int n, int opposite_max_deg) -> int64_t {
    int64_t per = (max_nodes_per_hop < 0) ? (int64_t)n : std::min<int64_t>(max_nodes_per_hop, n);
    if (hop == 1) per = std::min<int64_t>(per, opposite_max_deg);
    return std::min<int64_t>(n, 1 + (int64_t)hop * per);
  };
  const int64_t cap_u = side_cap(g->d.n_users, g->max_deg_v), cap_v = side_cap(g->d.n_items, g->max_deg_u);
  const int64_t slot = cap_u + cap_v;
  const int64_t node_cap = (int64_t)max_graphs * slot;
  const int64_t per_graph_e = std::min<int64_t>(2 * cap_u * cap_v, 2 * g->nnz);
  int64_t edge_cap = (int64_t)max_graphs * per_graph_e;
  if (node_cap >= (1 << 24)) IGMC_FAIL("node capacity exceeds the 24-bit source index of a CSR entry; lower the batch size");
  edge_cap = std::min<int64_t>(edge_cap, (int64_t)INT_MAX - 65536);
  if (g->max_rel * (2 * hop + 2) + (2 * hop + 1) > 65535) IGMC_FAIL("relation*label code exceeds uint16");
  igmc_batch* b = new igmc_batch();
  b->g = g;
  b->last_B = 0;
  b->side = nullptr;
  b->n_side = 0;
  b->side_src = nullptr;
  b->side_buf = nullptr;
  b->side_buf_cols = 0;
  b->lean = 0;
  b->ctrl = nullptr;
  BatchDev& d = b->d;
  Allocs& M = b->mem;
  const int Bc = max_graphs;
  int fail = 0;
  fail |= M.get(&d.node_off, Bc + 1) | M.get(&d.n_users, Bc) | M.get(&d.n_items, Bc) | M.get(&d.edge_cnt, Bc) |
          M.get(&d.edge_off, Bc + 1);
  fail |= M.get(&d.node_label, node_cap) | M.get(&d.node_gid, node_cap) | M.get(&d.node_graph, node_cap) |
          M.get(&d.row_ptr, node_cap + 1);
  fail |= M.get(&d.ecr, edge_cap) | M.get(&d.ecode, edge_cap) | M.get(&d.eflag, edge_cap) |
          M.get(&d.edst, edge_cap);
  fail |= M.get(&d.y, Bc) | M.get(&d.totals, 8) | M.get(&d.stamp, 4);
  d.relm = nullptr;
  d.relmT = nullptr;
  d.relmT_ld = (int)((cap_u + 3) & ~(size_t)3);
  d.max_rel = g->max_rel;
  d.relm_ld = (int)((cap_v + 3) & ~(size_t)3);
  // dense path: rows of the block fit one 256-entry super-chunk and block + row starts fit the default LDS window
  // (a byte of the block holds relation + 1 in bits 0-3 and the two keep bits of edge dropout in bits 4-5: <= 15 relations)
  if (g->max_rel + 1 <= (int)IGMC_RELM_CODE && cap_u <= 256 && cap_v <= 256 && cap_u * (size_t)d.relm_ld + slot * 4 <= 60 * 1024) fail |= M.get(&d.relm, (size_t)Bc * cap_u * d.relm_ld + (size_t)128 * d.relm_ld + 128);   // dense induced block per link (+ a tail pad: the subgraph kernel's lanes read up to 128 rows from a slot's start unguarded)
  // slots too large for the subgraph kernel (> 128 nodes a side) but with a dense block: the transposed copy feeds the
  // item-side workgroups of the dense per-layer kernels (denselayer.hip)
  // (IGMC_DL_ALWAYS=1: also for small slots -- lets tests run those kernels on small cases)
  {
    const char* da = getenv("IGMC_DL_ALWAYS");
    // (more than five relations: the subgraph kernel does not take the arena whatever its slots, the dense-layer kernels do)
    // ... and so for a layer-0 table of more than 32 rows (two hops: the graph's relations x 6 labels)
    const int nlab = 2 * hop + 2;
    if (d.relm && (cap_u > 128 || cap_v > 128 || g->max_rel + 1 > G2_NR || (g->max_rel + 1) * nlab + nlab + 1 > 32 || (da && atoi(da) == 1))) fail |= M.get(&d.relmT, (size_t)Bc * cap_v * d.relmT_ld);
  }
  fail |= M.get(&d.s_gid, Bc * slot) | M.get(&d.s_lab, Bc * slot) | M.get(&d.s_deg, Bc * slot) |
          M.get(&d.t_list, Bc * slot) | M.get(&d.t_dist, Bc * slot);
  if (fail) {
    M.release();
    delete b;
    IGMC_FAIL("hipMalloc failed (batch arena)");
  }
  HIPCHECK(hipMemset(d.totals, 0, 8 * sizeof(int32_t)));
  HIPCHECK(hipMemset(d.stamp, 0xFF, 4 * sizeof(int64_t)));        // -1: no batch of a known cursor in the arena
  d.cap_u = (int)cap_u;
  d.cap_v = (int)cap_v;
  d.slot = (int)slot;
  d.node_cap = (int)node_cap;
  d.edge_cap = (int)edge_cap;
  d.graph_cap = Bc;
  d.hop = hop;
  d.max_nodes_per_hop = max_nodes_per_hop;
  d.num_labels = 2 * hop + 2;
  *out = b;
  return 0;
}

extern "C" void igmc_batch_destroy(igmc_batch* b) {
  if (!b) return;
  b->mem.release();
  delete b;
}

// The transposed copy of the dense blocks for an arena whose slots would not get one by size (<= 128 nodes a side): what
// the dense-layer kernels (k_dl_fwd / k_dl_bwd / k_dl_layer) read on the item side.  For models the subgraph kernel does
// not take although the blocks exist -- the sort-pool family, side features -- so that their conv layers run on the matrix
// cores instead of the CSR row walkers.  A no-op without dense blocks; a batch already in the arena is dropped.
extern "C" int igmc_batch_want_transposed(igmc_batch* b) {
  if (!b) IGMC_FAIL("null arena");
  if (b->d.relmT || !b->d.relm) return 0;
  if (b->mem.get(&b->d.relmT, (size_t)b->d.graph_cap * b->d.cap_v * b->d.relmT_ld)) IGMC_FAIL("hipMalloc failed (transposed blocks)");
  b->last_B = 0;      // (a batch extracted before has no transposed blocks: the arena counts as empty until the next extraction)
  return 0;
}

extern "C" int igmc_extract_batch(const igmc_graph* g, igmc_batch* b, const int32_t* d_link_u,
                                  const int32_t* d_link_v, const float* d_link_y, const int32_t* d_link_idx,
                                  int first, int B, double sample_ratio, uint64_t seed, uint64_t epoch,
                                  void* stream) {
  if (!g || !b || b->g != g) IGMC_FAIL("batch does not belong to this graph");
  if (B <= 0 || B > b->d.graph_cap) IGMC_FAIL("B exceeds the batch capacity");
  if (!d_link_u || !d_link_v || !d_link_y) IGMC_FAIL("null link arrays");
  igmc_launch_extract(g->d, b->d, d_link_u, d_link_v, d_link_y, d_link_idx, first, B, 0, sample_ratio, seed, epoch,
                      b->ctrl, b->lean && b->d.relm, stream);
  if (b->side_src)      // side features of the target nodes travel with the extraction (reference :250-253)
    igmc_launch_side_gather(b->side_src, b->n_side, d_link_idx, first, B, b->ctrl, b->side_buf, stream);
  HIPCHECK(hipGetLastError());
  b->last_B = B;
  return 0;
}

extern "C" int igmc_extract_batch_replay(const igmc_graph* g, igmc_batch* b, int B, const int32_t* h_unodes,
                                         const uint8_t* h_udist, const int32_t* h_uoff, const int32_t* h_vnodes,
                                         const uint8_t* h_vdist, const int32_t* h_voff, const float* h_y,
                                         void* stream) {
  if (!g || !b || b->g != g) IGMC_FAIL("batch does not belong to this graph");
  if (B <= 0 || B > b->d.graph_cap) IGMC_FAIL("B exceeds the batch capacity");
  const BatchDev& d = b->d;
  std::vector<int32_t> tl((size_t)B * d.slot, 0), nu(B), nv(B);
  std::vector<uint8_t> td((size_t)B * d.slot, 0);
  for (int gi = 0; gi < B; ++gi) {
    const int cu = h_uoff[gi + 1] - h_uoff[gi], cv = h_voff[gi + 1] - h_voff[gi];
    if (cu < 1 || cv < 1 || cu > d.cap_u || cv > d.cap_v) IGMC_FAIL("replayed node list exceeds the slot capacity");
    nu[gi] = cu;
    nv[gi] = cv;
    for (int i = 0; i < cu; ++i) {
      const int id = h_unodes[h_uoff[gi] + i];
      if (id < 0 || id >= g->d.n_users) IGMC_FAIL("user id out of range");
      tl[(size_t)gi * d.slot + i] = id;
      td[(size_t)gi * d.slot + i] = h_udist[h_uoff[gi] + i];
    }
    for (int i = 0; i < cv; ++i) {
      const int id = h_vnodes[h_voff[gi] + i];
      if (id < 0 || id >= g->d.n_items) IGMC_FAIL("item id out of range");
      tl[(size_t)gi * d.slot + d.cap_u + i] = id;
      td[(size_t)gi * d.slot + d.cap_u + i] = h_vdist[h_voff[gi] + i];
    }
  }
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHECK(hipMemcpy(d.t_list, tl.data(), tl.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d.t_dist, td.data(), td.size(), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d.n_users, nu.data(), B * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d.n_items, nv.data(), B * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d.y, h_y, B * sizeof(float), hipMemcpyHostToDevice));
  HIPCHECK(hipMemset(d.stamp, 0xFF, 4 * sizeof(int64_t)));        // node sets from the host: no cursor to compare with
  igmc_launch_extract(g->d, b->d, nullptr, nullptr, nullptr, nullptr, 0, B, 1, 1.0, 0, 0, nullptr, 0, stream);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  b->last_B = B;
  return 0;
}

extern "C" int igmc_extract_batch_cached(const igmc_graph* g, igmc_batch* b, const int64_t* d_uoff, const int32_t* d_unodes,
                                         const uint8_t* d_udist, const int64_t* d_voff, const int32_t* d_vnodes,
                                         const uint8_t* d_vdist, const float* d_link_y, const int32_t* d_link_idx, int first,
                                         int B, void* stream) {
  if (!g || !b || b->g != g) IGMC_FAIL("batch does not belong to this graph");
  if (B <= 0 || B > b->d.graph_cap) IGMC_FAIL("B exceeds the batch capacity");
  if (!d_uoff || !d_unodes || !d_udist || !d_voff || !d_vnodes || !d_vdist || !d_link_y) IGMC_FAIL("null cache arrays");
  igmc_launch_load_nodes(b->d, d_uoff, d_unodes, d_udist, d_voff, d_vnodes, d_vdist, d_link_y, d_link_idx, first, B, b->ctrl,
                         stream);
  igmc_launch_extract(g->d, b->d, nullptr, nullptr, nullptr, nullptr, 0, B, 1, 1.0, 0, 0, nullptr, b->lean && b->d.relm, stream);
  if (b->side_src) igmc_launch_side_gather(b->side_src, b->n_side, d_link_idx, first, B, b->ctrl, b->side_buf, stream);
  HIPCHECK(hipGetLastError());
  b->last_B = B;
  return 0;
}

// ---- a group of arenas extracted in one launch per stage
struct igmc_batch_set {
  std::vector<igmc_batch*> arenas;
  BatchDev* d_views;        // device array of the arenas' views
};

extern "C" int igmc_batch_set_create(igmc_batch* const* batches, int count, igmc_batch_set** out) {
  if (!batches || count < 1 || !out) IGMC_FAIL("bad arguments");
  std::vector<BatchDev> views;
  for (int i = 0; i < count; ++i) {
    const igmc_batch* b = batches[i];
    if (!b) IGMC_FAIL("null arena");
    const BatchDev &d = b->d, &d0 = batches[0]->d;
    if (b->g != batches[0]->g || d.cap_u != d0.cap_u || d.cap_v != d0.cap_v || d.graph_cap != d0.graph_cap || d.hop != d0.hop ||
        d.max_nodes_per_hop != d0.max_nodes_per_hop || (d.relm != nullptr) != (d0.relm != nullptr) ||
        (d.relmT != nullptr) != (d0.relmT != nullptr))
      IGMC_FAIL("the arenas of a set must share graph and geometry");
    if (!d.relm) IGMC_FAIL("group extraction needs arenas with dense induced blocks");
    views.push_back(d);
  }
  HIPCHECK(hipSetDevice(batches[0]->g->device));
  igmc_batch_set* s = new igmc_batch_set();
  s->arenas.assign(batches, batches + count);
  s->d_views = nullptr;
  if (hipMalloc((void**)&s->d_views, views.size() * sizeof(BatchDev)) != hipSuccess) {
    delete s;
    IGMC_FAIL("hipMalloc failed (arena set)");
  }
  HIPCHECK(hipMemcpy(s->d_views, views.data(), views.size() * sizeof(BatchDev), hipMemcpyHostToDevice));
  *out = s;
  return 0;
}

extern "C" void igmc_batch_set_destroy(igmc_batch_set* s) {
  if (!s) return;
  if (s->d_views) hipFree(s->d_views);
  delete s;
}

extern "C" int igmc_extract_group(const igmc_graph* g, igmc_batch_set* s, int count, const int32_t* d_link_u,
                                  const int32_t* d_link_v, const float* d_link_y, const int32_t* d_link_idx, int sel0,
                                  int B, double sample_ratio, uint64_t seed, float drop_p, int force_undirected,
                                  uint64_t drop_seed, void* stream) {
  if (!g || !s || count < 1 || count > (int)s->arenas.size()) IGMC_FAIL("bad arguments");
  if (!d_link_u || !d_link_v || !d_link_y) IGMC_FAIL("null link arrays");
  const igmc_batch* b0 = s->arenas[0];
  if (b0->g != g) IGMC_FAIL("the set does not belong to this graph");
  if (B <= 0 || B > b0->d.graph_cap) IGMC_FAIL("B exceeds the batch capacity");
  for (int i = 0; i < count; ++i) {
    const igmc_batch* b = s->arenas[i];
    if (!b->lean || !b->ctrl || b->ctrl != b0->ctrl || b->side_src)
      IGMC_FAIL("group extraction needs lean arenas with one control block attached and no side-feature source");
  }
  igmc_launch_extract_set(g->d, s->d_views, b0->d, count, d_link_u, d_link_v, d_link_y, d_link_idx, sel0, B, sample_ratio,
                          seed, b0->ctrl, drop_p, force_undirected, drop_seed, stream);
  HIPCHECK(hipGetLastError());
  for (int i = 0; i < count; ++i) s->arenas[i]->last_B = B;
  return 0;
}

// A lean arena carries no collated CSR after an extraction: whoever needs it (inspection, the flag kernels, the
// per-layer model kernels) emits it first.  Emission is idempotent, and a hipGraph replay of the extraction leaves no
// host-side trace, so a lean arena re-emits on every such call.
static void ensure_csr(const igmc_batch* b, void* stream) {
  if (b->lean && b->d.relm && b->last_B > 0) igmc_launch_emit(b->d, b->last_B, stream);
}

extern "C" int igmc_batch_assume_size(igmc_batch* b, int B) {
  if (!b) IGMC_FAIL("null batch");
  if (B <= 0 || B > b->d.graph_cap) IGMC_FAIL("batch size out of range");
  b->last_B = B;
  return 0;
}

extern "C" int igmc_batch_set_lean(igmc_batch* b, int lean) {
  if (!b) IGMC_FAIL("null batch");
  b->lean = lean ? 1 : 0;
  return 0;
}

extern "C" int igmc_batch_edge_dropout(igmc_batch* b, float p, int force_undirected, uint64_t seed, uint64_t step,
                                       void* stream) {
  if (!b) IGMC_FAIL("null batch");
  if (b->lean && b->d.relm && b->last_B > 0) {
    // lean arena: the same draws taken on the dense blocks; a CSR emitted later derives its flags from them
    igmc_launch_relm_dropout(b->d, b->last_B, p, force_undirected, seed, step, b->ctrl, stream);
  } else {
    // the flag kernel walks the collated CSR (and mirrors the bits into the dense block)
    igmc_launch_edge_flags(b->d, p, force_undirected, seed, step, b->ctrl, stream);
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int igmc_batch_get_info(const igmc_batch* b, igmc_batch_info* out, void* stream) {
  if (!b || !out) IGMC_FAIL("null argument");
  ensure_csr(b, stream);
  int32_t t[8];
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHECK(hipMemcpy(t, b->d.totals, sizeof(t), hipMemcpyDeviceToHost));
  out->num_graphs = t[3];
  out->num_nodes = t[2] ? t[4] : t[0];
  out->num_edges = t[2] ? t[5] : t[1];
  out->overflow = t[2];
  out->node_capacity = b->d.node_cap;
  out->edge_capacity = b->d.edge_cap;
  out->num_labels = b->d.num_labels;
  out->hop = b->d.hop;
  return 0;
}

extern "C" int igmc_batch_set_edge_flags(igmc_batch* b, const uint8_t* h_flags, int64_t n) {
  if (!b || !h_flags) IGMC_FAIL("null argument");
  if (n > b->d.edge_cap) IGMC_FAIL("too many flags");
  ensure_csr(b, nullptr);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(b->d.eflag, h_flags, (size_t)n, hipMemcpyHostToDevice));
  igmc_launch_relm_flags(b->d, nullptr);        // the dense block mirrors the keep bits
  HIPCHECK(hipDeviceSynchronize());
  return 0;
}

extern "C" int igmc_batch_clear_edge_flags(igmc_batch* b) {
  if (!b) IGMC_FAIL("null batch");
  igmc_batch_info info;
  if (igmc_batch_get_info(b, &info, nullptr)) return 1;
  if (info.num_edges > 0) {
    HIPCHECK(hipMemset(b->d.eflag, 3, (size_t)info.num_edges));
    igmc_launch_relm_flags(b->d, nullptr);
    HIPCHECK(hipDeviceSynchronize());
  }
  return 0;
}

extern "C" int igmc_batch_download(const igmc_batch* b, int32_t* node_off, int32_t* n_users, uint8_t* node_label,
                                   int32_t* node_gid, int32_t* node_graph, int32_t* row_ptr, int32_t* col,
                                   uint8_t* erel, uint8_t* elab, uint8_t* eflag, float* y, void* stream) {
  igmc_batch_info info;
  if (igmc_batch_get_info(b, &info, stream)) return 1;
  if (info.overflow) IGMC_FAIL("batch arena overflow");
  const int B = info.num_graphs, N = info.num_nodes, E = info.num_edges;
  const BatchDev& d = b->d;
  if (node_off) HIPCHECK(hipMemcpy(node_off, d.node_off, (B + 1) * 4, hipMemcpyDeviceToHost));
  if (n_users) HIPCHECK(hipMemcpy(n_users, d.n_users, B * 4, hipMemcpyDeviceToHost));
  if (node_label && N) HIPCHECK(hipMemcpy(node_label, d.node_label, N, hipMemcpyDeviceToHost));
  if (node_gid && N) HIPCHECK(hipMemcpy(node_gid, d.node_gid, N * 4, hipMemcpyDeviceToHost));
  if (node_graph && N) HIPCHECK(hipMemcpy(node_graph, d.node_graph, N * 4, hipMemcpyDeviceToHost));
  if (row_ptr) HIPCHECK(hipMemcpy(row_ptr, d.row_ptr, (N + 1) * 4, hipMemcpyDeviceToHost));
  if ((col || erel) && E) {
    std::vector<uint32_t> ecr(E);
    HIPCHECK(hipMemcpy(ecr.data(), d.ecr, (size_t)E * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < E; ++e) {
      if (col) col[e] = (int32_t)(ecr[e] & 0xFFFFFFu);
      if (erel) erel[e] = (uint8_t)(ecr[e] >> 24);
    }
  }
  if (elab && E) {
    std::vector<uint16_t> code(E);
    HIPCHECK(hipMemcpy(code.data(), d.ecode, (size_t)E * 2, hipMemcpyDeviceToHost));
    for (int e = 0; e < E; ++e) elab[e] = (uint8_t)(code[e] % d.num_labels);
  }
  if (eflag && E) HIPCHECK(hipMemcpy(eflag, d.eflag, E, hipMemcpyDeviceToHost));
  if (y) HIPCHECK(hipMemcpy(y, d.y, B * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void* igmc_batch_device_ptr(const igmc_batch* b, int which) {
  if (!b) return nullptr;
  const BatchDev& d = b->d;
  switch (which) {
    case IGMC_BUF_NODE_OFF: return d.node_off;
    case IGMC_BUF_N_USERS: return d.n_users;
    case IGMC_BUF_NODE_LABEL: return d.node_label;
    case IGMC_BUF_NODE_GID: return d.node_gid;
    case IGMC_BUF_NODE_GRAPH: return d.node_graph;
    case IGMC_BUF_ROW_PTR: return d.row_ptr;
    case IGMC_BUF_ECR: return d.ecr;
    case IGMC_BUF_ECODE: return d.ecode;
    case IGMC_BUF_EFLAG: return d.eflag;
    case IGMC_BUF_Y: return d.y;
    case IGMC_BUF_TOTALS: return d.totals;
  }
  return nullptr;
}

extern "C" int igmc_batch_set_side_features(igmc_batch* b, const float* d_feat, int n_side) {
  if (!b) IGMC_FAIL("null batch");
  b->side = d_feat;
  b->n_side = n_side;
  b->side_src = nullptr;
  return 0;
}

extern "C" int igmc_batch_bind_side_source(igmc_batch* b, const float* d_side_all, int n_side) {
  if (!b) IGMC_FAIL("null batch");
  if (!d_side_all || n_side <= 0) {      // unbind
    b->side_src = nullptr;
    b->side = nullptr;
    b->n_side = 0;
    return 0;
  }
  if (b->side_buf_cols < n_side) {
    // (a wider re-bind carves a new buffer out of the arena's slabs; the old one stays part of them and is released with
    //  the arena -- slabs have no per-buffer free -- so grow geometrically to bound what repeated re-binds can take)
    HIPCHECK(hipSetDevice(b->g->device));
    const int cols = std::max(n_side, 2 * b->side_buf_cols);
    if (b->mem.get(&b->side_buf, (size_t)b->d.graph_cap * cols)) IGMC_FAIL("hipMalloc failed (side features)");
    b->side_buf_cols = cols;
  }
  b->side_src = d_side_all;
  b->side = b->side_buf;
  b->n_side = n_side;
  return 0;
}

// ------------------------------------------------------------------ model
extern "C" int igmc_model_create(int device, int num_relations, int num_bases, int num_labels, int n_side,
                                 int max_nodes, int max_edges, int max_graphs, igmc_model** out) {
  if (!out) IGMC_FAIL("null out");
  if (num_bases != 4) IGMC_FAIL("the gfx950 kernels are built for num_bases == 4 (reference Main.py:393)");
  if (num_relations < 1 || num_relations > 255) IGMC_FAIL("num_relations out of range");
  if (num_relations * num_labels + num_labels + 1 > 320) IGMC_FAIL("num_relations * num_labels too large for the layer-0 gradient kernel");
  if (num_labels < 2 || n_side < 0 || max_nodes < 1 || max_graphs < 1) IGMC_FAIL("bad sizes");
  HIPCHECK(hipSetDevice(device));
  igmc_model* m = new igmc_model();
  m->device = device;
  m->last_B = 0;
  m->last_training = 0;
  m->last_flags = 0;
  m->img_valid = 0;
  m->img_hint = 0;
  m->img_params = nullptr;
  ModelDev& d = m->d;
  d.img_current = 0;
  d.adam_m1 = nullptr;
  d.adam_m2 = nullptr;
  d.R = num_relations;
  d.Bs = 4;
  d.L = num_labels;
  d.S = n_side;
  d.D = 256 + n_side;
  int64_t off = 0;
  for (int l = 0; l < 4; ++l) {
    const int fin = l == 0 ? d.L : 32;
    d.off_basis[l] = off; off += (int64_t)4 * fin * 32;
    d.off_root[l] = off;  off += (int64_t)fin * 32;
    d.off_bias[l] = off;  off += 32;
    d.off_att[l] = off;   off += (int64_t)d.R * 4;
  }
  d.off_l1w = off; off += (int64_t)128 * d.D;
  d.off_l1b = off; off += 128;
  d.off_l2w = off; off += 128;
  d.off_l2b = off; off += 1;
  d.n_params = off;
  d.node_cap = max_nodes;
  d.edge_cap = max_edges;
  d.graph_cap = max_graphs;
  Allocs& M = m->mem;
  int fail = 0;
  const size_t N = (size_t)max_nodes, Bc = (size_t)max_graphs;
  for (int l = 0; l < 4; ++l) fail |= M.get(&d.h[l], N * 32) | M.get(&d.dpre[l], N * 32);
  fail |= M.get(&d.agg, N * 128);
  for (int l = 0; l < 3; ++l) fail |= M.get(&d.gagg[l], N * 128) | M.get(&d.Y[l], N * 128);
  fail |= M.get(&d.feat, Bc * d.D) | M.get(&d.a1, Bc * 128) | M.get(&d.lmask, Bc * 128) | M.get(&d.dz, Bc * 128) |
          M.get(&d.gfeat, Bc * d.D) | M.get(&d.err, Bc) | M.get(&d.cnt0, N * d.R * d.L);
  const size_t rows0 = (size_t)d.R * d.L + d.L + 1;
  // relation-space gradient tables of the subgraph kernel (graphstep2.hip), R <= 5 only
  d.ts_part = nullptr;
  d.ts_raw = nullptr;
  d.ts_stride = (d.R * 32 + 33) * 32;
  d.fin_stash = nullptr;
  d.datt_part = nullptr;
  if (d.R <= G2_NR * G2_NG_MAX)
  {   // (sums + d att partials in ONE allocation: a data-parallel step exchanges them as one span)
    fail |= M.get(&d.ts_part, (size_t)4 * IGMC_TS_BLOCKS * d.ts_stride) |
            M.get(&d.ts_raw, (size_t)4 * d.ts_stride + (size_t)4 * d.ts_stride / 32 * 4);
    if (!fail) d.datt_part = d.ts_raw + (size_t)4 * d.ts_stride;
  }
  if (d.R <= 128) fail |= M.get(&d.fin_stash, (size_t)4 * IGMC_STASH_LAYER + 16);     // weights-only stash of k_finalize_ts (both modes)
  d.g2_ex = nullptr;
  d.g2_fx = nullptr;
  d.g2_px = nullptr;
  d.g2_px_stride = 0;
  d.g2_w = nullptr;
  d.g2_graphs = 0;
  // exchange regions [32 features][256 nodes a side] of the one-launch dense layers (k_dl_fwd / k_dl_bwd); the subgraph
  // kernel uses the first 128 nodes of a region.  655 KB per subgraph slot: HBM is not what this path is short of.
  d.ex_nodes = 256;
  d.g2_ex_stride = (size_t)Bc * 2 * 32 * d.ex_nodes;
  const int wide = d.R <= G2_NR * G2_NG_MAX;      // relation groups of the dense-layer kernels (g2_image.h)
  // (both exchanges go through the L2 of one XCD: where the probe says workgroups b and b + 8 do not share one, neither kernel
  //  family is eligible and ~1 MB per subgraph slot of exchange regions would be allocated for nothing -- ADVICE r5)
  const int xcd_ok = igmc_g2_xcd_ok();
  if (wide && Bc <= 2048 && xcd_ok) {
    fail |= M.get(&d.g2_ex, 5 * d.g2_ex_stride) | M.get(&d.g2_fx, Bc * 256) | M.get(&d.g2_w, g2_w_words(d.R, d.L));
    d.g2_graphs = (int)Bc;
    // the subgraph kernel's plane exchange (R <= 5): 320 KB per subgraph slot
    if (d.R <= G2_NR && Bc <= 1024) {
      d.g2_px_stride = (size_t)Bc * 2 * 32768;
      fail |= M.get(&d.g2_px, 5 * d.g2_px_stride);
    } else if (d.R <= G2_NR) {
      static bool told = false;
      if (!told) fprintf(stderr, "[igmc] batches of more than 1024 subgraphs: no plane-exchange regions -- the dense-layer / per-layer kernels take the steps, not the subgraph kernel\n");
      told = true;
    }
  } else if (wide) {
    fail |= M.get(&d.g2_w, g2_w_words(d.R, d.L));      // weight images alone: the dense per-layer kernels (any head)
  }
  fail |= M.get(&d.gs_bar, 2 * Bc + 1);
  fail |= M.get(&d.gs_ts, 4);
  d.gs_err = d.gs_bar ? d.gs_bar + 2 * Bc : nullptr;
  fail |= M.get(&d.wg_part, (size_t)4 * IGMC_WG_BLOCKS * igmc_wg_stride()) |
          M.get(&d.gatt_part, (size_t)3 * IGMC_GATHER_BLOCKS * d.R * 4) |
          M.get(&d.l0_part, (size_t)IGMC_L0_BLOCKS * rows0 * 32) |
          M.get(&d.graw, (size_t)3 * igmc_wg_stride() + 3 * d.R * 4 + rows0 * 32) | M.get(&d.arr_part, 4);
  d.side = nullptr;
  d.ctrl = nullptr;
  d.dcat[0] = d.dcat[1] = d.dcat[2] = nullptr;
  fail |= M.get(&m->done_ctr, 8);      // [0] step-wide, [1..4] per conv layer (k_finalize)
  if (fail) {
    M.release();
    delete m;
    IGMC_FAIL("hipMalloc failed (model workspace)");
  }
  {
    hipStream_t st1, st2;
    if (hipStreamCreateWithFlags(&st1, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) != hipSuccess) {
      M.release();
      delete m;
      IGMC_FAIL("hipStreamCreate failed");
    }
    m->ax.s1 = st1;
    m->ax.s2 = st2;
    for (int i = 0; i < 8; ++i) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) IGMC_FAIL("hipEventCreate failed");
      m->ax.ev[i] = e;
    }
  }
  HIPCHECK(hipMemset(m->done_ctr, 0, 8 * sizeof(int)));
  HIPCHECK(hipMemset(m->d.gs_bar, 0, (2 * (size_t)max_graphs + 1) * sizeof(int)));
  {
    const unsigned long long ts0[4] = {~0ull, 0ull, 0ull, 0ull};
    HIPCHECK(hipMemcpy(m->d.gs_ts, ts0, sizeof(ts0), hipMemcpyHostToDevice));
  }
  if (m->d.g2_ex) {
    HIPCHECK(hipMemset(m->d.g2_ex, 0, 5 * m->d.g2_ex_stride * sizeof(unsigned long long)));
    HIPCHECK(hipMemset(m->d.g2_fx, 0, (size_t)max_graphs * 256 * sizeof(unsigned long long)));
    if (m->d.g2_px) HIPCHECK(hipMemset(m->d.g2_px, 0, 5 * m->d.g2_px_stride));      // flags 0 (never a launch's tag), planes finite
  }
  if (igmc_model_prepare(d)) {
    M.release();
    delete m;
    IGMC_FAIL("hipFuncSetAttribute failed");
  }
  *out = m;
  return 0;
}

// debug aid (tests): host copy of the keep mask the last TRAINING forward drew / was given for the 0.5 MLP dropout
// (reference models.py:212), uint8 [n <= max_graphs * 128].  Synchronises the device.
extern "C" int igmc_debug_lin_mask(const igmc_model* m, uint8_t* h_out, int64_t n) {
  if (!m || !h_out) IGMC_FAIL("null argument");
  if (n < 0 || n > (int64_t)m->d.graph_cap * 128) IGMC_FAIL("mask size out of range");
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(h_out, m->d.lmask, (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void igmc_model_destroy(igmc_model* m) {
  if (!m) return;
  for (int i = 0; i < 8; ++i) hipEventDestroy((hipEvent_t)m->ax.ev[i]);
  hipStreamDestroy((hipStream_t)m->ax.s1);
  hipStreamDestroy((hipStream_t)m->ax.s2);
  m->mem.release();
  delete m;
}

extern "C" int64_t igmc_param_count(const igmc_model* m) { return m ? m->d.n_params : 0; }

extern "C" int64_t igmc_param_offset(const igmc_model* m, int layer, int which, int64_t* count) {
  if (!m) return -1;
  const ModelDev& d = m->d;
  const int fin = (layer == 0) ? d.L : 32;
  int64_t off = -1, n = 0;
  switch (which) {
    case IGMC_P_BASIS: off = d.off_basis[layer & 3]; n = (int64_t)4 * fin * 32; break;
    case IGMC_P_ROOT: off = d.off_root[layer & 3]; n = (int64_t)fin * 32; break;
    case IGMC_P_BIAS: off = d.off_bias[layer & 3]; n = 32; break;
    case IGMC_P_ATT: off = d.off_att[layer & 3]; n = (int64_t)d.R * 4; break;
    case IGMC_P_LIN1_W: off = d.off_l1w; n = (int64_t)128 * d.D; break;
    case IGMC_P_LIN1_B: off = d.off_l1b; n = 128; break;
    case IGMC_P_LIN2_W: off = d.off_l2w; n = 128; break;
    case IGMC_P_LIN2_B: off = d.off_l2b; n = 1; break;
  }
  if (count) *count = n;
  return off;
}

// the per-layer row-walker kernels read the collated CSR; the matrix-core subgraph kernel does not
static void csr_for_model(const igmc_model* m, const igmc_batch* b, int dense_capable_call, void* stream) {
  if (!b->lean) return;
  G2Layout lay;
  int cs = 1;
  const int rows0 = m->d.R * m->d.L + m->d.L + 1;
  if (dense_capable_call && rows0 <= 32 && igmc_g2_eligible(m->d, b->d, b->last_B, &lay, &cs)) return;
  // dense per-layer path: it needs the node arrays only, and a lean extraction of such an arena (one with the transposed
  // block) has left them behind (k_emit_nodes in the extraction branch)
  if (igmc_dl_eligible(m->d, b->d, b->last_B) && b->d.relm && b->last_B > 0) return;
  if (dense_capable_call && igmc_dl_wide(m->d, b->d, b->last_B) && b->last_B > 0) return;      // (relation groups: same)
  ensure_csr(b, stream);
}

// 1 when the dense per-layer kernels (k_dl_layer) take the conv layers of this arena
extern "C" int igmc_model_dense_layers(const igmc_model* m, const igmc_batch* b, int B) {
  if (!m || !b) return 0;
  return (igmc_dl_eligible(m->d, b->d, B) || igmc_dl_wide(m->d, b->d, B)) ? 1 : 0;
}

extern "C" int igmc_model_dense_path(const igmc_model* m, const igmc_batch* b, int B) {
  if (!m || !b) return 0;
  G2Layout lay;
  int cs = 1;
  const int rows0 = m->d.R * m->d.L + m->d.L + 1;
  return (rows0 <= 32 && m->d.D % 16 == 0 && igmc_g2_eligible(m->d, b->d, B, &lay, &cs)) ? 1 : 0;
}

extern "C" int igmc_model_step_form(const igmc_model* m, const igmc_batch* b, int B) {
  if (!m || !b) return 0;
  if (igmc_model_dense_path(m, b, B)) return 1;
  if (igmc_dl_wide_gsplit(m->d, b->d, B)) return 3;
  if (igmc_dl_wide(m->d, b->d, B) || (igmc_dl_eligible(m->d, b->d, B) && igmc_dl_fwd_eligible(m->d, b->d, B))) return 2;
  return 0;
}

static int check_fit(igmc_model* m, const igmc_batch* b, std::string* why) {
  if (!m || !b) { *why = "null model or batch"; return 1; }
  if (b->last_B <= 0) { *why = "batch is empty (run igmc_extract_batch first)"; return 1; }
  if (b->d.node_cap > m->d.node_cap || b->d.graph_cap > m->d.graph_cap) { *why = "batch capacity exceeds the model workspace"; return 1; }
  if (b->d.num_labels != m->d.L) { *why = "num_labels mismatch (hop)"; return 1; }
  if (b->g->max_rel >= m->d.R) { *why = "graph has more relations than the model"; return 1; }
  if (m->d.S != b->n_side && m->d.S > 0) { *why = "side feature width mismatch"; return 1; }
  return 0;
}

extern "C" int igmc_model_forward(igmc_model* m, const float* d_params, const igmc_batch* b, int training,
                                  int use_edge_flags, const uint8_t* d_lin_mask, uint64_t seed, uint64_t step,
                                  float multiply_by, float* d_out, void* stream) {
  std::string why;
  if (check_fit(m, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out) IGMC_FAIL("null buffer");
  m->d.side = b->side;
  csr_for_model(m, b, !training, stream);
  ImgScope img(m, d_params);
  igmc_launch_forward(m->d, m->ax, b->d, d_params, b->last_B, training, use_edge_flags, d_lin_mask, seed, step, multiply_by,
                      d_out, stream);
  img.done_unchanged();
  HIPCHECK(hipGetLastError());
  m->last_B = b->last_B;
  m->last_training = training;
  m->last_flags = use_edge_flags;
  return 0;
}

extern "C" int igmc_model_backward(igmc_model* m, const float* d_params, const igmc_batch* b, const float* d_gout,
                                   float multiply_by, float* d_grad, void* stream) {
  std::string why;
  if (check_fit(m, b, &why)) IGMC_FAIL(why);
  if (!m->last_training || m->last_B != b->last_B) IGMC_FAIL("backward needs a preceding training-mode forward on this batch");
  if (!d_params || !d_gout || !d_grad) IGMC_FAIL("null buffer");
  igmc_launch_backward(m->d, m->ax, b->d, d_params, b->last_B, m->last_flags, d_gout, 0, 0.f, multiply_by, 2.f, 0.f, d_grad,
                       stream);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int igmc_model_loss_grad(igmc_model* m, const float* d_params, const igmc_batch* b, int use_edge_flags,
                                    const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float multiply_by,
                                    float ARR, float grad_scale, float arr_scale, float* d_out, float* d_grad,
                                    float* d_loss, void* stream) {
  std::string why;
  if (check_fit(m, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out || !d_grad) IGMC_FAIL("null buffer");
  m->d.side = b->side;
  csr_for_model(m, b, 1, stream);
  ImgScope img(m, d_params);
  igmc_launch_loss_grad(m->d, m->ax, b->d, (float*)d_params, b->last_B, use_edge_flags, d_lin_mask, seed, step,
                        multiply_by, ARR, grad_scale, arr_scale, d_out, d_grad, d_loss, nullptr, stream);
  img.done_unchanged();
  HIPCHECK(hipGetLastError());
  m->last_B = b->last_B;
  m->last_training = 1;
  m->last_flags = use_edge_flags;
  return 0;
}

extern "C" int igmc_adam_step(float* d_params, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, int64_t n,
                              int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              void* stream) {
  if (!d_params || !d_grad || !d_exp_avg || !d_exp_avg_sq || n <= 0 || step < 1) IGMC_FAIL("bad arguments");
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  igmc_launch_adam(d_params, d_grad, d_exp_avg, d_exp_avg_sq, n, (float)((double)lr / bc1),
                   (float)(1.0 / std::sqrt(bc2)), beta1, beta2, eps, weight_decay, nullptr, 0, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int igmc_sse_accumulate(const float* d_out, const igmc_batch* b, double* d_acc, void* stream) {
  if (!d_out || !b || !d_acc) IGMC_FAIL("null argument");
  igmc_launch_sse(b->d, d_out, d_acc, nullptr, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}
extern "C" int igmc_sse_accumulate_tick(const float* d_out, const igmc_batch* b, double* d_acc, int64_t* d_ctrl, void* stream) {
  if (!d_out || !b || !d_acc || !d_ctrl) IGMC_FAIL("null argument");
  igmc_launch_sse(b->d, d_out, d_acc, d_ctrl, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ device-side step control
extern "C" int igmc_ctrl_tick(int64_t* d_ctrl, void* stream) {
  if (!d_ctrl) IGMC_FAIL("null ctrl");
  igmc_launch_tick(d_ctrl, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}
extern "C" int igmc_ctrl_regroup(int64_t* d_ctrl, int M, int64_t first_cur, int64_t first_next, void* stream) {
  if (!d_ctrl || M < 1 || first_cur < 0 || first_next < 0) IGMC_FAIL("bad arguments");
  igmc_launch_regroup(d_ctrl, M, first_cur, first_next, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}
extern "C" int igmc_ctrl_gate(int64_t* d_ctrl, int q, int gk_min, double delay_us, int delay_always, double timeout_us,
                              void* stream) {
  if (!d_ctrl || gk_min < 0 || !(timeout_us >= 0.0) || !(delay_us >= 0.0) || delay_us > 1000.0) IGMC_FAIL("bad arguments");
  // (wall_clock64: 100 MHz)
  igmc_launch_gate(d_ctrl, q & 1, gk_min, (long long)(delay_us * 100.0), delay_always != 0, (long long)(timeout_us * 100.0), stream);
  HIPCHECK(hipGetLastError());
  return 0;
}
extern "C" int igmc_batch_set_ctrl(igmc_batch* b, const int64_t* d_ctrl) {
  if (!b) IGMC_FAIL("null batch");
  b->ctrl = d_ctrl;
  return 0;
}
extern "C" int igmc_model_set_ctrl(igmc_model* m, const int64_t* d_ctrl) {
  if (!m) IGMC_FAIL("null model");
  m->d.ctrl = d_ctrl;
  return 0;
}
// INVARIANT of the plane / row exchange regions (g2_px, g2_ex, g2_fx): every value in them is FINITE.  A consumer copies the
// whole plane image of the opposite side, rows past that side's extent included -- whatever an earlier launch left in the
// subgraph slot --, and relies on 0 * stale == 0 (the A-block bytes there are zero).  Steps that ran on non-finite parameters
// can leave NaN / Inf rows behind which would poison every later gather of the slot, even after the parameters are restored:
// the regions are cleared (flags included: 0 is never a launch's tag) whenever parameters are loaded (models.load_state_dict)
// and when an exchange timed out (igmc_model_check below).
extern "C" int igmc_model_reset_exchange(igmc_model* m, void* stream) {
  if (!m) IGMC_FAIL("null model");
  hipStream_t st = (hipStream_t)stream;
  if (m->d.g2_ex) {
    HIPCHECK(hipMemsetAsync(m->d.g2_ex, 0, 5 * m->d.g2_ex_stride * sizeof(unsigned long long), st));
    HIPCHECK(hipMemsetAsync(m->d.g2_fx, 0, (size_t)m->d.g2_graphs * 256 * sizeof(unsigned long long), st));
    if (m->d.g2_px) HIPCHECK(hipMemsetAsync(m->d.g2_px, 0, 5 * m->d.g2_px_stride, st));
  }
  return 0;
}
extern "C" int igmc_model_check(igmc_model* m, void* stream) {
  if (!m) IGMC_FAIL("null model");
  int v = 0;
  if (m->d.gs_err) {
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHECK(hipMemcpy(&v, m->d.gs_err, sizeof(int), hipMemcpyDeviceToHost));
    if (v) {
      HIPCHECK(hipMemset(m->d.gs_bar, 0, (2 * (size_t)m->d.graph_cap + 1) * sizeof(int)));
      (void)igmc_model_reset_exchange(m, stream);        // (sequence number and flags restart together; partial rows gone)
      HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
      IGMC_FAIL("a workgroup-cluster exchange of k_graph_step2 timed out (GPU shared with another job?): results of the "
                "affected steps are invalid; set IGMC_GS_CLUSTER=1 or IGMC_GRAPH_STEP=0");
    }
  }
  return 0;
}
extern "C" int igmc_adam_step_ctrl(float* d_params, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq,
                                   int64_t n, const int64_t* d_ctrl, void* stream) {
  if (!d_params || !d_grad || !d_exp_avg || !d_exp_avg_sq || n <= 0 || !d_ctrl) IGMC_FAIL("bad arguments");
  igmc_launch_adam(d_params, d_grad, d_exp_avg, d_exp_avg_sq, n, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, (int64_t*)d_ctrl, 0, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int igmc_step_finish(igmc_model* m, const igmc_batch* b, float* d_params, const float* d_grad,
                                float* d_exp_avg, float* d_exp_avg_sq, float ARR, float* d_loss, double* d_total,
                                int64_t* d_ctrl, int64_t step, float lr, float beta1, float beta2, float eps,
                                float weight_decay, void* stream) {
  if (!m || !b || !d_params || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_loss) IGMC_FAIL("bad arguments");
  if (!d_ctrl && step < 1) IGMC_FAIL("step must be >= 1");
  float step_size = 0.f, inv = 0.f;
  if (!d_ctrl) {
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    step_size = (float)((double)lr / bc1);
    inv = (float)(1.0 / std::sqrt(bc2));
  }
  igmc_launch_finish(m->d, b->d, d_params, d_grad, d_exp_avg, d_exp_avg_sq, step_size, inv, beta1, beta2, eps,
                     weight_decay, d_ctrl, ARR, d_loss, d_total, m->last_flags, stream);
  m->img_valid = 0;      // (the parameters moved, the weight images did not)
  m->img_hint = 0;
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int igmc_train_step(igmc_model* m, float* d_params, const igmc_batch* b, int use_edge_flags,
                               const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float multiply_by, float ARR,
                               float* d_out, float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_loss,
                               double* d_total, int64_t* d_ctrl, int64_t adam_t, float lr, float beta1, float beta2,
                               float eps, float weight_decay, void* stream) {
  std::string why;
  if (check_fit(m, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_loss) IGMC_FAIL("null buffer");
  if (!d_ctrl && adam_t < 1) IGMC_FAIL("adam_t must be >= 1");
  float step_size = 0.f, inv = 0.f;
  if (!d_ctrl) {
    const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_t);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)adam_t);
    step_size = (float)((double)lr / bc1);
    inv = (float)(1.0 / std::sqrt(bc2));
  }
  m->d.side = b->side;
  csr_for_model(m, b, 1, stream);
  ImgScope img(m, d_params);
  int emitted = 0;
  igmc_launch_train_step(m->d, m->ax, b->d, d_params, b->last_B, use_edge_flags, d_lin_mask, seed, step, multiply_by, ARR,
                         d_out, d_grad, d_exp_avg, d_exp_avg_sq, step_size, inv, beta1, beta2, eps, weight_decay, d_ctrl,
                         m->done_ctr, d_loss, d_total, stream, 0.f, nullptr, &emitted);
  img.done_updated(emitted);
  HIPCHECK(hipGetLastError());
  m->last_B = b->last_B;
  m->last_training = 1;
  m->last_flags = use_edge_flags;
  return 0;
}

// ------------------------------------------------------------------ gradient exchange (RCCL behind the C ABI)
#define IGMC_MAX_PEERS 16
struct igmc_comm {
  void* nccl;        // ncclComm_t (NULL: host-callback / peer communicator, or the emulation build's single rank)
  int rank, world, device;
  igmc_allreduce_fn host_fn;      // igmc_comm_create_host: the caller's own sum over the ranks
  void* host_user;
  // peer communicator (igmc_comm_peer_alloc / _connect): every rank publishes into its OWN buffer, mapped by the others
  unsigned long long* pub[IGMC_MAX_PEERS];      // [2 slots][cap] {f32, tag} words of rank r (pub[rank] = the local buffer)
  int64_t cap;                                  // floats a slot holds
  int* d_state;                                 // device: [0] launch sequence number, [1] workgroups done, [2] poll timed out
  int peer;                                     // 1 = peer communicator
  int connected;
  int fine;                                     // the local buffer is fine-grained device memory
};
#ifndef IGMC_HIPEMU
#include <dlfcn.h>
// RCCL's entry points, resolved at the first use.  The few types needed are restated here (rccl.h: ncclUniqueId = 128
// opaque bytes; ncclFloat = 7, ncclSum = 0; results: 0 = success) so that the build needs no RCCL headers.
struct NcclId { char internal[128]; };
struct RcclApi {
  int (*GetUniqueId)(NcclId*);
  int (*CommInitRank)(void**, int, NcclId, int);
  int (*CommDestroy)(void*);
  int (*CommCount)(const void*, int*);
  int (*CommUserRank)(const void*, int*);
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  const char* (*GetErrorString)(int);
  bool ok = false;
};
static RcclApi g_rccl;
static int rccl_load(std::string* why) {
  if (g_rccl.ok) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // the copy the process already holds, if any (same soname)
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { *why = std::string("cannot load librccl.so.1: ") + dlerror(); return 1; }
  struct { const char* name; void** slot; } syms[] = {
      {"ncclGetUniqueId", (void**)&g_rccl.GetUniqueId}, {"ncclCommInitRank", (void**)&g_rccl.CommInitRank},
      {"ncclCommDestroy", (void**)&g_rccl.CommDestroy}, {"ncclCommCount", (void**)&g_rccl.CommCount},
      {"ncclCommUserRank", (void**)&g_rccl.CommUserRank}, {"ncclAllReduce", (void**)&g_rccl.AllReduce},
      {"ncclGroupStart", (void**)&g_rccl.GroupStart}, {"ncclGroupEnd", (void**)&g_rccl.GroupEnd},
      {"ncclGetErrorString", (void**)&g_rccl.GetErrorString}};
  for (auto& sy : syms) {
    *sy.slot = dlsym(h, sy.name);
    if (!*sy.slot) { *why = std::string("librccl.so.1 lacks ") + sy.name; return 1; }
  }
  g_rccl.ok = true;
  return 0;
}
#define RCCLCHECK(expr)                                                                              \
  do {                                                                                               \
    int r_ = (expr);                                                                                 \
    if (r_ != 0) {                                                                                   \
      g_err = std::string(__func__) + ": " #expr " -> " + g_rccl.GetErrorString(r_);                 \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)
#endif

__global__ __launch_bounds__(IGMC_BLOCK) void k_scale_flat(float* p, int64_t n, float s) {
  for (int64_t i = (int64_t)blockIdx.x * IGMC_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * IGMC_BLOCK) p[i] *= s;
}

extern "C" int igmc_comm_unique_id(uint8_t* h_id128) {
  if (!h_id128) IGMC_FAIL("null id buffer");
#ifndef IGMC_HIPEMU
  std::string why;
  if (rccl_load(&why)) IGMC_FAIL(why);
  NcclId id;
  RCCLCHECK(g_rccl.GetUniqueId(&id));
  memcpy(h_id128, id.internal, 128);
#else
  memset(h_id128, 0, 128);
#endif
  return 0;
}

extern "C" int igmc_comm_create(const uint8_t* h_id128, int rank, int world, int device, igmc_comm** out) {
  if (!h_id128 || !out || world < 1 || rank < 0 || rank >= world) IGMC_FAIL("bad arguments");
  igmc_comm* c = new igmc_comm();
  c->nccl = nullptr;
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->host_fn = nullptr;
  c->host_user = nullptr;
  c->peer = 0;
  c->d_state = nullptr;
#ifndef IGMC_HIPEMU
  std::string why;
  if (rccl_load(&why)) { delete c; IGMC_FAIL(why); }
  HIPCHECK(hipSetDevice(device));
  NcclId id;
  memcpy(id.internal, h_id128, 128);
  const int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
  if (r != 0) {
    delete c;
    IGMC_FAIL(std::string("ncclCommInitRank -> ") + g_rccl.GetErrorString(r));
  }
#else
  if (world != 1) { delete c; IGMC_FAIL("the emulation build has no RCCL: one rank only"); }
#endif
  *out = c;
  return 0;
}

extern "C" int igmc_comm_create_host(igmc_allreduce_fn fn, void* user, int rank, int world, igmc_comm** out) {
  if (!fn || !out || world < 1 || rank < 0 || rank >= world) IGMC_FAIL("bad arguments");
  igmc_comm* c = new igmc_comm();
  c->nccl = nullptr;
  c->rank = rank;
  c->world = world;
  c->device = -1;
  c->host_fn = fn;
  c->host_user = user;
  c->peer = 0;
  c->d_state = nullptr;
  *out = c;
  return 0;
}

// ---- peer communicator: a one-shot all-reduce over peer-mapped buffers ------------------------------------------------
// The exchange of a step is ~60 k floats between a handful of GPUs of ONE node: a ring collective pays 2 (G - 1) hops of
// launch + link latency for it, while every rank can simply READ the other ranks' values -- xGMI is point to point.  Every
// rank owns a publish buffer (hipMalloc + hipIpcGetMemHandle, mapped by the others with hipIpcOpenMemHandle); a launch
// of k_peer_allreduce writes the rank's span into its own buffer as 8-byte {value, tag} words (flag in data: single-copy
// atomic, system scope), then reads the same words of EVERY rank in rank order -- polling a word until its tag is this
// launch's -- and leaves the sum in place.  Same order on every rank: bit-identical replicas.  No barrier, no second
// launch: about one xGMI round trip.  Two slots alternate: a rank can only be ONE launch ahead of the slowest one (it
// cannot finish launch s + 1 before every rank published s + 1, i.e. finished reading s), so slot s & 1 is never
// overwritten while someone still reads it.  The launch sequence number lives in device memory and is advanced by the
// launch's last workgroup -- a hipGraph replay carries no host-side counter.  Bounded polls raise d_state[2].
struct PeerArgs {
  unsigned long long* pub[IGMC_MAX_PEERS];
  int world, rank;
  int64_t cap;
  unsigned long long timeout_ticks;      // bound of a poll on the device's constant-rate clock (100 MHz)
  int* state;
  float* a;
  int64_t na;
  float* b;
  int64_t nb;
};
__global__ __launch_bounds__(IGMC_BLOCK) void k_peer_allreduce(PeerArgs p) {
#ifndef IGMC_HIPEMU
  const uint32_t seq = (uint32_t)__hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  const uint32_t seq = (uint32_t)p.state[0];
#endif
  const unsigned long long tag = (unsigned long long)seq << 32;
  const int64_t n = p.na + p.nb, base = (int64_t)(seq & 1u) * p.cap;
  const int64_t T = (int64_t)gridDim.x * IGMC_BLOCK, t0 = (int64_t)blockIdx.x * IGMC_BLOCK + threadIdx.x;
  unsigned long long* mine = p.pub[p.rank] + base;
  for (int64_t i = t0; i < n; i += T) {
    const float v = (i < p.na) ? p.a[i] : p.b[i - p.na];
    const unsigned long long w = tag | (unsigned long long)__float_as_uint(v);
#ifndef IGMC_HIPEMU
    __hip_atomic_store(mine + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    mine[i] = w;
#endif
  }
  // A rank may be late by far more than a kernel's length -- first-step capture, a checkpoint write on rank 0, a
  // time-sliced device: the polls are bounded by WALL-CLOCK time (p.timeout_ticks, a minute by default), not by an
  // iteration count.  A word that never arrives raises the sticky p.state[2]: the element is left as it was, every thread
  // that sees the flag stops summing, but elements summed BEFORE a thread gave up stay summed -- after a time-out the spans
  // are a mixture of sums and local values and MUST NOT be used (no foreign or torn value is ever written: a word counts
  // only with this launch's tag in it).  The gradient / Adam kernel of igmc_train_step_dp behind this launch reads the
  // flag and leaves parameters, moments and counters alone (AdamTail::skip); the host raises at its next check
  // (igmc_comm_check).  Callers of the bare igmc_allreduce_grads must call igmc_comm_check before they use the spans.
  //
  // CROSS-DEVICE visibility (the words of rank r live in r's HBM, fine-grained, mapped over xGMI): value and flag are ONE
  // 8-byte word written by ONE system-scope atomic store (global_store_dwordx2 sc0 sc1: written through, never parked in the
  // writer's L2) and read by ONE system-scope atomic load (sc0 sc1: never served from the reader's L2) -- single-copy
  // atomic, so a reader sees either the old word (old tag: polls again) or the whole new one.  Nothing else is published
  // through these buffers, so no release / acquire ordering BETWEEN words is needed: every word validates itself.
#ifndef IGMC_HIPEMU
  const unsigned long long t_start = wall_clock64();
#endif
  for (int64_t i = t0; i < n; i += T) {
    float s = 0.f;
    bool ok = true;
    for (int r = 0; r < p.world && ok; ++r) {
      const unsigned long long* src = p.pub[r] + base + i;
      unsigned long long w = 0;
      for (long it = 0;; ++it) {
#ifndef IGMC_HIPEMU
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
        w = *src;
#endif
        if ((w >> 32) == (unsigned long long)seq) break;
#ifndef IGMC_HIPEMU
        if ((it & 63) == 63 && wall_clock64() - t_start > p.timeout_ticks) ok = false;
        if (__hip_atomic_load(p.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ok = false;      // (another thread gave up)
#else
        if (it > (1L << 22)) ok = false;
#endif
        if (!ok) {
          p.state[2] = 1;
          break;
        }
#ifndef IGMC_HIPEMU
        __builtin_amdgcn_s_sleep(4);
#endif
      }
      s += __uint_as_float((uint32_t)w);
    }
    if (ok) {
      if (i < p.na) p.a[i] = s;
      else p.b[i - p.na] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#ifndef IGMC_HIPEMU
    if (__hip_atomic_fetch_add(p.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(p.state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p.state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#else
    if (p.state[1]++ == (int)gridDim.x - 1) {
      p.state[1] = 0;
      p.state[0] += 1;
    }
#endif
  }
}

extern "C" int igmc_comm_peer_alloc(int rank, int world, int device, int64_t max_floats, igmc_comm** out, uint8_t* h_handle64) {
  if (!out || !h_handle64 || world < 1 || world > IGMC_MAX_PEERS || rank < 0 || rank >= world || max_floats < 1) IGMC_FAIL("bad arguments");
  igmc_comm* c = new igmc_comm();
  c->nccl = nullptr;
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->host_fn = nullptr;
  c->host_user = nullptr;
  c->peer = 1;
  c->connected = 0;
  c->cap = (max_floats + 63) & ~(int64_t)63;
  for (int r = 0; r < IGMC_MAX_PEERS; ++r) c->pub[r] = nullptr;
  memset(h_handle64, 0, 64);
#ifndef IGMC_HIPEMU
  HIPCHECK(hipSetDevice(device));
  // FINE-GRAINED device memory where the runtime can share it (coherent between devices while kernels run: what RCCL's
  // own buffers are); else ordinary device memory -- the words are written and read with system-scope accesses either way
  const size_t bytes = (size_t)2 * c->cap * sizeof(unsigned long long);
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the 64 bytes of the C ABI");
  bool fine = hipExtMallocWithFlags((void**)&c->pub[rank], bytes, hipDeviceMallocFinegrained) == hipSuccess;
  if (fine && hipIpcGetMemHandle(&h, c->pub[rank]) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(c->pub[rank]);
    c->pub[rank] = nullptr;
    fine = false;
  }
  if (!fine) {
    (void)hipGetLastError();
    HIPCHECK(hipMalloc((void**)&c->pub[rank], bytes));
    HIPCHECK(hipIpcGetMemHandle(&h, c->pub[rank]));
  }
  c->fine = fine ? 1 : 0;
  HIPCHECK(hipMemset(c->pub[rank], 0, bytes));      // tag 0 is never a launch's
  HIPCHECK(hipMalloc((void**)&c->d_state, 4 * sizeof(int)));
  const int init[4] = {1, 0, 0, 0};
  HIPCHECK(hipMemcpy(c->d_state, init, sizeof(init), hipMemcpyHostToDevice));
  memcpy(h_handle64, &h, sizeof(h));
  HIPCHECK(hipDeviceSynchronize());
#else
  if (world != 1) { delete c; IGMC_FAIL("the emulation build has one rank only"); }
  c->pub[rank] = (unsigned long long*)calloc((size_t)2 * c->cap, sizeof(unsigned long long));
  c->d_state = (int*)calloc(4, sizeof(int));
  c->d_state[0] = 1;
#endif
  *out = c;
  return 0;
}

extern "C" int igmc_comm_peer_connect(igmc_comm* c, const uint8_t* h_handles) {
  if (!c || !c->peer || !h_handles) IGMC_FAIL("not a peer communicator");
#ifndef IGMC_HIPEMU
  HIPCHECK(hipSetDevice(c->device));
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, h_handles + (size_t)r * 64, sizeof(h));
    void* ptr = nullptr;
    HIPCHECK(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    c->pub[r] = (unsigned long long*)ptr;
  }
#endif
  c->connected = 1;
  return 0;
}

// 0 = the communicator's device-side state is sane; synchronises `stream`.  A peer communicator whose bounded poll ran out
// (a rank that never published: not co-resident, crashed, or peer memory that is not visible) reports it here.
extern "C" int igmc_comm_check(igmc_comm* c, void* stream) {
  if (!c) IGMC_FAIL("null communicator");
#ifndef IGMC_HIPEMU
  HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
  if (c->peer && c->d_state) {
    int st[4];
    HIPCHECK(hipMemcpy(st, c->d_state, sizeof(st), hipMemcpyDeviceToHost));
    if (st[2]) {
      // reported once: the flag is cleared so that a caller that recovers (or tears the job down in order) is not told again
      const int zero = 0;
      (void)hipMemcpy(c->d_state + 2, &zero, sizeof(int), hipMemcpyHostToDevice);
      IGMC_FAIL("peer all-reduce: another rank's words did not arrive within the time limit (IGMC_PEER_TIMEOUT_S); the spans of that "
                "launch were left unsummed and the optimiser step behind it was skipped");
    }
  }
#endif
  return 0;
}

// 0 none (one rank), 1 RCCL, 2 host callback, 3 peer-mapped buffers (4: in fine-grained device memory)
extern "C" int igmc_comm_kind(const igmc_comm* c) {
  if (!c) return 0;
  if (c->peer) return c->fine ? 4 : 3;
  if (c->host_fn) return 2;
  return c->nccl ? 1 : 0;
}

static int peer_sum2(igmc_comm* c, float* a, int64_t na, float* b, int64_t nb, void* stream) {
  if (!c->connected) { g_err = "peer communicator is not connected (igmc_comm_peer_connect)"; return 1; }
  if (c->world == 1 && !getenv("IGMC_PEER_ALWAYS")) return 0;      // one rank: the spans ARE the sums (IGMC_PEER_ALWAYS: test hook)
  if (!a) na = 0;
  if (!b) nb = 0;
  double tmo = 60.0;
  if (const char* e = getenv("IGMC_PEER_TIMEOUT_S")) tmo = atof(e) > 0 ? atof(e) : tmo;
  // spans beyond a slot go in pieces (each piece is a launch of its own: the sequence number advances per launch)
  while (na + nb > 0) {
    PeerArgs p;
    memset(&p, 0, sizeof(p));
    for (int r = 0; r < c->world; ++r) p.pub[r] = c->pub[r];
    p.world = c->world; p.rank = c->rank; p.cap = c->cap; p.state = c->d_state;
    p.timeout_ticks = (unsigned long long)(tmo * 1e8);
    const int64_t ta = na < c->cap ? na : c->cap;
    const int64_t tb = (nb < c->cap - ta) ? nb : c->cap - ta;
    p.a = a; p.na = ta; p.b = b; p.nb = tb;
    int grid = (int)((ta + tb + 4 * IGMC_BLOCK - 1) / (4 * IGMC_BLOCK));      // four words a thread; few workgroups: every
    if (grid > 64) grid = 64;                                                  // rank's launch must be on its chip at once
    if (grid < 1) grid = 1;
    IGMC_PLAUNCH("k_peer_allreduce", k_peer_allreduce, grid, IGMC_BLOCK, 0, stream, p);
    a += ta; na -= ta;
    b += tb; nb -= tb;
  }
#ifndef IGMC_HIPEMU
  if (hipGetLastError() != hipSuccess) { g_err = "k_peer_allreduce launch failed"; return 1; }
#endif
  return 0;
}

extern "C" void igmc_comm_destroy(igmc_comm* c) {
  if (!c) return;
#ifndef IGMC_HIPEMU
  if (c->nccl && g_rccl.ok) g_rccl.CommDestroy(c->nccl);
  if (c->peer) {
    for (int r = 0; r < c->world; ++r)
      if (c->pub[r]) {
        if (r == c->rank) hipFree(c->pub[r]);
        else hipIpcCloseMemHandle(c->pub[r]);
      }
    if (c->d_state) hipFree(c->d_state);
  }
#else
  if (c->peer) {
    free(c->pub[c->rank]);
    free(c->d_state);
  }
#endif
  delete c;
}

extern "C" int igmc_comm_info(const igmc_comm* c, int* rank, int* world) {
  if (!c) IGMC_FAIL("null communicator");
  int r = c->rank, w = c->world;
#ifndef IGMC_HIPEMU
  if (c->nccl) {
    RCCLCHECK(g_rccl.CommUserRank(c->nccl, &r));
    RCCLCHECK(g_rccl.CommCount(c->nccl, &w));
  }
#endif
  if (rank) *rank = r;
  if (world) *world = w;
  return 0;
}

// sum of up to two spans over the ranks, in place: ONE grouped collective (RCCL), or the host callback once per span
static int comm_sum2(void* user, float* a, int64_t na, float* b, int64_t nb, void* stream) {
  igmc_comm* c = (igmc_comm*)user;
  if (c->peer) return peer_sum2(c, a, na, b, nb, stream);
  if (c->host_fn) {
    if (a && na > 0 && c->host_fn(c->host_user, a, na, stream)) { g_err = "comm_sum2: the host all-reduce callback failed"; return 1; }
    if (b && nb > 0 && c->host_fn(c->host_user, b, nb, stream)) { g_err = "comm_sum2: the host all-reduce callback failed"; return 1; }
    return 0;
  }
#ifndef IGMC_HIPEMU
  if (!c->nccl) return 0;
  const bool two = a && na > 0 && b && nb > 0;
  if (two) RCCLCHECK(g_rccl.GroupStart());
  if (a && na > 0) RCCLCHECK(g_rccl.AllReduce(a, a, (size_t)na, /*ncclFloat*/ 7, /*ncclSum*/ 0, c->nccl, (hipStream_t)stream));
  if (b && nb > 0) RCCLCHECK(g_rccl.AllReduce(b, b, (size_t)nb, 7, 0, c->nccl, (hipStream_t)stream));
  if (two) RCCLCHECK(g_rccl.GroupEnd());
#endif
  return 0;
}

extern "C" int igmc_allreduce_grads(igmc_comm* c, float* d_flat_grad, int64_t n, float scale, void* stream) {
  if (!c || !d_flat_grad || n <= 0) IGMC_FAIL("bad arguments");
  if (comm_sum2(c, d_flat_grad, n, nullptr, 0, stream)) return 1;
  if (scale != 1.0f) {
    int grid = (int)((n + IGMC_BLOCK - 1) / IGMC_BLOCK);
    IGMC_PLAUNCH("k_scale_flat", k_scale_flat, grid > 1024 ? 1024 : grid, IGMC_BLOCK, 0, stream, d_flat_grad, n, scale);
    HIPCHECK(hipGetLastError());
  }
  return 0;
}

// One optimisation step of a data-parallel job (reference train_eval.py:157-177 per rank, gradients averaged over the
// ranks): igmc_train_step with the exchange INSIDE the step.  Where the step keeps its gradient sources in reduced form
// (igmc_step_exchange_inside: the subgraph kernel's tables, the per-layer path's basis-space sums) those are summed over
// the ranks between their reduction and the gradient / Adam kernel -- the kernels of the single-GPU step, one grouped
// collective more; elsewhere the flat gradient is formed, all-reduced and handed to the Adam kernel.  comm == NULL: one rank.
extern "C" int igmc_train_step_dp(igmc_model* m, igmc_comm* comm, float* d_params, const igmc_batch* b, int use_edge_flags,
                                  const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float multiply_by, float ARR,
                                  float* d_out, float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_loss,
                                  double* d_total, int64_t* d_ctrl, int64_t adam_t, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, void* stream) {
  std::string why;
  if (check_fit(m, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_loss) IGMC_FAIL("null buffer");
  if (!d_ctrl && adam_t < 1) IGMC_FAIL("adam_t must be >= 1");
  float step_size = 0.f, inv = 0.f;
  if (!d_ctrl) {
    const double bc1 = 1.0 - std::pow((double)beta1, (double)adam_t);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)adam_t);
    step_size = (float)((double)lr / bc1);
    inv = (float)(1.0 / std::sqrt(bc2));
  }
  const int world = comm ? comm->world : 1;
  const int B = b->last_B;
  const float gscale = 1.0f / ((float)B * (float)world);
  m->d.side = b->side;
  csr_for_model(m, b, 1, stream);
  ImgScope img(m, d_params);
  int emitted = 0;
  if (igmc_step_exchange_inside(m->d, b->d, B)) {
    StepExchange x = {comm_sum2, comm, (comm && comm->peer) ? comm->d_state + 2 : nullptr};
    // (the ARR term depends on the weights only: every rank adds it in full AFTER the exchange)
    const int rc = igmc_launch_train_step(m->d, m->ax, b->d, d_params, B, use_edge_flags, d_lin_mask, seed, step, multiply_by,
                                          ARR, d_out, d_grad, d_exp_avg, d_exp_avg_sq, step_size, inv, beta1, beta2, eps,
                                          weight_decay, d_ctrl, m->done_ctr, d_loss, d_total, stream, gscale,
                                          comm ? &x : nullptr, &emitted);
    img.done_updated(emitted);
    if (rc) return 1;
  } else {
    igmc_launch_loss_grad(m->d, m->ax, b->d, d_params, B, use_edge_flags, d_lin_mask, seed, step, multiply_by, ARR, gscale,
                          1.0f / (float)world, d_out, d_grad, nullptr, nullptr, stream);
    if (comm && comm_sum2(comm, d_grad, m->d.n_params, nullptr, 0, stream)) return 1;
    igmc_launch_finish(m->d, b->d, d_params, d_grad, d_exp_avg, d_exp_avg_sq, step_size, inv, beta1, beta2, eps,
                       weight_decay, d_ctrl, ARR, d_loss, d_total, use_edge_flags, stream);
    img.done_updated(0);
  }
  HIPCHECK(hipGetLastError());
  m->last_B = B;
  m->last_training = 1;
  m->last_flags = use_edge_flags;
  return 0;
}

// ------------------------------------------------------------------ sort-pool readout family (DGCNN_RS)
struct igmc_sortpool {
  igmc_model* m;
  SpDev d;
  float* pe;        // engine-layout copy of the conv parameters (layer 3 zero-padded to 32 columns)
  float* ge;        // engine-layout gradient scratch
  Allocs mem;
};

extern "C" int igmc_sortpool_create(igmc_model* m, int k, int max_nodes_per_graph, igmc_sortpool** out) {
  if (!m || !out) IGMC_FAIL("null argument");
  if (k < 10) IGMC_FAIL("sort-pool k must be >= 10 (reference models.py:73)");
  if (m->d.S != 0) IGMC_FAIL("the sort-pool readout takes no side features");
  if (max_nodes_per_graph < 2 || max_nodes_per_graph > 4096) IGMC_FAIL("subgraphs of 2..4096 nodes");
  HIPCHECK(hipSetDevice(m->device));
  igmc_sortpool* sp = new igmc_sortpool();
  sp->m = m;
  SpDev& d = sp->d;
  const ModelDev& md = m->d;
  d.k = k;
  d.Q1 = k / 2;                       // int((k - 2) / 2 + 1), reference models.py:83
  d.Q2 = d.Q1 - 5 + 1;
  if (d.Q2 < 1) { delete sp; IGMC_FAIL("k too small for Conv1d(16, 32, 5)"); }
  d.dense = d.Q2 * 32;
  d.nmax = max_nodes_per_graph;
  d.P = 1;
  while (d.P < d.nmax) d.P <<= 1;
  // true layout: layers 0..2 at the engine offsets, then layer 3 with one output column
  int64_t off = md.off_basis[3];
  d.t_basis3 = off; off += 4 * 32;
  d.t_root3 = off;  off += 32;
  d.t_bias3 = off;  off += 1;
  d.t_att3 = off;   off += (int64_t)md.R * 4;
  d.t_conv_end = off;
  d.t_c1w = off; off += 16 * 97 + 16;
  d.t_c2w = off; off += 32 * 16 * 5 + 32;
  d.t_l1w = off; off += (int64_t)128 * d.dense;
  d.t_l1b = off; off += 128;
  d.t_l2w = off; off += 128;
  d.t_l2b = off; off += 1;
  d.n_params = off;
  if (!igmc_sp_lds_ok(d)) { delete sp; IGMC_FAIL("k / subgraph size too large for the LDS plan of the sort-pool kernels"); }
  Allocs& M = sp->mem;
  const size_t Bc = (size_t)md.graph_cap, N = (size_t)md.node_cap;
  int fail = 0;
  fail |= M.get(&d.sel, Bc * k) | M.get(&d.y1, Bc * 16 * k) | M.get(&d.flat, Bc * d.dense) | M.get(&d.dflat, Bc * d.dense) | M.get(&d.lin_part, 8 * Bc) | M.get(&d.lin_ctr, Bc / 16 + 1) | M.get(&d.a1, Bc * 128) |
          M.get(&d.lmask, Bc * 128) | M.get(&d.dz, Bc * 128) | M.get(&d.dout, Bc) |
          M.get(&d.part_c1, Bc * (16 * 97 + 16)) | M.get(&d.part_c2, Bc * (32 * 16 * 5 + 32));
  for (int l = 0; l < 3; ++l) fail |= M.get(&d.dcat[l], N * 32);
  fail |= M.get(&sp->pe, (size_t)md.n_params) | M.get(&sp->ge, (size_t)md.n_params);
  if (fail) {
    M.release();
    delete sp;
    IGMC_FAIL("hipMalloc failed (sort-pool workspace)");
  }
  HIPCHECK(hipMemset(d.lin_ctr, 0, (Bc / 16 + 1) * sizeof(int)));        // arrival counters: zero between launches
  HIPCHECK(hipMemset(sp->pe, 0, (size_t)md.n_params * sizeof(float)));
  HIPCHECK(hipMemset(sp->ge, 0, (size_t)md.n_params * sizeof(float)));
  if (igmc_sp_prepare(d)) { M.release(); delete sp; IGMC_FAIL("hipFuncSetAttribute failed"); }
  *out = sp;
  return 0;
}

extern "C" void igmc_sortpool_destroy(igmc_sortpool* sp) {
  if (!sp) return;
  sp->mem.release();
  delete sp;
}

// [0..16): basis, root, bias, att of conv layers 0..3; then conv1.weight, conv1.bias, conv2.weight, conv2.bias,
// lin1.weight, lin1.bias, lin2.weight, lin2.bias; [24] = number of parameters, [25] = dense width, [26] = k
extern "C" int igmc_sortpool_layout(const igmc_sortpool* sp, int64_t* out27) {
  if (!sp || !out27) IGMC_FAIL("null argument");
  const ModelDev& md = sp->m->d;
  const SpDev& d = sp->d;
  for (int l = 0; l < 3; ++l) {
    out27[4 * l + 0] = md.off_basis[l]; out27[4 * l + 1] = md.off_root[l];
    out27[4 * l + 2] = md.off_bias[l];  out27[4 * l + 3] = md.off_att[l];
  }
  out27[12] = d.t_basis3; out27[13] = d.t_root3; out27[14] = d.t_bias3; out27[15] = d.t_att3;
  out27[16] = d.t_c1w; out27[17] = d.t_c1w + 16 * 97; out27[18] = d.t_c2w; out27[19] = d.t_c2w + 32 * 16 * 5;
  out27[20] = d.t_l1w; out27[21] = d.t_l1b; out27[22] = d.t_l2w; out27[23] = d.t_l2b;
  out27[24] = d.n_params; out27[25] = d.dense; out27[26] = d.k;
  return 0;
}

static int sp_check(igmc_sortpool* sp, const igmc_batch* b, std::string* why) {
  if (!sp) { *why = "null sort-pool workspace"; return 1; }
  if (check_fit(sp->m, b, why)) return 1;
  if (b->d.slot > sp->d.nmax) { *why = "subgraph slots larger than the sort-pool workspace was created for"; return 1; }
  return 0;
}

// reference DGCNN_RS.forward (models.py:142-167) on an extracted batch
extern "C" int igmc_sortpool_forward(igmc_sortpool* sp, const float* d_params, const igmc_batch* b, int training,
                                     int use_edge_flags, const uint8_t* d_lin_mask, uint64_t seed, uint64_t step,
                                     float* d_out, void* stream) {
  std::string why;
  if (sp_check(sp, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out) IGMC_FAIL("null buffer");
  igmc_model* m = sp->m;
  ensure_csr(b, stream);
  igmc_launch_sp_pack(m->d, sp->d, d_params, sp->pe, stream);
  igmc_launch_conv_forward(m->d, b->d, sp->pe, b->last_B, training, use_edge_flags, stream);
  igmc_launch_sp_forward(m->d, sp->d, b->d, d_params, b->last_B, training, d_lin_mask, seed, step, d_out, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}

// forward + MSE (+ ARR over the conv layers, reference train_eval.py:162-174) + backward: d_grad in the true layout
extern "C" int igmc_sortpool_loss_grad(igmc_sortpool* sp, const float* d_params, const igmc_batch* b, int use_edge_flags,
                                       const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float ARR,
                                       float grad_scale, float arr_scale, float* d_out, float* d_grad, float* d_loss,
                                       void* stream) {
  std::string why;
  if (sp_check(sp, b, &why)) IGMC_FAIL(why);
  if (!d_params || !d_out || !d_grad) IGMC_FAIL("null buffer");
  igmc_model* m = sp->m;
  const int B = b->last_B;
  ensure_csr(b, stream);
  igmc_launch_sp_pack(m->d, sp->d, d_params, sp->pe, stream);
  ModelDev md = m->d;        // (with the dense readout gradient: the conv kernels then know which backward form follows)
  for (int l = 0; l < 3; ++l) md.dcat[l] = sp->d.dcat[l];
  igmc_launch_conv_forward(md, b->d, sp->pe, B, 1, use_edge_flags, stream);
  igmc_launch_sp_forward(m->d, sp->d, b->d, d_params, B, 1, d_lin_mask, seed, step, d_out, stream);
  igmc_launch_sp_backward(m->d, sp->d, b->d, d_params, B, grad_scale > 0.f ? grad_scale : 1.f / (float)B, stream);
  igmc_launch_conv_backward(md, b->d, sp->pe, B, use_edge_flags, ARR * arr_scale, sp->ge, stream);
  igmc_launch_sp_wgrad(m->d, sp->d, b->d, B, sp->ge, d_grad, stream);
  if (d_loss) igmc_launch_loss(m->d, b->d, ARR, d_loss, stream);
  HIPCHECK(hipGetLastError());
  m->last_flags = use_edge_flags;
  return 0;
}

extern "C" int igmc_sortpool_step_finish(igmc_sortpool* sp, const igmc_batch* b, float* d_params, const float* d_grad,
                                         float* d_exp_avg, float* d_exp_avg_sq, float ARR, float* d_loss, double* d_total,
                                         int64_t* d_ctrl, int64_t step, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, void* stream) {
  if (!sp || !b || !d_params || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_loss) IGMC_FAIL("bad arguments");
  if (!d_ctrl && step < 1) IGMC_FAIL("step must be >= 1");
  float step_size = 0.f, inv = 0.f;
  if (!d_ctrl) {
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    step_size = (float)((double)lr / bc1);
    inv = (float)(1.0 / std::sqrt(bc2));
  }
  ModelDev md = sp->m->d;
  md.n_params = sp->d.n_params;          // Adam runs over the sort-pool family's own flat buffer
  igmc_launch_finish(md, b->d, d_params, d_grad, d_exp_avg, d_exp_avg_sq, step_size, inv, beta1, beta2, eps, weight_decay,
                     d_ctrl, ARR, d_loss, d_total, sp->m->last_flags, stream);
  HIPCHECK(hipGetLastError());
  return 0;
}
