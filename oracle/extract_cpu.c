/* extract_cpu.c -- TEST INFRASTRUCTURE: an OpenMP host twin of the engine's free-running subgraph extraction.
 *
 * A second, independent implementation of what igmc_extract_batch produces on the GPU (the igmc_cpu_* twins):
 * the reference's algorithm (util_functions.py:208-277 subgraph_extraction_labeling, :300-304 neighbors) with the engine's
 * stateless per-hop sampling (include/igmc_rng.h: the k candidates with the smallest igmc_sample_key -- CPython's
 * random.sample is not reproducible across devices) and the engine's node order (targets first, the rest in
 * ascending id order).  Written from the reference and igmc_rng.h, not from the HIP kernels: no bitmaps in LDS, no radix
 * select, no dense block -- sorted candidate lists, a sort by key, a merge of CSR rows.  The tests hold the HIP extraction
 * (emulator and GPU) to it bit for bit; nothing under igmc_amd/ loads it.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -I include -o oracle/_build/libextract_cpu.so oracle/extract_cpu.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "igmc_rng.h"

typedef struct {
  int n_users, n_items;
  const int64_t* u_ptr;   /* CSR users -> items: [n_users + 1] */
  const int32_t* u_idx;   /* item ids, ascending within a row */
  const uint8_t* u_rel;   /* relation of the rating */
  const int64_t* v_ptr;   /* CSC items -> users */
  const int32_t* v_idx;
} Graph;

static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}
typedef struct { uint32_t key; int32_t id; } Keyed;
static int cmp_keyed(const void* a, const void* b) {
  const uint32_t x = ((const Keyed*)a)->key, y = ((const Keyed*)b)->key;
  return (x > y) - (x < y);
}

/* sorted union of the neighbour lists of `fringe`, minus `visited` (a byte map, updated; what it gains is appended to
 * `touched`), -> out; returns the count */
static int expand(const int32_t* fringe, int nf, const int64_t* ptr, const int32_t* idx, uint8_t* visited, int32_t* out,
                  int32_t* touched, int* nt) {
  int n = 0;
  for (int f = 0; f < nf; ++f)
    for (int64_t e = ptr[fringe[f]]; e < ptr[fringe[f] + 1]; ++e) {
      const int32_t j = idx[e];
      if (!visited[j]) {        /* (visited is updated before sampling: reference :220-221) */
        visited[j] = 1;
        out[n++] = j;
        touched[(*nt)++] = j;     /* (to clear the map after the link) */
      }
    }
  qsort(out, (size_t)n, sizeof(int32_t), cmp_i32);
  return n;
}

/* the k candidates with the smallest sample key, left in ascending id order; returns k */
static int sample(int32_t* cand, int n, int k, uint64_t salt, Keyed* tmp) {
  if (k >= n) return n;
  for (int i = 0; i < n; ++i) {
    tmp[i].key = igmc_sample_key(salt, (uint32_t)cand[i]);
    tmp[i].id = cand[i];
  }
  qsort(tmp, (size_t)n, sizeof(Keyed), cmp_keyed);      /* (the key is a bijection of the id: no ties) */
  for (int i = 0; i < k; ++i) cand[i] = tmp[i].id;
  qsort(cand, (size_t)k, sizeof(int32_t), cmp_i32);
  return k;
}

/* per-thread scratch, allocated once per call */
typedef struct {
  uint8_t *vis_u, *vis_v, *du, *dv;     /* visited maps (all zero between links), hop distance of the discovered nodes */
  int32_t *cu, *cv, *lu, *lv, *loc_v;   /* candidates, discovered nodes, item id -> local index (all -1 between links) */
  int32_t *tu, *tv;                     /* ids set in the visited maps */
  Keyed* tmp;
} Scratch;
static void scratch_make(Scratch* S, const Graph* G) {
  const int nmax = G->n_users > G->n_items ? G->n_users : G->n_items;
  S->vis_u = (uint8_t*)calloc((size_t)G->n_users, 1);
  S->vis_v = (uint8_t*)calloc((size_t)G->n_items, 1);
  S->cu = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_users + 1));
  S->cv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_items + 1));
  S->lu = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_users + 1));
  S->lv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_items + 1));
  S->du = (uint8_t*)malloc((size_t)(G->n_users + 1));
  S->dv = (uint8_t*)malloc((size_t)(G->n_items + 1));
  S->tu = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_users + 1));
  S->tv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(G->n_items + 1));
  S->loc_v = (int32_t*)malloc(sizeof(int32_t) * (size_t)G->n_items);
  memset(S->loc_v, 0xFF, sizeof(int32_t) * (size_t)G->n_items);
  S->tmp = (Keyed*)malloc(sizeof(Keyed) * (size_t)(nmax + 1));
}
static void scratch_free(Scratch* S) {
  free(S->vis_u); free(S->vis_v); free(S->cu); free(S->cv); free(S->lu); free(S->lv); free(S->du); free(S->dv);
  free(S->loc_v); free(S->tmp); free(S->tu); free(S->tv);
}

/* One link.  users / items: [cap] ids (target first, then ascending), ulab / vlab: labels 2 d / 2 d + 1, edges: (u_local,
 * v_local, relation) triples of the induced subgraph without the target pair, user-major; returns the edge count or -1 when
 * a side outgrows cap / the edge buffer. */
static int64_t extract_one(const Graph* G, int u0, int v0, int hop, double sample_ratio, int mnph, uint64_t seed,
                           uint64_t epoch, uint64_t link_uid, int cap_u, int cap_v, int32_t* users, int32_t* items,
                           uint8_t* ulab, uint8_t* vlab, int* n_u, int* n_v, int32_t* edges, int64_t edge_cap, Scratch* S) {
  uint8_t *vis_u = S->vis_u, *vis_v = S->vis_v, *du = S->du, *dv = S->dv;
  int32_t *cu = S->cu, *cv = S->cv, *lu = S->lu, *lv = S->lv, *loc_v = S->loc_v;   /* lu / lv: discovery order */
  Keyed* tmp = S->tmp;
  int ntu = 0, ntv = 0;
  int nu = 1, nv = 1, fu0 = 0, fu1 = 1, fv0 = 0, fv1 = 1;
  int64_t ne = -1;
  lu[0] = u0; du[0] = 0; vis_u[u0] = 1;
  lv[0] = v0; dv[0] = 0; vis_v[v0] = 1;
  for (int dist = 1; dist <= hop; ++dist) {
    /* simultaneous swap (reference :217): the users' rows give item candidates and vice versa */
    const int cnt_v = expand(lu + fu0, fu1 - fu0, G->u_ptr, G->u_idx, vis_v, cv, S->tv, &ntv);
    const int cnt_u = expand(lv + fv0, fv1 - fv0, G->v_ptr, G->v_idx, vis_u, cu, S->tu, &ntu);
    int ku = cnt_u, kv = cnt_v;
    if (sample_ratio < 1.0) {      /* int(sample_ratio * len), reference :222-224 */
      ku = (int)(sample_ratio * (double)cnt_u);
      kv = (int)(sample_ratio * (double)cnt_v);
    }
    if (mnph >= 0) {               /* reference :225-229 */
      if (mnph < ku) ku = mnph;
      if (mnph < kv) kv = mnph;
    }
    ku = sample(cu, cnt_u, ku, igmc_sample_salt(seed, epoch, link_uid, (uint32_t)dist, 0), tmp);
    kv = sample(cv, cnt_v, kv, igmc_sample_salt(seed, epoch, link_uid, (uint32_t)dist, 1), tmp);
    if (ku == 0 && kv == 0) break; /* reference :230-231 */
    for (int i = 0; i < ku; ++i) { lu[nu + i] = cu[i]; du[nu + i] = (uint8_t)dist; }
    for (int i = 0; i < kv; ++i) { lv[nv + i] = cv[i]; dv[nv + i] = (uint8_t)dist; }
    fu0 = nu; fu1 = nu + ku; nu = fu1;
    fv0 = nv; fv1 = nv + kv; nv = fv1;
  }
  if (nu <= cap_u && nv <= cap_v) {
    /* final order: target first, the others in ascending id order (labels follow their node) */
    for (int side = 0; side < 2; ++side) {
      const int n = side ? nv : nu;
      int32_t* l = side ? lv : lu;
      uint8_t* d = side ? dv : du;
      int32_t* outn = side ? items : users;
      uint8_t* outl = side ? vlab : ulab;
      for (int i = 0; i < n; ++i) { tmp[i].key = (uint32_t)l[i]; tmp[i].id = d[i]; }
      qsort(tmp + 1, (size_t)(n - 1), sizeof(Keyed), cmp_keyed);
      for (int i = 0; i < n; ++i) {
        outn[i] = (int32_t)tmp[i].key;
        outl[i] = (uint8_t)(2 * tmp[i].id + side);
      }
    }
    /* induced edges: every rating (u, v) with both ends selected, except the target pair (reference :233-237) */
    for (int j = 0; j < nv; ++j) loc_v[items[j]] = j;
    ne = 0;
    for (int i = 0; i < nu && ne >= 0; ++i)
      for (int64_t e = G->u_ptr[users[i]]; e < G->u_ptr[users[i] + 1]; ++e) {
        const int j = loc_v[G->u_idx[e]];
        if (j < 0 || (i == 0 && j == 0)) continue;
        if (ne >= edge_cap) { ne = -1; break; }
        edges[3 * ne + 0] = i;
        edges[3 * ne + 1] = j;
        edges[3 * ne + 2] = G->u_rel[e];
        ++ne;
      }
    for (int j = 0; j < nv; ++j) loc_v[items[j]] = -1;
  }
  vis_u[u0] = 0;
  vis_v[v0] = 0;
  for (int i = 0; i < ntu; ++i) vis_u[S->tu[i]] = 0;
  for (int i = 0; i < ntv; ++i) vis_v[S->tv[i]] = 0;
  *n_u = nu;
  *n_v = nv;
  return ne;
}

/* threads of the next calls (0: OpenMP's default); returns the count in force */
int igmc_cpu_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

/* B links of the permutation positions first .. first + B - 1 (link_idx == NULL: the positions themselves), dynamically
 * scheduled over the OpenMP threads.  Per link g: n_u[g], n_v[g], users[g * cap_u ..], items[g * cap_v ..] (+ labels), edges[edge_off[g] ..] with
 * edge_off[g] = g * edges_per_link, n_e[g] = its count (-1: a capacity was too small).  Returns 0. */
int igmc_cpu_extract_batch(int n_users, int n_items, const int64_t* u_ptr, const int32_t* u_idx, const uint8_t* u_rel,
                           const int64_t* v_ptr, const int32_t* v_idx, const int32_t* link_u, const int32_t* link_v,
                           const int64_t* link_idx, int64_t first, int B, int hop, double sample_ratio, int max_nodes_per_hop,
                           uint64_t seed, uint64_t epoch, int cap_u, int cap_v, int64_t edges_per_link, int32_t* n_u,
                           int32_t* n_v, int32_t* users, int32_t* items, uint8_t* ulab, uint8_t* vlab, int64_t* n_e,
                           int32_t* edges) {
  Graph G = {n_users, n_items, u_ptr, u_idx, u_rel, v_ptr, v_idx};
#pragma omp parallel
  {
    Scratch S;
    scratch_make(&S, &G);
#pragma omp for schedule(dynamic, 4)
    for (int g = 0; g < B; ++g) {
      const int64_t pos = link_idx ? link_idx[first + g] : first + g;
      int nu = 0, nv = 0;
      n_e[g] = extract_one(&G, link_u[pos], link_v[pos], hop, sample_ratio, max_nodes_per_hop, seed, epoch,
                           (uint64_t)(uint32_t)pos, cap_u, cap_v, users + (size_t)g * cap_u, items + (size_t)g * cap_v,
                           ulab + (size_t)g * cap_u, vlab + (size_t)g * cap_v, &nu, &nv, edges + 3 * (size_t)g * edges_per_link,
                           edges_per_link, &S);
      n_u[g] = nu;
      n_v[g] = nv;
    }
    scratch_free(&S);
  }
  return 0;
}
