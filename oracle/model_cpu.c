/* model_cpu.c -- TEST INFRASTRUCTURE: an OpenMP host twin of the IGMC model step (forward, loss, backward, Adam).
 *
 * The second half of SURVEY.md 8(b)'s igmc_cpu_* twins (the first is extract_cpu.c): a second, independent implementation of
 * what igmc_model_forward / igmc_model_loss_grad / igmc_adam_step compute on the GPU, written from the reference's model code
 * (models.py:170-217 IGMC: 4 x (RGCNConv + tanh), concatenated states of the two target nodes, lin1 + ReLU + dropout(0.5) +
 * lin2, * multiply_by; train_eval.py:157-177: mse_loss + ARR * sum_l sum_r ||W_l[r+1] - W_l[r]||^2, torch.optim.Adam) and
 * the published PyG-1.4.2 RGCNConv (basis decomposition, aggr = add, root + bias) -- not from the HIP kernels and not in
 * the oracle's formulation either: where oracle/pyg_ref.py multiplies per edge and scatter-adds, and the kernels run dense
 * blocks through the matrix cores, this one is shaped for a CPU:
 *   - the subgraphs of a batch are independent until the parameter gradient: ONE SUBGRAPH PER OpenMP THREAD at a time,
 *     private gradient accumulators, one ordered reduction at the end (deterministic for a given thread count);
 *   - aggregate-then-transform: per node and relation the neighbour states are summed first (edges bucketed by
 *     (target, relation) with a counting sort), then ONE in x out product per non-empty bucket;
 *   - the backward pass recomputes the bucket sums instead of storing them and walks the transposed buckets for dX.
 * tests/test_model_twin.py holds it to the oracle (outputs, loss, every gradient, Adam) and the HIP path to both; bench.py
 * times extraction twin + this as a second CPU baseline.  Nothing under igmc_amd/ loads it.
 *
 * Flat parameter layout = the C ABI's (include/igmc_hip.h, igmc_param_offset): per conv layer basis [NB][in][out], root
 * [in][out], bias [out], att [R][NB]; then lin1.weight [hid][width], lin1.bias [hid], lin2.weight [hid], lin2.bias [1].
 *
 * Build: gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC -o oracle/_build/libmodel_cpu.so oracle/model_cpu.c -lm
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXL 8

typedef struct {
  int nl, R, NB, hid, width, md;          /* md = widest layer input / output */
  int dims[MAXL + 1];
  int64_t o_basis[MAXL], o_root[MAXL], o_bias[MAXL], o_att[MAXL], o_w1, o_b1, o_w2, o_b2, n_params;
  int64_t o_W[MAXL], n_W;                 /* composed W_l[r] = sum_b att[r][b] basis[b] (and their gradient accumulators) */
} Cfg;

static int cfg_make(Cfg* C, int nl, const int* dims, int R, int NB, int hid) {
  if (nl < 1 || nl > MAXL || R < 1 || NB < 1) return -1;
  memset(C, 0, sizeof(*C));
  C->nl = nl; C->R = R; C->NB = NB; C->hid = hid;
  int64_t off = 0, w = 0;
  for (int l = 0; l <= nl; ++l) {
    C->dims[l] = dims[l];
    if (dims[l] > C->md) C->md = dims[l];
  }
  for (int l = 0; l < nl; ++l) {
    const int64_t io = (int64_t)dims[l] * dims[l + 1];
    C->o_basis[l] = off; off += NB * io;
    C->o_root[l] = off;  off += io;
    C->o_bias[l] = off;  off += dims[l + 1];
    C->o_att[l] = off;   off += (int64_t)R * NB;
    C->o_W[l] = w;       w += R * io;
    C->width += 2 * dims[l + 1];
  }
  C->o_w1 = off; off += (int64_t)hid * C->width;
  C->o_b1 = off; off += hid;
  C->o_w2 = off; off += hid;
  C->o_b2 = off; off += 1;
  C->n_params = off;
  C->n_W = w;
  return 0;
}

int64_t igmc_cpu_param_count(int nl, const int* dims, int R, int NB, int hid) {
  Cfg C;
  return cfg_make(&C, nl, dims, R, NB, hid) ? -1 : C.n_params;
}

int igmc_cpu_model_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

/* W_l[r][i][o] = sum_b att[r][b] basis[b][i][o] (reference train_eval.py:169-170 spells the same product out) */
static void compose(const Cfg* C, const float* P, float* W) {
  for (int l = 0; l < C->nl; ++l) {
    const int64_t io = (int64_t)C->dims[l] * C->dims[l + 1];
    for (int r = 0; r < C->R; ++r) {
      float* w = W + C->o_W[l] + r * io;
      for (int64_t k = 0; k < io; ++k) w[k] = 0.f;
      for (int b = 0; b < C->NB; ++b) {
        const float a = P[C->o_att[l] + r * C->NB + b];
        const float* bs = P + C->o_basis[l] + b * io;
        for (int64_t k = 0; k < io; ++k) w[k] += a * bs[k];
      }
    }
  }
}
/* the transposed copies the backward pass walks (dX = W dPre: rows of W^T are contiguous in the input feature):
 * WT[o_W[l] + r io ..] = W_l[r]^T [out][in], RT[o_W[l] / R ..] = root_l^T */
static void transpose_all(const Cfg* C, const float* P, const float* W, float* WT, float* RT) {
  for (int l = 0; l < C->nl; ++l) {
    const int di = C->dims[l], dn = C->dims[l + 1];
    const int64_t io = (int64_t)di * dn;
    for (int r = 0; r <= C->R; ++r) {
      const float* w = r < C->R ? W + C->o_W[l] + r * io : P + C->o_root[l];
      float* t = r < C->R ? WT + C->o_W[l] + r * io : RT + C->o_W[l] / C->R;
      for (int i = 0; i < di; ++i)
        for (int o = 0; o < dn; ++o) t[(size_t)o * di + i] = w[(size_t)i * dn + o];
    }
  }
}

typedef struct {
  int cap_n, cap_e;
  int32_t *iptr, *inb, *optr, *onb, *fill;   /* buckets (target, relation) -> sources; (source, relation) -> targets */
  float *x0, *h, *g0, *g1;                  /* one-hot input, states of every layer [nl][n][md], two gradient planes */
  float *feat, *z, *a, *dfeat;
} Scratch;

static void scratch_make(Scratch* S, const Cfg* C, int cap_n, int cap_e) {
  const size_t nb = (size_t)cap_n * C->R + 2;
  S->cap_n = cap_n; S->cap_e = cap_e;
  S->iptr = (int32_t*)malloc(sizeof(int32_t) * nb);
  S->optr = (int32_t*)malloc(sizeof(int32_t) * nb);
  S->fill = (int32_t*)malloc(sizeof(int32_t) * nb);
  S->inb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cap_e + 1));
  S->onb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cap_e + 1));
  S->x0 = (float*)malloc(sizeof(float) * (size_t)cap_n * C->md);
  S->h = (float*)malloc(sizeof(float) * (size_t)C->nl * cap_n * C->md);
  S->g0 = (float*)malloc(sizeof(float) * (size_t)cap_n * C->md);
  S->g1 = (float*)malloc(sizeof(float) * (size_t)cap_n * C->md);
  S->feat = (float*)malloc(sizeof(float) * (size_t)C->width);
  S->dfeat = (float*)malloc(sizeof(float) * (size_t)C->width);
  S->z = (float*)malloc(sizeof(float) * (size_t)C->hid);
  S->a = (float*)malloc(sizeof(float) * (size_t)C->hid);
}
static void scratch_free(Scratch* S) {
  free(S->iptr); free(S->optr); free(S->fill); free(S->inb); free(S->onb); free(S->x0); free(S->h); free(S->g0); free(S->g1);
  free(S->feat); free(S->dfeat); free(S->z); free(S->a);
}

/* edges (key -> other end) bucketed by key * R + relation: ptr [n R + 1], nb [e] */
static void bucket(int n, int R, int e, const int32_t* key, const int32_t* other, const uint8_t* rel, int32_t base,
                   int32_t* ptr, int32_t* nb, int32_t* fill) {
  const int nbk = n * R;
  memset(ptr, 0, sizeof(int32_t) * (size_t)(nbk + 1));
  for (int k = 0; k < e; ++k) ++ptr[(key[k] - base) * R + rel[k] + 1];
  for (int k = 0; k < nbk; ++k) ptr[k + 1] += ptr[k];
  memcpy(fill, ptr, sizeof(int32_t) * (size_t)nbk);
  for (int k = 0; k < e; ++k) nb[fill[(key[k] - base) * R + rel[k]]++] = other[k] - base;
}

/* One subgraph: forward; with acc != NULL also the backward pass into the thread's accumulators
 * (acc[0 .. n_params): root / bias / lin slots of the flat layout, acc[n_params ..): dW_l[r]).  Returns 0; -1 when the
 * graph lacks a target node (label 0 = the user, label 1 = the item: reference models.py:205-206), -2 / -3 for a label /
 * an edge outside the subgraph's ranges. */
static int graph_pass(const Cfg* C, const float* P, const float* W, const float* WT, const float* RT, int n, const int32_t* lab, int e, const int32_t* src,
                      const int32_t* dst, const uint8_t* rel, int32_t base, float y, const uint8_t* mask, int training,
                      float mult, float gscale, float* out, float* acc, Scratch* S) {
  const int R = C->R, md = C->md, nl = C->nl;
  int tu = -1, tv = -1;
  for (int v = 0; v < n; ++v) {
    if (lab[v] == 0 && tu < 0) tu = v;
    if (lab[v] == 1 && tv < 0) tv = v;
  }
  if (tu < 0 || tv < 0) return -1;
  for (int v = 0; v < n; ++v)
    if (lab[v] < 0 || lab[v] >= C->dims[0]) return -2;                      /* a label beyond the one-hot width */
  for (int k = 0; k < e; ++k)                                               /* an edge that leaves its subgraph / a relation beyond R */
    if (src[k] < base || src[k] >= base + n || dst[k] < base || dst[k] >= base + n || rel[k] >= R) return -3;
  bucket(n, R, e, dst, src, rel, base, S->iptr, S->inb, S->fill);          /* messages flow source -> target */
  memset(S->x0, 0, sizeof(float) * (size_t)n * md);
  for (int v = 0; v < n; ++v) S->x0[(size_t)v * md + lab[v]] = 1.f;
  float agg[64], pre[64];
  /* ---- forward */
  for (int l = 0; l < nl; ++l) {
    const int di = C->dims[l], dn = C->dims[l + 1];
    const float* X = l ? S->h + (size_t)(l - 1) * n * md : S->x0;
    float* H = S->h + (size_t)l * n * md;
    const float *root = P + C->o_root[l], *bias = P + C->o_bias[l], *Wl = W + C->o_W[l];
    for (int v = 0; v < n; ++v) {
      const float* xv = X + (size_t)v * md;
      for (int o = 0; o < dn; ++o) pre[o] = bias[o];
      for (int i = 0; i < di; ++i) {
        const float xi = xv[i];
        if (xi == 0.f) continue;
        const float* rr = root + (size_t)i * dn;
        for (int o = 0; o < dn; ++o) pre[o] += xi * rr[o];
      }
      for (int r = 0; r < R; ++r) {
        const int p0 = S->iptr[v * R + r], p1 = S->iptr[v * R + r + 1];
        if (p0 == p1) continue;
        for (int i = 0; i < di; ++i) agg[i] = 0.f;
        for (int p = p0; p < p1; ++p) {
          const float* xs = X + (size_t)S->inb[p] * md;
          for (int i = 0; i < di; ++i) agg[i] += xs[i];
        }
        const float* w = Wl + (size_t)r * di * dn;
        for (int i = 0; i < di; ++i) {
          const float ai = agg[i];
          if (ai == 0.f) continue;
          for (int o = 0; o < dn; ++o) pre[o] += ai * w[(size_t)i * dn + o];
        }
      }
      for (int o = 0; o < dn; ++o) H[(size_t)v * md + o] = tanhf(pre[o]);
    }
  }
  /* ---- readout of the two target nodes + MLP (reference models.py:204-216) */
  const int half = C->width / 2, hid = C->hid;
  for (int l = 0, k = 0; l < nl; ++l)
    for (int o = 0; o < C->dims[l + 1]; ++o, ++k) {
      S->feat[k] = S->h[((size_t)l * n + tu) * md + o];
      S->feat[half + k] = S->h[((size_t)l * n + tv) * md + o];
    }
  const float *w1 = P + C->o_w1, *b1 = P + C->o_b1, *w2 = P + C->o_w2;
  float o2 = P[C->o_b2];
  for (int j = 0; j < hid; ++j) {
    float z = b1[j];
    const float* wj = w1 + (size_t)j * C->width;
    for (int k = 0; k < C->width; ++k) z += wj[k] * S->feat[k];
    S->z[j] = z;
    float a = z > 0.f ? z : 0.f;
    if (training) a = mask[j] ? a * 2.f : 0.f;          /* F.dropout(p = 0.5): keep / (1 - p) */
    S->a[j] = a;
    o2 += w2[j] * a;
  }
  *out = o2 * mult;
  if (!acc) return 0;
  /* ---- backward: d loss / d out = gscale * (out - y) (gscale = 2 / B for the batch mean) */
  const float dout = gscale * (*out - y) * mult;
  acc[C->o_b2] += dout;
  for (int k = 0; k < C->width; ++k) S->dfeat[k] = 0.f;
  for (int j = 0; j < hid; ++j) {
    acc[C->o_w2 + j] += dout * S->a[j];
    float dz = dout * w2[j];
    if (training) dz = mask[j] ? dz * 2.f : 0.f;
    if (!(S->z[j] > 0.f) || dz == 0.f) continue;
    acc[C->o_b1 + j] += dz;
    float* gw = acc + C->o_w1 + (size_t)j * C->width;
    const float* wj = w1 + (size_t)j * C->width;
    for (int k = 0; k < C->width; ++k) {
      gw[k] += dz * S->feat[k];
      S->dfeat[k] += dz * wj[k];
    }
  }
  bucket(n, R, e, src, dst, rel, base, S->optr, S->onb, S->fill);          /* transposed: source -> its targets */
  float *G = S->g0, *Gp = S->g1;
  memset(G, 0, sizeof(float) * (size_t)n * md);
  int koff = half;                      /* feature offset of layer l inside a half of the readout */
  for (int l = nl - 1; l >= 0; --l) {
    const int di = C->dims[l], dn = C->dims[l + 1];
    koff -= dn;
    for (int o = 0; o < dn; ++o) {       /* the readout's share of d h_l (tu == tv cannot happen: the labels differ) */
      G[(size_t)tu * md + o] += S->dfeat[koff + o];
      G[(size_t)tv * md + o] += S->dfeat[half + koff + o];
    }
    const float* X = l ? S->h + (size_t)(l - 1) * n * md : S->x0;
    const float* H = S->h + (size_t)l * n * md;
    const float *rootT = RT + C->o_W[l] / C->R, *WlT = WT + C->o_W[l];
    float *groot = acc + C->o_root[l], *gbias = acc + C->o_bias[l], *gW = acc + C->n_params + C->o_W[l];
    for (int v = 0; v < n; ++v) {        /* through tanh, then the parameter gradients of this node */
      float* gv = G + (size_t)v * md;
      const float* hv = H + (size_t)v * md;
      int any = 0;
      for (int o = 0; o < dn; ++o) {
        gv[o] *= 1.f - hv[o] * hv[o];
        any |= gv[o] != 0.f;
      }
      if (!any) continue;
      const float* xv = X + (size_t)v * md;
      for (int o = 0; o < dn; ++o) gbias[o] += gv[o];
      for (int i = 0; i < di; ++i) {
        const float xi = xv[i];
        if (xi == 0.f) continue;
        for (int o = 0; o < dn; ++o) groot[(size_t)i * dn + o] += xi * gv[o];
      }
      for (int r = 0; r < R; ++r) {
        const int p0 = S->iptr[v * R + r], p1 = S->iptr[v * R + r + 1];
        if (p0 == p1) continue;
        for (int i = 0; i < di; ++i) agg[i] = 0.f;
        for (int p = p0; p < p1; ++p) {
          const float* xs = X + (size_t)S->inb[p] * md;
          for (int i = 0; i < di; ++i) agg[i] += xs[i];
        }
        float* gw = gW + (size_t)r * di * dn;
        for (int i = 0; i < di; ++i) {
          const float ai = agg[i];
          if (ai == 0.f) continue;
          for (int o = 0; o < dn; ++o) gw[(size_t)i * dn + o] += ai * gv[o];
        }
      }
    }
    if (l == 0) break;
    for (int v = 0; v < n; ++v) {        /* d h_{l-1}[v] = root dPre[v] + sum_r W_r (sum of dPre over v's targets of relation r) */
      float* gp = Gp + (size_t)v * md;
      const float* gv = G + (size_t)v * md;
      for (int i = 0; i < di; ++i) gp[i] = 0.f;
      for (int o = 0; o < dn; ++o) {
        const float go = gv[o];
        if (go == 0.f) continue;
        const float* rt = rootT + (size_t)o * di;
        for (int i = 0; i < di; ++i) gp[i] += go * rt[i];
      }
      for (int r = 0; r < R; ++r) {
        const int p0 = S->optr[v * R + r], p1 = S->optr[v * R + r + 1];
        if (p0 == p1) continue;
        for (int o = 0; o < dn; ++o) pre[o] = 0.f;
        for (int p = p0; p < p1; ++p) {
          const float* gt = G + (size_t)S->onb[p] * md;
          for (int o = 0; o < dn; ++o) pre[o] += gt[o];
        }
        const float* wt = WlT + (size_t)r * di * dn;
        for (int o = 0; o < dn; ++o) {
          const float po = pre[o];
          if (po == 0.f) continue;
          for (int i = 0; i < di; ++i) gp[i] += po * wt[(size_t)o * di + i];
        }
      }
    }
    float* t = G; G = Gp; Gp = t;
  }
  return 0;
}

/* One batch of B subgraphs (a collated PyG batch: nodes and edges of graph g are node_off[g] .. node_off[g + 1] and
 * edge_off[g] .. edge_off[g + 1]; src / dst are batch-wide node ids; label[v] = the one-hot position of x[v]).
 *   training : 1 = lin_mask [B][hid] (uint8 keep flags of the 0.5 dropout) is applied; 0 = eval
 *   grad     : NULL = forward only; else the flat gradient of  mean_g (out_g - y_g)^2 + ARR * sum_l sum_r ||W_l[r+1] - W_l[r]||^2
 *   loss     : NULL or [2]: that loss, and the sum of squared errors
 * Returns 0; -1 bad configuration; -2 a malformed subgraph (no target pair, a label / edge / relation out of range). */
int igmc_cpu_model_loss_grad(int nl, const int* dims, int R, int NB, int hid, const float* params, int B,
                             const int64_t* node_off, const int32_t* label, const int64_t* edge_off, const int32_t* src,
                             const int32_t* dst, const uint8_t* rel, const float* y, const uint8_t* lin_mask, int training,
                             float multiply_by, float ARR, float* out, float* grad, double* loss) {
  Cfg C;
  if (cfg_make(&C, nl, dims, R, NB, hid) || C.md > 64) return -1;
  float* W = (float*)malloc(sizeof(float) * (size_t)C.n_W * 2 + sizeof(float) * (size_t)(C.n_W / R + 1));
  float *WT = W + C.n_W, *RT = WT + C.n_W;
  compose(&C, params, W);
  transpose_all(&C, params, W, WT, RT);
  int cap_n = 1, cap_e = 1, bad = 0;
  for (int g = 0; g < B; ++g) {
    if (node_off[g + 1] - node_off[g] > cap_n) cap_n = (int)(node_off[g + 1] - node_off[g]);
    if (edge_off[g + 1] - edge_off[g] > cap_e) cap_e = (int)(edge_off[g + 1] - edge_off[g]);
  }
  const int T = omp_get_max_threads();
  const size_t na = (size_t)C.n_params + (size_t)C.n_W;
  float* accs = grad ? (float*)calloc(na * (size_t)T, sizeof(float)) : NULL;
#pragma omp parallel num_threads(T)
  {
    Scratch S;
    scratch_make(&S, &C, cap_n, cap_e);
    float* acc = accs ? accs + na * (size_t)omp_get_thread_num() : NULL;
#pragma omp for schedule(static)
    for (int g = 0; g < B; ++g) {
      const int64_t n0 = node_off[g], e0 = edge_off[g];
      const int rc = graph_pass(&C, params, W, WT, RT, (int)(node_off[g + 1] - n0), label + n0, (int)(edge_off[g + 1] - e0), src + e0,
                                dst + e0, rel + e0, (int32_t)n0, y ? y[g] : 0.f, lin_mask ? lin_mask + (size_t)g * hid : NULL,
                                training && lin_mask, multiply_by, 2.f / (float)B, out + g, acc, &S);
      if (rc) {
#pragma omp atomic write
        bad = 1;
      }
    }
    scratch_free(&S);
  }
  if (bad) { free(W); free(accs); return -2; }
  double sse = 0.0, arr = 0.0;
  if (y)
    for (int g = 0; g < B; ++g) sse += ((double)out[g] - y[g]) * ((double)out[g] - y[g]);
  for (int l = 0; l < C.nl; ++l) {
    const int64_t io = (int64_t)C.dims[l] * C.dims[l + 1];
    for (int r = 0; r + 1 < R; ++r)
      for (int64_t k = 0; k < io; ++k) {
        const float d = W[C.o_W[l] + (r + 1) * io + k] - W[C.o_W[l] + r * io + k];
        arr += (double)d * d;
      }
  }
  if (loss) {
    loss[0] = sse / (B > 0 ? B : 1) + (double)ARR * arr;
    loss[1] = sse;
  }
  if (grad) {
    float* tot = accs;                                   /* thread 0's accumulators take the others, in thread order */
    for (int t = 1; t < T; ++t)
      for (size_t k = 0; k < na; ++k) tot[k] += accs[na * (size_t)t + k];
    memcpy(grad, tot, sizeof(float) * (size_t)C.n_params);
    for (int l = 0; l < C.nl; ++l) {
      const int64_t io = (int64_t)C.dims[l] * C.dims[l + 1];
      float* gW = tot + C.n_params + C.o_W[l];
      const float* Wl = W + C.o_W[l];
      if (ARR != 0.f)                                    /* d/dW_r of ARR * sum ||W[r+1] - W[r]||^2 */
        for (int r = 0; r < R; ++r)
          for (int64_t k = 0; k < io; ++k) {
            float d = 0.f;
            if (r > 0) d += Wl[r * io + k] - Wl[(r - 1) * io + k];
            if (r + 1 < R) d -= Wl[(r + 1) * io + k] - Wl[r * io + k];
            gW[r * io + k] += 2.f * ARR * d;
          }
      for (int b = 0; b < NB; ++b) {                      /* W_r = sum_b att[r][b] basis[b] */
        float* gb = grad + C.o_basis[l] + b * io;
        const float* bs = params + C.o_basis[l] + b * io;
        for (int64_t k = 0; k < io; ++k) gb[k] = 0.f;
        for (int r = 0; r < R; ++r) {
          const float a = params[C.o_att[l] + r * NB + b];
          const float* gw = gW + r * io;
          double dot = 0.0;
          for (int64_t k = 0; k < io; ++k) {
            gb[k] += a * gw[k];
            dot += (double)gw[k] * bs[k];
          }
          grad[C.o_att[l] + r * NB + b] = (float)dot;
        }
      }
    }
  }
  free(W);
  free(accs);
  return 0;
}

/* torch.optim.Adam (reference train_eval.py:54, 177): weight decay added to the gradient, bias corrections of step t >= 1 */
int igmc_cpu_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t t, float lr,
                       float beta1, float beta2, float eps, float weight_decay) {
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < n; ++k) {
    const float g = grad[k] + weight_decay * params[k];
    const float m = beta1 * exp_avg[k] + (1.f - beta1) * g;
    const float v = beta2 * exp_avg_sq[k] + (1.f - beta2) * g * g;
    exp_avg[k] = m;
    exp_avg_sq[k] = v;
    params[k] -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
  }
  return 0;
}

/* The extraction twin's raw output (extract_cpu.c: per link the users / items with their labels and the (u_local, v_local,
 * relation) triples) as the collated batch the model takes: nodes of a graph = its users then its items (reference
 * util_functions.py:280-297 construct_pyg_graph), every rating as two directed edges.  node_off / edge_off: [B + 1];
 * label, src, dst, rel sized by the caller (sum (n_u + n_v), 2 sum n_e).  Returns 0, or -1 when a link was not extracted. */
int igmc_cpu_collate(int B, int cap_u, int cap_v, int64_t edges_per_link, const int32_t* n_u, const int32_t* n_v,
                     const uint8_t* ulab, const uint8_t* vlab, const int64_t* n_e, const int32_t* edges, int64_t* node_off,
                     int64_t* edge_off, int32_t* label, int32_t* src, int32_t* dst, uint8_t* rel) {
  node_off[0] = 0;
  edge_off[0] = 0;
  for (int g = 0; g < B; ++g) {
    if (n_e[g] < 0) return -1;
    node_off[g + 1] = node_off[g] + n_u[g] + n_v[g];
    edge_off[g + 1] = edge_off[g] + 2 * n_e[g];
  }
#pragma omp parallel for schedule(static)
  for (int g = 0; g < B; ++g) {
    const int64_t n0 = node_off[g], e0 = edge_off[g];
    for (int i = 0; i < n_u[g]; ++i) label[n0 + i] = ulab[(size_t)g * cap_u + i];
    for (int j = 0; j < n_v[g]; ++j) label[n0 + n_u[g] + j] = vlab[(size_t)g * cap_v + j];
    const int32_t* ed = edges + 3 * (size_t)g * edges_per_link;
    for (int64_t k = 0; k < n_e[g]; ++k) {
      const int32_t u = (int32_t)(n0 + ed[3 * k]), v = (int32_t)(n0 + n_u[g] + ed[3 * k + 1]);
      src[e0 + 2 * k] = u; dst[e0 + 2 * k] = v; rel[e0 + 2 * k] = (uint8_t)ed[3 * k + 2];
      src[e0 + 2 * k + 1] = v; dst[e0 + 2 * k + 1] = u; rel[e0 + 2 * k + 1] = (uint8_t)ed[3 * k + 2];
    }
  }
  return 0;
}
