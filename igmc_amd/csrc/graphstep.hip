// graphstep.hip -- forward + loss + backward of ONE enclosing subgraph inside ONE workgroup (gfx950 / CDNA4).
//
// Message passing in IGMC never crosses enclosing subgraphs (reference models.py:190-217 runs the R-GCN stack on
// a block-diagonal batch), and with the per-hop cap of the headline configurations a subgraph has at most
// ~200 nodes: its [N,32] feature matrix is 26 KB.  MI355X has 160 KB of LDS per CU, so ONE 256-thread workgroup
// (one wave per SIMD, up to 512 VGPRs per lane) keeps the gather source of every layer in LDS and walks the
// whole stack for its subgraph, with workgroup barriers instead of kernel boundaries (each boundary costs
// >= 4.7 us on this part, and the per-layer kernels were latency-bound at 25-30 us for ~1 MB of traffic):
//
//   layer 0      : h0 = tanh([code histogram | onehot(label) | 1] @ T0)            (f32 MFMA, K = 32)
//   layers 1..3  : RELATION-space aggregate  T_r[i] = sum_{e in row i, rel r} x[src_e]  (rows are sorted by
//                  relation: 8 adds per edge and lane, a run is stored when it ends), then
//                  h = tanh([T_0..T_R-1 | x] @ [W_0; ..; W_R-1; root] + bias),  W_r = sum_b att[r,b] basis_b
//   head         : centre-node readout -> lin1 / ReLU / dropout / lin2 -> residual
//   backward     : the same walk on the transposed keep flags with dPre as the gather source,
//                  dX = [T' | dPre] @ [W_r^T ; root^T], tanh' epilogue, and the weight gradient as the table
//                  h_{l-1}^T [T' | dPre] = [dW_0..dW_R-1 | d root]  -- d basis_b = sum_r att[r,b] dW_r and
//                  d att[r,b] = <basis_b, dW_r> are formed from it by k_finalize (same code as the layer-0 table),
//                  so no per-edge or per-run d att work exists at all.
//
// Work unit = a BUNDLE of 16 rows of similar degree (rows are ranked by degree per subgraph); a wave gathers
// its bundle with one QUAD per row (lane j owns features 8j..8j+7: one ds_read_b128 pair per edge), multiplies
// the 16-row tile on MFMA and runs the epilogue without any workgroup barrier; bundles are assigned to the 4
// waves longest-first by a static schedule computed from the degrees (=> bit-reproducible).
// dPre / T never exist in HBM; h_l is written once (the backward re-reads 16-row chunks of it).
//
// Eligibility (else the per-layer kernels of model.hip run): R <= 5, layer-0 table <= 32 rows, no side
// features, slot (= max nodes of a subgraph) small enough for the LDS plan below (~300 nodes).
#include "launch.h"
#include <stdlib.h>
#include <stdio.h>

#define GS_THREADS 256      // 4 waves = one per SIMD: up to 512 VGPRs per lane, every weight fragment stays in registers
#define GS_NW 4
#define GS_NR 5             // relations the relation-space tile is built for (R <= GS_NR; missing ones are zero)
#define GS_KT (GS_NR * 32)  // tile width
#define GS_TP (GS_KT + 4)   // pitch of a wave's 16-row tile (conflict-light MFMA A-operand reads)
#define GS_KS (GS_NR * 8 + 8)   // MFMA k-steps of [T | x] @ [W_r ; root]
#define GS_WN (GS_NR * 2 + 2)   // 16-column tiles of the weight-gradient table
#define GS_HP 36            // pitch of a wave's 16-row h_{l-1} chunk
#define GS_SMAX 20          // max bundles (16 rows) per subgraph: nmax <= 320
#define GS_CMAX 4           // max workgroups per subgraph (cluster)
#define GS_WMAX (GS_NW * GS_CMAX)   // waves of a cluster
#define GS_INVALID 0xFFFFFFFFu
// keeps per-lane index arithmetic INSIDE the loop it is used in (LLVM otherwise hoists hundreds of loop-invariant
// addresses out of the layer / bundle loops and spills them)
#ifdef IGMC_HIPEMU
#define GS_OPAQUE(x) do { } while (0)
#else
#define GS_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

// phase clocks of workgroup 0 (debug aid, see igmc_debug_gs_clocks)
__device__ unsigned long long g_gs_clk[64];
#ifdef IGMC_HIPEMU
#define GS_STAMP(k) do { } while (0)
#define GS_WSTAMP(k) do { } while (0)
#define GS_CSTAMP(k) do { } while (0)
#else
#define GS_STAMP(k) do { if (a.timing && blockIdx.x == 0 && threadIdx.x == 0) g_gs_clk[k] = __builtin_readcyclecounter(); } while (0)
#define GS_WSTAMP(k) do { if (a.timing && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_gs_clk[(k) + (threadIdx.x >> 6)] = __builtin_readcyclecounter(); } while (0)
// member m of the cluster of subgraph 0, event k (0 start, 1 setup done, 2 before / 3 after the first barrier)
#define GS_CSTAMP(k) do { if (a.timing && a.cs > 1 && blockIdx.x % a.stride == 0 && threadIdx.x == 0) g_gs_clk[40 + (blockIdx.x / a.stride) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#endif

// tanh(x) = 1 - 2 / (1 + e^{2x}) on the transcendental unit (v_exp_f32 + v_rcp_f32): |error| ~1e-7 absolute
__device__ __forceinline__ float gs_tanh(float x) {
#ifdef IGMC_HIPEMU
  return tanhf(x);
#else
  return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x));
#endif
}

// entry Q of a 4-entry group held one-per-lane by a quad: DPP quad_perm broadcast (VALU, no LDS crossbar)
template <int Q>
__device__ __forceinline__ uint32_t gs_qbcast(uint32_t v) {
#ifdef IGMC_HIPEMU
  return __shfl(v, Q, 4);
#else
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, Q * 0x55, 0xf, 0xf, false);     // quad_perm:[Q,Q,Q,Q]
#endif
}

// one CSR entry of the batch (source node | relation), GS_INVALID beyond the run or when its keep bit is clear
template <bool FLAGS, bool TRANS>
__device__ __forceinline__ uint32_t gs_entry(const BatchDev& b, int e, int end) {
  uint32_t w = GS_INVALID;
  if (e < end && (!FLAGS || ((b.eflag[e] >> (TRANS ? 1 : 0)) & 1))) w = b.ecr[e];
  return w;
}
// BYTE offset of an entry's gather row from the gather source (padding / dropped entries: the zero row), computed
// ONCE by the lane that holds the entry; the quad then broadcasts the finished offset (1 DPP move + 1 add per lane
// and entry instead of mask / subtract / shift / compare / select in every lane for every entry)
__device__ __forceinline__ int gs_entoff(uint32_t wk, int zoffb, int nb) {
  return (wk != GS_INVALID) ? ((int)(wk & 0xFFFFFFu) - nb) * 128 : zoffb;
}
template <int Q>
__device__ __forceinline__ const float* gs_qrow(const char* srcj, int off) {
#ifdef IGMC_HIPEMU
  return (const float*)(srcj + __shfl(off, Q, 4));
#else
  return (const float*)(srcj + __builtin_amdgcn_mov_dpp(off, Q * 0x55, 0xf, 0xf, true));   // quad_perm:[Q,Q,Q,Q]
#endif
}

// Relation-space aggregate of one row by one QUAD (lane j owns features 8j..8j+7), gather source in LDS, result
// rows T_r written straight into the wave's tile row `trow`.  The relation loop is UNIFORM over the wave (every
// quad is in the same relation at the same time, each on its own run [rptr[r], rptr[r+1]) of its row), so there
// is no run-change branch: per edge one DPP broadcast, one address select, two ds_read_b128 and 8 adds; the
// row loads of the next 4-entry group are in flight while the current one is summed.
// (Measured alternatives, all slower on this part with one wave per SIMD: run-change branches in a row-order
// stream, 16-entry super-chunks with position masks, one OCTET per row with a single ds_read_b128 per edge --
// the loop is bound by per-slot address arithmetic and dependent-issue latency, not by LDS bandwidth.)
// One bundle = 16 rows x R relation runs = 16 R work units (any quad may fill any (row, relation) block of the
// tile).  The units are ranked by run length once per subgraph; round `it` gives quad q the unit of rank
// 16 it + q, so the 16 quads of a wave -- which advance in lockstep -- always work on runs of similar length and
// the ~100-entry target rows are spread over R quads instead of serialising one.
template <bool FLAGS, bool TRANS>
__device__ __forceinline__ void gs_gather(const BatchDev& b, const float* src, const float* zrow, int nb, float* tile,
                                          const int* relp, const int* order, const unsigned char* ulist, int b0, int N,
                                          int R, int qd, int j) {
  // unit of round `it` of this quad: tile slot, relation, run [beg, end) (empty beyond the subgraph's last row)
  auto unit = [&](int it, int& slot, int& r, int& beg, int& end) {
    const int u = ulist[qd + 16 * it];             // units of the bundle by decreasing run length: a round of 16
    slot = u >> 3;                                 // quads works on runs of similar length
    r = u & 7;
    beg = 0;
    end = 0;
    if (b0 + slot < N) {
      const int* rptr = relp + order[b0 + slot] * 8;
      beg = rptr[r];
      end = rptr[r + 1];
    }
  };
  const char* srcj = (const char*)(src + 8 * j);
  const int zoffb = (int)((const char*)zrow - (const char*)src);
  int slot, r, beg, end;
  unit(0, slot, r, beg, end);
  // 4 groups of 4 entries per iteration; the entries of group q of the NEXT iteration are requested right after
  // group q is consumed, i.e. a whole iteration before they are needed (the index loads come from L2 / HBM); the
  // first 16 entries of the NEXT round's run are requested at the start of the round: only round 0 waits for a
  // global load with nothing else to do
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w[q] = gs_entry<FLAGS, TRANS>(b, beg + 4 * q + j, end);
#pragma unroll 1
  for (int it = 0; it < R; ++it) {
    int slot2 = 0, r2 = 0, beg2 = 0, end2 = 0;
    uint32_t wn[4] = {GS_INVALID, GS_INVALID, GS_INVALID, GS_INVALID};
    if (it + 1 < R) {
      unit(it + 1, slot2, r2, beg2, end2);
#pragma unroll
      for (int q = 0; q < 4; ++q) wn[q] = gs_entry<FLAGS, TRANS>(b, beg2 + 4 * q + j, end2);
    }
    float tx[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) tx[f] = 0.f;
    if (beg < end) {
#pragma unroll 1
      for (int c0 = beg; c0 < end; c0 += 16) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {              // two 8-entry halves: up to 16 row loads in flight per wait
          if (c0 + 8 * h >= end) break;
          const int oa = gs_entoff(w[2 * h], zoffb, nb), ob = gs_entoff(w[2 * h + 1], zoffb, nb);
          const bool two = c0 + 8 * h + 4 < end;   // second group of the half present (quad-uniform)
          const float* p0 = gs_qrow<0>(srcj, oa);
          const float* p1 = gs_qrow<1>(srcj, oa);
          const float* p2 = gs_qrow<2>(srcj, oa);
          const float* p3 = gs_qrow<3>(srcj, oa);
          const float4 a0 = *(const float4*)p0, b0v = *(const float4*)(p0 + 4);
          const float4 a1 = *(const float4*)p1, b1v = *(const float4*)(p1 + 4);
          const float4 a2 = *(const float4*)p2, b2v = *(const float4*)(p2 + 4);
          const float4 a3 = *(const float4*)p3, b3v = *(const float4*)(p3 + 4);
          float4 a4, a5, a6, a7, b4v, b5v, b6v, b7v;      // only touched when the second group exists
          if (two) {
            const float* p4 = gs_qrow<0>(srcj, ob);
            const float* p5 = gs_qrow<1>(srcj, ob);
            const float* p6 = gs_qrow<2>(srcj, ob);
            const float* p7 = gs_qrow<3>(srcj, ob);
            a4 = *(const float4*)p4; b4v = *(const float4*)(p4 + 4);
            a5 = *(const float4*)p5; b5v = *(const float4*)(p5 + 4);
            a6 = *(const float4*)p6; b6v = *(const float4*)(p6 + 4);
            a7 = *(const float4*)p7; b7v = *(const float4*)(p7 + 4);
          }
          w[2 * h] = gs_entry<FLAGS, TRANS>(b, c0 + 16 + 8 * h + j, end);          // next iteration's entries
          w[2 * h + 1] = gs_entry<FLAGS, TRANS>(b, c0 + 16 + 8 * h + 4 + j, end);
          float t[8];
          t[0] = (a0.x + a1.x) + (a2.x + a3.x);
          t[1] = (a0.y + a1.y) + (a2.y + a3.y);
          t[2] = (a0.z + a1.z) + (a2.z + a3.z);
          t[3] = (a0.w + a1.w) + (a2.w + a3.w);
          t[4] = (b0v.x + b1v.x) + (b2v.x + b3v.x);
          t[5] = (b0v.y + b1v.y) + (b2v.y + b3v.y);
          t[6] = (b0v.z + b1v.z) + (b2v.z + b3v.z);
          t[7] = (b0v.w + b1v.w) + (b2v.w + b3v.w);
          if (two) {
            t[0] += (a4.x + a5.x) + (a6.x + a7.x);
            t[1] += (a4.y + a5.y) + (a6.y + a7.y);
            t[2] += (a4.z + a5.z) + (a6.z + a7.z);
            t[3] += (a4.w + a5.w) + (a6.w + a7.w);
            t[4] += (b4v.x + b5v.x) + (b6v.x + b7v.x);
            t[5] += (b4v.y + b5v.y) + (b6v.y + b7v.y);
            t[6] += (b4v.z + b5v.z) + (b6v.z + b7v.z);
            t[7] += (b4v.w + b5v.w) + (b6v.w + b7v.w);
          }
#pragma unroll
          for (int f = 0; f < 8; ++f) tx[f] += t[f];
        }
      }
    }
    // rows past the end of the subgraph get zeros (they are K entries of the weight-gradient product)
    *(float4*)(tile + slot * GS_TP + r * 32 + 8 * j) = make_float4(tx[0], tx[1], tx[2], tx[3]);
    *(float4*)(tile + slot * GS_TP + r * 32 + 8 * j + 4) = make_float4(tx[4], tx[5], tx[6], tx[7]);
    slot = slot2;
    r = r2;
    beg = beg2;
    end = end2;
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = wn[q];
  }
}

// ---- cluster of workgroups working on ONE subgraph --------------------------------------------------------
// With B = 50 subgraphs and one workgroup each, 206 of the 256 CUs idle.  `cs` workgroups (same XCD: block index
// = subgraph + stride * member, stride a multiple of 8) therefore share a subgraph: the bundles of every layer are
// scheduled over all their waves, each member publishes the rows it produced and every member loads the whole
// [N][32] matrix back into its LDS.
// The exchange is flag-in-data: every float travels as one 8-byte word {value, tag}, stored and loaded with
// agent-scope (sc1) relaxed atomics -- 8-byte accesses are single-copy atomic, so a word whose tag is the tag of
// THIS exchange carries this exchange's value, and the readers simply poll the data: one store -> load hand-off
// instead of  stores -> wait for their completion -> arrival counter -> poll the counter -> loads  (~7 us per
// exchange with the counter barrier, about half of it round trips that carry no data).  tag = 8 * launch sequence
// number + exchange index + 1: the sequence number lives in HBM and is advanced by the workgroup that finishes the
// launch LAST (by then every workgroup has read it); buffers start zeroed and 0 is never a tag.  Polls are BOUNDED:
// a missing member -- which a grid of <= 224 workgroups of one-per-CU size excludes on this part -- raises gs_err
// instead of hanging the GPU.
// (sc1 accesses meet in the coherent level without any cache-wide maintenance; agent-scope release / acquire fences
// instead write back and invalidate the whole L2 of the XCD and slowed the whole kernel 2-3x.)
__device__ __forceinline__ void gs_pub(float* p, float v) {
#ifndef IGMC_HIPEMU
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}
__device__ __forceinline__ float gs_sub(const float* p) {
#ifndef IGMC_HIPEMU
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
__device__ __forceinline__ void gs_ll_pub(unsigned long long* p, float v, uint32_t tag) {
#ifndef IGMC_HIPEMU
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
#else
  uint32_t bits;
  memcpy(&bits, &v, 4);
  *p = ((unsigned long long)tag << 32) | (unsigned long long)bits;
#endif
}
// All N x 32 words of an exchange into dst[N][32]: 16-byte sc1 loads (two words each), every pending request of a
// thread in flight before the single wait; words with another tag are requested again.  N <= 320 rows = 5120 pairs =
// at most 20 per thread.
__device__ __forceinline__ void gs_ll_reload(float* dst, const unsigned long long* gsrc, int N, uint32_t tag, int* err) {
#ifndef IGMC_HIPEMU
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* gp = (const u32x4*)gsrc;
  float2* d2 = (float2*)dst;
  const int n2 = N * 16, t0 = (int)threadIdx.x;
  uint32_t pend = 0;
#pragma unroll
  for (int u = 0; u < 20; ++u) pend |= (t0 + u * GS_THREADS < n2) ? (1u << u) : 0u;
  u32x4 v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0, v8 = 0, v9 = 0;
  u32x4 v10 = 0, v11 = 0, v12 = 0, v13 = 0, v14 = 0, v15 = 0, v16 = 0, v17 = 0, v18 = 0, v19 = 0;
#define GS_LD(V, U)                                                                            \
  if (pend & (1u << (U))) {                                                                    \
    const u32x4* p = gp + t0 + (U) * GS_THREADS;                                               \
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(V) : "v"(p) : "memory");         \
  }
#define GS_CK(V, U)                                                                            \
  if ((pend & (1u << (U))) && V.y == tag && V.w == tag) {                                      \
    d2[t0 + (U) * GS_THREADS] = make_float2(__uint_as_float(V.x), __uint_as_float(V.z));       \
    pend &= ~(1u << (U));                                                                      \
  }
  for (int it = 0;; ++it) {
    GS_LD(v0, 0) GS_LD(v1, 1) GS_LD(v2, 2) GS_LD(v3, 3) GS_LD(v4, 4)
    GS_LD(v5, 5) GS_LD(v6, 6) GS_LD(v7, 7) GS_LD(v8, 8) GS_LD(v9, 9)
    GS_LD(v10, 10) GS_LD(v11, 11) GS_LD(v12, 12) GS_LD(v13, 13) GS_LD(v14, 14)
    GS_LD(v15, 15) GS_LD(v16, 16) GS_LD(v17, 17) GS_LD(v18, 18) GS_LD(v19, 19)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9),
                   "+v"(v10), "+v"(v11), "+v"(v12), "+v"(v13), "+v"(v14), "+v"(v15), "+v"(v16), "+v"(v17), "+v"(v18),
                   "+v"(v19)
                 :
                 : "memory");
    GS_CK(v0, 0) GS_CK(v1, 1) GS_CK(v2, 2) GS_CK(v3, 3) GS_CK(v4, 4)
    GS_CK(v5, 5) GS_CK(v6, 6) GS_CK(v7, 7) GS_CK(v8, 8) GS_CK(v9, 9)
    GS_CK(v10, 10) GS_CK(v11, 11) GS_CK(v12, 12) GS_CK(v13, 13) GS_CK(v14, 14)
    GS_CK(v15, 15) GS_CK(v16, 16) GS_CK(v17, 17) GS_CK(v18, 18) GS_CK(v19, 19)
    if (!pend) break;
    if (it > (1 << 20)) {
      *err = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
#undef GS_LD
#undef GS_CK
#else
  // emulation (co-resident workgroups, see hipemu::Runtime::co_cs): poll word by word, yielding to the other
  // work-items -- of this and of the other resident workgroups -- while the tag is not this exchange's
  for (int k = (int)threadIdx.x; k < N * 32; k += GS_THREADS) {
    long spins = 0;
    for (;;) {
      const unsigned long long w = gsrc[k];
      if ((uint32_t)(w >> 32) == tag) {
        const uint32_t bits = (uint32_t)w;
        memcpy(&dst[k], &bits, 4);
        break;
      }
      if (++spins > (1L << 22)) {
        *err = 1;
        break;
      }
      hipemu::yield();
    }
  }
#endif
}

// [T | x](16 x 192) @ sW2 (192 x 32) of one bundle: GS_KS k-steps in groups of GS_KG; the operands of group g + 1 are
// requested before the MFMAs of group g are issued (left to itself the scheduler issues every LDS read right before
// the MFMA pair that needs it: ~52 cycles per MFMA instead of 32)
#define GS_KG 8
__device__ __forceinline__ void gs_transform(const float* T, const float* src, int rowA, const float2* sW2, int li,
                                             int kq, f32x4 (&acc)[2][4]) {
  static_assert(GS_KS % GS_KG == 0 && (GS_NR * 8) % GS_KG == 0, "k-step groups");
  float av[2][GS_KG];
  float2 bv[2][GS_KG];
  auto request = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < GS_KG; ++i) {
      const int s = g * GS_KG + i;
      av[buf][i] = (s < GS_NR * 8) ? T[li * GS_TP + 4 * s + kq] : src[rowA * 32 + 4 * (s - GS_NR * 8) + kq];
      bv[buf][i] = sW2[(4 * s + kq) * 16 + li];
    }
  };
  request(0, 0);
#pragma unroll
  for (int g = 0; g < GS_KS / GS_KG; ++g) {
    if (g + 1 < GS_KS / GS_KG) request(g + 1, (g + 1) & 1);
#ifndef IGMC_HIPEMU
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < GS_KG; ++i) {
      const int s = g * GS_KG + i;
      acc[0][s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][i], bv[g & 1][i].x, acc[0][s & 3], 0, 0, 0);
      acc[1][s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][i], bv[g & 1][i].y, acc[1][s & 3], 0, 0, 0);
    }
#ifndef IGMC_HIPEMU
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
#define GS_TRANSFORM(T, src, rowA, sW2, li, kq, acc) gs_transform(T, src, rowA, sW2, li, kq, acc)

template <bool FLAGS, bool TRAIN>
__global__ __launch_bounds__(GS_THREADS) void k_graph_step(BatchDev b, ModelDev m, const float* P, GsArgs a) {
  IGMC_DYN_SMEM(smem);
  float* S = (float*)smem;
  const GsLayout lay = a.lay;
  float* XA = S + lay.xa;                 // [nmax][32]  gather source / destination (ping-pong)
  float* XB = S + lay.xb;
  float* zrow = S + lay.zrow;             // [32] zeros: gather target of padding / dropped entries
  float* TILES = S + lay.tile;            // [4 waves][16][GS_TP] relation-space rows of the wave's bundle; between
                                          // layers: W_r staging [GS_NR*32][32], weight-gradient reduction buffer
  float* HSS = S + lay.hs;                // [4 waves][16][GS_HP]  h_{l-1} rows of the wave's bundle (backward)
  float* s_att = S + lay.att;             // [R][4]
  float* sT0 = S + lay.t0;                // [32][32]  layer-0 table: W0[r*L+c] | root0[c] | bias0 | 0
  int* cnt = (int*)(S + lay.cnt);         // [nmax][rlp]  kept in-edges of node i with code c: two 16-bit counters per word
  int* rp = (int*)(S + lay.rp);           // [nmax+1]  CSR row starts (global edge positions)
  int* slab = (int*)(S + lay.lab);        // [nmax]    node labels
  int* sdeg = (int*)(S + lay.deg);        // [nmax]    row lengths
  int* order = (int*)(S + lay.order);     // [nmax]    rows by decreasing degree: a bundle holds rows of similar length
  int* relp = (int*)(S + lay.relp);       // [nmax][8]  start of every relation run of a row (global edge positions), [R] = row end
  float2* sW2 = (float2*)(S + lay.wreg);  // [GS_KT + 32][16] B operand of the layer: element (k, n) of [W_r ; root] (backward:
                                          // its transpose) at [k][n & 15].{x: n < 16, y: n >= 16}
  int* sched = (int*)(S + lay.sched);     // [2 dirs][GS_WMAX waves][GS_SMAX] bundle lists, then [2][GS_WMAX] list lengths
  unsigned char* ulist = (unsigned char*)(S + lay.ulist);   // [GS_SMAX bundles][16 * GS_NR] units (slot << 3 | rel) by length
  float* sfeat = S + lay.head;            // [256] centre-node readout
  float* sgf = sfeat + 256;               // [256] d feat
  float* sa1 = sgf + 256;                 // [128]
  float* skeep = sa1 + 128;               // [128]
  float* sdz = skeep + 128;               // [128]
  float* sred = sdz + 128;                // [256]
  float* misc = sred + 256;               // [16]
  const int R = m.R, L = m.L, RL = R * L, LF = L * 32, na = R * 4, rlp = lay.rlp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qd = lane >> 2, j = lane & 3;           // quad of the wave (= row of its bundle), lane within it
  const int li = lane & 15, kq = lane >> 4;         // MFMA fragment coordinates
  float* T = TILES + wave * 16 * GS_TP;
  float* HS = HSS + wave * 16 * GS_HP;
  const int B = b.totals[3];
  const int ts = m.ts_stride;
  const int cs = a.cs;                               // workgroups per subgraph
  const int cm = (cs > 1) ? blockIdx.x / a.stride : 0;   // this workgroup's member index
  const int gw = cm * GS_NW + wave, nwt = cs * GS_NW; // wave of the cluster, waves of the cluster
  // cluster exchanges: tag of exchange x = tag0 + x (x = 0..2: h_1..h_3, 3..4: dPre_2, dPre_1)
#ifndef IGMC_HIPEMU
  const uint32_t tag0 = (cs > 1) ? (uint32_t)__hip_atomic_load(m.gs_bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8u + 1u : 0u;
#else
  const uint32_t tag0 = (cs > 1) ? (uint32_t)m.gs_bar[1] * 8u + 1u : 0u;
#endif
  const uint64_t step = m.ctrl ? (uint64_t)m.ctrl[IGMC_CTRL_STEP] : a.step;
#ifndef IGMC_HIPEMU
  // launch clock (profiling mode 2 only): duration of THIS launch = last workgroup's end - earliest workgroup's start,
  // on the constant-rate wall clock, measured where hipGraph replay leaves no room for events between kernels
  if (a.ts && tid == 0) atomicMin(a.ts, (unsigned long long)wall_clock64());
#endif
  GS_STAMP(0);
  GS_CSTAMP(0);

  // ---- the first subgraph's row starts / labels are requested BEFORE the layer-0 table is staged: the two
  //      dependent round trips (node_off -> row_ptr) overlap with the table arithmetic
  const int g_first = (cs > 1) ? blockIdx.x % a.stride : blockIdx.x;
  int pre_nb = 0, pre_N = 0, pre_rp0 = 0, pre_rp1 = 0, pre_lab = 0;
  if (g_first < B) {
    pre_nb = b.node_off[g_first];
    pre_N = b.node_off[g_first + 1] - pre_nb;
    if (tid <= pre_N) pre_rp0 = b.row_ptr[pre_nb + tid];
    if (tid + GS_THREADS <= pre_N) pre_rp1 = b.row_ptr[pre_nb + tid + GS_THREADS];
    if (tid < pre_N) pre_lab = b.node_label[pre_nb + tid];
  }
  // ---- layer-0 table, staged once per workgroup
  for (int i = tid; i < 1024; i += GS_THREADS) {
    const int c = i >> 5, f = i & 31;
    float s = 0.f;
    if (c < RL) {
      const int r = c / L, cf = (c % L) * 32 + f;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) s += P[m.off_att[0] + r * 4 + bb] * P[m.off_basis[0] + bb * LF + cf];
    } else if (c < RL + L) {
      s = P[m.off_root[0] + (c - RL) * 32 + f];
    } else if (c == RL + L) {
      s = P[m.off_bias[0] + f];
    }
    sT0[i] = s;
  }
  if (tid < 32) zrow[tid] = 0.f;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};        // layer-0 table gradient tile (code half, feature half) of this wave
  bool first_graph = true;
  GS_STAMP(56);

#pragma unroll 1
  for (int g = g_first; g < B; g += (cs > 1) ? B : gridDim.x) {
    const int nb = first_graph ? pre_nb : b.node_off[g];
    const int N = first_graph ? pre_N : b.node_off[g + 1] - nb;
    const int cu = b.n_users[g];
    const int nbun = (N + 15) >> 4;
    if (first_graph && N < 2 * GS_THREADS) {
      if (tid <= N) rp[tid] = pre_rp0;
      if (tid + GS_THREADS <= N) rp[tid + GS_THREADS] = pre_rp1;
      if (tid < N) slab[tid] = pre_lab;
      for (int i = tid + GS_THREADS; i < N; i += GS_THREADS) slab[i] = b.node_label[nb + i];
    } else {
      for (int i = tid; i <= N; i += GS_THREADS) rp[i] = b.row_ptr[nb + i];
      for (int i = tid; i < N; i += GS_THREADS) slab[i] = b.node_label[nb + i];
    }
    for (int i = tid; i < N * rlp; i += GS_THREADS) cnt[i] = 0;
    for (int i = tid; i < N * 8; i += GS_THREADS) relp[i] = 0;
    __syncthreads();
    GS_STAMP(57);
    for (int i = tid; i < ((N + 3) & ~3); i += GS_THREADS) sdeg[i] = (i < N) ? rp[i + 1] - rp[i] : -1;
    // layer-0 histogram: a flat pass over the subgraph's contiguous CSR range (edst = destination row)
    {
      const int e0 = rp[0], e1 = rp[N];
      for (int eb = e0 + tid; eb < e1; eb += 24 * GS_THREADS) {    // 24 entries (72 loads) in flight per thread:
        int code[24], row[24], rel[24], keep[24];                   // one round trip for a typical subgraph
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          const int e = eb + u * GS_THREADS;
          const bool in = e < e1;
          const int es = in ? e : e0;
          code[u] = b.ecode[es];                               // relation * L + label of the source node
          row[u] = in ? (int)b.edst[es] : -1;
          if (FLAGS) rel[u] = (int)(b.ecr[es] >> 24);
          keep[u] = FLAGS ? (b.eflag[es] & 1) : 1;
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          if (row[u] < 0) continue;
          // run lengths count every entry, dropped or not; without edge dropout they are sums of the histogram (below)
          if (FLAGS) atomicAdd(&relp[row[u] * 8 + rel[u] + 1], 1);
          if (keep[u]) atomicAdd(&cnt[row[u] * rlp + (code[u] >> 1)], 1 << ((code[u] & 1) * 16));
        }
      }
    }
    __syncthreads();
    if (!FLAGS) {
      for (int i = tid; i < N * R; i += GS_THREADS) {
        const int row_ = i / R, r = i - row_ * R;
        int n = 0;
        for (int c = r * L; c < r * L + L; ++c) n += (cnt[row_ * rlp + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFF;
        relp[row_ * 8 + r + 1] = n;
      }
      __syncthreads();
    }
    GS_STAMP(58);
    // run lengths -> run starts; rows by decreasing degree (rank counting, ties by index) unless every bundle gets a
    // wave of its own anyway (clustered launch: the bundle order then does not matter, natural order is kept)
    const bool ranked = nbun > nwt;
    for (int i = tid; i < N; i += GS_THREADS) {
      int acc = rp[i];
      for (int r = 0; r <= GS_NR; ++r) {
        acc += relp[i * 8 + r];
        relp[i * 8 + r] = acc;
      }
      const int di = sdeg[i];
      int rank = ranked ? 0 : i;
#pragma unroll 4
      for (int q = 0; ranked && q < N; q += 4) {
        const int4 d4 = *(const int4*)(sdeg + q);
        rank += (d4.x > di) || (d4.x == di && q < i);
        rank += (d4.y > di) || (d4.y == di && q + 1 < i);
        rank += (d4.z > di) || (d4.z == di && q + 2 < i);
        rank += (d4.w > di) || (d4.w == di && q + 3 < i);
      }
      order[rank] = i;
    }
    __syncthreads();
    GS_STAMP(59);
    // static longest-first schedule of the bundles over the 4 waves (depends only on the data => reproducible);
    // cost of a bundle = its longest row (edge steps) + the dense work that follows it
    if (!ranked) {
      if (tid < 2 * GS_WMAX) {          // bundle w -> wave w
        sched[tid * GS_SMAX] = tid & (GS_WMAX - 1);
        sched[2 * GS_WMAX * GS_SMAX + tid] = ((tid & (GS_WMAX - 1)) < nbun) ? 1 : 0;
      }
    } else if (tid < 2) {
      // (everything in registers with static indices: a dynamically indexed private array lives in scratch memory
      //  and made this loop the longest phase of the set-up)
      const int cdense = tid ? 100 : 50;
      int load[GS_WMAX], cntw[GS_WMAX];
#pragma unroll
      for (int w = 0; w < GS_WMAX; ++w) { load[w] = (w < nwt) ? 0 : 0x3fffffff; cntw[w] = 0; }
      int* sl = sched + tid * GS_WMAX * GS_SMAX;
      for (int k = 0; k < nbun; ++k) {
        const int cost = sdeg[order[k * 16]] + cdense;
        int best = 0, bl = load[0];
#pragma unroll
        for (int w = 1; w < GS_WMAX; ++w)
          if (load[w] < bl) { bl = load[w]; best = w; }
        int pos = 0;
#pragma unroll
        for (int w = 0; w < GS_WMAX; ++w)
          if (w == best) { pos = cntw[w]; cntw[w] += 1; load[w] += cost; }
        sl[best * GS_SMAX + pos] = k;
      }
#pragma unroll
      for (int w = 0; w < GS_WMAX; ++w) sched[2 * GS_WMAX * GS_SMAX + tid * GS_WMAX + w] = cntw[w];
    }
    __syncthreads();
    GS_STAMP(60);
    // unit lists of the bundles this workgroup will process (either direction), one wave per bundle
    {
      const int nun = 16 * R;
      for (int d = 0; d < 2; ++d) {
        const int ns = sched[2 * GS_WMAX * GS_SMAX + d * GS_WMAX + gw];
        for (int si = 0; si < ns; ++si) {
          const int bun = sched[(d * GS_WMAX + gw) * GS_SMAX + si];
          if (d == 1) {                                   // already done for the forward list of this very wave?
            bool dup = false;
            const int nf = sched[2 * GS_WMAX * GS_SMAX + gw];
            for (int q = 0; q < nf; ++q) dup |= sched[gw * GS_SMAX + q] == bun;
            if (dup) continue;
          }
          const int b0 = bun * 16;
          // run lengths of this lane's (up to two) units, bucketed at 31; a unit's rank = units in longer buckets +
          // earlier units of its own bucket (ballot prefix): deterministic, no LDS traffic
          int kb[2], un[2];
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            const int u = lane + 64 * p2;
            un[p2] = u;
            kb[p2] = -1;
            if (u < nun) {
              const int slot = u / R, r = u - slot * R;
              int len = 0;
              if (b0 + slot < N) {
                const int* rptr = relp + order[b0 + slot] * 8;
                len = rptr[r + 1] - rptr[r];
              }
              kb[p2] = len < 31 ? len : 31;
            }
          }
          int base = 0, rank0 = 0, rank1 = 0;
          const unsigned long long below = (1ull << lane) - 1ull;
          for (int bk = 31; bk >= 0; --bk) {
            const unsigned long long m0 = __ballot(kb[0] == bk), m1 = __ballot(kb[1] == bk);
            const int c0 = __popcll(m0), c1 = __popcll(m1);
            if (kb[0] == bk) rank0 = base + __popcll(m0 & below);
            if (kb[1] == bk) rank1 = base + c0 + __popcll(m1 & below);
            base += c0 + c1;
          }
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            const int u = un[p2];
            if (u < nun) {
              const int slot = u / R, r = u - slot * R;
              ulist[bun * 16 * GS_NR + (p2 ? rank1 : rank0)] = (unsigned char)((slot << 3) | r);
            }
          }
          IGMC_WAVE_SYNC();
        }
      }
    }
    __syncthreads();
    GS_STAMP(1);
    GS_CSTAMP(1);

    // weights of the NEXT conv layer to be staged: requested a phase ahead (before layer 0 / before the cluster
    // barrier of the previous layer), so that their round trip overlaps with compute and with the barrier
    float4 wb4[4], wr4;
    float watt = 0.f;
    auto wpre = [&](int l) {
      const float* basis = P + m.off_basis[l];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) wb4[bb] = *(const float4*)(basis + bb * 1024 + 4 * tid);
      wr4 = *(const float4*)(P + m.off_root[l] + 4 * tid);
      if (tid < na) watt = P[m.off_att[l] + tid];
    };
    wpre(1);
    // ================================================================ layer 0: h0 = tanh([cnt | onehot(label) | 1] @ T0)
    {
      float t0f[2][8];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 8; ++s) t0f[nt][s] = sT0[(4 * s + kq) * 32 + nt * 16 + li];
      // (clustered launches: EVERY member computes all of h_0 -- 13 small MFMA bundles -- which is cheaper than a
      //  cluster barrier + exchange; all members store identical values)
      for (int bun = wave; bun < nbun; bun += GS_NW) {
        const int b0 = bun * 16;
        const int prow = (b0 + li < N) ? b0 + li : N - 1;
        const int row = order[prow];
        const int lab = slab[row];
        f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int code = 4 * s + kq;
          const int cv = (cnt[row * rlp + ((code < RL) ? (code >> 1) : 0)] >> ((code & 1) * 16)) & 0xFFFF;
          const float av = (code < RL) ? (float)cv : ((code == RL + lab || code == RL + L) ? 1.f : 0.f);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t0f[0][s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t0f[1][s], c1, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int p2 = b0 + kq * 4 + rr;
          if (p2 < N) {
            const int orow = order[p2];
            const float v0 = gs_tanh(c0[rr]), v1 = gs_tanh(c1[rr]);
            XA[orow * 32 + li] = v0;
            XA[orow * 32 + 16 + li] = v1;
            if (TRAIN) {
              gs_pub(m.h[0] + (size_t)(nb + orow) * 32 + li, v0);
              gs_pub(m.h[0] + (size_t)(nb + orow) * 32 + 16 + li, v1);
            }
          }
        }
      }
    }
    __syncthreads();
    if (tid < 64) sfeat[(tid >> 5) * 128 + (tid & 31)] = XA[((tid >> 5) ? cu : 0) * 32 + (tid & 31)];
    GS_STAMP(2);

    // ================================================================ conv layers 1..3, forward
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
      const float* src = (l & 1) ? XA : XB;
      float* dst = (l & 1) ? XB : XA;
      // B operand of the layer, staged once: [W_0; ..; W_R-1; 0..; root], W_r = sum_b att[r,b] basis_b
      {
        if (tid < na) s_att[tid] = watt;
        const float4 (&b4)[4] = wb4;
        const float4 r4 = wr4;
        __syncthreads();
        float* sW = (float*)sW2;
        const int f = tid >> 3, n0 = (4 * tid) & 31;           // 4 consecutive outputs n0..n0+3 of input feature f
#pragma unroll
        for (int r = 0; r <= GS_NR; ++r) {
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (r == GS_NR) {
            v[0] = r4.x; v[1] = r4.y; v[2] = r4.z; v[3] = r4.w;
          } else if (r < R) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              const float at = s_att[r * 4 + bb];
              v[0] += at * b4[bb].x; v[1] += at * b4[bb].y; v[2] += at * b4[bb].z; v[3] += at * b4[bb].w;
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) sW[((r * 32 + f) * 16 + ((n0 + q) & 15)) * 2 + ((n0 + q) >> 4)] = v[q];
        }
      }
      const float bias0 = P[m.off_bias[l] + li], bias1 = P[m.off_bias[l] + 16 + li];
      __syncthreads();
      const int ns = sched[2 * GS_WMAX * GS_SMAX + gw];
#pragma unroll 1
      for (int si = 0; si < ns; ++si) {
        const int b0 = sched[gw * GS_SMAX + si] * 16;
        int lane_ = lane;
        GS_OPAQUE(lane_);
        const int qd_ = lane_ >> 2, j_ = lane_ & 3, li_ = lane_ & 15, kq_ = lane_ >> 4;
        if (l == 1 && si == 0) GS_STAMP(16);
        if (R < GS_NR) {
#pragma unroll
          for (int r = 0; r < GS_NR; ++r) {
            *(float4*)(T + qd_ * GS_TP + r * 32 + 8 * j_) = make_float4(0.f, 0.f, 0.f, 0.f);
            *(float4*)(T + qd_ * GS_TP + r * 32 + 8 * j_ + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          IGMC_WAVE_SYNC();
        }
        gs_gather<FLAGS, false>(b, src, zrow, nb, T, relp, order, ulist + (b0 >> 4) * 16 * GS_NR, b0, N, R, qd_, j_);
        IGMC_WAVE_SYNC();
        if (l == 1 && si == 0) GS_STAMP(17);
        const int rowA = order[(b0 + li_ < N) ? b0 + li_ : N - 1];
        f32x4 acc[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        GS_TRANSFORM(T, src, rowA, sW2, li_, kq_, acc);
        if (l == 1 && si == 0) GS_STAMP(18);
        // (the four output rows of a lane are consecutive bundle positions: one 16-byte read of their row numbers,
        //  entries beyond N are never used)
        const int4 orow4 = *(const int4*)(order + b0 + kq_ * 4);
        const int orows[4] = {orow4.x, orow4.y, orow4.z, orow4.w};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int p2 = b0 + kq_ * 4 + rr;
          if (p2 < N) {
            const int orow = orows[rr];
            const float v0 = gs_tanh((acc[0][0][rr] + acc[0][1][rr]) + (acc[0][2][rr] + acc[0][3][rr]) + bias0);
            const float v1 = gs_tanh((acc[1][0][rr] + acc[1][1][rr]) + (acc[1][2][rr] + acc[1][3][rr]) + bias1);
            dst[orow * 32 + li_] = v0;
            dst[orow * 32 + 16 + li_] = v1;
            if (cs > 1) {
              unsigned long long* xp = m.gs_ll + (size_t)(l - 1) * m.gs_ll_stride + (size_t)(nb + orow) * 32;
              gs_ll_pub(xp + li_, v0, tag0 + (l - 1));
              gs_ll_pub(xp + 16 + li_, v1, tag0 + (l - 1));
            }
            if (TRAIN) {                 // backward reads h_l row by row
              gs_pub(m.h[l] + (size_t)(nb + orow) * 32 + li_, v0);
              gs_pub(m.h[l] + (size_t)(nb + orow) * 32 + 16 + li_, v1);
            }
          }
        }
        IGMC_WAVE_SYNC();
        if (l == 1 && si == 0) GS_STAMP(19);
      }
      if (l == 1) GS_STAMP(20);
      if (l == 1) GS_WSTAMP(32);
      if (l < 3) wpre(l + 1);
      if (l == 1) GS_STAMP(22);
      if (l == 1) GS_CSTAMP(2);
      if (cs > 1) {      // every member needs all of h_l
        gs_ll_reload(dst, m.gs_ll + (size_t)(l - 1) * m.gs_ll_stride + (size_t)nb * 32, N, tag0 + (l - 1), m.gs_err);
      }
      if (l == 1) GS_CSTAMP(3);
      __syncthreads();
      if (tid < 64) sfeat[(tid >> 5) * 128 + l * 32 + (tid & 31)] = dst[((tid >> 5) ? cu : 0) * 32 + (tid & 31)];
      GS_STAMP(2 + l);
    }
    __syncthreads();

    // ================================================================ head: lin1 / ReLU / dropout / lin2 / residual
    {
      const int ju = tid >> 1, part = tid & 1;         // hidden unit, half of the fan-in
      const float* wrow = P + m.off_l1w + (int64_t)ju * 256 + part * 128;
      float4 w4[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) w4[q] = *(const float4*)(wrow + 4 * q);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const float4 f4 = *(const float4*)(sfeat + part * 128 + 4 * q);
        s += w4[q].x * f4.x + w4[q].y * f4.y + w4[q].z * f4.z + w4[q].w * f4.w;
      }
      s += __shfl_xor(s, 1, 4);
      if (part == 0) {
        float av = s + P[m.off_l1b + ju];
        av = av > 0.f ? av : 0.f;
        int keep = 1;
        if (TRAIN) {
          keep = a.inj_mask ? (int)a.inj_mask[g * 128 + ju]
                            : (int)(igmc_u01(igmc_unit_hash(a.seed, step, (uint32_t)g, (uint32_t)ju)) >= 0.5f);
          if (cm == 0) {
            m.a1[g * 128 + ju] = av;
            m.lmask[g * 128 + ju] = (uint8_t)keep;
          }
          sa1[ju] = av;
          skeep[ju] = keep ? 1.f : 0.f;
        }
        // F.dropout(p=0.5) in training: kept units scaled by 1/(1-p)
        sred[ju] = (TRAIN ? (keep ? av * 2.f : 0.f) : av) * P[m.off_l2w + ju];
      }
    }
    __syncthreads();
    if (wave == 0) {
      float s = sred[lane] + sred[lane + 64];
      s = igmc_wave_sum_f(s);
      if (lane == 0) {
        const float o = (s + P[m.off_l2b]) * a.mult;
        const float e = o - b.y[g];
        if (cm == 0) {
          a.out[g] = o;
          m.err[g] = e;
        }
        misc[0] = e;
      }
    }
    GS_STAMP(6);
    if (!TRAIN) {
      __syncthreads();
      continue;
    }
    if (TRAIN) {
      __syncthreads();
      // ---- d z, d feat = dz @ lin1.weight
      if (tid < 128) {
        const float dp = 2.f * misc[0] * a.grad_scale * a.mult;
        const float dzv = (sa1[tid] > 0.f && skeep[tid] != 0.f) ? dp * P[m.off_l2w + tid] * 2.f : 0.f;
        sdz[tid] = dzv;
        if (cm == 0) m.dz[g * 128 + tid] = dzv;
      }
      if (cm == 0) m.feat[(size_t)g * m.D + tid] = sfeat[tid];
      __syncthreads();
      {   // wave w takes hidden units 32w..32w+31, lane -> 4 fan-in columns; rows with dz == 0 (ReLU / dropout:
          // ~3/4 of them) are skipped wave-uniformly
        const float* w1 = P + m.off_l1w + 4 * lane;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long nz = __ballot(lane < 32 && sdz[32 * wave + (lane & 31)] != 0.f);   // this wave's live units
        while (nz) {                                   // wave-uniform: 8 weight rows in flight per round
          int q[8];
          float4 wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            q[u] = nz ? (int)__builtin_ctzll(nz) : -1;
            if (nz) nz &= nz - 1;
            wv[u] = (q[u] >= 0) ? *(const float4*)(w1 + (int64_t)(32 * wave + q[u]) * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float dzv = (q[u] >= 0) ? sdz[32 * wave + q[u]] : 0.f;
            s4.x += dzv * wv[u].x; s4.y += dzv * wv[u].y; s4.z += dzv * wv[u].z; s4.w += dzv * wv[u].w;
          }
        }
        *(float4*)(TILES + wave * 256 + 4 * lane) = s4;
      }
      __syncthreads();
      {
        const float v = (TILES[tid] + TILES[256 + tid]) + (TILES[512 + tid] + TILES[768 + tid]);
        sgf[tid] = v;
        if (cm == 0) m.gfeat[(size_t)g * m.D + tid] = v;
      }
      // dPre_3: only the two centre rows are non-zero
      for (int i = tid; i < N * 32; i += GS_THREADS) XA[i] = 0.f;
      __syncthreads();
      if (tid < 64) {
        const int side = tid >> 5, f = tid & 31;
        const float hv = sfeat[side * 128 + 96 + f];
        XA[(side ? cu : 0) * 32 + f] = sgf[side * 128 + 96 + f] * (1.f - hv * hv);
      }
      __syncthreads();
      GS_STAMP(7);

      // ============================================================== conv layers 3..1, backward
#pragma unroll 1
      for (int l = 3; l >= 1; --l) {
        const float* src = (l & 1) ? XA : XB;          // dPre_l
        float* dst = (l & 1) ? XB : XA;                // dPre_{l-1}
        {   // B operand of the layer, staged once: [W_r^T ; root^T]: element (k = r*32 + f_out, n = f_in) = W_r[f_in][f_out]
          if (tid < na) s_att[tid] = watt;
          const float4 (&b4)[4] = wb4;
          const float4 r4 = wr4;
          // d bias_l = column sums of dPre_l (fixed order: 8 row classes, then the classes in order)
          {
            const int n = tid & 31, part = tid >> 5;
            float sb = 0.f;
            // (clustered: member cm takes the rows of its residue class, the partial slots add up)
            for (int row = part * cs + cm; row < N; row += (GS_THREADS / 32) * cs) sb += src[row * 32 + n];
            sred[part * 32 + n] = sb;
          }
          __syncthreads();
          float* sW = (float*)sW2;
          const int f = tid >> 3, n0 = (4 * tid) & 31;         // W_r[f][n0..n0+3]  ->  k = r*32 + n0 + q, n = f
#pragma unroll
          for (int r = 0; r <= GS_NR; ++r) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r == GS_NR) {
              v[0] = r4.x; v[1] = r4.y; v[2] = r4.z; v[3] = r4.w;
            } else if (r < R) {
#pragma unroll
              for (int bb = 0; bb < 4; ++bb) {
                const float at = s_att[r * 4 + bb];
                v[0] += at * b4[bb].x; v[1] += at * b4[bb].y; v[2] += at * b4[bb].z; v[3] += at * b4[bb].w;
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) sW[((r * 32 + n0 + q) * 16 + (f & 15)) * 2 + (f >> 4)] = v[q];
          }
        }
        float* wpart = m.ts_part + ((size_t)l * IGMC_TS_BLOCKS + blockIdx.x) * ts;
        if (tid < 32) {
          float s = 0.f;
          for (int p = 0; p < GS_THREADS / 32; ++p) s += sred[p * 32 + tid];
          if (first_graph) wpart[(R * 32 + 32) * 32 + tid] = s;
          else wpart[(R * 32 + 32) * 32 + tid] += s;
        }
        f32x4 wacc[2][GS_WN];                          // h_{l-1}^T [T' | dPre_l] of this wave's bundles
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int nt = 0; nt < GS_WN; ++nt) wacc[m2][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        const int ns = sched[2 * GS_WMAX * GS_SMAX + GS_WMAX + gw];
        // clustered launches give every wave at most ONE bundle per layer: the four tiles of the workgroup then stay
        // in LDS until all waves are done, and the table product is split by OUTPUT tile (wave w: 6 of the 24 tiles,
        // K = the 64 rows of the four bundles) -- no cross-wave reduction at all
        const int* nsb = sched + 2 * GS_WMAX * GS_SMAX + GS_WMAX + cm * GS_NW;
        const bool split_out = nsb[0] <= 1 && nsb[1] <= 1 && nsb[2] <= 1 && nsb[3] <= 1;
        if (l == 3) GS_STAMP(23);
#pragma unroll 1
        for (int si = 0; si < ns; ++si) {
          const int b0 = sched[(GS_WMAX + gw) * GS_SMAX + si] * 16;
          int lane_ = lane;
          GS_OPAQUE(lane_);
          const int qd_ = lane_ >> 2, j_ = lane_ & 3, li_ = lane_ & 15, kq_ = lane_ >> 4;
          if (l == 3 && si == 0) GS_STAMP(24);
          // h_{l-1} rows of the bundle (zero beyond N: they are K entries of the weight-gradient product)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int o = lane_ + 64 * h2, rr = o >> 3, c4 = o & 7;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + rr < N) {
              const float* hp = m.h[l - 1] + (size_t)(nb + order[b0 + rr]) * 32 + 4 * c4;
              if (cs > 1) v = make_float4(gs_sub(hp), gs_sub(hp + 1), gs_sub(hp + 2), gs_sub(hp + 3));   // rows of other members
              else v = *(const float4*)hp;
            }
            *(float4*)(HS + rr * GS_HP + 4 * c4) = v;
          }
          if (l == 3 && si == 0) GS_STAMP(25);
          if (R < GS_NR) {
#pragma unroll
            for (int r = 0; r < GS_NR; ++r) {
              *(float4*)(T + qd_ * GS_TP + r * 32 + 8 * j_) = make_float4(0.f, 0.f, 0.f, 0.f);
              *(float4*)(T + qd_ * GS_TP + r * 32 + 8 * j_ + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            IGMC_WAVE_SYNC();
          }
          gs_gather<FLAGS, true>(b, src, zrow, nb, T, relp, order, ulist + (b0 >> 4) * 16 * GS_NR, b0, N, R, qd_, j_);
          IGMC_WAVE_SYNC();
          if (l == 3 && si == 0) GS_STAMP(26);
          // dX tile = [T' | dPre_l] @ [W_r^T ; root^T]
          const int rowA = order[(b0 + li_ < N) ? b0 + li_ : N - 1];
          f32x4 acc[2][4];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
          GS_TRANSFORM(T, src, rowA, sW2, li_, kq_, acc);
          if (l == 3 && si == 0) GS_STAMP(27);
          // weight-gradient table: K = the 16 rows of the bundle (4 k-steps), 2 x GS_WN output tiles
          if (!split_out)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int prk = b0 + 4 * s + kq_;
            const int rowK = order[(prk < N) ? prk : N - 1];
            const float a0 = HS[(4 * s + kq_) * GS_HP + li_], a1 = HS[(4 * s + kq_) * GS_HP + 16 + li_];
#pragma unroll
            for (int nt = 0; nt < GS_WN; ++nt) {
              const float bv = (nt < GS_NR * 2) ? T[(4 * s + kq_) * GS_TP + nt * 16 + li_]
                                                : src[rowK * 32 + (nt - GS_NR * 2) * 16 + li_];
              wacc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, wacc[0][nt], 0, 0, 0);
              wacc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, wacc[1][nt], 0, 0, 0);
            }
          }
          if (l == 3 && si == 0) GS_STAMP(28);
          // epilogue: + readout gradient on the centre rows, * tanh'(h_{l-1})
          const int4 orow4 = *(const int4*)(order + b0 + kq_ * 4);
          const int orows[4] = {orow4.x, orow4.y, orow4.z, orow4.w};
          float hx0[4], hx1[4];                       // h_{l-1} of the four output rows, requested together
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            hx0[rr] = HS[(kq_ * 4 + rr) * GS_HP + li_];
            hx1[rr] = HS[(kq_ * 4 + rr) * GS_HP + 16 + li_];
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int p2 = b0 + kq_ * 4 + rr;
            if (p2 < N) {
              const int orow = orows[rr];
              float v0 = (acc[0][0][rr] + acc[0][1][rr]) + (acc[0][2][rr] + acc[0][3][rr]);
              float v1 = (acc[1][0][rr] + acc[1][1][rr]) + (acc[1][2][rr] + acc[1][3][rr]);
              if (orow == 0 || orow == cu) {
                const float* gf = sgf + (orow == 0 ? 0 : 128) + (l - 1) * 32;
                v0 += gf[li_];
                v1 += gf[16 + li_];
              }
              const float x0 = hx0[rr], x1 = hx1[rr];
              const float d0 = v0 * (1.f - x0 * x0), d1 = v1 * (1.f - x1 * x1);
              dst[orow * 32 + li_] = d0;
              dst[orow * 32 + 16 + li_] = d1;
              if (cs > 1 && l > 1) {
                unsigned long long* xp = m.gs_ll + (size_t)(6 - l) * m.gs_ll_stride + (size_t)(nb + orow) * 32;
                gs_ll_pub(xp + li_, d0, tag0 + (6 - l));
                gs_ll_pub(xp + 16 + li_, d1, tag0 + (6 - l));
              }
            }
          }
          IGMC_WAVE_SYNC();
          if (l == 3 && si == 0) GS_STAMP(29);
        }
        if (l == 3) GS_STAMP(30);
        if (l == 3) GS_WSTAMP(36);
        if (split_out) {
          __syncthreads();                             // the four bundle tiles / h chunks of the workgroup are complete
          if (l == 3) GS_STAMP(13);
          f32x4 w6[6];
#pragma unroll
          for (int i6 = 0; i6 < 6; ++i6) w6[i6] = (f32x4){0.f, 0.f, 0.f, 0.f};
          // tile tt = 6 wave + i6 of the 2 x 12 output tiles: row half m2 = wave >> 1, column tile nt = 6 (wave & 1) + i6;
          // column tiles 0..9 are T' (relations), 10..11 dPre (root).  All 32 operands of a bundle are requested
          // before its 24 MFMAs (issued one by one, each MFMA waits for its own LDS read: 11.5 k -> cycles per layer)
          static_assert(GS_NR == 5 && GS_WN == 12, "tile split of the table product");
          const int m2w = wave >> 1, wo = wave & 1;
#pragma unroll 1
          for (int wb = 0; wb < GS_NW; ++wb) {
            if (nsb[wb] == 0) continue;
            const int b0 = sched[(GS_WMAX + cm * GS_NW + wb) * GS_SMAX] * 16;
            const float* Tb = TILES + wb * 16 * GS_TP;
            const float* Hb = HSS + wb * 16 * GS_HP;
            float av[4], bw[4][6];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const int prk = b0 + 4 * s4 + kq;
              const int rowK = order[(prk < N) ? prk : N - 1];
              av[s4] = Hb[(4 * s4 + kq) * GS_HP + m2w * 16 + li];
              const float* tb = Tb + (4 * s4 + kq) * GS_TP + li;
#pragma unroll
              for (int i6 = 0; i6 < 4; ++i6) bw[s4][i6] = tb[(wo * 6 + i6) * 16];
              const float* pr = wo ? src + rowK * 32 + li : tb + 64;       // column tiles 10, 11 (odd waves) / 4, 5
              bw[s4][4] = pr[0];
              bw[s4][5] = pr[16];
            }
#ifndef IGMC_HIPEMU
            __builtin_amdgcn_sched_barrier(0);           // keep the requests together (the scheduler sinks them back)
#endif
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
              for (int i6 = 0; i6 < 6; ++i6)
                w6[i6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bw[s4][i6], w6[i6], 0, 0, 0);
          }
          if (l == 3) GS_STAMP(14);
#pragma unroll
          for (int i6 = 0; i6 < 6; ++i6) {
            const int tt = wave * 6 + i6, m2 = tt / GS_WN, nt = tt % GS_WN;
            const int r = nt >> 1;                      // 32-column block: relation, or GS_NR = root
            if (r >= R && r < GS_NR) continue;
            float* pp = wpart + (kq * 4) * 32 + li + (r < R ? r : R) * 1024 + m2 * 512 + (nt & 1) * 16;
            if (first_graph) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = w6[i6][rr];
            } else {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) pp[rr * 32] += w6[i6][rr];
            }
          }
          __syncthreads();
        } else {
        // ---- combine the 4 waves' tables as (w0 + w1) + (w2 + w3) through LDS (tile + h-chunk regions are free now;
          //      element q of lane x lives at [q][x]: conflict-free, identical in every wave) and emit
          //      [dW_r (r < R) | d root] in the layout of the layer-0 table: row = r*32 + f_in (resp. R*32 + f_in)
          __syncthreads();
          {
            float* buf = TILES + (wave >> 1) * (2 * GS_WN * 4 * 64);
            if (wave & 1) {
  #pragma unroll
              for (int m2 = 0; m2 < 2; ++m2)
  #pragma unroll
                for (int nt = 0; nt < GS_WN; ++nt)
  #pragma unroll
                  for (int rr = 0; rr < 4; ++rr) buf[((m2 * GS_WN + nt) * 4 + rr) * 64 + lane] = wacc[m2][nt][rr];
            }
            __syncthreads();
            if (!(wave & 1)) {
  #pragma unroll
              for (int m2 = 0; m2 < 2; ++m2)
  #pragma unroll
                for (int nt = 0; nt < GS_WN; ++nt)
  #pragma unroll
                  for (int rr = 0; rr < 4; ++rr) wacc[m2][nt][rr] += buf[((m2 * GS_WN + nt) * 4 + rr) * 64 + lane];
            }
            __syncthreads();
            if (wave == 2) {
  #pragma unroll
              for (int m2 = 0; m2 < 2; ++m2)
  #pragma unroll
                for (int nt = 0; nt < GS_WN; ++nt)
  #pragma unroll
                  for (int rr = 0; rr < 4; ++rr) TILES[((m2 * GS_WN + nt) * 4 + rr) * 64 + lane] = wacc[m2][nt][rr];
            }
            __syncthreads();
            if (wave == 0) {
              float* wp = wpart + (kq * 4) * 32 + li;          // + (row block)*1024 + m2*512 + rr*32 + (nt & 1)*16
  #pragma unroll
              for (int m2 = 0; m2 < 2; ++m2)
  #pragma unroll
                for (int nt = 0; nt < GS_WN; ++nt) {
                  const int r = nt >> 1;                        // 32-column block: relation, or GS_NR = root
                  if (r >= R && r < GS_NR) continue;
                  float* pp = wp + (r < R ? r : R) * 1024 + m2 * 512 + (nt & 1) * 16;
                  float v[4];
  #pragma unroll
                  for (int rr = 0; rr < 4; ++rr) v[rr] = wacc[m2][nt][rr] + TILES[((m2 * GS_WN + nt) * 4 + rr) * 64 + lane];
                  if (first_graph) {
  #pragma unroll
                    for (int rr = 0; rr < 4; ++rr) pp[rr * 32] = v[rr];
                  } else {
  #pragma unroll
                    for (int rr = 0; rr < 4; ++rr) pp[rr * 32] += v[rr];
                  }
                }
            }
            __syncthreads();
          }
        }
        if (l == 3) GS_STAMP(15);
        if (l > 1) wpre(l - 1);
        if (l == 3) GS_STAMP(21);
        if (cs > 1 && l > 1) {    // every member needs all of dPre_{l-1} (dPre_0 is only used row by row, below)
          gs_ll_reload(dst, m.gs_ll + (size_t)(6 - l) * m.gs_ll_stride + (size_t)nb * 32, N, tag0 + (6 - l), m.gs_err);
          __syncthreads();
        }
        if (l == 3) GS_STAMP(31);
        GS_STAMP(11 - l);
      }

      // ============================================================== layer-0 table gradient (dPre_0 is in XB)
      // T0'[c][f] = sum_i [cnt | onehot | 1](i, c) dPre_0[i][f]; wave = (code half, feature half), K = all rows
      {   // K = the rows of THIS workgroup's bundles (their dPre_0 is in XB; clustered: the other members add theirs
          // through their own partial slots)
        const int m2 = wave >> 1, wn = wave & 1;
        const int code = m2 * 16 + li;
        for (int w2 = 0; w2 < GS_NW; ++w2) {
          const int gw2 = cm * GS_NW + w2;
          const int ns2 = sched[2 * GS_WMAX * GS_SMAX + GS_WMAX + gw2];
          for (int si = 0; si < ns2; ++si) {
            const int b0 = sched[(GS_WMAX + gw2) * GS_SMAX + si] * 16;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const int prow = b0 + 4 * s4 + kq;
              const int rc = order[(prow < N) ? prow : N - 1];
              const int cv = (cnt[rc * rlp + ((code < RL) ? (code >> 1) : 0)] >> ((code & 1) * 16)) & 0xFFFF;
              float av = (code < RL) ? (float)cv : ((code == RL + slab[rc] || code == RL + L) ? 1.f : 0.f);
              if (prow >= N) av = 0.f;
              const float bv = XB[rc * 32 + wn * 16 + li];
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
            }
          }
        }
      }
      first_graph = false;
      __syncthreads();
      GS_STAMP(11);
    }
  }

  if (TRAIN) {
    // ---- layer-0 table partial: rows c < R*L + L + 1 of [32 codes][32]
    float* part0 = m.ts_part + (size_t)blockIdx.x * ts;        // slice 0 of [4][IGMC_TS_BLOCKS][ts]
    const int m2 = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int c = m2 * 16 + kq * 4 + rr;
      if (c < RL + L + 1) part0[c * 32 + wn * 16 + li] = acc0[rr];
    }
  }
#ifndef IGMC_HIPEMU
  if (cs > 1 && tid == 0) {
    // the workgroup that finishes the launch LAST advances the sequence number: every workgroup has read it by then
    // (relaxed: only the counters themselves are communicated; the next launch starts after this one has completed)
    if (__hip_atomic_fetch_add(m.gs_bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(m.gs_bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(m.gs_bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#else
  if (cs > 1 && tid == 0) {
    if (m.gs_bar[0]++ == (int)gridDim.x - 1) {
      m.gs_bar[0] = 0;
      m.gs_bar[1] += 1;
    }
  }
#endif
#ifndef IGMC_HIPEMU
  if (a.ts && tid == 0) {
    const unsigned long long t1 = (unsigned long long)wall_clock64();
    if (atomicAdd(a.ts + 3, 1ull) == (unsigned long long)gridDim.x - 1ull) {      // last workgroup of the launch
      const unsigned long long t0 = __hip_atomic_load(a.ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(a.ts + 1, t1 - t0);
      atomicAdd(a.ts + 2, 1ull);
      __hip_atomic_store(a.ts, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ts + 3, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
  GS_STAMP(12);
}

// ------------------------------------------------------------------------------------------------ host side
int igmc_gs_layout(const ModelDev& m, const BatchDev& b, GsLayout* lay) {
  const int RL = m.R * m.L;
  if (m.S != 0 || m.D != 256 || m.R > GS_NR || RL + m.L + 1 > 32 || !m.ts_part) return 0;
  if (b.slot < 2) return 0;
  const int nmax = (b.slot + 15) & ~15;
  if (nmax > 16 * GS_SMAX) return 0;
  int o = 0;
  lay->nmax = nmax;
  lay->rlp = ((RL + 1) >> 1) | 1;            // words per histogram row (two 16-bit counters each), odd pitch
  lay->xa = o; o += nmax * 32;
  lay->xb = o; o += nmax * 32;
  lay->zrow = o; o += 32;
  lay->tile = o; o += GS_NW * 16 * GS_TP;    // >= GS_KT*32 (W_r staging) and >= 32*(GS_KT+32) (table reduction)
  lay->hs = o; o += GS_NW * 16 * GS_HP;
  lay->att = o; o += 64;
  o = (o + 3) & ~3;
  lay->t0 = o; o += 1024;
  lay->cnt = o; o += nmax * lay->rlp;
  lay->rp = o; o += nmax + 1;
  lay->lab = o; o += nmax;
  o = (o + 3) & ~3;
  lay->deg = o; o += nmax;
  lay->order = o; o += nmax;
  lay->sched = o; o += 2 * GS_WMAX * GS_SMAX + 2 * GS_WMAX;
  o = (o + 3) & ~3;
  lay->relp = o; o += nmax * 8;
  lay->wreg = o; o += (GS_KT + 32) * 32;
  lay->ulist = o; o += GS_SMAX * 16 * GS_NR / 4;
  o = (o + 3) & ~3;
  lay->head = o; o += 256 + 256 + 3 * 128 + 256 + 16;
  lay->words = o;
  return (size_t)o * 4 <= 160 * 1024;
}

// Default ON where eligible (IGMC_GRAPH_STEP=0 forces the per-layer kernels of model.hip).
static int igmc_gs_enabled() {
  const char* e = getenv("IGMC_GRAPH_STEP");      // read on every call: tests switch it per case
  return e ? atoi(e) : 1;
}

int igmc_gs_eligible(const ModelDev& m, const BatchDev& b, GsLayout* lay) {
  return igmc_gs_enabled() && igmc_gs_layout(m, b, lay);
}

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

void igmc_launch_graph_step(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                            const GsLayout& lay, const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult,
                            float grad_scale, float* out, void* stream) {
  GsArgs a;
  a.inj_mask = inj_mask;
  a.seed = seed;
  a.step = step;
  a.mult = mult;
  a.grad_scale = grad_scale;
  a.out = out;
  a.lay = lay;
  a.timing = getenv("IGMC_GS_TIMING") ? 1 : 0;
  a.ts = (g_igmc_prof_on == 2) ? m.gs_ts : nullptr;
  const int cs = igmc_gs_cluster(B);
  a.cs = cs;
  a.stride = (cs > 1) ? ((B + 7) & ~7) : 1;
  const int grid = (cs > 1) ? cs * a.stride : igmc_gs_grid(B);
  const size_t sm = (size_t)lay.words * 4;
#ifdef IGMC_HIPEMU
  if (cs > 1) {
    hipemu::rt().co_cs = cs;
    hipemu::rt().co_stride = a.stride;
  }
#endif
  if (getenv("IGMC_GS_TRACE")) fprintf(stderr, "[igmc] k_graph_step B=%d train=%d flags=%d nmax=%d lds=%zu cluster=%d grid=%d\n", B, training, use_flags, lay.nmax, sm, cs, grid);
  if (training) {
    if (use_flags) IGMC_PLAUNCH("k_graph_step", (k_graph_step<true, true>), grid, GS_THREADS, sm, stream, b, m, P, a);
    else IGMC_PLAUNCH("k_graph_step", (k_graph_step<false, true>), grid, GS_THREADS, sm, stream, b, m, P, a);
  } else {
    if (use_flags) IGMC_PLAUNCH("k_graph_fwd", (k_graph_step<true, false>), grid, GS_THREADS, sm, stream, b, m, P, a);
    else IGMC_PLAUNCH("k_graph_fwd", (k_graph_step<false, false>), grid, GS_THREADS, sm, stream, b, m, P, a);
  }
}

// dynamic LDS above 64 KB needs an explicit opt-in on HIP
int igmc_gs_prepare() {
  if (igmc_g2_prepare()) return 1;
#ifndef IGMC_HIPEMU
  const int mx = 160 * 1024;
  if (hipFuncSetAttribute((const void*)k_graph_step<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_graph_step<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_graph_step<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void*)k_graph_step<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess) return 1;
#endif
  return 0;
}

// debug aid: 1 when a cluster barrier of k_graph_step ever timed out on this model workspace
extern "C" int igmc_debug_gs_error(const void* model_dev_gs_err) {
  int v = 0;
#ifndef IGMC_HIPEMU
  if (!model_dev_gs_err) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(&v, model_dev_gs_err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
#endif
  return v;
}

// debug aid: phase clocks (shader cycles) of workgroup 0 of the last k_graph_step launched with IGMC_GS_TIMING set
extern "C" int igmc_debug_gs_clocks(unsigned long long* out, int n) {
#ifndef IGMC_HIPEMU
  if (n > 64) n = 64;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gs_clk), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return 1;
#else
  for (int i = 0; i < n; ++i) out[i] = 0;
#endif
  return 0;
}
