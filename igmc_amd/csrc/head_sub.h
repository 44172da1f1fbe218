// head_sub.h -- readout + MLP head of ONE subgraph by 512 threads (reference models.py:203-215; internal).
//
// The conv features of the two target rows (+ the link's side-feature row, models.py:208-209) -> lin1 / ReLU / dropout(0.5) /
// lin2 / residual -> (training) dz, d feat and dPre_3 on the target rows.  The body of k_head_sub (graphstep2.hip: a launch of
// its own behind the dense-layer forward) and of the head inside k_dl_bwd's set-up -- the same code, so the two launch forms
// agree bit for bit.
#pragma once
#include "model.h"

#ifdef IGMC_HIPEMU
#define IGMC_HS_SCHED_BARRIER() do { } while (0)
#else
#define IGMC_HS_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// LDS of the head: sfeat[D] | sa1[128] | skeep[128] | sdz[128] | sred[128] | part8[8 * 256] | misc[4]
static inline int igmc_head_sub_lds_floats(int D) { return ((D + 3) & ~3) + 4 * 128 + 8 * 256 + 4; }

// What the head loads that depends on nothing but the subgraph and the parameters: requested by the caller AHEAD of its own
// set-up (k_dl_bwd: in front of its block-row staging), so that the head's first two round trips -- the target rows' features,
// the wave's sixteen lin1 rows -- are over when the head starts.  512 threads = eight waves: wave w takes hidden units
// 16 w .. 16 w + 15, lane = 4 fan-in columns.
struct HeadPre {
  float fv, yv, l2b, l1b, l2w, l2w_t;
  float4 w4[16];
};

// nu / nv: first node of the subgraph's users / items in the collated batch (b.node_off[g], + b.n_users[g])
__device__ __forceinline__ void head_sub_prefetch(HeadPre& hp, const BatchDev& b, const ModelDev& m, const float* __restrict__ P,
                                                  int g, int tid, int nu, int nv) {
  const int D = m.D;
  const int lane = tid & 63, wave = (tid >> 6) & 7;
  const int t8 = tid & 255;
  const size_t trow = (size_t)((t8 >> 7) ? nv : nu) * 32 + (t8 & 31);          // feature t8: side, layer, column
  const int ju = 16 * wave + ((lane >> 1) & 15);
  hp.fv = m.h[(t8 >> 5) & 3][trow];
  hp.yv = b.y[g];
  hp.l2b = P[m.off_l2b];
  hp.l1b = P[m.off_l1b + ju];
  hp.l2w = P[m.off_l2w + ju];
  hp.l2w_t = P[m.off_l2w + (tid & 127)];
  const float* wrow = P + m.off_l1w + (int64_t)(16 * wave) * D + 4 * lane;
#pragma unroll
  for (int q = 0; q < 16; ++q) hp.w4[q] = *(const float4*)(wrow + (int64_t)q * D);
}

// The head of subgraph g by 512 threads (eight waves), on what head_sub_prefetch requested.
template <bool TRAIN>
__device__ __forceinline__ void head_sub_compute(const HeadPre& hp, const BatchDev& b, const ModelDev& m, const float* __restrict__ P,
                                                 int g, int tid, int nu, int nv, float* lds, const uint8_t* __restrict__ inj_mask,
                                                 uint64_t seed, uint64_t step, float mult, float grad_scale, float* __restrict__ out,
                                                 float* dpre3_alt = nullptr, size_t alt_bias = 0) {
  const int D = m.D, S = m.S;
  float* sfeat = lds;
  float* sa1 = sfeat + ((D + 3) & ~3);
  float* skeep = sa1 + 128;
  float* sdz = skeep + 128;
  float* sred = sdz + 128;
  float* part8 = sred + 128;
  float* misc = part8 + 8 * 256;
  const bool on = tid < 256;
  const int lane = tid & 63, wave = (tid >> 6) & 7;
  const int t8 = tid & 255;
  const size_t trow = (size_t)((t8 >> 7) ? nv : nu) * 32 + (t8 & 31);
  const float fv = hp.fv;
  if (on) {
    sfeat[tid] = fv;
    if (TRAIN) m.feat[(size_t)g * D + tid] = fv;
    for (int k = tid; k < S; k += 256) {
      const float sv = m.side[(size_t)g * S + k];
      sfeat[256 + k] = sv;
      if (TRAIN) m.feat[(size_t)g * D + 256 + k] = sv;
    }
  }
  __syncthreads();
  {
    // lin1 (256 [+ S] -> 128): the 16 per-lane partial dot products of the wave's units are reduced over the 64 lanes by a
    // transposing butterfly (lane bits 4..1 -> the unit, bits 0 and 5 summed last); the side-feature columns are a short dot
    // product of the unit's owner lane
    const int ju = 16 * wave + ((lane >> 1) & 15);
    const float4 f4 = *(const float4*)(sfeat + 4 * lane);
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = (hp.w4[q].x * f4.x + hp.w4[q].y * f4.y) + (hp.w4[q].z * f4.z + hp.w4[q].w * f4.w);
#define IGMC_HS_BFLY(H)                                                              \
    {                                                                                \
      const bool up = (lane & (2 * (H))) != 0;                                       \
      _Pragma("unroll") for (int i = 0; i < (H); ++i) {                              \
        const float send = up ? v[i] : v[i + (H)], keep = up ? v[i + (H)] : v[i];    \
        v[i] = keep + __shfl_xor(send, 2 * (H));                                     \
      }                                                                              \
    }
    IGMC_HS_BFLY(8) IGMC_HS_BFLY(4) IGMC_HS_BFLY(2) IGMC_HS_BFLY(1)
#undef IGMC_HS_BFLY
    float s = v[0] + __shfl_xor(v[0], 1);
    s += __shfl_xor(s, 32);
    if ((lane & 33) == 0) {
      if (S > 0) {
        const float* ws = P + m.off_l1w + (int64_t)ju * D + 256;
        float ss = 0.f;
        for (int k = 0; k < S; ++k) ss += ws[k] * sfeat[256 + k];
        s += ss;
      }
      float av = s + hp.l1b;
      av = av > 0.f ? av : 0.f;
      int keep = 1;
      if (TRAIN) {
        keep = inj_mask ? (int)inj_mask[g * 128 + ju]
                        : (int)(igmc_u01(igmc_unit_hash(seed, step, (uint32_t)g, (uint32_t)ju)) >= 0.5f);
        m.a1[g * 128 + ju] = av;
        m.lmask[g * 128 + ju] = (uint8_t)keep;
        sa1[ju] = av;
        skeep[ju] = keep ? 1.f : 0.f;
      }
      sred[ju] = (TRAIN ? (keep ? av * 2.f : 0.f) : av) * hp.l2w;       // F.dropout(p = 0.5): kept * 2
    }
  }
  __syncthreads();
  if (tid < 64) {
    float s = sred[lane] + sred[lane + 64];
    s = igmc_wave_sum_f(s);
    if (lane == 0) {
      const float o = (s + hp.l2b) * mult;
      out[g] = o;
      m.err[g] = o - hp.yv;
      misc[0] = o - hp.yv;
    }
  }
  if (!TRAIN) return;
  __syncthreads();
  if (tid < 128) {
    const float dp = 2.f * misc[0] * grad_scale * mult;
    const float dzv = (sa1[tid] > 0.f && skeep[tid] != 0.f) ? dp * hp.l2w_t * 2.f : 0.f;
    sdz[tid] = dzv;
    m.dz[g * 128 + tid] = dzv;
  }
  __syncthreads();
  {   // d feat = dz @ lin1.weight (conv columns): wave w takes hidden units 16 w .. 16 w + 15, lane -> 4 fan-in columns -- from
      // the rows the prefetch left in registers (round 6: a second, data-dependent round trip to the weights before; rows with
      // dz == 0 -- ReLU / dropout: ~3/4 of them -- add exact zeros, the non-zero rows in the same order as before)
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* dz4 = (const float4*)(sdz + 16 * wave);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 d4 = dz4[q4];
      const float dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 wv = hp.w4[4 * q4 + u];
        s4.x += dq[u] * wv.x; s4.y += dq[u] * wv.y; s4.z += dq[u] * wv.z; s4.w += dq[u] * wv.w;
      }
    }
    *(float4*)(part8 + wave * 256 + 4 * lane) = s4;
  }
  __syncthreads();
  if (on) {
    const float v = ((part8[tid] + part8[256 + tid]) + (part8[512 + tid] + part8[768 + tid])) +
                    ((part8[1024 + tid] + part8[1280 + tid]) + (part8[1536 + tid] + part8[1792 + tid]));
    m.gfeat[(size_t)g * D + tid] = v;
    if (((tid >> 5) & 3) == 3) {      // dPre_3: non-zero on the two target rows only
      m.dpre[3][trow] = v * (1.f - fv * fv);
      if (dpre3_alt) dpre3_alt[trow - alt_bias] = v * (1.f - fv * fv);      // (a caller's own copy of the target rows)
    }
  }
}
