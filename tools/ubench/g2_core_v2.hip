// Micro-benchmark of the two inner routines of graphstep2.hip (copied verbatim by tools/ubench/make_g2_core.py):
// cycles of g2_gather (4 k-steps) and g2_transform for one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define G2_NR 5
#define G2_KS 4
#define G2_NT 3
#define G2_XP 36
#define G2_WIMG (G2_NT * (G2_NR + 1) * 2 * 64 * 4)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef IGMC_HIPEMU
typedef __bf16 g2_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g2_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g2_f32x2 __attribute__((ext_vector_type(2)));
#endif

__device__ __forceinline__ f32x4 g2_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
#ifdef IGMC_HIPEMU
  return igmc_emu_mfma_16x16x32_bf16(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g2_bf16x8, a), __builtin_bit_cast(g2_bf16x8, b), c, 0, 0, 0);
#endif
}

// {bf16(x) | bf16(y) << 16}, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t g2_pk_bf16(float x, float y) {
#ifdef IGMC_HIPEMU
  return hipemu_f32_to_bf16_rne(x) | (hipemu_f32_to_bf16_rne(y) << 16);
#else
  g2_f32x2 v = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g2_bf16x2));
#endif
}
// the three bf16 terms of two f32 values: hi + mid + lo == x to 24 bits (each residual is exact in f32)
__device__ __forceinline__ void g2_split2(float x, float y, uint32_t& h, uint32_t& mi, uint32_t& lo) {
  h = g2_pk_bf16(x, y);
  const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xFFFF0000u);
  mi = g2_pk_bf16(rx, ry);
  const float sx = rx - __uint_as_float(mi << 16), sy = ry - __uint_as_float(mi & 0xFFFF0000u);
  lo = g2_pk_bf16(sx, sy);
}

#ifdef IGMC_HIPEMU
#define G2_SCHED_BARRIER() do { } while (0)
#else
#define G2_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
__device__ __forceinline__ void g2_gather(const uint32_t* pl, int kp, int nks, const uint32_t (&A)[G2_NR][G2_KS][4],
                                          int li, int kq, f32x4 (&acc)[G2_NR][2]) {
#pragma unroll
  for (int r = 0; r < G2_NR; ++r) {
    acc[r][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[r][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int tstride = 32 * kp >> 1;            // dwords per term
  const uint32_t* base = pl + (li * kp >> 1) + 4 * kq;
  const int toff = 16 * kp >> 1;               // second feature tile
  u32x4 pf[2][2 * G2_NT];
  auto request = [&](int s, int buf) {
#pragma unroll
    for (int sp = 0; sp < G2_NT; ++sp) {
      pf[buf][2 * sp] = *(const u32x4*)(base + sp * tstride + 16 * s);
      pf[buf][2 * sp + 1] = *(const u32x4*)(base + sp * tstride + toff + 16 * s);
    }
  };
  request(0, 0);
#pragma unroll
  for (int s = 0; s < G2_KS; ++s) {
    if (s < nks) {
      if (s + 1 < G2_KS) request(s + 1, (s + 1) & 1);      // unconditional (a k-step past the side reads bytes that are
      G2_SCHED_BARRIER();                                  // never used): a guarded request is sunk below the MFMAs
#pragma unroll
      for (int q = 0; q < 2 * G2_NT; ++q) {
#pragma unroll
        for (int r = 0; r < G2_NR; ++r) {
          const u32x4 af = {A[r][s][0], A[r][s][1], A[r][s][2], A[r][s][3]};
          acc[r][q & 1] = g2_mfma_bf16(pf[s & 1][q], af, acc[r][q & 1]);
        }
      }
      G2_SCHED_BARRIER();
    }
  }
}

__device__ __forceinline__ void g2_split8(const float (&v)[8], u32x4& ah, u32x4& am, u32x4& al) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t h, mi, lo;
    g2_split2(v[2 * q], v[2 * q + 1], h, mi, lo);
    ah[q] = h;
    am[q] = mi;
    al[q] = lo;
  }
}
__device__ __forceinline__ void g2_transform(const f32x4 (&acc)[G2_NR][2], const float* xrows, const uint32_t* sW, int li, int kq,
                                             f32x4 (&o)[2]) {
  f32x4 o0a = (f32x4){0.f, 0.f, 0.f, 0.f}, o0b = o0a, o1a = o0a, o1b = o0a;
  const float4 x0 = *(const float4*)(xrows + li * G2_XP + 4 * kq), x1 = *(const float4*)(xrows + li * G2_XP + 16 + 4 * kq);
  const u32x4* wf = (const u32x4*)sW + (kq * 16 + li);
  u32x4 bf[2][2 * G2_NT];
  auto request = [&](int g, int buf) {
#pragma unroll
    for (int t = 0; t < G2_NT; ++t) {
      bf[buf][2 * t] = wf[((t * (G2_NR + 1) + g) * 2 + 0) * 64];
      bf[buf][2 * t + 1] = wf[((t * (G2_NR + 1) + g) * 2 + 1) * 64];
    }
  };
  auto values = [&](int g, float (&v)[8]) {
    if (g < G2_NR) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        v[rr] = acc[g][0][rr];
        v[4 + rr] = acc[g][1][rr];
      }
    } else {
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
      v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
  };
  u32x4 ah[2], am[2], al[2];
  {
    float v[8];
    values(0, v);
    g2_split8(v, ah[0], am[0], al[0]);
  }
  request(0, 0);
#pragma unroll
  for (int g = 0; g <= G2_NR; ++g) {
    if (g < G2_NR) {
      request(g + 1, (g + 1) & 1);
      float v[8];
      values(g + 1, v);
      g2_split8(v, ah[(g + 1) & 1], am[(g + 1) & 1], al[(g + 1) & 1]);
    }
    const u32x4 (&b)[2 * G2_NT] = bf[g & 1];
    const u32x4 h = ah[g & 1], mi = am[g & 1], lo = al[g & 1];
    o0a = g2_mfma_bf16(h, b[0], o0a);
    o1a = g2_mfma_bf16(h, b[1], o1a);
    o0b = g2_mfma_bf16(h, b[2], o0b);
    o1b = g2_mfma_bf16(h, b[3], o1b);
    o0a = g2_mfma_bf16(mi, b[0], o0a);
    o1a = g2_mfma_bf16(mi, b[1], o1a);
    o0b = g2_mfma_bf16(h, b[4], o0b);
    o1b = g2_mfma_bf16(h, b[5], o1b);
    o0a = g2_mfma_bf16(lo, b[0], o0a);
    o1a = g2_mfma_bf16(lo, b[1], o1a);
    o0b = g2_mfma_bf16(mi, b[2], o0b);
    o1b = g2_mfma_bf16(mi, b[3], o1b);
#ifndef IGMC_HIPEMU
    // interleave: one MFMA, then up to four of the next block's VALU / LDS instructions
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x102, 4, 0);     // VALU | DS read
    }
#endif
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    o[0][rr] = o0a[rr] + o0b[rr];
    o[1][rr] = o1a[rr] + o1b[rr];
  }
}

__global__ __launch_bounds__(256) void k_core(unsigned long long* clk, float* out, int nks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* pl = (uint32_t*)smem;                       // planes [3][32][136] bf16
  uint32_t* sW = pl + 3 * 32 * 136 / 2;                 // staged weights
  float* xr = (float*)(sW + G2_WIMG);                   // [4][16][36]
  for (int i = threadIdx.x; i < 3 * 32 * 136 / 2 + G2_WIMG + 4 * 16 * 36; i += 256) pl[i] = 0x3c003c00u + (i & 3);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  uint32_t A[G2_NR][G2_KS][4];
  for (int r = 0; r < G2_NR; ++r)
    for (int s2 = 0; s2 < G2_KS; ++s2)
      for (int q = 0; q < 4; ++q) A[r][s2][q] = ((lane + r + s2 + q) & 1) ? 0x3F803F80u : 0x3F80u;
  f32x4 acc[G2_NR][2];
  unsigned long long t0 = __builtin_readcyclecounter();
  g2_gather(pl, 136, nks, A, li, kq, acc);
  unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 o[2];
  g2_transform(acc, xr + wave * 16 * 36, sW, li, kq, o);
  unsigned long long t2 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = o[0][0] + o[1][3] + acc[0][0][0];
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; }
}
int main() {
  unsigned long long* clk; float* out;
  hipMalloc(&clk, 16); hipMalloc(&out, 256 * 256 * 4);
  const size_t sm = (3 * 32 * 136 / 2 + G2_WIMG + 4 * 16 * 36) * 4;
  hipFuncSetAttribute((const void*)k_core, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  for (int grid : {1, 200}) {
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_core, grid, 256, sm, 0, clk, out, 4);
    hipDeviceSynchronize();
    unsigned long long c[2];
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    printf("grid %3d: g2_gather (120 MFMA) %llu cycles = %.1f per MFMA | g2_transform (72 MFMA) %llu cycles = %.1f per MFMA\n", grid, c[0], c[0] / 120.0, c[1], c[1] / 72.0);
  }
  return 0;
}
