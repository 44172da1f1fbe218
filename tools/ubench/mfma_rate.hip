// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 / 32x32x16 bf16 / 16x16x4 f32 from ONE wave per SIMD
// (256-thread workgroups, 512-register budget), independent accumulators.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SAME_A>
__global__ __launch_bounds__(256) void k_16(const u32x4* in, float* out, unsigned long long* clk, int iters) {
  u32x4 a[4], b[5];
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
  for (int i = 0; i < 5; ++i) b[i] = in[threadIdx.x + 64 * (i + 4)];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[SAME_A ? 0 : (i & 3)]),
                                                       __builtin_bit_cast(bf16x8, b[i % 5]), acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_32(const u32x4* in, float* out, unsigned long long* clk, int iters) {
  u32x4 a = in[threadIdx.x], b = in[threadIdx.x + 64];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_f32(const float* in, float* out, unsigned long long* clk, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, int per_iter, unsigned long long* dclk) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) launch(iters);
  hipDeviceSynchronize();
  unsigned long long c = 0;
  hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
  printf("%-44s %7.2f cycles per MFMA\n", name, (double)c / ((double)iters * per_iter));
}

int main() {
  u32x4* in;
  float* out;
  unsigned long long* clk;
  hipMalloc(&in, 64 * 16 * 16);
  hipMemset(in, 0, 64 * 16 * 16);
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&clk, 8);
  for (int grid : {1, 256}) {
    printf("grid %d workgroups of 256 threads (one wave per SIMD)\n", grid);
    run("16x16x32 bf16, 4 accumulators", [&](int it) { hipLaunchKernelGGL((k_16<4, 1>), grid, 256, 0, 0, in, out, clk, it); }, 4, clk);
    run("16x16x32 bf16, 10 accumulators", [&](int it) { hipLaunchKernelGGL((k_16<10, 1>), grid, 256, 0, 0, in, out, clk, it); }, 10, clk);
    run("16x16x32 bf16, 10 accumulators, 4 A operands", [&](int it) { hipLaunchKernelGGL((k_16<10, 0>), grid, 256, 0, 0, in, out, clk, it); }, 10, clk);
    run("16x16x32 bf16, 2 accumulators", [&](int it) { hipLaunchKernelGGL((k_16<2, 1>), grid, 256, 0, 0, in, out, clk, it); }, 2, clk);
    run("32x32x16 bf16, 4 accumulators", [&](int it) { hipLaunchKernelGGL(k_32, grid, 256, 0, 0, in, out, clk, it); }, 4, clk);
    run("16x16x4 f32, 4 accumulators", [&](int it) { hipLaunchKernelGGL(k_f32, grid, 256, 0, 0, (const float*)in, out, clk, it); }, 4, clk);
  }
  return 0;
}
