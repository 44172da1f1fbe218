// Issue rate of bf16 MFMAs from one wave per SIMD, hand-written instruction streams (no compiler moves / nops).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP10(x) x x x x x x x x x x
__global__ __launch_bounds__(256) void k_a16(unsigned long long* clk, int iters, float* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP10("v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n"
                       "v_mfma_f32_16x16x32_bf16 a[4:7], v[0:3], v[4:7], a[4:7]\n"
                       "v_mfma_f32_16x16x32_bf16 a[8:11], v[0:3], v[4:7], a[8:11]\n"
                       "v_mfma_f32_16x16x32_bf16 a[12:15], v[0:3], v[4:7], a[12:15]\n"
                       "v_mfma_f32_16x16x32_bf16 a[16:19], v[0:3], v[4:7], a[16:19]\n")
                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void k_a16v(unsigned long long* clk, int iters, float* out) {   // accumulators in VGPRs
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP10("v_mfma_f32_16x16x32_bf16 v[8:11], v[0:3], v[4:7], v[8:11]\n"
                       "v_mfma_f32_16x16x32_bf16 v[12:15], v[0:3], v[4:7], v[12:15]\n"
                       "v_mfma_f32_16x16x32_bf16 v[16:19], v[0:3], v[4:7], v[16:19]\n"
                       "v_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[4:7], v[20:23]\n"
                       "v_mfma_f32_16x16x32_bf16 v[24:27], v[0:3], v[4:7], v[24:27]\n")
                 ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void k_a32(unsigned long long* clk, int iters, float* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP10("v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n"
                       "v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]\n"
                       "v_mfma_f32_32x32x16_bf16 a[32:47], v[0:3], v[4:7], a[32:47]\n"
                       "v_mfma_f32_32x32x16_bf16 a[48:63], v[0:3], v[4:7], a[48:63]\n"
                       "v_mfma_f32_32x32x16_bf16 a[64:79], v[0:3], v[4:7], a[64:79]\n")
                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19",
                     "a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39",
                     "a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59",
                     "a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void k_af32(unsigned long long* clk, int iters, float* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile(REP10("v_mfma_f32_16x16x4_f32 a[0:3], v0, v4, a[0:3]\n"
                       "v_mfma_f32_16x16x4_f32 a[4:7], v0, v4, a[4:7]\n"
                       "v_mfma_f32_16x16x4_f32 a[8:11], v0, v4, a[8:11]\n"
                       "v_mfma_f32_16x16x4_f32 a[12:15], v0, v4, a[12:15]\n"
                       "v_mfma_f32_16x16x4_f32 a[16:19], v0, v4, a[16:19]\n")
                 ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[threadIdx.x] = 0.f;
}
int main() {
  unsigned long long* clk;
  float* out;
  hipMalloc(&clk, 8);
  hipMalloc(&out, 4096);
  const int iters = 400;
  struct { const char* n; void (*k)(unsigned long long*, int, float*); } ks[] = {
      {"16x16x32 bf16 (acc in AGPR), 5 independent", k_a16}, {"16x16x32 bf16 (acc in VGPR), 5 independent", k_a16v},
      {"32x32x16 bf16 (AGPR), 5 independent", k_a32}, {"16x16x4 f32 (AGPR), 5 independent", k_af32}};
  for (int grid : {1, 256})
    for (auto& e : ks) {
      for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(e.k, grid, 256, 0, 0, clk, iters, out);
      hipDeviceSynchronize();
      unsigned long long c = 0;
      hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
      printf("grid %3d  %-46s %6.2f cycles per MFMA\n", grid, e.n, (double)c / (iters * 50.0));
    }
  return 0;
}
