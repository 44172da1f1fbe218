// Micro-benchmark: (1) latency of DEPENDENT 16-byte buffer loads by cache policy (a pointer chase over 64 lines written
// by the host), (2) the pieces of a tagged-word hand-off between two workgroups: polling with / without a sleep, the
// word read by a returning atomic instead of a load, 8-byte words.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/xcd_latency.hip -o tools/ubench/xcd_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ u32x4 ld16(const void* p, int off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void st16(void* p, int off, u32x4 v) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
}

template <int AUX>
__global__ void k_chase(const uint32_t* chain, int n, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int off = 0;
  for (int rep = 0; rep < 2; ++rep) {          // second pass: lines are wherever the first pass left them
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) off = (int)ld16<AUX>(chain, off).x;
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[rep] = t1 - t0;
    out[2] = (unsigned long long)off;
  }
}

// MODE 0: poll with s_sleep(1); 1: poll without sleep; 2: poll with a returning atomic OR of zero (agent scope) on the
// first 8 bytes; 3: __hip_atomic_load agent scope 8 bytes
template <int SAUX, int LAUX, int MODE>
__global__ void k_pp(unsigned long long* buf, int wa, int wb, int rounds, unsigned long long* out) {
  const int me = (int)blockIdx.x;
  if (me != wa && me != wb) return;
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned long long* A = buf + 2 * lane;
  unsigned long long* B = buf + 4096 + 2 * lane;
  unsigned long long stale = 0, bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= rounds; ++i) {
    const u32x4 v = {(uint32_t)i, (uint32_t)i, (uint32_t)i, (uint32_t)i};
    if (me == wa) st16<SAUX>(A, 0, v);
    unsigned long long* src = (me == wa) ? B : A;
    for (int it = 0;; ++it) {
      uint32_t got;
      if (MODE == 2) got = (uint32_t)__hip_atomic_fetch_or(src, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 3) got = (uint32_t)__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else got = ld16<LAUX>(src, 0).x;
      if (got == (uint32_t)i) break;
      ++stale;
      if (it > (1 << 16)) { bad = 1; break; }
      if (MODE == 0) __builtin_amdgcn_s_sleep(1);
    }
    if (bad) break;
    if (me == wb) st16<SAUX>(B, 0, v);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0 && me == wa) { out[0] = t1 - t0; out[1] = stale; out[2] = bad; }
}

template <int AUX>
static void chase(const char* name, const uint32_t* chain, unsigned long long* out) {
  const int n = 512;
  hipMemset(out, 0, 64);
  hipLaunchKernelGGL((k_chase<AUX>), dim3(1), dim3(64), 0, 0, chain, n, out);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("dependent loads, %-14s: first pass %6.0f cycles per load, second pass %6.0f\n", name, h[0] / (double)n, h[1] / (double)n);
}
template <int SAUX, int LAUX, int MODE>
static void pp(const char* name, unsigned long long* buf, unsigned long long* out, int wa, int wb) {
  const int rounds = 2000;
  hipMemset(buf, 0, 8192 * 8);
  hipMemset(out, 0, 64);
  hipLaunchKernelGGL((k_pp<SAUX, LAUX, MODE>), dim3(64), dim3(256), 0, 0, buf, wa, wb, rounds, out);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("hand-off %-44s wg %2d <-> %2d: %7.1f ns per hand-off, %5.2f stale polls%s\n", name, wa, wb, h[0] * 10.0 / rounds / 2, h[1] / (double)rounds,
         h[2] ? "  ** GAVE UP **" : "");
}

int main() {
  const int n = 512;
  uint32_t* hc = (uint32_t*)calloc(n * 64, 4);            // 256-byte stride: line i -> line (i * 37 + 1) % n
  for (int i = 0; i < n; ++i) hc[i * 64] = (uint32_t)(((i * 37 + 1) % n) * 256);
  uint32_t* chain;
  unsigned long long *buf, *out;
  hipMalloc(&chain, n * 256);
  hipMemcpy(chain, hc, n * 256, hipMemcpyHostToDevice);
  hipMalloc(&buf, 8192 * 8);
  hipMalloc(&out, 64);
  chase<0>("plain", chain, out);
  chase<1>("sc0", chain, out);
  chase<2>("nt", chain, out);
  chase<16>("sc1", chain, out);
  chase<17>("sc0 sc1", chain, out);
  chase<0>("plain", chain, out);
  for (int pl = 0; pl < 2; ++pl) {
    const int wa = 3, wb = pl ? 11 : 4;
    pp<16, 16, 0>("sc1 / sc1, s_sleep(1)", buf, out, wa, wb);
    pp<16, 16, 1>("sc1 / sc1, no sleep", buf, out, wa, wb);
    pp<16, 16, 2>("sc1 store, returning atomic or (agent)", buf, out, wa, wb);
    pp<16, 16, 3>("sc1 store, atomic load (agent) 8 B", buf, out, wa, wb);
    pp<17, 17, 1>("sc0 sc1 / sc0 sc1, no sleep", buf, out, wa, wb);
    if (pl) pp<0, 2, 1>("plain / nt, no sleep", buf, out, wa, wb);
    if (pl) pp<0, 16, 1>("plain / sc1, no sleep", buf, out, wa, wb);
  }
  return 0;
}
