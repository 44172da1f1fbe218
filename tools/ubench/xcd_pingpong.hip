// Micro-benchmark: latency of a tagged-word hand-off between two workgroups, by cache policy of the store / load and by
// placement (same XCD: workgroups b and b + 8 of a launch -- workgroups are dealt to the eight XCDs round robin -- or
// neighbouring XCDs: b and b + 1).  The question behind it: can the members of a k_graph_step2 cluster exchange their
// rows through the XCD's own L2 (store sc0 / load sc0: write-through to L2, L1 bypassed) instead of the device-coherent
// level (sc1), and what does a round trip cost either way.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_pingpong.hip -o /tmp/xcd_pingpong && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void st16(unsigned long long* p, u32x4 v) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, 0, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ u32x4 ld16(const unsigned long long* p) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, AUX);
}

// workgroup `wa` and workgroup `wb` bounce a tag `rounds` times: every lane of wave 0 owns one 16-byte word (1 KB per
// direction, like a bundle's publish).  out[0] = wall-clock ticks (100 MHz) of the whole exchange in wa, out[1] = polls
// that found a stale tag, out[2] = 1 when a poll gave up (stale data served for ever: the policy is NOT coherent),
// out[3] / out[4] = XCC ids of the two workgroups.
template <int SAUX, int LAUX>
__global__ void k_pp(unsigned long long* buf, int wa, int wb, int rounds, unsigned long long* out) {
  const int me = (int)blockIdx.x;
  if (me != wa && me != wb) return;
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned long long* A = buf + 2 * lane;              // wa -> wb
  unsigned long long* B = buf + 4096 + 2 * lane;       // wb -> wa
  const unsigned long long xcc = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
  unsigned long long stale = 0, bad = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= rounds; ++i) {
    const u32x4 v = {(uint32_t)i, (uint32_t)i, (uint32_t)i, (uint32_t)i};
    if (me == wa) st16<SAUX>(A, v);
    const unsigned long long* src = (me == wa) ? B : A;
    for (int it = 0;; ++it) {
      const u32x4 w = ld16<LAUX>(src);
      if (w.x == (uint32_t)i && w.w == (uint32_t)i) break;
      ++stale;
      if (it > (1 << 16)) { bad = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (bad) break;
    if (me == wb) st16<SAUX>(B, v);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) {
    if (me == wa) { out[0] = t1 - t0; out[1] = stale; out[2] = bad; out[3] = xcc; }
    else { out[4] = xcc; out[5] = bad; }
  }
}

template <int SAUX, int LAUX>
static void run(const char* name, unsigned long long* buf, unsigned long long* out, int wa, int wb) {
  const int rounds = 2000;
  hipMemset(buf, 0, 8192 * 8);
  hipMemset(out, 0, 64);
  hipLaunchKernelGGL((k_pp<SAUX, LAUX>), dim3(64), dim3(256), 0, 0, buf, wa, wb, rounds, out);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("%-28s wg %2d <-> %2d (xcc %llu / %llu): %7.1f ns per round trip (two hand-offs), %5.2f stale polls per hand-off%s\n", name, wa, wb,
         h[3] & 15, h[4] & 15, h[0] * 10.0 / rounds, h[1] / (double)rounds, (h[2] || h[5]) ? "  ** GAVE UP: not coherent **" : "");
}

int main() {
  unsigned long long *buf, *out;
  hipMalloc(&buf, 8192 * 8);
  hipMalloc(&out, 64);
  for (int rep = 0; rep < 2; ++rep) {
    for (int pl = 0; pl < 2; ++pl) {
      const int wa = 3, wb = pl ? 3 + 8 : 4;
      printf("-- %s\n", pl ? "same XCD (b, b + 8)" : "neighbouring XCDs (b, b + 1)");
      run<16, 16>("store sc1 / load sc1", buf, out, wa, wb);
      run<17, 17>("store sc0 sc1 / load sc0 sc1", buf, out, wa, wb);
      run<1, 1>("store sc0 / load sc0", buf, out, wa, wb);
      run<0, 1>("store plain / load sc0", buf, out, wa, wb);
      run<0, 16>("store plain / load sc1", buf, out, wa, wb);
      run<1, 16>("store sc0 / load sc1", buf, out, wa, wb);
      run<16, 1>("store sc1 / load sc0", buf, out, wa, wb);
      run<0, 0>("store plain / load plain", buf, out, wa, wb);
      run<2, 2>("store nt / load nt", buf, out, wa, wb);
      run<3, 3>("store sc0 nt / load sc0 nt", buf, out, wa, wb);
    }
  }
  return 0;
}
