// Micro-benchmark: ONE-WAY latency of a tagged word from workgroup wa to workgroup wb (device wall clock, 10 ns ticks):
// issue of the store -> first poll that sees the tag, by store kind, poll kind and placement (same / neighbouring XCD).
// Also: store issue -> store acknowledged (vmcnt(0)) in the producer.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/xcd_oneway.hip -o tools/ubench/xcd_oneway.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

enum { ST_SC1 = 0, ST_SC01 = 1, ST_ATOMIC_AGENT = 2, ST_ATOMIC_SYS = 3, ST_ATOMIC_ADD = 4, ST_PLAIN = 5, ST_NT = 6 };
enum { LD_SC1 = 0, LD_SC01 = 1, LD_ATOMIC_AGENT = 2, LD_ATOMIC_OR = 3, LD_NT = 4, LD_SC0 = 5 };

template <int ST>
__device__ __forceinline__ void do_store(unsigned long long* p, unsigned long long v) {
  if (ST == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_SC01) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_NT) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (ST == ST_ATOMIC_AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (ST == ST_ATOMIC_SYS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if (ST == ST_ATOMIC_ADD) __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int LD>
__device__ __forceinline__ unsigned long long do_load(unsigned long long* p) {
  unsigned long long v;
  if (LD == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (LD == LD_SC01) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (LD == LD_NT) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (LD == LD_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (LD == LD_ATOMIC_AGENT) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else v = __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

template <int ST, int LD>
__global__ void k_ow(unsigned long long* word, unsigned long long* back, int wa, int wb, int rounds, unsigned long long* tiss,
                     unsigned long long* tack, unsigned long long* tseen, unsigned long long* npoll) {
  const int me = (int)blockIdx.x;
  if ((me != wa && me != wb) || threadIdx.x != 0) return;
  for (int i = 1; i <= rounds; ++i) {
    if (me == wa) {
      // wait until the consumer is polling for round i (it says so through `back`, read with an agent-scope atomic)
      while (__hip_atomic_load(back, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(i - 1)) __builtin_amdgcn_s_sleep(4);
      for (int d = 0; d < 40; ++d) __builtin_amdgcn_s_sleep(32);      // ~ 1 us: the consumer is certainly spinning
      const unsigned long long t0 = wall_clock64();
      do_store<ST>(word, ST == ST_ATOMIC_ADD ? 1ull : (unsigned long long)i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = wall_clock64();
      tiss[i] = t0;
      tack[i] = t1;
    } else {
      unsigned long long n = 0;
      for (;;) {
        const unsigned long long v = do_load<LD>(word);
        ++n;
        if (v == (unsigned long long)i) break;
        if (n > (1ull << 13)) break;
      }
      tseen[i] = wall_clock64();
      npoll[i] = n;
      __hip_atomic_store(back, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int ST, int LD>
static void ow(const char* name, unsigned long long* dev, int wa, int wb) {
  const int rounds = 100;
  hipMemset(dev, 0, 4 * 4096 * 8 + 4096);
  unsigned long long *word = dev, *back = dev + 64, *tiss = dev + 512, *tack = tiss + 4096, *tseen = tack + 4096, *npoll = tseen + 4096;
  hipLaunchKernelGGL((k_ow<ST, LD>), dim3(64), dim3(64), 0, 0, word, back, wa, wb, rounds, tiss, tack, tseen, npoll);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  static unsigned long long h[4 * 4096 + 512];
  hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
  double ack = 0, seen = 0, polls = 0, mx = 0;
  int bad = 0;
  for (int i = 11; i <= rounds; ++i) {
    const double s = (double)(long long)(h[512 + 2 * 4096 + i] - h[512 + i]) * 10.0;
    ack += (double)(h[512 + 4096 + i] - h[512 + i]) * 10.0;
    seen += s;
    if (s > mx) mx = s;
    polls += (double)h[512 + 3 * 4096 + i];
    if (h[512 + 3 * 4096 + i] > (1ull << 13)) bad = 1;
  }
  const int n = rounds - 10;
  printf("%-44s wg %2d -> %2d: store acknowledged %6.0f ns, seen by the poller %6.0f ns after issue (max %6.0f), %8.1f polls%s\n", name, wa, wb,
         ack / n, seen / n, mx, polls / n, bad ? "  ** never seen **" : "");
}

int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  unsigned long long* dev;
  hipMalloc(&dev, 4 * 4096 * 8 + 4096);
  for (int pl = 0; pl < 2; ++pl) {
    const int wa = 3, wb = pl ? 11 : 4;
    printf("-- %s\n", pl ? "same XCD (b, b + 8)" : "neighbouring XCDs (b, b + 1)");
    ow<ST_SC1, LD_SC1>("store sc1, load sc1", dev, wa, wb);
    ow<ST_SC01, LD_SC01>("store sc0 sc1, load sc0 sc1", dev, wa, wb);
    ow<ST_SC1, LD_ATOMIC_AGENT>("store sc1, atomic load (agent)", dev, wa, wb);
    ow<ST_ATOMIC_AGENT, LD_ATOMIC_AGENT>("atomic store (agent), atomic load (agent)", dev, wa, wb);
    ow<ST_ATOMIC_SYS, LD_ATOMIC_AGENT>("atomic store (system), atomic load (agent)", dev, wa, wb);
    ow<ST_ATOMIC_AGENT, LD_ATOMIC_OR>("atomic store (agent), returning atomic or", dev, wa, wb);
    ow<ST_NT, LD_SC1>("store nt, load sc1", dev, wa, wb);
    ow<ST_PLAIN, LD_SC1>("store plain, load sc1", dev, wa, wb);
    ow<ST_PLAIN, LD_NT>("store plain, load nt", dev, wa, wb);
    ow<ST_PLAIN, LD_SC0>("store plain, load sc0", dev, wa, wb);
    ow<ST_SC1, LD_NT>("store sc1, load nt", dev, wa, wb);
  }
  return 0;
}
