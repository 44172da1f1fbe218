// g2_prims.h -- the gfx950 primitives of graphstep2.hip: matrix-core step, byte-compare expansion, fast tanh, tagged
// exchange words (sc1 stores / compiler-tracked sc1 buffer loads, polled), launch-sequence and clock helpers -- each with
// its stand-in for the CPU emulation build (tools/hipemu, IGMC_HIPEMU: test infrastructure for the kernel LOGIC; what the
// stand-ins replace is verified by the GPU suite only).  The kernels in graphstep2.hip are written against these names and
// carry no build switch themselves.  (internal; included once, by graphstep2.hip, after g_g2_clk / g_g2_wg)
#pragma once

#ifdef IGMC_HIPEMU
#define G2_STAMP(k) do { } while (0)
#else
#define G2_STAMP(k)                                                                                        \
  do {                                                                                                     \
    if (a.timing && threadIdx.x == 0) {                                                                    \
      if (blockIdx.x == 0) g_g2_clk[k] = __builtin_readcyclecounter();                                     \
      else if (a.cs > 2 && (int)blockIdx.x == 8 * (a.cs / 2) && (k) < 40) g_g2_clk[64 + (k)] = __builtin_readcyclecounter(); \
    }                                                                                                      \
  } while (0)
#endif

// ... and of the one-launch dense layers (IGMC_DL_TIMING=<workgroup + 1>; igmc_debug_g2_clocks): k_dl_fwd -> slots 0..39,
// k_dl_bwd -> slots 40..127
#ifdef IGMC_HIPEMU
#define DL_STAMP(k) do { } while (0)
#else
#define DL_STAMP(k)                                                                               \
  do {                                                                                            \
    if (a.timing && threadIdx.x == 0 && (int)blockIdx.x == a.timing - 1 && (k) < 128)             \
      g_g2_clk[k] = __builtin_readcyclecounter();                                                 \
  } while (0)
#endif

// keeps per-lane index arithmetic INSIDE the phase it is used in (LLVM otherwise hoists hundreds of loop-invariant LDS
// addresses out of the layer loops and spills them)
#ifdef IGMC_HIPEMU
#define G2_OPAQUE(x) do { } while (0)
#else
#define G2_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef IGMC_HIPEMU
typedef __bf16 g2_bf16x8 __attribute__((ext_vector_type(8)));
#endif

__device__ __forceinline__ f32x4 g2_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
#ifdef IGMC_HIPEMU
  return igmc_emu_mfma_16x16x32_bf16(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g2_bf16x8, a), __builtin_bit_cast(g2_bf16x8, b), c, 0, 0, 0);
#endif
}

// four relm bytes (common.h: bits 0..3 = relation + 1, bits 4 / 5 = keep flags of the two directions) -> two dwords of bf16 pairs
// that are 1.0 where the byte's relation is r1 - 1 (and its keep bit is set)
template <bool FLAGS>
__device__ __forceinline__ void g2_expand4(uint32_t w, uint32_t r1, int keepbit, uint32_t& o01, uint32_t& o23) {
  uint32_t mk = w & (0x01010101u * IGMC_RELM_CODE);
  if (FLAGS) mk &= ((w >> keepbit) & 0x01010101u) * IGMC_RELM_CODE;
  const uint32_t t = mk ^ (0x01010101u * r1);
  const uint32_t eq = ~(t + 0x7F7F7F7Fu) & 0x80808080u;        // bit 7 of a byte set <=> the byte of t is zero (t <= 15)
  const uint32_t mask = (eq >> 7) * 0xFFu;                      // 0xFF per matching byte
#ifdef IGMC_HIPEMU
  o01 = ((mask & 0xFFu) ? 0x3F80u : 0u) | ((mask & 0xFF00u) ? 0x3F800000u : 0u);
  o23 = ((mask & 0xFF0000u) ? 0x3F80u : 0u) | ((mask & 0xFF000000u) ? 0x3F800000u : 0u);
#else
  o01 = __builtin_amdgcn_perm(mask, mask, 0x01010000u) & 0x3F803F80u;    // bytes [b0 b0 b1 b1]
  o23 = __builtin_amdgcn_perm(mask, mask, 0x03030202u) & 0x3F803F80u;    // bytes [b2 b2 b3 b3]
#endif
}

// ... the same in two steps: 0xFF in the bytes that match (kept resident), and the bf16 pairs of eight bytes from two masks
template <bool FLAGS>
__device__ __forceinline__ uint32_t g2_bytemask(uint32_t w, uint32_t r1, int keepbit) {
  uint32_t mk = w & (0x01010101u * IGMC_RELM_CODE);
  if (FLAGS) mk &= ((w >> keepbit) & 0x01010101u) * IGMC_RELM_CODE;
  const uint32_t t = mk ^ (0x01010101u * r1);
  const uint32_t eq = ~(t + 0x7F7F7F7Fu) & 0x80808080u;
  return (eq >> 7) * 0xFFu;
}
__device__ __forceinline__ u32x4 g2_mask_frag(uint32_t m0, uint32_t m1) {
  u32x4 f;
#ifdef IGMC_HIPEMU
  f[0] = ((m0 & 0xFFu) ? 0x3F80u : 0u) | ((m0 & 0xFF00u) ? 0x3F800000u : 0u);
  f[1] = ((m0 & 0xFF0000u) ? 0x3F80u : 0u) | ((m0 & 0xFF000000u) ? 0x3F800000u : 0u);
  f[2] = ((m1 & 0xFFu) ? 0x3F80u : 0u) | ((m1 & 0xFF00u) ? 0x3F800000u : 0u);
  f[3] = ((m1 & 0xFF0000u) ? 0x3F80u : 0u) | ((m1 & 0xFF000000u) ? 0x3F800000u : 0u);
#else
  f[0] = __builtin_amdgcn_perm(m0, m0, 0x01010000u) & 0x3F803F80u;
  f[1] = __builtin_amdgcn_perm(m0, m0, 0x03030202u) & 0x3F803F80u;
  f[2] = __builtin_amdgcn_perm(m1, m1, 0x01010000u) & 0x3F803F80u;
  f[3] = __builtin_amdgcn_perm(m1, m1, 0x03030202u) & 0x3F803F80u;
#endif
  return f;
}

__device__ __forceinline__ float g2_tanh(float x) {
#ifdef IGMC_HIPEMU
  return tanhf(x);
#else
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
#endif
}

// 16 bytes per lane, global -> LDS, without passing through registers (global_load_lds_dwordx4): the destination is the
// WAVE-UNIFORM base + 16 * lane, the source address is per lane.  Asynchronous: a pending one counts on the vector
// counter; the barrier (or wait) in front of the first read of the destination retires it.
template <int AUX = 0>          // 16 = sc1: past the CU's L1 (data another CU of the XCD has just written)
__device__ __forceinline__ void g2_glds16(const float4* src_wave, float4* lds_wave, int lane) {
#ifndef IGMC_HIPEMU
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_wave + lane),
                                   (__attribute__((address_space(3))) void*)lds_wave, 16, 0, AUX);
#else
  lds_wave[lane] = src_wave[lane];
#endif
}

// ---- the readout words of the subgraph kernel: 8-byte {f32, tag}, polled ------------------------------------------------
// (an ORDINARY 8-byte store: its readers -- g2_poll_f32, sc1 loads -- sit on the writer's XCD and find it in the shared L2
//  ~250 ns after issue; written through to memory with sc1 it took ~570 ns, profiles/r05_experiments/xcd_oneway.txt)
__device__ __forceinline__ void g2_pub_f32(unsigned long long* p, float v, uint32_t tag) {
#ifndef IGMC_HIPEMU
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
  uint32_t bits;
  memcpy(&bits, &v, 4);
  *p = ((unsigned long long)tag << 32) | (unsigned long long)bits;
#endif
}

// ---- plane exchange of the subgraph kernel (k_graph_step2) ------------------------------------------------------------
// The members of a cluster sit on ONE XCD (graphstep2.hip: workgroup -> (subgraph, member)), so the rows they exchange can
// stay in that XCD's L2: the producer writes bf16 term planes [term][feature][node] -- the very image the gather reads from
// LDS -- with ORDINARY stores (acknowledged by the L2 in ~120 ns; an sc1 store is written through to memory: ~430 ns, and a
// bulk poll of lines under write-through took microseconds, profiles/r05_experiments/xcd_*.txt), waits for their
// acknowledgement and raises ONE flag word per wave; a consumer wave polls the flags of the opposite side's bundles (sc1
// loads: past the CU's L1, served by the shared L2) and then copies the planes global -> LDS directly (global_load_lds).
// Region of one (exchange, subgraph, side): G2_PX_BYTES; planes at 0 (192 * kp bytes <= 26112), 64 flag words at G2_PX_FLAGS.
// The dense-layer kernels (<= 256 nodes a side: planes of up to 50688 bytes) use the same protocol in regions of DLX_PX_BYTES.
#define G2_PX_BYTES 32768
#define G2_PX_FLAGS 28672
#define DLX_PX_BYTES 65536
#define DLX_PX_FLAGS 61440
// the four rows node0 .. node0 + 3 (node0 % 4 == 0) of feature f: one 8-byte store per term
__device__ __forceinline__ void g2_publish_planes(unsigned char* px, int kp, int f, int node0, const float (&v)[4]) {
  uint32_t h0, m0, l0, h1, m1, l1;
  g2_split2(v[0], v[1], h0, m0, l0);
  g2_split2(v[2], v[3], h1, m1, l1);
  unsigned char* p = px + ((size_t)f * kp + node0) * 2;
  const size_t ts = (size_t)32 * kp * 2;
  *(uint2*)p = make_uint2(h0, h1);
  *(uint2*)(p + ts) = make_uint2(m0, m1);
  *(uint2*)(p + 2 * ts) = make_uint2(l0, l1);
}
// every store of the wave so far is in the L2; then its flag
__device__ __forceinline__ void g2_flag_raise(unsigned char* px, int slot, uint32_t tag, int lane, int flags_off = G2_PX_FLAGS) {
#ifndef IGMC_HIPEMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store((unsigned long long*)(px + flags_off) + slot, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
  if (lane == 0) ((unsigned long long*)(px + flags_off))[slot] = (unsigned long long)tag;
#endif
}
// the wave waits until the first n flags of the region carry this exchange's tag (lane i polls flag i)
__device__ __forceinline__ void g2_flags_wait(const unsigned char* px, int n, uint32_t tag, int lane, int* err, int flags_off = G2_PX_FLAGS) {
  const unsigned long long* fl = (const unsigned long long*)(px + flags_off);
  for (long it = 0;; ++it) {
#ifndef IGMC_HIPEMU
    const unsigned long long w = (lane < n) ? __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
#else
    const unsigned long long w = (lane < n) ? fl[lane] : (unsigned long long)tag;
#endif
    if (__ballot(w != (unsigned long long)tag) == 0ull) break;
    if (it > (1L << 20)) {
      *err = 1;
      break;
    }
#ifndef IGMC_HIPEMU
    __builtin_amdgcn_s_sleep(1);
#else
    hipemu::yield();
#endif
  }
}
// ... the same without writing past the image: the last piece is copied by the lanes that hold bytes of it only
__device__ __forceinline__ void g2_planes_load_exact(uint32_t* pl, const unsigned char* px, int bytes, int wave, int lane, int nwaves) {
  const int pieces = (bytes + 1023) >> 10;
  for (int c = wave; c < pieces; c += nwaves)
    if (c * 1024 + lane * 16 < bytes) g2_glds16<16>((const float4*)px + c * 64, (float4*)pl + c * 64, lane);
}
// planes of one side, global -> LDS, 1 KB pieces dealt to the waves of the workgroup (the LDS image is padded to whole pieces)
__device__ __forceinline__ void g2_planes_load(uint32_t* pl, const unsigned char* px, int kp, int wave, int lane, int nwaves) {
  const int pieces = (192 * kp + 1023) >> 10;
  for (int c = wave; c < pieces; c += nwaves) g2_glds16<16>((const float4*)px + c * 64, (float4*)pl + c * 64, lane);
}

// one 8-byte {f32, tag} word, polled
__device__ __forceinline__ float g2_poll_f32(const unsigned long long* p, uint32_t tag, int* err) {
  for (long it = 0;; ++it) {
#ifndef IGMC_HIPEMU
    const unsigned long long w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const unsigned long long w = *p;
#endif
    if ((uint32_t)(w >> 32) == tag) return __uint_as_float((uint32_t)w);
    if (it > (1L << 22)) {
      *err = 1;
      return 0.f;
    }
#ifndef IGMC_HIPEMU
    __builtin_amdgcn_s_sleep(2);
#else
    hipemu::yield();
#endif
  }
}

#ifdef IGMC_HIPEMU
#define G2_SCHED_BARRIER() do { } while (0)
#else
#define G2_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// ---- launch sequence number of the exchange tags, scheduling / wait points, clocks ---------------------------------------
__device__ __forceinline__ uint32_t g2_ld_seq(const int* gs_bar) {
#ifndef IGMC_HIPEMU
  return (uint32_t)__hip_atomic_load(gs_bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return (uint32_t)gs_bar[1];
#endif
}
// one arrival per workgroup; true in the LAST one, which has reset the counter and advanced the sequence number
__device__ __forceinline__ bool g2_last_workgroup_advances(int* gs_bar) {
#ifndef IGMC_HIPEMU
  if (__hip_atomic_fetch_add(gs_bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)gridDim.x - 1) return false;
  __hip_atomic_store(gs_bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(gs_bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
#else
  if (gs_bar[0]++ != (int)gridDim.x - 1) return false;
  gs_bar[0] = 0;
  gs_bar[1] += 1;
  return true;
#endif
}
__device__ __forceinline__ void g2_wait_vm0() {
#ifndef IGMC_HIPEMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ unsigned long long g2_wall_clock() {
#ifndef IGMC_HIPEMU
  return (unsigned long long)wall_clock64();
#else
  return 0ull;
#endif
}
__device__ __forceinline__ unsigned long long g2_xcc_id() {
#ifndef IGMC_HIPEMU
  return (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
#else
  return 0ull;
#endif
}
// device-side launch clock (igmc_profile_enable(2)): ts[0] = earliest workgroup start of the running launch, closed by the
// launch's last workgroup into ts[1] (sum of durations) / ts[2] (launches)
__device__ __forceinline__ void g2_clock_open(unsigned long long* ts) {
#ifndef IGMC_HIPEMU
  atomicMin(ts, (unsigned long long)wall_clock64());
#else
  (void)ts;
#endif
}
__device__ __forceinline__ void g2_clock_close(unsigned long long* ts, unsigned long long t1) {
#ifndef IGMC_HIPEMU
  const unsigned long long t0 = __hip_atomic_load(ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  atomicAdd(ts + 1, t1 - t0);
  atomicAdd(ts + 2, 1ull);
  __hip_atomic_store(ts, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  (void)ts; (void)t1;
#endif
}

// one multiply-add that stays ONE v_fmac_f32 (the compiler's packed-f32 chains gave run-to-run different sums on gfx950:
// see the note at its use in graphstep2.hip)
#ifdef IGMC_HIPEMU
#define DL_FMAC(acc_, a_, b_) ((acc_) += (a_) * (b_))
#else
#define DL_FMAC(acc_, a_, b_) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc_) : "v"(a_), "v"(b_))
#endif

// ---- exchange of the dense-layer kernels (k_dl_fwd / k_dl_bwd): regions of DLX_K nodes a side, 512-thread workgroups -----
#define DLX_K 256                 // nodes a side of the dense-layer kernels (an exchange region = 32 * DLX_K 8-byte words = DLX_PX_BYTES)
// launch sequence number of the exchange tags: advanced once per launch chain, after every workgroup has read it
__device__ __forceinline__ void dlx_seq_done(int* gs_bar, int self_seq) {
  if (threadIdx.x != 0 || !self_seq) return;
  (void)g2_last_workgroup_advances(gs_bar);
}

