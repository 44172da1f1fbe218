// hipemu.h -- single-threaded CPU emulation of the HIP subset used by igmc_amd/csrc/*.hip.
//
// TEST INFRASTRUCTURE ONLY.  The build container has no GPU, so kernel *logic*
// (indexing, barriers, wave collectives, MFMA fragment maps) is exercised on the
// host by compiling the very same .hip sources with g++ and this header
// (-DIGMC_HIPEMU -include tools/hipemu/hipemu.h).  The resulting
// libigmc_emu.so is loaded ONLY by tests (tests/emu/), never by the product
// package, bench.py or smoke(): igmc_amd refuses to run without the real
// gfx950 library.
//
// Execution model: one workgroup at a time; every work-item is a fiber (x86-64: a 7-instruction register/stack
// switch -- glibc's swapcontext makes an rt_sigprocmask system call per switch, which was 40 % of the suite's time;
// other hosts: ucontext).
// Fibers run until they reach a collective (block barrier / wave collective),
// where they yield to a round-robin scheduler.  Because a fiber runs far ahead
// of its neighbours between collectives, any code that silently relies on
// wave-lockstep execution without an explicit wave_sync()/__syncthreads() FAILS
// here -- deliberately stricter than hardware.
#pragma once
#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3

namespace hipemu {

constexpr int WAVE = 64;
constexpr size_t STACK = 256 * 1024;

struct WaveState {
  int active = 0, arrived = 0;
  unsigned gen = 0;
  // sub-wave groups (width 4 / 8 / 16 / 32 collectives rendezvous only inside their group, as on
  // hardware where a group-uniform branch keeps the whole group active together)
  int g_arrived[4][16] = {};
  unsigned g_gen[4][16] = {};
  // exchange slots for collectives: [class 0: w4, 1: w8, 2: w16, 3: w32, 4: w64][double buffer][lane]
  uint64_t slot[5][2][WAVE];
  uint64_t part[2] = {0, 0};   // lanes that took part in the current / previous full-wave exchange
  float fa[WAVE], fb[WAVE], fc[WAVE][4];
  uint32_t ba[WAVE][4], bb[WAVE][4];     // bf16 MFMA operands (8 bf16 per lane each)
};

#if defined(__x86_64__)
#define HIPEMU_ASM_SWITCH 1
// saves the callee-saved registers of the SysV ABI on the current stack, stores the stack pointer in *from_sp and
// continues on to_sp (weak: the header is included by several translation units of one library)
extern "C" void hipemu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.weak hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");
struct Ctx {
  void* sp = nullptr;
};
#else
struct Ctx {
  ucontext_t uc;
};
#endif

struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  bool done = true;
  int tid = 0;      // work-item index inside its workgroup
  int blk = 0;      // resident-workgroup slot (Runtime::blocks)
};

struct BlockState {
  int nthreads = 0, active = 0, arrived = 0;
  unsigned gen = 0;
  std::vector<WaveState> waves;
};

// one RESIDENT workgroup: its barrier / wave state, its block index and its dynamic LDS
struct BlockCtx {
  BlockState blk;
  uint3_emu bidx{0, 0, 0};
  unsigned char* dyn_smem = nullptr;
  size_t dyn_cap = 0;
};

struct Runtime {
  Ctx sched;
  std::vector<Fiber> fibers;          // [resident slot][work-item]
  std::vector<BlockCtx> blocks;       // resident workgroups: one, unless the launch asks for clusters (below)
  int nres = 1;
  int cur = -1;
  dim3 bdim, gdim;
  // cluster launch (k_graph_step): the NEXT launch runs the workgroups b, b + co_stride, ..., (co_cs of them) of a 1-D
  // grid TOGETHER, round-robin over all their work-items, so that they can exchange data through global memory with
  // polling loops that call hipemu::yield(); consumed (reset to 1) by that launch
  // co_block > 0: the members of a cluster are b, b + co_stride, .. INSIDE blocks of co_block = co_cs * co_stride consecutive
  // workgroups (cluster (j, x) = workgroups j * co_block + x + c * co_stride, x < co_stride): the layout that puts the members
  // of a cluster on ONE XCD of a round-robin dispatch
  int co_cs = 1, co_stride = 0, co_block = 0;
  std::function<void()> body;
  long n_switch = 0;
};

inline Runtime& rt() {
  static Runtime r;
  return r;
}

inline void ctx_switch(Ctx& from, Ctx& to) {
#ifdef HIPEMU_ASM_SWITCH
  hipemu_switch(&from.sp, to.sp);
#else
  swapcontext(&from.uc, &to.uc);
#endif
}

inline void yield() {
  Runtime& r = rt();
  r.n_switch++;
  ctx_switch(r.fibers[r.cur].ctx, r.sched);
}

inline BlockCtx& cur_block() {
  Runtime& r = rt();
  return r.blocks[r.fibers[r.cur].blk];
}

inline void fiber_main() {
  Runtime& r = rt();
  r.body();
  Fiber& f = r.fibers[r.cur];
  f.done = true;
  // leaving the kernel: stop counting towards barriers / wave collectives
  BlockState& b = r.blocks[f.blk].blk;
  WaveState& w = b.waves[f.tid / WAVE];
  w.active--;
  if (w.active > 0 && w.arrived == w.active) { w.arrived = 0; w.gen++; }
  b.active--;
  if (b.active > 0 && b.arrived == b.active) { b.arrived = 0; b.gen++; }
  ctx_switch(f.ctx, r.sched);
  abort();                 // a finished fiber is never resumed
}

inline void fiber_prepare(Fiber& f, Ctx& link) {
#ifdef HIPEMU_ASM_SWITCH
  (void)link;
  // initial frame: six callee-saved registers (zero), then the address hipemu_switch "returns" to; the entry point
  // sees the stack as after a call (rsp = 16n + 8)
  uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                           // return address of fiber_main (never used)
  *--sp = (void*)(void (*)())fiber_main;
  for (int i = 0; i < 6; ++i) *--sp = nullptr;
  f.ctx.sp = (void*)sp;
#else
  getcontext(&f.ctx.uc);
  f.ctx.uc.uc_stack.ss_sp = f.stack;
  f.ctx.uc.uc_stack.ss_size = STACK;
  f.ctx.uc.uc_link = &link.uc;
  makecontext(&f.ctx.uc, (void (*)())fiber_main, 0);
#endif
}

// runs the r.nres resident workgroups (r.blocks[0 .. nres)) to completion, round-robin over all their work-items
inline void run_resident(int nthreads) {
  Runtime& r = rt();
  const int total = r.nres * nthreads;
  if ((int)r.fibers.size() < total) r.fibers.resize(total);
  const int nw = (nthreads + WAVE - 1) / WAVE;
  for (int bi = 0; bi < r.nres; ++bi) {
    BlockState& b = r.blocks[bi].blk;
    b.nthreads = nthreads;
    b.active = nthreads;
    b.arrived = 0;
    b.waves.assign(nw, WaveState());
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = r.fibers[bi * nthreads + t];
      if (!f.stack) f.stack = (char*)malloc(STACK);
      f.done = false;
      f.tid = t;
      f.blk = bi;
      b.waves[t / WAVE].active++;
      fiber_prepare(f, r.sched);
    }
  }
  int remaining = total;
  long idle_rounds = 0;
  while (remaining > 0) {
    for (int t = 0; t < total; ++t) {
      Fiber& f = r.fibers[t];
      if (f.done) continue;
      r.cur = t;
      ctx_switch(r.sched, f.ctx);
      if (f.done) remaining--;
    }
    if (++idle_rounds > 50000000L) { fprintf(stderr, "hipemu: deadlock suspected\n"); abort(); }
  }
  r.cur = -1;
}

inline void prepare_block(BlockCtx& b, uint3_emu idx, size_t shmem) {
  b.bidx = idx;
  if (shmem > b.dyn_cap) {
    free(b.dyn_smem);
    b.dyn_smem = (unsigned char*)aligned_alloc(64, (shmem + 63) / 64 * 64);
    b.dyn_cap = shmem;
  }
  if (shmem) memset(b.dyn_smem, 0xCD, shmem);  // poison: uninitialised LDS reads show up
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  Runtime& r = rt();
  if (r.cur != -1) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
  r.body = body;
  r.bdim = block;
  r.gdim = grid;
  const int cs = r.co_cs, stride = r.co_stride, cblock = r.co_block;
  r.co_cs = 1;
  r.co_stride = 0;
  r.co_block = 0;
  int nthreads = block.x * block.y * block.z;
  if (cs > 1) {
    // clusters: members g, g + stride, .. of a 1-D grid are resident together (NOTE: function-scope `__shared__`
    // variables are plain statics here and would be shared by them -- cluster kernels use dynamic LDS only)
    // (stride < 0: the members of cluster g are the consecutive workgroups g * cs .. g * cs + cs - 1)
    if (grid.y != 1 || grid.z != 1 || stride == 0) { fprintf(stderr, "hipemu: cluster launches need a 1-D grid\n"); abort(); }
    if ((int)r.blocks.size() < cs) r.blocks.resize(cs);
    if (cblock > 0) {
      for (unsigned base = 0; base < grid.x; base += (unsigned)cblock)
        for (int x = 0; x < stride; ++x) {
          r.nres = 0;
          for (int c = 0; c < cs; ++c) {
            const unsigned b = base + (unsigned)(x + c * stride);
            if (b < grid.x) prepare_block(r.blocks[r.nres++], uint3_emu{b, 0, 0}, shmem);
          }
          if (r.nres) run_resident(nthreads);
        }
      r.nres = 1;
      return;
    }
    const int ngroups = (stride < 0) ? ((int)grid.x + cs - 1) / cs : stride;
    for (int g = 0; g < ngroups; ++g) {
      r.nres = 0;
      for (int c = 0; c < cs; ++c) {
        const unsigned b = (stride < 0) ? (unsigned)(g * cs + c) : (unsigned)(g + c * stride);
        if (b < grid.x) prepare_block(r.blocks[r.nres++], uint3_emu{b, 0, 0}, shmem);
      }
      if (r.nres) run_resident(nthreads);
    }
    r.nres = 1;
    return;
  }
  if (r.blocks.empty()) r.blocks.resize(1);
  r.nres = 1;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        prepare_block(r.blocks[0], uint3_emu{bx, by, bz}, shmem);
        run_resident(nthreads);
      }
}

inline uint3_emu cur_tid() {
  Runtime& r = rt();
  unsigned t = r.fibers[r.cur].tid;
  uint3_emu o;
  o.x = t % r.bdim.x;
  o.y = (t / r.bdim.x) % r.bdim.y;
  o.z = t / (r.bdim.x * r.bdim.y);
  return o;
}

inline void syncthreads() {
  BlockState& b = cur_block().blk;
  unsigned gen = b.gen;
  b.arrived++;
  if (b.arrived == b.active) { b.arrived = 0; b.gen++; return; }
  while (b.gen == gen) yield();
}

inline WaveState& my_wave() {
  Runtime& r = rt();
  const Fiber& f = r.fibers[r.cur];
  return r.blocks[f.blk].blk.waves[f.tid / WAVE];
}
inline int lane_id() { return rt().fibers[rt().cur].tid % WAVE; }

inline void wave_rendezvous() {
  WaveState& w = my_wave();
  unsigned gen = w.gen;
  w.arrived++;
  if (w.arrived == w.active) { w.arrived = 0; w.gen++; return; }
  while (w.gen == gen) yield();
}

// number of lanes of [lo, lo+width) in this wave that are still running
inline int group_active(int lo, int width) {
  Runtime& r = rt();
  const Fiber& f = r.fibers[r.cur];
  const int nthreads = r.blocks[f.blk].blk.nthreads, base = f.blk * nthreads;
  int wave = f.tid / WAVE, n = 0;
  for (int l = lo; l < lo + width; ++l) {
    int t = wave * WAVE + l;
    if (t < nthreads && !r.fibers[base + t].done) n++;
  }
  return n;
}

// Exchange one 64-bit value per lane among the lanes of the caller's `width`-group; returns the
// wave's slot array for that width class (valid until the NEXT collective of the same class).
inline const uint64_t* wave_exchange(uint64_t v, int width = WAVE) {
  WaveState& w = my_wave();
  const int lane = lane_id();
  if (width >= WAVE) {
    int buf = w.gen & 1;
    if (w.arrived == 0) w.part[buf] = 0;
    w.part[buf] |= (1ull << lane);
    w.slot[4][buf][lane] = v;
    wave_rendezvous();
    return w.slot[4][buf];
  }
  if (width != 4 && width != 8 && width != 16 && width != 32) { fprintf(stderr, "hipemu: unsupported shuffle width %d\n", width); abort(); }
  const int cls = (width == 4) ? 0 : (width == 8) ? 1 : (width == 16) ? 2 : 3;
  const int grp = lane / width;
  unsigned gen = w.g_gen[cls][grp];
  int buf = gen & 1;
  w.slot[cls][buf][lane] = v;
  w.g_arrived[cls][grp]++;
  if (w.g_arrived[cls][grp] == group_active(grp * width, width)) {
    w.g_arrived[cls][grp] = 0;
    w.g_gen[cls][grp]++;
  } else {
    while (w.g_gen[cls][grp] == gen) yield();
  }
  return w.slot[cls][buf];
}

template <typename T> inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "T too large");
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  return u;
}
template <typename T> inline T from_bits(uint64_t u) {
  T v;
  memcpy(&v, &u, sizeof(T));
  return v;
}

}  // namespace hipemu

#define threadIdx (hipemu::cur_tid())
#define blockIdx (hipemu::cur_block().bidx)
#define blockDim (hipemu::rt().bdim)
#define gridDim (hipemu::rt().gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave collectives (all non-exited lanes of the wave must call them) ----
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu::lane_id();
  const uint64_t* s = hipemu::wave_exchange(hipemu::to_bits(v), width);
  int base = lane & ~(width - 1);
  return hipemu::from_bits<T>(s[base + (src & (width - 1))]);
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = hipemu::lane_id();
  const uint64_t* s = hipemu::wave_exchange(hipemu::to_bits(v), width);
  int base = lane & ~(width - 1);
  int src = (lane ^ mask) & (width - 1);
  return hipemu::from_bits<T>(s[base + src]);
}
template <typename T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int lane = hipemu::lane_id();
  const uint64_t* s = hipemu::wave_exchange(hipemu::to_bits(v), width);
  int rel = lane & (width - 1);
  int src = (rel + (int)delta < width) ? lane + (int)delta : lane;
  return hipemu::from_bits<T>(s[src]);
}
template <typename T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  int lane = hipemu::lane_id();
  const uint64_t* s = hipemu::wave_exchange(hipemu::to_bits(v), width);
  int rel = lane & (width - 1);
  int src = (rel - (int)delta >= 0) ? lane - (int)delta : lane;
  return hipemu::from_bits<T>(s[src]);
}
static inline unsigned long long __ballot(int pred) {
  hipemu::WaveState& w = hipemu::my_wave();
  const int buf = w.gen & 1;
  const uint64_t* s = hipemu::wave_exchange(pred ? 1ull : 0ull);
  // lanes that already left the kernel did not take part and contribute 0
  unsigned long long m = 0;
  for (int l = 0; l < hipemu::WAVE; ++l)
    if (((w.part[buf] >> l) & 1ull) && s[l]) m |= (1ull << l);
  return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) {
  hipemu::WaveState& w = hipemu::my_wave();
  const int buf = w.gen & 1;
  unsigned long long m = __ballot(pred);
  return (m & w.part[buf]) == w.part[buf];
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline void __builtin_amdgcn_s_setprio(int) {}      // (wave issue priority: no meaning under the emulation)
static inline int __builtin_amdgcn_readfirstlane(int v) {
  hipemu::WaveState& w = hipemu::my_wave();
  const int buf = w.gen & 1;
  const uint64_t* s = hipemu::wave_exchange(hipemu::to_bits(v));
  for (int l = 0; l < hipemu::WAVE; ++l)
    if ((w.part[buf] >> l) & 1ull) return hipemu::from_bits<int>(s[l]);
  return v;
}
// explicit intra-wave ordering point (LDS written by one lane, read by another of the same wave)
static inline void igmc_emu_wave_sync() { hipemu::wave_rendezvous(); }

// ---- atomics (sequential execution => plain read-modify-write) ----
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

// ---- MFMA f32 16x16x4 (CDNA4 fragment map, cdna_hip_programming.md section 3) ----
// A: lane l holds A[i=l&15][k=l>>4];  B: lane l holds B[k=l>>4][j=l&15];
// C/D: 4 regs per lane: col = l&15, row = (l>>4)*4 + reg.  D = fma chain over k = 0..3.
typedef float igmc_f32x4 __attribute__((ext_vector_type(4)));
static inline igmc_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, igmc_f32x4 c, int, int, int) {
  hipemu::WaveState& w = hipemu::my_wave();
  int lane = hipemu::lane_id();
  w.fa[lane] = a;
  w.fb[lane] = b;
  hipemu::wave_rendezvous();
  igmc_f32x4 d;
  int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[row + 16 * k], w.fb[col + 16 * k], acc);
    d[r] = acc;
  }
  hipemu::wave_rendezvous();  // nobody overwrites fa/fb before all lanes have read them
  return d;
}

// ---- MFMA f32 16x16x32 bf16 (gfx950): A: lane l holds A[i=l&15][k=8*(l>>4)+e], e=0..7 (dword e/2, low half first);
//      B: lane l holds B[k=8*(l>>4)+e][j=l&15]; C/D as the 16x16 f32 map.  Products of bf16 values are exact in f32;
//      the k-sum is formed in double and rounded once (the hardware's internal order is not architecturally defined:
//      tests compare at fp32 tolerances). ----
typedef uint32_t igmc_u32x4 __attribute__((ext_vector_type(4)));
static inline float hipemu_bf16_to_f32(uint32_t h) {
  uint32_t u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline igmc_f32x4 igmc_emu_mfma_16x16x32_bf16(igmc_u32x4 a, igmc_u32x4 b, igmc_f32x4 c) {
  hipemu::WaveState& w = hipemu::my_wave();
  int lane = hipemu::lane_id();
  for (int q = 0; q < 4; ++q) {
    w.ba[lane][q] = a[q];
    w.bb[lane][q] = b[q];
  }
  hipemu::wave_rendezvous();
  igmc_f32x4 d;
  int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (lane >> 4) * 4 + r;
    double acc = c[r];
    for (int kq = 0; kq < 4; ++kq)
      for (int e = 0; e < 8; ++e) {
        uint32_t wa = w.ba[row + 16 * kq][e >> 1], wb = w.bb[col + 16 * kq][e >> 1];
        float fa = hipemu_bf16_to_f32((e & 1) ? (wa >> 16) : (wa & 0xFFFFu));
        float fb = hipemu_bf16_to_f32((e & 1) ? (wb >> 16) : (wb & 0xFFFFu));
        acc += (double)fa * (double)fb;
      }
    d[r] = (float)acc;
  }
  hipemu::wave_rendezvous();
  return d;
}
// round-to-nearest-even f32 -> bf16 (the v_cvt_pk_bf16_f32 rounding), finite inputs
static inline uint32_t hipemu_f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

static inline uint32_t __float_as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float __uint_as_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---- host runtime subset ----
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
