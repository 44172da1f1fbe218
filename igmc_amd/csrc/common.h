// common.h -- internal definitions shared by the gfx950 kernels and the C-ABI host layer.
#pragma once
#include <stdint.h>
#include <stddef.h>

// A byte of the dense induced block of a subgraph (extract.hip: k_relm; read by the subgraph / dense-layer kernels):
// bits 0..3 = relation + 1 (0: no edge; R <= 15), bits 4 / 5 = keep flags of the two directions under edge dropout
#define IGMC_RELM_CODE 0x0Fu
#define IGMC_RELM_KF 4
#define IGMC_RELM_KT 5
#define IGMC_RELM_KEEP 0x30u

#ifdef IGMC_HIPEMU
// tools/hipemu/hipemu.h is force-included (CPU emulation for kernel-logic tests only)
#define IGMC_DYN_SMEM(name) unsigned char* name = hipemu::cur_block().dyn_smem
#define IGMC_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })
#define IGMC_WAVE_SYNC() igmc_emu_wave_sync()
#define IGMC_GROUP16_SYNC() ((void)__shfl(0, 0, 16))
typedef igmc_f32x4 f32x4;
#else
#include <hip/hip_runtime.h>
#define IGMC_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define IGMC_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (size_t)(shmem), (hipStream_t)(stream), __VA_ARGS__)
// intra-wave ordering point for LDS traffic between lanes of one wave
#define IGMC_WAVE_SYNC()                                     \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// ordering point for LDS traffic between lanes of one 16-lane group (same wave: a fence is enough)
#define IGMC_GROUP16_SYNC() IGMC_WAVE_SYNC()
#endif

#include "../../include/igmc_hip.h"
#include "../../include/igmc_rng.h"

#define IGMC_F 32          // hidden width
#define IGMC_BS_MAX 4      // num_bases supported by the kernels (reference Main.py:393 uses 4)
#define IGMC_BLOCK 256

// ------------------------------------------------------------------ device views
struct GraphDev {
  int n_users, n_items;
  const int32_t* u_ptr;   // [n_users+1]  CSR  user -> items, rows sorted by (relation, item)
  const int32_t* u_idx;   // [nnz] item ids
  const uint8_t* u_rel;   // [nnz] relation id = rating label (0..R-1)
  const int32_t* v_ptr;   // [n_items+1]  CSC  item -> users, columns sorted by (relation, user)
  const int32_t* v_idx;   // [nnz] user ids
  const uint8_t* v_rel;   // [nnz]
};

// One collated batch.  Nodes of graph g occupy [node_off[g], node_off[g+1]): its n_users[g]
// user nodes first (target user = first), then the item nodes (target item = first).
struct BatchDev {
  int32_t* node_off;     // [Bcap+1]
  int32_t* n_users;      // [Bcap]
  int32_t* n_items;      // [Bcap]
  int32_t* edge_cnt;     // [Bcap]   directed edges of graph g
  int32_t* edge_off;     // [Bcap+1]
  uint8_t* node_label;   // [Ncap]   2*dist (user) / 2*dist+1 (item)   (reference util_functions.py:245)
  int32_t* node_gid;     // [Ncap]   original user / item id
  int32_t* node_graph;   // [Ncap]   PyG `batch` vector
  int32_t* row_ptr;      // [Ncap+1] dst-sorted CSR over all nodes of the batch
  uint32_t* ecr;         // [Ecap]   source node (batch-global index, 24 bits) | relation id << 24
  uint16_t* ecode;       // [Ecap]   relation * num_labels + label(source)  (layer-0 table index)
  uint16_t* edst;        // [Ecap]   destination row of the entry, local to its subgraph (flat per-edge passes)
  uint8_t* eflag;        // [Ecap]   bit0: edge col->row kept, bit1: edge row->col kept
  float* y;              // [Bcap]
  int32_t* totals;       // [8]: 0 N, 1 E, 2 overflow flag, 3 B
  // per-graph scratch slots (capacity cap_u + cap_v each)
  int32_t* s_gid;        // [Bcap * slot]  ids by local index (users at 0.., items at cap_u..)
  uint8_t* s_lab;        // [Bcap * slot]
  int32_t* s_deg;        // [Bcap * slot]
  int32_t* t_list;       // [Bcap * slot]  BFS discovery order (temporary)
  uint8_t* t_dist;       // [Bcap * slot]
  uint8_t* relm;         // [Bcap * cap_u * cap_v] dense (user local, item local) -> relation+1 (0 = no edge); NULL when
                         // the slots are too large (uncapped extraction): then the CSC-scanning kernels are used
  int relm_ld;           // row stride of relm (cap_v rounded up to 4 so that every block is dword-aligned)
  uint8_t* relmT;        // [Bcap * cap_v * relmT_ld] the same bytes transposed (item local, user local): item-side rows for
                         // the dense per-layer kernels (denselayer path, slots of 129..256 nodes a side); NULL otherwise
  int relmT_ld;          // row stride of relmT (cap_u rounded up to 4)
  int max_rel;           // largest relation id of the rating graph
  int cap_u, cap_v, slot;
  int node_cap, edge_cap, graph_cap;
  int hop, max_nodes_per_hop, num_labels;
  int64_t* stamp;        // [4]: 0 = the `first` the node-set kernel of the batch in this arena resolved, 1 = the key of its
                         // edge dropout (-1 = none); compared by the tick of the step that consumes the arena (igmc_hip.h)
};

// ------------------------------------------------------------------ device-side step control (igmc_hip.h)
// Control words are read with agent-scope loads: they are written by the step's last kernel on one stream while kernels of
// the extraction chain on another stream read their own words of the same cache lines.
__device__ __forceinline__ int64_t igmc_ctrl_ld(const int64_t* p) {
#ifdef IGMC_HIPEMU
  return *p;
#else
  return (int64_t)__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// selector sel = q | (i << 1)  ->  offset into the link permutation of batch i of the group of parity q
__device__ __forceinline__ int igmc_ctrl_first(const int64_t* ctrl, int sel) {
  return (int)(igmc_ctrl_ld(ctrl + ((sel & 1) ? IGMC_CTRL_FIRST_ODD : IGMC_CTRL_FIRST)) +
               (int64_t)(sel >> 1) * igmc_ctrl_ld(ctrl + IGMC_CTRL_BATCH));
}
// key of the edge-dropout draws of the batch that starts at `first`: (epoch, batch index)
__device__ __forceinline__ uint64_t igmc_ctrl_drop_key(const int64_t* ctrl, int first) {
  const int64_t B = igmc_ctrl_ld(ctrl + IGMC_CTRL_BATCH);
  return ((uint64_t)igmc_ctrl_ld(ctrl + IGMC_CTRL_EPOCH) << 32) ^ (uint64_t)((int64_t)first / (B > 0 ? B : 1));
}

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ int igmc_lane() { return threadIdx.x & 63; }

// The tail kernels take the views of the batch and of the model BY VALUE: ~1.3 KB of kernel arguments, 20 cache lines that no
// one has touched before the launch.  The compiler loads a field right in front of its first use and waits for it -- one scalar
// round trip to memory per line, ONE AFTER THE OTHER down the prologue (k_finalize_ts: 2.5 k cycles before its first vector
// load left).  This requests one dword of every line of the first BYTES bytes back to back and waits once; the compiler's own
// loads then hit the scalar cache.  (k_graph_step2's 416 bytes gained nothing from it: profiles/r06_experiments.)
template <int BYTES>
__device__ __forceinline__ void igmc_kernarg_warm() {
#ifndef IGMC_HIPEMU
  typedef const __attribute__((address_space(4))) uint32_t* kp_t;
  kp_t p = (kp_t)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t v[(BYTES + 63) / 64];
#pragma unroll
  for (int i = 0; i < (BYTES + 63) / 64; ++i) v[i] = p[16 * i];
#pragma unroll
  for (int i = 0; i < (BYTES + 63) / 64; ++i) asm volatile("" ::"s"(v[i]));
#endif
}

// exclusive scan over the 256 threads of a block; *total = sum.  sm: >= 8 ints of LDS.
__device__ __forceinline__ int igmc_block_scan_excl(int v, int* total, int* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) sm[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) {
    int s = sm[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ int igmc_wave_sum_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__device__ __forceinline__ int igmc_block_sum_i(int v, int* sm) {
  v = igmc_wave_sum_i(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  int tot = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; ++w) tot += sm[w];
  __syncthreads();
  return tot;
}
