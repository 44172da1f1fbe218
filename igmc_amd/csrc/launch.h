// launch.h -- host-side launcher declarations + optional per-kernel HIP-event timing (internal).
#pragma once
#include "model.h"
#include <string.h>

// extract.hip
size_t igmc_extract_smem_bytes(const GraphDev& g);
void igmc_launch_extract(const GraphDev& g, const BatchDev& b, const int32_t* link_u, const int32_t* link_v,
                         const float* link_y, const int32_t* link_idx, int first, int B, int replay,
                         double sample_ratio, uint64_t seed, uint64_t epoch, const int64_t* ctrl, int lean, void* stream);
void igmc_launch_extract_set(const GraphDev& g, const BatchDev* d_set, const BatchDev& b0, int count, const int32_t* link_u,
                             const int32_t* link_v, const float* link_y, const int32_t* link_idx, int sel0, int B,
                             double sample_ratio, uint64_t seed, const int64_t* ctrl, float drop_p, int force_undirected,
                             uint64_t drop_seed, void* stream);
void igmc_launch_emit(const BatchDev& b, int B, void* stream);
void igmc_launch_emit_nodes(const BatchDev& b, int B, void* stream);
void igmc_launch_load_nodes(const BatchDev& b, const int64_t* uoff, const int32_t* unodes, const uint8_t* udist,
                            const int64_t* voff, const int32_t* vnodes, const uint8_t* vdist, const float* link_y,
                            const int32_t* link_idx, int first, int B, const int64_t* ctrl, void* stream);
void igmc_launch_edge_flags(const BatchDev& b, float p, int force_undirected, uint64_t seed, uint64_t step,
                            const int64_t* ctrl, void* stream);
void igmc_launch_relm_flags(const BatchDev& b, void* stream);
void igmc_launch_relm_dropout(const BatchDev& b, int B, float p, int force_undirected, uint64_t seed, uint64_t step,
                              const int64_t* ctrl, void* stream);
void igmc_launch_tick(int64_t* ctrl, void* stream);
void igmc_launch_regroup(int64_t* ctrl, int M, int64_t first_cur, int64_t first_next, void* stream);
void igmc_launch_gate(int64_t* ctrl, int q, int gk_min, long long delay_ticks, int delay_always, long long timeout_ticks,
                      void* stream);
void igmc_launch_fill_u8(uint8_t* p, int64_t n, uint8_t v, void* stream);
int igmc_extract_prepare(size_t smem);

// model.hip
// auxiliary streams / events of a model: independent kernels of the step run as parallel branches (also when
// the step is being captured into a hipGraph: event waits become graph edges)
struct ModelAux {
  void* s1;      // Y products (needed only by the backward gathers)
  void* s2;      // weight-gradient products + lin1 weight gradient
  void* ev[8];
};
void igmc_launch_forward(const ModelDev& m, const ModelAux& ax, const BatchDev& b, const float* P, int B, int training,
                         int use_flags, const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult,
                         float* out, void* stream);
void igmc_launch_conv_forward(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                              void* stream);
void igmc_launch_conv_backward(const ModelDev& m, const BatchDev& b, const float* P, int B, int use_flags, float arr_coef,
                               float* grad, void* stream);
void igmc_launch_backward(const ModelDev& m, const ModelAux& ax, const BatchDev& b, const float* P, int B, int use_flags,
                          const float* gout, int from_err, float grad_scale, float mult, float drop_scale,
                          float arr_coef, float* grad, void* stream);
struct AdamTail;
// Data-parallel step: the REDUCED gradient sources of the step (relation-space tables + d att partials of the subgraph
// kernel, or the basis-space partial sums of the per-layer path; plus the lin1 / lin2 gradients in `grad`) are summed over
// the ranks between their reduction and the gradient / Adam kernel -- the step keeps its single-GPU kernels, the exchange
// is one grouped collective of two spans.  sum() returns 0 on success.
struct StepExchange {
  int (*sum)(void* user, float* a, int64_t na, float* b, int64_t nb, void* stream);
  void* user;
  const int* failed;      // device word the exchange raises when its sums are NOT in place (a peer never delivered): the
                          // gradient / Adam kernel behind it then leaves parameters, moments and step counters alone; or null
};
// 1 = a step on this arena keeps its gradient sources in exchangeable form (igmc_launch_loss_grad honours `xch`)
int igmc_step_exchange_inside(const ModelDev& m, const BatchDev& b, int B);
// returns 0, or the exchange's error code
int igmc_launch_loss_grad(const ModelDev& m, const ModelAux& ax, const BatchDev& b, float* P, int B, int use_flags,
                          const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult, float ARR,
                          float grad_scale, float arr_scale, float* out, float* grad, float* loss, const AdamTail* adam,
                          void* stream, const StepExchange* xch = nullptr, int* img_emitted = nullptr);
// (grad_scale 0: 1 / B; xch: see StepExchange -- only where igmc_step_exchange_inside() says so; *img_emitted = 1 when the
//  step's last kernel also left the weight images of the updated parameters in m.g2_w)
int igmc_launch_train_step(const ModelDev& m, const ModelAux& ax, const BatchDev& b, float* P, int B, int use_flags,
                           const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult, float ARR, float* out,
                           float* grad, float* m1, float* m2, float step_size, float inv_sqrt_bc2, float beta1,
                           float beta2, float eps, float wd, int64_t* ctrl, int* done, float* loss, double* total,
                           void* stream, float grad_scale = 0.f, const StepExchange* xch = nullptr, int* img_emitted = nullptr);
void igmc_launch_loss(const ModelDev& m, const BatchDev& b, float ARR, float* loss, void* stream);
void igmc_launch_sse(const BatchDev& b, const float* out, double* acc, int64_t* ctrl, void* stream);
int igmc_model_prepare(const ModelDev& m);
void igmc_launch_adam(float* p, const float* g, float* m1, float* m2, int64_t n, float step_size,
                      float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd, int64_t* ctrl, int tick,
                      void* stream);
void igmc_launch_finish(const ModelDev& m, const BatchDev& b, float* p, const float* g, float* m1, float* m2,
                        float step_size, float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd,
                        int64_t* ctrl, float ARR, float* loss, double* total, int use_flags, void* stream);

// graphstep2.hip: one cluster of workgroups per enclosing subgraph, relational aggregation on the matrix cores
int igmc_g2_xcd_ok();      // g2_compose.h: 1 = workgroups b and b + 8 of a launch share an XCD on the current device
struct G2Layout {      // LDS plan of graphstep2.hip, offsets in 4-byte words
  int kp, nsides, rmr, rmc, pside;
  int planes, ohp, lab, xo, hs, tile, hist, px, wreg, t0, att, head, words;
};
void igmc_launch_side_gather(const float* src, int S, const int32_t* link_idx, int first, int B, const int64_t* ctrl,
                             float* dst, void* stream);
int igmc_gs_grid(int B);
int igmc_gs_cluster(int B);
int igmc_gs_prepare();
int igmc_g2_eligible(const ModelDev& m, const BatchDev& b, int B, G2Layout* lay, int* cs_out);
// dense per-layer kernels (graphstep2.hip): slots of 129..256 nodes a side with a dense block + its transposed copy
int igmc_dl_eligible(const ModelDev& m, const BatchDev& b, int B);
int igmc_dl_grid(const BatchDev& b, int B);
void igmc_launch_g2_compose(const ModelDev& m, const float* P, void* stream);
void igmc_launch_dl_layer0(const ModelDev& m, const BatchDev& b, int B, int training, int use_flags, void* stream);
int igmc_dl_ts_eligible(const ModelDev& m, const BatchDev& b, int B);
int igmc_dl_fwd_eligible(const ModelDev& m, const BatchDev& b, int B);
int igmc_dl_bwd_eligible(const ModelDev& m, const BatchDev& b, int B);
int igmc_dl_wide(const ModelDev& m, const BatchDev& b, int B);
int igmc_dl_wide_gsplit(const ModelDev& m, const BatchDev& b, int B);
// (head != NULL: the launch runs the subgraphs' loss head itself -- no k_head_sub launch in front of it)
struct DlHead {
  const float* P;
  const uint8_t* inj_mask;
  uint64_t seed, step;
  float mult, grad_scale;
  float* out;
};
// (dense3: the sort-pool family's backward -- dPre_3 of every row from m.dpre[3], the readout gradient of layers 0..2 from m.dcat)
void igmc_launch_dl_bwd(const ModelDev& m, const BatchDev& b, int B, int use_flags, void* stream, const DlHead* head = nullptr,
                        int dense3 = 0);
void igmc_launch_dl_fwd(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                        float* zero_out, int self_seq, void* stream);
void igmc_launch_head_sub(const ModelDev& m, const BatchDev& b, const float* P, int B, const uint8_t* inj_mask, uint64_t seed,
                          uint64_t step, float mult, float grad_scale, float* out, void* stream);
void igmc_launch_dl_layer(const ModelDev& m, const BatchDev& b, const float* P, int B, int l, int bwd, int use_flags,
                          float* zero_out, void* stream, int tables = 0);
int igmc_launch_graph_step2(const ModelDev& m, const BatchDev& b, const float* P, int B, int training, int use_flags,
                             const G2Layout& lay, int cs, const uint8_t* inj_mask, uint64_t seed, uint64_t step, float mult,
                             float grad_scale, float* out, void* stream);
int igmc_g2_prepare();

// ---- per-kernel timing (HIP events on the launch stream; bench.py's roofline leg) ----
void igmc_prof_begin(const char* name, void* stream);
void igmc_prof_end(void* stream);
extern int g_igmc_prof_on;

#define IGMC_PLAUNCH(name, kern, grid, block, shmem, stream, ...)          \
  do {                                                                     \
    if (g_igmc_prof_on == 1) igmc_prof_begin(name, stream);                \
    IGMC_LAUNCH(kern, grid, block, shmem, stream, __VA_ARGS__);            \
    if (g_igmc_prof_on == 1) igmc_prof_end(stream);                        \
  } while (0)
