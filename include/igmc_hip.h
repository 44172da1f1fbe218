/* igmc_hip.h -- C ABI of libigmc_hip.so, the MI355X (gfx950) engine behind the IGMC hot path.
 *
 * The reference (muhanzhang/IGMC) has no FFI layer: its boundary is a Python call
 * surface.  Every entry point below therefore cites the reference Python
 * function(s) it replaces; the Python mirror in igmc_amd/{util_functions,models,
 * train_eval}.py binds them with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from igmc_last_error() (thread-local).
 *   - "d_" arguments are DEVICE pointers (HBM), "h_" arguments are HOST pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     are asynchronous on that stream unless documented otherwise.
 *   - handles own their HBM; borrowed buffers (parameters, gradients, outputs)
 *     stay owned by the caller.
 *   - no torch types anywhere in this interface.
 */
#ifndef IGMC_HIP_H
#define IGMC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct igmc_graph igmc_graph;   /* bipartite rating graph (CSR + CSC) resident in HBM */
typedef struct igmc_batch igmc_batch;   /* one extracted + collated batch of enclosing subgraphs */
typedef struct igmc_model igmc_model;   /* model geometry + activation / gradient workspace */

#define IGMC_HIDDEN 32        /* latent_dim = [32,32,32,32]  (reference Main.py:391) */
#define IGMC_NUM_LAYERS 4
#define IGMC_LIN1_OUT 128     /* reference models.py:25 */

const char* igmc_last_error(void);
int igmc_version(void);

/* ------------------------------------------------------------------ rating graph
 * Replaces SparseRowIndexer / SparseColIndexer construction
 * (reference util_functions.py:20-66, built in MyDataset/MyDynamicDataset.__init__
 * :72-73, :116-117).  Input = scipy CSR of the training rating matrix whose stored
 * values are rating-label + 1 (reference preprocessing.py:190-197); `h_rating`
 * holds label+1 as uint8 (1..R).  Rows are re-sorted by (relation, id) on the host
 * and both orientations are uploaded once. Synchronous. */
int igmc_graph_create(int n_users, int n_items, int64_t nnz,
                      const int32_t* h_indptr, const int32_t* h_indices, const uint8_t* h_rating,
                      int device, igmc_graph** out);
void igmc_graph_destroy(igmc_graph* g);
int64_t igmc_graph_hbm_bytes(const igmc_graph* g);

/* ------------------------------------------------------------------ batch arena
 * Capacity is fixed at creation from (max_graphs, hop, max_nodes_per_hop) and the
 * graph's sizes; nothing is allocated per step. */
int igmc_batch_create(const igmc_graph* g, int max_graphs, int hop, int max_nodes_per_hop,
                      igmc_batch** out);
void igmc_batch_destroy(igmc_batch* b);

/* Enclosing-subgraph extraction + labelling + collation for B links.
 * Replaces MyDynamicDataset.get -> subgraph_extraction_labeling + construct_pyg_graph
 * (reference util_functions.py:138-145, :208-297) and the PyG DataLoader collate
 * (reference train_eval.py:44-51) for a whole batch, in HBM.
 *   d_link_u/d_link_v/d_link_y : dataset-wide link arrays (user id, item id, rating VALUE
 *                                class_values[label], reference :247)
 *   d_link_idx                 : permutation of dataset positions; the batch is
 *                                d_link_idx[first .. first+B-1]  (NULL = identity)
 *   sample_ratio, max_nodes_per_hop(from create), hop : reference :222-229
 *   seed, epoch                : counter-based sampler key (seed, epoch, link position, hop, side);
 *                                a static dataset (MyDataset) passes a constant epoch.
 */
int igmc_extract_batch(const igmc_graph* g, igmc_batch* b,
                       const int32_t* d_link_u, const int32_t* d_link_v, const float* d_link_y,
                       const int32_t* d_link_idx, int first, int B,
                       double sample_ratio, uint64_t seed, uint64_t epoch, void* stream);

/* Parity mode: node sets are supplied (e.g. by the reference / oracle), only the
 * induced-edge, labelling and collation stages run on the GPU.
 *   h_unodes/h_vnodes : concatenated global ids, target first within each graph
 *   h_udist/h_vdist   : hop distance per node;  h_uoff/h_voff : B+1 offsets.  Synchronous. */
int igmc_extract_batch_replay(const igmc_graph* g, igmc_batch* b, int B,
                              const int32_t* h_unodes, const uint8_t* h_udist, const int32_t* h_uoff,
                              const int32_t* h_vnodes, const uint8_t* h_vdist, const int32_t* h_voff,
                              const float* h_y, void* stream);

/* Static dataset (reference MyDataset + links2subgraphs, util_functions.py:69-110, :148-205: subgraphs extracted once
 * and cached in <root>/processed/data.pt).  The native cache = the node sets of every link with their hop distances,
 * packed in HBM: d_uoff / d_voff [n_links + 1] offsets into d_unodes / d_vnodes (global ids, target first) and d_udist /
 * d_vdist.  The batch d_link_idx[first .. first + B - 1] is rebuilt from it: induced edges, labels and collation run on
 * the GPU as in igmc_extract_batch (asynchronous; honours the control block and the lean mode). */
int igmc_extract_batch_cached(const igmc_graph* g, igmc_batch* b,
                              const int64_t* d_uoff, const int32_t* d_unodes, const uint8_t* d_udist,
                              const int64_t* d_voff, const int32_t* d_vnodes, const uint8_t* d_vdist,
                              const float* d_link_y, const int32_t* d_link_idx, int first, int B, void* stream);

/* A GROUP of batches extracted in one launch per stage (no reference counterpart: the reference extracts subgraph by subgraph
 * in worker processes, util_functions.py:138-145).  Extraction is a dependent chain per link, so its throughput is the number
 * of links in flight: the batches sel0 + 2 i (i = 0 .. count-1; selectors q | (i << 1) of the device-side step control) go
 * into the arenas of `set` with the kernels of igmc_extract_batch launched ONCE over all of them, followed -- drop_p > 0 --
 * by their edge dropout (igmc_batch_edge_dropout with step = the batch's selector).  Every arena of the set must belong to
 * `g`, share one geometry, carry dense induced blocks, be lean (igmc_batch_set_lean), have the same control block attached
 * and no side-feature source; anything else is an error (callers then extract arena by arena). */
typedef struct igmc_batch_set igmc_batch_set;
int igmc_batch_set_create(igmc_batch* const* batches, int count, igmc_batch_set** out);
void igmc_batch_set_destroy(igmc_batch_set* s);
int igmc_extract_group(const igmc_graph* g, igmc_batch_set* s, int count,
                       const int32_t* d_link_u, const int32_t* d_link_v, const float* d_link_y,
                       const int32_t* d_link_idx, int sel0, int B, double sample_ratio, uint64_t seed,
                       float drop_p, int force_undirected, uint64_t drop_seed, void* stream);

/* Edge dropout (reference models.py:193-198 -> PyG dropout_adj): fills the per-entry keep
 * flags (bit0: edge col->row kept, bit1: edge row->col kept) from a counter-based hash of
 * (seed, step, graph, user id, item id, direction).  p = drop probability. */
int igmc_batch_edge_dropout(igmc_batch* b, float p, int force_undirected,
                            uint64_t seed, uint64_t step, void* stream);
/* Parity mode: inject the flags (host array, one byte per CSR entry). Synchronous. */
int igmc_batch_set_edge_flags(igmc_batch* b, const uint8_t* h_flags, int64_t n);
int igmc_batch_clear_edge_flags(igmc_batch* b);   /* no dropout: every edge kept */

/* Batch introspection (synchronises the stream; used by the Python Data view and tests). */
typedef struct igmc_batch_info {
  int32_t num_graphs, num_nodes, num_edges, overflow;
  int32_t node_capacity, edge_capacity, num_labels, hop;
} igmc_batch_info;
int igmc_batch_get_info(const igmc_batch* b, igmc_batch_info* out, void* stream);
/* Copies to host (any pointer may be NULL):
 *  node_off[B+1], n_users[B], node_label[N](u8), node_gid[N], node_graph[N], row_ptr[N+1],
 *  col[E], erel[E](u8), elab[E](u8), eflag[E](u8), y[B] */
int igmc_batch_download(const igmc_batch* b, int32_t* node_off, int32_t* n_users, uint8_t* node_label,
                        int32_t* node_gid, int32_t* node_graph, int32_t* row_ptr, int32_t* col,
                        uint8_t* erel, uint8_t* elab, uint8_t* eflag, float* y, void* stream);
/* Device pointers of the collated batch (for zero-copy views): index by IGMC_BUF_*. */
enum { IGMC_BUF_NODE_OFF = 0, IGMC_BUF_N_USERS, IGMC_BUF_NODE_LABEL, IGMC_BUF_NODE_GID,
       IGMC_BUF_NODE_GRAPH, IGMC_BUF_ROW_PTR, IGMC_BUF_ECR, IGMC_BUF_ECODE,
       IGMC_BUF_EFLAG, IGMC_BUF_Y, IGMC_BUF_TOTALS, IGMC_BUF_COUNT };
void* igmc_batch_device_ptr(const igmc_batch* b, int which);
/* Optional side features of the two target nodes (reference util_functions.py:250-253,
 * models.py:208-209): d_feat[B, n_side] fp32, borrowed for the next forward. */
int igmc_batch_set_side_features(igmc_batch* b, const float* d_feat, int n_side);
/* The same, resolved on the device: d_side_all[n_links, n_side] holds, for EVERY link of the dataset, the feature
 * rows of its two target nodes ([u_features[link_u] | v_features[link_v]], reference MyDynamicDataset.get ->
 * util_functions.py:272-275); every following igmc_extract_batch on this arena gathers the rows of the batch's
 * links (same d_link_idx / first / control-block indexing as the extraction) into an arena-owned buffer that the
 * model then reads -- nothing happens on the host per step, so the step stays hipGraph-capturable.  NULL unbinds. */
int igmc_batch_bind_side_source(igmc_batch* b, const float* d_side_all, int n_side);
/* Lean extraction (arenas with a per-hop cap only; ignored otherwise): igmc_extract_batch stops after the node sets,
 * labels, ratings and the dense induced blocks -- everything the matrix-core subgraph kernel reads -- and the collated
 * CSR (reference construct_pyg_graph + Batch.from_data_list, util_functions.py:280-297) is emitted on demand by the
 * calls that need it (get_info / download / edge flags / model calls that run the per-layer kernels).  The training
 * loop sets it when igmc_model_dense_path() says the model will take the dense path for this arena and batch size. */
int igmc_batch_set_lean(igmc_batch* b, int lean);
/* The arena holds a batch of B subgraphs that a REPLAYED launch extracted (a captured igmc_extract_batch / igmc_extract_group
 * leaves no host-side trace when its hipGraph is replayed): the model calls take the batch size of an arena from its last
 * extraction call on the host, which may be an older one with another size (the ragged last batch of an epoch).  Callers
 * that consume a prefetched arena outside the graph that filled it say so here.  No reference counterpart. */
int igmc_batch_assume_size(igmc_batch* b, int B);
/* Keep the transposed copy of the dense induced blocks too (arenas get one by themselves when a side exceeds 128 nodes):
 * the item-side operand of the dense-layer kernels, for models the subgraph kernel does not take although the blocks exist
 * (sort-pool readout, side features) -- their conv layers then run on the matrix cores instead of walking CSR rows.
 * No-op without dense blocks; a batch already in the arena is dropped (extract again).  No reference counterpart. */
int igmc_batch_want_transposed(igmc_batch* b);

/* ------------------------------------------------------------------ model
 * Flat fp32 parameter buffer layout (offsets in floats; query with igmc_param_offset):
 *   for l in 0..3:  conv{l}.basis [Bs, Fin_l, 32]   conv{l}.root [Fin_l, 32]
 *                   conv{l}.bias [32]               conv{l}.att [R, Bs]
 *   lin1.weight [128, 256+n_side]  lin1.bias [128]  lin2.weight [1,128]  lin2.bias [1]
 * with Fin_0 = 2*hop+2, Fin_l = 32.  Shapes = PyG-1.4.2 RGCNConv / torch.nn.Linear, i.e. the
 * reference state_dict keys convs.{l}.{basis,att,root,bias}, lin1.*, lin2.* (Main.py:36-45). */
enum { IGMC_P_BASIS = 0, IGMC_P_ROOT, IGMC_P_BIAS, IGMC_P_ATT,
       IGMC_P_LIN1_W, IGMC_P_LIN1_B, IGMC_P_LIN2_W, IGMC_P_LIN2_B };
int igmc_model_create(int device, int num_relations, int num_bases, int num_labels /*2*hop+2*/,
                      int n_side_features, int max_nodes, int max_edges, int max_graphs,
                      igmc_model** out);
void igmc_model_destroy(igmc_model* m);
int64_t igmc_param_count(const igmc_model* m);
/* offset (floats) and element count of a tensor; layer is ignored for lin1/lin2. */
int64_t igmc_param_offset(const igmc_model* m, int layer, int which, int64_t* count);

/* IGMC.forward (reference models.py:190-217): 4x (RGCNConv + tanh), centre-node readout,
 * lin1+ReLU+dropout(0.5)+lin2, * multiply_by.  Writes d_out[B].
 *   training      : 0 = eval (no dropout), 1 = train
 *   d_lin_mask    : optional injected keep-mask for the 0.5 dropout, uint8 [B,128] (parity);
 *                   NULL = draw from the counter-based hash (seed, step)
 *   use_edge_flags: 1 = honour the batch's edge keep flags (training with adj_dropout>0) */
int igmc_model_forward(igmc_model* m, const float* d_params, const igmc_batch* b,
                       int training, int use_edge_flags, const uint8_t* d_lin_mask,
                       uint64_t seed, uint64_t step, float multiply_by,
                       float* d_out, void* stream);
/* Backward of the last igmc_model_forward(training=1) given d_gout[B] = dLoss/d_out.
 * Accumulates nothing: d_grad (flat, same layout as params) is overwritten. */
int igmc_model_backward(igmc_model* m, const float* d_params, const igmc_batch* b,
                        const float* d_gout, float multiply_by, float* d_grad, void* stream);

/* One optimisation step's loss + gradient (reference train_eval.py:158-175):
 * (d_loss may be NULL when igmc_step_finish will produce it.)
 * forward, loss = mse_loss(out, y) (mean over the B graphs * loss_scale) + ARR * sum_l sum_r
 * ||W_l[r+1]-W_l[r]||^2, backward.  d_loss[0] = loss, d_loss[1] = sum of squared errors.
 * `grad_scale` multiplies the data-term gradient (1/B for the reference; 1/(global B) under
 * data parallelism, where `arr_scale` = 1/world_size keeps the all-reduced ARR term exact). */
int igmc_model_loss_grad(igmc_model* m, const float* d_params, const igmc_batch* b,
                         int use_edge_flags, const uint8_t* d_lin_mask, uint64_t seed, uint64_t step,
                         float multiply_by, float ARR, float grad_scale, float arr_scale,
                         float* d_out, float* d_grad, float* d_loss, void* stream);

/* Fused Adam over the flat buffer (reference train_eval.py:54,177: torch.optim.Adam,
 * betas (0.9,0.999), eps 1e-8, weight_decay added to the gradient).  `step` is 1-based. */
int igmc_adam_step(float* d_params, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq,
                   int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                   float weight_decay, void* stream);

/* ------------------------------------------------------------------ device-side step control
 * A hipGraph replay cannot change kernel arguments, so the per-step scalars live in HBM instead:
 * d_ctrl is an int64[IGMC_CTRL_WORDS] device buffer (slots 8..14 hold doubles, bit-cast):
 *   [0] step   [1] cursor_0   [2] epoch   [3] adam_t   [4] batch size B   [5] internal arrival counter (keep 0)
 *   [6] cursor_1   [7] k = steps done in this epoch
 *   [8] lr  [9] beta1  [10] beta2  [11] eps  [12] weight_decay   [13] lr/(1-beta1^t)  [14] 1/sqrt(1-beta2^t)
 *   [15] M = steps per GROUP (0 is read as 1)   [16] gk = steps done in the current group   [17] gq = parity of the
 *   current group   [18] sync_err (bit 1: a step consumed an arena whose stamp is not the batch of its cursor; bit 2:
 *   ... whose edge-dropout key is not that batch's)
 * Steps run in GROUPS of M: the batches of a group sit in M arenas (one set per group parity), extracted while the
 * previous group trained.  cursor_q = offset into the link permutation of the FIRST batch of the group of parity q; batch i
 * of that group starts at cursor_q + i * B.  A tick (end of a step, igmc_ctrl_tick or the step's last kernel) does
 * step += 1, k += 1, adam_t += 1, slots 13/14 recomputed, gk += 1 and -- when gk reaches M -- cursor_gq += 2 * M * B,
 * gk = 0, gq ^= 1: the cursor a concurrent prefetch of the NEXT group reads (cursor_{1-gq}) is never written while it may
 * be read.  M = 1 is the plain even / odd double buffer.
 * Once attached (igmc_batch_set_ctrl), the host `first` of igmc_extract_batch / igmc_extract_batch_cached and the host
 * `step` of igmc_batch_edge_dropout become a SELECTOR sel = q | (i << 1): first = cursor_q + i * B, epoch = ctrl.epoch,
 * dropout key = (epoch, first / B); the forward's MLP dropout uses ctrl.step.  Every extraction stamps its arena with the
 * `first` (and every edge dropout with the key) it resolved; the tick of the step that consumed the arena compares the
 * stamp with cursor_gq + gk * B and raises sync_err on a mismatch (igmc_model_check reports it).  NULL detaches. */
enum { IGMC_CTRL_STEP = 0, IGMC_CTRL_FIRST = 1, IGMC_CTRL_EPOCH = 2, IGMC_CTRL_ADAM_T = 3, IGMC_CTRL_BATCH = 4,
       IGMC_CTRL_DONE = 5, IGMC_CTRL_FIRST_ODD = 6, IGMC_CTRL_K = 7,
       IGMC_CTRL_LR = 8, IGMC_CTRL_BETA1 = 9, IGMC_CTRL_BETA2 = 10, IGMC_CTRL_EPS = 11, IGMC_CTRL_WD = 12,
       IGMC_CTRL_STEP_SIZE = 13, IGMC_CTRL_INV_SQRT_BC2 = 14, IGMC_CTRL_GROUP = 15, IGMC_CTRL_GK = 16,
       IGMC_CTRL_GQ = 17, IGMC_CTRL_SYNC_ERR = 18, IGMC_CTRL_GATE_TIMEOUTS = 19, IGMC_CTRL_WORDS = 24 };
int igmc_ctrl_tick(int64_t* d_ctrl, void* stream);
/* Starts a new group at the current position (no reference counterpart): M steps per group, gk = 0, gq = 0,
 * cursor_0 = first_cur (the batch of the next step), cursor_1 = first_next (first batch of the group after it). */
int igmc_ctrl_regroup(int64_t* d_ctrl, int M, int64_t first_cur, int64_t first_next, void* stream);
/* Pacing gate for work on ANOTHER stream (no reference counterpart): a one-wave kernel on `stream` that ends once gk_min
 * steps of the running group of parity q are done (ctrl.gk >= gk_min), or that group is over (ctrl.gq != q), or timeout_us
 * microseconds have passed -- whichever comes first; when it had to wait for the step counter (or delay_always != 0) it ends
 * delay_us (<= 1000) later, so that the step which has just begun has its workgroups on the chip before whatever is queued
 * behind the gate on `stream` (the extraction of the next group's batches) starts beside it: an extraction launch dispatched
 * TOGETHER with the subgraph kernel costs that step ~14 us, one dispatched 10 us behind it ~3 (profiles/r05_experiments).
 * No graph edge leaves the step chain.  A hint, not a dependency: the work behind the gate must be correct whenever it runs.
 * A gate that gives up counts itself in ctrl[IGMC_CTRL_GATE_TIMEOUTS]: where the two streams are not served concurrently (a
 * profiler or AMD_SERIALIZE_KERNEL serialising dispatches, both streams on one hardware queue) every gate would cost its
 * timeout -- the caller reads the counter and goes back to pacing by graph edges (StepGraph.check). */
int igmc_ctrl_gate(int64_t* d_ctrl, int q, int gk_min, double delay_us, int delay_always, double timeout_us, void* stream);
int igmc_batch_set_ctrl(igmc_batch* b, const int64_t* d_ctrl);
int igmc_model_set_ctrl(igmc_model* m, const int64_t* d_ctrl);
/* Adam + loss/epoch-total epilogue in ONE launch (the step's last kernel): updates d_params like
 * igmc_adam_step, writes d_loss[0..1] like igmc_model_loss_grad (call that one with d_loss = NULL), adds
 * loss*num_graphs to d_total[0] (reference train_eval.py:176), and -- when d_ctrl is given -- takes the Adam
 * scalars from the control block and ADVANCES it for the next step (the tick rides in this kernel). */
int igmc_step_finish(igmc_model* m, const igmc_batch* b, float* d_params, const float* d_grad,
                     float* d_exp_avg, float* d_exp_avg_sq, float ARR, float* d_loss, double* d_total,
                     int64_t* d_ctrl, int64_t step, float lr, float beta1, float beta2, float eps,
                     float weight_decay, void* stream);
/* The whole single-GPU optimisation step on an extracted batch (reference train_eval.py:158-177: forward, loss,
 * backward, optimizer.step) in the minimum number of launches: the last kernel applies Adam to the parameters
 * whose gradients it finalises, emits d_loss[0..1], adds loss*num_graphs to d_total[0] and (with d_ctrl) advances
 * the control block.  grad_scale = 1/B.  Data-parallel runs use igmc_model_loss_grad + all-reduce +
 * igmc_step_finish instead. */
int igmc_train_step(igmc_model* m, float* d_params, const igmc_batch* b, int use_edge_flags,
                    const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float multiply_by, float ARR,
                    float* d_out, float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_loss,
                    double* d_total, int64_t* d_ctrl, int64_t adam_t, float lr, float beta1, float beta2,
                    float eps, float weight_decay, void* stream);

/* Weight images.  The subgraph kernel and the dense per-layer kernels read each conv layer's weights as staged images
 * (W_r = sum_b att[r,b] basis_b per relation, split into bf16 terms in MFMA fragment order; the layer-0 table), composed
 * from d_params by a small kernel at the start of every forward / step.  igmc_train_step[_dp]'s last kernel (gradient +
 * Adam) ALSO writes the images of the parameters it has just updated, and an evaluation leaves the images of its
 * parameters behind -- so a caller that knows d_params has not been touched since the PREVIOUS call on this model (the
 * next step of a training loop, the next batch of an evaluation) says so, and that call starts with the subgraph kernel:
 * one launch less per step.  One-shot: applies to the next igmc_model_forward / igmc_model_loss_grad / igmc_train_step /
 * igmc_train_step_dp on `m` only, and only if the library itself left the images of the same d_params pointer behind
 * (else it composes as usual).  Never assert it after changing the parameters by other means (igmc_adam_step, a
 * checkpoint load, a broadcast).  Reference: there is none -- PyG's RGCNConv forms W_r on every forward (models.py:200). */
int igmc_model_weights_unchanged(igmc_model* m, int on);
int igmc_adam_step_ctrl(float* d_params, const float* d_grad, float* d_exp_avg, float* d_exp_avg_sq,
                        int64_t n, const int64_t* d_ctrl, void* stream);

/* ------------------------------------------------------------------ gradient exchange (data parallelism)
 * The reference has no distributed code (single process, single device: train_eval.py:20); the hot path shards by LINKS
 * (SURVEY.md 8(e)): one process per GPU, graph + parameters replicated, and the only exchange of a step is ONE sum
 * all-reduce of the flat gradient buffer (49 233 floats at R = 5) between the gradient kernels and the Adam kernel.
 * It is an RCCL ncclAllReduce enqueued on the caller's stream -- capturable into the step's hipGraph like any kernel --
 * on a communicator owned by this library (RCCL is resolved with dlopen("librccl.so.1") at the first use: a process
 * that never creates a communicator never loads it).
 *   igmc_comm_unique_id : rank 0 draws the 128-byte RCCL id; the caller hands it to the other ranks (any transport)
 *   igmc_comm_create    : collective over the `world` ranks; `device` = this rank's GPU.  world == 1 is valid.
 *   igmc_comm_info      : rank / size AS SEEN BY RCCL (bench.py checks them against the launcher's)
 *   igmc_allreduce_grads: d_flat_grad[i] = scale * sum over ranks of d_flat_grad[i], in place, asynchronous on `stream`.
 *                         The gradient kernels already scale by 1/(B * world) (igmc_model_loss_grad), so scale = 1.
 *   igmc_comm_create_host: a communicator whose sum is the CALLER's: fn(user, d_buf, n, stream) must leave the sum over the
 *                         `world` ranks in d_buf (in place, ordered on `stream`), return 0 on success.  For a host that already
 *                         owns a process group (torch.distributed, MPI): the library's steps then run on it unchanged.
 *   igmc_train_step_dp  : igmc_train_step of a data-parallel job, the exchange INSIDE the step: the step's reduced gradient
 *                         sources (the subgraph kernel's relation-space tables, or the per-layer path's basis-space sums, plus
 *                         the lin1 / lin2 gradients: about the size of the flat gradient) are summed over the ranks between
 *                         their reduction and the gradient / Adam kernel, as ONE grouped collective -- the kernels of the
 *                         single-GPU step and nothing else; arenas whose step keeps no such form (generic sequence) form the
 *                         flat gradient, all-reduce it and run the Adam kernel.  Loss terms are scaled by 1/(B * world), the
 *                         ARR term is added by every rank after the exchange.  comm == NULL: one rank (= igmc_train_step).
 *                         d_loss / d_total stay per rank (this rank's batches), as in igmc_train_step.
 *   igmc_comm_peer_alloc / igmc_comm_peer_connect: a communicator whose sum is a ONE-SHOT all-reduce over peer-mapped
 *                         buffers (the ranks of ONE node): every rank publishes its span into its own device buffer as
 *                         {value, tag} words and reads every rank's words in rank order -- one launch, about one xGMI round
 *                         trip, bit-identical replicas, capturable.  _alloc creates the rank's buffer (slots of `max_floats`)
 *                         and returns its 64-byte IPC handle; the caller hands the handles of all ranks (world x 64 bytes,
 *                         rank order, any transport) to _connect.  Used like any other communicator afterwards.
 *                         A one-rank peer communicator launches nothing (the spans are the sums).  The polls are bounded by
 *                         WALL-CLOCK time (IGMC_PEER_TIMEOUT_S, default 60 s): a word that never arrives raises a sticky
 *                         device word -- the spans of that launch are then PARTLY summed (elements finished before the
 *                         time-out hold sums, the others their local values; never a foreign or torn value) and must not be
 *                         used -- and the gradient / Adam kernel of igmc_train_step_dp behind the exchange touches no
 *                         parameter, moment or step counter.  After a bare igmc_allreduce_grads call igmc_comm_check.
 *   igmc_comm_check     : synchronises `stream`; fails -- once -- if a rank's words did not arrive in time in a peer exchange
 *                         since the last check; igmc_comm_kind: 0 one rank, 1 RCCL, 2 host callback, 3 peer-mapped buffers (4: in fine-grained memory).
 */
typedef struct igmc_comm igmc_comm;
typedef int (*igmc_allreduce_fn)(void* user, float* d_buf, int64_t n, void* stream);
int igmc_comm_unique_id(uint8_t* h_id128);
int igmc_comm_create(const uint8_t* h_id128, int rank, int world, int device, igmc_comm** out);
int igmc_comm_create_host(igmc_allreduce_fn fn, void* user, int rank, int world, igmc_comm** out);
int igmc_comm_peer_alloc(int rank, int world, int device, int64_t max_floats, igmc_comm** out, uint8_t* h_handle64);
int igmc_comm_peer_connect(igmc_comm* c, const uint8_t* h_handles);
int igmc_comm_check(igmc_comm* c, void* stream);
int igmc_comm_kind(const igmc_comm* c);
void igmc_comm_destroy(igmc_comm* c);
int igmc_comm_info(const igmc_comm* c, int* rank, int* world);
int igmc_allreduce_grads(igmc_comm* c, float* d_flat_grad, int64_t n, float scale, void* stream);
int igmc_train_step_dp(igmc_model* m, igmc_comm* comm, float* d_params, const igmc_batch* b, int use_edge_flags,
                       const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float multiply_by, float ARR, float* d_out,
                       float* d_grad, float* d_exp_avg, float* d_exp_avg_sq, float* d_loss, double* d_total,
                       int64_t* d_ctrl, int64_t adam_t, float lr, float beta1, float beta2, float eps, float weight_decay,
                       void* stream);

/* Eval reduction helper (reference train_eval.py:195): d_acc[0] += sum_g (out-y)^2, d_acc[1] += B. */
int igmc_sse_accumulate(const float* d_out, const igmc_batch* b, double* d_acc, void* stream);
/* ... and, in the same launch, the tick that ends an evaluation step of the grouped pipeline (igmc_ctrl_tick): an evaluation step
 * is then the forward launch + this one. */
int igmc_sse_accumulate_tick(const float* d_out, const igmc_batch* b, double* d_acc, int64_t* d_ctrl, void* stream);

/* Per-kernel timing of the last call (HIP events on the launch stream); names/ms arrays are
 * filled up to `cap` (ms = total over `calls` launches of that kernel since the last fetch);
 * returns the number of distinct kernels recorded, or <0 on error. */
int igmc_profile_enable(int on);
int igmc_profile_fetch(char names[][48], float* ms, int* calls, int cap);
/* igmc_profile_enable(2): no events (legal inside a hipGraph capture); instead every k_graph_step launch enqueued or
 * captured from then on clocks itself on the device (earliest workgroup start -> last workgroup end, constant-rate
 * wall clock), which is how bench.py times the dominant kernel UNDER graph replay.  Returns the launches and their
 * mean duration since the last reset; synchronises the device. */
int igmc_profile_gs_clock(const igmc_model* m, int64_t* launches, double* mean_us, int reset);

/* Health check of the workspace (synchronises `stream`): fails when a bounded device-side wait of the
 * one-workgroup-per-subgraph step kernel ever timed out since the last check.  No reference counterpart. */
int igmc_model_check(igmc_model* m, void* stream);
/* 1 when forward / loss_grad / train_step on (this arena, batch size B) run the matrix-core subgraph kernel, which
 * reads the dense blocks only (see igmc_batch_set_lean); 0 otherwise.  No reference counterpart. */
int igmc_model_dense_path(const igmc_model* m, const igmc_batch* b, int B);
/* Which kernels a training step on (this arena, batch size B) is made of -- what a caller that overlaps the extraction of
 * the next batches with the steps needs to know to pace it (igmc_ctrl_gate): 1 = the subgraph kernel (igmc_model_dense_path),
 * 2 = the one-launch dense-layer kernels, 3 = those in their group-split form (two relation groups at once), 0 = the
 * per-layer kernels.  No reference counterpart. */
int igmc_model_step_form(const igmc_model* m, const igmc_batch* b, int B);
/* Clears the row / plane exchange regions of the subgraph and dense-layer kernels (enqueued on `stream`).  They must only ever
 * hold finite values (a consumer copies whole plane images, stale rows of earlier launches included, and multiplies them by
 * zero block entries): call it after steps that ran on non-finite parameters, e.g. when parameters are restored. */
int igmc_model_reset_exchange(igmc_model* m, void* stream);

/* 1 when the arena's slots (129..256 nodes a side, dense block + transposed copy) send the conv layers of the per-layer
 * sequence to the matrix-core layer kernels (k_dl_layer) instead of the CSR row walkers. */
int igmc_model_dense_layers(const igmc_model* m, const igmc_batch* b, int B);

/* ---- Sort-pool readout family: DGCNN_RS (reference models.py:123-167 on the DGCNN base :63-120; Main.py:364-380).
 * Four R-GCN layers with latent_dim [32, 32, 32, 1] (the conv kernels of igmc_model), concat (97 channels),
 * global_sort_pool(k) (PyG 1.4.2: nodes of a graph by the last channel, descending; first k rows, zero padding),
 * Conv1d(1,16,97,97) + ReLU, MaxPool1d(2,2), Conv1d(16,32,5,1) + ReLU, lin1 (dense -> 128) + ReLU, dropout(0.5), lin2.
 * Parameters live in ONE flat buffer in the "true" layout reported by igmc_sortpool_layout (offsets of
 * convs.{0..3}.{basis,root,bias,att}, conv1d_params1.{weight,bias}, conv1d_params2.{weight,bias}, lin1.{weight,bias},
 * lin2.{weight,bias}; [24] parameter count, [25] dense width 32 * (k/2 - 4), [26] k).  `m` supplies the conv workspace
 * (create it with n_side = 0); max_nodes_per_graph >= the arena's slot size (users + items of one subgraph). */
typedef struct igmc_sortpool igmc_sortpool;
int igmc_sortpool_create(igmc_model* m, int k, int max_nodes_per_graph, igmc_sortpool** out);
void igmc_sortpool_destroy(igmc_sortpool* sp);
int igmc_sortpool_layout(const igmc_sortpool* sp, int64_t* out27);
/* replaces DGCNN_RS.forward (models.py:142-167); training != 0: dropout active, activations kept for the backward */
int igmc_sortpool_forward(igmc_sortpool* sp, const float* d_params, const igmc_batch* b, int training,
                          int use_edge_flags, const uint8_t* d_lin_mask, uint64_t seed, uint64_t step,
                          float* d_out, void* stream);
/* replaces out = model(data); loss = mse (+ ARR over model.convs, train_eval.py:162-174); loss.backward():
 * d_grad (true layout) is overwritten, d_loss[0] = loss, d_loss[1] = sum of squared errors. */
int igmc_sortpool_loss_grad(igmc_sortpool* sp, const float* d_params, const igmc_batch* b, int use_edge_flags,
                            const uint8_t* d_lin_mask, uint64_t seed, uint64_t step, float ARR, float grad_scale,
                            float arr_scale, float* d_out, float* d_grad, float* d_loss, void* stream);

/* optimizer.step() + loss bookkeeping of a sort-pool step in ONE launch, like igmc_step_finish for IGMC: Adam over the
 * sort-pool family's flat buffer (igmc_sortpool_layout [24] parameters), d_loss[0..1], d_total[0] += loss * num_graphs and,
 * with d_ctrl, the Adam scalars from / the tick of the control block -- so that the step is hipGraph-capturable
 * (call igmc_sortpool_loss_grad with d_loss = NULL). */
int igmc_sortpool_step_finish(igmc_sortpool* sp, const igmc_batch* b, float* d_params, const float* d_grad,
                              float* d_exp_avg, float* d_exp_avg_sq, float ARR, float* d_loss, double* d_total,
                              int64_t* d_ctrl, int64_t step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IGMC_HIP_H */
