/* reagent_hip.h — C ABI of libreagent_hip.so: the MI355X (gfx950) batch-RL training-step kernels.
 *
 * ReAgent (the reference) has no FFI on this path: every op is an eager torch call made from
 * Python.  This header therefore DEFINES the boundary; each entry point names the reference
 * code it replaces (paths relative to the ReAgent tree) so a maintainer can bind it from the
 * reference side with ctypes.
 *
 * Conventions
 *  - plain device pointers + sizes, no torch types; all matrices row-major, leading dimensions in
 *    ELEMENTS; every pointer is device memory unless the comment says host.
 *  - caller owns all memory (PyTorch caching allocator); nothing is allocated or freed here.
 *  - work is enqueued on `stream` only; no entry point synchronises.
 *  - return 0 on success, negative RG_E* for argument errors detected on the host before any
 *    launch, positive = hipError_t of a failed launch.  Never throws, never exits.
 *  - `precision`: RG_PREC_F32 = fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32, parity mode);
 *    RG_PREC_BF16 = bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The
 *    element type of every `void*` activation/weight operand is float resp. bf16 accordingly.
 */
#ifndef REAGENT_HIP_H_
#define REAGENT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rg_stream_t; /* hipStream_t */

enum { RG_OK = 0, RG_EINVAL = -1, RG_EALIGN = -2, RG_EUNSUPPORTED = -3, RG_EWORKSPACE = -4 };
enum { RG_PREC_F32 = 0, RG_PREC_BF16 = 1, RG_PREC_BF16X3 = 2 /* fused stack only: rg_mlp_desc.x3 */ };
enum { RG_DT_F32 = 0, RG_DT_BF16 = 1 };
enum {
  RG_ACT_LINEAR = 0, RG_ACT_RELU = 1, RG_ACT_LEAKY_RELU = 2, RG_ACT_TANH = 3, RG_ACT_SIGMOID = 4,
  RG_ACT_SOFTPLUS = 5 /* reagent/models/fully_connected_network.py:37-44 ACTIVATION_MAP */
};
enum { RG_LOSS_MSE = 0, RG_LOSS_HUBER = 1 }; /* reagent/training/dqn_trainer_base.py:146-155 */

const char* rg_strerror(int code);
int rg_abi_version(void);

/* ---- FullyConnected layer ops ------------------------------------------------------------ */

/* y = act(x · w^T + bias).  Replaces nn.Linear + activation,
 * reagent/models/fully_connected_network.py:101-153.
 * x [batch, in] (ldx), w [out, in] (ldw, nn.Linear layout), bias [out] fp32 or NULL.
 * Outputs (each nullable): y [batch, out] compute type; y32 [batch, out] fp32 (both use ldy);
 * yt [out, batch] (ldyt) transposed copy in compute type (input of rg_fc_wgrad / mask of
 * rg_fc_dgrad). */
int rg_fc_forward(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                  void* y, float* y32, int64_t ldy, void* yt, int64_t ldyt, int batch,
                  int out_features, int in_features, int act, int precision, rg_stream_t stream);

/* dx = (dz · w) ⊙ act'(h_below).  Replaces autograd's AddmmBackward (input grad) +
 * Relu/Tanh/...Backward of the layer below.
 * dz [batch, out] (lddz); wt = w^T [in, out] (ldwt); ht = saved transposed output of the layer
 * below [in, batch] (ldht) or NULL when that "layer" is the network input / linear.
 * Outputs (nullable): dx [batch, in] compute type, dx32 fp32 (lddx), dxt [in, batch] (lddxt). */
int rg_fc_dgrad(const void* dz, int64_t lddz, const void* wt, int64_t ldwt, const void* ht,
                int64_t ldht, int act_below, void* dx, float* dx32, int64_t lddx, void* dxt,
                int64_t lddxt, int batch, int in_features, int out_features, int precision,
                rg_stream_t stream);

/* dw [out, in] (contiguous fp32) = dz^T · x ; db [out] = column sums of dz (nullable).
 * Replaces autograd's AddmmBackward (weight + bias grads).  Inputs are the TRANSPOSED
 * copies: dzt [out, batch] (lddzt), xt [in, batch] (ldxt).  Deterministic: fixed split of
 * the batch axis, partial slabs in `workspace`, ordered second-stage sum. */
size_t rg_fc_wgrad_workspace_bytes(int out_features, int in_features, int batch, int precision);
int rg_fc_wgrad(const void* dzt, int64_t lddzt, const void* xt, int64_t ldxt, float* dw, float* db,
                void* workspace, size_t workspace_bytes, int out_features, int in_features,
                int batch, int precision, rg_stream_t stream);

/* src [rows, cols] (ld_src, dtype src_dt) -> dst [rows, cols] (ld_dst) and/or
 * dst_t [cols, rows] (ld_t), both of dtype dst_dt.  Used to stage fp32 master weights /
 * network inputs into the compute type and its transposed twin. */
int rg_transpose_cast(const void* src, int src_dt, int64_t ld_src, int rows, int cols, void* dst,
                      int64_t ld_dst, void* dst_t, int64_t ld_t, int dst_dt, rg_stream_t stream);

/* nn.LayerNorm(n) between a Linear and its activation — FullyConnectedNetwork's use_layer_norm option
 * (reagent/models/fully_connected_network.py:128-130).  z [batch, n] fp32 = the Linear's output (rg_fc_forward with
 * RG_ACT_LINEAR); y (nullable, element type y_dtype) and / or y32 (nullable) = act(((z - mean) / sqrt(var + eps)) *
 * gamma + beta) with the row's mean and biased variance; mean / rstd [batch] (nullable) are what the backward reads. */
int rg_layer_norm_forward(const float* z, int64_t ldz, const float* gamma, const float* beta, double eps, int act,
                          int batch, int n, void* y, int y_dtype, int64_t ldy, float* y32, int64_t ldy32, float* mean,
                          float* rstd, rg_stream_t stream);
/* g [batch, n] = d loss / d (LayerNorm output, i.e. with the activation's derivative already applied): writes
 * dz (nullable, dz_dtype) and / or dz32 = d loss / d z, dgamma [n] and dbeta [n] (overwritten; per-workgroup partials in
 * `workspace`, summed in a fixed order). */
size_t rg_layer_norm_backward_workspace_bytes(int batch, int n);
int rg_layer_norm_backward(const float* g, int64_t ldg, const float* z, int64_t ldz, const float* mean, const float* rstd,
                           const float* gamma, int batch, int n, void* dz, int dz_dtype, int64_t lddz, float* dz32,
                           int64_t lddz32, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                           rg_stream_t stream);

/* nn.BatchNorm1d(n) on a layer's INPUT — FullyConnectedNetwork's use_batch_norm option (SlateBatchNorm1d on
 * [batch, features], reagent/models/fully_connected_network.py:48-64,107-108).  x, y [batch, n] fp32.
 * training != 0: y = (x - mean_b) / sqrt(var_b + eps) * gamma + beta with the batch mean and BIASED variance, which are
 *   kept in save_mean / save_rstd [n] for the backward; running_mean / running_var (nullable) move by `momentum`
 *   toward the batch mean and the UNBIASED variance (torch.nn.functional.batch_norm), `stat_updates` times (1; 2 for a
 *   module the reference evaluates twice on the same batch, actor.py:215-231; 0 leaves them alone).  `workspace` holds the
 *   fixed-order fp64 column partials (rg_batch_norm_workspace_bytes).
 * training == 0: the running statistics normalise; save_* and workspace are not touched.  gamma / beta nullable. */
size_t rg_batch_norm_workspace_bytes(int batch, int n);
int rg_batch_norm_forward(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int training, int stat_updates, double momentum, double eps, int batch, int n, float* y,
                          int64_t ldy, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                          rg_stream_t stream);
/* g = d loss / d y.  training != 0: mean / rstd are the forward's save_mean / save_rstd and
 *   dx = gamma * rstd * (g - mean_b(g) - xhat * mean_b(g * xhat)); training == 0: mean = running_mean, rstd null,
 *   running_var + eps give the scale and dx = gamma * rstd * g.  dgamma = sum_b g * xhat, dbeta = sum_b g
 *   (each nullable, overwritten); dx nullable. */
int rg_batch_norm_backward(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* gamma,
                           const float* mean, const float* rstd, const float* running_var, int training, double eps,
                           int batch, int n, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, rg_stream_t stream);

/* nn.Dropout(p) after a layer's activation (fully_connected_network.py:139-141), training mode.
 * generate != 0: keep [batch * n] bytes ~ Bernoulli(1 - p) from Philox4x32-10 keyed by `seed`, counter (element / 4,
 *   offset) — a different `offset` per call gives an independent mask — and y = x * keep / (1 - p).
 * generate == 0: `keep` is read (the backward pass: x = d loss / d y, y = d loss / d x).  x may alias y. */
int rg_dropout(const float* x, int64_t ldx, int batch, int n, double p, int generate, uint64_t seed, uint64_t offset,
               uint8_t* keep, float* y, int64_t ldy, rg_stream_t stream);

/* dz = dy * act'(z) written through the activation output y = act(z), fp32 [rows, cols]: turns the
 * gradient w.r.t. a non-linear OUTPUT layer (FullyConnectedActor's tanh head,
 * reagent/models/actor.py:71-75) into the pre-activation gradient the backward entry points take. */
int rg_act_backward(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int act, float* dz,
                    int64_t ld_dz, int rows, int cols, rg_stream_t stream);

/* ---- fused FullyConnected stack (bf16 throughput path) ------------------------------------ */

/* Whole-network kernels for stacks whose hidden layers share one width in {256, 512}, input
 * width <= 512 and output width <= 256 (rg_mlp_fused_supported): a 128-row activation tile stays
 * in LDS across all layers; weights stream from HBM/L2 in MFMA B-fragment order
 * (rg_stage_weights_frag); what backward needs is saved in MFMA C-fragment order
 * (rg_frag_elems(batch, width) bf16 elements per saved matrix), which rg_fc_wgrad_frag
 * consumes directly as MFMA operands.  Replaces FullyConnectedNetwork.forward
 * (reagent/models/fully_connected_network.py:157-163) and its autograd backward. */
#define RG_MLP_MAX_LAYERS 6
typedef struct {
  int32_t n_layers;
  int32_t dims[RG_MLP_MAX_LAYERS + 1];        /* dims[0] = input features, dims[l+1] = out of layer l */
  int32_t acts[RG_MLP_MAX_LAYERS];            /* RG_ACT_* per layer; backward takes d loss / d PRE-activation of the last layer (rg_act_backward) */
  const void* wfrag_fwd[RG_MLP_MAX_LAYERS];   /* rg_stage_weights_frag outputs */
  const void* wfrag_bwd[RG_MLP_MAX_LAYERS];
  const float* bias[RG_MLP_MAX_LAYERS];
  void* act_frag[RG_MLP_MAX_LAYERS];          /* [l] = saved INPUT of layer l, C-fragment order */
  void* dz_frag[RG_MLP_MAX_LAYERS];           /* [l] = d loss / d pre-activation output of layer l */
  void* act_sign[RG_MLP_MAX_LAYERS];          /* optional, [l] (l >= 1) = one bit per element of act_frag[l]
                                               * (value > 0), rg_sign_bytes(batch, dims[l]) bytes, written by a
                                               * saving forward; when layer l-1 is ReLU / leaky ReLU the backward
                                               * reads these 16 B per lane instead of act_frag[l] */
  float* db[RG_MLP_MAX_LAYERS];               /* backward output: bias gradients [dims[l+1]] (nullable) */
  const float* w[RG_MLP_MAX_LAYERS];          /* fp32 master weights [dims[l+1], dims[l]] (stage_weights_fused) */
  float* dw[RG_MLP_MAX_LAYERS];               /* weight gradients, same shape (wgrad_fused) */
  int32_t x3;                                 /* 0: bf16 operands.  1: split-bf16 ("bf16x3", RG_PREC_BF16X3) — every
                                               * operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a product
                                               * is hi*hi + hi*lo + lo*hi on the bf16 MFMA pipe with fp32 accumulation
                                               * (~2^-16 relative per product: fp32-class results).  All
                                               * fragment buffers (wfrag_*, act_frag, dz_frag) then hold TWO planes, hi
                                               * then lo, i.e. twice the element counts rg_*_elems report; the kernels
                                               * work on 64-row tiles (both planes of the activation tile share the LDS) */
  int32_t dx_only;                            /* backward: 1 = only the input gradient is wanted (a frozen network, e.g.
                                               * SAC's critics in the actor step, sac_trainer.py:262-279): dz_frag is
                                               * neither required nor written and no weight gradient may follow */
  /* Two-panel network input (FullyConnectedCritic: cat(state, action), reagent/models/critic.py:79-92) without
   * materialising the concatenation: when x2 != NULL the forward reads input columns [0, x_split) from its `x`
   * argument and columns [x_split, dims[0]) from x2 (row pitch ldx2, element type x2_dtype); x_split a multiple of 32. */
  const void* x2;
  int64_t ldx2;
  int32_t x_split;
  /* backward: dx32 receives d loss / d input columns [dx_col0, dims[0]) only (dx32[0] is column dx_col0; a
   * multiple of 32) — SAC's actor step needs the action columns of the critic's input gradient, not the state's */
  int32_t dx_col0;
  int32_t x2_dtype;                           /* element type of x2 (RG_DT_*): the panels may differ — network-ready bf16
                                               * state rows from the sampler next to fp32 actions */
  int32_t wgrad_flags;                        /* ABI 10 (was reserved, 0).  rg_mlp_wgrad_fused: bit 0 = this launch shares the chip
                                               * with another launch on a second stream (the QR-DQN trunk beside the grouped
                                               * head's weight gradient): every split of a layer keeps the same length — the
                                               * uneven plan leans on the launch's own dispatch order */
  /* forward only: row r of the batch the kernels work on reads input row rowmap[r] of x (-1: an all-zero row);
   * `batch` is then the length of rowmap.  Lets a stack run in "grouped space" (rows sorted by a key and padded to
   * whole 128-row tiles, qr_grouped.hip) without materialising the permuted input. */
  const int32_t* rowmap;
  /* Grouped OUTPUT layer (qr_grouped.hip): with tile_key != NULL the last layer is one of n_groups layers
   * [dims[L], dims[L-1]] chosen per ROW of the grouped space: group g owns the rows [row_begin[g], row_begin[g + 1])
   * (ABI 9; rows past row_begin[n_groups] belong to no group and have no output), tile_key[tile] = the first group with rows
   * in a 128-row tile (-1: none) — a tile that spans several groups runs the layer once per group on that group's rows.
   * wfrag_fwd[L-1] / wfrag_bwd[L-1] / bias[L-1] point at group 0 and advance by group_stride_fwd /
   * group_stride_bwd elements / dims[L] per group (rg_group_weights_stage lays them out so).  out_scatter != 0
   * writes output row r of the forward to out32[rowmap[r]] (rows with rowmap[r] < 0 are dropped).  The backward
   * reduces the last layer's bias gradient per group (db[L-1]: [n_groups * dims[L]]; its workspace holds n_groups more
   * partial rows for that layer) and leaves the last layer's weight gradient to rg_group_head_wgrad; dz_frag[L-1], its
   * operand, is a fragment matrix of rows + 32 * n_groups rows: group g's 32-row blocks are written g blocks late, a
   * block two groups share once per group with the other group's rows zeroed.
   * ABI 8: also for split-bf16 stacks (x3): per group [hi plane | lo plane], the strides cover both. */
  const int32_t* tile_key;
  const int32_t* row_begin;
  int32_t n_groups;
  int32_t out_scatter;
  int64_t group_stride_fwd, group_stride_bwd;
  /* ABI 6 — the step's launch-bound tails folded into the weight gradient's reduce launch.
   * defer_db != 0: rg_mlp_backward_fused leaves the bias-gradient partials in its workspace and launches no column
   *   reduce; rg_mlp_wgrad_fused, handed that workspace in db_partials, sums them into db[] in the launch that sums its
   *   own split partials (same arithmetic, one launch instead of two).  With a grouped output layer (ABI 9) the last
   *   layer's per-group reduce is still launched by rg_mlp_backward_fused; the other layers' partials wait for the
   *   rg_mlp_wgrad_fused call on the trunk (the first n_layers - 1 layers: same workspace layout).
   * sum_in != NULL: that launch also writes sum_out[0] = sum_scale * sum(sum_in[0 .. sum_n)) — the mean loss of a step
   *   from the loss head's per-workgroup partials (rg_reduce_sum's arithmetic). */
  int32_t defer_db;
  int32_t sum_n;
  const float* db_partials;
  const float* sum_in;
  float* sum_out;
  double sum_scale;
} rg_mlp_desc; /* host struct */

int rg_mlp_fused_supported(const rg_mlp_desc* d);
size_t rg_frag_elems(int rows, int cols);                  /* bf16 elements of a C-fragment matrix */
size_t rg_sign_bytes(int rows, int cols);                  /* bytes of an act_sign plane */
size_t rg_wfrag_elems(int out_features, int in_features);  /* bf16 elements of a B-fragment weight */
/* w [out, in] fp32 (nn.Linear layout) -> wfrag_fwd (rg_wfrag_elems(out,in)) and/or
 * wfrag_bwd = fragments of w^T (rg_wfrag_elems(in,out)); zero padded. */
int rg_stage_weights_frag(const float* w, int out_features, int in_features, void* wfrag_fwd,
                          void* wfrag_bwd, rg_stream_t stream);
/* out32 [batch, dims[L]] = network(x); x [batch, dims[0]] row-major of dtype x_dtype (RG_DT_*).
 * save = 1 additionally writes act_frag[0..L-1] and the act_sign planes (everything backward + wgrad read).
 * save = 2 writes only what a dx_only backward reads: the act_sign planes, and act_frag[l] of the layers whose
 * activation gradient is not a sign test (tanh, sigmoid, softplus) — for a ReLU stack 16 B per lane per layer
 * instead of the activations themselves. */
int rg_mlp_forward_fused(const rg_mlp_desc* d, const void* x, int x_dtype, int64_t ldx, int batch,
                         float* out32, int64_t ldo, int save, rg_stream_t stream);
/* Given dout32 = d loss / d out32: writes dz_frag[0..L-1] (needs act_frag[1..L-1] from a saving
 * forward of the same batch) and the bias gradients d->db[l] (column sums of dZ_l, reduced
 * deterministically from per-workgroup partials in `workspace`); dx32 (nullable) = d loss / d x,
 * fp32 [batch, dims[0]].  With d->dx_only only dx32 is produced (see rg_mlp_desc). */
size_t rg_mlp_backward_fused_workspace_bytes(const rg_mlp_desc* d, int batch);
int rg_mlp_backward_fused(const rg_mlp_desc* d, const float* dout32, int64_t lddo, int batch,
                          float* dx32, int64_t lddx, void* workspace, size_t workspace_bytes,
                          rg_stream_t stream);
/* dw [out, in] fp32 = dz^T x from C-fragment operands dz_frag (batch x out) and x_frag
 * (batch x in).  Deterministic split over the batch. */
size_t rg_fc_wgrad_frag_workspace_bytes(int out_features, int in_features, int batch);
int rg_fc_wgrad_frag(const void* dz_frag, const void* x_frag, int out_features, int in_features,
                     int batch, float* dw, void* workspace, size_t workspace_bytes,
                     rg_stream_t stream);

/* Whole-stack variants: every layer's weights staged by ONE launch (d->w -> d->wfrag_fwd, and
 * d->wfrag_bwd when need_bwd), every layer's weight gradient by ONE wgrad launch + ONE reduce
 * (d->dz_frag, d->act_frag -> d->dw).  The weight gradient is a deterministic split over the batch (same inputs, same
 * bits); the workspace holds the splits' partial tiles — fp32 for split-bf16 stacks, bf16 in accumulator-tile order for
 * bf16 stacks (round 5: half the bytes; 2^-9 of a PARTIAL sum, far inside what bf16 operands cost the gradient) — and is
 * sized by rg_mlp_wgrad_fused_workspace_bytes for either form.  Split-bf16 stacks multiply BOTH planes of dZ (three
 * MFMAs per tile pair: fp32-class gradients) unless the library runs with RG_X3_DZ_PLANES=1 (dZ as one plane, two
 * MFMAs, ~2e-3 relative on dW: an opt-in). */
int rg_mlp_stage_weights_fused(const rg_mlp_desc* d, int need_bwd, rg_stream_t stream);
size_t rg_mlp_wgrad_fused_workspace_bytes(const rg_mlp_desc* d, int batch);
int rg_mlp_wgrad_fused(const rg_mlp_desc* d, int batch, void* workspace, size_t workspace_bytes,
                       rg_stream_t stream);

/* Optimizer step fused with weight staging for a fused stack: rg_adam_step on the flat parameter slab
 * of the online network, rg_soft_update of the equally laid out target slab (target NULL = none), and
 * the bf16 re-staging of the updated weights into wfrag_fwd / wfrag_bwd (online) and target_wfrag_fwd
 * — same arithmetic per element as the separate entry points, one launch instead of four.
 * w_off / b_off: element offsets of layer l's weight [dims[l+1], dims[l]] and bias [dims[l+1]] inside
 * the slabs.  The fragment buffers must have been staged once (their padding is not rewritten in
 * wfrag_bwd); any of the three fragment pointers of a layer may be NULL. */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* target;
  int32_t n_layers;
  int32_t dims[RG_MLP_MAX_LAYERS + 1];
  int64_t w_off[RG_MLP_MAX_LAYERS];
  int64_t b_off[RG_MLP_MAX_LAYERS];
  void* wfrag_fwd[RG_MLP_MAX_LAYERS];
  void* wfrag_bwd[RG_MLP_MAX_LAYERS];
  void* target_wfrag_fwd[RG_MLP_MAX_LAYERS];
  int32_t x3;  /* ABI 6: split-bf16 stacks — every fragment buffer is [hi plane | lo plane] (rg_mlp_desc.x3), both
                * planes of all three fragment sets are re-staged: lo = bf16(w - hi) as rg_mlp_stage_weights_fused */
  int32_t group_rows[RG_MLP_MAX_LAYERS]; /* ABI 7: group_rows[l] > 0 = layer l is a GROUPED layer (QR-DQN's A x N output
                * layer, reagent/training/qrdqn_trainer.py:108-194, as dims[l+1] / group_rows[l] independent
                * [group_rows[l], dims[l]] layers): its three fragment buffers are laid out as rg_group_weights_stage
                * writes them (group g at g * rg_group_wfrag_elems elements; ABI 8: with x3 a group's set is
                * [hi plane | lo plane], twice that). */
  /* ABI 7, replayed steps (a captured HIP graph must not need launch arguments that change from step to step):
   * sched_pre_ticked != 0 (_sched entry point only): the step is already counted in sched[0] — by the sampler launch of the
   * same step, rg_replay_dqn_batch_pooled — so the coefficients of step sched[0] apply and no rg_sched_tick follows;
   * post_tick != NULL: *post_tick = (*post_tick + 1) % post_tick_mod when the launch is done with it (the cursor of the
   * index pool the next step's sampler reads). */
  int32_t sched_pre_ticked;
  int32_t post_tick_mod;
  int64_t* post_tick;
} rg_mlp_update_desc; /* host struct */
int rg_mlp_update_fused(const rg_mlp_update_desc* d, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double bias_correction1, double bias_correction2_sqrt,
                        double grad_scale, double tau, rg_stream_t stream);

/* ---- replay buffer ------------------------------------------------------------------------ */

/* n-step bookkeeping of ReplayBuffer.sample_transition_batch,
 * reagent/replay_memory/circular_replay_buffer.py:652-663,676-678,741-747,759-774.
 * indices [B] int64; terminal [C] uint8; reward [C] fp32; decays [update_horizon] fp32 host-
 * computed gamma**k (so the product order matches the reference bit for bit).
 * Outputs: steps [B] int64; next_indices [B] int64; out_terminal [B] uint8; out_reward [B] fp32. */
int rg_replay_nstep(const int64_t* indices, const uint8_t* terminal, const float* reward,
                    const float* decays, int64_t capacity, int update_horizon, int batch,
                    int64_t* steps, int64_t* next_indices, uint8_t* out_terminal,
                    float* out_reward, rg_stream_t stream);

/* One launch gathers up to RG_MAX_GATHER_COLS dense columns
 * (`store[key][stack_indices]` of circular_replay_buffer.py:749-757 + sample_to_output :133-141).
 * For column c: dst_c[b, e, s] = src_c[(idx_c[b] - (stack-1) + s) mod capacity, e]
 * with e in [0,row_elems_c), elem_bytes_c in {1,2,4,8}; stack==1 degenerates to a row copy. */
#define RG_MAX_GATHER_COLS 16
typedef struct {
  const void* src;        /* [capacity, row_elems] */
  void* dst;              /* [batch, row_elems, stack] */
  const int64_t* indices; /* [batch] */
  int32_t row_elems;
  int32_t elem_bytes;
  /* optional normalize-on-gather epilogue (fp32 source columns, stack == 1): `norm` points to
   * row_elems rg_norm_col descriptors on the device (one per element, 1:1 ops — no ENUM), applied
   * with presence = 1 while the row streams through (Preprocessor.forward fused into the gather);
   * out_dtype RG_DT_F32 or RG_DT_BF16 selects the element type of dst. */
  const void* norm;
  const float* norm_quantiles;
  int32_t out_dtype;
  int32_t reserved;
} rg_gather_col;
int rg_replay_gather(const rg_gather_col* cols /*host*/, int ncols, int64_t capacity, int stack,
                     int batch, rg_stream_t stream);

/* ---- sum tree for prioritized replay ------------------------------------------------------ */

/* Device-resident SumTree (reagent/replay_memory/sum_tree.py:30-189) backing
 * PrioritizedReplayBuffer (reagent/replay_memory/prioritized_replay_buffer.py:30-185).
 * `tree`: rg_sumtree_nodes(capacity) doubles in heap order (level d at offset 2^d - 1, leaves at
 * level depth = rg_sumtree_depth(capacity) = ceil(log2(capacity))), zero-initialised by the caller.
 * Updates of <= 32 pairs (or with claim == NULL) repeat the reference's delta accumulation exactly;
 * larger batches write the leaves and rebuild node = left + right level by level (identical
 * whenever the sums are exact in fp64, last-place differences otherwise). */
int rg_sumtree_depth(int64_t capacity);
size_t rg_sumtree_nodes(int64_t capacity);
/* SumTree.set for n (index, value) pairs applied in order (a later pair overrides an earlier one on
 * the same index), set_priority :146-157.  `claim`: int32 [2^depth] all -1, scratch for n > 32
 * (returned all -1); nullable -> the in-order single-thread walk is used for any n.
 * Indices outside [0, capacity) are skipped, never dereferenced (the reference raises IndexError; the host
 * wrapper raises it too whenever the indices are host data). */
int rg_sumtree_set(double* tree, int depth, int64_t capacity, const int64_t* indices, const double* values,
                   int n, int* claim, rg_stream_t stream);
/* SumTree.sample :97-131 for n query values in [0, 1] (each scaled by the root, then the
 * reference's descent); stratified_sample :133-153 = queries drawn one per segment by the caller. */
int rg_sumtree_sample(const double* tree, int depth, const double* query01, int n,
                      int64_t* out_indices, rg_stream_t stream);
/* leaf values: out32 (float32, get_priority :159-180) and/or out64; either nullable; 0 for an index outside
 * [0, capacity) */
int rg_sumtree_get(const double* tree, int depth, int64_t capacity, const int64_t* indices, int n,
                   float* out32, double* out64, rg_stream_t stream);

/* DiscreteDqnInputMaker (reagent/gym/preprocessors/trainer_preprocessor.py:72-97,118-158):
 * action / next_action [B] int64 -> one-hot fp32 [B, A] (next_action rows zeroed where terminal),
 * not_terminal [B] = 1 - terminal, action_probability [B] = exp(log_prob) (nullable). */
int rg_make_dqn_input(const int64_t* action, const int64_t* next_action, const uint8_t* terminal,
                      const float* log_prob, int batch, int num_actions, float* action_onehot,
                      float* next_action_onehot, float* not_terminal, float* action_probability,
                      rg_stream_t stream);
/* Sparse replay elements — IDListMetadata / IDScoreListMetadata.sample_to_output (circular_replay_buffer.py:144-274).
 * A feature's lists sit in padded slots ids [capacity, width] (+ scores [capacity, width]) with lens [capacity].
 * rg_ragged_offsets: offsets [batch] = exclusive prefix sums of lens[indices[b]], total [1] = their sum.
 * rg_ragged_copy: ids_out (and scores_out when scores != NULL) receive the sampled rows back to back at `offsets`. */
int rg_ragged_offsets(const int32_t* lens, const int64_t* indices, int batch, int32_t* offsets, int32_t* total,
                      rg_stream_t stream);
int rg_ragged_copy(const int64_t* ids, const float* scores, int width, const int32_t* lens, const int64_t* indices,
                   const int32_t* offsets, int batch, int64_t* ids_out, float* scores_out, rg_stream_t stream);
/* PolicyNetworkInputMaker.__call__ (gym/preprocessors/trainer_preprocessor.py:161-227, dense path) in one launch:
 * action_out / next_action_out [B, A] = rescale_actions(training/utils.py:13-29) from the environment's range to the
 * training range, next_action rows of terminal transitions zeroed; not_terminal [B] = 1 - terminal;
 * action_probability [B] (nullable) = exp(log_prob).  ranges [4 * A] = prev_min | prev_max | new_min | new_max per
 * action dimension.  Same operation order and roundings as the torch expression it replaces. */
int rg_make_policy_input(const float* action, int64_t lda, const float* next_action, int64_t ldna, const uint8_t* terminal,
                         const float* log_prob, const float* ranges, int batch, int action_dim, float* action_out,
                         float* next_action_out, float* not_terminal, float* action_probability, rg_stream_t stream);

/* ---- dense feature normalization ---------------------------------------------------------- */

/* Preprocessor.forward, reagent/preprocessing/preprocessor.py:115-170 (+ _preprocess_* :197-525).
 * One descriptor per OUTPUT column (ENUM features expand to several).  x [B, n_in] fp32 (ldx),
 * presence [B, n_in] uint8 (ldp).  out [B, n_out] fp32 (ldo).
 * p0..p3 per op:  CONTINUOUS mean,stddev | BOXCOX shift,lambda,mean,stddev | ENUM value |
 * QUANTILE q_offset,q_count (into `quantiles`),-,- | CONTINUOUS_ACTION min_serving,scaling,min_training */
enum {
  RG_NORM_BINARY = 0, RG_NORM_PROBABILITY = 1, RG_NORM_CONTINUOUS = 2, RG_NORM_BOXCOX = 3,
  RG_NORM_ENUM = 4, RG_NORM_QUANTILE = 5, RG_NORM_CONTINUOUS_ACTION = 6,
  RG_NORM_DISCRETE_ACTION = 7, RG_NORM_DO_NOT_PREPROCESS = 8, RG_NORM_CLIP_LOG = 9
};
typedef struct {
  int32_t op;
  int32_t in_col;
  float p0, p1, p2, p3;
} rg_norm_col;
int rg_normalize_dense(const float* x, int64_t ldx, const uint8_t* presence, int64_t ldp,
                       const rg_norm_col* cols /*device*/, int n_out, const float* quantiles,
                       float* out, int64_t ldo, int batch, rg_stream_t stream);

/* ---- offline table -> training batch ------------------------------------------------------- */

/* The post-timeline dataset (column schema of select_relevant_columns,
 * reagent/data/oss_data_fetcher.py:293-336) as one device array per column.  rg_table_dqn_batch
 * replaces the data loader's row fetch plus DiscreteDqnBatchPreprocessor.forward
 * (reagent/preprocessing/batch_preprocessor.py:35-66) for a batch of row indices, in one launch:
 * Preprocessor.forward on state / next_state (cols / quantiles as for rg_normalize_dense, n_out
 * output columns), one-hot action / next_action (next_action == n_actions -> all zeros), not_terminal
 * = max of possible_next_actions_mask, pass-through columns.  Nullable table columns: presence
 * (= all present), time_diff / step (= 1), action_probability (= 1), mdp_id / sequence_number (= 0),
 * possible_actions_mask (= all ones).  Nullable outputs: time_diff, step, action_probability, mdp_id,
 * sequence_number, possible_actions_mask.  Masks and presence are bytes (0 / 1). */
typedef struct {
  const float* state_features;                 /* [n_rows, n_features] */
  const uint8_t* state_features_presence;      /* [n_rows, n_features] */
  const float* next_state_features;
  const uint8_t* next_state_features_presence;
  const int64_t* action;                       /* [n_rows] index into the action names */
  const int64_t* next_action;                  /* [n_rows], n_actions = no next action */
  const float* reward;                         /* [n_rows] */
  const float* action_probability;
  const int64_t* time_diff;
  const int64_t* step;
  const int64_t* mdp_id;
  const int64_t* sequence_number;
  const uint8_t* possible_actions_mask;        /* [n_rows, n_actions] */
  const uint8_t* possible_next_actions_mask;
  int64_t n_rows;
  int32_t n_features;
  int32_t n_actions;
} rg_dqn_table; /* host struct */
typedef struct {
  void* state;                                 /* [batch, n_out] fp32 or bf16 (state_dtype) */
  void* next_state;
  float* action;                               /* [batch, n_actions] one-hot */
  float* next_action;
  float* reward;                               /* [batch] */
  float* time_diff;
  float* step;
  float* not_terminal;
  float* possible_actions_mask;                /* [batch, n_actions] */
  float* possible_next_actions_mask;
  float* action_probability;
  int64_t* mdp_id;
  int64_t* sequence_number;
  int32_t state_dtype;                         /* RG_DT_F32 / RG_DT_BF16 */
  int32_t reserved;
} rg_dqn_batch_out; /* host struct */
int rg_table_dqn_batch(const rg_dqn_table* table, const int64_t* indices, int batch, const rg_norm_col* cols,
                       int n_out, const float* quantiles, const rg_dqn_batch_out* out, rg_stream_t stream);
/* The replay store as the DQN step reads it (stack_size 1, dense fp32 observations; gym schema of
 * SURVEY.md §8 a1).  rg_replay_dqn_batch = ReplayBuffer.sample_transition_batch
 * (reagent/replay_memory/circular_replay_buffer.py:614-706: n-step steps / next index / terminal /
 * discounted reward, state and next_state rows) + DiscreteDqnInputMaker
 * (reagent/gym/preprocessors/trainer_preprocessor.py:100-158: one-hots, next_action zeroed on terminal,
 * not_terminal, exp(log_prob), masks at idx / next idx or ones) in ONE launch, optionally with
 * Preprocessor.forward on both state matrices (cols = n_features 1:1 descriptors, all present; NULL =
 * raw fp32 rows).  Bit-identical to rg_replay_nstep + rg_replay_gather + rg_make_dqn_input.
 * RG_EUNSUPPORTED (n_features % 4, > 512 features, unaligned rows): use those three instead. */
typedef struct {
  const float* observation;            /* [capacity, n_features] */
  const int64_t* action;               /* [capacity] */
  const float* reward;                 /* [capacity] */
  const uint8_t* terminal;             /* [capacity] */
  const float* log_prob;               /* [capacity], nullable (action_probability = 1) */
  const float* possible_actions_mask;  /* [capacity, n_actions], nullable (all ones) */
  const int64_t* mdp_id;               /* nullable */
  const int64_t* sequence_number;      /* nullable */
  const float* decays;                 /* [update_horizon] gamma**k, as for rg_replay_nstep */
  int64_t capacity;
  int32_t n_features;
  int32_t n_actions;
  int32_t update_horizon;
  int32_t reserved;
} rg_replay_view; /* host struct */
int rg_replay_dqn_batch(const rg_replay_view* view, const int64_t* indices, int batch, const rg_norm_col* cols,
                        const float* quantiles, const rg_dqn_batch_out* out, rg_stream_t stream);
/* ABI 7: the same launch for a replayed step: index_pool [pool rows][batch] int64, the row sampled is *cursor (device; advanced
 * by the step's rg_mlp_update_fused_sched through rg_mlp_update_desc.post_tick); pre_tick_sched (nullable): the device-resident
 * Adam schedule whose step count this launch advances (sched[0] += 1, see rg_mlp_update_desc.sched_pre_ticked). */
int rg_replay_dqn_batch_pooled(const rg_replay_view* view, const int64_t* index_pool, const int64_t* cursor,
                               double* pre_tick_sched, int batch, const rg_norm_col* cols, const float* quantiles,
                               const rg_dqn_batch_out* out, rg_stream_t stream);

/* ABI 11.  The continuous-action twin of rg_replay_dqn_batch: ReplayBuffer.sample_transition_batch
 * (reagent/replay_memory/circular_replay_buffer.py:614-706, stack_size 1, dense fp32 observations and [capacity, action_dim]
 * fp32 actions) + PolicyNetworkInputMaker (reagent/gym/preprocessors/trainer_preprocessor.py:175-227: rescale_actions of action
 * and next_action into the training range, next_action rows of terminal transitions zero, not_terminal = 1 - terminal,
 * action_probability = exp(log_prob)) in ONE launch, optionally with Preprocessor.forward on both state matrices.
 * Bit-identical to rg_replay_nstep + rg_replay_gather + rg_make_policy_input (same operations, same roundings).
 * ranges [4 * action_dim] (device) = prev_min | prev_max | new_min | new_max, as for rg_make_policy_input.
 * RG_EUNSUPPORTED (n_features % 4, > 512 features, unaligned rows): use those three instead. */
typedef struct {
  const float* observation;  /* [capacity, n_features] */
  const float* action;       /* [capacity, action_dim] */
  const float* reward;       /* [capacity] */
  const uint8_t* terminal;   /* [capacity] */
  const float* log_prob;     /* [capacity], nullable (action_probability = 1) */
  const float* decays;       /* [update_horizon] gamma**k, as for rg_replay_nstep */
  const float* ranges;       /* [4 * action_dim] */
  int64_t capacity;
  int32_t n_features;
  int32_t action_dim;
  int32_t update_horizon;
  int32_t reserved;
} rg_policy_replay_view; /* host struct */
typedef struct {
  void* state;                /* [batch, n_features] fp32 or bf16 (state_dtype) */
  void* next_state;
  float* action;              /* [batch, action_dim] rescaled */
  float* next_action;         /* [batch, action_dim] rescaled, zero rows for terminal transitions */
  float* reward;              /* [batch] n-step discounted sum */
  float* not_terminal;        /* [batch] */
  float* action_probability;  /* [batch], nullable */
  int32_t state_dtype;        /* RG_DT_F32 / RG_DT_BF16 */
  int32_t reserved;
} rg_policy_batch_out; /* host struct */
int rg_replay_policy_batch(const rg_policy_replay_view* view, const int64_t* indices, int batch, const rg_norm_col* cols,
                           const float* quantiles, const rg_policy_batch_out* out, rg_stream_t stream);

/* *bad_flag (device int, zeroed by the caller) becomes 1 if a sampled row holds an action outside
 * [0, n_actions) or a next_action outside [0, n_actions] (F.one_hot would raise), 2 if an index is
 * outside [0, n_rows). */
int rg_table_check_actions(const rg_dqn_table* table, const int64_t* indices, int batch, int* bad_flag,
                           rg_stream_t stream);

/* ---- loss heads --------------------------------------------------------------------------- */

/* DQN TD head: boost_rewards (reagent/training/dqn_trainer_base.py:216-241),
 * compute_discount_tensor (reagent/training/dqn_trainer.py:166-177),
 * get_max_q_values_with_target (dqn_trainer_base.py:33-77), compute_td_loss tail
 * (dqn_trainer.py:201-238) and d loss / d q, fused in one pass over the batch.
 * q, qn_online, qn_target [B, A] fp32 contiguous; action, next_mask [B, A] fp32
 * (next_mask = possible_next_actions_mask when maxq_learning, next_action when SARSA);
 * reward, not_terminal [B] fp32; reward_boosts [A] fp32 or NULL;
 * discount = gamma, or gamma ** gamma_exponent[b] when gamma_exponent != NULL (time_diff / step).
 * Outputs: dq [B, A] fp32 = d(mean loss)/d q; loss_partials [rg_dqn_head_partials(B)] fp32 whose
 * ordered sum / B is the loss (rg_reduce_sum finishes it); next_q [B], next_idx [B] int64 and
 * q_sel [B] (nullable) for logging/tests. */
int rg_dqn_head_partials(int batch);
int rg_dqn_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
                const float* next_mask, const float* reward, const float* reward_boosts,
                const float* not_terminal, double gamma, const float* gamma_exponent, int batch,
                int num_actions, int double_q, int loss_type, float* dq, float* loss_partials,
                float* next_q, int64_t* next_idx, float* q_sel, rg_stream_t stream);

/* Batch-constrained q-learning (reagent/training/dqn_trainer.py:209-215 with
 * get_valid_actions_from_imitator, reagent/training/imitator_training.py:12-25): mask [B, A] (in place)
 * *= (softmax(imitator_logits)[b, a] / max_a softmax(imitator_logits)[b, :] >= drop_threshold). */
int rg_bcq_filter(const float* imitator_logits, int batch, int num_actions, double drop_threshold, float* mask,
                  rg_stream_t stream);

/* CPE heads of the DQN step, _calculate_cpes (reagent/training/dqn_trainer_base.py:338-452) with
 * masked_softmax (reagent/core/torch_utils.py:62-73): reward-network MSE and CPE q-network loss on the
 * logged action's column of each of the M metrics, and both output gradients.
 * reward_est, q_cpe, q_cpe_tgt_next [B, M*A] fp32 = reward_network(state), q_network_cpe(state),
 * q_network_cpe_target(next_state); next_scores [B, A] = q_network(next_state) (after the q step);
 * next_mask [B, A] = possible_next_actions_mask (maxq) or next_action (SARSA); action [B, A] one-hot;
 * reward [B] (unboosted), extra_metrics [B, M-1] or NULL (M == 1); discount as in rg_dqn_head.
 * Outputs: d_reward_est, d_q_cpe [B, M*A] = d(mean loss)/d output; reward_partials, cpe_partials
 * [rg_dqn_head_partials(B)] whose ordered sums / (B*M) are the two losses; propensities_out [B, A]
 * (nullable) = model propensities of the next states. */
int rg_cpe_head(const float* reward_est, const float* q_cpe, const float* q_cpe_tgt_next,
                const float* next_scores, const float* next_mask, const float* action,
                const float* reward, const float* extra_metrics, const float* not_terminal, double gamma,
                const float* gamma_exponent, double temperature, int batch, int num_actions,
                int num_metrics, int loss_type, float* d_reward_est, float* d_q_cpe,
                float* reward_partials, float* cpe_partials, float* propensities_out,
                rg_stream_t stream);

/* C51 head, reagent/training/c51_trainer.py:98-187 (+ CategoricalDQN.log_dist,
 * reagent/models/categorical_dqn.py:36-38).  q / qn_online / qn_target [B, A*N] fp32 = logits viewed
 * (B, A, N); qn_online NULL = select the next action with the target net.  next_mask [B, A] =
 * possible_next_actions_mask (maxq != 0) or next_action (SARSA); support [N] = linspace(qmin, qmax, N).
 * Outputs: dq [B, A*N] = d loss / d logits; loss_partials [B] whose sum is the loss; all_q [B, A]
 * (nullable) = expected values of the current distributions.  Limits: N <= 1024, A <= 256. */
int rg_c51_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
                const float* next_mask, const float* reward, const float* reward_boosts,
                const float* not_terminal, double gamma, const float* gamma_exponent, const float* support,
                double qmin, double qmax, int batch, int num_actions, int num_atoms, int maxq, float* dq,
                float* loss_partials, float* all_q, rg_stream_t stream);

/* Dueling aggregation, reagent/models/dueling_q_network.py:96-107: value [B, num_atoms], advantage [B, A * num_atoms]
 * viewed (B, A, N) (num_atoms = 1 for plain DQN): q = value + advantage - mean over (actions, atoms) of advantage;
 * rg_dueling_split is its adjoint (dadvantage = dq - mean(dq), dvalue[b, n] = sum over actions of dq[b, a, n]). */
int rg_dueling_combine(const float* value, int64_t ldv, const float* advantage, int64_t lda, int batch,
                       int num_actions, int num_atoms, float* q, int64_t ldq, rg_stream_t stream);
int rg_dueling_split(const float* dq, int64_t lddq, int batch, int num_actions, int num_atoms, float* dadvantage,
                     int64_t ldda, float* dvalue, int64_t lddv, rg_stream_t stream);

/* ---- QR-DQN with a grouped output layer (qr_grouped.hip; reagent/training/qrdqn_trainer.py:108-160) ---------
 * The [A * N, H] output layer is treated as A layers [N, H] ("groups"); the batch rows are sorted by the action
 * whose quantiles are needed — "grouped space", dense (ABI 9: no padding between the groups, ceil(batch / 128) tiles) or with
 * each action's rows padded to whole 128-row tiles (rounds 2-3), described by
 *   rowmap    [128 * n_tiles] int32 : batch row of each grouped row, -1 for padding
 *   tile_key  [n_tiles] int32       : first group with rows in each tile, -1 for none
 *   row_begin [n_groups + 1] int32  : first grouped row of each group ([n_groups] = end of the last group's range)
 * all built on the device.  Forward and input gradient of the grouped layer run inside the fused stack kernels
 * (rg_mlp_desc.rowmap / tile_key); h_frag below is the saved input of the stack's last layer (act_frag[L-1]) and
 * dz_frag the dz_frag[L-1] that backward wrote.
 * rg_group_weights_stage : w [n_groups * group_rows, in] fp32 -> per-group B fragments of W_g (forward) and of
 *                          W_g^T (input gradient); rg_group_wfrag_elems(...) bf16 elements per group.
 * rg_wide_head_mean      : wbar [n_groups, in], bbar [n_groups] = mean over each group's rows of w and b — the
 *                          per-action mean over quantiles is the linear layer (wbar, bbar) (fp32, row order).
 * rg_qr_select_action    : key[b] = arg max_a (q[b,a] - 1e9 (1 - mask[b,a])) (maxq != 0, :210-214) or the position of
 *                          the 1 in mask[b,:] (SARSA, mask = next_action), num_actions for an all-zero row.
 * rg_qr_compact_head     : quantile-Huber loss (:143-160, :217-218) of z [grouped rows] against
 *                          T = reward (+ boost of action row_key[rowmap[r]], ABI 9) + gamma^e * not_terminal * zt[rowmap[r], :];
 *                          dz [padded_rows, lddz] (padding rows and columns zero), loss_partials [padded_rows];
 *                          tile_losses (nullable) [padded_rows / 128] = their sums per 128-row tile.  num_atoms <= 256.
 *                          O(N log N) per row (sorted targets + prefix sums), not the N x N pair loop.
 * rg_group_head_wgrad    : dw [n_groups * group_rows, in] = per group dz^T h over the group's rows. */
/* rg_group_rows: the grouped space of `key` [batch] int32 in [0, n_groups] (n_groups = "no group": dropped) — a
 * stable counting sort (rows keep batch order inside a group).  dense != 0 (ABI 9): the groups follow each other without
 * padding, n_tiles >= ceil(batch / 128); dense == 0: every group starts on a tile, n_tiles >= ceil(batch / 128) + n_groups. */
size_t rg_group_rows_workspace_bytes(int batch, int n_groups);
int rg_group_rows(const int32_t* key, int batch, int n_groups, int n_tiles, int dense, int32_t* rowmap, int32_t* tile_key,
                  int32_t* row_begin, void* workspace, size_t workspace_bytes, rg_stream_t stream);
size_t rg_group_wfrag_elems(int group_rows, int in_features, int transposed);
/* ABI 8: x3 != 0 = split-bf16 — a group's fragment set is [hi plane | lo plane] (lo = bf16(w - hi)), 2 *
 * rg_group_wfrag_elems(...) elements per group; rg_mlp_desc.group_stride_* then counts both planes. */
int rg_group_weights_stage(const float* w, int n_groups, int group_rows, int in_features, int x3, void* wfrag_fwd,
                           void* wfrag_bwd, rg_stream_t stream);
int rg_wide_head_mean(const float* w, const float* b, int n_groups, int group_rows, int in_features, float* wbar,
                      float* bbar, rg_stream_t stream);
/* ABI 7: the same, and the means also written as the forward B fragments of the [n_groups, in] mean layer (the slots
 * rg_stage_weights_frag(wbar, n_groups, in, wfrag_fwd, NULL) fills for rows < n_groups; wfrag_fwd staged once before) */
/* ABI 8: x3 != 0 also writes the lo plane (rg_wfrag_elems(n_groups, in) elements behind the hi plane) */
int rg_wide_head_mean_staged(const float* w, const float* b, int n_groups, int group_rows, int in_features, float* wbar,
                             float* bbar, void* wfrag_fwd, int x3, rg_stream_t stream);
int rg_qr_select_action(const float* q, int64_t ldq, const float* mask, int batch, int num_actions, int maxq,
                        int32_t* key, rg_stream_t stream);
/* ABI 7: rg_qr_select_action + rg_group_rows(key, n_groups = num_actions) in the launches of the latter (the key is
 * evaluated by the counting launch and written to `key`): two launches instead of four for the grouped space of a*
 * (qrdqn_trainer.py:210-214) or of the logged action */
int rg_qr_select_group_rows(const float* q, int64_t ldq, const float* mask, int batch, int num_actions, int maxq,
                            int32_t* key, int n_tiles, int dense, int32_t* rowmap, int32_t* tile_key, int32_t* row_begin,
                            void* workspace, size_t workspace_bytes, rg_stream_t stream);
int rg_qr_compact_head(const float* z, int64_t ldz, const float* zt, int64_t ldzt, const int32_t* rowmap,
                       const int32_t* row_key, int padded_rows, const float* reward, const float* reward_boosts,
                       const float* not_terminal, double gamma, const float* gamma_exponent,
                       const float* quantiles, int batch, int num_atoms, float* dz, int64_t lddz,
                       float* loss_partials, float* tile_losses, rg_stream_t stream);
size_t rg_group_head_wgrad_workspace_bytes(int n_groups, int group_rows, int in_features, int splits);
/* ABI 8: x3 != 0 = split-bf16 operands, each [hi plane | lo plane] over `rows` (the grouped space's row count: the lo
 * planes start rg_frag_elems(rows + 32 * n_groups, group_rows) / rg_frag_elems(rows, in_features) elements in), three MFMAs
 * per product.  ABI 9: group g reads the blocks [row_begin[g] / 32, ceil(row_begin[g + 1] / 32)) of h_frag and the same
 * blocks + g of dz_frag (rg_mlp_desc: how the backward launch writes them) */
int rg_group_head_wgrad(const void* dz_frag, const void* h_frag, const int32_t* row_begin, int n_groups,
                        int group_rows, int in_features, int splits, int x3, int rows, float* dw, void* workspace,
                        size_t workspace_bytes, rg_stream_t stream);

/* Discrete CRR heads, reagent/training/discrete_crr_trainer.py.  All matrices [B, A] fp32 contiguous.
 * rg_crr_critic_head = compute_target_q_values (:191-206) + compute_td_loss (:208-212) for one or two
 * critics: target = reward (+ sum_a action * reward_boosts) + gamma * not_terminal * min_k sum_a
 * Qk_target(s', a) * softmax(next_logits)_a ; loss_k = mean((sum_a Qk(s, a) * action - target)^2).
 * q2 / q2_next_target / dq2 / partials2 are all NULL for a single critic.  target_out [B] nullable.
 * partials*: rg_crr_partials(B) floats whose sum / B is the loss.
 * rg_crr_actor_head = compute_actor_loss (:214-285): q = q1_network(state) AFTER its optimizer step,
 * logits = actor scores, logged_prob [B] = extras.action_probability (read only if entropy_coeff > 0).
 * dlogits = d actor_loss / d logits; plain_partials sum / B = actor_loss_without_reg;
 * entropy_partials sum / B = the entropy term (actor_loss = plain + entropy_coeff * entropy). */
int rg_crr_partials(int batch);
int rg_crr_critic_head(const float* q1, const float* q2, const float* q1_next_target, const float* q2_next_target,
                       const float* next_logits, const float* action, const float* reward,
                       const float* reward_boosts, const float* not_terminal, double gamma, int batch,
                       int num_actions, float* target_out, float* dq1, float* dq2, float* partials1,
                       float* partials2, rg_stream_t stream);
int rg_crr_actor_head(const float* q, const float* logits, const float* action, const float* logged_prob,
                      double beta, double max_weight, double entropy_coeff, double clip_limit, int batch,
                      int num_actions, float* dlogits, float* plain_partials, float* entropy_partials,
                      rg_stream_t stream);

/* QR-DQN head, reagent/training/qrdqn_trainer.py:108-160 (+ argmax_with_mask :210-214, huber
 * :217-218, quantiles :70-73).  q / qn_online / qn_target [B, A*N] fp32 = network outputs viewed
 * (B, A, N); qn_online NULL = select the next action with the target net (double_q off).
 * next_mask [B, A] = possible_next_actions_mask (maxq != 0) or next_action (SARSA).
 * Outputs: dq [B, A*N] = d loss / d q; loss_partials [B] whose sum is the loss; all_q [B, A]
 * (nullable) = mean over atoms of q (the trainer's logged `all_q_values`).  The (N, B, N) pair
 * tensor of the reference is never materialised.  Limits: N <= 1024, A <= 256. */
int rg_qr_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
               const float* next_mask, const float* reward, const float* reward_boosts,
               const float* not_terminal, double gamma, const float* gamma_exponent,
               const float* quantiles, int batch, int num_actions, int num_atoms, int maxq, float* dq,
               float* loss_partials, float* all_q, rg_stream_t stream);

/* ---- SAC: Gaussian policy head and loss heads ---------------------------------------------- */

/* GaussianFullyConnectedActor.forward, reagent/models/actor.py:215-231 (+ get_log_prob :233-261),
 * given the FC output loc_scale [B, 2A] (ldls) and the N(0,1) draw `noise` [B, A] the reference
 * takes from torch.randn_like (:217): action [B, A] (lda) = clamp(tanh(loc + noise*exp(scale_log))),
 * log_prob [B] (nullable), squashed_mean [B, A] (nullable). */
int rg_gaussian_head_forward(const float* loc_scale, int64_t ldls, const float* noise, int batch,
                             int action_dim, float* action, int64_t lda, float* log_prob,
                             float* squashed_mean, rg_stream_t stream);
/* get_log_prob(state, squashed_action) for a given action, actor.py:233-261. */
int rg_gaussian_log_prob(const float* loc_scale, int64_t ldls, const float* action, int64_t lda,
                         int batch, int action_dim, float* log_prob, rg_stream_t stream);
/* Backward of rg_gaussian_head_forward: g_action [B, A] (nullable) and g_log_prob [B] (nullable) are
 * d loss / d action and d loss / d log_prob; d_loc_scale [B, 2A] = d loss / d loc_scale. */
int rg_gaussian_head_backward(const float* loc_scale, int64_t ldls, const float* noise,
                              const float* g_action, int64_t ldga, const float* g_log_prob, int batch,
                              int action_dim, float* d_loc_scale, int64_t lddls, rg_stream_t stream);
/* Same with the action-embedding KLD term's gradient (sac_trainer.py:282-306) folded in: kld_coef (nullable) [2A] =
 * (c0, c1) from rg_sac_kld; the term's gradient c0_d + c1_d x reaches the action (kld_on_mean == 0) or, through
 * the squashed mean x = clamp(tanh(loc)), loc alone (kld_on_mean != 0). */
int rg_gaussian_head_backward_kld(const float* loc_scale, int64_t ldls, const float* noise, const float* g_action,
                                  int64_t ldga, const float* g_log_prob, int batch, int action_dim,
                                  float* d_loc_scale, int64_t lddls, const float* kld_coef, int kld_on_mean,
                                  rg_stream_t stream);
/* The KLD term itself, sac_trainer.py:282-306: per action dimension d, m = mean over the batch of x[:, d] and v its
 * unbiased variance (x = the sampled action, or with squash != 0 clamp(tanh(x)) of the actor's loc output: the
 * squashed mean); kld = 0.5 * sum_d ((v + (m - emb_mean)^2) / emb_var - 1 + log emb_var - log v).
 * Writes coef [2A] (the gradient coefficients above, `weight` folded in), kld_terms [A] (scratch: per-dimension
 * terms), kld [1] (nullable) and adds weight * kld to loss_inout [1] (nullable: the actor loss mean). */
int rg_sac_kld(const float* x, int64_t ldx, int squash, int batch, int action_dim, const float* emb_mean,
               const float* emb_var, double weight, float* coef, float* kld_terms, float* kld, float* loss_inout,
               rg_stream_t stream);

/* Critic segment of SACTrainer.train_step_gen, reagent/training/sac_trainer.py:217-248:
 * y = r + gamma * (min(q1_t, q2_t) - alpha * clamp(log_prob', -2, 2)) * not_done (y = r if gamma == 0);
 * loss_i = mse(q_i, y), dq_i = d loss_i / d q_i.  All per-transition arrays are [B] fp32; alpha is a
 * device fp64 scalar; q2* may be NULL (single critic).  *_partials: [rg_sac_partials(B)]. */
int rg_sac_partials(int batch);
int rg_sac_critic_head(const float* q1, const float* q2, const float* q1_target, const float* q2_target,
                       const float* log_prob_next, const float* reward, const float* not_terminal,
                       double gamma, const double* alpha, int batch, float* target, float* dq1,
                       float* dq2, float* loss1_partials, float* loss2_partials, rg_stream_t stream);
/* Actor segment, sac_trainer.py:254-280: loss = mean(alpha*clamp(log_prob,-2,2) - min(q1a, q2a));
 * outputs d loss / d log_prob, d loss / d q1a, d loss / d q2a, loss partials and partial sums of
 * (clamp(log_prob) + target_entropy) for the temperature segment (:311-320). */
int rg_sac_actor_head(const float* log_prob, const float* q1_actor, const float* q2_actor, const double* alpha,
                      double target_entropy, int batch, const float* v_cur, int crr_mode, double crr_p0,
                      double crr_clamp, int backprop_log_prob, float* g_log_prob, float* dq1_actor,
                      float* dq2_actor, float* loss_partials, float* entropy_partials, rg_stream_t stream);
/* grad[0] = d alpha_loss / d log_alpha = -mean(clamp(log_prob) + target_entropy); alpha_loss
 * (nullable) = -(log_alpha * mean(...)).  fp64 like the reference's log_alpha (:124-126). */
int rg_sac_alpha_grad(const float* entropy_partials, int batch, const double* log_alpha, double* grad,
                      double* alpha_loss, rg_stream_t stream);
/* TD3 target-policy smoothing, reagent/training/td3_trainer.py:141-146:
 * out[b, :A] (row pitch ld_out) = clamp(next_actor[b] + clamp(noise[b] * noise_variance, noise_clip_lo,
 * noise_clip_hi), lo, hi)   (the reference's `noise.clamp(*self.noise_clip_range)`);
 * noise [B, A] contiguous ~ N(0, 1). */
int rg_td3_target_action(const float* next_actor, int64_t ld_a, const float* noise, double noise_variance,
                         double noise_clip_lo, double noise_clip_hi, double lo, double hi, float* out,
                         int64_t ld_out, int batch, int action_dim, rg_stream_t stream);

/* torch.optim.Adam arithmetic in fp64; exp_param_out (nullable) = exp(param) (alpha = exp(log_alpha)). */
int rg_adam_step_f64(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n,
                     double lr, double beta1, double beta2, double eps, double bias_correction1,
                     double bias_correction2_sqrt, double* exp_param_out, rg_stream_t stream);
/* out[b, c] = a[b, c] + b[b, c] (b nullable) on strided [batch, cols] views (sums the two critics'
 * action-input gradients). */
int rg_add_cols(const float* a, int64_t lda, const float* b, int64_t ldb, int batch, int cols,
                float* out, int64_t ldo, rg_stream_t stream);

/* out[0] = scale * sum_i in[i], summed in index order by one workgroup (deterministic). */
int rg_reduce_sum(const float* in, int n, float scale, float* out, rg_stream_t stream);

/* ---- optimizer ---------------------------------------------------------------------------- */

/* torch.optim.Adam single-tensor step (torch/optim/adam.py _single_tensor_adam, the arithmetic
 * behind reagent/optimizer/uninferrable_optimizers.py:23-33) over one flat fp32 slab.
 * bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) are computed by the
 * caller in double like the reference does.  grad_scale multiplies g first (1/world for DP). */
int rg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay,
                 double bias_correction1, double bias_correction2_sqrt, double grad_scale,
                 rg_stream_t stream);

/* Graph-safe Adam steps.  A captured HIP graph bakes launch ARGUMENTS in, and bias_correction1/2 change every
 * step: the `_sched` variants read them (and lr) from a device-resident schedule instead, so one captured
 * step replays as step t, t+1, ...  Layout of `sched` (doubles, written by the host):
 *   [0] steps applied so far   [1] lr   [2] n = table entries   [3] reserved
 *   [4 + 2(t-1)], [5 + 2(t-1)] = 1 - beta1^t, sqrt(1 - beta2^t) for t = 1..n (steps past n use entry n; the
 *   host extends the table until both have reached 1.0), computed in double as for the scalar entry points.
 * A launch applies step [0] + 1; rg_sched_tick (one thread: [0] += 1) is enqueued after the last launch of
 * that step.  Arithmetic per element is that of rg_adam_step / rg_mlp_update_fused / rg_adam_step_f64 with the
 * same coefficients: identical bits. */
int rg_adam_step_sched(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                       const double* sched, rg_stream_t stream);
int rg_mlp_update_fused_sched(const rg_mlp_update_desc* d, double beta1, double beta2, double eps,
                              double weight_decay, double grad_scale, double tau, const double* sched,
                              rg_stream_t stream);
int rg_adam_step_f64_sched(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n,
                           double beta1, double beta2, double eps, const double* sched,
                           double* exp_param_out, rg_stream_t stream);
int rg_sched_tick(double* sched, rg_stream_t stream);
/* ABI 11: the ticks of up to RG_MAX_TICKS DISTINCT schedules in one launch (a SAC step updates four optimizers — three networks and
 * the temperature — each with its own schedule: one launch at the end of the step instead of four between its segments).
 * `scheds` is a host array of n device pointers. */
#define RG_MAX_TICKS 8
int rg_sched_tick_many(double* const* scheds, int n, rg_stream_t stream);

/* SoftUpdate.step, reagent/optimizer/soft_update.py:60-70: tgt = tau*src + (1-tau)*tgt */
int rg_soft_update(float* target, const float* source, int64_t n, double tau, rg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REAGENT_HIP_H_ */
