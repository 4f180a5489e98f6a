/* pyz.h -- C-ABI of the MI355X (gfx950) backend for the Pyesian.optimizers hot path.
 *
 * The reference (leoelm/Bayesian_inference_for_NN, "Pyesian") has no FFI: the hot
 * path sits behind the Python abstract class `Optimizer`
 * (Pyesian/optimizers/Optimizer.py:14-165) whose subclasses run eager
 * TensorFlow ops.  This header is the boundary a maintainer would bind with
 * ctypes from inside those `step()` methods (INTEGRATION.md shows the stubs).
 * Every entry point cites the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C linkage, plain pointers and sizes; no torch / HIP types.
 *   - every function returns int: 0 = PYZ_OK, <0 = error; pyz_last_error()
 *     returns the message of the calling thread's most recent failure.
 *   - pointers named d_* are DEVICE pointers (hipMalloc / torch-ROCm storage)
 *     owned by the caller; h_* are host pointers.  `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream).  Calls enqueue work and
 *     return; scalar outputs live in device memory the caller reads after
 *     synchronising its stream.
 *   - a pyz_mlp handle owns its device workspace and is confined to one host
 *     thread at a time.
 *   - flat parameter order (HMC.py:177-183, SVGD.py:159-160,230-239,
 *     nn/BayesianModel.py:73-77): for each Dense layer, kernel (in x out,
 *     row-major) then bias (out).  "P" below is a particle/chain/sample count:
 *     parameter matrices are (P, D) row-major.
 *   - all arithmetic is float32 (the reference's dtype) unless stated.
 */
#ifndef PYZ_H
#define PYZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYZ_VERSION 302 /* 0.3.2 */

#define PYZ_OK 0
#define PYZ_E_INVALID (-1) /* bad argument / unsupported combination */
#define PYZ_E_SHAPE (-2)   /* batch / particle count exceeds the plan */
#define PYZ_E_HIP (-3)     /* HIP runtime error (message has the HIP string) */
#define PYZ_E_OOM (-4)     /* device allocation failed */
#define PYZ_E_NODEV (-5)   /* no gfx950 device visible */
#define PYZ_E_NAN (-6)     /* pyz_check_finite: a step produced a NaN / Inf loss */

/* activations of a Dense layer (Keras names) */
#define PYZ_ACT_LINEAR 0
#define PYZ_ACT_RELU 1
#define PYZ_ACT_TANH 2
#define PYZ_ACT_SIGMOID 3
#define PYZ_ACT_SOFTMAX 4 /* last layer only */

/* losses produced by Dataset.loss() (Pyesian/datasets/Dataset.py:152-159) */
#define PYZ_LOSS_SCCE 0 /* SparseCategoricalCrossentropy on a softmax last layer; labels int32 (B) */
#define PYZ_LOSS_MSE 1  /* MeanSquaredError; targets float32 (B, out) */

/* SVGD bandwidth: pass this as `gamma` for the median heuristic of SVGD.baseline__kernel (SVGD.py:165-181):
 * h = sqrt(0.5 median(sqdist) / log(M + 1)), K = exp(-sqdist / (2 h^2)), repulsion (-K X + X rowsum K) / h^2,
 * evaluated on the snapshot d_all (PYZ_SWEEP_JACOBI, at most 64 particles, counts in multiples of four). */
#define PYZ_SVGD_GAMMA_MEDIAN (-1.0f)

/* SVGD sweep order */
#define PYZ_SWEEP_GAUSS_SEIDEL 0 /* the reference: particle i sees updated rows 0..i-1 (SVGD.py:100-123) */
#define PYZ_SWEEP_JACOBI 1       /* all rows from one snapshot (multi-GPU mode) */

typedef struct pyz_mlp pyz_mlp;

int pyz_version(void);
const char *pyz_last_error(void);
/* number of visible HIP devices (does not initialise a context beyond the count) */
int pyz_device_count(void);

/* ---- plan -------------------------------------------------------------------
 * Dense stack parsed from the Keras JSON the reference passes to
 * Optimizer.compile (Optimizer.py:43-62).  dims has n_layers+1 entries.
 * max_batch bounds every batch (training, validation, prediction) and
 * max_particles every P passed later; the workspace is sized for both. */
int pyz_mlp_create(int n_layers, const int32_t *h_dims, const int32_t *h_acts, int loss,
                   int max_batch, int max_particles, pyz_mlp **out);
int pyz_mlp_destroy(pyz_mlp *mlp);
int64_t pyz_mlp_param_count(const pyz_mlp *mlp);
int64_t pyz_mlp_workspace_bytes(const pyz_mlp *mlp);

/* ---- G1: Keras Dense forward (called at SGLD.py:55, SGD.py:57, HMC.py:155,
 * BBB.py:144, SVGD.py:106; BayesianModel.py:124).  d_out is (P, batch, out):
 * the model output (softmax applied when the last activation is softmax).
 * d_row_idx (optional, int32[batch]) gathers rows of d_x: row m of the batch
 * is d_x[d_row_idx[m]]. */
int pyz_mlp_forward(pyz_mlp *mlp, const float *d_theta, int n_particles, const float *d_x,
                    const int32_t *d_row_idx, int batch, float *d_out, void *stream);

/* ---- G1-G3: forward + loss + tape.gradient (SGLD.py:54-64, HMC.py:128-136,
 * BBB.py:140-173, SVGD.py:104-111).  d_loss is float32[P] (mean loss over the
 * batch); d_grad is (P, D) or NULL (loss only: the validation passes of
 * BBB.py:203-209 and SVGD.py:126-129). */
int pyz_mlp_loss_grad(pyz_mlp *mlp, const float *d_theta, int n_particles, const float *d_x,
                      const void *d_y, const int32_t *d_row_idx, int batch, float *d_grad,
                      float *d_loss, void *stream);

/* ---- S1: SGD.step update (SGD.py:56-69): theta <- theta - lr * grad. */
int pyz_sgd_step(pyz_mlp *mlp, float *d_theta, const float *d_x, const void *d_y,
                 const int32_t *d_row_idx, int batch, float lr, float *d_loss, void *stream);

/* ---- SWAG.step (Pyesian/optimizers/SWAG.py:43-94; survey 8f rank 2): theta <- theta - lr * grad;
 * when update_moments != 0 (n % frequency == 0): mean / sq_mean running moments with count n and
 * d_dev_row (float32[D], one row of the k x D deviation matrix, may be NULL) <- theta - mean.
 * d_loss receives the batch loss. */
int pyz_swag_step(pyz_mlp *mlp, float *d_theta, float *d_mean, float *d_sq_mean, float *d_dev_row,
                  const float *d_x, const void *d_y, const int32_t *d_row_idx, int batch, float lr,
                  int64_t n, int update_moments, float *d_loss, void *stream);

/* ---- L2/L3: SGLD.step (SGLD.py:54-95).  noise = lr * z, z ~ N(0,1) from the
 * library's Philox stream (seed, step n) or from d_unit_noise (float32[D]) when
 * given; theta += -lr * (grad + noise); mean/sq_mean running moments with
 * count n.  d_loss receives the batch loss. */
int pyz_sgld_step(pyz_mlp *mlp, float *d_theta, float *d_mean, float *d_sq_mean, const float *d_x,
                  const void *d_y, const int32_t *d_row_idx, int batch, float lr, int64_t n,
                  uint64_t seed, const float *d_unit_noise, float *d_loss, void *stream);

/* Device-resident multi-step SGLD (the Optimizer.train loop, Optimizer.py:121-134,
 * without per-step host work): step s in [0, n_steps) uses rows
 * d_row_idx[(slot0+s)*max_batch .. +h_batch_sizes[s]) of the resident data set,
 * learning rate h_lr[s] and count n0+s, and writes its batch loss to
 * d_losses[slot0+s].  With use_graph != 0 (and a non-NULL stream) 32 steps are
 * captured once into a hipGraph and replayed; the per-step scalars live in device
 * memory and are advanced by the last kernel of each step. */
int pyz_sgld_run(pyz_mlp *mlp, float *d_theta, float *d_mean, float *d_sq_mean, const float *d_x,
                 const void *d_y, const int32_t *d_row_idx, const int32_t *h_batch_sizes,
                 const float *h_lr, int n_steps, int64_t n0, int64_t slot0, uint64_t seed,
                 float *d_losses, int use_graph, void *stream);

/* Device-resident multi-step SGD (SGD.py:42-69 under the train loop of Optimizer.py:121-134): the
 * arguments of pyz_sgld_run without the moment vectors, count and seed; step s applies
 * theta <- theta - h_lr[s] * grad on its batch and writes the batch loss to d_losses[slot0+s].
 * Needs the fused step (last layer of at most 32 units). */
int pyz_sgd_run(pyz_mlp *mlp, float *d_theta, const float *d_x, const void *d_y,
                const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr,
                int n_steps, int64_t slot0, float *d_losses, int use_graph, void *stream);

/* Device-resident multi-step SWAG (SWAG.py:43-94 under the train loop): step s has count n0 + s; when the
 * count is a multiple of `frequency` the moments are updated and theta - mean goes to row
 * min(ceil(count / frequency), k - 1) of d_dev (float32 (k, D), row = one column of the reference's
 * deviation matrix).  The bookkeeping assumes the chain started at count 0 with every step run through
 * pyz_swag_step / pyz_swag_run.  Needs the fused step. */
int pyz_swag_run(pyz_mlp *mlp, float *d_theta, float *d_mean, float *d_sq_mean, float *d_dev, int k,
                 int frequency, const float *d_x, const void *d_y, const int32_t *d_row_idx,
                 const int32_t *h_batch_sizes, const float *h_lr, int n_steps, int64_t n0,
                 int64_t slot0, float *d_losses, int use_graph, void *stream);

/* What the last pyz_sgld_run / pyz_sgd_run / pyz_swag_run call on this plan did: steps that ran inside replayed
 * hipGraphs, steps launched eagerly, and the number of graph launches.  With use_graph != 0 on a non-NULL
 * stream a run of any length is replayed from graphs: chunks of 32 steps (PYZ_GRAPH_STEPS) and one graph for
 * the remainder, captured for its exact length and kept (seven lengths, least recently used one replaced). */
int pyz_last_run_info(const pyz_mlp *mlp, int32_t *h_graph_steps, int32_t *h_eager_steps,
                      int32_t *h_graph_launches);

/* Non-finite sentinel (the reference prints the loss every step, Optimizer.py:123, so a diverged chain is
 * seen at once; here losses stay on the device): every kernel that finalises a step's loss counts NaN / Inf
 * results.  Synchronises `stream`, returns PYZ_E_NAN if any step since the last call was non-finite (the
 * count is in the message) and resets the count. */
int pyz_check_finite(pyz_mlp *mlp, void *stream);

/* Measurement (bench.py roofline leg): between pyz_probe_begin and pyz_probe_end every kernel the calling
 * thread launches through this library carries a start / stop event pair of its own (hipExtLaunchKernelGGL:
 * the dispatch's begin / end timestamps, i.e. the duration rocprofv3 --kernel-trace reports; no packet is
 * added between the kernels of a pipeline).  While a probe is open the *_run entry points launch eagerly.
 * pyz_probe_end synchronises `stream` and returns, in launch order, each launch's duration in microseconds
 * (h_us[max_launches]) and the kernel expression of its launch site (h_names: max_launches rows of
 * name_stride bytes, NUL terminated; may be NULL); *h_n = launches recorded. */
int pyz_probe_begin(int max_launches);
int pyz_probe_end(void *stream, float *h_us, char *h_names, int name_stride, int *h_n);

/* ---- B2-B4: BBB.step (BBB.py:128-201).  d_mu / d_rho are the variational
 * parameters (D each); eps ~ N(0,1) from Philox (seed, step) or d_eps.
 * Writes the sampled weights to d_w (D, used by the validation pass), the
 * cost (loss + alpha * (log q - log p)) to d_cost[0] and the data loss to
 * d_cost[1].  prior_rho is the raw rho: sigma_p = softplus(prior_rho).  A list-valued
 * GaussianPrior (GaussianPrior.py:50-69) is passed as the optional per-element vectors
 * d_prior_mean_vec / d_prior_rho_vec (float32[D]; NULL = the scalars). */
int pyz_bbb_step(pyz_mlp *mlp, float *d_mu, float *d_rho, float *d_w, const float *d_x,
                 const void *d_y, const int32_t *d_row_idx, int batch, float lr, float alpha,
                 float prior_mean, float prior_rho, const float *d_prior_mean_vec,
                 const float *d_prior_rho_vec, int64_t step, uint64_t seed,
                 const float *d_eps, float *d_cost, void *stream);

/* The BBB train loop (BBB.step inside Optimizer.train, Optimizer.py:121-134) as ONE device-resident run of n_steps
 * steps, replayed from captured hipGraphs like pyz_sgld_run: d_row_idx (slots, max_batch) / h_batch_sizes / h_lr as
 * there; step i of the call is optimizer step step0 + i (its Philox step) and writes {cost, data loss, log q - log p}
 * to d_costs[4 (slot0 + i) ..].  Needs a last layer of at most 32 units.  val_plan (optional; a second plan of the same
 * model with max_batch >= n_val): on the steps BBB.py:203 validates (step % 10 != 0) the validation split d_val_x /
 * d_val_y is forwarded through the weights the step sampled and its mean loss goes to d_val_losses[slot0 + i].
 * Results equal n_steps calls of pyz_bbb_step (+ pyz_mlp_loss_grad on the validation plan) bit for bit. */
int pyz_bbb_run(pyz_mlp *mlp, float *d_mu, float *d_rho, float *d_w, const float *d_x, const void *d_y,
                const int32_t *d_row_idx, const int32_t *h_batch_sizes, const float *h_lr, int n_steps,
                float alpha, float prior_mean, float prior_rho, const float *d_prior_mean_vec,
                const float *d_prior_rho_vec, int64_t step0, int64_t slot0, uint64_t seed, float *d_costs,
                pyz_mlp *val_plan, const float *d_val_x, const void *d_val_y, int n_val,
                float *d_val_losses, int use_graph, void *stream);

/* ---- H2-H5: one HMC proposal (HMC.py:74-104) for P independent chains on the
 * full training split (d_x, d_y, n_rows).  d_q (P, D) is updated in place when
 * accepted.  Momentum p = m * z with z from Philox (seed, step, chain) or
 * d_unit_p (P, D).  h_uniform[P] are the host uniforms of random.random();
 * burning != 0 forces acceptance.  Outputs (device, float32): d_stats (P, 8) =
 * {accepted, loss, U0, K0, U1, K1, log_ratio, 0}.  prior_sigma is the raw rho
 * (negative => NaN potential, every non-burn proposal rejected, HMC.py:149-159).
 * d_prior_mean_vec / d_prior_sigma_vec: optional per-element prior (float32[D], shared by the
 * chains) for list-valued priors; NULL = the scalars.
 * Small 2-layer models (inputs, classes <= 8, hidden + classes <= 64) run with the chain state in LDS:
 * one workgroup per chain for many chains, or -- for at most 16 chains and >= 192 rows -- row slices
 * spread over up to 32 workgroups per chain with one launch per gradient evaluation, replayed as a
 * hipGraph when `stream` is not the default stream.  Results do not depend on the path beyond float32
 * summation order. */
int pyz_hmc_step(pyz_mlp *mlp, float *d_q, int n_chains, const float *d_x, const void *d_y,
                 int n_rows, int L, float epsilon, float m, float prior_mean, float prior_sigma,
                 const float *d_prior_mean_vec, const float *d_prior_sigma_vec,
                 int burning, const float *h_uniform, int64_t step, uint64_t seed,
                 const float *d_unit_p, float *d_stats, void *stream);

/* ---- V2-V4: SVGD.step (SVGD.py:84-141).  d_particles (P_local, D) float32 are
 * this rank's rows [row0, row0+P_local) of the (M, D) particle matrix; d_all
 * (M, D) is the matrix the kernel row is evaluated against (== d_particles on
 * one GPU; the all-gathered snapshot under PYZ_SWEEP_JACOBI).  d_adam_m/v are
 * the Keras-legacy-Adam slots (P_local, D); t is the 1-based Adam step.  The
 * squared distances, the RBF kernel and the repulsion sum are evaluated in float64.  gamma > 0 is
 * the fixed bandwidth (reference: 1.0); PYZ_SVGD_GAMMA_MEDIAN selects the median heuristic.  d_loss[0] = sum_i loss_i / M over the
 * local rows.  Under PYZ_SWEEP_JACOBI with M <= 64 and local rows in multiples of four (row0 too) the
 * sweep reads the particle matrix once per pass for all rows (squared distances through the Gram matrix on the
 * float64 matrix cores); a shard then gets the rows of the whole-matrix call.  Under PYZ_SWEEP_GAUSS_SEIDEL with M <= 64 and D <= 196 608 the sweep is one launch
 * per particle that reads the matrix once; otherwise two launches per particle.  The paths agree within
 * float32 rounding of phi. */
int pyz_svgd_step(pyz_mlp *mlp, float *d_particles, int n_local, const float *d_all, int n_total,
                  int row0, float *d_adam_m, float *d_adam_v, const float *d_x, const void *d_y,
                  const int32_t *d_row_idx, int batch, float lr, float gamma, int64_t t, int sweep,
                  float *d_loss, void *stream);

/* The two phases of pyz_svgd_step as calls of their own, for callers that overlap the exchange of the particle
 * matrix (the RCCL all-gather of the Jacobi sweep) with phase 1: pyz_svgd_gradients computes the loss gradients
 * of the local particles (SVGD.py:104-111; they stay inside the plan) and needs no other rank's rows;
 * pyz_svgd_sweep does the rest (kernel rows, repulsion, Adam, d_loss) and is the first reader of d_all. */
int pyz_svgd_gradients(pyz_mlp *mlp, const float *d_particles, int n_local, const float *d_x, const void *d_y,
                       const int32_t *d_row_idx, int batch, void *stream);
int pyz_svgd_sweep(pyz_mlp *mlp, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                   float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, int sweep, float *d_loss,
                   void *stream);

/* pyz_svgd_sweep under PYZ_SWEEP_JACOBI (at most 64 particles, row0 and n_local multiples of four) in its two halves,
 * for callers that place work -- or a collective of their own -- between them (SVGD.py:54-68,183-202 is a function of
 * the particle matrix alone; SVGD.py:112-123 needs the gradients too):
 *   pyz_svgd_kernel_matrix  squared distances of rows [row0, row0 + n_local) of the snapshot d_all (M, D) against all
 *                           of it (float64), the bandwidth when gamma == PYZ_SVGD_GAMMA_MEDIAN, K rows and their sums;
 *                           they stay inside the plan.  It touches neither the gradients nor the losses of phase 1:
 *                           it may run on ANOTHER stream than pyz_svgd_gradients, at the same time;
 *   pyz_svgd_combine        phi, the legacy Adam step of every local row, d_loss -- ordered by the caller after BOTH
 *                           (stream order or events); same d_all, rows and gamma as the kernel-matrix call.
 * kernel_matrix + combine on one stream is what pyz_svgd_sweep runs for such shapes (bit-identical results).
 * Under PYZ_SWEEP_JACOBI d_particles is only WRITTEN (the rows' current values are read from rows [row0, ...) of
 * d_all -- by pyz_svgd_step's gradient pass too): a one-GPU caller may alternate two (M, D) buffers instead of copying
 * the matrix every step.  (pyz_svgd_gradients takes the current rows explicitly.)
 * PYZ_E_INVALID for shapes the all-rows-at-once kernels do not take (use pyz_svgd_sweep), and from pyz_svgd_combine /
 * pyz_svgd_sweep when the plan does not hold what they consume (another entry point used its buffers in between).
 * A plan serves one stream at a time, with this one exception. */
int pyz_svgd_kernel_matrix(pyz_mlp *mlp, const float *d_all, int n_total, int row0, int n_local, float gamma,
                           void *stream);
int pyz_svgd_combine(pyz_mlp *mlp, float *d_particles, int n_local, const float *d_all, int n_total, int row0,
                     float *d_adam_m, float *d_adam_v, float lr, float gamma, int64_t t, float *d_loss,
                     void *stream);

/* The distance pass of pyz_svgd_kernel_matrix split over the ELEMENTS of the particles, for sharded runs (each
 * rank reads D / world of the gathered matrix instead of all of it; the exchange is PYZ_SVGD_GROUPS x 32 KB, the caller's):
 * the blocks of the pass form PYZ_SVGD_GROUPS groups of consecutive blocks;
 *   pyz_svgd_gram_groups           partial squared distances of ALL pairs over groups [g_lo, g_hi) of d_all (M, D), M a multiple
 *                                  of four <= 64, summed per group into d_groups[(g * 64 + i) * 64 + j] (float64; entries of
 *                                  other groups are not touched) -- rank r of a world that divides 8 takes groups
 *                                  [8 r / world, 8 (r + 1) / world) and all-gathers its slice;
 *   pyz_svgd_kernel_matrix_groups  pyz_svgd_kernel_matrix with the complete d_groups (PYZ_SVGD_GROUPS, 64, 64) in place of the
 *                                  pass over d_all (still named: pyz_svgd_combine checks it).
 * pyz_svgd_kernel_matrix sums its own partials in the same order (groups, then a fixed tree over the groups), so both
 * routes give the same bits for any split of the groups over ranks (SVGD.py:183-202 per pair, float64). */
#define PYZ_SVGD_GROUPS 8
int pyz_svgd_gram_groups(pyz_mlp *mlp, const float *d_all, int n_total, int g_lo, int g_hi, double *d_groups, void *stream);
int pyz_svgd_kernel_matrix_groups(pyz_mlp *mlp, const double *d_groups, const float *d_all, int n_total, int row0,
                                  int n_local, float gamma, void *stream);

/* ---- R1: BayesianModel.predict (BayesianModel.py:106-129): S weight draws
 * d_weights (S, D) -> d_samples (S, n, out) with NaN -> 0, d_mean (n, out). */
int pyz_predict(pyz_mlp *mlp, const float *d_weights, int n_samples, const float *d_x, int n,
                float *d_samples, float *d_mean, void *stream);

/* ---- noise: d_out[i] = mean + std * z_i, z from Philox stream (seed, stream, step).
 * (tf.random.normal at SGLD.py:67, HMC.py:171; tfp samplers at BBB.py:234-237,
 * SVGD.py:154, distributions/tf/TensorflowProbabilityDistribution.py:55-58.) */
int pyz_fill_normal(float *d_out, int64_t n, uint64_t seed, uint32_t stream_id, uint32_t step,
                    float mean, float std, void *stream);

/* n_rows draws of a vector Normal(d_loc, d_scale) (float32[len]) into columns [col0, col0 + len) of the
 * row-major (n_rows, row_stride) matrix d_out: the weight draws of BayesianModel._sample_weights
 * (BayesianModel.py:63-77) for Normal posteriors, made where pyz_predict reads them.  Draw r uses the
 * Philox counter (seed, stream_id, first_draw + r). */
int pyz_sample_normal_rows(float *d_out, int64_t n_rows, int64_t row_stride, int64_t col0, int64_t len,
                           const float *d_loc, const float *d_scale, uint64_t seed,
                           uint32_t stream_id, uint32_t first_draw, void *stream);

/* ---- device memory for callers that have no allocator of their own (a TensorFlow / NumPy host process binding
 * this library from the reference's step(), INTEGRATION.md route B; torch-ROCm callers pass their tensors'
 * storage instead).  pyz_upload / pyz_download copy between a HOST buffer and device memory and return when
 * the host buffer may be reused / holds the data; pyz_sync waits for `stream`. */
int pyz_malloc(size_t bytes, void **d_out);
int pyz_free(void *d_ptr);
int pyz_upload(void *d_dst, const void *h_src, size_t bytes, void *stream);
int pyz_download(void *h_dst, const void *d_src, size_t bytes, void *stream);
int pyz_sync(void *stream);

/* ---- exchange step of the sharded SVGD run without a collective library ("verify a direct algorithm, else a
 * peer-write fallback"): every rank WRITES its rows straight into every peer's gathered matrix (peer-mapped memory over
 * xGMI, the caller's copies) and then its slot of the peer's flag array; a consumer parks this on the stream that reads the
 * gathered matrix: the stream goes on once d_flags[0 .. n) are all >= value (system-scope loads, one wave), i.e. once
 * every rank's rows of that step have landed.  Gives up after spin_limit polls without progress: *d_fail (optional) is set to
 * 1 and the stream goes on.  PYZ_E_INVALID for n outside [1, 64]. */
int pyz_wait_flags(const uint64_t *d_flags, int n, uint64_t value, int spin_limit, int *d_fail, void *stream);

/* ---- measurement hook (bench.py roofline leg): launch `iters` times ONE kernel of the
 * gradient step on the workspace left by the last pyz_mlp_loss_grad call with the same
 * arguments.  kind: 0 = k_dense_fwd of `layer` (G1), 1 = k_dense_bwd_data of `layer`,
 * 2 = k_wgrad_all, the weight gradients of ALL layers (G3; `layer` ignored), written to
 * d_grad (P, D). */
int pyz_bench_dense_kernel(pyz_mlp *mlp, int kind, int layer, const float *d_theta, int n_particles,
                           const float *d_x, const int32_t *d_row_idx, int batch, float *d_grad,
                           int iters, void *stream);

/* ---- diagnostic build only (-DPYZ_STAMPS, csrc/libpyz_stamps.so): copy the in-kernel
 * phase stamps {shader clock, 100 MHz clock} x 8 slots x 256 workgroups x 4 kernels to
 * h_out (uint64).  The shipped library returns PYZ_E_INVALID. */
int pyz_debug_stamps(uint64_t *h_out, int64_t n_words);

/* ---- diagnostic: the lane layout of v_mfma_f64_16x16x4_f64's result as probed on this device:
 * h_out512[4 l + r] = row, h_out512[256 + 4 l + r] = column of register r of lane l (k_svgd_gram_tile runs only
 * when it is row = l / 16 + 4 r, column = l % 16). */
int pyz_debug_mfma_f64_layout(int32_t *h_out512);

#ifdef __cplusplus
}
#endif
#endif /* PYZ_H */
