/*
 * eigentraj.h -- C ABI of libetamd.so, the MI355X (gfx950) implementation of the
 * EigenTrajectory SVD-descriptor hot path.
 *
 * The reference (InhwanBae/EigenTrajectory) is pure Python/PyTorch and has no FFI
 * of its own; the entry points below are what a binding for
 * this path has to cover, one per reference call site:
 *
 *   et_norm_project            EigenTrajectory/descriptor.py:144-160 (projection)
 *                              + normalizer.py:17-51, fused with the moving/static
 *                              routing of EigenTrajectory/model.py:73-90
 *   et_anchor_reconstruct_fwd  EigenTrajectory/anchor.py:76-88 + descriptor.py:162-176
 *                              + normalizer.py:53-62, routed like model.py:98-105
 *   et_anchor_reconstruct_bwd  autograd of the above w.r.t. C_pred (training)
 *   et_fit_gram, et_eigh_topk  descriptor.py:91-114 truncated_SVD via the 2T x 2T Gram
 *                              matrix (parameter_initialization, descriptor.py:116-142)
 *   et_kmeans_*                EigenTrajectory/kmeans.py:59-272 (BatchKMeans); used for
 *                              anchor generation (anchor.py:54-74)
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *  - tensors are contiguous fp32; obs (N,T_obs,2), pred (N,T_pred,2) row-major "NTC";
 *    coefficients are k-major: C (k,N) and C (k,N,S); labels are int64;
 *  - the caller owns every buffer; the library never allocates device memory.
 *    Scratch is passed in as a workspace whose size comes from *_workspace_bytes().  The only memory the library
 *    keeps is host-side and private: per host thread and device a 4-slot pinned staging ring (4 x 96 B + a 64-byte
 *    progress mailbox the single-GPU Lloyd kernel writes into) + 4 events
 *    for the non-blocking convergence polling of the Lloyd loops (csrc/et_hostring.h), and the timing events of
 *    et_kmeans_fit when timing is requested -- created on first use, released when that host thread ends;
 *  - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work and
 *    never synchronise unless documented;
 *  - return value: ET_OK or an ET_ERR_* code; no exceptions cross this boundary;
 *    N == 0 is valid and a no-op; NaN/Inf in the data propagate like in the reference
 *    (normalizer.py:28-29) and are not errors, except in k-means (see below).
 *
 * mode (which descriptor a row uses; the reference keeps two ETDescriptor objects,
 * model.py:29-30):
 *   ET_MODE_STATIC 0  every row: norm_sca=False descriptor (U_*_s, A_s)
 *   ET_MODE_MOVING 1  every row: norm_sca=True  descriptor (U_*_m, A_m)
 *   ET_MODE_SPLIT  2  per row: moving iff ||(obs[-1]-obs[-3])/2|| > static_dist (model.py:46,73)
 *   ET_MODE_IDENTITY 3 rows are used as they are (already normalised input / normalised output)
 */
#ifndef EIGENTRAJ_H
#define EIGENTRAJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ET_ABI_VERSION 3  /* 2: et_kmeans_timing grew `iterations` (32 bytes); 3: et_set_option replaces the ET_* environment switches */

#define ET_OK 0
#define ET_ERR_INVALID_ARG 1  /* bad shape / null pointer / misaligned buffer */
#define ET_ERR_HIP 2          /* a HIP runtime call or launch failed */
#define ET_ERR_UNSUPPORTED 3  /* dimensions outside the compiled range */
#define ET_ERR_WORKSPACE 4    /* workspace too small */
#define ET_ERR_BAD_DATA 5     /* k-means input contains NaN/Inf */
#define ET_ERR_RCCL 6         /* RCCL could not be loaded, or a collective / communicator call failed */

#define ET_MODE_STATIC 0
#define ET_MODE_MOVING 1
#define ET_MODE_SPLIT 2
#define ET_MODE_IDENTITY 3 /* no normalisation at all: bare to_ET_space / to_Euclidean_space
                              (descriptor.py:59-89) with the U_*_s / A_s operands */

#define ET_MAX_T 32  /* max T_obs, T_pred (2T <= 64 rows for the one-wavefront eigensolver) */
#define ET_MAX_K 32  /* max descriptor rank k */
#define ET_KMEANS_MAX_D 32
#define ET_KMEANS_MAX_CLUSTERS 255

typedef void *et_stream_t;
typedef void *et_comm_t; /* an ncclComm_t (RCCL); NULL = a single shard, no collective is enqueued */

int et_abi_version(void);
const char *et_status_string(int status);
/* name of the GPU arch the kernels were compiled for ("gfx950") */
const char *et_compiled_arch(void);

/* ---- tuning switches -----------------------------------------------------------------------------
 * The data path keeps no state between calls and reads nothing from the environment.  The ONE piece of process-wide
 * mutable configuration is this table of measurement aids / test levers (csrc/et_options.hip): every setting selects
 * between forms that are tested to give the same bits, the defaults are the shipped configuration, and a switch is read
 * once at the start of the call it affects (set it before the call, from one thread).  Keys (value as text):
 *   kmeans_packed_min      >= 1024: shards with at least this many points iterate on the packed f16 copy (default 131072 = 2^17)
 *   kmeans_packed          0: trace-less fits keep the fp32 filter body              (default 1)
 *   kmeans_pack_fused      0: the packed copy is written by a pass of its own         (default 1: inside iteration 0)
 *   kmeans_argmax          f: matrix-core filter + exact certification (default) | v: the exact scan only
 *   kmeans_init_tiles      0: farthest-first steps look at every point                (default 1: 256-point tile summaries)
 *   kmeans_filter_threads  256..1024, multiple of 64: workgroup size of the Lloyd kernels (default 0: chosen per shard)
 *   kmeans_chain_copies    1 / 2 / 4 / 8 copies of the per-iteration delta table in a single-GPU fit's one-launch-per-iteration loop (default 2;
 *                          a sharded fit always uses one: its wire format)
 *   kmeans_loop_grid       > 0: at most this many workgroups for the one-launch-per-iteration Lloyd kernel (default 0)
 *   kmeans_loop            a: auto (default) | c: one launch per iteration | p: one persistent launch per fit
 *   reforder_filter_min_lp 4..9: the reference-order Lloyd kernel certifies labels with the matrix-core filter from this cascade level
 *                          power on (L = 2^lp; 4: always, 5: N > 4.2e6, 9: never = default -- built and tested equal, but at 1e7 points
 *                          the launch is bound by its vector instructions either way and the filter's registers halve the
 *                          wavefronts per CU: 138 against 119 us per iteration)
 *   reforder_init_skip_min points (default 2^21): from this shard size on the reference-order farthest-first tests a point's running
 *                          similarity and nearest centroid (5 bytes) before it reads its coordinates (28); same picks
 *   reforder_single_update 1 (default): reference-order fits of at most 8 level-2 blocks update in one 1024-thread workgroup | 0: grid form
 *   metrics_form           a: auto (default) | t: vector-ALU tile kernel for every S | f: fp32 matrix instructions only
 * et_set_option returns ET_ERR_INVALID_ARG for an unknown key or a value the key does not take; et_get_option writes the
 * current value as text.  (The Python binding forwards environment variables ET_OPT_<KEY> once, at load.) */
int et_set_option(const char *key, const char *value);
int et_get_option(const char *key, char *value, size_t value_bytes);

/* ---- TrajNorm (EigenTrajectory/normalizer.py) -------------------------------------------
 * et_norm_params   normalizer.py:17-29: ori (N,1,2), rot (N,2,2) = [[c,-s],[s,c]], sca (N,1,1)
 *                  = 2/||obs[-1]-obs[-3]||; any output may be NULL (flag off).
 * et_normalize     normalizer.py:42-51: ((traj - ori) @ rot) * sca, each step skipped when its
 *                  pointer is NULL.  traj/out (N,T,2); in-place allowed.
 * et_denormalize   normalizer.py:53-62: (traj / sca) @ rot^T + ori. */
int et_norm_params(const float *obs, int64_t N, int T, float *ori, float *rot, float *sca, et_stream_t stream);
/* the same tensors from the compact state nrm (4,N) = ox, oy, dx, dy cached by et_norm_project */
int et_norm_params_from_nrm(const float *nrm, int64_t N, float *ori, float *rot, float *sca, et_stream_t stream);
int et_normalize(const float *traj, int64_t N, int T, const float *ori, const float *rot, const float *sca,
                 float *out, et_stream_t stream);
int et_denormalize(const float *traj, int64_t N, int T, const float *ori, const float *rot, const float *sca,
                   float *out, et_stream_t stream);

/* ---- projection -------------------------------------------------------------------
 * C_obs[j][n]  = sum_f U_obs[f][j]  * normalize(obs[n])[f]      (k,N)
 * C_pred[j][n] = sum_f U_pred[f][j] * normalize(pred[n])[f]     (k,N), if pred != NULL
 * nrm (4,N)    = ox, oy, dx, dy with (ox,oy)=obs[n,-1], (dx,dy)=obs[n,-1]-obs[n,-3]: the
 *                state TrajNorm caches between projection and reconstruction
 *                (normalizer.py:15,20-28); rows 0-1 are model.py:86-87's obs_ori.
 * flag (N)     = 1 for rows routed to the moving descriptor.
 * U_* are (2T,k) row-major like the reference's nn.Parameter (descriptor.py:26-27);
 * the pair a mode does not use may be NULL.  C_pred, nrm, flag may be NULL. */
int et_norm_project(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                    const float *U_obs_m, const float *U_pred_m, const float *U_obs_s, const float *U_pred_s,
                    int mode, float static_dist,
                    float *C_obs, float *C_pred, float *nrm, uint8_t *flag, et_stream_t stream);
/* ... with one more optional output, pose (5,N) = ox, oy, c sca, s sca, +-1/sca: the normaliser of a row in the form the
 * fused error metric consumes (et_anchor_reconstruct_metrics_pose) -- origin, the heading rotation (normalizer.py:24-26)
 * already multiplied by the scale (:28), and 1 / scale with the row's moving (sign bit set) / static decision
 * (model.py:46,73).  20 B per row on top of the projection's 224; pose == NULL is et_norm_project. */
int et_norm_project_pose(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                         const float *U_obs_m, const float *U_pred_m, const float *U_obs_s, const float *U_pred_s,
                         int mode, float static_dist,
                         float *C_obs, float *C_pred, float *nrm, uint8_t *flag, float *pose, et_stream_t stream);

/* Scene form of the projection (one scene batch of model.py:73-90, obs only, N <= ET_SCENE_MAX_N): ONE single-workgroup
 * launch that also produces obs_ori (2,N) = last observed positions minus their mean over the scene (model.py:86-89) --
 * the reference's real workload is N <= 57 pedestrians per forward, i.e. launch-bound.  flag may be NULL. */
#define ET_SCENE_MAX_N 16384
int et_scene_project(const float *obs, int64_t N, int T_obs, int k, const float *U_obs_m, const float *U_obs_s,
                     int mode, float static_dist, float *C_obs, float *nrm, float *obs_ori, uint8_t *flag,
                     et_stream_t stream);

/* ---- training form of a wrapper call on one scene (model.py:58-125 with pred_traj; utils/trainer.py:126-152) ----------
 * N <= ET_SCENE_MAX_N, single-workgroup launches (the work is microseconds; what costs is launches and framework
 * operators).  mode: ET_MODE_STATIC / MOVING / SPLIT.
 *   et_scene_project_train  et_scene_project + C_gt (k,N): the ground truth `pred` (N,T_pred,2) projected with the
 *                           observation's normaliser and its row's U_pred (descriptor.py:144-160, model.py:73-83).
 *   et_wrapper_losses_fwd   recon (S,N,T_pred,2) as et_anchor_reconstruct_fwd, and in the same launch
 *                             losses[0] loss_eigentraj     = mean_n min_s ||A[:,s] + C[:,n,s] - C_gt[:,n]||   (model.py:119)
 *                             losses[1] loss_euclidean_ade = mean_n min_s mean_t ||recon[s,n,t] - gt[n,t]||   (model.py:120-121)
 *                             losses[2] loss_euclidean_fde = mean_n min_s ||recon[s,n,-1] - gt[n,-1]||        (model.py:122-123)
 *                           best (3,N): the per-pedestrian minima, arg (3,N) int32: the sample attaining each.
 *   et_wrapper_losses_bwd   dC (k,N,S) = sum_i g_i * d losses[i] / d C (g_*: the upstream gradient of each loss, a device
 *                           scalar, NULL = not differentiated); each term reaches only its arg-min sample (torch.amin's
 *                           gradient). */
int et_scene_project_train(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                           const float *U_obs_m, const float *U_obs_s, const float *U_pred_m, const float *U_pred_s,
                           int mode, float static_dist, float *C_obs, float *nrm, float *obs_ori, float *C_gt,
                           uint8_t *flag, et_stream_t stream);
int et_wrapper_losses_fwd(const float *C, int64_t N, int S, int k, int T_pred, const float *nrm, const float *A_m,
                          const float *A_s, const float *U_pred_m, const float *U_pred_s, int mode, float static_dist,
                          const float *C_gt, const float *gt, float *recon, float *best, int32_t *arg, float *losses,
                          et_stream_t stream);
int et_wrapper_losses_bwd(const float *g_eigentraj, const float *g_ade, const float *g_fde, const float *C, int64_t N, int S,
                          int k, int T_pred, const float *nrm,
                          const float *A_m, const float *A_s, const float *U_pred_m, const float *U_pred_s, int mode,
                          float static_dist, const float *C_gt, const float *gt, const float *recon, const int32_t *arg,
                          float *dC, et_stream_t stream);

/* ---- anchor refinement + reconstruction -------------------------------------------
 * out[s][n] = denormalize( reshape( U_pred . (C[:,n,s] + A[:,s]) ) )      (S,N,T_pred,2)
 * C (k,N,S); A_m/A_s (k,S) or NULL (no anchor add); normaliser state comes from `nrm`
 * (4,N) as written by et_norm_project, or, when nrm == NULL, is recomputed from obs. */
int et_anchor_reconstruct_fwd(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                              const float *obs, const float *nrm,
                              const float *A_m, const float *A_s, const float *U_pred_m, const float *U_pred_s,
                              int mode, float static_dist, float *out, et_stream_t stream);

/* Fused evaluation form (no counterpart in the reference, which materialises recon_traj and calls
 * utils/metrics.py:73-102): best-of-S displacement errors against gt (N,T_pred,2) without writing the
 * trajectories:  ade[n] = min_s mean_t ||out[s][n][t] - gt[n][t]||,  fde[n] = min_s ||out[s][n][T-1] - gt[n][T-1]||.
 * A NaN anywhere in a trajectory's samples makes its two results NaN (torch.min's propagation).  For T_pred = 12, k = 6,
 * 12 <= S <= 64 the contraction runs on the matrix cores from two-term f16 splits of both operands: results within 1e-6
 * (relative to the largest) of et_anchor_reconstruct_fwd + the metrics; operands beyond f16's range take fp32 instructions. */
int et_anchor_reconstruct_metrics(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                                  const float *obs, const float *nrm,
                                  const float *A_m, const float *A_s, const float *U_pred_m, const float *U_pred_s,
                                  int mode, float static_dist, const float *gt, float *ade, float *fde,
                                  et_stream_t stream);
/* ... with the normaliser from `pose` (5,N) as written by et_norm_project_pose under the same mode and static_dist: the
 * matrix-core kernel (T_pred = 12, k = 6, 12 <= S <= 64) then re-derives nothing per pass (no square root, reciprocal or
 * selects for the rotation and the scale: ~11 % of its vector instructions).  Shapes that kernel does not take fall back to
 * nrm / obs exactly as et_anchor_reconstruct_metrics (so pass them too unless the shape is known); pose == NULL is that call. */
int et_anchor_reconstruct_metrics_pose(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                                       const float *obs, const float *nrm, const float *pose,
                                       const float *A_m, const float *A_s, const float *U_pred_m, const float *U_pred_s,
                                       int mode, float static_dist, const float *gt, float *ade, float *fde,
                                       et_stream_t stream);

/* dC[j][n][s] = sum_f U_pred[f][j] * ((dtraj[s][n] @ R_n) / sca_n)[f]          (k,N,S) */
int et_anchor_reconstruct_bwd(const float *dtraj, int64_t N, int S, int k, int T_obs, int T_pred,
                              const float *obs, const float *nrm,
                              const float *U_pred_m, const float *U_pred_s,
                              int mode, float static_dist, float *dC, et_stream_t stream);

/* ---- fit ----------------------------------------------------------------------------
 * Gram matrices of the normalised trajectories routed to descriptor `which`
 * (1 moving / 0 static) under `mode`:  G_obs (2T_obs,2T_obs), G_pred (2T_pred,2T_pred)
 * fp64, full symmetric; *count = rows used (int64, device).  A data-sharded fit sums
 * G_obs, G_pred and count over ranks (RCCL all-reduce) before et_eigh_topk. */
size_t et_fit_gram_workspace_bytes(int64_t N, int T_obs, int T_pred);
int et_fit_gram(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred,
                int mode, float static_dist, int which,
                double *G_obs, double *G_pred, int64_t *count,
                void *workspace, size_t workspace_bytes, et_stream_t stream);

/* The fit of ONE descriptor in a single call (descriptor.py:116-142 parameter_initialization = normalise -> SVD of the obs
 * rows + SVD of the pred rows): et_fit_gram followed by the eigensolve of both matrices, the matrix assembly folded into
 * the eigensolver's launch (three launches in all).  U_obs (2T_obs,k), U_pred (2T_pred,k),
 * sigma_obs / sigma_pred (k) as et_eigh_topk writes them; G_obs, G_pred, count: optional outputs (may be NULL), the same
 * bits et_fit_gram returns.  Any (T_obs, T_pred) and N = 0 are accepted (they run the two calls one after the other). */
size_t et_fit_descriptor_workspace_bytes(int64_t N, int T_obs, int T_pred);
int et_fit_descriptor(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                      int mode, float static_dist, int which,
                      float *U_obs, float *U_pred, float *sigma_obs, float *sigma_pred,
                      double *G_obs, double *G_pred, int64_t *count,
                      void *workspace, size_t workspace_bytes, et_stream_t stream);

/* Top-k eigenpairs of a symmetric n x n fp64 matrix (n <= 64), parallel-order (round-robin) Jacobi in one
 * workgroup: U (n,k) fp32 = eigenvectors by descending eigenvalue, each signed so its
 * largest-|.| component is positive; sigma[k] = sqrt(max(lambda,0)) -- U[:, :k], S[:k]
 * of torch.linalg.svd at descriptor.py:110-113 up to sign. */
int et_eigh_topk(const double *G, int n, int k, float *U, float *sigma, et_stream_t stream);

/* The same for `batch` (<= ET_EIGH_MAX_BATCH) independent matrices in ONE launch, one workgroup each:
 * G[i] (n[i] x n[i]) -> U[i] (n[i], k[i]), sigma[i] (k[i]).  The arrays themselves are host memory, the
 * matrices and outputs device memory.  A fit solves its obs / pred (x moving / static) Gram matrices
 * side by side this way (the reference runs its torch.linalg.svd calls one after the other,
 * model.py:50-52). */
#define ET_EIGH_MAX_BATCH 8
int et_eigh_topk_batch(int batch, const double *const *G, const int *n, const int *k, float *const *U,
                       float *const *sigma, et_stream_t stream);

/* ---- BatchKMeans ----------------------------------------------------------------------
 * X (d,N) d-major points (= C_pred (k,N)); centroids (d,K); labels int64 (N).
 * Similarity is kmeans.py:71-74 in the reference's operation order; the argmax takes
 * the first maximum and lets NaN win (torch.max).  Per-cluster sums are exact 64-bit
 * fixed-point integers (state.frac fractional bits), hence identical for any
 * partition of the points over workgroups or GPUs. */
typedef struct et_kmeans_state {
    double max_abs_x;   /* max |x| over ALL shards (all-reduce MAX before et_kmeans_begin) */
    double max_abs_c;   /* max |centroid| of the current centroids (NaN ignored)           */
    int64_t n_total;    /* number of points over all shards                                */
    int64_t frac;       /* fractional bits of the coordinate accumulators                   */
    int64_t sim_frac;   /* fractional bits of the similarity (inertia) accumulator          */
    int64_t iter;       /* Lloyd iterations executed                                       */
    int64_t done;       /* 1 once error <= tol (kmeans.py:239); later steps are no-ops     */
    int64_t bad_input;  /* 1 if X holds NaN/Inf (all-reduce MAX)                           */
    double error;       /* kmeans.py:45-51 of the last update                              */
    double inertia;     /* kmeans.py:53-57 of the last assignment                          */
    int64_t fast_ok;    /* 0/1/2: which arg-max kernel the next assignment may use          */
    int64_t min_nz_x_bits; /* fp32 bit pattern of the smallest non-zero |x| (all-reduce MIN) */
} et_kmeans_state;

/* kmeans.py:59-76 euc_sim for one batch element: a (d,m), b (d,n) -> y (m,n) */
int et_euc_sim(const float *a, const float *b, int d, int64_t m, int64_t n, float *y, et_stream_t stream);
/* ... and for the reference's leading batch dimensions in one launch: contiguous a (B,d,m), b (B,d,n) -> y (B,m,n) */
int et_euc_sim_batch(const float *a, const float *b, int64_t batch, int d, int64_t m, int64_t n, float *y,
                     et_stream_t stream);

/* number of int64 in a partials block: d*K sums (d-major), K counts, sim_sum, nan_count */
size_t et_kmeans_partials_len(int d, int K);
/* scratch of every k-means call on a shard of N points: ~5 B per point (labels, running best similarity) + O(d K); for
 * d = 6, 3 <= K <= 32, N >= 131072 (2^17) and N % 4 == 0 another 46 B per point: the packed copy of the points that the trace-less
 * Lloyd iterations of et_kmeans_fit / et_kmeans_fit_sharded read instead of X (csrc/et_kmeans.hip: kmeans_pack_kernel) */
size_t et_kmeans_workspace_bytes(int64_t N, int d, int K);

/* max |x| and non-finite flag of this shard -> state->max_abs_x, state->bad_input
 * (the rest of *state is zeroed). */
int et_kmeans_scan(const float *X, int64_t N, int d, et_kmeans_state *state, et_stream_t stream);
/* fix frac / sim_frac from the (all-reduced) max_abs_x, n_total and the initial centroids
 * (d,K); reset iter/done (done = 1 straight away when bad_input is set). */
int et_kmeans_begin(et_kmeans_state *state, int64_t n_total, const float *centroids, int d, int K,
                    et_stream_t stream);

/* ABI limit: the farthest-first keys and the filter kernel's queue carry 32-bit GLOBAL point indices, so the number
 * of points over all shards of one k-means problem must be < 2^32 (checked: ET_ERR_INVALID_ARG).
 * farthest-first initialisation (kmeans.py:78-112).  best (N) fp32 scratch.
 *   step i (1 <= i < K): similarity of every local point to centroid column i-1 of C0,
 *   running max into best, local arg-min -> cand: {uint64 key, d floats} where
 *   key = orderable(best) << 32 | (index_base + local index); smaller key wins and a
 *   sharded run picks the minimum key over ranks.  cand must hold 8 + 4*d bytes.
 * et_kmeans_init_set copies a point (device, d floats) into column `col` of C0 (d,K). */
int et_kmeans_init_step(const float *X, int64_t N, int d, int K, int i, const float *C0, float *best,
                        int64_t index_base, void *cand, void *workspace, size_t workspace_bytes,
                        et_stream_t stream);
int et_kmeans_init_set(float *C0, int d, int K, int col, const float *point, et_stream_t stream);
/* sharded step, after the all-gather of the ranks' candidate records (n_cands records, stride_bytes apart,
 * 8-byte aligned): the record with the smallest key becomes column `col` of C0 -- the same on every rank. */
int et_kmeans_init_select(const void *cands, int n_cands, int stride_bytes, int d, int K, int col, float *C0,
                          et_stream_t stream);
/* gather X[:, local_index] -> point (d floats, device) */
int et_kmeans_gather_point(const float *X, int64_t N, int d, int64_t local_index, float *point,
                           et_stream_t stream);
/* single-GPU convenience: the whole initialisation, first centroid = X[:, first_index] */
int et_kmeans_init_farthest(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0,
                            void *workspace, size_t workspace_bytes, et_stream_t stream);

/* one Lloyd half-step on a shard (kmeans.py:230 + the sums of :231, :234): labels_u8 (N)
 * and this shard's exact partials; no-op when state->done.  From the second iteration on only
 * the points whose label changed update the sums (exact integer deltas) -- for d = 6, K <= 32 a
 * matrix-core filter first proves, per point, that the label cannot change (bit-identical results) --
 * so `labels_u8` and `workspace` (which holds the running totals) must be the buffers of the previous
 * iteration of the same fit, unmodified; `partials` receives a copy of the totals and is the caller's to
 * overwrite (e.g. all-reduce in place).  With given_labels != NULL (int64, N) the labels are taken as they are instead of
 * computed (compute_centroids, kmeans.py:160-198, as a public method). */
int et_kmeans_assign_accumulate(const float *X, int64_t N, int d, int K, const et_kmeans_state *state,
                                const float *centroids, const int64_t *given_labels, uint8_t *labels_u8,
                                int64_t *partials, void *workspace, size_t workspace_bytes, et_stream_t stream);
/* centroid update + error/inertia/convergence from (all-reduced) partials
 * (kmeans.py:180-182, 45-57, 239).  centroids updated in place; trace (max_iter,2) fp32
 * receives (error, inertia) at row state->iter when not NULL. */
int et_kmeans_update(et_kmeans_state *state, const int64_t *partials, int d, int K, float tol,
                     float *centroids, float *trace, et_stream_t stream);
/* BatchKMeans.fit with l > 1 problems stops them TOGETHER, on the sum of their errors (kmeans.py:228-240: `error` is one
 * sum over the (l, d, K) centroid tensors).  Step-API driver: run et_kmeans_update of every problem with a negative
 * tolerance (no error meets it), then this: states = device array of the n_problems state-block pointers; sets every
 * state's `done` to (sum of the states' errors <= tol).  Later steps of all problems are then no-ops. */
int et_kmeans_joint_done(et_kmeans_state *const *states, int n_problems, float tol, et_stream_t stream);
/* widen the uint8 labels of the last assignment to the reference's int64 */
int et_kmeans_labels_i64(const uint8_t *labels_u8, int64_t N, int64_t *labels, et_stream_t stream);

/* optional kernel timing of et_kmeans_fit: HIP events recorded on `stream` around the dominant kernel of the path.
 * Default (all iterations of the fit in ONE persistent launch, kmeans_lloyd_persist_kernel): the duration of that
 * launch, assign_launches = 1, iterations = the Lloyd iterations it ran.  One launch per iteration (ET_KMEANS_LOOP=chain,
 * shapes the persistent kernel does not take): a sample of the launches -- the first one separately, then of every eight
 * launches a run of four between one pair of events (an event record between two kernels costs a dispatch gap on both
 * sides; assign_launches counts the launches inside the pairs).  Filled after the final sync. */
typedef struct et_kmeans_timing {
    double assign_ms;        /* summed duration of the timed launches                                                 */
    int64_t assign_launches; /* number of launches in assign_ms                                                       */
    double first_assign_ms;  /* per-iteration form only: launch 0 (plain exact scan + full accumulation), else 0      */
    int64_t iterations;      /* Lloyd iterations (24 B of coordinates per point each) that assign_ms covers           */
} et_kmeans_timing;

/* single-GPU fit of one batch element from given initial centroids (kmeans.py:228-240):
 * centroids (d,K) in/out, labels int64 (N) out (NULL: not wanted), trace (max_iter,2) fp32 (error, inertia) per iteration or NULL:
 * without a trace the per-iteration inertia is not evaluated (the reference only prints it, kmeans.py:236) and
 * state.inertia -- the inertia of the last assignment -- comes from one extra pass after the loop (same bits).
 * *state_host receives the final state,
 * *timing_host (may be NULL) the assign-kernel timing.  Synchronises the stream once, at the
 * end (the reference syncs every iteration at kmeans.py:239; here the convergence flag is polled
 * without blocking, a few iterations behind the launches). */
int et_kmeans_fit(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                  int64_t *labels, float *trace, et_kmeans_state *state_host, et_kmeans_timing *timing_host,
                  void *workspace, size_t workspace_bytes, et_stream_t stream);

/* `batch` independent fits in one call (each stops on its own error, unlike BatchKMeans' joint stop): the n_init
 * initialisations of the reference's sklearn anchor clustering (anchor.py:65-71).  X: problem b's points at X + b *
 * x_stride floats (x_stride = 0: all problems cluster the same points); centroids (batch,d,K) in/out; labels
 * (batch,N) int64 or NULL; states_host[batch].  For d = 6, 3 <= K <= 32 and shards that leave room for at least two
 * problems on the device the problems run side by side as the y dimension of ONE persistent launch
 * (csrc/et_kmeans.hip), in chunks when they do not all fit; any other shape: one et_kmeans_fit after the other.
 * workspace: et_kmeans_batch_workspace_bytes (= batch x et_kmeans_workspace_bytes).  Synchronises the stream. */
size_t et_kmeans_batch_workspace_bytes(int64_t N, int d, int K, int64_t batch);
int et_kmeans_fit_batch(const float *X, int64_t x_stride, int64_t N, int d, int K, int64_t batch, int max_iter, float tol,
                        float *centroids, int64_t *labels, et_kmeans_state *states_host, void *workspace,
                        size_t workspace_bytes, et_stream_t stream);

/* ---- opt-in: BatchKMeans in the REFERENCE's own fp32 summation orders (csrc/et_kmeans_reforder.hip) ----------
 * et_kmeans_fit sums the per-cluster coordinates exactly, which makes the result independent of launch geometry and GPU
 * count but lets a whole run drift away from the reference's (kmeans.py:180-182 sums fp32 in ATen's cascade order; whole-run
 * label equality with the imported reference: 96/96 runs here against 76/96 with exact sums).  These entry
 * points restate ATen's CPU orders for kmeans.py:73-74 (norms), :180-182 (cluster sums) and :45-51 (error) literally.
 * Single GPU, any d <= 32, K <= 255.  d = 6, K <= 32, 1024 <= N < 2^29 (the anchor clustering's shape): ONE launch per
 * Lloyd iteration -- ATen's cascade is a fixed tree over index ranges, so its levels are evaluated in parallel without
 * changing an addition (level 0: one work item per chunk lane and coordinate; level 1: inside a workgroup; levels 2, 3,
 * the centroid update and the stop flag: by the workgroups that arrive last) --, no host synchronisation inside the loop.
 * Any other shape: plain kernels, serial where the reference's order is serial, one synchronisation per iteration.
 * workspace: et_kmeans_reforder_workspace_bytes (~ 25 B per point for the fast form's permuted copy). */
size_t et_kmeans_reforder_workspace_bytes(int64_t N, int d, int K);
/* kmeans.py:59-76 with both norms in torch's order: every bit of the reference's euc_sim. a (d,m), b (d,n) -> y (m,n) */
int et_euc_sim_reforder(const float *a, const float *b, int d, int64_t m, int64_t n, float *y, et_stream_t stream);
/* kmeans.py:78-112, re-evaluating euc_sim against all current centroids at every step like the reference */
int et_kmeans_init_farthest_reforder(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0,
                                     void *workspace, size_t workspace_bytes, et_stream_t stream);
/* kmeans.py:143-158 get_labels: labels int64 (N), maxsims (N) (either may be NULL) */
int et_kmeans_predict_reforder(const float *X, int64_t N, int d, const float *centroids, int K, int64_t *labels,
                               float *maxsims, void *workspace, size_t workspace_bytes, et_stream_t stream);
/* kmeans.py:228-240 from given initial centroids: centroids (d,K) in/out, labels int64 (N) or NULL, trace (max_iter,2) or
 * NULL, *state_host: iter, done, error, inertia (the inertia is this build's fp64 sum: the reference only prints it).
 * The fast form synchronises the stream once, at the end; the plain kernels every iteration (kmeans.py:239). */
int et_kmeans_fit_reforder(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                           int64_t *labels, float *trace, et_kmeans_state *state_host, void *workspace,
                           size_t workspace_bytes, et_stream_t stream);
/* BatchKMeans.fit's own loop for `batch` (= l) problems (kmeans.py:228-240): all problems iterate together and stop
 * TOGETHER on the error summed over the whole contiguous (l, d, K) centroid tensor in ATen's inner-sum order.  X: problem
 * b's points at X + b * x_stride floats; centroids (batch, d, K) in/out; labels (batch, N) int64 or NULL; trace (batch,
 * max_iter, 2) or NULL (row = joint error, this problem's inertia); states_host[batch]; *timing_host (may be NULL):
 * assign_ms = the whole loop between two events, assign_launches = launches enqueued.  batch > 1 takes the fast form's
 * shapes only (d = 6, K <= 32, 1024 <= N < 2^29, batch <= 64; the workspace query returns 0 otherwise). */
size_t et_kmeans_reforder_batch_workspace_bytes(int64_t N, int d, int K, int64_t batch);
int et_kmeans_fit_reforder_batch(const float *X, int64_t x_stride, int64_t N, int d, int K, int64_t batch, int max_iter,
                                 float tol, float *centroids, int64_t *labels, float *trace, et_kmeans_state *states_host,
                                 et_kmeans_timing *timing_host, void *workspace, size_t workspace_bytes, et_stream_t stream);

/* kmeans.py:261-272 predict: labels int64 (N); maxsims (N) optional */
int et_kmeans_predict(const float *X, int64_t N, int d, const float *centroids, int K, int64_t *labels,
                      float *maxsims, et_stream_t stream);
/* kmeans.py:143-158 get_labels on a batch in one launch: element b's points at X + b * x_stride floats (d N for a
 * contiguous (B,d,N) tensor, 0: the same points for every element), centroids (B,d,K) -> labels / maxsims (B,N) */
int et_kmeans_predict_batch(const float *X, int64_t x_stride, int64_t batch, int64_t N, int d, const float *centroids,
                            int K, int64_t *labels, float *maxsims, et_stream_t stream);

/* ---- anchor clustering as the reference runs it: sklearn KMeans(init='k-means++', n_init=10) ----------
 * EigenTrajectory/anchor.py:65-71 hands the coefficients to sklearn.cluster.KMeans (third-party; its published
 * algorithm is restated, see csrc/et_kmeanspp.hip).  The Lloyd iterations are et_kmeans_fit; these two entry
 * points are the parts sklearn does around them, in the arithmetic of its float32 code path.
 *
 * et_center_columns   KMeans.fit's pre-processing, in place on X (d,N): mean[j] = fp32 sum over the points in
 *                     index order / N (numpy's add.reduce(axis=0) order), X[j] -= mean[j], and
 *                     *tol = rel_tol * mean_j(var_j) (sklearn `_tolerance`).  mean (d), tol (1): device.
 *                     workspace >= 2 * 4 * ET_KMEANS_MAX_D bytes.
 * et_kmeanspp_seed    greedy k-means++ (n_trials = 2 + floor(ln K) candidates per centre): centers (d,K) fp32 and
 *                     indices (K) int64 of the chosen points.  `uniforms` (device, float64) holds the
 *                     1 + (K-1)*n_trials draws of the seeding in the order sklearn consumes its RandomState:
 *                     [0] picks the first centre (index floor(u*N)), then n_trials thresholds per centre.
 *                     Enqueues 4K-1 launches, no synchronisation. */
int et_center_columns(float *X, int64_t N, int d, float rel_tol, float *mean, float *tol,
                      void *workspace, size_t workspace_bytes, et_stream_t stream);
size_t et_kmeanspp_workspace_bytes(int64_t N, int d, int n_trials);
int et_kmeanspp_seed(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms,
                     float *centers, int64_t *indices, void *workspace, size_t workspace_bytes, et_stream_t stream);
/* `batch` seedings of the SAME points side by side (the n_init initialisations), as the y dimension of the same 4K-1
 * launches: uniforms (batch, 1+(K-1)*n_trials), centers (batch,d,K), indices (batch,K);
 * workspace = batch x et_kmeanspp_workspace_bytes. */
size_t et_kmeanspp_batch_workspace_bytes(int64_t N, int d, int n_trials, int64_t batch);
int et_kmeanspp_seed_batch(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms, int64_t batch,
                           float *centers, int64_t *indices, void *workspace, size_t workspace_bytes, et_stream_t stream);

/* ---- data-sharded fit and k-means: one process per GPU, RCCL over xGMI (csrc/et_sharded.hip) ------------------------
 * The reference has no distributed code; these are the entry points of SURVEY.md §8(b) "with an optional
 * ncclComm_t for the sharded variants".  `comm` is an ncclComm_t -- the caller's own, or one made with et_comm_* below
 * (ncclGetUniqueId on rank 0, the 128 bytes handed to the other ranks by any side channel, ncclCommInitRank everywhere).
 * RCCL is bound at run time with dlopen: an RCCL the process has already mapped (PyTorch's) is reused, otherwise
 * librccl.so.1 is searched; et_comm_load(path) forces a specific library.  Every rank must make the same calls in the
 * same order; all collectives are enqueued on `stream` (a few KB each, latency-bound).  comm == NULL: single shard. */
#define ET_COMM_UNIQUE_ID_BYTES 128
int et_comm_load(const char *librccl_path_or_null);
int et_comm_unique_id(void *id128_host);
int et_comm_init_rank(const void *id128_host, int nranks, int rank, et_comm_t *comm); /* on the CURRENT device */
int et_comm_destroy(et_comm_t comm);
int et_comm_info(et_comm_t comm, int *nranks, int *rank);

/* et_fit_gram over all ranks' rows: the local pass + one grouped all-reduce(SUM) of G_obs, G_pred (fp64) and count. */
int et_fit_gram_sharded(const float *obs, const float *pred, int64_t N_local, int T_obs, int T_pred,
                        int mode, float static_dist, int which,
                        double *G_obs, double *G_pred, int64_t *count,
                        void *workspace, size_t workspace_bytes, et_comm_t comm, et_stream_t stream);

/* workspace of the two calls below (>= et_kmeans_workspace_bytes + the gathered candidate records) */
size_t et_kmeans_sharded_workspace_bytes(int64_t N_local, int d, int K, int nranks);
/* farthest-first initialisation over all ranks' points (kmeans.py:78-112): C0 (d,K) identical on every rank.
 * first_index is GLOBAL; this rank's points are the global indices [index_base, index_base + N_local); best (N_local)
 * fp32 scratch.  One all-gather of an (8 + 4d)-byte record per rank and new centroid. */
int et_kmeans_init_farthest_sharded(const float *X, int64_t N_local, int d, int K, int64_t first_index,
                                    int64_t index_base, float *C0, float *best,
                                    void *workspace, size_t workspace_bytes, et_comm_t comm, et_stream_t stream);
/* Lloyd iterations over all ranks' points from given centroids (identical on every rank; updated in place):
 * per iteration ONE kernel launch (the previous iteration's update in its prologue, then the assignment) and ONE
 * in-place all-reduce(SUM) of the exact int64 delta table (16 (d K + K + 2) values, 18 KB for d = 6, K = 20), all on
 * `stream`, no host round trip inside the loop (the convergence flag is looked at a few iterations late, on the same
 * copy on every rank).  Shards that form does not take (d != 6, K > 32, N_local < 1024 or not a multiple of 4 on ANY
 * rank -- decided together, one all-reduce(MIN) and one stream synchronisation before the loop) run assignment kernels,
 * an all-reduce of d K + K + 2 values and an update kernel instead.  state / labels_u8 (N_local + 3) / partials
 * (et_kmeans_partials_len) are caller-owned device buffers; labels (N_local) int64 may be NULL.  Synchronises the
 * stream at the end; *state_host receives the final state. */
int et_kmeans_fit_sharded(const float *X, int64_t N_local, int64_t N_total, int d, int K, int max_iter, float tol,
                          float *centroids, int64_t *labels, float *trace,
                          et_kmeans_state *state, uint8_t *labels_u8, int64_t *partials,
                          et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                          et_comm_t comm, et_stream_t stream);

/* The REFERENCE-ORDER Lloyd iterations (et_kmeans_fit_reforder above) over shards, d = 6, K <= 32, 1024 <= N_total < 2^29.
 * The order of ATen's cascade sum is a property of the whole array, but its tree is made of index ranges: rank r holds the
 * points [sum of n_locals[0..r), + n_locals[r]) and every rank before the last non-empty one holds a whole number of
 * level-2 blocks (n_locals[r] a multiple of et_kmeans_reforder_shard_block(N_total, d, K) = 4 L^3 points, L the level
 * step of N_total: 16 384 / 131 072 / 1 048 576 points for L = 16 / 32 / 64; zero is a multiple); ranks after it are empty.  Each rank runs
 * levels 0 .. 2 over its own blocks; per iteration ONE all-gather of a record per rank (a row of d K sums + K counts per
 * block, the end of the array's partial sums) and the same sequential level 3 on every rank: centroids, labels, error and
 * iteration count equal et_kmeans_fit_reforder on the whole array bit for bit (inertia: to fp32 rounding).  n_locals is a
 * HOST array of nranks sizes, identical on every rank; labels (n_locals[rank]) int64 or NULL; *state_host the final
 * state.  comm == NULL: nranks must be 1.  Anything else than the shapes above: ET_ERR_UNSUPPORTED / ET_ERR_INVALID_ARG. */
#define ET_REFORDER_MAX_RANKS 64
int64_t et_kmeans_reforder_shard_block(int64_t N_total, int d, int K); /* 0: shape not taken */
size_t et_kmeans_reforder_sharded_workspace_bytes(const int64_t *n_locals, int nranks, int rank, int d, int K);
int et_kmeans_fit_reforder_sharded(const float *X, const int64_t *n_locals, int nranks, int rank, int d, int K, int max_iter,
                                   float tol, float *centroids, int64_t *labels, float *trace,
                                   et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                                   et_comm_t comm, et_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EIGENTRAJ_H */
