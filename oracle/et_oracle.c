/*
 * et_oracle.c -- CPU restatement of the EigenTrajectory SVD-descriptor hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (eigentrajectory_amd/); only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg may use it, and there
 * only as the checker.  Parity status: PINNED against golden vectors captured
 * by importing the reference in the build container (tools/make_golden.py ->
 * tests/golden/; checked by tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * restates.  Plain C99, scalar, single thread.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC
 * so that every fused multiply-add in the result is one written as fmaf()/fma()
 * below -- the HIP kernels are written to the same operation order wherever the
 * result must be bit-exact (k-means similarities, fixed-point sums, Jacobi).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ETO_OK 0
#define ETO_EINVAL 1

/* ------------------------------------------------------------------------- */
/* TrajNorm (EigenTrajectory/normalizer.py)                                    */
/* ------------------------------------------------------------------------- */

/* normalizer.py:17-29 calculate_params.  One row of obs (T,2) ->
 * nrm[0..5] = ox, oy, cos, sin, sca, r   (sca = 1 when use_sca == 0). */
static void eto_row_params(const float *row, int T, int use_sca, float *p)
{
    const float ox = row[2 * (T - 1)], oy = row[2 * (T - 1) + 1];
    const float dx = ox - row[2 * (T - 3)], dy = oy - row[2 * (T - 3) + 1];
    const float th = atan2f(dy, dx);       /* normalizer.py:24 */
    p[0] = ox;
    p[1] = oy;
    p[2] = cosf(th);                       /* normalizer.py:25-26 */
    p[3] = sinf(th);
    const float r = sqrtf(dx * dx + dy * dy);
    p[5] = r;
    p[4] = use_sca ? (1.0f / r) * 2.0f : 1.0f; /* normalizer.py:28 */
}

/* model.py:46 / :73  mask = ||(obs[-1]-obs[-3])/2|| > static_dist */
static int eto_row_moving(const float *row, int T, float static_dist)
{
    const float hx = (row[2 * (T - 1)] - row[2 * (T - 3)]) / 2.0f;
    const float hy = (row[2 * (T - 1) + 1] - row[2 * (T - 3) + 1]) / 2.0f;
    return sqrtf(hx * hx + hy * hy) > static_dist;
}

/* normalizer.py:42-51 normalize: ((traj - ori) @ R) * sca, R = [[c,-s],[s,c]] */
static void eto_row_normalize(const float *row, int T, const float *p, int use_sca, float *out)
{
    for (int t = 0; t < T; ++t) {
        const float tx = row[2 * t] - p[0], ty = row[2 * t + 1] - p[1];
        float x = tx * p[2] + ty * p[3];
        float y = tx * (-p[3]) + ty * p[2];
        if (use_sca) {
            x = x * p[4];
            y = y * p[4];
        }
        out[2 * t] = x;
        out[2 * t + 1] = y;
    }
}

/* normalizer.py:53-62 denormalize: (traj / sca) @ R^T + ori */
static void eto_row_denormalize(const float *v, int T, const float *p, int use_sca, float *out)
{
    for (int t = 0; t < T; ++t) {
        float x = v[2 * t], y = v[2 * t + 1];
        if (use_sca) {
            x = x / p[4];
            y = y / p[4];
        }
        out[2 * t] = (x * p[2] + y * (-p[3])) + p[0];
        out[2 * t + 1] = (x * p[3] + y * p[2]) + p[1];
    }
}

/* Public, array forms (used by the golden-vector tests G1). */
int eto_norm_params(const float *obs, int64_t N, int T, int use_sca,
                    float *ori /*N,2*/, float *rot /*N,2,2*/, float *sca /*N or NULL*/)
{
    if (T < 3) return ETO_EINVAL;
    for (int64_t n = 0; n < N; ++n) {
        float p[6];
        eto_row_params(obs + n * 2 * T, T, use_sca, p);
        ori[2 * n] = p[0];
        ori[2 * n + 1] = p[1];
        rot[4 * n + 0] = p[2];
        rot[4 * n + 1] = -p[3]; /* normalizer.py:25 row 0 = [cos, -sin] */
        rot[4 * n + 2] = p[3];
        rot[4 * n + 3] = p[2];
        if (sca) sca[n] = p[4];
    }
    return ETO_OK;
}

int eto_normalize(const float *obs, const float *traj, int64_t N, int T_obs, int T, int use_sca, float *out)
{
    if (T_obs < 3) return ETO_EINVAL;
    for (int64_t n = 0; n < N; ++n) {
        float p[6];
        eto_row_params(obs + n * 2 * T_obs, T_obs, use_sca, p);
        eto_row_normalize(traj + n * 2 * T, T, p, use_sca, out + n * 2 * T);
    }
    return ETO_OK;
}

int eto_denormalize(const float *obs, const float *traj_norm, int64_t N, int T_obs, int T, int use_sca, float *out)
{
    if (T_obs < 3) return ETO_EINVAL;
    for (int64_t n = 0; n < N; ++n) {
        float p[6];
        eto_row_params(obs + n * 2 * T_obs, T_obs, use_sca, p);
        eto_row_denormalize(traj_norm + n * 2 * T, T, p, use_sca, out + n * 2 * T);
    }
    return ETO_OK;
}

int eto_moving_flags(const float *obs, int64_t N, int T, float static_dist, uint8_t *flag)
{
    if (T < 3) return ETO_EINVAL;
    for (int64_t n = 0; n < N; ++n) flag[n] = (uint8_t)eto_row_moving(obs + n * 2 * T, T, static_dist);
    return ETO_OK;
}

/* mode: 0 = every row uses the static descriptor (norm_sca=False),
 *       1 = every row uses the moving descriptor (norm_sca=True),
 *       2 = per-row split by static_dist (model.py:46-48, 73-77). */
static int eto_row_mode(const float *row, int T, int mode, float static_dist)
{
    return mode == 2 ? eto_row_moving(row, T, static_dist) : mode;
}

/* ------------------------------------------------------------------------- */
/* ETDescriptor (EigenTrajectory/descriptor.py)                                */
/* ------------------------------------------------------------------------- */

/* descriptor.py:144-160 projection = normalize_trajectory (:29-44) + to_ET_space
 * (:59-73) for obs (and pred), fused with the wrapper's moving/static routing
 * (model.py:73-90).  U_* are (2T, k) row-major like the nn.Parameter.
 * Outputs: C_obs (k,N), C_pred (k,N) k-major; nrm (4,N) = ox, oy, dx, dy
 * (rows 0-1 are the reference's obs_ori before the scene-mean subtraction,
 * model.py:86-89); flag (N) = 1 for moving rows. */
int eto_norm_project(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                     const float *U_obs_m, const float *U_pred_m, const float *U_obs_s, const float *U_pred_s,
                     int mode, float static_dist,
                     float *C_obs, float *C_pred, float *nrm, uint8_t *flag)
{
    if (T_obs < 3 || T_obs > 64 || T_pred > 64 || k < 1) return ETO_EINVAL;
    float buf[128];
    for (int64_t n = 0; n < N; ++n) {
        const float *row = obs + n * 2 * T_obs;
        const int mv = eto_row_mode(row, T_obs, mode, static_dist);
        float p[6];
        eto_row_params(row, T_obs, mv, p);
        if (nrm) {
            nrm[0 * N + n] = p[0];
            nrm[1 * N + n] = p[1];
            nrm[2 * N + n] = p[0] - row[2 * (T_obs - 3)];
            nrm[3 * N + n] = p[1] - row[2 * (T_obs - 3) + 1];
        }
        if (flag) flag[n] = (uint8_t)mv;
        if (C_obs) {
            const float *U = mv ? U_obs_m : U_obs_s;
            eto_row_normalize(row, T_obs, p, mv, buf);
            for (int j = 0; j < k; ++j) {
                float acc = 0.0f;
                for (int f = 0; f < 2 * T_obs; ++f) acc = fmaf(U[f * k + j], buf[f], acc);
                C_obs[(int64_t)j * N + n] = acc;
            }
        }
        if (pred && C_pred) {
            const float *U = mv ? U_pred_m : U_pred_s;
            eto_row_normalize(pred + n * 2 * T_pred, T_pred, p, mv, buf);
            for (int j = 0; j < k; ++j) {
                float acc = 0.0f;
                for (int f = 0; f < 2 * T_pred; ++f) acc = fmaf(U[f * k + j], buf[f], acc);
                C_pred[(int64_t)j * N + n] = acc;
            }
        }
    }
    return ETO_OK;
}

/* anchor.py:76-88 (C_anchor[:,None,:] + C_pred) fused with descriptor.py:162-176
 * reconstruction (to_Euclidean_space :75-89 + denormalize normalizer.py:53-62),
 * routed per row like model.py:98-105.  C (k,N,S), A_* (k,S) or NULL,
 * U_pred_* (2T,k), out (S,N,T,2). */
int eto_anchor_reconstruct(const float *C, int64_t N, int S, int k, int T_obs, int T_pred, const float *obs,
                           const float *A_m, const float *A_s, const float *U_m, const float *U_s,
                           int mode, float static_dist, float *out)
{
    if (T_obs < 3 || T_pred > 64 || k > 64) return ETO_EINVAL;
    float v[128], c[64];
    for (int64_t n = 0; n < N; ++n) {
        const float *row = obs + n * 2 * T_obs;
        const int mv = eto_row_mode(row, T_obs, mode, static_dist);
        float p[6];
        eto_row_params(row, T_obs, mv, p);
        const float *U = mv ? U_m : U_s;
        const float *A = mv ? A_m : A_s;
        for (int s = 0; s < S; ++s) {
            for (int j = 0; j < k; ++j) {
                const float cj = C[((int64_t)j * N + n) * S + s];
                c[j] = A ? A[j * S + s] + cj : cj; /* anchor.py:87 */
            }
            for (int f = 0; f < 2 * T_pred; ++f) {
                float acc = 0.0f;
                for (int j = 0; j < k; ++j) acc = fmaf(U[f * k + j], c[j], acc);
                v[f] = acc;
            }
            eto_row_denormalize(v, T_pred, p, mv, out + (((int64_t)s * N + n) * T_pred) * 2);
        }
    }
    return ETO_OK;
}

/* Backward of eto_anchor_reconstruct w.r.t. C (autograd of descriptor.py:173-175;
 * U, A and the normaliser state are detached: descriptor.py:87, anchor.py:87).
 * dtraj (S,N,T,2) -> dC (k,N,S). */
int eto_anchor_reconstruct_bwd(const float *dtraj, int64_t N, int S, int k, int T_obs, int T_pred, const float *obs,
                               const float *U_m, const float *U_s, int mode, float static_dist, float *dC)
{
    if (T_obs < 3 || T_pred > 64) return ETO_EINVAL;
    float g[128];
    for (int64_t n = 0; n < N; ++n) {
        const float *row = obs + n * 2 * T_obs;
        const int mv = eto_row_mode(row, T_obs, mode, static_dist);
        float p[6];
        eto_row_params(row, T_obs, mv, p);
        const float *U = mv ? U_m : U_s;
        for (int s = 0; s < S; ++s) {
            const float *gt = dtraj + (((int64_t)s * N + n) * T_pred) * 2;
            for (int t = 0; t < T_pred; ++t) {
                const float gx = gt[2 * t], gy = gt[2 * t + 1];
                float x = gx * p[2] + gy * p[3];      /* d/d(x) of x*c - y*s, x*s + y*c */
                float y = gx * (-p[3]) + gy * p[2];
                if (mv) {
                    x = x / p[4];
                    y = y / p[4];
                }
                g[2 * t] = x;
                g[2 * t + 1] = y;
            }
            for (int j = 0; j < k; ++j) {
                float acc = 0.0f;
                for (int f = 0; f < 2 * T_pred; ++f) acc = fmaf(U[f * k + j], g[f], acc);
                dC[((int64_t)j * N + n) * S + s] = acc;
            }
        }
    }
    return ETO_OK;
}

/* ------------------------------------------------------------------------- */
/* Fit: Gram matrices + Jacobi eigendecomposition                             */
/* (replaces torch.linalg.svd at descriptor.py:109-114: U = eigvecs of M M^T,  */
/*  sigma = sqrt(eigvals)).                                                    */
/* ------------------------------------------------------------------------- */

/* G_obs (2To x 2To) and G_pred (2Tp x 2Tp) in fp64 for the rows routed to
 * descriptor `which` (1 = moving, 0 = static) under `mode`; count = rows used.
 * descriptor.py:131 (normalise with obs-derived params) + :109 (M = X^T). */
int eto_fit_gram(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred,
                 int mode, float static_dist, int which, double *G_obs, double *G_pred, int64_t *count)
{
    if (T_obs < 3 || T_obs > 64 || T_pred > 64) return ETO_EINVAL;
    const int Do = 2 * T_obs, Dp = 2 * T_pred;
    float xo[128], xp[128];
    memset(G_obs, 0, sizeof(double) * Do * Do);
    memset(G_pred, 0, sizeof(double) * Dp * Dp);
    int64_t cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        const float *row = obs + n * 2 * T_obs;
        const int mv = eto_row_mode(row, T_obs, mode, static_dist);
        if (mv != which) continue;
        float p[6];
        eto_row_params(row, T_obs, mv, p);
        eto_row_normalize(row, T_obs, p, mv, xo);
        eto_row_normalize(pred + n * 2 * T_pred, T_pred, p, mv, xp);
        for (int i = 0; i < Do; ++i)
            for (int j = 0; j < Do; ++j) G_obs[i * Do + j] += (double)xo[i] * (double)xo[j];
        for (int i = 0; i < Dp; ++i)
            for (int j = 0; j < Dp; ++j) G_pred[i * Dp + j] += (double)xp[i] * (double)xp[j];
        ++cnt;
    }
    *count = cnt;
    return ETO_OK;
}

/* Parallel-order (round-robin) Jacobi, fp64, symmetric n x n (n <= 64).  Every round applies
 * n/2 disjoint rotations J = J_1 (+) ... (+) J_{n/2}:  A <- J^T A J, as "all row updates, then all
 * column updates".  The pairing is the circle method of a round-robin tournament.  The HIP
 * kernel et_eigh_topk executes the same arithmetic element by element and must agree bit for
 * bit.  A is destroyed; on return evals[i] = A[i][i], V columns = eigenvectors. */
#define ETO_JACOBI_MAX_SWEEPS 30
static void eto_jacobi(double *A, int n, double *V)
{
    const int m = (n + 1) & ~1; /* players, even; index n (if any) is a dummy */
    int pp[32], qq[32], act[32];
    double cc[32], ss[32];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < ETO_JACOBI_MAX_SWEEPS; ++sweep) {
        /* converged when the largest off-diagonal magnitude is below 1e-10 of the largest diagonal one
         * (maxima, so the test does not depend on any summation order).  Convergence is quadratic: the sweep
         * that meets 1e-10 at its head would end near 1e-20; 1e-15 (rounds 1-5) cost one more sweep for the same
         * fp32 U (csrc/et_fit.hip: kJacobiStop is the same constant). */
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            if (fabs(A[i * n + i]) > diag) diag = fabs(A[i * n + i]);
            for (int j = i + 1; j < n; ++j)
                if (fabs(A[i * n + j]) > off) off = fabs(A[i * n + j]);
        }
        if (off <= 1e-10 * diag) break;
        for (int r = 0; r < m - 1; ++r) {
            for (int i = 0; i < m / 2; ++i) {
                int a, b;
                if (i == 0) {
                    a = m - 1;
                    b = r;
                } else {
                    a = (r + i) % (m - 1);
                    b = (r + (m - 1) - i) % (m - 1);
                }
                const int p = a < b ? a : b, q = a < b ? b : a;
                pp[i] = p;
                qq[i] = q;
                act[i] = 0;
                if (q >= n) continue;
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double app = A[p * n + p], aqq = A[q * n + q];
                /* tan(phi) = t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written with
                 * alpha = aqq - app, beta = 2 apq as t = sgn |beta| / D, D = |alpha| + sqrt(alpha^2 + beta^2); then
                 * c = 1 / sqrt(1 + t^2) = D / g and s = t c = sgn |beta| / g with g = sqrt(D^2 + beta^2): two square
                 * roots and one level of divisions on the critical path instead of three divisions and two roots */
                const double alpha = aqq - app, beta = 2.0 * apq;
                const double h = sqrt(alpha * alpha + beta * beta);
                const double D = fabs(alpha) + h;
                const double g = sqrt(D * D + beta * beta);
                const double sgn = (alpha == 0.0 || ((alpha > 0.0) == (beta > 0.0))) ? 1.0 : -1.0;
                cc[i] = D / g;
                ss[i] = sgn * fabs(beta) / g;
                act[i] = 1;
            }
            for (int i = 0; i < m / 2; ++i) { /* rows p,q of every pair */
                if (!act[i]) continue;
                const int p = pp[i], q = qq[i];
                const double c = cc[i], s_ = ss[i];
                for (int j = 0; j < n; ++j) {
                    const double apj = A[p * n + j], aqj = A[q * n + j];
                    A[p * n + j] = c * apj - s_ * aqj;
                    A[q * n + j] = s_ * apj + c * aqj;
                }
            }
            for (int i = 0; i < m / 2; ++i) { /* columns p,q of every pair, and of V */
                if (!act[i]) continue;
                const int p = pp[i], q = qq[i];
                const double c = cc[i], s_ = ss[i];
                for (int j = 0; j < n; ++j) {
                    const double ajp = A[j * n + p], ajq = A[j * n + q];
                    A[j * n + p] = c * ajp - s_ * ajq;
                    A[j * n + q] = s_ * ajp + c * ajq;
                    const double vjp = V[j * n + p], vjq = V[j * n + q];
                    V[j * n + p] = c * vjp - s_ * vjq;
                    V[j * n + q] = s_ * vjp + c * vjq;
                }
            }
            for (int i = 0; i < m / 2; ++i) {
                if (!act[i]) continue;
                A[pp[i] * n + qq[i]] = 0.0;
                A[qq[i] * n + pp[i]] = 0.0;
            }
        }
    }
}

/* Top-k eigenpairs of symmetric G (n x n, fp64) -> U (n,k) fp32 row-major like
 * nn.Parameter U_*_trunc (descriptor.py:26-27,113), sigma[k] = sqrt(lambda)
 * (descriptor.py:113 S[:k]); descending order; sign convention (LAPACK's is
 * unspecified): the largest-|.| component of every vector is positive. */
int eto_eigh_topk(const double *G, int n, int k, float *U, float *sigma)
{
    if (n < 1 || n > 64 || k < 1 || k > n) return ETO_EINVAL;
    double *A = (double *)malloc(sizeof(double) * n * n * 2);
    double *V = A + n * n;
    int used[64];
    memcpy(A, G, sizeof(double) * n * n);
    eto_jacobi(A, n, V);
    memset(used, 0, sizeof(used));
    for (int j = 0; j < k; ++j) {
        int best = -1;
        for (int i = 0; i < n; ++i)
            if (!used[i] && (best < 0 || A[i * n + i] > A[best * n + best])) best = i;
        used[best] = 1;
        const double lam = A[best * n + best];
        sigma[j] = (float)sqrt(lam > 0.0 ? lam : 0.0);
        int im = 0;
        for (int i = 1; i < n; ++i)
            if (fabs(V[i * n + best]) > fabs(V[im * n + best])) im = i;
        const double sgn = V[im * n + best] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) U[i * k + j] = (float)(sgn * V[i * n + best]);
    }
    free(A);
    return ETO_OK;
}

/* ------------------------------------------------------------------------- */
/* BatchKMeans (EigenTrajectory/kmeans.py)                                     */
/* ------------------------------------------------------------------------- */

/* kmeans.py:59-76 euc_sim for one (a, b) pair, d-major inputs:
 *   y = a^T b ; y *= 2 ; y -= |a|^2 ; y -= |b|^2
 * a^T b is an fmaf chain over d starting from 0 (what the reference's sgemm
 * does on this host: G8), norms are sequential sums of rounded squares. */
static inline float eto_sqnorm(const float *x, int64_t stride, int d)
{
    float s = 0.0f;
    for (int i = 0; i < d; ++i) {
        const float v = x[i * stride];
        s = s + v * v;
    }
    return s;
}

/* torch's own order for `x.pow(2).sum(dim=-2)` (kmeans.py:73-74) -- ATen's CPU sum kernel, outer reduction over the d rows
 * of a (d, count) tensor (SumKernel.cpp, Vectorized<float> of 8 in the build container's torch 2.10.0): the columns are
 * handled in blocks of 32 (4 vectors) whose sums run over the rows in cascade order (sequential for d < 16: what
 * eto_sqnorm does); the count % 32 columns after the last full block go through row_sum instead: rows dealt onto 4
 * lanes (row mod 4), the d % 4 leftover rows added to lane 0, then lanes 1, 2, 3 added to lane 0.  For d = 6:
 * ((((s0 + s4) + s5) + s1) + s2) + s3.  |b|^2 of K = 20 centroids takes that order in every column, |a|^2 of the points
 * only in the last N % 32.  Verified bit for bit against torch (tests/test_oracle_golden.py). */
static float eto_cascade_f32(const float *v, int64_t stride, int64_t size)
{
    int lp = 0;
    {
        int64_t w = size - 1;
        int l = 0;
        while (size > 2 && w > 0) { w >>= 1; ++l; }
        if (size <= 2) l = 1;
        lp = l / 4 > 4 ? l / 4 : 4;
    }
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    while (i + step <= size) {
        for (int64_t q = 0; q < step; ++q, ++i) acc[0] = acc[0] + v[i * stride];
        for (int lv = 1; lv < 4; ++lv) {
            acc[lv] = acc[lv] + acc[lv - 1];
            acc[lv - 1] = 0.f;
            if ((i & (lmask << (lv * lp))) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] = acc[0] + v[i * stride];
    for (int lv = 1; lv < 4; ++lv) acc[0] = acc[0] + acc[lv];
    return acc[0];
}

static float eto_row_sum_f32(const float *v, int64_t size)
{
    const int64_t s4 = size / 4;
    float lane[4];
    for (int k = 0; k < 4; ++k) lane[k] = eto_cascade_f32(v + k, 4, s4);
    for (int64_t i = s4 * 4; i < size; ++i) lane[0] = lane[0] + v[i];
    for (int k = 1; k < 4; ++k) lane[0] = lane[0] + lane[k];
    return lane[0];
}

float eto_inner_sum_f32(const float *v, int64_t size);

/* |x|^2 of column `pos` of `count`: ref = 0 this build's sequential order, ref = 1 torch's (above) */
static inline float eto_sqnorm_at(const float *x, int64_t stride, int d, int64_t pos, int64_t count, int ref)
{
    if (!ref) return eto_sqnorm(x, stride, d);
    /* fewer than 8 columns (one vector): the scalar kernel, blocks of 4 columns instead of 32 */
    const int64_t seq_cols = count < 8 ? count / 4 * 4 : count / 32 * 32;
    float sq[64];
    for (int i = 0; i < d; ++i) {
        const float v = x[i * stride];
        sq[i] = v * v;
    }
    /* a single column of d >= 8 rows is a CONTIGUOUS reduction for ATen: its inner-sum order */
    if (count == 1 && d >= 8) return eto_inner_sum_f32(sq, d);
    return pos < seq_cols ? eto_cascade_f32(sq, 1, d) : eto_row_sum_f32(sq, d);
}

static inline float eto_sim(const float *a, int64_t sa, float an, const float *b, int64_t sb, float bn, int d)
{
    float y = 0.0f;
    for (int i = 0; i < d; ++i) y = fmaf(a[i * sa], b[i * sb], y);
    y = y * 2.0f;
    y = y - an;
    y = y - bn;
    return y;
}

static int eto_euc_sim_impl(const float *a, const float *b, int d, int64_t m, int64_t n, float *y, int ref)
{
    if (d < 1 || d > 64) return ETO_EINVAL;
    for (int64_t i = 0; i < m; ++i) {
        const float an = eto_sqnorm_at(a + i, m, d, i, m, ref);
        for (int64_t j = 0; j < n; ++j) {
            const float bn = eto_sqnorm_at(b + j, n, d, j, n, ref);
            y[i * n + j] = eto_sim(a + i, m, an, b + j, n, bn, d);
        }
    }
    return ETO_OK;
}
int eto_euc_sim(const float *a, const float *b, int d, int64_t m, int64_t n, float *y) { return eto_euc_sim_impl(a, b, d, m, n, y, 0); }
/* ... with both norms in torch's order: every bit of the reference's euc_sim */
int eto_euc_sim_ref(const float *a, const float *b, int d, int64_t m, int64_t n, float *y) { return eto_euc_sim_impl(a, b, d, m, n, y, 1); }

/* torch.max semantics (kmeans.py:156): NaN beats everything, first index wins. */
static inline int eto_gt_nanmax(float cand, float best)
{
    return (cand > best) || (isnan(cand) && !isnan(best));
}

/* kmeans.py:143-158 get_labels: labels (int64) and maxsims for X (d,N) vs C (d,K) */
static int eto_kmeans_assign_impl(const float *X, int64_t N, int d, const float *C, int K, int64_t *labels, float *maxsims,
                                  int ref)
{
    if (K < 1 || K > 255 || d < 1 || d > 64) return ETO_EINVAL;
    float bn[256];
    for (int j = 0; j < K; ++j) bn[j] = eto_sqnorm_at(C + j, K, d, j, K, ref);
    for (int64_t n = 0; n < N; ++n) {
        const float an = eto_sqnorm_at(X + n, N, d, n, N, ref);
        float best = eto_sim(X + n, N, an, C, K, bn[0], d);
        int lb = 0;
        for (int j = 1; j < K; ++j) {
            const float y = eto_sim(X + n, N, an, C + j, K, bn[j], d);
            if (eto_gt_nanmax(y, best)) {
                best = y;
                lb = j;
            }
        }
        if (labels) labels[n] = lb;
        if (maxsims) maxsims[n] = best;
    }
    return ETO_OK;
}
int eto_kmeans_assign(const float *X, int64_t N, int d, const float *C, int K, int64_t *labels, float *maxsims)
{
    return eto_kmeans_assign_impl(X, N, d, C, K, labels, maxsims, 0);
}
int eto_kmeans_assign_ref(const float *X, int64_t N, int d, const float *C, int K, int64_t *labels, float *maxsims)
{
    return eto_kmeans_assign_impl(X, N, d, C, K, labels, maxsims, 1);
}

/* kmeans.py:78-112 kmeanspp: c_0 = X[:, first_index]; c_i = the point whose
 * max similarity to c_0..c_{i-1} is smallest (argmin: first index on ties,
 * NaN counts as smallest like torch.argmin).  index_out[i] = chosen indices. */
static int eto_kmeans_init_farthest_impl(const float *X, int64_t N, int d, int K, int64_t first_index,
                                         float *C0 /*d,K*/, int64_t *index_out /*K or NULL*/, int ref)
{
    if (N < 1 || first_index < 0 || first_index >= N) return ETO_EINVAL;
    float *best = (float *)malloc(sizeof(float) * (size_t)N);
    float cb[64];
    if (d > 64) { free(best); return ETO_EINVAL; }
    int64_t idx = first_index;
    for (int i = 0; i < K; ++i) {
        for (int t = 0; t < d; ++t) {
            cb[t] = X[(int64_t)t * N + idx];
            C0[t * K + i] = cb[t];
        }
        if (index_out) index_out[i] = idx;
        if (i == K - 1) break;
        const float bn = eto_sqnorm(cb, 1, d);
        /* reference order: the reference re-evaluates euc_sim against ALL i + 1 current centroids at every step
         * (kmeans.py:94-96), and the order of a centroid's |b|^2 depends on its column and on the column count */
        float bns[256];
        if (ref)
            for (int j = 0; j <= i; ++j) bns[j] = eto_sqnorm_at(C0 + j, K, d, j, i + 1, 1);
        int64_t arg = 0;
        float argv = 0.0f;
        for (int64_t n = 0; n < N; ++n) {
            float b;
            if (ref) {
                const float an = eto_sqnorm_at(X + n, N, d, n, N, 1);
                b = eto_sim(X + n, N, an, C0, K, bns[0], d);
                for (int j = 1; j <= i; ++j) {
                    const float y = eto_sim(X + n, N, an, C0 + j, K, bns[j], d);
                    if (eto_gt_nanmax(y, b)) b = y;
                }
            } else {
                const float an = eto_sqnorm(X + n, N, d);
                const float y = eto_sim(X + n, N, an, cb, 1, bn, d);
                b = (i == 0) ? y : best[n];
                if (i > 0 && eto_gt_nanmax(y, b)) b = y;
            }
            best[n] = b;
            if (n == 0 || (b < argv && !isnan(argv)) || (isnan(b) && !isnan(argv))) {
                arg = n;
                argv = b;
            }
        }
        idx = arg;
    }
    free(best);
    return ETO_OK;
}
int eto_kmeans_init_farthest(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0, int64_t *index_out)
{
    return eto_kmeans_init_farthest_impl(X, N, d, K, first_index, C0, index_out, 0);
}
int eto_kmeans_init_farthest_ref(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0, int64_t *index_out)
{
    return eto_kmeans_init_farthest_impl(X, N, d, K, first_index, C0, index_out, 1);
}

/* ---- exactly-associative fixed-point accumulation (design of this build) ----
 * The reference sums per-cluster coordinates in fp32 in torch's reduction
 * order (kmeans.py:180-182), which no parallel implementation can reproduce.
 * This build defines the sums exactly instead: every x is converted to a
 * 64-bit fixed-point integer with `frac` fractional bits (truncation toward
 * zero, exact because the scale is a power of two), integers are summed
 * (associative, so any block/GPU partition gives identical bits), and the mean
 * is rounded once.  frac = 62 - E - bits(N) where max|x| < 2^E. */
static int eto_exponent_above(double m)
{ /* smallest E with m < 2^E (E = 0 for m == 0) */
    if (!(m > 0.0)) return 0;
    int e;
    frexp(m, &e); /* m = f * 2^e, f in [0.5,1) -> m < 2^e */
    return e;
}

static int eto_bits_for(int64_t n)
{
    int b = 0;
    while (((int64_t)1 << b) <= n && b < 62) ++b;
    return b;
}

static inline int64_t eto_to_fixed(float x, int frac)
{
    return (int64_t)ldexp((double)x, frac); /* exact scale, C cast truncates toward zero */
}

int eto_kmeans_frac_bits(double max_abs, int64_t n_total) { return 62 - eto_exponent_above(max_abs) - eto_bits_for(n_total); }

/* similarity bound for the inertia accumulator: |sim| <= 4 d m^2, m = max(|x|,|c|) */
int eto_kmeans_sim_frac_bits(double max_abs_x, double max_abs_c, int d, int64_t n_total)
{
    const double m = max_abs_x > max_abs_c ? max_abs_x : max_abs_c;
    return 62 - eto_exponent_above(4.0 * d * m * m) - eto_bits_for(n_total);
}

/* One Lloyd step on a shard: labels (kmeans.py:230), then exact partial sums
 * for compute_centroids (:231) and calculate_inertia (:234).
 * sums (d,K) int64, counts (K) int64, sim_sum int64, nan_count int64. */
static int eto_kmeans_assign_accumulate_impl(const float *X, int64_t N, int d, const float *C, int K, int frac, int sim_frac,
                                             int64_t *labels, int64_t *sums, int64_t *counts, int64_t *sim_sum,
                                             int64_t *nan_count, int ref)
{
    if (K < 1 || K > 255 || d < 1 || d > 64) return ETO_EINVAL;
    float bn[256];
    for (int j = 0; j < K; ++j) bn[j] = eto_sqnorm_at(C + j, K, d, j, K, ref);
    memset(sums, 0, sizeof(int64_t) * d * K);
    memset(counts, 0, sizeof(int64_t) * K);
    int64_t ss = 0, nn = 0;
    for (int64_t n = 0; n < N; ++n) {
        const float an = eto_sqnorm_at(X + n, N, d, n, N, ref);
        float best = eto_sim(X + n, N, an, C, K, bn[0], d);
        int lb = 0;
        for (int j = 1; j < K; ++j) {
            const float y = eto_sim(X + n, N, an, C + j, K, bn[j], d);
            if (eto_gt_nanmax(y, best)) {
                best = y;
                lb = j;
            }
        }
        labels[n] = lb;
        counts[lb] += 1;
        for (int t = 0; t < d; ++t) sums[t * K + lb] += eto_to_fixed(X[(int64_t)t * N + n], frac);
        if (isnan(best) || isinf(best)) nn += 1;
        else ss += eto_to_fixed(best, sim_frac);
    }
    *sim_sum = ss;
    *nan_count = nn;
    return ETO_OK;
}
int eto_kmeans_assign_accumulate(const float *X, int64_t N, int d, const float *C, int K, int frac, int sim_frac,
                                 int64_t *labels, int64_t *sums, int64_t *counts, int64_t *sim_sum, int64_t *nan_count)
{
    return eto_kmeans_assign_accumulate_impl(X, N, d, C, K, frac, sim_frac, labels, sums, counts, sim_sum, nan_count, 0);
}

/* kmeans.py:45-51 sums the d K squared differences with torch's fp32 reduction.  The summation ORDER is this build's own
 * choice (the result is compared with the reference's at a tolerance): fp64, blocks of 256 consecutive terms (zero padded),
 * each reduced by the balanced tree x[i] += x[i + s] for s = 1, 2, 4, ..., 128 -- seven dependent additions, a wavefront's
 * natural reduction -- and the block results added in block order. */
static double eto_error_sum(const float *sq, int n)
{
    double total = 0.0;
    for (int b0 = 0; b0 < n; b0 += 256) {
        double x[256];
        for (int i = 0; i < 256; ++i) x[i] = b0 + i < n ? (double)sq[b0 + i] : 0.0;
        for (int s = 1; s < 256; s <<= 1)
            for (int i = 0; i + s < 256; i += 2 * s) x[i] = x[i] + x[i + s];
        total = total + x[0];
    }
    return total;
}

/* Centroid update + convergence scalars from (all-reduced) exact sums:
 * kmeans.py:180-182 (mean; empty cluster -> 0/0 = NaN), :45-51 (error),
 * :53-57 (inertia), :239 (error <= tol).  Returns converged flag in *done. */
int eto_kmeans_update(const int64_t *sums, const int64_t *counts, int64_t sim_sum, int64_t nan_count,
                      int64_t n_total, int d, int K, int frac, int sim_frac, float tol,
                      const float *C_old, float *C_new, float *error, float *inertia, int *done)
{
    float *sq = (float *)malloc(sizeof(float) * (size_t)d * K);
    if (!sq) return ETO_EINVAL;
    for (int t = 0; t < d; ++t)
        for (int j = 0; j < K; ++j) {
            float c;
            if (counts[j] == 0) c = NAN;
            else c = (float)(ldexp((double)sums[t * K + j], -frac) / (double)counts[j]);
            C_new[t * K + j] = c;
            const float diff = C_old[t * K + j] - c;
            sq[t * K + j] = diff * diff;
        }
    *error = (float)eto_error_sum(sq, d * K);
    free(sq);
    *inertia = nan_count > 0 ? NAN : (float)(-(ldexp((double)sim_sum, -sim_frac) / (double)n_total));
    *done = (*error <= tol) ? 1 : 0;
    return ETO_OK;
}

/* max |x| ignoring NaN; *bad is set when any element is NaN or +-inf */
static double eto_max_abs(const float *x, int64_t n, int *bad)
{
    double m = 0.0;
    int b = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double a = fabs((double)x[i]);
        if (a > m) m = a;
        if (!(a <= 3.4028234663852886e38)) b = 1;
    }
    if (bad) *bad = b;
    return m;
}

/* kmeans.py:200-259 fit for one batch element, one redo, given initial centroids.
 * labels = labels of the LAST assignment (:230, pre-update centroids), centroids =
 * last update; trace (max_iter,2) gets (error, inertia) per iteration. */
int eto_kmeans_fit(const float *X, int64_t N, int d, int K, const float *C_init, int max_iter, float tol,
                   float *centroids, int64_t *labels, int *n_iter, float *error, float *inertia, float *trace)
{
    if (K < 1 || K > 255 || d < 1 || d > 64 || N < 1) return ETO_EINVAL;
    int64_t *sums = (int64_t *)malloc(sizeof(int64_t) * (d * K + K));
    int64_t *counts = sums + d * K;
    float *cur = (float *)malloc(sizeof(float) * d * K * 2);
    float *nxt = cur + d * K;
    memcpy(cur, C_init, sizeof(float) * d * K);
    int bad = 0;
    const double mx = eto_max_abs(X, (int64_t)d * N, &bad);
    if (bad) { free(sums); free(cur); return ETO_EINVAL; } /* non-finite data: undefined in this build */
    const int frac = eto_kmeans_frac_bits(mx, N);
    int it = 0, done = 0;
    float err = 0.0f, ine = 0.0f;
    for (it = 0; it < max_iter; ++it) {
        const int sfrac = eto_kmeans_sim_frac_bits(mx, eto_max_abs(cur, d * K, NULL), d, N);
        int64_t ss, nn;
        eto_kmeans_assign_accumulate(X, N, d, cur, K, frac, sfrac, labels, sums, counts, &ss, &nn);
        eto_kmeans_update(sums, counts, ss, nn, N, d, K, frac, sfrac, tol, cur, nxt, &err, &ine, &done);
        memcpy(cur, nxt, sizeof(float) * d * K);
        if (trace) {
            trace[2 * it] = err;
            trace[2 * it + 1] = ine;
        }
        if (done) {
            ++it;
            break;
        }
    }
    memcpy(centroids, cur, sizeof(float) * d * K);
    *n_iter = it;
    *error = err;
    *inertia = ine;
    free(sums);
    free(cur);
    return ETO_OK;
}

/* ------------------------------------------------------------------------- */
/* kmeans.py:180-182 in the REFERENCE's summation order ("reference-order" sums) */
/* ------------------------------------------------------------------------- */
/* `(data.unsqueeze(-1) * mask.unsqueeze(-3)).sum(dim=-2)` reduces a (d, N, K) fp32 tensor over N with ATen's CPU sum
 * kernel (aten/src/ATen/native/cpu/SumKernel.cpp; third-party, not under /root/reference; torch 2.10.0 in the build
 * container): every output column (t, j) is the "cascade sum" of its N terms  x[t,n]·[label_n == j]:
 *   row_sum:        the terms are dealt round-robin onto 4 lanes (n mod 4; the N mod 4 leftover terms are added to
 *                   lane 0 afterwards, then lanes 1, 2, 3 are added to lane 0 in that order);
 *   multi_row_sum:  per lane a 4-level cascade over its size = N/4 terms with level_step = 2^max(4, ceil_log2(size)/4):
 *                   acc0 sums level_step consecutive terms and is dumped into acc1; acc1 into acc2 every level_step
 *                   dumps; acc2 into acc3 likewise; leftover terms go to acc0; result ((acc0 + acc1) + acc2) + acc3.
 * The order per column does not depend on the vector width or the thread count (threads split the output columns, never
 * the reduced dimension).  Pinned bit for bit against torch on random inputs by tests/test_oracle_golden.py
 * (test_reforder_sums_equal_torch) and through whole BatchKMeans runs by the G7c fixture.
 * A term that is not a member contributes x·0 = ±0, which leaves a running fp32 sum that started at +0 unchanged, so
 * only members are added. */
static int eto_ceil_log2(int64_t x)
{
    if (x <= 2) return 1;
    int l = 0;
    int64_t v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

static float eto_cascade_lane(const float *x, const int64_t *labels, int j, int64_t size, int lane)
{
    const int lp_raw = eto_ceil_log2(size) / 4;
    const int lp = lp_raw > 4 ? lp_raw : 4;
    const int64_t step = (int64_t)1 << lp, lmask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t i = 0;
    while (i + step <= size) {
        for (int64_t q = 0; q < step; ++q, ++i) {
            const int64_t n = 4 * i + lane;
            if (labels[n] == j) acc[0] = acc[0] + x[n];
        }
        for (int lv = 1; lv < 4; ++lv) {
            acc[lv] = acc[lv] + acc[lv - 1];
            acc[lv - 1] = 0.f;
            if ((i & (lmask << (lv * lp))) != 0) break;
        }
    }
    for (; i < size; ++i) {
        const int64_t n = 4 * i + lane;
        if (labels[n] == j) acc[0] = acc[0] + x[n];
    }
    for (int lv = 1; lv < 4; ++lv) acc[0] = acc[0] + acc[lv];
    return acc[0];
}

/* sums (d,K) fp32 of the members of every cluster, in the reference's order */
int eto_kmeans_reforder_sums(const float *X, int64_t N, int d, int K, const int64_t *labels, float *sums)
{
    if (K < 1 || K > 255 || d < 1 || d > 64 || N < 0) return ETO_EINVAL;
    const int64_t size = N / 4;
    for (int t = 0; t < d; ++t)
        for (int j = 0; j < K; ++j) {
            const float *x = X + (int64_t)t * N;
            float lane[4];
            for (int k = 0; k < 4; ++k) lane[k] = eto_cascade_lane(x, labels, j, size, k);
            for (int64_t n = size * 4; n < N; ++n)
                if (labels[n] == j) lane[0] = lane[0] + x[n];
            for (int k = 1; k < 4; ++k) lane[0] = lane[0] + lane[k];
            sums[t * K + j] = lane[0];
        }
    return ETO_OK;
}

/* `diff.sum()` of kmeans.py:50 on the contiguous d K squared differences -- ATen's inner (contiguous) reduction: the
 * first size / 8 * 8 values as 8-float vectors through row_sum (lane-wise), the rest added to a scalar that starts at 0,
 * then the 8 lanes of the vector sum added to it in lane order.  Verified bit for bit against torch. */
float eto_inner_sum_f32(const float *v, int64_t size)
{
    if (size < 8) return eto_row_sum_f32(v, size); /* less than one vector: the scalar kernel's row_sum */
    const int64_t nv = size / 8;
    float lanes[8];
    for (int l = 0; l < 8; ++l) {
        /* row_sum over nv vectors for lane l: vectors dealt onto 4 ilp slots */
        const int64_t s4 = nv / 4;
        float slot[4];
        for (int k = 0; k < 4; ++k) slot[k] = eto_cascade_f32(v + 8 * k + l, 32, s4);
        for (int64_t i = s4 * 4; i < nv; ++i) slot[0] = slot[0] + v[8 * i + l];
        for (int k = 1; k < 4; ++k) slot[0] = slot[0] + slot[k];
        lanes[l] = slot[0];
    }
    float acc = 0.f;
    for (int64_t i = nv * 8; i < size; ++i) acc = acc + v[i];
    for (int l = 0; l < 8; ++l) acc = acc + lanes[l];
    return acc;
}

/* kmeans.py:200-259 like eto_kmeans_fit, with the centroids of kmeans.py:180-182 formed as the reference forms them:
 * fp32 cascade sum / (float)count (0/0 = NaN for an empty cluster), the norms inside euc_sim in torch's order
 * (eto_sqnorm_at) and the error of kmeans.py:45-51 in torch's order (eto_inner_sum_f32).  The inertia (only printed by
 * the reference) stays this build's exact sum. */
int eto_kmeans_fit_reforder(const float *X, int64_t N, int d, int K, const float *C_init, int max_iter, float tol,
                            float *centroids, int64_t *labels, int *n_iter, float *error, float *inertia, float *trace)
{
    if (K < 1 || K > 255 || d < 1 || d > 64 || N < 1) return ETO_EINVAL;
    int64_t *sums = (int64_t *)malloc(sizeof(int64_t) * (d * K + K));
    int64_t *counts = sums + d * K;
    float *cur = (float *)malloc(sizeof(float) * d * K * 4);
    float *nxt = cur + d * K, *fs = cur + 2 * d * K, *sq = cur + 3 * d * K;
    memcpy(cur, C_init, sizeof(float) * d * K);
    int bad = 0;
    const double mx = eto_max_abs(X, (int64_t)d * N, &bad);
    if (bad) { free(sums); free(cur); return ETO_EINVAL; }
    const int frac = eto_kmeans_frac_bits(mx, N);
    int it = 0, done = 0;
    float err = 0.0f, ine = 0.0f;
    for (it = 0; it < max_iter; ++it) {
        const int sfrac = eto_kmeans_sim_frac_bits(mx, eto_max_abs(cur, d * K, NULL), d, N);
        int64_t ss, nn;
        eto_kmeans_assign_accumulate_impl(X, N, d, cur, K, frac, sfrac, labels, sums, counts, &ss, &nn, 1);
        eto_kmeans_reforder_sums(X, N, d, K, labels, fs);
        for (int e = 0; e < d * K; ++e) {
            const float c = fs[e] / (float)counts[e % K];
            nxt[e] = c;
            const float diff = cur[e] - c;
            sq[e] = diff * diff;
        }
        err = eto_inner_sum_f32(sq, d * K);
        ine = nn > 0 ? NAN : (float)(-(ldexp((double)ss, -sfrac) / (double)N));
        done = (err <= tol) ? 1 : 0;
        memcpy(cur, nxt, sizeof(float) * d * K);
        if (trace) {
            trace[2 * it] = err;
            trace[2 * it + 1] = ine;
        }
        if (done) {
            ++it;
            break;
        }
    }
    memcpy(centroids, cur, sizeof(float) * d * K);
    *n_iter = it;
    *error = err;
    *inertia = ine;
    free(sums);
    free(cur);
    return ETO_OK;
}

/* kmeans.py:228-240 for a batch of l problems in the reference's orders: every iteration assigns and updates ALL problems
 * (eto_kmeans_fit_reforder's step), then ONE error -- `diff.sum()` over the whole contiguous (l, d, K) tensor of squared
 * centroid differences, ATen's inner-sum order (kmeans.py:45-51, 232) -- is compared with tol (kmeans.py:239): all problems
 * stop in the same iteration.  X (l,d,N), C_init / centroids (l,d,K), labels (l,N), inertia (l), trace (max_iter,2):
 * (joint error, mean of the problems' inertias -- kmeans.py:234 is one mean over the batch). */
int eto_kmeans_fit_reforder_batch(const float *X, int64_t N, int d, int K, int l, const float *C_init, int max_iter, float tol,
                                  float *centroids, int64_t *labels, int *n_iter, float *error, float *inertia, float *trace)
{
    if (K < 1 || K > 255 || d < 1 || d > 64 || N < 1 || l < 1) return ETO_EINVAL;
    const int dk = d * K;
    int64_t *sums = (int64_t *)malloc(sizeof(int64_t) * (dk + K));
    int64_t *counts = sums + dk;
    float *cur = (float *)malloc(sizeof(float) * dk * (size_t)l * 2 + sizeof(float) * dk);
    float *sq = cur + (size_t)dk * l, *fs = sq + (size_t)dk * l;
    double *mx = (double *)malloc(sizeof(double) * l);
    memcpy(cur, C_init, sizeof(float) * dk * (size_t)l);
    for (int b = 0; b < l; ++b) {
        int bad = 0;
        mx[b] = eto_max_abs(X + (int64_t)b * d * N, (int64_t)d * N, &bad);
        if (bad) { free(sums); free(cur); free(mx); return ETO_EINVAL; }
    }
    int it = 0, done = 0;
    float err = 0.0f;
    for (it = 0; it < max_iter; ++it) {
        double ine_mean = 0.0;
        for (int b = 0; b < l; ++b) {
            const float *Xb = X + (int64_t)b * d * N;
            float *cb = cur + (size_t)b * dk;
            const int frac = eto_kmeans_frac_bits(mx[b], N);
            const int sfrac = eto_kmeans_sim_frac_bits(mx[b], eto_max_abs(cb, dk, NULL), d, N);
            int64_t ss, nn;
            eto_kmeans_assign_accumulate_impl(Xb, N, d, cb, K, frac, sfrac, labels + (int64_t)b * N, sums, counts, &ss, &nn, 1);
            eto_kmeans_reforder_sums(Xb, N, d, K, labels + (int64_t)b * N, fs);
            for (int e = 0; e < dk; ++e) {
                const float c = fs[e] / (float)counts[e % K];
                const float diff = cb[e] - c;
                sq[(size_t)b * dk + e] = diff * diff;
                cb[e] = c;
            }
            inertia[b] = nn > 0 ? NAN : (float)(-(ldexp((double)ss, -sfrac) / (double)N));
            ine_mean += (double)inertia[b];
        }
        err = eto_inner_sum_f32(sq, (int64_t)dk * l);
        done = (err <= tol) ? 1 : 0;
        if (trace) {
            trace[2 * it] = err;
            trace[2 * it + 1] = (float)(ine_mean / l);
        }
        if (done) {
            ++it;
            break;
        }
    }
    memcpy(centroids, cur, sizeof(float) * dk * (size_t)l);
    *n_iter = it;
    *error = err;
    free(sums);
    free(cur);
    free(mx);
    return ETO_OK;
}
