// et_train.hip -- the TRAINING form of a wrapper call on one scene (EigenTrajectory/model.py:58-125 with pred_traj;
// utils/trainer.py:126-152 is the loop that calls it): projection of the observations AND of the ground truth, and --
// after the predictor -- anchor refinement + reconstruction + the three loss terms of model.py:119-123 in one launch,
// with their gradient w.r.t. the predictor's output in one more.
//
// The reference's training step is one scene (N <= a few hundred pedestrians) per call: the work is microseconds, the
// cost is launches and framework operators (the composite form of this build: ~15 operators forward, as many autograd
// nodes backward, 550 us per 57-pedestrian scene against 32 us for the inference form).  Here:
//   et_scene_project_train   C_obs (k,N), nrm (4,N), scene-centred obs_ori (2,N), C_gt (k,N) = projection of the ground
//                            truth with the observation's normaliser (descriptor.py:144-160), flag (N)      [1 launch]
//   et_wrapper_losses_fwd    recon (S,N,T,2) = denorm(U (A[:,s] + C[:,n,s])) (anchor.py:76-88, descriptor.py:162-176) and
//                            loss_eigentraj     = mean_n min_s || A[:,s] + C[:,n,s] - C_gt[:,n] ||_2        (model.py:119)
//                            loss_euclidean_ade = mean_n min_s mean_t || recon[s,n,t] - gt[n,t] ||_2        (model.py:120-121)
//                            loss_euclidean_fde = mean_n min_s || recon[s,n,T-1] - gt[n,T-1] ||_2           (model.py:122-123)
//                            + the arg-min sample of each term per pedestrian (for the backward)             [1 launch]
//   et_wrapper_losses_bwd    dC (k,N,S) for given d(loss) factors: the three terms only reach the sample that attains
//                            their minimum (torch.amin's gradient; ties have measure zero), through
//                            d recon / d C = (g @ R) / sca @ U (ops._reconstruct_bwd's map)                  [1 launch]
// One workgroup (scenes: N <= ET_SCENE_MAX_N); lane = pedestrian; the means are fixed-order sums (deterministic).
#include "et_common.h"

namespace et {

constexpr int kTrThreads = 256;

__device__ __forceinline__ RowNorm nrm_row(const float *__restrict__ nrm, int N, int n, int mode, float static_dist) {
    return row_norm(nrm[n], nrm[N + n], nrm[2 * N + n], nrm[3 * N + n], mode, static_dist);
}

// fixed-order sum of one value per thread over the workgroup -> every thread
__device__ __forceinline__ float block_sum(float v, float *sW) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0) sW[threadIdx.x / kWave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < kTrThreads / kWave; ++w) t += sW[w];
    return t;
}

__global__ __launch_bounds__(kTrThreads) void scene_project_train_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int N, int T_obs, int T_pred, int k,
    const float *__restrict__ U_obs_m, const float *__restrict__ U_obs_s, const float *__restrict__ U_pred_m,
    const float *__restrict__ U_pred_s, int mode, float static_dist, float *__restrict__ C_obs, float *__restrict__ nrm,
    float *__restrict__ obs_ori, float *__restrict__ C_gt, uint8_t *__restrict__ flag) {
    __shared__ float sW[kTrThreads / kWave];
    float sx = 0.f, sy = 0.f;
    for (int n = threadIdx.x; n < N; n += kTrThreads) {
        const float *row = obs + (int64_t)n * 2 * T_obs;
        const float ox = row[2 * (T_obs - 1)], oy = row[2 * (T_obs - 1) + 1];
        const float dx = ox - row[2 * (T_obs - 3)], dy = oy - row[2 * (T_obs - 3) + 1];
        const RowNorm p = row_norm(ox, oy, dx, dy, mode, static_dist);
        nrm[n] = ox;
        nrm[N + n] = oy;
        nrm[2 * N + n] = dx;
        nrm[3 * N + n] = dy;
        if (flag) flag[n] = (uint8_t)p.mv;
        sx += ox;
        sy += oy;
        const float *U = p.mv ? U_obs_m : U_obs_s;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
            for (int t = 0; t < T_obs; ++t) {
                float a, b;
                normalize_point(p, row[2 * t], row[2 * t + 1], a, b);
                acc = fmaf(U[(2 * t) * k + j], a, acc);
                acc = fmaf(U[(2 * t + 1) * k + j], b, acc);
            }
            C_obs[(int64_t)j * N + n] = acc;
        }
        const float *prow = pred + (int64_t)n * 2 * T_pred;
        const float *Up = p.mv ? U_pred_m : U_pred_s;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
            for (int t = 0; t < T_pred; ++t) {
                float a, b;
                normalize_point(p, prow[2 * t], prow[2 * t + 1], a, b);
                acc = fmaf(Up[(2 * t) * k + j], a, acc);
                acc = fmaf(Up[(2 * t + 1) * k + j], b, acc);
            }
            C_gt[(int64_t)j * N + n] = acc;
        }
    }
    const float mx = block_sum(sx, sW) / (float)N;
    const float my = block_sum(sy, sW) / (float)N;
    for (int n = threadIdx.x; n < N; n += kTrThreads) {  // this thread wrote nrm[n], nrm[N + n] itself
        obs_ori[n] = nrm[n] - mx;
        obs_ori[N + n] = nrm[N + n] - my;
    }
}

// K = coefficient count (<= ET_MAX_K), T = prediction steps (<= ET_MAX_T): runtime, the loops are short
__global__ __launch_bounds__(kTrThreads) void wrapper_losses_fwd_kernel(
    const float *__restrict__ C, int N, int S, int k, int T, const float *__restrict__ nrm, const float *__restrict__ A_m,
    const float *__restrict__ A_s, const float *__restrict__ U_m, const float *__restrict__ U_s, int mode, float static_dist,
    const float *__restrict__ C_gt, const float *__restrict__ gt, float *__restrict__ recon, float *__restrict__ best,
    int *__restrict__ arg, float *__restrict__ losses) {
    __shared__ float sW[kTrThreads / kWave];
    float se = 0.f, sa = 0.f, sf = 0.f;
    const float inf = __int_as_float(0x7f800000);
    for (int n = threadIdx.x; n < N; n += kTrThreads) {
        const RowNorm p = nrm_row(nrm, N, n, mode, static_dist);
        const float *A = p.mv ? A_m : A_s;
        const float *U = p.mv ? U_m : U_s;
        float cg[ET_MAX_K];
        for (int j = 0; j < k; ++j) cg[j] = C_gt[(int64_t)j * N + n];
        const float *g = gt + (int64_t)n * 2 * T;
        float be = inf, ba = inf, bf = inf;
        int ie = 0, ia = 0, ifd = 0;
        for (int s = 0; s < S; ++s) {
            float cp[ET_MAX_K];
            float e2 = 0.f;
            for (int j = 0; j < k; ++j) {
                cp[j] = (A ? A[j * S + s] : 0.f) + C[((int64_t)j * N + n) * S + s];  // anchor.py:87
                const float df = cp[j] - cg[j];
                e2 = fmaf(df, df, e2);
            }
            const float e = sqrtf(e2);
            float ad = 0.f, fd = 0.f;
            float *out = recon + (((int64_t)s * N + n) * T) * 2;
            for (int t = 0; t < T; ++t) {
                float a = 0.f, b = 0.f;
                for (int j = 0; j < k; ++j) {  // descriptor.py:86-88
                    a = fmaf(U[(2 * t) * k + j], cp[j], a);
                    b = fmaf(U[(2 * t + 1) * k + j], cp[j], b);
                }
                float x, y;
                denormalize_point(p, a, b, x, y);
                out[2 * t] = x;
                out[2 * t + 1] = y;
                const float ex = x - g[2 * t], ey = y - g[2 * t + 1];
                const float dist = sqrtf(fmaf(ex, ex, ey * ey));
                ad += dist;
                fd = dist;  // the last step's stays
            }
            ad = ad / (float)T;
            // torch.amin / argmin: first minimum; a NaN takes over (and stays)
            if (e < be || (e != e && be == be)) be = e, ie = s;
            if (ad < ba || (ad != ad && ba == ba)) ba = ad, ia = s;
            if (fd < bf || (fd != fd && bf == bf)) bf = fd, ifd = s;
        }
        best[n] = be;
        best[N + n] = ba;
        best[2 * N + n] = bf;
        arg[n] = ie;
        arg[N + n] = ia;
        arg[2 * N + n] = ifd;
        se += be;
        sa += ba;
        sf += bf;
    }
    const float te = block_sum(se, sW), ta = block_sum(sa, sW), tf = block_sum(sf, sW);
    if (threadIdx.x == 0) {
        losses[0] = te / (float)N;
        losses[1] = ta / (float)N;
        losses[2] = tf / (float)N;
    }
}

// dC = sum over the three terms of g_term / N * d(term's minimum)/dC; g_* = the upstream gradient of each loss (device
// scalars; NULL = that term was not differentiated)
__global__ __launch_bounds__(kTrThreads) void wrapper_losses_bwd_kernel(
    const float *__restrict__ g_e, const float *__restrict__ g_ade, const float *__restrict__ g_fde,
    const float *__restrict__ C, int N, int S, int k, int T, const float *__restrict__ nrm,
    const float *__restrict__ A_m, const float *__restrict__ A_s, const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, const float *__restrict__ C_gt, const float *__restrict__ gt,
    const float *__restrict__ recon, const int *__restrict__ arg, float *__restrict__ dC) {
    const float ge = (g_e ? *g_e : 0.f) / (float)N, ga = (g_ade ? *g_ade : 0.f) / (float)N / (float)T,
                gf = (g_fde ? *g_fde : 0.f) / (float)N;
    for (int n = threadIdx.x; n < N; n += kTrThreads) {
        for (int j = 0; j < k; ++j)
            for (int s = 0; s < S; ++s) dC[((int64_t)j * N + n) * S + s] = 0.f;
        const RowNorm p = nrm_row(nrm, N, n, mode, static_dist);
        const float *A = p.mv ? A_m : A_s;
        const float *U = p.mv ? U_m : U_s;
        const float *g = gt + (int64_t)n * 2 * T;
        {   // coefficient term: (cp - cg) / ||cp - cg|| at its arg-min sample
            const int s = arg[n];
            float df[ET_MAX_K];
            float e2 = 0.f;
            for (int j = 0; j < k; ++j) {
                const float cp = (A ? A[j * S + s] : 0.f) + C[((int64_t)j * N + n) * S + s];
                df[j] = cp - C_gt[(int64_t)j * N + n];
                e2 = fmaf(df[j], df[j], e2);
            }
            const float e = sqrtf(e2);
            if (e > 0.f)
                for (int j = 0; j < k; ++j) dC[((int64_t)j * N + n) * S + s] += ge * df[j] / e;
        }
        // displacement terms: d ||r - g|| / d r = (r - g) / ||r - g||, pulled back through the denormalisation and U
        auto pull = [&](int s, int t, float w) {
            const float *r = recon + (((int64_t)s * N + n) * T + t) * 2;
            const float ex = r[0] - g[2 * t], ey = r[1] - g[2 * t + 1];
            const float dist = sqrtf(fmaf(ex, ex, ey * ey));
            if (!(dist > 0.f)) return;
            float a, b;
            denormalize_point_bwd(p, w * ex / dist, w * ey / dist, a, b);
            for (int j = 0; j < k; ++j)
                dC[((int64_t)j * N + n) * S + s] += fmaf(U[(2 * t) * k + j], a, U[(2 * t + 1) * k + j] * b);
        };
        const int sa = arg[N + n], sf = arg[2 * N + n];
        for (int t = 0; t < T; ++t) pull(sa, t, ga);
        pull(sf, T - 1, gf);
    }
}

}  // namespace et

using namespace et;

static bool tr_dims_ok(int64_t N, int S, int k, int T_obs, int T_pred) {
    return N >= 0 && N <= ET_SCENE_MAX_N && S >= 1 && k >= 1 && k <= ET_MAX_K && T_obs >= 3 && T_obs <= ET_MAX_T &&
           T_pred >= 1 && T_pred <= ET_MAX_T;
}

extern "C" int et_scene_project_train(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                                      const float *U_obs_m, const float *U_obs_s, const float *U_pred_m,
                                      const float *U_pred_s, int mode, float static_dist, float *C_obs, float *nrm,
                                      float *obs_ori, float *C_gt, uint8_t *flag, et_stream_t stream) {
    if (!tr_dims_ok(N, 1, k, T_obs, T_pred) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if (!obs || !pred || !C_obs || !nrm || !obs_ori || !C_gt) return ET_ERR_INVALID_ARG;
    const bool m = mode == ET_MODE_MOVING || mode == ET_MODE_SPLIT, s = mode != ET_MODE_MOVING;
    if ((m && (!U_obs_m || !U_pred_m)) || (s && (!U_obs_s || !U_pred_s))) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scene_project_train_kernel, dim3(1), dim3(kTrThreads), 0, (hipStream_t)stream, obs, pred, (int)N, T_obs,
                       T_pred, k, U_obs_m, U_obs_s, U_pred_m, U_pred_s, mode, static_dist, C_obs, nrm, obs_ori, C_gt, flag);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_wrapper_losses_fwd(const float *C, int64_t N, int S, int k, int T_pred, const float *nrm, const float *A_m,
                                     const float *A_s, const float *U_pred_m, const float *U_pred_s, int mode,
                                     float static_dist, const float *C_gt, const float *gt, float *recon, float *best,
                                     int32_t *arg, float *losses, et_stream_t stream) {
    if (!tr_dims_ok(N, S, k, 3, T_pred) || N < 1 || mode < 0 || mode > 2) return ET_ERR_INVALID_ARG;
    if (!C || !nrm || !C_gt || !gt || !recon || !best || !arg || !losses) return ET_ERR_INVALID_ARG;
    const bool m = mode != ET_MODE_STATIC, s = mode != ET_MODE_MOVING;
    if ((m && !U_pred_m) || (s && !U_pred_s)) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(wrapper_losses_fwd_kernel, dim3(1), dim3(kTrThreads), 0, (hipStream_t)stream, C, (int)N, S, k, T_pred, nrm,
                       A_m, A_s, U_pred_m, U_pred_s, mode, static_dist, C_gt, gt, recon, best, (int *)arg, losses);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_wrapper_losses_bwd(const float *g_eigentraj, const float *g_ade, const float *g_fde, const float *C,
                                     int64_t N, int S, int k, int T_pred,
                                     const float *nrm, const float *A_m, const float *A_s, const float *U_pred_m,
                                     const float *U_pred_s, int mode, float static_dist, const float *C_gt,
                                     const float *gt, const float *recon, const int32_t *arg, float *dC,
                                     et_stream_t stream) {
    if (!tr_dims_ok(N, S, k, 3, T_pred) || N < 1 || mode < 0 || mode > 2) return ET_ERR_INVALID_ARG;
    if (!C || !nrm || !C_gt || !gt || !recon || !arg || !dC) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(wrapper_losses_bwd_kernel, dim3(1), dim3(kTrThreads), 0, (hipStream_t)stream, g_eigentraj, g_ade, g_fde, C,
                       (int)N, S, k,
                       T_pred, nrm, A_m, A_s, U_pred_m, U_pred_s, mode, static_dist, C_gt, gt, recon, (const int *)arg, dC);
    ET_LAUNCH_CHECK();
    return ET_OK;
}
