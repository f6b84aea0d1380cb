// et_trajnorm.hip -- stand-alone TrajNorm operations (EigenTrajectory/normalizer.py) and
// euc_sim (EigenTrajectory/kmeans.py:59-76) as public entry points.  The fused kernels in
// et_descriptor.hip never materialise these tensors; these exist because the reference exposes
// them (TrajNorm.calculate_params / normalize / denormalize are called directly by
// script/descriptor_evaluation.py:32-36 and the plot scripts).
#include "et_common.h"

namespace et {

constexpr int kTnThreads = 256;

// normalizer.py:17-29.  One lane per trajectory; reads 16 B of each row.
__global__ __launch_bounds__(kTnThreads) void norm_params_kernel(const float *__restrict__ obs, int64_t N, int T,
                                                                 float *__restrict__ ori, float *__restrict__ rot,
                                                                 float *__restrict__ sca) {
    const int64_t n = (int64_t)blockIdx.x * kTnThreads + threadIdx.x;
    if (n >= N) return;
    const float *row = obs + n * 2 * T;
    const float ox = row[2 * (T - 1)], oy = row[2 * (T - 1) + 1];
    const float dx = ox - row[2 * (T - 3)], dy = oy - row[2 * (T - 3) + 1];
    const RowNorm p = row_norm(ox, oy, dx, dy, ET_MODE_MOVING, 0.f);
    if (ori) {
        ori[2 * n] = ox;
        ori[2 * n + 1] = oy;
    }
    if (rot) {  // normalizer.py:25-26: [[cos, -sin], [sin, cos]]
        rot[4 * n] = p.c;
        rot[4 * n + 1] = -p.s;
        rot[4 * n + 2] = p.s;
        rot[4 * n + 3] = p.c;
    }
    if (sca) sca[n] = p.sca;
}

// Same outputs from the compact state the projection kernel caches: nrm (4,N) = ox, oy, dx, dy.
__global__ __launch_bounds__(kTnThreads) void norm_params_nrm_kernel(const float *__restrict__ nrm, int64_t N,
                                                                     float *__restrict__ ori, float *__restrict__ rot,
                                                                     float *__restrict__ sca) {
    const int64_t n = (int64_t)blockIdx.x * kTnThreads + threadIdx.x;
    if (n >= N) return;
    const float ox = nrm[n], oy = nrm[N + n];
    const RowNorm p = row_norm(ox, oy, nrm[2 * N + n], nrm[3 * N + n], ET_MODE_MOVING, 0.f);
    if (ori) {
        ori[2 * n] = ox;
        ori[2 * n + 1] = oy;
    }
    if (rot) {
        rot[4 * n] = p.c;
        rot[4 * n + 1] = -p.s;
        rot[4 * n + 2] = p.s;
        rot[4 * n + 3] = p.c;
    }
    if (sca) sca[n] = p.sca;
}

// One lane per point (n, t): unit-stride float2 traffic on traj / out.
template <bool INVERSE>
__global__ __launch_bounds__(kTnThreads) void traj_transform_kernel(const float *__restrict__ traj, int64_t N, int T,
                                                                    const float *__restrict__ ori,
                                                                    const float *__restrict__ rot,
                                                                    const float *__restrict__ sca,
                                                                    float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kTnThreads + threadIdx.x;
    if (i >= N * T) return;
    const int64_t n = i / T;
    const float2 v = reinterpret_cast<const float2 *>(traj)[i];
    float x = v.x, y = v.y;
    if (!INVERSE) {
        if (ori) {  // normalizer.py:45-46
            x = x - ori[2 * n];
            y = y - ori[2 * n + 1];
        }
        if (rot) {  // :47-48  traj @ R
            const float r00 = rot[4 * n], r01 = rot[4 * n + 1], r10 = rot[4 * n + 2], r11 = rot[4 * n + 3];
            const float a = x * r00 + y * r10, b = x * r01 + y * r11;
            x = a;
            y = b;
        }
        if (sca) {  // :49-50
            x = x * sca[n];
            y = y * sca[n];
        }
    } else {
        if (sca) {  // normalizer.py:56-57
            x = x / sca[n];
            y = y / sca[n];
        }
        if (rot) {  // :58-59  traj @ R^T
            const float r00 = rot[4 * n], r01 = rot[4 * n + 1], r10 = rot[4 * n + 2], r11 = rot[4 * n + 3];
            const float a = x * r00 + y * r01, b = x * r10 + y * r11;
            x = a;
            y = b;
        }
        if (ori) {  // :60-61
            x = x + ori[2 * n];
            y = y + ori[2 * n + 1];
        }
    }
    reinterpret_cast<float2 *>(out)[i] = make_float2(x, y);
}

// kmeans.py:59-76; lane = (i, j) entry, j fastest (unit-stride stores); blockIdx.y = batch element (the reference's
// leading dimensions: contiguous (B, d, m), (B, d, n) -> (B, m, n))
__global__ __launch_bounds__(kTnThreads) void euc_sim_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             int d, int64_t m, int64_t n, float *__restrict__ y) {
    a += (int64_t)blockIdx.y * d * m;
    b += (int64_t)blockIdx.y * d * n;
    y += (int64_t)blockIdx.y * m * n;
    const int64_t e = (int64_t)blockIdx.x * kTnThreads + threadIdx.x;
    if (e >= m * n) return;
    const int64_t i = e / n, j = e - i * n;
    float an = 0.f, bn = 0.f, acc = 0.f;
    for (int t = 0; t < d; ++t) {
        const float av = a[(int64_t)t * m + i], bv = b[(int64_t)t * n + j];
        an = an + av * av;
        bn = bn + bv * bv;
        acc = fmaf(av, bv, acc);
    }
    acc = acc * 2.0f;
    acc = acc - an;
    acc = acc - bn;
    y[e] = acc;
}

}  // namespace et

using namespace et;

extern "C" int et_norm_params(const float *obs, int64_t N, int T, float *ori, float *rot, float *sca,
                              et_stream_t stream) {
    if (N < 0 || T < 3 || T > ET_MAX_T || (N > 0 && !obs)) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    hipLaunchKernelGGL(norm_params_kernel, dim3((unsigned)ceil_div(N, kTnThreads)), dim3(kTnThreads), 0,
                       (hipStream_t)stream, obs, N, T, ori, rot, sca);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_norm_params_from_nrm(const float *nrm, int64_t N, float *ori, float *rot, float *sca,
                                       et_stream_t stream) {
    if (N < 0 || (N > 0 && !nrm)) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    hipLaunchKernelGGL(norm_params_nrm_kernel, dim3((unsigned)ceil_div(N, kTnThreads)), dim3(kTnThreads), 0,
                       (hipStream_t)stream, nrm, N, ori, rot, sca);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

static int traj_transform(bool inverse, const float *traj, int64_t N, int T, const float *ori, const float *rot,
                          const float *sca, float *out, et_stream_t stream) {
    if (N < 0 || T < 1 || (N > 0 && (!traj || !out))) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if ((reinterpret_cast<uintptr_t>(traj) & 7u) || (reinterpret_cast<uintptr_t>(out) & 7u)) return ET_ERR_INVALID_ARG;
    const unsigned grid = (unsigned)ceil_div(N * T, kTnThreads);
    if (inverse)
        hipLaunchKernelGGL((traj_transform_kernel<true>), dim3(grid), dim3(kTnThreads), 0, (hipStream_t)stream, traj, N,
                           T, ori, rot, sca, out);
    else
        hipLaunchKernelGGL((traj_transform_kernel<false>), dim3(grid), dim3(kTnThreads), 0, (hipStream_t)stream, traj, N,
                           T, ori, rot, sca, out);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_normalize(const float *traj, int64_t N, int T, const float *ori, const float *rot, const float *sca,
                            float *out, et_stream_t stream) {
    return traj_transform(false, traj, N, T, ori, rot, sca, out, stream);
}

extern "C" int et_denormalize(const float *traj, int64_t N, int T, const float *ori, const float *rot,
                              const float *sca, float *out, et_stream_t stream) {
    return traj_transform(true, traj, N, T, ori, rot, sca, out, stream);
}

extern "C" int et_euc_sim_batch(const float *a, const float *b, int64_t batch, int d, int64_t m, int64_t n, float *y,
                                et_stream_t stream) {
    if (d < 1 || m < 0 || n < 0 || batch < 0 || batch > 65535 || (batch * m * n > 0 && (!a || !b || !y)))
        return ET_ERR_INVALID_ARG;
    if (batch * m * n == 0) return ET_OK;
    hipLaunchKernelGGL(euc_sim_kernel, dim3((unsigned)ceil_div(m * n, kTnThreads), (unsigned)batch), dim3(kTnThreads), 0,
                       (hipStream_t)stream, a, b, d, m, n, y);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_euc_sim(const float *a, const float *b, int d, int64_t m, int64_t n, float *y, et_stream_t stream) {
    return et_euc_sim_batch(a, b, 1, d, m, n, y, stream);
}
