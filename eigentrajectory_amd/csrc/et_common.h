// et_common.h -- shared device helpers for the gfx950 kernels of libetamd.so.
//
// Built with -ffp-contract=off: every fused multiply-add below is an explicit
// fmaf()/fma(), so the arithmetic that has to be bit-identical to the CPU oracle
// (k-means similarities, fixed-point sums, Jacobi rotations) is spelled out, not
// left to the contraction pass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eigentraj.h"

#define ET_HIP_TRY(expr)                         \
    do {                                         \
        if ((expr) != hipSuccess) return ET_ERR_HIP; \
    } while (0)

#define ET_LAUNCH_CHECK()                                    \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return ET_ERR_HIP; \
    } while (0)

namespace et {

constexpr int kWave = 64;  // CDNA wavefront width

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Normaliser state of one trajectory (EigenTrajectory/normalizer.py:17-29).
struct RowNorm {
    float ox, oy;  // origin = last observed point
    float c, s;    // cos / sin of the heading angle
    float sca;     // 2 / ||d||  (1 for the static descriptor)
    float inv;     // 1 / sca
    int mv;        // 1 = moving descriptor (norm_sca=True)
};

// (ox,oy) = obs[-1]; (dx,dy) = obs[-1]-obs[-3].  mode: ET_MODE_*.
__device__ __forceinline__ RowNorm row_norm(float ox, float oy, float dx, float dy, int mode, float static_dist) {
    RowNorm p;
    if (mode == ET_MODE_IDENTITY) {
        p.ox = 0.f;
        p.oy = 0.f;
        p.c = 1.f;
        p.s = 0.f;
        p.sca = 1.f;
        p.inv = 1.f;
        p.mv = 0;
        return p;
    }
    p.ox = ox;
    p.oy = oy;
    int mv = mode;
    if (mode == ET_MODE_SPLIT) {  // model.py:46 / :73
        const float hx = dx * 0.5f, hy = dy * 0.5f;
        mv = sqrtf(hx * hx + hy * hy) > static_dist ? 1 : 0;
    }
    p.mv = mv;
    // normalizer.py:24-26 builds R from theta = atan2(dy, dx), cos(theta), sin(theta).  The same unit
    // vector is (dx, dy) / ||d||, which costs one sqrt and two divisions instead of three
    // transcendental calls per trajectory and is at least as accurate (the reference's fp32 theta
    // already carries 6e-8 relative error); atan2(0, 0) = 0 gives the identity for motionless rows.
    const float r = sqrtf(dx * dx + dy * dy);
    const bool still = !(r > 0.0f);
    p.c = still ? (isnan(r) ? r : 1.0f) : dx / r;
    p.s = still ? (isnan(r) ? r : 0.0f) : dy / r;
    p.sca = mv ? (1.0f / r) * 2.0f : 1.0f;  // normalizer.py:28
    p.inv = mv ? 1.0f / p.sca : 1.0f;
    return p;
}

// normalizer.py:42-51: ((p - ori) @ R) * sca with R = [[c,-s],[s,c]]
__device__ __forceinline__ void normalize_point(const RowNorm &p, float x, float y, float &xn, float &yn) {
    const float tx = x - p.ox, ty = y - p.oy;
    float a = tx * p.c + ty * p.s;
    float b = tx * (-p.s) + ty * p.c;
    if (p.mv) {
        a = a * p.sca;
        b = b * p.sca;
    }
    xn = a;
    yn = b;
}

// normalizer.py:53-62: (v / sca) @ R^T + ori   (the division is done as * (1/sca))
__device__ __forceinline__ void denormalize_point(const RowNorm &p, float x, float y, float &xo, float &yo) {
    if (p.mv) {
        x = x * p.inv;
        y = y * p.inv;
    }
    xo = (x * p.c + y * (-p.s)) + p.ox;
    yo = (x * p.s + y * p.c) + p.oy;
}

// backward of denormalize_point w.r.t. (x,y): (g @ R) / sca
__device__ __forceinline__ void denormalize_point_bwd(const RowNorm &p, float gx, float gy, float &dx, float &dy) {
    float a = gx * p.c + gy * p.s;
    float b = gx * (-p.s) + gy * p.c;
    if (p.mv) {
        a = a * p.inv;
        b = b * p.inv;
    }
    dx = a;
    dy = b;
}

}  // namespace et
