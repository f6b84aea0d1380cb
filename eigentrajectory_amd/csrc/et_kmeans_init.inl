// et_kmeans_init.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): farthest-first initialisation kernels (kmeans.py:88-112).
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
// NaN first) encoded as a 64-bit key so that a plain unsigned min is the reduction.
// ------------------------------------------------------------------------------------------
//
// Steps >= 2 skip the coordinate read of every point that provably keeps its running maximum (Elkan's
// triangle inequality, made rigorous for the computed fp32 similarities): with l = nearest[n] the centroid that
// holds best[n], Delta_l <= ||c_new - c_l|| and E >= the rounding error of any computed similarity of this shard
// (2^-19 (R + C)^2 with R = sqrt(d) max|x| >= every local ||x|| and C = the largest centroid norm so far),
//     ||x - c_l|| <= sqrt(E - best[n])   and   ||x - c_new|| >= Delta_l - ||x - c_l||,
// so  Delta_l >= 2 sqrt(E - best[n])  implies  y_new <= -||x - c_new||^2 + E <= best[n]:  the strict `>` of the
// update cannot fire and best / nearest stay as they are.  Such a point costs 5 B (best + nearest) instead of
// 32 B; farthest-first picks are far from everything by construction, so most points qualify.  best[] is only
// written when it changes.  max|x| is collected by step 1, which reads everything anyway.
// PERSIST (not instantiated any more: tools/archive/lost_forms/kmeans_init_persist.hip.txt): the body inside ONE launch for all steps,
// separated by a fence-free grid barrier: everything that
// crosses workgroups inside the launch -- the workgroup keys, the centroid columns workgroup 0 stores -- is then written
// and read with device-scope atomics (served by the memory side: no cache fence); best / nearest / the tile summaries
// are only re-read by the wavefront that wrote them (the tile -> wavefront map is fixed).
template <int D, bool PERSIST>
__device__ __forceinline__ void init_step_body(const float *__restrict__ X, int64_t N, int d_rt, int K, int step,
                                               const float *C0, float *__restrict__ best, uint8_t *__restrict__ nearest,
                                               unsigned *__restrict__ max_abs_bits, int64_t index_base,
                                               unsigned long long *block_keys, const unsigned long long *prev_keys,
                                               int n_prev, float *C0_rw, unsigned char *cand, uint4 *__restrict__ meta,
                                               int meta_valid) {
    const int d = D ? D : d_rt;
    __shared__ float sc[ET_KMEANS_MAX_D + 1];
    __shared__ float sDelta[ET_KMEANS_MAX_CLUSTERS + 1];
    __shared__ unsigned long long sKey[kKmThreads / 64];
    __shared__ unsigned sMax[kKmThreads / 64];
    __shared__ unsigned sCmax;  // fp32 bits of the largest centroid norm among columns 0 .. step-1
    // the earlier centroids this thread will measure the new one against (columns < step - 1 are final): requested now,
    // so that their round trip overlaps the key reduction and the gather of the new centroid
    float cprev[D ? D : 1];
    if constexpr (D != 0) {
        const int jc = (int)threadIdx.x < step - 1 ? (int)threadIdx.x : 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
            cprev[i] = PERSIST ? __hip_atomic_load(&C0[i * K + jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : C0[i * K + jc];
    }
    // the tile summaries of this wavefront's first 64 tiles do not depend on the new centroid either: requested now, their
    // round trip (the fourth dependent one of a step) runs under the prologue's
    const int lane = (int)(threadIdx.x & 63);
    const bool vec = step > 1 && ((reinterpret_cast<uintptr_t>(best) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(nearest) & 3u) == 0);
    const int64_t n4 = vec ? N / 4 : 0;
    const int64_t n_tiles = (n4 + 63) >> 6, n_waves = (int64_t)gridDim.x * (kKmThreads / 64);
    const int64_t per = (n_tiles + n_waves - 1) / n_waves;
    const int64_t t_begin = ((int64_t)blockIdx.x * (kKmThreads / 64) + (threadIdx.x >> 6)) * per;
    const int64_t t_end = t_begin + per < n_tiles ? t_begin + per : n_tiles;
    uint4 meta0 = make_uint4(0u, 0u, 0u, 0u);
    if (meta && vec && meta_valid && t_begin + lane < t_end) meta0 = meta[t_begin + lane];
    if (prev_keys) {
        // Single-GPU path: centroid step-1 has not been stored yet -- every workgroup derives it from the previous
        // step's workgroup keys (the same minimum everywhere), workgroup 0 also stores it.  Two short round trips
        // in the prologue instead of a pick launch between two steps.
        unsigned long long key = ~0ull;
        for (int b0 = 0; b0 < n_prev; b0 += 4 * kKmThreads) {  // four keys per thread in flight (usually all there are)
            unsigned long long k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * kKmThreads + (int)threadIdx.x;
                k4[u] = PERSIST ? __hip_atomic_load(&prev_keys[b < n_prev ? b : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : prev_keys[b < n_prev ? b : 0];
                if (b >= n_prev) k4[u] = ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) key = k4[u] < key ? k4[u] : key;
        }
        key = wave_min_u64_lane0(key);  // (register exchanges: six ds_bpermute levels on a 64-bit key were ~800 cycles)
        if ((threadIdx.x & 63) == 0) sKey[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kKmThreads / 64; ++w) key = sKey[w] < key ? sKey[w] : key;
            const int64_t local = (int64_t)(unsigned)(key & 0xffffffffull) - index_base;
            sCmax = 0u;
            float bn = 0.f;
            const bool ok = key != ~0ull && local >= 0 && local < N;
            float pt[D ? D : 1];
            if constexpr (D != 0) {  // the winner's coordinates: all loads first (the stores below may alias for the compiler)
#pragma unroll
                for (int i = 0; i < D; ++i) pt[i] = X[(int64_t)i * N + (ok ? local : 0)];
            }
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i) {
                if (i >= d) break;
                float v;
                if constexpr (D != 0) v = ok ? pt[i] : __int_as_float(0x7fc00000);
                else v = ok ? X[(int64_t)i * N + local] : __int_as_float(0x7fc00000);
                sc[i] = v;
                bn = bn + v * v;
                if (blockIdx.x == 0) {
                    if (PERSIST) __hip_atomic_store(&C0_rw[i * K + (step - 1)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else C0_rw[i * K + (step - 1)] = v;
                    reinterpret_cast<float *>(cand + 8)[i] = v;
                }
            }
            sc[d] = bn;
            if (blockIdx.x == 0) *reinterpret_cast<unsigned long long *>(cand) = key;
        }
    } else if (threadIdx.x == 0) {
        sCmax = 0u;
        float bn = 0.f;
        for (int i = 0; i < d; ++i) {
            const float v = C0[i * K + (step - 1)];
            sc[i] = v;
            bn = bn + v * v;
        }
        sc[d] = bn;
    }
    __syncthreads();
    // lower bounds of the distances from the new centroid to the earlier ones, upper bound of the centroid norms
    for (int j = threadIdx.x; j < step; j += kKmThreads) {
        double s2 = 0.0, n2 = 0.0;
#pragma unroll
        for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i) {
            if (i >= d) break;
            double cj;
            if constexpr (D != 0) cj = j == step - 1 ? (double)sc[i] : (double)cprev[i];  // step <= K < blockDim.x: j == threadIdx.x
            else cj = j == step - 1 ? (double)sc[i]
                                    : (double)(PERSIST ? __hip_atomic_load(&C0[i * K + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : C0[i * K + j]);
            const double t = (double)sc[i] - cj;
            s2 += t * t;
            n2 += cj * cj;
        }
        // the SQUARE of a lower bound of the distance: the skip test below is delta >= 2 sqrt(E - b), evaluated as
        // delta^2 >= 4 (E - b) with E - b >= 0 (no square root per point; both sides carry their margins)
        sDelta[j] = (float)(s2 * (1.0 - 4e-6)) * (1.0f - 1e-6f);
        const float nj = (float)(sqrt(n2) * (1.0 + 1e-6)) * (1.0f + 1e-6f);
        atomicMax(&sCmax, nj == nj ? __float_as_uint(nj) : 0x7f800000u);  // NaN centroid: +inf, nothing is skipped
    }
    __syncthreads();
    float c[D ? D : ET_KMEANS_MAX_D];
#pragma unroll
    for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
        if (i < d) c[i] = sc[i];
    const float bn = sc[d];
    // E: bound on |computed similarity - (-||x - c||^2)| for this shard's points; +inf (never skip) if unknown
    float E = __int_as_float(0x7f800000);
    if (step > 1) {
        const float R = sqrtf((float)d) * __uint_as_float(*max_abs_bits) * 1.0001f + __uint_as_float(sCmax);
        E = R * R * 1.9073486328125e-6f;  // 2^-19 (R + C)^2
        if (!(E <= 3.0e38f)) E = __int_as_float(0x7f800000);
    }
    unsigned long long key = ~0ull;
    float mabs = 0.f;
    // one point: full evaluation unless `skip`; returns the (possibly updated) running maximum
    auto visit = [&](int64_t n, float b, bool skip, float &b_out, int &lab_out) {
        if (!skip) {
            float an = 0.f, y = 0.f;
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
                if (i < d) {
                    const float v = X[(int64_t)i * N + n];
                    an = an + v * v;
                    y = fmaf(v, c[i], y);
                    if (step == 1) mabs = fmaxf(mabs, fabsf(v));  // NaN ignored; a NaN point never gets skipped anyway
                }
            y = y * 2.0f;
            y = y - an;
            y = y - bn;
            if (step == 1 || gt_nanmax(y, b)) {
                b = y;
                best[n] = b;
                nearest[n] = (uint8_t)(step - 1);
                lab_out = step - 1;
            }
        }
        b_out = b;
        const unsigned long long k = ((unsigned long long)orderable(b) << 32) | (unsigned)(index_base + n);
        key = k < key ? k : key;
    };
    const int64_t stride = (int64_t)gridDim.x * kKmThreads;
    const int64_t tid = (int64_t)blockIdx.x * kKmThreads + threadIdx.x;
    // steps >= 2 look at four points per lane through one 16-B load of best[] and one 4-B load of nearest[]
    // Steps >= 3 first look at a 16-byte summary of each tile of 256 points (the smallest key, the largest running
    // similarity, the set of nearest centroids -- written by the step before): if the skip test holds for the tile's
    // WORST values it holds for every point in it (E - b and the product are monotone in b, the distance bound is the
    // smallest over the labels present), nothing in the tile changes, and its smallest key is the stored one -- the tile
    // costs 16 bytes instead of 1280.  A wavefront owns a contiguous run of tiles; its LANES test up to 64 of them at once
    // (one summary each: one round trip for the whole run, not one per tile), then the whole wavefront goes through the
    // tiles that failed, point by point as before.  After a farthest-first pick almost every tile passes: the sweep of a
    // step was 9 of its 18 us, all of it reading best[] and nearest[].
    auto sweep_tile = [&](int64_t tile) {  // the whole wavefront: four points per lane, and the tile's new summary
        const int64_t g = tile * 64 + lane;
        const bool act = g < n4;
        unsigned long long tkey = ~0ull;
        float tmax = -__int_as_float(0x7f800000);
        unsigned tmask = 0u;
        if (act) {
            const float4 b4 = reinterpret_cast<const float4 *>(best)[g];
            const unsigned l4 = reinterpret_cast<const unsigned *>(nearest)[g];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            const unsigned long long before = key;
            key = ~0ull;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // (a NaN or +inf in b, E or Delta makes the comparison false: full evaluation)
                const float w = E - bb[v];
                const int lab = (int)((l4 >> (8 * v)) & 0xffu);
                const bool skip = w >= 0.0f && sDelta[lab] >= 4.0001f * w;
                float b_after = bb[v];
                int lab_after = lab;
                visit(4 * g + v, bb[v], skip, b_after, lab_after);
                tmax = b_after != b_after ? __int_as_float(0x7f800000) : fmaxf(tmax, b_after);
                tmask |= 1u << (lab_after & 31);
            }
            tkey = key;
            key = tkey < before ? tkey : before;
        }
        if (meta) {
            tkey = wave_min_u64_lane0(tkey);
#define ET_DOWN(O)                                                                          \
    do {                                                                                    \
        tmax = fmaxf(tmax, __uint_as_float(lane_down_u32<O>(__float_as_uint(tmax))));       \
        tmask |= lane_down_u32<O>(tmask);                                                   \
    } while (0)
            ET_DOWN(32);
            ET_DOWN(16);
            ET_DOWN(8);
            ET_DOWN(4);
            ET_DOWN(2);
            ET_DOWN(1);
#undef ET_DOWN
            if (lane == 0)
                meta[tile] = make_uint4((unsigned)(tkey & 0xffffffffull), (unsigned)(tkey >> 32), __float_as_uint(tmax), tmask);
        }
    };
    if (meta && vec) {
        for (int64_t tb = t_begin; tb < t_end; tb += 64) {  // (wave-uniform)
            const int64_t mine = tb + lane;
            bool todo_mine = mine < t_end;
            if (meta_valid && todo_mine) {
                const uint4 m = tb == t_begin ? meta0 : meta[mine];
                const unsigned ob = m.y;  // orderable(b_min) -> b_min
                const float b_min = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob), b_max = __uint_as_float(m.z);
                float dmin = __int_as_float(0x7f800000);
                for (unsigned bits = m.w; bits; bits &= bits - 1u) dmin = fminf(dmin, sDelta[__builtin_ctz(bits)]);
                // (a NaN anywhere makes a comparison false: the tile is looked at point by point)
                if (m.w != 0u && E - b_max >= 0.0f && dmin >= 4.0001f * (E - b_min)) {
                    const unsigned long long k = ((unsigned long long)m.y << 32) | m.x;
                    key = k < key ? k : key;
                    todo_mine = false;
                }
            }
            for (unsigned long long todo = __ballot(todo_mine); todo; todo &= todo - 1ull)
                sweep_tile(tb + __builtin_ctzll(todo));
        }
    } else {
        for (int64_t g = tid; g < n4; g += stride) {
            const float4 b4 = reinterpret_cast<const float4 *>(best)[g];
            const unsigned l4 = reinterpret_cast<const unsigned *>(nearest)[g];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // (a NaN or +inf in b, E or Delta makes the comparison false: full evaluation)
                const float w = E - bb[v];
                const bool skip = w >= 0.0f && sDelta[(l4 >> (8 * v)) & 0xffu] >= 4.0001f * w;
                float b_after;
                int lab_after = 0;
                visit(4 * g + v, bb[v], skip, b_after, lab_after);
            }
        }
    }
    for (int64_t n = 4 * n4 + tid; n < N; n += stride) {
        float b = 0.f;
        bool skip = false;
        if (step > 1) {
            b = best[n];
            const float w = E - b;
            skip = w >= 0.0f && sDelta[(int)nearest[n]] >= 4.0001f * w;
        }
        float b_after;
        int lab_after;
        visit(n, b, skip, b_after, lab_after);
    }
    key = wave_min_u64_lane0(key);
    if (step == 1) {  // (the largest |x|: made in the first step only)
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<32>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<16>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<8>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<4>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<2>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<1>(__float_as_uint(mabs))));
    }
    if ((threadIdx.x & 63) == 0) {
        sKey[threadIdx.x >> 6] = key;
        sMax[threadIdx.x >> 6] = __float_as_uint(mabs);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mb = sMax[0];
        for (int w = 1; w < kKmThreads / 64; ++w) {
            key = sKey[w] < key ? sKey[w] : key;
            mb = sMax[w] > mb ? sMax[w] : mb;
        }
        if (PERSIST) __hip_atomic_store(&block_keys[blockIdx.x], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else block_keys[blockIdx.x] = key;
        // non-negative floats order like their bits; only a workgroup that would raise the maximum touches it
        if (step == 1 && mb > __hip_atomic_load(max_abs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_abs_bits, mb);
    }
}

template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeans_init_step_kernel(const float *__restrict__ X, int64_t N, int d_rt,
                                                                      int K, int step, const float *__restrict__ C0,
                                                                      float *__restrict__ best,
                                                                      uint8_t *__restrict__ nearest,
                                                                      unsigned *__restrict__ max_abs_bits,
                                                                      int64_t index_base,
                                                                      unsigned long long *__restrict__ block_keys,
                                                                      const unsigned long long *__restrict__ prev_keys,
                                                                      int n_prev, float *C0_rw, unsigned char *cand,
                                                                      uint4 *__restrict__ meta, int meta_valid) {
    init_step_body<D, false>(X, N, d_rt, K, step, C0, best, nearest, max_abs_bits, index_base, block_keys, prev_keys, n_prev,
                             C0_rw, cand, meta, meta_valid);
}

// reduce the workgroup keys; candidate record = {key, d floats of the winning local point}
__global__ __launch_bounds__(kKmThreads) void kmeans_init_pick_kernel(const float *__restrict__ X, int64_t N, int d,
                                                                      const unsigned long long *__restrict__ block_keys,
                                                                      int n_blocks, int64_t index_base,
                                                                      unsigned char *__restrict__ cand, float *C0_out, int K,
                                                                      int col) {
    __shared__ unsigned long long sKey[kKmThreads / 64];
    unsigned long long key = ~0ull;
    for (int b = threadIdx.x; b < n_blocks; b += kKmThreads) key = block_keys[b] < key ? block_keys[b] : key;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) sKey[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKmThreads / 64; ++w) key = sKey[w] < key ? sKey[w] : key;
        *reinterpret_cast<unsigned long long *>(cand) = key;
        float *pt = reinterpret_cast<float *>(cand + 8);
        const int64_t local = (int64_t)(unsigned)(key & 0xffffffffull) - index_base;
        for (int i = 0; i < d; ++i) {
            pt[i] = (key != ~0ull && local >= 0 && local < N) ? X[(int64_t)i * N + local] : __int_as_float(0x7fc00000);
            if (C0_out) C0_out[i * K + col] = pt[i];  // single-GPU path: the candidate IS the new centroid
        }
    }
}

// sharded farthest-first step: the smallest 64-bit key among the ranks' candidate records (value first, then global
// index: the same winner on every rank) becomes centroid `col`.  One wavefront; replaces a handful of tensor ops.
__global__ void kmeans_init_select_kernel(const unsigned char *__restrict__ cands, int n_cands, int stride, int d, int K,
                                          int col, float *__restrict__ C0) {
    const int lane = threadIdx.x;
    unsigned long long key = ~0ull;
    int who = 0;
    for (int r = lane; r < n_cands; r += 64) {
        const unsigned long long k = *reinterpret_cast<const unsigned long long *>(cands + (size_t)r * stride);
        if (k < key) {
            key = k;
            who = r;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o);
        const int ow = __shfl_xor(who, o);
        if (ok < key || (ok == key && ow < who)) {
            key = ok;
            who = ow;
        }
    }
    if (lane < d) C0[lane * K + col] = *reinterpret_cast<const float *>(cands + (size_t)who * stride + 8 + 4 * lane);
}

__global__ void kmeans_init_set_kernel(float *__restrict__ C0, int d, int K, int col, const float *__restrict__ point) {
    const int i = threadIdx.x;
    if (i < d) C0[i * K + col] = point[i];
}

__global__ void kmeans_gather_point_kernel(const float *__restrict__ X, int64_t N, int d, int64_t idx,
                                           float *__restrict__ point) {
    const int i = threadIdx.x;
    if (i < d) point[i] = X[(int64_t)i * N + idx];
}

// farthest-first, before its first step, in ONE launch: the first centroid = point `idx` (-> column 0 of C0 and the
// candidate record's point slot) and the running max |x| cleared (a gather kernel, a set kernel and a memset were three
// ~5 us packets with a kernel boundary each)
__global__ void kmeans_init_first_kernel(const float *__restrict__ X, int64_t N, int d, int K, int64_t idx,
                                         float *__restrict__ C0, float *__restrict__ point, unsigned *__restrict__ maxabs) {
    const int i = threadIdx.x;
    if (i < d) {
        const float v = X[(int64_t)i * N + idx];
        point[i] = v;
        C0[i * K] = v;
    }
    if (i == 0) *maxabs = 0u;
}

// the state block before a scan: all zero, "no non-zero value yet" = +inf (two memsets were two packets)
__global__ void kmeans_state_reset_kernel(et_kmeans_state *state) {
    constexpr int kWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) reinterpret_cast<unsigned *>(state)[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<unsigned *>(&state->min_nz_x_bits) = 0x7f800000u;
}

static int km_grid(int64_t work_items) {
    const int64_t b = ceil_div(work_items, (int64_t)kKmThreads);
    return (int)(b < 1 ? 1 : (b > kKmMaxBlocks ? kKmMaxBlocks : b));
