// et_reforder_fast_assign.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): the fast form: assignment of a group by matrix-core certification, and the first half of an iteration (reforder_groups_kernel: assignment + cascade levels 0 and 1).
// ---- the assignment of a group by CERTIFICATION (iterations >= 1, no NaN possible): csrc/et_kmeans.hip's matrix-core filter
//      ("Lloyd half-step for iterations >= 1": that is where the bounds are derived) on the quads of the permuted copy.  Per
//      point the second largest of the f16-MFMA upper bounds u_j >= Y_j + |x|^2 is compared with the exact Y_l + |x|^2 of the
//      point's OLD label l (one fmaf chain, kmeans.py:71-74 with the norms in ATen's orders -- the bound E1 on the chain's
//      rounding holds for any order of the six-term norm sums): if it exceeds every other cluster's bound the reference's
//      arg-max is l, strictly, and Y_l is its maximum similarity.  Every other point (1-3 % per iteration) goes on a
//      workgroup queue and gets the exact scan afterwards, four threads per point.  The same labels as the exact scan of
//      every point, by construction; what it saves is the scan: ~300 vector + 16 matrix instructions per 256 points
//      instead of ~720 vector instructions. ----
#ifdef ET_EXP_RF_CHECK
__device__ unsigned g_rf_check[64];
#endif
template <int NREGS>
__device__ __forceinline__ double assign_group_filter(const float4 *__restrict__ x4, int L2, const float *sC, int K, float sg,
                                                      unsigned *sLab, unsigned *__restrict__ LTg, unsigned short *sQ, int q_cap,
                                                      int *sQn, bool &ok) {
    const int tid = (int)threadIdx.x, lane = tid & 63, half = lane >> 5, col = lane & 31;
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;  // (1 + 2^-9) v + 2^-23 survives the rtz to f16
    const float sg2 = sg * sg;
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);
    // A operands (loop invariant): this lane feeds accumulator row m = col, k-half = half; cluster j sits in register j >> 1
    // of half j & 1 (rows of clusters >= K: -60000) -- csrc/et_kmeans.hip, filter_assign_body
    u32x4 a1 = {0u, 0u, 0u, 0u}, a2 = {0u, 0u, 0u, 0u};
    {
        const int j = 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
        unsigned ch[3] = {0u, 0u, 0u}, cl[3] = {0u, 0u, 0u};
        float nb = -60000.0f;
        if (j < K) {
#pragma unroll
            for (int p = 0; p < 3; ++p) split_f16(sC[j * 8 + 2 * p], sC[j * 8 + 2 * p + 1], 2.0f * sg, ch[p], cl[p]);
            nb = -sC[j * 8 + 6] * sg2;
        }
        const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
        const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
        unsigned ebd = 0u;
        if (j < K) {
            const float cj = sqrtf(sC[j * 8 + 6]) * sg * 1.001f;
            ebd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(3.0517578125e-5f * cj, kUp, kTiny),
                                                                           fmaf(fmaf(cj, 1.52587890625e-5f, 9.5367431640625e-7f) * cj, kUp, kTiny)));
        }
        a1 = u32x4{ch[0], ch[1], ch[2], half == 0 ? bnd : ebd};
        a2 = u32x4{cl[0], cl[1], cl[2], 0u};
    }
    const f16x8 A1 = __builtin_bit_cast(f16x8, a1), A2 = __builtin_bit_cast(f16x8, a2);
    double sim = 0.0;
    for (int qi = tid; qi < L2; qi += kFThreads) {  // (L2 mod 384 = 256: whole wavefronts run the last round)
        const unsigned old_packed = LTg[qi];
        unsigned undecided = 0u;
        // Register-lean on purpose (the first form held the quad's 24 coordinates and both tiles' 32 accumulators: 122
        // registers = two workgroups per CU, and lost to the exact scan): two points at a time from 8-byte loads (the lines
        // are in the L1 after the first), the two 32-point tiles of a step one after the other.
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            float2 v[kD];
#pragma unroll
            for (int i = 0; i < kD; ++i) v[i] = reinterpret_cast<const float2 *>(x4 + i * L2 + qi)[hq];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * hq + qq;
                float x[kD];
#pragma unroll
                for (int i = 0; i < kD; ++i) x[i] = qq == 0 ? v[i].x : v[i].y;
                float an = x[0] * x[0];  // kmeans.py:73, a full block's column: rows in sequence
#pragma unroll
                for (int i = 1; i < kD; ++i) an = an + x[i] * x[i];
                const float rs = fmaf(__builtin_amdgcn_sqrtf(an) * sg, kUp, kTiny);  // >= sg ||x||
                unsigned w[7];  // {xh01, xh23, xh45, xl01, xl23, xl45, (r, 1)}
#pragma unroll
                for (int p = 0; p < 3; ++p) split_f16(x[2 * p], x[2 * p + 1], sg, w[p], w[3 + p]);
                w[6] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(rs, kUp, kTiny), 1.0f));
                const unsigned ones = 0x14003c00u;  // {1, 2^-10}: partners of {hi, lo * 2^10} of -|c|^2
                u32x4 bLo, bUp;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const auto r = p < 3 ? __builtin_amdgcn_permlane32_swap(w[p], w[3 + p], false, false)
                                         : __builtin_amdgcn_permlane32_swap(ones, w[6], false, false);
                    bLo[p] = r[0];
                    bUp[p] = r[1];
                }
                float bL, sL, bU, sU;
                {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(f16x8, bLo), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, __builtin_bit_cast(f16x8, bLo), acc, 0, 0, 0);
                    top2<NREGS>(acc, bL, sL);
                }
                {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, __builtin_bit_cast(f16x8, bUp), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, __builtin_bit_cast(f16x8, bUp), acc, 0, 0, 0);
                    top2<NREGS>(acc, bU, sU);
                }
                const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(bL), __float_as_uint(bU), false, false);
                const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(sL), __float_as_uint(sU), false, false);
                const float b0 = __uint_as_float(rb[0]), b1 = __uint_as_float(rb[1]);
                const float s0 = __uint_as_float(rq[0]), s1 = __uint_as_float(rq[1]);
                const float second = vmed3(b0, b1, vmax(s0, s1));  // second largest upper bound u_j
                // exact similarity to the old label's centroid, kmeans.py:71-74
                const int ol = (int)((old_packed >> (8 * q)) & 0xffu);
                const float4 r0 = s4[2 * ol], r1 = s4[2 * ol + 1];
                float y = fmaf(x[0], r0.x, 0.f);
                y = fmaf(x[1], r0.y, y);
                y = fmaf(x[2], r0.z, y);
                y = fmaf(x[3], r0.w, y);
                y = fmaf(x[4], r1.x, y);
                y = fmaf(x[5], r1.y, y);
                y = y * 2.0f;
                y = y - an;
                y = y - r1.z;
                // keep <=> (Y_l + |x|^2) sg^2 exceeds every other cluster's upper bound: w - second > eps(r) + rounding of w
                const float wv = (y + an) * sg2;
                const float th = fmaf(fabsf(wv), 2.384185791015625e-7f,
                                      fmaf(rs, fmaf(rs, 1.52587890625e-5f, 9.5367431640625e-7f), 2.3283064365386963e-10f));
#ifdef ET_EXP_RF_ALL_UNDECIDED
                const bool keep = false;
#else
                const bool keep = wv - second > th;
#endif
                sim = sim + (keep ? (double)y : 0.0);
                undecided |= keep ? 0u : (1u << q);
            }
        }
        sLab[qi] = old_packed;  // (the bytes of undecided points are replaced below)
        if (undecided) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((undecided >> q) & 1u) {
                    const int slot = atomicAdd(sQn, 1);
                    if (slot < q_cap) sQ[slot] = (unsigned short)(qi * 4 + q);
                }
        }
    }
    __syncthreads();
    // ---- the undecided points: exact arg-max, four threads per point (clusters sub, sub + 4, ...; first maximum wins) ----
    const int nq = *sQn, sub = tid & 3;
    ok = nq <= q_cap;  // (more undecided points than the queue holds: the caller runs the exact scan of the whole group)
    if (!ok) return 0.0;
    for (int base = 0; base < nq; base += kFThreads / 4) {
        const int e = base + (tid >> 2);
        const bool act = e < nq;
        const int pid = sQ[act ? e : 0], qi = pid >> 2, q = pid & 3;
        float x[kD];
#pragma unroll
        for (int i = 0; i < kD; ++i) x[i] = reinterpret_cast<const float *>(x4 + i * L2 + qi)[q];
        float an = x[0] * x[0];
#pragma unroll
        for (int i = 1; i < kD; ++i) an = an + x[i] * x[i];
        float best = -__builtin_inff();
        int lb = 0x7fffffff;
        for (int j = sub; j < K; j += 4) {
            const float4 r0 = s4[2 * j], r1 = s4[2 * j + 1];
            float y = fmaf(x[0], r0.x, 0.f);
            y = fmaf(x[1], r0.y, y);
            y = fmaf(x[2], r0.z, y);
            y = fmaf(x[3], r0.w, y);
            y = fmaf(x[4], r1.x, y);
            y = fmaf(x[5], r1.y, y);
            y = y * 2.0f;
            y = y - an;
            y = y - r1.z;
            if (y > best) {
                best = y;
                lb = j;
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int ol = __shfl_xor(lb, o);
            if (ob > best || (ob == best && ol < lb)) {
                best = ob;
                lb = ol;
            }
        }
        if (act && sub == 0) {
            reinterpret_cast<uint8_t *>(sLab)[pid] = (uint8_t)lb;
            reinterpret_cast<uint8_t *>(LTg)[pid] = (uint8_t)lb;
            sim = sim + (double)best;
        }
    }
    __syncthreads();
    return sim;
}

// ---- one Lloyd iteration, first half: assignment + levels 0 and 1.  Workgroup g < G: group g; workgroup G: the tail ----
// NREGS = 0: the exact scan of every point (L = 16: four workgroups per CU); 10 / 16 (K <= 20 / 32): iterations >= 1 certify
// the labels with the matrix-core filter (L >= 32; more registers: fewer wavefronts per CU, far fewer instructions)
template <int NREGS>
__global__ __launch_bounds__(kFThreads, NREGS ? 6 : 7) void reforder_groups_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, dk = kD * K;
    unsigned char *ws = a.ws + (int64_t)blockIdx.y * a.ws_stride;
    const et_kmeans_state *state = at<et_kmeans_state>(ws, a.lay.state);
    // (the centroids are requested together with the flag: one round trip to memory, not two)
    float cpre[kD];
    {
        const float *cen0 = at<float>(ws, a.lay.cen);
#pragma unroll
        for (int i = 0; i < kD; ++i) cpre[i] = cen0[i * K + (tid < K ? tid : 0)];
    }
    const int64_t done0 = state->done, iter0 = state->iter;
    const double max_abs_x = state->max_abs_x;
    if (done0) return;  // the whole batch stopped in an earlier launch (kmeans.py:239), or bad input was flagged
    const float *X = a.X + (int64_t)blockIdx.y * a.x_stride;
    const Geo &geo = a.geo;
    const int lp = geo.lp;
    const int L = 1 << lp, L2 = L * L, RB = L / 4;
    const int64_t N = geo.N;
    unsigned *cnt = at<unsigned>(ws, a.lay.cnt);
    float4 *S1 = at<float4>(ws, a.lay.S1);
    float4 *T = at<float4>(ws, a.lay.T);
    double *Sin = at<double>(ws, a.lay.Sin);
    const float4 *XT4 = at<const float4>(ws, a.lay.XT);
    unsigned *LT32 = at<unsigned>(ws, a.lay.LT);
    uint8_t *tail_lab = at<uint8_t>(ws, a.lay.tail);

    __shared__ __attribute__((aligned(16))) float sC[(kFMaxK + 1) * 8];  // (+ a row the arg-max loop's last prefetch may read)
    __shared__ unsigned sCnt[kFMaxK];
    __shared__ double sWsum[8];
    const int TR = a.tiles_per_round;
    float *sAcc = reinterpret_cast<float *>(smem);                                          // [tile in round][coordinate][row][64 chains]
    unsigned *sLab = reinterpret_cast<unsigned *>(smem + acc_region_bytes(K, 1 << a.geo.lp, TR));  // a group's labels, one word per quad
    // the tail's workgroup is dispatched FIRST: it is as long as any other and at the highest index it used to start when the
    // last slot freed up, alone on the chip for its whole 20 us (N = 1e7: the launch ended 113 us after it began, the groups 93)
    const int64_t gidx = blockIdx.x == 0 ? geo.G : (int64_t)blockIdx.x - 1;
    const bool is_tail = gidx == geo.G;
    [[maybe_unused]] const int who = is_tail ? 1 : (gidx == 0 ? 0 : 9);
    RF_STAMP(who, 0);

    // ---- prologue: centroid rows with |c_j|^2 in ATen's order for column j of K (kmeans.py:74), NaN / overflow test ----
    int bad = 0;
    if (tid < K) {
        float sq[kMaxD];
#pragma unroll
        for (int i = 0; i < kD; ++i) {
            const float v = cpre[i];
            sC[tid * 8 + i] = v;
            sq[i] = v * v;
            bad |= !(fabsf(v) < 1e18f);
        }
        sC[tid * 8 + 6] = sqnorm_at(sq, kD, tid, K);
        sC[tid * 8 + 7] = 0.f;
    }
    __shared__ unsigned sMaxC;
    __shared__ int sQn;
    if (tid < kFMaxK) sCnt[tid] = 0u;
    if (tid == 0) {
        sMaxC = 0u;
        sQn = 0;
    }
    __syncthreads();
    if (tid < K) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < kD; ++i) m = fmaxf(m, fabsf(cpre[i]));
        atomicMax(&sMaxC, __float_as_uint(m));  // (non-negative floats order like their bit patterns; a NaN sets `bad`)
    }
    const bool nans = __syncthreads_or(bad) != 0 || !(max_abs_x < 1e18);
    double sim = 0.0;
    float4 acc1 = make_float4(0.f, 0.f, 0.f, 0.f), acc0 = acc1;
    RF_STAMP(who, 1);

    if (!is_tail) {
        // ---- assignment of the group's 4 L^2 points (kmeans.py:143-158): a quad = four consecutive steps of one chain ----
        const float4 *x4 = XT4 + gidx * kD * L2;
        bool filtered = false;
        if constexpr (NREGS > 0) {
            // power-of-two scale: every |x| sg, |c| sg < 32 (csrc/et_kmeans.hip, filter_assign_body); the first iteration (no
            // labels yet), a possible NaN or a scale whose square leaves the fp32 range: the exact scan decides
            const int e_max = exponent_above(fmax(max_abs_x, (double)__uint_as_float(sMaxC)));
            if (iter0 > 0 && !nans && K >= 3 && e_max >= -40 && e_max <= 60) {
                const int q_cap = min(4 * L2, (int)((size_t)TR * kD * (K + 1) * 64 * sizeof(float) / sizeof(unsigned short)));
                sim = assign_group_filter<NREGS>(x4, L2, sC, K, ldexpf(1.0f, 5 - e_max), sLab, LT32 + gidx * L2,
                                                 reinterpret_cast<unsigned short *>(sAcc), q_cap, &sQn, filtered);
                if (!filtered) sim = 0.0;
            }
        }
        for (int qi = tid; qi < L2 && !filtered; qi += kFThreads) {
            float4 xv[kD];
#pragma unroll
            for (int i = 0; i < kD; ++i) xv[i] = x4[i * L2 + qi];
            int lb[4];
            float bv[4];
            if (!nans) {
                quad_best(xv, sC, K, lb, bv);
            } else {  // (an empty cluster's NaN centroid, or magnitudes near the fp32 range: torch.max's NaN rule, point by point)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float x[1][kD], an[1], b1[1];
                    int l1[1];
#pragma unroll
                    for (int i = 0; i < kD; ++i) x[0][i] = p == 0 ? xv[i].x : (p == 1 ? xv[i].y : (p == 2 ? xv[i].z : xv[i].w));
                    float sacc = x[0][0] * x[0][0];
#pragma unroll
                    for (int i = 1; i < kD; ++i) sacc = sacc + x[0][i] * x[0][i];
                    an[0] = sacc;
                    points_best<true, 1>(x, an, sC, K, l1, b1);
                    lb[p] = l1[0];
                    bv[p] = b1[0];
                }
            }
            const unsigned packed = (unsigned)lb[0] | ((unsigned)lb[1] << 8) | ((unsigned)lb[2] << 16) | ((unsigned)lb[3] << 24);
            sLab[qi] = packed;
            LT32[gidx * L2 + qi] = packed;
#pragma unroll
            for (int p = 0; p < 4; ++p) sim = sim + (double)bv[p];
        }
        __syncthreads();
        for (int qi = tid; qi < L2; qi += kFThreads) {  // points per cluster, from the final labels
            const unsigned l4 = sLab[qi];
            atomicAdd(&sCnt[l4 & 255u], 1u);
            atomicAdd(&sCnt[(l4 >> 8) & 255u], 1u);
            atomicAdd(&sCnt[(l4 >> 16) & 255u], 1u);
            atomicAdd(&sCnt[l4 >> 24], 1u);
        }
        RF_STAMP(who, 2);
        RF_STAMP_MAX(0, 8);
        cascade_levels([&](int q, int rb, int ln, int i) { return x4[i * L2 + (q * RB + rb) * 64 + ln]; }, sLab, sAcc, K, L, TR, L, L,
                       acc1, acc0);
        if (tid < dk) S1[gidx * dk + tid] = acc1;
        RF_STAMP(who, 3);
        RF_STAMP_MAX(0, 9);
    } else {
        // ---- the tail: the points tail0 .. N-1 where they lie in X -- the chunks of the partial group (level 1 of their
        //      level-0 sums -> T[0 .. d K)), the lane terms after the last full chunk (level 0 -> T[d K ..)), and the
        //      N mod 4 points after the lanes' ranges (their labels -> T[2 d K]) ----
        const int64_t tail0 = geo.tail0, size = N / 4;
        const int nt = (int)(N - tail0);
        const int pc = (int)(geo.full_chunks - geo.G * L);    // full chunks of the partial group (< L)
        const int rem = (int)(size - geo.full_chunks * L);    // lane terms after them (< L)
        const int n_all = pc + (rem > 0 ? 1 : 0);
        // (the tail's label bytes live in the accumulators' space until level 0 clears it: a region of their own made the
        // launch's LDS 41.9 KB at L = 32 -- three workgroups per CU instead of four)
        uint8_t *sTail = reinterpret_cast<uint8_t *>(sAcc);
        for (int m0 = 2 * tid; m0 < nt; m0 += 2 * kFThreads) {  // two points per thread side by side
            float x[2][kD], an[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int64_t n = tail0 + (m0 + p < nt ? m0 + p : m0);
                float sq[kMaxD];
#pragma unroll
                for (int i = 0; i < kD; ++i) {
                    x[p][i] = X[(int64_t)i * N + n];
                    sq[i] = x[p][i] * x[p][i];
                }
                an[p] = sqnorm_at(sq, kD, n, N);  // the last N mod 32 columns take the 4-lane order
            }
            int lb[2];
            float bv[2];
            if (nans) points_best<true, 2>(x, an, sC, K, lb, bv);
            else points_best<false, 2>(x, an, sC, K, lb, bv);
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (m0 + p < nt) {
                    sTail[m0 + p] = (uint8_t)lb[p];
                    tail_lab[m0 + p] = (uint8_t)lb[p];
                    atomicAdd(&sCnt[lb[p]], 1u);
                    sim = sim + (double)bv[p];
                }
        }
        __syncthreads();
        RF_STAMP(who, 2);
        // the label words of the chains' steps; a step past the lane's range gets the dummy row
        const int tiles = (n_all + 15) >> 4;
        for (int w = tid; w < tiles * RB * 64; w += kFThreads) {
            const int t = w & 63, rb = (w >> 6) % RB, q = (w >> 6) / RB;
            const int c = q * 16 + (t >> 2), k = t & 3;
            unsigned word = 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * rb + u;
                const bool real = c < pc || (c == pc && r < rem);
                const unsigned lb = real ? (unsigned)sTail[4 * (c * L + r) + k] : (unsigned)K;
                word |= lb << (8 * u);
            }
            sLab[w] = word;
        }
        unsigned lw = 0u;  // labels of the N mod 4 leftover points, for the workgroup that combines the lanes
        if (tid == 0)
            for (int64_t n = size * 4; n < N; ++n) lw |= (unsigned)sTail[n - tail0] << (8 * (int)(n & 3));
        __syncthreads();
        const int64_t lim = N - tail0;
        cascade_levels(
            [&](int q, int rb, int ln, int i) {
                const float *x = X + (int64_t)i * N + tail0;
                const int64_t m0 = 4 * ((int64_t)(q * 16 + (ln >> 2)) * L + 4 * rb) + (ln & 3);
                float4 v;
                v.x = m0 < lim ? x[m0] : 0.f;
                v.y = m0 + 4 < lim ? x[m0 + 4] : 0.f;
                v.z = m0 + 8 < lim ? x[m0 + 8] : 0.f;
                v.w = m0 + 12 < lim ? x[m0 + 12] : 0.f;
                return v;
            },
            sLab, sAcc, K, L, TR, n_all, pc, acc1, acc0);
        if (tid < dk) {
            T[tid] = acc1;
            T[dk + tid] = acc0;
        }
        if (tid == 0) T[2 * dk] = make_float4(__uint_as_float(lw), 0.f, 0.f, 0.f);
        RF_STAMP(who, 3);
    }
    // ---- this workgroup's counts and similarity sum (read by the next kernel) ----
    sim = wave_sum_f64(sim);
    if (lane == 0) sWsum[wave] = sim;
    __syncthreads();
    if (tid < kFMaxK) cnt[gidx * kFMaxK + tid] = sCnt[tid];
    if (tid == 0) {
        double s = sWsum[0];
        for (int w = 1; w < kFThreads / 64; ++w) s = s + sWsum[w];
        Sin[gidx] = s;
    }
    RF_STAMP(who, 4);
    RF_STAMP_MAX(0, 10);
}

// ATen's inner (contiguous) sum (inner_sum_f32) of v[0..size) in LDS, its 32 (vector lane, slot) cascades side by side;
// scratch: 40 floats of LDS.  Called by a whole workgroup (>= 64 threads); the result is returned to every thread.
__device__ __forceinline__ float inner_sum_parallel(const float *v, int size, float *scratch) {
    const int tid = (int)threadIdx.x;
    if (size < 8) {
        if (tid == 0) scratch[0] = row_sum_f32(v, size);
        __syncthreads();
        const float r = scratch[0];
        __syncthreads();
        return r;
    }
    const int nv = size / 8, s4 = nv / 4;
    if (tid < 32) scratch[tid] = cascade_f32(v + 8 * (tid >> 3) + (tid & 7), 32, s4);  // slot k = tid / 8 of lane l = tid % 8
    __syncthreads();
    if (tid < 8) {
        float s = scratch[tid];
        for (int i = s4 * 4; i < nv; ++i) s = s + v[8 * i + tid];
        for (int k = 1; k < 4; ++k) s = s + scratch[8 * k + tid];
        scratch[32 + tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float acc = 0.f;
        for (int i = nv * 8; i < size; ++i) acc = acc + v[i];
        for (int l = 0; l < 8; ++l) acc = acc + scratch[32 + l];
        scratch[0] = acc;
    }
    __syncthreads();
    const float r = scratch[0];
    __syncthreads();
    return r;
}
