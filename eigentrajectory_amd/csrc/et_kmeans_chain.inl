// et_kmeans_chain.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): one launch per iteration: the stand-alone filter kernel, the delta fold, update_body, kmeans_lloyd_chain_kernel and its finalize kernel.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
template <int NREGS>
__global__ __launch_bounds__(kFilterMaxThreads) void kmeans_assign_filter_kernel(
    const float *__restrict__ X, int64_t N, int K, const et_kmeans_state *__restrict__ state,
    const float *__restrict__ cen, uint8_t *__restrict__ labels, long long *__restrict__ block_partials,
    long long *__restrict__ lanes) {
    if (state->done) return;
    filter_assign_body<NREGS>(X, N, K, state, cen, labels, block_partials, lanes);
}

// Fold the workgroup deltas into the shard's running totals: one workgroup per entry, unit-stride
// reads.  Cluster sums / counts accumulate across iterations (deltas), the similarity sum and
// the NaN count are per-iteration quantities and are overwritten.
__global__ __launch_bounds__(kKmThreads) void kmeans_reduce_partials_kernel(const long long *__restrict__ block_partials,
                                                                            int n_blocks, int plen, int full,
                                                                            const et_kmeans_state *__restrict__ state,
                                                                            long long *totals, long long *partials) {
    if (state->done) return;
    __shared__ long long sW[kKmThreads / 64];
    const int e = blockIdx.x;
    long long s = 0;
    for (int b = threadIdx.x; b < n_blocks; b += kKmThreads) s += block_partials[(size_t)e * n_blocks + b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sW[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKmThreads / 64; ++w) s += sW[w];
        // the running totals stay in the workspace; the caller's buffer receives a copy it may overwrite (all-reduce
        // in place)
        const bool running = !full && state->iter > 0 && e < plen - 2;
        const long long tot = running ? totals[e] + s : s;
        totals[e] = tot;
        partials[e] = tot;
    }
}

// centroid update + convergence scalars from the (all-reduced) exact sums.  One workgroup.
// `partials` may live in global memory or in LDS (flat addressing); `pre` = the state block if the caller has
// already loaded it.
// `last` (may be null): {d*K floats, then one int64 at the next 8-byte boundary} receives the centroids and sim_frac
// the assignment just consumed was made with -- what kmeans_inertia_kernel needs to evaluate its inertia afterwards.
// x[lane + O] for the lanes that are multiples of 2 O (what a level of a "x[i] += x[i + O]" tree needs), without the LDS
// crossbar: inside a row of 16 lanes a DPP row shift, across rows v_permlane16_swap / v_permlane32_swap.  (__shfl_down is a
// ds_bpermute per 32-bit half and ~130 cycles per level; the update's two reduction trees were ~800 cycles of every
// launch's prologue, profiles/r04k_lloyd_launch_stamps.txt.)
template <int O>
__device__ __forceinline__ unsigned lane_down_u32(unsigned v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "a power of two below the wavefront size");
    if constexpr (O < 16) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + O, 0xf, 0xf, true);  // row_shl:O
    } else if constexpr (O == 16) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // second result: rows (1, 1, 3, 3)
        return r.y;
    } else {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // second result: (upper half, upper half)
        return r.y;
    }
}
template <int O>
__device__ __forceinline__ double lane_down_f64(double v) {
    const unsigned lo = lane_down_u32<O>((unsigned)__double2loint(v)), hi = lane_down_u32<O>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
template <int O>
__device__ __forceinline__ unsigned long long lane_down_u64(unsigned long long v) {
    const unsigned lo = lane_down_u32<O>((unsigned)v), hi = lane_down_u32<O>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// the minimum of a 64-bit key over the wavefront, in lane 0 (register exchanges only)
__device__ __forceinline__ unsigned long long wave_min_u64_lane0(unsigned long long key) {
    unsigned long long o;
    o = lane_down_u64<32>(key); key = o < key ? o : key;
    o = lane_down_u64<16>(key); key = o < key ? o : key;
    o = lane_down_u64<8>(key); key = o < key ? o : key;
    o = lane_down_u64<4>(key); key = o < key ? o : key;
    o = lane_down_u64<2>(key); key = o < key ? o : key;
    o = lane_down_u64<1>(key); key = o < key ? o : key;
    return key;
}

struct NoSideWork {
    __device__ __forceinline__ void operator()(int) const {}
};
// `side(w)`: work for wavefront 2 + w of the workgroup, run beside the reductions (between the update's two barriers) --
// it may read the new centroids in `cen`
template <class Side = NoSideWork>
__device__ __forceinline__ void update_body(et_kmeans_state *state, const long long *partials, int d, int K, float tol,
                                            float *cen, float *trace, const et_kmeans_state *pre = nullptr,
                                            float *last = nullptr, bool need_inertia = true,
                                            const float *pk_hdr = nullptr, int *pk_bad = nullptr, Side side = Side()) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sSq = reinterpret_cast<float *>(smem_raw);  // d*K squared differences
    float *sNew = sSq + d * K;
    // every global value the serial tail needs is fetched up front (one round trip instead of a chain of them)
    const et_kmeans_state st = pre ? *pre : *state;
    const long long sim_sum = partials[d * K + K], nan_count = partials[d * K + K + 1];
    const int frac = (int)st.frac;
    const double inv_scale = ldexp(1.0, -frac);
    for (int e = tx; e < d * K; e += (int)blockDim.x) {
        const int j = e % K;
        const long long cnt = partials[d * K + j];
        float c;
        if (cnt == 0) c = __int_as_float(0x7fc00000);  // 0/0 (kmeans.py:182)
        else c = (float)(((double)partials[e] * inv_scale) / (double)cnt);
        const float prev = cen[e];
        const float diff = prev - c;  // kmeans.py:48
        sSq[e] = diff * diff;         // :49
        sNew[e] = c;
        cen[e] = c;
        if (last) last[e] = prev;
    }
    if (last && tx == 0) *reinterpret_cast<long long *>(last + ((d * K + 1) & ~1)) = (long long)st.sim_frac;
    __syncthreads();
    // max |c| (NaN ignored), smallest non-zero |c| and a non-finite flag, reduced by the first wavefront
    __shared__ float sRed[3];
    __shared__ double sErr;
    if (blockDim.x <= 64 || (tx >> 6) == 1) {
        // kmeans.py:50 in the oracle's fixed order (oracle/et_oracle.c: eto_error_sum): fp64, blocks of 256 consecutive
        // terms, each a balanced tree x[i] += x[i + s], s = 1 ... 128, block results added in block order.  A lane holds
        // four consecutive terms (levels s = 1, 2), the lanes combine through shuffles (s = 4 ... 128): seven dependent
        // additions instead of the d K of a running sum (1.7 us of every Lloyd launch's prologue with d K = 120).
        // It runs on the second wavefront next to the reductions below.
        const int l = tx & 63, dk = d * K;
        double total = 0.0;
        for (int b0 = 0; b0 < dk; b0 += 256) {
            float f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = b0 + 4 * l + q < dk ? sSq[b0 + 4 * l + q] : 0.f;
            double t = ((double)f[0] + (double)f[1]) + ((double)f[2] + (double)f[3]);
            // t += t[lane + o], o = 1 ... 32: valid in the lanes that are multiples of 2 o
            t = t + lane_down_f64<1>(t);
            t = t + lane_down_f64<2>(t);
            t = t + lane_down_f64<4>(t);
            t = t + lane_down_f64<8>(t);
            t = t + lane_down_f64<16>(t);
            t = t + lane_down_f64<32>(t);
            total = total + t;
        }
        if (l == 0) sErr = total;
    }
    if (tx >= 128) side((int)(tx >> 6) - 2);
    if (tx < 64) {
        float mx = 0.f;
        unsigned mn = 0x7f800000u;
        int bad = 0;
        // (d K <= 192 for the shapes the chained kernel takes: up to three values per lane, requested from LDS together with
        // the packed copy's header -- a loop with a dependent header read per value was 1 000 of this phase's 1 600 cycles)
        float h[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pk_hdr) {
#pragma unroll
            for (int q = 0; q < 7; ++q) h[q] = pk_hdr[q];
        }
        for (int e0 = tx; e0 < d * K; e0 += 3 * 64) {
            float v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = sNew[e0 + 64 * u < d * K ? e0 + 64 * u : e0];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = e0 + 64 * u;
                if (e >= d * K) break;
                const float a = fabsf(v[u]);
                if (!(a <= 3.402823466e+38f)) bad = 1;
                if (a > mx) mx = a;  // false for NaN: ignored, like the oracle
                const unsigned b = (unsigned)__float_as_int(a);
                if (a <= 3.402823466e+38f && b != 0u && b < mn) mn = b;
                if (pk_hdr) {  // packed_assign_body's range test of the new centroids (bit 1 of `bad`), while they are at hand
                    float mu = h[0];
#pragma unroll
                    for (int q = 1; q < 6; ++q) mu = (q < d && e >= q * K) ? h[q] : mu;  // mu[e / K]
                    if (!(fabsf((v[u] - mu) * h[6]) < 31.0f)) bad |= 2;
                }
            }
        }
        // (order independent: any tree; lane 0 ends up with the result)
#define ET_DOWN(O)                                                                        \
    do {                                                                                  \
        mx = fmaxf(mx, __uint_as_float(lane_down_u32<O>(__float_as_uint(mx))));           \
        const unsigned other = lane_down_u32<O>(mn);                                      \
        mn = other < mn ? other : mn;                                                     \
        bad |= (int)lane_down_u32<O>((unsigned)bad);                                      \
    } while (0)
        ET_DOWN(32);
        ET_DOWN(16);
        ET_DOWN(8);
        ET_DOWN(4);
        ET_DOWN(2);
        ET_DOWN(1);
#undef ET_DOWN
        if (tx == 0) {
            sRed[0] = mx;
            sRed[1] = __int_as_float((int)mn);
            sRed[2] = (bad & 1) ? 1.f : 0.f;
            if (pk_bad) *pk_bad = bad >> 1;
        }
    }
    __syncthreads();
    if (tx == 0) {
        const float error = (float)sErr;
        const int64_t n_total = st.n_total;
        // (need_inertia == false: a trace-less fit's launches, which do not accumulate the similarity sum -- the inertia of
        // the last assignment is evaluated after the loop -- and the workgroups that publish nothing: an fp64 division
        // less on the serial tail of every launch's prologue)
        float inertia = (float)st.inertia;
        if (need_inertia) {
            if (nan_count > 0) inertia = __int_as_float(0x7fc00000);
            else inertia = (float)(-(((double)sim_sum * ldexp(1.0, -(int)st.sim_frac)) / (double)n_total));  // :57
        }
        const double mc = (double)sRed[0];
        const double mx = st.max_abs_x;
        state->max_abs_c = mc;
        state->sim_frac = sim_frac_bits(mx, mc, d, n_total);
        int64_t fast = 0;
        if (sRed[2] == 0.f && mx < 1e18 && mc < 1e18) {
            const unsigned lim = 0x26800000u;  // 2^-50, see the fast_ok levels above
            fast = ((unsigned)__float_as_int(sRed[1]) >= lim && (unsigned long long)st.min_nz_x_bits >= lim) ? 2 : 1;
        }
        state->fast_ok = fast;
        if (trace) {
            trace[2 * st.iter] = error;
            trace[2 * st.iter + 1] = inertia;
        }
        state->error = (double)error;
        state->inertia = (double)inertia;
        state->iter = st.iter + 1;
        state->done = (error <= tol) ? 1 : 0;  // kmeans.py:239 (NaN -> keep going)
    }
}

__global__ __launch_bounds__(kKmThreads) void kmeans_update_kernel(et_kmeans_state *state,
                                                                   const long long *__restrict__ partials, int d, int K,
                                                                   float tol, float *__restrict__ cen,
                                                                   float *__restrict__ trace) {
    if (state->done) return;
    update_body(state, partials, d, K, tol, cen, trace);
}

// BatchKMeans.fit on a batch of l > 1 problems (kmeans.py:228-240): ONE error -- the squared centroid movement summed
// over all problems (kmeans.py:45-51 on the (l, d, K) tensors) -- is compared with the tolerance and all problems stop
// together.  The step API runs the problems side by side with a tolerance no error can meet; this kernel, after their
// updates, sums the per-problem errors (fp64, problem order; each is the fp32 value the update stored) and sets every
// problem's convergence flag from the sum.  One wavefront.
__global__ void kmeans_joint_done_kernel(et_kmeans_state *const *__restrict__ states, int n, float tol) {
    if (threadIdx.x != 0) return;
    if (states[0]->done) return;  // (the flags are only ever set together)
    double sum = 0.0;
    for (int b = 0; b < n; ++b) sum += states[b]->error;
    const int64_t done = ((float)sum <= tol) ? 1 : 0;  // kmeans.py:239 (NaN -> keep going)
    for (int b = 0; b < n; ++b) states[b]->done = done;
}

// Large shards, single-GPU fit: ONE launch per Lloyd iteration and NO serial section between two iterations.
//
// A launch first applies the update of the PREVIOUS iteration's assignment and then makes its own assignment:
// every workgroup folds the 16 copies of the exact integer totals the previous launch's workgroups added their
// deltas onto, and computes the new centroids, error and convergence flag ITSELF, straight into its LDS staging --
// identical integers in, identical results in every workgroup, so nobody waits for a "last" workgroup (the ticket +
// fence + one-workgroup fold and update + dispatch gap of the two-phase form cost ~9 of the ~20 us that an iteration
// takes besides streaming the points).  Workgroup 0 also publishes the results (state, centroids, totals, trace,
// the centroids of the last assignment).  Nothing a workgroup reads is written during the same launch:
//   state / centroids / totals   two copies, launch t reads copy t % 2 and workgroup 0 writes copy (t+1) % 2
//   the 16-copy delta table       three copies: launch t reads t % 3 (filled by launch t-1), adds onto (t+1) % 3 and
//                                workgroup 0 clears (t+2) % 3 (read by launch t-1, to be filled by launch t+1)
// The assignment of the final iteration is followed by kmeans_chain_finalize_kernel (its update, once).
struct LloydChain {
    const et_kmeans_state *st_rd;
    et_kmeans_state *st_wr;
    const float *cen_rd;
    float *cen_wr;
    const long long *tot_rd;
    long long *tot_wr;
    const long long *lanes_rd;
    long long *lanes_wr;
    long long *lanes_zero;
    float *last;
    unsigned long long *mail;  // host-visible progress word (et_hostring.h: mailbox), or nullptr
    // sharded loop: ONE copy of the delta table, entries adjacent (what travels over the wire between two launches is
    // then the d K + K + 2 int64 that carry the information, 1.1 KB, not the 16-copy table)
    int compact;
    int copies;  // compact copies of the delta table in use (1: sharded loop -- the wire format; a power of two <= 8 else)
    int vec_ok;  // this shard's rows allow 16-byte loads (N % 4 == 0, aligned) and it has >= 1024 points: filter body
    LloydPacked pk;  // pk.xh != nullptr: trace-less iterations run on the packed copy (packed_assign_body)
};

// fold the 16 copies of every total (layout [entry][copy]: a linear, coalesced sweep, 16 adjacent lanes per entry) and
// add the running totals of the earlier iterations -> sTot (LDS)
// The fold of the kAccLanes copies of the delta table onto the previous totals, in two halves so that the caller can
// put its other loads between them: fold_issue() requests every value (clamped indices keep the register arrays out
// of scratch memory), fold_combine() sums.  d = 6, K <= 32 with the filter kernels' 768 / 1024 threads needs 3 ... 5
// sweeps of blockDim.x entries; fold_lanes() is the plain loop for any other shape.
constexpr int kTimedRun = 4;  // launches between the two events of a timed sample of the chained loop
constexpr int kFoldSweeps = 5;
struct FoldRegs {
    long long v[kFoldSweeps], prev[kFoldSweeps];
};
__device__ __forceinline__ bool fold_fits(int plen) { return kFoldSweeps * (int)blockDim.x >= plen * kAccLanes; }
__device__ __forceinline__ void fold_issue(const long long *__restrict__ lanes, const long long *__restrict__ tot_prev,
                                           int plen, FoldRegs &r, bool compact = false, int copies = 1) {
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    if (compact) {  // entries adjacent: one load per entry and copy (plen <= 226 <= blockDim.x), all requested together
        const int ci = (int)threadIdx.x < plen ? (int)threadIdx.x : 0;
        const int pitch = compact_pitch(plen);
        long long v = lanes[ci];
        long long x[3] = {0, 0, 0};
        if (copies > 1) x[0] = lanes[ci + pitch];
        if (copies > 2) {
            x[1] = lanes[ci + 2 * pitch];
            x[2] = lanes[ci + 3 * pitch];
        }
        long long y = 0;
        for (int c = 4; c < copies; ++c) y += lanes[ci + c * pitch];
        r.v[0] = ((v + x[0]) + (x[1] + x[2])) + y;  // (integers: any order)
        r.prev[0] = tot_prev[ci];
        return;
    }
#pragma unroll
    for (int it = 0; it < kFoldSweeps; ++it) {
        const int idx = it * n_threads + (int)threadIdx.x;
        const int ci = idx < total ? idx : 0;
        r.v[it] = lanes[ci];
        r.prev[it] = tot_prev[ci / kAccLanes];
    }
}
__device__ __forceinline__ void fold_combine(const FoldRegs &r, bool have_prev, int plen, long long *sTot,
                                             bool compact = false) {
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    if (compact) {
        const int e = (int)threadIdx.x;
        if (e < plen) sTot[e] = ((have_prev && e < plen - 2) ? r.prev[0] : 0) + r.v[0];
        return;
    }
#pragma unroll
    for (int it = 0; it < kFoldSweeps; ++it) {
        const int idx = it * n_threads + (int)threadIdx.x;
        if (it * n_threads >= total) break;  // uniform
        long long x = idx < total ? r.v[it] : 0;
#pragma unroll
        for (int o = kAccLanes / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (idx < total && (idx & (kAccLanes - 1)) == 0) {
            const int e = idx / kAccLanes;
            sTot[e] = ((have_prev && e < plen - 2) ? r.prev[it] : 0) + x;
        }
    }
}
__device__ __forceinline__ void fold_lanes(const long long *__restrict__ lanes, const long long *__restrict__ tot_prev,
                                           bool have_prev, int plen, long long *sTot, bool compact = false, int copies = 1) {
    if (compact) {
        for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) {
            long long v = lanes[e];
            for (int c = 1; c < copies; ++c) v += lanes[e + c * compact_pitch(plen)];
            sTot[e] = ((have_prev && e < plen - 2) ? tot_prev[e] : 0) + v;
        }
        return;
    }
    if (fold_fits(plen)) {
        FoldRegs r;
        fold_issue(lanes, tot_prev, plen, r);
        fold_combine(r, have_prev, plen, sTot);
        return;
    }
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    for (int base = 0; base < total; base += n_threads) {
        const int idx = base + (int)threadIdx.x;
        long long v = idx < total ? lanes[idx] : 0;
#pragma unroll
        for (int o = kAccLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (idx < total && (idx & (kAccLanes - 1)) == 0) {
            const int e = idx / kAccLanes;
            sTot[e] = ((have_prev && e < plen - 2) ? tot_prev[e] : 0) + v;
        }
    }
}

template <int NREGS, bool SIM, bool FIRST = false>
__global__ __launch_bounds__(kFilterMaxThreads) void kmeans_lloyd_chain_kernel(
    const float *__restrict__ X, int64_t N, int K, const LloydChain ch, uint8_t *__restrict__ labels, float tol,
    float *trace, int has_pending) {
    constexpr int d = 6;
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ et_kmeans_state sSt;
    // scratch of this prologue, inside the area the assignment's LDS queues use later: the folded totals past the
    // 2 d K floats update_body works in, then the centroids (old -> new, in place)
    long long *sTot = reinterpret_cast<long long *>(smem_raw) + 512;
    float *sCen = reinterpret_cast<float *>(sTot + ((plen + 1) & ~1));
    const bool wg0 = blockIdx.x == 0;
    // (no local copy of the state block: a by-value et_kmeans_state whose address is taken ends up in scratch memory,
    // and a kernel with a private segment pays for it at every wavefront launch)
    // Everything the prologue needs from memory is requested at once -- the convergence flag, the centroids (d K <= 192
    // <= blockDim.x values), the delta table and the previous totals: one round trip, not three dependent ones.
    KM_PSTAMP(0);
    const int64_t done0 = ch.st_rd->done, iter0 = ch.st_rd->iter;
    __shared__ float sPkHdr[12];
    __shared__ u32x4 sPkTab[64];  // packed_assign_body's matrix operand per lane, when made beside the update
    bool tables_ready = false;
    __shared__ int sPkBad;  // the packed copy's range test of the new centroids, made by update_body (-1: not made)
    if (threadIdx.x == 0) sPkBad = -1;
    float pk_word = 0.f;
    if constexpr (!SIM) {  // the packed copy's header (9 words), with the other prologue loads
        if (ch.pk.xh && threadIdx.x < 9) pk_word = reinterpret_cast<const float *>(ch.pk.hdr)[threadIdx.x];
    }
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    const unsigned st_word = reinterpret_cast<const unsigned *>(ch.st_rd)[(int)threadIdx.x < kStateWords ? (int)threadIdx.x : 0];
    const float cen0 = ch.cen_rd[(int)threadIdx.x < d * K ? (int)threadIdx.x : 0];
    FoldRegs fr;
    // (the delta table of the chained loop is ALWAYS the compact one-copy form -- host side, chain_for() --: a constant here,
    // so that the sweeps of the 16-copy form are not compiled in; their register arrays, indexed under a runtime flag, ended
    // up in scratch memory: a store -> load round trip in every launch's prologue and a private segment per wavefront)
    fold_issue(ch.lanes_rd, ch.tot_rd, plen, fr, true, ch.copies);
    if (done0) {  // converged earlier (or bad input flagged before the loop): keep the published copies in step
        if (wg0) {
            if (threadIdx.x == 0) {
                *ch.st_wr = *ch.st_rd;
                if (ch.mail)  // the host stops launching as soon as it reads the flag (it would otherwise spin for it)
                    __hip_atomic_store(ch.mail, (1ull << 63) | (unsigned long long)iter0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
            for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) ch.cen_wr[e] = ch.cen_rd[e];
            for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) ch.tot_wr[e] = ch.tot_rd[e];
        }
        return;
    }
#ifdef ET_EXP_WAITSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KM_PSTAMP(1);
    if ((int)threadIdx.x < d * K) sCen[threadIdx.x] = cen0;
    if (!SIM && threadIdx.x < 9) sPkHdr[threadIdx.x] = pk_word;
    if ((int)threadIdx.x < kStateWords) reinterpret_cast<unsigned *>(&sSt)[threadIdx.x] = st_word;  // the state block, word by word
    if (has_pending) {
        fold_combine(fr, iter0 > 0, plen, sTot, true);
        __syncthreads();
        KM_PSTAMP(2);
        // (trace-less fit on the packed copy: the next assignment's tables are made by wavefronts 2, 3 and 4 beside the
        // update's reductions -- 1.3 us of every launch's prologue when they followed it)
        const bool side_tables = !SIM && ch.pk.xh && ch.vec_ok && blockDim.x >= 320;
        auto side = [&](int w) {
            if (side_tables && w < 3) packed_tables_side(w, (int)(threadIdx.x & 63), sCen, sPkHdr, K, sPkTab);
        };
        update_body(&sSt, sTot, d, K, tol, sCen, wg0 ? trace : nullptr, nullptr, wg0 ? ch.last : nullptr, SIM && wg0,
                    (!SIM && ch.pk.xh) ? sPkHdr : nullptr, &sPkBad, side);  // reads its copy in LDS
        tables_ready = side_tables;
        KM_PSTAMP(3);
    }
    __syncthreads();
    const int64_t done1 = sSt.done;  // (requested together: the flag of the update just applied and its range test)
    const int range_bad = sPkBad;
    if (wg0) {  // publish (read by the next launch, the host's convergence polling and the finalize kernel)
        if (threadIdx.x == 0) {
            *ch.st_wr = sSt;
            if (ch.mail)
                __hip_atomic_store(ch.mail, ((unsigned long long)(sSt.done != 0) << 63) | (unsigned long long)sSt.iter,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) ch.cen_wr[e] = sCen[e];
        if (has_pending)
            for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) ch.tot_wr[e] = sTot[e];
        for (int i = threadIdx.x; i < ch.copies * compact_pitch(plen); i += (int)blockDim.x) ch.lanes_zero[i] = 0;  // (the compact copies in use)
        // sCen / sTot lie inside the area the assignment bodies clear for their accumulators: this workgroup's other
        // wavefronts must not start clearing while the ones above still read (uniform per workgroup: only workgroup 0 waits)
        __syncthreads();
    }
    KM_PSTAMP(4);
    if (done1) return;  // the update just applied met the tolerance: no further assignment (kmeans.py:239)
    const int copy_mask = -ch.copies;  // (-1: one compact copy)
    // a shard whose rows do not allow 16-byte loads (sharded runs cut the points anywhere), or a tiny one: the plain exact
    // scan, one point per lane, inside the same launch -- which loop form a sharded fit takes then depends on (d, K)
    // alone and every rank knows it without asking the others
    if (ch.vec_ok) {
        if constexpr (!SIM) {
            if (ch.pk.xh) {
                packed_assign_body<NREGS, FIRST>(ch.pk, sPkHdr, X, N, K, &sSt, sCen, labels, ch.lanes_wr, copy_mask, range_bad,
                                          tables_ready ? sPkTab : nullptr);
                return;
            }
        }
        filter_assign_body<NREGS, SIM>(X, N, K, &sSt, sCen, labels, nullptr, ch.lanes_wr, copy_mask);
    } else {
        assign_body_valu<6, 1>(X, N, d, K, &sSt, sCen, nullptr, labels, nullptr, ch.lanes_wr, copy_mask);
    }
}

// After the loop: the update that belongs to the last assignment (if one is pending), into the caller's buffers.
__global__ __launch_bounds__(kKmThreads) void kmeans_chain_finalize_kernel(const LloydChain ch, et_kmeans_state *state,
                                                                           long long *partials, float *cen, int d, int K,
                                                                           float tol, float *trace, int has_pending,
                                                                           long long *sim_total, int last_was_sim = 0) {
    if (sim_total && threadIdx.x < 2) sim_total[threadIdx.x] = 0;  // for the inertia pass that follows a trace-less fit
    // sim_total[2]: the pending assignment was made by a launch that accumulated the similarity sum (the trace-less loop's
    // LAST launch when it runs to max_iter) -- the update below turns it into the inertia and the inertia pass is skipped
    if (sim_total && threadIdx.x == 2) sim_total[2] = (last_was_sim && has_pending && !ch.st_rd->done) ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int plen = d * K + K + 2;
    long long *sTot = reinterpret_cast<long long *>(smem_raw) + 512;
    for (int e = threadIdx.x; e < d * K; e += kKmThreads) cen[e] = ch.cen_rd[e];
    if (ch.st_rd->done || !has_pending) {
        if (threadIdx.x == 0) *state = *ch.st_rd;
        for (int e = threadIdx.x; e < plen; e += kKmThreads) partials[e] = ch.tot_rd[e];
        return;
    }
    fold_lanes(ch.lanes_rd, ch.tot_rd, ch.st_rd->iter > 0, plen, sTot, ch.compact != 0, ch.copies);
    __syncthreads();
    for (int e = threadIdx.x; e < plen; e += kKmThreads) partials[e] = sTot[e];
    if (threadIdx.x == 0) *state = *ch.st_rd;
    __syncthreads();
    update_body(state, sTot, d, K, tol, cen, trace, ch.st_rd, ch.last);
}
