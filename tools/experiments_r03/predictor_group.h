// predictor_group.h -- the GROUP form of the predictor kernels (included by predictor.hip inside its namespace).
//
// Row split alone (predictor.hip) gives every utterance n workgroups, each streaming 1/n of the 2.67 MB weight set per
// frame through its CU's L2 port; at the 128-utterance share of config 5 that is n = 2 and 1.33 MB per CU and frame,
// and the port (measured 54 of 64 B/clk while it streams) is two thirds of the frame.  Here U utterances form a group
// on n workgroups: each workgroup evaluates its slice of the gate rows for ALL U utterances from ONE pass over its
// 1/n of the weights (U fmaf chains per loaded weight quad), so with the same number of workgroups (B n / U) the bytes
// per CU and frame fall by U: U = 2, n = 4 -> 0.67 MB; U = 4, n = 8 -> 0.33 MB.  Every chain is the chain of the
// single-workgroup form (same k order, same segments, same trees): bit-identical results.
//   * the slices of the new states change hands as in the row split (tagged 8-byte granules), U slices per hop;
//   * output layer and ReLU run for all U utterances on every workgroup (same inputs, same bits);
//   * the encoder's searches are NOT replicated: utterance u's residual, searches and outputs belong to workgroup
//     u * (n / U) of the group, and the next input row travels to the others in a third hop per frame.
// Reference: Wavernn.forward / Wavernn.encoder (wavernn.py:69-95, 150-254), as predictor.hip.

constexpr int PR = 3 * MAX_H1 / 2;  // gate rows of one workgroup's slice of a GRU (n >= 2)
constexpr int GW = FPC_GW;               // rolling window of the shared chains

template <int U>
struct __attribute__((aligned(16))) GrpLds : SearchLds {
    float x[U][MAX_IN];
    float h1[U][MAX_H1];
    float h2[U][MAX_H2];
    float pi[U][4][PR];  // segment sums [utterance][segment][row of this workgroup's slice: gate * Hs + unit]
    float ph[U][4][PR];
    float pf[U][8][MAX_FC];
    float relu[U][MAX_H2];
    float fo[U][MAX_FC];
#ifdef FPC_PRED_PROF
    long long pprof[12], plast;
#endif
};

// the load of ld4(), placed after the fmaf's of all U chains (see ld4_after)
template <int U>
__device__ __forceinline__ v4f ld4_after_g(const float* q, const float4 (&a)[U]) {
    v4f r;
    if constexpr (U == 1)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(q), "v"(a[0].x), "v"(a[0].y), "v"(a[0].z), "v"(a[0].w));
    else if constexpr (U == 2)
        asm volatile("global_load_dwordx4 %0, %1, off"
                     : "=v"(r)
                     : "v"(q), "v"(a[0].x), "v"(a[0].y), "v"(a[0].z), "v"(a[0].w), "v"(a[1].x), "v"(a[1].y), "v"(a[1].z),
                       "v"(a[1].w));
    else {
        static_assert(U == 4, "group sizes: 1, 2, 4");
        asm volatile("global_load_dwordx4 %0, %1, off"
                     : "=v"(r)
                     : "v"(q), "v"(a[0].x), "v"(a[0].y), "v"(a[0].z), "v"(a[0].w), "v"(a[1].x), "v"(a[1].y), "v"(a[1].z),
                       "v"(a[1].w), "v"(a[2].x), "v"(a[2].y), "v"(a[2].z), "v"(a[2].w), "v"(a[3].x), "v"(a[3].y),
                       "v"(a[3].z), "v"(a[3].w));
    }
    return r;
}
// the input values of 4 consecutive k for the U utterances (one ds_read_b128 each: v, vs and k are multiples of 4 floats)
template <int U>
__device__ __forceinline__ void read_hv(float (&hv)[U][4], const float* v, int vs) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(v + u * vs);
        hv[u][0] = t.x;
        hv[u][1] = t.y;
        hv[u][2] = t.z;
        hv[u][3] = t.w;
    }
}
template <int U, int J = 0>
__device__ __forceinline__ void last_window_g(float4 (&a)[U], const float* v, int vs, v4f (&w)[GW]) {
    if constexpr (J < GW) {
        float hv[U][4];
        read_hv<U>(hv, v + J, vs);
        landed<GW - 1 - J>(w[J]);
#pragma unroll
        for (int u = 0; u < U; ++u) fma4(a[u], hv[u][0], w[J]);
        landed<GW - 2 - J>(w[J + 1]);
#pragma unroll
        for (int u = 0; u < U; ++u) fma4(a[u], hv[u][1], w[J + 1]);
        landed<GW - 3 - J>(w[J + 2]);
#pragma unroll
        for (int u = 0; u < U; ++u) fma4(a[u], hv[u][2], w[J + 2]);
        landed<GW - 4 - J>(w[J + 3]);
#pragma unroll
        for (int u = 0; u < U; ++u) fma4(a[u], hv[u][3], w[J + 3]);
        last_window_g<U, J + 4>(a, v, vs, w);
    }
}
// chain4 for U input vectors v, v + vs, ... (floats): one pass over the weights, U accumulator quads
template <int U>
__device__ __forceinline__ void chain4g(const float* __restrict__ wT, const float* v, int vs, int K, int R, int r, v4f& a0,
                                        float4 (&a)[U]) {
    static_assert(GW % 4 == 0, "window: whole ds_read_b128 blocks");
    const float* q = wT + r;
    const int nb = K / GW;
    int rem = K - nb * GW;
    v4f w[GW], wt[CT];
    if (nb > 0) {
#pragma unroll
        for (int j = 0; j < GW; ++j, q += R) w[j] = ld4(q);
    }
    if (nb > 0)
        landed<GW - 1>(a0);
    else
        landed<0>(a0);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = make_float4(a0.x, a0.y, a0.z, a0.w);
    for (int b = 0; b + 1 < nb; ++b, v += GW) {
#pragma unroll
        for (int jj = 0; jj < GW; jj += 4) {
            float hv[U][4];
            read_hv<U>(hv, v + jj, vs);
#pragma unroll
            for (int j = 0; j < 4; ++j, q += R) {
                landed<GW - 1>(w[jj + j]);
#pragma unroll
                for (int u = 0; u < U; ++u) fma4(a[u], hv[u][j], w[jj + j]);
                w[jj + j] = ld4_after_g<U>(q, a);
            }
        }
    }
    const int t0 = rem < CT ? rem : CT;
#pragma unroll
    for (int j = 0; j < CT; ++j)
        if (j < t0) wt[j] = ld4(q + (size_t)j * R);
    q += (size_t)t0 * R;
    if (nb > 0) {
        last_window_g<U>(a, v, vs, w);
        v += GW;
    }
    while (rem > 0) {
#pragma unroll
        for (int j = 0; j < CT; ++j)
            if (j < rem) {
                landed<0>(wt[j]);
#pragma unroll
                for (int u = 0; u < U; ++u) fma4(a[u], v[u * vs + j], wt[j]);
            }
        rem -= CT;
        v += CT;
        if (rem > 0) {
#pragma unroll
            for (int j = 0; j < CT; ++j)
                if (j < rem) wt[j] = ld4(q + (size_t)j * R);
            q += (size_t)CT * R;
        }
    }
}

// both mat-vecs of a GRU layer for this workgroup's slice of the units and all U utterances (gru_rows)
template <int U>
__device__ __forceinline__ void gru_rows_g(const float* __restrict__ wiT, const float* __restrict__ whT,
                                           const float* __restrict__ bi, const float* __restrict__ bh, const float* x, int xs,
                                           int K, const float* h, int hs, int H, GrpLds<U>& L, int tid, int nsplit, int half) {
    const int R = 3 * H;
    const int Qg = H / 4 / nsplit, Hs = 4 * Qg;
    const int Q = 3 * Qg;
    const int Si = segments(K), Sh = segments(H);
    const int n_h = Q * Sh, n_all = n_h + Q * Si;
    for (int it = tid; it < n_all; it += NT) {
        const bool is_h = it < n_h;
        const int j = is_h ? it : it - n_h;
        const int q = j % Q, sg = j / Q;
        const int gate = q / Qg, qq = q - gate * Qg;
        const int len = (is_h ? H : K) / (is_h ? Sh : Si), k0 = sg * len, r = gate * H + 4 * (half * Qg + qq);
        v4f a0 = {0.f, 0.f, 0.f, 0.f};
        if (sg == 0) a0 = ld4(&(is_h ? bh : bi)[r]);
        float4 a[U];
        chain4g<U>((is_h ? whT : wiT) + (size_t)k0 * R, (is_h ? h : x) + k0, is_h ? hs : xs, len, R, r, a0, a);
        const int lr = gate * Hs + 4 * qq;
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<float4*>(&(is_h ? L.ph : L.pi)[u][sg][lr]) = a[u];
    }
    __syncthreads();
}
__device__ __forceinline__ float seg_tree_g(const float (*p)[PR], int S, int row) {
    if (S == 4) return (p[0][row] + p[1][row]) + (p[2][row] + p[3][row]);
    if (S == 2) return p[0][row] + p[1][row];
    return p[0][row];
}
template <int U>
__device__ __forceinline__ void gru_gates_g(int K, float* h, int hs, int H, GrpLds<U>& L, int tid, int nsplit, int half) {
    const int Si = segments(K), Sh = segments(H);
    const int Hs = H / nsplit;
    for (int idx = tid; idx < U * Hs; idx += NT) {
        const int u = idx / Hs, ii = idx - u * Hs, i = half * Hs + ii;
        const float gir = seg_tree_g(L.pi[u], Si, ii), giz = seg_tree_g(L.pi[u], Si, Hs + ii),
                    gin = seg_tree_g(L.pi[u], Si, 2 * Hs + ii);
        const float ghr = seg_tree_g(L.ph[u], Sh, ii), ghz = seg_tree_g(L.ph[u], Sh, Hs + ii),
                    ghn = seg_tree_g(L.ph[u], Sh, 2 * Hs + ii);
        const float r = fpc_sigmoidf(gir + ghr);
        const float z = fpc_sigmoidf(giz + ghz);
        const float n = fpc_tanhf(fmaf(r, ghn, gin));
        h[u * hs + i] = fmaf(z, h[u * hs + i] - n, n);
    }
    __syncthreads();
}
// one hop of the group: the U slices of this workgroup go out under a new epoch, every other slice is picked up
// (granules [u][H]; ends with a barrier)
template <int U>
__device__ __forceinline__ void hop_g(float* h, int hs, int H, SplitCtx& X, unsigned long long* g, int tid) {
    const int Hs = H / X.n, mine = X.half * Hs;
    const unsigned epoch = ++X.epoch;
    if (!X.withhold)
        for (int idx = tid; idx < U * Hs; idx += NT) {
            const int u = idx / Hs, i = mine + idx - u * Hs;
            store_granule(&g[u * H + i], epoch, h[u * hs + i]);
        }
    bool gave_up = false;
    const int Ho = H - Hs;
    for (int idx = tid; idx < U * Ho; idx += NT) {
        const int u = idx / Ho, ii = idx - u * Ho, i = ii < mine ? ii : ii + Hs;
        h[u * hs + i] = await_granule(&g[u * H + i], epoch, X, gave_up);
    }
    if (__syncthreads_or(gave_up)) X.dead = true;
}
// the third hop of the encoder: the next input rows of the utterances this workgroup owns go out, the others come in
// (granules [u][in]; owner of utterance u: workgroup u * (n / U) of the group)
template <int U>
__device__ __forceinline__ bool owns_g(const SplitCtx& X, int u) { return X.half == u * (X.n / U); }
template <int U>
__device__ __forceinline__ void hop_x_g(float (*x)[MAX_IN], int Cc, SplitCtx& X, unsigned long long* g, int tid) {
    const unsigned epoch = ++X.epoch;
    // (all stores first, in a loop of their own: store and spin lanes of one wave in two arms of a branch would let the
    //  spinning arm run first and wait for a partner that is waiting for this wave's stores)
    if (!X.withhold)
        for (int idx = tid; idx < U * Cc; idx += NT) {
            const int u = idx / Cc, c = idx - u * Cc;
            if (owns_g<U>(X, u)) store_granule(&g[u * Cc + c], epoch, x[u][c]);
        }
    bool gave_up = false;
    for (int idx = tid; idx < U * Cc; idx += NT) {
        const int u = idx / Cc, c = idx - u * Cc;
        if (!owns_g<U>(X, u)) x[u][c] = await_granule(&g[u * Cc + c], epoch, X, gave_up);
    }
    if (__syncthreads_or(gave_up)) X.dead = true;
}

// one frame of Wavernn.forward for the U utterances of the group: L.x -> L.fo, states in L.h1 / L.h2
template <int U>
__device__ __forceinline__ void pred_step_g(const PredDev& P, GrpLds<U>& L, int tid, SplitCtx& X) {
    PSTAMP(0)
    gru_rows_g<U>(P.w1i, P.w1h, P.b1i, P.b1h, &L.x[0][0], MAX_IN, P.in, &L.h1[0][0], MAX_H1, P.h1, L, tid, X.n, X.half);
    PSTAMP(1)
    gru_gates_g<U>(P.in, &L.h1[0][0], MAX_H1, P.h1, L, tid, X.n, X.half);
    PSTAMP(2)
    hop_g<U>(&L.h1[0][0], MAX_H1, P.h1, X, X.g1, tid);
    PSTAMP(3)
    gru_rows_g<U>(P.w2i, P.w2h, P.b2i, P.b2h, &L.h1[0][0], MAX_H1, P.h1, &L.h2[0][0], MAX_H2, P.h2, L, tid, X.n, X.half);
    PSTAMP(4)
    gru_gates_g<U>(P.h1, &L.h2[0][0], MAX_H2, P.h2, L, tid, X.n, X.half);
    PSTAMP(5)
    hop_g<U>(&L.h2[0][0], MAX_H2, P.h2, X, X.g2, tid);
    PSTAMP(6)
    for (int idx = tid; idx < U * P.h2; idx += NT) {
        const int u = idx / P.h2, i = idx - u * P.h2;
        L.relu[u][i] = L.h2[u][i] > 0.0f ? L.h2[u][i] : 0.0f;
    }
    __syncthreads();
    const int Sf = (P.h2 % 8 == 0 && P.h2 >= 64) ? 8 : 1;
    const int lenf = P.h2 / Sf, per = P.fc * Sf;
    for (int idx = tid; idx < U * per; idx += NT) {
        const int u = idx / per, j = idx - u * per, o = j % P.fc, sg = j / P.fc;
        L.pf[u][sg][o] =
            chain1(P.fcw + (size_t)sg * lenf * P.fc, L.relu[u] + sg * lenf, lenf, P.fc, o, sg == 0 ? P.fcb[o] : 0.0f);
    }
    __syncthreads();
    for (int idx = tid; idx < U * P.fc; idx += NT) {
        const int u = idx / P.fc, o = idx - u * P.fc;
        float acc = L.pf[u][0][o];
        if (Sf == 8)
            acc = ((L.pf[u][0][o] + L.pf[u][1][o]) + (L.pf[u][2][o] + L.pf[u][3][o])) +
                  ((L.pf[u][4][o] + L.pf[u][5][o]) + (L.pf[u][6][o] + L.pf[u][7][o]));
        const float t = fpc_tanhf(acc);
        L.fo[u][o] = t + t;
    }
    __syncthreads();
    PSTAMP(7)
}

// the group's granules: the region of its U utterances ([b][2][h1 + h2], split_args) re-cut as [U][h1], [U][h2], [U][in]
__device__ __forceinline__ SplitCtx group_ctx(const SplitArgs& S, const PredDev& P, int grp, int half) {
    SplitCtx X;
    X.n = S.n;
    X.half = half;
    X.err = S.err;
    X.limit = S.limit;
    X.withhold = S.withhold != 0 && grp == 0 && half == S.n - 1;
    X.dead = (status_load(S.err) & FPC_ST_TIMEOUT) != 0u;
    X.g1 = S.g + (size_t)grp * S.U * 2 * (P.h1 + P.h2);
    X.g2 = X.g1 + (size_t)S.U * P.h1;
    X.g3 = X.g2 + (size_t)S.U * P.h2;  // [U][in] (in <= h1)
    return X;
}

template <int U>
__global__ __launch_bounds__(NT) void k_forward_g(const PredDev P, const float* __restrict__ x, int Lf, float* h1, float* h2,
                                                  float* __restrict__ y, const SplitArgs S) {
    __shared__ GrpLds<U> L;
    const int grp = blockIdx.x / S.n, half = blockIdx.x % S.n, tid = threadIdx.x, b0 = grp * U;
    SplitCtx X = group_ctx(S, P, grp, half);
    const bool writer = half == 0;
    for (int idx = tid; idx < U * P.h1; idx += NT) L.h1[idx / P.h1][idx % P.h1] = h1[(size_t)b0 * P.h1 + idx];
    for (int idx = tid; idx < U * P.h2; idx += NT) L.h2[idx / P.h2][idx % P.h2] = h2[(size_t)b0 * P.h2 + idx];
    __syncthreads();
#ifdef FPC_PRED_PROF
    if (tid == 0) {
        for (int i = 0; i < 12; ++i) L.pprof[i] = 0;
        L.plast = __builtin_readcyclecounter();
    }
#endif
    int t = 0;
    for (; t < Lf; ++t) {
        for (int idx = tid; idx < U * P.in; idx += NT) {
            const int u = idx / P.in, c = idx - u * P.in;
            L.x[u][c] = x[((size_t)(b0 + u) * Lf + t) * P.in + c];
        }
        __syncthreads();
        pred_step_g<U>(P, L, tid, X);
        if (X.dead) break;
        if (writer)
            for (int idx = tid; idx < U * P.fc; idx += NT) {
                const int u = idx / P.fc, o = idx - u * P.fc;
                y[((size_t)(b0 + u) * Lf + t) * P.fc + o] = L.fo[u][o];
            }
    }
    __syncthreads();
#ifdef FPC_PRED_PROF
    if (tid == 0 && blockIdx.x == gridDim.x / 2)
        for (int i = 0; i < 8; ++i) S.err[1 + i] = (unsigned)(L.pprof[i] / (Lf > 0 ? Lf : 1));
#endif
    if (X.dead) {  // fail loudly (k_forward)
        if (writer) {
            const float qnan = __uint_as_float(0x7fc00000u);
            for (int u = 0; u < U; ++u)
                for (size_t k = (size_t)t * P.fc + tid; k < (size_t)Lf * P.fc; k += NT)
                    y[(size_t)(b0 + u) * Lf * P.fc + k] = qnan;
            for (int idx = tid; idx < U * P.h1; idx += NT) h1[(size_t)b0 * P.h1 + idx] = qnan;
            for (int idx = tid; idx < U * P.h2; idx += NT) h2[(size_t)b0 * P.h2 + idx] = qnan;
        }
        return;
    }
    if (writer) {
        for (int idx = tid; idx < U * P.h1; idx += NT) h1[(size_t)b0 * P.h1 + idx] = L.h1[idx / P.h1][idx % P.h1];
        for (int idx = tid; idx < U * P.h2; idx += NT) h2[(size_t)b0 * P.h2 + idx] = L.h2[idx / P.h2][idx % P.h2];
    }
}

template <int U>
__global__ __launch_bounds__(NT) void k_encode_g(const PredDev P, const CbDev C, const EncArgs A, const SplitArgs S) {
    __shared__ GrpLds<U> L;
    const int grp = blockIdx.x / S.n, half = blockIdx.x % S.n, tid = threadIdx.x, b0 = grp * U;
    SplitCtx X = group_ctx(S, P, grp, half);
    const int step = S.n / U;  // (n >= U: every utterance of the group has a workgroup of its own for the searches)
    const int myu = half % step == 0 ? half / step : -1;
    for (int idx = tid; idx < U * MAX_H1; idx += NT) (&L.h1[0][0])[idx] = 0.0f;
    for (int idx = tid; idx < U * MAX_H2; idx += NT) (&L.h2[0][0])[idx] = 0.0f;
    for (int idx = tid; idx < U * MAX_IN; idx += NT) (&L.x[0][0])[idx] = 0.0f;
    __syncthreads();
    int i = 0;
    for (; i < A.Lf; ++i) {
        pred_step_g<U>(P, L, tid, X);
        if (X.dead) break;
        // residual, searches and outputs of an utterance on its owner only; the next input rows change hands
        if (myu >= 0) encode_frame(L, L.fo[myu], L.x[myu], P, C, A, S.err, (size_t)(b0 + myu) * A.Lf + i, true, tid);
        hop_x_g<U>(L.x, P.in, X, X.g3, tid);
        if (X.dead) {  // (the frame's outputs of the owned utterances are written; poison starts with the next one)
            ++i;
            break;
        }
    }
    if (X.dead && myu >= 0) encode_poison(P, A, b0 + myu, i, tid);
}

// the receiver's frame tail for one utterance: threads (c < Cc) of one wave (k_decode_feat)
__device__ __forceinline__ void decode_frame(const float* fo, float* xn, const PredDev& P, const CbDev& C,
                                             const float* __restrict__ pitch, const int* __restrict__ idx,
                                             float* __restrict__ c_out, int* bad, size_t fi, bool store, int c) {
    const int Cc = P.in, F = P.fc;
    if (c < F) {
        const int* ix = idx + fi * 4;
        float rq = 0.0f;
        if (c == 0) {
            const int k = ix[0];
            if (k >= 0) {
                if (k < C.n_hi)
                    rq = (float)C.scl_hi[k];
                else if (C.scl_lo && k - C.n_hi < C.n_lo)
                    rq = (float)C.scl_lo[k - C.n_hi];
                else if (store)
                    atomicOr(bad, 1);
            }
        } else {
            const int d = c - 1, k1 = ix[1], k2 = ix[2], k3 = ix[3];
            if (k1 >= 0) {
                if (k1 >= C.N_hi0 || (C.S_hi == 2 && (k2 < 0 || k2 >= C.N_hi1))) {
                    if (store) atomicOr(bad, 1);
                } else {
                    const double e0 = C.vq_hi0_r[(size_t)k1 * NDIM + d];
                    rq = (float)(C.S_hi == 2 ? e0 + C.vq_hi1_r[(size_t)k2 * NDIM + d] : e0);
                }
            } else if (k3 >= 0) {
                if (!C.vq_lo_r || k3 >= C.N_lo) {
                    if (store) atomicOr(bad, 1);
                } else {
                    rq = (float)C.vq_lo_r[(size_t)k3 * NDIM + d];
                }
            }
        }
        const float cn = fo[c] + rq;
        if (store) c_out[fi * Cc + c] = cn;
        xn[c] = cn;
    } else if (c < Cc) {
        const float v = pitch[fi * (Cc - F) + (c - F)];
        if (store) c_out[fi * Cc + c] = v;
        xn[c] = v;
    }
}

template <int U>
__global__ __launch_bounds__(NT) void k_decode_feat_g(const PredDev P, const CbDev C, const float* __restrict__ pitch,
                                                      const int* __restrict__ idx, int Lf, float* __restrict__ c_out,
                                                      int* bad, const SplitArgs S) {
    __shared__ GrpLds<U> L;
    const int grp = blockIdx.x / S.n, half = blockIdx.x % S.n, tid = threadIdx.x, b0 = grp * U;
    SplitCtx X = group_ctx(S, P, grp, half);
    const bool writer = half == 0;
    for (int idx2 = tid; idx2 < U * MAX_H1; idx2 += NT) (&L.h1[0][0])[idx2] = 0.0f;
    for (int idx2 = tid; idx2 < U * MAX_H2; idx2 += NT) (&L.h2[0][0])[idx2] = 0.0f;
    for (int idx2 = tid; idx2 < U * MAX_IN; idx2 += NT) (&L.x[0][0])[idx2] = 0.0f;
    __syncthreads();
    int i = 0;
    for (; i < Lf; ++i) {
        pred_step_g<U>(P, L, tid, X);
        if (X.dead) break;
        // every workgroup looks the residuals of all U utterances up itself (a wave per utterance): no third hop
        if (tid < U * 64) {
            const int u = tid >> 6;
            decode_frame(L.fo[u], L.x[u], P, C, pitch, idx, c_out, bad, (size_t)(b0 + u) * Lf + i, writer, tid & 63);
        }
        __syncthreads();
    }
    if (X.dead && writer) {
        const float qnan = __uint_as_float(0x7fc00000u);
        for (int u = 0; u < U; ++u)
            for (size_t k = (size_t)i * P.in + tid; k < (size_t)Lf * P.in; k += NT)
                c_out[(size_t)(b0 + u) * Lf * P.in + k] = qnan;
    }
}
