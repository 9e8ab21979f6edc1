// predictor_bwd_ws.h -- back-propagation through time of the training step with the TRANSPOSED weights stationary on chip
// and the batch on the matrix cores (included by predictor.hip behind k_train_bwd; the forward's counterpart: predictor_ws.h).
//
// k_train_bwd gives every utterance its own workgroups and streams the three transposed matrices (W2i, W2h, W1h: 2.4 MB) from
// L2 once per frame and utterance; at the reference's batch (100 x 150, train_frame.py:198-204) it was more than half of the
// step.  Here a GROUP of 16 utterances (the M dimension of one v_mfma_f32_16x16x4_f32 tile) runs on the 32 workgroups of one
// XCD, exactly as in the forward kernels, and workgroup s owns for the whole launch
//   GRU1 units 12 s .. 12 s + 11: their gate gradients, and the 12 COLUMNS of W1h^T / W2i^T that produce dh1 of those units,
//   GRU2 units  4 s ..  4 s + 3 : their gate gradients, and the  4 columns of W2h^T that produce dh2 of those units,
// as MFMA B operands in REGISTERS (wave w < 4: rows 288 w .. 288 w + 287 of W1h, 72 k-steps, the last 24 of them in LDS;
// wave 4 + w: rows 96 w .. 96 w + 95 of W2i and of W2h, 24 + 24 k-steps).  The A operands are the gate-gradient vectors of ALL
// units of the 16 utterances, as images in LDS ([row][utterance]: lane l of k-step j reads image[64 j + l]): they change hands
// as 16-byte granules {epoch, 3 values} -- one per (unit, utterance) -- through the group's granule block, two sets used by
// step parity (a workgroup publishes step k + 1 of a recurrence once its gather of step k is complete, i.e. once every partner
// has published step k, which a partner does only with step k - 1 whole in its LDS: the set of parity k + 1 is free).
//
// TWO TRACKS, because the two recurrences are almost independent of each other going backward -- dh2(t) needs dpre(t) and
// dh2n only; dh1(t) needs W2i^T g2i(t) and dh1n -- and each is a strictly serial loop of its own
//   publish the gate gradients -> hop (one L2 round trip) -> transposed product on the matrix cores -> gate gradients:
//   track B, waves 4-7, step k = 0 .. L-1, GRU2 at frame t = L-1-k:
//     gates (wave 4): dh2n = fma(dh2(t+1), z2(t+1), tree(W2h^T g2h(t+1)));  dh2 = relu'(h2) * (fc_w^T dpre) + dh2n;
//            gate gradients -> dgi2 / dgh2 (kept for the weight gradients), published;
//     hop:   the other 31 workgroups' GRU2 granules into the image g2;
//     MFMA:  W2i^T g2i(t) -> pa2[k & 1] (track A reads it one step later), W2h^T g2h(t) -> pb;
//   track A, waves 0-3, step k = 1 .. L, GRU1 at frame t = L-k:
//     hop:   the GRU1 granules of frame t + 1 (published at the end of step k - 1) into the image g1;
//     MFMA:  W1h^T g1h(t+1) -> pa1, four chains of 72;
//     gates (waves 0-2): dh1n = fma(dh1(t+1), z1(t+1), tree(pa1));  dh1 = tree(W2i^T g2i(t)) + dh1n;  -> dgi1 / dgh1, published.
// The tracks meet in two LDS counters only (track A waits for the pa2 of its frame; track B does not overwrite a pa2 buffer
// before track A has read it), the waves of a track in two more -- no workgroup barrier inside the loop, so the round trip of
// one recurrence hides behind the products of the other.  (One track, one hop and one barrier pair per step -- the first form
// of this kernel -- took 17.6 kcycles per step, 4 k of them waiting for granules and 2.4 k at the barrier:
// profiles/r05_ablations.txt item 10.)  Products in the row segments of oracle/fpc_oracle.c (matvec_t: 4 segments, each a
// row-ordered fmaf chain from 0 = what the f32 MFMA accumulates, the segment sums added as a balanced tree by the gate threads).
// Every value is formed by the operations and in the order of k_train_bwd and of orc_train_step (gru_bwd): losses, gradients
// and parameters are bit-identical to both (tests).  The saved activations a gate thread needs are fetched a step ahead.
// Residency, give-up and fallback: as the forward kernels (ws_hello decides GO / FALLBACK per group before anything is
// written; k_train_bwd, one workgroup per utterance, serves the groups that fell back).
// Reference: train_frame.py:53-120 (loss.backward() of the teacher-forced step).

#ifndef FPC_BW_STAMP_TID
#define FPC_BW_STAMP_TID 0
#endif
constexpr int BQ1 = WV1;                       // granules per workgroup, GRU1 items (unit, utterance): {epoch, drpre, dzpre, dnpre * r}
constexpr int BQ2 = WV2;                       // ... GRU2 items: {epoch, drpre, dzpre, dnpre} (the receiver multiplies by r itself)
constexpr int BQ = BQ1 + BQ2;                  // 256
constexpr int BGRANULES = WNS + 2 * WNS * BQ;  // hello | set 0 | set 1: 16 416 granules = 262 656 bytes per group
constexpr int BTN = NT / 2;                    // threads of a track
static_assert(WNS * WG == 2 * BTN, "gather assignment: thread p of a track = utterance p % 16 of the source workgroups p / 16 and p / 16 + 16");
enum { BSIG_A = 0, BSIG_BM, BSIG_B3, BSIG_AD, BSIG_AP, BSIG_BP, BSIG_N };

struct __attribute__((aligned(16))) BwLds {
    float g1[3 * WH1 * WG];  // image of g1h(t+1): [gate * 384 + unit][utterance]
    float g2[4 * WH2 * WG];  // image of g2(t): rows gate * 128 + unit (gate < 3) = g2i; rows 384 + unit = dnpre * r (g2h's n rows)
    float pa1[4][256];       // W1h^T g1h: [row segment][ws_tile(column, utterance)]
    float pa2[2][4][256];    // W2i^T g2i, by step parity
    float pb[4][256];        // W2h^T g2h (columns < 4)
    float w1[4][24 * 64];    // waves 0-3 keep k-steps 0 .. 47 of their 72 in registers; 48 .. 71 here: [wave][k-step - 48][lane]
    float r2[WH2 * WG];      // r2 of every GRU2 unit at track B's frame [unit][utterance] (the gather forms dnpre * r)
    float dp[WG * WIN];      // dpre of track B's frame [utterance][output]
    float fcw[WU2][WIN];     // rows 4 s .. 4 s + 3 of the output layer's weights ([H2][F] in the device layout)
    int sig[BSIG_N];         // wave counters: the waves of a track among themselves (A, BM, B3), and the two tracks (B3, AD)
    int dead, dead_latch, hello, same_xcd;
};

__device__ __forceinline__ float bw_tree(const float (&p)[4][256], int v) { return (p[0][v] + p[1][v]) + (p[2][v] + p[3][v]); }

// N granules per lane, granule i at g0 + (i / SPLIT) * 16 BQ + (i % SPLIT) * 16 (source workgroups s and s + 16, SPLIT items of
// each), polled until every wanted tag of the WAVE shows `epoch`; `own`: the source (0 / 1) that is this workgroup itself (its
// items are in the image already), -1: neither.  false: the wait was given up, or the workgroup is dead already.
// "this wave has published": a hint for the waves of the track that have no gate items -- a wave that polls the granule block
// before the partners can have published only re-issues its loads, and those loads queue in front of the publishing wave's
// stores in the compute unit's one vector-memory pipeline (measured: six stores of the gate wave took 2.6 kcycles to issue
// beside three polling waves).  Relaxed: it orders nothing (a release would wait for the store's acknowledgement).
__device__ __forceinline__ void bw_hint(int* s) {
    if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void bw_await_hint(int* s, int target, int* dead) {
    unsigned spins = 0;
    while (__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if ((++spins & 7u) == 0 && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int N, int SPLIT>
__device__ __forceinline__ bool bw_poll(const WsCtx& X, BwLds& L, int g0, unsigned epoch, int own, u32x4 (&v)[N]) {
    if (ws_dead(L)) return false;
    unsigned spins = 0;
    unsigned long long tm0 = 0, last = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i)
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, (g0 + (i / SPLIT) * 16 * BQ + (i % SPLIT) * 16) * 16, 0, 16);
#pragma unroll
        for (int i = 0; i < N; ++i) all &= i / SPLIT == own || v[i].x == epoch;
        if (__all(all)) return true;
        if (ws_dead(L)) return false;
        if ((++spins & 63u) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (tm0 == 0 || now - last > spin_rearm_gap(X.limit)) tm0 = now;  // (this wave was descheduled: await_granule)
            last = now;
            if (now - tm0 > X.limit || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                ws_give_up(X, L);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");  // (the loads are re-issued every round)
    }
}

__global__ __launch_bounds__(NT) void k_train_bwd_ws(const PredDev P, const BwdW W, int Lf, const TrainBufs T, const WsArgs S) {
    __shared__ BwLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
#ifdef FPC_WS_PROF
    const long long t_entry = __builtin_readcyclecounter();
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice, BGRANULES);
    if (tid == 0) L.dead = (status_load(S.err) & FPC_ST_TIMEOUT) != 0u ? 1 : 0;  // (a failed handle waits for nobody)
    if (tid < BSIG_N) L.sig[tid] = 0;
    // ---- this wave's B operands: rows of the torch-layout matrices (BwdW), this workgroup's columns ----
    float wB[48];
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 48; ++j) wB[j] = c < WU1 ? W.w1h[(size_t)(288 * wave + 4 * j + q) * WH1 + WU1 * slice + c] : 0.0f;
        for (int j = 48; j < 72; ++j)
            L.w1[wave][(j - 48) * 64 + lane] = c < WU1 ? W.w1h[(size_t)(288 * wave + 4 * j + q) * WH1 + WU1 * slice + c] : 0.0f;
    } else {
        const int sg = wave - 4;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            wB[j] = c < WU1 ? W.w2i[(size_t)(96 * sg + 4 * j + q) * WH1 + WU1 * slice + c] : 0.0f;
            wB[24 + j] = c < WU2 ? W.w2h[(size_t)(96 * sg + 4 * j + q) * WH2 + WU2 * slice + c] : 0.0f;
        }
    }
    if (tid < WU2 * WFC) L.fcw[tid / WFC][tid % WFC] = P.fcw[(size_t)(WU2 * slice + tid / WFC) * WFC + tid % WFC];
    __syncthreads();
    ws_hello(X, L, S, tid);
    if (X.fallback) return;  // (group-uniform, nothing written yet) k_train_bwd behind this launch serves the group
    // ---- the thread within its track: gate item (unit ij of the workgroup's, utterance iu); gather (sources gs / gs + 16, utterance gu) ----
    const bool trackA = wave < 4;
    const int p = tid & (BTN - 1);
    const int ij = p >> 4, iu = p & 15, tv = ws_tile(ij, iu);
    const bool gate = trackA ? p < BQ1 : p < BQ2;                    // (waves 0-2; wave 4)
    const bool live = gate && iu < X.nu;                             // (a part-filled group: the missing utterances are zeros)
    const int unit = trackA ? WU1 * slice + ij : WU2 * slice + ij, H = trackA ? WH1 : WH2;
    const size_t nrow = (size_t)(X.b0 + (iu < X.nu ? iu : 0)) * Lf;  // sample index of frame 0 of the thread's utterance
    const int gs = p >> 4, gu = p & 15;
    // saved activations of the thread's item at frame t (0 outside the batch / the sequence)
    struct Act {
        float r, z, n, hn, hp, h;
    };
    auto load_act = [&](int t) {
        Act a{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live && t >= 0 && t < Lf) {
            const size_t o = (nrow + t) * H + unit;
            if (trackA) {
                a.r = T.r1[o], a.z = T.z1[o], a.n = T.n1[o], a.hn = T.hn1[o], a.hp = T.h1p[o];
            } else {
                a.r = T.r2[o], a.z = T.z2[o], a.n = T.n2[o], a.hn = T.hn2[o], a.hp = T.h2p[o], a.h = T.h2[o];
            }
        }
        return a;
    };
#ifdef FPC_WS_PROF
    long long bprof[6] = {0, 0, 0, 0, 0, 0}, blast = __builtin_readcyclecounter();
    const long long t_loop = blast;
#define BSTAMPW(i)                                           \
    if (tid == FPC_BW_STAMP_TID) {                           \
        const long long now_ = __builtin_readcyclecounter(); \
        bprof[i] += now_ - blast;                            \
        blast = now_;                                        \
    }
#else
#define BSTAMPW(i)
#endif
#ifdef FPC_BW_PROF_A2  // (the stamps of track A inside its hop instead: hint wait | poll 1 | writes 1 | poll 2 | writes 2 | the rest)
#define BSTAMPA(i)
#define BSTAMPE(i) BSTAMPW(i)
#else
#define BSTAMPA(i) BSTAMPW(i)
#define BSTAMPE(i)
#endif
#ifdef FPC_BW_PROF_B2  // (the stamps of track B inside its gates instead: pb tree | dot product | gate math + publication | kept stores | prefetch issue | the rest)
#define BSTAMPB(i)
#define BSTAMPD(i) BSTAMPW(i)
#else
#define BSTAMPB(i) BSTAMPW(i)
#define BSTAMPD(i)
#endif
    // the gate gradients of one item from dh and its saved activations: {drpre, dzpre, dnpre, dnpre * r}
    auto gate_grads = [&](const Act& a, float dh) {
        const float dn_raw = dh * (1.0f - a.z);
        const float dnpre = dn_raw * fmaf(-a.n, a.n, 1.0f);
        const float dz_raw = dh * (a.hp - a.n);
        const float dzpre = dz_raw * (a.z * (1.0f - a.z));
        const float drpre = (dnpre * a.hn) * (a.r * (1.0f - a.r));
        return f32x4ws{drpre, dzpre, dnpre, dnpre * a.r};
    };
    // ... kept for the weight gradients (k_grad_tn); behind the publication: one in-order memory queue per wave
    auto keep = [&](const f32x4ws& g, int t) {
        if (live) {
            float* gi = trackA ? T.dgi1 : T.dgi2;
            float* gh = trackA ? T.dgh1 : T.dgh2;
            const size_t o = (nrow + t) * 3 * H + unit;
            gi[o] = g[0], gi[o + H] = g[1], gi[o + 2 * H] = g[2];
            gh[o] = g[0], gh[o + H] = g[1], gh[o + 2 * H] = g[3];
        }
    };
    float dhp = 0.0f, zp = 0.0f;  // dh and z of the thread's item at the frame of the previous step
    if (trackA) {
        // =============================== track A: GRU1, frame t = L - k in step k = 1 .. L ===============================
        int nsync = 0;
        auto sync = [&]() {
            ++nsync;
            df_signal(&L.sig[BSIG_A]);
            return df_wait(&L.sig[BSIG_A], 4 * nsync, &L.dead);
        };
        Act an = load_act(Lf - 1);
        for (int k = 1; k <= Lf; ++k) {
            const int t = Lf - k;
            const Act a = an;
            if (k >= 2) {
                // ---- the hop: GRU1 items of frame t + 1 (published in step k - 1: epoch k, set (k - 1) & 1), 12 per source ----
                const int pset = WNS + ((k - 1) & 1) * WNS * BQ;
                bool ok = true;
                bw_await_hint(&L.sig[BSIG_AP], 3 * (k - 1), &L.dead);  // (this workgroup's gate waves have published step k - 1)
                BSTAMPE(0)
#pragma unroll 1
                for (int hf = 0; hf < 2 && ok; ++hf) {
                    const int s = gs + 16 * hf;
                    u32x4 v[WU1];
                    ok = bw_poll<WU1, WU1>(X, L, pset + s * BQ + gu, (unsigned)k, s == slice ? 0 : -1, v);
                    if (hf == 0) {
                        BSTAMPE(1)
                    } else {
                        BSTAMPE(3)
                    }
                    if (ok && s != slice) {
                        float* img = L.g1 + (WU1 * s) * WG + gu;
#pragma unroll
                        for (int j = 0; j < WU1; ++j) {
                            img[(0 * WH1 + j) * WG] = __uint_as_float(v[j].y);
                            img[(1 * WH1 + j) * WG] = __uint_as_float(v[j].z);
                            img[(2 * WH1 + j) * WG] = __uint_as_float(v[j].w);
                        }
                    }
                    if (hf == 0) {
                        BSTAMPE(2)
                    } else {
                        BSTAMPE(4)
                    }
                }
                BSTAMPA(0)
                if (!ok || !sync()) break;  // the image whole
                BSTAMPA(1)
                an = load_act(t - 1);  // (consumed in the next step; issued here, where the vector-memory pipeline is idle)
                // ---- W1h^T g1h(t + 1), rows 288 wave ..: one chain of 72 dependent MFMAs (the canonical order is a chain) ----
                f32x4ws acc = {0.f, 0.f, 0.f, 0.f};
                const float* img = L.g1 + 288 * wave * WG + lane;
                const float* wl = L.w1[wave] + lane;
                // (the scheduler interleaves these reads with the chain, eight at a time; fencing all of them in front of the chain
                //  with scheduling barriers was measured slower: step 2.91 against 2.81 ms, same box)
                float av[3][24], bl[24];
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                    for (int j = 0; j < 24; ++j) av[ch][j] = img[64 * (24 * ch + j)];
#pragma unroll
                for (int j = 0; j < 24; ++j) {
                    av[2][j] = img[64 * (48 + j)];
                    bl[j] = wl[64 * j];
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int j = 0; j < 24; ++j) acc = ws_mfma(av[ch][j], ch < 2 ? wB[24 * ch + j] : bl[j], acc);
                ws_put(L.pa1[wave], lane, acc);
                BSTAMPA(2)
            }
            if (k < 2) an = load_act(t - 1);
            if (!sync()) break;  // pa1 whole (and nobody reads the image any more)
            BSTAMPA(3)
            // ---- the gates (waves 0-2): W2i^T g2i(t) is track B's product of its step k - 1 ----
            if (wave < 3) {
                if (!df_wait(&L.sig[BSIG_B3], 4 * k, &L.dead)) break;
                BSTAMPA(4)
                const float dhn = k >= 2 ? fmaf(dhp, zp, bw_tree(L.pa1, tv)) : 0.0f;  // dh1n(t): gru_bwd's last line for frame t + 1
                const float dh = bw_tree(L.pa2[(k - 1) & 1], tv) + dhn;
                df_signal(&L.sig[BSIG_AD]);  // (this wave has read the pa2 buffer: its lanes' values are in registers)
                const f32x4ws g = gate_grads(a, dh);
                dhp = dh;
                zp = a.z;
                if (k < Lf) {  // (the last step -- frame 0 -- feeds no product)
                    ws_store(X, WNS + (k & 1) * WNS * BQ + slice * BQ + p, (unsigned)k + 1u, g[0], g[1], g[3]);
                    bw_hint(&L.sig[BSIG_AP]);
                    L.g1[(0 * WH1 + unit) * WG + iu] = g[0];
                    L.g1[(1 * WH1 + unit) * WG + iu] = g[1];
                    L.g1[(2 * WH1 + unit) * WG + iu] = g[3];
                }
                keep(g, t);
                BSTAMPA(5)
            }
            BSTAMPE(5)
        }
    } else {
        // =============================== track B: GRU2, frame t = L - 1 - k in step k = 0 .. L - 1 ===============================
        const int sg = wave - 4;
        int nmid = 0;
        // dpre and r2 of the step's frame go through LDS (fetched a step ahead, stored behind the gather):
        // thread p: dpre values p and (p < 32) 256 + p of the 288; r2 of units 8 (p / 16) .. + 7 of utterance p % 16
        float dpv[2];
        f32x4ws r2v[2];
        auto load_stage = [&](int t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = p + 256 * i, du = e / WFC, dout = e - du * WFC;
                dpv[i] = (e < WG * WFC && du < X.nu && t >= 0) ? T.dpre[((size_t)(X.b0 + du) * Lf + t) * WFC + dout] : 0.0f;
            }
            const f32x4ws zero = {0.f, 0.f, 0.f, 0.f};
            const bool in = gu < X.nu && t >= 0;
            const f32x4ws* src = reinterpret_cast<const f32x4ws*>(T.r2 + ((size_t)(X.b0 + (in ? gu : 0)) * Lf + (in ? t : 0)) * WH2 + 8 * gs);
            r2v[0] = in ? src[0] : zero;
            r2v[1] = in ? src[1] : zero;
        };
        auto stage = [&]() {
            L.dp[p] = dpv[0];
            if (p < WG * WFC - 256) L.dp[256 + p] = dpv[1];
#pragma unroll
            for (int i = 0; i < 8; ++i) L.r2[(8 * gs + i) * WG + gu] = r2v[i >> 2][i & 3];
        };
        Act an = load_act(Lf - 1);
        load_stage(Lf - 1);
        stage();
        df_signal(&L.sig[BSIG_BM]);
        ++nmid;
        (void)df_wait(&L.sig[BSIG_BM], 4 * nmid, &L.dead);
        for (int k = 0; k < Lf; ++k) {
            const int t = Lf - 1 - k;
            const unsigned epoch = (unsigned)k + 1u;
            const int set = WNS + (k & 1) * WNS * BQ;
            const Act a = an;
            // ---- the gates (wave 4) ----
            if (sg == 0) {
                const float dhn = k >= 1 ? fmaf(dhp, zp, bw_tree(L.pb, tv)) : 0.0f;
                BSTAMPD(0)
                float dr = 0.0f;
#pragma unroll
                for (int o = 0; o < WFC; ++o) dr = fmaf(L.fcw[ij][o], L.dp[iu * WFC + o], dr);
                const float dh = (a.h > 0.0f ? dr : 0.0f) + dhn;
                BSTAMPD(1)
                const f32x4ws g = gate_grads(a, dh);
                dhp = dh;
                zp = a.z;
                ws_store(X, set + slice * BQ + BQ1 + p, epoch, g[0], g[1], g[2]);
                bw_hint(&L.sig[BSIG_BP]);
                L.g2[(0 * WH2 + unit) * WG + iu] = g[0];
                L.g2[(1 * WH2 + unit) * WG + iu] = g[1];
                L.g2[(2 * WH2 + unit) * WG + iu] = g[2];
                L.g2[(3 * WH2 + unit) * WG + iu] = g[3];
                BSTAMPD(2)
                keep(g, t);
                BSTAMPD(3)
            }
            an = load_act(t - 1);  // (behind the publication; consumed a step later)
            load_stage(t - 1);
            BSTAMPD(4)
            BSTAMPB(0)
            // ---- the hop: GRU2 items of this frame, 4 of each of the sources gs and gs + 16 ----
            {
                u32x4 v[2 * WU2];
                const int own = gs == slice ? 0 : gs + 16 == slice ? 1 : -1;
                bw_await_hint(&L.sig[BSIG_BP], k + 1, &L.dead);  // (this workgroup's gate wave has published this step)
                if (!bw_poll<2 * WU2, WU2>(X, L, set + gs * BQ + BQ1 + gu, epoch, own, v)) break;
                BSTAMPB(1)
#pragma unroll
                for (int i = 0; i < 2 * WU2; ++i)
                    if (i / WU2 != own) {
                        const int u2 = WU2 * (gs + 16 * (i / WU2)) + i % WU2;
                        const float v2 = __uint_as_float(v[i].w);
                        L.g2[(0 * WH2 + u2) * WG + gu] = __uint_as_float(v[i].y);
                        L.g2[(1 * WH2 + u2) * WG + gu] = __uint_as_float(v[i].z);
                        L.g2[(2 * WH2 + u2) * WG + gu] = v2;
                        L.g2[(3 * WH2 + u2) * WG + gu] = v2 * L.r2[u2 * WG + gu];  // dgh[2 H + i] = dnpre * r[i] (gru_bwd)
                    }
            }
            ++nmid;
            df_signal(&L.sig[BSIG_BM]);
            if (!df_wait(&L.sig[BSIG_BM], 4 * nmid, &L.dead)) break;  // the image whole; dp and r2 of this frame read
            BSTAMPB(2)
            stage();
            // (the pa2 buffer of this parity: read by track A in its step k - 1)
            if (k >= 2 && !df_wait(&L.sig[BSIG_AD], 3 * (k - 1), &L.dead)) break;
            BSTAMPB(3)
            // ---- W2i^T g2i(t) and W2h^T g2h(t), rows 96 sg ..: two chains of 24, interleaved ----
            {
                f32x4ws ai = {0.f, 0.f, 0.f, 0.f}, ah = {0.f, 0.f, 0.f, 0.f};
                const float* img = L.g2 + 96 * sg * WG + lane;
                float av[24], bv[24];
#pragma unroll
                for (int j = 0; j < 24; ++j) {
                    av[j] = img[64 * j];
                    // (g2h = g2i in the r and z rows; its n rows 256 .. 383 sit 128 rows further down the image)
                    bv[j] = img[64 * j + ((96 * sg + 4 * j >= 2 * WH2) ? WH2 * WG : 0)];
                }
#pragma unroll
                for (int j = 0; j < 24; ++j) {
                    ai = ws_mfma(av[j], wB[j], ai);
                    ah = ws_mfma(bv[j], wB[24 + j], ah);
                }
                ws_put(L.pa2[k & 1][sg], lane, ai);
                ws_put(L.pb[sg], lane, ah);
            }
            BSTAMPB(4)
            df_signal(&L.sig[BSIG_B3]);
            if (!df_wait(&L.sig[BSIG_B3], 4 * (k + 1), &L.dead)) break;  // pb, dp and r2 whole for the next step's gates and gather
            BSTAMPB(5)
            BSTAMPD(5)
        }
    }
#ifdef FPC_WS_PROF
    if (tid == FPC_BW_STAMP_TID && blockIdx.x == 8 * 5)
    {
        for (int i = 0; i < 6; ++i) S.err[44 + i] = (unsigned)(bprof[i] / (Lf > 0 ? Lf : 1));
        S.err[50] = (unsigned)((t_loop - t_entry) / 1000);                                   // kilocycles before the loop
        S.err[51] = (unsigned)((__builtin_readcyclecounter() - t_loop) / 1000);             // ... in the loop
        S.err[52] = (unsigned)((__builtin_amdgcn_s_memrealtime() - rt_entry) / 100);        // microseconds in the kernel
    }
#endif
    // (a launch that gave up leaves its status bit: the step's Adam update is skipped, the host reports FPC_ERR_TIMEOUT)
}
