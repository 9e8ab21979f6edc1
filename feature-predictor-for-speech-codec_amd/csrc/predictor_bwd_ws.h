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
// wave 4 + w: rows 96 w .. 96 w + 95 of W2i and of W2h, 24 + 24 k-steps).  The A operands are the gate-gradient vectors of ALL units of the 16 utterances, as images
// in LDS ([row][utterance]: lane l of k-step j reads image[64 j + l]): they change hands once per step as 16-byte granules
// {epoch, 3 values} -- one per (unit, utterance) -- through the group's granule block, two sets used by step parity (a
// workgroup publishes step k + 1 once its gather of step k is complete, i.e. once every partner has published step k, which a
// partner does only with step k - 1 whole in its LDS: the set of parity k + 1 is free).
//
// ONE hop per step, because the two recurrences are independent of each other going backward -- dh2(t) needs dpre(t) and
// dh2n only, dh1(t) needs W2i^T g2i(t) and dh1n -- so step k works on GRU2 at frame t = L - 1 - k and on GRU1 at frame t + 1:
//   phase 1 (gate threads, registers): dh2n(t) = fma(dh2(t+1), z2(t+1), tree(W2h^T g2h(t+1)));  dh2 = relu'(h2) * (fc_w^T dpre) + dh2n;
//           gate gradients of GRU2 at t -> dgi2 / dgh2 (kept for the weight gradients), published;
//           dh1n(t+1) = fma(dh1(t+2), z1(t+2), tree(W1h^T g1h(t+2)));  dh1 = tree(W2i^T g2i(t+1)) + dh1n;
//           gate gradients of GRU1 at t + 1 -> dgi1 / dgh1, published;
//   hop: every workgroup gathers the other 31 workgroups' granules into its images;
//   phase 2 (all eight waves, MFMA): W2i^T g2i(t), W2h^T g2h(t), W1h^T g1h(t+1) for this workgroup's columns, in the row
//           segments of oracle/fpc_oracle.c (matvec_t: 4 segments, each a row-ordered fmaf chain from 0 = what the f32 MFMA
//           accumulates, the segment sums added as a balanced tree by the gate threads in the next phase 1).
// Every value is formed by the operations and in the order of k_train_bwd and of orc_train_step (gru_bwd): losses, gradients
// and parameters are bit-identical to both (tests).  The saved activations a gate thread needs are fetched a step ahead.
// Residency, give-up and fallback: as the forward kernels (ws_hello decides GO / FALLBACK per group before anything is
// written; k_train_bwd, one workgroup per utterance, serves the groups that fell back).
// Reference: train_frame.py:53-120 (loss.backward() of the teacher-forced step).

#ifndef FPC_BW_STAMP_TID
#define FPC_BW_STAMP_TID 0
#endif
constexpr int BQ1 = WV1;                     // granules per workgroup, GRU1 items (unit, utterance): {epoch, drpre, dzpre, dnpre * r}
constexpr int BQ2 = WV2;                     // ... GRU2 items: {epoch, drpre, dzpre, dnpre} (the receiver multiplies by r itself)
constexpr int BQ = BQ1 + BQ2;                // 256
constexpr int BGRANULES = WNS + 2 * WNS * BQ;  // hello | set 0 | set 1: 16 416 granules = 262 656 bytes per group
constexpr int BNG = WNS * BQ / NT;           // granules a thread gathers per step: 16 (one per pair of source workgroups)
static_assert(WNS * BQ % NT == 0 && NT == 2 * BQ, "gather assignment: thread p = item p % 256 of workgroups 2 i + p / 256");

struct __attribute__((aligned(16))) BwLds {
    float g1[3 * WH1 * WG];  // image of g1h(t+1): [gate * 384 + unit][utterance]
    float g2[4 * WH2 * WG];  // image of g2(t): rows gate * 128 + unit (gate < 3) = g2i; rows 384 + unit = dnpre * r (g2h's n rows)
    float pa1[4][256];       // W1h^T g1h: [row segment][ws_tile(column, utterance)]
    float pa2[4][256];       // W2i^T g2i
    float pb[4][256];        // W2h^T g2h (columns < 4)
    float w1[4][24 * 64];    // waves 0-3 keep k-steps 0 .. 47 of their 72 in registers; 48 .. 71 here: [wave][k-step - 48][lane]
    float r2[WH2 * WG];      // r2 of every GRU2 unit at the step's frame [unit][utterance] (the gather forms dnpre * r)
    float dp[WG][WIN];       // dpre of the step's frame [utterance][output]
    float fcw[WU2][WIN];     // rows 4 s .. 4 s + 3 of the output layer's weights ([H2][F] in the device layout)
    int dead, dead_latch, hello, same_xcd;
};

__device__ __forceinline__ float bw_tree(const float (&p)[4][256], int v) { return (p[0][v] + p[1][v]) + (p[2][v] + p[3][v]); }

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
    // ---- roles of phase 1: thread < 192: GRU1 item (unit 12 s + j, utterance u); 192 .. 255: GRU2 item (unit 4 s + j, u) ----
    const bool is1 = tid < BQ1, is2 = tid >= BQ1 && tid < BQ;
    const int it = is1 ? tid : tid - BQ1, ij = it >> 4, iu = it & 15;
    const bool live = (is1 || is2) && iu < X.nu;                     // (a part-filled group: the missing utterances are zeros)
    const int unit = is1 ? WU1 * slice + ij : WU2 * slice + ij, H = is1 ? WH1 : WH2;
    const size_t nrow = (size_t)(X.b0 + (iu < X.nu ? iu : 0)) * Lf;  // sample index of frame 0 of the thread's utterance
    const int tv = ws_tile(ij, iu);
    // ---- the gather's assignment: thread p takes item p % 256 of the source workgroups 2 i + p / 256, i < 16 ----
    const int gitem = tid & (BQ - 1), gs0 = tid >> 8;
    const bool g_is1 = gitem < BQ1;
    const int gj = g_is1 ? gitem >> 4 : (gitem - BQ1) >> 4, gu = gitem & 15;
    // saved activations of the thread's item at frame t (0 outside the batch / the sequence)
    struct Act {
        float r, z, n, hn, hp, h;
    };
    auto load_act = [&](int t) {
        Act a{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live && t >= 0 && t < Lf) {
            const size_t o = (nrow + t) * H + unit;
            if (is1) {
                a.r = T.r1[o], a.z = T.z1[o], a.n = T.n1[o], a.hn = T.hn1[o], a.hp = T.h1p[o];
            } else {
                a.r = T.r2[o], a.z = T.z2[o], a.n = T.n2[o], a.hn = T.hn2[o], a.hp = T.h2p[o], a.h = T.h2[o];
            }
        }
        return a;
    };
    // dpre and r2 of the step's frame go through LDS (one value / four values per thread, fetched a step ahead, stored behind the
    // products: a gate thread would hold 18 registers of dpre, a gathering thread 16 of r2)
    const int du = tid / WFC, dout = tid - du * WFC;  // thread < 288: dpre[utterance du][output dout]
    auto load_dpv = [&](int t) { return (tid < WG * WFC && du < X.nu && t >= 0) ? T.dpre[((size_t)(X.b0 + du) * Lf + t) * WFC + dout] : 0.0f; };
    f32x4ws r2v;
    auto load_r2v = [&](int t) {  // thread p: units 4 (p / 16) .. + 3 ... of utterance p % 16: r2 image [unit][utterance]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int un = 4 * (tid >> 4) + i, uu = tid & 15;
            r2v[i] = (uu < X.nu && t >= 0) ? T.r2[((size_t)(X.b0 + uu) * Lf + t) * WH2 + un] : 0.0f;
        }
    };
    auto stage = [&](float dpv) {
        if (tid < WG * WFC) L.dp[du][dout] = dpv;
#pragma unroll
        for (int i = 0; i < 4; ++i) L.r2[(4 * (tid >> 4) + i) * WG + (tid & 15)] = r2v[i];
    };
    Act an = load_act(is1 ? Lf : Lf - 1);  // (GRU1 has nothing to do in step 0)
    float dpv = load_dpv(Lf - 1);
    load_r2v(Lf - 1);
    stage(dpv);
    __syncthreads();
    float dhp = 0.0f, zp = 0.0f;  // dh and z of the thread's item at the frame of the previous step
#ifdef FPC_WS_PROF
    long long bprof[6] = {0, 0, 0, 0, 0, 0}, blast = __builtin_readcyclecounter();
    const long long t_loop = blast;
#define BSTAMPW(i)                                             \
    if (tid == FPC_BW_STAMP_TID) {                             \
        const long long now_ = __builtin_readcyclecounter();   \
        bprof[i] += now_ - blast;                              \
        blast = now_;                                          \
    }
#else
#define BSTAMPW(i)
#endif
    int k = 0;
    for (; k <= Lf; ++k) {
        const int t2 = Lf - 1 - k, t1 = t2 + 1;  // frames of GRU2 and of GRU1 in this step
        const bool do2 = t2 >= 0, do1 = k >= 1;
        const unsigned epoch = (unsigned)k + 1u;
        const int set = WNS + (k & 1) * WNS * BQ;
        // ---- phase 1 ----
        const Act a = an;
        if ((is1 && do1) || (is2 && do2)) {
            const int t = is1 ? t1 : t2;
            float dhn = 0.0f, dh;
            if (is1) {
                if (k >= 2) dhn = fmaf(dhp, zp, bw_tree(L.pa1, tv));  // dh1n(t1): gru_bwd's last line for frame t1 + 1
                dh = bw_tree(L.pa2, tv) + dhn;                        // (k = 1: pa2 holds W2i^T g2i(L - 1))
            } else {
                if (k >= 1) dhn = fmaf(dhp, zp, bw_tree(L.pb, tv));
                float dr = 0.0f;
#pragma unroll
                for (int o = 0; o < WFC; ++o) dr = fmaf(L.fcw[ij][o], L.dp[iu][o], dr);
                dh = (a.h > 0.0f ? dr : 0.0f) + dhn;
            }
            const float dn_raw = dh * (1.0f - a.z);
            const float dnpre = dn_raw * fmaf(-a.n, a.n, 1.0f);
            const float dz_raw = dh * (a.hp - a.n);
            const float dzpre = dz_raw * (a.z * (1.0f - a.z));
            const float drpre = (dnpre * a.hn) * (a.r * (1.0f - a.r));
            const float dnr = dnpre * a.r;
            dhp = dh;
            zp = a.z;
            if (live) {  // kept for the weight gradients (k_grad_tn)
                float* gi = is1 ? T.dgi1 : T.dgi2;
                float* gh = is1 ? T.dgh1 : T.dgh2;
                const size_t o = (nrow + t) * 3 * H + unit;
                gi[o] = drpre, gi[o + H] = dzpre, gi[o + 2 * H] = dnpre;
                gh[o] = drpre, gh[o + H] = dzpre, gh[o + 2 * H] = dnr;
            }
            if (k < Lf) {  // (the last step -- GRU1 at frame 0 -- feeds no product)
                if (is1) {
                    ws_store(X, set + slice * BQ + it, epoch, drpre, dzpre, dnr);
                    L.g1[(0 * WH1 + unit) * WG + iu] = drpre;
                    L.g1[(1 * WH1 + unit) * WG + iu] = dzpre;
                    L.g1[(2 * WH1 + unit) * WG + iu] = dnr;
                } else {
                    ws_store(X, set + slice * BQ + BQ1 + it, epoch, drpre, dzpre, dnpre);
                    L.g2[(0 * WH2 + unit) * WG + iu] = drpre;
                    L.g2[(1 * WH2 + unit) * WG + iu] = dzpre;
                    L.g2[(2 * WH2 + unit) * WG + iu] = dnpre;
                    L.g2[(3 * WH2 + unit) * WG + iu] = dnr;
                }
            }
        }
        if (k == Lf) break;
        BSTAMPW(0)
        // ---- the hop: the other 31 workgroups' items of this step into the images ----
        // (thread p polls item p % 256 of the source workgroups 2 i + p / 256 until every wanted tag of the WAVE shows the epoch;
        //  the granule index is formed where it is used: an index array would be 16 more live registers)
        // (in two halves of 8 granules: 16 at once keep 64 registers of payload live next to the wave's B operands, and the
        //  allocator answers by leaving operands in scratch memory -- 63 registers, reloaded inside the MFMA chains every step)
        {
            const bool want = g_is1 ? do1 : do2;
            bool ok = !ws_dead(L);
#pragma unroll 1
#ifndef FPC_BW_NH
#define FPC_BW_NH (BNG / 2)
#endif
            for (int hf = 0; hf < BNG / FPC_BW_NH && ok; ++hf) {
                constexpr int NH = FPC_BW_NH;
                u32x4 v[NH];
                const int g0 = set + (gs0 + 2 * NH * hf) * BQ + gitem;  // + 2 BQ per i
                if (!__all(!want)) {
                    unsigned spins = 0;
                    unsigned long long tm0 = 0, last = 0;
                    for (;;) {
                        bool all = true;
#pragma unroll
                        for (int i = 0; i < NH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, (g0 + 2 * BQ * i) * 16, 0, 16);
#pragma unroll
                        for (int i = 0; i < NH; ++i) all &= !want || 2 * (i + NH * hf) + gs0 == slice || v[i].x == epoch;
                        if (__all(all)) break;
                        if (ws_dead(L)) {
                            ok = false;
                            break;
                        }
                        if ((++spins & 63u) == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                            if (tm0 == 0 || now - last > spin_rearm_gap(X.limit)) tm0 = now;  // (this wave was descheduled: await_granule)
                            last = now;
                            if (now - tm0 > X.limit || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                                ws_give_up(X, L);
                                ok = false;
                                break;
                            }
                        }
                        __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");  // (the loads are re-issued every round)
                    }
                }
                BSTAMPW(1)
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const int s = 2 * (i + NH * hf) + gs0;
                    if (ok && want && s != slice) {
                        const float v0 = __uint_as_float(v[i].y), v1 = __uint_as_float(v[i].z), v2 = __uint_as_float(v[i].w);
                        if (g_is1) {
                            const int u1 = WU1 * s + gj;
                            L.g1[(0 * WH1 + u1) * WG + gu] = v0;
                            L.g1[(1 * WH1 + u1) * WG + gu] = v1;
                            L.g1[(2 * WH1 + u1) * WG + gu] = v2;
                        } else {
                            const int u2 = WU2 * s + gj;
                            L.g2[(0 * WH2 + u2) * WG + gu] = v0;
                            L.g2[(1 * WH2 + u2) * WG + gu] = v1;
                            L.g2[(2 * WH2 + u2) * WG + gu] = v2;
                            L.g2[(3 * WH2 + u2) * WG + gu] = v2 * L.r2[u2 * WG + gu];  // dgh[2 H + i] = dnpre * r[i] (gru_bwd)
                        }
                    }
                }
                BSTAMPW(2)
            }
        }
        // ONE barrier: the images are whole behind it, and it carries the give-up flag -- thread 0 copies the flag in front of it,
        // everybody acts on the copy (a wait given up behind the copy shows a step later: every poll fails at once from then
        // on, the status bit is set, the step's update is skipped either way)
        if (tid == 0) L.dead_latch = __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_barrier();
        if (L.dead_latch != 0) break;
        BSTAMPW(3)
        // ---- the next step's activations on their way (consumed in the next phase 1, behind the products) ----
        an = load_act(is1 ? t1 - 1 : t2 - 1);
        dpv = load_dpv(t2 - 1);
        load_r2v(t2 - 1);
        // ---- phase 2: this workgroup's columns of the three transposed products ----
        if (wave < 4) {
            if (do1) {  // W1h^T g1h(t1), rows 288 wave ..: one chain of 72 dependent MFMAs (the canonical order is a chain)
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
            }
        } else if (do2) {  // W2i^T g2i(t2) and W2h^T g2h(t2), rows 96 sg ..: two chains of 24, interleaved
            const int sg = wave - 4;
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
            ws_put(L.pa2[sg], lane, ai);
            ws_put(L.pb[sg], lane, ah);
        }
        BSTAMPW(4)
        stage(dpv);  // (the gather of this step has read r2, the gate threads dpre: the next step's values go in)
        lds_barrier();
        BSTAMPW(5)
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
