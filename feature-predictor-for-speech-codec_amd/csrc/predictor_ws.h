// predictor_ws.h -- the predictor with its weights STATIONARY on chip and the batch on the matrix cores
// (included by predictor.hip inside its namespace, after predictor_df.h).
//
// The row-split kernels (predictor.hip, predictor_df.h) give every utterance its own workgroups and stream the whole
// 2.67 MB weight set from L2 once per frame and utterance: 337 MB per frame at 128 utterances, 79 % of the L2's bandwidth
// for 170 MFLOP.  Here a GROUP of 16 utterances (= the M dimension of one v_mfma_f32_16x16x4_f32 tile) runs on the 32
// workgroups of one XCD (one per CU; 8 groups = 128 utterances fill the chip), and workgroup s keeps 1/32 of every
// matrix for the whole launch:
//   GRU1 units 12 s .. 12 s + 11  -> 36 gate rows of W1h (384 x 36, 55 kB) in the REGISTERS of the four background waves
//                                    (wave w holds input segment w: 24 k-steps x 3 gate tiles = 72 B-operand registers),
//                                    36 rows of W1i (20 x 36) in LDS,
//   GRU2 units  4 s ..  4 s + 3   -> 12 gate rows of W2i (384 x 12) and W2h (128 x 12) in LDS,
//   the output layer (128 x 18)   -> in LDS, evaluated by every workgroup (its 64 MFMAs cost less than a hop).
// Per frame a workgroup evaluates its gate rows for all 16 utterances at once: A operand = the state image in LDS
// ([k][utterance]: lane l of k-step j reads image[64 j + l]), B operand = the weights, one wave per INPUT SEGMENT, the
// accumulator initialised with the bias for segment 0 and with 0 for the others, segment sums added as a balanced tree by the
// gate threads -- the canonical order of oracle/fpc_oracle.c (matvec_seg) and of every other predictor kernel here, and
// v_mfma_f32_16x16x4_f32 accumulates as a k-ordered fmaf chain: results are bit-identical to them (tests: forward,
// encoder and receiver against the row-split kernels and against the oracle).
// After each GRU the 12 (4) x 16 new state values of a workgroup go to the other 31 as 16-byte granules {epoch, 3 values}
// (one 16-byte store is one request to the L2: tag and values arrive together; 2 048 granules = 32 kB per hop and
// workgroup instead of 48 kB as 8-byte granules), published with plain stores when all 32 workgroups of the group report
// the same XCD (they stay in that XCD's L2, where the partners' L1-bypassing sc1 loads find them), else written through
// (sc1 stores): placement is arranged for (blocks 8 apart share an XCD under the observed round-robin dealing) and
// checked at run time, never assumed -- a different dealing costs speed, not correctness.
// Roles (512 threads): waves 0-3 walk the frame's chain -- I(t) = W1i x(t), GRU1 gates, hop 1, C(t) = W2i h1(t), GRU2
// gates, hop 2, output layer -- waves 4-7 compute the recurrent products one frame ahead (A(t+1) = W1h h1(t) after hop 1,
// B(t+1) = W2h h2(t) after hop 2) and take half of hop 1's gather; the roles meet through LDS counters (predictor_df.h).
// The encoder's frame tail (residual, thresholds, searches: encode_frame) runs on the workgroup that OWNS the utterance
// (workgroup s < 16 owns utterance s of the group) and the next input row goes round as a third hop.
// Shapes: the reference's production predictor only (20 -> 384 -> 128 -> 18, README.md:26; train_frame.py:198-200);
// other shapes run the row-split kernels.
// Reference: Wavernn.forward / Wavernn.encoder (models/wavernn.py:63-102, 165-256).

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4ws __attribute__((ext_vector_type(4)));

constexpr int WG = 16;    // utterances per group (one MFMA M tile)
constexpr int WNS = 32;   // workgroups per group (the CUs of one XCD)
constexpr int WIN = 20, WH1 = 384, WH2 = 128, WFC = 18;
constexpr int WU1 = WH1 / WNS, WU2 = WH2 / WNS;  // units per workgroup: 12, 4
constexpr int WV1 = WU1 * WG, WV2 = WU2 * WG;    // state values per workgroup and hop: 192, 64
constexpr int WQ1 = WV1 / 3, WQ2 = (WV2 + 2) / 3;  // 16-byte granules per workgroup and hop: 64, 22
constexpr int WQX = WFC / 3;                     // granules of an utterance's next input row (hop 3): 6
constexpr int WFG = 4, WBG = 4;                  // waves per role
constexpr int WFGT = WFG * 64;
// granule block of a group, in 16-byte units: hello | h1 | h2 | next input
constexpr int WOFF_HELLO = 0, WOFF_H1 = WNS, WOFF_H2 = WOFF_H1 + WNS * WQ1, WOFF_X = WOFF_H2 + WNS * WQ2,
              WGRANULES = WOFF_X + WG * WQX;  // 2 880 granules = 46 080 bytes
static_assert(NT == 512, "predictor_ws.h is written for 8 waves per workgroup");
static_assert(WH1 / 4 == 96 && WH2 / 2 == 64 && WFC == NDIM + 1, "production shape");
enum { WSIG_A = 0, WSIG_B, WSIG_H1, WSIG_H2, WSIG_P1, WSIG_FG, WNSIG };

struct WsArgs {
    int B, ngroups;
    u32x4* g;                  // [ngroups][WGRANULES], zeroed before the launch
    unsigned* err;             // the handle's status word
    unsigned long long limit;  // give-up bound of one spin, s_memrealtime ticks
    int withhold;              // test hook: the last workgroup of group 0 never publishes
    int no_fast;               // FPC_FAST_HOP=0: always the write-through path (tests run both)
};

struct __attribute__((aligned(16))) WsLds : SearchLds {
    float x[WIN * WG];        // state images [k][utterance]
    float h1[WH1 * WG];
    float h2[WH2 * WG];
    float pI[3][256];         // segment sums as the MFMA leaves them: [gate][unit * 16 + utterance]
    float pA[4][3][256];      // [segment][gate][...]
    float pC[4][256];         // [segment][(gate * 4 + unit) * 16 + utterance]
    float pB[2][256];
    float pF[8][2][256];      // [segment][tile][row in tile * 16 + utterance]
    float fo[WG][WIN];        // predictions [utterance][row < 18]
    float xn[MAX_IN];         // the owner's next input row
    float w1i[WIN * 3 * WU1];  // [k][gate * 12 + unit]
    float w2i[WH1 * 3 * WU2];  // [k][gate * 4 + unit]
    float w2h[WH2 * 3 * WU2];
    float fcw[WH2 * WFC];      // [k][row]
    int sig[WNSIG];
    int dead;
    int same_xcd;
};

struct WsCtx {
    __amdgpu_buffer_rsrc_t rs;  // this group's granule block
    int slice, nu, b0;          // this workgroup's slice, valid utterances of the group, first utterance
    unsigned* err;
    unsigned long long limit;
    bool fast, withhold;
};

__device__ __forceinline__ bool ws_dead(WsLds& L) {
    return __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
}
__device__ __forceinline__ void ws_give_up(const WsCtx& X, WsLds& L) {
    status_or(X.err, FPC_ST_TIMEOUT);
    __hip_atomic_store(&L.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// barrier of the foreground waves only
__device__ __forceinline__ void ws_fg_sync(WsLds& L, int& fg_epoch) {
    ++fg_epoch;
    df_signal(&L.sig[WSIG_FG]);
    (void)df_wait(&L.sig[WSIG_FG], WFG * fg_epoch, &L.dead);
}
__device__ __forceinline__ void ws_store(const WsCtx& X, int granule, unsigned epoch, float v0, float v1, float v2) {
    if (X.withhold) return;
    const u32x4 w = {epoch, __float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2)};
    if (X.fast)
        __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, granule * 16, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, granule * 16, 0, 16);  // sc1: write-through
}
// N granules per lane (granule index, or -1: none), polled until every wanted tag of the WAVE equals `epoch`; false: the
// wait was given up (timeout, or the workgroup is dead already)
template <int N>
__device__ __forceinline__ bool ws_poll(const WsCtx& X, WsLds& L, const int (&gi)[N], unsigned epoch, u32x4 (&v)[N]) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, (gi[j] < 0 ? 0 : gi[j]) * 16, 0, 16);
#pragma unroll
        for (int j = 0; j < N; ++j) ok &= gi[j] < 0 || v[j].x == epoch;
        if (__all(ok)) return true;
        if (ws_dead(L)) return false;
        if ((++spins & 63u) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (now - t0 > X.limit || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                ws_give_up(X, L);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");  // (the loads are re-issued every round)
    }
}

__device__ __forceinline__ bool ws_role(int ngroups, int& group, int& slice) {
    const int i = blockIdx.x, x = i % 8, m = i / 8;
    slice = m % WNS;
    group = (m / WNS) * 8 + x;
    return group < ngroups;
}

// per-wave constants of the matrix products (registers for the whole launch)
struct WsRegs {
    float wA[3][24];   // background wave w: W1h rows of this workgroup, input segment w, as MFMA B operands
    float bA[3];       // (segment 0 only) b_hh of GRU1
    float bB;          // b_hh of GRU2 (background wave 0)
    float bI;          // foreground wave w < 3: b_ih of GRU1, gate w
    float bC;          // foreground wave 0: b_ih of GRU2
    float bF[2];       // foreground wave 0: output bias, tiles 0 and 1
};

// copies this workgroup's weight slices to LDS / registers (all threads; no barrier inside)
__device__ __forceinline__ void ws_load_weights(const PredDev& P, WsLds& L, WsRegs& R, int slice, int tid) {
    const int wave = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
    for (int i = tid; i < WIN * 36; i += NT) {
        const int k = i / 36, r = i - k * 36, g = r / WU1, u = r - g * WU1;
        L.w1i[i] = P.w1i[(size_t)k * 3 * WH1 + g * WH1 + WU1 * slice + u];
    }
    for (int i = tid; i < WH1 * 12; i += NT) {
        const int k = i / 12, r = i - k * 12, g = r / WU2, u = r - g * WU2;
        L.w2i[i] = P.w2i[(size_t)k * 3 * WH2 + g * WH2 + WU2 * slice + u];
    }
    for (int i = tid; i < WH2 * 12; i += NT) {
        const int k = i / 12, r = i - k * 12, g = r / WU2, u = r - g * WU2;
        L.w2h[i] = P.w2h[(size_t)k * 3 * WH2 + g * WH2 + WU2 * slice + u];
    }
    for (int i = tid; i < WH2 * WFC; i += NT) L.fcw[i] = P.fcw[i];
    const int fw = wave, bw = wave - WFG;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        R.bA[g] = 0.0f;
#pragma unroll
        for (int j = 0; j < 24; ++j) R.wA[g][j] = 0.0f;
    }
    R.bB = R.bI = R.bC = 0.0f;
    R.bF[0] = R.bF[1] = 0.0f;
    if (wave >= WFG) {
        if (c < WU1) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (bw == 0) R.bA[g] = P.b1h[g * WH1 + WU1 * slice + c];
#pragma unroll
                for (int j = 0; j < 24; ++j)
                    R.wA[g][j] = P.w1h[(size_t)(96 * bw + 4 * j + q) * 3 * WH1 + g * WH1 + WU1 * slice + c];
            }
        }
        if (bw == 0 && c < 12) R.bB = P.b2h[(c / WU2) * WH2 + WU2 * slice + (c % WU2)];
    } else {
        if (fw < 3 && c < WU1) R.bI = P.b1i[fw * WH1 + WU1 * slice + c];
        if (fw == 0) {
            if (c < 12) R.bC = P.b2i[(c / WU2) * WH2 + WU2 * slice + (c % WU2)];
            R.bF[0] = P.fcb[c];
            if (c < WFC - 16) R.bF[1] = P.fcb[16 + c];
        }
    }
}

// once per launch (all threads; ends with barriers): X.fast, or L.dead when a partner never shows up
__device__ __forceinline__ void ws_hello(WsCtx& X, WsLds& L, const WsArgs& S, int tid) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xfu) + 1u;
    const unsigned tag = 0xffffffffu;  // (no frame's epoch)
    if (tid == 0) {
        L.same_xcd = 1;
        if (!X.withhold) {  // general path: nothing is known about the placement yet
            const u32x4 w = {tag, xcc, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, (WOFF_HELLO + X.slice) * 16, 0, 16);
        }
    }
    __syncthreads();
    if (tid < 64) {
        int gi[1] = {tid < WNS && tid != X.slice ? WOFF_HELLO + tid : -1};
        u32x4 v[1];
        const bool got = ws_poll<1>(X, L, gi, tag, v);
        if (got && gi[0] >= 0 && v[0].y != xcc) L.same_xcd = 0;
    }
    __syncthreads();
    X.fast = L.same_xcd != 0 && !ws_dead(L) && S.no_fast == 0;
}

// ---- the matrix products (one wave each; `lane` = c + 16 q) ----
__device__ __forceinline__ f32x4ws ws_mfma(float a, float b, f32x4ws c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void ws_put(float* p, int lane, const f32x4ws& acc) {
    *reinterpret_cast<f32x4ws*>(&p[(lane & 15) * 16 + 4 * (lane >> 4)]) = acc;  // utterances 4 q .. 4 q + 3 of column c
}
// A = W1h h1: background wave bw = input segment, three gate tiles from register weights
__device__ __forceinline__ void ws_A(WsLds& L, const WsRegs& R, int bw, int lane) {
    f32x4ws acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4ws{R.bA[g], R.bA[g], R.bA[g], R.bA[g]};
    const float* hs = L.h1 + 96 * bw * WG + lane;
#pragma unroll
    for (int j = 0; j < 24; ++j) {
        const float a = hs[64 * j];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = ws_mfma(a, R.wA[g][j], acc[g]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) ws_put(L.pA[bw][g], lane, acc[g]);
}
// B = W2h h2: background waves 0, 1 = input segments of 64
__device__ __forceinline__ void ws_B(WsLds& L, const WsRegs& R, int bw, int lane) {
    const int c = lane & 15, q = lane >> 4, cc = c < 12 ? c : 11;
    f32x4ws acc = {R.bB, R.bB, R.bB, R.bB};
    const float* hs = L.h2 + 64 * bw * WG + lane;
    const float* ws = L.w2h + (64 * bw + q) * 12 + cc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = ws_mfma(hs[64 * j], ws[48 * j], acc);
    ws_put(L.pB[bw], lane, acc);
}
// I = W1i x: foreground wave fw < 3 = gate tile (one segment of 20 inputs)
__device__ __forceinline__ void ws_I(WsLds& L, const WsRegs& R, int fw, int lane) {
    const int c = lane & 15, q = lane >> 4, cc = c < WU1 ? c : WU1 - 1;
    f32x4ws acc = {R.bI, R.bI, R.bI, R.bI};
    const float* xs = L.x + lane;
    const float* ws = L.w1i + q * 36 + fw * WU1 + cc;
#pragma unroll
    for (int j = 0; j < WIN / 4; ++j) acc = ws_mfma(xs[64 * j], ws[144 * j], acc);
    ws_put(L.pI[fw], lane, acc);
}
// C = W2i h1: foreground wave fw = input segment of 96
__device__ __forceinline__ void ws_C(WsLds& L, const WsRegs& R, int fw, int lane) {
    const int c = lane & 15, q = lane >> 4, cc = c < 12 ? c : 11;
    f32x4ws acc = {R.bC, R.bC, R.bC, R.bC};
    const float* hs = L.h1 + 96 * fw * WG + lane;
    const float* ws = L.w2i + (96 * fw + q) * 12 + cc;
#pragma unroll
    for (int j = 0; j < 24; ++j) acc = ws_mfma(hs[64 * j], ws[48 * j], acc);
    ws_put(L.pC[fw], lane, acc);
}
// output layer on relu(h2): foreground wave fw = input segments 2 fw, 2 fw + 1 (16 inputs each), two row tiles
__device__ __forceinline__ void ws_F(WsLds& L, const WsRegs& R, int fw, int lane) {
    const int c = lane & 15, q = lane >> 4, c1 = c < WFC - 16 ? 16 + c : 16;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int sg = 2 * fw + s2;
        const float b0 = sg == 0 ? R.bF[0] : 0.0f, b1 = sg == 0 ? R.bF[1] : 0.0f;
        f32x4ws a0 = {b0, b0, b0, b0}, a1 = {b1, b1, b1, b1};
        const float* hs = L.h2 + 16 * sg * WG + lane;
        const float* ws = L.fcw + (16 * sg + q) * WFC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hv = hs[64 * j];
            const float a = hv > 0.0f ? hv : 0.0f;  // (the rectified state, formed in the chain)
            a0 = ws_mfma(a, ws[4 * WFC * j + c], a0);
            a1 = ws_mfma(a, ws[4 * WFC * j + c1], a1);
        }
        ws_put(L.pF[sg][0], lane, a0);
        ws_put(L.pF[sg][1], lane, a1);
    }
}

// hop 1 gather, this wave's share: wave wv of 8 takes the slices wv, wv + 8, wv + 16, wv + 24 (lane = granule)
__device__ __forceinline__ bool ws_gather1(const WsCtx& X, WsLds& L, int wv, int lane, unsigned epoch, bool guard_A, int t) {
    int gi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = wv + 8 * j;
        gi[j] = s == X.slice ? -1 : WOFF_H1 + s * WQ1 + lane;
    }
    u32x4 v[4];
    if (!ws_poll<4>(X, L, gi, epoch, v)) return false;
    // (a background wave may be here before its neighbours have finished A(t) on the old image)
    if (guard_A && !df_wait(&L.sig[WSIG_A], WBG * (t + 1), &L.dead)) return false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = wv + 8 * j;
        if (s != X.slice) {
            float* d = L.h1 + s * WV1 + 3 * lane;
            d[0] = __uint_as_float(v[j].y);
            d[1] = __uint_as_float(v[j].z);
            d[2] = __uint_as_float(v[j].w);
        }
    }
    return true;
}
// hop 2 gather by the 256 foreground threads
__device__ __forceinline__ bool ws_gather2(const WsCtx& X, WsLds& L, int ft, unsigned epoch) {
    int gi[3], sl[3], e[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = ft + WFGT * j;
        sl[j] = i / WQ2;
        e[j] = i - sl[j] * WQ2;
        gi[j] = (i < WNS * WQ2 && sl[j] != X.slice) ? WOFF_H2 + i : -1;
    }
    u32x4 v[3];
    if (!ws_poll<3>(X, L, gi, epoch, v)) return false;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (gi[j] >= 0) {
            float* d = L.h2 + sl[j] * WV2 + 3 * e[j];
            d[0] = __uint_as_float(v[j].y);
            if (3 * e[j] + 1 < WV2) d[1] = __uint_as_float(v[j].z);
            if (3 * e[j] + 2 < WV2) d[2] = __uint_as_float(v[j].w);
        }
    }
    return true;
}

// FOREGROUND, frame t: L.x = x(t) -> L.fo (all 16 utterances), states in L.h1 / L.h2; false: the launch is dead
__device__ __forceinline__ bool ws_foreground(const WsCtx& X, WsLds& L, const WsRegs& R, int t, int ft, int& fg_epoch) {
    const int fw = ft >> 6, lane = ft & 63;
    const unsigned epoch = (unsigned)t + 1u;
    if (fw < 3) ws_I(L, R, fw, lane);
    ws_fg_sync(L, fg_epoch);
    if (!df_wait(&L.sig[WSIG_A], WBG * (t + 1), &L.dead)) return false;  // A(t): prologue, then one round per frame
    // ---- GRU1 gates of this workgroup's 12 units x 16 utterances: 48 values per wave; torch.nn.GRU rows [r; z; n]
    {
        const int base = X.slice * WV1 + 48 * fw;
        if (lane < 48) {
            const int v = 48 * fw + lane;
            const float ghr = (L.pA[0][0][v] + L.pA[1][0][v]) + (L.pA[2][0][v] + L.pA[3][0][v]);
            const float ghz = (L.pA[0][1][v] + L.pA[1][1][v]) + (L.pA[2][1][v] + L.pA[3][1][v]);
            const float ghn = (L.pA[0][2][v] + L.pA[1][2][v]) + (L.pA[2][2][v] + L.pA[3][2][v]);
            const float r = fpc_sigmoidf(L.pI[0][v] + ghr);
            const float z = fpc_sigmoidf(L.pI[1][v] + ghz);
            const float n = fpc_tanhf(fmaf(r, ghn, L.pI[2][v]));
            const float hp = L.h1[base + lane];
            L.h1[base + lane] = fmaf(z, hp - n, n);
        }
        // (the same wave reads what it has just written: one in-order LDS queue per wave)
        if (lane < 16) {
            const float* s = &L.h1[base + 3 * lane];
            ws_store(X, WOFF_H1 + X.slice * WQ1 + 16 * fw + lane, epoch, s[0], s[1], s[2]);
        }
    }
    df_signal(&L.sig[WSIG_P1]);  // (the background starts polling now, not before)
    if (!ws_gather1(X, L, fw, lane, epoch, false, t)) return false;
    df_signal(&L.sig[WSIG_H1]);
    if (!df_wait(&L.sig[WSIG_H1], NW * (t + 1), &L.dead)) return false;  // h1(t) whole in LDS
    ws_C(L, R, fw, lane);
    ws_fg_sync(L, fg_epoch);
    if (!df_wait(&L.sig[WSIG_B], 2 * (t + 1), &L.dead)) return false;  // B(t)
    if (fw == 0) {  // GRU2 gates: 4 units x 16 utterances
        const int base = X.slice * WV2;
        const int v = lane;
        const float gir = (L.pC[0][v] + L.pC[1][v]) + (L.pC[2][v] + L.pC[3][v]);
        const float giz = (L.pC[0][64 + v] + L.pC[1][64 + v]) + (L.pC[2][64 + v] + L.pC[3][64 + v]);
        const float gin = (L.pC[0][128 + v] + L.pC[1][128 + v]) + (L.pC[2][128 + v] + L.pC[3][128 + v]);
        const float ghr = L.pB[0][v] + L.pB[1][v];
        const float ghz = L.pB[0][64 + v] + L.pB[1][64 + v];
        const float ghn = L.pB[0][128 + v] + L.pB[1][128 + v];
        const float r = fpc_sigmoidf(gir + ghr);
        const float z = fpc_sigmoidf(giz + ghz);
        const float n = fpc_tanhf(fmaf(r, ghn, gin));
        const float hp = L.h2[base + v];
        L.h2[base + v] = fmaf(z, hp - n, n);
        if (lane < WQ2) {
            const int i0 = 3 * lane, i1 = i0 + 1 < WV2 ? i0 + 1 : WV2 - 1, i2 = i0 + 2 < WV2 ? i0 + 2 : WV2 - 1;
            ws_store(X, WOFF_H2 + X.slice * WQ2 + lane, epoch, L.h2[base + i0], L.h2[base + i1], L.h2[base + i2]);
        }
    }
    if (!ws_gather2(X, L, ft, epoch)) return false;
    df_signal(&L.sig[WSIG_H2]);
    if (!df_wait(&L.sig[WSIG_H2], WFG * (t + 1), &L.dead)) return false;  // h2(t) whole in LDS
    ws_F(L, R, fw, lane);
    ws_fg_sync(L, fg_epoch);
    for (int i = ft; i < WFC * WG; i += WFGT) {
        const int row = i >> 4, u = i & 15, tile = row >> 4, o = ((row & 15) << 4) + u;
        const float acc = ((L.pF[0][tile][o] + L.pF[1][tile][o]) + (L.pF[2][tile][o] + L.pF[3][tile][o])) +
                          ((L.pF[4][tile][o] + L.pF[5][tile][o]) + (L.pF[6][tile][o] + L.pF[7][tile][o]));
        const float tt = fpc_tanhf(acc);
        L.fo[u][row] = tt + tt;  // the "dual" FC is the same Linear summed twice (wavernn.py:89-92)
    }
    ws_fg_sync(L, fg_epoch);
    return !ws_dead(L);
}

// BACKGROUND, frame t: half of hop 1's gather, then A(t+1) and B(t+1)
__device__ __forceinline__ bool ws_background(const WsCtx& X, WsLds& L, const WsRegs& R, int t, bool last, int bt) {
    const int bw = bt >> 6, lane = bt & 63;
    if (!df_wait(&L.sig[WSIG_P1], WFG * (t + 1), &L.dead)) return false;
    if (!ws_gather1(X, L, WFG + bw, lane, (unsigned)t + 1u, true, t)) return false;
    df_signal(&L.sig[WSIG_H1]);
    if (!df_wait(&L.sig[WSIG_H1], NW * (t + 1), &L.dead)) return false;
    if (!last) ws_A(L, R, bw, lane);
    df_signal(&L.sig[WSIG_A]);
    if (bw < 2) {
        if (!df_wait(&L.sig[WSIG_H2], WFG * (t + 1), &L.dead)) return false;
        if (!last) ws_B(L, R, bw, lane);
        df_signal(&L.sig[WSIG_B]);
    }
    return true;
}

// everything before frame 0 (all threads; ends with a barrier): counters, weights, hello, A(0), B(0).
// The state images and x(0) have been written (and a barrier passed) by the caller.
__device__ __forceinline__ void ws_prologue(const PredDev& P, WsCtx& X, WsLds& L, WsRegs& R, const WsArgs& S, int tid) {
    if (tid < WNSIG) L.sig[tid] = 0;
    if (tid == 0) L.dead = (status_load(S.err) & FPC_ST_TIMEOUT) != 0u ? 1 : 0;  // (a failed handle waits for nobody)
    ws_load_weights(P, L, R, X.slice, tid);
    __syncthreads();
    ws_hello(X, L, S, tid);
    if (tid >= WFGT) {
        const int bw = (tid - WFGT) >> 6, lane = tid & 63;
        ws_A(L, R, bw, lane);
        df_signal(&L.sig[WSIG_A]);
        if (bw < 2) {
            ws_B(L, R, bw, lane);
            df_signal(&L.sig[WSIG_B]);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ WsCtx ws_ctx(const WsArgs& S, int group, int slice) {
    WsCtx X;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(S.g + (size_t)group * WGRANULES), 0, WGRANULES * 16, 0x00020000);
    X.slice = slice;
    X.b0 = group * WG;
    X.nu = S.B - X.b0 < WG ? S.B - X.b0 : WG;
    X.err = S.err;
    X.limit = S.limit;
    X.fast = false;
    X.withhold = S.withhold != 0 && group == 0 && slice == WNS - 1;
    return X;
}

__global__ __launch_bounds__(NT) void k_forward_ws(const PredDev P, const float* __restrict__ x, int Lf, float* h1, float* h2,
                                                   float* __restrict__ y, const WsArgs S) {
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) {  // state images [k][u] from [utterance][k]
        const int u = i / WH1, k = i - u * WH1;
        L.h1[k * WG + u] = u < X.nu ? h1[(size_t)(X.b0 + u) * WH1 + k] : 0.0f;
    }
    for (int i = tid; i < WH2 * WG; i += NT) {
        const int u = i / WH2, k = i - u * WH2;
        L.h2[k * WG + u] = u < X.nu ? h2[(size_t)(X.b0 + u) * WH2 + k] : 0.0f;
    }
    for (int i = tid; i < WIN * WG; i += NT) {
        const int u = i / WIN, k = i - u * WIN;
        L.x[k * WG + u] = (u < X.nu && Lf > 0) ? x[(size_t)(X.b0 + u) * Lf * WIN + k] : 0.0f;
    }
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    const bool owner = slice < X.nu;  // this workgroup stores the outputs of utterance `slice` of the group
    if (tid < WFGT) {
        __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
        int fg_epoch = 0;
        for (int t = 0; t < Lf; ++t) {
            float xn[2] = {0.0f, 0.0f};  // (teacher forcing: the next input rows are fetched while this frame runs)
            if (t + 1 < Lf) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = tid + WFGT * j, u = i / WIN, k = i - u * WIN;
                    if (i < WIN * WG && u < X.nu) xn[j] = x[((size_t)(X.b0 + u) * Lf + t + 1) * WIN + k];
                }
            }
            if (!ws_foreground(X, L, R, t, tid, fg_epoch)) break;
            if (owner && tid < WFC) y[((size_t)(X.b0 + slice) * Lf + t) * WFC + tid] = L.fo[slice][tid];
            if (t + 1 < Lf) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int i = tid + WFGT * j, u = i / WIN, k = i - u * WIN;
                    if (i < WIN * WG) L.x[k * WG + u] = xn[j];
                }
                ws_fg_sync(L, fg_epoch);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        for (int tb = 0; tb < Lf; ++tb)
            if (!ws_background(X, L, R, tb, tb + 1 == Lf, tid - WFGT)) break;
    }
    __syncthreads();
    // new states of this workgroup's units; a launch that gave up fails loudly: NaN outputs and states, FPC_ERR_TIMEOUT
    const bool dead = ws_dead(L);
    const float qnan = __uint_as_float(0x7fc00000u);
    if (dead && owner)
        for (size_t k = tid; k < (size_t)Lf * WFC; k += NT) y[(size_t)(X.b0 + slice) * Lf * WFC + k] = qnan;
    for (int v = tid; v < WV1; v += NT) {
        const int c = v >> 4, u = v & 15;
        if (u < X.nu) h1[(size_t)(X.b0 + u) * WH1 + WU1 * slice + c] = dead ? qnan : L.h1[slice * WV1 + v];
    }
    for (int v = tid; v < WV2; v += NT) {
        const int c = v >> 4, u = v & 15;
        if (u < X.nu) h2[(size_t)(X.b0 + u) * WH2 + WU2 * slice + c] = dead ? qnan : L.h2[slice * WV2 + v];
    }
}

// hop 3: the owners' next input rows (18 values = 6 granules each) -> x image; threads p < 96 of waves 0, 1
__device__ __forceinline__ bool ws_gather3(const WsCtx& X, WsLds& L, int p, unsigned epoch) {
    const int u = p / WQX, e = p - u * WQX;
    int gi[1] = {(p < WG * WQX && u < X.nu && u != X.slice) ? WOFF_X + p : -1};
    u32x4 v[1];
    if (!ws_poll<1>(X, L, gi, epoch, v)) return false;
    if (gi[0] >= 0) {
        L.x[(3 * e + 0) * WG + u] = __uint_as_float(v[0].y);
        L.x[(3 * e + 1) * WG + u] = __uint_as_float(v[0].z);
        L.x[(3 * e + 2) * WG + u] = __uint_as_float(v[0].w);
    }
    return true;
}
// the owner's next input row L.xn goes out and into its own column of the x image (threads of wave 0)
__device__ __forceinline__ void ws_publish_x(const WsCtx& X, WsLds& L, int lane, unsigned epoch) {
    if (lane < WQX) ws_store(X, WOFF_X + X.slice * WQX + lane, epoch, L.xn[3 * lane], L.xn[3 * lane + 1], L.xn[3 * lane + 2]);
    if (lane < WIN) L.x[lane * WG + X.slice] = L.xn[lane];
}

__global__ __launch_bounds__(NT) void k_encode_ws(const PredDev P, const CbDev C, const EncArgs A, const WsArgs S) {
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) L.h1[i] = 0.0f;  // h = None -> zeros (wavernn.py:182)
    for (int i = tid; i < WH2 * WG; i += NT) L.h2[i] = 0.0f;
    for (int i = tid; i < WIN * WG; i += NT) L.x[i] = 0.0f;   // c_in[:, 0, :] is all zero (wavernn.py:177-178)
    const bool scl_in_lds = C.n_hi + C.n_lo <= SCLC;
    if (scl_in_lds) {
        for (int k = tid; k < C.n_hi; k += NT) L.sclc[k] = C.scl_hi[k];
        for (int k = tid; k < C.n_lo; k += NT) L.sclc[C.n_hi + k] = C.scl_lo[k];
    }
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    const bool owner = slice < X.nu;
    const int b = X.b0 + slice;  // the owned utterance
    int fg_epoch = 0;
    int i = 0;
    for (; i < A.Lf; ++i) {
        const unsigned epoch = (unsigned)i + 1u;
        // this frame's feature row of the owned utterance (one column per thread), and the pitch columns of every
        // utterance of the group, which pass through to the next input (wavernn.py:178): fetched before the step
        const float fv = (owner && tid < WIN) ? A.feat[((size_t)b * A.Lf + i) * WIN + tid] : 0.0f;
        float pv = 0.0f;
        if (tid < (WIN - WFC) * WG) {
            const int u = tid / (WIN - WFC), k = WFC + tid % (WIN - WFC);
            if (u < X.nu) pv = A.feat[((size_t)(X.b0 + u) * A.Lf + i) * WIN + k];
        }
        if (tid < WFGT) {
            __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
            (void)ws_foreground(X, L, R, i, tid, fg_epoch);
            __builtin_amdgcn_s_setprio(0);
        } else {
            (void)ws_background(X, L, R, i, i + 1 == A.Lf, tid - WFGT);
        }
        if (__syncthreads_or(ws_dead(L))) break;  // both roles meet: the searches take the whole workgroup
        if (owner) {
            encode_frame(L, L.fo[slice], L.xn, P, C, A, S.err, (size_t)b * A.Lf + i, fv, true, tid, scl_in_lds);
            if (tid < 64) ws_publish_x(X, L, tid, epoch);
        }
        if (tid < (WIN - WFC) * WG) {
            const int u = tid / (WIN - WFC), k = WFC + tid % (WIN - WFC);
            if (!(owner && u == slice)) L.x[k * WG + u] = pv;
        }
        bool ok = true;
        if (tid < 128) ok = ws_gather3(X, L, tid, epoch);
        if (__syncthreads_or(!ok)) break;
    }
    if (i < A.Lf && owner) encode_poison(P, A, b, i, tid);
}

__global__ __launch_bounds__(NT) void k_decode_feat_ws(const PredDev P, const CbDev C, const float* __restrict__ pitch,
                                                       const int* __restrict__ idx, int Lf, float* __restrict__ c_out,
                                                       int* bad, const WsArgs S) {
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) L.h1[i] = 0.0f;
    for (int i = tid; i < WH2 * WG; i += NT) L.h2[i] = 0.0f;
    for (int i = tid; i < WIN * WG; i += NT) L.x[i] = 0.0f;
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    const bool owner = slice < X.nu;
    const int b = X.b0 + slice;
    int fg_epoch = 0;
    int i = 0;
    for (; i < Lf; ++i) {
        const unsigned epoch = (unsigned)i + 1u;
        float pv = 0.0f;  // the pitch columns are side information of the receiver
        if (tid < (WIN - WFC) * WG) {
            const int u = tid / (WIN - WFC), k = tid % (WIN - WFC);
            if (u < X.nu) pv = pitch[((size_t)(X.b0 + u) * Lf + i) * (WIN - WFC) + k];
        }
        if (tid < WFGT) {
            __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
            (void)ws_foreground(X, L, R, i, tid, fg_epoch);
            __builtin_amdgcn_s_setprio(0);
        } else {
            (void)ws_background(X, L, R, i, i + 1 == Lf, tid - WFGT);
        }
        if (__syncthreads_or(ws_dead(L))) break;
        if (owner && tid < 64) {  // the residual is a lookup: one wave rebuilds the owned utterance's next input row
            decode_frame(L.fo[slice], L.xn, P, C, pitch, idx, c_out, bad, (size_t)b * Lf + i, true, tid);
            ws_publish_x(X, L, tid, epoch);
        }
        if (tid < (WIN - WFC) * WG) {
            const int u = tid / (WIN - WFC), k = WFC + tid % (WIN - WFC);
            if (!(owner && u == slice)) L.x[k * WG + u] = pv;
        }
        bool ok = true;
        if (tid >= 64 && tid < 192) ok = ws_gather3(X, L, tid - 64, epoch);
        if (__syncthreads_or(!ok)) break;
    }
    if (i < Lf && owner) {  // fail loudly: NaN from this frame on; the host reports FPC_ERR_TIMEOUT
        const float qnan = __uint_as_float(0x7fc00000u);
        for (size_t k = (size_t)i * WIN + tid; k < (size_t)Lf * WIN; k += NT) c_out[(size_t)b * Lf * WIN + k] = qnan;
    }
}
