// predictor_ws.h -- the predictor with its weights STATIONARY on chip and the batch on the matrix cores
// (included by predictor.hip inside its namespace, after predictor_df.h).
//
// The row-split kernels (predictor.hip, predictor_df.h) give every utterance its own workgroups and stream the whole
// 2.67 MB weight set from L2 once per frame and utterance: 337 MB per frame at 128 utterances, 79 % of the L2's bandwidth
// for 170 MFLOP.  Here a GROUP of 16 utterances (= the M dimension of one v_mfma_f32_16x16x4_f32 tile) runs on the 32
// workgroups of one XCD (one per CU; 8 groups = 128 utterances fill the chip), and workgroup s keeps 1/32 of every
// matrix for the whole launch:
//   GRU1 units 12 s .. 12 s + 11  -> 36 gate rows of W1h (384 x 36, 55 kB) and of W1i (20 x 36),
//   GRU2 units  4 s ..  4 s + 3   -> 12 gate rows of W2i (384 x 12) and W2h (128 x 12),
//   the output layer (128 x 18)   -> whole, evaluated by the workgroups that own an utterance (64 MFMAs: less than a hop),
// all in LDS (92 kB of weights in rows of 12 floats, so that the four k of an MFMA step read disjoint banks; registers
// would hold W1h as well -- 72 per lane of the background waves -- but the encoder's float64 searches need them).
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
// The encoder's frame tail (residual, thresholds, searches) is distributed over the group's workgroups (predictor_wsd.h);
// the receiver's (a table lookup) runs on the workgroup that OWNS the utterance (workgroup s < 16 owns utterance s of the
// group) and the next input row goes round as a third hop.
// Shapes: the reference's production predictor only (20 -> 384 -> 128 -> 18, README.md:26; train_frame.py:198-200);
// other shapes run the row-split kernels.
// Reference: Wavernn.forward / Wavernn.encoder (models/wavernn.py:63-102, 165-256).

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4ws __attribute__((ext_vector_type(4)));

#ifndef FPC_WS_A_SPLIT
#define FPC_WS_A_SPLIT 0  // the teacher-forced forward's A(t+1): first unit's chain beside C(t), the rest behind it (ws_A)
#endif
#ifndef FPC_WS_HOIST
#define FPC_WS_HOIST 1  // teacher-forced forward: lane-constant address arithmetic may leave the frame loop (the closed loops keep it inside: registers)
#endif
#ifndef FPC_WS_POLL_DELAY
#define FPC_WS_POLL_DELAY 0  // x 64 cycles before the first poll of a hop
#endif
constexpr int WG = 16;    // utterances per group (one MFMA M tile)
constexpr int WNS = 32;   // workgroups per group (the CUs of one XCD)
constexpr int WIN = 20, WH1 = 384, WH2 = 128, WFC = 18;
constexpr int WU1 = WH1 / WNS, WU2 = WH2 / WNS;  // units per workgroup: 12, 4
constexpr int WV1 = WU1 * WG, WV2 = WU2 * WG;    // state values per workgroup and hop: 192, 64
constexpr int WQ1 = WV1 / 3, WQ2 = (WV2 + 2) / 3;  // 16-byte granules per workgroup and hop: 64, 22
constexpr int WFG = 4, WBG = 4;                  // waves per role
constexpr int W1G = WIN * WU1 + 20;                // pitch of a gate's block of W1i in LDS
constexpr int WXP = 17;                          // pitch of a row of the input image (ws_xi)
constexpr int WPF = 20;                          // pitch of an utterance's 18 output rows in pFa (16-byte stores, 8 lanes: 8 bank quads)
constexpr int WFGT = WFG * 64;
// granule block of a group, in 16-byte units: hello | h1 | h2
// (the state hops have TWO granule sets each, used by frame parity: a workgroup publishes h1(t+1) once its gather of h1(t) is
//  complete, i.e. once every partner has published h1(t) -- which a partner does only with h1(t-1) whole in its LDS: the set
//  of parity t+1 is free.  Likewise h2(t+1) follows the own gather of h2(t), the partners' h2(t) their gather of h2(t-1).
//  With ONE set the publish of a frame had to wait for the previous frame's OTHER hop to prove that, which put hop 2 on the
//  teacher-forced forward's loop although nothing on that loop needs h2: profiles/r05_ablations.txt)
constexpr int WOFF_HELLO = 0, WOFF_H1 = WNS, WOFF_H2 = WOFF_H1 + 2 * WNS * WQ1,
              // | the distributed searches' results [utterance][workgroup][5], first and second stage (predictor_wsd.h)
              WOFF_G1 = WOFF_H2 + 2 * WNS * WQ2, WOFF_G2 = WOFF_G1 + WG * WNS * SURV,
              WGRANULES = WOFF_G2 + WG * WNS * SURV;  // 10 656 granules = 170 496 bytes
static_assert(NT == 512, "predictor_ws.h is written for 8 waves per workgroup");
static_assert(WH1 / 4 == 96 && WH2 / 2 == 64 && WFC == NDIM + 1, "production shape");
enum { WSIG_A = 0, WSIG_B, WSIG_H1, WSIG_H2, WSIG_P1, WSIG_P2, WSIG_C, WSIG_I, WSIG_X, WSIG_FB, WNSIG };

struct WsArgs {
    int B, ngroups;
    u32x4* g;                  // [ngroups][WGRANULES], zeroed before the launch
    unsigned* err;             // the handle's status word
    unsigned long long limit;  // give-up bound of one spin, s_memrealtime ticks
    int withhold;              // test hook: the last workgroup of group 0 never publishes (2: not even its hello)
    int no_fast;               // FPC_FAST_HOP=0: always the write-through path (tests run both)
    unsigned* dec;             // [ngroups] decision words (0 undecided, WS_GO, WS_FALLBACK: ws_hello), zeroed with the granules
    unsigned long long hello_limit;  // bound of the first wait for the partners' hello (ticks): behind it the group falls back
};

struct __attribute__((aligned(16))) WsLds : SearchLds {
    float x[WIN * WXP + 4];   // input image [k][utterance], rows WXP = 17 floats apart (ws_xi); 344 floats: the next member stays 16-byte aligned
    float h1[WH1 * WG];
    float h2[WH2 * WG];
    float pI[2][3][256];      // segment sums as the MFMA leaves them: [frame parity][gate][ws_tile(unit, utterance)]
    float pA[4][3][256];      // [segment][gate][...]
    float pC[4][256];         // [segment][ws_tile(gate * 4 + unit, utterance)]
    float pB[2][256];
    float pFl[2][8][2][16];   // the training forward's off-chain output layer: [frame parity][segment][tile][row in tile], owned utterance
    float fo[WG][WIN];        // predictions [utterance][row < 18]
    float w1i[3 * W1G];        // [gate][k][unit], gates 260 floats apart (240: gates 0 and 2 in the same banks for ws_I_own's lanes)
    float w2i[WH1 * 3 * WU2];  // [k / 16][k % 4][gate * 4 + unit][(k % 16) / 4]
    float w2h[WH2 * 3 * WU2];
    float fcw[WH2 * 16];       // output layer [k][row < 16] (an MFMA tile's B operand: lanes (k % 4, row) read 32 banks)
    float fcw2[2 * WH2];       // ... rows 16, 17: [row - 16][k] (16-byte reads of four k)
    float h2own[WH2];          // the owned utterance's column of the h2 image, as it arrives (ws_F / ws_F_late: rows 16, 17)
    float fcb[WIN];            // ... bias [row < 18]
    int sig[WNSIG];
    int dead;
    int dead_latch;  // the value every thread acts on at the end of a frame (read once, between two barriers)
    int same_xcd;
    int hello;       // the group's decision as this workgroup adopted it (ws_hello)
    // the distributed tail (predictor_wsd.h): every workgroup of a group serves every utterance of the group
    // (16-byte aligned: the tail reads coordinate PAIRS; at an 8-byte offset -- one counter fewer in `sig` did that -- the encoder
    //  lost 5 %)
    __attribute__((aligned(16))) double cbs[3][WNS][NDIM + 1];  // this workgroup's entries 32 m + slice of the books hi stage 1, hi stage 2, lo (18 doubles
                                   // apart: 16-byte reads of coordinate pairs, conflict-free; 17 apart was measured slower --
                                   // 17 eight-byte reads per distance instead of 9 reads: profiles/r05_ablations.txt)
    float pFa[8][WG * WPF];        // output-layer segment sums of all 16 utterances [segment][utterance * 20 + row]
    float rsa[WG][WIN];            // residuals [utterance][row]
    __attribute__((aligned(16))) double xs[WG][NDIM + 1];       // first-stage targets
    double xq2[WG][SURV][NDIM + 1];  // second-stage targets
    double dl[WG][WNS];            // a half-wave's 32 distances (local ranking, heads of the gathered lists)
    unsigned dh[WG][WNS];          // ... and their high words

#ifdef FPC_WS_PROF
    long long wprof[32], wlast, wlast_bg;  // diagnostic builds: cycles per stage, foreground [0..13) + [19..24), background [13..19)
#endif
};
#ifdef FPC_WS_PROF
#define WSTAMP(k)                                            \
    if (threadIdx.x == 0) {                                  \
        const long long now_ = __builtin_readcyclecounter(); \
        L.wprof[k] += now_ - L.wlast;                        \
        L.wlast = now_;                                      \
    }
#ifdef FPC_WS_PROF_TAIL
#define WBSTAMP(k)
#else
#ifndef FPC_WS_BSTAMP_WAVE
#define FPC_WS_BSTAMP_WAVE 1
#endif
#define WBSTAMP(k)                                           \
    if (threadIdx.x == WFGT + 64 * FPC_WS_BSTAMP_WAVE) {     \
        const long long now_ = __builtin_readcyclecounter(); \
        L.wprof[k] += now_ - L.wlast_bg;                     \
        L.wlast_bg = now_;                                   \
    }
#endif
#define WPROF_INIT()                                              \
    if (threadIdx.x == 0) {                                       \
        for (int k_ = 0; k_ < 32; ++k_) L.wprof[k_] = 0;          \
        L.wlast = L.wlast_bg = __builtin_readcyclecounter();      \
    }                                                             \
    __syncthreads();
#define WPROF_DUMP(frames)                                                                             \
    __syncthreads();                                                                                   \
    if (threadIdx.x < 32 && blockIdx.x == 8 * 5)                                                        \
        S.err[1 + threadIdx.x] = (unsigned)(L.wprof[threadIdx.x] / ((frames) > 0 ? (frames) : 1));
#else
#define WSTAMP(k)
#define WBSTAMP(k)
#define WPROF_INIT()
#define WPROF_DUMP(frames)
#endif

// what the training forward keeps per sample n = utterance * L + frame for the backward pass (TrainBufs, k_train_fwd):
// the activations of this workgroup's units for every utterance of the group; tanh outputs by the utterance's owner
struct WsSave {
    float *h1p, *r1, *z1, *n1, *hn1, *h1;         // [N][384]
    float *h2p, *r2, *z2, *n2, *hn2, *h2, *relu;  // [N][128]
    float* th;                                    // [N][18]
};
struct WsCtx {
    __amdgpu_buffer_rsrc_t rs;  // this group's granule block
    int Lf;                     // frames of the launch (the training forward's sample index)
    int slice, nu, b0;          // this workgroup's slice, valid utterances of the group, first utterance
    int own;                    // utterance of the group whose prediction this workgroup needs (-1: none)
    unsigned* err;
    unsigned long long limit, hello_limit;
    unsigned* dec;              // this group's decision word
    bool fast, withhold, withhold_hello;
    bool fallback;              // the group decided WS_FALLBACK: this launch leaves its utterances to the row-split launch behind it
};

// The end of a frame in the closed-loop kernels: has any wait of this frame been given up?  Every poll that fails sets
// L.dead first; the answer must be the same for every thread (they leave the frame loop together), so one thread latches
// the flag between two barriers (__syncthreads_or costs three barriers and a cross-lane reduction: ~900 cycles per use).
template <class LT>
__device__ __forceinline__ bool ws_frame_dead(LT& L, int tid) {
    lds_barrier();
    if (tid == 0) L.dead_latch = __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_barrier();
    return L.dead_latch != 0;
}
// a zero the compiler cannot see through: added to a lane index at the top of a frame's role function, it keeps the frame's
// address arithmetic inside the frame loop (hoisted out of it, the per-lane offsets of every unrolled access stay live across
// the encoder's searches and spill)
__device__ __forceinline__ int ws_opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
template <class LT>
__device__ __forceinline__ bool ws_dead(LT& L) {
    return __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
}
template <class LT>
__device__ __forceinline__ void ws_give_up(const WsCtx& X, LT& L) {
    status_or(X.err, FPC_ST_TIMEOUT);
    __hip_atomic_store(&L.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
template <int N, class LT>
__device__ __forceinline__ bool ws_poll(const WsCtx& X, LT& L, const int (&gi)[N], unsigned epoch, u32x4 (&v)[N]) {
    {  // (nothing wanted by any lane: no round trip)
        bool none = true;
#pragma unroll
        for (int j = 0; j < N; ++j) none &= gi[j] < 0;
        if (__all(none)) return true;
    }
    unsigned spins = 0;
    unsigned long long t0 = 0, last = 0;
#if FPC_WS_POLL_DELAY
    // (a poll issued at once reads the L2 before the partners' stores have landed there and costs a second round trip: the
    //  first poll waits for about the time a store takes to arrive)
    __builtin_amdgcn_s_sleep(FPC_WS_POLL_DELAY);
#endif
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
            if (t0 == 0 || now - last > spin_rearm_gap(X.limit)) t0 = now;  // (this wave was descheduled: await_granule)
            last = now;
            if (now - t0 > X.limit || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                ws_give_up(X, L);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");  // (the loads are re-issued every round)
    }
}

// ws_poll with work of the caller's between the first round's loads and their check (the work rides on the L2 round trip)
template <int N, class F>
__device__ __forceinline__ bool ws_poll_over(const WsCtx& X, WsLds& L, const int (&gi)[N], unsigned epoch, u32x4 (&v)[N], F&& mid) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(X.rs, (gi[j] < 0 ? 0 : gi[j]) * 16, 0, 16);
    mid();
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) ok &= gi[j] < 0 || v[j].x == epoch;
    if (__all(ok)) return true;
    asm volatile("" ::: "memory");
    return ws_poll<N>(X, L, gi, epoch, v);
}

__device__ __forceinline__ bool ws_role(int ngroups, int& group, int& slice) {
    const int i = blockIdx.x, x = i % 8, m = i / 8;
    slice = m % WNS;
    group = (m / WNS) * 8 + x;
    return group < ngroups;
}

// per-wave constants of the matrix products (registers for the whole launch)
struct WsRegs {
    float wA[4][24];   // background wave bw = 1..3: W1h as MFMA B operands -- the three gate tiles of input segment bw - 1 and gate
                       // tile bw - 1 of segment 3, 24 k-steps each (55 kB per workgroup that LDS has no room for next to the
                       // encoder's distributed searches; the chain's other weights stay in LDS)
    float bA[3];       // background wave 1 (input segment 0): b_hh of GRU1
    float bB;          // b_hh of GRU2 (background wave 0)
    float bI;          // foreground wave w < 3: b_ih of GRU1, gate w (the forward's first frame)
    float bIo;         // foreground wave w, column c < 9 of its own input tile: gate c / 3, unit 3 w + c % 3 (ws_I_own)
    float bI3[3];      // background wave 0: all three gates (teacher-forced forward: I(t+1) off the chain)
    float bC;          // foreground wave 0: b_ih of GRU2
    float bF[2];       // foreground wave 0: output bias, tiles 0 and 1
    float bF16[2];     // output bias of rows 16, 17 (the distributed tail's output layer: every foreground thread)
};

// copies this workgroup's weight slices to LDS / registers (all threads; no barrier inside)
__device__ __forceinline__ void ws_load_weights(const PredDev& P, WsLds& L, WsRegs& R, int slice, int tid) {
    const int wave = tid >> 6, lane = tid & 63, c = lane & 15;
    for (int i = tid; i < 3 * WIN * WU1; i += NT) {
        const int g = i / (WIN * WU1), r = i - g * WIN * WU1, k = r / WU1, u = r - k * WU1;
        L.w1i[g * W1G + r] = P.w1i[(size_t)k * 3 * WH1 + g * WH1 + WU1 * slice + u];
    }
    for (int i = tid; i < WH1 * 12; i += NT) {
        const int k = i / 12, r = i - k * 12, g = r / WU2, u = r - g * WU2;
        L.w2i[((k / 16) * 4 + (k & 3)) * 48 + r * 4 + ((k & 15) >> 2)] = P.w2i[(size_t)k * 3 * WH2 + g * WH2 + WU2 * slice + u];
    }
    for (int i = tid; i < WH2 * 12; i += NT) {
        const int k = i / 12, r = i - k * 12, g = r / WU2, u = r - g * WU2;
        L.w2h[i] = P.w2h[(size_t)k * 3 * WH2 + g * WH2 + WU2 * slice + u];
    }
    for (int i = tid; i < WH2 * WFC; i += NT) {
        const int k = i / WFC, r = i - k * WFC;
        if (r < 16)
            L.fcw[k * 16 + r] = P.fcw[i];
        else
            L.fcw2[(r - 16) * WH2 + k] = P.fcw[i];
    }
    if (tid < WIN) L.fcb[tid] = tid < WFC ? P.fcb[tid] : 0.0f;
    const int fw = wave, bw = wave - WFG;
    R.bA[0] = R.bA[1] = R.bA[2] = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 24; ++j) R.wA[g][j] = 0.0f;
    R.bB = R.bI = R.bIo = R.bC = 0.0f;
    R.bI3[0] = R.bI3[1] = R.bI3[2] = 0.0f;
    R.bF[0] = P.fcb[c];
    R.bF[1] = P.fcb[16 + (lane & 1)];  // (rows 16, 17 = columns 0, 1 of the second tile; its other columns are dropped)
    R.bF16[0] = P.fcb[16];
    R.bF16[1] = P.fcb[17];
    if (wave >= WFG) {
        if (bw == 1 && c < WU1) {  // (input segment 0)
#pragma unroll
            for (int g = 0; g < 3; ++g) R.bA[g] = P.b1h[g * WH1 + WU1 * slice + c];
        }
        if (bw >= 1) {
            const int sg = bw - 1, q = lane >> 4;
#pragma unroll
            for (int j = 0; j < 24; ++j) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    R.wA[g][j] = c < WU1 ? P.w1h[(size_t)(96 * sg + 4 * j + q) * 3 * WH1 + g * WH1 + WU1 * slice + c] : 0.0f;
                R.wA[3][j] = c < WU1 ? P.w1h[(size_t)(96 * 3 + 4 * j + q) * 3 * WH1 + sg * WH1 + WU1 * slice + c] : 0.0f;
            }
        }
        if (bw == 0 && c < 12) R.bB = P.b2h[(c / WU2) * WH2 + WU2 * slice + (c % WU2)];
        if (bw == 0 && c < WU1) {
#pragma unroll
            for (int g = 0; g < 3; ++g) R.bI3[g] = P.b1i[g * WH1 + WU1 * slice + c];
        }
    } else {
        if (fw < 3 && c < WU1) R.bI = P.b1i[fw * WH1 + WU1 * slice + c];
        if (c < 9) R.bIo = P.b1i[(c / 3) * WH1 + WU1 * slice + 3 * fw + c % 3];
        if (fw == 0 && c < 12) R.bC = P.b2i[(c / WU2) * WH2 + WU2 * slice + (c % WU2)];
    }
}

// Once per launch (all threads; ends with barriers): are all 32 workgroups of the group RESIDENT?  The frame loop needs
// them all at once (each spins for the others' granules), and nothing guarantees it: another kernel -- of another process,
// or this library's own vocoder launch on a side stream -- may hold CUs for as long as it runs.  So the group DECIDES, once,
// before anything is computed or written:
//   * every workgroup publishes a hello granule {tag, XCD id} and polls the other 31, for at most `hello_limit` (10 ms);
//   * the first workgroup to see all 31 tries WS_GO on the group's decision word, the first to run out of patience tries
//     WS_FALLBACK (one compare-and-swap from 0: the first attempt wins, every workgroup adopts the winner);
//   * GO: all 32 have published, hence are resident and stay so (a workgroup is not preempted out of a running launch
//     short of a queue preemption, which the spins' clocks allow for): a workgroup that lost patience goes back to waiting
//     -- for the frame loop's own bound now (1 s), a real failure if it expires (FPC_ERR_TIMEOUT as before);
//   * FALLBACK: every workgroup of the group returns at once -- also those dispatched later, which find the word set --
//     and the launch queued behind this one on the stream (the row-split kernels, one workgroup per utterance, no partner
//     to wait for: SplitArgs.only) serves exactly the groups that decided so.  Slower, never wrong, never a timeout.
// Also: X.fast (all 32 on one XCD), or L.dead when the handle has failed already.
__device__ __forceinline__ unsigned ws_dec_load(const WsCtx& X) {
    return __hip_atomic_load((gu32*)X.dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ws_dec_try(const WsCtx& X, unsigned want) {  // the decision after this attempt (whole wave)
    unsigned seen = 0u;
    if ((threadIdx.x & 63) == 0)  // ONE compare-and-swap per workgroup (64 lanes' worth on one word took 0.16 ms per launch)
        (void)__hip_atomic_compare_exchange_strong((gu32*)X.dec, &seen, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
    return seen == 0u ? want : seen;
}
template <class LT>
__device__ __forceinline__ void ws_hello(WsCtx& X, LT& L, const WsArgs& S, int tid) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xfu) + 1u;
    const unsigned tag = 0xffffffffu;  // (no frame's epoch)
    if (tid == 0) {
        L.same_xcd = 1;
        L.hello = 0;
        if (!X.withhold_hello && !ws_dead(L) && ws_dec_load(X) != WS_FALLBACK) {  // general path: nothing is known about the placement yet
            const u32x4 w = {tag, xcc, 0u, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, (WOFF_HELLO + X.slice) * 16, 0, 16);
        }
    }
    __syncthreads();
    if (tid < 64 && !ws_dead(L)) {
        const int gi = (tid < WNS && tid != X.slice) ? WOFF_HELLO + tid : -1;
        unsigned dec = 0u;
        bool patient = false;  // the group has decided GO: the partners exist, wait for them like the frame loop does
        for (;;) {
            unsigned spins = 0;
            unsigned long long t0 = 0, last = 0;
            bool all = false, out = false;
            for (;;) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(X.rs, (gi < 0 ? 0 : gi) * 16, 0, 16);
                dec = ws_dec_load(X);
                // (the test hook's workgroup stands for one that is not resident: it decides nothing, it follows)
                all = !X.withhold_hello && __all(gi < 0 || v.x == tag);
                if (all) {
                    if (gi >= 0 && v.y != xcc) L.same_xcd = 0;
                    break;
                }
                if (dec == WS_FALLBACK) break;
                if (dec == WS_GO) patient = true;
                if ((++spins & 63u) == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    const unsigned long long lim = (patient || X.withhold_hello) ? X.limit : X.hello_limit;
                    if (t0 == 0 || now - last > spin_rearm_gap(lim)) t0 = now;
                    last = now;
                    if (now - t0 > lim || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                        out = true;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            if (all) {
                dec = ws_dec_try(X, WS_GO);
            } else if (out && !patient && !X.withhold_hello) {
                dec = ws_dec_try(X, WS_FALLBACK);
                if (dec == WS_GO) {  // (somebody saw all 32 a moment ago: they are there)
                    patient = true;
                    continue;
                }
            } else if (out) {  // the partners were there and stopped answering (or the handle failed meanwhile): a real failure
                ws_give_up(X, L);
            }
            break;
        }
        if (tid == 0) L.hello = (int)dec;
    }
    __syncthreads();
    X.fallback = L.hello == (int)WS_FALLBACK;
    X.fast = L.same_xcd != 0 && !ws_dead(L) && S.no_fast == 0;
}

// ---- the matrix products (one wave each; `lane` = c + 16 q) ----
__device__ __forceinline__ f32x4ws ws_mfma(float a, float b, f32x4ws c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// A product tile in LDS: [column c][16 utterances], the four-utterance blocks of a column XOR-swizzled by the column.  The
// accumulator of lane (c, q) = utterances 4 q .. 4 q + 3 of column c goes out as ONE 16-byte store: the 8 lanes of a store
// group (c = 0..7 of one q) hit 8 different bank quads (a plain 16-float pitch: two, a 4-way conflict on every store of every
// product); a gate thread's 4-byte read covers two columns of different parity x 16 utterances = 32 different banks.
#ifndef FPC_WS_TILESWZ
#define FPC_WS_TILESWZ 1
#endif
__device__ __forceinline__ int ws_tile(int c, int u) { return FPC_WS_TILESWZ ? c * 16 + (u ^ (((c >> 1) & 3) << 2)) : c * 16 + u; }
// The input image x: [k][utterance] with rows 17 floats apart.  The closed loops write it one utterance per half-wave, lane m =
// row m, the forward 20 consecutive floats of an utterance per 20 lanes: at a 16-float pitch one utterance's rows live in two
// banks (9- and 16-way conflicts), at 17 in 20 different ones; as an MFMA operand lane (utterance, k % 4) still reads at a
// constant offset per k-step.  (An XOR swizzle of a 16-float pitch also removes the conflicts but needs the address of every
// k-step formed separately, on the closed loops' chain: encode 4.05 against 4.01 ms, same box.)
__device__ __forceinline__ int ws_xi(int k, int u) { return k * WXP + u; }
__device__ __forceinline__ void ws_put(float* p, int lane, const f32x4ws& acc) {
    *reinterpret_cast<f32x4ws*>(&p[ws_tile(lane & 15, 4 * (lane >> 4))]) = acc;  // utterances 4 q .. 4 q + 3 of column c
}
// A = W1h h1 on the background waves 1-3 (wave 0 of the role does no matrix work: it shares its SIMD with the foreground
// wave that evaluates the GRU2 gates and polls the hops).  The f32 MFMA runs at the vector ALU's rate and keeps the SIMD's
// vector issue busy while it does -- whatever else runs on that SIMD meanwhile takes 3-8x as long (measured: the chain's own
// product 1.0k -> 3.4k cycles, the gates 0.8k -> 2.7k, a poll loop 1.4k -> 3.3k) -- so the 12 (segment, gate tile) units of
// this product go to three SIMDs, 4 units = 96 MFMAs each: wave bw takes the three gate tiles of input segment bw - 1 and gate
// tile bw - 1 of segment 3, four independent accumulators per k-step, state operands read up front, weights one step ahead.
// SPLIT (the teacher-forced forward, whose frame loop runs THROUGH this product: gates1 -> hop 1 -> A(t+1) -> gates1): the
// first unit as ONE dependent chain while the foreground's C(t) uses the same matrix pipe (a single chain leaves every second
// slot to the other wave; the four interleaved chains took four of five and stretched C from 1.0k to 3.4k cycles), the
// other three units interleaved once C(t) is through (`mid` waits for it).
template <bool SPLIT = false, class F>
__device__ __forceinline__ bool ws_A(WsLds& L, const WsRegs& R, int bw, int lane, F&& mid) {
    const int sg = bw - 1;  // 0..2
    f32x4ws acc[4];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4ws{R.bA[g], R.bA[g], R.bA[g], R.bA[g]};
    acc[3] = f32x4ws{0.f, 0.f, 0.f, 0.f};
    const float* hs = L.h1 + 96 * sg * WG + lane;
    const float* h3 = L.h1 + 96 * 3 * WG + lane;
    float a[24], b[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) {
        a[j] = hs[64 * j];
        b[j] = h3[64 * j];
    }
    if (SPLIT) {
#pragma unroll
        for (int j = 0; j < 24; ++j) acc[0] = ws_mfma(a[j], R.wA[0][j], acc[0]);
        if (!mid()) return false;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
#pragma unroll
            for (int g = 1; g < 3; ++g) acc[g] = ws_mfma(a[j], R.wA[g][j], acc[g]);
            acc[3] = ws_mfma(b[j], R.wA[3][j], acc[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = ws_mfma(a[j], R.wA[g][j], acc[g]);
            acc[3] = ws_mfma(b[j], R.wA[3][j], acc[3]);
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) ws_put(L.pA[sg][g], lane, acc[g]);
    ws_put(L.pA[3][sg], lane, acc[3]);
    return true;
}
__device__ __forceinline__ void ws_A(WsLds& L, const WsRegs& R, int bw, int lane) {
    (void)ws_A<false>(L, R, bw, lane, [] { return true; });
}
// B(t) = W2h h2(t-1), both input segments of 64 (two independent chains), by background wave 0 while hop 1 is in the air:
// its SIMD partner, the foreground's wave 0, only sleeps on a counter then
__device__ __forceinline__ void ws_B(WsLds& L, const WsRegs& R, int lane) {
    const int c = lane & 15, q = lane >> 4, cc = c < 12 ? c : 11;
    f32x4ws a0 = {R.bB, R.bB, R.bB, R.bB}, a1 = {0.f, 0.f, 0.f, 0.f};
    const float* hs = L.h2 + lane;
    const float* ws = L.w2h + q * 12 + cc;
    float h0[16], h1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        h0[j] = hs[64 * j];
        h1[j] = hs[64 * (16 + j)];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        a0 = ws_mfma(h0[j], ws[48 * j], a0);
        a1 = ws_mfma(h1[j], ws[48 * (16 + j)], a1);
    }
    ws_put(L.pB[0], lane, a0);
    ws_put(L.pB[1], lane, a1);
}
// I = W1i x (one segment of 20 inputs): gate tile `g` = foreground wave (no background matrix work runs at the start of a frame)
__device__ __forceinline__ void ws_I(WsLds& L, float bias, int g, int lane, int buf) {
    const int c = lane & 15, q = lane >> 4, cc = c < WU1 ? c : WU1 - 1;
    f32x4ws acc = {bias, bias, bias, bias};
    const float* ws = L.w1i + g * W1G + q * WU1 + cc;
#pragma unroll
    for (int j = 0; j < WIN / 4; ++j) acc = ws_mfma(L.x[ws_xi(4 * j + q, c)], ws[4 * WU1 * j], acc);  // (ws_I: the forward only)
    ws_put(L.pI[buf][g], lane, acc);
}
// The closed loop's form of I (x(t) is the previous frame's result: I is on the chain): foreground wave fw evaluates the
// columns ITS gate threads need -- 3 gates x units 3 fw .. 3 fw + 2 = 9 columns of one tile -- so the gates follow in the same
// wave without a barrier of the role.  Sums at p[ws_tile(gate * 3 + unit - 3 fw, utterance)], p = the wave's 144 floats.
__device__ __forceinline__ void ws_I_own(WsLds& L, float bias, int fw, int lane, float* p) {
    const int c = lane & 15, q = lane >> 4, cc = c < 9 ? c : 8;
    f32x4ws acc = {bias, bias, bias, bias};
    const float* ws = L.w1i + (cc / 3) * W1G + q * WU1 + 3 * fw + cc % 3;
#pragma unroll
    for (int j = 0; j < WIN / 4; ++j) acc = ws_mfma(L.x[ws_xi(4 * j + q, c)], ws[4 * WU1 * j], acc);
    if (c < 9) *reinterpret_cast<f32x4ws*>(&p[ws_tile(c, 4 * q)]) = acc;
}
// C = W2i h1: foreground wave fw = input segment of 96 (one chain of 24 dependent MFMAs: operands read ahead)
// (ws_C_weights: this lane's B operands, read while hop 1 is still in the air)
__device__ __forceinline__ void ws_C_weights(WsLds& L, int fw, int lane, f32x4ws (&w)[6]) {
    const int c = lane & 15, q = lane >> 4, cc = c < 12 ? c : 11;
    const f32x4ws* ws = reinterpret_cast<const f32x4ws*>(L.w2i) + (6 * fw * 4 + q) * 12 + cc;  // + 48 per block of 16 inputs
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) w[kb] = ws[48 * kb];
}
__device__ __forceinline__ void ws_C(WsLds& L, const WsRegs& R, int fw, int lane, const f32x4ws (&w)[6]) {
    f32x4ws acc = {R.bC, R.bC, R.bC, R.bC};
    const float* hs = L.h1 + 96 * fw * WG + lane;
    float a[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) a[j] = hs[64 * j];
#pragma unroll
    for (int j = 0; j < 24; ++j) acc = ws_mfma(a[j], w[j >> 2][j & 3], acc);
    ws_put(L.pC[fw], lane, acc);
}
// rows 16, 17 of the output layer for the owned utterance: 16 lanes = (row, input segment of 16), k-ordered fmaf chains from
// the bias (segment 0) or 0 -- what an f32 MFMA accumulates.  Operands by 16-byte reads from L.h2own (the utterance's column
// of the state image, copied as it arrives: in the image its 128 values live in TWO banks, and round 4's reads from there
// were 8-way conflicts, 40 % of the forward's conflict cycles) and from the weights transposed to [row][k].
__device__ __forceinline__ void ws_F_rows(WsLds& L, const WsRegs& R, float (&pF)[8][2][16], int lane) {
    const int row = lane & 1, rsg = (lane >> 1) & 7;
    f32x4ws h[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = *reinterpret_cast<const f32x4ws*>(&L.h2own[16 * rsg + 4 * i]);
        w[i] = *reinterpret_cast<const f32x4ws*>(&L.fcw2[row * WH2 + 16 * rsg + 4 * i]);
    }
    float a = rsg == 0 ? R.bF[1] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) a = fmaf(h[i][k] > 0.0f ? h[i][k] : 0.0f, w[i][k], a);
    if (lane < 16) pF[rsg][1][row] = a;
}
// The teacher-forced forward's output layer, off the chain: nothing of frame t + 1 depends on the prediction of frame t, so
// background waves 1-3 evaluate it behind A(t + 1), when hop 2 of frame t is over -- h2(t) stays whole in LDS until the GRU2
// gates of frame t + 1, which come behind hop 1 of that frame, i.e. behind these waves' own gather -- while the foreground
// starts the next frame (its SIMDs run no matrix product then).
// Wave w3 = 0, 1, 2: input segments {0, 1, 2}, {3, 4, 5}, {6, 7} (independent chains of four MFMAs), rows 16, 17 on wave 2.
// (One wave for all eight segments -- background wave 0, behind a hop-2 gather of its own -- was measured: that wave becomes
// the frame's pole, 15.8k cycles; a second MFMA tile for rows 16, 17: the waves come late to hop 1: profiles/r05_ablations.txt)
__device__ __forceinline__ void ws_F_late(WsLds& L, const WsRegs& R, float (&pF)[8][2][16], int w3, int lane, int own) {
    const int c = lane & 15, q = lane >> 4;
    const int s0 = 3 * w3, ns = w3 == 2 ? 2 : 3;
    float hv[3][4], wv[3][4];
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
        const int sg = s0 + (s2 < ns ? s2 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h = L.h2[(16 * sg + 4 * j) * WG + lane];
            hv[s2][j] = h > 0.0f ? h : 0.0f;
            wv[s2][j] = L.fcw[(16 * sg + 4 * j + q) * 16 + c];
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
        if (s2 < ns) {
            const int sg = s0 + s2;
            const float b0 = sg == 0 ? R.bF[0] : 0.0f;
            f32x4ws a0 = {b0, b0, b0, b0};
#pragma unroll
            for (int j = 0; j < 4; ++j) a0 = ws_mfma(hv[s2][j], wv[s2][j], a0);
            if (q == (own >> 2)) {
                const int r = own & 3;
                pF[sg][0][c] = r == 0 ? a0[0] : (r == 1 ? a0[1] : (r == 2 ? a0[2] : a0[3]));
            }
        }
    }
    if (w3 == 2) ws_F_rows(L, R, pF, lane);
}
// ... and its last step for row `row` < 18 (the segment sums are complete: every wave of ws_F_late has passed a barrier since)
template <bool TANH_ONLY = false>
__device__ __forceinline__ float ws_F_out(const float (&pF)[8][2][16], int row) {
    const int tile = row >> 4, o = row & 15;
    const float acc = ((pF[0][tile][o] + pF[1][tile][o]) + (pF[2][tile][o] + pF[3][tile][o])) +
                      ((pF[4][tile][o] + pF[5][tile][o]) + (pF[6][tile][o] + pF[7][tile][o]));
    const float tt = fpc_tanhf(acc);
    return TANH_ONLY ? tt : tt + tt;  // the "dual" FC is the same Linear summed twice (wavernn.py:89-92)
}

// hop 1 gather by six waves (foreground 1-3, background 1-3): p < 384, six granules each; the waves of SIMD 0 have other
// work meanwhile (background wave 0 computes B(t), the foreground's wave 0 must not take issue slots from it)
template <class F>
__device__ __forceinline__ bool ws_gather1(const WsCtx& X, WsLds& L, int p, unsigned epoch, bool guard_A, int t, F&& mid) {
    constexpr int NG = (WNS * WQ1 + 383) / 384;
    const int set = WOFF_H1 + (t & 1) * WNS * WQ1;
    int gi[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = p + 384 * j;
        gi[j] = (i < WNS * WQ1 && i / WQ1 != X.slice) ? set + i : -1;
    }
    u32x4 v[NG];
    if (!ws_poll_over<NG>(X, L, gi, epoch, v, mid)) return false;
    // (a background wave may be here before its neighbours have finished A(t) on the old image)
    if (guard_A && !df_wait(&L.sig[WSIG_A], 3 * (t + 1), &L.dead)) return false;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        if (gi[j] >= 0) {
            float* d = L.h1 + 3 * (gi[j] - set);  // granule e of slice s holds values 3 e .. 3 e + 2 of that slice
            d[0] = __uint_as_float(v[j].y);
            d[1] = __uint_as_float(v[j].z);
            d[2] = __uint_as_float(v[j].w);
        }
    }
    return true;
}
__device__ __forceinline__ bool ws_gather1(const WsCtx& X, WsLds& L, int p, unsigned epoch, bool guard_A, int t) {
    return ws_gather1(X, L, p, epoch, guard_A, t, [] {});
}
// hop 2 gather: SHARE = 2: by the two waves of SIMD 0 (foreground wave 0 and background wave 0: p < 128, six granules each);
// SHARE = 1: by background wave 0 alone (p < 64, eleven granules: the teacher-forced forward, whose foreground does not need
// h2(t) and whose frame is as long as its wave 0's path).  The owned utterance's values also go to L.h2own (ws_F_rows).
template <int SHARE>
__device__ __forceinline__ bool ws_gather2(const WsCtx& X, WsLds& L, int p, unsigned epoch) {
    // granule i = p + NP j of the hop's WNS x WQ2: consecutive lanes take consecutive granules (their image writes are
    // consecutive floats: conflict-free; a slice per group of four lanes -- slices are 64 floats apart -- was an 8-way conflict
    // on every write); slice = i / 22 by a multiply and a shift, exact for i < 1 489 (22 x 2 979 = 65 538)
    constexpr int NP = 64 * SHARE, NG = (WNS * WQ2 + NP - 1) / NP;
    static_assert(WQ2 == 22 && WNS * WQ2 < 1489, "slice of a granule: (i * 2979) >> 16");
    const int set = WOFF_H2 + (int)((epoch - 1u) & 1u) * WNS * WQ2;  // (epoch = frame + 1)
    int gi[NG], sl[NG], e[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = p + NP * j;
        sl[j] = (i * 2979) >> 16;
        e[j] = i - sl[j] * WQ2;
        gi[j] = (i < WNS * WQ2 && sl[j] != X.slice) ? set + i : -1;
    }
    u32x4 v[NG];
    if (!ws_poll<NG>(X, L, gi, epoch, v)) return false;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        if (gi[j] >= 0) {
            const int w0 = 3 * e[j];  // the granule's values w0 .. w0 + 2 of that slice, value w = unit * 16 + utterance
            float* d = L.h2 + sl[j] * WV2 + w0;
            const float v0 = __uint_as_float(v[j].y), v1 = __uint_as_float(v[j].z), v2 = __uint_as_float(v[j].w);
            d[0] = v0;
            if (w0 + 1 < WV2) d[1] = v1;
            if (w0 + 2 < WV2) d[2] = v2;
            // (three consecutive values = three consecutive utterances: at most one is the owned one)
            const int i = (X.own - w0) & 15;
            if (X.own >= 0 && i < 3 && w0 + i < WV2) L.h2own[WU2 * sl[j] + ((w0 + i) >> 4)] = i == 0 ? v0 : (i == 1 ? v1 : v2);
        }
    }
    return true;
}

// FOREGROUND, frame t: L.x = x(t) -> L.fo[slice] on the owners (written by threads ft < 18: a reader in another wave needs a
// barrier first), states in L.h1 / L.h2; false: the launch is dead
// EARLY_I (the teacher-forced forward): I(t) has been computed one frame ahead by background wave 0 (k_forward_ws)
__device__ __forceinline__ void wsd_F(WsLds& L, const WsRegs& R, int fw, int lane, int ft);  // predictor_wsd.h
__device__ __forceinline__ void wsd_receive_tail(const WsCtx& X, WsLds& L, const PredDev& P, const CbDev& C, const int* __restrict__ idx,
                                                 float* __restrict__ c_out, int* bad, int Lf, int frame, float pv, int tid);  // predictor_wsd.h
// FC_ALL (the distributed encoder tail): the output layer's segment sums of all 16 utterances, no prediction formed here
// FC_LATE (the teacher-forced forward): no output layer here -- the background evaluates it off the chain (ws_background)
// SAVE (the training forward): the gates' activations of every sample go to *sv
template <bool EARLY_I = false, bool FC_ALL = false, bool FC_LATE = false, bool SAVE = false>
__device__ __forceinline__ bool ws_foreground(const WsCtx& X, WsLds& L, const WsRegs& R, int t, int ft0, const WsSave* sv = nullptr) {
    const int ft = ft0 + (FPC_WS_HOIST && EARLY_I ? 0 : ws_opaque_zero());
    const int fw = ft0 >> 6, lane = ft & 63;
    const unsigned epoch = (unsigned)t + 1u;
    // (EARLY_I: the caller has waited for I(t) AND A(t) in one poll loop -- k_forward_ws at the top of its frame loop, before x(t + 1)
    //  goes into the image)
    if (!EARLY_I) {
        ws_I_own(L, R.bIo, fw, lane, &L.pI[0][0][0] + 144 * fw);
        WSTAMP(0)
        if (!df_wait(&L.sig[WSIG_A], 3 * (t + 1), &L.dead)) return false;  // A(t): prologue, then one round per frame
    }
    WSTAMP(1)
    // ---- GRU1 gates of this workgroup's 12 units x 16 utterances: 48 values per wave; torch.nn.GRU rows [r; z; n]
    float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        const int base = X.slice * WV1 + 48 * fw;
        if (lane < 48) {
            const int j = lane >> 4, u = lane & 15;
            const int v = ws_tile(3 * fw + j, u);  // unit 3 fw + j of the workgroup's 12, utterance u
            const float ghr = (L.pA[0][0][v] + L.pA[1][0][v]) + (L.pA[2][0][v] + L.pA[3][0][v]);
            const float ghz = (L.pA[0][1][v] + L.pA[1][1][v]) + (L.pA[2][1][v] + L.pA[3][1][v]);
            const float ghn = (L.pA[0][2][v] + L.pA[1][2][v]) + (L.pA[2][2][v] + L.pA[3][2][v]);
            // (I: by gate tile a frame ahead, or this wave's own tile -- columns gate * 3 + j -- read back by the wave that wrote it)
            const float* pw = &L.pI[0][0][0] + 144 * fw;
            const float ir = EARLY_I ? L.pI[t & 1][0][v] : pw[ws_tile(j, u)];
            const float iz = EARLY_I ? L.pI[t & 1][1][v] : pw[ws_tile(3 + j, u)];
            const float in = EARLY_I ? L.pI[t & 1][2][v] : pw[ws_tile(6 + j, u)];
            const float r = fpc_sigmoidf(ir + ghr);
            const float z = fpc_sigmoidf(iz + ghz);
            const float n = fpc_tanhf(fmaf(r, ghn, in));
            const float hp = L.h1[base + lane];
            const float hn = fmaf(z, hp - n, n);
            L.h1[base + lane] = hn;
            if (SAVE) sa[0] = hp, sa[1] = r, sa[2] = z, sa[3] = n, sa[4] = ghn, sa[5] = hn;
        }
        // (the same wave reads what it has just written: one in-order LDS queue per wave)
        if (lane < 16) {
            const float* s = &L.h1[base + 3 * lane];
            ws_store(X, WOFF_H1 + (t & 1) * WNS * WQ1 + X.slice * WQ1 + 16 * fw + lane, epoch, s[0], s[1], s[2]);
        }
    }
    df_signal(&L.sig[WSIG_P1]);  // (the background starts polling now, not before)
    // the training forward's saved activations: BEHIND the publication (a wave's vector-memory instructions leave in order:
    // six stores in front of the granule are six stores' worth of issue time on the hop)
    if (SAVE && lane < 48 && (lane & 15) < X.nu) {
        const size_t o = ((size_t)(X.b0 + (lane & 15)) * X.Lf + t) * WH1 + WU1 * X.slice + 3 * fw + (lane >> 4);
        sv->h1p[o] = sa[0];
        sv->r1[o] = sa[1];
        sv->z1[o] = sa[2];
        sv->n1[o] = sa[3];
        sv->hn1[o] = sa[4];
        sv->h1[o] = sa[5];
    }
    WSTAMP(2)
    if (fw != 0) {
        if (!ws_gather1(X, L, ft - 64, epoch, false, t)) return false;
        df_signal(&L.sig[WSIG_H1]);
    }
    WSTAMP(3)
    f32x4ws wC[6];
    ws_C_weights(L, fw, lane, wC);
    if (!df_wait(&L.sig[WSIG_H1], 6 * (t + 1), &L.dead)) return false;  // h1(t) whole in LDS
    WSTAMP(4)
    ws_C(L, R, fw, lane, wC);
    df_signal(&L.sig[WSIG_C]);  // (closed loops: A(t+1) starts behind C(t) -- side by side on one matrix pipe the chain's product took 3.4k cycles)
    WSTAMP(5)
    // the GRU2 gates need all four segments of C(t) and B(t); only their wave waits (one poll loop for both counters; round 4
    // had a barrier of the role and a second wait here: two LDS round trips on every wave)
    if (fw == 0 && !df_wait2(&L.sig[WSIG_C], WFG * (t + 1), &L.sig[WSIG_B], t + 1, &L.dead)) return false;
    WSTAMP(6)
    if (fw == 0) {  // GRU2 gates: 4 units x 16 utterances
        const int base = X.slice * WV2;
        const int v = lane;
        // columns gate * 4 + unit: the swizzle of column 4 + unit is that of column unit with bit 3 flipped, of column 8 + unit the same
        const int vr = ws_tile(v >> 4, v & 15), vz = FPC_WS_TILESWZ ? (vr ^ 8) + 64 : vr + 64, vn = vr + 128;
        const float gir = (L.pC[0][vr] + L.pC[1][vr]) + (L.pC[2][vr] + L.pC[3][vr]);
        const float giz = (L.pC[0][vz] + L.pC[1][vz]) + (L.pC[2][vz] + L.pC[3][vz]);
        const float gin = (L.pC[0][vn] + L.pC[1][vn]) + (L.pC[2][vn] + L.pC[3][vn]);
        const float ghr = L.pB[0][vr] + L.pB[1][vr];
        const float ghz = L.pB[0][vz] + L.pB[1][vz];
        const float ghn = L.pB[0][vn] + L.pB[1][vn];
        const float r = fpc_sigmoidf(gir + ghr);
        const float z = fpc_sigmoidf(giz + ghz);
        const float n = fpc_tanhf(fmaf(r, ghn, gin));
        const float hp = L.h2[base + v];
        const float hn = fmaf(z, hp - n, n);
        L.h2[base + v] = hn;
        if ((v & 15) == X.own) L.h2own[WU2 * X.slice + (v >> 4)] = hn;
        if (lane < WQ2) {
            const int i0 = 3 * lane, i1 = i0 + 1 < WV2 ? i0 + 1 : WV2 - 1, i2 = i0 + 2 < WV2 ? i0 + 2 : WV2 - 1;
            ws_store(X, WOFF_H2 + (t & 1) * WNS * WQ2 + X.slice * WQ2 + lane, epoch, L.h2[base + i0], L.h2[base + i1], L.h2[base + i2]);
        }
        df_signal(&L.sig[WSIG_P2]);
        if (!SAVE && sv != nullptr && sv->relu != nullptr && (v & 15) < X.nu)  // the plain forward: what k_out_layer reads
            sv->relu[((size_t)(X.b0 + (v & 15)) * X.Lf + t) * WH2 + WU2 * X.slice + (v >> 4)] = hn > 0.0f ? hn : 0.0f;
        if (SAVE && (v & 15) < X.nu) {  // (behind the publication, as GRU1's)
            const size_t o = ((size_t)(X.b0 + (v & 15)) * X.Lf + t) * WH2 + WU2 * X.slice + (v >> 4);
            sv->h2p[o] = hp;
            sv->r2[o] = r;
            sv->z2[o] = z;
            sv->n2[o] = n;
            sv->hn2[o] = ghn;
            sv->h2[o] = hn;
            sv->relu[o] = hn > 0.0f ? hn : 0.0f;
        }
    }
    WSTAMP(7)
    if (fw == 0) {
        if (!ws_gather2<2>(X, L, lane, epoch)) return false;
        df_signal(&L.sig[WSIG_H2]);
    }
    WSTAMP(8)
    // h2(t) whole in LDS: for the output layer of the closed loops.  The teacher-forced forward's foreground needs nothing of
    // it (the background evaluates the output layer and B(t+1) = W2h h2(t), behind its own waits for both halves of the
    // gather; the granule sets of the two frame parities make the next frame's publish independent of this hop: WOFF_H1)
    if (FC_LATE) return true;  // (a launch that died shows at the next frame's first wait)
    if (!df_wait(&L.sig[WSIG_H2], 2 * (t + 1), &L.dead)) return false;
    WSTAMP(9)
    if (FC_ALL) wsd_F(L, R, fw, lane, ft);  // (the closed loops: every workgroup evaluates the layer for all 16 utterances)
    WSTAMP(11)
    return !ws_dead(L);
}

// BACKGROUND, frame t: wave 0: B(t) and half of hop 2's gather; waves 1-3: half of hop 1's gather, then A(t+1)
// FC_LATE: y_late = the owned utterance's output row of frame t
template <bool EARLY_I = false, bool FC_LATE = false, bool TANH_ONLY = false>
__device__ __forceinline__ bool ws_background(const WsCtx& X, WsLds& L, const WsRegs& R, int t, bool last, int bt,
                                              float* y_late = nullptr) {
    const int bw = bt >> 6, lane = (bt + (FPC_WS_HOIST && EARLY_I ? 0 : ws_opaque_zero())) & 63;
    if (!df_wait(&L.sig[WSIG_P1], WFG * (t + 1), &L.dead)) return false;
    WBSTAMP(13)
    if (bw == 0) {  // the two waves of SIMD 0 run no matrix product beside the chain: B(t) under hop 1, then hop 2's gather
        if (!df_wait(&L.sig[WSIG_H2], 2 * t, &L.dead)) return false;  // h2(t-1) whole (the foreground's half of the gather too)
        ws_B(L, R, lane);
        df_signal(&L.sig[WSIG_B]);
        WBSTAMP(14)
        if (EARLY_I) {  // the next frame's input product, still under hop 1 (x(t+1) has been in LDS since the frame began)
            if (!df_wait(&L.sig[WSIG_X], WFG * (t + 1), &L.dead)) return false;
            if (!last) {
#pragma unroll
                for (int g = 0; g < 3; ++g) ws_I(L, R.bI3[g], g, lane, (t + 1) & 1);
            }
            df_signal(&L.sig[WSIG_I]);
        }
        WBSTAMP(15)
        if (!df_wait(&L.sig[WSIG_P2], t + 1, &L.dead)) return false;
        WBSTAMP(17)
        if (!ws_gather2<2>(X, L, 64 + lane, (unsigned)t + 1u)) return false;
        df_signal(&L.sig[WSIG_H2]);
        return true;
    }
    if (!ws_gather1(X, L, 192 + 64 * (bw - 1) + lane, (unsigned)t + 1u, true, t)) return false;
    df_signal(&L.sig[WSIG_H1]);
    WBSTAMP(14)
    if (!df_wait(&L.sig[WSIG_H1], 6 * (t + 1), &L.dead)) return false;
    WBSTAMP(15)
    // Closed loops: behind C(t) -- the chain's product comes first, A(t+1) has the whole frame tail to finish in.  Teacher-forced
    // forward (FPC_WS_A_SPLIT): the first unit's chain beside C(t), the rest behind it.
    if (FC_LATE && FPC_WS_A_SPLIT) {
        if (!last && !ws_A<true>(L, R, bw, lane, [&] { return df_wait(&L.sig[WSIG_C], WFG * (t + 1), &L.dead); })) return false;
    } else {
        if (!df_wait(&L.sig[WSIG_C], WFG * (t + 1), &L.dead)) return false;
        WBSTAMP(17)
        if (!last) ws_A(L, R, bw, lane);
    }
    df_signal(&L.sig[WSIG_A]);
    WBSTAMP(16)
    if (FC_LATE && X.own >= 0 && y_late != nullptr) {
        if (!df_wait(&L.sig[WSIG_H2], 2 * (t + 1), &L.dead)) return false;  // h2(t) whole in LDS
        ws_F_late(L, R, L.pFl[t & 1], bw - 1, lane, X.own);
        df_signal(&L.sig[WSIG_FB]);
        if (bw == 3) {
            if (!df_wait(&L.sig[WSIG_FB], 3 * (t + 1), &L.dead)) return false;
            if (lane < WFC) y_late[lane] = ws_F_out<TANH_ONLY>(L.pFl[t & 1], lane);
        }
    }
    WBSTAMP(18)
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
        if (bw >= 1) {
            ws_A(L, R, bw, lane);
            df_signal(&L.sig[WSIG_A]);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ WsCtx ws_ctx(const WsArgs& S, int group, int slice, int granules = WGRANULES) {
    WsCtx X;
    X.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(S.g + (size_t)group * granules), 0, granules * 16, 0x00020000);
    X.slice = slice;
    X.Lf = 0;
    X.b0 = group * WG;
    X.nu = S.B - X.b0 < WG ? S.B - X.b0 : WG;
    X.err = S.err;
    X.limit = S.limit;
    X.hello_limit = S.hello_limit;
    X.dec = S.dec + group;
    X.fallback = false;
    X.fast = false;
    X.own = slice < X.nu ? slice : -1;
    X.withhold = S.withhold != 0 && group == 0 && slice == WNS - 1;
    X.withhold_hello = X.withhold && S.withhold == 2;
    return X;
}

// TRAIN: the training forward (k_train_fwd's contract): states from zero, the kept activations to `sv`, tanh outputs to
// sv.th instead of predictions to y, no final states
template <bool TRAIN>
__global__ __launch_bounds__(NT) void k_forward_ws(const PredDev P, const float* __restrict__ x, int Lf, float* h1, float* h2,
                                                   float* __restrict__ y, const WsArgs S, const WsSave sv) {
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    X.Lf = Lf;
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) {  // state images [k][u] from [utterance][k]
        const int u = i / WH1, k = i - u * WH1;
        L.h1[k * WG + u] = (!TRAIN && u < X.nu) ? h1[(size_t)(X.b0 + u) * WH1 + k] : 0.0f;
    }
    for (int i = tid; i < WH2 * WG; i += NT) {
        const int u = i / WH2, k = i - u * WH2;
        L.h2[k * WG + u] = (!TRAIN && u < X.nu) ? h2[(size_t)(X.b0 + u) * WH2 + k] : 0.0f;
    }
    for (int i = tid; i < WIN * WG; i += NT) {
        const int u = i / WIN, k = i - u * WIN;
        L.x[ws_xi(k, u)] = (u < X.nu && Lf > 0) ? x[(size_t)(X.b0 + u) * Lf * WIN + k] : 0.0f;
    }
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    if (X.fallback) return;  // (workgroup- and group-uniform, nothing written yet) the row-split launch behind this one serves the group
    const bool owner = slice < X.nu;  // this workgroup stores the outputs of utterance `slice` of the group
    // Teacher forcing: every input row is known in advance, so the input product I(t+1) = W1i x(t+1) leaves the chain -- the
    // foreground puts x(t+1) into LDS as soon as frame t begins (I(t) has been taken from x(t) a frame earlier), background
    // wave 0 computes I(t+1) under hop 1 of frame t.  I(0) here, by the foreground's waves 0-2.
    if (tid < 3 * 64 && Lf > 0) ws_I(L, R.bI, tid >> 6, tid & 63, 0);
    __syncthreads();
    if (tid == 0) L.sig[WSIG_I] = 1;
    __syncthreads();
    WPROF_INIT()
    // this thread's two values of an input row of the group: where they come from (frame 0) and where they go in the image
    const float* xsrc[2];
    int xdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + WFGT * j, u = i / WIN, k = i - u * WIN;
        const bool have = tid < WFGT && i < WIN * WG;
        xsrc[j] = (have && u < X.nu) ? x + (size_t)(X.b0 + u) * Lf * WIN + k : nullptr;
        xdst[j] = have ? ws_xi(k, u) : -1;
    }
    auto x_row = [&](int t, float (&xr)[2]) {  // (0 beyond the end and for the utterances a part-filled group lacks)
#pragma unroll
        for (int j = 0; j < 2; ++j) xr[j] = (t < Lf && xsrc[j] != nullptr) ? xsrc[j][t * WIN] : 0.0f;
    };
    if (tid < WFGT) {
        __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
        float xa[2];
        x_row(1, xa);
        for (int t = 0; t < Lf; ++t) {
            // x(t+1) into LDS (x(t) is not read any more: I(t) is done), then the row after it on its way
            if (!df_wait2(&L.sig[WSIG_I], t + 1, &L.sig[WSIG_A], 3 * (t + 1), &L.dead)) break;  // I(t) and A(t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (xdst[j] >= 0) L.x[xdst[j]] = xa[j];
            df_signal(&L.sig[WSIG_X]);
            x_row(t + 2, xa);
            if (!ws_foreground<true, false, true, TRAIN>(X, L, R, t, tid, &sv)) break;
            WSTAMP(12)
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        for (int tb = 0; tb < Lf; ++tb)
            if (!ws_background<true, true, TRAIN>(X, L, R, tb, tb + 1 == Lf, tid - WFGT,  // (training: tanh row of the owned utterance;
                                                  //  the plain forward's output layer is k_out_layer, over all frames at once)
                                                  (owner && TRAIN) ? sv.th + ((size_t)(X.b0 + slice) * Lf + tb) * WFC : nullptr))
                break;
    }
    WPROF_DUMP(Lf)
    __syncthreads();
    if (TRAIN) return;  // (a launch that gave up leaves its status bit: the step's Adam update is skipped, FPC_ERR_TIMEOUT)
    // new states of this workgroup's units; a launch that gave up fails loudly: NaN outputs and states, FPC_ERR_TIMEOUT
    const bool dead = ws_dead(L);
    const float qnan = __uint_as_float(0x7fc00000u);
    // (y: k_out_layer, behind this launch, writes the predictions of the groups that ran -- NaN for a group marked dead)
    if (dead && tid == 0) __hip_atomic_store((gu32*)X.dec, WS_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int v = tid; v < WV1; v += NT) {
        const int c = v >> 4, u = v & 15;
        if (u < X.nu) h1[(size_t)(X.b0 + u) * WH1 + WU1 * slice + c] = dead ? qnan : L.h1[slice * WV1 + v];
    }
    for (int v = tid; v < WV2; v += NT) {
        const int c = v >> 4, u = v & 15;
        if (u < X.nu) h2[(size_t)(X.b0 + u) * WH2 + WU2 * slice + c] = dead ? qnan : L.h2[slice * WV2 + v];
    }
}

// float64 squared distance in numpy's pairwise order (vq_func.py:18): target in LDS, entry in registers
__device__ __forceinline__ double ws_dist(const double* x, const double (&c)[NDIM + 1]) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double d = x[j] - c[j];
        r[j] = d * d;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double d = x[8 + j] - c[8 + j];
        const double dd = d * d;
        r[j] = r[j] + dd;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const double d = x[16] - c[16];
    const double dd = d * d;
    res = res + dd;
    return res;
}

__global__ __launch_bounds__(NT) void k_decode_feat_ws(const PredDev P, const CbDev C, const float* __restrict__ pitch,
                                                       const int* __restrict__ idx, int Lf, float* __restrict__ c_out,
                                                       int* bad, const WsArgs S) {
    // The receiver in the encoder's distributed form (predictor_wsd.h): EVERY workgroup evaluates the output layer of all 16
    // utterances (one MFMA tile gives all 16 columns for the price of one) and rebuilds every next input row itself from the
    // symbols -- a lookup -- so the frame has two hops, not three, and no workgroup does more than another.  (The first form
    // had the workgroup owning an utterance compute the tile for its one column, rebuild the row, publish it, and everybody
    // gather it: 2.16 ms at 128 x 300 against 1.82 ms for the ENCODER without quantisation, which does strictly more.)
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    X.own = -1;  // (the output layer runs for all utterances on every workgroup: ws_foreground<false, true>)
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) L.h1[i] = 0.0f;
    for (int i = tid; i < WH2 * WG; i += NT) L.h2[i] = 0.0f;
    for (int i = tid; i < WIN * WXP + 4; i += NT) L.x[i] = 0.0f;
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    if (X.fallback) return;  // (workgroup- and group-uniform, nothing written yet) the row-split launch behind this one serves the group
    int i = 0;
    // thread (u, m) = column m of utterance u; the pitch columns (m = 18, 19) are side information of the receiver
    const int u = tid >> 5, m = tid & 31;
    const float* pvp = nullptr;
    if (m >= WFC && m < WIN && u < X.nu) pvp = pitch + (size_t)(X.b0 + u) * Lf * (WIN - WFC) + (m - WFC);
    for (; i < Lf; ++i) {
        const float pv = pvp != nullptr ? pvp[(size_t)i * (WIN - WFC)] : 0.0f;  // fetched before the step
        if (tid < WFGT) {
            __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
            (void)ws_foreground<false, true>(X, L, R, i, tid);
            __builtin_amdgcn_s_setprio(0);
        } else {
            (void)ws_background(X, L, R, i, i + 1 == Lf, tid - WFGT);
        }
        lds_barrier();  // both roles meet: the tail takes the whole workgroup
        wsd_receive_tail(X, L, P, C, idx, c_out, bad, Lf, i, pv, tid);
        if (tid == 0) L.dead_latch = __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_barrier();  // the next input rows are in LDS; the give-up flag as every thread acts on it
        if (L.dead_latch != 0) break;
    }
    if (i < Lf && slice < X.nu) {  // fail loudly: NaN from this frame on (its rows came from a failed hop); FPC_ERR_TIMEOUT on the host
        const float qnan = __uint_as_float(0x7fc00000u);
        const int b = X.b0 + slice;
        for (size_t k = (size_t)i * WIN + tid; k < (size_t)Lf * WIN; k += NT) c_out[(size_t)b * Lf * WIN + k] = qnan;
    }
}

// the codebook-usage histograms of an encode call (cb_tot of Wavernn.encoder, wavernn.py:221-240) from its symbols
// idx [frames][4] = {scalar code (+ n_hi when from the below-threshold book), stage 1, stage 2, below-threshold entry};
// -1 = not coded, -2 = frame refused or poisoned (not counted)
__global__ __launch_bounds__(256) void k_hist_symbols(const CbDev C, const int* __restrict__ idx, size_t frames, int nbins,
                                                      unsigned long long* hist) {
    // a few workgroups, each counting its frames into an LDS histogram (32-bit: a workgroup sees < 2^32 frames) and adding the
    // bins it touched to the caller's 64-bit one.  (One 64-bit global atomic per symbol -- the first form -- took 0.16 ms for the
    // 38 400 frames of 128 x 300: the 256 scalar codes are hot bins.)
    extern __shared__ unsigned bins[];
    for (int k = threadIdx.x; k < nbins; k += 256) bins[k] = 0u;
    __syncthreads();
    const int off_sl = C.n_hi, off_v0 = off_sl + C.n_lo, off_v1 = off_v0 + C.N_hi0, off_vl = off_v1 + (C.S_hi == 2 ? C.N_hi1 : 0);
    for (size_t f = (size_t)blockIdx.x * 256 + threadIdx.x; f < frames; f += (size_t)gridDim.x * 256) {
        const int4 s = *reinterpret_cast<const int4*>(&idx[f * 4]);
        if (s.x >= 0 && s.x < nbins) atomicAdd(&bins[s.x], 1u);  // (codes of the below-threshold book follow the others: same slot arithmetic)
        if (s.y >= 0 && off_v0 + s.y < nbins) atomicAdd(&bins[off_v0 + s.y], 1u);
        if (s.z >= 0 && C.S_hi == 2 && off_v1 + s.z < nbins) atomicAdd(&bins[off_v1 + s.z], 1u);
        if (s.w >= 0 && off_vl + s.w < nbins) atomicAdd(&bins[off_vl + s.w], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256)
        if (bins[k] != 0u) atomicAdd(&hist[k], (unsigned long long)bins[k]);
}
