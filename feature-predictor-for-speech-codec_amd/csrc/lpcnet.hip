// lpcnet.hip -- LPCNet-style vocoder for gfx950 (MI355X), hand-written HIP.
//
// What it replaces: the per-sample Keras loop of xiph/LPCNet training_tf2/test_lpcnet.py
// that the reference invokes from its README (README.md:47).  That code is NOT in
// /root/reference; the algorithm and the canonical evaluation orders are specified in
// DESIGN.md ("Vocoder numerics") and restated on the CPU in oracle/fpc_oracle.c.
// Fragments the reference restates itself: mu-law src/utils.py:16-31, pdf shaping
// src/train.py:79-92, period index src/synthesis.py:103, LPC taps src/utils.py:91-114,
// de-emphasis src/models/wavenet.py:188.
//
// Kernels
//   k_embed_tables   one-off: embed_sig x GRU_A input kernel -> three [256][1152] tables
//   k_frame_dense    frame-rate layers (conv k=3 / dense) as k-ordered fmaf chains
//   k_decode         persistent per-utterance sample loop: one 1024-thread workgroup
//                    (16 wave64) per utterance; sparse GRU_A / GRU_B weights live in
//                    VGPRs for the whole utterance, recurrent state and the dual-FC
//                    table live in LDS, HBM is touched only for the per-frame
//                    conditioning vectors, the embedding-table rows and the PCM output.
#include "fpc_common.h"
#include <algorithm>

namespace {

constexpr int RNN_A = 384, RNN_B = 16, COND = 128, EMB = 128;
constexpr int GA = 3 * RNN_A;  // 1152
constexpr int GB = 3 * RNN_B;  // 48
constexpr int NTHREADS = 512, NWAVES = 8;
constexpr int NROWGRP = GA / 8;  // 144 groups of 8 gate rows
constexpr int FC_STRIDE = 20;    // floats per (node,channel): 16 weights, bias, factor, pad

// ---------------------------------------------------------------------------------
// one-off table build: tab[s][e][row] = sum_k embed_sig[e][k] * ga_k[s*128+k][row]
// (float64, k ascending, rounded once: identical to oracle/fpc_oracle.c orc_lpcnet_create)
// ---------------------------------------------------------------------------------
__global__ void k_embed_tables(const float* __restrict__ embed, const float* __restrict__ ga_k,
                               float* __restrict__ tab) {
    const int e = blockIdx.x, s = blockIdx.y;
    for (int row = threadIdx.x; row < GA; row += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < EMB; ++k) {
            const double t = (double)embed[e * EMB + k] * (double)ga_k[(size_t)(s * EMB + k) * GA + row];
            acc = acc + t;
        }
        tab[((size_t)s * 256 + e) * GA + row] = (float)acc;
    }
}

// ---------------------------------------------------------------------------------
// frame-rate layers.  y[f][o] = act(bias[o] + sum_k x_f[k] W[k][o]) as a k-ordered
// fmaf chain (the order a gfx950 f32 MFMA accumulates in, so an MFMA version stays
// bit-identical).  MODE 0: x_f = x[f][0..K)   MODE 1: 'same' conv, K = 3*C,
// x_f[tap*C+c] = x[f+tap-1][c] with zero rows outside the utterance.
// MODE 2: first layer input built on the fly from the 36-float feature frame:
//         20 features | 64-dim pitch embedding (conv, C = 84).
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ void k_frame_dense(const float* __restrict__ x, int ldx, int C, int K,
                              const float* __restrict__ W, const float* __restrict__ bias, int N,
                              float* __restrict__ y, int T, int do_tanh,
                              const float* __restrict__ embed_pitch) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int f = blockIdx.x;  // frame index over B*T
    const int t = f % T;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float v;
        if (MODE == 0) {
            v = x[(size_t)f * ldx + k];
        } else {
            const int tap = k / C, c = k - tap * C;
            const int tt = t + tap - 1;
            if (tt < 0 || tt >= T) {
                v = 0.0f;
            } else if (MODE == 1) {
                v = x[(size_t)(f + tap - 1) * ldx + c];
            } else {
                const float* fr = x + (size_t)(f + tap - 1) * FPC_NB_FEATURES;
                v = c < FPC_NB_USED_FEATURES
                        ? fr[c]
                        : embed_pitch[fpc_period_index(fr[18]) * 64 + (c - FPC_NB_USED_FEATURES)];
            }
        }
        xs[k] = v;
    }
    __syncthreads();
    for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < N; o += gridDim.y * blockDim.x) {
        float acc = bias[o];
#pragma unroll 4
        for (int k = 0; k < K; ++k) acc = fmaf(xs[k], W[(size_t)k * N + o], acc);
        y[(size_t)f * N + o] = do_tanh ? fpc_tanhf(acc) : acc;
    }
}

// ---------------------------------------------------------------------------------
// decode kernel: one 512-thread workgroup (8 wave64, 2 per SIMD, 256 VGPRs each) per
// utterance.  Per output sample, six workgroup barriers separate the phases
//   A  issue embedding-row gather | sparse GRU_A mat-vec (weights in VGPRs) | side chains
//   B  GRU_A gates (384 lanes)                      C  GRU_B (half-wave = unit)
//   D1 dual-FC node probabilities (510 lanes)       D2 tree pdf + per-leaf candidates
//   EF (wave 0) normaliser, tail cut, scan, draw, publish next-step control block
// ---------------------------------------------------------------------------------
struct DecodeParams {
    const float* tab;       // [3][256][1152]
    const float* cfa;       // [B][T][1152]  GRU_A conditioning product (+biases)
    const float* cfb;       // [B][T][48]    GRU_B conditioning product (+biases)
    const float* features;  // [B][T][36]
    const unsigned long long* seeds;
    int16_t* pcm;  // [B][T*160]
    int T;
    const float* lane_w;     // [128][512] sparse GRU_A weights: 2 leaves x 2 blocks x 8x4
    const int* lane_meta;    // [6][512]   4 column blocks, row group, lane-in-group | lanes<<8
    const float* lane_wb;    // [36][512]  GRU_B input weights [gate][12 inputs]
    const float* ub;         // [16][48]
    const float* diag;       // [1152]
    const float* brn_a;      // [384]
    const float* brn_b;      // [16]
    const float* fc_tab;     // [256][2][FC_STRIDE]
    const float* ulaw_tab;   // [256]
    int wave_maxQ[NWAVES];
    unsigned long long* stamps;  // diagnostic only
};

struct __attribute__((aligned(16))) DecodeLds {
    float s1[RNN_A];
    float rec[GA];
    float diag[GA];
    float brn_a[RNN_A];
    float ub[RNN_B * GB];
    float fc[256 * 2 * FC_STRIDE];
    float uframe[FPC_FRAME_SIZE];
    float q[256];
    float p[256];
    float cand_pcm[256];
    float cand_pred[256];
    int cand_e[256];  // e_sig | e_pred << 8
    float s2[RNN_B];
    float brn_b[RNN_B];
    float hist[16];
    // control block written by the winning lane / the LPC chain lane
    int e_sig, e_pred, e_exc, pad0;
    float pred, partial, a1n, mem;
};

__device__ __forceinline__ float bfly_sum(float v, int width) {
    for (int s = 1; s < width; s <<= 1) v = v + __shfl_xor(v, s);
    return v;
}

// STAMP=true is a diagnostic build (env FPC_DECODE_STAMPS=1): wave 0 / lane 0 of block 0
// accumulates s_memtime deltas per phase into P.stamps; never used for timing claims.
#define FPC_STAMP(k)                                              \
    if (STAMP) {                                                  \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[k] += now_ - st_last;                              \
        st_last = now_;                                           \
    }

template <bool STAMP>
__global__ __launch_bounds__(NTHREADS) void k_decode(const DecodeParams P) {
    __shared__ DecodeLds L;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = STAMP ? __builtin_readcyclecounter() : 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, T = P.T;
    const unsigned long long seed = P.seeds[b];

    // ---- weights that stay in registers for the whole utterance ----
    float w[128];
#pragma unroll
    for (int j = 0; j < 128; ++j) w[j] = P.lane_w[j * NTHREADS + tid];
    int colb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) colb[j] = P.lane_meta[j * NTHREADS + tid];
    const int grp = P.lane_meta[4 * NTHREADS + tid];
    const int lanem = P.lane_meta[5 * NTHREADS + tid];
    const int lq = lanem & 0xff, lQ = lanem >> 8;
    float wb[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) wb[j] = P.lane_wb[j * NTHREADS + tid];
    int maxQ = 1;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i)
        if (i == __builtin_amdgcn_readfirstlane(wave)) maxQ = P.wave_maxQ[i];
    const float my_ulaw = P.ulaw_tab[tid & 255];

    // ---- LDS init ----
    for (int i = tid; i < RNN_A; i += NTHREADS) {
        L.s1[i] = 0.0f;
        L.brn_a[i] = P.brn_a[i];
    }
    for (int i = tid; i < GA; i += NTHREADS) {
        L.rec[i] = 0.0f;
        L.diag[i] = P.diag[i];
    }
    for (int i = tid; i < RNN_B * GB; i += NTHREADS) L.ub[i] = P.ub[i];
    for (int i = tid; i < 256 * 2 * FC_STRIDE; i += NTHREADS) L.fc[i] = P.fc_tab[i];
    if (tid < RNN_B) {
        L.s2[tid] = 0.0f;
        L.brn_b[tid] = P.brn_b[tid];
        L.hist[tid] = 0.0f;
    }
    if (tid == 0) {
        L.e_sig = 128;
        L.e_pred = 128;
        L.e_exc = 128;
        L.pred = -0.0f;
        L.partial = 0.0f;
        L.a1n = 0.0f;
        L.mem = 0.0f;
    }
    int16_t* out = P.pcm + (size_t)b * T * FPC_FRAME_SIZE;
    if (tid < FPC_LPC_ORDER + 1) out[tid] = 0;  // test_lpcnet.py skips order+1 samples
    __syncthreads();

    const int fnode = tid >> 1, fch = tid & 1;  // dual-FC role: (tree node, channel)
    const float* fcw = &L.fc[(fnode * 2 + fch) * FC_STRIDE];
    const int half = lane >> 5, hl = lane & 31;  // GRU_B role: unit 2*wave+half, leaves 2hl,2hl+1
    const int unitB = 2 * wave + half;

    for (int fr = 0; fr < T; ++fr) {
        const float* feat = P.features + ((size_t)b * T + fr) * FPC_NB_FEATURES;
        const float shape_e = fpc_shape_exponent(feat[19]);
        const float* cfa = P.cfa + ((size_t)b * T + fr) * GA;
        const float* cfb = P.cfb + ((size_t)b * T + fr) * GB;
        if (tid < FPC_FRAME_SIZE)
            L.uframe[tid] = fpc_philox_uniform(seed, (uint32_t)(fr * FPC_FRAME_SIZE + tid));
        // uframe is first read in phase EF, behind several barriers

        for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
            const int t = fr * FPC_FRAME_SIZE + i;

            // ================= phase A: gather issue | sparse mat-vec | side chains =========
            float g_t[9], g_c[3];
            if (tid < RNN_A) {
                const unsigned o0 = (unsigned)L.e_sig * GA + tid;
                const unsigned o1 = (256u + (unsigned)L.e_pred) * GA + tid;
                const unsigned o2 = (512u + (unsigned)L.e_exc) * GA + tid;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    g_t[g * 3 + 0] = P.tab[o0 + g * RNN_A];
                    g_t[g * 3 + 1] = P.tab[o1 + g * RNN_A];
                    g_t[g * 3 + 2] = P.tab[o2 + g * RNN_A];
                    g_c[g] = cfa[(unsigned)(g * RNN_A + tid)];
                }
            } else {
#pragma unroll
                for (int g = 0; g < 9; ++g) g_t[g] = 0.0f;
                g_c[0] = g_c[1] = g_c[2] = 0.0f;
            }
            {
                float acc[8];
#pragma unroll
                for (int lf = 0; lf < 2; ++lf) {  // two canonical leaves of two blocks each
                    const float4 ha = *reinterpret_cast<const float4*>(&L.s1[colb[2 * lf] * 4]);
                    const float4 hb = *reinterpret_cast<const float4*>(&L.s1[colb[2 * lf + 1] * 4]);
                    const float* wl = &w[lf * 64];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float a = 0.0f;
                        a = fmaf(wl[r * 4 + 0], ha.x, a);
                        a = fmaf(wl[r * 4 + 1], ha.y, a);
                        a = fmaf(wl[r * 4 + 2], ha.z, a);
                        a = fmaf(wl[r * 4 + 3], ha.w, a);
                        a = fmaf(wl[32 + r * 4 + 0], hb.x, a);
                        a = fmaf(wl[32 + r * 4 + 1], hb.y, a);
                        a = fmaf(wl[32 + r * 4 + 2], hb.z, a);
                        a = fmaf(wl[32 + r * 4 + 3], hb.w, a);
                        acc[r] = lf == 0 ? a : acc[r] + a;  // first tree level is in-lane
                    }
                }
                for (int s = 1; s < maxQ; s <<= 1) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float o = __shfl_down(acc[r], s);
                        if (lq + s < lQ) acc[r] = acc[r] + o;
                    }
                }
                if (grp >= 0 && lq == 0) {
                    const int gate = grp / (RNN_A / 8), rb = grp - gate * (RNN_A / 8);
                    const int row0 = gate * RNN_A + rb * 8;
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        L.rec[row0 + r] = fmaf(L.diag[row0 + r], L.s1[rb * 8 + r], acc[r]);
                }
            }
            // GRU_B recurrent part of this wave's two units: balanced tree over 16 products
            float ub_z, ub_r, ub_n;
            {
                const int g = (lane >> 4) < 3 ? (lane >> 4) : 0, k = lane & 15;
                const float s2k = L.s2[k];
                const float pa = bfly_sum(L.ub[k * GB + g * RNN_B + 2 * wave] * s2k, 16);
                const float pb = bfly_sum(L.ub[k * GB + g * RNN_B + 2 * wave + 1] * s2k, 16);
                const float za = __shfl(pa, 0), ra = __shfl(pa, 16), na = __shfl(pa, 32);
                const float zb = __shfl(pb, 0), rb2 = __shfl(pb, 16), nb = __shfl(pb, 32);
                ub_z = half ? zb : za;
                ub_r = half ? rb2 : ra;
                ub_n = half ? nb : na;
            }
            // LPC history chain for the NEXT sample (all taps except the newest)
            if (tid == NTHREADS - 1) {
                int frn = (t + 1) / FPC_FRAME_SIZE;
                frn = frn < T ? frn : T - 1;
                const float* a = P.features + ((size_t)b * T + frn) * FPC_NB_FEATURES +
                                 (FPC_NB_FEATURES - FPC_LPC_ORDER);
                float acc = 0.0f;
#pragma unroll
                for (int k = FPC_LPC_ORDER; k >= 2; --k)
                    acc = fmaf(a[k - 1], L.hist[(t + 1 - k) & 15], acc);
                L.partial = acc;
                L.a1n = a[0];
            }
            // input-side gate sums (the gather has landed behind the mat-vec by now)
            const float gz = ((g_t[0] + g_t[1]) + g_t[2]) + g_c[0];
            const float gr = ((g_t[3] + g_t[4]) + g_t[5]) + g_c[1];
            const float gn = ((g_t[6] + g_t[7]) + g_t[8]) + g_c[2];
            FPC_STAMP(0)
            __syncthreads();  // B0
            FPC_STAMP(6)

            // ================= phase B: GRU_A gates =====================================
            if (tid < RNN_A) {
                const float h = L.s1[tid];
                const float z = fpc_sigmoidf(gz + L.rec[tid]);
                const float r = fpc_sigmoidf(gr + L.rec[RNN_A + tid]);
                const float n = fpc_tanhf(fmaf(r, L.rec[2 * RNN_A + tid] + L.brn_a[tid], gn));
                L.s1[tid] = fmaf(z, h - n, n);
            }
            FPC_STAMP(1)
            __syncthreads();  // B1
            FPC_STAMP(6)

            // ================= phase C: GRU_B (half-wave = unit) =========================
            {
                const float4 h0 = *reinterpret_cast<const float4*>(&L.s1[12 * hl]);
                const float4 h1 = *reinterpret_cast<const float4*>(&L.s1[12 * hl + 4]);
                const float4 h2 = *reinterpret_cast<const float4*>(&L.s1[12 * hl + 8]);
                float a3[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float* wg = &wb[g * 12];
                    float a = 0.0f, c = 0.0f;  // leaves 2hl and 2hl+1 (6 inputs each)
                    a = fmaf(wg[0], h0.x, a);
                    a = fmaf(wg[1], h0.y, a);
                    a = fmaf(wg[2], h0.z, a);
                    a = fmaf(wg[3], h0.w, a);
                    a = fmaf(wg[4], h1.x, a);
                    a = fmaf(wg[5], h1.y, a);
                    c = fmaf(wg[6], h1.z, c);
                    c = fmaf(wg[7], h1.w, c);
                    c = fmaf(wg[8], h2.x, c);
                    c = fmaf(wg[9], h2.y, c);
                    c = fmaf(wg[10], h2.z, c);
                    c = fmaf(wg[11], h2.w, c);
                    a3[g] = bfly_sum(a + c, 32);
                }
                if (hl == 0) {
                    const float cfb_z = cfb[unitB], cfb_r = cfb[RNN_B + unitB], cfb_n = cfb[2 * RNN_B + unitB];
                    const float so = L.s2[unitB];
                    const float z = fpc_sigmoidf((a3[0] + cfb_z) + ub_z);
                    const float r = fpc_sigmoidf((a3[1] + cfb_r) + ub_r);
                    const float n = fpc_tanhf(fmaf(r, ub_n + L.brn_b[unitB], a3[2] + cfb_n));
                    L.s2[unitB] = fmaf(z, so - n, n);
                }
            }
            FPC_STAMP(2)
            __syncthreads();  // B2
            FPC_STAMP(6)

            // ================= phase D1: dual FC -> node probabilities ====================
            {
                float acc = fcw[16];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 wv = *reinterpret_cast<const float4*>(&fcw[4 * k4]);
                    const float4 sv = *reinterpret_cast<const float4*>(&L.s2[4 * k4]);
                    acc = fmaf(wv.x, sv.x, acc);
                    acc = fmaf(wv.y, sv.y, acc);
                    acc = fmaf(wv.z, sv.z, acc);
                    acc = fmaf(wv.w, sv.w, acc);
                }
                const float tt = fpc_tanhf(acc);
                const float ff = fcw[17];
                const float to = __shfl_xor(tt, 1), fo = __shfl_xor(ff, 1);
                // v = fma(f1, t1, f0*t0) evaluated identically on both lanes of the pair
                const float v = fch == 0 ? fmaf(fo, to, ff * tt) : fmaf(ff, tt, fo * to);
                if (fch == 0) L.q[fnode] = fpc_sigmoidf(v);
            }
            FPC_STAMP(3)
            __syncthreads();  // B3
            FPC_STAMP(6)

            // ================= phase D2: tree pdf and per-leaf candidates (256 lanes) ======
            if (tid < 256) {
                float p = 1.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float qq = L.q[(1 << j) + (tid >> (8 - j))];
                    p = p * (((tid >> (7 - j)) & 1) ? qq : 1.0f - qq);
                }
                if (shape_e > 0.0f) p = fpc_shape_pow(p, shape_e);
                L.p[tid] = p;
                // what the control block becomes if this leaf wins the draw
                const float cpcm = L.pred + my_ulaw;
                const float cpred = -fmaf(L.a1n, cpcm, L.partial);
                L.cand_pcm[tid] = cpcm;
                L.cand_pred[tid] = cpred;
                L.cand_e[tid] = fpc_lin2ulaw(cpcm) | (fpc_lin2ulaw(cpred) << 8);
            }
            FPC_STAMP(4)
            __syncthreads();  // B4
            FPC_STAMP(6)

            // ================= phase EF (wave 0): normaliser, tail cut, scan, draw ==========
            if (wave == 0) {
                const float4 p4 = *reinterpret_cast<const float4*>(&L.p[4 * lane]);
                const float S1 = bfly_sum((p4.x + p4.y) + (p4.z + p4.w), 64);
                const float thr = 0.002f * S1;
                float c0 = p4.x - thr, c1 = p4.y - thr, c2 = p4.z - thr, c3 = p4.w - thr;
                c0 = c0 > 0.0f ? c0 : 0.0f;
                c1 = c1 > 0.0f ? c1 : 0.0f;
                c2 = c2 > 0.0f ? c2 : 0.0f;
                c3 = c3 > 0.0f ? c3 : 0.0f;
                c1 = c0 + c1;  // sequential prefix inside the lane's 4 leaves
                c2 = c1 + c2;
                c3 = c2 + c3;
                float I = c3;  // Kogge-Stone over the 64 lane totals
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) {
                    const float o = __shfl_up(I, s);
                    if (lane >= s) I = I + o;
                }
                const float S2 = __shfl(I, 63);
                const float rthr = L.uframe[i] * S2;
                const int lw = __popcll(__ballot(lane < 63 && I <= rthr));  // winning lane
                const float Iprev = __shfl_up(I, 1);
                if (lane == lw) {
                    const float O = lane > 0 ? Iprev : 0.0f;
                    int cnt = ((O + c0) <= rthr) + ((O + c1) <= rthr) + ((O + c2) <= rthr) +
                              ((O + c3) <= rthr);
                    cnt = cnt > 3 ? 3 : cnt;
                    const int exc = 4 * lane + cnt;
                    const int ce = L.cand_e[exc];
                    const float cpcm = L.cand_pcm[exc];
                    L.e_sig = ce & 0xff;
                    L.e_pred = ce >> 8;
                    L.e_exc = exc;
                    L.pred = L.cand_pred[exc];
                    L.hist[t & 15] = cpcm;
                    const float mem = fmaf(FPC_PREEMPH, L.mem, cpcm);
                    L.mem = mem;
                    out[t] = fpc_pcm16(mem);
                }
            }
            FPC_STAMP(5)
            __syncthreads();  // B5
            FPC_STAMP(6)
        }
    }
    if (STAMP && blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 8; ++k) P.stamps[k] = st_acc[k];
}

}  // namespace

// =====================================================================================
// host side
// =====================================================================================
struct fpc_lpcnet {
    int device = 0;
    fpc::DevBuf embed_pitch, conv1_k, conv1_b, conv2_k, conv2_b, d1_k, d1_b, d2_k, d2_b;
    fpc::DevBuf ga_k, gb_k, bias_a, bias_b, tab;
    fpc::DevBuf lane_w, lane_meta, lane_wb, ub, diag, brn_a, brn_b, fc_tab, ulaw_tab;
    int wave_maxQ[NWAVES];
    int nblocks = 0, nleaves = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

static size_t ws_floats_per_frame() { return 128 + 128 + 128 + GA + GB; }

extern "C" long long fpc_lpcnet_workspace_bytes(const fpc_lpcnet* m, int B, int T) {
    (void)m;
    if (B <= 0 || T <= 0) return 0;
    return (long long)B * T * (long long)ws_floats_per_frame() * 4 + 256;
}

extern "C" int fpc_lpcnet_create(const fpc_lpcnet_weights* w, fpc_lpcnet** out) {
    FPC_REQUIRE(w && out, "fpc_lpcnet_create: null argument");
    if (!fpc::have_device()) {
        fpc::set_error("fpc_lpcnet_create: no HIP device (libfpcodec has no CPU fallback)");
        return FPC_ERR_NO_DEVICE;
    }
    const float* const* ptrs = reinterpret_cast<const float* const*>(w);
    for (size_t i = 0; i < sizeof(*w) / sizeof(float*); ++i)
        FPC_REQUIRE(ptrs[i], "fpc_lpcnet_create: weight pointer %zu is null", i);
    fpc_lpcnet* m = new fpc_lpcnet();
    FPC_HIP(hipGetDevice(&m->device));

    auto up = [&](fpc::DevBuf& d, const float* src, size_t n) -> hipError_t {
        hipError_t e = d.alloc(n * 4);
        if (e != hipSuccess) return e;
        return hipMemcpy(d.p, src, n * 4, hipMemcpyHostToDevice);
    };
    FPC_HIP(up(m->embed_pitch, w->embed_pitch, 256 * 64));
    FPC_HIP(up(m->conv1_k, w->conv1_kernel, 3 * 84 * 128));
    FPC_HIP(up(m->conv1_b, w->conv1_bias, 128));
    FPC_HIP(up(m->conv2_k, w->conv2_kernel, 3 * 128 * 128));
    FPC_HIP(up(m->conv2_b, w->conv2_bias, 128));
    FPC_HIP(up(m->d1_k, w->dense1_kernel, 128 * 128));
    FPC_HIP(up(m->d1_b, w->dense1_bias, 128));
    FPC_HIP(up(m->d2_k, w->dense2_kernel, 128 * 128));
    FPC_HIP(up(m->d2_b, w->dense2_bias, 128));
    FPC_HIP(up(m->ga_k, w->gru_a_kernel, 512 * GA));
    FPC_HIP(up(m->gb_k, w->gru_b_kernel, 512 * GB));

    // folded biases (single float adds, same as the oracle)
    std::vector<float> bias_a(GA), brn_a(RNN_A), bias_b(GB), brn_b(RNN_B), diag(GA);
    for (int r = 0; r < GA; ++r)
        bias_a[r] = r < 2 * RNN_A ? w->gru_a_bias[r] + w->gru_a_bias[GA + r] : w->gru_a_bias[r];
    for (int i = 0; i < RNN_A; ++i) brn_a[i] = w->gru_a_bias[GA + 2 * RNN_A + i];
    for (int o = 0; o < GB; ++o)
        bias_b[o] = o < 2 * RNN_B ? w->gru_b_bias[o] + w->gru_b_bias[GB + o] : w->gru_b_bias[o];
    for (int i = 0; i < RNN_B; ++i) brn_b[i] = w->gru_b_bias[GB + 2 * RNN_B + i];
    for (int g = 0; g < 3; ++g)
        for (int i = 0; i < RNN_A; ++i)
            diag[g * RNN_A + i] = w->gru_a_recurrent[(size_t)i * GA + g * RNN_A + i];
    FPC_HIP(m->bias_a.upload(bias_a));
    FPC_HIP(m->brn_a.upload(brn_a));
    FPC_HIP(m->bias_b.upload(bias_b));
    FPC_HIP(m->brn_b.upload(brn_b));
    FPC_HIP(m->diag.upload(diag));

    // ---- block-sparse packing of the GRU_A recurrent matrix (8 outputs x 4 inputs) ----
    struct Grp {
        int id;
        std::vector<int> cols;
    };
    std::vector<Grp> grps(NROWGRP);
    for (int g = 0; g < NROWGRP; ++g) {
        const int gate = g / (RNN_A / 8), rb = g % (RNN_A / 8);
        grps[g].id = g;
        for (int cb = 0; cb < RNN_A / 4; ++cb) {
            bool nz = false;
            for (int r = 0; r < 8 && !nz; ++r)
                for (int c = 0; c < 4; ++c) {
                    const int in = cb * 4 + c, o = rb * 8 + r;
                    if (in != o && w->gru_a_recurrent[(size_t)in * GA + gate * RNN_A + o] != 0.0f) {
                        nz = true;
                        break;
                    }
                }
            if (nz) grps[g].cols.push_back(cb);
        }
        m->nblocks += (int)grps[g].cols.size();
    }
    // canonical leaf = 2 consecutive blocks; a lane owns 2 consecutive leaves (4 blocks);
    // the lanes of one row group are consecutive lanes of one wave
    std::vector<int> order(NROWGRP);
    for (int g = 0; g < NROWGRP; ++g) order[g] = g;
    auto lanes_of = [&](int g) {  // an empty row group still owns one (all-zero) lane
        const int n = ((int)grps[g].cols.size() + 3) / 4;
        return n > 0 ? n : 1;
    };
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        const int la = lanes_of(a), lb = lanes_of(b);
        return la != lb ? la > lb : a < b;
    });
    int wave_fill[NWAVES] = {0};
    for (int i = 0; i < NWAVES; ++i) m->wave_maxQ[i] = 1;
    std::vector<float> lane_w((size_t)128 * NTHREADS, 0.0f);
    std::vector<int> lane_meta(6 * NTHREADS, 0);
    for (int l = 0; l < NTHREADS; ++l) {
        lane_meta[4 * NTHREADS + l] = -1;
        lane_meta[5 * NTHREADS + l] = (1 << 8);
    }
    for (int oi = 0; oi < NROWGRP; ++oi) {
        const int g = order[oi], Q = lanes_of(g);
        int best = -1;
        for (int wv = 0; wv < NWAVES; ++wv)  // least-filled wave that still has room
            if (wave_fill[wv] + Q <= 64 && (best < 0 || wave_fill[wv] < wave_fill[best])) best = wv;
        if (best < 0) {
            fpc::set_error(
                "fpc_lpcnet_create: recurrent matrix too dense for the register-resident "
                "layout (%d blocks of 8x4; capacity %d)", m->nblocks, 4 * NTHREADS);
            delete m;
            return FPC_ERR_CAPACITY;
        }
        const int gate = g / (RNN_A / 8), rb = g % (RNN_A / 8);
        for (int q = 0; q < Q; ++q) {
            const int l = best * 64 + wave_fill[best] + q;
            lane_meta[4 * NTHREADS + l] = g;
            lane_meta[5 * NTHREADS + l] = q | (Q << 8);
            for (int bb = 0; bb < 4; ++bb) {
                const int bi = 4 * q + bb;
                if (bi >= (int)grps[g].cols.size()) continue;  // weights stay 0, column 0
                const int cb = grps[g].cols[bi];
                lane_meta[bb * NTHREADS + l] = cb;
                for (int r = 0; r < 8; ++r)
                    for (int c = 0; c < 4; ++c) {
                        const int in = cb * 4 + c, o = rb * 8 + r;
                        lane_w[(size_t)(bb * 32 + r * 4 + c) * NTHREADS + l] =
                            in == o ? 0.0f : w->gru_a_recurrent[(size_t)in * GA + gate * RNN_A + o];
                    }
            }
        }
        wave_fill[best] += Q;
        if (Q > m->wave_maxQ[best]) m->wave_maxQ[best] = Q;
        m->nleaves += Q;
    }
    FPC_HIP(m->lane_w.upload(lane_w));
    FPC_HIP(m->lane_meta.upload(lane_meta));

    // GRU_B input weights: half-wave = unit, lane = 12 consecutive inputs (2 leaves of 6)
    std::vector<float> lane_wb(36 * NTHREADS);
    for (int wv = 0; wv < NWAVES; ++wv)
        for (int l = 0; l < 64; ++l) {
            const int unit = 2 * wv + (l >> 5), hl = l & 31;
            for (int g = 0; g < 3; ++g)
                for (int k = 0; k < 12; ++k)
                    lane_wb[(g * 12 + k) * NTHREADS + wv * 64 + l] =
                        w->gru_b_kernel[(size_t)(12 * hl + k) * GB + g * RNN_B + unit];
        }
    FPC_HIP(m->lane_wb.upload(lane_wb));
    FPC_HIP(up(m->ub, w->gru_b_recurrent, RNN_B * GB));

    std::vector<float> fc(256 * 2 * FC_STRIDE, 0.0f);
    for (int j = 0; j < 256; ++j)
        for (int ch = 0; ch < 2; ++ch) {
            float* d = &fc[(j * 2 + ch) * FC_STRIDE];
            for (int k = 0; k < RNN_B; ++k) d[k] = w->md_kernel[((size_t)j * RNN_B + k) * 2 + ch];
            d[16] = w->md_bias[j * 2 + ch];
            d[17] = w->md_factor[j * 2 + ch];
        }
    FPC_HIP(m->fc_tab.upload(fc));
    std::vector<float> ulaw(256);
    for (int u = 0; u < 256; ++u) ulaw[u] = fpc_ulaw2lin(u);
    FPC_HIP(m->ulaw_tab.upload(ulaw));

    // embedding x kernel tables on the device
    fpc::DevBuf embed_sig;
    FPC_HIP(up(embed_sig, w->embed_sig, 256 * EMB));
    FPC_HIP(m->tab.alloc((size_t)3 * 256 * GA * 4));
    hipLaunchKernelGGL(k_embed_tables, dim3(256, 3), dim3(384), 0, 0, embed_sig.as<float>(),
                       m->ga_k.as<float>(), m->tab.as<float>());
    FPC_HIP(hipGetLastError());
    FPC_HIP(hipDeviceSynchronize());
    FPC_HIP(hipEventCreate(&m->ev0));
    FPC_HIP(hipEventCreate(&m->ev1));
    *out = m;
    return FPC_OK;
}

extern "C" void fpc_lpcnet_destroy(fpc_lpcnet* m) {
    if (!m) return;
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    delete m;
}

namespace {
struct CondBufs {
    float *x1, *x2, *x3, *cfa, *cfb;
};
CondBufs carve(void* ws, int B, int T) {
    float* p = static_cast<float*>(ws);
    const size_t F = (size_t)B * T;
    CondBufs c;
    c.x1 = p;
    c.x2 = c.x1 + F * 128;
    c.x3 = c.x2 + F * 128;
    c.cfa = c.x3 + F * 128;
    c.cfb = c.cfa + F * GA;
    return c;
}

int run_condition(fpc_lpcnet* m, const float* feat, int B, int T, const CondBufs& c, float* cfeat,
                  hipStream_t st) {
    const int F = B * T;
    // conv1 (84 ch, built on the fly from the feature frame) -> x1
    hipLaunchKernelGGL(k_frame_dense<2>, dim3(F, 1), dim3(128), 3 * 84 * 4, st, feat, 0, 84, 3 * 84,
                       m->conv1_k.as<float>(), m->conv1_b.as<float>(), 128, c.x1, T, 1,
                       m->embed_pitch.as<float>());
    // conv2 -> x2
    hipLaunchKernelGGL(k_frame_dense<1>, dim3(F, 1), dim3(128), 3 * 128 * 4, st, c.x1, 128, 128,
                       3 * 128, m->conv2_k.as<float>(), m->conv2_b.as<float>(), 128, c.x2, T, 1,
                       (const float*)nullptr);
    // dense1 -> x3, dense2 -> cfeat
    hipLaunchKernelGGL(k_frame_dense<0>, dim3(F, 1), dim3(128), 128 * 4, st, c.x2, 128, 128, 128,
                       m->d1_k.as<float>(), m->d1_b.as<float>(), 128, c.x3, T, 1,
                       (const float*)nullptr);
    hipLaunchKernelGGL(k_frame_dense<0>, dim3(F, 1), dim3(128), 128 * 4, st, c.x3, 128, 128, 128,
                       m->d2_k.as<float>(), m->d2_b.as<float>(), 128, cfeat, T, 1,
                       (const float*)nullptr);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}
}  // namespace

extern "C" int fpc_lpcnet_condition(fpc_lpcnet* m, const float* features_dev, int B, int T,
                                    float* cfeat_dev, void* workspace_dev, fpc_stream s) {
    FPC_REQUIRE(m && features_dev && cfeat_dev && workspace_dev, "fpc_lpcnet_condition: null argument");
    FPC_REQUIRE(B > 0 && T > 0, "fpc_lpcnet_condition: bad shape B=%d T=%d", B, T);
    return run_condition(m, features_dev, B, T, carve(workspace_dev, B, T), cfeat_dev,
                         static_cast<hipStream_t>(s));
}

extern "C" int fpc_lpcnet_synthesize(fpc_lpcnet* m, const float* features_dev, int B, int T,
                                     const uint64_t* seeds_dev, int16_t* pcm_dev,
                                     void* workspace_dev, fpc_stream s) {
    FPC_REQUIRE(m && features_dev && seeds_dev && pcm_dev && workspace_dev,
                "fpc_lpcnet_synthesize: null argument");
    FPC_REQUIRE(B > 0 && T > 0 && (long long)T * FPC_FRAME_SIZE < (1ll << 31),
                "fpc_lpcnet_synthesize: bad shape B=%d T=%d", B, T);
    hipStream_t st = static_cast<hipStream_t>(s);
    const CondBufs c = carve(workspace_dev, B, T);
    float* cfeat = c.x1;  // x1 is free again once conv2 has run
    // cfeat cannot alias a live buffer: run conv1->x1, conv2->x2, d1->x3, d2->x1
    int rc = run_condition(m, features_dev, B, T, c, cfeat, st);
    if (rc != FPC_OK) return rc;
    const int F = B * T;
    // conditioning products with the cfeat rows of both GRU input kernels
    hipLaunchKernelGGL(k_frame_dense<0>, dim3(F, 3), dim3(384), 128 * 4, st, cfeat, 128, 128, 128,
                       m->ga_k.as<float>() + (size_t)3 * EMB * GA, m->bias_a.as<float>(), GA, c.cfa,
                       T, 0, (const float*)nullptr);
    hipLaunchKernelGGL(k_frame_dense<0>, dim3(F, 1), dim3(64), 128 * 4, st, cfeat, 128, 128, 128,
                       m->gb_k.as<float>() + (size_t)RNN_A * GB, m->bias_b.as<float>(), GB, c.cfb, T,
                       0, (const float*)nullptr);
    DecodeParams P;
    P.tab = m->tab.as<float>();
    P.cfa = c.cfa;
    P.cfb = c.cfb;
    P.features = features_dev;
    P.seeds = reinterpret_cast<const unsigned long long*>(seeds_dev);
    P.pcm = pcm_dev;
    P.T = T;
    P.lane_w = m->lane_w.as<float>();
    P.lane_meta = m->lane_meta.as<int>();
    P.lane_wb = m->lane_wb.as<float>();
    P.ub = m->ub.as<float>();
    P.diag = m->diag.as<float>();
    P.brn_a = m->brn_a.as<float>();
    P.brn_b = m->brn_b.as<float>();
    P.fc_tab = m->fc_tab.as<float>();
    P.ulaw_tab = m->ulaw_tab.as<float>();
    for (int i = 0; i < NWAVES; ++i) P.wave_maxQ[i] = m->wave_maxQ[i];
    const bool stamp = getenv("FPC_DECODE_STAMPS") != nullptr;
    fpc::DevBuf stamps;
    P.stamps = nullptr;
    if (stamp) {
        FPC_HIP(stamps.alloc(8 * sizeof(unsigned long long)));
        P.stamps = stamps.as<unsigned long long>();
    }
    FPC_HIP(hipEventRecord(m->ev0, st));
    if (stamp)
        hipLaunchKernelGGL(k_decode<true>, dim3(B), dim3(NTHREADS), 0, st, P);
    else
        hipLaunchKernelGGL(k_decode<false>, dim3(B), dim3(NTHREADS), 0, st, P);
    FPC_HIP(hipEventRecord(m->ev1, st));
    FPC_HIP(hipGetLastError());
    if (stamp) {  // diagnostic path only: synchronises
        unsigned long long h[8];
        FPC_HIP(hipStreamSynchronize(st));
        FPC_HIP(hipMemcpy(h, stamps.p, sizeof h, hipMemcpyDeviceToHost));
        const double n = (double)T * FPC_FRAME_SIZE - 17;
        fprintf(stderr, "[fpc stamps] cycles/sample (wave0): A=%.0f B=%.0f C=%.0f D1=%.0f D2=%.0f EF=%.0f barrier-wait=%.0f\n",
                h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n);
    }
    m->timed = true;
    return FPC_OK;
}

extern "C" float fpc_lpcnet_last_decode_ms(fpc_lpcnet* m) {
    if (!m || !m->timed) return -1.0f;
    if (hipEventSynchronize(m->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, m->ev0, m->ev1) != hipSuccess) return -1.0f;
    return ms;
}
