// lpcnet_decode2.h -- the sample loop for MORE UTTERANCES THAN COMPUTE UNITS: two utterances per workgroup
// (included by lpcnet.hip behind lpcnet_decode.h, whose helpers it uses; gfx950).
//
// k_decode (one utterance per workgroup) is a latency chain: per sample its vector issue is busy 62 % of the time and the
// LDS pipe 43 % (profiles/r06_counters.json).  With more utterances than CUs the chip runs the grid in rounds; this kernel
// instead walks TWO utterances through one workgroup in lockstep -- the same four barriers per sample (five if either
// utterance's frame is voiced), every phase carrying both -- so that every barrier, L2 round trip and LDS hop is paid once
// per PAIR of samples and a wave has the other utterance's arithmetic to issue while one waits for LDS:
//   waves 0-3  "sampler":  window (GRU_B recurrent sums, LPC taps, leaf candidates), GRU_B, dual FC of BOTH utterances;
//                          the draw of utterance 0 on wave 0, of utterance 1 on wave 1, side by side
//   waves 4-7  "mat-vec":  gates of units 0..255 of both utterances (six table rows in flight)        + the sparse products
//   waves 8-11 "mat-vec":  gates of units 256..383 of utterance 0 (waves 8, 9) / 1 (waves 10, 11)      + the sparse products
//   (three gate jobs per SIMD); the two sparse products of a pair of samples run one after the other on one accumulator set.
// Shared by both utterances: the sparse GRU_A weights (116 registers per mat-vec lane + 12 in LDS, read once per product --
// the twelve registers are the gate phase's head-room), GRU_B's (72 registers per sampler lane), the dual FC's (in LDS here,
// one ds_read_b128 serves both; the 36 registers they hold in k_decode are what the second utterance's accumulators live in),
// the activation table, diagonal and bias rows.
// Partial sums of the sparse product: lanes 2k, 2k+1 of a row group add their eight row sums with one DPP add per value
// (level 1 of the canonical tree) before lane 2k publishes: 2 planes for the update / reset gates, 4 for the candidate gate
// (fpc_lpcnet_create pads the row groups to even widths in a second placement; an all-zero lane is the tree's +0 padding).
// Table activations and mu-law lookups come in two halves (issue / finish): waves issue in order, so what hides an LDS
// round trip is the other utterance's arithmetic placed between a read and its use, by hand (sched_barrier fences).
// The same holds for the loops' LDS reads: GRU_B's state quads are read on a fixed schedule four quads ahead of the products
// that consume them, the mat-vec waves' state columns and LDS-resident weights a few column steps ahead (FPC2_STEPS).
// GRU_B's gates run on half-rows (lanes 0..7 of a unit's row finish utterance 0, lanes 8..15 utterance 1); a unit's
// per-frame and per-model constants are packed (cfa4, cdiag: one ds_read_b128 per gate job).
// The PCM goes out through a pointer the compiler can see is GLOBAL (kernarg base + offset): a FLAT store in the sample
// loop leaves the wait-count pass with a pending flat access, after which every LDS wait in the loop becomes lgkmcnt(0).
// Every value is produced by the same operations in the same order as in k_decode (and in
// oracle/fpc_oracle.c::orc_lpcnet_synthesize): the PCM is bit-identical, whichever kernel decodes an utterance.
// Launched by fpc_lpcnet_synthesize for batches larger than the device's CU count (fpc_lpcnet_set_pairing overrides);
// 512 x 3 s in 102.7 ms against 131.7 ms as two rounds of k_decode (1.28x).  What was tried, what each step bought and what
// bounds it now: profiles/r06_ablations.txt.
#pragma once

// A plane holds the sums of TWO adjacent lanes of a row group (lanes 2k, 2k+1 add their eight row sums -- the first level
// of the canonical tree -- with one DPP add per value before lane 2k publishes): half the planes, half the LDS traffic.
constexpr int PN = RNN_A + 4;         // plane stride of planes 2, 3 (candidate-gate rows only); +4 as PSTRIDE
constexpr int PART_LO = 2 * PSTRIDE;  // planes 0, 1: all 1152 gate rows
constexpr int PART_HI = 2 * PN;       // planes 2, 3
#ifndef FPC2_ABL
#define FPC2_ABL 0  // timing-only ablations, results wrong (bits: 1 no sparse-product FMAs, 2 no window work, 4 gates of utterance 0 only
                    // in the two-utterance path, 8 / 16 / 32 no GRU_B products / butterfly / gate activations); reported by fpc_build_info
#endif
#ifndef FPC2_GPRIO_PAIR
#define FPC2_GPRIO_PAIR 3
#endif
#ifndef FPC2_GPRIO_SINGLE
#define FPC2_GPRIO_SINGLE 3
#endif
#ifndef FPC2_WPRIO
#define FPC2_WPRIO 0  // priority of the sampler waves' window work (the gate waves run at 3)
#endif
#ifndef FPC2_SPRIO_GB
#define FPC2_SPRIO_GB 3  // priority of the sampler waves in the GRU_B phase ...
#endif
#ifndef FPC2_SPRIO_FC
#define FPC2_SPRIO_FC 3  // ... and in the dual-FC phase (the draw runs at 3)
#endif
#ifndef FPC2_PRIO3
#define FPC2_PRIO3 1
#endif
#ifndef FPC2_N1
#define FPC2_N1 14  // column steps (of 32: 16 of utterance 0, then 16 of utterance 1) of the sparse product under GRU_B ...
#endif
#ifndef FPC2_N2
#define FPC2_N2 11  // ... and under the dual FC; the rest under the draw
#endif

// Field order matters (as in DecodeLds): everything a lane addresses with a lane-constant register plus a constant sits in
// the first 64 KB, where the constant folds into the DS instruction's 16-bit offset field -- both utterances' copies, so one
// address register serves both; the activation table's base (indexed by a computed value) folds likewise.  Behind them
// the arrays that are addressed through a base register anyway.
struct __attribute__((aligned(16))) PairStream {
    float s1[RNN_A + 4];  // units 192..383 sit 4 floats up (s1_at): GRU_B's sixteen 24-unit slices then start in sixteen
                          // different bank quads (6 kl + (kl >> 3) mod 16), its reads are conflict-free
    float4 cfa4[RNN_A];  // this frame's GRU_A conditioning values of a unit as (z, r, h, -): one ds_read_b128 per gate job
    float s2[RNN_B];
    float hist[16];
    unsigned o_sig, o_pred, o_exc;
    float pred;
    float4 qq[128];
    float4 cand[256];
};
__device__ __forceinline__ constexpr int s1_at(int unit) { return unit + (unit >= RNN_A / 2 ? 4 : 0); }
struct __attribute__((aligned(16))) PairStreamFar {
    float p[256];
    float uframe[FPC_FRAME_SIZE];
    float part[PART_LO + PART_HI];
};
struct __attribute__((aligned(16))) Decode2Lds {
    PairStream S[2];
    float4 cdiag[RNN_A];  // per unit (diagonal z, r, h; recurrent bias of the candidate gate): one ds_read_b128 per gate job
    float ulaw_thr[64];
    float2 tt[FPC_TANH_TABLE_SIZE - 1];
    PairStreamFar F[2];
    float4 fcw[9 * NSAMP];  // dual-FC weights of node = lane: chunk c = (fcw[2c], fcw[2c+1]) of k_decode's register array
    float4 lw[3 * NMAT];    // the last twelve sparse weights of every mat-vec lane (w2[58..63] of k_decode): see k_decode2
};
static_assert(offsetof(Decode2Lds, tt) < 65536, "the activation table's base must fold into a DS offset");
static_assert(sizeof(Decode2Lds) <= 160 * 1024, "Decode2Lds exceeds the LDS of a CU");

// six independent row butterflies interleaved (see row_bfly16x3): DPP reads come >= 5 instructions behind the write
__device__ __forceinline__ void row_bfly16x6(float& a, float& b, float& c, float& d, float& e, float& f) {
#define FPC_B6(CTRL)                                                              \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" FPC_B6("quad_perm:[1,0,3,2]") FPC_B6("quad_perm:[2,3,0,1]") FPC_B6("row_half_mirror")
                     FPC_B6("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef FPC_B6
}

// the lane's index in its wave, produced where it is needed (two VALU instructions; asm volatile: not hoisted, so no
// register -- or scratch slot -- is held across the sample loop for it)
__device__ __forceinline__ unsigned lane_index_here() {
    unsigned x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// N table activations at once: all indices, all ds_read_b64, then all interpolations -- one LDS round trip for the
// batch (the compiler otherwise emits read / wait / use per activation); same operations per value as lut_scaled
template <int N>
__device__ __forceinline__ void lut_batch(const float2* T2, const float (&x)[N], const float scale, float (&y)[N]) {
    float f[N];
    float2 td[N];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float u = fminf(fabsf(x[k]) * scale, 4095.99976f);
        f[k] = __builtin_amdgcn_fractf(u);
        td[k] = T2[(uint32_t)u];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N; ++k) y[k] = copysignf(fmaf(f[k], td[k].y, td[k].x), x[k]);
}

// A table activation in two halves, so that the two utterances' chains can be skewed by half a step: the wave issues in
// order, and what hides an LDS round trip is the OTHER utterance's arithmetic placed between the read and its use.
struct LutReq {
    float x, f;
    float2 td;
};
__device__ __forceinline__ void lut_issue(const float2* T2, LutReq& q, const float x, const float scale) {
    const float u = fminf(fabsf(x) * scale, 4095.99976f);
    q.x = x;
    q.f = __builtin_amdgcn_fractf(u);
    q.td = T2[(uint32_t)u];
}
__device__ __forceinline__ float lut_finish(const LutReq& q) { return copysignf(fmaf(q.f, q.td.y, q.td.x), q.x); }

// fpc_lin2ulaw_tab (include/fpc_numerics.h) in two halves, for the same reason as LutReq: the table pair of the value's
// bin is requested, other work runs, the level is finished.  Integer arithmetic on the same bits: the same level.
struct UlawReq {
    float x;
    uint32_t iv;
    float2 tc;  // (threshold inside the bin, thresholds below the bin)
};
__device__ __forceinline__ void ulaw_issue(const float* tab, UlawReq& q, const float x) {
    q.x = x;
    q.iv = __float_as_uint(fmaf(255.0f / 32768.0f, fabsf(x), 1.0f));
    q.tc = reinterpret_cast<const float2*>(tab)[(q.iv >> 18) & 31u];
}
__device__ __forceinline__ unsigned ulaw_finish(const UlawReq& q) {
    const int e = (int)(q.iv >> 23) - 127;
    const float m = __uint_as_float((q.iv & 0x007fffffu) | 0x3f800000u);
    int K = 16 * e + (int)q.tc.y + (m >= q.tc.x ? 1 : 0);
    K = K > 128 ? 128 : K;
    const int u = q.x < 0.0f ? 128 - K : 128 + K;
    return (unsigned)(u > 255 ? 255 : u);
}
// eight independent row butterflies interleaved (see row_bfly16x6)
__device__ __forceinline__ void row_bfly16x8(float& a, float& b, float& c, float& d, float& e, float& f, float& g, float& h) {
#define FPC_B8(CTRL)                                                              \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" FPC_B8("quad_perm:[1,0,3,2]") FPC_B8("quad_perm:[2,3,0,1]") FPC_B8("row_half_mirror")
                     FPC_B8("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
#undef FPC_B8
}

// the draw of one utterance by one wave (k_decode's drawing wave): normaliser, tail cut, scan, search, control block
__device__ __forceinline__ float pair_draw(PairStream& S, const float4 p4, const bool sharpened, const float uf,
                                           const int lane, const int t) {
    float thr = 0.002f;
    if (sharpened) {
        float rs = row_bfly16((p4.x + p4.y) + (p4.z + p4.w));
        rs = add_bcast<DPP_BCAST15, 0xa>(rs);
        rs = add_bcast<DPP_BCAST31, 0xc>(rs);
        thr = 0.002f * lane_val(rs, 63);
    }
    const float c0 = __builtin_amdgcn_fmed3f(p4.x - thr, 0.0f, 1.0f);
    const float c1 = __builtin_amdgcn_fmed3f(p4.y - thr, 0.0f, 1.0f);
    const float c2 = __builtin_amdgcn_fmed3f(p4.z - thr, 0.0f, 1.0f);
    const float c3 = __builtin_amdgcn_fmed3f(p4.w - thr, 0.0f, 1.0f);
    const float P1 = c0 + c1, s23 = c2 + c3;
    const float P2 = P1 + c2, P3 = P1 + s23;
    float I = P3;
    I = I + dpp_f<DPP_ROW_SHR + 1>(I);
    I = I + dpp_f<DPP_ROW_SHR + 2>(I);
    I = I + dpp_f<DPP_ROW_SHR + 4>(I);
    I = I + dpp_f<DPP_ROW_SHR + 8>(I);
    I = add_bcast<DPP_BCAST15, 0xa>(I);
    I = add_bcast<DPP_BCAST31, 0xc>(I);
    const float rthr = uf * lane_val(I, 63);
    const float O = dpp_f<DPP_WAVE_SHR1>(I);
    const unsigned long long m0 = __builtin_amdgcn_fcmpf(O + c0, rthr, 5 /* FCMP_OLE */);
    const unsigned long long m1 = __builtin_amdgcn_fcmpf(O + P1, rthr, 5);
    const unsigned long long m2 = __builtin_amdgcn_fcmpf(O + P2, rthr, 5);
    const unsigned long long m3 = __builtin_amdgcn_fcmpf(I, rthr, 5);
    int exc = (__popcll(m0) + __popcll(m1)) + (__popcll(m2) + __popcll(m3));
    exc = exc > 255 ? 255 : exc;
    float4 cd = S.cand[exc];
    asm volatile("" : "+v"(cd.x), "+v"(cd.y), "+v"(cd.z), "+v"(cd.w));
    if (lane == 0) {
        *reinterpret_cast<float4*>(&S.o_sig) =
            make_float4(cd.z, cd.w, __uint_as_float((512u + (unsigned)exc) * (unsigned)GA), cd.y);
        S.hist[t & 15] = cd.x;
    }
    return cd.x;
}

template <bool STAMP, int QZR>
__global__ __launch_bounds__(NTHREADS) void k_decode2(const DecodeParams P, const int B) {
    __shared__ Decode2Lds L;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave: an SGPR)
    const int T = P.T;
    // utterances 2 * block and 2 * block + 1; an odd batch's last workgroup decodes its one utterance twice (both
    // copies compute and store the same values)
    const int bs[2] = {2 * (int)blockIdx.x, 2 * (int)blockIdx.x + 1 < B ? 2 * (int)blockIdx.x + 1 : 2 * (int)blockIdx.x};

    // ---- LDS init (k_decode's, per utterance) ----
    const bool resume = P.state != nullptr && P.f0 > 0;
    float* rec[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) rec[s] = P.state != nullptr ? P.state + (size_t)bs[s] * STATE_FLOATS : nullptr;
    for (int i = tid; i < RNN_A; i += NTHREADS) {
#pragma unroll
        for (int s = 0; s < 2; ++s) L.S[s].s1[s1_at(i)] = resume ? rec[s][i] : 0.0f;
        L.cdiag[i] = make_float4(P.diag[i], P.diag[RNN_A + i], P.diag[2 * RNN_A + i], P.brn_a[i]);
    }
    for (int i = tid; i < PART_LO + PART_HI; i += NTHREADS) L.F[0].part[i] = L.F[1].part[i] = 0.0f;
    if (tid < 64) L.ulaw_thr[tid] = k_ulaw_thr[tid];
    for (int k = tid; k < FPC_TANH_TABLE_SIZE - 1; k += NTHREADS) {
        const float t0 = fpc_tanh_table_entry(k), t1 = fpc_tanh_table_entry(k + 1);
        L.tt[k] = make_float2(t0, t1 - t0);
    }
    for (int i = tid; i < 9 * NSAMP; i += NTHREADS) {
        // chunk c of node sl: rows 2c, 2c+1 of k_decode's fcw[18] = (channel 0, channel 1) pairs of inputs 2c, 2c+1;
        // chunk 8 = (bias pair, factor pair)
        const int c = i / NSAMP, sl = i - c * NSAMP;
        float4 v;
        if (c < 8) {
            v.x = P.lane_fc[(2 * c) * NSAMP + sl];
            v.y = P.lane_fc[(16 + 2 * c) * NSAMP + sl];
            v.z = P.lane_fc[(2 * c + 1) * NSAMP + sl];
            v.w = P.lane_fc[(16 + 2 * c + 1) * NSAMP + sl];
        } else {
            v.x = P.lane_fc[32 * NSAMP + sl];
            v.y = P.lane_fc[33 * NSAMP + sl];
            v.z = P.lane_fc[34 * NSAMP + sl];
            v.w = P.lane_fc[35 * NSAMP + sl];
        }
        L.fcw[i] = v;
    }
    if (tid < 2 * RNN_B) {
        const int s = tid >> 4, k = tid & 15;
        L.S[s].s2[k] = resume ? rec[s][RNN_A + k] : 0.0f;
        L.S[s].hist[k] = resume ? rec[s][RNN_A + 16 + k] : 0.0f;
    }
    if (tid < 2) {
        PairStream& S = L.S[tid];
        if (resume) {
            *reinterpret_cast<float4*>(&S.o_sig) = *reinterpret_cast<const float4*>(&rec[tid][RNN_A + 32]);
        } else {
            S.o_sig = 128u * GA;
            S.o_pred = (256u + 128u) * GA;
            S.o_exc = (512u + 128u) * GA;
            S.pred = -0.0f;
        }
    }
    int16_t* out[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        out[s] = P.pcm + (size_t)bs[s] * T * FPC_FRAME_SIZE;
        if (tid < FPC_LPC_ORDER + 1 && P.f0 == 0) out[s][tid] = 0;
    }
    __syncthreads();

    if (wave >= 4) {
        // =========================== mat-vec role ===========================
        const int ml_ = tid - NSAMP;
        // 116 of the lane's 128 sparse weights live in registers; the last twelve (the pairs 58..63: column 2 of block 3 for
        // the rows 4..7, column 3 for all rows) are read from LDS once per product -- three ds_read_b128 at a lane-consecutive
        // address, issued eight columns ahead.  The twelve registers are what lets the gate phase hold both utterances'
        // table rows and a round of plane reads at once (it ran out of registers otherwise: spills and copy chains)
        f2 w2[58];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int bc = j >> 2, rp = j & 3;
            const int bb = bc >> 2, c = bc & 3;
            const f2 w = mk2(P.lane_w[(bb * 32 + (2 * rp) * 4 + c) * NMAT + ml_],
                             P.lane_w[(bb * 32 + (2 * rp + 1) * 4 + c) * NMAT + ml_]);
            if (j < 58)
                w2[j] = w;
            else
                reinterpret_cast<f2*>(&L.lw[((j - 58) >> 1) * NMAT + ml_])[(j - 58) & 1] = w;  // (read back by this lane only)
        }
        const unsigned colp_ = (unsigned)P.lane_meta[ml_];
        const unsigned metap_ = (unsigned)P.lane_meta[NMAT + ml_];
        // Gate jobs (unit, utterance): waves 4..7 evaluate units 0..255 of BOTH utterances (lane = unit), waves 8, 9 units
        // 256..383 of utterance 0, waves 10, 11 the same units of utterance 1 -- every SIMD then carries three jobs beside
        // its sampler wave (with all 768 jobs on waves 4..9, SIMDs 0 and 1 carried four, SIMDs 2 and 3 two)
        const int gu_ = wave < 10 ? ml_ : ml_ - 128;  // the unit of this lane's gate job(s)
        const unsigned gu_wave0 = (unsigned)__builtin_amdgcn_readfirstlane(gu_ & ~63);  // first unit of this wave (SGPR)
        const unsigned s1_pad = gu_wave0 >= RNN_A / 2 ? 4u : 0u;  // (s1_at of this wave's units: wave-uniform, 192 = 3 waves)
        typedef __attribute__((address_space(3))) float lds_float;
        // where this lane's 8 partial row sums of utterance 0 go (utterance 1: + sizeof(PairStreamFar)); 0 = no group
        unsigned paddr_ = 0u;
        if ((metap_ >> 16) != 0) {
            const int grp = (int)(metap_ >> 16) - 1;
            const int gate = grp / (RNN_A / 8), rb = grp - gate * (RNN_A / 8);
            const int q = (int)(metap_ & 0xff), pq = q >> 1;  // lane q of the group; its pair's plane
            float* dst = pq < 2 ? &L.F[0].part[pq * PSTRIDE + gate * RNN_A + rb * 8]
                                : &L.F[0].part[PART_LO + (pq - 2) * PN + rb * 8];  // (planes 2, 3: candidate gate only)
            if ((q & 1) == 0) paddr_ = (unsigned)(size_t)(lds_float*)dst;  // (the odd lane's sums go out through its neighbour)
        }
        paddr_ = opaque(paddr_);

        // steps k = 0..31 of the pair's sparse products: utterance k >> 4, column k & 15.  One accumulator set: an
        // utterance's sums are published as its last column is done
#define FPC2_STEPS(FROM, TO)                                                                                      \
    _Pragma("unroll") for (int k = (FROM); k < (TO); ++k) {                                                       \
        const int s = k >> 4, bc = k & 15;                                                                        \
        /* state reads run ahead of their columns: the second half's (columns 8..15) four columns early, the LDS-resident   \
           weights eight, the NEXT utterance's first half (its registers are free after column 7) four columns before  \
           this product ends -- only utterance 0's first read, right behind barrier Y, is waited for */               \
        if (bc == 0) {                                                                                            \
            _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f);                      \
        }                                                                                                         \
        if ((bc == 0 && s == 0) || (bc == 12 && s == 0)) {                                                        \
            const int sn = bc == 0 ? 0 : 1;                                                                       \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 ha = *reinterpret_cast<const float4*>(&L.S[sn].s1[(colp & 0xff) * 4]);                   \
            const float4 hb = *reinterpret_cast<const float4*>(&L.S[sn].s1[((colp >> 8) & 0xff) * 4]);            \
            hv0[0] = ha.x, hv0[1] = ha.y, hv0[2] = ha.z, hv0[3] = ha.w;                                           \
            hv0[4] = hb.x, hv0[5] = hb.y, hv0[6] = hb.z, hv0[7] = hb.w;                                           \
        }                                                                                                         \
        if (bc == 4) {                                                                                            \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 hc = *reinterpret_cast<const float4*>(&L.S[s].s1[((colp >> 16) & 0xff) * 4]);            \
            const float4 hd = *reinterpret_cast<const float4*>(&L.S[s].s1[(colp >> 24) * 4]);                     \
            hv1[0] = hc.x, hv1[1] = hc.y, hv1[2] = hc.z, hv1[3] = hc.w;                                           \
            hv1[4] = hd.x, hv1[5] = hd.y, hv1[6] = hd.z, hv1[7] = hd.w;                                           \
        }                                                                                                         \
        if (bc == 8) {                                                                                            \
            const unsigned mlx = opaque((unsigned)ml_);                                                           \
            const float4 l0 = L.lw[mlx], l1 = L.lw[NMAT + mlx], l2 = L.lw[2 * NMAT + mlx];                        \
            wl[0] = mk2(l0.x, l0.y), wl[1] = mk2(l0.z, l0.w), wl[2] = mk2(l1.x, l1.y);                            \
            wl[3] = mk2(l1.z, l1.w), wl[4] = mk2(l2.x, l2.y), wl[5] = mk2(l2.z, l2.w);                            \
        }                                                                                                         \
        _Pragma("unroll") for (int rp = 0; rp < (FPC2_ABL & 1 ? 0 : 4); ++rp) {                                   \
            if (bc < 8)                                                                                           \
                acc[rp] = fma2(w2[bc * 4 + rp], splat2(hv0[bc]), acc[rp]);                                        \
            else                                                                                                  \
                a[rp] = fma2(bc * 4 + rp < 58 ? w2[bc * 4 + rp < 58 ? bc * 4 + rp : 0] : wl[bc * 4 + rp - 58],        \
                             splat2(hv1[bc - 8]), a[rp]);                                                         \
        }                                                                                                         \
        if (bc == 15) {                                                                                           \
            _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) acc[rp] = acc[rp] + a[rp];                           \
            _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                                    \
                acc[rp].x = acc[rp].x + dpp_f<DPP_ROW_SHL + 1>(acc[rp].x); /* + the next lane's: tree level 1 */  \
                acc[rp].y = acc[rp].y + dpp_f<DPP_ROW_SHL + 1>(acc[rp].y);                                        \
            }                                                                                                     \
            if (paddr_ != 0u) {                                                                                   \
                typedef float v4f __attribute__((ext_vector_type(4)));                                            \
                typedef __attribute__((address_space(3))) v4f lds_v4f;                                            \
                lds_v4f* pp = (lds_v4f*)(size_t)(paddr_ + (unsigned)(s * sizeof(PairStreamFar)));                    \
                v4f lo, hi;                                                                                       \
                lo.x = acc[0].x, lo.y = acc[0].y, lo.z = acc[1].x, lo.w = acc[1].y;                               \
                hi.x = acc[2].x, hi.y = acc[2].y, hi.z = acc[3].x, hi.w = acc[3].y;                               \
                pp[0] = lo;                                                                                       \
                pp[1] = hi;                                                                                       \
            }                                                                                                     \
        }                                                                                                         \
    }                                                                                                             \
    _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                                            \
        pin(acc[rp]);                                                                                             \
        pin(a[rp]);                                                                                               \
    }
        if (P.state != nullptr) {
            if (resume) {  // the sparse products of the states the chunk starts from
                f2 acc[4], a[4], wl[6];
                float hv0[8], hv1[8];
                FPC2_STEPS(0, 32)
            }
            __syncthreads();
        }
        for (int fr = P.f0; fr < P.f1; ++fr) {
            const bool voiced =
                fpc_shape_exponent(P.features[((size_t)bs[0] * T + fr) * FPC_NB_FEATURES + 19]) > 0.0f ||
                fpc_shape_exponent(P.features[((size_t)bs[1] * T + fr) * FPC_NB_FEATURES + 19]) > 0.0f;
            // this frame's conditioning rows of the unit(s) whose gates this lane evaluates: written and read by the same lane
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (wave < 8 || (wave >= 10) == (s == 1)) {
                    const float* cfa = P.cfa + ((size_t)bs[s] * P.cf_T + (fr - P.f0)) * GA;
                    L.S[s].cfa4[gu_] = make_float4(cfa[gu_], cfa[RNN_A + gu_], cfa[2 * RNN_A + gu_], 0.0f);
                }
            }
            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                const int st_t = fr * FPC_FRAME_SIZE + i;
                const bool stamp_on = STAMP && blockIdx.x == 0 && st_t >= FPC_STAMP_T0 && st_t < FPC_STAMP_T0 + FPC_STAMP_NS;
                // ---- X..Y: both utterances' table rows gathered together, gates of both ----
                // (both gate paths are on the sample-to-sample chain; the one-utterance path is the shorter one)
                if (wave >= 8)
                    __builtin_amdgcn_s_setprio(FPC2_GPRIO_SINGLE);
                else
                    __builtin_amdgcn_s_setprio(FPC2_GPRIO_PAIR);
                struct F3 {
                    float x, y, z;
                };
                const char* tabc = reinterpret_cast<const char*>(P.tab);
                if (wave >= 8) {
                    // one utterance's gates of unit ml (k_decode's gate phase on the pair planes)
                    const unsigned ml = lane_index_here() + gu_wave0;
                    PairStream& S = L.S[wave >= 10 ? 1 : 0];
                    const float* pl = &L.F[wave >= 10 ? 1 : 0].part[ml];
                    const unsigned oa = S.o_sig, ob = S.o_pred, oc = S.o_exc;
                    const F3 ta = *reinterpret_cast<const F3*>(tabc + (size_t)((oa + 3u * ml) * 4u));
                    const F3 tb = *reinterpret_cast<const F3*>(tabc + (size_t)((ob + 3u * ml) * 4u));
                    const F3 tc = *reinterpret_cast<const F3*>(tabc + (size_t)((oc + 3u * ml) * 4u));
                    __builtin_amdgcn_sched_barrier(0);
                    const float h_own = S.s1[ml + 4];  // (units 256..383: s1_at = + 4)
                    float unb, uz, ur;
                    {
                        const float4 cd = L.cdiag[ml];
                        const float dn = cd.z, bn = cd.w;
                        const float n0 = pl[2 * RNN_A], n1 = pl[2 * RNN_A + PSTRIDE], n2 = pl[PART_LO], n3 = pl[PART_LO + PN];
                        const float dz = cd.x, dr = cd.y;
                        const float z0 = pl[0], r0 = pl[RNN_A];
                        float tz = z0, tr = r0;
                        if (QZR == 4) {
                            const float z1 = pl[PSTRIDE], r1 = pl[RNN_A + PSTRIDE];
                            __builtin_amdgcn_sched_barrier(0);
                            tz = z0 + z1;
                            tr = r0 + r1;
                        } else {
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        unb = fmaf(dn, h_own, (n0 + n1) + (n2 + n3)) + bn;
                        uz = fmaf(dz, h_own, tz);
                        ur = fmaf(dr, h_own, tr);
                    }
                    const float4 cf = S.cfa4[ml];
                    const float cz = cf.x, cr = cf.y, cn = cf.z;
                    __builtin_amdgcn_sched_barrier(0);
                    LutReq qz, qr, qn;
                    lut_issue(L.tt, qz, (((ta.x + tb.x) + tc.x) + cz) + uz, 256.0f);
                    lut_issue(L.tt, qr, (((ta.y + tb.y) + tc.y) + cr) + ur, 256.0f);
                    const float gn = ((ta.z + tb.z) + tc.z) + cn;
                    __builtin_amdgcn_sched_barrier(0);
                    const float z = fmaf(0.5f, lut_finish(qz), 0.5f);
                    lut_issue(L.tt, qn, fmaf(fmaf(0.5f, lut_finish(qr), 0.5f), unb, gn), 512.0f);
                    __builtin_amdgcn_sched_barrier(0);
                    const float n = lut_finish(qn);
                    S.s1[ml + 4] = fmaf(z, h_own - n, n);
                } else {
                    const unsigned ml = lane_index_here() + gu_wave0;
                    F3 ta[2], tb[2], tc[2];
#pragma unroll
                    for (int s = 0; s < (FPC2_ABL & 4 ? 1 : 2); ++s) {
                        const unsigned oa = L.S[s].o_sig, ob = L.S[s].o_pred, oc = L.S[s].o_exc;
                        ta[s] = *reinterpret_cast<const F3*>(tabc + (size_t)((oa + 3u * ml) * 4u));
                        tb[s] = *reinterpret_cast<const F3*>(tabc + (size_t)((ob + 3u * ml) * 4u));
                        tc[s] = *reinterpret_cast<const F3*>(tabc + (size_t)((oc + 3u * ml) * 4u));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // while the six rows are in flight: recurrent terms of the unit's three rows (diagonal + the upper levels of
                    // the tree over the pair planes).  The diagonal / bias values once for both utterances; each utterance's
                    // state value and eight plane values as one round of reads, the second requested before the first is used
                    const float4 cd = L.cdiag[ml];
                    const float dz = cd.x, dr = cd.y, dn = cd.z, bn = cd.w;
                    float h_own[2], cz[2], cr[2], cn[2], unb[2], uz[2], ur[2];
#pragma unroll
                    for (int s = 0; s < (FPC2_ABL & 4 ? 1 : 2); ++s) {
                        const float* pl = &L.F[s].part[ml];
                        float pz[2], pr[2], pn[4];
                        h_own[s] = L.S[s].s1[ml + s1_pad];
                        pn[0] = pl[2 * RNN_A], pn[1] = pl[2 * RNN_A + PSTRIDE], pn[2] = pl[PART_LO], pn[3] = pl[PART_LO + PN];
                        pz[0] = pl[0], pr[0] = pl[RNN_A];
                        if (QZR == 4) pz[1] = pl[PSTRIDE], pr[1] = pl[RNN_A + PSTRIDE];
                        __builtin_amdgcn_sched_barrier(0);
                        unb[s] = fmaf(dn, h_own[s], (pn[0] + pn[1]) + (pn[2] + pn[3])) + bn;
                        uz[s] = fmaf(dz, h_own[s], QZR == 4 ? pz[0] + pz[1] : pz[0]);
                        ur[s] = fmaf(dr, h_own[s], QZR == 4 ? pr[0] + pr[1] : pr[0]);
                        const float4 cf = L.S[s].cfa4[ml];
                        cz[s] = cf.x, cr[s] = cf.y, cn[s] = cf.z;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // gates: utterance 1 half a step behind utterance 0 (its arithmetic hides the other's table read)
                    LutReq qz[2], qr[2], qn[2];
                    float gn[2];
#pragma unroll
                    for (int s = 0; s < (FPC2_ABL & 4 ? 1 : 2); ++s) {
                        lut_issue(L.tt, qz[s], (((ta[s].x + tb[s].x) + tc[s].x) + cz[s]) + uz[s], 256.0f);
                        lut_issue(L.tt, qr[s], (((ta[s].y + tb[s].y) + tc[s].y) + cr[s]) + ur[s], 256.0f);
                        gn[s] = ((ta[s].z + tb[s].z) + tc[s].z) + cn[s];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    float zg[2];
#pragma unroll
                    for (int s = 0; s < (FPC2_ABL & 4 ? 1 : 2); ++s) {
                        zg[s] = fmaf(0.5f, lut_finish(qz[s]), 0.5f);
                        lut_issue(L.tt, qn[s], fmaf(fmaf(0.5f, lut_finish(qr[s]), 0.5f), unb[s], gn[s]), 512.0f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < (FPC2_ABL & 4 ? 1 : 2); ++s) {
                        const float n = lut_finish(qn[s]);
                        L.S[s].s1[ml + s1_pad] = fmaf(zg[s], h_own[s] - n, n);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#if FPC_PRIO
                // the third wave of each SIMD (8..11) would get what the sampler wave and the second wave leave over:
                // its product steps end each phase last.  One priority step above the second wave evens them out
                if (FPC2_PRIO3 && wave >= 8)
                    __builtin_amdgcn_s_setprio(FPC2_PRIO3);
                else
                    __builtin_amdgcn_s_setprio(0);
#endif
                FPC_BARRIER(0)  // Y
                f2 acc[4], a[4], wl[6];
                float hv0[8], hv1[8];
                FPC2_STEPS(0, FPC2_N1)
                FPC_BARRIER(1)  // Z1
                FPC2_STEPS(FPC2_N1, FPC2_N1 + FPC2_N2)
                FPC_BARRIER(2)  // Z2
                if (voiced) {
                    FPC_BARRIER(3)  // Z3
                }
                FPC2_STEPS(FPC2_N1 + FPC2_N2, 32)
                FPC_BARRIER(4)  // X
            }
        }
#undef FPC2_STEPS
#pragma unroll
        for (int s = 0; s < 2; ++s)  // (the carried state: by the lanes that hold the gate jobs)
            if (rec[s] != nullptr && (wave < 8 || (wave >= 10) == (s == 1))) rec[s][gu_] = L.S[s].s1[s1_at(gu_)];
    } else {
        // =========================== sampler role ===========================
        __builtin_amdgcn_s_setprio(3);
        const int sl = tid;
        const int u = sl >> 4, kl = sl & 15;
        const int dw = wave & 1;  // the utterance this wave draws (waves 0, 1) and whose de-emphasis state it tracks
        f2 wB[3][6][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    wB[g][m][h] = mk2(P.lane_wb[(g * 24 + 4 * m + 2 * h) * NSAMP + sl],
                                      P.lane_wb[(g * 24 + 4 * m + 2 * h + 1) * NSAMP + sl]);
        const float ub0 = P.lane_ub[sl], ub1 = P.lane_ub[NSAMP + sl], ub2 = P.lane_ub[2 * NSAMP + sl];
        const float brnb = P.brn_b[u];
        const float my_ulaw = P.ulaw_tab[sl];
        // this wave's utterance's PCM: kernel argument + offset, so that the store in the sample loop is a global_store -- as a
        // FLAT store (what indexing a pointer array by the wave gives) it leaves the waitcnt bookkeeping with a pending flat
        // operation, and every LDS wait of the loop then becomes lgkmcnt(0): the staged reads lose their stagger
        int16_t* const out_dw = P.pcm + (size_t)(dw ? bs[1] : bs[0]) * T * FPC_FRAME_SIZE;
        float mem = resume ? rec[dw][RNN_A + 36] : 0.0f;
        float pcm_new = 0.0f;
        // GRU_B's gates: after the row butterfly all 16 lanes of a unit's row hold both utterances' three sums; lanes 0..7
        // of the row evaluate utterance 0's gates, lanes 8..15 utterance 1's (one triple of table activations per lane
        // instead of two), and lanes 0 / 8 store the new state value
        const bool hb = (kl & 8) != 0;
        float s2_mine = resume ? rec[hb ? 1 : 0][RNN_A + u] : 0.0f;
        if (P.state != nullptr) __syncthreads();

        for (int fr = P.f0; fr < P.f1; ++fr) {
            float shape_e[2], a_cur[2], a_nxt[2], a0_cur[2], a0_nxt[2];
            float cfb_z, cfb_r, cfb_n;  // of this lane's utterance (hb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* feat = P.features + ((size_t)bs[s] * T + fr) * FPC_NB_FEATURES;
                shape_e[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fpc_shape_exponent(feat[19]))));
                const float* cfb = P.cfb + ((size_t)bs[s] * P.cf_T + (fr - P.f0)) * GB;
                if (hb == (s == 1)) cfb_z = cfb[u], cfb_r = cfb[RNN_B + u], cfb_n = cfb[2 * RNN_B + u];
                const float* fa = feat + (FPC_NB_FEATURES - FPC_LPC_ORDER);
                const float* fan = fa + (fr + 1 < T ? FPC_NB_FEATURES : 0);
                a_cur[s] = fa[kl], a_nxt[s] = fan[kl];
                a0_cur[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fa[0])));
                a0_nxt[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fan[0])));
                if (sl < FPC_FRAME_SIZE)
                    L.F[s].uframe[sl] = fpc_philox_uniform(P.seeds[bs[s]], (uint32_t)(fr * FPC_FRAME_SIZE + sl));
            }
            const bool voiced = shape_e[0] > 0.0f || shape_e[1] > 0.0f;

            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                const int t = fr * FPC_FRAME_SIZE + i;
                const int st_t = t;
                const bool stamp_on = STAMP && blockIdx.x == 0 && st_t >= FPC_STAMP_T0 && st_t < FPC_STAMP_T0 + FPC_STAMP_NS;
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(FPC2_WPRIO);
#endif
                // ---- X..Y: GRU_B recurrent parts, LPC taps, leaf candidates of both utterances: one round of state reads,
                //      one eight-way butterfly (GRU_B's three recurrent sums and the tap tree, per utterance), the four
                //      mu-law levels requested together ----
                float ub_z[2], ub_r[2], ub_n[2];
#if !(FPC2_ABL & 2)
                {
                    const bool lastsmp = i == FPC_FRAME_SIZE - 1;
                    float s2k[2], hk[2], prd[2], tap[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        s2k[s] = L.S[s].s2[kl];
                        hk[s] = L.S[s].hist[(t - kl) & 15];
                        prd[s] = L.S[s].pred;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        ub_z[s] = ub0 * s2k[s], ub_r[s] = ub1 * s2k[s], ub_n[s] = ub2 * s2k[s];
                        const float am = lastsmp ? a_nxt[s] : a_cur[s];
                        tap[s] = kl ? am * hk[s] : 0.0f;
                    }
                    row_bfly16x8(ub_z[0], ub_r[0], ub_n[0], tap[0], ub_z[1], ub_r[1], ub_n[1], tap[1]);
                    float cpcm[2], cpred[2];
                    UlawReq qs[2], qp[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float a0 = lastsmp ? a0_nxt[s] : a0_cur[s];
                        cpcm[s] = prd[s] + my_ulaw;
                        cpred[s] = -fmaf(a0, cpcm[s], tap[s]);
                        ulaw_issue(L.ulaw_thr, qs[s], cpcm[s]);
                        ulaw_issue(L.ulaw_thr, qp[s], cpred[s]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned es = ulaw_finish(qs[s]);
                        const unsigned ep = 256u + ulaw_finish(qp[s]);
                        L.S[s].cand[sl] = make_float4(cpcm[s], cpred[s], __uint_as_float((es << 10) + (es << 7)),
                                                      __uint_as_float((ep << 10) + (ep << 7)));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#else
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float s2k = L.S[s].s2[kl];
                    ub_z[s] = ub0 * s2k, ub_r[s] = ub1 * s2k, ub_n[s] = ub2 * s2k;
                }
                row_bfly16x6(ub_z[0], ub_r[0], ub_n[0], ub_z[1], ub_r[1], ub_n[1]);
#endif
#if FPC2_ABL & 2
                L.S[0].cand[sl] = L.S[1].cand[sl] = make_float4(0.0f, 0.0f, __uint_as_float(128u * GA), __uint_as_float(384u * GA));
#endif
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(FPC2_SPRIO_GB);
#endif
                FPC_BARRIER(0)  // Y
                // ---- Y..Z1: GRU_B of both utterances (one weight register feeds two chains) ----
                const unsigned slv = opaque((unsigned)sl);
                float4 wa[2], wb[2], bf;  // dual-FC weight chunks in flight (loaded from the end of this phase on)
#define FPC2_FC_LOADW(k4)                                                                \
    wa[(k4) & 1] = L.fcw[(2 * (k4)) * NSAMP + slv];                                      \
    wb[(k4) & 1] = L.fcw[(2 * (k4) + 1) * NSAMP + slv];
                {
                    const unsigned klv = (unsigned)kl;
                    f2 acc[2][3][2];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[s][g][0] = acc[s][g][1] = splat2(0.0f);
#if !(FPC2_ABL & 8)
                    {
                        // twelve state reads (six per utterance) through FOUR register quads, always four reads ahead of the
                        // products: the quads are consumed in the order A0..A3 B0..B3 A4 A5 B4 B5 (per utterance ascending: the
                        // chains' order), and the quad a step frees takes the read four steps ahead -- every read has three
                        // steps (18 packed FMAs) in front of its use.  (Left to itself the compiler alternates the utterances
                        // and consumes each late read right after issuing it: eight exposed LDS round trips.)
                        float4 hq[4];
#define FPC2_GB_STEP_S(q) ((q) < 4 ? 0 : (q) < 8 ? 1 : (q) < 10 ? 0 : 1)
#define FPC2_GB_STEP_M(q) ((q) < 4 ? (q) : (q) < 8 ? (q) - 4 : (q) < 10 ? (q) - 4 : (q) - 6)
#define FPC2_GB_READ(q) \
    hq[(q) & 3] = *reinterpret_cast<const float4*>(&L.S[FPC2_GB_STEP_S(q)].s1[24 * klv + 4 * (klv >> 3) + 4 * FPC2_GB_STEP_M(q)]);
                        FPC2_GB_READ(0)
                        FPC2_GB_READ(1)
                        FPC2_GB_READ(2)
                        FPC2_GB_READ(3)
#pragma unroll
                        for (int q = 0; q < 12; ++q) {
                            __builtin_amdgcn_sched_barrier(0);
                            const int s = FPC2_GB_STEP_S(q), m = FPC2_GB_STEP_M(q);
                            const float4 h4 = hq[q & 3];
#pragma unroll
                            for (int g = 0; g < 3; ++g) {
                                acc[s][g][0] = fma2(wB[g][m][0], mk2(h4.x, h4.y), acc[s][g][0]);
                                acc[s][g][1] = fma2(wB[g][m][1], mk2(h4.z, h4.w), acc[s][g][1]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (q + 4 < 12) {
                                FPC2_GB_READ(q + 4)
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#undef FPC2_GB_READ
#undef FPC2_GB_STEP_S
#undef FPC2_GB_STEP_M
                    }
#endif
                    float a3[2][3];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const f2 pr = acc[s][g][0] + acc[s][g][1];
                            a3[s][g] = pr.x + pr.y;
                        }
                    if (!(FPC2_ABL & 16)) row_bfly16x6(a3[0][0], a3[0][1], a3[0][2], a3[1][0], a3[1][1], a3[1][2]);
                    // the accumulators are dead: the dual FC's first weight chunks (they depend on nothing) start their way
                    // from LDS here, under the gates and the barrier
                    FPC2_FC_LOADW(0)
                    FPC2_FC_LOADW(1)
                    bf = L.fcw[8 * NSAMP + slv];
                    // gates of this lane's utterance
                    {
                        const float xz = hb ? a3[1][0] : a3[0][0], xr = hb ? a3[1][1] : a3[0][1], xn = hb ? a3[1][2] : a3[0][2];
                        const float uz_ = hb ? ub_z[1] : ub_z[0], ur_ = hb ? ub_r[1] : ub_r[0], un_ = hb ? ub_n[1] : ub_n[0];
#if FPC2_ABL & 32
                        const float z = 0.5f + 1e-3f * xz, n = 1e-3f * (xr + xn);
#else
                        LutReq qz, qr, qn;
                        lut_issue(L.tt, qz, (xz + cfb_z) + uz_, 256.0f);
                        lut_issue(L.tt, qr, (xr + cfb_r) + ur_, 256.0f);
                        __builtin_amdgcn_sched_barrier(0);
                        const float z = fmaf(0.5f, lut_finish(qz), 0.5f);
                        lut_issue(L.tt, qn, fmaf(fmaf(0.5f, lut_finish(qr), 0.5f), un_ + brnb, xn + cfb_n), 512.0f);
                        __builtin_amdgcn_sched_barrier(0);
                        const float n = lut_finish(qn);
#endif
                        s2_mine = fmaf(z, s2_mine - n, n);
                        if ((kl & 7) == 0) (hb ? L.S[1].s2 : L.S[0].s2)[u] = s2_mine;
                    }
                }
#if FPC_PRIO
                if (FPC2_SPRIO_FC != FPC2_SPRIO_GB) __builtin_amdgcn_s_setprio(FPC2_SPRIO_FC);
#endif
                FPC_BARRIER(1)  // Z1
                // ---- Z1..Z2: dual FC of tree node `sl`, both utterances; weights from LDS, read once, two chunks ahead ----
                {
                    float4 sv[2][2];
#define FPC2_FC_LOADS(k4)                                                                \
    sv[(k4) & 1][0] = *reinterpret_cast<const float4*>(&L.S[0].s2[4 * (k4)]);            \
    sv[(k4) & 1][1] = *reinterpret_cast<const float4*>(&L.S[1].s2[4 * (k4)]);
                    FPC2_FC_LOADS(0)
                    FPC2_FC_LOADS(1)
                    f2 a01[2], b01[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) a01[s] = mk2(bf.x, bf.y), b01[s] = splat2(0.0f);
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        __builtin_amdgcn_sched_barrier(0);
                        const float4 w0 = wa[k4 & 1], w1 = wb[k4 & 1];
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const float4 v = sv[k4 & 1][s];
                            a01[s] = fma2(mk2(w0.x, w0.y), splat2(v.x), a01[s]);
                            b01[s] = fma2(mk2(w0.z, w0.w), splat2(v.y), b01[s]);
                            a01[s] = fma2(mk2(w1.x, w1.y), splat2(v.z), a01[s]);
                            b01[s] = fma2(mk2(w1.z, w1.w), splat2(v.w), b01[s]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (k4 < 2) {  // (the buffers just consumed take the chunks two steps ahead)
                            FPC2_FC_LOADW(k4 + 2)
                            FPC2_FC_LOADS(k4 + 2)
                        }
                    }
#undef FPC2_FC_LOADS
                    // tanh of both channels, factor sum, sigmoid: utterance 1 half a step behind utterance 0
                    LutReq q0[2], q1[2], qq[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const f2 c = a01[s] + b01[s];
                        lut_issue(L.tt, q0[s], c.x, 512.0f);
                        lut_issue(L.tt, q1[s], c.y, 512.0f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float t0 = lut_finish(q0[s]), t1 = lut_finish(q1[s]);
                        lut_issue(L.tt, qq[s], fmaf(bf.w, t1, bf.z * t0), 256.0f);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float qv = fmaf(0.5f, lut_finish(qq[s]), 0.5f);
                        reinterpret_cast<float2*>(L.S[s].qq)[slv] = make_float2(1.0f - qv, qv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#undef FPC2_FC_LOADW
                const float uf = L.F[dw].uframe[i];
#if FPC_PRIO
                if (FPC2_SPRIO_FC != 3) __builtin_amdgcn_s_setprio(3);
#endif
                FPC_BARRIER(2)  // Z2
                // ---- voiced frames: leaf probability + sharpening on all 256 lanes, per voiced utterance ----
                if (voiced) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (shape_e[s] > 0.0f) {
                            const float* qf = reinterpret_cast<const float*>(L.S[s].qq);
                            const unsigned slv = opaque((unsigned)sl);
                            float f[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] = qf[(2u << j) + (slv >> (7 - j))];
                            const float p = ((((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5])) * f[6]) * f[7];
                            L.F[s].p[slv] = fpc_shape_pow(p, shape_e[s]);
                        }
                    }
                    FPC_BARRIER(3)  // Z3
                }
                // ---- waves 0 / 1: the draw of utterance 0 / 1 ----
                auto draw_phase = [&](PairStream& S, const float* pp, const float she) {
                    const bool sharp = she > 0.0f;
                    float4 p4;
                    if (sharp) {
                        p4 = *reinterpret_cast<const float4*>(&pp[4 * lane]);
                    } else {
                        const float* qf = reinterpret_cast<const float*>(S.qq);
                        const unsigned lv = (unsigned)lane;
                        float f[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) f[j] = qf[(2u << j) + (lv >> (5 - j))];
                        const float2 q6 = *reinterpret_cast<const float2*>(&qf[128u + 2u * lv]);
                        const float4 q7 = *reinterpret_cast<const float4*>(&qf[256u + 4u * lv]);
                        const float pre = ((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5]);
                        const float lo = pre * q6.x, hi = pre * q6.y;
                        p4.x = lo * q7.x;
                        p4.y = lo * q7.y;
                        p4.z = hi * q7.z;
                        p4.w = hi * q7.w;
                    }
                    pcm_new = pair_draw(S, p4, sharp, uf, lane, t);
                };
                if (wave == 0)
                    draw_phase(L.S[0], L.F[0].p, shape_e[0]);
                else if (wave == 1)
                    draw_phase(L.S[1], L.F[1].p, shape_e[1]);
                FPC_BARRIER(4)  // X
                if (wave < 2) {
                    mem = fmaf(FPC_PREEMPH, mem, pcm_new);
                    if (lane == 0) out_dw[t] = fpc_pcm16(mem);
                }
            }
        }
        if (sl < 2 * RNN_B) {
            const int s = sl >> 4, k = sl & 15;
            if (rec[s] != nullptr) {
                rec[s][RNN_A + k] = L.S[s].s2[k];
                rec[s][RNN_A + 16 + k] = L.S[s].hist[k];
            }
        }
        if (wave < 2 && lane == 0 && rec[dw] != nullptr) {
            *reinterpret_cast<float4*>(&rec[dw][RNN_A + 32]) = *reinterpret_cast<const float4*>(&L.S[dw].o_sig);
            rec[dw][RNN_A + 36] = mem;
        }
    }
}
#undef FPC_BARRIER
#undef FPC_STAMP
