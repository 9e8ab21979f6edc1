// lpcnet_decode.h -- the persistent per-utterance sample loop of the LPCNet-style vocoder
// (included by lpcnet.hip only; gfx950).
//
// One 768-thread workgroup (12 wave64, 3 per SIMD, 168 VGPRs each) per utterance:
//   waves 0-3  "sampler":  GRU_B, dual FC, tree pdf, draw   (GRU_B and FC weights in VGPRs)
//   waves 4-11 "mat-vec":  embedding-row gather + GRU_A gates (384 "gate lanes"), and the block-sparse
//                          recurrent product for the NEXT sample (4 blocks of 8x4 weights per lane in VGPRs)
// Four workgroup barriers per output sample (X, Y, Z1, Z2; voiced frames add Z3):
//   X  control block (table-row offsets selected by the drawn sample) and all partial sums published
//        M: gather 3 table rows; under the gather, reduce the partial sums of the sparse product of the
//           unit's three rows (balanced tree) + diagonal; GRU_A gates -> s1'
//        S: GRU_B recurrent part, LPC taps 2..16 (DPP tree), per-leaf candidate records of this sample
//   Y  s1' ready
//        S: GRU_B -> s2'                           M: 6 of the 16 columns of the sparse product of s1'
//   Z1 s2' ready
//        S: dual FC -> 255 node probabilities      M: 4 more columns
//   Z2 q ready
//        S wave 0 (unvoiced frame): 4 leaf probabilities per lane, tail cut, scan, draw, control block
//        (voiced frame: all 256 sampler lanes compute leaf probability + sharpening, barrier Z3, then
//         wave 0 normalises and draws)              M: last 6 columns, partial sums -> LDS
// The sparse product (61 % of the algorithmic FLOPs) is hidden completely under the sampler phases;
// HBM is touched only for the gathered table rows (L2-resident), the per-frame conditioning rows and
// 2 bytes of PCM per sample.
//
// Canonical evaluation orders (DESIGN.md "Vocoder numerics") are those of
// oracle/fpc_oracle.c::orc_lpcnet_synthesize; results are bit-identical.
#pragma once

constexpr int STATE_FLOATS = 512;  // per-stream record of a chunked pass (DecodeParams::state)
constexpr int NTHREADS = 768;
constexpr int NSAMP = 256;  // sampler lanes (waves 0-3)
constexpr int NMAT = 512;   // mat-vec lanes (waves 4-11)
constexpr int NMW = NMAT / 64;
// stride between the partial-sum planes of consecutive lanes q of a row group: +4 floats so that the
// lanes of one group (consecutive lanes of a wave) start their 16-byte stores in different bank groups
constexpr int PSTRIDE = GA + 4;
#ifndef FPC_PRIO
#define FPC_PRIO 3  // priority of the mat-vec waves while they gather and gate (the sampler waves drop to 0 there)
#endif
#ifndef FPC_NA
#define FPC_NA 6  // sparse-product columns (of 16) computed under GRU_B ...
#endif
#ifndef FPC_PCM_WHERE
#define FPC_PCM_WHERE 0  // de-emphasis + PCM store: 0 drawing wave behind barrier X, 1 drawing wave before it, 2 wave 3 behind it
#endif
#ifndef FPC_NB
#define FPC_NB 4  // ... and under the dual FC; the rest runs under the draw
#endif

struct DecodeParams {
    const float* tab;       // [3][256][384][3]  embedding x input-kernel tables, gate-interleaved
    const float* cfa;       // [B][cf_T][1152]  GRU_A conditioning product (+biases) of frames f0 .. f0 + cf_T - 1
    const float* cfb;       // [B][cf_T][48]    GRU_B conditioning product (+biases)
    const float* features;  // [B][T][36]
    const unsigned long long* seeds;
    int16_t* pcm;  // [B][T*160]
    int T;
    int f0, f1, cf_T;  // this launch decodes frames [f0, f1) (the whole utterance: 0, T, T)
    float* state;      // [B][STATE_FLOATS] per-stream record carried between the launches of a chunked pass (nullptr: none):
                       // s1[384] | s2[16] | history ring[16] | control block[4] | de-emphasis state
    const float* lane_w;     // [128][512] sparse GRU_A weights: 2 leaves x 2 blocks x 8x4
    const int* lane_meta;    // [2][512]   packed column blocks; (group+1)<<16 | lanes<<8 | lane
    const float* lane_wb;    // [72][256]  GRU_B input weights [gate][24 inputs] of (unit, slice)
    const float* lane_ub;    // [3][256]   GRU_B recurrent weights ub[k][gate*16+unit]
    const float* lane_fc;    // [36][256]  dual-FC of node = lane: 2x16 weights, 2 bias, 2 factor
    const float* diag;       // [1152]
    const float* brn_a;      // [384]
    const float* brn_b;      // [16]
    const float* ulaw_tab;   // [256]
    unsigned* stamps;  // diagnostic only: [FPC_STAMP_NS][12 waves][8 slots]
};

__device__ const float k_ulaw_thr[64] = FPC_ULAW_TABLE_INIT;

// Field order matters: lane-indexed arrays sit in the first 64 KB (their base folds into the DS
// instruction's offset field instead of a VGPR), the activation table at offset 0 (ds_read2 has 8-bit offsets).
struct __attribute__((aligned(16))) DecodeLds {
    float2 tt[FPC_TANH_TABLE_SIZE - 1];  // fpc_tanh_lut table as (T[k], T[k+1] - T[k]) pairs, built at kernel start
    float s1[RNN_A];
    float cfa[GA];            // this frame's GRU_A conditioning rows [z|r|h][unit]
    float diag[GA];
    float brn_a[RNN_A];
    float s2[RNN_B];
    float hist[16];
    // control block written by the winning lane / the LPC chain lane
    unsigned o_sig, o_pred, o_exc;  // float offsets of the three table rows to gather next
    float pred;                     // prediction of the next sample
    float4 qq[128];  // node n's branch factors as the pair (1 - q[n], q[n]) at floats 2n, 2n+1 (16-byte aligned rows)
    float p[256];
    float4 cand[256];  // per leaf, if it wins the draw: (pcm, next prediction, bits of o_sig, bits of o_pred)
    float ulaw_thr[64];  // fpc_lin2ulaw_tab table
    float uframe[FPC_FRAME_SIZE];
    float part[16 * PSTRIDE];      // partial row sums of the sparse product: [lane q of the row group][gate row]
};

// ---- DPP helpers (gfx9 DPP controls; invalid source lanes read 0) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138;

// balanced (adjacent-pair) sum over each aligned row of 16 lanes; result in every lane
__device__ __forceinline__ float row_bfly16(float v) {
    v = v + dpp_f<DPP_XOR1>(v);
    v = v + dpp_f<DPP_XOR2>(v);
    v = v + dpp_f<DPP_HALF_MIRROR>(v);
    v = v + dpp_f<DPP_MIRROR>(v);
    return v;
}
// three independent row butterflies interleaved: every DPP read of a register comes >= 2 instructions
// after its last write (the gfx9 VALU-write -> DPP-read hazard), so no s_nop and no separate v_mov_dpp
__device__ __forceinline__ void row_bfly16x3(float& a, float& b, float& c) {
#define FPC_B3(CTRL)                                                              \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" FPC_B3("quad_perm:[1,0,3,2]") FPC_B3("quad_perm:[2,3,0,1]") FPC_B3("row_half_mirror")
                     FPC_B3("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c));
#undef FPC_B3
}
// v + (value broadcast from the last lane of the previous row(s)), written only to the rows in ROWMASK
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float add_bcast(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
constexpr int DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;
__device__ __forceinline__ float lane_val(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float x, float y) {
    f2 r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ f2 splat2(float v) { return mk2(v, v); }
// pins a value where it is computed (the compiler would otherwise sink the whole computation to its only use)
__device__ __forceinline__ void pin(f2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f32

// opaque copy: the compiler recomputes what derives from it instead of hoisting + spilling
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

// fpc_tanh_lut_scaled / fpc_tanh_lut / fpc_sigmoid_lut (include/fpc_numerics.h) on the pair table: one
// ds_read_b64, v_fract for the interpolation weight (u - trunc(u) exactly, u >= 0); bit-identical results
__device__ __forceinline__ float lut_scaled(const float2* T2, float x, float scale) {
    const float u = fminf(fabsf(x) * scale, 4095.99976f);
    const float f = __builtin_amdgcn_fractf(u);
    const float2 td = T2[(uint32_t)u];
    return copysignf(fmaf(f, td.y, td.x), x);
}
__device__ __forceinline__ float lut_tanh(const float2* T2, float x) { return lut_scaled(T2, x, 512.0f); }
__device__ __forceinline__ float lut_sigmoid(const float2* T2, float x) {
    return fmaf(0.5f, lut_scaled(T2, x, 256.0f), 0.5f);
}

// zero-padded balanced (adjacent-pair) tree over the QP partial sums of one gate row; p0 = &part[0][row]
template <int QP>
__device__ __forceinline__ float part_tree(const float* p0) {
    float v[QP];
#pragma unroll
    for (int k = 0; k < QP; ++k) v[k] = p0[k * PSTRIDE];
#pragma unroll
    for (int w = QP; w > 1; w >>= 1)
#pragma unroll
        for (int k = 0; k < w / 2; ++k) v[k] = v[2 * k] + v[2 * k + 1];
    return v[0];
}
// STAMP=true is a diagnostic build (env FPC_DECODE_STAMPS=1): for samples FPC_STAMP_T0 .. +FPC_STAMP_NS of block 0,
// lane 0 of every wave stores raw s_memtime values: slot k = arrival at barrier k (0 Y, 1 Z1, 2 Z2, 3 Z3, 4 X; the
// release is taken as the last wave's arrival).  No registers are held between stamps; perturbs the timing a little
// (each stamp waits for the wave's LDS operations), never timed.
#define FPC_STAMP_T0 400
#define FPC_STAMP_NS 64
#define FPC_STAMP(k)                                                                                       \
    if (STAMP && stamp_on) {                                                                               \
        const unsigned now_ = (unsigned)__builtin_readcyclecounter();                                      \
        if (lane == 0) P.stamps[((st_t - FPC_STAMP_T0) * (NTHREADS / 64) + wave) * 8 + (k)] = now_;        \
    }
#define FPC_SYNC() __syncthreads()
#define FPC_BARRIER(k) \
    FPC_STAMP(k)       \
    FPC_SYNC();

// QZR / QN: partial-sum planes read per update/reset-gate row and per candidate-gate row (powers of two
// >= the widest row group of those gates; planes no lane owns hold +0)
template <bool STAMP, int QZR, int QN>
__global__ __launch_bounds__(NTHREADS) void k_decode(const DecodeParams P) {
    __shared__ DecodeLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, T = P.T;

    // ---- LDS init ----
    // a chunk after the first resumes from the record its predecessor left (same values, so the same samples)
    const bool resume = P.state != nullptr && P.f0 > 0;
    float* const rec = P.state != nullptr ? P.state + (size_t)b * STATE_FLOATS : nullptr;
    for (int i = tid; i < RNN_A; i += NTHREADS) {
        L.s1[i] = resume ? rec[i] : 0.0f;
        L.brn_a[i] = P.brn_a[i];
    }
    for (int i = tid; i < GA; i += NTHREADS) {
        L.diag[i] = P.diag[i];
    }
    for (int i = tid; i < 16 * PSTRIDE; i += NTHREADS) L.part[i] = 0.0f;  // slots no lane owns stay +0 (exact padding)
    if (tid < 64) L.ulaw_thr[tid] = k_ulaw_thr[tid];
    for (int k = tid; k < FPC_TANH_TABLE_SIZE - 1; k += NTHREADS) {
        const float t0 = fpc_tanh_table_entry(k), t1 = fpc_tanh_table_entry(k + 1);
        L.tt[k] = make_float2(t0, t1 - t0);
    }
    if (tid < RNN_B) {
        L.s2[tid] = resume ? rec[RNN_A + tid] : 0.0f;
        L.hist[tid] = resume ? rec[RNN_A + 16 + tid] : 0.0f;
    }
    if (tid == 0) {
        if (resume) {
            *reinterpret_cast<float4*>(&L.o_sig) = *reinterpret_cast<const float4*>(&rec[RNN_A + 32]);
        } else {
            L.o_sig = 128u * GA;
            L.o_pred = (256u + 128u) * GA;
            L.o_exc = (512u + 128u) * GA;
            L.pred = -0.0f;
        }
    }
    int16_t* out = P.pcm + (size_t)b * T * FPC_FRAME_SIZE;
    if (tid < FPC_LPC_ORDER + 1 && P.f0 == 0) out[tid] = 0;  // test_lpcnet.py skips order+1 samples
    __syncthreads();

    if (wave >= 4) {
        // =========================== mat-vec role ===========================
        const int ml_ = tid - NSAMP;
        // w2[(block*4 + col)*4 + rp] = weights of rows (2rp, 2rp+1) at column `col` of block `block`:
        // one v_pk_fma_f32 advances two row chains by one column with the h value broadcast
        f2 w2[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int bc = j >> 2, rp = j & 3;  // bc = block*4 + col
            const int bb = bc >> 2, c = bc & 3;
            w2[j] = mk2(P.lane_w[(bb * 32 + (2 * rp) * 4 + c) * NMAT + ml_],
                         P.lane_w[(bb * 32 + (2 * rp + 1) * 4 + c) * NMAT + ml_]);
        }
        const unsigned colp_ = (unsigned)P.lane_meta[ml_];
        const unsigned metap_ = (unsigned)P.lane_meta[NMAT + ml_];
        const bool gate_lane = ml_ < RNN_A;
        // where this lane's 8 partial row sums go: part[lane q of the group][first row of the group]
        // (kept as a finished LDS byte address: one VGPR, no per-sample address math; 0 = lane owns no group)
        typedef __attribute__((address_space(3))) float lds_float;
        unsigned paddr_ = 0u;
        if ((metap_ >> 16) != 0) {
            const int grp = (int)(metap_ >> 16) - 1;
            const int gate = grp / (RNN_A / 8), rb = grp - gate * (RNN_A / 8);
            paddr_ = (unsigned)(size_t)(lds_float*)&L.part[(int)(metap_ & 0xff) * PSTRIDE + gate * RNN_A + rb * 8];
        }
        paddr_ = opaque(paddr_);

#define FPC_COLS(FROM, TO)                                                                                        \
    _Pragma("unroll") for (int bc = (FROM); bc < (TO); ++bc) {                                                    \
        if (bc == 0 || (bc == (FROM) && bc < 8 && false)) {                                                       \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 ha = *reinterpret_cast<const float4*>(&L.s1[(colp & 0xff) * 4]);                         \
            const float4 hb = *reinterpret_cast<const float4*>(&L.s1[((colp >> 8) & 0xff) * 4]);                  \
            hv0[0] = ha.x, hv0[1] = ha.y, hv0[2] = ha.z, hv0[3] = ha.w;                                           \
            hv0[4] = hb.x, hv0[5] = hb.y, hv0[6] = hb.z, hv0[7] = hb.w;                                           \
        }                                                                                                         \
        if (bc == 8) {                                                                                            \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 hc = *reinterpret_cast<const float4*>(&L.s1[((colp >> 16) & 0xff) * 4]);                 \
            const float4 hd = *reinterpret_cast<const float4*>(&L.s1[(colp >> 24) * 4]);                          \
            hv1[0] = hc.x, hv1[1] = hc.y, hv1[2] = hc.z, hv1[3] = hc.w;                                           \
            hv1[4] = hd.x, hv1[5] = hd.y, hv1[6] = hd.z, hv1[7] = hd.w;                                           \
        }                                                                                                         \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                                 \
            if (bc < 8)                                                                                           \
                acc[rp] = fma2(w2[bc * 4 + rp], splat2(hv0[bc]), acc[rp]);                                        \
            else                                                                                                  \
                a[rp] = fma2(w2[bc * 4 + rp], splat2(hv1[bc - 8]), a[rp]);                                        \
        }                                                                                                         \
    }                                                                                                             \
    _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                                            \
        pin(acc[rp]);                                                                                             \
        pin(a[rp]);                                                                                               \
    }
#define FPC_PUBLISH()                                                                                             \
    _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) acc[rp] = acc[rp] + a[rp]; /* the in-lane tree level */         \
    if (paddr_ != 0u) { /* publish this lane's 8 partial row sums */                                                  \
        typedef float v4f __attribute__((ext_vector_type(4)));                                                        \
        typedef __attribute__((address_space(3))) v4f lds_v4f;                                                        \
        lds_v4f* pp = (lds_v4f*)(size_t)paddr_; /* 32-byte aligned: two ds_write_b128 */                              \
        v4f lo, hi;                                                                                                   \
        lo.x = acc[0].x, lo.y = acc[0].y, lo.z = acc[1].x, lo.w = acc[1].y;                                           \
        hi.x = acc[2].x, hi.y = acc[2].y, hi.z = acc[3].x, hi.w = acc[3].y;                                           \
        pp[0] = lo;                                                                                                   \
        pp[1] = hi;                                                                                                   \
    }
        if (P.state != nullptr) {  // (uniform) a resumed chunk: the sparse product of the state it starts from, as the
            if (resume) {          // last sample of the chunk before left it in LDS
                f2 acc[4], a[4];
                float hv0[8], hv1[8];
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f);
                FPC_COLS(0, 16)
                FPC_PUBLISH()
            }
            __syncthreads();
        }
        for (int fr = P.f0; fr < P.f1; ++fr) {
            // voiced frames (pdf sharpening on) keep a separate parallel leaf phase: one more barrier
            const bool voiced = fpc_shape_exponent(P.features[((size_t)b * T + fr) * FPC_NB_FEATURES + 19]) > 0.0f;
            if (gate_lane) {  // this frame's conditioning values of unit ml: written and read by the same lane
                const float* cfa = P.cfa + ((size_t)b * P.cf_T + (fr - P.f0)) * GA;
                L.cfa[ml_] = cfa[ml_];
                L.cfa[RNN_A + ml_] = cfa[RNN_A + ml_];
                L.cfa[2 * RNN_A + ml_] = cfa[2 * RNN_A + ml_];
            }
            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                const int st_t = fr * FPC_FRAME_SIZE + i;
                const bool stamp_on = STAMP && blockIdx.x == 0 && st_t >= FPC_STAMP_T0 && st_t < FPC_STAMP_T0 + FPC_STAMP_NS;
                // ---- X..Y: gather the three embedding-table rows, GRU_A gates ----
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(FPC_PRIO);  // the gates are on the sample-to-sample critical path
#endif
                if (gate_lane) {
                    const unsigned ml = opaque((unsigned)ml_);
                    struct F3 {
                        float x, y, z;
                    };
                    unsigned oa = L.o_sig, ob = L.o_pred, oc = L.o_exc;
                    // uniform base + 32-bit byte offset (the global_load saddr form: no 64-bit VALU address math)
                    const char* tabc = reinterpret_cast<const char*>(P.tab);
                    const F3 ta = *reinterpret_cast<const F3*>(tabc + (size_t)((oa + 3u * ml) * 4u));
                    const F3 tb = *reinterpret_cast<const F3*>(tabc + (size_t)((ob + 3u * ml) * 4u));
                    const F3 tc = *reinterpret_cast<const F3*>(tabc + (size_t)((oc + 3u * ml) * 4u));
                    // while the gather is in flight: recurrent terms of the three rows of unit ml =
                    // diagonal + tree over the row group's partial sums (written before barrier X)
                    const float h_own = L.s1[ml];
                    const float unb =
                        fmaf(L.diag[2 * RNN_A + ml], h_own, part_tree<QN>(&L.part[2 * RNN_A + ml])) + L.brn_a[ml];
                    __builtin_amdgcn_sched_barrier(0);  // bounds the live registers next to the 128 weight VGPRs
                    const float uz = fmaf(L.diag[ml], h_own, part_tree<QZR>(&L.part[ml]));
                    const float ur = fmaf(L.diag[RNN_A + ml], h_own, part_tree<QZR>(&L.part[RNN_A + ml]));
                    const float cz = L.cfa[ml], cr = L.cfa[RNN_A + ml], cn = L.cfa[2 * RNN_A + ml];
                    const float gz = ((ta.x + tb.x) + tc.x) + cz;
                    const float gr = ((ta.y + tb.y) + tc.y) + cr;
                    const float gn = ((ta.z + tb.z) + tc.z) + cn;
                    const float z = lut_sigmoid(L.tt, gz + uz);
                    const float r = lut_sigmoid(L.tt, gr + ur);
                    const float n = lut_tanh(L.tt, fmaf(r, unb, gn));
                    const float h_new = fmaf(z, h_own - n, n);
                    L.s1[ml] = h_new;
                }
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                FPC_BARRIER(0)  // Y
                // ---- the sparse product of s1' with this lane's 4 blocks (16 columns of 8 rows), sliced under
                //      the sampler phases: FPC_NA columns under GRU_B, FPC_NB under the dual FC, the rest and the
                //      store of the partial sums under the draw ----
                f2 acc[4], a[4];
                float hv0[8], hv1[8];
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f);
                FPC_COLS(0, FPC_NA)
                FPC_BARRIER(1)  // Z1
                FPC_COLS(FPC_NA, FPC_NA + FPC_NB)
                FPC_BARRIER(2)  // Z2
                if (voiced) {
                    FPC_BARRIER(3)  // Z3 (voiced frames only: the sampler waves' leaf phase ends here)
                }
                FPC_COLS(FPC_NA + FPC_NB, 16)
                FPC_PUBLISH()
                FPC_BARRIER(4)  // X
            }
        }
#undef FPC_COLS
#undef FPC_PUBLISH
        if (rec != nullptr && gate_lane) rec[ml_] = L.s1[ml_];  // (written by this lane)
    } else {
        // =========================== sampler role ===========================
        __builtin_amdgcn_s_setprio(3);  // the sample-to-sample critical path lives in these waves
        const int sl = tid;                    // 0..255
        const int u = sl >> 4, kl = sl & 15;   // GRU_B: unit, 24-input slice
        const unsigned long long seed = P.seeds[b];
        // GRU_B input weights of this lane's 24 inputs (lane_wb[gate*24 + k] = weight of input 24*kl + k):
        // wB[g][m][h] pairs the weights of inputs 4m+2h and 4m+2h+1, i.e. of the leaf pair (2h, 2h+1) at
        // step m: one v_pk_fma_f32 per half of a float4 of state advances two leaves of one gate
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
        // dual-FC weights of tree node sl: 16 (channel 0, channel 1) pairs, then bias pair and factor pair
        f2 fcw[18];
#pragma unroll
        for (int k = 0; k < 16; ++k) fcw[k] = mk2(P.lane_fc[k * NSAMP + sl], P.lane_fc[(16 + k) * NSAMP + sl]);
        fcw[16] = mk2(P.lane_fc[32 * NSAMP + sl], P.lane_fc[33 * NSAMP + sl]);
        fcw[17] = mk2(P.lane_fc[34 * NSAMP + sl], P.lane_fc[35 * NSAMP + sl]);
        float mem = resume ? rec[RNN_A + 36] : 0.0f;     // de-emphasis state (tracked by every lane of the drawing wave)
        float pcm_new = 0.0f; // the sample just drawn (drawing wave)
        float s2_own = resume ? rec[RNN_A + u] : 0.0f;  // state of unit u, replicated over the 16 lanes of its row
        if (P.state != nullptr) __syncthreads();  // (the mat-vec waves' product of the resumed state)

        for (int fr = P.f0; fr < P.f1; ++fr) {
            const float* feat = P.features + ((size_t)b * T + fr) * FPC_NB_FEATURES;
            const float shape_e = fpc_shape_exponent(feat[19]);
            const float* cfb = P.cfb + ((size_t)b * P.cf_T + (fr - P.f0)) * GB;
            const float cfb_z = cfb[u], cfb_r = cfb[RNN_B + u], cfb_n = cfb[2 * RNN_B + u];
            const float* fa = feat + (FPC_NB_FEATURES - FPC_LPC_ORDER);  // this frame's LPC, and the next frame's
            const float* fan = fa + (fr + 1 < T ? FPC_NB_FEATURES : 0);
            const float a_cur = fa[kl], a_nxt = fan[kl], a0_cur = fa[0], a0_nxt = fan[0];
            if (sl < FPC_FRAME_SIZE)  // first read behind barrier Z3 of this frame's first sample
                L.uframe[sl] = fpc_philox_uniform(seed, (uint32_t)(fr * FPC_FRAME_SIZE + sl));

            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                const int t = fr * FPC_FRAME_SIZE + i;
                const int st_t = t;
                const bool stamp_on = STAMP && blockIdx.x == 0 && st_t >= FPC_STAMP_T0 && st_t < FPC_STAMP_T0 + FPC_STAMP_NS;
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(0);  // window work has slack; the gate waves do not
#endif
                // ---- X..Y (the mat-vec waves gather + gate): everything that only needs the
                //      previous draw: GRU_B recurrent part, LPC history chain, leaf candidates ----
                const float s2k = L.s2[kl];
                float ub_z = ub0 * s2k, ub_r = ub1 * s2k, ub_n = ub2 * s2k;
                row_bfly16x3(ub_z, ub_r, ub_n);
                {
                    // prediction of the NEXT sample: taps 2..16 as a balanced tree over the 16 lanes of
                    // the row (lane kl holds tap kl+1, lane 0 contributes 0), the newest tap by one fma
                    const bool lastsmp = i == FPC_FRAME_SIZE - 1;  // next sample belongs to the next frame
                    const float am = lastsmp ? a_nxt : a_cur;
                    const float a0 = lastsmp ? a0_nxt : a0_cur;
                    const float hk = L.hist[(t - kl) & 15];
                    const float part = row_bfly16(kl ? am * hk : 0.0f);
                    // what the control block becomes if leaf `sl` wins this sample's draw
                    const float cpcm = L.pred + my_ulaw;
                    const float cpred = -fmaf(a0, cpcm, part);
                    // ... with the float offsets of the table rows of its signal and prediction levels (x GA = 1024 + 128)
                    const unsigned es = (unsigned)fpc_lin2ulaw_tab(cpcm, L.ulaw_thr);
                    const unsigned ep = 256u + (unsigned)fpc_lin2ulaw_tab(cpred, L.ulaw_thr);
                    L.cand[sl] = make_float4(cpcm, cpred, __uint_as_float((es << 10) + (es << 7)),
                                             __uint_as_float((ep << 10) + (ep << 7)));
                }
#if FPC_PRIO
                __builtin_amdgcn_s_setprio(3);
#endif
                FPC_BARRIER(0)  // Y
                // ---- Y..Z1: GRU_B (row of 16 lanes = unit; lane = 24 inputs = 4 leaves of 6, leaf of
                //      float4 component c takes inputs c + 4m) ----
                {
                    const unsigned klv = (unsigned)kl;  // (24 * kl: a lane constant the compiler keeps in a register)
                    f2 acc[3][2];
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g][0] = acc[g][1] = splat2(0.0f);
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const float4 h4 = *reinterpret_cast<const float4*>(&L.s1[24 * klv + 4 * m]);
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            acc[g][0] = fma2(wB[g][m][0], mk2(h4.x, h4.y), acc[g][0]);
                            acc[g][1] = fma2(wB[g][m][1], mk2(h4.z, h4.w), acc[g][1]);
                        }
                    }
                    float a3[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {  // (leaf x + leaf z) + (leaf y + leaf w)
                        const f2 pr = acc[g][0] + acc[g][1];
                        a3[g] = pr.x + pr.y;
                    }
                    row_bfly16x3(a3[0], a3[1], a3[2]);
                    const float z = lut_sigmoid(L.tt, (a3[0] + cfb_z) + ub_z);
                    const float r = lut_sigmoid(L.tt, (a3[1] + cfb_r) + ub_r);
                    const float n = lut_tanh(L.tt, fmaf(r, ub_n + brnb, a3[2] + cfb_n));
                    s2_own = fmaf(z, s2_own - n, n);
                    if (kl == 0) L.s2[u] = s2_own;
                }
                FPC_BARRIER(1)  // Z1
                // ---- Z1..Z2: dual FC of tree node `sl` ----
                {
                    const unsigned slv = opaque((unsigned)sl);
                    f2 a01 = fcw[16], b01 = splat2(0.0f);  // both channels advance together; even / odd inputs
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const float4 sv = *reinterpret_cast<const float4*>(&L.s2[4 * k4]);
                        a01 = fma2(fcw[4 * k4], splat2(sv.x), a01);
                        b01 = fma2(fcw[4 * k4 + 1], splat2(sv.y), b01);
                        a01 = fma2(fcw[4 * k4 + 2], splat2(sv.z), a01);
                        b01 = fma2(fcw[4 * k4 + 3], splat2(sv.w), b01);
                    }
                    a01 = a01 + b01;
                    const float t0 = lut_tanh(L.tt, a01.x), t1 = lut_tanh(L.tt, a01.y);
                    const float v = fmaf(fcw[17].y, t1, fcw[17].x * t0);
                    const float qv = lut_sigmoid(L.tt, v);
                    // both branch factors of the node: a leaf reads the one its bit selects, no select on the draw's chain
                    reinterpret_cast<float2*>(L.qq)[slv] = make_float2(1.0f - qv, qv);
                }
                const float uf = L.uframe[i];  // this sample's uniform: fetched under the FC phase, not behind the scan
                FPC_BARRIER(2)  // Z2
                float4 p4;  // wave 0: probabilities of leaves 4*lane .. 4*lane+3
                // branch factor j (root = 0) of a leaf is float 2*(2^j + (leaf >> (8-j))) + bit_(7-j)(leaf)
                //                                            = 2*2^j + (leaf >> (7-j)) of qq
                const float* qf = reinterpret_cast<const float*>(L.qq);
                if (shape_e > 0.0f) {
                    // ---- voiced frame, Z2..Z3: leaf probability + sharpening, 256 lanes ----
                    {
                        const unsigned slv = opaque((unsigned)sl);  // (recomputed per sample: no registers to spare)
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = qf[(2u << j) + (slv >> (7 - j))];
                        const float p = ((((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5])) * f[6]) * f[7];
                        L.p[slv] = fpc_shape_pow(p, shape_e);
                    }
                    FPC_BARRIER(3)  // Z3
                    if (wave == 0) p4 = *reinterpret_cast<const float4*>(&L.p[4 * lane]);
                } else if (wave == 0) {
                    // ---- unvoiced frame: the drawing wave builds its 4 leaves per lane straight from the
                    //      factor pairs (same product order per leaf); no separate leaf phase, no barrier ----
                    const unsigned lv = (unsigned)lane;  // leaves 4*lv .. 4*lv+3: factors 0..5 shared
                    float f[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) f[j] = qf[(2u << j) + (lv >> (5 - j))];
                    const float2 q6 = *reinterpret_cast<const float2*>(&qf[128u + 2u * lv]);   // node 64 + lv
                    const float4 q7 = *reinterpret_cast<const float4*>(&qf[256u + 4u * lv]);   // nodes 128 + 2 lv, + 1
                    const float pre = ((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5]);
                    const float lo = pre * q6.x, hi = pre * q6.y;
                    p4.x = lo * q7.x;
                    p4.y = lo * q7.y;
                    p4.z = hi * q7.z;
                    p4.w = hi * q7.w;
                }
                // ---- (wave 0): normaliser, tail cut, scan, draw, publish ----
                if (wave == 0) {
                    float thr = 0.002f;  // the tree pdf sums to 1 by construction: only sharpened pdfs are totalled
                    if (shape_e > 0.0f) {
                        float rs = row_bfly16((p4.x + p4.y) + (p4.z + p4.w));
                        rs = add_bcast<DPP_BCAST15, 0xa>(rs);  // rows 1,3 += rows 0,2
                        rs = add_bcast<DPP_BCAST31, 0xc>(rs);  // row 3 = (r2+r3)+(r0+r1): the balanced total
                        thr = 0.002f * lane_val(rs, 63);
                    }
                    // max(p - thr, 0) as one v_sub_f32 with the output clamp (p - thr <= 1 always, so the upper
                    // clamp never acts; NaN -> 0 on both forms)
                    const float c0 = __builtin_amdgcn_fmed3f(p4.x - thr, 0.0f, 1.0f);
                    const float c1 = __builtin_amdgcn_fmed3f(p4.y - thr, 0.0f, 1.0f);
                    const float c2 = __builtin_amdgcn_fmed3f(p4.z - thr, 0.0f, 1.0f);
                    const float c3 = __builtin_amdgcn_fmed3f(p4.w - thr, 0.0f, 1.0f);
                    // prefixes inside the lane's 4 leaves, two levels deep: c0 | c0+c1 | (c0+c1)+c2 | (c0+c1)+(c2+c3)
                    const float P1 = c0 + c1, s23 = c2 + c3;
                    const float P2 = P1 + c2, P3 = P1 + s23;
                    float I = P3;  // Kogge-Stone inside each row of 16 lanes
                    I = I + dpp_f<DPP_ROW_SHR + 1>(I);
                    I = I + dpp_f<DPP_ROW_SHR + 2>(I);
                    I = I + dpp_f<DPP_ROW_SHR + 4>(I);
                    I = I + dpp_f<DPP_ROW_SHR + 8>(I);
                    I = add_bcast<DPP_BCAST15, 0xa>(I);  // block offsets by row broadcasts
                    I = add_bcast<DPP_BCAST31, 0xc>(I);
                    const float rthr = uf * lane_val(I, 63);
                    // the draw = number of leaves whose inclusive prefix is <= the threshold: four compares to lane
                    // masks, four s_bcnt1 (the last leaf of a lane carries the scan value I itself)
                    const float O = dpp_f<DPP_WAVE_SHR1>(I);  // exclusive offset of the lane (lane 0: +0)
                    const unsigned long long m0 = __builtin_amdgcn_fcmpf(O + c0, rthr, 5 /* FCMP_OLE */);
                    const unsigned long long m1 = __builtin_amdgcn_fcmpf(O + P1, rthr, 5);
                    const unsigned long long m2 = __builtin_amdgcn_fcmpf(O + P2, rthr, 5);
                    const unsigned long long m3 = __builtin_amdgcn_fcmpf(I, rthr, 5);
                    int exc = (__popcll(m0) + __popcll(m1)) + (__popcll(m2) + __popcll(m3));
                    exc = exc > 255 ? 255 : exc;
                    float4 cd = L.cand[exc];  // one broadcast read: what the control block becomes
                    asm volatile("" : "+v"(cd.x), "+v"(cd.y), "+v"(cd.z), "+v"(cd.w));  // (keeps it one ds_read_b128 up here)
                    if (lane == 0) {
                        // control block {o_sig, o_pred, o_exc, pred}: one 16-byte store, first thing after the read
                        *reinterpret_cast<float4*>(&L.o_sig) =
                            make_float4(cd.z, cd.w, __uint_as_float((512u + (unsigned)exc) * (unsigned)GA), cd.y);
                        L.hist[t & 15] = cd.x;
                    }
                    pcm_new = cd.x;
#if FPC_PCM_WHERE == 1
                    mem = fmaf(FPC_PREEMPH, mem, pcm_new);
                    if (lane == 0) out[t] = fpc_pcm16(mem);
#endif
                }
                FPC_BARRIER(4)  // X
                // behind the barrier, off the sample-to-sample chain: de-emphasis and the PCM store
#if FPC_PCM_WHERE == 0
                if (wave == 0) {
                    mem = fmaf(FPC_PREEMPH, mem, pcm_new);
                    if (lane == 0) out[t] = fpc_pcm16(mem);
                }
#elif FPC_PCM_WHERE == 2
                if (wave == 3) {  // a sampler wave that does not draw picks the sample up from the history ring
                    mem = fmaf(FPC_PREEMPH, mem, L.hist[t & 15]);
                    if (lane == 0) out[t] = fpc_pcm16(mem);
                }
#endif
            }
        }
        if (rec != nullptr) {  // (behind the last sample's barrier X nothing writes these any more)
            if (sl < RNN_B) {
                rec[RNN_A + sl] = L.s2[sl];
                rec[RNN_A + 16 + sl] = L.hist[sl];
            }
            if (sl == 0) {
                *reinterpret_cast<float4*>(&rec[RNN_A + 32]) = *reinterpret_cast<const float4*>(&L.o_sig);
                rec[RNN_A + 36] = mem;
            }
        }
    }
}
// (FPC_BARRIER / FPC_STAMP stay defined for lpcnet_decode2.h, which undefines them)
