// lpcnet_decode.h -- the persistent per-utterance sample loop of the LPCNet-style vocoder
// (included by lpcnet.hip only; gfx950).
//
// One 512-thread workgroup (8 wave64, 2 per SIMD, up to 256 VGPRs each) per utterance, persistent over all samples.
// What shapes it (profiles/r03_ubench_issue.txt): a wave issues one vector instruction per ~4.6 cycles whatever its
// dependences, a packed fma or a DPP op occupies its SIMD's pipe for ~4.25 cycles, and a register-starved role gets a
// serialised schedule.  So every phase of the sample-to-sample chain is spread over ALL eight waves (fewest
// instructions per wave), every wave keeps the same ~190 resident registers, and the packed-fma bulk (the sparse
// product) runs on seven waves while the eighth draws:
//   every lane   4 blocks (8x4) of the block-sparse recurrent matrix of GRU_A              (128 VGPRs)
//                GRU_B: 32 lanes per unit, 12 inputs x 3 gates per lane                     (36 VGPRs)
//                dual FC: lane = (tree node, channel), 16 inputs                            (18 VGPRs)
//   lanes 0..47 of every wave: GRU_A gates of 48 units ("gate lanes")
//   wave 0: draws;  wave 1: de-emphasis + PCM store
// Four workgroup barriers per output sample (X, Y, Z1, Z2; voiced frames add Z3):
//   X  control block (byte offsets of the three table rows selected by the drawn sample) and the partial sums of the
//      sparse product published
//        gate lanes gather their 3 x 12 bytes of the three embedding x kernel table rows; in the shadow of that L2
//        round trip: recurrent terms (tree over the row group's partial-sum planes + diagonal), the recurrent part of
//        GRU_B, wave 0 the LPC taps 2..16 of the next prediction; then gates -> s1'
//   Y  s1' ready:   GRU_B (18 packed fmas per lane, 5-level DPP tree) -> s2'
//   Z1 s2' ready:   dual FC -> 255 branch-factor pairs
//   Z2 factors ready
//        wave 0: leaf probabilities, tail cut, scan, draw; the winner's record (signal, next prediction, their mu-law
//                levels by ballot + popcount) and the control block
//        waves 1-7: sparse product of s1' (16 columns x 4 row pairs per lane), 8 partial row sums per lane -> LDS
//        (voiced frame: 256 lanes compute leaf probability + sharpening first, barrier Z3)
// HBM is touched only for the gathered table rows (L2-resident), the per-frame conditioning rows and 2 bytes of PCM
// per sample.
//
// Canonical evaluation orders (DESIGN.md "Vocoder numerics") are those of
// oracle/fpc_oracle.c::orc_lpcnet_synthesize; results are bit-identical.
#pragma once

constexpr int NTHREADS = 512;
constexpr int NMAT = 512;       // lanes that can carry 4 blocks of the sparse product
constexpr int UPW = RNN_A / 8;  // gate units per wave (lanes 0..47)
constexpr int DRAW_WAVE = 0, PCM_WAVE = 1;
// stride between the partial-sum planes of consecutive lanes q of a row group: +4 floats so that the
// lanes of one group (consecutive lanes of a wave) start their 16-byte stores in different bank groups
constexpr int PSTRIDE = GA + 4;

struct DecodeParams {
    const float* tab;       // [3][256][384][3]  embedding x input-kernel tables, gate-interleaved
    const float* cfa;       // [B][T][1152]  GRU_A conditioning product (+biases)
    const float* cfb;       // [B][T][48]    GRU_B conditioning product (+biases)
    const float* features;  // [B][T][36]
    const unsigned long long* seeds;
    int16_t* pcm;  // [B][T*160]
    int T;
    const float* lane_w;     // [128][512] sparse GRU_A weights: 2 leaves x 2 blocks x 8x4
    const int* lane_meta;    // [2][512]   packed column blocks; (group+1)<<16 | lanes<<8 | lane
    const float* lane_wb;    // [36][512]  GRU_B input weights of lane (unit, slice kl, half h): [gate][m][c0|c1]
    const float* lane_ub;    // [3][512]   GRU_B recurrent weights ub[k][gate*16+unit], k = lane & 15
    const float* lane_fc;    // [18][512]  dual-FC of lane (node, channel): 16 weights, bias, factor
    const float* diag;       // [1152]
    const float* brn_a;      // [384]
    const float* brn_b;      // [16]
    const float* ulaw_tab;   // [256]
    unsigned* stamps;  // diagnostic only: [FPC_STAMP_NS][8 waves][16 slots]
};

__device__ const float k_ulaw_thr[64] = FPC_ULAW_TABLE_INIT;

// position of state unit i in L.s1: inside each aligned group of four the order is (0, 2, 1, 3), so that the two
// inputs a GRU_B lane multiplies with one packed fma -- components (0,2) or (1,3) of a float4 of state, the leaf pair
// of the canonical order -- are one aligned 8-byte read
__device__ __forceinline__ unsigned s1_pos(unsigned i) { return (i & ~3u) | ((i & 1u) << 1) | ((i >> 1) & 1u); }

// Every field sits in the first 64 KB for the lane-indexed arrays (bases fold into the DS instructions' offset
// fields), the activation table at offset 0.
struct __attribute__((aligned(16))) DecodeLds {
    float2 tt[FPC_TANH_TABLE_SIZE - 1];  // fpc_tanh_lut table as (T[k], T[k+1] - T[k]) pairs, built at kernel start
    float s1[RNN_A];          // GRU_A state, units at s1_pos()
    float cfa[GA];            // this frame's GRU_A conditioning rows [z|r|h][unit]
    float diag[GA];
    float brn_a[RNN_A];
    float s2[RNN_B];
    float hist[16];
    // control block written by the drawing wave: byte offsets of the three table rows to gather next
    unsigned o_sig, o_pred, o_exc;
    float pred;      // prediction of the next sample
    float4 qq[128];  // node n's branch factors as the pair (1 - q[n], q[n]) at floats 2n, 2n+1 (16-byte aligned rows)
    float p[256];
    float ulaw[256];     // fpc_ulaw2lin table
    float uframe[FPC_FRAME_SIZE];
    float cfb[GB];       // this frame's GRU_B conditioning rows
    float brn_b[RNN_B];
    float part[16 * PSTRIDE];  // partial row sums of the sparse product: [lane q of the row group][gate row]
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
// three independent butterflies interleaved: every DPP read of a register comes >= 2 instructions
// after its last write (the gfx9 VALU-write -> DPP-read hazard), so no s_nop and no separate v_mov_dpp
#define FPC_B3(CTRL)                                                              \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// ... over each aligned row of 16 lanes; result in every lane of the row
__device__ __forceinline__ void row_bfly16x3(float& a, float& b, float& c) {
    asm volatile("s_nop 1\n\t" FPC_B3("quad_perm:[1,0,3,2]") FPC_B3("quad_perm:[2,3,0,1]") FPC_B3("row_half_mirror")
                     FPC_B3("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c));
}
// ... over each aligned pair of rows (32 lanes); the total lands in the ODD row of the pair (rows 1 and 3 add the
// total of rows 0 and 2, broadcast from their last lane)
__device__ __forceinline__ void pair_bfly32x3(float& a, float& b, float& c) {
    asm volatile("s_nop 1\n\t" FPC_B3("quad_perm:[1,0,3,2]") FPC_B3("quad_perm:[2,3,0,1]") FPC_B3("row_half_mirror")
                     FPC_B3("row_mirror")
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 : "+v"(a), "+v"(b), "+v"(c));
}
#undef FPC_B3
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
// a value loaded from global memory before the sample loop, re-defined by an (empty) instruction once it has
// landed: its uses inside the loop then need no vector-memory wait (s_waitcnt vmcnt counts in issue order, so a
// wait for such a value inside the loop would also wait for the table gather in flight)
__device__ __forceinline__ float landed(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
// a register the compiler must treat as defined (no instruction): keeps a value that only gate lanes
// produce from being zero-filled for the other lanes at the top of every sample
__device__ __forceinline__ float undef_f() {
    float v;
    asm volatile("" : "=v"(v));
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

// fpc_lin2ulaw_tab (include/fpc_numerics.h) of a wave-uniform x by a ballot: its K is by definition the number of the
// 128 thresholds 2^e c_i (e = 0..7, c_i = 2^((i+.5)/16) as the table's floats) that v = fl(1 + 255|x|/32768) reaches;
// lane l holds thresholds c_(l&15) 2^(l>>4) and c_(l&15) 2^(4+(l>>4)): two compares, two s_bcnt1 (tests/test_host_cpu.py
// checks the count form against the table form on every threshold's neighbourhood)
__device__ __forceinline__ unsigned ulaw_level_ballot(float x, float thr_lo, float thr_hi) {
    const float v = fmaf(255.0f / 32768.0f, fabsf(x), 1.0f);
    const unsigned long long m0 = __builtin_amdgcn_fcmpf(v, thr_lo, 3 /* FCMP_OGE */);
    const unsigned long long m1 = __builtin_amdgcn_fcmpf(v, thr_hi, 3);
    const int K = __popcll(m0) + __popcll(m1);
    const int u = x < 0.0f ? 128 - K : 128 + K;
    return (unsigned)(u > 255 ? 255 : u);
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
// lane 0 of every wave stores raw s_memtime values: slot 2k = arrival at barrier k (k: 0 Y, 1 Z1, 2 Z2, 3 Z3, 4 X;
// the release is taken as the last wave's arrival); with -DFPC_DRAW_STAMPS the drawing wave also stamps inside its
// Z2..X stretch (slots 10..15).
// No registers are held between stamps; perturbs the timing a little (each stamp waits for the wave's LDS
// operations), never timed.
#define FPC_STAMP_T0 400
#define FPC_STAMP_NS 64
#define FPC_STAMP(k)                                                                                    \
    if (STAMP && stamp_on) {                                                                            \
        const unsigned now_ = (unsigned)__builtin_readcyclecounter();                                   \
        if (lane == 0) P.stamps[((t - FPC_STAMP_T0) * 8 + wave) * 16 + (k)] = now_;                     \
    }
#ifdef FPC_DRAW_STAMPS
#define FPC_DSTAMP(k) FPC_STAMP(10 + (k))
#else
#define FPC_DSTAMP(k)
#endif
#define FPC_BARRIER(k)   \
    FPC_STAMP(2 * (k))   \
    __syncthreads();

struct F3 {
    float x, y, z;
};

// ---- the sparse product of s1' with this lane's 4 blocks (16 columns of 8 rows): columns [FROM, TO); a float4 of
//      state holds the units (0, 2, 1, 3) of its group (s1_pos) ----
#define FPC_COLS(FROM, TO)                                                                                        \
    _Pragma("unroll") for (int bc = (FROM); bc < (TO); ++bc) {                                                    \
        if (bc == 0) {                                                                                            \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 ha = *reinterpret_cast<const float4*>(&L.s1[(colp & 0xff) * 4]);                         \
            const float4 hb = *reinterpret_cast<const float4*>(&L.s1[((colp >> 8) & 0xff) * 4]);                  \
            hv0[0] = ha.x, hv0[1] = ha.z, hv0[2] = ha.y, hv0[3] = ha.w;                                           \
            hv0[4] = hb.x, hv0[5] = hb.z, hv0[6] = hb.y, hv0[7] = hb.w;                                           \
        }                                                                                                         \
        if (bc == 8) {                                                                                            \
            const unsigned colp = opaque(colp_);                                                                  \
            const float4 hc = *reinterpret_cast<const float4*>(&L.s1[((colp >> 16) & 0xff) * 4]);                 \
            const float4 hd = *reinterpret_cast<const float4*>(&L.s1[(colp >> 24) * 4]);                          \
            hv1[0] = hc.x, hv1[1] = hc.z, hv1[2] = hc.y, hv1[3] = hc.w;                                           \
            hv1[4] = hd.x, hv1[5] = hd.z, hv1[6] = hd.y, hv1[7] = hd.w;                                           \
        }                                                                                                         \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                                        \
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
// ---- the in-lane tree level (leaf pair) and this lane's 8 partial row sums -> its plane ----
#define FPC_PUBLISH()                                                                           \
    {                                                                                           \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) acc[rp] = acc[rp] + a[rp];             \
        if (paddr_ != 0u) {                                                                     \
            typedef float v4f __attribute__((ext_vector_type(4)));                              \
            typedef __attribute__((address_space(3))) v4f lds_v4f;                              \
            lds_v4f* pp = (lds_v4f*)(size_t)paddr_; /* 32-byte aligned: two ds_write_b128 */    \
            v4f lo, hi;                                                                         \
            lo.x = acc[0].x, lo.y = acc[0].y, lo.z = acc[1].x, lo.w = acc[1].y;                 \
            hi.x = acc[2].x, hi.y = acc[2].y, hi.z = acc[3].x, hi.w = acc[3].y;                 \
            pp[0] = lo;                                                                         \
            pp[1] = hi;                                                                         \
        }                                                                                       \
    }
#define FPC_SPARSE()                                                          \
    {                                                                         \
        f2 acc[4], a[4];                                                      \
        float hv0[8], hv1[8];                                                 \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f); \
        FPC_COLS(0, 16)                                                       \
        FPC_PUBLISH()                                                         \
    }

// QZR / QN: partial-sum planes read per update/reset-gate row and per candidate-gate row (powers of two
// >= the widest row group of those gates; planes no lane owns hold +0).
// W0: the drawing wave's lanes carry blocks too (matrices that need more than 448 lanes): it multiplies them
// behind its draw, which lengthens the sample.
template <bool STAMP, int QZR, int QN, bool W0>
__global__ __launch_bounds__(NTHREADS) void k_decode(const DecodeParams P) {
    __shared__ DecodeLds L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (provably wave-uniform: role branches are scalar branches)
    const int b = blockIdx.x, T = P.T;

    // ---- LDS init ----
    for (int i = tid; i < RNN_A; i += NTHREADS) {
        L.s1[i] = 0.0f;
        L.brn_a[i] = P.brn_a[i];
    }
    for (int i = tid; i < GA; i += NTHREADS) L.diag[i] = P.diag[i];
    for (int i = tid; i < 16 * PSTRIDE; i += NTHREADS) L.part[i] = 0.0f;  // slots no lane owns stay +0 (exact padding)
    if (tid < 256) L.ulaw[tid] = P.ulaw_tab[tid];
    for (int k = tid; k < FPC_TANH_TABLE_SIZE - 1; k += NTHREADS) {
        const float t0 = fpc_tanh_table_entry(k), t1 = fpc_tanh_table_entry(k + 1);
        L.tt[k] = make_float2(t0, t1 - t0);
    }
    if (tid < RNN_B) {
        L.s2[tid] = 0.0f;
        L.hist[tid] = 0.0f;
        L.brn_b[tid] = P.brn_b[tid];
    }
    if (tid == 0) {
        L.o_sig = 128u * GA * 4u;
        L.o_pred = (256u + 128u) * GA * 4u;
        L.o_exc = (512u + 128u) * GA * 4u;
        L.pred = -0.0f;
    }
    int16_t* out = P.pcm + (size_t)b * T * FPC_FRAME_SIZE;
    if (tid < FPC_LPC_ORDER + 1) out[tid] = 0;  // test_lpcnet.py skips order+1 samples

    // ---- every lane: its 4 blocks of the sparse matrix ----
    // w2[(block*4 + col)*4 + rp] = weights of rows (2rp, 2rp+1) at column `col` of block `block`:
    // one v_pk_fma_f32 advances two row chains by one column with the h value broadcast
    f2 w2[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const int bc = j >> 2, rp = j & 3;  // bc = block*4 + col
        const int bb = bc >> 2, c = bc & 3;
        w2[j] = mk2(P.lane_w[(bb * 32 + (2 * rp) * 4 + c) * NMAT + tid], P.lane_w[(bb * 32 + (2 * rp + 1) * 4 + c) * NMAT + tid]);
    }
    const unsigned colp_ = (unsigned)P.lane_meta[tid];
    const unsigned metap_ = (unsigned)P.lane_meta[NMAT + tid];
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

    // ---- every wave: 48 gate lanes ----
    const bool gate_lane = lane < UPW;
    const unsigned unit = (unsigned)(wave * UPW + (gate_lane ? lane : 0));
    const unsigned upos = s1_pos(unit);  // where the unit's state sits in L.s1
    const unsigned voff = 12u * unit;    // bytes of (unit, z|r|h) inside a table row
    const char* tabc = reinterpret_cast<const char*>(P.tab);

    // ---- GRU_B: 32 lanes (two DPP rows) per unit; lane (kl, h) multiplies the leaf pair (components h and h+2 of the
    //      float4s of state of input slice kl): wb[g][m] = weights of inputs 24 kl + 4 m + h and + h + 2 ----
    const int gu = tid >> 5, gkl = (tid >> 1) & 15, gh = tid & 1;
    f2 wb[3][6];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int m = 0; m < 6; ++m)
            wb[g][m] = mk2(P.lane_wb[((g * 6 + m) * 2) * NMAT + tid], P.lane_wb[((g * 6 + m) * 2 + 1) * NMAT + tid]);
    const unsigned gsoff = (unsigned)(24 * gkl + 2 * gh);  // float index of the lane's first state pair
    const bool gru_out = (lane & 16) != 0;                 // the odd row of the unit's pair of rows holds the totals
    const int k16 = lane & 15;
    const float ub0 = P.lane_ub[tid], ub1 = P.lane_ub[NMAT + tid], ub2 = P.lane_ub[2 * NMAT + tid];
    float s2_own = 0.0f;                          // state of unit gu, replicated over the 16 lanes of its odd row
    float ub_z = 0.0f, ub_r = 0.0f, ub_n = 0.0f;  // recurrent part of GRU_B

    // ---- dual FC: lane = (node, channel) ----
    const int fnode = tid >> 1, fch = tid & 1;
    float fw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) fw[k] = P.lane_fc[k * NMAT + tid];
    const float fbias = P.lane_fc[16 * NMAT + tid], ffac = P.lane_fc[17 * NMAT + tid];

    // ---- the drawing wave: thresholds of the mu-law level count (lane l: c_(l&15) 2^(l>>4) and x 16) ----
    float thr_lo = 0.0f, thr_hi = 0.0f;
    {
        int seen = -1;
        float ci = 4.0f;
        for (int bin = 0; bin < 32; ++bin) {  // the i-th threshold < 4 of the 32-bin table is c_i
            const float tb = k_ulaw_thr[2 * bin];
            if (tb < 4.0f) {
                ++seen;
                if (seen == k16) ci = tb;
            }
        }
        thr_lo = ci * (float)(1 << (lane >> 4));
        thr_hi = thr_lo * 16.0f;
    }
    const unsigned long long seed = P.seeds[b];
    float mem = 0.0f;  // de-emphasis state (wave 1)
    const int t_first = FPC_LPC_ORDER + 1;
    float lpc_part = 0.0f;  // drawing wave: taps 2..16 of the next prediction
    __syncthreads();

    for (int fr = 0; fr < T; ++fr) {
        const float* feat = P.features + ((size_t)b * T + fr) * FPC_NB_FEATURES;
        // voiced frames (pdf sharpening on) keep a separate parallel leaf phase: one more barrier
        const float shape_e = landed(fpc_shape_exponent(feat[19]));
        const bool voiced = shape_e > 0.0f;
        const float* fa = feat + (FPC_NB_FEATURES - FPC_LPC_ORDER);  // this frame's LPC, and the next frame's
        const float* fan = fa + (fr + 1 < T ? FPC_NB_FEATURES : 0);
        const float a_cur = landed(fa[k16]), a_nxt = landed(fan[k16]), a0_cur = landed(fa[0]), a0_nxt = landed(fan[0]);
        if (tid < FPC_FRAME_SIZE)  // first read by the drawing wave behind barrier Z2 of this frame's first sample
            L.uframe[tid] = fpc_philox_uniform(seed, (uint32_t)(fr * FPC_FRAME_SIZE + tid));
        if (tid >= 256 && tid < 256 + GB) L.cfb[tid - 256] = P.cfb[((size_t)b * T + fr) * GB + (tid - 256)];
        if (gate_lane) {  // this frame's conditioning values of the unit: written and read by the same lane
            const float* cfa = P.cfa + ((size_t)b * T + fr) * GA;
            L.cfa[unit] = cfa[unit];
            L.cfa[RNN_A + unit] = cfa[RNN_A + unit];
            L.cfa[2 * RNN_A + unit] = cfa[2 * RNN_A + unit];
        }
        // (L.cfb is first read behind barrier Y of the frame's first sample, L.cfa by its writer)

        for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
            const int t = fr * FPC_FRAME_SIZE + i;
            const bool stamp_on = STAMP && blockIdx.x == 0 && t >= FPC_STAMP_T0 && t < FPC_STAMP_T0 + FPC_STAMP_NS;
            // ================= X..Y: gather + GRU_A gates =================
            __builtin_amdgcn_s_setprio(3);
            const uint4 ctl = *reinterpret_cast<const uint4*>(&L.o_sig);  // all lanes: one broadcast read
            F3 ta, tb, tc;
            ta.x = ta.y = ta.z = tb.x = tb.y = tb.z = tc.x = tc.y = tc.z = undef_f();
            float h_own = undef_f(), uz = undef_f(), ur = undef_f(), unb = undef_f(), cz = undef_f(), cr = undef_f(),
                  cn = undef_f();
            if (gate_lane) {
                // uniform base + 32-bit byte offset (the global_load saddr form: no 64-bit VALU address math)
                ta = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.x + voff));
                tb = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.y + voff));
                tc = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.z + voff));
                // while the gather is in flight: recurrent terms of the three rows of the unit =
                // diagonal + tree over the row group's partial sums (written before barrier X)
                h_own = L.s1[upos];
                unb = fmaf(L.diag[2 * RNN_A + unit], h_own, part_tree<QN>(&L.part[2 * RNN_A + unit])) + L.brn_a[unit];
                uz = fmaf(L.diag[unit], h_own, part_tree<QZR>(&L.part[unit]));
                ur = fmaf(L.diag[RNN_A + unit], h_own, part_tree<QZR>(&L.part[RNN_A + unit]));
                cz = L.cfa[unit], cr = L.cfa[RNN_A + unit], cn = L.cfa[2 * RNN_A + unit];
            }
            {  // still in the gather's shadow: recurrent part of GRU_B (balanced tree over the 16 products, per row)
                const float s2k = L.s2[k16];
                ub_z = ub0 * s2k, ub_r = ub1 * s2k, ub_n = ub2 * s2k;
                row_bfly16x3(ub_z, ub_r, ub_n);
            }
            if (wave == DRAW_WAVE) {
                // prediction of the NEXT sample: taps 2..16 as a balanced tree over the 16 lanes of
                // the row (lane k holds tap k+1, lane 0 contributes 0); the newest tap comes with the draw
                const float am = i == FPC_FRAME_SIZE - 1 ? a_nxt : a_cur;  // next sample belongs to the next frame
                const float hk = L.hist[(t - k16) & 15];
                lpc_part = row_bfly16(k16 ? am * hk : 0.0f);
            }
            if (gate_lane) {
                const float gz = ((ta.x + tb.x) + tc.x) + cz;
                const float gr = ((ta.y + tb.y) + tc.y) + cr;
                const float gn = ((ta.z + tb.z) + tc.z) + cn;
                const float z = lut_sigmoid(L.tt, gz + uz);
                const float r = lut_sigmoid(L.tt, gr + ur);
                const float n = lut_tanh(L.tt, fmaf(r, unb, gn));
                L.s1[upos] = fmaf(z, h_own - n, n);
            }
            FPC_BARRIER(0)  // Y
            // ================= Y..Z1: GRU_B =================
            {
                f2 gacc[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) gacc[g] = splat2(0.0f);
                const unsigned so = opaque(gsoff);
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const f2 sv = *reinterpret_cast<const f2*>(&L.s1[so + 4 * m]);  // units (h, h+2) of the float4
#pragma unroll
                    for (int g = 0; g < 3; ++g) gacc[g] = fma2(wb[g][m], sv, gacc[g]);
                }
                // leaf pair (x+z or y+w) in the lane, (x+z)+(y+w) across the lane pair, then the 16 slices
                float a3z = gacc[0].x + gacc[0].y, a3r = gacc[1].x + gacc[1].y, a3n = gacc[2].x + gacc[2].y;
                pair_bfly32x3(a3z, a3r, a3n);
                if (gru_out) {
                    const float z = lut_sigmoid(L.tt, (a3z + L.cfb[gu]) + ub_z);
                    const float r = lut_sigmoid(L.tt, (a3r + L.cfb[RNN_B + gu]) + ub_r);
                    const float n = lut_tanh(L.tt, fmaf(r, ub_n + L.brn_b[gu], a3n + L.cfb[2 * RNN_B + gu]));
                    s2_own = fmaf(z, s2_own - n, n);
                    if (k16 == 0) L.s2[gu] = s2_own;
                }
            }
            FPC_BARRIER(1)  // Z1
            // ================= Z1..Z2: dual FC, lane = (node, channel) =================
            {
                float dacc = fbias, dodd = 0.0f;  // two chains: bias + even inputs, odd inputs
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 sv = *reinterpret_cast<const float4*>(&L.s2[4 * k4]);
                    dacc = fmaf(fw[4 * k4], sv.x, dacc);
                    dodd = fmaf(fw[4 * k4 + 1], sv.y, dodd);
                    dacc = fmaf(fw[4 * k4 + 2], sv.z, dacc);
                    dodd = fmaf(fw[4 * k4 + 3], sv.w, dodd);
                }
                const float tch = lut_tanh(L.tt, dacc + dodd);
                const float prod = ffac * tch;              // channel 0: f0 t0
                const float p0 = dpp_f<DPP_XOR1>(prod);     // channel-1 lane: its node's f0 t0
                const float v = fmaf(ffac, tch, p0);        // channel 1: fmaf(f1, t1, f0 t0)
                const float qv = lut_sigmoid(L.tt, v);
                // both branch factors of the node: a leaf reads the one its bit selects, no select on the draw's chain
                if (fch) reinterpret_cast<float2*>(L.qq)[fnode] = make_float2(1.0f - qv, qv);
            }
            FPC_BARRIER(2)  // Z2
            // branch factor j (root = 0) of a leaf is float 2*(2^j + (leaf >> (8-j))) + bit_(7-j)(leaf)
            //                                            = 2*2^j + (leaf >> (7-j)) of qq
            const float* qf = reinterpret_cast<const float*>(L.qq);
            if (voiced) {
                // ---- voiced frame, Z2..Z3: leaf probability + sharpening, lanes 0..255 ----
                if (wave < 4) {
                    const unsigned slv = opaque((unsigned)tid);
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = qf[(2u << j) + (slv >> (7 - j))];
                    const float p = ((((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5])) * f[6]) * f[7];
                    L.p[slv] = fpc_shape_pow(p, shape_e);
                }
                FPC_BARRIER(3)  // Z3
            }
            if (wave == DRAW_WAVE) {
                // ================= the draw =================
                const float uf = L.uframe[i];  // this sample's uniform
                float4 p4;                     // probabilities of leaves 4*lane .. 4*lane+3
                float thr = 0.002f;  // the tree pdf sums to 1 by construction: only sharpened pdfs are totalled
                if (voiced) {
                    p4 = *reinterpret_cast<const float4*>(&L.p[4 * lane]);
                    float rs = row_bfly16((p4.x + p4.y) + (p4.z + p4.w));
                    rs = add_bcast<DPP_BCAST15, 0xa>(rs);  // rows 1,3 += rows 0,2
                    rs = add_bcast<DPP_BCAST31, 0xc>(rs);  // row 3 = (r2+r3)+(r0+r1): the balanced total
                    thr = 0.002f * lane_val(rs, 63);
                } else {
                    // unvoiced frame: 4 leaves per lane straight from the factor pairs (same product order per
                    // leaf); no separate leaf phase, no barrier
                    const unsigned lv = (unsigned)lane;  // leaves 4*lv .. 4*lv+3: factors 0..5 shared
                    float f[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) f[j] = qf[(2u << j) + (lv >> (5 - j))];
                    const float2 q6 = *reinterpret_cast<const float2*>(&qf[128u + 2u * lv]);   // node 64 + lv
                    const float4 q7 = *reinterpret_cast<const float4*>(&qf[256u + 4u * lv]);   // nodes 128 + 2 lv, + 1
                    FPC_DSTAMP(0)
                    const float pre = ((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5]);
                    const float lo = pre * q6.x, hi = pre * q6.y;
                    p4.x = lo * q7.x;
                    p4.y = lo * q7.y;
                    p4.z = hi * q7.z;
                    p4.w = hi * q7.w;
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
                FPC_DSTAMP(1)
                float I = P3;  // Kogge-Stone inside each row of 16 lanes
                I = I + dpp_f<DPP_ROW_SHR + 1>(I);
                I = I + dpp_f<DPP_ROW_SHR + 2>(I);
                I = I + dpp_f<DPP_ROW_SHR + 4>(I);
                I = I + dpp_f<DPP_ROW_SHR + 8>(I);
                I = add_bcast<DPP_BCAST15, 0xa>(I);  // block offsets by row broadcasts
                I = add_bcast<DPP_BCAST31, 0xc>(I);
                const float rthr = uf * lane_val(I, 63);
                FPC_DSTAMP(2)
                // the draw = number of leaves whose inclusive prefix is <= the threshold: four compares to lane
                // masks, four s_bcnt1 (the last leaf of a lane carries the scan value I itself)
                const float O = dpp_f<DPP_WAVE_SHR1>(I);  // exclusive offset of the lane (lane 0: +0)
                const unsigned long long m0 = __builtin_amdgcn_fcmpf(O + c0, rthr, 5 /* FCMP_OLE */);
                const unsigned long long m1 = __builtin_amdgcn_fcmpf(O + P1, rthr, 5);
                const unsigned long long m2 = __builtin_amdgcn_fcmpf(O + P2, rthr, 5);
                const unsigned long long m3 = __builtin_amdgcn_fcmpf(I, rthr, 5);
                int exc = (__popcll(m0) + __popcll(m1)) + (__popcll(m2) + __popcll(m3));
                exc = exc > 255 ? 255 : exc;
                FPC_DSTAMP(3)
                // the winner's record, wave-uniform: signal, next prediction (newest tap on top of the tree taken
                // in the gather's shadow), the mu-law levels of both
                const float cpcm = __uint_as_float(ctl.w) + L.ulaw[exc];
                const float a0 = i == FPC_FRAME_SIZE - 1 ? a0_nxt : a0_cur;
                const float cpred = -fmaf(a0, cpcm, lpc_part);
                const unsigned es = ulaw_level_ballot(cpcm, thr_lo, thr_hi);
                const unsigned ep = ulaw_level_ballot(cpred, thr_lo, thr_hi);
                FPC_DSTAMP(4)
                if (lane == 0) {
                    // control block {o_sig, o_pred, o_exc, pred}: one 16-byte store
                    const unsigned rowb = (unsigned)(GA * 4);
                    *reinterpret_cast<uint4*>(&L.o_sig) =
                        make_uint4(es * rowb, (256u + ep) * rowb, (512u + (unsigned)exc) * rowb, __float_as_uint(cpred));
                    L.hist[t & 15] = cpcm;
                }
                FPC_DSTAMP(5)
                if (W0) {
                    __builtin_amdgcn_s_setprio(0);
                    FPC_SPARSE()
                }
            } else {
                // ================= background: this lane's share of the sparse product of s1' =================
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_sleep(1);  // the drawing wave's factor reads reach the LDS first
                FPC_SPARSE()
                if (wave == PCM_WAVE && t > t_first) {
                    // off the sample-to-sample chain: de-emphasis and PCM store of the PREVIOUS sample (its value
                    // sits in the history ring since the last barrier X)
                    mem = fmaf(FPC_PREEMPH, mem, L.hist[(t - 1) & 15]);
                    if (lane == 0) out[t - 1] = fpc_pcm16(mem);
                }
            }
            FPC_BARRIER(4)  // X
        }
    }
    if (wave == PCM_WAVE) {  // the last sample (behind the last barrier X)
        const int t = T * FPC_FRAME_SIZE;
        if (t > t_first) {
            mem = fmaf(FPC_PREEMPH, mem, L.hist[(t - 1) & 15]);
            if (lane == 0) out[t - 1] = fpc_pcm16(mem);
        }
    }
}
#undef FPC_BARRIER
#undef FPC_STAMP
#undef FPC_DSTAMP
#undef FPC_COLS
#undef FPC_PUBLISH
#undef FPC_SPARSE
