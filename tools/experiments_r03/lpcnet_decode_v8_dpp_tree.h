// lpcnet_decode.h -- the persistent per-utterance sample loop of the LPCNet-style vocoder
// (included by lpcnet.hip only; gfx950).
//
// One 512-thread workgroup (8 wave64, 2 per SIMD, up to 256 VGPRs each) per utterance.  Every lane carries 4 blocks
// (8x4) of the block-sparse recurrent matrix of GRU_A in VGPRs; every wave owns 48 GRU_A units (lanes 0..47 = "gate
// lanes").  The two waves of a SIMD have complementary roles and take turns on the sample-to-sample chain:
//   waves 0-3 "S": dual FC (lane = tree node), draw (wave 0), per-leaf candidate records ("window")
//   waves 4-7 "M": GRU_B (16 lanes per unit, weights in VGPRs)
// Four workgroup barriers per output sample (X, Y, Z1, Z2; voiced frames add Z3):
//   X  control block (byte offsets of the three table rows selected by the drawn sample) published
//        all: gate lanes gather their 3 x 12 bytes of the three embedding x kernel table rows; in the shadow of that
//             L2 round trip: the recurrent terms (3 reduced row sums + diagonal), and
//             S: candidate records of the 256 leaves (LPC taps 2..16, mu-law levels of signal and prediction)
//             M: recurrent part of GRU_B
//             then gates -> s1'
//   Y  s1' ready
//        M: GRU_B -> s2'           S (background): sparse product of s1' (16 columns x 4 row pairs), row sums reduced
//                                     over the lanes of a row group by masked DPP adds, one 32-byte store per group
//   Z1 s2' ready
//        S: dual FC -> 255 branch-factor pairs      M (background): its share of the sparse product
//   Z2 factors ready
//        wave 0: leaf probabilities, tail cut, scan, draw, control block      M: reduction + store of its row sums
//        wave 1: de-emphasis + PCM store of the previous sample
//        (voiced frame: all 256 S lanes compute leaf probability + sharpening first, barrier Z3)
// HBM is touched only for the gathered table rows (L2-resident), the per-frame conditioning rows and 2 bytes of PCM
// per sample.
//
// Canonical evaluation orders (DESIGN.md "Vocoder numerics") are those of
// oracle/fpc_oracle.c::orc_lpcnet_synthesize; results are bit-identical.
#pragma once

constexpr int NTHREADS = 512;
constexpr int NSAMP = 256;      // lanes of one role: S lanes = tree nodes / leaves, M lanes = (GRU_B unit, input slice)
constexpr int NMAT = 512;       // every lane carries 4 blocks of the sparse product
constexpr int UPW = RNN_A / 8;  // gate units per wave (lanes 0..47)
#ifndef FPC_M_SPLIT
#define FPC_M_SPLIT 16  // sparse-product columns (of 16) an M wave computes under the dual FC; the rest under the draw
#endif

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
    const float* lane_wb;    // [72][256]  GRU_B input weights [gate][24 inputs] of (unit, slice)
    const float* lane_ub;    // [3][256]   GRU_B recurrent weights ub[k][gate*16+unit]
    const float* lane_fc;    // [36][256]  dual-FC of node = lane: 2x16 weights, 2 bias, 2 factor
    const float* diag;       // [1152]
    const float* brn_a;      // [384]
    const float* brn_b;      // [16]
    const float* ulaw_tab;   // [256]
    unsigned long long* stamps;  // diagnostic only
};

__device__ const float k_ulaw_thr[64] = FPC_ULAW_TABLE_INIT;

// Every field sits in the first 64 KB (bases fold into the DS instructions' offset fields), the activation
// table at offset 0.
struct __attribute__((aligned(16))) DecodeLds {
    float2 tt[FPC_TANH_TABLE_SIZE - 1];  // fpc_tanh_lut table as (T[k], T[k+1] - T[k]) pairs, built at kernel start
    float s1[RNN_A];
    float usum[GA];           // row sums of the sparse product [z|r|h][unit] (the oracle's `tsum`)
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
    float4 cand[256];  // per leaf, if it wins the draw: (pcm, next prediction, o_sig, o_pred)
    float ulaw_thr[64];  // fpc_lin2ulaw_tab table
    float uframe[FPC_FRAME_SIZE];
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
// one level of the zero-padded balanced tree over the lanes q = 0..Q-1 of a row group (consecutive lanes of one
// 16-lane row), for the 8 row sums a lane holds:  v[q] += v[q + S]  where the lane's mask m is 1.0 (q % 2S == 0
// and q + S < Q), unchanged where it is 0.0 (x * 1 + v and x * 0 + v are exact; sums are never -0).  Eight
// independent registers: each DPP read comes 8 instructions after the register's last write.
#define FPC_TREE_LEVEL(S, v, m)                                                                            \
    asm volatile("s_nop 1\n\t"                                                                             \
                 "v_fmac_f32_dpp %0, %0, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %1, %1, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %2, %2, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %3, %3, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %4, %4, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %5, %5, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %6, %6, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 "v_fmac_f32_dpp %7, %7, %8 row_shl:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"    \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                 : "v"(m))
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

// STAMP=true is a diagnostic build (env FPC_DECODE_STAMPS=1): lane 0 of every wave w of block 0
// (slots 16w..16w+9) accumulates s_memtime deltas: slot 2k = work before barrier k,
// slot 2k+1 = wait at barrier k (k: 0 Y, 1 Z1, 2 Z2, 3 Z3, 4 X); perturbs the timing, never timed.
#define FPC_STAMP(k)                                                  \
    if (STAMP) {                                                      \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        st_acc[k] += now_ - st_last;                                  \
        st_last = now_;                                               \
    }
#define FPC_BARRIER(k)   \
    FPC_STAMP(2 * (k))   \
    __syncthreads();     \
    FPC_STAMP(2 * (k) + 1)

struct F3 {
    float x, y, z;
};

// ---- gate lanes, first half (right behind barrier X): gather the three table rows of the unit and, while they
//      are in flight, its recurrent terms = diagonal + reduced row sums of the sparse product ----
#define FPC_GATE_ISSUE()                                                                                          \
    const uint4 ctl = *reinterpret_cast<const uint4*>(&L.o_sig); /* all lanes: one broadcast read */              \
    F3 ta = {0.f, 0.f, 0.f}, tb = ta, tc = ta;                                                                    \
    float h_own = 0.f, uz = 0.f, ur = 0.f, unb = 0.f, cz = 0.f, cr = 0.f, cn = 0.f;                               \
    if (gate_lane) {                                                                                              \
        /* uniform base + 32-bit byte offset (the global_load saddr form: no 64-bit VALU address math) */         \
        ta = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.x + voff));                                         \
        tb = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.y + voff));                                         \
        tc = *reinterpret_cast<const F3*>(tabc + (size_t)(ctl.z + voff));                                         \
        h_own = L.s1[unit];                                                                                       \
        unb = fmaf(L.diag[2 * RNN_A + unit], h_own, L.usum[2 * RNN_A + unit]) + L.brn_a[unit];                    \
        uz = fmaf(L.diag[unit], h_own, L.usum[unit]);                                                             \
        ur = fmaf(L.diag[RNN_A + unit], h_own, L.usum[RNN_A + unit]);                                             \
        cz = L.cfa[unit], cr = L.cfa[RNN_A + unit], cn = L.cfa[2 * RNN_A + unit];                                 \
    }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
// ---- gate lanes, second half: gates -> s1' ----
#define FPC_GATE_FINISH()                                              \
    __builtin_amdgcn_sched_barrier(0);                                 \
    if (gate_lane) {                                                   \
        const float gz = ((ta.x + tb.x) + tc.x) + cz;                  \
        const float gr = ((ta.y + tb.y) + tc.y) + cr;                  \
        const float gn = ((ta.z + tb.z) + tc.z) + cn;                  \
        const float z = lut_sigmoid(L.tt, gz + uz);                    \
        const float r = lut_sigmoid(L.tt, gr + ur);                    \
        const float n = lut_tanh(L.tt, fmaf(r, unb, gn));              \
        L.s1[unit] = fmaf(z, h_own - n, n);                            \
    }

// ---- the sparse product of s1' with this lane's 4 blocks (16 columns of 8 rows): columns [FROM, TO) ----
#define FPC_COLS(FROM, TO)                                                                                        \
    _Pragma("unroll") for (int bc = (FROM); bc < (TO); ++bc) {                                                    \
        if (bc == 0) {                                                                                            \
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
// ---- in-lane tree level (leaf pair), tree over the lanes of the row group, one 32-byte store per group ----
#define FPC_REDUCE_STORE()                                                                  \
    {                                                                                       \
        float v[8];                                                                         \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp) {                                  \
            const f2 s = acc[rp] + a[rp];                                                   \
            v[2 * rp] = s.x, v[2 * rp + 1] = s.y;                                           \
        }                                                                                   \
        if (LV >= 1 && need1) FPC_TREE_LEVEL(1, v, tm1);                                    \
        if (LV >= 2 && need2) FPC_TREE_LEVEL(2, v, tm2);                                    \
        if (LV >= 3 && need4) FPC_TREE_LEVEL(4, v, tm4);                                    \
        if (LV >= 4 && need8) FPC_TREE_LEVEL(8, v, tm8);                                    \
        if (paddr_ != 0u) { /* lane q = 0 of a row group: its 8 row sums */                \
            typedef float v4f __attribute__((ext_vector_type(4)));                          \
            typedef __attribute__((address_space(3))) v4f lds_v4f;                          \
            lds_v4f* pp = (lds_v4f*)(size_t)paddr_; /* 32-byte aligned: two ds_write_b128 */ \
            v4f lo, hi;                                                                     \
            lo.x = v[0], lo.y = v[1], lo.z = v[2], lo.w = v[3];                             \
            hi.x = v[4], hi.y = v[5], hi.z = v[6], hi.w = v[7];                             \
            pp[0] = lo;                                                                     \
            pp[1] = hi;                                                                     \
        }                                                                                   \
    }

// LV: levels of the tree over the lanes of a row group = ceil(log2(widest row group in lanes)), 0..4
template <bool STAMP, int LV>
__global__ __launch_bounds__(NTHREADS) void k_decode(const DecodeParams P) {
    __shared__ DecodeLds L;
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = STAMP ? __builtin_readcyclecounter() : 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, T = P.T;

    // ---- LDS init ----
    for (int i = tid; i < RNN_A; i += NTHREADS) {
        L.s1[i] = 0.0f;
        L.brn_a[i] = P.brn_a[i];
    }
    for (int i = tid; i < GA; i += NTHREADS) {
        L.diag[i] = P.diag[i];
        L.usum[i] = 0.0f;  // U . 0
    }
    if (tid < 64) L.ulaw_thr[tid] = k_ulaw_thr[tid];
    for (int k = tid; k < FPC_TANH_TABLE_SIZE - 1; k += NTHREADS) {
        const float t0 = fpc_tanh_table_entry(k), t1 = fpc_tanh_table_entry(k + 1);
        L.tt[k] = make_float2(t0, t1 - t0);
    }
    if (tid < RNN_B) {
        L.s2[tid] = 0.0f;
        L.hist[tid] = 0.0f;
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
    // lane q of Q of its row group: masks of the tree levels, and where lane 0 stores the group's 8 row sums
    // (a finished LDS byte address: one VGPR, no per-sample address math; 0 = not a storing lane)
    typedef __attribute__((address_space(3))) float lds_float;
    unsigned paddr_ = 0u;
    float tm1 = 0.0f, tm2 = 0.0f, tm4 = 0.0f, tm8 = 0.0f;
    if ((metap_ >> 16) != 0) {
        const int grp = (int)(metap_ >> 16) - 1, q = (int)(metap_ & 0xff), Q = (int)((metap_ >> 8) & 0xff);
        const int gate = grp / (RNN_A / 8), rb = grp - gate * (RNN_A / 8);
        if (q == 0) paddr_ = (unsigned)(size_t)(lds_float*)&L.usum[gate * RNN_A + rb * 8];
        tm1 = (q % 2 == 0 && q + 1 < Q) ? 1.0f : 0.0f;
        tm2 = (q % 4 == 0 && q + 2 < Q) ? 1.0f : 0.0f;
        tm4 = (q % 8 == 0 && q + 4 < Q) ? 1.0f : 0.0f;
        tm8 = (q % 16 == 0 && q + 8 < Q) ? 1.0f : 0.0f;
    }
    paddr_ = opaque(paddr_);
    // levels no row group of this wave needs are skipped (wave-uniform)
    const bool need1 = __ballot(tm1 != 0.0f) != 0ull, need2 = __ballot(tm2 != 0.0f) != 0ull;
    const bool need4 = __ballot(tm4 != 0.0f) != 0ull, need8 = __ballot(tm8 != 0.0f) != 0ull;
    // ---- every wave: 48 gate lanes ----
    const bool gate_lane = lane < UPW;
    const unsigned unit = (unsigned)(wave * UPW + (gate_lane ? lane : 0));
    const unsigned voff = 12u * unit;  // bytes of (unit, z|r|h) inside a table row
    const char* tabc = reinterpret_cast<const char*>(P.tab);
    __syncthreads();

    if (wave >= 4) {
        // =========================== M role: GRU_B ===========================
        const int ml = tid - NSAMP;           // 0..255
        const int u = ml >> 4, kl = ml & 15;  // GRU_B: unit, 24-input slice
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
                    wB[g][m][h] = mk2(P.lane_wb[(g * 24 + 4 * m + 2 * h) * NSAMP + ml],
                                      P.lane_wb[(g * 24 + 4 * m + 2 * h + 1) * NSAMP + ml]);
        const float ub0 = P.lane_ub[ml], ub1 = P.lane_ub[NSAMP + ml], ub2 = P.lane_ub[2 * NSAMP + ml];
        const float brnb = P.brn_b[u];
        float s2_own = 0.0f;  // state of unit u, replicated over the 16 lanes of its row

        for (int fr = 0; fr < T; ++fr) {
            // voiced frames (pdf sharpening on) keep a separate parallel leaf phase: one more barrier
            const bool voiced = fpc_shape_exponent(P.features[((size_t)b * T + fr) * FPC_NB_FEATURES + 19]) > 0.0f;
            const float* cfb = P.cfb + ((size_t)b * T + fr) * GB;
            const float cfb_z = cfb[u], cfb_r = cfb[RNN_B + u], cfb_n = cfb[2 * RNN_B + u];
            if (gate_lane) {  // this frame's conditioning values of the unit: written and read by the same lane
                const float* cfa = P.cfa + ((size_t)b * T + fr) * GA;
                L.cfa[unit] = cfa[unit];
                L.cfa[RNN_A + unit] = cfa[RNN_A + unit];
                L.cfa[2 * RNN_A + unit] = cfa[2 * RNN_A + unit];
            }
            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                // ---- X..Y: gather + gates; in the gather's shadow the recurrent part of GRU_B ----
                __builtin_amdgcn_s_setprio(3);
                FPC_GATE_ISSUE()
                const float s2k = L.s2[kl];
                float ub_z = ub0 * s2k, ub_r = ub1 * s2k, ub_n = ub2 * s2k;
                row_bfly16x3(ub_z, ub_r, ub_n);
                FPC_GATE_FINISH()
                FPC_BARRIER(0)  // Y
                // ---- Y..Z1: GRU_B (row of 16 lanes = unit; lane = 24 inputs = 4 leaves of 6, leaf of
                //      float4 component c takes inputs c + 4m) ----
                {
                    const unsigned klv = (unsigned)kl;  // (24 * kl: a lane constant the compiler keeps in a register)
                    f2 gacc[3][2];
#pragma unroll
                    for (int g = 0; g < 3; ++g) gacc[g][0] = gacc[g][1] = splat2(0.0f);
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const float4 h4 = *reinterpret_cast<const float4*>(&L.s1[24 * klv + 4 * m]);
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            gacc[g][0] = fma2(wB[g][m][0], mk2(h4.x, h4.y), gacc[g][0]);
                            gacc[g][1] = fma2(wB[g][m][1], mk2(h4.z, h4.w), gacc[g][1]);
                        }
                    }
                    float a3[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {  // (leaf x + leaf z) + (leaf y + leaf w)
                        const f2 pr = gacc[g][0] + gacc[g][1];
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
                // ---- background from here: this lane's share of the sparse product of s1' ----
                __builtin_amdgcn_s_setprio(0);
                f2 acc[4], a[4];
                float hv0[8], hv1[8];
#pragma unroll
                for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f);
                FPC_COLS(0, FPC_M_SPLIT)
                FPC_BARRIER(2)  // Z2
                if (voiced) {
                    FPC_BARRIER(3)  // Z3 (voiced frames only: the S waves' leaf phase ends here)
                }
                FPC_COLS(FPC_M_SPLIT, 16)
                FPC_REDUCE_STORE()
                FPC_BARRIER(4)  // X
            }
        }
        if (STAMP && blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 10; ++k) P.stamps[16 * wave + k] = st_acc[k];
    } else {
        // =========================== S role: dual FC, draw, window ===========================
        const int sl = tid;       // 0..255: tree node of the dual FC, leaf of the window
        const int kl = sl & 15;   // LPC tap of the window tree
        const unsigned long long seed = P.seeds[b];
        const float my_ulaw = P.ulaw_tab[sl];
        // dual-FC weights of tree node sl: 16 (channel 0, channel 1) pairs, then bias pair and factor pair
        f2 fcw[18];
#pragma unroll
        for (int k = 0; k < 16; ++k) fcw[k] = mk2(P.lane_fc[k * NSAMP + sl], P.lane_fc[(16 + k) * NSAMP + sl]);
        fcw[16] = mk2(P.lane_fc[32 * NSAMP + sl], P.lane_fc[33 * NSAMP + sl]);
        fcw[17] = mk2(P.lane_fc[34 * NSAMP + sl], P.lane_fc[35 * NSAMP + sl]);
        float mem = 0.0f;  // de-emphasis state (wave 1)
        const int t_first = FPC_LPC_ORDER + 1;

        for (int fr = 0; fr < T; ++fr) {
            const float* feat = P.features + ((size_t)b * T + fr) * FPC_NB_FEATURES;
            const float shape_e = fpc_shape_exponent(feat[19]);
            const float* fa = feat + (FPC_NB_FEATURES - FPC_LPC_ORDER);  // this frame's LPC, and the next frame's
            const float* fan = fa + (fr + 1 < T ? FPC_NB_FEATURES : 0);
            const float a_cur = fa[kl], a_nxt = fan[kl], a0_cur = fa[0], a0_nxt = fan[0];
            if (sl < FPC_FRAME_SIZE)  // first read behind barrier Z1 of this frame's first sample
                L.uframe[sl] = fpc_philox_uniform(seed, (uint32_t)(fr * FPC_FRAME_SIZE + sl));
            if (gate_lane) {  // this frame's conditioning values of the unit: written and read by the same lane
                const float* cfa = P.cfa + ((size_t)b * T + fr) * GA;
                L.cfa[unit] = cfa[unit];
                L.cfa[RNN_A + unit] = cfa[RNN_A + unit];
                L.cfa[2 * RNN_A + unit] = cfa[2 * RNN_A + unit];
            }

            for (int i = (fr == 0 ? FPC_LPC_ORDER + 1 : 0); i < FPC_FRAME_SIZE; ++i) {
                const int t = fr * FPC_FRAME_SIZE + i;
                // ---- X..Y: gather + gates; in the gather's shadow everything of this sample's draw that only
                //      needs the previous draw: LPC history chain and the 256 leaf candidates ----
                __builtin_amdgcn_s_setprio(3);
                FPC_GATE_ISSUE()
                {
                    // prediction of the NEXT sample: taps 2..16 as a balanced tree over the 16 lanes of
                    // the row (lane kl holds tap kl+1, lane 0 contributes 0), the newest tap by one fma
                    const bool lastsmp = i == FPC_FRAME_SIZE - 1;  // next sample belongs to the next frame
                    const float am = lastsmp ? a_nxt : a_cur;
                    const float a0 = lastsmp ? a0_nxt : a0_cur;
                    const float hk = L.hist[(t - kl) & 15];
                    const float part = row_bfly16(kl ? am * hk : 0.0f);
                    // what the control block becomes if leaf `sl` wins this sample's draw
                    const float cpcm = __uint_as_float(ctl.w) + my_ulaw;
                    const float cpred = -fmaf(a0, cpcm, part);
                    // ... with the byte offsets of the table rows of its signal and prediction levels (x GA x 4 = 4096 + 512)
                    const unsigned es = (unsigned)fpc_lin2ulaw_tab(cpcm, L.ulaw_thr);
                    const unsigned ep = 256u + (unsigned)fpc_lin2ulaw_tab(cpred, L.ulaw_thr);
                    L.cand[sl] = make_float4(cpcm, cpred, __uint_as_float((es << 12) + (es << 9)),
                                             __uint_as_float((ep << 12) + (ep << 9)));
                }
                FPC_GATE_FINISH()
                FPC_BARRIER(0)  // Y
                // ---- Y..Z1 (background: the M waves run GRU_B): this lane's share of the sparse product ----
                __builtin_amdgcn_s_setprio(0);
                {
                    f2 acc[4], a[4];
                    float hv0[8], hv1[8];
#pragma unroll
                    for (int rp = 0; rp < 4; ++rp) acc[rp] = a[rp] = splat2(0.0f);
                    FPC_COLS(0, 16)
                    FPC_REDUCE_STORE()
                }
                __builtin_amdgcn_s_setprio(3);
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
                        const unsigned slv = opaque((unsigned)sl);
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
                            make_float4(cd.z, cd.w, __uint_as_float((512u + (unsigned)exc) * (unsigned)(GA * 4)), cd.y);
                        L.hist[t & 15] = cd.x;
                    }
                } else if (wave == 1) {
                    // off the sample-to-sample chain: de-emphasis and PCM store of the PREVIOUS sample (its value
                    // sits in the history ring since the last barrier X)
                    if (t > t_first) {
                        mem = fmaf(FPC_PREEMPH, mem, L.hist[(t - 1) & 15]);
                        if (lane == 0) out[t - 1] = fpc_pcm16(mem);
                    }
                }
                FPC_BARRIER(4)  // X
            }
        }
        if (wave == 1) {  // the last sample (behind the last barrier X)
            const int t = T * FPC_FRAME_SIZE;
            if (t > t_first) {
                mem = fmaf(FPC_PREEMPH, mem, L.hist[(t - 1) & 15]);
                if (lane == 0) out[t - 1] = fpc_pcm16(mem);
            }
        }
        if (STAMP && blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 10; ++k) P.stamps[16 * wave + k] = st_acc[k];
    }
}
#undef FPC_BARRIER
#undef FPC_STAMP
#undef FPC_GATE_ISSUE
#undef FPC_GATE_FINISH
#undef FPC_COLS
#undef FPC_REDUCE_STORE
