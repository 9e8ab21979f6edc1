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
//   k_frame_mfma     frame-rate layers (conv k=3 / dense) on f32 MFMA (k-ordered fmaf chains)
//   k_decode         persistent per-utterance sample loop (lpcnet_decode.h): one 768-thread workgroup
//                    (12 wave64: 4 sampler + 8 mat-vec waves) per utterance; sparse GRU_A, GRU_B and
//                    dual-FC weights live in VGPRs for the whole utterance, recurrent state, activation
//                    table and the partial sums of the sparse product live in LDS, HBM is touched only
//                    for the per-frame conditioning vectors, the embedding-table rows and the PCM output.
//   k_decode2        the same loop for TWO utterances per workgroup (lpcnet_decode2.h): batches larger than the CU count
#include "fpc_common.h"
#include <string>
#include <algorithm>
#include <memory>

namespace {

constexpr int RNN_A = 384, RNN_B = 16, EMB = 128;
constexpr int GA = 3 * RNN_A;  // 1152
constexpr int GB = 3 * RNN_B;  // 48
constexpr int NROWGRP = GA / 8;  // 144 groups of 8 gate rows

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
        // stored gate-interleaved [table][entry][unit][z,r,h]: a gate lane fetches 12 contiguous bytes
        tab[((size_t)s * 256 + e) * GA + (row % RNN_A) * 3 + row / RNN_A] = (float)acc;
    }
}

// ---------------------------------------------------------------------------------
// frame-rate layers on the matrix cores.  y[f][o] = act(bias[o] + sum_k x_f[k] W[k][o]);
// v_mfma_f32_16x16x4_f32 accumulates D = A.B + C as a k-ordered fmaf chain starting from C, so
// with C = bias and k ascending the result is bit-identical to the canonical chain of the CPU
// oracle (dense_chain in oracle/fpc_oracle.c).  A = 16 frames x 4 k (from an LDS-staged tile of
// one utterance), B = 4 k x 16 outputs (weights straight from L2), 2 output tiles per wave.
//   MODE 0: x_f = x[f][0..K)                     (dense, K == C)
//   MODE 1: 'same' k=3 conv, K = 3*C, x_f[tap*C+c] = x[f+tap-1][c], zero rows outside the utterance
//   MODE 2: like 1 with the input row built on the fly from the 36-float feature frame:
//           20 features | 64-dim pitch embedding (C = 84)
// block = 256 threads = 4 waves = 16 frames x 128 outputs; grid = (B * ceil(frames/16), ceil(N/128)).
// Frames are ABSOLUTE frame numbers of the utterance (FrameView): a launch evaluates frames [o0, o1), reads the input
// rows it needs from a buffer that holds frames x_t0 .. x_t0 + x_T - 1 of every utterance (zero rows outside [0, T):
// the 'same' padding of the utterance, not of the chunk) and writes into a buffer that holds y_t0 .. y_t0 + y_T - 1
// -- the whole utterance in one launch (x_t0 = y_t0 = o0 = 0, x_T = y_T = o1 = T) or a chunk of it with its halo rows.
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct FrameView {
    int T;           // frames of an utterance
    int o0, o1;      // frames evaluated by this launch
    int x_t0, x_T;   // input buffer: first frame held, frames per utterance
    int y_t0, y_T;   // output buffer
};

template <int MODE>
__global__ __launch_bounds__(256) void k_frame_mfma(const float* __restrict__ x, int ldx, int C, int K,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ bias, int N,
                                                    float* __restrict__ y, const FrameView V, int do_tanh,
                                                    const float* __restrict__ embed_pitch) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [18][Cs]
    const int Cs = ((C + 29) / 32) * 32 + 2;                    // row stride = 2 mod 32: conflict-free A reads
    const int T = V.T;
    const int tiles = (V.o1 - V.o0 + 15) / 16;
    const int b = blockIdx.x / tiles, t0 = V.o0 + (blockIdx.x % tiles) * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- stage rows t0-1 .. t0+16 of this utterance (zeros outside [0,T)) ----
    const int nrows = MODE == 0 ? 16 : 18, roff = MODE == 0 ? 0 : -1;
    for (int e = tid; e < nrows * C; e += 256) {
        const int rr = e / C, c = e - rr * C;
        const int tt = t0 + rr + roff;
        float v = 0.0f;
        if (tt >= 0 && tt < T) {
            const size_t f = (size_t)b * V.x_T + (tt - V.x_t0);
            if (MODE == 2) {
                const float* fr = x + f * FPC_NB_FEATURES;
                v = c < FPC_NB_USED_FEATURES
                        ? fr[c]
                        : embed_pitch[fpc_period_index(fr[18]) * 64 + (c - FPC_NB_USED_FEATURES)];
            } else {
                v = x[f * ldx + c];
            }
        }
        xs[rr * Cs + c] = v;
    }
    __syncthreads();
    // ---- two 16x16 output tiles per wave ----
    const int n0 = blockIdx.y * 128 + wave * 32;
    if (n0 >= N) return;  // (wave-uniform) nothing to do for this wave
    const int fi = lane & 15, kq = lane >> 4;
    const int col0 = n0 + fi, col1 = n0 + 16 + fi;
    const bool has1 = n0 + 16 < N;
    const float b0 = bias[col0], b1 = has1 ? bias[col1] : 0.0f;
    f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {b1, b1, b1, b1};
    const int c1 = has1 ? col1 : col0;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int tap = MODE == 0 ? 0 : k0 / C;  // C % 4 == 0: the 4 k's of a step share a tap
        const int cc = k0 - tap * C;
        const float a = xs[(fi + tap) * Cs + cc + kq];
        const float w0 = W[(size_t)(k0 + kq) * N + col0];
        const float w1 = W[(size_t)(k0 + kq) * N + c1];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w1, acc1, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tt = t0 + kq * 4 + r;
        if (tt < V.o1) {
            const size_t f = (size_t)b * V.y_T + (tt - V.y_t0);
            y[f * N + col0] = do_tanh ? fpc_tanhf(acc0[r]) : acc0[r];
            if (has1) y[f * N + col1] = do_tanh ? fpc_tanhf(acc1[r]) : acc1[r];
        }
    }
}

#include "lpcnet_decode.h"
#include "lpcnet_decode2.h"

}  // namespace

// =====================================================================================
// host side
// =====================================================================================
// compile-time tunables of this translation unit that differ from the shipped defaults (fpc_build_info, api.hip)
namespace fpc {
void lpcnet_build_info(std::string& out) {
    char b[64];
#define FPC_TUNE(name, value, dflt)                            \
    if ((value) != (dflt)) {                                   \
        snprintf(b, sizeof b, " " name "=%d", (int)(value));   \
        out += b;                                              \
    }
    FPC_TUNE("FPC_PRIO", FPC_PRIO, 3)
    FPC_TUNE("FPC_NA", FPC_NA, 6)
    FPC_TUNE("FPC_NB", FPC_NB, 4)
    FPC_TUNE("FPC_PCM_WHERE", FPC_PCM_WHERE, 0)
    FPC_TUNE("FPC2_N1", FPC2_N1, 14)
    FPC_TUNE("FPC2_N2", FPC2_N2, 11)
    FPC_TUNE("FPC2_PRIO3", FPC2_PRIO3, 1)
    FPC_TUNE("FPC2_WPRIO", FPC2_WPRIO, 0)
    FPC_TUNE("FPC2_SPRIO_GB", FPC2_SPRIO_GB, 3)
    FPC_TUNE("FPC2_SPRIO_FC", FPC2_SPRIO_FC, 3)
    FPC_TUNE("FPC2_GPRIO_PAIR", FPC2_GPRIO_PAIR, 3)
    FPC_TUNE("FPC2_GPRIO_SINGLE", FPC2_GPRIO_SINGLE, 3)
    FPC_TUNE("FPC2_ABL", FPC2_ABL, 0)
#undef FPC_TUNE
}
}  // namespace fpc
struct fpc_lpcnet {
    int device = 0;
    fpc::DevBuf embed_pitch, conv1_k, conv1_b, conv2_k, conv2_b, d1_k, d1_b, d2_k, d2_b;
    fpc::DevBuf ga_k, gb_k, bias_a, bias_b, tab;
    fpc::DevBuf lane_w, lane_meta, lane_wb, lane_ub, lane_fc, diag, brn_a, brn_b, ulaw_tab;
    fpc::DevBuf lane_w2, lane_meta2;  // k_decode2's placement (row groups padded to an even number of lanes)
    bool pair_ok = false;             // k_decode2 can serve this model
    int lds_read_cycles2 = 0;
    int gate_qp[3];
    int nblocks = 0, nleaves = 0;
    int lds_read_cycles_greedy = 0, lds_read_cycles = 0;  // LDS cycles of the mat-vec state reads per sample (diagnostic)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int chunk = 0;  // frames per chunk of fpc_lpcnet_synthesize (0: the whole utterance in one pass)
    int pairing = 0;  // fpc_lpcnet_set_pairing: 0 two utterances per workgroup when B exceeds the CU count, 1 always, -1 never
    int cus = 0;      // compute units of the device
    int last_streams_per_wg = 0;  // what the last fpc_lpcnet_synthesize launched (fpc_lpcnet_last_streams_per_workgroup)
};

static size_t ws_floats_per_frame() { return 128 + 128 + 128 + GA + GB; }
// frames per pass of fpc_lpcnet_synthesize: T, or the handle's chunk when the utterance is longer than that
static int pass_frames(const fpc_lpcnet* m, int T) { return (m && m->chunk > 0 && m->chunk < T) ? m->chunk : T; }

extern "C" long long fpc_lpcnet_workspace_bytes(const fpc_lpcnet* m, int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    const int C = pass_frames(m, T);
    if (C == T) return (long long)B * T * (long long)ws_floats_per_frame() * 4 + 256;
    // chunked: the first conv's rows with a halo frame either side, the carried state; independent of T
    return (long long)B * ((long long)C * (long long)ws_floats_per_frame() + 2 * 128 + STATE_FLOATS) * 4 + 256;
}

extern "C" int fpc_lpcnet_set_chunk_frames(fpc_lpcnet* m, int frames) {
    FPC_REQUIRE(m && frames >= 0, "fpc_lpcnet_set_chunk_frames: bad argument");
    m->chunk = frames;
    return FPC_OK;
}

// kernel instance by the widest row group: (update/reset gates, candidate gate) partial-sum planes
static int decode_variant_of(const int (&gate_qp)[3]) {
    const int qzr = gate_qp[0] > gate_qp[1] ? gate_qp[0] : gate_qp[1], qn = gate_qp[2];
    if (qzr <= 2 && qn <= 8) return 208;
    if (qzr <= 4 && qn <= 8) return 408;
    return 1616;
}

// the default mapping's split (fpcodec.h): how many of B utterances go two per workgroup on a device of `cus` compute units
extern "C" int fpc_lpcnet_paired_utterances(int B, int cus) {
    if (cus <= 0) cus = 256;
    if (B <= cus) return 0;
    int NP = 0;
    double best = 1e30;
    for (int p = 0; p <= (B + 2 * cus - 1) / (2 * cus); ++p) {
        const int np = std::min(B, 2 * cus * p), rest = B - np;
        const double cost = 1.56 * p + (rest + cus - 1) / cus;  // a paired round of 2 x cus utterances ~ 1.56 plain rounds of cus
        if (cost < best - 1e-9) best = cost, NP = np;
    }
    return NP;
}

extern "C" int fpc_lpcnet_set_pairing(fpc_lpcnet* m, int mode) {
    FPC_REQUIRE(m && mode >= -1 && mode <= 1, "fpc_lpcnet_set_pairing: bad argument");
    m->pairing = mode;
    return FPC_OK;
}

extern "C" int fpc_lpcnet_last_streams_per_workgroup(const fpc_lpcnet* m) { return m ? m->last_streams_per_wg : -1; }

extern "C" void fpc_lpcnet_destroy(fpc_lpcnet* m);

extern "C" int fpc_lpcnet_create(const fpc_lpcnet_weights* w, fpc_lpcnet** out) {
    FPC_REQUIRE(w && out, "fpc_lpcnet_create: null argument");
    if (!fpc::have_device()) {
        fpc::set_error("fpc_lpcnet_create: no HIP device (libfpcodec has no CPU fallback)");
        return FPC_ERR_NO_DEVICE;
    }
    const float* const* ptrs = reinterpret_cast<const float* const*>(w);
    for (size_t i = 0; i < sizeof(*w) / sizeof(float*); ++i)
        FPC_REQUIRE(ptrs[i], "fpc_lpcnet_create: weight pointer %zu is null", i);
    // owned until the last step succeeds: every early return below frees the buffers and events
    struct Del {
        void operator()(fpc_lpcnet* x) const { fpc_lpcnet_destroy(x); }
    };
    std::unique_ptr<fpc_lpcnet, Del> own(new fpc_lpcnet());
    fpc_lpcnet* m = own.get();
    FPC_HIP(hipGetDevice(&m->device));
    if (const char* e = getenv("FPC_LPCNET_CHUNK")) m->chunk = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("FPC_LPCNET_PAIRING")) m->pairing = atoi(e) > 0 ? 1 : atoi(e) < 0 ? -1 : 0;
    FPC_HIP(hipDeviceGetAttribute(&m->cus, hipDeviceAttributeMultiprocessorCount, m->device));

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
    // ---- placement of the row groups on the mat-vec lanes.  Twice: the layout of k_decode, and (even = true) the one of
    // k_decode2, whose lanes 2k, 2k+1 of a row group add their partial sums before publishing, so every group is padded
    // to an even number of lanes (an all-zero lane contributes +0, the padding of the canonical tree).  Returns 0, 1
    // (even only: the padded groups do not fit) or a negative status.
    auto pack = [&](const bool even, fpc::DevBuf& dst_w, fpc::DevBuf& dst_meta, int (&gate_qp)[3], int& nleaves,
                    int& lds_greedy, int& lds_cycles) -> int {
        // canonical leaf = 2 consecutive blocks; a mat-vec lane owns 2 consecutive leaves (4 blocks);
        // the lanes of one row group are consecutive lanes of ONE 16-lane DPP row
        constexpr int NROWS16 = NMAT / 16;
        std::vector<int> order(NROWGRP);
        for (int g = 0; g < NROWGRP; ++g) order[g] = g;
        auto lanes_of = [&](int g) {  // an empty row group still owns one (all-zero) lane
            int n = ((int)grps[g].cols.size() + 3) / 4;
            n = n > 0 ? n : 1;
            return even ? (n + 1) & ~1 : n;  // (k_decode2: lanes 2k, 2k+1 of a group add their sums before publishing)
        };
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            const int la = lanes_of(a), lb = lanes_of(b);
            return la != lb ? la > lb : a < b;
        });
        int row_fill[NROWS16] = {0};
        int gate_q[3] = {1, 1, 1};
        nleaves = 0;
        std::vector<int> first_lane(NROWGRP, 0);  // placement: first lane of each row group
        // a lane may hold its two leaves in either order (block slots 0,1 <-> 2,3): their sums are added in the lane,
        // and a + b == b + a exactly, so the order is free for the placement search below
        std::vector<char> leaf_flip(NROWGRP * 16, 0);
        for (int oi = 0; oi < NROWGRP; ++oi) {
            const int g = order[oi], Q = lanes_of(g);
            int best = -1;
            for (int rw = 0; rw < NROWS16; ++rw)  // least-filled 16-lane row that still has room
                if (row_fill[rw] + Q <= 16 && (best < 0 || row_fill[rw] < row_fill[best])) best = rw;
            if (even && (Q > 16 || best < 0)) return 1;  // no room for the padded groups: the handle decodes one utterance per workgroup
            if (Q > 16 || best < 0) {
                fpc::set_error(
                    "fpc_lpcnet_create: recurrent matrix too dense for the register-resident layout "
                    "(%d blocks of 8x4, row group %d needs %d lanes; capacity %d blocks, 64 per row group)",
                    m->nblocks, g, Q, 4 * NMAT);
                return FPC_ERR_CAPACITY;
            }
            first_lane[g] = best * 16 + row_fill[best];
            row_fill[best] += Q;
            const int gate = g / (RNN_A / 8);
            if (Q > gate_q[gate]) gate_q[gate] = Q;
            nleaves += Q;
        }
        // ---- LDS bank conflicts of the state reads (speed only; the results do not depend on the placement).  Every
        // sample each mat-vec lane fetches the 4 inputs of each of its 4 blocks with one ds_read_b128 at float 4 * cb of
        // the state: a wave's read is served in four groups of 16 lanes, one LDS cycle per group when the lanes' addresses
        // fall into different bank quads (cb mod 16) or coincide (MI355X_MICROARCH.md, LDS).  Row groups of equal width
        // swap places while that lowers the cycle count (deterministic local search, a few ms at create).
        {
            static const int kGroupOf[64] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                             2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
            std::vector<int> lane_cb((size_t)NMAT * 4, 0);  // column block of (lane, block slot); 0 = padding reads block 0
            auto emit = [&](int g) {
                const int Q = lanes_of(g);
                for (int q = 0; q < Q; ++q)
                    for (int bb = 0; bb < 4; ++bb) {
                        const int bi = 4 * q + (leaf_flip[g * 16 + q] ? bb ^ 2 : bb);
                        lane_cb[(size_t)(first_lane[g] + q) * 4 + bb] = bi < (int)grps[g].cols.size() ? grps[g].cols[bi] : 0;
                    }
            };
            for (int g = 0; g < NROWGRP; ++g) emit(g);
            auto wave_cost = [&](int wv) {
                int cost = 0;
                for (int bb = 0; bb < 4; ++bb)
                    for (int pg = 0; pg < 4; ++pg) {
                        int seen[16][8], cnt[16] = {0}, worst = 1;
                        for (int l = 0; l < 64; ++l) {
                            if (kGroupOf[l] != pg) continue;
                            const int cb = lane_cb[(size_t)(wv * 64 + l) * 4 + bb], qd = (even ? cb + (cb >= RNN_A / 8) : cb) & 15;
                            bool dup = false;
                            for (int k = 0; k < cnt[qd]; ++k) dup = dup || seen[qd][k] == cb;
                            if (!dup && cnt[qd] < 8) seen[qd][cnt[qd]++] = cb;
                            if (cnt[qd] > worst) worst = cnt[qd];
                        }
                        cost += worst;
                    }
                return cost;
            };
            int cost[NMAT / 64], total = 0;
            for (int wv = 0; wv < NMAT / 64; ++wv) total += cost[wv] = wave_cost(wv);
            lds_greedy = total;
            // rows as ordered lists of groups; a move swaps two groups (any widths, any rows) if both rows keep <= 16 lanes
            std::vector<std::vector<int>> rows(NROWS16);
            std::vector<int> row_of(NROWGRP);
            {
                std::vector<int> by_lane(NROWGRP);
                for (int g = 0; g < NROWGRP; ++g) by_lane[g] = g;
                std::sort(by_lane.begin(), by_lane.end(), [&](int x, int y) { return first_lane[x] < first_lane[y]; });
                for (int g : by_lane) {
                    rows[first_lane[g] / 16].push_back(g);
                    row_of[g] = first_lane[g] / 16;
                }
            }
            auto fill_of = [&](int rw) {
                int f = 0;
                for (int g : rows[rw]) f += lanes_of(g);
                return f;
            };
            auto relayout = [&](int rw) {  // lanes of the row's groups in list order; lanes behind them read block 0
                for (int l = rw * 16; l < rw * 16 + 16; ++l)
                    for (int bb = 0; bb < 4; ++bb) lane_cb[(size_t)l * 4 + bb] = 0;
                int off = 0;
                for (int g : rows[rw]) {
                    first_lane[g] = rw * 16 + off;
                    off += lanes_of(g);
                    emit(g);
                }
            };
            unsigned rng = 12345u;
            auto next = [&]() { return rng = rng * 1664525u + 1013904223u, rng >> 8; };
            for (int it = 0; it < 80000; ++it) {
                const int g1 = (int)(next() % NROWGRP), g2 = (int)(next() % NROWGRP);
                if (it & 1) {  // flip the leaf order of one lane of g1
                    const int q = (int)(next() % (unsigned)lanes_of(g1)), wv = first_lane[g1] / 64;
                    leaf_flip[g1 * 16 + q] ^= 1;
                    emit(g1);
                    const int c = wave_cost(wv);
                    if (c <= cost[wv]) {
                        total += c - cost[wv];
                        cost[wv] = c;
                    } else {
                        leaf_flip[g1 * 16 + q] ^= 1;
                        emit(g1);
                    }
                    continue;
                }
                if (g1 == g2) continue;
                const int r1 = row_of[g1], r2 = row_of[g2];
                if (r1 != r2 && (fill_of(r1) - lanes_of(g1) + lanes_of(g2) > 16 || fill_of(r2) - lanes_of(g2) + lanes_of(g1) > 16))
                    continue;
                auto do_swap = [&]() {
                    auto& a1 = rows[row_of[g1]];
                    auto& a2 = rows[row_of[g2]];
                    auto i1 = std::find(a1.begin(), a1.end(), g1);
                    auto i2 = std::find(a2.begin(), a2.end(), g2);
                    std::iter_swap(i1, i2);
                    std::swap(row_of[g1], row_of[g2]);
                    relayout(r1);
                    if (r2 != r1) relayout(r2);
                };
                do_swap();
                const int w1 = r1 / 4, w2 = r2 / 4;
                const int c1 = wave_cost(w1), c2 = w2 != w1 ? wave_cost(w2) : 0;
                const int delta = c1 + c2 - cost[w1] - (w2 != w1 ? cost[w2] : 0);
                if (delta <= 0) {
                    cost[w1] = c1;
                    if (w2 != w1) cost[w2] = c2;
                    total += delta;
                } else {
                    do_swap();  // (swaps back: g1 now sits where g2 was)
                }
            }
            lds_cycles = total;  // ideal: 8 waves x 4 reads x 4 groups = 128
        }
        std::vector<float> lane_w((size_t)128 * NMAT, 0.0f);
        std::vector<int> lane_meta(2 * NMAT, 0);
        for (int l = 0; l < NMAT; ++l) lane_meta[NMAT + l] = (1 << 8);  // no group, 1 lane, lane 0
        for (int g = 0; g < NROWGRP; ++g) {
            const int Q = lanes_of(g);
            const int gate = g / (RNN_A / 8), rb = g % (RNN_A / 8);
            for (int q = 0; q < Q; ++q) {
                const int l = first_lane[g] + q;
                lane_meta[NMAT + l] = q | (Q << 8) | ((g + 1) << 16);
                unsigned colp = 0;
                for (int bb = 0; bb < 4; ++bb) {
                    const int bi = 4 * q + (leaf_flip[g * 16 + q] ? bb ^ 2 : bb);
                    if (bi >= (int)grps[g].cols.size()) continue;  // weights stay 0, column block 0
                    const int cb = grps[g].cols[bi];
                    colp |= (unsigned)(even ? cb + (cb >= RNN_A / 8) : cb) << (8 * bb);  // (k_decode2: float4 index in its padded state)
                    for (int r = 0; r < 8; ++r)
                        for (int c = 0; c < 4; ++c) {
                            const int in = cb * 4 + c, o = rb * 8 + r;
                            lane_w[(size_t)(bb * 32 + r * 4 + c) * NMAT + l] =
                                in == o ? 0.0f : w->gru_a_recurrent[(size_t)in * GA + gate * RNN_A + o];
                        }
                }
                lane_meta[l] = (int)colp;
            }
        }
        for (int g = 0; g < 3; ++g) {
            int qp = 1;
            while (qp < gate_q[g]) qp <<= 1;
            gate_qp[g] = qp;
        }
        FPC_HIP(dst_w.upload(lane_w));
        FPC_HIP(dst_meta.upload(lane_meta));
        return 0;
    };
    {
        const int rc = pack(false, m->lane_w, m->lane_meta, m->gate_qp, m->nleaves, m->lds_read_cycles_greedy,
                            m->lds_read_cycles);
        if (rc != 0) return rc;
        int qp2[3], nl2 = 0, g2 = 0, c2 = 0;
        const int rc2 = pack(true, m->lane_w2, m->lane_meta2, qp2, nl2, g2, c2);
        if (rc2 < 0) return rc2;
        m->pair_ok = rc2 == 0 && decode_variant_of(m->gate_qp) != 1616;
        m->lds_read_cycles2 = c2;
    }

    // sampler-lane weights: lane = (unit u, slice kl) for GRU_B, lane = tree node for the dual FC
    std::vector<float> lane_wb(72 * NSAMP), lane_ub(3 * NSAMP), lane_fc(36 * NSAMP);
    for (int sl = 0; sl < NSAMP; ++sl) {
        const int u = sl >> 4, kl = sl & 15;
        for (int g = 0; g < 3; ++g) {
            for (int k = 0; k < 24; ++k)
                lane_wb[(g * 24 + k) * NSAMP + sl] = w->gru_b_kernel[(size_t)(24 * kl + k) * GB + g * RNN_B + u];
            lane_ub[g * NSAMP + sl] = w->gru_b_recurrent[kl * GB + g * RNN_B + u];
        }
        for (int ch = 0; ch < 2; ++ch) {
            for (int k = 0; k < RNN_B; ++k)
                lane_fc[(ch * 16 + k) * NSAMP + sl] = w->md_kernel[((size_t)sl * RNN_B + k) * 2 + ch];
            lane_fc[(32 + ch) * NSAMP + sl] = w->md_bias[sl * 2 + ch];
            lane_fc[(34 + ch) * NSAMP + sl] = w->md_factor[sl * 2 + ch];
        }
    }
    FPC_HIP(m->lane_wb.upload(lane_wb));
    FPC_HIP(m->lane_ub.upload(lane_ub));
    FPC_HIP(m->lane_fc.upload(lane_fc));
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
    *out = own.release();
    return FPC_OK;
}

namespace {
int decode_variant(const fpc_lpcnet* m) { return decode_variant_of(m->gate_qp); }
}  // namespace

extern "C" int fpc_lpcnet_kernel_variant(const fpc_lpcnet* m) { return m ? decode_variant(m) : -1; }

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
// C = frames per pass (T, or a chunk: x1 then holds a halo frame either side and `state` the carried records)
CondBufs carve(void* ws, int B, int C, bool chunked, float** state = nullptr) {
    float* p = static_cast<float*>(ws);
    const size_t F = (size_t)B * C;
    CondBufs c;
    c.x1 = p;
    c.x2 = c.x1 + (chunked ? (size_t)B * (C + 2) : F) * 128;
    c.x3 = c.x2 + F * 128;
    c.cfa = c.x3 + F * 128;
    c.cfb = c.cfa + F * GA;
    if (state) *state = c.cfb + F * GB;
    return c;
}
FrameView whole(int T) { return FrameView{T, 0, T, 0, T, 0, T}; }

template <int MODE>
void launch_frame(const float* x, int ldx, int C, int K, const float* W, const float* bias, int N, float* y,
                  int B, const FrameView& V, int do_tanh, const float* embed_pitch, hipStream_t st) {
    const int Cs = ((C + 29) / 32) * 32 + 2;
    const dim3 grid(B * ((V.o1 - V.o0 + 15) / 16), (N + 127) / 128);
    hipLaunchKernelGGL(k_frame_mfma<MODE>, grid, dim3(256), (size_t)18 * Cs * sizeof(float), st, x, ldx, C, K, W,
                       bias, N, y, V, do_tanh, embed_pitch);
}

// frames [f0, f1) of every utterance; the buffers hold C frames per utterance from frame f0 on (x1: C + 2 from f0 - 1 on
// when the pass is a chunk -- the second conv reads a frame either side, which the first conv evaluates again), cfeat
// likewise C from f0 on
// (cf_t0, cf_T: the view of the cfeat buffer -- the chunk's own rows, or the caller's whole [B][T][128])
int run_condition(fpc_lpcnet* m, const float* feat, int B, int T, int f0, int f1, int C, const CondBufs& c, float* cfeat,
                  int cf_t0, int cf_T, hipStream_t st) {
    const float* none = nullptr;
    const bool chunked = C != T;
    const int h0 = chunked ? std::max(0, f0 - 1) : 0, h1 = chunked ? std::min(T, f1 + 1) : T;
    const int x1_t0 = chunked ? f0 - 1 : 0, x1_T = chunked ? C + 2 : T;
    const FrameView v1{T, h0, h1, 0, T, x1_t0, x1_T};    // features (whole) -> x1 with halo
    const FrameView v2{T, f0, f1, x1_t0, x1_T, f0, C};   // x1 -> x2
    const FrameView v3{T, f0, f1, f0, C, f0, C};         // x2 -> x3 -> cfeat
    // conv1 (84 ch, built on the fly from the feature frame) -> x1, conv2 -> x2, dense1 -> x3, dense2 -> cfeat
    launch_frame<2>(feat, 0, 84, 3 * 84, m->conv1_k.as<float>(), m->conv1_b.as<float>(), 128, c.x1, B, v1, 1,
                    m->embed_pitch.as<float>(), st);
    launch_frame<1>(c.x1, 128, 128, 3 * 128, m->conv2_k.as<float>(), m->conv2_b.as<float>(), 128, c.x2, B, v2, 1,
                    none, st);
    launch_frame<0>(c.x2, 128, 128, 128, m->d1_k.as<float>(), m->d1_b.as<float>(), 128, c.x3, B, v3, 1, none, st);
    const FrameView v4{T, f0, f1, f0, C, cf_t0, cf_T};
    launch_frame<0>(c.x3, 128, 128, 128, m->d2_k.as<float>(), m->d2_b.as<float>(), 128, cfeat, B, v4, 1, none, st);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}
}  // namespace

extern "C" int fpc_lpcnet_condition(fpc_lpcnet* m, const float* features_dev, int B, int T,
                                    float* cfeat_dev, void* workspace_dev, fpc_stream s) {
    FPC_REQUIRE(m && features_dev && cfeat_dev && workspace_dev, "fpc_lpcnet_condition: null argument");
    FPC_REQUIRE(B > 0 && T > 0, "fpc_lpcnet_condition: bad shape B=%d T=%d", B, T);
    const int C = pass_frames(m, T);
    const CondBufs c = carve(workspace_dev, B, C, C != T);
    for (int f0 = 0; f0 < T; f0 += C) {
        const int rc = run_condition(m, features_dev, B, T, f0, std::min(T, f0 + C), C, c, cfeat_dev, 0, T,
                                     static_cast<hipStream_t>(s));
        if (rc != FPC_OK) return rc;
    }
    return FPC_OK;
}

extern "C" int fpc_lpcnet_synthesize(fpc_lpcnet* m, const float* features_dev, int B, int T,
                                     const uint64_t* seeds_dev, int16_t* pcm_dev,
                                     void* workspace_dev, fpc_stream s) {
    FPC_REQUIRE(m && features_dev && seeds_dev && pcm_dev && workspace_dev,
                "fpc_lpcnet_synthesize: null argument");
    FPC_REQUIRE(B > 0 && T > 0 && (long long)T * FPC_FRAME_SIZE < (1ll << 31),
                "fpc_lpcnet_synthesize: bad shape B=%d T=%d", B, T);
    hipStream_t st = static_cast<hipStream_t>(s);
    // One pass over the whole utterance, or (fpc_lpcnet_set_chunk_frames) passes of C frames: the frame-rate layers and
    // the conditioning products of a chunk, then the sample loop over it with the per-stream state carried in a record --
    // the workspace holds one chunk, whatever T is.
    const int C = pass_frames(m, T);
    const bool chunked = C != T;
    float* state = nullptr;
    const CondBufs c = carve(workspace_dev, B, C, chunked, &state);
    float* cfeat = c.x1;  // x1 is free again once conv2 has run
    DecodeParams P;
    P.tab = m->tab.as<float>();
    P.cfa = c.cfa;
    P.cfb = c.cfb;
    P.cf_T = C;
    P.state = chunked ? state : nullptr;
    P.features = features_dev;
    P.seeds = reinterpret_cast<const unsigned long long*>(seeds_dev);
    P.pcm = pcm_dev;
    P.T = T;
    P.lane_w = m->lane_w.as<float>();
    P.lane_meta = m->lane_meta.as<int>();
    P.lane_wb = m->lane_wb.as<float>();
    P.lane_ub = m->lane_ub.as<float>();
    P.lane_fc = m->lane_fc.as<float>();
    P.diag = m->diag.as<float>();
    P.brn_a = m->brn_a.as<float>();
    P.brn_b = m->brn_b.as<float>();
    P.ulaw_tab = m->ulaw_tab.as<float>();
    const bool stamp = getenv("FPC_DECODE_STAMPS") != nullptr;
    if (stamp)
        fprintf(stderr, "[fpc stamps] state reads of the sparse product: %d LDS cycles per sample after placement search (%d greedy, 128 conflict-free)\n",
                m->lds_read_cycles, m->lds_read_cycles_greedy);
    fpc::DevBuf stamps;
    P.stamps = nullptr;
    constexpr int kWaves = NTHREADS / 64, kStampWords = FPC_STAMP_NS * kWaves * 8;
    if (stamp) {
        FPC_HIP(stamps.alloc(kStampWords * sizeof(unsigned)));
        FPC_HIP(hipMemsetAsync(stamps.p, 0, kStampWords * sizeof(unsigned), st));
        P.stamps = stamps.as<unsigned>();
    }
    const int variant = decode_variant(m);
    // More utterances than compute units: k_decode2 walks two utterances through each workgroup (lpcnet_decode2.h; the
    // packed partial-sum planes exist for the instances whose update / reset row groups are <= 4 lanes wide).  Same PCM.
    // The batch is decoded as its first NP utterances on k_decode2 (NP / 2 workgroups) and the rest on k_decode.  Default
    // policy (pairing 0): a round of k_decode2 (2 x CUs utterances) costs about 1.56 rounds of k_decode (CUs utterances
    // each); take the number of full-or-partial pair rounds p that minimises 1.56 p + ceil(rest / CUs) -- B <= CUs: none;
    // CUs < B <= 2 CUs: all paired; 2 CUs < B <= 3 CUs: one pair round + one round of k_decode; ...  pairing 1: all paired.
    int NP = 0;
    if (m->pair_ok && B > 1 && m->pairing > 0) NP = B;
    if (m->pair_ok && m->pairing == 0) NP = fpc_lpcnet_paired_utterances(B, m->cus);
    const bool pair = NP > 0;
    m->last_streams_per_wg = pair ? 2 : 1;
    DecodeParams P2 = P;  // k_decode2's launch: its own placement of the row groups
    P2.lane_w = m->lane_w2.as<float>();
    P2.lane_meta = m->lane_meta2.as<int>();
    // k_decode's launch over the utterances behind the paired ones: the kernel indexes by workgroup, so the bases move
    DecodeParams P1 = P;
    P1.features += (size_t)NP * T * FPC_NB_FEATURES;
    P1.seeds += NP;
    P1.pcm += (size_t)NP * T * FPC_FRAME_SIZE;
    P1.cfa += (size_t)NP * C * GA;
    P1.cfb += (size_t)NP * C * GB;
    if (P1.state) P1.state += (size_t)NP * STATE_FLOATS;
    const int B1 = B - NP;
#define FPC_LAUNCH2(QZR)                                                                                      \
    do {                                                                                                      \
        if (stamp)                                                                                            \
            hipLaunchKernelGGL((k_decode2<true, QZR>), dim3((NP + 1) / 2), dim3(NTHREADS), 0, st, P2, NP);    \
        else                                                                                                  \
            hipLaunchKernelGGL((k_decode2<false, QZR>), dim3((NP + 1) / 2), dim3(NTHREADS), 0, st, P2, NP);   \
    } while (0)
#define FPC_LAUNCH(QZR, QN)                                                                       \
    do {                                                                                          \
        if (stamp)                                                                                \
            hipLaunchKernelGGL((k_decode<true, QZR, QN>), dim3(B1), dim3(NTHREADS), 0, st, P1);   \
        else                                                                                      \
            hipLaunchKernelGGL((k_decode<false, QZR, QN>), dim3(B1), dim3(NTHREADS), 0, st, P1);  \
    } while (0)
    for (int f0 = 0; f0 < T; f0 += C) {
        const int f1 = std::min(T, f0 + C);
        // cfeat cannot alias a live buffer: run conv1->x1, conv2->x2, d1->x3, d2->x1
        const int rc = run_condition(m, features_dev, B, T, f0, f1, C, c, cfeat, f0, C, st);
        if (rc != FPC_OK) return rc;
        // conditioning products with the cfeat rows of both GRU input kernels
        const FrameView v{T, f0, f1, f0, C, f0, C};
        launch_frame<0>(cfeat, 128, 128, 128, m->ga_k.as<float>() + (size_t)3 * EMB * GA, m->bias_a.as<float>(), GA,
                        c.cfa, B, v, 0, nullptr, st);
        launch_frame<0>(cfeat, 128, 128, 128, m->gb_k.as<float>() + (size_t)RNN_A * GB, m->bias_b.as<float>(), GB,
                        c.cfb, B, v, 0, nullptr, st);
        P1.f0 = P2.f0 = f0;
        P1.f1 = P2.f1 = f1;
        if (f0 == 0) FPC_HIP(hipEventRecord(m->ev0, st));  // (chunked: the span from the first sample loop to the last)
        if (pair && variant == 208)
            FPC_LAUNCH2(2);
        else if (pair)
            FPC_LAUNCH2(4);
        if (B1 > 0) {
            if (variant == 208)
                FPC_LAUNCH(2, 8);
            else if (variant == 408)
                FPC_LAUNCH(4, 8);
            else
                FPC_LAUNCH(16, 16);
        }
    }
#undef FPC_LAUNCH
#undef FPC_LAUNCH2
    FPC_HIP(hipEventRecord(m->ev1, st));
    FPC_HIP(hipGetLastError());
    if (stamp && (long long)T * FPC_FRAME_SIZE >= FPC_STAMP_T0 + FPC_STAMP_NS) {  // diagnostic path only: synchronises
        std::vector<unsigned> h(kStampWords);
        FPC_HIP(hipStreamSynchronize(st));
        FPC_HIP(hipMemcpy(h.data(), stamps.p, kStampWords * sizeof(unsigned), hipMemcpyDeviceToHost));
        // per wave: work before barrier k = arrival(k) - release(previous barrier), wait = release(k) - arrival(k),
        // release(k) taken as the last wave's arrival; barrier order within a sample: Y(0) Z1(1) Z2(2) [Z3(3)] X(4);
        // averages over the unvoiced stamped samples
        static const char* nm[5] = {"X..Y", "Y..Z1", "Z1..Z2", "Z2..Z3", "..X"};
        auto slot = [&](int sidx, int w, int k) { return h[((size_t)sidx * kWaves + w) * 8 + k]; };
        auto release = [&](int sidx, int k) {  // last arrival (wrap-safe: relative to wave 0)
            const unsigned base = slot(sidx, 0, k);
            int best = 0;
            for (int w = 1; w < kWaves; ++w) best = std::max(best, (int)(slot(sidx, w, k) - base));
            return base + (unsigned)best;
        };
        double phase[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < kWaves; ++w) {
            double work[5] = {0, 0, 0, 0, 0}, wait[5] = {0, 0, 0, 0, 0};
            int n = 0;
            for (int sidx = 1; sidx < FPC_STAMP_NS; ++sidx) {
                if (slot(sidx, w, 3) != 0 || slot(sidx - 1, w, 4) == 0 || slot(sidx, w, 4) == 0) continue;  // voiced / not stamped
                unsigned last = release(sidx - 1, 4);
                for (int k = 0; k < 5; ++k) {
                    if (k == 3) continue;
                    const unsigned rel = release(sidx, k);
                    work[k] += (double)(int)(slot(sidx, w, k) - last);
                    wait[k] += (double)(int)(rel - slot(sidx, w, k));
                    if (w == 0) phase[k] += (double)(int)(rel - last);
                    last = rel;
                }
                ++n;
            }
            if (n == 0) continue;
            fprintf(stderr, "[fpc stamps] wave %2d (%s), cycles/sample work|wait:", w, w < 4 ? "sampler" : "mat-vec");
            for (int k = 0; k < 5; ++k)
                if (k != 3) fprintf(stderr, "  %s %.0f|%.0f", nm[k], work[k] / n, wait[k] / n);
            fprintf(stderr, "  (%d samples)\n", n);
            if (w == 0 && slot(1, 0, 5) != 0) {  // (k_decode2: two stamps inside the GRU_B phase of the sampler waves)
                double d5 = 0, d6 = 0;
                int n2 = 0;
                for (int sidx = 1; sidx < FPC_STAMP_NS; ++sidx) {
                    if (slot(sidx, w, 3) != 0 || slot(sidx, w, 5) == 0) continue;
                    const unsigned rel = release(sidx, 0);
                    d5 += (double)(int)(slot(sidx, w, 5) - rel);
                    d6 += (double)(int)(slot(sidx, w, 6) - rel);
                    ++n2;
                }
                if (n2) fprintf(stderr, "[fpc stamps] wave 0 inside Y..Z1: products done at +%.0f, butterfly done at +%.0f\n", d5 / n2, d6 / n2);
            }
            if (w == 0)
                fprintf(stderr, "[fpc stamps] phase lengths (release to release): X..Y %.0f  Y..Z1 %.0f  Z1..Z2 %.0f  ..X %.0f  sum %.0f\n",
                        phase[0] / n, phase[1] / n, phase[2] / n, phase[4] / n, (phase[0] + phase[1] + phase[2] + phase[4]) / n);
        }
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
