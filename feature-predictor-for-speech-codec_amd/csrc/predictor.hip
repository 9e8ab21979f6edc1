// predictor.hip -- GRU feature predictor, closed-loop residual encoder and the
// scalar / multi-stage M-best vector quantizers, hand-written HIP for gfx950.
//
// Reference interfaces replaced (paths under /root/reference/src):
//   Wavernn.forward        models/wavernn.py:63-102
//   Wavernn.encoder        models/wavernn.py:165-256   (mask=None)
//   vq_quantize_mbest      quantization/vq_func.py:10-24
//   quantize_mstage        quantization/vq_func.py:82-131 (S in {1,2})
//   vq_quantize            quantization/vq_func.py:134-164
//   scl_quantize           quantization/vq_func.py:167-185
//
// One workgroup per utterance runs the whole closed loop (predict -> residual ->
// threshold -> quantize -> feed back) on chip: the reference's per-frame, per-element
// device->host->NumPy->device round trip (wavernn.py:217-238) is gone.  Matrix rows are
// evaluated as k-ordered fmaf chains from the bias (what a gfx950 f32 MFMA accumulates),
// distances in float64 with numpy's pairwise association, so results match the CPU
// oracle bit for bit and the reference's indices exactly.
#include <algorithm>
#include "fpc_common.h"
#include <atomic>
#include <string>
#include <memory>
#include <cmath>

namespace {

// threads per workgroup of every kernel here: 8 wave64 = 2 per SIMD, so a lane may hold 256 registers -- the encoder's
// float64 searches stop spilling (240 registers; 168 + 45 spilled at 9 waves) and k_encode goes from 7.4 to 6.9 ms at
// 128 x 300; the teacher-forced forward loses 4 % of its streaming waves' bandwidth (3.65 -> 3.8 ms)
#ifndef FPC_NT
#define FPC_NT 512
#endif
#ifndef FPC_CD
#define FPC_CD 16
#endif
constexpr int NT = FPC_NT;
constexpr int NW = NT / 64;
constexpr int MAX_H1 = 512, MAX_H2 = 256, MAX_IN = 64, MAX_FC = 32;
constexpr int NDIM = 17, SURV = 5;
constexpr int SCLC = 512;  // scalar codes kept in LDS by the encoder (4 kB)

// ---- row split: one utterance on NSPLIT workgroups (SURVEY 2.1 "row-sliced weights-stationary across CUs") ----
// Every frame streams the whole 2.67 MB weight set through ONE CU's L2 port (25 us at ~108 GB/s), 300 frames in
// sequence: the port, not the arithmetic, sets the time.  With the utterance on two workgroups, each evaluates the
// gate rows of half of the units of both GRUs (half of the bytes through its port) and the halves of the new state
// are exchanged after each GRU as 8-byte {epoch, value} granules: one write-through (sc1) store per value, polled
// by the partner with sc1 loads until the tag matches -- no fence, no flag (cdna_hip_programming.md G16, form R2).
// The output layer, thresholds and searches run redundantly on both (same inputs, same code: same bits); half 0
// writes the outputs.  Granules are zeroed before every launch, epochs count 1.. within it.  A spin that does not
// see its tag within FPC_SPIN_LIMIT (1 s of wall clock, s_memrealtime) gives up: it sets bit 0 of the handle's
// STATUS WORD (host-mapped memory, sticky), every workgroup that notices stops waiting, the kernels store NaN / idx -2
// from that frame on and skip the Adam update, and the host turns the word into FPC_ERR_TIMEOUT (at the next call on
// the handle, at every call that synchronises anyway, and in fpc_predictor_status) -- a launch can fail, loudly, but
// never hang.  Placement-independent: nothing assumes which CUs or XCDs the workgroups land on, only that all of a
// group get dispatched (in-order dispatch, grid <= CUs: the process is assumed to own the GPU, fpcodec.h).
typedef __attribute__((address_space(1))) unsigned long long gu64;
struct SplitCtx {
    int n = 1, half = 0;            // workgroups per utterance, this workgroup's slice
    unsigned long long* g1 = nullptr;  // [H1] granules of the new GRU1 state of this utterance
    unsigned long long* g2 = nullptr;  // [H2]
    unsigned long long* g3 = nullptr;  // [H1], g4: [H2]: second set (the training backward has ONE hop per frame and
    unsigned long long* g4 = nullptr;  // alternates between the sets, so a set is rewritten only two hops later)
    unsigned* err = nullptr;        // the handle's status word (host-mapped): bit 0 spin timeout, bit 1 non-finite residual
    unsigned long long limit = 0;   // give-up bound of one spin in s_memrealtime ticks (100 MHz)
    unsigned epoch = 0;             // last epoch used
    bool dead = false;              // a spin gave up: no further waiting in this workgroup
    bool withhold = false;          // test hook (FPC_TEST_WITHHOLD_PUBLISH=1): this workgroup never publishes
    bool fast = false;              // every workgroup of the utterance reported the same XCD: plain stores (predictor_df.h)
    bool no_fast = false;           // ... unless FPC_FAST_HOP=0 forbids them
};
constexpr unsigned FPC_ST_TIMEOUT = 1u, FPC_ST_NONFINITE = 2u;
// A spin reads the clock every 64 polls (~10-100 us apart while the wave runs).  A gap between two reads beyond this bound
// means the wave itself was descheduled (queue preemption, a debugger) and says nothing about the partner: the spin's
// clock starts again.  A quarter of the give-up bound, but never below 1 ms -- with a short bound (FPC_SPIN_LIMIT_US below
// ~0.4 ms) every ordinary gap would exceed limit / 4, the clock would restart at every check and the spin never give up.
__device__ __forceinline__ unsigned long long spin_rearm_gap(unsigned long long limit) {
    const unsigned long long q = limit / 4;
    return q > 100000ull ? q : 100000ull;  // s_memrealtime ticks (100 MHz)
}
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ unsigned status_load(const unsigned* w) {
    return __hip_atomic_load((gu32*)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void status_or(unsigned* w, unsigned bits) {
    (void)__hip_atomic_fetch_or((gu32*)w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// publish this workgroup's slice of h[0..H) (h in LDS) under a new epoch ...
__device__ __forceinline__ void publish_slice(const float* h, int H, SplitCtx& X, unsigned long long* g, int tid,
                                              bool new_epoch = true) {
    const int Hs = H / X.n, mine = X.half * Hs;
    if (new_epoch) ++X.epoch;  // (false: a second array handed over in the same hop)
    const unsigned epoch = X.epoch;
    if (X.withhold) return;    // (test hook: the partners' spins must time out)
    for (int i = tid; i < Hs; i += NT) store_granule(&g[mine + i], epoch, h[mine + i]);
}
// wait for the value tagged `epoch` in granule *g (NaN if the wait is given up or the workgroup is already dead)
__device__ __forceinline__ float await_granule(unsigned long long* g, unsigned epoch, SplitCtx& X, bool& gave_up) {
    unsigned long long x = 0x7fc00000ull;  // NaN unless the partner's value arrives: poisons everything downstream
    unsigned spins = 0;
    unsigned long long t0 = 0, last = 0;
    while (!X.dead) {
        const unsigned long long v = __hip_atomic_load((gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == epoch) {
            x = v;
            break;
        }
        // every 64 polls (~10-100 us): has the partner been missing for X.limit of wall clock, or has another
        // workgroup already given up?  Then stop waiting, here and from now on -- the launch must end quickly
        // and loudly, never hang the GPU
        if ((++spins & 63u) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            // (two time reads are ~10-100 us apart while the wave runs; a much larger gap means the wave itself was
            //  descheduled -- queue preemption, a debugger -- and says nothing about the partner: the clock starts again)
            if (t0 == 0 || now - last > spin_rearm_gap(X.limit)) t0 = now;
            last = now;
            if (now - t0 > X.limit || (status_load(X.err) & FPC_ST_TIMEOUT) != 0u) {
                status_or(X.err, FPC_ST_TIMEOUT);
                gave_up = true;
                break;
            }
        }
#ifndef FPC_XCHG_NOSLEEP
        __builtin_amdgcn_s_sleep(1);
#endif
    }
    return __uint_as_float((unsigned)x);
}
// ... and pick every other slice of that epoch up (ends with a barrier); independent work may sit between the two
__device__ __forceinline__ void consume_slices(float* h, int H, SplitCtx& X, unsigned long long* g, int tid) {
    const int Hs = H / X.n, mine = X.half * Hs;  // this workgroup's slice of the units; every other slice is read
    const unsigned epoch = X.epoch;
    bool gave_up = false;
    for (int ii = tid; ii < H - Hs; ii += NT) {
        const int i = ii < mine ? ii : ii + Hs;
        h[i] = await_granule(&g[i], epoch, X, gave_up);
    }
    if (__syncthreads_or(gave_up)) X.dead = true;
}
__device__ __forceinline__ void exchange_halves(float* h, int H, SplitCtx& X, unsigned long long* g, int tid) {
    publish_slice(h, H, X, g, tid);
    consume_slices(h, H, X, g, tid);
}

struct PredDev {
    int in, h1, h2, fc;
    const float *w1i, *w1h, *b1i, *b1h;  // transposed: [K][3H]
    const float *w2i, *w2h, *b2i, *b2h;
    const float *fcw, *fcb;  // [H2][fc]
};

struct CbDev {
    int S_hi, N_hi0, N_hi1, N_lo, n_hi, n_lo;
    const double *vq_hi0, *vq_hi1, *vq_lo;     // transposed [17][N]
    const double *vq_hi0_r, *vq_hi1_r, *vq_lo_r;  // row-major [N][17] (entry fetch)
    const double *scl_hi, *scl_lo;
};

// workgroup barrier for data that changes hands through LDS only: __syncthreads also waits for the wave's outstanding
// global stores (vmcnt), a round trip the frame tail does not need
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// scratch of the residual searches (one utterance at a time)
struct __attribute__((aligned(16))) SearchLds {
    float rs[MAX_FC];           // residual of the frame: rs[0] scalar, rs[1..17] the VQ target
    double xq[SURV][NDIM];      // search targets (stage 1: xq[0]; stage 2: 5 residuals)
    double wd[SURV][NW][SURV];  // per-wave M-best lists
    int wi[SURV][NW][SURV];
    double od[SURV][SURV];      // merged M-best lists
    int oi[SURV][SURV];
    double sd[NW];              // scalar arg-min scratch
    int si[NW];
    int res_i[4];
    double qv[NDIM];
    double qs;
    double sclc[SCLC];  // the scalar codebooks (above-threshold codes, then the below-threshold ones) when they fit: k_encode_df
#ifdef FPC_VQ_PROF
    long long prof[16];  // diagnostic builds only: cycle stamps of the search phases
#endif
};
#ifdef FPC_VQ_PROF
#define VQ_STAMP(k) \
    if (tid == 0) L.prof[k] = clock64();
#else
#define VQ_STAMP(k)
#endif
// four adjacent rows (r..r+3, R % 4 == 0) advance together: one 16-byte load per k serves 4 chains (dword loads
// would be bound by the texture-address unit, not by L2).  The loads run as a ROLLING WINDOW of CD k-steps: a
// register is refilled with k + CD as soon as k has been used, so CD - 1 loads stay in flight for the whole chain
// (a block that loads CD, waits for all and then computes pays one L2 round trip per block: 6 per 96-long chain,
// measured 1 400-1 750 cycles each, and the CU's L2 port idles meanwhile).  Same k order: same bits.
// Round 4: PLAIN loads, which the compiler counts itself, and scheduling group barriers that ask for the interleaving
// (the fmaf's of the oldest window register, then its refill).  Rounds 2-3 wrote the loads and their waits as inline
// assembly (global_load_dwordx4 + s_waitcnt vmcnt(N) tied to the register): faster at 576 threads, but the compiler
// does not count such loads, and a window register that it copied between its load and its wait gave wrong results
// when the surrounding code changed shape (profiles/r03_predictor_two_roles.txt).  The kernels that still used that
// form -- the phase kernels (the tests' reference form) and the training step -- are correct by construction now.
constexpr int CD = FPC_CD;  // k-steps of 16-byte loads in flight per thread (multiple of 4)
__device__ __forceinline__ void fma4(float4& a, float hv, const float4& w) {
    a.x = fmaf(hv, w.x, a.x);
    a.y = fmaf(hv, w.y, a.y);
    a.z = fmaf(hv, w.z, a.z);
    a.w = fmaf(hv, w.w, a.w);
}
// a: the chain's start value (bias or 0) on entry, the chain's sum on exit
__device__ __forceinline__ void chain4(const float* __restrict__ wT, const float* v, int K, int R, int r, float4& a) {
    const float* q = wT + r;  // ONE running address, advanced by a row per load (the loads go out in k order)
    const int nb = K / CD, rem = K - nb * CD;
    float4 w[CD];
    float hv[CD];
    if (nb > 0) {
#pragma unroll
        for (int j = 0; j < CD; ++j, q += R) w[j] = *reinterpret_cast<const float4*>(q);
    }
    for (int b = 0; b + 1 < nb; ++b, v += CD) {
#pragma unroll
        for (int j = 0; j < CD; ++j) hv[j] = v[j];
#pragma unroll
        for (int j = 0; j < CD; ++j, q += R) {
            fma4(a, hv[j], w[j]);
            w[j] = *reinterpret_cast<const float4*>(q);
        }
#pragma unroll
        for (int j = 0; j < CD; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // the step's arithmetic (4 fmaf + the address)
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one load
        }
    }
    if (nb > 0) {
#pragma unroll
        for (int j = 0; j < CD; ++j) hv[j] = v[j];
#pragma unroll
        for (int j = 0; j < CD; ++j) fma4(a, hv[j], w[j]);
        v += CD;
    }
    for (int k = 0; k < rem; ++k, q += R) fma4(a, v[k], *reinterpret_cast<const float4*>(q));
}

// Rows are evaluated in S contiguous segments of the input, one thread per (4 adjacent rows, segment): segment 0
// is a k-ordered fmaf chain from the bias, the others start from 0, and the segment sums are added as a
// balanced tree (oracle: fpc_segments / matvec_seg).  S depends on the row length only.
__device__ __forceinline__ int segments(int K) {
    const int S = K >= 256 ? 4 : (K >= 64 ? 2 : 1);
    return K % S == 0 ? S : 1;
}
// ---- float64 squared distance with numpy's pairwise association (vq_func.py:18) ----
__device__ __forceinline__ double dist17(const double* x, const double* __restrict__ cbT, int N, int e) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double d = x[j] - cbT[(size_t)j * N + e];
        r[j] = d * d;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double d = x[8 + j] - cbT[(size_t)(8 + j) * N + e];
        const double dd = d * d;
        r[j] = r[j] + dd;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const double d = x[16] - cbT[(size_t)16 * N + e];
    const double dd = d * d;
    res = res + dd;
    return res;
}

// (distance, index) minimum over the wave, ties to the lower index; every lane gets the result.  The first four
// exchange steps stay inside the 16-lane rows (DPP: no LDS crossbar), only the last two cross rows.  The result
// does not depend on the pairing order (a minimum of a set).
template <int CTRL>
__device__ __forceinline__ void dpp_pair(double d, int i, double& od, int& oi) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    od = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ void take_min(double& d, int& i, double od, int oi) {
    // branch-free (three compares, two scalar mask ops, three selects): no exec-mask juggling per exchange step
    const bool take = (od < d) | ((od == d) & (oi < i));
    d = take ? od : d;
    i = take ? oi : i;
}
template <int CTRL, int ROWMASK, bool BOUND>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)b, (int)(unsigned)b, CTRL, ROWMASK, 0xf, BOUND);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(b >> 32), (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xf, BOUND);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ double min_f64(double a, double b) {  // distances are never NaN: plain v_min_f64
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (distance, index) minimum over the wave, ties to the lower index; every lane gets the result.  Two passes of
// cheap exchanges instead of one pass of compound compares: the smallest distance (v_min_f64; four steps inside the
// 16-lane rows, two row broadcasts, lane 63 holds the wave's), then the smallest index among the lanes that hold
// exactly that distance (v_min_i32, DPP-fused).  The result does not depend on the pairing order (a minimum of a set).
__device__ __forceinline__ void wave_argmin(double& d, int& i) {
    double m = d;
    m = min_f64(m, dpp_mov_f64<0xB1, 0xf, false>(m));   // quad_perm [1,0,3,2]
    m = min_f64(m, dpp_mov_f64<0x4E, 0xf, false>(m));   // quad_perm [2,3,0,1]
    m = min_f64(m, dpp_mov_f64<0x141, 0xf, false>(m));  // row_half_mirror
    m = min_f64(m, dpp_mov_f64<0x140, 0xf, false>(m));  // row_mirror
    m = min_f64(m, dpp_mov_f64<0x142, 0xa, false>(m));  // row_bcast:15 -> rows 1,3
    m = min_f64(m, dpp_mov_f64<0x143, 0xc, false>(m));  // row_bcast:31 -> rows 2,3
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    const double dmin = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    int c = d == dmin ? i : 0x7fffffff;
    c = min(c, dpp_mov_i32<0xB1, 0xf>(c));
    c = min(c, dpp_mov_i32<0x4E, 0xf>(c));
    c = min(c, dpp_mov_i32<0x141, 0xf>(c));
    c = min(c, dpp_mov_i32<0x140, 0xf>(c));
    c = min(c, dpp_mov_i32<0x142, 0xa>(c));
    c = min(c, dpp_mov_i32<0x143, 0xc>(c));
    d = dmin;
    i = __builtin_amdgcn_readlane(c, 63);
}

// per-wave M-best (5 smallest by (distance, index)) of search `srch` -> L.wd/L.wi
__device__ void wave_mbest(SearchLds& L, int srch, const double* __restrict__ cbT, int N, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    double d5[SURV];
    int i5[SURV];
#pragma unroll
    for (int m = 0; m < SURV; ++m) {
        d5[m] = INFINITY;
        i5[m] = 0x7fffffff;
    }
    for (int e = tid; e < N; e += NT) {  // ascending e: strict < keeps the lower index first
        double d = dist17(L.xq[srch], cbT, N, e);
        int ix = e;
#pragma unroll
        for (int m = 0; m < SURV; ++m) {
            if (d < d5[m]) {
                const double td = d5[m];
                const int ti = i5[m];
                d5[m] = d;
                i5[m] = ix;
                d = td;
                ix = ti;
            }
        }
    }
#pragma unroll 1
    for (int rnd = 0; rnd < SURV; ++rnd) {
        double d = d5[0];
        int ix = i5[0];
        wave_argmin(d, ix);
        if (i5[0] == ix && ix != 0x7fffffff) {  // this lane's head won: pop it
#pragma unroll
            for (int m = 0; m < SURV - 1; ++m) {
                d5[m] = d5[m + 1];
                i5[m] = i5[m + 1];
            }
            d5[SURV - 1] = INFINITY;
            i5[SURV - 1] = 0x7fffffff;
        }
        if (lane == 0) {
            L.wd[srch][wave][rnd] = d;
            L.wi[srch][wave][rnd] = ix;
        }
    }
}

// merge the per-wave lists of search `srch` (executed by ONE wave) -> L.od/L.oi
__device__ void merge_mbest(SearchLds& L, int srch, int lane) {
    double d = INFINITY;
    int ix = 0x7fffffff;
    if (lane < NW * SURV) {
        d = L.wd[srch][lane / SURV][lane % SURV];
        ix = L.wi[srch][lane / SURV][lane % SURV];
    }
#pragma unroll 1
    for (int rnd = 0; rnd < SURV; ++rnd) {
        double md = d;
        int mi = ix;
        wave_argmin(md, mi);
        if (ix == mi) {
            d = INFINITY;
            ix = 0x7fffffff;
        }
        if (lane == 0) {
            L.od[srch][rnd] = md;
            L.oi[srch][rnd] = mi;
        }
    }
}

// Running (distance, index) minimum of T targets over this thread's entries (e = tid, tid + NT, ...: ascending, so
// strict < keeps the lower index), the entry's coordinates loaded once for all targets; then the minimum of the
// workgroup: per-wave exchange, one LDS hop, NW candidates per target.  Only the BEST entry per target is needed
// where the reference's M-best list is never read past its head: in 1-stage searches (vq_func.py:93-95) and in
// stage 2, whose merge-insert (:110-125) only ever returns the path at position 0 = the smallest total error,
// earlier survivors winning ties (strict <).
template <int T>
__device__ __forceinline__ void block_argmin(SearchLds& L, const double* __restrict__ cbT, int N, int tid) {
    // N <= 2 * NT: a thread owns entries tid and tid + NT; their coordinates stay in registers across the targets
    const int wave = tid >> 6, lane = tid & 63;
    const int e0 = tid, e1 = tid + NT;
    const bool has0 = e0 < N, has1 = e1 < N;
    double c0[NDIM], c1[NDIM];
#pragma unroll
    for (int j = 0; j < NDIM; ++j) {
        c0[j] = has0 ? cbT[(size_t)j * N + e0] : 0.0;
        c1[j] = has1 ? cbT[(size_t)j * N + e1] : 0.0;
    }
    auto dist = [](const double* x, const double* c) {
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
    };
#pragma unroll 1
    for (int k = 0; k < T; ++k) {  // (not unrolled: one target's 17 coordinates live at a time, one exchange network)
        double x[NDIM];
#pragma unroll
        for (int j = 0; j < NDIM; ++j) x[j] = L.xq[k][j];
        double d = INFINITY;
        int ix = 0x7fffffff;
        if (has0) {
            d = dist(x, c0);
            ix = e0;
        }
        if (has1) {  // ascending entries: strict < keeps the lower index
            const double d1 = dist(x, c1);
            if (d1 < d) {
                d = d1;
                ix = e1;
            }
        }
        wave_argmin(d, ix);
        if (lane == 0) {
            L.wd[k][wave][0] = d;
            L.wi[k][wave][0] = ix;
        }
    }
    __syncthreads();
    if (tid < T) {
        double d = L.wd[tid][0][0];
        int ix = L.wi[tid][0][0];
        for (int w = 1; w < NW; ++w) take_min(d, ix, L.wd[tid][w][0], L.wi[tid][w][0]);
        L.od[tid][0] = d;
        L.oi[tid][0] = ix;
    }
}

// stage-1 M-best of a stage of <= 2 * NT entries: a thread owns entries tid and tid + NT (coordinates loaded
// together), its sorted pair feeds the per-wave rounds -> L.wd/L.wi[0]; merge_mbest(L, 0, .) finishes
__device__ __forceinline__ void wave_mbest2(SearchLds& L, const double* __restrict__ cbT, int N, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const int e0 = tid, e1 = tid + NT;
    const bool has0 = e0 < N, has1 = e1 < N;
    double c0[NDIM], c1[NDIM], x[NDIM];
#pragma unroll
    for (int j = 0; j < NDIM; ++j) {
        c0[j] = has0 ? cbT[(size_t)j * N + e0] : 0.0;
        c1[j] = has1 ? cbT[(size_t)j * N + e1] : 0.0;
        x[j] = L.xq[0][j];
    }
    auto dist = [](const double* xx, const double* c) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double d = xx[j] - c[j];
            r[j] = d * d;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double d = xx[8 + j] - c[8 + j];
            const double dd = d * d;
            r[j] = r[j] + dd;
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        const double d = xx[16] - c[16];
        const double dd = d * d;
        res = res + dd;
        return res;
    };
    double da = has0 ? dist(x, c0) : INFINITY, db = has1 ? dist(x, c1) : INFINITY;
    int ia = has0 ? e0 : 0x7fffffff, ib = has1 ? e1 : 0x7fffffff;
    if (db < da) {  // sorted pair; equal distances keep the lower index (e0 < e1) first
        const double t = da;
        da = db;
        db = t;
        const int u = ia;
        ia = ib;
        ib = u;
    }
#pragma unroll 1
    for (int rnd = 0; rnd < SURV; ++rnd) {
        double d = da;
        int ix = ia;
        wave_argmin(d, ix);
        if (ia == ix && ix != 0x7fffffff) {  // this lane's head won: pop it
            da = db;
            ia = ib;
            db = INFINITY;
            ib = 0x7fffffff;
        }
        if (lane == 0) {
            L.wd[0][wave][rnd] = d;
            L.wi[0][wave][rnd] = ix;
        }
    }
}

// quantize_mstage (vq_func.py:82-131) on L.rs[1..17]; result in L.qv, L.res_i[0..1].
// Block-uniform control flow; ends with a barrier.
__device__ __forceinline__ void vq_mstage(SearchLds& L, int S, const double* cb0T, const double* cb0R, int N0,
                          const double* cb1T, const double* cb1R, int N1, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    VQ_STAMP(0)
    if (tid < NDIM) L.xq[0][tid] = (double)L.rs[1 + tid];
    __syncthreads();
    if (S == 1 && N0 <= 2 * NT) {  // the nearest entry is all a 1-stage search returns
        block_argmin<1>(L, cb0T, N0, tid);
    } else if (N0 <= 2 * NT) {  // per-wave M-best of the threads' entry pairs, merged by one wave
        wave_mbest2(L, cb0T, N0, tid);
        __syncthreads();
        VQ_STAMP(1)
        if (wave == 0) merge_mbest(L, 0, lane);
    } else {
        wave_mbest(L, 0, cb0T, N0, tid);
        __syncthreads();
        if (wave == 0) merge_mbest(L, 0, lane);
    }
    __syncthreads();
    if (S == 1) {
        if (tid < NDIM) L.qv[tid] = cb0R[(size_t)L.oi[0][0] * NDIM + tid];
        if (tid == 0) {
            L.res_i[0] = L.oi[0][0];
            L.res_i[1] = -1;
        }
        __syncthreads();
        return;
    }
    VQ_STAMP(2)
    // stage 2: residual of every survivor, searched concurrently
    int s1[SURV];
#pragma unroll
    for (int k = 0; k < SURV; ++k) s1[k] = L.oi[0][k];
    __syncthreads();  // everyone has read stage-1 results before od/oi are reused
    if (tid < SURV * NDIM) {
        const int k = tid / NDIM, d = tid % NDIM;
        L.xq[k][d] = (double)L.rs[1 + d] - cb0R[(size_t)s1[k] * NDIM + d];
    }
    __syncthreads();
    VQ_STAMP(3)
    if (N1 <= 2 * NT) {
        block_argmin<SURV>(L, cb1T, N1, tid);  // the five searches share one pass over the entries
    } else {  // larger stages: strided M-best search per survivor (only the heads are used below)
        for (int k = 0; k < SURV; ++k) wave_mbest(L, k, cb1T, N1, tid);
        __syncthreads();
        if (wave < SURV) merge_mbest(L, wave, lane);
    }
    __syncthreads();
    VQ_STAMP(5)
    if (tid == 0) {
        // head of the merge-insert of candidate paths (vq_func.py:110-125): survivor k's best stage-2 entry replaces
        // the running best only if its total error is strictly smaller (earlier survivors win ties)
        int bk = 0;
        double g = L.od[0][0];
        for (int k = 1; k < SURV; ++k)
            if (L.od[k][0] < g) {
                g = L.od[k][0];
                bk = k;
            }
        L.res_i[0] = s1[bk];
        L.res_i[1] = L.oi[bk][0];
    }
    __syncthreads();
    VQ_STAMP(6)
    if (tid < NDIM)
        L.qv[tid] = cb0R[(size_t)L.res_i[0] * NDIM + tid] + cb1R[(size_t)L.res_i[1] * NDIM + tid];
    __syncthreads();
    VQ_STAMP(7)
}

// scl_quantize (vq_func.py:167-185): first arg-min of (x-c)^2 in float64 -> L.qs, L.res_i[2]
__device__ void scl_search(SearchLds& L, float xv, const double* __restrict__ codes, int n, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    double bd = INFINITY;
    int bi = 0x7fffffff;
    const double v = (double)xv;
    if (n <= 256) {  // small codebooks: one wave, no cross-wave round
        if (wave == 0) {
            for (int c = lane; c < n; c += 64) {
                const double df = v - codes[c];
                const double d = df * df;
                if (d < bd) {
                    bd = d;
                    bi = c;
                }
            }
            wave_argmin(bd, bi);
            if (lane == 0) {
                L.res_i[2] = bi;
                L.qs = codes[bi];
            }
        }
        __syncthreads();
        return;
    }
    for (int c = tid; c < n; c += NT) {
        const double df = v - codes[c];
        const double d = df * df;
        if (d < bd) {
            bd = d;
            bi = c;
        }
    }
    wave_argmin(bd, bi);
    if (lane == 0) {
        L.sd[wave] = bd;
        L.si[wave] = bi;
    }
    __syncthreads();
    if (wave == 0) {
        double d = lane < NW ? L.sd[lane] : INFINITY;
        int ix = lane < NW ? L.si[lane] : 0x7fffffff;
        wave_argmin(d, ix);
        if (lane == 0) {
            L.res_i[2] = ix;
            L.qs = codes[ix];
        }
    }
    __syncthreads();
}

// scl_search with the codes in LDS (L.sclc[off .. off + n)): the same scan, the same arg-min, no L2 round trips -- the
// global form costs the frame two of them (the scan's loads, then the winner's code), ~3.3k cycles on every frame
__device__ __forceinline__ void scl_search_lds(SearchLds& L, float xv, int off, int n, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    double bd = INFINITY;
    int bi = 0x7fffffff;
    const double v = (double)xv;
    if (n <= 256) {
        if (wave == 0) {
            for (int c = lane; c < n; c += 64) {
                const double df = v - L.sclc[off + c];
                const double d = df * df;
                if (d < bd) {
                    bd = d;
                    bi = c;
                }
            }
            wave_argmin(bd, bi);
            if (lane == 0) {
                L.res_i[2] = bi;
                L.qs = L.sclc[off + bi];
            }
        }
        lds_barrier();
        return;
    }
    for (int c = tid; c < n; c += NT) {
        const double df = v - L.sclc[off + c];
        const double d = df * df;
        if (d < bd) {
            bd = d;
            bi = c;
        }
    }
    wave_argmin(bd, bi);
    if (lane == 0) {
        L.sd[wave] = bd;
        L.si[wave] = bi;
    }
    lds_barrier();
    if (wave == 0) {
        double d = lane < NW ? L.sd[lane] : INFINITY;
        int ix = lane < NW ? L.si[lane] : 0x7fffffff;
        wave_argmin(d, ix);
        if (lane == 0) {
            L.res_i[2] = ix;
            L.qs = L.sclc[off + ix];
        }
    }
    lds_barrier();
}

// ---------------------------------------------------------------------------------
struct SplitArgs {
    int n;                    // workgroups per utterance (1, 2, 4 or 8)
    unsigned long long* g;    // [B][2][h1 + h2] exchange granules, zeroed before the launch (n > 1)
    unsigned* err;            // the handle's status word (host-mapped, sticky; always valid)
    unsigned long long limit; // give-up bound of one spin, s_memrealtime ticks
    int withhold;             // test hook: the last slice of utterance 0 never publishes
    int no_fast;              // FPC_FAST_HOP=0: the write-through exchange even when every slice sits on one XCD (tests run both)
    const unsigned* only;     // the weights-stationary launch's fallback (predictor_ws.h, ws_hello): run only the utterances
                              // whose group of 16 decided WS_FALLBACK there ([groups] decision words), nullptr: all
};
constexpr unsigned WS_GO = 1u, WS_FALLBACK = 2u, WS_DEAD = 3u;  // (WS_DEAD: a workgroup of the group gave up -- k_forward_ws for k_out_layer)
// (workgroup-uniform) is utterance b left to this launch?
__device__ __forceinline__ bool split_wanted(const SplitArgs& S, int b) {
    return S.only == nullptr || __hip_atomic_load((gu32*)(S.only + b / 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == WS_FALLBACK;
}
__device__ __forceinline__ SplitCtx split_ctx(const SplitArgs& S, const PredDev& P, int b, int half) {
    SplitCtx X;
    X.n = S.n;
    X.half = half;
    X.err = S.err;
    X.limit = S.limit;
    X.withhold = S.withhold != 0 && b == 0 && half == S.n - 1 && S.n > 1;
    X.no_fast = S.no_fast != 0;
    // a handle whose status word is already set (an earlier launch failed and the host has not cleared it yet)
    // does not wait for anybody: its outputs are poison anyway
    X.dead = S.n > 1 && (status_load(S.err) & FPC_ST_TIMEOUT) != 0u;
    if (S.n > 1) {
        X.g1 = S.g + (size_t)b * 2 * (P.h1 + P.h2);
        X.g2 = X.g1 + P.h1;
        X.g3 = X.g2 + P.h2;
        X.g4 = X.g3 + P.h1;
    }
    return X;
}

struct EncArgs {
    const float* feat;
    int Lf;
    float l1, l2;
    int qtz;
    float *c_in, *r, *r_qtz, *r_under, *ind1, *ind2;
    int* idx;
    unsigned long long* hist;
    const float* mask;  // [B][L][2] indicators handed in (wavernn.py:209-211), or nullptr: the thresholds decide (:201-207)
};

// The frame's tail of Wavernn.encoder for ONE utterance (wavernn.py:196-252), run by the whole workgroup: residual of the
// prediction fo[0..F), thresholds, searches, outputs of frame fi (stored when `store`), next input -> xn[0..Cc).
// Ends with a barrier.
// fv: this thread's column of the frame's feature row (tid < Cc), fetched by the caller before the predictor step so
// that its latency is off the closed loop.
__device__ __forceinline__ void encode_frame(SearchLds& L, const float* fo, float* xn, const PredDev& P, const CbDev& C,
                                             const EncArgs& A, unsigned* err, size_t fi, float fv, bool store, int tid,
                                             bool scl_in_lds = false) {
    const int Cc = P.in, F = P.fc;
    const int off_sl = C.n_hi, off_v0 = off_sl + C.n_lo, off_v1 = off_v0 + C.N_hi0,
              off_vl = off_v1 + (C.S_hi == 2 ? C.N_hi1 : 0);
    // the input-mask mode (:209-211): the caller's indicators replace the thresholds' (a wave-uniform load, mask mode only)
    const bool masked = A.mask != nullptr;
    const float m1 = masked ? A.mask[fi * 2] : 0.0f, m2 = masked ? A.mask[fi * 2 + 1] : 0.0f;
    if (tid < F) L.rs[tid] = fv - fo[tid];  // :196
    __syncthreads();
    float sabs = 0.0f;
    for (int d = 1; d < F; ++d) sabs += fabsf(L.rs[d]);
    const int i1 = masked ? (m1 != 0.0f) : (fabsf(L.rs[0]) > A.l1);  // :202 | :210 (`if ind1[k, 0]`: any non-zero value)
    const int i2 = masked ? (m2 != 0.0f) : (sabs > A.l2);            // :206 | :211
    // a NaN / infinite residual (non-finite features or weights) has no nearest entry: the arg-min would come back
    // as 0x7fffffff and be used as an address.  Such a frame is not searched: status bit 1, symbols -2
    // (every thread evaluates the same LDS values: workgroup-uniform without a barrier)
    const bool nonfinite = !(fabsf(L.rs[0]) <= 3.0e38f) || !(sabs <= 3.0e38f);
    if (nonfinite && A.qtz && tid == 0 && store) status_or(err, FPC_ST_NONFINITE);
    float rq = 0.0f;                       // this thread's r_qtz[d] (tid < F)
    int ix0 = -1, ix1 = -1, ix2 = -1, ix3 = -1;
    if (nonfinite) ix0 = ix1 = ix2 = ix3 = -2;
    if (A.qtz && !nonfinite) {
        if (i1 || C.scl_lo) {  // :218-225
            if (scl_in_lds)
                scl_search_lds(L, L.rs[0], i1 ? 0 : C.n_hi, i1 ? C.n_hi : C.n_lo, tid);
            else
                scl_search(L, L.rs[0], i1 ? C.scl_hi : C.scl_lo, i1 ? C.n_hi : C.n_lo, tid);
            if (tid == 0) {
                rq = (float)L.qs;
                ix0 = L.res_i[2] + (i1 ? 0 : C.n_hi);
                if (A.hist && store) atomicAdd(&A.hist[(i1 ? 0 : off_sl) + L.res_i[2]], 1ull);
            }
        }
        if (i2 || C.vq_lo) {  // :229-240: above the threshold the 1- or 2-stage book, below it the 1-stage one
            // (one call site: the search is one copy of code, inlined with global pointers)
            vq_mstage(L, i2 ? C.S_hi : 1, i2 ? C.vq_hi0 : C.vq_lo, i2 ? C.vq_hi0_r : C.vq_lo_r,
                      i2 ? C.N_hi0 : C.N_lo, i2 ? C.vq_hi1 : nullptr, i2 ? C.vq_hi1_r : nullptr,
                      i2 ? C.N_hi1 : 0, tid);
            if (tid >= 1 && tid < F) rq = (float)L.qv[tid - 1];
            if (tid == 0) {
                if (i2) {
                    ix1 = L.res_i[0];
                    ix2 = L.res_i[1];
                    if (A.hist && store) {
                        atomicAdd(&A.hist[off_v0 + ix1], 1ull);
                        if (C.S_hi == 2) atomicAdd(&A.hist[off_v1 + ix2], 1ull);
                    }
                } else {
                    ix3 = L.res_i[0];
                    if (A.hist && store) atomicAdd(&A.hist[off_vl + ix3], 1ull);
                }
            }
        }
    }
    if (tid < F) {
        const float rs = L.rs[tid];
        const int ind = tid == 0 ? i1 : i2;
        float rv, ru, cn;
        if (A.qtz) {
            rv = rs;  // un-thresholded residual (:197)
            ru = 0.0f;
            cn = fo[tid] + rq;  // :242
        } else {                  // :244-252 (mask mode: the products with the mask's own values)
            const float mv = tid == 0 ? m1 : m2;
            ru = rs * (masked ? 1.0f - mv : (float)(1 - ind));
            rv = rs * (masked ? mv : (float)ind);
            cn = fo[tid] + rv;
        }
        if (store) {
            A.r[fi * F + tid] = rv;
            A.r_qtz[fi * F + tid] = rq;
            A.r_under[fi * F + tid] = ru;
            A.c_in[fi * Cc + tid] = cn;
        }
        xn[tid] = cn;
    } else if (tid < Cc) {  // pitch columns pass through (:178)
        if (store) A.c_in[fi * Cc + tid] = fv;
        xn[tid] = fv;
    }
    if (tid == 0 && store) {
        A.ind1[fi] = masked ? 0.0f : (float)i1;  // (the reference fills the indicator outputs from the thresholds only)
        A.ind2[fi] = masked ? 0.0f : (float)i2;
        if (A.idx) {
            A.idx[fi * 4 + 0] = ix0;
            A.idx[fi * 4 + 1] = ix1;
            A.idx[fi * 4 + 2] = ix2;
            A.idx[fi * 4 + 3] = ix3;
        }
    }
    lds_barrier();  // (the next input row is in LDS; the frame's global stores need not have landed)
}
// a launch that gave up at frame i0 fails loudly: NaN and symbols -2 from that frame on for utterance b (the histograms
// are not touched any more); the host reports FPC_ERR_TIMEOUT
__device__ __forceinline__ void encode_poison(const PredDev& P, const EncArgs& A, int b, int i0, int tid) {
    const int Cc = P.in, F = P.fc;
    const float qnan = __uint_as_float(0x7fc00000u);
    for (size_t k = (size_t)i0 * F + tid; k < (size_t)A.Lf * F; k += NT) {
        const size_t o = (size_t)b * A.Lf * F + k;
        A.r[o] = qnan;
        A.r_qtz[o] = qnan;
        A.r_under[o] = qnan;
    }
    for (size_t k = (size_t)i0 * Cc + tid; k < (size_t)A.Lf * Cc; k += NT) A.c_in[(size_t)b * A.Lf * Cc + k] = qnan;
    for (size_t k = (size_t)i0 + tid; k < (size_t)A.Lf; k += NT) {
        const size_t o = (size_t)b * A.Lf + k;
        A.ind1[o] = qnan;
        A.ind2[o] = qnan;
        if (A.idx)
            for (int c = 0; c < 4; ++c) A.idx[o * 4 + c] = -2;
    }
}

// the receiver's frame tail for one utterance: threads (c < Cc) of one wave (k_decode_feat)
__device__ __forceinline__ void decode_frame(const float* fo, float* xn, const PredDev& P, const CbDev& C,
                                             const float* __restrict__ pitch, const int* __restrict__ idx,
                                             float* __restrict__ c_out, int* bad, size_t fi, bool store, int c) {
    const int Cc = P.in, F = P.fc;
    if (c < F) {
        const int* ix = idx + fi * 4;
        float rq = 0.0f;
        if (c == 0) {
            const int k = ix[0];
            if (k >= 0) {
                if (k < C.n_hi)
                    rq = (float)C.scl_hi[k];
                else if (C.scl_lo && k - C.n_hi < C.n_lo)
                    rq = (float)C.scl_lo[k - C.n_hi];
                else if (store)
                    atomicOr(bad, 1);
            }
        } else {
            const int d = c - 1, k1 = ix[1], k2 = ix[2], k3 = ix[3];
            if (k1 >= 0) {
                if (k1 >= C.N_hi0 || (C.S_hi == 2 && (k2 < 0 || k2 >= C.N_hi1))) {
                    if (store) atomicOr(bad, 1);
                } else {
                    const double e0 = C.vq_hi0_r[(size_t)k1 * NDIM + d];
                    rq = (float)(C.S_hi == 2 ? e0 + C.vq_hi1_r[(size_t)k2 * NDIM + d] : e0);
                }
            } else if (k3 >= 0) {
                if (!C.vq_lo_r || k3 >= C.N_lo) {
                    if (store) atomicOr(bad, 1);
                } else {
                    rq = (float)C.vq_lo_r[(size_t)k3 * NDIM + d];
                }
            }
        }
        const float cn = fo[c] + rq;
        if (store) c_out[fi * Cc + c] = cn;
        xn[c] = cn;
    } else if (c < Cc) {
        const float v = pitch[fi * (Cc - F) + (c - F)];
        if (store) c_out[fi * Cc + c] = v;
        xn[c] = v;
    }
}

#include "predictor_df.h"
#include "predictor_ws.h"
#include "predictor_wsd.h"

// stand-alone quantizers: one workgroup per input row
__global__ __launch_bounds__(NT) void k_vq(const CbDev C, int which, const float* __restrict__ r, double* qr,
                                           int* idx) {
    __shared__ SearchLds L;
    const int n = blockIdx.x, tid = threadIdx.x;
    float rv = 0.0f;
    if (tid < NDIM) L.rs[1 + tid] = rv = r[(size_t)n * NDIM + tid];
    // a NaN / infinite row has no nearest entry (the arg-min would come back as 0x7fffffff and be used as an address):
    // NaN out, symbols -2
    if (__syncthreads_or(!(fabsf(rv) <= 3.0e38f))) {
        if (tid < NDIM) qr[(size_t)n * NDIM + tid] = __longlong_as_double(0x7ff8000000000000ll);
        if (tid < 2 && idx) idx[n * 2 + tid] = -2;
        return;
    }
    const bool hi = which == 0;
    vq_mstage(L, hi ? C.S_hi : 1, hi ? C.vq_hi0 : C.vq_lo, hi ? C.vq_hi0_r : C.vq_lo_r, hi ? C.N_hi0 : C.N_lo,
              hi ? C.vq_hi1 : nullptr, hi ? C.vq_hi1_r : nullptr, hi ? C.N_hi1 : 0, tid);
    if (tid < NDIM) qr[(size_t)n * NDIM + tid] = L.qv[tid];
#ifdef FPC_VQ_PROF
    __syncthreads();
    if (tid < 8) qr[(size_t)n * NDIM + tid] = (double)(L.prof[tid] - L.prof[0]);
#endif
    if (tid < 2 && idx) idx[n * 2 + tid] = L.res_i[tid];
}

__global__ __launch_bounds__(NT) void k_scl(const CbDev C, int which, const float* __restrict__ x, double* q,
                                            int* idx) {
    __shared__ SearchLds L;
    const int n = blockIdx.x, tid = threadIdx.x;
    if (!(fabsf(x[n]) <= 3.0e38f)) {  // (workgroup-uniform) no nearest code of a NaN: NaN out, symbol -2
        if (tid == 0) {
            q[n] = __longlong_as_double(0x7ff8000000000000ll);
            if (idx) idx[n] = -2;
        }
        return;
    }
    scl_search(L, x[n], which == 0 ? C.scl_hi : C.scl_lo, which == 0 ? C.n_hi : C.n_lo, tid);
    if (tid == 0) {
        q[n] = L.qs;
        if (idx) idx[n] = L.res_i[2];
    }
}

// ---------------------------------------------------------------------------------
// Predictor training step (src/train_frame.py:53-120, the live branch): teacher-forced
// forward with the activations kept, MSE against the next frame, back-propagation through time, weight
// gradients on the matrix cores, Adam.  Every evaluation order is the one of oracle/fpc_oracle.c
// (orc_train_step): results are bit-identical to it; the oracle is pinned to torch autograd + torch.optim.Adam.
// Weights, gradients and Adam moments live in the transposed layouts the inference kernels use ([K][3H]).
// ---------------------------------------------------------------------------------
struct TrainBufs {              // per sample n = b*L + t
    float *h1p, *r1, *z1, *n1, *hn1, *h1;         // [N][H1]
    float *h2p, *r2, *z2, *n2, *hn2, *h2, *relu;  // [N][H2]
    float *th, *dpre;                             // [N][F]
    float *dgi1, *dgh1;                           // [N][3*H1]
    float *dgi2, *dgh2;                           // [N][3*H2]
    double* lossb;                                // [B]
};

// the training forward as two roles (predictor_df.h): k_train_fwd's frames with the latency chain on three waves and the
// recurrent products streamed by the others; same chains, same kept activations
__global__ __launch_bounds__(NT) void k_train_fwd_df(const PredDev P, const float* __restrict__ feat, int Lf, const TrainBufs T,
                                                     const SplitArgs S) {
    __shared__ DfLds L;
    const int b = blockIdx.x / S.n, half = blockIdx.x % S.n, tid = threadIdx.x;
    if (!split_wanted(S, b)) return;
    SplitCtx X = split_ctx(S, P, b, half);
    const bool writer = half == 0;
    for (int i = tid; i < P.h1; i += NT) L.h1[i] = 0.0f;
    for (int i = tid; i < P.h2; i += NT) L.h2[i] = 0.0f;
    if (tid < P.in && Lf > 0) L.x[tid] = feat[(size_t)b * Lf * P.in + tid];
    __syncthreads();
    const DfStep D = df_setup(P, L, S.n);
    df_prologue(P, D, L, tid, S.n, half);
    if (tid < FGT) {
        __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
        int fg_epoch = 0;
        for (int t = 0; t < Lf; ++t) {
            const size_t n = (size_t)b * Lf + t;
            float xn = 0.0f;
            if (t + 1 < Lf && tid < P.in) xn = feat[(n + 1) * P.in + tid];
            if (writer) {  // the states the frame starts from (whole in LDS since the previous frame's hops)
                for (int i = tid; i < P.h1; i += FGT) T.h1p[n * P.h1 + i] = L.h1[i];
                for (int i = tid; i < P.h2; i += FGT) T.h2p[n * P.h2 + i] = L.h2[i];
            }
            const DfSave sv{T.r1, T.z1, T.n1, T.hn1, T.h1, T.r2, T.z2, T.n2, T.hn2, T.h2, T.relu, T.th, n, writer};
            if (!df_foreground<true>(P, D, L, X, t, t + 1 == Lf, tid, fg_epoch, &sv)) break;
            if (t + 1 < Lf) {
                if (tid < P.in) L.x[tid] = xn;
                fg_sync(L, fg_epoch);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        for (int tb = 0; tb < Lf; ++tb)
            if (!df_background(D, L, S.n, half, tb, tb + 1 == Lf, tid - FGT)) break;
    }
    // (a launch that gave up leaves its status bit: the step's Adam update is skipped, the host reports FPC_ERR_TIMEOUT)
}

// dL/dpre of every frame and the utterance's share of the loss (float64, frames then outputs in order)
__global__ __launch_bounds__(256) void k_train_loss(const float* __restrict__ feat, int in, int F, int Lf, float scale,
                                                    const TrainBufs T) {
    // dpre = d loss / d (pre-activation of the output layer); the loss itself -- a strictly sequential float64 sum per utterance,
    // 13 us on one thread -- is not needed by anything on the device: it rides beside the weight-gradient tiles (k_grad_tn)
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < Lf * F; e += 256) {
        const int t = e / F, o = e - t * F;
        const size_t n = (size_t)b * Lf + t;
        float dp = 0.0f;
        if (t + 1 < Lf) {
            const float th = T.th[n * F + o];
            const float y = th + th;
            const float diff = y - feat[(n + 1) * in + o];
            const float g = diff * scale;
            dp = (g + g) * fmaf(-th, th, 1.0f);
        }
        T.dpre[n * F + o] = dp;
    }
}
// the squared error of utterance b: sum over (frame, output) in order of (double) diff^2, one add after the other (orc_train_step);
// one wave: 64 elements per round are fetched and squared side by side, the adds stay in sequence (every lane carries the sum)
__device__ __forceinline__ void loss_sum(const float* __restrict__ feat, int in, int F, int Lf, const TrainBufs& T, int b) {
    const int lane = threadIdx.x & 63, ne = (Lf - 1) * F;
    double lb = 0.0;
    for (int e0 = 0; e0 < ne; e0 += 64) {
        const int e = e0 + lane;
        double d2 = 0.0;
        if (e < ne) {
            const int t = e / F, o = e - t * F;
            const size_t n = (size_t)b * Lf + t;
            const float th = T.th[n * F + o];
            const float y = th + th;
            const float diff = y - feat[(n + 1) * in + o];
            d2 = (double)diff * (double)diff;
        }
        const int cnt = ne - e0 < 64 ? ne - e0 : 64;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const double v = __shfl(d2, j);
            if (j < cnt) lb += v;
        }
    }
    if (lane == 0) T.lossb[b] = lb;
}

struct BwdLds {
    float dh1n[MAX_H1], dh2n[MAX_H2], dh1[MAX_H1], dh2[MAX_H2];
    float g1i[3 * MAX_H1], g1h[3 * MAX_H1], g2i[3 * MAX_H2], g2h[3 * MAX_H2];
    float pa[4][MAX_H1];  // segment sums of the transposed products [segment][k]
    float pb[4][MAX_H1];  // (same row pitch as pa: one helper serves both)
    float dp[MAX_FC];
};

// torch-layout copies ([3H][K], row r contiguous over k) of the matrices whose transposed products the
// backward pass needs: with them `sum_r W[r][k] d[r]` is a coalesced chain like the forward's
struct BwdW {
    const float *w2i, *w2h, *w1h;
};

// all (4 adjacent k, row segment) work items of one transposed product: part[sg][k] = chain over the rows
// of segment sg of W[r][k] d[r] from 0
__device__ __forceinline__ void tprod_items(const float* __restrict__ W, const float* d, int rows, int cols,
                                            float (*part)[MAX_H1], int item, int nsplit, int half) {
    // (row split: this workgroup's slice of the output columns k; every chain is the one of the unsplit form)
    const int Qs = cols / 4 / nsplit, S = segments(rows), len = rows / S;
    const int q = half * Qs + item % Qs, sg = item / Qs;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    chain4(W + (size_t)sg * len * cols, d + sg * len, len, cols, 4 * q, a);
    *reinterpret_cast<float4*>(&part[sg][4 * q]) = a;
}
__device__ __forceinline__ float tree4(const float (*p)[MAX_H1], int S, int k) {
    if (S == 4) return (p[0][k] + p[1][k]) + (p[2][k] + p[3][k]);
    if (S == 2) return p[0][k] + p[1][k];
    return p[0][k];
}

__device__ __forceinline__ void gate_grads(float dh, float r, float z, float nn, float hn, float hp, float* gi, float* gh,
                                           int H, int i) {
    const float dn_raw = dh * (1.0f - z);
    const float dnpre = dn_raw * fmaf(-nn, nn, 1.0f);
    const float dz_raw = dh * (hp - nn);
    const float dzpre = dz_raw * (z * (1.0f - z));
    const float drpre = (dnpre * hn) * (r * (1.0f - r));
    gi[i] = drpre;
    gi[H + i] = dzpre;
    gi[2 * H + i] = dnpre;
    gh[i] = drpre;
    gh[H + i] = dzpre;
    gh[2 * H + i] = dnpre * r;
}

// back-propagation through time of one utterance; the transposed products W^T d run on torch-layout copies
// of the matrices as coalesced, row-segmented chains
__global__ __launch_bounds__(NT) void k_train_bwd(const PredDev P, const BwdW W, int Lf, const TrainBufs T,
                                                  const SplitArgs Sp) {
    __shared__ BwdLds S;
    const int b = blockIdx.x / Sp.n, half = blockIdx.x % Sp.n, tid = threadIdx.x;
    if (!split_wanted(Sp, b)) return;
    SplitCtx X = split_ctx(Sp, P, b, half);
    const bool writer = half == 0;
    const int ns = X.n;  // row split: each workgroup owns a slice of the columns k of the three transposed products
    const int H1 = P.h1, H2 = P.h2, F = P.fc;
    for (int i = tid; i < H1; i += NT) S.dh1n[i] = 0.0f;
    for (int i = tid; i < H2; i += NT) S.dh2n[i] = 0.0f;
    __syncthreads();
    for (int t = Lf - 1; t >= 0; --t) {
        const size_t n = (size_t)b * Lf + t;
        if (tid < F) S.dp[tid] = T.dpre[n * F + tid];
        __syncthreads();
        for (int i = tid; i < H2; i += NT) {  // output layer back, then the gates of GRU2
            float dr = 0.0f;
            for (int o = 0; o < F; ++o) dr = fmaf(P.fcw[(size_t)i * F + o], S.dp[o], dr);
            const float dh = (T.h2[n * H2 + i] > 0.0f ? dr : 0.0f) + S.dh2n[i];
            S.dh2[i] = dh;
            gate_grads(dh, T.r2[n * H2 + i], T.z2[n * H2 + i], T.n2[n * H2 + i], T.hn2[n * H2 + i], T.h2p[n * H2 + i],
                       S.g2i, S.g2h, H2, i);
        }
        __syncthreads();
        if (writer)
            for (int i = tid; i < 3 * H2; i += NT) {
                T.dgi2[n * 3 * H2 + i] = S.g2i[i];
                T.dgh2[n * 3 * H2 + i] = S.g2h[i];
            }
        {  // W_ih2^T dgi2 -> dh1 (part), W_hh2^T dgh2 -> dh2 of frame t-1: (4 adjacent k, row segment) items
            const int S2 = segments(3 * H2), na = (H1 / 4 / ns) * S2, nb = (H2 / 4 / ns) * S2;
            for (int it = tid; it < na + nb; it += NT) {
                if (it < na)
                    tprod_items(W.w2i, S.g2i, 3 * H2, H1, S.pa, it, ns, half);
                else
                    tprod_items(W.w2h, S.g2h, 3 * H2, H2, S.pb, it - na, ns, half);
            }
            __syncthreads();
            const int H1s = H1 / ns, H2s = H2 / ns;
            for (int ww = tid; ww < H1s + H2s; ww += NT) {
                if (ww < H1s) {
                    const int w = half * H1s + ww;
                    S.dh1[w] = tree4(S.pa, S2, w) + S.dh1n[w];
                } else {
                    const int k = half * H2s + (ww - H1s);
                    S.dh2n[k] = fmaf(S.dh2[k], T.z2[n * H2 + k], tree4(S.pb, S2, k));
                }
            }
        }
        __syncthreads();
        if (ns > 1) {
            // both vectors change hands in ONE hop per frame, through the granule set of the frame's parity: a set is
            // rewritten two hops later, after the partners' intervening publish proves they have read it
            unsigned long long* ga = (t & 1) ? X.g3 : X.g1;
            unsigned long long* gb = (t & 1) ? X.g4 : X.g2;
            publish_slice(S.dh1, H1, X, ga, tid);
            publish_slice(S.dh2n, H2, X, gb, tid, false);
            consume_slices(S.dh1, H1, X, ga, tid);
            consume_slices(S.dh2n, H2, X, gb, tid);
        }
        for (int i = tid; i < H1; i += NT)
            gate_grads(S.dh1[i], T.r1[n * H1 + i], T.z1[n * H1 + i], T.n1[n * H1 + i], T.hn1[n * H1 + i],
                       T.h1p[n * H1 + i], S.g1i, S.g1h, H1, i);
        __syncthreads();
        if (writer)
            for (int i = tid; i < 3 * H1; i += NT) {
                T.dgi1[n * 3 * H1 + i] = S.g1i[i];
                T.dgh1[n * 3 * H1 + i] = S.g1h[i];
            }
        {
            const int S1 = segments(3 * H1), n1 = (H1 / 4 / ns) * S1, H1s = H1 / ns;
            for (int it = tid; it < n1; it += NT) tprod_items(W.w1h, S.g1h, 3 * H1, H1, S.pa, it, ns, half);
            __syncthreads();
            for (int kk = tid; kk < H1s; kk += NT) {
                const int k = half * H1s + kk;
                S.dh1n[k] = fmaf(S.dh1[k], T.z1[n * H1 + k], tree4(S.pa, S1, k));
            }
        }
        __syncthreads();  // (dh1n is only ever read at this workgroup's own columns: it stays local)
    }
}

#include "predictor_bwd_ws.h"

// C[k][r] = sum over n (ascending, fmaf chain from 0 = what the f32 MFMA accumulates) of A[n][k] * D[n][r]:
// the gradient of a weight matrix in the transposed layout.  One wave = 16 k-rows x 64 r-columns
// (1 A fragment, 4 D fragments per 4 samples), 4 waves per block.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct GradJobs {  // the five weight matrices of the predictor in one launch (blockIdx.z = job)
    const float* A[5];
    const float* D[5];
    float* C[5];
    float* bsum[5];
    float* part[5];  // [GSEG][K][R] segment sums
    int K[5], R[5];
};
constexpr int GSEG = 8;  // sample segments per gradient tile (oracle: ORC_GSEG)
#ifndef FPC_GKT
#define FPC_GKT 4
#endif
constexpr int GKT = FPC_GKT;  // 16-row k tiles per wave of k_grad_tn
// bias gradient of column r of job `job`: plain sum over the samples in ascending order (grad_w of the oracle)
__device__ __forceinline__ void colsum(const GradJobs& J, int N, int job, int r) {
    const float* __restrict__ D = J.D[job];
    float* __restrict__ out = J.bsum[job];
    const int R = J.R[job];
    if (r >= R) return;
    float s = 0.0f;
    int n = 0;
    for (; n + 64 <= N; n += 64) {  // the adds stay in order; 64 loads in flight ahead of them (a wave's limit; latency-bound)
        float v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = D[(size_t)(n + u) * R + r];
#pragma unroll
        for (int u = 0; u < 64; ++u) s = s + v[u];
    }
    for (; n < N; ++n) s = s + D[(size_t)n * R + r];
    out[r] = s;
}
// Which tile a workgroup computes (k_grad_tn's 1-D grid).  The workgroups of a launch go to the XCDs round-robin by linear
// index, and every XCD has its own L2: workgroup l takes sample segment l % 8 -- with 8 segments (ORC_GSEG) and 8 XCDs ALL tiles
// of a segment, of every job, run on ONE XCD, where the rows of A and D they share are fetched into the L2 once.  (As a 3-D
// grid, tiles sharing D sat on different XCDs: TCC_MISS 8.7 M of 9.0 M requests, 1.1 GB per launch from beyond the L2 for
// 240 MB of operands.)  Slots = the (job, column block, row block) tiles of one segment, the largest job first; behind them
// `ncol` workgroups of bias sums and B / 4 of loss sums (strictly sequential adds, latency-bound on a few waves: beside the tiles
// they cost nothing).
struct GradMap {
    unsigned char job[128], bx[128], by[128];
    unsigned char cjob[32], cbx[32];
    int nslots, ncol;
};
static_assert(GSEG == 8, "k_grad_tn: one sample segment per XCD");
__global__ __launch_bounds__(256) void k_grad_tn(const GradJobs J, int N, int seglen, const GradMap M, int Lf, const TrainBufs T) {
    const int lin = blockIdx.x;
    if (lin >= GSEG * M.nslots) {
        const int c = lin - GSEG * M.nslots;
        if (c < M.ncol) {
            colsum(J, N, M.cjob[c], M.cbx[c] * 256 + threadIdx.x);
        } else {  // the utterances' losses, four per workgroup (one wave each): J.A[0] = the features, K[0] = their width, R[4] = fc
            const int b = 4 * (c - M.ncol) + (threadIdx.x >> 6);
            if (b < N / Lf) loss_sum(J.A[0], J.K[0], J.R[4], Lf, T, b);
        }
        return;
    }
    const int sg = lin % GSEG, slot = lin / GSEG;
    const int job = M.job[slot], bxi = M.bx[slot], byi = M.by[slot];
    const float* __restrict__ A = J.A[job];
    const float* __restrict__ D = J.D[job];
    const int K = J.K[job], R = J.R[job];
    float* __restrict__ C = J.part[job] + (size_t)sg * K * R;
    const int nbeg = sg * seglen, nend = (sg + 1) * seglen < N ? (sg + 1) * seglen : N;
    // one wave = GKT x 16 k-rows x 64 r-columns: GKT A fragments + 4 D fragments per 4 samples feed 4 GKT MFMAs (round 5: GKT = 4,
    // 0.5 loads per MFMA; 16 k-rows per wave was 1.25 and L2-bound at 21 % of the f32 MFMA peak)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = byi * 16 * GKT, r0 = (bxi * 4 + wave) * 64;
    if (r0 >= R) return;
    const int fi = lane & 15, kq = lane >> 4;
    f32x4 acc[GKT][4];
#pragma unroll
    for (int kt = 0; kt < GKT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef FPC_GUN
#define FPC_GUN 4
#endif
    constexpr int UN = FPC_GUN;  // groups of 4 samples per fetch (the MFMA chains stay in sample order)
    auto fetch = [&](float (&a)[UN][GKT], float (&d)[UN][4], int n0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int n = n0 + 4 * u + kq;
            const bool vn = n < nend;
#pragma unroll
            for (int kt = 0; kt < GKT; ++kt) {
                const int ka = k0 + 16 * kt + fi;
                a[u][kt] = (vn && ka < K) ? A[(size_t)n * K + ka] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rc = r0 + 16 * j + fi;
                d[u][j] = (vn && rc < R) ? D[(size_t)n * R + rc] : 0.0f;
            }
        }
    };
    auto products = [&](const float (&a)[UN][GKT], const float (&d)[UN][4]) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int kt = 0; kt < GKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[kt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][kt], d[u][j], acc[kt][j], 0, 0, 0);
    };
    // (two register buffers -- the next samples on their way during the products -- measured equal without the XCD map and
    //  slower with it, 2.05 against 2.00 ms per step: 240 VGPRs leave one wave per SIMD; profiles/r05_ablations.txt item 11)
    for (int n0 = nbeg; n0 < nend; n0 += 4 * UN) {
        float a[UN][GKT], d[UN][4];
        fetch(a, d, n0);
        products(a, d);
    }
#pragma unroll
    for (int kt = 0; kt < GKT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) {  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
                const int kr = k0 + 16 * kt + kq * 4 + v, rc = r0 + 16 * j + fi;
                if (kr < K && rc < R) C[(size_t)kr * R + rc] = acc[kt][j][v];
            }
}

// The teacher-forced forward's output layer over ALL frames at once: y[n] = 2 tanh(fc(relu(h2[n]))) for the samples n = b L + t
// of the groups the weights-stationary launch served.  Inside k_forward_ws the layer cost the workgroup that owns an utterance
// 40 MFMAs per frame for ONE useful column of a 16-column tile, on the background waves that also gather hop 1: those
// workgroups were the frame's pole (forward 1.33 -> 1.25 ms without it).  Here the tile's M dimension is 16 SAMPLES: the same
// products in the same order -- per output row 8 input segments of 16, each a k-ordered fmaf chain (segment 0 from the bias)
// = what the f32 MFMA accumulates, the segment sums as a balanced tree, rows 16 and 17 on a second tile -- for 1/16 of the
// matrix work.  relu(h2) comes from the history k_forward_ws writes; one wave = 16 samples through an LDS tile.
__global__ __launch_bounds__(256) void k_out_layer(const PredDev P, const float* __restrict__ relu, int N, int Lf,
                                                   const unsigned* __restrict__ dec, float* __restrict__ y) {
    constexpr int PT = WH2 + 4;  // row pitch of the sample tile (16-byte rows, two-way conflicts at most)
    __shared__ __attribute__((aligned(16))) float tile[4][16 * PT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    float w0[8][4], w1[8][4];  // B operands: fc weights [k][output], outputs 0 .. 15 and 16, 17 (zero beyond)
#pragma unroll
    for (int sg = 0; sg < 8; ++sg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * sg + 4 * j + q;
            w0[sg][j] = P.fcw[(size_t)k * WFC + c];
            w1[sg][j] = 16 + c < WFC ? P.fcw[(size_t)k * WFC + 16 + c] : 0.0f;
        }
    const float b0 = P.fcb[c], b1 = 16 + c < WFC ? P.fcb[16 + c] : 0.0f;
    float* tl = tile[wave];
    for (int n0 = (blockIdx.x * 4 + wave) * 16; n0 < N; n0 += gridDim.x * 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // 16 rows x 128 floats = 512 float4: 8 per lane
            const int e = lane + 64 * i, row = e >> 5, c4 = e & 31;
            f32x4ws v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + row < N) v = *reinterpret_cast<const f32x4ws*>(relu + (size_t)(n0 + row) * WH2 + 4 * c4);
            *reinterpret_cast<f32x4ws*>(tl + row * PT + 4 * c4) = v;
        }
        // (one wave writes and reads its own tile: one in-order LDS queue per wave)
        f32x4ws a0[8], a1[8];
#pragma unroll
        for (int sg = 0; sg < 8; ++sg) {
            const float i0 = sg == 0 ? b0 : 0.0f, i1 = sg == 0 ? b1 : 0.0f;
            a0[sg] = f32x4ws{i0, i0, i0, i0};
            a1[sg] = f32x4ws{i1, i1, i1, i1};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = tl[c * PT + 16 * sg + 4 * j + q];  // A[sample c][k]
                a0[sg] = ws_mfma(h, w0[sg][j], a0[sg]);
                a1[sg] = ws_mfma(h, w1[sg][j], a1[sg]);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {  // C/D layout: column = lane & 15 (output), rows 4 q + v (samples)
            const int n = n0 + 4 * q + v;
            if (n >= N) continue;
            const int b = n / Lf;
            const unsigned gd = __hip_atomic_load((gu32*)(dec + b / 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gd == WS_DEAD) {  // a workgroup of the group gave up: the group's outputs fail loudly (fpcodec.h), whole
                const float qnan = __uint_as_float(0x7fc00000u);
                y[(size_t)n * WFC + c] = qnan;
                if (16 + c < WFC) y[(size_t)n * WFC + 16 + c] = qnan;
                continue;
            }
            if (gd != WS_GO) continue;  // (WS_FALLBACK: the row-split launch has written this utterance's rows)
            const float s0 = ((a0[0][v] + a0[1][v]) + (a0[2][v] + a0[3][v])) + ((a0[4][v] + a0[5][v]) + (a0[6][v] + a0[7][v]));
            const float t0 = fpc_tanhf(s0);
            y[(size_t)n * WFC + c] = t0 + t0;  // the "dual" FC is the same Linear summed twice (wavernn.py:89-92)
            if (16 + c < WFC) {
                const float s1 = ((a1[0][v] + a1[1][v]) + (a1[2][v] + a1[3][v])) + ((a1[4][v] + a1[5][v]) + (a1[6][v] + a1[7][v]));
                const float t1 = fpc_tanhf(s1);
                y[(size_t)n * WFC + 16 + c] = t1 + t1;
            }
        }
    }
}

// dst[r][k] = src[k][r]: the torch-layout copies the backward pass reads, at the first step (k_adam keeps them current)
__global__ void k_transpose(const float* __restrict__ src, int K, int R, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K * R) return;
    const int r = (int)(i / K), k = (int)(i - (size_t)r * K);
    dst[i] = src[(size_t)k * R + r];
}

// segment sums -> gradient, balanced tree; one thread also latches the handle's status word for the Adam launches behind it:
// the word lives in host memory (one PCIe round trip per reader), so it is copied into device memory once per step.
// latch[0] = the status word, latch[1] = the number of steps whose update was applied (the host's step counter, and with it
// Adam's bias corrections, follows this count after a failed step: fpc_trainer_step)
__global__ void k_grad_reduce(const GradJobs J, const unsigned* status, unsigned* latch) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const unsigned st = status_load(status);
        latch[0] = st;
        if (st == 0u) latch[1] += 1u;
    }
    const int job = blockIdx.y;
    const size_t n = (size_t)J.K[job] * J.R[job];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = J.part[job];
    J.C[job][i] = ((p[i] + p[n + i]) + (p[2 * n + i] + p[3 * n + i])) + ((p[4 * n + i] + p[5 * n + i]) + (p[6 * n + i] + p[7 * n + i]));
}

// torch.optim.Adam, single-tensor path (betas 0.9 / 0.999, eps 1e-8, no weight decay), the ten parameter tensors in ONE launch
// (blockIdx -> tensor through the prefix of their block counts); the three matrices the backward pass reads transposed get their
// torch-layout copies refreshed by the thread that updates the element (thirteen launches of ~4.6 us each were 3 % of the step)
struct AdamJobs {
    float* p[10];
    float* m[10];
    float* v[10];
    const float* g[10];
    float* tp[10];  // the transposed copy [R][K] of a [K][R] matrix, or null
    int tK[10], tR[10];
    unsigned n[10], blk0[11];
};
__global__ void k_adam(const AdamJobs J, float step_size, float bc2_sqrt, const unsigned* __restrict__ latch) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 10; ++j) k += blockIdx.x >= J.blk0[j];
    const unsigned i = (blockIdx.x - J.blk0[k]) * blockDim.x + threadIdx.x;
    if (i >= J.n[k]) return;
    if (*latch != 0u) return;  // the step's forward or backward gave up: the weights stay as they are
    float* __restrict__ p = J.p[k];
    float* __restrict__ m = J.m[k];
    float* __restrict__ v = J.v[k];
    const float gi = J.g[k][i];
    const float mi = fmaf(0.1f, gi - m[i], m[i]);
    const float vi = fmaf(0.001f, gi * gi, v[i] * 0.999f);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
    const float pn = p[i] - step_size * (mi / denom);
    p[i] = pn;
    if (J.tp[k] != nullptr) {  // element [kk][r] of the [K][R] matrix -> [r][kk] of its copy
        const unsigned kk = i / (unsigned)J.tR[k], r = i - kk * (unsigned)J.tR[k];
        J.tp[k][(size_t)r * J.tK[k] + kk] = pn;
    }
}

}  // namespace


// =====================================================================================
// host side
// =====================================================================================
// compile-time tunables of this translation unit that differ from the shipped defaults (fpc_build_info, api.hip)
namespace fpc {
void predictor_build_info(std::string& out) {
    char b[64];
#define FPC_TUNE(name, value, dflt)                            \
    if ((value) != (dflt)) {                                   \
        snprintf(b, sizeof b, " " name "=%d", (int)(value));   \
        out += b;                                              \
    }
    FPC_TUNE("FPC_NT", FPC_NT, 512)
    FPC_TUNE("FPC_CD", FPC_CD, 16)
    FPC_TUNE("FPC_DFW", FPC_DFW, 16)
    FPC_TUNE("FPC_FG_PRIO", FPC_FG_PRIO, 2)
    FPC_TUNE("FPC_BSTAMP_WAVE", FPC_BSTAMP_WAVE, 0)
    FPC_TUNE("FPC_WS_A_SPLIT", FPC_WS_A_SPLIT, 0)
    FPC_TUNE("FPC_WS_POLL_DELAY", FPC_WS_POLL_DELAY, 0)
#undef FPC_TUNE
#ifdef FPC_WAVES_EU
    out += " FPC_WAVES_EU";
#endif
#ifdef FPC_XCHG_NOSLEEP
    out += " FPC_XCHG_NOSLEEP";
#endif
#ifdef FPC_PRED_PROF
    out += " FPC_PRED_PROF";
#endif
#ifdef FPC_VQ_PROF
    out += " FPC_VQ_PROF";
#endif
#ifdef FPC_WS_PROF
    out += " FPC_WS_PROF";
#endif
}
}  // namespace fpc
struct fpc_predictor {
    PredDev d;
    fpc::DevBuf buf[10];
    fpc::DevBuf flag;  // one int: "a symbol lay outside its codebook" (fpc_decode_features), allocated once per handle
    fpc::DevBuf xg;    // row-split exchange granules [B][2][h1 + h2] x 8 bytes, grown on demand
    fpc::DevBuf wsg;   // weights-stationary form (predictor_ws.h): granule blocks [groups][WGRANULES] x 16 bytes
    fpc::DevBuf h2hist; // ... relu(h2) of every sample of a forward call [B][L][128] (k_out_layer), grown on demand
    fpc::DevBuf wsidx; // ... and the symbols of an encode call whose caller passes no idx buffer (the histograms are counted from them)
    // status word: host-mapped pinned memory the kernels OR failure bits into (FPC_ST_*); sticky until
    // fpc_predictor_status() clears it; read by the host without a synchronisation at the start of every call
    unsigned* status_host = nullptr;
    unsigned* status_dev = nullptr;
    // launches on this handle share its granule block: a call on another stream first waits for the last launch
    hipEvent_t last_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool launched = false;
    int last_ws_groups = 0;  // groups of the handle's last weights-stationary launch (fpc_predictor_fallback_groups)
    size_t last_ws_gbytes = 0;  // ... and where its decision words start in wsg
    int forced_split = 0;  // fpc_predictor_set_split: 0 automatic, 1 off, 2/4/8 exactly that many workgroups per utterance
    int num_cus = 0;
    int occ_df = 1;  // workgroups of the two-role kernels that one CU holds (LDS-bound: 1)
    int fail_epoch = 0;    // failures reported and cleared by fpc_predictor_status so far (a trainer re-reads its step count)
    std::atomic<int> refs{1};  // the creator's handle + one per live fpc_trainer built on it (fpc_predictor_destroy only drops a reference)
    ~fpc_predictor() {
        if (last_ev) (void)hipEventDestroy(last_ev);
        if (status_host) (void)hipHostFree(status_host);
    }
};
// the sticky status word as an error code (no synchronisation: what the device has reported so far)
static int status_error(const fpc_predictor* p, const char* who) {
    const unsigned st = p->status_host ? *(volatile unsigned*)p->status_host : 0u;
    if (st & FPC_ST_TIMEOUT) {
        fpc::set_error("%s: a row-split exchange timed out (a workgroup of an utterance's group was not dispatched "
                       "or stalled for > 1 s: is the GPU shared?); the outputs of that launch are NaN / -2; "
                       "fpc_predictor_status() clears the condition, fpc_predictor_set_split(p, 1) avoids the exchange", who);
        return FPC_ERR_TIMEOUT;
    }
    if (st & FPC_ST_NONFINITE) {
        fpc::set_error("%s: a non-finite residual reached a quantizer (NaN / infinity in the features or weights); "
                       "those frames carry the symbols -2; fpc_predictor_status() clears the condition", who);
        return FPC_ERR_NONFINITE;
    }
    return FPC_OK;
}
// every launch on the handle goes through these two: cross-stream ordering on the shared granule block
static int before_launch(fpc_predictor* p, hipStream_t st) {
    if (p->launched && p->last_stream != st) FPC_HIP(hipStreamWaitEvent(st, p->last_ev, 0));
    return FPC_OK;
}
static int after_launch(fpc_predictor* p, hipStream_t st) {
    FPC_HIP(hipEventRecord(p->last_ev, st));
    p->last_stream = st;
    p->launched = true;
    return FPC_OK;
}
// Row split (2, 4 or 8 workgroups per utterance) while the batch leaves CUs idle; fpc_predictor_set_split or
// FPC_PRED_SPLIT=0/1/2/4/8 force it off / to a count (tests run every form).  Prepares the zeroed granule block on the
// stream.  The automatic choice assumes that this process owns the GPU (every workgroup of a group must be resident).
static int split_args(fpc_predictor* p, int B, hipStream_t st, SplitArgs* out) {
    out->n = 1;
    out->g = nullptr;
    out->err = p->status_dev;
    out->limit = 100000000ull;  // 1 s of s_memrealtime (100 MHz)
    out->withhold = 0;
    out->no_fast = 0;
    out->only = nullptr;
    if (const char* lim = getenv("FPC_SPIN_LIMIT_US")) {  // test hook: a shorter give-up bound
        const long us = atol(lim);
        if (us > 0) out->limit = (unsigned long long)us * 100ull;
    }
    if (const char* wh = getenv("FPC_TEST_WITHHOLD_PUBLISH")) out->withhold = wh[0] == '1';  // test hook
    if (const char* fh = getenv("FPC_FAST_HOP")) out->no_fast = fh[0] == '0';  // the write-through exchange everywhere
    const char* env = getenv("FPC_PRED_SPLIT");  // 0: never; 2/4/8: exactly that many (tests); unset: as many as leave
    int n = 1;                                    // every workgroup a CU of its own, up to 8
    for (int c = 2; c <= 8; c *= 2)
        if (p->d.h1 % (4 * c) == 0 && p->d.h2 % (4 * c) == 0 && c * B <= p->num_cus) n = c;
    int f = p->forced_split;
    if (env && env[0] >= '0' && env[0] <= '8') f = env[0] == '0' ? 1 : env[0] - '0';
    // (forced: every workgroup of the launch must be resident at once -- a group that straddles the residency boundary would
    //  spin for workgroups that are not dispatched yet -- so the bound is what the chosen kernel form's LDS lets a CU hold:
    //  one workgroup of the two-role kernels (~140 kB), two or three of the phase kernels)
    const int occ = p->occ_df;
    if (f == 1)
        n = 1;
    else if ((f == 2 || f == 4 || f == 8) && p->d.h1 % (4 * f) == 0 && p->d.h2 % (4 * f) == 0 && f * B <= occ * p->num_cus)
        n = f;
    {
        const int rc = before_launch(p, st);
        if (rc != FPC_OK) return rc;
    }
    if (n == 1) return FPC_OK;
    const size_t bytes = (size_t)B * 2 * (p->d.h1 + p->d.h2) * sizeof(unsigned long long);
    if (p->xg.bytes < bytes) {
        if (p->xg.p) {
            FPC_HIP(hipDeviceSynchronize());  // nothing may still be polling the old block (on any stream)
            p->xg.release();  // (bytes = 0: a failed allocation below leaves no size behind)
        }
        FPC_HIP(p->xg.alloc(bytes));
    }
    FPC_HIP(hipMemsetAsync(p->xg.p, 0, bytes, st));  // tags start at 0; epochs count from 1 within the launch
    out->n = n;
    out->g = p->xg.as<unsigned long long>();
    return FPC_OK;
}

// The weights-stationary kernels (predictor_ws.h) serve the reference's production shape on a whole MI355X (8 XCDs x 32
// CUs) whenever the caller has not pinned a row split: FPC_PRED_WS=0 turns them off (tests compare the forms bit for
// bit), a pinned split (fpc_predictor_set_split, FPC_PRED_SPLIT) or FPC_PRED_DF=0 selects the row-split kernels.
static bool ws_wanted(const fpc_predictor* p) {
    if (p->d.in != WIN || p->d.h1 != WH1 || p->d.h2 != WH2 || p->d.fc != WFC) return false;
    if (p->num_cus < 8 * WNS || p->forced_split != 0) return false;
    if (getenv("FPC_PRED_SPLIT")) return false;
    if (const char* e = getenv("FPC_PRED_DF"))
        if (e[0] == '0') return false;
    if (const char* e = getenv("FPC_PRED_WS"))
        if (e[0] == '0') return false;
    return true;
}
struct fpc_codebooks {
    CbDev d;
    fpc::DevBuf buf[8];
    int hist_size = 0;
};
// the encoder's distributed frame tail (wsd_tail) holds 32 entries of every book per workgroup (1 024 per book over the
// group's 32 workgroups) and the scalar codes in LDS
static bool ws_codebooks_fit(const fpc_codebooks* cb) {
    if (!cb) return true;
    const CbDev& c = cb->d;
    return c.N_hi0 <= 2 * NT && c.N_hi1 <= 2 * NT && c.N_lo <= 2 * NT && c.n_hi <= 256 && c.n_lo <= 256 && c.n_hi + c.n_lo <= SCLC;
}
// granule blocks zeroed on the stream, test hooks read; the grid is ws_grid(args) workgroups
static int ws_args(fpc_predictor* p, int B, hipStream_t st, WsArgs* out, int granules = WGRANULES) {
    out->B = B;
    out->ngroups = (B + WG - 1) / WG;
    out->err = p->status_dev;
    out->limit = 100000000ull;  // 1 s of s_memrealtime (100 MHz)
    out->withhold = 0;
    out->no_fast = 0;
    out->hello_limit = 1000000ull;  // 10 ms: the partners of a group that has the chip to itself show up within microseconds
    if (const char* lim = getenv("FPC_SPIN_LIMIT_US")) {  // test hook: a shorter give-up bound
        const long us = atol(lim);
        if (us > 0) out->limit = (unsigned long long)us * 100ull;
    }
    if (const char* lim = getenv("FPC_HELLO_LIMIT_US")) {  // test hook: a shorter / longer residency bound
        const long us = atol(lim);
        if (us > 0) out->hello_limit = (unsigned long long)us * 100ull;
    }
    // test hooks: 1 the last workgroup of group 0 never publishes a frame's values (the frame loop's give-up), 2 ("hello")
    // not even its hello (stands for a workgroup that is not resident: the group falls back)
    if (const char* wh = getenv("FPC_TEST_WITHHOLD_PUBLISH")) out->withhold = wh[0] == '1' ? 1 : (wh[0] == 'h' ? 2 : 0);
    if (const char* fh = getenv("FPC_FAST_HOP")) out->no_fast = fh[0] == '0';  // the write-through path everywhere
    {
        const int rc = before_launch(p, st);
        if (rc != FPC_OK) return rc;
    }
    const size_t gbytes = (size_t)out->ngroups * granules * sizeof(u32x4);
    const size_t bytes = gbytes + (((size_t)out->ngroups * sizeof(unsigned) + 255) & ~(size_t)255);  // + the decision words
    if (p->wsg.bytes < bytes) {
        if (p->wsg.p) {
            FPC_HIP(hipDeviceSynchronize());  // nothing may still be polling the old block (on any stream)
            p->wsg.release();  // (bytes = 0: a failed allocation below leaves no size behind)
        }
        FPC_HIP(p->wsg.alloc(bytes));
    }
    FPC_HIP(hipMemsetAsync(p->wsg.p, 0, bytes, st));  // tags start at 0; epochs count from 1 within the launch
    out->g = p->wsg.as<u32x4>();
    out->dec = reinterpret_cast<unsigned*>(static_cast<char*>(p->wsg.p) + gbytes);
    p->last_ws_groups = out->ngroups;
    p->last_ws_gbytes = gbytes;
    return FPC_OK;
}
// the launch behind every weights-stationary launch: the row-split kernels, one workgroup per utterance (no partner to
// wait for), for exactly the groups whose workgroups could not all become resident (ws_hello) -- none, normally: B
// workgroups that read one word and return
static SplitArgs ws_fallback_args(const fpc_predictor* p, const WsArgs& wa) {
    SplitArgs sp;
    sp.n = 1;
    sp.g = nullptr;
    sp.err = p->status_dev;
    sp.limit = wa.limit;
    sp.withhold = 0;
    sp.no_fast = 0;
    sp.only = wa.dec;
    return sp;
}
#ifdef FPC_WS_PROF
// diagnostic builds: cycles per frame and stage of workgroup 5 of group 0 (thread 0 = foreground, thread 256 = background)
static void ws_prof_print(fpc_predictor* p, const char* who, int B, hipStream_t st) {
    (void)hipStreamSynchronize(st);
    volatile unsigned* w = (volatile unsigned*)p->status_host;
    fprintf(stderr, "%s B=%d cycles/frame FG: I %u waitA %u gates1 %u gather1 %u waitH1 %u C %u waitB %u gates2 %u gather2 %u waitH2 %u "
            "fc %u out %u tail %u | rendezvous %u frame-tail(wsd: rows+outputs) %u hop3(wsd: last barrier) %u | BG wave 1: waitP1 %u gather1 %u waitH1 %u A %u waitC %u (unused %u) "
            "toRendezvous %u rest %u | tail stamps 25..31 (wsd: residual, stage 1, scalar, gather 1, survivors, stage 2, gather 2): %u %u %u %u %u %u %u\n", who, B, w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13],
            w[21], w[22], w[23], w[14], w[15], w[16], w[17], w[18], w[19], w[20], w[24], w[26], w[27], w[28], w[29], w[30], w[31], w[32]);
    for (int k = 1; k < 40; ++k) w[k] = 0;
}
#endif
// 32 workgroups per group, the groups dealt over the XCDs (ws_role): 256 workgroups per round of 8 groups
static unsigned ws_grid(const WsArgs& a) { return 8u * WNS * (unsigned)((a.ngroups + 7) / 8); }

static void predictor_unref(fpc_predictor* p) {
    if (p && p->refs.fetch_sub(1) <= 1) delete p;
}


static std::vector<float> transpose_f(const float* src, int rows, int cols) {
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = src[(size_t)r * cols + c];
    return t;
}

extern "C" int fpc_predictor_create(const fpc_predictor_weights* w, fpc_predictor** out) {
    FPC_REQUIRE(w && out, "fpc_predictor_create: null argument");
    if (!fpc::have_device()) {
        fpc::set_error("fpc_predictor_create: no HIP device (libfpcodec has no CPU fallback)");
        return FPC_ERR_NO_DEVICE;
    }
    FPC_REQUIRE(w->in_features > 0 && w->in_features <= MAX_IN && w->gru_units1 > 0 &&
                    w->gru_units1 <= MAX_H1 && w->gru_units2 > 0 && w->gru_units2 <= MAX_H2 &&
                    w->fc_units > 0 && w->fc_units <= MAX_FC && w->fc_units <= w->in_features &&
                    w->gru_units1 % 4 == 0 && w->gru_units2 % 4 == 0,
                "fpc_predictor_create: unsupported sizes in=%d h1=%d h2=%d fc=%d", w->in_features,
                w->gru_units1, w->gru_units2, w->fc_units);
    FPC_REQUIRE(w->rnn1_weight_ih && w->rnn1_weight_hh && w->rnn1_bias_ih && w->rnn1_bias_hh &&
                    w->rnn2_weight_ih && w->rnn2_weight_hh && w->rnn2_bias_ih && w->rnn2_bias_hh &&
                    w->fc_weight && w->fc_bias,
                "fpc_predictor_create: null weight pointer");
    std::unique_ptr<fpc_predictor> own(new fpc_predictor());  // freed on every early return below
    fpc_predictor* p = own.get();
    const int in = w->in_features, h1 = w->gru_units1, h2 = w->gru_units2, fc = w->fc_units;
    p->d.in = in;
    p->d.h1 = h1;
    p->d.h2 = h2;
    p->d.fc = fc;
    auto upv = [&](fpc::DevBuf& b, const std::vector<float>& v, const float** dst) -> hipError_t {
        hipError_t e = b.upload(v);
        *dst = b.as<float>();
        return e;
    };
    auto vec = [](const float* s, size_t n) { return std::vector<float>(s, s + n); };
    FPC_HIP(upv(p->buf[0], transpose_f(w->rnn1_weight_ih, 3 * h1, in), &p->d.w1i));
    FPC_HIP(upv(p->buf[1], transpose_f(w->rnn1_weight_hh, 3 * h1, h1), &p->d.w1h));
    FPC_HIP(upv(p->buf[2], vec(w->rnn1_bias_ih, 3 * h1), &p->d.b1i));
    FPC_HIP(upv(p->buf[3], vec(w->rnn1_bias_hh, 3 * h1), &p->d.b1h));
    FPC_HIP(upv(p->buf[4], transpose_f(w->rnn2_weight_ih, 3 * h2, h1), &p->d.w2i));
    FPC_HIP(upv(p->buf[5], transpose_f(w->rnn2_weight_hh, 3 * h2, h2), &p->d.w2h));
    FPC_HIP(upv(p->buf[6], vec(w->rnn2_bias_ih, 3 * h2), &p->d.b2i));
    FPC_HIP(upv(p->buf[7], vec(w->rnn2_bias_hh, 3 * h2), &p->d.b2h));
    FPC_HIP(upv(p->buf[8], transpose_f(w->fc_weight, fc, h2), &p->d.fcw));
    FPC_HIP(upv(p->buf[9], vec(w->fc_bias, fc), &p->d.fcb));
    FPC_HIP(p->flag.alloc(sizeof(int)));
    {
        void* hp = nullptr;
        FPC_HIP(hipHostMalloc(&hp, 256, hipHostMallocMapped));
        p->status_host = static_cast<unsigned*>(hp);
        memset(hp, 0, 256);
        void* dp = nullptr;
        FPC_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        p->status_dev = static_cast<unsigned*>(dp);
        FPC_HIP(hipEventCreateWithFlags(&p->last_ev, hipEventDisableTiming));
    }
    {
        int dev = 0;
        FPC_HIP(hipGetDevice(&dev));
        FPC_HIP(hipDeviceGetAttribute(&p->num_cus, hipDeviceAttributeMultiprocessorCount, dev));
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_encode_df, NT, 0) == hipSuccess && o > 0) p->occ_df = o;
    }
    *out = own.release();
    return FPC_OK;
}

extern "C" void fpc_predictor_destroy(fpc_predictor* p) { predictor_unref(p); }

extern "C" int fpc_predictor_forward(fpc_predictor* p, const float* x_dev, int B, int L,
                                     float* h1_dev, float* h2_dev, float* y_dev, fpc_stream s) {
    FPC_REQUIRE(p && x_dev && h1_dev && h2_dev && y_dev, "fpc_predictor_forward: null argument");
    FPC_REQUIRE(B > 0 && L >= 0, "fpc_predictor_forward: bad shape B=%d L=%d", B, L);
    if (const int se = status_error(p, "fpc_predictor_forward")) return se;  // an earlier launch on the handle failed
    if (ws_wanted(p)) {
        WsArgs wa;
        const int rcw = ws_args(p, B, static_cast<hipStream_t>(s), &wa);
        if (rcw != FPC_OK) return rcw;
        const size_t hbytes = sizeof(float) * (size_t)B * (size_t)(L > 0 ? L : 1) * WH2;  // relu(h2) of every sample, for k_out_layer
        if (p->h2hist.bytes < hbytes) {
            FPC_HIP(hipDeviceSynchronize());  // (a launch in flight -- on this stream or another -- may still read the old block)
            p->h2hist.release();  // (bytes = 0: if the allocation fails, a later, smaller call allocates again instead of
            FPC_HIP(p->h2hist.alloc(hbytes));  //  launching on a block that is not there)
        }
        FPC_REQUIRE(p->h2hist.p != nullptr, "fpc_predictor_forward: no scratch block");
        WsSave hs{};
        hs.relu = p->h2hist.as<float>();
        hipLaunchKernelGGL(k_forward_ws<false>, dim3(ws_grid(wa)), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, x_dev, L, h1_dev,
                           h2_dev, y_dev, wa, hs);
        hipLaunchKernelGGL(k_forward_df, dim3(B), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, x_dev, L, h1_dev, h2_dev, y_dev,
                           ws_fallback_args(p, wa));
        if (L > 0) {
            const int N = B * L, groups16 = (N + 15) / 16;
            hipLaunchKernelGGL(k_out_layer, dim3((unsigned)std::min(1024, (groups16 + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(s),
                               p->d, p->h2hist.as<float>(), N, L, wa.dec, y_dev);
        }
        FPC_HIP(hipGetLastError());
#ifdef FPC_WS_PROF
        ws_prof_print(p, "k_forward_ws", B, static_cast<hipStream_t>(s));
#endif
        return after_launch(p, static_cast<hipStream_t>(s));
    }
    SplitArgs sp;
    const int rc = split_args(p, B, static_cast<hipStream_t>(s), &sp);
    if (rc != FPC_OK) return rc;
    hipLaunchKernelGGL(k_forward_df, dim3(B * sp.n), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, x_dev, L, h1_dev, h2_dev,
                       y_dev, sp);
    FPC_HIP(hipGetLastError());
#ifdef FPC_PRED_PROF
    (void)hipStreamSynchronize(static_cast<hipStream_t>(s));
    {
        const volatile unsigned* w = (const volatile unsigned*)p->status_host;
        fprintf(stderr, "k_forward_df B=%d n=%d cycles/frame FG: I %u waitA %u gates1 %u hop1 %u Ashare %u waitB %u waitC %u gates2 %u hop2 %u fc %u"
                " | BG: waitH1 %u C %u A %u waitH2 %u B %u\n", B, sp.n, w[1], w[2], w[3], w[4], w[5], w[10], w[6], w[7], w[8], w[9], w[11],
                w[12], w[13], w[14], w[15]);
    }
#endif
    return after_launch(p, static_cast<hipStream_t>(s));
}

extern "C" int fpc_predictor_status(fpc_predictor* p) {
    FPC_REQUIRE(p, "fpc_predictor_status: null handle");
    FPC_HIP(hipDeviceSynchronize());  // everything launched on the handle so far has reported
    const int rc = status_error(p, "fpc_predictor_status");
    *(volatile unsigned*)p->status_host = 0u;
    if (rc != FPC_OK) p->fail_epoch += 1;
    return rc;
}

extern "C" int fpc_predictor_fallback_groups(fpc_predictor* p) {
    FPC_REQUIRE(p, "fpc_predictor_fallback_groups: null handle");
    if (p->last_ws_groups <= 0 || !p->wsg.p) return 0;
    FPC_HIP(hipDeviceSynchronize());
    std::vector<unsigned> dec((size_t)p->last_ws_groups);
    FPC_HIP(hipMemcpy(dec.data(), static_cast<char*>(p->wsg.p) + p->last_ws_gbytes, dec.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int n = 0;
    for (unsigned d : dec) n += d == WS_FALLBACK;
    return n;
}

extern "C" int fpc_predictor_set_split(fpc_predictor* p, int n) {
    FPC_REQUIRE(p, "fpc_predictor_set_split: null handle");
    FPC_REQUIRE(n == 0 || n == 1 || n == 2 || n == 4 || n == 8, "fpc_predictor_set_split: n = %d (0 automatic, 1 off, 2, 4 or 8)", n);
    p->forced_split = n;
    return FPC_OK;
}

extern "C" int fpc_codebooks_create(const double* vq_hi, int S_hi, const int* N_hi,
                                    const double* vq_lo, int N_lo, const double* scl_hi, int n_hi,
                                    const double* scl_lo, int n_lo, fpc_codebooks** out) {
    FPC_REQUIRE(out && vq_hi && N_hi && scl_hi, "fpc_codebooks_create: null argument");
    // the reference's quantize_mstage only works for 1 or 2 stages (vq_func.py:111 breaks at 3)
    FPC_REQUIRE(S_hi == 1 || S_hi == 2, "fpc_codebooks_create: %d stages unsupported (1 or 2)", S_hi);
    FPC_REQUIRE(N_hi[0] >= SURV && (S_hi == 1 || N_hi[1] >= SURV) && n_hi > 0,
                "fpc_codebooks_create: codebook smaller than %d survivors", SURV);
    FPC_REQUIRE((vq_lo == nullptr) == (N_lo == 0) && (scl_lo == nullptr) == (n_lo == 0),
                "fpc_codebooks_create: below-threshold codebook pointer/size mismatch");
    FPC_REQUIRE(N_lo == 0 || N_lo >= SURV, "fpc_codebooks_create: vq_lo smaller than %d survivors", SURV);
    if (!fpc::have_device()) {
        fpc::set_error("fpc_codebooks_create: no HIP device (libfpcodec has no CPU fallback)");
        return FPC_ERR_NO_DEVICE;
    }
    fpc_codebooks* c = new fpc_codebooks();
    memset(&c->d, 0, sizeof c->d);
    c->d.S_hi = S_hi;
    c->d.N_hi0 = N_hi[0];
    c->d.N_hi1 = S_hi == 2 ? N_hi[1] : 0;
    c->d.N_lo = N_lo;
    c->d.n_hi = n_hi;
    c->d.n_lo = n_lo;
    auto up_cb = [&](int slot, const double* src, int N, const double** dT, const double** dR) -> hipError_t {
        std::vector<double> t((size_t)N * NDIM), r(src, src + (size_t)N * NDIM);
        for (int e = 0; e < N; ++e)
            for (int d = 0; d < NDIM; ++d) t[(size_t)d * N + e] = src[(size_t)e * NDIM + d];
        hipError_t e1 = c->buf[slot].upload(t);
        if (e1 != hipSuccess) return e1;
        *dT = c->buf[slot].as<double>();
        e1 = c->buf[slot + 1].upload(r);
        if (e1 != hipSuccess) return e1;
        *dR = c->buf[slot + 1].as<double>();
        return e1;
    };
    FPC_HIP(up_cb(0, vq_hi, N_hi[0], &c->d.vq_hi0, &c->d.vq_hi0_r));
    if (S_hi == 2)
        FPC_HIP(up_cb(2, vq_hi + (size_t)N_hi[0] * NDIM, N_hi[1], &c->d.vq_hi1, &c->d.vq_hi1_r));
    if (vq_lo) FPC_HIP(up_cb(4, vq_lo, N_lo, &c->d.vq_lo, &c->d.vq_lo_r));
    {
        std::vector<double> v(scl_hi, scl_hi + n_hi);
        FPC_HIP(c->buf[6].upload(v));
        c->d.scl_hi = c->buf[6].as<double>();
    }
    if (scl_lo) {
        std::vector<double> v(scl_lo, scl_lo + n_lo);
        FPC_HIP(c->buf[7].upload(v));
        c->d.scl_lo = c->buf[7].as<double>();
    }
    c->hist_size = n_hi + n_lo + c->d.N_hi0 + c->d.N_hi1 + N_lo;
    *out = c;
    return FPC_OK;
}

extern "C" void fpc_codebooks_destroy(fpc_codebooks* c) { delete c; }
extern "C" int fpc_codebooks_hist_size(const fpc_codebooks* c) { return c ? c->hist_size : 0; }

extern "C" int fpc_encode(fpc_predictor* p, const fpc_codebooks* cb, const float* feat_dev, int B,
                          int L, float l1, float l2, int qtz, float* c_in_dev, float* r_dev,
                          float* r_qtz_dev, float* r_under_dev, float* ind1_dev, float* ind2_dev,
                          int32_t* idx_dev, unsigned long long* hist_dev, const float* mask_dev, fpc_stream s) {
    FPC_REQUIRE(p && feat_dev && c_in_dev && r_dev && r_qtz_dev && r_under_dev && ind1_dev && ind2_dev,
                "fpc_encode: null argument");
    FPC_REQUIRE(B > 0 && L >= 0, "fpc_encode: bad shape B=%d L=%d", B, L);
    FPC_REQUIRE(!qtz || cb, "fpc_encode: qtz=1 needs codebooks");
    FPC_REQUIRE(p->d.fc == NDIM + 1, "fpc_encode: fc_units must be 18 (c0 + 17-dim VQ), got %d", p->d.fc);
    if (L == 0) return FPC_OK;
    if (const int se = status_error(p, "fpc_encode")) return se;  // an earlier launch on the handle failed
    CbDev cd;
    memset(&cd, 0, sizeof cd);
    if (cb) cd = cb->d;
    EncArgs a{feat_dev, L,  l1, l2, qtz ? 1 : 0, c_in_dev, r_dev, r_qtz_dev, r_under_dev, ind1_dev, ind2_dev,
              idx_dev,  hist_dev, mask_dev};
    if (ws_wanted(p) && ws_codebooks_fit(cb)) {
        WsArgs wa;
        const int rcw = ws_args(p, B, static_cast<hipStream_t>(s), &wa);
        if (rcw != FPC_OK) return rcw;
        if (!a.idx) {  // (the kernel always writes the symbols; the caller may not want them)
            const size_t need = (size_t)B * L * 4 * sizeof(int);
            if (p->wsidx.bytes < need) {
                if (p->wsidx.p) {
                    FPC_HIP(hipDeviceSynchronize());
                    p->wsidx.release();
                }
                FPC_HIP(p->wsidx.alloc(need));
            }
            a.idx = p->wsidx.as<int>();
        }
        // (the frame tail is distributed over the group's workgroups: predictor_wsd.h)
        hipLaunchKernelGGL(k_encode_wsd, dim3(ws_grid(wa)), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, cd, a, wa);
        {  // (the fallback's symbols are counted with everybody's by k_hist_symbols below, not by its own atomics)
            EncArgs af = a;
            af.hist = nullptr;
            hipLaunchKernelGGL(k_encode_df, dim3(B), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, cd, af, ws_fallback_args(p, wa));
        }
        FPC_HIP(hipGetLastError());
        if (hist_dev && qtz) {
            const size_t frames = (size_t)B * L;
            const unsigned hblocks = (unsigned)std::min<size_t>(32, (frames + 2047) / 2048);
            hipLaunchKernelGGL(k_hist_symbols, dim3(hblocks), dim3(256), sizeof(unsigned) * (size_t)cb->hist_size, static_cast<hipStream_t>(s),
                               cd, a.idx, frames, cb->hist_size, hist_dev);
            FPC_HIP(hipGetLastError());
        }
#ifdef FPC_WS_PROF
        ws_prof_print(p, qtz ? "k_encode_wsd qtz=1" : "k_encode_wsd qtz=0", B,
                      static_cast<hipStream_t>(s));
#endif
        return after_launch(p, static_cast<hipStream_t>(s));
    }
    SplitArgs sp;
    const int rc = split_args(p, B, static_cast<hipStream_t>(s), &sp);
    if (rc != FPC_OK) return rc;
    hipLaunchKernelGGL(k_encode_df, dim3(B * sp.n), dim3(NT), 0, static_cast<hipStream_t>(s), p->d, cd, a, sp);
#ifdef FPC_PRED_PROF
    {
        (void)hipStreamSynchronize(static_cast<hipStream_t>(s));
        const volatile unsigned* w = (const volatile unsigned*)p->status_host;
        fprintf(stderr, "k_encode_df B=%d n=%d qtz=%d cycles/frame FG: I %u waitA %u gates1 %u hop1 %u C %u waitB %u gates2 %u hop2 %u out %u | rendezvous %u"
                " frame tail + searches %u | BG: waitH1 %u A %u (signal %u) B %u\n", B, sp.n, qtz, w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8],
                w[9], w[10], w[11], w[12], w[13], w[14], w[15]);
        for (int k = 1; k < 16; ++k) ((volatile unsigned*)p->status_host)[k] = 0;
    }
#endif
    FPC_HIP(hipGetLastError());
    return after_launch(p, static_cast<hipStream_t>(s));
}

extern "C" int fpc_decode_features(fpc_predictor* p, const fpc_codebooks* cb, const float* pitch_dev,
                                   const int32_t* idx_dev, int B, int L, float* c_out_dev, fpc_stream s) {
    FPC_REQUIRE(p && cb && pitch_dev && idx_dev && c_out_dev, "fpc_decode_features: null argument");
    FPC_REQUIRE(B > 0 && L >= 0, "fpc_decode_features: bad shape B=%d L=%d", B, L);
    FPC_REQUIRE(p->d.fc == NDIM + 1, "fpc_decode_features: fc_units must be 18 (c0 + 17-dim VQ), got %d", p->d.fc);
    if (L == 0) return FPC_OK;
    if (const int se = status_error(p, "fpc_decode_features")) return se;  // an earlier launch on the handle failed
    hipStream_t st = static_cast<hipStream_t>(s);
    // the flag lives in the handle (no per-call hipMalloc/hipFree: hipFree synchronises the whole device); the
    // check itself needs this stream's result, so the call still ends with a sync of this one stream
    FPC_HIP(hipMemsetAsync(p->flag.p, 0, sizeof(int), st));
    SplitArgs sp;
    sp.n = 1;
    if (ws_wanted(p)) {
        WsArgs wa;
        const int rcw = ws_args(p, B, st, &wa);
        if (rcw != FPC_OK) return rcw;
        hipLaunchKernelGGL(k_decode_feat_ws, dim3(ws_grid(wa)), dim3(NT), 0, st, p->d, cb->d, pitch_dev, idx_dev, L, c_out_dev,
                           p->flag.as<int>(), wa);
        hipLaunchKernelGGL(k_decode_feat_df, dim3(B), dim3(NT), 0, st, p->d, cb->d, pitch_dev, idx_dev, L, c_out_dev,
                           p->flag.as<int>(), ws_fallback_args(p, wa));
    } else if (const int rc = split_args(p, B, st, &sp)) {
        return rc;
    } else
        hipLaunchKernelGGL(k_decode_feat_df, dim3(B * sp.n), dim3(NT), 0, st, p->d, cb->d, pitch_dev, idx_dev, L, c_out_dev,
                           p->flag.as<int>(), sp);
    FPC_HIP(hipGetLastError());
    int h = 0;
    FPC_HIP(hipMemcpyAsync(&h, p->flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    {
        const int rc2 = after_launch(p, st);
        if (rc2 != FPC_OK) return rc2;
    }
    FPC_HIP(hipStreamSynchronize(st));
    if (const int se = status_error(p, "fpc_decode_features")) return se;  // this launch's exchange gave up
    FPC_REQUIRE(h == 0, "fpc_decode_features: a symbol lies outside its codebook (corrupt stream or wrong codebooks)");
    return FPC_OK;
}

// ---- training ----
struct fpc_trainer {
    fpc_predictor* p = nullptr;
    int maxB = 0, maxL = 0, step = 0;
    fpc::DevBuf ws, grad[10], m[10], v[10], lossb, wt[3], gpart[5];  // wt: torch-layout copies of w2i, w2h, w1h
    fpc::DevBuf latch;  // device copy of the handle's status word, refreshed once per step, + the count of applied updates
    int seen_fail_epoch = 0;  // the handle's failure count when the step counter was last known to match the applied updates
    TrainBufs T;
    size_t sz[10];
};

static float* param_ptr(fpc_predictor* p, int k) { return p->buf[k].as<float>(); }

extern "C" int fpc_trainer_create(fpc_predictor* p, int max_B, int max_L, fpc_trainer** out) {
    FPC_REQUIRE(p && out, "fpc_trainer_create: null argument");
    FPC_REQUIRE(max_B > 0 && max_L > 1, "fpc_trainer_create: bad shape B=%d L=%d (L >= 2: the loss compares with the next frame)",
                max_B, max_L);
    FPC_REQUIRE(p->d.h1 % 4 == 0 && p->d.h2 % 4 == 0, "fpc_trainer_create: gru units must be multiples of 4");
    FPC_REQUIRE((size_t)max_L * p->d.fc * sizeof(float) <= 60 * 1024,
                "fpc_trainer_create: %d frames x %d outputs do not fit the loss kernel's LDS staging (<= 60 KB)", max_L, p->d.fc);
    std::unique_ptr<fpc_trainer> own(new fpc_trainer());  // freed on every early return below
    fpc_trainer* t = own.get();
    t->p = p;
    t->maxB = max_B;
    t->maxL = max_L;
    const int in = p->d.in, H1 = p->d.h1, H2 = p->d.h2, F = p->d.fc;
    const size_t N = (size_t)max_B * max_L;
    const size_t sz[10] = {(size_t)3 * H1 * in, (size_t)3 * H1 * H1, (size_t)3 * H1, (size_t)3 * H1, (size_t)3 * H2 * H1,
                           (size_t)3 * H2 * H2, (size_t)3 * H2, (size_t)3 * H2, (size_t)F * H2, (size_t)F};
    for (int k = 0; k < 10; ++k) {
        t->sz[k] = sz[k];
        FPC_HIP(t->grad[k].alloc(sz[k] * 4));
        FPC_HIP(t->m[k].alloc(sz[k] * 4));
        FPC_HIP(t->v[k].alloc(sz[k] * 4));
        FPC_HIP(hipMemset(t->m[k].p, 0, sz[k] * 4));
        FPC_HIP(hipMemset(t->v[k].p, 0, sz[k] * 4));
    }
    const size_t per = (size_t)6 * H1 + 7 * H2 + 2 * F + 6 * H1 + 6 * H2;
    FPC_HIP(t->ws.alloc(N * per * 4));
    FPC_HIP(t->lossb.alloc(sizeof(double) * (size_t)max_B));
    FPC_HIP(t->latch.alloc(2 * sizeof(unsigned)));
    FPC_HIP(hipMemset(t->latch.p, 0, 2 * sizeof(unsigned)));
    t->seen_fail_epoch = p->fail_epoch;
    {
        const int wi[5] = {0, 1, 4, 5, 8};
        for (int j = 0; j < 5; ++j) FPC_HIP(t->gpart[j].alloc(sz[wi[j]] * 4 * GSEG));
    }
    FPC_HIP(t->wt[0].alloc(sz[4] * 4));
    FPC_HIP(t->wt[1].alloc(sz[5] * 4));
    FPC_HIP(t->wt[2].alloc(sz[1] * 4));
    t->step = -1;  // the copies are filled by the first step
    float* q = t->ws.as<float>();
    auto take = [&](size_t n) {
        float* r = q;
        q += n;
        return r;
    };
    TrainBufs& T = t->T;
    T.h1p = take(N * H1), T.r1 = take(N * H1), T.z1 = take(N * H1), T.n1 = take(N * H1), T.hn1 = take(N * H1), T.h1 = take(N * H1);
    T.h2p = take(N * H2), T.r2 = take(N * H2), T.z2 = take(N * H2), T.n2 = take(N * H2), T.hn2 = take(N * H2), T.h2 = take(N * H2);
    T.relu = take(N * H2);
    T.th = take(N * F), T.dpre = take(N * F);
    T.dgi1 = take(N * 3 * H1), T.dgh1 = take(N * 3 * H1), T.dgi2 = take(N * 3 * H2), T.dgh2 = take(N * 3 * H2);
    T.lossb = t->lossb.as<double>();
    p->refs.fetch_add(1);  // the trainer keeps its predictor alive: fpc_predictor_destroy before fpc_trainer_destroy is safe
    *out = own.release();
    return FPC_OK;
}

extern "C" void fpc_trainer_destroy(fpc_trainer* t) {
    if (!t) return;
    predictor_unref(t->p);
    delete t;
}

extern "C" int fpc_trainer_step(fpc_trainer* t, const float* feat_dev, int B, int L, double lr, float* loss_host,
                                fpc_stream s) {
    FPC_REQUIRE(t && feat_dev, "fpc_trainer_step: null argument");
    FPC_REQUIRE(B > 0 && B <= t->maxB && L > 1 && L <= t->maxL, "fpc_trainer_step: shape B=%d L=%d outside the created %d x %d",
                B, L, t->maxB, t->maxL);
    hipStream_t st = static_cast<hipStream_t>(s);
    fpc_predictor* p = t->p;
    if (const int se = status_error(p, "fpc_trainer_step")) return se;  // an earlier launch on the handle failed
    if (p->fail_epoch != t->seen_fail_epoch && t->step >= 0) {
        // a failure has been reported and cleared since the last step: steps launched while the status word was set skipped
        // their update (k_adam), so the step count -- Adam's bias corrections -- continues from the updates actually applied
        unsigned applied = 0;
        FPC_HIP(hipMemcpy(&applied, t->latch.as<unsigned>() + 1, sizeof applied, hipMemcpyDeviceToHost));  // (synchronises)
        t->step = (int)applied;
        t->seen_fail_epoch = p->fail_epoch;
    }
    const PredDev& P = p->d;
    const int in = P.in, H1 = P.h1, H2 = P.h2, F = P.fc;
    const int N = B * L;
    const TrainBufs& T = t->T;  // laid out for maxB x maxL; rows are addressed by n = b*L + t < maxB*maxL
    auto refresh = [&]() {  // torch-layout copies for the backward pass
        const int kk[3] = {4, 5, 1}, Kd[3] = {H1, H2, H1}, Rd[3] = {3 * H2, 3 * H2, 3 * H1};
        for (int c = 0; c < 3; ++c)
            hipLaunchKernelGGL(k_transpose, dim3((unsigned)((t->sz[kk[c]] + 255) / 256)), dim3(256), 0, st, param_ptr(p, kk[c]),
                               Kd[c], Rd[c], t->wt[c].as<float>());
    };
    if (t->step < 0) {
        refresh();
        t->step = 0;
    }
    SplitArgs sp;  // backward (and the row-split forward) run an utterance on 2-8 workgroups while the batch leaves CUs idle
    if (ws_wanted(p)) {
        // the forward on the weights-stationary kernel (predictor_ws.h): 16 utterances a group, the kept activations stored
        // by the workgroup that evaluates them -- the same values as k_train_fwd's, bit for bit (tests)
        WsArgs wa;
        const int rc = ws_args(p, B, st, &wa);
        if (rc != FPC_OK) return rc;
        const WsSave sv{T.h1p, T.r1, T.z1, T.n1, T.hn1, T.h1, T.h2p, T.r2, T.z2, T.n2, T.hn2, T.h2, T.relu, T.th};
        hipLaunchKernelGGL(k_forward_ws<true>, dim3(ws_grid(wa)), dim3(NT), 0, st, P, feat_dev, L, nullptr, nullptr, nullptr, wa, sv);
        hipLaunchKernelGGL(k_train_fwd_df, dim3(B), dim3(NT), 0, st, P, feat_dev, L, T, ws_fallback_args(p, wa));
    } else {
        const int rc = split_args(p, B, st, &sp);
        if (rc != FPC_OK) return rc;
        hipLaunchKernelGGL(k_train_fwd_df, dim3(B * sp.n), dim3(NT), 0, st, P, feat_dev, L, T, sp);
    }
    const double cnt = (double)B * (L - 1) * F;
    hipLaunchKernelGGL(k_train_loss, dim3(B), dim3(256), 0, st, feat_dev, in, F, L,
                       (float)(2.0 / cnt), T);
    const BwdW bw{t->wt[0].as<float>(), t->wt[1].as<float>(), t->wt[2].as<float>()};
    if (ws_wanted(p) && !fpc::env_is_one("FPC_TRAIN_BWD_ROWSPLIT")) {
        // back-propagation on the weights-stationary kernel (predictor_bwd_ws.h); behind it, for the groups whose workgroups
        // could not all become resident, the row-split kernel with one workgroup per utterance
        WsArgs wb;
        const int rc = ws_args(p, B, st, &wb, BGRANULES);  // fresh granules (zeroed behind the forward launches on the stream)
        if (rc != FPC_OK) return rc;
        if (const char* wh = getenv("FPC_TEST_WITHHOLD_PUBLISH")) wb.withhold = wh[0] == 'b' ? 1 : wb.withhold;  // test hook: this kernel alone
        hipLaunchKernelGGL(k_train_bwd_ws, dim3(ws_grid(wb)), dim3(NT), 0, st, P, bw, L, T, wb);
#ifdef FPC_WS_PROF
        {
            (void)hipStreamSynchronize(st);
            volatile unsigned* w = (volatile unsigned*)p->status_host;
            fprintf(stderr, FPC_BW_STAMP_TID < 256
                    ? "k_train_bwd_ws B=%d cycles/step (thread %d of workgroup 5, group 0; track A): hop %u image sync %u chain of 72 %u "
                      "product sync %u wait for track B %u gates %u | before the loop %u kcycles, loop %u kcycles, kernel %u us\n"
                    : "k_train_bwd_ws B=%d cycles/step (thread %d of workgroup 5, group 0; track B): gates %u poll %u image writes + sync %u "
                      "stage + wait for track A %u chains of 24 %u product sync %u | before the loop %u kcycles, loop %u kcycles, kernel %u us\n",
                    B, FPC_BW_STAMP_TID, w[44], w[45], w[46], w[47], w[48], w[49], w[50], w[51], w[52]);
            for (int k2 = 40; k2 < 56; ++k2) w[k2] = 0;
        }
#endif
        hipLaunchKernelGGL(k_train_bwd, dim3(B), dim3(NT), 0, st, P, bw, L, T, ws_fallback_args(p, wb));
    } else {
        const int rc = split_args(p, B, st, &sp);  // fresh granules (zeroed behind the forward launch on the stream)
        if (rc != FPC_OK) return rc;
        hipLaunchKernelGGL(k_train_bwd, dim3(B * sp.n), dim3(NT), 0, st, P, bw, L, T, sp);
    }
    struct G {
        const float* A;
        int K;
        const float* D;
        int R;
        int w, b;
    } gs[5] = {{feat_dev, in, T.dgi1, 3 * H1, 0, 2}, {T.h1p, H1, T.dgh1, 3 * H1, 1, 3}, {T.h1, H1, T.dgi2, 3 * H2, 4, 6},
               {T.h2p, H2, T.dgh2, 3 * H2, 5, 7}, {T.relu, H2, T.dpre, F, 8, 9}};
    GradJobs J;
    int maxR = 0, maxK = 0;
    for (int j = 0; j < 5; ++j) {
        const G& g = gs[j];
        J.A[j] = g.A, J.D[j] = g.D, J.C[j] = t->grad[g.w].as<float>(), J.bsum[j] = t->grad[g.b].as<float>();
        J.K[j] = g.K, J.R[j] = g.R;
        J.part[j] = t->gpart[j].as<float>();
        maxR = g.R > maxR ? g.R : maxR;
        maxK = g.K > maxK ? g.K : maxK;
    }
    const int seglen = ((N + 4 * GSEG - 1) / (4 * GSEG)) * 4;
    GradMap M;
    M.nslots = 0, M.ncol = 0;
    {
        int order[5] = {0, 1, 2, 3, 4};  // the jobs with the most tiles first (they are what a late start would leave running alone)
        auto tiles = [&](int j) { return ((J.K[j] + 16 * GKT - 1) / (16 * GKT)) * ((J.R[j] + 255) / 256); };
        std::sort(order, order + 5, [&](int a, int b) { return tiles(a) > tiles(b); });
        for (int o = 0; o < 5; ++o) {
            const int j = order[o];
            for (int y = 0; y < (J.K[j] + 16 * GKT - 1) / (16 * GKT); ++y)
                for (int x = 0; x < (J.R[j] + 255) / 256; ++x) {
                    if (M.nslots >= 128) {
                        fpc::set_error("fpc_trainer_step: more than 128 gradient tiles per sample segment");
                        return FPC_ERR_CAPACITY;
                    }
                    M.job[M.nslots] = (unsigned char)j, M.bx[M.nslots] = (unsigned char)x, M.by[M.nslots] = (unsigned char)y;
                    ++M.nslots;
                }
        }
        for (int j = 0; j < 5; ++j)
            for (int x = 0; x < (J.R[j] + 255) / 256; ++x) {
                if (M.ncol >= 32) {
                    fpc::set_error("fpc_trainer_step: more than 32 bias-sum workgroups");
                    return FPC_ERR_CAPACITY;
                }
                M.cjob[M.ncol] = (unsigned char)j, M.cbx[M.ncol] = (unsigned char)x, ++M.ncol;
            }
    }
    hipLaunchKernelGGL(k_grad_tn, dim3(GSEG * M.nslots + M.ncol + (B + 3) / 4), dim3(256), 0, st, J, N, seglen, M, L, T);
    hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)(((size_t)maxK * maxR + 255) / 256), 5), dim3(256), 0, st, J, p->status_dev,
                       t->latch.as<unsigned>());
    t->step += 1;
    const double bc1 = 1.0 - pow(0.9, t->step), bc2 = 1.0 - pow(0.999, t->step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    {
        AdamJobs A;
        const int kk[3] = {4, 5, 1}, Kd[3] = {H1, H2, H1}, Rd[3] = {3 * H2, 3 * H2, 3 * H1};  // (as refresh())
        unsigned blocks = 0;
        for (int k = 0; k < 10; ++k) {
            A.p[k] = param_ptr(p, k), A.m[k] = t->m[k].as<float>(), A.v[k] = t->v[k].as<float>(), A.g[k] = t->grad[k].as<float>();
            A.tp[k] = nullptr, A.tK[k] = 1, A.tR[k] = 1;
            A.n[k] = (unsigned)t->sz[k];
            A.blk0[k] = blocks;
            blocks += (unsigned)((t->sz[k] + 255) / 256);
        }
        A.blk0[10] = blocks;
        for (int c = 0; c < 3; ++c) A.tp[kk[c]] = t->wt[c].as<float>(), A.tK[kk[c]] = Kd[c], A.tR[kk[c]] = Rd[c];
        hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, st, A, step_size, bc2_sqrt, t->latch.as<unsigned>());
    }
    FPC_HIP(hipGetLastError());
    {
        const int rc = after_launch(p, st);
        if (rc != FPC_OK) return rc;
    }
    if (loss_host) {
        std::vector<double> lb(B);
        FPC_HIP(hipMemcpyAsync(lb.data(), T.lossb, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, st));
        FPC_HIP(hipStreamSynchronize(st));
        if (const int se = status_error(p, "fpc_trainer_step")) return se;  // this step gave up: no Adam update was made
        double loss = 0.0;
        for (int b = 0; b < B; ++b) loss += lb[b];
        *loss_host = (float)(loss / cnt);
    }
    return FPC_OK;
}

// what = 0: parameters, 1: gradients of the last step; host arrays in torch layouts (fpc_predictor_weights)
extern "C" int fpc_trainer_export(fpc_trainer* t, int what, const fpc_predictor_weights* out_host) {
    FPC_REQUIRE(t && out_host && (what == 0 || what == 1), "fpc_trainer_export: bad argument");
    const int in = t->p->d.in, H1 = t->p->d.h1, H2 = t->p->d.h2, F = t->p->d.fc;
    float* dst[10] = {(float*)out_host->rnn1_weight_ih, (float*)out_host->rnn1_weight_hh, (float*)out_host->rnn1_bias_ih,
                      (float*)out_host->rnn1_bias_hh,   (float*)out_host->rnn2_weight_ih, (float*)out_host->rnn2_weight_hh,
                      (float*)out_host->rnn2_bias_ih,   (float*)out_host->rnn2_bias_hh,   (float*)out_host->fc_weight,
                      (float*)out_host->fc_bias};
    const int rows[10] = {3 * H1, 3 * H1, 0, 0, 3 * H2, 3 * H2, 0, 0, F, 0};   // torch rows of the matrices
    const int cols[10] = {in, H1, 0, 0, H1, H2, 0, 0, H2, 0};
    FPC_HIP(hipDeviceSynchronize());
    if (const int se = status_error(t->p, "fpc_trainer_export")) return se;  // a step on the handle gave up
    for (int k = 0; k < 10; ++k) {
        FPC_REQUIRE(dst[k], "fpc_trainer_export: null output pointer %d", k);
        std::vector<float> h(t->sz[k]);
        const void* src = what == 0 ? (const void*)param_ptr(t->p, k) : t->grad[k].p;
        FPC_HIP(hipMemcpy(h.data(), src, t->sz[k] * 4, hipMemcpyDeviceToHost));
        if (rows[k] == 0) {
            memcpy(dst[k], h.data(), t->sz[k] * 4);
        } else {  // device layout [cols][rows] -> torch [rows][cols]
            for (int r = 0; r < rows[k]; ++r)
                for (int c = 0; c < cols[k]; ++c) dst[k][(size_t)r * cols[k] + c] = h[(size_t)c * rows[k] + r];
        }
    }
    return FPC_OK;
}

extern "C" int fpc_vq_quantize(const fpc_codebooks* cb, int which, const float* r_dev, int n,
                               double* qr_dev, int32_t* idx_dev, fpc_stream s) {
    FPC_REQUIRE(cb, "fpc_vq_quantize: null codebook handle");
    FPC_REQUIRE(which == 0 || (which == 1 && cb->d.vq_lo), "fpc_vq_quantize: codebook %d not loaded", which);
    if (n <= 0) return FPC_OK;  // empty input: nothing to do (pointers may be null)
    FPC_REQUIRE(r_dev && qr_dev, "fpc_vq_quantize: null argument");
    hipLaunchKernelGGL(k_vq, dim3(n), dim3(NT), 0, static_cast<hipStream_t>(s), cb->d, which, r_dev, qr_dev,
                       idx_dev);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}

extern "C" int fpc_scl_quantize(const fpc_codebooks* cb, int which, const float* x_dev, int n,
                                double* q_dev, int32_t* idx_dev, fpc_stream s) {
    FPC_REQUIRE(cb, "fpc_scl_quantize: null codebook handle");
    FPC_REQUIRE(which == 0 || (which == 1 && cb->d.scl_lo), "fpc_scl_quantize: codebook %d not loaded", which);
    if (n <= 0) return FPC_OK;
    FPC_REQUIRE(x_dev && q_dev, "fpc_scl_quantize: null argument");
    hipLaunchKernelGGL(k_scl, dim3(n), dim3(NT), 0, static_cast<hipStream_t>(s), cb->d, which, x_dev, q_dev,
                       idx_dev);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}
