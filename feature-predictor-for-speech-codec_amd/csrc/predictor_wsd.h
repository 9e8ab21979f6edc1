// predictor_wsd.h -- the encoder's frame tail DISTRIBUTED over the 32 workgroups of a group (included after predictor_ws.h).
//
// With the tail on the utterance's own workgroups (round 4's owner + helper pairs, since removed) a frame of the group lasts
// as long as its slowest utterance, and one of 16 is almost always above the threshold: 2 x (1 024 + 5 x 1 024 entries x 51
// float64 operations) on two CUs while 30 others wait, plus 16 x 139 kB of codebook per stage through the XCD's L2 -- 24.7k
// cycles per frame (profiles/r04_ablations.txt).  Here every workgroup keeps 1/32 of every codebook in LDS for the launch (entries
// 32 m + slice: 32 per book, 13.8 kB) and serves EVERY utterance of the group: the TARGETS move, not the codebooks.
//   thread (u = tid >> 5, m = tid & 31): utterance u of the group, entry m of this workgroup's slice -- or, when results
//   come back, the list of workgroup m for utterance u: a half-wave per utterance throughout.
//   1. output layer, residuals and thresholds of all 16 utterances on every workgroup (an MFMA tile has them anyway);
//   2. first stage: one distance per thread, the half-wave's five best (one for a 1-stage search) -> 16-byte granules
//      {epoch, distance, index} [utterance][workgroup][5];
//   3. every workgroup gathers every list (thread (u, m) polls workgroup m's list for utterance u) and takes the five
//      smallest of the 32 sorted lists -- five half-wave arg-min rounds with the winner's list popped -- so all 32 know
//      the survivors without another hop;
//   4. second stage: (utterance, survivor, entry) triples dealt over the threads, 512 per round, the half-wave's best
//      per (utterance, survivor) -> granules; gathered and reduced the same way;
//   5. every workgroup forms every utterance's quantized residual and next input row itself (no third hop); workgroup u
//      stores utterance u's outputs.
// Every choice is a minimum over (float64 distance in numpy's pairwise order, index) exactly as vq_func.py:10-24,110-125
// makes it: results are bit-identical to encode_frame (the row-split kernels' tail) and to the oracle (tests).

// (distance, index) minimum over each HALF of the wave (lanes 0-31, 32-63), ties to the lower index; every lane gets its
// half's.  The distance: one float64 pass (four steps inside the 16-lane rows, one row broadcast).  The index: where exactly
// one lane of a half holds the smallest distance -- the rule: equal float64 distances of different entries are rare -- it is
// read from that lane (a ballot and two scalar lane reads); otherwise the smallest index among the holders by a second pass.
// (Three 32-bit passes over the distance's words instead of the float64 pass were measured slower: the passes are chains of
// dependent DPP steps, and there would be three of them.)
// Fast path first: distances are >= +0, so their bit patterns order like the values and the HIGH WORDS order them weakly:
// one pass of 32-bit minima (one DPP-fused instruction a step), and where exactly one lane of each half holds the smallest
// high word that lane holds the strictly smallest distance -- its (distance, index) is read out by lane number.  Two
// distances that agree in sign, exponent and 20 mantissa bits (or a half that is all +inf) take the float64 passes below.
__device__ __forceinline__ void hw_argmin(double& d, int& i, int lane) {
    {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
        const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
        unsigned h = hi;
        h = min(h, (unsigned)dpp_mov_i32<0xB1, 0xf>((int)h));
        h = min(h, (unsigned)dpp_mov_i32<0x4E, 0xf>((int)h));
        h = min(h, (unsigned)dpp_mov_i32<0x141, 0xf>((int)h));
        h = min(h, (unsigned)dpp_mov_i32<0x140, 0xf>((int)h));
        h = min(h, (unsigned)dpp_mov_i32<0x142, 0xa>((int)h));
        const unsigned h0 = (unsigned)__builtin_amdgcn_readlane((int)h, 31), h1 = (unsigned)__builtin_amdgcn_readlane((int)h, 63);
        const bool up = lane >= 32;
        const unsigned long long bal = __ballot(hi == (up ? h1 : h0));
        const unsigned b0 = (unsigned)bal, b1 = (unsigned)(bal >> 32);
        if (__popc(b0) <= 1 && __popc(b1) <= 1) {  // (wave-uniform)
            const int l0 = b0 != 0u ? __ffs((int)b0) - 1 : 0, l1 = 32 + (b1 != 0u ? __ffs((int)b1) - 1 : 0);
            const unsigned lo0 = (unsigned)__builtin_amdgcn_readlane((int)lo, l0), hi0 = (unsigned)__builtin_amdgcn_readlane((int)hi, l0);
            const unsigned lo1 = (unsigned)__builtin_amdgcn_readlane((int)lo, l1), hi1 = (unsigned)__builtin_amdgcn_readlane((int)hi, l1);
            const int i0 = __builtin_amdgcn_readlane(i, l0), i1 = __builtin_amdgcn_readlane(i, l1);
            d = __longlong_as_double((long long)(((unsigned long long)(up ? hi1 : hi0) << 32) | (up ? lo1 : lo0)));
            i = up ? i1 : i0;
            return;
        }
    }
    double m = d;
    m = min_f64(m, dpp_mov_f64<0xB1, 0xf, false>(m));   // quad_perm [1,0,3,2]
    m = min_f64(m, dpp_mov_f64<0x4E, 0xf, false>(m));   // quad_perm [2,3,0,1]
    m = min_f64(m, dpp_mov_f64<0x141, 0xf, false>(m));  // row_half_mirror
    m = min_f64(m, dpp_mov_f64<0x140, 0xf, false>(m));  // row_mirror: every lane holds its row's minimum
    m = min_f64(m, dpp_mov_f64<0x142, 0xa, false>(m));  // row_bcast:15 -> rows 1, 3: the half's minimum
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    const unsigned lo0 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 31), hi0 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 31);
    const unsigned lo1 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi1 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    const bool up = lane >= 32;
    const double dmin = __longlong_as_double((long long)(((unsigned long long)(up ? hi1 : hi0) << 32) | (up ? lo1 : lo0)));
    int c = d == dmin ? i : 0x7fffffff;
    c = min(c, dpp_mov_i32<0xB1, 0xf>(c));
    c = min(c, dpp_mov_i32<0x4E, 0xf>(c));
    c = min(c, dpp_mov_i32<0x141, 0xf>(c));
    c = min(c, dpp_mov_i32<0x140, 0xf>(c));
    c = min(c, dpp_mov_i32<0x142, 0xa>(c));
    const int i0 = __builtin_amdgcn_readlane(c, 31), i1 = __builtin_amdgcn_readlane(c, 63);
    const int r = up ? i1 : i0;
    d = dmin;
    i = r;
}
// float64 squared distance, numpy's pairwise order (vq_func.py:18), both operands in LDS
__device__ __forceinline__ double wsd_dist(const double* x, const double* c) {
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
__device__ __forceinline__ void wsd_put(const WsCtx& X, int g, unsigned epoch, double d, int ix) {
    if (X.withhold) return;
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    const u32x4 w = {epoch, (unsigned)b, (unsigned)(b >> 32), (unsigned)ix};
    if (X.fast)
        __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, g * 16, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(w, X.rs, g * 16, 0, 16);
}
// this thread's list of n <= 5 results (granules base .. base + n - 1); a list that never arrives reads as "no entry"
__device__ __forceinline__ void wsd_get(const WsCtx& X, WsLds& L, int base, int n, unsigned epoch, double (&d)[SURV], int (&ix)[SURV]) {
    int gi[SURV];
#pragma unroll
    for (int k = 0; k < SURV; ++k) gi[k] = k < n ? base + k : -1;
    u32x4 v[SURV];
    const bool ok = ws_poll<SURV>(X, L, gi, epoch, v);
#pragma unroll
    for (int k = 0; k < SURV; ++k) {
        const bool have = ok && k < n && v[k].x == epoch;
        d[k] = have ? __longlong_as_double((long long)(((unsigned long long)v[k].z << 32) | v[k].y)) : INFINITY;
        ix[k] = have ? (int)v[k].w : 0x7fffffff;
    }
}
__device__ __forceinline__ int wsd_clamp(int i, int n) { return i < 0 ? 0 : (i < n ? i : n - 1); }

// output layer of ALL 16 utterances: rows 0-15 as one MFMA tile per input segment (foreground wave fw: segments 2 fw, 2 fw + 1),
// rows 16, 17 as fmaf chains (32 (row, utterance) pairs x 8 segments on the 256 foreground threads)
__device__ __forceinline__ void wsd_F(WsLds& L, const WsRegs& R, int fw, int lane, int ft) {
    const int c = lane & 15, q = lane >> 4;
    float hv[2][4], wv[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int sg = 2 * fw + s2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h = L.h2[(16 * sg + 4 * j) * WG + lane];
            hv[s2][j] = h > 0.0f ? h : 0.0f;
            wv[s2][j] = L.fcw[(16 * sg + 4 * j + q) * 16 + c];
        }
    }
    const int rsg = ft >> 5, row = 16 + ((ft >> 4) & 1), ru = ft & 15;  // item ft: (segment, row 16 | 17, utterance)
    float rh[16], rw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rh[k] = L.h2[(16 * rsg + k) * WG + ru];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4ws w4 = *reinterpret_cast<const f32x4ws*>(&L.fcw2[(row - 16) * WH2 + 16 * rsg + 4 * i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) rw[4 * i + k] = w4[k];
    }
    // The tile TRANSPOSED -- weights as the A operand, states as B (the two operand layouts are the same: lane = (index, k)),
    // same products in the same k order, same bits -- so that lane (c, q) ends up with rows 4 q .. 4 q + 3 of UTTERANCE c:
    // one 16-byte store per lane to [utterance][row], where the tail's half-wave (lane m = row m of its utterance) reads 18
    // consecutive floats.  (Rows of 16 utterances at a 16-float pitch put one utterance's 18 rows into two banks: a 9-way
    // conflict on each of the tail's eight reads, by all eight waves at once -- 1.1k cycles of the LDS pipe per frame.)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int sg = 2 * fw + s2;
        f32x4ws a0 = {0.f, 0.f, 0.f, 0.f};
        if (sg == 0) a0 = *reinterpret_cast<const f32x4ws*>(&L.fcb[4 * q]);
#pragma unroll
        for (int j = 0; j < 4; ++j) a0 = ws_mfma(wv[s2][j], hv[s2][j], a0);
        *reinterpret_cast<f32x4ws*>(&L.pFa[sg][c * WPF + 4 * q]) = a0;  // utterance c, rows 4 q .. 4 q + 3
    }
    float a = rsg == 0 ? (row == 16 ? R.bF16[0] : R.bF16[1]) : 0.0f;  // (no indexed access: the struct stays in registers)
#pragma unroll
    for (int k = 0; k < 16; ++k) a = fmaf(rh[k] > 0.0f ? rh[k] : 0.0f, rw[k], a);
    L.pFa[rsg][ru * WPF + row] = a;
}

// the frame's tail for all utterances of the group (all 512 threads); `fv`: feat[u][frame][m] of thread (u, m < 20);
// `mk`: the frame's two mask values of utterance u (mask mode, wavernn.py:209-211)
__device__ __forceinline__ void wsd_tail(const WsCtx& X, WsLds& L, const CbDev& C, const EncArgs& A, unsigned* err, int frame,
                                         float fv, float2 mk, int tid0, unsigned epoch) {
    const int tid = tid0 + ws_opaque_zero(), lane = tid & 63, u = tid >> 5, m = tid & 31;
    const bool uv = u < X.nu;
    // ---- prediction, residual, first-stage target of utterance u: lane m < 18 = row m (wavernn.py:195-196).  Everything an
    // utterance's half-wave reads from here on it has written itself (one in-order LDS queue per wave): no barrier.
    if (m < WFC) {
        const int o = u * WPF + m;
        const float acc = ((L.pFa[0][o] + L.pFa[1][o]) + (L.pFa[2][o] + L.pFa[3][o])) +
                          ((L.pFa[4][o] + L.pFa[5][o]) + (L.pFa[6][o] + L.pFa[7][o]));
        const float tt = fpc_tanhf(acc);
        const float f = tt + tt;  // the "dual" FC is the same Linear summed twice (wavernn.py:89-92)
        const float r = fv - f;
        L.fo[u][m] = f;
        L.rsa[u][m] = r;
        if (m >= 1) L.xs[u][m - 1] = (double)r;
    } else if (m < WIN) {  // the pitch columns pass through (wavernn.py:178; stored with the other outputs: a store here
        L.x[ws_xi(m, u)] = fv;  // would stand in front of the next wait for a load)
    }
    WSTAMP(25)
    // ---- thresholds (:202,:206): every lane of the utterance's half-wave evaluates the same values ----
    float sabs = 0.0f;
    for (int d = 1; d < WFC; ++d) sabs += fabsf(L.rsa[u][d]);
    const float r0 = L.rsa[u][0];
    const bool masked = A.mask != nullptr;  // (`if ind1[k, 0]`, :218: any non-zero value)
    const int i1 = masked ? (mk.x != 0.0f) : (fabsf(r0) > A.l1), i2 = masked ? (mk.y != 0.0f) : (sabs > A.l2);
    const bool nonfinite = !(fabsf(r0) <= 3.0e38f) || !(sabs <= 3.0e38f);
    const bool live = uv && A.qtz && !nonfinite;
    const bool do_scl = live && (i1 || C.scl_lo), do_vq = live && (i2 || C.vq_lo);
    const bool two = do_vq && i2 && C.S_hi == 2;
    const int cb0 = i2 ? 0 : 2, N0 = i2 ? C.N_hi0 : C.N_lo;
    const double* cb0R = i2 ? C.vq_hi0_r : C.vq_lo_r;

    if (uv && nonfinite && A.qtz && m == 0 && X.slice == u) status_or(err, FPC_ST_NONFINITE);
    // ---- first stage: this workgroup's 32 entries against utterance u ----
    const int e = WNS * m + X.slice;
    if (do_vq) {
        const double d0 = e < N0 ? wsd_dist(L.xs[u], L.cbs[cb0][m]) : INFINITY;
        const int e0 = e < N0 ? e : 0x7fffffff;
        const int g1 = WOFF_G1 + (u * WNS + X.slice) * SURV;
        if (!two) {  // the nearest entry is all a 1-stage search returns (vq_func.py:93-95)
            double d = d0;
            int ix = e0;
            hw_argmin(d, ix, lane);
            if (m == 0) wsd_put(X, g1, epoch, d, ix);
        } else {  // the half-wave's five best, ranked by counting (ties: the lower entry first; five arg-min rounds with the
            // winner struck out were measured slower: 4.65 against 4.50 ms at 128 x 300)
            // First on the distances' high words (they order the distances weakly: hw_argmin): rank = lanes with a smaller
            // high word, exact for every lane whose high word no other lane shares -- and two lanes that share one get the
            // same count, so five distinct ranks 0..4 in the half prove the five best exact.  Otherwise the float64 count.
            const unsigned hw = (unsigned)((unsigned long long)__double_as_longlong(d0) >> 32);
            L.dl[u][m] = d0;  // (read back by the same wave: one in-order LDS queue per wave)
            L.dh[u][m] = hw;
            int rank = 0;
#pragma unroll
            for (int j4 = 0; j4 < WNS / 4; ++j4) {
                const uint4 hj = reinterpret_cast<const uint4*>(L.dh[u])[j4];
                rank += (int)(hj.x < hw) + (int)(hj.y < hw) + (int)(hj.z < hw) + (int)(hj.w < hw);
            }
            bool sure = true;
#pragma unroll
            for (int r = 0; r < SURV; ++r) {
                const unsigned long long bal = __ballot(rank == r);
                sure &= __popc((unsigned)bal) <= 1 && __popc((unsigned)(bal >> 32)) <= 1;
            }
            if (!sure) {  // (wave-uniform)
                rank = 0;
#pragma unroll 8
                for (int j = 0; j < WNS; ++j) {
                    const double dj = L.dl[u][j];
                    rank += (dj < d0) | ((dj == d0) & (j < m));
                }
            }
            if (rank < SURV) wsd_put(X, g1 + rank, epoch, d0, e0);
        }
    }
    WSTAMP(26)
    // scalar quantizer of every utterance (scl_quantize, vq_func.py:167-185): codes in LDS, a half-wave per utterance
    float rq0 = 0.0f;
    int ix0 = -1;
    if (do_scl) {
        const int off = i1 ? 0 : C.n_hi, n = i1 ? C.n_hi : C.n_lo;
        double bd = INFINITY;
        int bi = 0x7fffffff;
        const double v = (double)r0;
        for (int c = m; c < n; c += WNS) {
            const double df = v - L.sclc[off + c];
            const double d = df * df;
            if (d < bd) {
                bd = d;
                bi = c;
            }
        }
        hw_argmin(bd, bi, lane);
        bi = wsd_clamp(bi, n);
        rq0 = (float)L.sclc[off + bi];
        ix0 = bi + (i1 ? 0 : C.n_hi);
    }
    WSTAMP(27)
    // ---- every workgroup's list for utterance u comes in (thread = list of workgroup m) ----
    double ld[SURV];
    int li[SURV];
    wsd_get(X, L, WOFF_G1 + (u * WNS + m) * SURV, do_vq ? (two ? SURV : 1) : 0, epoch, ld, li);
    WSTAMP(28)
    int s[SURV] = {0, 0, 0, 0, 0};  // the survivors (1-stage: s[0] = the nearest entry)
    if (do_vq) {
        if (!two) {
            double d = ld[0];
            int ix = li[0];
            hw_argmin(d, ix, lane);
            s[0] = wsd_clamp(ix, N0);
        } else {  // the five smallest of 32 sorted lists: five rounds of arg-min over the lists' heads, the winner's list popped
            // (a survivor's first-stage row -- lane m < 17: element m -- is asked for as soon as the survivor is known: the
            //  L2 round trips ride under the remaining rounds)
            double row[SURV];
#pragma unroll
            for (int r = 0; r < SURV; ++r) {
                double d = ld[0];
                int ix = li[0];
                hw_argmin(d, ix, lane);
                if (ix == li[0] && ix != 0x7fffffff) {
#pragma unroll
                    for (int k = 0; k + 1 < SURV; ++k) {
                        ld[k] = ld[k + 1];
                        li[k] = li[k + 1];
                    }
                    ld[SURV - 1] = INFINITY;
                    li[SURV - 1] = 0x7fffffff;
                }
                s[r] = wsd_clamp(ix, N0);
                row[r] = m < NDIM ? cb0R[(size_t)s[r] * NDIM + m] : 0.0;
            }
            // second-stage targets: residual of every survivor (vq_func.py:103-108)
            if (m < NDIM) {
                const double xv = L.xs[u][m];
#pragma unroll
                for (int k = 0; k < SURV; ++k) L.xq2[u][k][m] = xv - row[k];
            }
        }
    }
    WSTAMP(29)
    // ---- second stage.  The reference keeps, per survivor k, the entry with the smallest total error (ties: the lower index)
    // and lets a later survivor win only with a strictly smaller error (vq_func.py:110-125): the winner is the minimum over
    // (error, k, index).  So this thread runs its entry against all five targets of utterance u, keeps its best by that order,
    // the half-wave reduces once, and ONE granule per utterance and workgroup goes out (key = k * 2048 + index).  (Dealing
    // the (utterance, survivor) pairs over all 16 half-waves instead -- one distance per thread and round -- needs two
    // workgroup barriers more and was measured equal: 4.52 ms both.)
    int w1 = 0, bk = 0;
    if (two) {
        double bd = INFINITY;
        int bkey = 0x7fffffff;
        if (e < C.N_hi1) {
            double c2[NDIM + 1];  // (this thread's entry stays in registers across the five targets)
#pragma unroll
            for (int j = 0; j < NDIM + 1; ++j) c2[j] = L.cbs[1][m][j];
#pragma unroll 1
            for (int k = 0; k < SURV; ++k) {
                const double d = ws_dist(L.xq2[u][k], c2);
                if (d < bd) {  // (ascending k: strict < keeps the earlier survivor)
                    bd = d;
                    bkey = k * 2048 + e;
                }
            }
        }
        hw_argmin(bd, bkey, lane);
        if (m == 0) wsd_put(X, WOFF_G2 + (u * WNS + X.slice) * SURV, epoch, bd, bkey);
    }
    WSTAMP(30)
    double qd[SURV];
    int qi[SURV];
    wsd_get(X, L, WOFF_G2 + (u * WNS + m) * SURV, two ? 1 : 0, epoch, qd, qi);
    if (two) {
        double d = qd[0];
        int key = qi[0];
        hw_argmin(d, key, lane);
        bk = (key >> 11) & 7;
        bk = bk < SURV ? bk : 0;
        w1 = wsd_clamp(key & 2047, C.N_hi1);
    }
    WSTAMP(31)
    // ---- quantized residual, next input row (:242 / :244-252), outputs ----
    const int sb = s[0] * (bk == 0) + s[1] * (bk == 1) + s[2] * (bk == 2) + s[3] * (bk == 3) + s[4] * (bk == 4);
    float rq = 0.0f;  // lane m = row: r_qtz[row]
    if (m == 0) rq = rq0;
    if (do_vq && m >= 1 && m < WFC) {
        const int d = m - 1;
        const double q = two ? cb0R[(size_t)sb * NDIM + d] + C.vq_hi1_r[(size_t)w1 * NDIM + d] : cb0R[(size_t)s[0] * NDIM + d];
        rq = (float)q;
    }
    const bool owner = X.slice == u && uv;
    const size_t fi = (size_t)(X.b0 + u) * A.Lf + frame;
    if (m < WFC) {
        const float rs = L.rsa[u][m], f = L.fo[u][m];
        const int ind = m == 0 ? i1 : i2;
        float rv, ru, cn;
        if (A.qtz) {
            rv = rs;  // un-thresholded residual (:197)
            ru = 0.0f;
            cn = f + rq;
        } else {  // (:244-252; mask mode: the products with the mask's own values)
            const float mv = m == 0 ? mk.x : mk.y;
            ru = rs * (masked ? 1.0f - mv : (float)(1 - ind));
            rv = rs * (masked ? mv : (float)ind);
            cn = f + rv;
        }
        L.x[ws_xi(m, u)] = uv ? cn : 0.0f;
        if (owner) {
            A.r[fi * WFC + m] = rv;
            A.r_qtz[fi * WFC + m] = rq;
            A.r_under[fi * WFC + m] = ru;
            A.c_in[fi * WIN + m] = cn;
        }
    }
    if (owner && m >= WFC && m < WIN) A.c_in[fi * WIN + m] = fv;
    if (owner && m == 0) {
        A.ind1[fi] = masked ? 0.0f : (float)i1;  // (the reference fills the indicator outputs from the thresholds only)
        A.ind2[fi] = masked ? 0.0f : (float)i2;
        int o0 = ix0, o1 = -1, o2 = -1, o3 = -1;
        if (do_vq) {
            if (i2) {
                o1 = two ? sb : s[0];
                o2 = two ? w1 : -1;
            } else {
                o3 = s[0];
            }
        }
        if (nonfinite) o0 = o1 = o2 = o3 = -2;
        *reinterpret_cast<int4*>(&A.idx[fi * 4]) = make_int4(o0, o1, o2, o3);
    }
}

// The RECEIVER's frame tail (k_decode_feat_ws): thread (u, m) = row m of utterance u.  Prediction from the output layer's segment
// sums (wsd_F), quantized residual looked up from the frame's four symbols exactly as decode_frame does (scalar code | first +
// second stage of the upper books | the lower book; -1: none; a symbol outside its book raises `bad`, reported by the workgroup
// that owns the utterance), next input row into the image; the owner stores the row.
__device__ __forceinline__ void wsd_receive_tail(const WsCtx& X, WsLds& L, const PredDev& P, const CbDev& C, const int* __restrict__ idx,
                                                 float* __restrict__ c_out, int* bad, int Lf, int frame, float pv, int tid) {
    const int u = tid >> 5, m = tid & 31;
    const bool uv = u < X.nu, owner = uv && X.slice == u;
    const size_t fi = (size_t)(X.b0 + (uv ? u : 0)) * Lf + frame;
    if (m < WFC) {
        const int o = u * WPF + m;
        const float acc = ((L.pFa[0][o] + L.pFa[1][o]) + (L.pFa[2][o] + L.pFa[3][o])) +
                          ((L.pFa[4][o] + L.pFa[5][o]) + (L.pFa[6][o] + L.pFa[7][o]));
        const float tt = fpc_tanhf(acc);
        const float f = tt + tt;  // the "dual" FC is the same Linear summed twice (wavernn.py:89-92)
        float rq = 0.0f;
        if (uv) {
            const int* ix = idx + fi * 4;
            if (m == 0) {
                const int k = ix[0];
                if (k >= 0) {
                    if (k < C.n_hi)
                        rq = (float)C.scl_hi[k];
                    else if (C.scl_lo && k - C.n_hi < C.n_lo)
                        rq = (float)C.scl_lo[k - C.n_hi];
                    else if (owner)
                        atomicOr(bad, 1);
                }
            } else {
                const int d = m - 1, k1 = ix[1], k2 = ix[2], k3 = ix[3];
                if (k1 >= 0) {
                    if (k1 >= C.N_hi0 || (C.S_hi == 2 && (k2 < 0 || k2 >= C.N_hi1))) {
                        if (owner) atomicOr(bad, 1);
                    } else {
                        const double e0 = C.vq_hi0_r[(size_t)k1 * NDIM + d];
                        rq = (float)(C.S_hi == 2 ? e0 + C.vq_hi1_r[(size_t)k2 * NDIM + d] : e0);
                    }
                } else if (k3 >= 0) {
                    if (!C.vq_lo_r || k3 >= C.N_lo) {
                        if (owner) atomicOr(bad, 1);
                    } else {
                        rq = (float)C.vq_lo_r[(size_t)k3 * NDIM + d];
                    }
                }
            }
        }
        const float cn = f + rq;
        L.x[ws_xi(m, u)] = uv ? cn : 0.0f;
        if (owner) c_out[fi * WIN + m] = cn;
    } else if (m < WIN) {
        L.x[ws_xi(m, u)] = uv ? pv : 0.0f;
        if (owner) c_out[fi * WIN + m] = pv;
    }
}

__global__ __launch_bounds__(NT) void k_encode_wsd(const PredDev P, const CbDev C, const EncArgs A, const WsArgs S) {
    __shared__ WsLds L;
    const int tid = threadIdx.x;
    int group, slice;
    if (!ws_role(S.ngroups, group, slice)) return;
    WsCtx X = ws_ctx(S, group, slice);
    X.own = -1;  // (the output layer runs for all utterances on every workgroup: ws_foreground<false, true>)
    WsRegs R;
    for (int i = tid; i < WH1 * WG; i += NT) L.h1[i] = 0.0f;  // h = None -> zeros (wavernn.py:182)
    for (int i = tid; i < WH2 * WG; i += NT) L.h2[i] = 0.0f;
    for (int i = tid; i < WIN * WXP + 4; i += NT) L.x[i] = 0.0f;   // c_in[:, 0, :] is all zero (wavernn.py:177-178)
    for (int k = tid; k < C.n_hi; k += NT) L.sclc[k] = C.scl_hi[k];
    for (int k = tid; k < C.n_lo; k += NT) L.sclc[C.n_hi + k] = C.scl_lo[k];
    for (int i = tid; i < 3 * WNS * (NDIM + 1); i += NT) {  // this workgroup's entries 32 m + slice of the three books
        const int cb = i / (WNS * (NDIM + 1)), r = i - cb * WNS * (NDIM + 1), mm = r / (NDIM + 1), d = r - mm * (NDIM + 1);
        const double* src = cb == 0 ? C.vq_hi0_r : (cb == 1 ? C.vq_hi1_r : C.vq_lo_r);
        const int N = cb == 0 ? C.N_hi0 : (cb == 1 ? C.N_hi1 : C.N_lo), e = WNS * mm + slice;
        L.cbs[cb][mm][d] = (src != nullptr && e < N && d < NDIM) ? src[(size_t)e * NDIM + d] : 0.0;
    }
    __syncthreads();
    ws_prologue(P, X, L, R, S, tid);
    if (X.fallback) return;  // (workgroup- and group-uniform, nothing written yet) the row-split launch behind this one serves the group
    int i = 0;
    // this thread's feature value of a frame: thread (u, m) = column m < 20 of utterance u -- rows 0-17 for the residuals, the
    // pitch columns pass through to the next input (wavernn.py:178) -- where it is in frame 0 (nullptr: none)
    const float* fvp = nullptr;
    if ((tid & 31) < WIN && (tid >> 5) < X.nu) fvp = A.feat + (size_t)(X.b0 + (tid >> 5)) * A.Lf * WIN + (tid & 31);
    const float2* mkp = nullptr;  // mask mode: the utterance's two indicators of a frame (every lane of its half-wave)
    if (A.mask != nullptr && (tid >> 5) < X.nu) mkp = reinterpret_cast<const float2*>(A.mask) + (size_t)(X.b0 + (tid >> 5)) * A.Lf;
    WPROF_INIT()
    for (; i < A.Lf; ++i) {
        const unsigned epoch = (unsigned)i + 1u;
        const float fv = fvp != nullptr ? fvp[i * WIN] : 0.0f;  // fetched before the step
        const float2 mk = mkp != nullptr ? mkp[i] : make_float2(0.0f, 0.0f);
        if (tid < WFGT) {
            __builtin_amdgcn_s_setprio(FPC_FG_PRIO);
            (void)ws_foreground<false, true>(X, L, R, i, tid);
            __builtin_amdgcn_s_setprio(0);
        } else {
            (void)ws_background(X, L, R, i, i + 1 == A.Lf, tid - WFGT);
        }
        WBSTAMP(19)
        lds_barrier();  // both roles meet: the tail takes the whole workgroup
        WSTAMP(20)
        wsd_tail(X, L, C, A, S.err, i, fv, mk, tid, epoch);
        WSTAMP(21)
        // The frame's last barrier (the next input rows are in LDS) carries the give-up flag as well: one thread copies it
        // before the barrier, everybody acts on the copy behind it.  A wait given up behind the copy shows a frame later, or in
        // the check behind the loop: every wait fails at once from then on, the launch poisons its outputs either way.
        if (tid == 0) L.dead_latch = __hip_atomic_load(&L.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_barrier();
        if (L.dead_latch != 0) break;
        WSTAMP(22)
    }
    // A wait given up BEHIND a frame's copy of the flag (a wave still in the tail's gathers when thread 0 copied) has written
    // that frame's outputs from a failed gather and shows in the next frame's copy only: a launch that leaves the loop at
    // frame i poisons from frame i - 1 on (fpcodec.h: NaN / -2 "from that frame on", and such frames in no histogram).
    if (i < A.Lf)
        i = i > 0 ? i - 1 : 0;
    else if (A.Lf > 0 && ws_frame_dead(L, tid))
        i = A.Lf - 1;  // (given up in the last frame's tail, behind its copy)
    WPROF_DUMP(A.Lf)
    if (i < A.Lf && slice < X.nu) encode_poison(P, A, X.b0 + slice, i, tid);
}
