/*
 * fpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of haiciyang/Feature-predictor-for-speech-codec.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped path (libfpcodec.so) never links or calls it.
 *
 * Parity status
 *   predictor / encoder / VQ / scalar-Q / ceps2lpc / mu-law / lpc_pred / entropy:
 *     PINNED -- checked against tests/golden/ (npz files), which were produced by running
 *     the reference's own Python in the build container (tests/golden/make_golden.py).
 *   LPCNet vocoder (orc_lpcnet_*): PARITY UNPINNED -- xiph/LPCNet is a third-party
 *     dependency referenced only by URL (README.md:13-15,47), no version pin, source
 *     absent from /root/reference.  The code below restates the published algorithm
 *     (lpcnet.py / mdense.py / ulaw.py / test_lpcnet.py of the training_tf2 tree) and is
 *     anchored on the fragments the reference itself restates (cited per function).
 *
 * Every function cites the reference file:line it follows (paths under
 * /root/reference).  Build: see oracle/Makefile (-ffp-contract=off is required).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "fpc_numerics.h"

#define EXPORT __attribute__((visibility("default")))

/* ======================================================================
 * 1. Feature predictor  (src/models/wavernn.py)
 * ==================================================================== */

typedef struct {
    int in, h1, h2, fc;
    const float *w1_ih, *w1_hh, *b1_ih, *b1_hh;
    const float *w2_ih, *w2_hh, *b2_ih, *b2_hh;
    const float *fc_w, *fc_b;
} orc_pred;

/* The predictor's rows are evaluated in S contiguous segments of the input (the kernel gives every
 * segment its own thread): segment 0 is a k-ordered fmaf chain from the bias (the order a gfx950 f32 MFMA accumulates in), the others
 * start from 0, and the
 * segment sums are added as a balanced tree.  S depends on the row length only. */
static int fpc_segments(int K) {
    const int S = K >= 256 ? 4 : (K >= 64 ? 2 : 1);
    return K % S == 0 ? S : 1;
}
static void matvec_seg(const float* W, const float* bias, const float* x, int rows, int cols, int S, float* y) {
    const int len = cols / S;
    for (int r = 0; r < rows; ++r) {
        float part[8];
        const float* w = W + (size_t)r * cols;
        for (int sgm = 0; sgm < S; ++sgm) {
            float acc = sgm == 0 ? bias[r] : 0.0f;
            for (int k = sgm * len; k < (sgm + 1) * len; ++k) acc = fmaf(x[k], w[k], acc);
            part[sgm] = acc;
        }
        for (int st = 1; st < S; st <<= 1)
            for (int q = 0; q + st < S; q += 2 * st) part[q] = part[q] + part[q + st];
        y[r] = part[0];
    }
}

/* torch.nn.GRU cell, gate rows [r; z; n]  (wavernn.py:37-38,71,76) */
static void gru_cell(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                     const float* x, int in, float* h, int H, float* gi, float* gh) {
    matvec_seg(w_ih, b_ih, x, 3 * H, in, fpc_segments(in), gi);
    matvec_seg(w_hh, b_hh, h, 3 * H, H, fpc_segments(H), gh);
    for (int i = 0; i < H; ++i) {
        const float r = fpc_sigmoidf(gi[i] + gh[i]);
        const float z = fpc_sigmoidf(gi[H + i] + gh[H + i]);
        const float n = fpc_tanhf(fmaf(r, gh[2 * H + i], gi[2 * H + i]));
        h[i] = fmaf(z, h[i] - n, n); /* (1-z)*n + z*h */
    }
}

/* One frame of Wavernn.forward (wavernn.py:63-102): rnn1 -> rnn2 -> relu ->
 * Linear+tanh applied to two copies and summed (:89-92) == 2*tanh(.) */
static void pred_step(const orc_pred* p, const float* x, float* h1, float* h2, float* y,
                      float* scratch) {
    float* gi = scratch;
    float* gh = scratch + 3 * (p->h1 > p->h2 ? p->h1 : p->h2); /* (either layer may be the wider one) */
    gru_cell(p->w1_ih, p->w1_hh, p->b1_ih, p->b1_hh, x, p->in, h1, p->h1, gi, gh);
    gru_cell(p->w2_ih, p->w2_hh, p->b2_ih, p->b2_hh, h1, p->h1, h2, p->h2, gi, gh);
    float* relu = gi;
    for (int i = 0; i < p->h2; ++i) relu[i] = h2[i] > 0.0f ? h2[i] : 0.0f;
    float pre[64];
    const int Sfc = (p->h2 % 8 == 0 && p->h2 >= 64) ? 8 : 1; /* the few output rows: 8 segments each */
    matvec_seg(p->fc_w, p->fc_b, relu, p->fc, p->h2, Sfc, pre);
    for (int o = 0; o < p->fc; ++o) {
        const float t = fpc_tanhf(pre[o]);
        y[o] = t + t;
    }
}

EXPORT int orc_predictor_forward(const orc_pred* p, const float* x, int B, int L, float* h1,
                                 float* h2, float* y) {
    float* scratch = (float*)malloc(sizeof(float) * 6 * (size_t)(p->h1 > p->h2 ? p->h1 : p->h2));
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t)
            pred_step(p, x + ((size_t)b * L + t) * p->in, h1 + (size_t)b * p->h1,
                      h2 + (size_t)b * p->h2, y + ((size_t)b * L + t) * p->fc, scratch);
    free(scratch);
    return 0;
}

/* ======================================================================
 * 2. Quantizers  (src/quantization/vq_func.py)
 * ==================================================================== */
#define ORC_NDIM 17
#define ORC_SURV 5 /* vq_func.py:3 */

/* squared distance in float64 with numpy's pairwise-sum association for a
 * contiguous 17-element reduction (np.sum(..., -1) at vq_func.py:18):
 * 8 running sums over elements j and j+8, combined as a balanced tree, then +a[16]. */
static double dist17(const double* x, const double* c) {
    double r[8];
    for (int j = 0; j < 8; ++j) {
        const double d = x[j] - c[j];
        r[j] = d * d;
    }
    for (int j = 0; j < 8; ++j) {
        const double d = x[8 + j] - c[8 + j];
        r[j] += d * d;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const double d = x[16] - c[16];
    res += d * d;
    return res;
}

/* vq_quantize_mbest (vq_func.py:10-24): the 5 entries with smallest distance,
 * ties resolved to the lower index (Python's stable sorted over arange). */
EXPORT void orc_vq_mbest(const double* cb, int n, const double* x, int* idx, double* dist) {
    for (int m = 0; m < ORC_SURV; ++m) {
        idx[m] = -1;
        dist[m] = INFINITY;
    }
    for (int e = 0; e < n; ++e) {
        const double d = dist17(x, cb + (size_t)e * ORC_NDIM);
        int pos = ORC_SURV;
        while (pos > 0 && d < dist[pos - 1]) --pos; /* strict: earlier index wins ties */
        if (pos < ORC_SURV) {
            for (int j = ORC_SURV - 1; j > pos; --j) {
                dist[j] = dist[j - 1];
                idx[j] = idx[j - 1];
            }
            dist[pos] = d;
            idx[pos] = e;
        }
    }
}

/* quantize_mstage (vq_func.py:82-131) for S in {1,2} (S>=3 is broken in the
 * reference).  x: 17 float32 promoted to float64. */
EXPORT void orc_quantize_mstage(const float* x32, int S, const int* n_entries, const double* cb,
                                double* qx, int* index_out) {
    double x[ORC_NDIM];
    for (int d = 0; d < ORC_NDIM; ++d) x[d] = (double)x32[d];
    int idx0[ORC_SURV];
    double d0[ORC_SURV];
    const double* cb0 = cb;
    orc_vq_mbest(cb0, n_entries[0], x, idx0, d0);
    if (S == 1) {
        for (int d = 0; d < ORC_NDIM; ++d) qx[d] = cb0[(size_t)idx0[0] * ORC_NDIM + d];
        index_out[0] = idx0[0];
        return;
    }
    const double* cb1 = cb + (size_t)n_entries[0] * ORC_NDIM;
    int best0[ORC_SURV], best1[ORC_SURV];
    double glob[ORC_SURV];
    for (int k = 0; k < ORC_SURV; ++k) {
        double diff[ORC_NDIM];
        for (int d = 0; d < ORC_NDIM; ++d) diff[d] = x[d] - cb0[(size_t)idx0[k] * ORC_NDIM + d];
        int ci[ORC_SURV];
        double cd[ORC_SURV];
        orc_vq_mbest(cb1, n_entries[1], diff, ci, cd);
        if (k == 0) { /* vq_func.py:110-113 */
            for (int m = 0; m < ORC_SURV; ++m) {
                best0[m] = idx0[0];
                best1[m] = ci[m];
                glob[m] = cd[m];
            }
        } else if (cd[0] < glob[ORC_SURV - 1]) { /* :115-125 single forward merge pass */
            int m = 0;
            for (int p = 0; p < ORC_SURV; ++p) {
                if (cd[m] < glob[p]) {
                    for (int j = ORC_SURV - 1; j > p; --j) {
                        glob[j] = glob[j - 1];
                        best0[j] = best0[j - 1];
                        best1[j] = best1[j - 1];
                    }
                    glob[p] = cd[m];
                    best0[p] = idx0[k];
                    best1[p] = ci[m];
                    ++m;
                }
            }
        }
    }
    for (int d = 0; d < ORC_NDIM; ++d) /* :127-129: 0 + CB0[i0] + CB1[i1] */
        qx[d] = cb0[(size_t)best0[0] * ORC_NDIM + d] + cb1[(size_t)best1[0] * ORC_NDIM + d];
    index_out[0] = best0[0];
    index_out[1] = best1[0];
}

/* vq_quantize (vq_func.py:134-164) on a batch; hist = per-stage usage counts */
EXPORT void orc_vq_quantize(const float* r, int n, int S, const int* n_entries, const double* cb,
                            double* qr, int* idx /*[n,2]*/, double* hist /*sum n_entries*/) {
    for (int i = 0; i < n; ++i) {
        int ix[2] = {-1, -1};
        orc_quantize_mstage(r + (size_t)i * ORC_NDIM, S, n_entries, cb, qr + (size_t)i * ORC_NDIM,
                            ix);
        if (idx) {
            idx[2 * i] = ix[0];
            idx[2 * i + 1] = ix[1];
        }
        if (hist) {
            hist[ix[0]] += 1.0;
            if (S == 2) hist[n_entries[0] + ix[1]] += 1.0;
        }
    }
}

/* scl_quantize (vq_func.py:167-185): first arg-min of (x - code)^2 in float64 */
EXPORT void orc_scl_quantize(const float* x, int n, const double* codes, int n_codes, double* q,
                             int* idx, double* hist) {
    for (int i = 0; i < n; ++i) {
        const double v = (double)x[i];
        int best = 0;
        double bd = INFINITY;
        for (int c = 0; c < n_codes; ++c) {
            const double d = (v - codes[c]) * (v - codes[c]);
            if (d < bd) {
                bd = d;
                best = c;
            }
        }
        q[i] = codes[best];
        if (idx) idx[i] = best;
        if (hist) hist[best] += 1.0;
    }
}

/* cal_entropy (src/generate_qtz_features.py:94-101): bits/symbol of a usage histogram */
EXPORT double orc_cal_entropy(const double* hist, int n) {
    double tot = 0.0;
    for (int i = 0; i < n; ++i) tot += hist[i];
    double ent = 0.0;
    for (int i = 0; i < n; ++i) {
        const double p = hist[i] / tot;
        ent += -p * log2(p + 1e-20);
    }
    return ent;
}

/* ======================================================================
 * 3. Closed-loop encoder  (src/models/wavernn.py:165-256, mask=None)
 * ==================================================================== */
typedef struct {
    int S_hi;
    int N_hi[2];
    const double* vq_hi; /* stages back to back */
    int N_lo;
    const double* vq_lo; /* may be NULL */
    int n_hi;
    const double* scl_hi;
    int n_lo;
    const double* scl_lo; /* may be NULL */
} orc_codebooks;

EXPORT int orc_encode(const orc_pred* p, const orc_codebooks* cb, const float* feat, int B, int L,
                      float l1, float l2, int qtz, float* c_in_out /*[B,L,20]*/, float* r_out,
                      float* r_qtz_out, float* r_under_out, float* ind1_out, float* ind2_out,
                      int* idx_out /*[B,L,4]*/, double* hist) {
    const int C = p->in, F = p->fc; /* 20, 18 */
    float* scratch = (float*)malloc(sizeof(float) * 6 * (size_t)(p->h1 > p->h2 ? p->h1 : p->h2));
    float* h1 = (float*)malloc(sizeof(float) * p->h1);
    float* h2 = (float*)malloc(sizeof(float) * p->h2);
    float* cin = (float*)malloc(sizeof(float) * C);
    float* fo = (float*)malloc(sizeof(float) * F);
    int off_sl = 0, off_v0 = 0, off_v1 = 0, off_vl = 0;
    if (cb) {
        off_sl = cb->n_hi;
        off_v0 = off_sl + cb->n_lo;
        off_v1 = off_v0 + cb->N_hi[0];
        off_vl = off_v1 + (cb->S_hi == 2 ? cb->N_hi[1] : 0);
    }
    for (int b = 0; b < B; ++b) {
        memset(h1, 0, sizeof(float) * p->h1); /* h=None -> zeros (:182) */
        memset(h2, 0, sizeof(float) * p->h2);
        memset(cin, 0, sizeof(float) * C); /* c_in[:,0,:] is all zero incl. pitch (:177-178) */
        for (int i = 0; i < L; ++i) {
            const size_t fi = ((size_t)b * L + i);
            const float* f = feat + fi * C;
            pred_step(p, cin, h1, h2, fo, scratch); /* :194-195 */
            float rs[18];
            for (int d = 0; d < F; ++d) rs[d] = f[d] - fo[d]; /* :196 */
            float s = 0.0f;
            for (int d = 1; d < F; ++d) s += fabsf(rs[d]);
            const int i1 = fabsf(rs[0]) > l1; /* :202 */
            const int i2 = s > l2;            /* :206 */
            ind1_out[fi] = (float)i1;
            ind2_out[fi] = (float)i2;
            float* r = r_out + fi * F;
            float* rq = r_qtz_out + fi * F;
            float* ru = r_under_out + fi * F;
            int* ix = idx_out ? idx_out + fi * 4 : NULL;
            if (ix) ix[0] = ix[1] = ix[2] = ix[3] = -1;
            for (int d = 0; d < F; ++d) rq[d] = ru[d] = 0.0f;
            float* cnext = c_in_out + fi * C;
            if (qtz) {
                for (int d = 0; d < F; ++d) r[d] = rs[d]; /* :197 (un-thresholded) */
                double q;
                int qi;
                if (i1) { /* :218-221 */
                    orc_scl_quantize(rs, 1, cb->scl_hi, cb->n_hi, &q, &qi, hist);
                    rq[0] = (float)q;
                    if (ix) ix[0] = qi;
                } else if (cb->scl_lo) { /* :222-225 */
                    orc_scl_quantize(rs, 1, cb->scl_lo, cb->n_lo, &q, &qi, hist ? hist + off_sl : NULL);
                    rq[0] = (float)q;
                    if (ix) ix[0] = cb->n_hi + qi;
                }
                double qv[ORC_NDIM];
                int vi[2];
                if (i2) { /* :229-234 */
                    orc_vq_quantize(rs + 1, 1, cb->S_hi, cb->N_hi, cb->vq_hi, qv, vi,
                                    hist ? hist + off_v0 : NULL);
                    for (int d = 0; d < ORC_NDIM; ++d) rq[1 + d] = (float)qv[d];
                    if (ix) {
                        ix[1] = vi[0];
                        ix[2] = vi[1];
                    }
                } else if (cb->vq_lo) { /* :235-240 */
                    orc_vq_quantize(rs + 1, 1, 1, &cb->N_lo, cb->vq_lo, qv, vi,
                                    hist ? hist + off_vl : NULL);
                    for (int d = 0; d < ORC_NDIM; ++d) rq[1 + d] = (float)qv[d];
                    if (ix) ix[3] = vi[0];
                }
                for (int d = 0; d < F; ++d) cnext[d] = fo[d] + rq[d]; /* :242 */
            } else {                                                  /* :244-252 */
                ru[0] = rs[0] * (float)(1 - i1);
                r[0] = rs[0] * (float)i1;
                for (int d = 1; d < F; ++d) {
                    ru[d] = rs[d] * (float)(1 - i2);
                    r[d] = rs[d] * (float)i2;
                }
                for (int d = 0; d < F; ++d) cnext[d] = fo[d] + r[d];
            }
            for (int d = F; d < C; ++d) cnext[d] = f[d]; /* pitch passes through (:178) */
            memcpy(cin, cnext, sizeof(float) * C);
        }
    }
    free(scratch);
    free(h1);
    free(h2);
    free(cin);
    free(fo);
    return 0;
}

/* Receiver side of orc_encode (the reference's own Wavernn.decoder, wavernn.py:367-379,
 * is dead code): rebuild c_in[:,1:,:] from the transmitted symbols alone.  idx [B,L,4] as written by
 * orc_encode ({scalar idx (+n_hi when from the below-threshold codebook), vq stage 1, vq stage 2,
 * below-threshold vq idx}, -1 = not coded), pitch [B,L,2] = feat[:,:,18:20] (side information).
 * Same predictor steps and the same float64 -> float32 dequantisation as the encoder, so the output
 * equals the encoder's c_in bit for bit. */
EXPORT int orc_decode_features(const orc_pred* p, const orc_codebooks* cb, const float* pitch, const int* idx,
                               int B, int L, float* c_out /*[B,L,20]*/) {
    const int C = p->in, F = p->fc;
    float* scratch = (float*)malloc(sizeof(float) * 6 * (size_t)(p->h1 > p->h2 ? p->h1 : p->h2));
    float* h1 = (float*)malloc(sizeof(float) * p->h1);
    float* h2 = (float*)malloc(sizeof(float) * p->h2);
    float* cin = (float*)malloc(sizeof(float) * C);
    float* fo = (float*)malloc(sizeof(float) * F);
    int rc = 0;
    for (int b = 0; b < B && rc == 0; ++b) {
        memset(h1, 0, sizeof(float) * p->h1);
        memset(h2, 0, sizeof(float) * p->h2);
        memset(cin, 0, sizeof(float) * C);
        for (int i = 0; i < L; ++i) {
            const size_t fi = ((size_t)b * L + i);
            const int* ix = idx + fi * 4;
            pred_step(p, cin, h1, h2, fo, scratch);
            float rq[18];
            for (int d = 0; d < F; ++d) rq[d] = 0.0f;
            if (ix[0] >= 0) {
                if (ix[0] < cb->n_hi)
                    rq[0] = (float)cb->scl_hi[ix[0]];
                else if (cb->scl_lo && ix[0] - cb->n_hi < cb->n_lo)
                    rq[0] = (float)cb->scl_lo[ix[0] - cb->n_hi];
                else
                    rc = -1;
            }
            if (ix[1] >= 0) {
                if (ix[1] >= cb->N_hi[0] || (cb->S_hi == 2 && (ix[2] < 0 || ix[2] >= cb->N_hi[1]))) {
                    rc = -1;
                } else {
                    const double* e0 = cb->vq_hi + (size_t)ix[1] * ORC_NDIM;
                    const double* e1 = cb->S_hi == 2 ? cb->vq_hi + ((size_t)cb->N_hi[0] + ix[2]) * ORC_NDIM : NULL;
                    for (int d = 0; d < ORC_NDIM; ++d) rq[1 + d] = (float)(e1 ? e0[d] + e1[d] : e0[d]);
                }
            } else if (ix[3] >= 0) {
                if (!cb->vq_lo || ix[3] >= cb->N_lo)
                    rc = -1;
                else
                    for (int d = 0; d < ORC_NDIM; ++d) rq[1 + d] = (float)cb->vq_lo[(size_t)ix[3] * ORC_NDIM + d];
            }
            float* cnext = c_out + fi * C;
            for (int d = 0; d < F; ++d) cnext[d] = fo[d] + rq[d];
            for (int d = F; d < C; ++d) cnext[d] = pitch[fi * (C - F) + (d - F)];
            memcpy(cin, cnext, sizeof(float) * C);
        }
    }
    free(scratch);
    free(h1);
    free(h2);
    free(cin);
    free(fo);
    return rc;
}

/* ======================================================================
 * 4. cepstrum -> LPC  (src/ceps2lpc/ceps2lpc_vct.py)
 * ==================================================================== */
#define NB_BANDS 18
#define FREQ_SIZE 161
#define WINDOW_SIZE 320

static const float COMPENSATION[NB_BANDS] = {/* ceps2lpc_vct.py:23-25 */
                                             0.8f,      1.0f,      1.0f,  1.0f,  1.0f, 1.0f,
                                             1.0f,      1.0f,      0.666667f, 0.5f, 0.5f, 0.5f,
                                             0.333333f, 0.25f,     0.25f, 0.2f,  0.166667f, 0.173913f};
static const int EBAND5MS[NB_BANDS] = {/* :47-50 */
                                       0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};

static float g_dct[NB_BANDS][NB_BANDS];
static double g_cos[17][FREQ_SIZE];
static int g_tables_ready = 0;

static void ceps_tables(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < NB_BANDS; ++i)
        for (int j = 0; j < NB_BANDS; ++j) { /* :27-32 (float32 argument, float32 cosine) */
            const float arg = (float)((i + 0.5) * j * M_PI / NB_BANDS);
            float c = (float)cos((double)arg);
            if (j == 0) c *= (float)sqrt(0.5);
            g_dct[i][j] = c;
        }
    for (int k = 0; k < 17; ++k)
        for (int m = 0; m < FREQ_SIZE; ++m)
            g_cos[k][m] = cos(2.0 * M_PI * (double)((m * k) % WINDOW_SIZE) / (double)WINDOW_SIZE);
    g_tables_ready = 1;
}

/* ceps2lpc_v (:122-162) for one frame; returns the final Levinson error. */
static float ceps2lpc_row(const float* ceps, float* lpc, float* rc) {
    float in[NB_BANDS], Ex[NB_BANDS];
    for (int j = 0; j < NB_BANDS; ++j) in[j] = ceps[j] + (j == 0 ? 4.0f : 0.0f); /* :128-133 */
    const float s = (float)sqrt(2.0 / NB_BANDS);
    for (int i = 0; i < NB_BANDS; ++i) { /* idct :35-43 (mul then add, no fusing) */
        float sm = 0.0f;
        for (int j = 0; j < NB_BANDS; ++j) {
            const float t = in[j] * g_dct[i][j];
            sm = sm + t;
        }
        Ex[i] = fpc_exp10f(sm * s) * COMPENSATION[i]; /* :134 */
    }
    float X[FREQ_SIZE]; /* interp_band_gain :45-57; bin 160 stays 0 */
    for (int m = 0; m < FREQ_SIZE; ++m) X[m] = 0.0f;
    for (int i = 0; i < NB_BANDS - 1; ++i) {
        const int bs = (EBAND5MS[i + 1] - EBAND5MS[i]) * 4;
        for (int j = 0; j < bs; ++j) {
            const float frac = (float)((double)j / (double)bs);
            const float a = (1.0f - frac) * Ex[i];
            const float b = frac * Ex[i + 1];
            X[EBAND5MS[i] * 4 + j] = a + b;
        }
    }
    float ac[17]; /* irfft(n=320)[:17] (:140-143), evaluated as a float64 cosine sum */
    for (int k = 0; k < 17; ++k) {
        double acc = (double)X[0];
        for (int m = 1; m < FREQ_SIZE - 1; ++m) {
            const double t = 2.0 * (double)X[m] * g_cos[k][m];
            acc = acc + t;
        }
        acc = acc + (double)X[FREQ_SIZE - 1] * g_cos[k][FREQ_SIZE - 1];
        ac[k] = (float)(acc / (double)WINDOW_SIZE);
    }
    {
        const float t = ac[0] * 0.0001f; /* :147 */
        const float u = t + (float)(320.0 / 12.0 / 38.0);
        ac[0] = ac[0] + u;
    }
    for (int i = 1; i < 17; ++i) ac[i] = ac[i] * (float)(1.0 - 0.00006 * i * i); /* :150-151 */
    /* _celt_lpc_s (:60-88) */
    float error = ac[0];
    for (int i = 0; i < 16; ++i) {
        lpc[i] = 0.0f;
        if (rc) rc[i] = 0.0f;
    }
    if (ac[0] != 0.0f) {
        for (int i = 0; i < 16; ++i) {
            float rr = 0.0f;
            for (int j = 0; j < i; ++j) {
                const float t = lpc[j] * ac[i - j];
                rr = rr + t;
            }
            rr = rr + ac[i + 1];
            const float r = -rr / error;
            if (rc) rc[i] = r;
            lpc[i] = r;
            for (int j = 0; j < (i + 1) / 2; ++j) {
                const float t1 = lpc[j], t2 = lpc[i - 1 - j];
                const float m1 = r * t2, m2 = r * t1;
                lpc[j] = t1 + m1;
                lpc[i - 1 - j] = t2 + m2;
            }
            const float rr2 = r * r;
            const float dec = rr2 * error;
            error = error - dec;
            if (error < ac[0] / 1024.0f) break;
            if (error < 0.001f * ac[0]) break;
        }
    }
    return error;
}

EXPORT int orc_ceps2lpc(const float* ceps, int N, int stride, float* lpc, float* e, float* rc) {
    ceps_tables();
    for (int n = 0; n < N; ++n) {
        const float err = ceps2lpc_row(ceps + (size_t)n * stride, lpc + (size_t)n * 16,
                                       rc ? rc + (size_t)n * 16 : NULL);
        if (e) e[n] = err;
    }
    return 0;
}

/* ======================================================================
 * 5. Reference's own float mu-law / LPC predictor restatements (pins)
 * ==================================================================== */
/* utils.l2u (src/utils.py:19-24): no rounding, clip to [0,255] */
EXPORT void orc_l2u_ref(const float* x, int n, float* u) {
    const float scale = 255.0f / 32768.0f;
    for (int i = 0; i < n; ++i) {
        const float s = x[i] > 0 ? 1.0f : (x[i] < 0 ? -1.0f : 0.0f);
        float v = s * (128.0f * logf(1.0f + scale * fabsf(x[i])) / (float)log(256.0));
        v = 128.0f + v;
        u[i] = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    }
}
/* utils.u2l (src/utils.py:26-31) */
EXPORT void orc_u2l_ref(const float* u, int n, float* x) {
    const float scale_1 = 32768.0f / 255.0f;
    for (int i = 0; i < n; ++i) {
        const float v = u[i] - 128.0f;
        const float s = v > 0 ? 1.0f : (v < 0 ? -1.0f : 0.0f);
        x[i] = s * scale_1 * (expf(fabsf(v) / 128.0f * (float)log(256.0)) - 1.0f);
    }
}
/* utils.lpc_pred (src/utils.py:91-114): pred[t] = -sum_k lpc[t/160][k] x[t-k] */
EXPORT void orc_lpc_pred_ref(const float* x, const float* lpc, int B, int F, int frame, float* pred) {
    const int N = F * frame;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < N; ++t) {
            float acc = 0.0f;
            for (int k = 0; k < 16; ++k) {
                const float xv = (t - k >= 0) ? x[(size_t)b * N + t - k] : 0.0f;
                acc += lpc[((size_t)b * F + t / frame) * 16 + k] * xv;
            }
            pred[(size_t)b * N + t] = -acc;
        }
}
/* canonical integer mu-law / table used by the vocoder (fpc_numerics.h) */
EXPORT int orc_lin2ulaw(float x) { return fpc_lin2ulaw(x); }
EXPORT float orc_ulaw2lin(int u) { return fpc_ulaw2lin(u); }
EXPORT float orc_tanh(float x) { return fpc_tanhf(x); }
EXPORT float orc_sigmoid(float x) { return fpc_sigmoidf(x); }
EXPORT float orc_exp(float x) { return fpc_expf(x); }
EXPORT float orc_log(float x) { return fpc_logf(x); }
EXPORT float orc_philox_uniform(uint64_t seed, uint32_t t) { return fpc_philox_uniform(seed, t); }
EXPORT int orc_period_index(float f) { return fpc_period_index(f); }

/* ======================================================================
 * 6. LPCNet vocoder -- PARITY UNPINNED (see header).  Canonical evaluation
 *    orders are specified in DESIGN.md section "Vocoder numerics".
 * ==================================================================== */
#define RNN_A 384
#define RNN_B 16
#define COND 128
#define EMB 128
#define GA (3 * RNN_A) /* 1152 */
#define GB (3 * RNN_B) /* 48 */
#define BLK_R 8        /* sparse block: 8 outputs ... */
#define BLK_C 4        /* ... x 4 inputs */
#define LEAF_BLOCKS 2  /* canonical leaf = 2 consecutive blocks of a row group */
#define NROWGRP (GA / BLK_R) /* 144 */

typedef struct {
    const float* embed_pitch;
    const float *conv1_k, *conv1_b, *conv2_k, *conv2_b;
    const float *d1_k, *d1_b, *d2_k, *d2_b;
    const float* embed_sig;
    const float *ga_k, *ga_r, *ga_b;
    const float *gb_k, *gb_r, *gb_b;
    const float *md_k, *md_b, *md_f;
} orc_lpcnet_weights;

typedef struct {
    orc_lpcnet_weights w;
    float* tab[3];   /* [256][1152] embed_sig . kernel rows of sig/pred/exc */
    float* bias_a;   /* [1152] input bias (+ recurrent bias for z,r) */
    float* brn_a;    /* [384]  recurrent bias of the candidate gate */
    float* diag;     /* [3][384] */
    int grp_nblk[NROWGRP];
    int* grp_cols[NROWGRP];   /* column-block indices, ascending */
    float* grp_w[NROWGRP];    /* [nblk][8][4] */
    float* bias_b;   /* [48] */
    float* brn_b;    /* [16] */
    float ulaw_tab[256];
    int nblocks;
} orc_lpcnet;

EXPORT orc_lpcnet* orc_lpcnet_create(const orc_lpcnet_weights* w) {
    orc_lpcnet* m = (orc_lpcnet*)calloc(1, sizeof(orc_lpcnet));
    m->w = *w;
    /* embedding x input-kernel tables, float64 accumulation in k order, rounded once */
    for (int s = 0; s < 3; ++s) {
        m->tab[s] = (float*)malloc(sizeof(float) * 256 * GA);
        for (int e = 0; e < 256; ++e)
            for (int row = 0; row < GA; ++row) {
                double acc = 0.0;
                for (int k = 0; k < EMB; ++k) {
                    const double t = (double)w->embed_sig[e * EMB + k] *
                                     (double)w->ga_k[(size_t)(s * EMB + k) * GA + row];
                    acc = acc + t;
                }
                m->tab[s][(size_t)e * GA + row] = (float)acc;
            }
    }
    m->bias_a = (float*)malloc(sizeof(float) * GA);
    m->brn_a = (float*)malloc(sizeof(float) * RNN_A);
    for (int row = 0; row < GA; ++row)
        m->bias_a[row] = row < 2 * RNN_A ? w->ga_b[row] + w->ga_b[GA + row] : w->ga_b[row];
    for (int i = 0; i < RNN_A; ++i) m->brn_a[i] = w->ga_b[GA + 2 * RNN_A + i];
    m->bias_b = (float*)malloc(sizeof(float) * GB);
    m->brn_b = (float*)malloc(sizeof(float) * RNN_B);
    for (int o = 0; o < GB; ++o)
        m->bias_b[o] = o < 2 * RNN_B ? w->gb_b[o] + w->gb_b[GB + o] : w->gb_b[o];
    for (int i = 0; i < RNN_B; ++i) m->brn_b[i] = w->gb_b[GB + 2 * RNN_B + i];
    /* block-sparse recurrent matrix: 8 outputs x 4 inputs, diagonal kept apart */
    m->diag = (float*)malloc(sizeof(float) * GA);
    for (int g = 0; g < 3; ++g)
        for (int i = 0; i < RNN_A; ++i) m->diag[g * RNN_A + i] = w->ga_r[(size_t)i * GA + g * RNN_A + i];
    for (int grp = 0; grp < NROWGRP; ++grp) {
        const int g = grp / (RNN_A / BLK_R), rb = grp % (RNN_A / BLK_R);
        int cols[RNN_A / BLK_C];
        int n = 0;
        for (int cb = 0; cb < RNN_A / BLK_C; ++cb) {
            int nz = 0;
            for (int r = 0; r < BLK_R; ++r)
                for (int c = 0; c < BLK_C; ++c) {
                    const int in = cb * BLK_C + c, out = rb * BLK_R + r;
                    if (in == out) continue;
                    if (w->ga_r[(size_t)in * GA + g * RNN_A + out] != 0.0f) nz = 1;
                }
            if (nz) cols[n++] = cb;
        }
        m->grp_nblk[grp] = n;
        m->grp_cols[grp] = (int*)malloc(sizeof(int) * (n ? n : 1));
        m->grp_w[grp] = (float*)malloc(sizeof(float) * 32 * (n ? n : 1));
        for (int b = 0; b < n; ++b) {
            m->grp_cols[grp][b] = cols[b];
            for (int r = 0; r < BLK_R; ++r)
                for (int c = 0; c < BLK_C; ++c) {
                    const int in = cols[b] * BLK_C + c, out = rb * BLK_R + r;
                    m->grp_w[grp][b * 32 + r * 4 + c] =
                        in == out ? 0.0f : w->ga_r[(size_t)in * GA + g * RNN_A + out];
                }
        }
        m->nblocks += n;
    }
    for (int u = 0; u < 256; ++u) m->ulaw_tab[u] = fpc_ulaw2lin(u);
    return m;
}

EXPORT int orc_lpcnet_nblocks(const orc_lpcnet* m) { return m->nblocks; }

EXPORT void orc_lpcnet_destroy(orc_lpcnet* m) {
    if (!m) return;
    for (int s = 0; s < 3; ++s) free(m->tab[s]);
    for (int g = 0; g < NROWGRP; ++g) {
        free(m->grp_cols[g]);
        free(m->grp_w[g]);
    }
    free(m->bias_a);
    free(m->brn_a);
    free(m->diag);
    free(m->bias_b);
    free(m->brn_b);
    free(m);
}

/* dense layer as k-ordered fmaf chain from the bias, x:[K] W:[K][N] (Keras (in,out)) */
static void dense_chain(const float* x, const float* W, const float* b, int K, int N, float* y,
                        int do_tanh) {
    for (int o = 0; o < N; ++o) {
        float acc = b[o];
        for (int k = 0; k < K; ++k) acc = fmaf(x[k], W[(size_t)k * N + o], acc);
        y[o] = do_tanh ? fpc_tanhf(acc) : acc;
    }
}

/* frame-rate network "enc" of lpcnet.py: pitch embedding, two k=3 'same' convs
 * (tanh), two dense (tanh).  features: [T,36]; cfeat: [T,128] */
EXPORT void orc_lpcnet_condition(const orc_lpcnet* m, const float* feat, int T, float* cfeat) {
    const orc_lpcnet_weights* w = &m->w;
    const int C0 = FPC_NB_USED_FEATURES + 64; /* 84 */
    float* x0 = (float*)calloc((size_t)(T + 2) * C0, sizeof(float));
    float* x1 = (float*)calloc((size_t)(T + 2) * COND, sizeof(float));
    float* x2 = (float*)malloc(sizeof(float) * COND);
    float* x3 = (float*)malloc(sizeof(float) * COND);
    for (int t = 0; t < T; ++t) {
        const float* f = feat + (size_t)t * FPC_NB_FEATURES;
        float* x = x0 + (size_t)(t + 1) * C0;
        for (int c = 0; c < FPC_NB_USED_FEATURES; ++c) x[c] = f[c];
        const int pidx = fpc_period_index(f[18]); /* src/synthesis.py:103 */
        for (int c = 0; c < 64; ++c) x[FPC_NB_USED_FEATURES + c] = w->embed_pitch[pidx * 64 + c];
    }
    for (int t = 0; t < T; ++t) /* conv1: taps t-1,t,t+1 as one 252-long chain */
        dense_chain(x0 + (size_t)t * C0, w->conv1_k, w->conv1_b, 3 * C0, COND,
                    x1 + (size_t)(t + 1) * COND, 1);
    for (int t = 0; t < T; ++t) {
        dense_chain(x1 + (size_t)t * COND, w->conv2_k, w->conv2_b, 3 * COND, COND, x2, 1);
        dense_chain(x2, w->d1_k, w->d1_b, COND, COND, x3, 1);
        dense_chain(x3, w->d2_k, w->d2_b, COND, COND, cfeat + (size_t)t * COND, 1);
    }
    free(x0);
    free(x1);
    free(x2);
    free(x3);
}

/* zero-padded balanced binary tree over P partial sums (in place, result in a[0]) */
static float tree_reduce(float* a, int P) {
    for (int s = 1; s < P; s <<= 1)
        for (int p = 0; p + s < P; p += 2 * s) a[p] = a[p] + a[p + s];
    return a[0];
}

/* probability of mu-law level v under the 8-level binary tree with node probabilities q[1..255] */
static float tree_leaf_prob(const float* q, int v) {
    float f[8];
    int node = 1;
    for (int l = 0; l < 8; ++l) {
        const int bit = (v >> (7 - l)) & 1;
        const float qq = q[node];
        f[l] = bit ? qq : 1.0f - qq;
        node = 2 * node + bit;
    }
    return ((((f[0] * f[1]) * (f[2] * f[3])) * (f[4] * f[5])) * f[6]) * f[7];
}

static float g_tanh_tab[FPC_TANH_TABLE_SIZE + 3];
static int g_tanh_ready = 0;
static const float* tanh_tab(void) {
    if (!g_tanh_ready) {
        for (int k = 0; k < FPC_TANH_TABLE_SIZE; ++k) g_tanh_tab[k] = fpc_tanh_table_entry(k);
        g_tanh_ready = 1;
    }
    return g_tanh_tab;
}

/* sample loop of test_lpcnet.py for one utterance.  Optional traces:
 *   exc_out [T*160] uint8 (0 for skipped samples), pcm_f [T*160] float */
EXPORT void orc_lpcnet_synthesize(const orc_lpcnet* m, const float* feat, int T, uint64_t seed,
                                  int16_t* pcm_out, uint8_t* exc_out, float* pcm_f_out) {
    const orc_lpcnet_weights* w = &m->w;
    const float* TT = tanh_tab();
    float* cfeat = (float*)malloc(sizeof(float) * (size_t)T * COND);
    orc_lpcnet_condition(m, feat, T, cfeat);
    float s1[RNN_A], s2[RNN_B], cfa[GA], cfb[GB], q[256], p[256], c[256], tmp[256];
    float hist[FPC_LPC_ORDER]; /* hist[k] = pcm[t-1-k] */
    memset(s1, 0, sizeof s1);
    memset(s2, 0, sizeof s2);
    memset(hist, 0, sizeof hist);
    int e_sig = 128, e_exc = 128; /* fexc init (test_lpcnet.py) */
    float mem = 0.0f;
    int skip = FPC_LPC_ORDER + 1;
    for (int fr = 0; fr < T; ++fr) {
        const float* f = feat + (size_t)fr * FPC_NB_FEATURES;
        const float* a = f + FPC_NB_FEATURES - FPC_LPC_ORDER;
        const float* cf = cfeat + (size_t)fr * COND;
        const float shape_e = fpc_shape_exponent(f[19]);
        /* frame-rate conditioning products (cfeat rows 384.. of both GRU kernels) */
        for (int row = 0; row < GA; ++row) {
            float acc = m->bias_a[row];
            for (int k = 0; k < COND; ++k) acc = fmaf(cf[k], w->ga_k[(size_t)(3 * EMB + k) * GA + row], acc);
            cfa[row] = acc;
        }
        for (int o = 0; o < GB; ++o) {
            float acc = m->bias_b[o];
            for (int k = 0; k < COND; ++k) acc = fmaf(cf[k], w->gb_k[(size_t)(RNN_A + k) * GB + o], acc);
            cfb[o] = acc;
        }
        for (int i = 0; i < FPC_FRAME_SIZE; ++i) {
            const int t = fr * FPC_FRAME_SIZE + i;
            if (i < skip) {
                pcm_out[t] = 0;
                if (exc_out) exc_out[t] = 0;
                if (pcm_f_out) pcm_f_out[t] = 0.0f;
                continue;
            }
            /* LPC prediction (src/utils.py:91-114 sign/tap order) */
            float prod[FPC_LPC_ORDER]; /* older taps: balanced tree; newest tap: one fma on top */
            prod[0] = 0.0f;
            for (int k = 1; k < FPC_LPC_ORDER; ++k) prod[k] = a[k] * hist[k];
            const float pred = -fmaf(a[0], hist[0], tree_reduce(prod, FPC_LPC_ORDER));
            const int e_pred = fpc_lin2ulaw(pred);
            /* GRU_A: sparse recurrent product, canonical leaf/tree order */
            float u[GA];
            for (int grp = 0; grp < NROWGRP; ++grp) {
                const int n = m->grp_nblk[grp];
                const int P = (n + LEAF_BLOCKS - 1) / LEAF_BLOCKS;
                for (int r = 0; r < BLK_R; ++r) {
                    float part[RNN_A / BLK_C / LEAF_BLOCKS + 1];
                    for (int l = 0; l < P; ++l) {
                        float sacc = 0.0f;
                        for (int b = l * LEAF_BLOCKS; b < n && b < (l + 1) * LEAF_BLOCKS; ++b) {
                            const float* wb = m->grp_w[grp] + b * 32 + r * 4;
                            const float* hv = s1 + m->grp_cols[grp][b] * BLK_C;
                            for (int cc = 0; cc < BLK_C; ++cc) sacc = fmaf(wb[cc], hv[cc], sacc);
                        }
                        part[l] = sacc;
                    }
                    const float tsum = P > 0 ? tree_reduce(part, P) : 0.0f;
                    const int g = grp / (RNN_A / BLK_R), unit = (grp % (RNN_A / BLK_R)) * BLK_R + r;
                    u[g * RNN_A + unit] = fmaf(m->diag[g * RNN_A + unit], s1[unit], tsum);
                }
            }
            float s1n[RNN_A];
            for (int j = 0; j < RNN_A; ++j) {
                float gi[3];
                for (int g = 0; g < 3; ++g) {
                    const int row = g * RNN_A + j;
                    gi[g] = ((m->tab[0][(size_t)e_sig * GA + row] + m->tab[1][(size_t)e_pred * GA + row]) +
                             m->tab[2][(size_t)e_exc * GA + row]) + cfa[row];
                }
                const float z = fpc_sigmoid_lut(TT, gi[0] + u[j]);
                const float r = fpc_sigmoid_lut(TT, gi[1] + u[RNN_A + j]);
                const float n = fpc_tanh_lut(TT, fmaf(r, u[2 * RNN_A + j] + m->brn_a[j], gi[2]));
                s1n[j] = fmaf(z, s1[j] - n, n);
            }
            memcpy(s1, s1n, sizeof s1);
            /* GRU_B: 64 leaves of 6 inputs, balanced tree.  Leaf l = 4*kl + j of input slice kl (24 inputs)
             * takes inputs 24*kl + c(j) + 4*m, m = 0..5, with c = {0, 2, 1, 3}: the kernel advances the
             * leaf pairs (c, c+1) with one packed fma per float4 of state and adds the pairs crosswise */
            float s2n[RNN_B];
            float gb[GB], ub[GB];
            for (int o = 0; o < GB; ++o) {
                float part[64];
                for (int l = 0; l < 64; ++l) {
                    float sacc = 0.0f;
                    static const int comp[4] = {0, 2, 1, 3};
                    for (int mm = 0; mm < 6; ++mm) {
                        const int k = 24 * (l >> 2) + comp[l & 3] + 4 * mm;
                        sacc = fmaf(w->gb_k[(size_t)k * GB + o], s1[k], sacc);
                    }
                    part[l] = sacc;
                }
                gb[o] = tree_reduce(part, 64) + cfb[o];
                float prod[RNN_B]; /* recurrent part: balanced tree over the 16 products */
                for (int k = 0; k < RNN_B; ++k) prod[k] = w->gb_r[(size_t)k * GB + o] * s2[k];
                ub[o] = tree_reduce(prod, RNN_B);
            }
            for (int j = 0; j < RNN_B; ++j) {
                const float z = fpc_sigmoid_lut(TT, gb[j] + ub[j]);
                const float r = fpc_sigmoid_lut(TT, gb[RNN_B + j] + ub[RNN_B + j]);
                const float n = fpc_tanh_lut(TT, fmaf(r, ub[2 * RNN_B + j] + m->brn_b[j], gb[2 * RNN_B + j]));
                s2n[j] = fmaf(z, s2[j] - n, n);
            }
            memcpy(s2, s2n, sizeof s2);
            /* dual fully-connected (mdense.py) -> node probabilities */
            q[0] = 0.0f;
            for (int j = 1; j < 256; ++j) {
                float tc[2];
                for (int ch = 0; ch < 2; ++ch) {
                    /* two chains: bias + even inputs, odd inputs; then their sum */
                    float dacc = w->md_b[j * 2 + ch], dodd = 0.0f;
                    for (int k = 0; k < RNN_B; k += 2) {
                        dacc = fmaf(w->md_k[((size_t)j * RNN_B + k) * 2 + ch], s2[k], dacc);
                        dodd = fmaf(w->md_k[((size_t)j * RNN_B + k + 1) * 2 + ch], s2[k + 1], dodd);
                    }
                    tc[ch] = fpc_tanh_lut(TT, dacc + dodd);
                }
                const float v = fmaf(w->md_f[j * 2 + 1], tc[1], w->md_f[j * 2] * tc[0]);
                q[j] = fpc_sigmoid_lut(TT, v);
            }
            /* 8-level binary tree -> pdf over 256 mu-law levels; branch factors f0..f7 MSB first, multiplied
             * as (((f0 f1)(f2 f3))(f4 f5)) f6 f7 */
            for (int v = 0; v < 256; ++v) {
                const float pv = tree_leaf_prob(q, v);
                p[v] = shape_e > 0.0f ? fpc_shape_pow(pv, shape_e) : pv; /* src/train.py:82 */
            }
            /* train.py:83-85 without the division: the cut is 0.002 of the pdf's total.  Without
             * sharpening the tree pdf sums to 1 by construction, so the total is not computed */
            float thr = 0.002f;
            if (shape_e > 0.0f) {
                memcpy(tmp, p, sizeof p);
                thr = 0.002f * tree_reduce(tmp, 256);
            }
            for (int v = 0; v < 256; ++v) {
                const float d = p[v] - thr;
                p[v] = d > 0.0f ? d : 0.0f;
            }
            /* three-level inclusive scan.  Inside each aligned group of 4 leaves the prefixes are two levels
             * deep: c0 | c0+c1 | (c0+c1)+c2 | (c0+c1)+(c2+c3); Kogge-Stone over the 16 group totals of each
             * aligned block of 64 leaves; block offsets as below */
            float I64[64];
            for (int g4 = 0; g4 < 64; ++g4) {
                const float* pg = p + 4 * g4;
                const float P1 = pg[0] + pg[1], s23 = pg[2] + pg[3];
                c[4 * g4] = pg[0];
                c[4 * g4 + 1] = P1;
                c[4 * g4 + 2] = P1 + pg[2];
                c[4 * g4 + 3] = P1 + s23;
                I64[g4] = c[4 * g4 + 3];
            }
            float R4[4], O4[3];
            for (int row = 0; row < 4; ++row) {
                float* x = I64 + 16 * row;
                for (int d = 1; d < 16; d <<= 1) {
                    float nx[16];
                    for (int l = 0; l < 16; ++l) nx[l] = l >= d ? x[l] + x[l - d] : x[l];
                    memcpy(x, nx, sizeof nx);
                }
                R4[row] = x[15];
            }
            O4[0] = 0.0f;
            O4[1] = R4[0];
            O4[2] = R4[0] + R4[1];
            /* block offsets in the order of the DPP row-broadcast scan: blocks 1 and 3 first take the
             * total of their left neighbour, then blocks 2 and 3 take the total of blocks 0+1 */
            for (int l = 0; l < 16; ++l) {
                I64[16 + l] = I64[16 + l] + R4[0];
                I64[32 + l] = I64[32 + l] + O4[2];
                I64[48 + l] = (I64[48 + l] + R4[2]) + O4[2];
            }
            const float S2 = I64[63];
            const float rthr = fpc_philox_uniform(seed, (uint32_t)t) * S2;
            /* the draw = number of leaves whose inclusive prefix is <= the threshold (inverse CDF, one uniform per
             * sample).  Prefix of leaf 4g+j, j < 3: scan value of group g-1 (0 for g = 0) + the in-group prefix;
             * the last leaf of a group carries the group's scan value itself */
            int cntall = 0;
            for (int g4 = 0; g4 < 64; ++g4) {
                const float O = g4 > 0 ? I64[g4 - 1] : 0.0f;
                for (int j = 0; j < 3; ++j)
                    if (O + c[4 * g4 + j] <= rthr) ++cntall;
                if (I64[g4] <= rthr) ++cntall;
            }
            const int exc = cntall > 255 ? 255 : cntall;
            /* synthesis filter + de-emphasis (wavenet.py:188 coefficient) */
            const float pcm = pred + m->ulaw_tab[exc];
            for (int k = FPC_LPC_ORDER - 1; k > 0; --k) hist[k] = hist[k - 1];
            hist[0] = pcm;
            e_sig = fpc_lin2ulaw(pcm);
            e_exc = exc;
            mem = fmaf(FPC_PREEMPH, mem, pcm);
            pcm_out[t] = fpc_pcm16(mem);
            if (exc_out) exc_out[t] = (uint8_t)exc;
            if (pcm_f_out) pcm_f_out[t] = pcm;
        }
        skip = 0;
    }
    free(cfeat);
}

/* the vocoder's pdf shaping + tail cut of ONE pdf, exactly as orc_lpcnet_synthesize applies it (src/train.py:79-92
 * without the divisions): out[v] = max(p'[v] - .002 * S1, 0) with p' = p * p**e (e = max(0, 1.5 corr - .5)) and
 * S1 = balanced total of p' when e > 0, S1 = 1 (not computed) otherwise.  Test hook for golden G9. */
EXPORT void orc_shape_cut(const float* p_in, float pitch_corr, float* out) {
    const float shape_e = fpc_shape_exponent(pitch_corr);
    float p[256], tmp[256];
    for (int v = 0; v < 256; ++v) p[v] = shape_e > 0.0f ? fpc_shape_pow(p_in[v], shape_e) : p_in[v];
    float thr = 0.002f;
    if (shape_e > 0.0f) {
        memcpy(tmp, p, sizeof p);
        thr = 0.002f * tree_reduce(tmp, 256);
    }
    for (int v = 0; v < 256; ++v) {
        const float d = p[v] - thr;
        out[v] = d > 0.0f ? d : 0.0f;
    }
}

/* debugging/known-answer helper: pdf of one step from given node probabilities */
EXPORT void orc_tree_pdf(const float* q, float* p) {
    for (int v = 0; v < 256; ++v) p[v] = tree_leaf_prob(q, v);
}

/* ====================================================================
 * Codebook training: src/quantization/cb_func.py
 * ==================================================================== */

/* Training vectors arrive as float32 (train_cb.py:170-178) for the first stage and as float64 for
 * later stages (the residual `qr - r` of train_cb.py:191-192): data_f64 selects how `data` is read. */
static double cb_at(const void* data, int data_f64, size_t k) {
    return data_f64 ? ((const double*)data)[k] : (double)((const float*)data)[k];
}

/* find_nearest (cb_func.py:56-68): dist = np.sum((data - codebook)**2, -1) in float64 (float32 data is
 * broadcast against the float64 codebook), np.argmin over entries = first minimum.  17 dimensions:
 * numpy's pairwise association as in dist17 above. */
EXPORT int orc_cb_find_nearest(const void* data, int data_f64, int nv, int nd, const double* cb, int e, int* idx) {
    if (nd != ORC_NDIM) return -1;
    for (int i = 0; i < nv; ++i) {
        double x[ORC_NDIM];
        for (int j = 0; j < ORC_NDIM; ++j) x[j] = cb_at(data, data_f64, (size_t)i * nd + j);
        double best = INFINITY;
        int bi = 0;
        for (int n = 0; n < e; ++n) {
            const double d = dist17(x, cb + (size_t)n * nd);
            if (n == 0 || d < best) { /* np.argmin: first index of the minimum (NaN-free input) */
                best = d;
                bi = n;
            }
        }
        idx[i] = bi;
    }
    return 0;
}

/* update (cb_func.py:71-100): nearest entry per vector, then per entry the float64 sum of its
 * members accumulated in index order, divided by (count + 1e-20).  count_out may be NULL. */
EXPORT int orc_cb_update(const void* data, int data_f64, int nv, int nd, const double* cb_in, int e, double* cb_out,
                         double* count_out) {
    int* idx = (int*)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
    if (orc_cb_find_nearest(data, data_f64, nv, nd, cb_in, e, idx) != 0) {
        free(idx);
        return -1;
    }
    double* count = (double*)calloc((size_t)e, sizeof(double));
    for (size_t k = 0; k < (size_t)e * nd; ++k) cb_out[k] = 0.0;
    for (int i = 0; i < nv; ++i) {
        const int n = idx[i];
        count[n] += 1.0;
        for (int j = 0; j < nd; ++j) cb_out[(size_t)n * nd + j] += cb_at(data, data_f64, (size_t)i * nd + j);
    }
    for (int n = 0; n < e; ++n)
        for (int j = 0; j < nd; ++j) cb_out[(size_t)n * nd + j] /= count[n] + 1e-20;
    if (count_out) memcpy(count_out, count, sizeof(double) * (size_t)e);
    free(count);
    free(idx);
    return 0;
}

/* np.mean(data, 0) of a C-contiguous (nv, nd) array (cb_func.py:34): accumulation row after row in the
 * array's own precision, division by nv in that precision; widened to float64 on assignment */
EXPORT void orc_cb_mean0(const void* data, int data_f64, int nv, int nd, double* out) {
    for (int j = 0; j < nd; ++j) {
        if (data_f64) {
            const double* d = (const double*)data;
            double s = d[j];
            for (int i = 1; i < nv; ++i) s = s + d[(size_t)i * nd + j];
            out[j] = s / (double)nv;
        } else {
            const float* d = (const float*)data;
            float s = d[j];
            for (int i = 1; i < nv; ++i) s = s + d[(size_t)i * nd + j];
            out[j] = (double)(s / (float)nv);
        }
    }
}

/* ====================================================================
 * Predictor training step: src/train_frame.py:53-120, the live branch (batch_idx <= 10):
 * teacher-forced Wavernn.forward over the whole window, nn.MSELoss(feat_out[:, :-1], feat[:, 1:, :18]),
 * autograd, torch.optim.Adam(lr).  Restated with explicit evaluation orders the kernels follow:
 *  - forward = pred_step above (segmented chains), activations kept per frame;
 *  - backward through time per utterance; transposed products (W^T d) as fmaf chains over the rows from 0;
 *  - weight gradients = 8 contiguous segments of the samples n = b*L + t, each an fmaf chain from 0 (what an f32
 *    MFMA accumulates), added as a balanced tree; bias gradients = plain sums over n;
 *  - Adam as torch's single-tensor path (lerp for the first moment), scalars in double on the host.
 * ==================================================================== */
typedef struct {
    float *w1_ih, *w1_hh, *b1_ih, *b1_hh, *w2_ih, *w2_hh, *b2_ih, *b2_hh, *fc_w, *fc_b;
} orc_params; /* torch layouts: weight [3H][K], bias [3H], fc_w [fc][H2], fc_b [fc] */

static void gru_fwd_save(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x,
                         int in, const float* h_prev, int H, float* r, float* z, float* n, float* hn, float* h,
                         float* gi, float* gh) {
    matvec_seg(w_ih, b_ih, x, 3 * H, in, fpc_segments(in), gi);
    matvec_seg(w_hh, b_hh, h_prev, 3 * H, H, fpc_segments(H), gh);
    for (int i = 0; i < H; ++i) {
        r[i] = fpc_sigmoidf(gi[i] + gh[i]);
        z[i] = fpc_sigmoidf(gi[H + i] + gh[H + i]);
        hn[i] = gh[2 * H + i];
        n[i] = fpc_tanhf(fmaf(r[i], hn[i], gi[2 * H + i]));
        h[i] = fmaf(z[i], h_prev[i] - n[i], n[i]);
    }
}

/* d_in[k] = sum_row W[row][k] d[row]: the rows in fpc_segments(rows) contiguous segments, each an fmaf chain
 * from 0, segment sums added as a balanced tree (one thread per (4 adjacent k, segment) in the kernel) */
static void matvec_t(const float* W, const float* d, int rows, int cols, float* out) {
    const int S = fpc_segments(rows), len = rows / S;
    for (int k = 0; k < cols; ++k) {
        float part[8];
        for (int sg = 0; sg < S; ++sg) {
            float acc = 0.0f;
            for (int r = sg * len; r < (sg + 1) * len; ++r) acc = fmaf(W[(size_t)r * cols + k], d[r], acc);
            part[sg] = acc;
        }
        for (int st = 1; st < S; st <<= 1)
            for (int q = 0; q + st < S; q += 2 * st) part[q] = part[q] + part[q + st];
        out[k] = part[0];
    }
}

static void gru_bwd(const float* w_ih, const float* w_hh, int in, int H, const float* dh, const float* r,
                    const float* z, const float* n, const float* hn, const float* h_prev, float* dgi, float* dgh,
                    float* dh_prev, float* dx) {
    for (int i = 0; i < H; ++i) {
        const float dn_raw = dh[i] * (1.0f - z[i]);
        const float dnpre = dn_raw * fmaf(-n[i], n[i], 1.0f);
        const float dz_raw = dh[i] * (h_prev[i] - n[i]);
        const float dzpre = dz_raw * (z[i] * (1.0f - z[i]));
        const float drpre = (dnpre * hn[i]) * (r[i] * (1.0f - r[i]));
        dgi[i] = drpre;
        dgi[H + i] = dzpre;
        dgi[2 * H + i] = dnpre;
        dgh[i] = drpre;
        dgh[H + i] = dzpre;
        dgh[2 * H + i] = dnpre * r[i];
    }
    matvec_t(w_hh, dgh, 3 * H, H, dh_prev);
    for (int i = 0; i < H; ++i) dh_prev[i] = fmaf(dh[i], z[i], dh_prev[i]);
    if (dx) matvec_t(w_ih, dgi, 3 * H, in, dx);
}

/* dW[row][k] = sum over the samples n of d[n][row] * a[n][k]: the samples in 8 contiguous segments (lengths a
 * multiple of 4), each an fmaf chain from 0 (what an f32 MFMA accumulates), the segment sums added as a
 * balanced tree;  db[row] = plain sum over n of d[n][row] */
#define ORC_GSEG 8
static void grad_w(const float* d, const float* a, size_t N, int rows, int cols, float* dW, float* db) {
    const size_t seg = ((N + 4 * ORC_GSEG - 1) / (4 * ORC_GSEG)) * 4;
    for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < cols; ++k) {
            float part[ORC_GSEG];
            for (int sg = 0; sg < ORC_GSEG; ++sg) {
                float acc = 0.0f;
                const size_t n1 = (sg + 1) * seg < N ? (sg + 1) * seg : N;
                for (size_t n = sg * seg; n < n1; ++n) acc = fmaf(d[n * rows + r], a[n * cols + k], acc);
                part[sg] = acc;
            }
            dW[(size_t)r * cols + k] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        }
        float s = 0.0f;
        for (size_t n = 0; n < N; ++n) s = s + d[n * rows + r];
        db[r] = s;
    }
}

static void adam(float* p, float* m, float* v, const float* g, size_t n, float step_size, float bc2_sqrt) {
    for (size_t i = 0; i < n; ++i) {
        m[i] = fmaf(0.1f, g[i] - m[i], m[i]);              /* exp_avg.lerp_(grad, 1 - beta1) */
        v[i] = fmaf(0.001f, g[i] * g[i], v[i] * 0.999f);   /* exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2) */
        const float denom = sqrtf(v[i]) / bc2_sqrt + 1e-8f;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* One training step.  in/h1/h2/fc = 20/384/128/18 style sizes; params updated in place; m, v: Adam moments
 * (10 arrays each, same shapes); step counts from 1.  grads (10 arrays) may be NULL.  Returns the loss. */
EXPORT float orc_train_step(int in, int H1, int H2, int F, orc_params* P, orc_params* M, orc_params* V,
                            orc_params* G, const float* feat, int B, int L, double lr, int step) {
    const size_t N = (size_t)B * L;
    float* x = (float*)calloc(N * in, 4);
    float *h1p = calloc(N * H1, 4), *r1 = calloc(N * H1, 4), *z1 = calloc(N * H1, 4), *n1 = calloc(N * H1, 4),
          *hn1 = calloc(N * H1, 4), *h1 = calloc(N * H1, 4);
    float *h2p = calloc(N * H2, 4), *r2 = calloc(N * H2, 4), *z2 = calloc(N * H2, 4), *n2 = calloc(N * H2, 4),
          *hn2 = calloc(N * H2, 4), *h2 = calloc(N * H2, 4), *relu = calloc(N * H2, 4);
    float *th = calloc(N * F, 4), *dpre = calloc(N * F, 4);
    float *dgi1 = calloc(N * 3 * H1, 4), *dgh1 = calloc(N * 3 * H1, 4), *dgi2 = calloc(N * 3 * H2, 4),
          *dgh2 = calloc(N * 3 * H2, 4);
    float* gi = (float*)malloc(4 * 3 * (size_t)(H1 > H2 ? H1 : H2));
    float* gh = (float*)malloc(4 * 3 * (size_t)(H1 > H2 ? H1 : H2));
    const int Sfc = (H2 % 8 == 0 && H2 >= 64) ? 8 : 1;
    /* ---- forward, activations kept ---- */
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < L; ++t) {
            const size_t n = (size_t)b * L + t;
            memcpy(x + n * in, feat + n * in, 4 * (size_t)in);
            if (t > 0) {
                memcpy(h1p + n * H1, h1 + (n - 1) * H1, 4 * (size_t)H1);
                memcpy(h2p + n * H2, h2 + (n - 1) * H2, 4 * (size_t)H2);
            }
            gru_fwd_save(P->w1_ih, P->w1_hh, P->b1_ih, P->b1_hh, x + n * in, in, h1p + n * H1, H1, r1 + n * H1,
                         z1 + n * H1, n1 + n * H1, hn1 + n * H1, h1 + n * H1, gi, gh);
            gru_fwd_save(P->w2_ih, P->w2_hh, P->b2_ih, P->b2_hh, h1 + n * H1, H1, h2p + n * H2, H2, r2 + n * H2,
                         z2 + n * H2, n2 + n * H2, hn2 + n * H2, h2 + n * H2, gi, gh);
            for (int i = 0; i < H2; ++i) relu[n * H2 + i] = h2[n * H2 + i] > 0.0f ? h2[n * H2 + i] : 0.0f;
            float pre[64];
            matvec_seg(P->fc_w, P->fc_b, relu + n * H2, F, H2, Sfc, pre);
            for (int o = 0; o < F; ++o) th[n * F + o] = fpc_tanhf(pre[o]);
        }
    /* ---- loss (float64 accumulation per utterance, then over utterances) and dL/dpre ---- */
    const double cnt = (double)B * (L - 1) * F;
    const float scale = (float)(2.0 / cnt);
    double loss = 0.0;
    for (int b = 0; b < B; ++b) {
        double lb = 0.0;
        for (int t = 0; t + 1 < L; ++t) {
            const size_t n = (size_t)b * L + t;
            for (int o = 0; o < F; ++o) {
                const float y = th[n * F + o] + th[n * F + o];
                const float diff = y - feat[(n + 1) * in + o];
                lb += (double)diff * (double)diff;
                const float g = diff * scale;
                dpre[n * F + o] = (g + g) * fmaf(-th[n * F + o], th[n * F + o], 1.0f);
            }
        }
        loss += lb;
    }
    loss /= cnt;
    /* ---- backward through time ---- */
    float *dh1n = calloc(H1, 4), *dh2n = calloc(H2, 4), *dh = calloc(H1 > H2 ? H1 : H2, 4), *dx2 = calloc(H1, 4),
          *tmp = calloc(H1 > H2 ? H1 : H2, 4);
    for (int b = 0; b < B; ++b) {
        memset(dh1n, 0, 4 * (size_t)H1);
        memset(dh2n, 0, 4 * (size_t)H2);
        for (int t = L - 1; t >= 0; --t) {
            const size_t n = (size_t)b * L + t;
            matvec_t(P->fc_w, dpre + n * F, F, H2, tmp); /* d relu */
            for (int i = 0; i < H2; ++i) dh[i] = (h2[n * H2 + i] > 0.0f ? tmp[i] : 0.0f) + dh2n[i];
            gru_bwd(P->w2_ih, P->w2_hh, H1, H2, dh, r2 + n * H2, z2 + n * H2, n2 + n * H2, hn2 + n * H2, h2p + n * H2,
                    dgi2 + n * 3 * H2, dgh2 + n * 3 * H2, dh2n, dx2);
            for (int i = 0; i < H1; ++i) dh[i] = dx2[i] + dh1n[i];
            gru_bwd(P->w1_ih, P->w1_hh, in, H1, dh, r1 + n * H1, z1 + n * H1, n1 + n * H1, hn1 + n * H1, h1p + n * H1,
                    dgi1 + n * 3 * H1, dgh1 + n * 3 * H1, dh1n, NULL);
        }
    }
    /* ---- parameter gradients ---- */
    orc_params g;
    const size_t sz[10] = {(size_t)3 * H1 * in, (size_t)3 * H1 * H1, (size_t)3 * H1, (size_t)3 * H1, (size_t)3 * H2 * H1,
                           (size_t)3 * H2 * H2, (size_t)3 * H2, (size_t)3 * H2, (size_t)F * H2, (size_t)F};
    float** gp = (float**)&g;
    for (int k = 0; k < 10; ++k) gp[k] = (float*)calloc(sz[k], 4);
    grad_w(dgi1, x, N, 3 * H1, in, g.w1_ih, g.b1_ih);
    grad_w(dgh1, h1p, N, 3 * H1, H1, g.w1_hh, g.b1_hh);
    grad_w(dgi2, h1, N, 3 * H2, H1, g.w2_ih, g.b2_ih);
    grad_w(dgh2, h2p, N, 3 * H2, H2, g.w2_hh, g.b2_hh);
    grad_w(dpre, relu, N, F, H2, g.fc_w, g.fc_b);
    /* ---- Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8) ---- */
    const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    float **pp = (float**)P, **mp = (float**)M, **vp = (float**)V, **go = G ? (float**)G : NULL;
    for (int k = 0; k < 10; ++k) {
        adam(pp[k], mp[k], vp[k], gp[k], sz[k], step_size, bc2_sqrt);
        if (go) memcpy(go[k], gp[k], 4 * sz[k]);
        free(gp[k]);
    }
    float* fr[] = {x, h1p, r1, z1, n1, hn1, h1, h2p, r2, z2, n2, hn2, h2, relu, th, dpre, dgi1, dgh1, dgi2, dgh2, gi,
                   gh, dh1n, dh2n, dh, dx2, tmp};
    for (size_t k = 0; k < sizeof fr / sizeof fr[0]; ++k) free(fr[k]);
    return (float)loss;
}

/* ---- the two loops of oracle/kmeans1d_oracle.py's Lloyd iteration that numpy makes slow at the production sizes
 * (k = 256, n = 400 000: a 400 000 x 256 float64 matrix per E-step, np.add.at per block in the M-step).  The SAME
 * operations in the same order as kmeans1d_oracle._assign / _sums (tests/test_host_cpu.py holds the two equal); the numpy
 * functions stay the definition.  scikit-learn (third party, _k_means_lloyd.pyx): label = argmin_j c_j^2 + (-2)(x c_j), the
 * first minimum on ties. ---- */
EXPORT void orc_km_assign(const double* x, long long n, const double* centers, int k, int* labels) {
    for (long long i = 0; i < n; ++i) {
        int best = 0;
        double bd = centers[0] * centers[0] + (-2.0 * (x[i] * centers[0]));
        for (int j = 1; j < k; ++j) {
            const double d = centers[j] * centers[j] + (-2.0 * (x[i] * centers[j]));
            if (d < bd) {
                bd = d;
                best = j;
            }
        }
        labels[i] = best;
    }
}
/* cluster sums and counts: per block of `ch` points the members' values one after the other in index order, then the blocks'
 * partial sums one after the other (tot = tot + ps) */
EXPORT void orc_km_sums(const double* x, const int* labels, long long n, int k, int ch, double* tot, double* cnt,
                        double* ps /* [k] scratch */, double* pc /* [k] scratch */) {
    for (int j = 0; j < k; ++j) tot[j] = cnt[j] = 0.0;
    for (long long i0 = 0; i0 < n; i0 += ch) {
        for (int j = 0; j < k; ++j) ps[j] = pc[j] = 0.0;
        const long long i1 = i0 + ch < n ? i0 + ch : n;
        for (long long i = i0; i < i1; ++i) {
            ps[labels[i]] = ps[labels[i]] + x[i];
            pc[labels[i]] = pc[labels[i]] + 1.0;
        }
        for (int j = 0; j < k; ++j) {
            tot[j] = tot[j] + ps[j];
            cnt[j] = cnt[j] + pc[j];
        }
    }
}
