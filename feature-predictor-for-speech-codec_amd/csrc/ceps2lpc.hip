// ceps2lpc.hip -- Bark cepstrum -> 16 LPC coefficients, batched over frames (gfx950).
//
// Replaces ceps2lpc_v of the reference (src/ceps2lpc/ceps2lpc_vct.py:122-162 with idct :35-43,
// interp_band_gain :45-57, _celt_lpc_s :60-88), which runs a Python-level Levinson loop per
// frame on the CPU.  Here one lane owns one frame; the 161-bin spectrum of each lane is
// staged in LDS (row stride 161 dwords: conflict free), the 320-point inverse real FFT is
// evaluated for the 17 needed lags as a float64 cosine sum.  Operation order is identical to
// oracle/fpc_oracle.c::ceps2lpc_row so results match the CPU oracle bit for bit.
#include "fpc_common.h"
#include <mutex>

namespace {

constexpr int NB = 18, FREQ = 161, WIN = 320, ORDER = 16, NLAG = 17, TPB = 64;

__constant__ float c_comp[NB] = {0.8f, 1.0f, 1.0f,      1.0f,  1.0f,  1.0f, 1.0f,      1.0f,     0.666667f,
                                 0.5f, 0.5f, 0.5f,      0.333333f, 0.25f, 0.25f, 0.2f, 0.166667f, 0.173913f};
__constant__ int c_eband[NB] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};

__global__ __launch_bounds__(TPB) void k_ceps2lpc(const float* __restrict__ ceps, int N, int stride,
                                                  const float* __restrict__ dct,   // [18][18]
                                                  const double* __restrict__ cosT, // [17][161]
                                                  float idct_scale, float floor_add,
                                                  float* __restrict__ lpc_out, float* __restrict__ e_out,
                                                  float* __restrict__ rc_out) {
    __shared__ float Xs[TPB][FREQ];
    const int n = blockIdx.x * TPB + threadIdx.x;
    if (n >= N) return;
    float* X = Xs[threadIdx.x];
    const float* c = ceps + (size_t)n * stride;
    float in[NB], Ex[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) in[j] = c[j] + (j == 0 ? 4.0f : 0.0f);
    for (int i = 0; i < NB; ++i) {
        float sm = 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float t = in[j] * dct[i * NB + j];
            sm = sm + t;
        }
        Ex[i] = fpc_exp10f(sm * idct_scale) * c_comp[i];
    }
    for (int m = 0; m < FREQ; ++m) X[m] = 0.0f;
    for (int i = 0; i < NB - 1; ++i) {
        const int bs = (c_eband[i + 1] - c_eband[i]) * 4;
        for (int j = 0; j < bs; ++j) {
            const float frac = (float)((double)j / (double)bs);
            const float a = (1.0f - frac) * Ex[i];
            const float b = frac * Ex[i + 1];
            X[c_eband[i] * 4 + j] = a + b;
        }
    }
    float ac[NLAG];
    for (int k = 0; k < NLAG; ++k) {
        double acc = (double)X[0];
        for (int m = 1; m < FREQ - 1; ++m) {
            const double t = 2.0 * (double)X[m] * cosT[k * FREQ + m];
            acc = acc + t;
        }
        const double tl = (double)X[FREQ - 1] * cosT[k * FREQ + FREQ - 1];
        acc = acc + tl;
        ac[k] = (float)(acc / (double)WIN);
    }
    {
        const float t = ac[0] * 0.0001f;
        const float u = t + floor_add;
        ac[0] = ac[0] + u;
    }
#pragma unroll
    for (int i = 1; i < NLAG; ++i) ac[i] = ac[i] * (float)(1.0 - 0.00006 * i * i);
    float lpc[ORDER], rc[ORDER];
#pragma unroll
    for (int i = 0; i < ORDER; ++i) {
        lpc[i] = 0.0f;
        rc[i] = 0.0f;
    }
    float error = ac[0];
    if (ac[0] != 0.0f) {
        bool done = false;
#pragma unroll
        for (int i = 0; i < ORDER; ++i) {
            if (!done) {
                float rr = 0.0f;
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    const float t = lpc[j] * ac[i - j];
                    rr = rr + t;
                }
                rr = rr + ac[i + 1];
                const float r = -rr / error;
                rc[i] = r;
                lpc[i] = r;
#pragma unroll
                for (int j = 0; j < (i + 1) / 2; ++j) {
                    const float t1 = lpc[j], t2 = lpc[i - 1 - j];
                    const float m1 = r * t2, m2 = r * t1;
                    lpc[j] = t1 + m1;
                    lpc[i - 1 - j] = t2 + m2;
                }
                const float rr2 = r * r;
                const float dec = rr2 * error;
                error = error - dec;
                if (error < ac[0] / 1024.0f) done = true;
                if (error < 0.001f * ac[0]) done = true;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ORDER; ++i) {
        lpc_out[(size_t)n * ORDER + i] = lpc[i];
        if (rc_out) rc_out[(size_t)n * ORDER + i] = rc[i];
    }
    if (e_out) e_out[n] = error;
}

struct Tables {
    fpc::DevBuf dct, cosT;
    bool ready = false;
};
// heap-allocated and never freed: a static DevBuf's destructor would call hipFree during static destruction,
// after the HIP runtime may already be gone; a few KB per device live until the process ends
Tables* g_tab = nullptr;
std::mutex g_mu;

}  // namespace

extern "C" int fpc_ceps2lpc(const float* ceps_dev, int N, int stride, float* lpc_dev, float* e_dev,
                            float* rc_dev, fpc_stream s) {
    FPC_REQUIRE(ceps_dev && lpc_dev, "fpc_ceps2lpc: null argument");
    FPC_REQUIRE(N >= 0 && stride >= NB, "fpc_ceps2lpc: bad shape N=%d stride=%d", N, stride);
    if (!fpc::have_device()) {
        fpc::set_error("fpc_ceps2lpc: no HIP device (libfpcodec has no CPU fallback)");
        return FPC_ERR_NO_DEVICE;
    }
    if (N == 0) return FPC_OK;
    int dev = 0;
    FPC_HIP(hipGetDevice(&dev));
    FPC_REQUIRE(dev < 16, "fpc_ceps2lpc: device index %d unsupported", dev);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_tab) g_tab = new Tables[16];
        Tables& t = g_tab[dev];
        if (!t.ready) {  // same table formulas as the oracle (ceps2lpc_vct.py:27-32)
            std::vector<float> dct(NB * NB);
            for (int i = 0; i < NB; ++i)
                for (int j = 0; j < NB; ++j) {
                    const float arg = (float)((i + 0.5) * j * M_PI / NB);
                    float c = (float)cos((double)arg);
                    if (j == 0) c *= (float)sqrt(0.5);
                    dct[i * NB + j] = c;
                }
            std::vector<double> cs(NLAG * FREQ);
            for (int k = 0; k < NLAG; ++k)
                for (int m = 0; m < FREQ; ++m)
                    cs[k * FREQ + m] = cos(2.0 * M_PI * (double)((m * k) % WIN) / (double)WIN);
            FPC_HIP(t.dct.upload(dct));
            FPC_HIP(t.cosT.upload(cs));
            t.ready = true;
        }
    }
    const Tables& t = g_tab[dev];
    hipLaunchKernelGGL(k_ceps2lpc, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, static_cast<hipStream_t>(s),
                       ceps_dev, N, stride, t.dct.as<float>(), t.cosT.as<double>(),
                       (float)sqrt(2.0 / NB), (float)(320.0 / 12.0 / 38.0), lpc_dev, e_dev, rc_dev);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}
