// cb_train.hip -- codebook training primitives (splitting LBG / k-means), hand-written HIP for gfx950.
//
// Reference interfaces replaced (paths under /root/reference/src):
//   find_nearest   quantization/cb_func.py:56-68    float64 squared distances, first minimum
//   update         quantization/cb_func.py:71-100   nearest entry, float64 sums in index order, / (count + 1e-20)
//   np.mean(data, 0) of vq_train  quantization/cb_func.py:34
// (the splitting schedule of vq_train, cb_func.py:28-54, stays on the host: it draws from numpy's global RNG)
// Training vectors are float32 for the first stage (train_cb.py:170-178) and float64 for later stages (the
// residual `qr - r`, train_cb.py:191-192): the kernels that read them are instantiated for both.
//
// The reference accumulates `codebook[n] += data[i]` for i = 0..nv-1; floating-point addition does not commute
// with reordering, so the sums are taken in exactly that order: the assignment is turned into a STABLE counting
// sort by entry (per-block histograms -> offsets -> in-order scatter), after which the members of an entry are
// contiguous (the rows are moved into a staging buffer) and ascending, and one 17-lane group adds them up one
// after the other in float64.
#include "fpc_common.h"

namespace {

constexpr int ND = 17;          // code_dims of the production codebooks (train_cb.py: cfg['code_dims'])
constexpr int AT = 256;         // threads of the assignment kernel
constexpr int VPT = 2;          // vectors per thread there (entry loads amortised over both)
constexpr int CH = 2048;        // vectors per histogram / scatter block
constexpr int MAXE = 4096;      // entries (LDS histogram of CH-blocks: 16 KB)

// numpy's pairwise association of a contiguous 17-term float64 sum (as dist17 of predictor.hip)
__device__ __forceinline__ double sqdist17(const double* x, const double* __restrict__ c) {
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
    return res + dd;
}

// nearest entry of every vector: thread = VPT vectors in registers (float64), entries streamed by
// wave-uniform addresses (scalar loads / broadcast), strict `<` keeps the first minimum like np.argmin
template <class T>
__global__ __launch_bounds__(AT) void k_cb_assign(const T* __restrict__ data, int nv, const double* __restrict__ cb,
                                                  int e, int* __restrict__ idx) {
    const int i0 = (blockIdx.x * AT + threadIdx.x) * VPT;
    double x[VPT][ND];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int i = i0 + v < nv ? i0 + v : nv - 1;
#pragma unroll
        for (int j = 0; j < ND; ++j) x[v][j] = (double)data[(size_t)i * ND + j];
    }
    double best[VPT];
    int bi[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        best[v] = sqdist17(x[v], cb);
        bi[v] = 0;
    }
    for (int n = 1; n < e; ++n) {
        const double* c = cb + (size_t)n * ND;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const double d = sqdist17(x[v], c);
            if (d < best[v]) {
                best[v] = d;
                bi[v] = n;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VPT; ++v)
        if (i0 + v < nv) idx[i0 + v] = bi[v];
}

// histogram of one block of CH consecutive vectors
__global__ __launch_bounds__(256) void k_cb_hist(const int* __restrict__ idx, int nv, int e, int* __restrict__ hist) {
    __shared__ int h[MAXE];
    for (int c = threadIdx.x; c < e; c += 256) h[c] = 0;
    __syncthreads();
    const int i0 = blockIdx.x * CH;
    for (int k = threadIdx.x; k < CH && i0 + k < nv; k += 256) atomicAdd(&h[idx[i0 + k]], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < e; c += 256) hist[(size_t)blockIdx.x * e + c] = h[c];
}

// hist[b][c] -> first output slot of block b's members of entry c; count[c]; base[c]
__global__ __launch_bounds__(1024) void k_cb_offsets(int* __restrict__ hist, int nb, int e, int* __restrict__ base,
                                                     double* __restrict__ count) {
    __shared__ int tot[MAXE];
    __shared__ int carry;
    for (int c = threadIdx.x; c < e; c += 1024) {
        int run = 0;
        for (int b = 0; b < nb; ++b) {
            const int h = hist[(size_t)b * e + c];
            hist[(size_t)b * e + c] = run;
            run += h;
        }
        tot[c] = run;
        if (count) count[c] = (double)run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive scan over entries (e <= 4096: negligible next to the rest)
        int run = 0;
        for (int c = 0; c < e; ++c) {
            const int t = tot[c];
            tot[c] = run;
            run += t;
        }
        carry = run;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < e; c += 1024) {
        base[c] = tot[c];
        for (int b = 0; b < nb; ++b) hist[(size_t)b * e + c] += tot[c];
    }
    if (threadIdx.x == 0) base[e] = carry;
}

// stable scatter: one wave per block of CH vectors walks them in index order, 64 at a time; lanes with the
// same entry take consecutive slots in lane order.  The rows themselves are moved (not just their indices), so
// that the summation below streams contiguous memory instead of chasing two dependent gathers per member.
template <class T>
__global__ __launch_bounds__(64) void k_cb_scatter(const T* __restrict__ data, const int* __restrict__ idx, int nv, int e,
                                                   const int* __restrict__ blockbase, T* __restrict__ sorted) {
    __shared__ int cur[MAXE];
    const int lane = threadIdx.x;
    for (int c = lane; c < e; c += 64) cur[c] = blockbase[(size_t)blockIdx.x * e + c];
    __syncthreads();
    const int i0 = blockIdx.x * CH;
    for (int k = 0; k < CH && i0 + k < nv; k += 64) {
        const int i = i0 + k + lane;
        const bool valid = i < nv;
        const int c = valid ? idx[i] : -1;
        unsigned long long todo = __ballot(valid);
        int pos = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int cl = __shfl(c, leader);
            const unsigned long long same = __ballot(c == cl);
            if (c == cl) pos = cur[cl] + __popcll(same & ((1ull << lane) - 1ull));
            __syncthreads();  // single wave: orders the reads of cur[cl] before its update
            if (lane == leader) cur[cl] += __popcll(same);
            __syncthreads();
            todo &= ~same;
        }
        if (valid) {
            T row[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) row[j] = data[(size_t)i * ND + j];
#pragma unroll
            for (int j = 0; j < ND; ++j) sorted[(size_t)pos * ND + j] = row[j];
        }
    }
}

// per entry: float64 sum of its members in ascending index order (contiguous rows of `sorted`),
// / (count + 1e-20).  The add chain is strictly sequential -- that is the specification -- and the reference's
// splitting schedule leaves cells with half of all vectors, so one chain per dimension is as parallel as it
// gets; what can overlap is the memory: a 256-thread block per entry streams the rows through a double
// buffer in LDS (coalesced copies of chunk k+1 in flight while 17 lanes add chunk k).
template <class T>
__global__ __launch_bounds__(256) void k_cb_sum(const T* __restrict__ sorted, const int* __restrict__ base, int e,
                                                double* __restrict__ cb_out) {
    constexpr int CHK = sizeof(T) == 4 ? 512 : 256;  // rows per chunk: 2 x 34 KB of LDS either way
    constexpr int PER = CHK * ND / 256;               // elements per thread per chunk
    __shared__ T buf[2][CHK * ND];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int b0 = base[c], n = base[c + 1] - b0;
    const T* src = sorted + (size_t)b0 * ND;
    const size_t total = (size_t)n * ND;
    const int nchunks = (n + CHK - 1) / CHK;
    double acc = 0.0;
    T stage[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const size_t k = (size_t)u * 256 + tid;
        if (k < total) buf[0][k] = src[k];
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const size_t nxt = (size_t)(ch + 1) * CHK * ND;
        if (ch + 1 < nchunks) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const size_t k = nxt + (size_t)u * 256 + tid;
                stage[u] = k < total ? src[k] : (T)0;
            }
        }
        if (tid < ND) {
            const T* b = buf[ch & 1] + tid;
            const int rows = n - ch * CHK < CHK ? n - ch * CHK : CHK;
            int r = 0;
            for (; r + 16 <= rows; r += 16) {
                T v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = b[(r + u) * ND];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = acc + (double)v[u];
            }
            for (; r < rows; ++r) acc = acc + (double)b[r * ND];
        }
        if (ch + 1 < nchunks) {
#pragma unroll
            for (int u = 0; u < PER; ++u) buf[(ch + 1) & 1][(size_t)u * 256 + tid] = stage[u];
        }
        __syncthreads();
    }
    if (tid < ND) cb_out[(size_t)c * ND + tid] = acc / ((double)n + 1e-20);
}

// np.mean(data, 0): accumulation row after row and division in the array's own precision
template <class T>
__global__ __launch_bounds__(64) void k_cb_mean0(const T* __restrict__ data, int nv, double* __restrict__ out) {
    const int dim = threadIdx.x;
    if (dim >= ND) return;
    T s = data[dim];
    int i = 1;
    for (; i + 32 <= nv; i += 32) {
        T v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = data[(size_t)(i + u) * ND + dim];
#pragma unroll
        for (int u = 0; u < 32; ++u) s = s + v[u];
    }
    for (; i < nv; ++i) s = s + data[(size_t)i * ND + dim];
    out[dim] = (double)(s / (T)nv);
}

struct CbWs {
    int *idx, *hist, *base;
    void* sorted;  // [nv][ND] rows grouped by entry, in index order inside each entry
};
inline size_t a256(size_t n) { return (n + 255) / 256 * 256; }
inline int nblocks(int nv) { return (nv + CH - 1) / CH; }
CbWs carve_cb(void* ws, int nv, int e) {
    char* p = static_cast<char*>(ws);
    CbWs w;
    w.idx = reinterpret_cast<int*>(p);
    p += a256(sizeof(int) * (size_t)nv);
    w.sorted = p;
    p += a256(sizeof(double) * (size_t)nv * ND);
    w.hist = reinterpret_cast<int*>(p);
    p += a256(sizeof(int) * (size_t)nblocks(nv) * e);
    w.base = reinterpret_cast<int*>(p);
    return w;
}

int check_shape(const char* fn, int nv, int nd, int e) {
    FPC_REQUIRE(fpc::have_device(), "%s: no HIP device (this library has no CPU fallback)", fn);
    FPC_REQUIRE(nd == ND, "%s: %d dimensions requested, this build trains %d-dimensional codebooks only", fn, nd, ND);
    FPC_REQUIRE(nv > 0 && e > 0 && e <= MAXE, "%s: bad shape nv=%d entries=%d (entries <= %d)", fn, nv, e, MAXE);
    return FPC_OK;
}

}  // namespace

extern "C" long long fpc_cb_workspace_bytes(int nv, int e) {
    if (nv <= 0 || e <= 0) return 0;
    return (long long)(a256(sizeof(int) * (size_t)nv) + a256(sizeof(double) * (size_t)nv * ND) +
                       a256(sizeof(int) * (size_t)nblocks(nv) * e) +
                       a256(sizeof(int) * (size_t)(e + 1)));
}

extern "C" int fpc_cb_find_nearest(const void* data_dev, int data_f64, int nv, int nd, const double* cb_dev, int e,
                                   int* idx_dev, fpc_stream s) {
    FPC_REQUIRE(data_dev && cb_dev && idx_dev, "fpc_cb_find_nearest: null argument");
    if (int rc = check_shape("fpc_cb_find_nearest", nv, nd, e)) return rc;
    const dim3 grid((nv + AT * VPT - 1) / (AT * VPT));
    hipStream_t st = static_cast<hipStream_t>(s);
    if (data_f64)
        hipLaunchKernelGGL(k_cb_assign<double>, grid, dim3(AT), 0, st, static_cast<const double*>(data_dev), nv, cb_dev, e,
                           idx_dev);
    else
        hipLaunchKernelGGL(k_cb_assign<float>, grid, dim3(AT), 0, st, static_cast<const float*>(data_dev), nv, cb_dev, e,
                           idx_dev);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}

extern "C" int fpc_cb_update(const void* data_dev, int data_f64, int nv, int nd, const double* cb_in_dev, int e,
                             double* cb_out_dev, double* count_dev, void* workspace_dev, fpc_stream s) {
    FPC_REQUIRE(data_dev && cb_in_dev && cb_out_dev && workspace_dev, "fpc_cb_update: null argument");
    if (int rc = check_shape("fpc_cb_update", nv, nd, e)) return rc;
    hipStream_t st = static_cast<hipStream_t>(s);
    const CbWs w = carve_cb(workspace_dev, nv, e);
    const int nb = nblocks(nv);
    const dim3 grid((nv + AT * VPT - 1) / (AT * VPT));
    if (data_f64)
        hipLaunchKernelGGL(k_cb_assign<double>, grid, dim3(AT), 0, st, static_cast<const double*>(data_dev), nv, cb_in_dev,
                           e, w.idx);
    else
        hipLaunchKernelGGL(k_cb_assign<float>, grid, dim3(AT), 0, st, static_cast<const float*>(data_dev), nv, cb_in_dev,
                           e, w.idx);
    hipLaunchKernelGGL(k_cb_hist, dim3(nb), dim3(256), 0, st, w.idx, nv, e, w.hist);
    hipLaunchKernelGGL(k_cb_offsets, dim3(1), dim3(1024), 0, st, w.hist, nb, e, w.base, count_dev);
    if (data_f64) {
        hipLaunchKernelGGL(k_cb_scatter<double>, dim3(nb), dim3(64), 0, st, static_cast<const double*>(data_dev), w.idx,
                           nv, e, w.hist, static_cast<double*>(w.sorted));
        hipLaunchKernelGGL(k_cb_sum<double>, dim3(e), dim3(256), 0, st, static_cast<const double*>(w.sorted),
                           w.base, e, cb_out_dev);
    } else {
        hipLaunchKernelGGL(k_cb_scatter<float>, dim3(nb), dim3(64), 0, st, static_cast<const float*>(data_dev), w.idx, nv,
                           e, w.hist, static_cast<float*>(w.sorted));
        hipLaunchKernelGGL(k_cb_sum<float>, dim3(e), dim3(256), 0, st, static_cast<const float*>(w.sorted),
                           w.base, e, cb_out_dev);
    }
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}

extern "C" int fpc_cb_mean0(const void* data_dev, int data_f64, int nv, int nd, double* out_dev, fpc_stream s) {
    FPC_REQUIRE(data_dev && out_dev, "fpc_cb_mean0: null argument");
    if (int rc = check_shape("fpc_cb_mean0", nv, nd, 1)) return rc;
    hipStream_t st = static_cast<hipStream_t>(s);
    if (data_f64)
        hipLaunchKernelGGL(k_cb_mean0<double>, dim3(1), dim3(64), 0, st, static_cast<const double*>(data_dev), nv, out_dev);
    else
        hipLaunchKernelGGL(k_cb_mean0<float>, dim3(1), dim3(64), 0, st, static_cast<const float*>(data_dev), nv, out_dev);
    FPC_HIP(hipGetLastError());
    return FPC_OK;
}
