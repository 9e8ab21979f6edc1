// kmeans1d.hip -- the scalar codebooks' k-means (one feature), hand-written HIP for gfx950.
//
// Reference interface replaced (path under /root/reference/src):
//   train_cb.py:219-226 (commented out there):
//       KMeans(n_clusters=cfg['scl_clusters'], random_state=0).fit(np.array(scl_res).flatten()[:, None]).cluster_centers_
// i.e. scikit-learn's KMeans with its defaults: k-means++ seeding (sklearn/cluster/_kmeans.py, _kmeans_plusplus: the first
// seed uniform, every further seed the best of 2 + int(ln k) candidates drawn with probability proportional to the squared
// distance to the nearest seed so far), Lloyd iterations (_kmeans_single_lloyd: stop when the labels repeat or when the summed
// squared centre shift is <= tol = 1e-4 var(x); the E-step once more if it stopped on tol), n_init runs, the run of lowest
// inertia wins unless it is the same clustering.  scikit-learn is not part of the reference tree; what is restated here is
// its published algorithm, operation by operation where the operation decides an outcome:
//   distance of a seed c to x (sklearn.metrics.pairwise._euclidean_distances with precomputed norms):
//       max(((-2 (c x)) + fl(c c)) + fl(x x), 0)
//   E-step (lloyd_iter_chunked_dense: gemm(alpha = -2, beta = 1) onto the centres' squared norms, first minimum):
//       argmin_j  fl(c_j c_j) + (-2) (x c_j)
//   M-step: centre = sum * (1 / count) (_average_centers); shift_j = sqrt((old - new)^2), tolerance on sum_j shift_j^2.
// The random draws of the seeding do not depend on the data: the host draws them from numpy's RandomState in sklearn's order
// (choice for the first seed, uniform(size = trials) per further seed) and hands them over, so a run picks the seeds sklearn
// picks.  What differs from sklearn is the ASSOCIATION of the long float64 sums (potentials, cluster sums, inertia): sklearn's
// are BLAS / OpenMP reductions whose order is not specified (and not reproducible across thread counts); here every sum has
// one fixed shape -- per block of 2 048 points thread t adds the points t, t + 256, .. in index order, then a halving tree over
// the 256 threads, the block sums one after the other; cluster sums: per block the members in index order, then the blocks in order -- restated in oracle/kmeans1d_oracle.py (fit / kmeans1d), to which this file is bit-identical; the oracle is
// pinned to sklearn itself (same seeds, centres to 1e-9) in the CPU suite.
#include "fpc_common.h"

namespace {

constexpr int KT = 256;        // threads per block
constexpr int PPT = 8;         // consecutive points per thread
constexpr int CH = KT * PPT;   // points per block
constexpr int MAXK = 2048;     // centres (two LDS arrays of doubles in the E-step: 32 KB)
constexpr int MAXT = 16;       // candidates per seed (2 + int(ln k): 9 at k = 2048)

// _euclidean_distances(c, x, squared=True) with both squared norms precomputed
__device__ __forceinline__ double seed_dist(double c, double cc, double x, double xx) {
    const double t = c * x;
    const double d = ((-2.0 * t) + cc) + xx;
    return d > 0.0 ? d : 0.0;
}

// the block's sum of one value per thread: halving tree (thread t += thread t + h, h = 128 .. 1); result in every thread
__device__ __forceinline__ double block_tree(double v, double* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int h = KT / 2; h >= 1; h >>= 1) {
        if (t < h) sh[t] = sh[t] + sh[t + h];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// xx[i] = fl(x[i] x[i]) (row_norms(X, squared=True))
__global__ __launch_bounds__(KT) void k_km_norms(const double* __restrict__ x, long long n, double* __restrict__ xx) {
    const long long i = (long long)blockIdx.x * KT + threadIdx.x;
    if (i < n) xx[i] = x[i] * x[i];
}

// closest[i] = (first ? dist : min(closest[i], dist)) to the seed x[*seed]; part[b] = the block's sum of the new values
__global__ __launch_bounds__(KT) void k_kpp_apply(const double* __restrict__ x, const double* __restrict__ xx, long long n,
                                                  const int* __restrict__ seed, int first, double* __restrict__ closest,
                                                  double* __restrict__ part) {
    __shared__ double sh[KT];
    const int s = *seed;
    const double c = x[s], cc = xx[s];
    const long long i0 = (long long)blockIdx.x * CH + threadIdx.x;  // thread t: points t, t + 256, .. of the block (coalesced)
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const long long i = i0 + (long long)j * KT;
        if (i < n) {
            double d = seed_dist(c, cc, x[i], xx[i]);
            if (!first) {
                const double o = closest[i];
                d = o < d ? o : d;  // np.minimum
            }
            closest[i] = d;
            acc = acc + d;
        }
    }
    const double tot = block_tree(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// the candidates of one seed: prefix[b] = part[0] + .. + part[b] one after the other, pot = prefix[nblk - 1];
// candidate t = np.searchsorted(cumsum(closest), u[t] * pot) restated on three levels: the first BLOCK whose inclusive prefix
// reaches the value; inside it thread j's start Q[j] = the block's exclusive prefix + the sums S[0] .. S[j-1] of the threads
// before it (S = a thread's 8 points added in index order; Q accumulated one thread after the other); the candidate is the
// first point, in index order, whose running value (Q[j] + its thread's points up to it, one after the other) reaches the
// value -- the point behind the block if none does (the block's tree sum and these running sums differ in the last bits) --
// clipped to n - 1.  (A plain one-after-the-other running sum over the block's 2 048 points -- the first form of this kernel --
// is one dependent global load per point: 0.3 ms per seed, 95 % of the seeding.)
__global__ __launch_bounds__(KT) void k_kpp_pick(const double* __restrict__ part, int nblk, const double* __restrict__ closest,
                                                 long long n, const double* __restrict__ u, int trials, double* __restrict__ prefix,
                                                 double* __restrict__ pot, int* __restrict__ cand) {
    __shared__ double total;
    __shared__ double S[KT], Q[KT];
    __shared__ int hit[KT], slo[MAXT];
    __shared__ double chunk[512];
    const int t = threadIdx.x;
    {  // the prefix, 512 block sums at a time through LDS (thread 0 adds them one after the other)
        double acc = 0.0;
        for (int b0 = 0; b0 < nblk; b0 += 512) {
            const int m = nblk - b0 < 512 ? nblk - b0 : 512;
            for (int b = t; b < m; b += KT) chunk[b] = part[b0 + b];
            __syncthreads();
            if (t == 0) {
                for (int b = 0; b < m; ++b) {
                    acc = acc + chunk[b];
                    chunk[b] = acc;
                }
            }
            __syncthreads();
            for (int b = t; b < m; b += KT) prefix[b0 + b] = chunk[b];
            __syncthreads();
        }
        if (t == 0) {
            total = acc;
            *pot = acc;
        }
    }
    __syncthreads();
    if (t < trials) {  // first b with prefix[b] >= rv, every trial's search side by side
        const double rv = u[t] * total;
        int lo = 0, hi = nblk;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] >= rv)
                hi = mid;
            else
                lo = mid + 1;
        }
        slo[t] = lo;
    }
    __syncthreads();
    for (int tr = 0; tr < trials; ++tr) {
        const double rv = u[tr] * total;
        const int lo = slo[tr];
        if (lo >= nblk) {  // (block-uniform)
            if (t == 0) cand[tr] = (int)(n - 1);
            continue;
        }
        const long long i0 = (long long)lo * CH + (long long)t * PPT;
        double v[PPT];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            v[j] = i0 + j < n ? closest[i0 + j] : 0.0;
            s = s + v[j];
        }
        S[t] = s;
        __syncthreads();
        if (t == 0) {
            double acc = lo > 0 ? prefix[lo - 1] : 0.0;
            for (int j = 0; j < KT; ++j) {
                Q[j] = acc;
                acc = acc + S[j];
            }
        }
        __syncthreads();
        int h = CH;  // offset of the thread's first point whose running value reaches rv
        double acc = Q[t];
#pragma unroll
        for (int j = PPT - 1; j >= 0; --j) {
            double run = acc;
#pragma unroll
            for (int q = 0; q <= j; ++q) run = run + v[q];  // (Q + v0 + .. + vj, one after the other)
            if (i0 + j < n && run >= rv) h = t * PPT + j;
        }
        hit[t] = h;
        __syncthreads();
        for (int w = KT / 2; w >= 1; w >>= 1) {
            if (t < w) hit[t] = hit[t + w] < hit[t] ? hit[t + w] : hit[t];
            __syncthreads();
        }
        if (t == 0) {
            long long pick = (long long)lo * CH + hit[0];  // (no hit: CH = the point behind the block)
            const long long end = (long long)(lo + 1) * CH < n ? (long long)(lo + 1) * CH : n;
            if (pick > end) pick = end;
            if (pick > n - 1) pick = n - 1;
            cand[tr] = (int)pick;
        }
        __syncthreads();
    }
}

// the potential every candidate would leave: partT[t][b] = the block's sum of min(closest, dist to candidate t)
__global__ __launch_bounds__(KT) void k_kpp_eval(const double* __restrict__ x, const double* __restrict__ xx, long long n,
                                                 const double* __restrict__ closest, const int* __restrict__ cand, int trials,
                                                 int nblk, double* __restrict__ partT) {
    __shared__ double sh[KT];
    __shared__ double sc[MAXT], scc[MAXT];
    if (threadIdx.x < trials) {
        sc[threadIdx.x] = x[cand[threadIdx.x]];
        scc[threadIdx.x] = xx[cand[threadIdx.x]];
    }
    __syncthreads();
    const long long i0 = (long long)blockIdx.x * CH + threadIdx.x;
    double px[PPT], pxx[PPT], pc[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const long long i = i0 + (long long)j * KT;
        const bool in = i < n;
        px[j] = in ? x[i] : 0.0;
        pxx[j] = in ? xx[i] : 0.0;
        pc[j] = in ? closest[i] : 0.0;
    }
    for (int t = 0; t < trials; ++t) {
        const double c = sc[t], cc = scc[t];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (i0 + (long long)j * KT < n) {
                const double d = seed_dist(c, cc, px[j], pxx[j]);
                acc = acc + (pc[j] < d ? pc[j] : d);
            }
        const double tot = block_tree(acc, sh);
        if (threadIdx.x == 0) partT[(size_t)t * nblk + blockIdx.x] = tot;
    }
}

// the best candidate (np.argmin of the potentials: block sums added one after the other) becomes seed number `c`
// (the block sums come into LDS 512 at a time, all trials side by side; thread t adds trial t's in order)
constexpr int BCH = 512;
__global__ __launch_bounds__(KT) void k_kpp_best(const double* __restrict__ partT, int nblk, int trials, const int* __restrict__ cand,
                                                 const double* __restrict__ x, int c, int* __restrict__ seeds, double* __restrict__ centers) {
    __shared__ double buf[MAXT][BCH];
    __shared__ double pots[MAXT];
    const int t = threadIdx.x;
    double acc = 0.0;
    for (int b0 = 0; b0 < nblk; b0 += BCH) {
        const int m = nblk - b0 < BCH ? nblk - b0 : BCH;
        for (int q = t; q < trials * BCH; q += KT) {
            const int tr = q / BCH, b = q - tr * BCH;
            if (b < m) buf[tr][b] = partT[(size_t)tr * nblk + b0 + b];
        }
        __syncthreads();
        if (t < trials)
            for (int b = 0; b < m; ++b) acc = acc + buf[t][b];
        __syncthreads();
    }
    if (t < trials) pots[t] = acc;
    __syncthreads();
    if (t == 0) {
        int best = 0;
        for (int j = 1; j < trials; ++j)
            if (pots[j] < pots[best]) best = j;
        seeds[c] = cand[best];
        centers[c] = x[cand[best]];
    }
}

// seed 0
__global__ void k_kpp_first(const double* __restrict__ x, int id, int* __restrict__ seeds, double* __restrict__ centers) {
    seeds[0] = id;
    centers[0] = x[id];
}

// E-step: label = first minimum of fl(c c) + (-2) (x c); flag[0] |= a label differs from the previous iteration's
__global__ __launch_bounds__(KT) void k_km_assign(const double* __restrict__ x, long long n, const double* __restrict__ centers, int k,
                                                  const int* __restrict__ old_labels, int* __restrict__ labels, int* __restrict__ flag) {
    __shared__ double sc[MAXK], sc2[MAXK];
    for (int j = threadIdx.x; j < k; j += KT) {
        const double c = centers[j];
        sc[j] = c;
        sc2[j] = c * c;
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * KT + threadIdx.x;
    if (i >= n) return;
    const double xv = x[i];
    double best = sc2[0] + (-2.0 * (xv * sc[0]));
    int bj = 0;
    for (int j = 1; j < k; ++j) {
        const double d = sc2[j] + (-2.0 * (xv * sc[j]));
        if (d < best) {
            best = d;
            bj = j;
        }
    }
    labels[i] = bj;
    if (old_labels[i] != bj) *flag = 1;
}

// M-step in two levels, every add in index order: (1) per chunk of 2 048 points and cluster, the members' values one after the
// other (thread j walks the chunk's labels in LDS -- every lane reads the same element: a broadcast -- for the clusters j, j + 256,
// ..); (2) per cluster, the chunks' partial sums one after the other.  (One workgroup per cluster scanning ALL labels -- the first
// form -- read n k labels from the L2 per iteration: 1.85 ms at n = 2 M, k = 256.)
__global__ __launch_bounds__(KT) void k_km_psums(const double* __restrict__ x, const int* __restrict__ labels, long long n, int k,
                                                 double* __restrict__ psum, double* __restrict__ pcnt) {
    __shared__ double xv[CH];
    __shared__ int lab[CH];
    const long long i0 = (long long)blockIdx.x * CH;
    const int m = (int)(n - i0 < CH ? n - i0 : CH);
    for (int i = threadIdx.x; i < m; i += KT) {
        xv[i] = x[i0 + i];
        lab[i] = labels[i0 + i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += KT) {
        double s = 0.0, c = 0.0;
        for (int i = 0; i < m; ++i)
            if (lab[i] == j) {
                s = s + xv[i];
                c = c + 1.0;
            }
        psum[(size_t)blockIdx.x * k + j] = s;
        pcnt[(size_t)blockIdx.x * k + j] = c;
    }
}
__global__ __launch_bounds__(KT) void k_km_sums(const double* __restrict__ psum, const double* __restrict__ pcnt, int nblk, int k,
                                                double* __restrict__ sum, double* __restrict__ cnt) {
    const int j = blockIdx.x * KT + threadIdx.x;
    if (j >= k) return;
    double s = 0.0, c = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s = s + psum[(size_t)b * k + j];
        c = c + pcnt[(size_t)b * k + j];
    }
    sum[j] = s;
    cnt[j] = c;
}

// _relocate_empty_clusters_dense: the m empty clusters, in index order, take the m points farthest from their centres, farthest
// first (sklearn takes them from an argpartition, whose order among the m is not specified; lower index first on equal
// distances); the labels stay as they are; nothing is relocated when every point sits on its centre.  One block; nothing to do
// (the usual case) when no cluster is empty.
__global__ __launch_bounds__(1024) void k_km_relocate(const double* __restrict__ x, const int* __restrict__ labels, long long n,
                                                      const double* __restrict__ centers_old, int k, double* __restrict__ sum,
                                                      double* __restrict__ cnt, int* __restrict__ taken) {
    __shared__ int n_empty;
    __shared__ int empty[MAXK];  // np.where(weight_in_clusters == 0)[0], taken before any relocation
    __shared__ double bv[1024];
    __shared__ long long bi[1024];
    if (threadIdx.x == 0) {
        int m = 0;
        for (int j = 0; j < k; ++j)
            if (cnt[j] == 0.0) empty[m++] = j;
        n_empty = m;
    }
    __syncthreads();
    const int m = n_empty;
    if (m == 0) return;
    for (int r = 0; r < m; ++r) {
        double v = -1.0;
        long long vi = n;
        for (long long i = threadIdx.x; i < n; i += 1024) {
            bool used = false;
            for (int q = 0; q < r; ++q) used |= taken[q] == (int)i;
            if (used) continue;
            const double d = x[i] - centers_old[labels[i]];
            const double dd = d * d;
            if (dd > v) {
                v = dd;
                vi = i;
            }
        }
        bv[threadIdx.x] = v;
        bi[threadIdx.x] = vi;
        __syncthreads();
        for (int h = 512; h >= 1; h >>= 1) {
            if (threadIdx.x < h) {
                const double ov = bv[threadIdx.x + h];
                const long long oi = bi[threadIdx.x + h];
                if (ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) {
                    bv[threadIdx.x] = ov;
                    bi[threadIdx.x] = oi;
                }
            }
            __syncthreads();
        }
        // "Happens when there are more clusters than non-duplicate samples. Relocating is pointless in this case." (sklearn)
        if (r == 0 && !(bv[0] > 0.0)) return;  // (np.max(distances) == 0; block-uniform: every thread reads the same bv[0])
        if (threadIdx.x == 0 && bi[0] < n) {
            const long long p = bi[0];
            const int e = empty[r], old = labels[p];
            taken[r] = (int)p;
            sum[old] = sum[old] - x[p];
            sum[e] = x[p];
            cnt[e] = 1.0;
            cnt[old] = cnt[old] - 1.0;
        }
        __syncthreads();
    }
}

// _average_centers and _center_shift: new centre = sum * (1 / count); a cluster that is STILL empty (more clusters than
// distinct values) goes where sklearn puts it: "at the location of the biggest cluster" = entry argmax(count) of the array it
// is averaging in place, in index order -- the averaged centre if that cluster comes before it, its raw SUM if it comes after.
// scal[0] = sum_j (sqrt((old - new)^2))^2, one after the other
__global__ __launch_bounds__(KT) void k_km_average(const double* __restrict__ sum, const double* __restrict__ cnt,
                                                   const double* __restrict__ centers_old, int k, double* __restrict__ centers_new,
                                                   double* __restrict__ shift2, double* __restrict__ scal) {
    __shared__ int am;
    if (threadIdx.x == 0) {
        int a = 0;
        for (int j = 1; j < k; ++j)
            if (cnt[j] > cnt[a]) a = j;  // np.argmax: the first maximum
        am = a;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += KT) {
        double c;
        if (cnt[j] > 0.0) {
            const double alpha = 1.0 / cnt[j];
            c = sum[j] * alpha;
        } else if (j > am && cnt[am] > 0.0) {
            const double alpha = 1.0 / cnt[am];
            c = sum[am] * alpha;
        } else {
            c = sum[am];
        }
        centers_new[j] = c;
        const double d = centers_old[j] - c;
        const double s = sqrt(d * d);
        shift2[j] = s * s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc = acc + shift2[j];
        scal[0] = acc;
    }
}

// _inertia_dense: part[b] = the block's sum of (x - centre of its label)^2
__global__ __launch_bounds__(KT) void k_km_inertia(const double* __restrict__ x, const int* __restrict__ labels, long long n,
                                                   const double* __restrict__ centers, double* __restrict__ part) {
    __shared__ double sh[KT];
    const long long i0 = (long long)blockIdx.x * CH + threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const long long i = i0 + (long long)j * KT;
        if (i < n) {
            const double d = x[i] - centers[labels[i]];
            acc = acc + d * d;
        }
    }
    const double tot = block_tree(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void k_km_total(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc = acc + part[b];
    *out = acc;
}

// _is_same_clustering: is labels -> best a function?  (every writer of map[a] writes best[i]; if two disagree, one of them sees
// the other's value in the check)
__global__ __launch_bounds__(KT) void k_km_map(const int* __restrict__ labels, const int* __restrict__ best, long long n, int* __restrict__ map) {
    const long long i = (long long)blockIdx.x * KT + threadIdx.x;
    if (i < n) map[labels[i]] = best[i];
}
__global__ __launch_bounds__(KT) void k_km_mapcheck(const int* __restrict__ labels, const int* __restrict__ best, long long n,
                                                    const int* __restrict__ map, int* __restrict__ differs) {
    const long long i = (long long)blockIdx.x * KT + threadIdx.x;
    if (i < n && map[labels[i]] != best[i]) *differs = 1;
}

}  // namespace

extern "C" int fpc_kmeans1d(const double* x_dev, long long n, int k, int n_init, int trials, const long long* first_ids,
                            const double* uniforms, double tol, int max_iter, double* centers_host, double* inertia_host,
                            int* n_iter_host, int* seeds_host, fpc_stream s) {
    using namespace fpc;
    FPC_REQUIRE(x_dev && first_ids && centers_host, "fpc_kmeans1d: null argument");
    FPC_REQUIRE(n >= 1 && n < (1ll << 28), "fpc_kmeans1d: n = %lld outside 1 .. 2^28 - 1", n);
    FPC_REQUIRE(k >= 1 && k <= MAXK && k <= n, "fpc_kmeans1d: k = %d outside 1 .. min(%d, n)", k, MAXK);
    FPC_REQUIRE(n_init >= 1 && max_iter >= 1, "fpc_kmeans1d: n_init = %d, max_iter = %d", n_init, max_iter);
    FPC_REQUIRE(trials >= 1 && trials <= MAXT, "fpc_kmeans1d: trials = %d outside 1 .. %d", trials, MAXT);
    FPC_REQUIRE(k == 1 || uniforms, "fpc_kmeans1d: the seeding's uniform draws are missing");
    for (int r = 0; r < n_init; ++r) FPC_REQUIRE(first_ids[r] >= 0 && first_ids[r] < n, "fpc_kmeans1d: first_ids[%d] = %lld", r, first_ids[r]);
    if (!have_device()) {
        set_error("fpc_kmeans1d: no HIP device");
        return FPC_ERR_NO_DEVICE;
    }
    hipStream_t st = static_cast<hipStream_t>(s);
    const int nblk = (int)((n + CH - 1) / CH), nthr = (int)((n + KT - 1) / KT);
    const size_t nu = (size_t)n_init * (size_t)(k > 1 ? k - 1 : 0) * trials;
    DevBuf xx, closest, part, partT, prefix, u, cand, seeds, centers[2], best_centers, labels[2], best_labels, sum, cnt, shift2,
        scal, flags, map, taken, psum, pcnt;
    FPC_HIP(xx.alloc(sizeof(double) * n));
    FPC_HIP(closest.alloc(sizeof(double) * n));
    FPC_HIP(part.alloc(sizeof(double) * nblk));
    FPC_HIP(partT.alloc(sizeof(double) * (size_t)nblk * trials));
    FPC_HIP(prefix.alloc(sizeof(double) * nblk));
    FPC_HIP(u.alloc(sizeof(double) * (nu ? nu : 1)));
    FPC_HIP(cand.alloc(sizeof(int) * MAXT));
    FPC_HIP(seeds.alloc(sizeof(int) * k));
    for (int i = 0; i < 2; ++i) {
        FPC_HIP(centers[i].alloc(sizeof(double) * k));
        FPC_HIP(labels[i].alloc(sizeof(int) * n));
    }
    FPC_HIP(best_centers.alloc(sizeof(double) * k));
    FPC_HIP(best_labels.alloc(sizeof(int) * n));
    FPC_HIP(psum.alloc(sizeof(double) * (size_t)nblk * k));
    FPC_HIP(pcnt.alloc(sizeof(double) * (size_t)nblk * k));
    FPC_HIP(sum.alloc(sizeof(double) * k));
    FPC_HIP(cnt.alloc(sizeof(double) * k));
    FPC_HIP(shift2.alloc(sizeof(double) * k));
    FPC_HIP(scal.alloc(sizeof(double) * 4));  // [0] shift total, [1] potential, [2] inertia
    FPC_HIP(flags.alloc(sizeof(int) * 4));    // [0] a label changed, [1] the clusterings differ
    FPC_HIP(map.alloc(sizeof(int) * k));
    FPC_HIP(taken.alloc(sizeof(int) * k));
    if (nu) FPC_HIP(hipMemcpyAsync(u.p, uniforms, sizeof(double) * nu, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_km_norms, dim3(nthr), dim3(KT), 0, st, x_dev, n, xx.as<double>());

    double best_inertia = 0.0;
    int best_iter = 0;
    for (int r = 0; r < n_init; ++r) {
        // ---- k-means++ ----
        hipLaunchKernelGGL(k_kpp_first, dim3(1), dim3(1), 0, st, x_dev, (int)first_ids[r], seeds.as<int>(), centers[0].as<double>());
        hipLaunchKernelGGL(k_kpp_apply, dim3(nblk), dim3(KT), 0, st, x_dev, xx.as<double>(), n, seeds.as<int>(), 1, closest.as<double>(),
                           part.as<double>());
        for (int c = 1; c < k; ++c) {
            const double* uc = u.as<double>() + ((size_t)r * (k - 1) + (c - 1)) * trials;
            hipLaunchKernelGGL(k_kpp_pick, dim3(1), dim3(KT), 0, st, part.as<double>(), nblk, closest.as<double>(), n, uc, trials,
                               prefix.as<double>(), scal.as<double>() + 1, cand.as<int>());
            hipLaunchKernelGGL(k_kpp_eval, dim3(nblk), dim3(KT), 0, st, x_dev, xx.as<double>(), n, closest.as<double>(), cand.as<int>(),
                               trials, nblk, partT.as<double>());
            hipLaunchKernelGGL(k_kpp_best, dim3(1), dim3(KT), 0, st, partT.as<double>(), nblk, trials, cand.as<int>(), x_dev, c,
                               seeds.as<int>(), centers[0].as<double>());
            hipLaunchKernelGGL(k_kpp_apply, dim3(nblk), dim3(KT), 0, st, x_dev, xx.as<double>(), n, seeds.as<int>() + c, 0,
                               closest.as<double>(), part.as<double>());
        }
        FPC_HIP(hipGetLastError());
        if (seeds_host) FPC_HIP(hipMemcpyAsync(seeds_host + (size_t)r * k, seeds.p, sizeof(int) * k, hipMemcpyDeviceToHost, st));
        // ---- Lloyd ----
        int cur = 0, lab = 0, last = 0, iters = 0;  // lab: the buffer the next E-step writes (the other one = labels_old)
        bool strict = false;
        FPC_HIP(hipMemsetAsync(labels[1].p, 0xFF, sizeof(int) * n, st));  // labels_old = -1
        for (int it = 0; it < max_iter; ++it) {
            FPC_HIP(hipMemsetAsync(flags.p, 0, sizeof(int) * 4, st));
            hipLaunchKernelGGL(k_km_assign, dim3(nthr), dim3(KT), 0, st, x_dev, n, centers[cur].as<double>(), k, labels[lab ^ 1].as<int>(),
                               labels[lab].as<int>(), flags.as<int>());
            hipLaunchKernelGGL(k_km_psums, dim3(nblk), dim3(KT), 0, st, x_dev, labels[lab].as<int>(), n, k, psum.as<double>(),
                               pcnt.as<double>());
            hipLaunchKernelGGL(k_km_sums, dim3((k + KT - 1) / KT), dim3(KT), 0, st, psum.as<double>(), pcnt.as<double>(), nblk, k,
                               sum.as<double>(), cnt.as<double>());
            hipLaunchKernelGGL(k_km_relocate, dim3(1), dim3(1024), 0, st, x_dev, labels[lab].as<int>(), n, centers[cur].as<double>(), k,
                               sum.as<double>(), cnt.as<double>(), taken.as<int>());
            hipLaunchKernelGGL(k_km_average, dim3(1), dim3(KT), 0, st, sum.as<double>(), cnt.as<double>(), centers[cur].as<double>(), k,
                               centers[cur ^ 1].as<double>(), shift2.as<double>(), scal.as<double>());
            FPC_HIP(hipGetLastError());
            int changed = 0;
            double shift = 0.0;
            FPC_HIP(hipMemcpyAsync(&changed, flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
            FPC_HIP(hipMemcpyAsync(&shift, scal.p, sizeof(double), hipMemcpyDeviceToHost, st));
            FPC_HIP(hipStreamSynchronize(st));
            cur ^= 1;  // centers, centers_new = centers_new, centers
            iters = it + 1;
            last = lab;
            if (!changed) {  // np.array_equal(labels, labels_old)
                strict = true;
                break;
            }
            if (shift <= tol) break;
            lab ^= 1;  // labels_old[:] = labels
        }
        if (!strict) {  // the E-step once more, so that the labels match the final centres (labels_old is only compared against)
            hipLaunchKernelGGL(k_km_assign, dim3(nthr), dim3(KT), 0, st, x_dev, n, centers[cur].as<double>(), k, labels[last].as<int>(),
                               labels[last ^ 1].as<int>(), flags.as<int>());
            last ^= 1;
        }
        hipLaunchKernelGGL(k_km_inertia, dim3(nblk), dim3(KT), 0, st, x_dev, labels[last].as<int>(), n, centers[cur].as<double>(),
                           part.as<double>());
        hipLaunchKernelGGL(k_km_total, dim3(1), dim3(1), 0, st, part.as<double>(), nblk, scal.as<double>() + 2);
        FPC_HIP(hipGetLastError());
        double inertia = 0.0;
        FPC_HIP(hipMemcpyAsync(&inertia, scal.as<double>() + 2, sizeof(double), hipMemcpyDeviceToHost, st));
        FPC_HIP(hipStreamSynchronize(st));
        bool take = r == 0;
        if (!take && inertia < best_inertia) {
            FPC_HIP(hipMemsetAsync(flags.as<int>() + 1, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_km_map, dim3(nthr), dim3(KT), 0, st, labels[last].as<int>(), best_labels.as<int>(), n, map.as<int>());
            hipLaunchKernelGGL(k_km_mapcheck, dim3(nthr), dim3(KT), 0, st, labels[last].as<int>(), best_labels.as<int>(), n, map.as<int>(),
                               flags.as<int>() + 1);
            int differs = 0;
            FPC_HIP(hipMemcpyAsync(&differs, flags.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            FPC_HIP(hipStreamSynchronize(st));
            take = differs != 0;
        }
        if (take) {
            best_inertia = inertia;
            best_iter = iters;
            FPC_HIP(hipMemcpyAsync(best_labels.p, labels[last].p, sizeof(int) * n, hipMemcpyDeviceToDevice, st));
            FPC_HIP(hipMemcpyAsync(best_centers.p, centers[cur].p, sizeof(double) * k, hipMemcpyDeviceToDevice, st));
        }
        // (the next run seeds into centers[0]: k_kpp_first / k_kpp_best overwrite every entry)
    }
    FPC_HIP(hipMemcpyAsync(centers_host, best_centers.p, sizeof(double) * k, hipMemcpyDeviceToHost, st));
    FPC_HIP(hipStreamSynchronize(st));
    if (inertia_host) *inertia_host = best_inertia;
    if (n_iter_host) *n_iter_host = best_iter;
    return FPC_OK;
}
