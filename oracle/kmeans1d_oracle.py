"""CPU restatement (numpy) of the scalar codebooks' k-means -- TEST INFRASTRUCTURE, not the product: only tests/ may import it.

What is restated: scikit-learn's `KMeans(n_clusters=k, random_state=0).fit(v[:, None])` for ONE feature, the call the reference
keeps (commented out) at /root/reference/src/train_cb.py:219-226.  scikit-learn is a third-party dependency that is not part of
the reference tree (installed here: 1.7.2); its published algorithm (sklearn/cluster/_kmeans.py: KMeans.fit, _kmeans_plusplus,
_kmeans_single_lloyd; _k_means_lloyd.pyx: _update_chunk_dense; _k_means_common.pyx: _relocate_empty_clusters_dense,
_average_centers, _center_shift, _inertia_dense; metrics/pairwise.py: _euclidean_distances) is followed operation by operation
where an operation decides an outcome (seed distances, the E-step's argmin, centre = sum * (1 / count), the stopping rules).
The long float64 sums have no specified order in sklearn (BLAS / OpenMP reductions); here they have the ONE fixed shape
csrc/kmeans1d.hip uses -- per block of 2048 points thread t adds points t, t + 256, .., a halving tree over 256 threads, block sums
one after the other; cluster sums per block in index order, then the blocks in order --
so the HIP kernels are compared with this file bit for bit, and this file is pinned to sklearn itself in the CPU suite
(tests/test_host_cpu.py: the same seeds index for index, centres to 1e-9)."""
import numpy as np

KT, PPT = 256, 8
CH = KT * PPT


def draws(n, k, n_init, seed=0):
    """the random draws of sklearn's seeding, in its order, from RandomState(seed): they do not depend on the data
    (KMeans.fit: one RandomState for all runs; _kmeans_plusplus: choice(n, p = weights / sum) then uniform(size = trials) per
    further seed).  Returns first_ids (n_init,), uniforms (n_init, k - 1, trials), trials"""
    rs = np.random.RandomState(seed)
    trials = 2 + int(np.log(k))
    p = np.ones(n) / np.ones(n).sum()
    first = np.empty(n_init, dtype=np.int64)
    u = np.empty((n_init, max(k - 1, 0), trials))
    for r in range(n_init):
        first[r] = rs.choice(n, p=p)
        for c in range(k - 1):
            u[r, c] = rs.uniform(size=trials)
    return first, u, trials


def _block_sums(v):
    """per block of 2048 points: thread t adds the points t, t + 256, .. in index order, then the halving tree over 256 threads"""
    n = len(v)
    nblk = (n + CH - 1) // CH
    a = np.zeros(nblk * CH)
    a[:n] = v
    a = a.reshape(nblk, PPT, KT)
    acc = np.zeros((nblk, KT))
    for j in range(PPT):
        acc = acc + a[:, j, :]
    h = KT // 2
    while h >= 1:
        acc[:, :h] = acc[:, :h] + acc[:, h:2 * h]
        h //= 2
    return acc[:, 0].copy()


def _seq(v):
    """v[0] + v[1] + ... one after the other"""
    return float(np.cumsum(v)[-1])


def _seed_dist(c, cc, x, xx):
    d = ((-2.0 * (c * x)) + cc) + xx
    return np.where(d > 0.0, d, 0.0)


def kmeans_plusplus(x, xx, k, first_id, u):
    n = len(x)
    seeds = np.empty(k, dtype=np.int64)
    seeds[0] = first_id
    closest = _seed_dist(x[first_id], xx[first_id], x, xx)
    part = _block_sums(closest)
    nblk = len(part)
    for c in range(1, k):
        prefix = np.cumsum(part)
        total = prefix[-1]
        cand = []
        for t in range(u.shape[1]):
            rv = u[c - 1, t] * total
            lo = int(np.searchsorted(prefix, rv, side="left"))
            pick = n - 1
            if lo < nblk:
                # inside the block: thread j starts from Q[j] = the exclusive prefix + the sums of the threads before it (a
                # thread's sum = its 8 points in index order; Q one thread after the other); the candidate is the first point
                # whose running value (Q[j] + its thread's points up to it) reaches rv, the point behind the block if none
                base = prefix[lo - 1] if lo > 0 else 0.0
                i0, i1 = lo * CH, min((lo + 1) * CH, n)
                blk = np.zeros(CH)
                blk[:i1 - i0] = closest[i0:i1]
                blk = blk.reshape(KT, PPT)
                S = np.zeros(KT)
                for j in range(PPT):
                    S = S + blk[:, j]
                Q = np.cumsum(np.concatenate([[base], S]))[:KT]
                run = np.empty((KT, PPT))
                acc = Q.copy()
                for j in range(PPT):
                    acc = acc + blk[:, j]
                    run[:, j] = acc
                ok = (run >= rv).reshape(-1)
                ok[i1 - i0:] = False
                hit = np.nonzero(ok)[0]
                pick = min(i0 + int(hit[0]) if len(hit) else i1, n - 1)
            cand.append(pick)
        pots, mins = [], []
        for p in cand:
            d = _seed_dist(x[p], xx[p], x, xx)
            m = np.where(closest < d, closest, d)
            mins.append(m)
            pots.append(_seq(_block_sums(m)))
        best = int(np.argmin(np.array(pots)))
        seeds[c] = cand[best]
        closest = mins[best]
        part = _block_sums(closest)
    return seeds


def _assign(x, centers):
    c2 = centers * centers
    d = c2[None, :] + (-2.0 * (x[:, None] * centers[None, :]))
    return np.argmin(d, axis=1).astype(np.int32)


def _sums(x, labels, k):
    """cluster sums and counts: per block of 2048 points the members' values one after the other in index order, then the blocks'
    partial sums one after the other"""
    n = len(x)
    tot, cnt = np.zeros(k), np.zeros(k)
    for i0 in range(0, n, CH):
        ps, pc = np.zeros(k), np.zeros(k)
        np.add.at(ps, labels[i0:i0 + CH], x[i0:i0 + CH])  # unbuffered, in order: ps[l] = ps[l] + x
        np.add.at(pc, labels[i0:i0 + CH], 1.0)
        tot = tot + ps
        cnt = cnt + pc
    return tot, cnt


def _assign_c(x, centers):
    """_assign through oracle/fpc_oracle.c::orc_km_assign: the same operations per (point, centre), the first minimum"""
    import ctypes as C
    from oracle import oracle as O
    labels = np.empty(len(x), dtype=np.int32)
    c = np.ascontiguousarray(centers, dtype=np.float64)
    O.lib().orc_km_assign(x.ctypes.data_as(C.c_void_p), C.c_longlong(len(x)), c.ctypes.data_as(C.c_void_p), C.c_int(len(c)),
                          labels.ctypes.data_as(C.c_void_p))
    return labels


def _sums_c(x, labels, k):
    """_sums through oracle/fpc_oracle.c::orc_km_sums: the same adds in the same order"""
    import ctypes as C
    from oracle import oracle as O
    tot, cnt, ps, pc = np.zeros(k), np.zeros(k), np.zeros(k), np.zeros(k)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    O.lib().orc_km_sums(x.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), C.c_longlong(len(x)), C.c_int(k), C.c_int(CH),
                        tot.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p),
                        pc.ctypes.data_as(C.c_void_p))
    return tot, cnt


def _relocate(x, labels, centers_old, s, cnt):
    empty = np.nonzero(cnt == 0.0)[0]
    if len(empty) == 0:
        return
    d = x - centers_old[labels]
    dist = d * d
    if dist.max() == 0:
        return
    # the farthest first, the lower index on equal distances (sklearn takes them from an argpartition: order unspecified)
    order = np.lexsort((np.arange(len(x)), -dist))[:len(empty)]
    for e, p in zip(empty, order):
        old = labels[p]
        s[old] = s[old] - x[p]
        s[e] = x[p]
        cnt[e] = 1.0
        cnt[old] = cnt[old] - 1.0


def _average(s, cnt):
    """_average_centers: sum * (1 / count) in place, in index order; a cluster that is still empty (more clusters than distinct
    values) is put "at the location of the biggest cluster" = entry argmax(count) of the array as it stands at that moment --
    the averaged centre if that cluster has been passed, its raw sum if not"""
    new = s.copy()
    am = int(np.argmax(cnt))
    for j in range(len(s)):
        if cnt[j] > 0.0:
            new[j] = new[j] * (1.0 / cnt[j])
        else:
            new[j] = new[am]
    return new


def _same_clustering(l1, l2, k):
    mapping = np.full(k, -1, dtype=np.int64)
    for a, b in zip(l1, l2):
        if mapping[a] == -1:
            mapping[a] = b
        elif mapping[a] != b:
            return False
    return True


def kmeans1d(x, k, first_ids, uniforms, tol, max_iter=300, fast=False):
    """x: centred float64 values.  Returns (centers (k,), inertia, n_iter, seeds (n_init, k)) of the winning run.
    fast: the E-step's argmin and the M-step's ordered sums through their C twins (orc_km_assign / orc_km_sums: the same
    operations in the same order, held equal to the numpy forms by tests/test_host_cpu.py) -- what makes the production
    sizes (k = 256, n = 400 000) run in seconds per iteration instead of a minute"""
    assign, sums = (_assign_c, _sums_c) if fast else (_assign, _sums)
    x = np.ascontiguousarray(x, dtype=np.float64)
    xx = x * x
    best = None
    all_seeds = []
    for r in range(len(first_ids)):
        seeds = kmeans_plusplus(x, xx, k, int(first_ids[r]), uniforms[r])
        all_seeds.append(seeds)
        centers = x[seeds].copy()
        labels_old = np.full(len(x), -1, dtype=np.int32)
        strict = False
        for it in range(max_iter):
            labels = assign(x, centers)
            s, cnt = sums(x, labels, k)
            _relocate(x, labels, centers, s, cnt)
            new = _average(s, cnt)
            d = centers - new
            sh = np.sqrt(d * d)
            shift = _seq(sh * sh)
            centers = new
            if np.array_equal(labels, labels_old):
                strict = True
                break
            if shift <= tol:
                break
            labels_old = labels
        if not strict:
            labels = assign(x, centers)
        d = x - centers[labels]
        inertia = _seq(_block_sums(d * d))
        if best is None or (inertia < best[1] and not _same_clustering(labels, best[3], k)):
            best = (centers, inertia, it + 1, labels)
    return best[0], best[1], best[2], np.array(all_seeds)


def fit(values, k, n_init=10, seed=0, max_iter=300, fast=False):
    """sklearn.cluster.KMeans(n_clusters=k, random_state=seed, n_init=n_init).fit(values[:, None]).cluster_centers_ restated:
    the tolerance from the data as given (KMeans._tol, before the mean is subtracted), the mean subtracted, added back at the end"""
    v = np.asarray(values, dtype=np.float64).reshape(-1, 1).copy()
    tol = float(np.mean(np.var(v, axis=0)) * 1e-4)
    mean = v.mean(axis=0)
    v -= mean
    first, u, _ = draws(len(v), k, n_init, seed)
    centers, inertia, n_iter, seeds = kmeans1d(v[:, 0], k, first, u, tol, max_iter, fast)
    return (centers[:, None] + mean), inertia, n_iter, seeds
