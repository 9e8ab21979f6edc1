"""Codebook-training driver with the reference's `train_cb.py` surface (src/train_cb.py:48-226).

What the reference does per batch of utterances (`__main__`, :160-217):
    residuals of the closed-loop encoder WITHOUT quantisation        (:165-170; the shipped `mask_enc` call raises
                                                                      AttributeError, SURVEY App. C: the live equivalent
                                                                      is the commented `model.encoder(..., qtz=0)` :165-167
                                                                      = Wavernn.encoder's qtz=False branch, wavernn.py:244-252)
    scalar residuals != 0 collected for the scalar codebooks          (:174-175)
    rows of the last `code_dims` columns, all-zero rows dropped       (:179-186)
    first batch:  per stage  vq_train -> quantize -> r = qr - r       (:191-199)
    later batches: per stage 10 x update -> quantize -> r = qr - r    (:201-209)
    np.save('../codebooks/ceps_vq_codebook_{note}.npy', codebook)     (:217)
    scalar codebooks = KMeans centres (n, 1)                           (:219-226, commented in the reference)
Here the harvest stays on the device (encoder kernel + stable row compaction by a boolean mask), the stage loop
calls the GPU `cb_func` (same call surface as the reference's), and the files are written in the reference's
formats: (S, N, 17) float64 (object array of S stages when the sizes differ) and (n, 1) float64."""
import os

import numpy as np
import torch

from . import cb_func


def harvest(model, cfg, feat, l1=0.09, l2=0.28, train_bl=False, code_dims=17):
    """residual rows of one batch `feat` (B, L, 20): (rows (n, code_dims) float32 on the device, zero rows dropped
    in order; scalar residuals above / below the threshold, zeros dropped) -- train_cb.py:165-186"""
    c_in, r, r_qtz, r_bl, _, _, _ = model.encoder(cfg=cfg, feat=feat, mask=None, l1=l1, l2=l2, qtz=False)
    scl = r[:, :, 0].reshape(-1)
    scl_bl = r_bl[:, :, 0].reshape(-1)
    src = r_bl if train_bl else r
    rows = src[:, :, -code_dims:].reshape(-1, code_dims)
    keep = rows.abs().sum(1) != 0  # sum(abs(row)) != 0 (:186): zero iff every element is zero
    return rows[keep].contiguous(), scl[scl != 0], scl_bl[scl_bl != 0]


def train_stages(codebook, r, n_entries, first_batch, verbose=False):
    """the stage loop of one batch (train_cb.py:189-209) on residual rows `r`; returns the updated stage list and
    the final residual.  `r` is float32 rows for the first stage and the float64 `qr - r` afterwards."""
    codebook = list(codebook)
    for i in range(len(codebook)):
        if first_batch:
            codebook[i] = cb_func.vq_train(r, codebook[i], n_entries[i], verbose=verbose)
        else:
            for _ in range(10):
                codebook[i] = cb_func.update(r, codebook[i], n_entries[i], verbose=verbose)
        r_host = r.cpu().numpy() if isinstance(r, torch.Tensor) else np.asarray(r)
        qr = cb_func.quantize(codebook[i], r)
        r = qr - r_host  # float64 (:192,:206)
    return codebook, r


def save_codebook(path, codebook):
    """np.save(path, codebook) as train_cb.py:217 does for a list of stages: (S, N, 17) float64 when the stages
    have one size, otherwise an object array of S stages (what vq_func.py:141 loads with allow_pickle=True)"""
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    stages = [np.asarray(c, dtype=np.float64) for c in codebook]
    if len({s.shape for s in stages}) == 1:
        np.save(path, np.stack(stages))
    else:
        arr = np.empty(len(stages), dtype=object)
        for k, s in enumerate(stages):
            arr[k] = s
        np.save(path, arr, allow_pickle=True)


def kmeans_draws(n, k, n_init, seed=0):
    """the random draws of scikit-learn's k-means++ seeding in its order, from numpy's RandomState(seed): one RandomState for
    all runs (KMeans.fit), per run `choice(n, p = 1 / n)` for the first seed and `uniform(size = trials)` for every further
    one (_kmeans_plusplus), trials = 2 + int(ln k).  The draws do not depend on the data: handing them to fpc_kmeans1d makes a
    run pick the seeds sklearn's `KMeans(random_state=seed)` picks.  Returns (first_ids int64 (n_init,), uniforms float64
    (n_init, k - 1, trials), trials)"""
    rs = np.random.RandomState(seed)
    trials = 2 + int(np.log(k))
    w = np.ones(n)
    p = w / w.sum()
    # RandomState.choice(n, p=p) = searchsorted(normalised cumsum of p, one random_sample(), side="right"): the O(n) part is the
    # same for every run, so it is formed once (tests: equal to calling choice itself)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    first = np.empty(n_init, dtype=np.int64)
    u = np.empty((n_init, max(k - 1, 0), trials), dtype=np.float64)
    for r in range(n_init):
        first[r] = min(int(cdf.searchsorted(rs.random_sample(), side="right")), n - 1)
        if k > 1:
            u[r] = rs.uniform(size=(k - 1) * trials).reshape(k - 1, trials)  # (= k - 1 draws of `trials` each: one stream)
    return first, u, trials


def train_scalar_codebook(values, n_clusters, n_init=10, backend="gpu", max_iter=300, details=None):
    """(n_clusters, 1) float64 centres of the scalar residuals: the reference's commented
    `KMeans(n_clusters, random_state=0).fit(values[:, None]).cluster_centers_` (train_cb.py:219-226; n_init = 10 is the
    default of the scikit-learn releases that call was written for).
    backend "gpu": fpc_kmeans1d (csrc/kmeans1d.hip: seeding and Lloyd iterations on the device, the seeding's random draws
    from numpy's RandomState(0) in sklearn's order) -- the same seeds as sklearn, centres equal to rounding (tests: 1e-9), and
    reproducible run to run; fails loudly without the library or a GPU.  backend "sklearn": the reference's call itself, on
    the host.  `details`: a dict that receives inertia, n_iter and the seeds' indices (gpu backend)"""
    k = int(n_clusters)
    v = np.asarray(values, dtype=np.float64).reshape(-1, 1).copy()
    if backend == "sklearn":
        from sklearn.cluster import KMeans
        return KMeans(n_clusters=k, random_state=0, n_init=n_init, max_iter=max_iter).fit(v).cluster_centers_.astype(np.float64)
    if backend != "gpu":
        raise ValueError(f"backend {backend!r}: 'gpu' or 'sklearn'")
    import ctypes as C
    from . import _lib
    _lib.require_gpu()
    n = v.shape[0]
    if k > n:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")  # (sklearn's message)
    tol = float(np.mean(np.var(v, axis=0)) * 1e-4)  # KMeans._tol: from the data as given
    mean = v.mean(axis=0)                            # "subtract of mean of x for more accurate distance computations"
    v -= mean
    first, u, trials = kmeans_draws(n, k, n_init)
    x = torch.from_numpy(v[:, 0].copy()).cuda()
    centers = np.empty(k, dtype=np.float64)
    inertia, n_iter = C.c_double(0.0), C.c_int(0)
    seeds = np.empty((n_init, k), dtype=np.int32)
    _lib.check(_lib.lib().fpc_kmeans1d(C.c_void_p(x.data_ptr()), n, k, n_init, trials, first.ctypes.data_as(C.c_void_p),
                                       u.ctypes.data_as(C.c_void_p), tol, int(max_iter), centers.ctypes.data_as(C.c_void_p),
                                       C.byref(inertia), C.byref(n_iter), seeds.ctypes.data_as(C.c_void_p),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fpc_kmeans1d")
    if details is not None:
        details.update(inertia=inertia.value, n_iter=n_iter.value, seeds=seeds, tol=tol, centred=centers.copy())
    return centers[:, None] + mean


def train(model, cfg, batches, verbose=False):
    """The loop of train_cb.py:160-217 over `batches` (iterable of (B, L, 20) normalised feature tensors, the
    `nm_c[:, 2:-2, :-16]` of :156).  cfg keys as in the reference (:54-96): stages, n_entries, code_dims, note,
    train_bl, cb_path (continue from a file), scl_clusters / scl_clusters_bl (optional).  Returns the stage list and
    writes ../codebooks/ceps_vq_codebook_{note}.npy (+ the scalar codebooks when the cluster counts are given)."""
    stages, n_entries, code_dims = cfg["stages"], cfg["n_entries"], cfg.get("code_dims", 17)
    if cfg.get("cb_path"):
        loaded = np.load(cfg["cb_path"], allow_pickle=True)
        codebook = [np.asarray(loaded[i], dtype=np.float64) for i in range(stages)]
    else:
        codebook = [np.zeros((n_entries[i], code_dims)) for i in range(stages)]  # :127-130
    scl_res, scl_res_bl = [], []
    print('training:', '../codebooks/ceps_vq_codebook_{}.npy'.format(cfg['note']))
    for batch_idx, feat in enumerate(batches):
        rows, scl, scl_bl = harvest(model, cfg, feat, cfg.get("l1", 0.09), cfg.get("l2", 0.28),
                                    bool(cfg.get("train_bl")), code_dims)
        scl_res.append(scl.cpu().numpy())
        scl_res_bl.append(scl_bl.cpu().numpy())
        print('Finish residual calculating of epoch {}'.format(batch_idx))
        codebook, r = train_stages(codebook, rows, n_entries, batch_idx == 0 and not cfg.get("cb_path"), verbose)
        print('Epoch: {}, Err: {}'.format(batch_idx, float(np.sum(r * r))))
    out = '../codebooks/ceps_vq_codebook_{}.npy'.format(cfg['note'])
    save_codebook(out, codebook)
    if cfg.get("scl_clusters"):
        np.save('../codebooks/scalar_center_{}_{}_tr.npy'.format(cfg['scl_clusters'], cfg['note']),
                train_scalar_codebook(np.concatenate(scl_res), cfg['scl_clusters']))
    if cfg.get("scl_clusters_bl"):
        np.save('../codebooks/scalar_center_{}_{}_bl_tr.npy'.format(cfg['scl_clusters_bl'], cfg['note']),
                train_scalar_codebook(np.concatenate(scl_res_bl), cfg['scl_clusters_bl']))
    return codebook
