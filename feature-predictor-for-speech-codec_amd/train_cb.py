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


def train_scalar_codebook(values, n_clusters):
    """(n_clusters, 1) float64 centres of the scalar residuals: the reference's commented
    `KMeans(n_clusters, random_state=0).fit(values[:, None]).cluster_centers_` (train_cb.py:219-226)"""
    from sklearn.cluster import KMeans
    v = np.asarray(values, dtype=np.float64).reshape(-1, 1)
    return KMeans(n_clusters=int(n_clusters), random_state=0, n_init=10).fit(v).cluster_centers_.astype(np.float64)


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
