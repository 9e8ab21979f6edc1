"""Scalar-codebook k-means (train_cb.py:219-226) at a realistic size: fpc_kmeans1d against scikit-learn on the box's host cores."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd import train_cb
from sklearn.cluster import KMeans
n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400000, 256)
rs = np.random.RandomState(5)
v = (rs.laplace(size=n) * 0.1).astype(np.float32).astype(np.float64)
train_cb.train_scalar_codebook(v[:5000], 8, n_init=1)  # (library load, first launches)
for n_init in (1, 10):
    d = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = train_cb.train_scalar_codebook(v, k, n_init=n_init, details=d)
    dt = time.perf_counter() - t0
    print(f"fpc_kmeans1d n={n} k={k} n_init={n_init}: {dt:.3f} s (host draws + upload + device), {d['n_iter']} Lloyd iterations in the winning run, inertia {d['inertia']:.6g}")
t0 = time.perf_counter()
km = KMeans(n_clusters=k, random_state=0, n_init=1).fit(v[:, None])
ds = time.perf_counter() - t0
c1 = train_cb.train_scalar_codebook(v, k, n_init=1)
print(f"sklearn KMeans n_init=1: {ds:.2f} s ({km.n_iter_} iterations) -> about {10 * ds:.0f} s at n_init=10; max |centre difference| to fpc_kmeans1d {np.abs(c1 - km.cluster_centers_).max():.3g}")
