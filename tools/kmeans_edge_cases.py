"""k-means of the scalar codebooks: degenerate inputs on the device against the oracle (bit for bit) and scikit-learn (centres)."""
import sys; sys.path.insert(0, '.')
import numpy as np, warnings
warnings.filterwarnings("ignore")
from fpcodec_amd import train_cb
from oracle import kmeans1d_oracle as KO
from sklearn.cluster import KMeans
rs = np.random.RandomState(11)
cases = [("n=k=5", rs.normal(size=5), 5), ("n=300,k=256", rs.normal(size=300), 256), ("n=2048", rs.normal(size=2048), 3),
         ("n=2049", rs.normal(size=2049), 3), ("n=4097", rs.normal(size=4097), 7), ("all equal", np.full(1000, 0.25), 4),
         ("two values", np.repeat([1.0, -2.0], 700), 3), ("int-valued", rs.randint(0, 6, size=5000).astype(np.float64), 6),
         ("n=100000,k=100", (rs.laplace(size=100000) * .1).astype(np.float32).astype(np.float64), 100)]
bad = 0
for name, v, k in cases:
    d = {}
    c = train_cb.train_scalar_codebook(v, k, n_init=3, details=d)
    oc, oin, oit, oseeds = KO.fit(v, k, n_init=3)
    km = KMeans(n_clusters=k, random_state=0, n_init=3).fit(v[:, None])
    same = np.array_equal(c, oc) and np.array_equal(d["seeds"], oseeds) and d["inertia"] == oin and d["n_iter"] == oit
    dsk = np.abs(np.sort(c[:, 0]) - np.sort(km.cluster_centers_[:, 0])).max()
    print(f"{name:18s} oracle {'identical' if same else 'DIFFERENT'}; sklearn: max |dc| {dsk:.2e}, iters {d['n_iter']} / {km.n_iter_}, inertia {d['inertia']:.6g} / {km.inertia_:.6g}")
    bad += not same
sys.exit(1 if bad else 0)
