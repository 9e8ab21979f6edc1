"""Seeded synthetic weights, codebooks and features.

The reference ships no checkpoints, codebooks, features or audio (its .gitignore
excludes them all), so every input of tests and benchmarks is generated here from
``numpy.random.default_rng`` seeds.  Shapes follow the reference:

* predictor state_dict   src/models/wavernn.py:24-52   (10 tensors, PyTorch layout)
* codebooks              src/train_cb.py:125-130,217-221 (float64, (S,N,17) / (n,1))
* predictor features     src/datasets/dataset_syn.py:86-97 ((L,20) = feat/24.1)
* vocoder weights        xiph/LPCNet lpcnet.py (Keras layouts)
"""
import numpy as np

MAXI = 24.1  # src/synthesis_qtz.py:37, src/datasets/dataset_syn.py:27

SEED_WEIGHTS = 1001
SEED_CODEBOOKS = 1002
SEED_FEATURES = 1003
SEED_SAMPLING = 1004
SEED_VOCODER = 1005

# fc scale / feature sigmas chosen so the thresholded keep-rates land near the
# reference's training target of 0.3 (src/train_frame.py:204); achieved rates are
# recorded in DESIGN.md.
FC_SCALE = 0.02
SIGMA0 = 0.085
SIGMAK = 0.043


def predictor_state_dict(in_features=20, gru_units1=384, gru_units2=128, fc_units=18,
                         seed=SEED_WEIGHTS):
    """Random predictor weights in state_dict order, U(-1/sqrt(H), 1/sqrt(H)) like
    torch.nn.GRU/Linear default init; the FC layer is scaled by FC_SCALE."""
    rng = np.random.default_rng(seed)

    def u(shape, h):
        k = 1.0 / np.sqrt(h)
        return rng.uniform(-k, k, size=shape).astype(np.float32)

    h1, h2 = gru_units1, gru_units2
    sd = {
        "rnn1.weight_ih_l0": u((3 * h1, in_features), h1),
        "rnn1.weight_hh_l0": u((3 * h1, h1), h1),
        "rnn1.bias_ih_l0": u((3 * h1,), h1),
        "rnn1.bias_hh_l0": u((3 * h1,), h1),
        "rnn2.weight_ih_l0": u((3 * h2, h1), h2),
        "rnn2.weight_hh_l0": u((3 * h2, h2), h2),
        "rnn2.bias_ih_l0": u((3 * h2,), h2),
        "rnn2.bias_hh_l0": u((3 * h2,), h2),
        "dual_fc.0.weight": (u((fc_units, h2), h2) * np.float32(FC_SCALE)).astype(np.float32),
        "dual_fc.0.bias": (u((fc_units,), h2) * np.float32(FC_SCALE)).astype(np.float32),
    }
    return sd


def codebooks(seed=SEED_CODEBOOKS):
    """float64 codebooks in the reference's file shapes."""
    rng = np.random.default_rng(seed)
    return {
        "vq_hi": np.stack([rng.normal(0, 0.05, (1024, 17)), rng.normal(0, 0.02, (1024, 17))]),
        "vq_lo": rng.normal(0, 0.02, (1, 512, 17)),
        "scl_hi": rng.normal(0, 0.1, (256, 1)),
        "scl_lo": rng.normal(0, 0.03, (16, 1)),
    }


def predictor_features(B, L=300, seed=SEED_FEATURES, utt0=0):
    """(B,L,20) float32 in the normalised domain (/24.1): AR(1) cepstra, held
    integer pitch period, uniform pitch correlation."""
    out = np.zeros((B, L, 20), np.float32)
    sig = np.array([SIGMA0] + [SIGMAK * 0.9 ** k for k in range(1, 18)])
    for b in range(B):
        rng = np.random.default_rng(seed + utt0 + b)
        c = np.zeros(18)
        eps = rng.normal(size=(L, 18))
        for t in range(L):
            c = 0.95 * c + np.sqrt(1 - 0.95 ** 2) * sig * eps[t]
            out[b, t, :18] = c
        P = np.repeat(rng.integers(40, 256, size=(L + 9) // 10), 10)[:L]
        out[b, :, 18] = ((P - 100.1) / 50.0) / MAXI
        out[b, :, 19] = rng.uniform(-0.4, 0.4, size=L) / MAXI
    return out


def lpcnet_weights(seed=SEED_VOCODER, density=(0.05, 0.05, 0.2)):
    """Random LPCNet weights in Keras layouts.  GRU_A's recurrent matrix is pruned to
    8x4 blocks (8 outputs x 4 inputs) per gate at the given densities, diagonal kept,
    like LPCNet's Sparsify callback."""
    rng = np.random.default_rng(seed)

    def glorot(shape, fan_in, fan_out):
        s = np.sqrt(2.0 / (fan_in + fan_out))
        return rng.normal(0, s, size=shape).astype(np.float32)

    w = {
        "embed_pitch": rng.uniform(-0.05, 0.05, (256, 64)).astype(np.float32),
        "conv1_kernel": glorot((3, 84, 128), 3 * 84, 128),
        "conv1_bias": np.zeros(128, np.float32),
        "conv2_kernel": glorot((3, 128, 128), 3 * 128, 128),
        "conv2_bias": np.zeros(128, np.float32),
        "dense1_kernel": glorot((128, 128), 128, 128),
        "dense1_bias": rng.normal(0, 0.01, 128).astype(np.float32),
        "dense2_kernel": glorot((128, 128), 128, 128),
        "dense2_bias": rng.normal(0, 0.01, 128).astype(np.float32),
        "embed_sig": rng.uniform(-1.7, 1.7, (256, 128)).astype(np.float32) * np.float32(0.1),
        "gru_a_kernel": glorot((512, 1152), 512, 384),
        "gru_a_bias": rng.normal(0, 0.05, (2, 1152)).astype(np.float32),
        "gru_b_kernel": glorot((512, 48), 512, 16),
        "gru_b_recurrent": glorot((16, 48), 16, 16),
        "gru_b_bias": rng.normal(0, 0.05, (2, 48)).astype(np.float32),
        "md_kernel": glorot((256, 16, 2), 16, 256) * np.float32(4.0),
        "md_bias": rng.normal(0, 0.3, (256, 2)).astype(np.float32),
        "md_factor": rng.uniform(0.5, 1.5, (256, 2)).astype(np.float32),
    }
    # bias the tree towards excitation values near 128 (small residual), as a trained
    # LPCNet does: nodes under the MSB=1 half prefer bit 0, nodes under MSB=0 prefer bit 1
    for n in range(2, 256):
        top = (n >> (n.bit_length() - 2)) & 1
        w["md_bias"][n] += np.float32(-1.2 if top else 1.2)
    N = 384
    rec = glorot((N, 3 * N), N, N) * np.float32(2.0)
    for g, d in enumerate(density):
        A = rec[:, g * N:(g + 1) * N].copy()  # (in, out)
        diag = np.diag(A).copy()
        np.fill_diagonal(A, 0)
        At = A.T  # (out, in)
        blk = At.reshape(N // 8, 8, N // 4, 4)
        energy = (blk * blk).sum(axis=(1, 3)).reshape(-1)
        nkeep = int(round(energy.size * d))
        thresh = np.sort(energy)[energy.size - nkeep]
        mask = (energy >= thresh).reshape(N // 8, N // 4)
        mask = np.repeat(np.repeat(mask, 8, axis=0), 4, axis=1)
        At = At * mask
        A = At.T.copy()
        A[np.arange(N), np.arange(N)] = diag
        rec[:, g * N:(g + 1) * N] = A
    w["gru_a_recurrent"] = np.ascontiguousarray(rec.astype(np.float32))
    return w


def vocoder_features_raw(B, T=300, seed=SEED_FEATURES, utt0=0):
    """Stand-alone (B,T,36) vocoder features that do not need the encoder: cepstra and
    pitch from predictor_features x 24.1, LPC columns left zero (callers fill them
    with ceps2lpc)."""
    f20 = predictor_features(B, T, seed, utt0) * np.float32(MAXI)
    out = np.zeros((B, T, 36), np.float32)
    out[:, :, :20] = f20
    return out


def seeds(B, utt0=0):
    return (np.arange(B, dtype=np.uint64) + np.uint64(SEED_SAMPLING + utt0)).astype(np.uint64)


def peaked_cepstra():
    """(8,20) un-normalised cepstra whose band spectrum has one dominant band; several of
    them make the Levinson recursion of ceps2lpc stop early (ceps2lpc_vct.py:82-85)."""
    T = np.array([[np.cos((i + .5) * j * np.pi / 18) * (np.sqrt(.5) if j == 0 else 1.0)
                   for j in range(18)] for i in range(18)])
    rows = []
    for band, hi, lo in [(5, 6, -2), (3, 7, -3), (10, 5, -4), (0, 8, -2), (8, 4, -4), (2, 3, 0),
                         (12, 6, -1), (6, 9, -3)]:
        E = np.full(18, float(lo))
        E[band] = hi
        c = (E @ T) * np.sqrt(2.0 / 18)
        c[0] -= 4.0
        rows.append(np.concatenate([c, [0.0, 0.0]]))
    return np.array(rows, np.float32)


def cb_training_vectors(n, seed=SEED_CODEBOOKS, seed_offset=0, ndims=17):
    """Residual-like training vectors for the codebook trainer (train_cb.py:170-178 hands over float32
    rows): a mixture of 12 anisotropic Gaussian clusters around zero, float32."""
    rng = np.random.default_rng(seed + 1000 + seed_offset)
    centres = rng.normal(0, 0.12, (12, ndims))
    scale = rng.uniform(0.01, 0.05, (12, ndims))
    which = rng.integers(0, 12, n)
    return (centres[which] + rng.normal(0, 1, (n, ndims)) * scale[which]).astype(np.float32)
