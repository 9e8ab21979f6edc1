"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the reference goldens."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    from fpcodec_amd import _lib
    _lib.require_gpu()
    return torch


@pytest.fixture(scope="module")
def cb_paths(synth):
    d = tempfile.mkdtemp()
    c = synth.codebooks()
    p = {}
    for k, v in c.items():
        p[k] = os.path.join(d, k + ".npy")
        np.save(p[k], v)
    rag = np.empty(2, dtype=object)
    rag[0] = c["vq_hi"][0]
    rag[1] = c["vq_hi"][1][:512]
    p["ragged"] = os.path.join(d, "ragged.npy")
    np.save(p["ragged"], rag, allow_pickle=True)
    return p


@pytest.fixture(scope="module")
def model(torch_cuda, synth):
    from fpcodec_amd.wavernn import Wavernn
    m = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    m.load_state_dict(synth.predictor_state_dict())
    return m


def test_forward_vs_golden_and_oracle(torch_cuda, model, synth, golden, oracle):
    torch = torch_cuda
    g = golden("g1_forward")
    x = synth.predictor_features(1, 300)
    y, h1, h2 = model.forward(torch.from_numpy(x))
    assert np.abs(y.cpu().numpy() - g["y_1x300"]).max() < 1e-5  # north_star tolerance
    assert np.abs(h1.cpu().numpy() - g["h1_1x300"]).max() < 1e-5
    P = oracle.Predictor(synth.predictor_state_dict())
    yo, h1o, h2o = P.forward(x)
    assert np.array_equal(y.cpu().numpy(), yo)  # same fmaf chains -> bit-identical
    assert np.array_equal(h2.cpu().numpy()[0], h2o)
    # stepwise with carried state (wavernn.py:194)
    a = b = None
    for t in range(4):
        yy, a, b = model.forward(torch.from_numpy(x[:, t:t + 1]), a, b)
        assert np.abs(yy.cpu().numpy()[:, 0] - g["step_y"][:, t]).max() < 1e-5


@pytest.mark.parametrize("tag,B,L,utt0,full,qtz", [
    ("full_1x300", 1, 300, 0, True, True), ("full_4x40", 4, 40, 20, True, True),
    ("hi_4x40", 4, 40, 20, False, True), ("raw_4x40", 4, 40, 20, True, False)])
def test_encoder_vs_golden(torch_cuda, model, synth, golden, cb_paths, tag, B, L, utt0, full, qtz):
    torch = torch_cuda
    g = golden("g2_encoder")
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"],
               bl_scl_cb_path=cb_paths["scl_lo"] if full else "", bl_cb_path=cb_paths["vq_lo"] if full else "")
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=utt0))
    c_in, r, r_qtz, r_under, ind1, ind2, cb_tot = model.encoder(cfg, feat, None, 0.09, 0.28, None, None, qtz)
    assert np.array_equal(ind1.cpu().numpy(), g[f"{tag}_ind1"])
    assert np.array_equal(ind2.cpu().numpy(), g[f"{tag}_ind2"])
    for i in range(5):
        ref = np.atleast_1d(g[f"{tag}_hist{i}"])
        got = np.atleast_1d(cb_tot[i])
        if ref.size == 1:
            assert got.sum() == 0
        else:
            assert np.array_equal(got, ref), i
    for k, v in (("c_in", c_in), ("r", r), ("r_qtz", r_qtz), ("r_under", r_under)):
        assert np.abs(v.cpu().numpy() - g[f"{tag}_{k}"]).max() < 1e-5, k


def test_encoder_bit_identical_to_oracle(torch_cuda, model, synth, oracle, cb_paths):
    torch = torch_cuda
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    P = oracle.Predictor(synth.predictor_state_dict())
    feat = synth.predictor_features(6, 120, utt0=100)
    o = P.encode(feat, CB, 0.09, 0.28, True)
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"],
               bl_scl_cb_path=cb_paths["scl_lo"], bl_cb_path=cb_paths["vq_lo"])
    out = model.encoder(cfg, torch.from_numpy(feat), None, 0.09, 0.28, qtz=True, return_indices=True)
    assert np.array_equal(out[7].cpu().numpy(), o["idx"])
    for k, v in (("c_in", out[0]), ("r", out[1]), ("r_qtz", out[2])):
        assert np.array_equal(v.cpu().numpy(), o[k]), k
    hs = CB.split_hist(o["hist"])
    for i in range(5):
        assert np.array_equal(np.atleast_1d(out[6][i]), hs[i])


def test_quantizers_vs_golden(torch_cuda, synth, golden, cb_paths):
    from fpcodec_amd.vq_func import vq_quantize, scl_quantize
    g = golden("g3_quant")
    rng = np.random.default_rng(7)
    r = rng.normal(0, 0.05, (256, 17)).astype(np.float32)
    for tag, key in (("s2", "vq_hi"), ("ragged", "ragged"), ("s1", "vq_lo")):
        qr, hs, idx = vq_quantize(r, cb_paths[key], return_indices=True)
        assert np.array_equal(qr, g[f"{tag}_qr"]), tag
        for i, h in enumerate(hs):
            assert np.array_equal(h, g[f"{tag}_hist{i}"])
        assert np.array_equal(idx[:, :len(hs)], g[f"{tag}_idx"][:, :len(hs)])
    xs = rng.normal(0, 0.1, (256, 1)).astype(np.float32)
    for tag, key in (("hi", "scl_hi"), ("lo", "scl_lo")):
        q, h = scl_quantize(xs, cb_paths[key])
        assert np.array_equal(q, g[f"scl_{tag}_q"])
        assert np.array_equal(h, g[f"scl_{tag}_hist"])


def test_vq_edge_cases(torch_cuda, cb_paths):
    from fpcodec_amd.vq_func import vq_quantize
    qr, hs = vq_quantize(np.zeros((0, 17), np.float32), cb_paths["vq_hi"])  # empty input
    assert qr.shape == (0, 17) and hs[0].sum() == 0
    d = tempfile.mkdtemp()
    cb = np.zeros((1, 8, 17))
    cb[0, 3] = cb[0, 5] = 0.25  # duplicate entries: lower index wins (stable sort, vq_func.py:20)
    p = os.path.join(d, "tie.npy")
    np.save(p, cb)
    qr, hs, idx = vq_quantize(np.full((1, 17), 0.25, np.float32), p, return_indices=True)
    assert idx[0, 0] == 3


def test_ceps2lpc_vs_golden_and_oracle(torch_cuda, golden, oracle, synth):
    torch = torch_cuda
    from fpcodec_amd.ceps2lpc import ceps2lpc_v
    g = golden("g4_ceps2lpc")
    feats = g["feats36"][0]
    e, lpc, rc = ceps2lpc_v(torch.from_numpy(feats[:, :20].copy()))
    assert np.abs(lpc.cpu().numpy() - g["lpc"]).max() < 2e-4
    lo, eo, rco = oracle.ceps2lpc(feats[:, :20])
    assert np.array_equal(lpc.cpu().numpy(), lo)
    assert float(e) == float(eo[-1])
    assert np.array_equal(rc.cpu().numpy().astype(np.float32), rco[-1])
    pk = synth.peaked_cepstra()
    e2, lpc2, rc2 = ceps2lpc_v(torch.from_numpy(pk), all_rows=True)
    lo2, eo2, _ = oracle.ceps2lpc(pk)
    assert np.array_equal(lpc2.cpu().numpy(), lo2)
    assert np.array_equal(lpc2.cpu().numpy() == 0, g["peaked_lpc"] == 0)  # early exits at the same order


def _voc_features(synth, oracle, B, T, utt0=0):
    f = synth.vocoder_features_raw(B, T, utt0=utt0)
    f[:, :, 20:] = oracle.ceps2lpc(f.reshape(-1, 36)[:, :20])[0].reshape(B, T, 16)
    return f


@pytest.fixture(scope="module")
def vocoder(torch_cuda, synth):
    from fpcodec_amd.lpcnet import LPCNet
    w = synth.lpcnet_weights()
    return LPCNet(w), w


def test_lpcnet_condition_bit_identical(torch_cuda, vocoder, synth, oracle):
    voc, w = vocoder
    f = _voc_features(synth, oracle, 2, 12)
    cf = voc.condition(f).cpu().numpy()
    orc = oracle.LPCNet(w)
    for b in range(2):
        assert np.array_equal(cf[b], orc.condition(f[b]))


def test_lpcnet_decode_bit_identical_small(torch_cuda, vocoder, synth, oracle):
    voc, w = vocoder
    B, T = 3, 6
    f = _voc_features(synth, oracle, B, T)
    sd = synth.seeds(B)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    orc = oracle.LPCNet(w)
    for b in range(B):
        ref = orc.synthesize(f[b], int(sd[b]))
        nz = np.nonzero(pcm[b] != ref)[0]
        assert nz.size == 0, f"utt {b}: first mismatch at sample {nz[:5]}"
    assert (pcm[:, :17] == 0).all()


def test_lpcnet_config2_single_stream_3s(torch_cuda, vocoder, synth, oracle):
    """BASELINE config 2: one 3 s utterance, fixed RNG, bit-compared with the oracle."""
    voc, w = vocoder
    f = _voc_features(synth, oracle, 1, 300)
    sd = synth.seeds(1)
    pcm = voc.synthesize(f, sd).cpu().numpy()[0]
    ref = oracle.LPCNet(w).synthesize(f[0], int(sd[0]))
    nz = np.nonzero(pcm != ref)[0]
    assert nz.size == 0, f"first mismatch at {nz[:5]}"


def test_lpcnet_batch_invariance(torch_cuda, vocoder, synth, oracle):
    """shard invariance: an utterance decodes identically whatever its batch position"""
    voc, w = vocoder
    f = _voc_features(synth, oracle, 5, 8, utt0=40)
    sd = synth.seeds(5, utt0=40)
    full = voc.synthesize(f, sd).cpu().numpy()
    one = voc.synthesize(f[3:4], sd[3:4]).cpu().numpy()
    assert np.array_equal(full[3], one[0])
    perm = np.array([4, 2, 0, 1, 3])
    p2 = voc.synthesize(f[perm], sd[perm]).cpu().numpy()
    assert np.array_equal(p2, full[perm])
