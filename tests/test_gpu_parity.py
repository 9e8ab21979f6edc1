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


def test_predictor_other_dimensions_bit_identical_to_oracle(torch_cuda, synth, oracle, cb_paths, monkeypatch):
    """the predictor kernels are generic in the layer sizes (in <= 64, gru_units1 <= 512, gru_units2 <= 256, fc <= 32): forward
    incl. carried states at five other shapes -- different segment counts per row (1, 2, 4), the input product from its LDS
    copy or streamed, row quads that do not fill a wave -- and the closed-loop encoder at the shapes it supports (20 inputs, 18
    outputs), against the CPU oracle bit for bit, on 1 / 2 / 4 workgroups per utterance (the row-split two-role kernels: the
    generic-shape family)"""
    from fpcodec_amd.wavernn import Wavernn
    torch = torch_cuda
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    rng = np.random.default_rng(77)
    for dims in ((20, 256, 64, 18), (20, 128, 256, 18), (20, 512, 32, 18), (64, 192, 96, 24), (36, 64, 16, 12)):
        inf, h1, h2, fc = dims
        sd = synth.predictor_state_dict(inf, h1, h2, fc, seed=900 + h1 + h2)
        P = oracle.Predictor(sd)
        B, L = 5, 14
        if inf == 20:
            feat = synth.predictor_features(B, L, utt0=8100)
        else:
            feat = (rng.normal(size=(B, L, inf)) * 0.1).astype(np.float32)
        y0, a0, b0 = P.forward(feat)
        y1, a1, b1 = P.forward(feat[:, :4], a0, b0)
        enc0 = P.encode(feat, CB, 0.09, 0.28, True) if (inf, fc) == (20, 18) else None
        m = Wavernn(inf, h1, h2, fc)
        m.load_state_dict(sd)
        x = torch.from_numpy(feat).cuda()
        for n in ("0", "2", "4"):
            if int(n) and (h1 % (4 * int(n)) or h2 % (4 * int(n))):
                continue
            monkeypatch.setenv("FPC_PRED_SPLIT", n)
            y, a, b = m.forward(x)
            y2, a2, b2 = m.forward(x[:, :4].contiguous(), a, b)
            for got, ref in ((y, y0), (a, a0), (b, b0), (y2, y1), (a2, a1), (b2, b1)):
                assert np.array_equal(got.cpu().numpy().reshape(ref.shape), ref), (dims, n)
            if enc0 is not None:
                out = m.encoder(cfg, x, None, 0.09, 0.28, qtz=True, return_indices=True)
                assert np.array_equal(out[7].cpu().numpy(), enc0["idx"]), (dims, n)
                for k, v in (("c_in", out[0]), ("r", out[1]), ("r_qtz", out[2])):
                    assert np.array_equal(v.cpu().numpy(), enc0[k]), (dims, n, k)


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
    assert np.abs(lpc.cpu().numpy() - g["lpc"]).max() < 2.5e-5  # 10 x the measured 2.4e-6 (FFT vs cosine sum)
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


@pytest.mark.parametrize("density,variant", [((0.02, 0.02, 0.10), 208), ((0.05, 0.05, 0.20), 408),
                                             ((0.05, 0.05, 0.26), 1616)])
def test_lpcnet_decode_kernel_instances(torch_cuda, synth, oracle, density, variant):
    """every decode-kernel instance (row groups 2/4/16 lanes wide) against the oracle, on frames
    forced fully unvoiced (constant tail cut) and fully voiced (sharpened pdf, fifth barrier)"""
    from fpcodec_amd.lpcnet import LPCNet
    w = synth.lpcnet_weights(density=density)
    voc = LPCNet(w)
    assert voc.kernel_variant() == variant
    orc = oracle.LPCNet(w)
    B, T = 2, 4
    f = _voc_features(synth, oracle, B, T)
    f[0, :, 19] = -0.4   # pitch gain -> exponent 0 on every frame
    f[1, :, 19] = 0.9    # -> exponent 0.85 on every frame
    sd = synth.seeds(B, utt0=40)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    for b in range(B):
        ref = orc.synthesize(f[b], int(sd[b]))
        nz = np.nonzero(pcm[b] != ref)[0]
        assert nz.size == 0, f"variant {variant} utt {b}: first mismatch at sample {nz[:5]}"


def test_lpcnet_too_dense_is_refused(torch_cuda, synth):
    """a recurrent matrix that does not fit the register-resident layout fails loudly at create"""
    from fpcodec_amd._lib import FpcError
    from fpcodec_amd.lpcnet import LPCNet
    with pytest.raises(FpcError, match="too dense"):
        LPCNet(synth.lpcnet_weights(density=(0.3, 0.3, 0.5)))
    # the same refusal at the end of the `[Saved_Model]` route: Keras-named weights -> tools/h5_to_npz mapping -> .npz -> load
    import importlib.util
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("h5_to_npz", os.path.join(root, "tools", "h5_to_npz.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for density, ok in (((0.05, 0.05, 0.2), True), ((0.3, 0.3, 0.5), False)):
        w = synth.lpcnet_weights(density=density)
        named = {f"model_weights/{layer}/{layer}/{weight}:0": w[key] for key, layer, weight, shape in m.MAPPING}
        path = os.path.join(tempfile.mkdtemp(), "m.npz")
        np.savez(path, **m.from_keras_named(named))
        if ok:
            assert LPCNet.load(path).kernel_variant() == 408
        else:
            with pytest.raises(FpcError, match="too dense"):
                LPCNet.load(path)


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


def test_lpcnet_config3_full_batch_properties(torch_cuda, vocoder, synth, oracle):
    """BASELINE config 3 at full size (256 x 3 s, one workgroup per CU): utterances repeated with the
    same seed must come out identical from every CU, an utterance must not depend on its neighbours,
    and three rows are bit-compared with the oracle."""
    voc, w = vocoder
    B, T, nu = 256, 300, 8
    base = _voc_features(synth, oracle, nu, T, utt0=900)
    f = np.tile(base, (B // nu, 1, 1))
    sd = np.tile(synth.seeds(nu, utt0=900), B // nu)
    sd[200:] += 7  # the last 56 get their own random streams
    pcm = voc.synthesize(f, sd).cpu().numpy()
    assert pcm.shape == (B, T * 160) and (pcm[:, :17] == 0).all()
    for r in range(1, 200 // nu):  # replicas of the first 8 on other CUs
        assert np.array_equal(pcm[r * nu:(r + 1) * nu], pcm[:nu]), f"replica block {r} differs"
    assert not np.array_equal(pcm[200], pcm[0])  # another seed, another waveform
    orc = oracle.LPCNet(w)
    for b in (0, 133, 255):
        ref = orc.synthesize(f[b], int(sd[b]))
        nz = np.nonzero(pcm[b] != ref)[0]
        assert nz.size == 0, f"utt {b}: first mismatch at sample {nz[:5]}"


def test_lpcnet_more_utterances_than_cus(torch_cuda, vocoder, synth, oracle):
    """600 workgroups on 256 CUs: the grid runs in several rounds; rows spot-checked against the oracle"""
    voc, w = vocoder
    B, T, nu = 600, 2, 6
    base = _voc_features(synth, oracle, nu, T, utt0=300)
    f = np.tile(base, (B // nu, 1, 1))
    sd = synth.seeds(B, utt0=300)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    orc = oracle.LPCNet(w)
    for b in (0, 255, 256, 599):
        assert np.array_equal(pcm[b], orc.synthesize(f[b], int(sd[b]))), f"utt {b}"


@pytest.mark.parametrize("density,variant", [((0.02, 0.02, 0.10), 208), ((0.05, 0.05, 0.20), 408),
                                             ((0.05, 0.05, 0.26), 1616)])
def test_lpcnet_two_utterances_per_workgroup_small(torch_cuda, synth, oracle, density, variant):
    """k_decode2 (fpc_lpcnet_set_pairing, lpcnet_decode2.h) against the oracle and against k_decode: an odd batch (the last
    workgroup decodes its one utterance twice), pairs whose members are voiced / unvoiced in different frames (the fifth
    barrier is taken when either is), both instances of the kernel; a model whose row groups are too wide for the packed
    planes (1616) keeps one utterance per workgroup whatever is asked"""
    from fpcodec_amd.lpcnet import LPCNet
    w = synth.lpcnet_weights(density=density)
    voc = LPCNet(w)
    assert voc.kernel_variant() == variant
    orc = oracle.LPCNet(w)
    B, T = 5, 5
    f = _voc_features(synth, oracle, B, T, utt0=11)
    f[0, :, 19] = -0.4         # unvoiced throughout, paired with ...
    f[1, :, 19] = 0.9          # ... a voiced utterance
    f[2, 1:3, 19] = 0.8        # a pair voiced in different frames
    f[3, 2:4, 19] = 0.7
    sd = synth.seeds(B, utt0=11)
    ref = np.stack([orc.synthesize(f[b], int(sd[b])) for b in range(B)])
    for mode in (-1, 1):
        voc.set_pairing(mode)
        pcm = voc.synthesize(f, sd).cpu().numpy()
        assert voc.last_streams_per_workgroup() == (2 if mode == 1 and variant != 1616 else 1)
        nz = np.argwhere(pcm != ref)
        assert nz.size == 0, f"variant {variant} pairing {mode}: first mismatch at {nz[:3].tolist()}"
    # one utterance alone is never paired; the default pairs only when the batch exceeds the compute units
    voc.set_pairing(1)
    assert np.array_equal(voc.synthesize(f[2:3], sd[2:3]).cpu().numpy()[0], ref[2]) and voc.last_streams_per_workgroup() == 1
    voc.set_pairing(0)
    voc.synthesize(f, sd)
    assert voc.last_streams_per_workgroup() == 1


def test_lpcnet_512_utterances_two_per_workgroup_full_size(torch_cuda, vocoder, synth, oracle):
    """B = 512 x 3 s on one GPU (configs 4 / 5 on fewer than 8 GPUs; SURVEY.md 7: "more than one stream per CU is the
    lever for B > 256"): by default two utterances per workgroup; every sample equal to the rounds of k_decode, four rows
    equal to the oracle; B = 256 still takes one utterance per workgroup"""
    import concurrent.futures as cf
    voc, w = vocoder
    B, T, nu = 512, 300, 8
    base = _voc_features(synth, oracle, nu, T, utt0=1200)
    base[1, 0::2, 19] = 0.9   # one of the eight alternates voiced / unvoiced frames
    f = np.tile(base, (B // nu, 1, 1))
    sd = synth.seeds(B, utt0=1200)
    voc.set_pairing(0)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    assert voc.last_streams_per_workgroup() == 2
    t_pair = voc.last_decode_ms()
    voc.set_pairing(-1)
    rounds = voc.synthesize(f, sd).cpu().numpy()
    assert voc.last_streams_per_workgroup() == 1
    t_rounds = voc.last_decode_ms()
    voc.set_pairing(0)
    assert np.array_equal(pcm, rounds), "k_decode2 and k_decode disagree"
    assert t_pair < t_rounds, f"two per workgroup {t_pair:.1f} ms, rounds of k_decode {t_rounds:.1f} ms"
    orc = oracle.LPCNet(w)
    rows = (0, 255, 257, 511)
    with cf.ThreadPoolExecutor(4) as ex:
        refs = list(ex.map(lambda b: orc.synthesize(f[b], int(sd[b])), rows))
    for b, ref in zip(rows, refs):
        nz = np.nonzero(pcm[b] != ref)[0]
        assert nz.size == 0, f"utt {b}: first mismatch at sample {nz[:5]}"
    voc.synthesize(f[:256], sd[:256])
    assert voc.last_streams_per_workgroup() == 1


def test_vocoder_long_utterances_two_per_workgroup(torch_cuda, vocoder, synth, oracle):
    """two 15-second utterances through one workgroup of k_decode2 (see test_vocoder_long_utterance_parity)"""
    import concurrent.futures as cf
    voc, w = vocoder
    T = 1500
    f = _voc_features(synth, oracle, 2, T, utt0=17)
    f[1, 100:900, 19] = 0.8
    sd = synth.seeds(2, utt0=17)
    voc.set_pairing(1)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    assert voc.last_streams_per_workgroup() == 2
    voc.set_pairing(0)
    orc = oracle.LPCNet(w)
    with cf.ThreadPoolExecutor(2) as ex:
        refs = list(ex.map(lambda b: orc.synthesize(f[b], int(sd[b])), range(2)))
    for b in range(2):
        nz = np.nonzero(pcm[b] != refs[b])[0]
        assert nz.size == 0, f"utt {b}: first mismatch at sample {nz[:5]}"


def test_lpcnet_shortest_inputs(torch_cuda, vocoder, synth, oracle):
    """one frame (143 audible samples after the 17 skipped ones) and a single utterance"""
    voc, w = vocoder
    f = _voc_features(synth, oracle, 2, 1, utt0=77)
    sd = synth.seeds(2, utt0=77)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    orc = oracle.LPCNet(w)
    for b in range(2):
        assert np.array_equal(pcm[b], orc.synthesize(f[b], int(sd[b])))
    one = voc.synthesize(f[1:2], sd[1:2]).cpu().numpy()
    assert np.array_equal(one[0], pcm[1])


def test_synthesis_qtz_harness_vs_reference_golden(torch_cuda, synth, golden, cb_paths, tmp_path, monkeypatch):
    """rows a9/a10: the synthesis_qtz.py driver (checkpoint file -> encoder -> x24.1 -> ceps2lpc ->
    36-float frames, src/synthesis_qtz.py:67-166) against the reference's own output (G2+G4)."""
    torch = torch_cuda
    from fpcodec_amd.config import default_cfg
    from fpcodec_amd.synthesis_qtz import synthesis
    label, epoch = "0722_001326", 4000  # README.md:44
    src = tmp_path / "src"
    (tmp_path / "saved_models" / label).mkdir(parents=True)
    src.mkdir()
    sd = {k: torch.from_numpy(v) for k, v in synth.predictor_state_dict().items()}
    torch.save(sd, tmp_path / "saved_models" / label / f"{label}_{epoch}.pth")  # src/utils.py:134,146
    monkeypatch.chdir(src)  # the reference runs with src/ as cwd (relative ../saved_models, ../samples)
    cfg = default_cfg()
    cfg.update(model_label_f=label, epoch_f=epoch, gru_units1=384, gru_units2=128, fc_units=18, l1=0.09, l2=0.28,
               qtz=True, note="t", total_secs=3, scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"],
               bl_scl_cb_path=cb_paths["scl_lo"], bl_cb_path=cb_paths["vq_lo"])
    nm = np.zeros((1, 300, 36), np.float32)
    nm[:, :, :20] = synth.predictor_features(1, 300)
    out = synthesis(cfg, utterances=[("1272-128104-0001", torch.from_numpy(nm))])
    name, feats, r = out[0]
    g4, g2 = golden("g4_ceps2lpc"), golden("g2_encoder")
    assert feats.shape == (1, 300, 36)
    assert np.abs(feats.cpu().numpy()[..., :20] - g4["feats36"][..., :20]).max() < 1e-4  # 24.1 x 1e-5
    assert np.abs(feats.cpu().numpy()[..., 20:] - g4["feats36"][..., 20:]).max() < 2.5e-5
    assert np.abs(r.cpu().numpy() - g2["full_1x300_r"]).max() < 1e-5
    saved = np.load(tmp_path / "samples" / label / "1272-128104-0001_r_t.npy")  # synthesis_qtz.py:166 naming
    assert saved.shape == (1, 300, 18)


def test_end_to_end_config5_pipeline_and_bitrate(torch_cuda, model, vocoder, synth, oracle, cb_paths):
    """BASELINE config 5 in miniature: encode + VQ -> ceps2lpc -> LPCNet decode, GPU vs the oracle
    pipeline bit for bit, and the bitrate figure from the codebook-usage entropies."""
    torch = torch_cuda
    from fpcodec_amd.synthesis_qtz import encode_features
    from fpcodec_amd.vq_func import cal_entropy
    voc, w = vocoder
    B, L = 3, 40
    nm = np.zeros((B, L, 36), np.float32)
    nm[:, :, :20] = synth.predictor_features(B, L, utt0=300)
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"], l1=0.09, l2=0.28, qtz=True)
    feats, r, ind1, ind2, cb_tot = encode_features(model, cfg, torch.from_numpy(nm))
    sd = synth.seeds(B, utt0=300)
    pcm = voc.synthesize(feats, sd).cpu().numpy()
    # oracle pipeline
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    o = oracle.Predictor(synth.predictor_state_dict()).encode(nm[:, :, :20], CB, 0.09, 0.28, True)
    cin = o["c_in"] * np.float32(synth.MAXI)
    lpc = oracle.ceps2lpc(cin.reshape(-1, 20))[0].reshape(B, L, 16)
    f36 = np.concatenate([cin, lpc], -1)
    assert np.array_equal(feats.cpu().numpy(), f36)
    orc = oracle.LPCNet(w)
    for b in range(B):
        assert np.array_equal(pcm[b], orc.synthesize(f36[b], int(sd[b])))
    # bits per frame = flag bits + sum over used codebooks of usage-weighted entropy (generate_qtz_features.py:94-101,202)
    hs = CB.split_hist(o["hist"])
    for i in range(5):
        assert np.array_equal(np.atleast_1d(cb_tot[i]), hs[i])
    ent = [cal_entropy(h) if np.sum(h) > 0 else 0.0 for h in cb_tot]
    n = B * L
    bits = sum(e * np.sum(h) for e, h in zip(ent, cb_tot)) / n
    assert 5.0 < bits < 30.0  # scalar (<=8) + 2-stage VQ (<=20) weighted by keep-rates, + below-threshold books


def test_lpcnet_cli_three_args(torch_cuda, vocoder, synth, oracle, tmp_path):
    """`test_lpcnet.py [model] [features] [out]` surface (README.md:47): raw f32 in, raw int16 out"""
    from fpcodec_amd import lpcnet as L
    voc, w = vocoder
    voc.save(str(tmp_path / "model.npz"))
    f = _voc_features(synth, oracle, 1, 4)[0]
    f.astype("<f4").tofile(tmp_path / "feat.f32")
    assert L.main([str(tmp_path / "model.npz"), str(tmp_path / "feat.f32"), str(tmp_path / "out.pcm")]) == 0
    pcm = np.fromfile(tmp_path / "out.pcm", dtype="<i2")
    ref = oracle.LPCNet(w).synthesize(f, 0)
    assert pcm.size == 4 * 160 - 17 and np.array_equal(pcm, ref[17:])
    assert L.main(["only", "two"]) == 2


# ---- codebook training on the GPU ----
def test_cb_primitives_vs_golden_and_oracle(torch_cuda, synth, golden, oracle):
    from fpcodec_amd import cb_func
    g = golden("g7_cb_train")
    data = synth.cb_training_vectors(3000)
    cb0 = synth.cb_training_vectors(40, seed_offset=1).astype(np.float64)[:32]
    assert np.array_equal(cb_func.find_nearest(data, cb0), g["idx"])
    cb1, count = cb_func.update(data, cb0, 32, return_count=True)
    assert np.array_equal(cb1, g["cb1"]) and count.sum() == 3000
    assert np.array_equal(cb_func.quantize(cb1, data[:500]), g["qd"])
    far = cb0.copy()
    far[5] += 100.0
    far[17] -= 100.0
    cb2, count2 = cb_func.update(data, far, 32, return_count=True)
    assert np.array_equal(cb2, g["cb2"]) and count2[5] == 0 and not cb2[5].any()
    assert np.array_equal(cb_func.mean0(data), np.mean(data, 0).astype(np.float64))
    # ragged sizes: one vector, one entry, sizes off the block boundaries
    for nv, e in ((1, 1), (1, 7), (65, 3), (2049, 33), (4097, 1)):
        d = synth.cb_training_vectors(nv, seed_offset=nv)
        c = synth.cb_training_vectors(e, seed_offset=1000 + e).astype(np.float64)
        assert np.array_equal(cb_func.find_nearest(d, c), oracle.cb_find_nearest(d, c)), (nv, e)
        assert np.array_equal(cb_func.update(d, c, e), oracle.cb_update(d, c, e)), (nv, e)


def test_cb_vq_train_vs_reference_golden(torch_cuda, synth, golden):
    from fpcodec_amd import cb_func
    g = golden("g7_cb_train")
    data = synth.cb_training_vectors(3000)
    np.random.seed(20221104)
    assert np.array_equal(cb_func.vq_train(data, np.zeros((24, 17)), 24), g["cbt"])
    r = data.copy()
    np.random.seed(7)
    for i in range(2):
        c = cb_func.vq_train(r, np.zeros((8, 17)), 8)
        assert np.array_equal(c, g[f"stage{i}"])
        r = cb_func.quantize(c, r) - r
    assert np.array_equal(r, g["r_final"])


def test_cb_update_production_size_properties(torch_cuda, synth, oracle):
    """400 000 vectors x 1024 entries (a 5000-utterance batch at keep-rate 0.3): size-independent
    properties, and three cells recomputed in float64 on the host in index order"""
    torch = torch_cuda
    from fpcodec_amd import cb_func
    nv, e = 400_000, 1024
    data = synth.cb_training_vectors(nv, seed_offset=5)
    cb = data[::nv // e][:e].astype(np.float64) + 1e-3
    d = torch.from_numpy(data).cuda()
    idx = cb_func.find_nearest(d, cb)
    new, count = cb_func.update(d, cb, e, return_count=True)
    assert count.sum() == nv and np.array_equal(count, np.bincount(idx, minlength=e).astype(np.float64))
    sel = np.random.default_rng(0).integers(0, nv, 400)  # the chosen entry really is the (first) nearest
    dist = ((data[sel].astype(np.float64)[:, None, :] - cb[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(idx[sel], dist.argmin(1))
    for n in (0, 511, 1023):
        acc = np.zeros(17)
        for i in np.nonzero(idx == n)[0]:
            acc += data[i]
        assert np.array_equal(new[n], acc / (count[n] + 1e-20))
    again = cb_func.update(d, cb, e)  # deterministic
    assert np.array_equal(again, new)
    d2 = float(((cb_func.quantize(new, data[:20000]) - data[:20000]) ** 2).sum())
    d1 = float(((cb_func.quantize(cb, data[:20000]) - data[:20000]) ** 2).sum())
    assert d2 <= d1  # a Lloyd step does not increase the distortion (up to the re-assignment)


# ---- receiver side on the GPU ----
@pytest.mark.parametrize("which", ["full", "hi_only", "one_stage"])
def test_decode_features_closes_the_loop(torch_cuda, model, synth, oracle, cb_paths, which):
    """encode -> symbols -> bits -> symbols -> decode gives back the encoder's own reconstruction bit for bit"""
    torch = torch_cuda
    from fpcodec_amd import bitstream
    from fpcodec_amd.vq_func import load_codebooks
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    if which == "hi_only":
        cfg.update(bl_scl_cb_path="", bl_cb_path="")
    if which == "one_stage":
        cfg.update(cb_path=cb_paths["vq_lo"], bl_cb_path="")
    feat = synth.predictor_features(5, 90, utt0=800)
    out = model.encoder(cfg, torch.from_numpy(feat), None, 0.09, 0.28, qtz=True, return_indices=True)
    c_in, idx = out[0].cpu().numpy(), out[7].cpu().numpy()
    sizes = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg.get("bl_cb_path") or None,
                           cfg.get("bl_scl_cb_path") or None).sizes
    rebuilt = np.stack([bitstream.unpack(bitstream.pack(idx[b], sizes)[0], feat.shape[1], sizes) for b in range(5)])
    assert np.array_equal(rebuilt, idx)
    dec = model.decode_indices(cfg, rebuilt, feat[:, :, 18:20]).cpu().numpy()
    assert np.array_equal(dec, c_in)
    if which == "full":
        c = synth.codebooks()
        CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
        assert np.array_equal(dec, oracle.Predictor(synth.predictor_state_dict()).decode(CB, idx, feat[:, :, 18:20]))
        bad = rebuilt.copy()
        bad[0, 3, 0] = 100000  # not a symbol of any scalar codebook
        from fpcodec_amd._lib import FpcError
        with pytest.raises(FpcError, match="outside its codebook"):
            model.decode_indices(cfg, bad, feat[:, :, 18:20])


# ---- predictor training step on the GPU ----
def test_train_step_vs_torch_golden_and_oracle(torch_cuda, synth, golden, oracle):
    """two Adam steps: GPU == oracle bit for bit (losses, every gradient, every parameter); both within the
    tolerances of tests/test_oracle_golden.py of torch autograd + torch.optim.Adam on the reference's Wavernn"""
    torch = torch_cuda
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    from test_oracle_golden import check_train_against_golden
    m = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    m.load_state_dict(synth.predictor_state_dict())
    feat = synth.predictor_features(6, 40, utt0=4000)
    tr = Trainer(m, lr=1e-4, max_batch=8, max_frames=64)
    ref = oracle.Trainer(synth.predictor_state_dict(), lr=1e-4)
    l0, r0 = tr.step(feat), ref.step(feat)
    g1 = tr.gradients()
    for k in g1:
        assert np.array_equal(g1[k], ref.g[k]), k
    l1, r1 = tr.step(feat), ref.step(feat)
    assert (l0, l1) == (r0, r1)
    tr.sync()
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    for k in sd:
        assert np.array_equal(sd[k], ref.p[k]), k
    check_train_against_golden(golden("g8_train_step"), (l0, l1), g1, sd)
    # the updated weights are live in the inference entry points
    y = m.forward(torch.from_numpy(feat[:1]))[0].cpu().numpy()
    yo = oracle.Predictor(sd).forward(feat[:1])[0]
    assert np.array_equal(y, yo)


def test_train_backward_weights_stationary_equals_row_split_and_oracle(torch_cuda, synth, oracle, monkeypatch):
    """back-propagation on the weights-stationary kernel (csrc/predictor_bwd_ws.h: 16 utterances a group on the 32
    workgroups of an XCD, the transposed slices resident, W^T d on f32 MFMA, the two recurrences as two concurrent tracks of
    four waves) against the row-split kernel (FPC_TRAIN_BWD_ROWSPLIT=1: one utterance per workgroup) -- loss, every gradient,
    every parameter after two steps, bit for bit -- for 2 full groups + a part-filled one, for the shortest sequence (2
    frames), for more groups than the chip has XCDs (130 utterances: 9 groups) and on both hop paths; and against the CPU
    oracle at 19 x 9 (a full and a part-filled group).  (tools/stress_train_bwd.py: eight more shapes, repeated)"""
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn

    def run(feat, steps=2):
        m = Wavernn(20, 384, 128, 18)
        m.load_state_dict(synth.predictor_state_dict())
        tr = Trainer(m, lr=1e-3, max_batch=feat.shape[0], max_frames=feat.shape[1])
        losses = [tr.step(feat) for _ in range(steps)]
        g = tr.gradients()
        tr.sync()
        sd = m.state_dict()
        return [np.float32(losses)] + [g[k] for k in sorted(g)] + [sd[k].numpy() for k in sorted(sd)]

    for B, L in ((37, 21), (16, 2), (3, 5), (130, 7)):
        feat = synth.predictor_features(B, L, utt0=4400)
        monkeypatch.setenv("FPC_TRAIN_BWD_ROWSPLIT", "1")
        ref = run(feat)
        monkeypatch.delenv("FPC_TRAIN_BWD_ROWSPLIT")
        for fast in ("1", "0"):
            monkeypatch.setenv("FPC_FAST_HOP", fast)
            got = run(feat)
            for k, (a, b) in enumerate(zip(ref, got)):
                assert not np.isnan(b).any() and np.array_equal(a, b), (B, L, fast, k)
        monkeypatch.delenv("FPC_FAST_HOP")
    feat = synth.predictor_features(19, 9, utt0=4500)
    got = run(feat, steps=1)
    ref = oracle.Trainer(synth.predictor_state_dict(), lr=1e-3)
    l0 = ref.step(feat)
    assert np.float32(l0) == got[0][0]
    for k, name in enumerate(sorted(ref.g)):
        assert np.array_equal(got[1 + k], ref.g[name]), name
    for k, name in enumerate(sorted(ref.p)):
        assert np.array_equal(got[1 + len(ref.g) + k], ref.p[name]), name


def test_train_step_at_the_reference_batch_bitwise(torch_cuda, synth, oracle, monkeypatch):
    """the reference's batch (train_frame.py:188-204: 100 utterances x 150 frames = 7 groups of 16, the last one part-filled,
    150 steps of the two-track backward): loss, every gradient and every parameter over THREE steps bit for bit between the
    weights-stationary backward kernel and the row-split one (FPC_TRAIN_BWD_ROWSPLIT=1), and the first step bit for bit
    against the CPU oracle (one host thread, running beside the GPU work)"""
    import threading
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    feat = synth.predictor_features(100, 150, utt0=6000)
    ref = oracle.Trainer(synth.predictor_state_dict(), lr=1e-3)
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("loss", ref.step(feat)))
    th.start()

    def run(steps):
        m = Wavernn(20, 384, 128, 18)
        m.load_state_dict(synth.predictor_state_dict())
        tr = Trainer(m, lr=1e-3, max_batch=100, max_frames=150)
        losses, first = [], None
        for k in range(steps):
            losses.append(tr.step(feat))
            if k == 0:
                g0 = tr.gradients()
                tr.sync()
                first = [g0[n] for n in sorted(g0)] + [v.numpy().copy() for _, v in sorted(m.state_dict().items())]
        g = tr.gradients()
        tr.sync()
        sd = m.state_dict()
        return [np.float32(losses)] + [g[n] for n in sorted(g)] + [sd[n].numpy() for n in sorted(sd)], first

    monkeypatch.setenv("FPC_TRAIN_BWD_ROWSPLIT", "1")
    rs, _ = run(3)
    monkeypatch.delenv("FPC_TRAIN_BWD_ROWSPLIT")
    ws, first = run(3)
    for k, (a, b) in enumerate(zip(rs, ws)):
        assert not np.isnan(b).any() and np.array_equal(a, b), k
    th.join()
    assert np.float32(box["loss"]) == ws[0][0]
    names_g, names_p = sorted(ref.g), sorted(ref.p)
    for k, name in enumerate(names_g):
        assert np.array_equal(first[k], ref.g[name]), name
    for k, name in enumerate(names_p):
        assert np.array_equal(first[len(names_g) + k], ref.p[name]), name


def test_train_converges_at_reference_batch(torch_cuda, synth):
    """train_frame.py:188-192 shapes (batch 100 x 150 frames): the loss falls over a few steps"""
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    m = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    m.load_state_dict(synth.predictor_state_dict())
    feat = synth.predictor_features(100, 150, utt0=6000)
    tr = Trainer(m)
    losses = [tr.step(feat) for _ in range(6)]
    from fpcodec_amd._lib import FpcError
    m.load_state_dict(synth.predictor_state_dict())  # drops the device handle the trainer was built on
    with pytest.raises(FpcError, match="reloaded"):
        tr.step(feat)
    assert all(np.isfinite(losses)) and all(b < a for a, b in zip(losses, losses[1:])), losses  # lr 1e-4: ~0.3 % a step


def test_config5_one_gpu_share_full_size(torch_cuda, model, vocoder, synth, oracle, cb_paths):
    """BASELINE config 5 at one GPU's full share (128 DISTINCT utterances x 300 frames, all in one launch, more
    workgroups than a single wave of the encoder fits per XCD): every utterance's reconstructed features and symbols
    bit for bit against the oracle (host threads), the codebook-usage histograms summed over all 128 against the
    oracle's, the keep-rates of the synthetic material near the reference's target of 0.3 (train_frame.py:204), and
    three utterances' PCM bit for bit."""
    import concurrent.futures as cf
    torch = torch_cuda
    from fpcodec_amd.synthesis_qtz import encode_features
    voc, w = vocoder
    B, L = 128, 300
    f20 = synth.predictor_features(B, L, utt0=5000)
    assert len({f20[b].tobytes() for b in range(B)}) == B  # all distinct
    nm = np.zeros((B, L, 36), np.float32)
    nm[:, :, :20] = f20
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"], l1=0.09, l2=0.28, qtz=True)
    nm_d = torch.from_numpy(nm).cuda()
    feats, r, ind1, ind2, cb_tot = encode_features(model, cfg, nm_d)
    enc = model.encoder(cfg, nm_d[:, :, :20], None, 0.09, 0.28, qtz=True, return_indices=True)
    sd = synth.seeds(B, utt0=5000)
    pcm = voc.synthesize(feats, sd)
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    pred = oracle.Predictor(synth.predictor_state_dict())
    with cf.ThreadPoolExecutor(16) as ex:  # the oracle on host threads, 8 utterances per call
        parts = list(ex.map(lambda k: pred.encode(f20[8 * k:8 * k + 8], CB, 0.09, 0.28, True), range(B // 8)))
    cin = np.concatenate([o["c_in"] for o in parts]) * np.float32(synth.MAXI)
    idx_o = np.concatenate([o["idx"] for o in parts])
    lpc = oracle.ceps2lpc(cin.reshape(-1, 20))[0].reshape(B, L, 16)
    f36 = np.concatenate([cin, lpc], -1)
    assert np.array_equal(feats.cpu().numpy(), f36)
    assert np.array_equal(enc[7].cpu().numpy(), idx_o)
    hs = CB.split_hist(sum(o["hist"] for o in parts))
    for i in range(5):
        assert np.array_equal(np.atleast_1d(cb_tot[i]), hs[i])
    keep = [float(ind1.sum()) / (B * L), float(ind2.sum()) / (B * L)]
    assert 0.25 <= keep[0] <= 0.35 and 0.25 <= keep[1] <= 0.35, keep
    orc = oracle.LPCNet(w)
    pcm_h = pcm.cpu().numpy()
    for b in (0, 77, 127):
        assert np.array_equal(pcm_h[b], orc.synthesize(f36[b], int(sd[b])))


def test_bench_rank_path_two_ranks_one_gpu(torch_cuda, synth, tmp_path):
    """bench.py's own multi-rank path, started the way the driver starts it (torch.distributed.run, one process per
    rank, here 2 ranks on the one GPU with the gloo rehearsal backend): the JSON line reports both ranks, the sample
    count is the sum of the shards, and each rank's PCM block equals an unsharded decode of the same utterances.
    (The RCCL branch itself needs 2 GPUs and is only executed by the driver's scaling run.)"""
    import json
    import socket
    import subprocess
    import sys
    torch = torch_cuda
    from fpcodec_amd.ceps2lpc import ceps2lpc_v
    from fpcodec_amd.lpcnet import LPCNet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dump = str(tmp_path / "pcm")
    env = dict(os.environ, FPC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--streams", "8", "--secs", "1",
                        "--no-cpu-baseline", "--dump-pcm", dump],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    S, T = 8, 100
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["streams_per_gpu"] == S
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 2 * S * (T * 160 - 17)) < 1.0  # samples of both ranks
    assert j["e2e"]["ranks"] == 2 and j["e2e"]["utterances"] == 256 and 5.0 < j["e2e"]["bits_per_frame"] < 30.0
    # unsharded decode of the same global utterance list
    nuniq = min(S, 16)
    base = synth.vocoder_features_raw(nuniq, T, utt0=0)
    feats = torch.from_numpy(np.stack([base[k % nuniq] for k in range(2 * S)])).cuda()
    feats[:, :, 20:] = ceps2lpc_v(feats.reshape(-1, 36)[:, :20].contiguous())[1].reshape(2 * S, T, 16)
    full = LPCNet(synth.lpcnet_weights()).synthesize(feats, synth.seeds(2 * S)).cpu().numpy()
    for rank in range(2):
        z = np.load(os.path.join(dump, f"rank{rank}.npz"))
        assert (int(z["lo"]), int(z["hi"])) == (rank * S, (rank + 1) * S)
        assert np.array_equal(z["pcm"], full[rank * S:(rank + 1) * S])


def test_bench_rccl_branch_single_rank(torch_cuda, tmp_path):
    """the RCCL (`nccl`) branch of bench.py -- process group on the device, barrier, all_reduce of the report,
    all_gather_object of the rank records -- has no multi-GPU box to run on during the round, so it is executed here with
    ONE rank through the launcher the driver uses (FPC_BENCH_FORCE_DIST=1): the same calls on device tensors, world size 1"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, FPC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FPC_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "1", "--steps", "1", "--warmup", "1", "--streams", "8", "--secs", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["distinct_gpus"] == 1 and len(j["ranks"]) == 1 and j["ranks"][0]["pci_bus_id"] != "unknown"
    assert j["e2e"]["ranks"] == 1 and j["e2e"]["utterances"] == 128
    pr = j["e2e"]["predictor_roofline"]  # the predictor kernels against the f32 MFMA peak
    assert pr["bound"] == "mfma" and pr["kernel"] == "k_encode_wsd" and 0.0 < pr["frac"] < 1.0 and 0.0 < pr["forward_frac"] < 1.0
    assert abs(pr["achieved"] - pr["algorithmic_flop"] / (pr["kernel_ms"] * 1e-3) / 1e12) < 1e-6


def test_train_cb_stage_loop_vs_reference_golden(torch_cuda, synth, golden, tmp_path):
    """G10: the batch loop of src/train_cb.py:160-217 re-enacted with the reference's own cb_func (first batch
    vq_train per stage, second batch 10 x update per stage, stage sizes 12 and 6) -- the GPU driver reproduces both
    stages and the final residual bit for bit, and writes a file the repo's vq_quantize loads."""
    torch = torch_cuda
    from fpcodec_amd import train_cb, vq_func
    g = golden("g10_train_cb_loop")
    n_entries = [12, 6]
    codebook = [np.zeros((n, 17)) for n in n_entries]
    np.random.seed(31)
    errs = []
    for batch_idx in range(2):
        r = synth.cb_training_vectors(1500, seed_offset=50 + batch_idx)
        r[::5] = 0.0
        rows = torch.from_numpy(r).cuda()
        rows = rows[rows.abs().sum(1) != 0].contiguous()  # the device-side compaction of train_cb.harvest
        codebook, res = train_cb.train_stages(codebook, rows, n_entries, batch_idx == 0)
        errs.append(float(np.sum(res * res)))
    assert np.array_equal(codebook[0], g["stage0"]) and np.array_equal(codebook[1], g["stage1"])
    assert np.array_equal(res, g["r_last"]) and np.array_equal(np.array(errs), g["errs"])
    path = str(tmp_path / "codebooks" / "ceps_vq_codebook_t.npy")
    train_cb.save_codebook(path, codebook)  # unequal stage sizes -> object array (vq_func.py:141 allow_pickle)
    x = synth.cb_training_vectors(64, seed_offset=77)
    qr, hist = vq_func.vq_quantize(x, path)
    assert qr.shape == (64, 17) and [len(h) for h in hist] == n_entries and sum(h.sum() for h in hist) == 128
    same = [np.zeros((4, 17)), np.ones((4, 17))]
    train_cb.save_codebook(str(tmp_path / "same.npy"), same)
    assert np.load(str(tmp_path / "same.npy")).shape == (2, 4, 17)


def test_train_cb_harvest_vs_oracle(torch_cuda, model, synth, oracle, cb_paths):
    """residual harvest of train_cb.py:165-186 (encoder qtz=False, zero rows dropped, scalar residuals != 0), kept on
    the device: equal to the oracle encoder's qtz=False outputs compacted on the host the way the reference does"""
    torch = torch_cuda
    from fpcodec_amd import train_cb
    B, L = 5, 60
    feat = synth.predictor_features(B, L, utt0=900)
    cfg = dict(scl_cb_path="", cb_path="", bl_scl_cb_path="", bl_cb_path="")
    o = oracle.Predictor(synth.predictor_state_dict()).encode(feat, None, 0.09, 0.28, qtz=False)
    for bl in (False, True):
        rows, scl, scl_bl = train_cb.harvest(model, cfg, torch.from_numpy(feat).cuda(), 0.09, 0.28, train_bl=bl)
        src = (o["r_under"] if bl else o["r"])[:, :, -17:].reshape(-1, 17)
        want = np.array([src[i] for i in range(len(src)) if sum(abs(src[i])) != 0])
        assert rows.is_cuda and np.array_equal(rows.cpu().numpy(), want)
        assert 0 < len(want) < B * L
        assert np.array_equal(scl.cpu().numpy(), np.array([k for k in o["r"][:, :, 0].flatten() if k != 0]))
        assert np.array_equal(scl_bl.cpu().numpy(), np.array([k for k in o["r_under"][:, :, 0].flatten() if k != 0]))


def test_predictor_row_split_equals_single_workgroup_form(torch_cuda, model, synth, cb_paths, monkeypatch):
    """the predictor on 2, 4 or 8 workgroups per utterance (row split, slices of the new states exchanged as tagged
    granules; taken automatically while the batch leaves CUs idle) and on one workgroup per utterance give the same
    bits: forward incl. carried states, and the closed-loop encoder incl. symbols and histograms -- at 1, 7 and 128
    utterances (up to 512 co-resident workgroups) and over 300 frames (600 exchanges per group)"""
    torch = torch_cuda
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    for B, L, modes in ((1, 300, ("0", "2", "4", "8")), (7, 50, ("0", "2", "8")), (128, 60, ("0", "2", "4"))):
        feat = torch.from_numpy(np.tile(synth.predictor_features(min(B, 8), L, utt0=1200), (B // 8 + 1, 1, 1))[:B].copy()).cuda()
        out = {}
        for mode in modes:
            monkeypatch.setenv("FPC_PRED_SPLIT", mode)
            y, h1, h2 = model.forward(feat)
            y2, h1b, h2b = model.forward(feat[:, :5], h1, h2)  # carried states
            enc = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
            torch.cuda.synchronize()
            out[mode] = [t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)] + [t.cpu().numpy() for t in enc[:6]] + \
                list(enc[6]) + [enc[7].cpu().numpy()]
        for mode in modes[1:]:
            assert not any(np.isnan(a).any() for a in out[mode] if a.dtype.kind == "f")  # no spin gave up
            for a, b in zip(out["0"], out[mode]):
                assert np.array_equal(a, b), (B, L, mode)


def test_row_split_forms_equal_oracle_and_each_other(torch_cuda, model, synth, oracle, cb_paths, monkeypatch):
    """the generic-shape family (csrc/predictor_df.h: three waves walk the frame's latency chain, the others stream the next
    frame's recurrent products, LDS counters instead of workgroup barriers) on 1, 2, 4 and 8 workgroups per utterance:
    forward incl. carried states, encoder with and without quantisation incl. symbols and histograms, receiver -- against
    the CPU oracle at 7 utterances and against the one-workgroup form bit for bit at 1 / 7 / 128 / 200 utterances
    (1 workgroup: the input product streams from L2; 2-8: from its LDS copy)"""
    torch = torch_cuda
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])

    def run(feat):
        y, h1, h2 = model.forward(feat)
        y2, h1b, h2b = model.forward(feat[:, :5], h1, h2)
        enc = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
        enc2 = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)
        dec = model.decode_indices(cfg, enc[7], feat[:, :, 18:].contiguous())
        torch.cuda.synchronize()
        return ([t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)] + [t.cpu().numpy() for t in enc[:6]] + list(enc[6]) +
                [enc[7].cpu().numpy()] + [t.cpu().numpy() for t in enc2[:6]] + [dec.cpu().numpy()])

    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    P = oracle.Predictor(synth.predictor_state_dict())
    for B, L in ((1, 30), (7, 40), (128, 60), (200, 20)):
        f = synth.predictor_features(B, L, utt0=7000)
        feat = torch.from_numpy(f).cuda()
        monkeypatch.setenv("FPC_PRED_SPLIT", "0")
        ref = run(feat)
        if B == 7:
            y0, a0, b0 = P.forward(f)
            o = P.encode(f, CB, 0.09, 0.28, True)
            assert np.array_equal(ref[0], y0) and np.array_equal(ref[1].reshape(a0.shape), a0)
            assert np.array_equal(ref[6], o["c_in"]) and np.array_equal(ref[8], o["r_qtz"]) and np.array_equal(ref[17], o["idx"])
            assert np.array_equal(ref[-1], o["c_in"])  # the receiver closes the loop
        for n in ("2", "4", "8"):
            if B * int(n) > 512:
                continue
            monkeypatch.setenv("FPC_PRED_SPLIT", n)
            got = run(feat)
            assert len(got) == len(ref)
            for k, (a, b) in enumerate(zip(ref, got)):
                assert np.array_equal(a, b), (B, L, n, k)


def test_predictor_row_split_under_uneven_load(torch_cuda, model, vocoder, synth, oracle, cb_paths):
    """the row-split exchange with the chip busy and the workgroups of a group NOT starting together: a 256-workgroup
    vocoder launch (one per CU, too much LDS to share a CU) is queued on a side stream right before the split encoder,
    so the encoder's workgroups trickle onto CUs as the decoder's finish and spin for partners that are dispatched
    later; repeated with the launch order swapped.  Every output must equal the quiet run's (and no spin may give up)."""
    torch = torch_cuda
    voc, w = vocoder
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    B, L = 128, 40
    feat = torch.from_numpy(np.tile(synth.predictor_features(8, L, utt0=1700), (B // 8, 1, 1)).copy()).cuda()
    quiet = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    torch.cuda.synchronize()
    vf = torch.from_numpy(np.tile(_voc_features(synth, oracle, 4, 10), (64, 1, 1)).copy()).cuda()
    sd = torch.from_numpy(synth.seeds(256).astype(np.int64)).cuda()
    pcm = torch.empty(256, 1600, dtype=torch.int16, device="cuda")
    side = torch.cuda.Stream()
    for order in (0, 1):
        if order == 0:
            with torch.cuda.stream(side):
                voc.synthesize(vf, sd, out=pcm)
            busy = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
        else:
            busy = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
            with torch.cuda.stream(side):
                voc.synthesize(vf, sd, out=pcm)
        torch.cuda.synchronize()
        for a, b in zip(quiet[:6] + (quiet[7],), busy[:6] + (busy[7],)):
            assert torch.equal(a, b), order
        for a, b in zip(quiet[6], busy[6]):
            assert np.array_equal(a, b)


def test_train_step_row_split_equals_single_workgroup_form(torch_cuda, synth, monkeypatch):
    """the training step with forward and backward on 2 / 4 / 8 workgroups per utterance (slices of the states and of the
    back-propagated vectors exchanged as granules) against one workgroup per utterance: losses, every gradient and the
    parameters after two steps, bit for bit -- with the forward as the two-role row-split kernel and (mode "ws", the default)
    on the weights-stationary kernels"""
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    feat = synth.predictor_features(6, 40, utt0=4200)
    out = {}
    for mode in ("0", "2", "4", "8", "ws"):
        if mode == "ws":  # the shipped default: the weights-stationary kernels (one part-filled group of 16)
            monkeypatch.delenv("FPC_PRED_SPLIT", raising=False)
        else:
            monkeypatch.setenv("FPC_PRED_SPLIT", mode)
        m = Wavernn(20, 384, 128, 18)
        m.load_state_dict(synth.predictor_state_dict())
        tr = Trainer(m, lr=1e-3, max_batch=6, max_frames=40)
        l0 = tr.step(feat)
        g = tr.gradients()
        l1 = tr.step(feat)
        sd = m.state_dict()
        out[mode] = ([np.float32(l0), np.float32(l1)] + [g[k] for k in sorted(g)] + [sd[k].numpy() for k in sorted(sd)])
    for mode in ("2", "4", "8", "ws"):
        for a, b in zip(out["0"], out[mode]):
            assert not np.isnan(b).any() and np.array_equal(a, b), mode


@pytest.mark.parametrize("hop", ["fast", "general"])
def test_row_split_timeout_is_an_error_not_garbage(torch_cuda, synth, cb_paths, monkeypatch, hop):
    """(both exchange paths of the row-split kernels: plain stores on one XCD and, FPC_FAST_HOP=0, write-through)
    a row-split exchange that gives up must fail loudly at the ABI (round-2 review item 3): with the test hooks
    (FPC_TEST_WITHHOLD_PUBLISH: the last slice of utterance 0 never publishes; FPC_SPIN_LIMIT_US: 20 ms instead of 1 s)
    the partner's spin times out ONCE, deterministically.  The asynchronous entry point itself returns FPC_OK; the
    handle's sticky status word turns the next call on the handle and fpc_predictor_status into FPC_ERR_TIMEOUT; the
    failing launch's float outputs are NaN and its symbols -2 (never finite-but-wrong); clearing restores the handle."""
    import ctypes as C
    from fpcodec_amd import _lib
    from fpcodec_amd._lib import FpcError
    from fpcodec_amd.wavernn import Wavernn
    from fpcodec_amd.vq_func import load_codebooks
    torch = torch_cuda
    m = Wavernn(20, 384, 128, 18)
    m.load_state_dict(synth.predictor_state_dict())
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    B, L = 2, 30
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=5100)).cuda()
    monkeypatch.setenv("FPC_PRED_SPLIT", "2")
    if hop == "general":  # the write-through exchange
        monkeypatch.setenv("FPC_FAST_HOP", "0")
    good = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    yg, _, _ = m.forward(feat)
    torch.cuda.synchronize()
    # --- the raw ABI: the failing launch returns FPC_OK, its outputs are poison, the status call reports and clears ---
    L_ = _lib.lib()
    h = m._handle()
    cb = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg["bl_cb_path"], cfg["bl_scl_cb_path"])
    bufs = [torch.zeros(B, L, n, device="cuda") for n in (20, 18, 18, 18, 1, 1)]
    idx = torch.zeros(B, L, 4, device="cuda", dtype=torch.int32)
    hist = torch.zeros(cb.hist_size, device="cuda", dtype=torch.int64)
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", "1")
    monkeypatch.setenv("FPC_SPIN_LIMIT_US", "20000")
    rc = L_.fpc_encode(h, cb.handle, feat.data_ptr(), B, L, 0.09, 0.28, 1, *[b.data_ptr() for b in bufs], idx.data_ptr(),
                       hist.data_ptr(), None, _lib.stream_ptr())
    assert rc == 0  # asynchronous: the launch itself is accepted
    torch.cuda.synchronize()
    monkeypatch.delenv("FPC_TEST_WITHHOLD_PUBLISH")
    # utterance 0 gave up in its first frame: NaN floats, symbols -2, never finite-but-wrong values
    assert torch.isnan(bufs[0][0]).all() and torch.isnan(bufs[1][0]).all() and (idx[0] == -2).all()
    # the next call on the handle is refused with the timeout code, and keeps being refused until the status is read
    y = torch.empty(B, L, 18, device="cuda")
    s1, s2 = torch.zeros(B, 384, device="cuda"), torch.zeros(B, 128, device="cuda")
    for _ in range(2):
        rc = L_.fpc_predictor_forward(h, feat.data_ptr(), B, L, s1.data_ptr(), s2.data_ptr(), y.data_ptr(), _lib.stream_ptr())
        assert rc == -5 and b"timed out" in L_.fpc_last_error()
    assert L_.fpc_predictor_status(h) == -5
    assert L_.fpc_predictor_status(h) == 0  # cleared
    # --- the Python surface raises; afterwards the handle works again, bit for bit ---
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", "1")
    with pytest.raises(FpcError, match="timed out"):
        m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    yb, _, _ = m.forward(feat)  # (status already cleared by the raise: this launch is accepted and fails again)
    with pytest.raises(FpcError, match="timed out"):
        m.check()
    assert torch.isnan(yb[0]).all()
    monkeypatch.delenv("FPC_TEST_WITHHOLD_PUBLISH")
    monkeypatch.delenv("FPC_SPIN_LIMIT_US")
    again = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    ya, _, _ = m.forward(feat)
    m.check()
    for a, b in zip(good[:6] + (good[7], yg), again[:6] + (again[7], ya)):
        assert torch.equal(a, b)
    for a, b in zip(good[6], again[6]):
        assert np.array_equal(a, b)
    # --- a pinned split of 1 (shared-GPU deployments) never exchanges: the hook has nothing to withhold ---
    monkeypatch.delenv("FPC_PRED_SPLIT")
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", "1")
    m.set_split(1)
    one = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    m.set_split(0)
    monkeypatch.delenv("FPC_TEST_WITHHOLD_PUBLISH")
    assert torch.equal(one[0], good[0]) and torch.equal(one[7], good[7])


def test_nonfinite_residual_is_refused_not_searched(torch_cuda, synth, cb_paths):
    """a NaN in the features has no nearest codebook entry: the frame is not searched (the arg-min of NaN distances
    would be used as an address), its symbols are -2 and the call fails with the non-finite code;
    the stand-alone quantizers answer NaN / -2 for such a row"""
    from fpcodec_amd._lib import FpcError
    from fpcodec_amd.wavernn import Wavernn
    from fpcodec_amd.vq_func import vq_quantize, scl_quantize
    torch = torch_cuda
    m = Wavernn(20, 384, 128, 18)
    m.load_state_dict(synth.predictor_state_dict())
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    feat = torch.from_numpy(synth.predictor_features(2, 12, utt0=5200)).cuda()
    feat[1, 5, 3] = float("nan")
    with pytest.raises(FpcError, match="non-finite"):
        m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    clean = torch.from_numpy(synth.predictor_features(2, 12, utt0=5200)).cuda()
    out = m.encoder(cfg, clean, None, 0.09, 0.28, qtz=True, return_indices=True)  # the handle is usable again
    assert not torch.isnan(out[0]).any()
    r = synth.predictor_features(1, 4, utt0=5300)[0, :, :17].copy()
    r[2, 7] = np.inf
    qr, _, ix = vq_quantize(r, cb_paths["vq_hi"], return_indices=True)
    assert np.isnan(qr[2]).all() and (np.asarray(ix)[2] == -2).all() and np.isfinite(qr[[0, 1, 3]]).all()
    q, _, ix = scl_quantize(np.array([[0.1], [np.nan]], np.float32), cb_paths["scl_hi"], return_indices=True)
    assert np.isnan(q[1, 0]) and np.asarray(ix).reshape(-1)[1] == -2 and np.isfinite(q[0, 0])


def test_one_handle_on_two_streams(torch_cuda, model, synth, cb_paths, monkeypatch):
    """launches of ONE predictor handle issued back to back on two streams (they share the handle's granule block):
    the second waits on the device for the first; both equal the single-stream results"""
    torch = torch_cuda
    monkeypatch.setenv("FPC_PRED_SPLIT", "2")
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    fa = torch.from_numpy(synth.predictor_features(16, 120, utt0=5400)).cuda()
    fb = torch.from_numpy(synth.predictor_features(16, 120, utt0=5500)).cuda()
    ya, _, _ = model.forward(fa)
    yb, _, _ = model.forward(fb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            y1, _, _ = model.forward(fa)
        with torch.cuda.stream(s2):
            y2, _, _ = model.forward(fb)
        torch.cuda.synchronize()
        model.check()
        assert torch.equal(y1, ya) and torch.equal(y2, yb)


def test_injected_quantizers_are_honoured(torch_cuda, model, synth, cb_paths):
    """`Wavernn.encoder` calls whatever quantizers it is handed (reference wavernn.py:165,219-240): callables that are
    not this package's own run frame by frame on the rows the reference hands them.  A pass-through wrapper of the
    built-in quantizers must reproduce the fused kernel bit for bit; a quantizer of its own (everything -> 0) must
    show up in r_qtz, c_in and cb_tot; asking for codebook symbols from a foreign quantizer is refused."""
    from fpcodec_amd import vq_func
    from fpcodec_amd._lib import FpcError
    torch = torch_cuda
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    feat = torch.from_numpy(synth.predictor_features(3, 25, utt0=5600)).cuda()
    fused = model.encoder(cfg, feat, None, 0.09, 0.28, vq_func.vq_quantize, vq_func.scl_quantize, qtz=True)
    calls = {"vq": 0, "scl": 0}

    def vq_wrap(r, path):
        calls["vq"] += 1
        assert r.shape == (1, 17) and r.dtype == np.float32
        return vq_func.vq_quantize(r, path)

    def scl_wrap(d, path):
        calls["scl"] += 1
        assert d.shape == (1, 1)
        return vq_func.scl_quantize(d, path)

    slow = model.encoder(cfg, feat, None, 0.09, 0.28, vq_wrap, scl_wrap, qtz=True)
    assert calls["vq"] == 3 * 25 and calls["scl"] == 3 * 25  # both codebook pairs are set: every row is quantized
    for a, b in zip(fused[:6], slow[:6]):
        assert torch.equal(a, b)
    for a, b in zip(fused[6], slow[6]):
        assert np.array_equal(np.asarray(a), np.asarray(b))

    def vq_zero(r, path):
        return np.zeros((1, 17)), [np.ones(3), np.ones(3)]

    def scl_zero(d, path):
        return np.zeros((1, 1)), np.ones(2)

    z = model.encoder(cfg, feat, None, 0.09, 0.28, vq_zero, scl_zero, qtz=True)
    assert float(z[2].abs().max()) == 0.0 and not torch.equal(z[0], fused[0])
    n1, n2 = int(z[4].sum()), int(z[5].sum())
    assert z[6][0].sum() == 2 * n1 and z[6][1].sum() == 2 * (75 - n1) and z[6][2].sum() == 3 * n2 and z[6][4].sum() == 3 * (75 - n2)
    with pytest.raises(FpcError, match="return_indices"):
        model.encoder(cfg, feat, None, 0.09, 0.28, vq_zero, scl_zero, qtz=True, return_indices=True)


@pytest.mark.parametrize("pairing", [0, 1])
def test_vocoder_stress_parity_randomised(torch_cuda, synth, oracle, pairing):
    """(pairing 0: one utterance per workgroup, k_decode; 1: two per workgroup, k_decode2 -- the third weight set selects
    k_decode2<2>, the first two k_decode2<4>)
    the vocoder guard that used to be a tool (tools/stress_parity.py), now in the driver-run suite: three weight sets
    with other seeds and densities than the benchmark's (they select different decode-kernel instances and
    placements), pitch correlations drawn over the whole range (a wide mix of voiced and unvoiced frames), random
    64-bit seeds: 3 x 16 utterances x 40 frames = 306 k samples, every one equal to the oracle's"""
    import concurrent.futures as cf
    from fpcodec_amd.lpcnet import LPCNet
    total = 0
    for wseed, dens in ((1004, (0.05, 0.05, 0.2)), (77, (0.03, 0.06, 0.18)), (5, (0.02, 0.02, 0.1))):
        w = synth.lpcnet_weights(seed=wseed, density=dens)
        voc, orc = LPCNet(w), oracle.LPCNet(w)
        voc.set_pairing(pairing)
        B, T = 16, 40
        f = synth.vocoder_features_raw(B, T, utt0=wseed * 10)
        f[:, :, 19] = np.random.default_rng(wseed).uniform(-0.5, 1.0, (B, T))
        f[:, :, 20:] = oracle.ceps2lpc(f.reshape(-1, 36)[:, :20])[0].reshape(B, T, 16)
        sd = np.random.default_rng(wseed + 1).integers(0, 2 ** 62, B).astype(np.uint64)
        pcm = voc.synthesize(f, sd).cpu().numpy()
        assert voc.last_streams_per_workgroup() == 1 + pairing
        with cf.ThreadPoolExecutor(16) as ex:
            refs = list(ex.map(lambda b: orc.synthesize(f[b], int(sd[b])), range(B)))
        for b in range(B):
            nz = np.nonzero(pcm[b] != refs[b])[0]
            assert nz.size == 0, f"weights {wseed} {dens} utt {b}: first mismatch at sample {nz[:5]}"
        # the chunked pass (state carried between launches, the resumed sparse product) on every kernel instance
        voc.set_chunk_frames(7 + wseed % 5)
        assert np.array_equal(voc.synthesize(f, sd).cpu().numpy(), pcm), f"weights {wseed}: chunked pass differs"
        total += B * (T * 160 - 17)
    assert total > 300000


def test_vocoder_long_utterance_parity(torch_cuda, vocoder, synth, oracle):
    """one 15-second utterance (1 500 frames, 240 k samples; tools/long_parity.py): no drift between the kernel and the
    oracle over five times the benchmark's length (history ring, frame switches, de-emphasis state, 32-bit sample
    counters of the RNG)"""
    voc, w = vocoder
    T = 1500
    f = _voc_features(synth, oracle, 1, T, utt0=7)
    sd = synth.seeds(1, utt0=7)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    ref = oracle.LPCNet(w).synthesize(f[0], int(sd[0]))
    nz = np.nonzero(pcm[0] != ref)[0]
    assert nz.size == 0, f"first mismatch at sample {nz[:5]}"
    assert int(np.abs(pcm[0].astype(np.int32)).max()) > 100  # a live signal, not silence


@pytest.mark.parametrize("pairing", [0, 1])
def test_vocoder_chunked_pass_same_samples_bounded_workspace(torch_cuda, synth, oracle, pairing):
    """(pairing 1: the same on k_decode2 -- three utterances = one pair and a workgroup that decodes its one utterance twice)
    fpc_lpcnet_set_chunk_frames: the frame-rate layers, the conditioning products and the sample loop chunk by chunk
    with the per-stream state carried in a record -- the same samples bit for bit as the one-pass form and as the oracle
    (chunks of 1, 5 and 7 frames over 23: ragged last chunk, halo frames at both ends of the utterance, a voiced and an
    unvoiced stream), the conditioning vectors too, and a workspace that does not grow with T"""
    from fpcodec_amd.lpcnet import LPCNet
    w = synth.lpcnet_weights()
    voc = LPCNet(w)
    voc.set_pairing(pairing)
    B, T = 3, 23
    f = _voc_features(synth, oracle, B, T, utt0=3)
    f[1, :, 19] = 0.9   # a voiced stream: the sharpened-pdf path through every chunk
    sd = synth.seeds(B, utt0=3)
    whole = voc.synthesize(f, sd).cpu().numpy()
    cf_whole = voc.condition(f).cpu().numpy()
    ref = oracle.LPCNet(w).synthesize(f[0], int(sd[0]))
    assert np.array_equal(whole[0], ref)
    for chunk in (1, 5, 7, 23, 50):
        voc.set_chunk_frames(chunk)
        got = voc.synthesize(f, sd).cpu().numpy()
        nz = np.argwhere(got != whole)
        assert nz.size == 0, f"chunk {chunk}: first mismatch at {nz[:3].tolist()}"
        assert np.array_equal(voc.condition(f).cpu().numpy(), cf_whole), chunk
    voc.set_chunk_frames(50)
    per_frame = (128 * 3 + 1152 + 48) * 4
    assert voc.workspace_bytes(256, 300) == voc.workspace_bytes(256, 30000) == 256 * (50 * per_frame + 768 * 4) + 256
    assert voc.workspace_bytes(256, 40) == 256 * 40 * per_frame + 256      # shorter than a chunk: one pass
    voc.set_chunk_frames(0)
    assert voc.workspace_bytes(256, 300) == 256 * 300 * per_frame + 256
    # five times the benchmark's length in chunks of 50 (30 launches of the sample loop) against the one-pass form
    T = 1500
    f = _voc_features(synth, oracle, 2, T, utt0=7)
    sd = synth.seeds(2, utt0=7)
    one = voc.synthesize(f, sd).cpu().numpy()
    voc.set_chunk_frames(50)
    assert np.array_equal(voc.synthesize(f, sd).cpu().numpy(), one)


# ---- the weights-stationary predictor kernels (csrc/predictor_ws.h): the shipped form for the production shape ----
def _ws_cfgs(cb_paths, tmp):
    """codebook configurations: the reference's four books, only the above-threshold ones, a 1-stage book, ragged stages, and a
    degenerate first stage (300 copies of one entry: far more than 64 candidates at the bound, ties decided by index)"""
    full = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
                bl_cb_path=cb_paths["vq_lo"])
    hi = dict(full, bl_scl_cb_path="", bl_cb_path="")
    one = dict(full, cb_path=cb_paths["vq_lo"], bl_cb_path="")
    rag = dict(full, cb_path=cb_paths["ragged"])
    c = np.load(cb_paths["vq_hi"]).copy()
    c[0, 100:400] = c[0, 7]
    c[1, 600:700] = c[1, 3]
    p = os.path.join(tmp, "degenerate.npy")
    np.save(p, c)
    deg = dict(full, cb_path=p)
    # near ties: both stages are clouds of width 1e-7 around a few centres, so the best distances agree in their high 32 bits
    # (sign, exponent, 20 mantissa bits) but not in the low ones -- the selections' high-word fast paths must hand over to the
    # float64 passes exactly then (predictor_wsd.h: hw_argmin, the first stage's rank)
    rng = np.random.default_rng(99)
    n = np.load(cb_paths["vq_hi"]).copy()
    for st in range(n.shape[0]):
        centres = n[st, :8].copy()
        n[st] = centres[rng.integers(0, 8, n.shape[1])] + 1e-7 * rng.standard_normal(n[st].shape)
    pn = os.path.join(tmp, "near_ties.npy")
    np.save(pn, n)
    near = dict(full, cb_path=pn)
    return {"full": full, "hi_only": hi, "one_stage": one, "ragged": rag, "degenerate": deg, "near_ties": near}


def test_weights_stationary_kernels_equal_row_split_forms(torch_cuda, model, synth, oracle, cb_paths, monkeypatch, tmp_path):
    """the shipped predictor kernels for the production shape (groups of 16 utterances on the 32 workgroups of an XCD, weights
    resident in LDS, gate rows on f32 MFMA, 16-byte granule hops, the encoder's tail distributed over the group) against the
    generic-shape family (FPC_PRED_WS=0: the two-role row-split kernels, one workgroup per utterance) and the oracle: forward incl.
    carried states, encoder with and without quantisation incl. symbols and histograms, receiver -- bit for bit, for 1 / 7 /
    16 / 33 / 128 / 200 utterances (partly filled groups, two rounds of groups), on BOTH hop paths (FPC_FAST_HOP=0: write-through
    stores, the path a placement on several XCDs takes) and for six codebook configurations (the last two live on the
    selections' float64 fallbacks: exact ties, and distances that agree in their high words only)"""
    torch = torch_cuda
    cfgs = _ws_cfgs(cb_paths, str(tmp_path))

    def run(feat, cfg):
        y, h1, h2 = model.forward(feat)
        y2, h1b, h2b = model.forward(feat[:, :5], h1, h2)
        enc = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
        enc2 = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)
        dec = model.decode_indices(cfg, enc[7], feat[:, :, 18:].contiguous())
        torch.cuda.synchronize()
        return ([t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)] + [t.cpu().numpy() for t in enc[:6]] + list(enc[6]) +
                [enc[7].cpu().numpy()] + [t.cpu().numpy() for t in enc2[:6]] + [dec.cpu().numpy()])

    for B, L, names in ((1, 30, ("full",)), (7, 40, ("full", "hi_only", "one_stage", "ragged", "degenerate", "near_ties")), (16, 33, ("full",)),
                        (33, 25, ("full", "ragged")), (128, 60, ("full",)), (200, 20, ("full", "degenerate"))):
        feat = torch.from_numpy(synth.predictor_features(B, L, utt0=7000)).cuda()
        for name in names:
            cfg = cfgs[name]
            monkeypatch.setenv("FPC_PRED_WS", "0")
            monkeypatch.setenv("FPC_PRED_SPLIT", "0")
            ref = run(feat, cfg)
            monkeypatch.delenv("FPC_PRED_WS")
            monkeypatch.delenv("FPC_PRED_SPLIT")
            for fast in ("1", "0"):
                monkeypatch.setenv("FPC_FAST_HOP", fast)
                got = run(feat, cfg)
                assert len(got) == len(ref)
                for k, (a, b) in enumerate(zip(ref, got)):
                    assert np.array_equal(a, b), (B, L, name, fast, k)
            monkeypatch.delenv("FPC_FAST_HOP")
    # ... and the oracle itself (one partly filled group, the reference's four codebooks)
    feat = synth.predictor_features(5, 50, utt0=7100)
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    o = oracle.Predictor(synth.predictor_state_dict()).encode(feat, CB, 0.09, 0.28, True)
    enc = model.encoder(cfgs["full"], torch.from_numpy(feat).cuda(), None, 0.09, 0.28, qtz=True, return_indices=True)
    assert np.array_equal(enc[0].cpu().numpy(), o["c_in"]) and np.array_equal(enc[2].cpu().numpy(), o["r_qtz"])
    assert np.array_equal(enc[7].cpu().numpy(), o["idx"])


def test_weights_stationary_timeout_is_an_error_not_garbage(torch_cuda, synth, cb_paths, monkeypatch):
    """the give-up path of the weights-stationary kernels: the last workgroup of group 0 never publishes
    (FPC_TEST_WITHHOLD_PUBLISH), its 31 partners time out once (20 ms instead of 1 s) -- the launch is accepted, every output
    of the group is NaN / -2 (never finite-but-wrong), the handle reports FPC_ERR_TIMEOUT until the status call has cleared it,
    and the next launch is bit-identical to an undisturbed one; a second group of the same launch is untouched"""
    from fpcodec_amd import _lib
    from fpcodec_amd._lib import FpcError
    from fpcodec_amd.wavernn import Wavernn
    torch = torch_cuda
    m = Wavernn(20, 384, 128, 18)
    m.load_state_dict(synth.predictor_state_dict())
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    B, L = 20, 12  # two groups: 16 + 4 utterances
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=5300)).cuda()
    good = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    yg, h1g, _ = m.forward(feat)
    dg = m.decode_indices(cfg, good[7], feat[:, :, 18:].contiguous())
    torch.cuda.synchronize()
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", "1")
    monkeypatch.setenv("FPC_SPIN_LIMIT_US", "20000")
    with pytest.raises(FpcError, match="timed out"):
        m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    yb, h1b, _ = m.forward(feat)  # (status cleared by the raise: accepted, fails again)
    with pytest.raises(FpcError, match="timed out"):
        m.check()
    h1b, h1g = h1b.reshape(B, -1), h1g.reshape(B, -1)
    assert torch.isnan(yb[:16]).all() and torch.isnan(h1b[:16]).all()
    assert torch.equal(yb[16:], yg[16:]) and torch.equal(h1b[16:], h1g[16:])  # the other group never waited for anybody
    with pytest.raises(FpcError, match="timed out"):
        m.decode_indices(cfg, good[7], feat[:, :, 18:].contiguous())
    # the raw ABI: poison in the encoder's outputs of group 0, the symbols -2
    L_ = _lib.lib()
    from fpcodec_amd.vq_func import load_codebooks
    cb = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg["bl_cb_path"], cfg["bl_scl_cb_path"])
    bufs = [torch.zeros(B, L, n, device="cuda") for n in (20, 18, 18, 18, 1, 1)]
    idx = torch.zeros(B, L, 4, device="cuda", dtype=torch.int32)
    hist = torch.zeros(cb.hist_size, device="cuda", dtype=torch.int64)
    assert L_.fpc_predictor_status(m._handle()) in (0, -5)  # (whatever the failed decode left: cleared)
    rc = L_.fpc_encode(m._handle(), cb.handle, feat.data_ptr(), B, L, 0.09, 0.28, 1, *[b.data_ptr() for b in bufs],
                       idx.data_ptr(), hist.data_ptr(), None, _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isnan(bufs[0][:16]).all() and torch.isnan(bufs[1][:16]).all() and (idx[:16] == -2).all()
    assert torch.equal(bufs[0][16:], good[0][16:]) and torch.equal(idx[16:], good[7][16:])
    assert int(hist.sum()) == int((idx[16:] >= 0).sum())  # only the healthy group's symbols were counted
    assert L_.fpc_predictor_status(m._handle()) == -5 and L_.fpc_predictor_status(m._handle()) == 0
    monkeypatch.delenv("FPC_TEST_WITHHOLD_PUBLISH")
    monkeypatch.delenv("FPC_SPIN_LIMIT_US")
    again = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    ya, _, _ = m.forward(feat)
    da = m.decode_indices(cfg, again[7], feat[:, :, 18:].contiguous())
    m.check()
    for a, b in zip(good[:6] + (good[7], yg, dg), again[:6] + (again[7], ya, da)):
        assert torch.equal(a, b)
    for a, b in zip(good[6], again[6]):
        assert np.array_equal(a, b)


def test_ws_group_that_cannot_become_resident_falls_back_not_fails(torch_cuda, model, synth, cb_paths, monkeypatch):
    """the co-residency assumption of the weights-stationary kernels fails SAFE (fpcodec.h "Kernel forms"): with the test hook
    FPC_TEST_WITHHOLD_PUBLISH=hello the last workgroup of group 0 stands for one that never becomes resident -- its 31
    partners run out of patience (FPC_HELLO_LIMIT_US: 3 ms instead of 10), the group decides WS_FALLBACK, every workgroup of
    it returns before anything is written, and the row-split launch queued behind serves exactly that group: forward incl.
    carried states, encoder incl. symbols and histograms, receiver and two training steps are bit-identical to the quiet
    run, nothing raises, and the diagnostic counts one group (of two) -- group 1 stays on the weights-stationary kernels"""
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    torch = torch_cuda
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    B, L = 20, 30
    f = synth.predictor_features(B, L, utt0=7300)
    feat = torch.from_numpy(f).cuda()

    def run():
        y, h1, h2 = model.forward(feat)
        nfb = [model.fallback_groups()]
        y2, h1b, h2b = model.forward(feat[:, :5], h1, h2)
        enc = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
        nfb.append(model.fallback_groups())
        dec = model.decode_indices(cfg, enc[7], feat[:, :, 18:].contiguous())
        nfb.append(model.fallback_groups())
        m = Wavernn(20, 384, 128, 18)
        m.load_state_dict(synth.predictor_state_dict())
        tr = Trainer(m, lr=1e-3, max_batch=B, max_frames=L)
        losses = [tr.step(f), tr.step(f)]
        nfb.append(m.fallback_groups())
        sd = m.state_dict()
        return ([t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)] + [t.cpu().numpy() for t in enc[:6]] + list(enc[6]) +
                [enc[7].cpu().numpy(), dec.cpu().numpy(), np.float32(losses)] + [sd[k].numpy() for k in sorted(sd)]), nfb

    quiet, n0 = run()
    assert n0 == [0, 0, 0, 0]
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", "hello")
    monkeypatch.setenv("FPC_HELLO_LIMIT_US", "3000")
    for fast in ("1", "0"):
        monkeypatch.setenv("FPC_FAST_HOP", fast)
        busy, n1 = run()
        model.check()  # nothing was reported: a fallback is not a failure
        assert n1 == [1, 1, 1, 1], n1
        assert len(busy) == len(quiet)
        for k, (a, b) in enumerate(zip(quiet, busy)):
            assert np.array_equal(a, b), (fast, k)


def test_encode_beside_the_librarys_own_long_decode(torch_cuda, model, vocoder, synth, oracle, cb_paths, monkeypatch):
    """the pipeline the package itself suggests -- encode batch k + 1 while batch k is being decoded on a side stream: a
    200-stream vocoder launch holds 200 of the 256 CUs for ~35 ms, far longer than the give-up bounds (shortened to 2 ms /
    4 ms by the hooks), so an fpc_encode of 128 utterances issued beside it gets 7 of each group's 32 workgroups placed.  It
    must return the quiet run's bits -- not FPC_ERR_TIMEOUT, not NaN -- by deciding for the fallback (round-4 review item 5)"""
    import time
    torch = torch_cuda
    voc, w = vocoder
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    B, L = 128, 40
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=7400)).cuda()
    quiet = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    assert model.fallback_groups() == 0
    T = 150
    vf = torch.from_numpy(np.tile(_voc_features(synth, oracle, 4, T), (50, 1, 1)).copy()).cuda()
    sd = torch.from_numpy(synth.seeds(200).astype(np.int64)).cuda()
    pcm = torch.empty(200, T * 160, dtype=torch.int16, device="cuda")
    voc.synthesize(vf, sd, out=pcm)  # (warm: workspace allocated)
    torch.cuda.synchronize()
    monkeypatch.setenv("FPC_HELLO_LIMIT_US", "2000")
    monkeypatch.setenv("FPC_SPIN_LIMIT_US", "4000")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        voc.synthesize(vf, sd, out=pcm)
    time.sleep(0.004)  # the decode kernel is on the chip by now (its frame-rate kernels take < 1 ms)
    busy = model.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)  # (raises on any reported failure)
    nfb = model.fallback_groups()
    torch.cuda.synchronize()
    model.check()
    assert nfb > 0, "the decode did not crowd the encoder out: the test exercised nothing"
    for a, b in zip(quiet[:6] + (quiet[7],), busy[:6] + (busy[7],)):
        assert torch.equal(a, b)
    for a, b in zip(quiet[6], busy[6]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("hook", ["1", "backward"])
def test_trainer_step_that_gives_up_updates_nothing_and_the_count_follows(torch_cuda, synth, monkeypatch, hook):
    """the training step whose forward (weights-stationary kernel: one workgroup of group 0 withholds, its partners time out
    after 20 ms) gave up -- or, hook "backward", whose forward is whole and whose BACKWARD kernel alone gives up (its two
    tracks leave their loops independently: the wave whose wait timed out sets the workgroup's flag, every LDS-counter wait
    of either track reads it): the call fails with the timeout code, NO parameter moves (the Adam launches read the latched
    status word), and after the error has been reported the next steps continue with Adam's bias corrections of the updates
    actually applied -- parameters bit-identical to a trainer that never saw the failure (the step counter)"""
    from fpcodec_amd._lib import FpcError
    from fpcodec_amd.train_frame import Trainer
    from fpcodec_amd.wavernn import Wavernn
    feat = synth.predictor_features(6, 30, utt0=4300)

    def fresh():
        m = Wavernn(20, 384, 128, 18)
        m.load_state_dict(synth.predictor_state_dict())
        return m, Trainer(m, lr=1e-3, max_batch=6, max_frames=30)

    m0, t0 = fresh()
    ref_losses = [t0.step(feat) for _ in range(3)]
    t0.sync()
    ref = {k: v.numpy().copy() for k, v in m0.state_dict().items()}

    m1, t1 = fresh()
    l0 = t1.step(feat)
    t1.sync()
    before = {k: v.numpy().copy() for k, v in m1.state_dict().items()}
    monkeypatch.setenv("FPC_TEST_WITHHOLD_PUBLISH", hook)
    monkeypatch.setenv("FPC_SPIN_LIMIT_US", "20000")
    with pytest.raises(FpcError, match="timed out"):
        t1.step(feat)
    monkeypatch.delenv("FPC_TEST_WITHHOLD_PUBLISH")
    monkeypatch.delenv("FPC_SPIN_LIMIT_US")
    try:
        m1.check()  # (reports and clears whatever the failed step left)
    except FpcError:
        pass
    t1.sync()
    after = {k: v.numpy() for k, v in m1.state_dict().items()}
    for k in before:
        assert np.array_equal(before[k], after[k]), f"{k} moved in a step that gave up"
    l1, l2 = t1.step(feat), t1.step(feat)
    assert [l0, l1, l2] == ref_losses
    t1.sync()
    for k, v in m1.state_dict().items():
        assert np.array_equal(v.numpy(), ref[k]), k


def test_encoder_mask_mode_and_multi_stage_lo_vs_reference_golden(torch_cuda, model, cb_paths, tmp_path, monkeypatch):
    """golden G12, generated by tests/golden/make_golden_modes.py from the reference's own `encoder`: the input-mask mode
    (wavernn.py:209-211) with and without quantisation -- on the fused kernels through fpc_encode's mask_dev (ABI 2), in both
    kernel families (weights-stationary and, FPC_PRED_WS=0, row-split: same bits), and through the host loop that serves
    injected quantizer callables -- and a 2-stage below-threshold codebook (wavernn.py:235-240: quantize_mstage over both
    stages, cb_tot[4] += the last stage's histogram; host loop)"""
    torch = torch_cuda
    from fpcodec_amd import vq_func
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_encoder_modes.npz"))
    import fpcodec_amd
    synth = fpcodec_amd.synth
    feat = synth.predictor_features(3, 30, utt0=40)
    mask = (np.random.default_rng(1098).random((3, 30, 2, 1)) < 0.4).astype(np.float32)
    rng = np.random.default_rng(1099)
    lo2 = np.stack([rng.normal(0, 0.02, (512, 17)), rng.normal(0, 0.01, (512, 17))])
    p_lo2 = os.path.join(str(tmp_path), "vq_lo2.npy")
    np.save(p_lo2, lo2)
    cfg = dict(scl_cb_path=cb_paths["scl_hi"], cb_path=cb_paths["vq_hi"], bl_scl_cb_path=cb_paths["scl_lo"],
               bl_cb_path=cb_paths["vq_lo"])
    wrapped = dict(vq_quantize=lambda r, p: vq_func.vq_quantize(r, p), scl_quantize=lambda d, p: vq_func.scl_quantize(d, p))
    runs = {"mask_qtz": (cfg, mask, True, {}), "mask_raw": (cfg, mask, False, {}), "lo2": (dict(cfg, bl_cb_path=p_lo2), None, True, {}),
            "mask_qtz/host loop": (cfg, mask, True, wrapped), "mask_qtz/(B, L, 2)": (cfg, mask[..., 0], True, {})}
    fused = {}
    for name, (c, m, qtz, kw) in runs.items():
        tag = name.split("/")[0]
        for ws in ("1", "0"):
            if ws == "0":
                if name in ("lo2", "mask_qtz/host loop"):
                    continue
                monkeypatch.setenv("FPC_PRED_WS", "0")
            out = model.encoder(c, torch.from_numpy(feat), None if m is None else torch.from_numpy(m), 0.09, 0.28, qtz=qtz, **kw)
            monkeypatch.delenv("FPC_PRED_WS", raising=False)
            for n, v in zip(["c_in", "r", "r_qtz", "r_under"], out[:4]):
                assert np.abs(v.cpu().numpy() - g[f"{tag}_{n}"]).max() <= 1e-5, (name, ws, n)  # the north_star's tolerance
            for n, v in zip(["ind1", "ind2"], out[4:6]):
                assert np.array_equal(v.cpu().numpy().reshape(g[f"{tag}_{n}"].shape), g[f"{tag}_{n}"]), (name, ws, n)
            for i, h in enumerate(out[6]):
                assert np.array_equal(np.asarray(h, dtype=np.float64), g[f"{tag}_hist{i}"]), (name, ws, i)  # symbols: exact
            if name in ("mask_qtz", "mask_raw"):  # the two kernel families agree bit for bit
                got = [v.cpu().numpy() for v in out[:6]]
                if ws == "1":
                    fused[name] = got
                else:
                    for a, b in zip(fused[name], got):
                        assert np.array_equal(a, b), (name, "ws vs row split")
    # the mask mode leaves the indicator outputs at zero (the reference only fills them from the thresholds)
    assert not g["mask_qtz_ind1"].any() and g["lo2_ind1"].any()
    # a mask of the wrong shape is refused, not reinterpreted
    from fpcodec_amd._lib import FpcError
    with pytest.raises(FpcError, match="mask of shape"):
        model.encoder(cfg, torch.from_numpy(feat), torch.zeros(3, 30), 0.09, 0.28)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(5000, 8), (12000, 32), (3000, 64), (2049, 5), (40000, 16)])
def test_scalar_codebook_kmeans_equals_oracle_bitwise_and_sklearn(torch_cuda, n, k):
    """fpc_kmeans1d (csrc/kmeans1d.hip) = the reference's commented KMeans call (train_cb.py:219-226): seeds, iteration count,
    inertia and centres BIT-IDENTICAL to oracle/kmeans1d_oracle.py (which the CPU suite pins to scikit-learn), and against
    scikit-learn itself: the same seeds as its kmeans_plusplus, centres to 1e-9 (its sums have no specified order), the same
    result twice (sklearn's OpenMP reductions do not promise that)"""
    from sklearn.cluster import KMeans
    from fpcodec_amd import train_cb
    from oracle import kmeans1d_oracle as KO
    rs = np.random.RandomState(100 + k)
    v = (rs.laplace(size=n) * 0.1).astype(np.float32).astype(np.float64)
    d = {}
    c = train_cb.train_scalar_codebook(v, k, n_init=3, details=d)
    assert c.shape == (k, 1) and c.dtype == np.float64
    oc, oinertia, oiter, oseeds = KO.fit(v, k, n_init=3)
    assert np.array_equal(d["seeds"], oseeds)
    assert d["n_iter"] == oiter and d["inertia"] == oinertia
    assert np.array_equal(c, oc)
    km = KMeans(n_clusters=k, random_state=0, n_init=3).fit(v[:, None])
    assert np.abs(c - km.cluster_centers_).max() < 1e-9 and d["n_iter"] == km.n_iter_
    d2 = {}
    c2 = train_cb.train_scalar_codebook(v, k, n_init=3, details=d2)
    assert np.array_equal(c, c2) and d2["inertia"] == d["inertia"]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [256, 16])
def test_scalar_codebook_kmeans_production_sizes(torch_cuda, k):
    """the sizes the path uses (README.md:29: 256 scalar entries above the threshold, 16 below; train_cb.py:219-226 on the
    residuals of a corpus: n = 400 000): the seeding's three-level candidate search and the two-level M-step over 196 blocks of
    2 048 points.  Seeds, iteration count, inertia and centres bit-identical to the oracle (its Lloyd loops through their C
    twins, which tests/test_host_cpu.py holds equal to the numpy definition), and scikit-learn itself to 1e-9"""
    from sklearn.cluster import KMeans
    from fpcodec_amd import train_cb
    from oracle import kmeans1d_oracle as KO
    n = 400000
    rs = np.random.RandomState(100 + k)
    v = (rs.laplace(size=n) * 0.1).astype(np.float32).astype(np.float64)
    d = {}
    c = train_cb.train_scalar_codebook(v, k, n_init=2, details=d)
    oc, oinertia, oiter, oseeds = KO.fit(v, k, n_init=2, fast=True)
    assert np.array_equal(d["seeds"], oseeds)
    assert d["n_iter"] == oiter and d["inertia"] == oinertia
    assert np.array_equal(c, oc)
    km = KMeans(n_clusters=k, random_state=0, n_init=2).fit(v[:, None])
    assert np.abs(c - km.cluster_centers_).max() < 1e-9 and d["n_iter"] == km.n_iter_


@pytest.mark.gpu
def test_scalar_codebook_kmeans_edges(torch_cuda):
    """k = 1 (no seeding draws), heavy duplicates (fewer distinct values than a block), an empty cluster relocated (two equal
    values seeded: impossible through k-means++, which never draws a point at distance 0 -- reached through first_ids /
    uniforms that force it), refusals"""
    import ctypes as C
    from fpcodec_amd import _lib, train_cb
    from oracle import kmeans1d_oracle as KO
    rs = np.random.RandomState(7)
    v = rs.normal(size=3000)
    c = train_cb.train_scalar_codebook(v, 1, n_init=2)
    assert abs(c[0, 0] - v.mean()) < 1e-12
    dup = np.round(rs.normal(size=6000) * 4) / 4  # ~30 distinct values
    d = {}
    c = train_cb.train_scalar_codebook(dup, 6, n_init=2, details=d)
    oc, oin, oit, oseeds = KO.fit(dup, 6, n_init=2)
    assert np.array_equal(c, oc) and np.array_equal(d["seeds"], oseeds) and d["inertia"] == oin
    # more clusters than distinct values: the surplus centres stay empty (every point sits on its centre: no relocation) and go
    # "to the location of the biggest cluster" (sklearn's _average_centers); n = k; all values equal
    for v, k in ((np.repeat([1.0, -2.0], 700), 3), (np.repeat([3.0, -1.0, 0.5], [50, 900, 20]), 5), (np.full(1000, 0.25), 4),
                 (np.arange(5.0) ** 2, 5), (rs.randint(0, 6, size=5000).astype(np.float64), 6)):
        d = {}
        c = train_cb.train_scalar_codebook(v, k, n_init=2, details=d)
        oc, oin, oit, oseeds = KO.fit(v, k, n_init=2)
        assert np.array_equal(c, oc) and np.array_equal(d["seeds"], oseeds) and d["inertia"] == oin and d["n_iter"] == oit, (k, len(v))
    with pytest.raises(ValueError, match="should be >= n_clusters"):
        train_cb.train_scalar_codebook(np.arange(3.0), 4)
    L = _lib.lib()
    # the relocation: all candidates of the second seed drawn at u = 0 (-> point 0) behind first seed 0: two equal centres, the
    # second one wins no point in the first E-step and takes the point farthest from its centre (sklearn's rule, the oracle's)
    xs = rs.laplace(size=5000) * 0.1
    xs -= xs.mean()
    k, trials = 5, 3
    first = np.zeros(1, dtype=np.int64)
    u = rs.uniform(size=(1, k - 1, trials))
    u[0, 0] = 0.0
    tol = float(np.var(xs) * 1e-4)
    oc, oin, oit, oseeds = KO.kmeans1d(xs, k, first, u, tol)
    assert oseeds[0][0] == oseeds[0][1] == 0
    xd = torch_cuda.from_numpy(xs).cuda()
    out, inertia, n_iter, seeds = np.zeros(k), C.c_double(), C.c_int(), np.zeros((1, k), dtype=np.int32)
    _lib.check(L.fpc_kmeans1d(C.c_void_p(xd.data_ptr()), len(xs), k, 1, trials, first.ctypes.data_as(C.c_void_p),
                              u.ctypes.data_as(C.c_void_p), tol, 300, out.ctypes.data_as(C.c_void_p), C.byref(inertia),
                              C.byref(n_iter), seeds.ctypes.data_as(C.c_void_p), None), "fpc_kmeans1d")
    assert np.array_equal(seeds, oseeds) and np.array_equal(out, oc) and inertia.value == oin and n_iter.value == oit
    x = torch_cuda.zeros(10, dtype=torch_cuda.float64, device="cuda")
    out = np.zeros(4)
    rc = L.fpc_kmeans1d(C.c_void_p(x.data_ptr()), 10, 4, 1, 3, first.ctypes.data_as(C.c_void_p), None, 0.0, 300,
                        out.ctypes.data_as(C.c_void_p), None, None, None, None)
    assert rc == -1 and b"uniform draws" in L.fpc_last_error()
    rc = L.fpc_kmeans1d(C.c_void_p(x.data_ptr()), 10, 4000, 1, 3, first.ctypes.data_as(C.c_void_p), None, 0.0, 300,
                        out.ctypes.data_as(C.c_void_p), None, None, None, None)
    assert rc == -1 and b"k = 4000" in L.fpc_last_error()
