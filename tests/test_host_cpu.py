"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol of
include/fpcodec.h, fails loudly without a GPU, config/feature-file helpers, oracle
known-answer tests for the vocoder pieces (parity unpinned -> self-consistency), and the
world_size-2 gloo rehearsal of the multi-GPU sharding."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from fpcodec_amd import _lib
    return _lib


def test_abi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "fpcodec.h")).read()
    declared = set(re.findall(r"FPC_API [^;(]*?\b(fpc_[a-z0-9_]+)\(", hdr))
    assert declared and declared == set(built.SYMBOLS)
    L = built.lib()
    for s in declared:
        assert hasattr(L, s), s
    m = re.search(r"#define FPC_ABI_VERSION (\d+)", hdr)
    assert m and L.fpc_abi_version() == int(m.group(1)) == built.ABI_VERSION == 4
    # the shipped library is built without -D tunables and says so (a variant build lists them: tools/build_variant.sh)
    assert L.fpc_build_info() == b"fpcodec abi 4 gfx950"


def test_binding_names_a_library_of_another_abi_version(built, tmp_path, monkeypatch):
    """a library of another ABI version (an older build reached through FPC_LIB_PATH) is refused by VERSION, before any
    other symbol is looked up: a stand-in .so that exports nothing but fpc_abi_version() = 1"""
    import subprocess
    src = tmp_path / "old.c"
    src.write_text("int fpc_abi_version(void) { return 1; }\n")
    so = tmp_path / "libold.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(built, "LIB_PATH", str(so))
    monkeypatch.setattr(built, "_lib", None)
    with pytest.raises(built.FpcError, match="ABI version 1, this binding needs 4"):
        built.lib()


def test_default_mapping_of_utterances_onto_workgroups(built):
    """fpc_lpcnet_paired_utterances: the split fpc_lpcnet_synthesize makes by default (include/fpcodec.h).  Nothing paired up to the
    CU count; everything up to twice that; then paired rounds plus one plain round where that is cheaper; never dearer than rounds
    of one utterance per workgroup (a paired round of 2 x CUs counts 1.56 plain rounds)"""
    f = built.lib().fpc_lpcnet_paired_utterances
    cus = 256
    assert [f(b, cus) for b in (1, 2, 255, 256)] == [0, 0, 0, 0]
    assert [f(b, cus) for b in (257, 300, 384, 512)] == [257, 300, 384, 512]
    assert [f(b, cus) for b in (513, 600, 768)] == [512, 512, 512]       # one paired round + one plain round
    assert [f(b, cus) for b in (769, 1000, 1024)] == [769, 1000, 1024]   # two paired rounds
    assert f(1100, cus) == 1024 and f(10, 4) == 8 and f(9, 4) == 8 and f(5, 4) == 5
    for b in range(1, 3000, 7):
        npair = f(b, cus)
        assert 0 <= npair <= b and (npair == b or npair % (2 * cus) == 0)
        cost = 1.56 * -(-npair // (2 * cus)) + -(-(b - npair) // cus)
        assert cost <= -(-b // cus) + 1e-9, b


def test_device_buffer_is_empty_after_a_failed_allocation(built):
    """DevBuf::alloc wrote the size before hipMalloc; a failed grow left p == NULL with the new size, and a later,
    smaller forward call would have launched on the null block.  fpc_selftest exercises the failure (no device here: every
    hipMalloc fails) and checks pointer and size"""
    L = built.lib()
    assert L.fpc_selftest() == 0, L.fpc_last_error()


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = built.lib()
    assert L.fpc_device_count() == 0
    h = C.c_void_p()
    w = built.PredictorWeights(20, 384, 128, 18, *([1] * 10))
    rc = L.fpc_predictor_create(C.byref(w), C.byref(h))
    assert rc == -3 and b"no HIP device" in L.fpc_last_error()
    with pytest.raises(built.FpcError):
        built.require_gpu()
    import fpcodec_amd
    m = fpcodec_amd.Wavernn(20, 384, 128, 18)
    m.load_state_dict(fpcodec_amd.synth.predictor_state_dict())
    with pytest.raises(built.FpcError):
        m.forward(torch.zeros(1, 2, 20))


def test_state_dict_contract(synth):
    import fpcodec_amd
    m = fpcodec_amd.Wavernn(20, 384, 128, 18)
    sd = synth.predictor_state_dict()
    assert list(m.shapes().keys()) == list(sd.keys())  # src/models/wavernn.py:24-52 order
    assert sum(v.size for v in sd.values()) == 667410
    bad = dict(sd)
    bad.pop("dual_fc.0.bias")
    with pytest.raises(KeyError):
        m.load_state_dict(bad)
    bad = dict(sd)
    bad["rnn1.weight_ih_l0"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        m.load_state_dict(bad)


def test_cfg_overrides():
    from fpcodec_amd.config import parse_overrides
    cfg = parse_overrides("with cfg.l1=0.09 cfg.l2=0.28 cfg.qtz=True cfg.note=abc cfg.total_secs=3".split())
    assert cfg["l1"] == 0.09 and cfg["qtz"] is True and cfg["note"] == "abc"
    assert cfg["total_secs"] * cfg["sr"] // cfg["n_sample_seg"] == 20  # synthesis_qtz.py:97-98


def test_codebook_file_formats(tmp_path, synth):
    from fpcodec_amd.vq_func import read_vq_file, read_scl_file, cal_entropy
    c = synth.codebooks()
    p = tmp_path / "cb.npy"
    np.save(p, c["vq_hi"])
    st = read_vq_file(str(p))
    assert len(st) == 2 and st[0].shape == (1024, 17) and st[0].dtype == np.float64
    rag = np.empty(2, dtype=object)
    rag[0], rag[1] = c["vq_hi"][0], c["vq_hi"][1][:512]
    np.save(p, rag, allow_pickle=True)
    st = read_vq_file(str(p))
    assert st[1].shape == (512, 17)
    np.save(p, c["vq_hi"][0])  # 2-D files crash the reference (vq_func.py:143-146)
    with pytest.raises(ValueError):
        read_vq_file(str(p))
    np.save(p, c["scl_hi"])
    assert read_scl_file(str(p)).shape == (256,)
    h = np.array([1.0, 1.0, 2.0, 0.0])
    assert abs(cal_entropy(h) - 1.5) < 1e-12 and h[2] == 2.0  # does not mutate (the reference does)


def test_feature_file_readers(tmp_path):
    from fpcodec_amd.lpcnet import read_features
    a = np.arange(5 * 36, dtype=np.float32).reshape(5, 36)
    a.tofile(tmp_path / "f.f32")
    np.save(tmp_path / "f.npy", a[None])
    assert np.array_equal(read_features(str(tmp_path / "f.f32")), a)
    assert np.array_equal(read_features(str(tmp_path / "f.npy")), a)


# ---------------- vocoder oracle known-answer tests (parity unpinned) ----------------
def test_philox_known_answer(oracle):
    # Random123 Philox4x32-10 KAT: counter 0, key 0 -> first word 0x6627e8d5
    u = oracle.lib().orc_philox_uniform(0, 0)
    assert u == np.float32((0x6627e8d5 >> 8) * 2.0 ** -24)


def test_tree_pdf_properties(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    q = rng.uniform(0.05, 0.95, 256).astype(np.float32)
    p = np.zeros(256, np.float32)
    L.orc_tree_pdf(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))
    assert abs(p.sum() - 1.0) < 1e-5
    v = 0b10110010
    node, ref = 1, 1.0
    for l in range(8):
        bit = (v >> (7 - l)) & 1
        ref *= q[node] if bit else 1 - q[node]
        node = 2 * node + bit
    assert abs(p[v] - ref) < 1e-7
    q[:] = 0.5  # all-zero dual-FC weights -> sigmoid(0) -> uniform pdf
    L.orc_tree_pdf(q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))
    assert np.all(p == np.float32(1 / 256))


def test_ulaw_roundtrip_and_activations(oracle):
    L = oracle.lib()
    assert all(L.orc_lin2ulaw(L.orc_ulaw2lin(u)) == u for u in range(256))
    assert L.orc_lin2ulaw(0.0) == 128 and L.orc_lin2ulaw(-0.0) == 128
    assert L.orc_lin2ulaw(1e9) == 255 and L.orc_lin2ulaw(-1e9) == 0
    xs = np.linspace(-12, 12, 4001)
    t = np.array([L.orc_tanh(float(x)) for x in xs])
    assert np.abs(t - np.tanh(xs)).max() < 5e-7
    s = np.array([L.orc_sigmoid(float(x)) for x in xs])
    assert np.abs(s - 1 / (1 + np.exp(-xs))).max() < 5e-7


def test_lpc_only_impulse_response(oracle, synth):
    """zero network weights => excitation index fixed by the uniform pdf; the synthesis filter
    then must equal scipy.signal.lfilter([1],[1,a...]) on u2l(exc) (src/utils.py:91-114 sign)."""
    from scipy.signal import lfilter
    from fpcodec_amd import _lib
    w = {k: np.zeros(s, np.float32) for k, s in _lib.LPCNET_SHAPES.items()}
    orc = oracle.LPCNet(w)
    T = 3
    feat = np.zeros((T, 36), np.float32)
    a = np.array([-0.9, 0.2] + [0.0] * 14, np.float32)
    feat[:, 20:] = a
    pcm, exc, pf = orc.synthesize(feat, 7, trace=True)
    e = np.array([oracle.lib().orc_ulaw2lin(int(v)) for v in exc[17:]], np.float64)
    ref = lfilter([1.0], np.concatenate([[1.0], a.astype(np.float64)]), e)
    assert np.abs(pf[17:] - ref).max() < 1e-2 * max(1.0, np.abs(ref).max())
    # uniform pdf: inverse-CDF of the Philox uniform
    u = np.array([oracle.lib().orc_philox_uniform(7, t) for t in range(17, T * 160)])
    assert np.abs(exc[17:].astype(int) - np.floor(u * 256)).max() <= 1


def test_vocoder_oracle_determinism_and_seed(oracle, synth):
    w = synth.lpcnet_weights()
    orc = oracle.LPCNet(w)
    assert orc.nblocks == 1382  # 4608 blocks x (.05,.05,.2)
    f = synth.vocoder_features_raw(1, 5)[0]
    f[:, 20:] = oracle.ceps2lpc(f[:, :20])[0]
    a = orc.synthesize(f, 1004)
    assert np.array_equal(a, orc.synthesize(f, 1004))
    assert not np.array_equal(a, orc.synthesize(f, 1005))
    assert (a[:17] == 0).all() and np.abs(a[17:].astype(int)).max() > 0


# ---------------- multi-GPU sharding rehearsal on CPU (gloo, world_size 2) ----------------
_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["FPC_ROOT"])
import fpcodec_amd
from fpcodec_amd.parallel import shard_range, gather_report
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, T = 6, 3
lo, hi = shard_range(B, rank, world)
synth = fpcodec_amd.synth
w = synth.lpcnet_weights(); orc = O.LPCNet(w)
feats = synth.vocoder_features_raw(B, T)
feats[:, :, 20:] = O.ceps2lpc(feats.reshape(-1, 36)[:, :20])[0].reshape(B, T, 16)
seeds = synth.seeds(B)
# stand-in decode for the CPU rehearsal: the oracle plays the role of the per-rank GPU decode
local = np.stack([orc.synthesize(feats[b], int(seeds[b])) for b in range(lo, hi)])
rep = gather_report(elapsed_s=0.5 + rank, samples=local.size, hist=np.full(4, rank + 1, np.int64))
if rank == 0:
    assert rep["samples"] == B * T * 160 and rep["elapsed_s"] == 0.5 + world - 1
    assert np.array_equal(rep["hist"], np.full(4, sum(range(1, world + 1))))
# shard invariance: same PCM as an unsharded decode
full = np.stack([orc.synthesize(feats[b], int(seeds[b])) for b in range(B)])
assert np.array_equal(local, full[lo:hi])
print("rank", rank, "ok", lo, hi)
dist.destroy_process_group()
'''


def test_gloo_world2_sharding(tmp_path, oracle):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, FPC_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    import socket
    sk = socket.socket()  # a free port (a fixed one collides when two test runs share a host)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0 ok 0 3" in r.stdout and "rank 1 ok 3 6" in r.stdout


# ---- receiver side: bitstream pack/unpack and the decoder restatement ----
def test_bitstream_roundtrip_and_oracle_decoder(oracle, synth):
    from fpcodec_amd import bitstream
    c = synth.codebooks()
    P = oracle.Predictor(synth.predictor_state_dict())
    feat = synth.predictor_features(3, 60, utt0=700)
    for CB in (oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"]),   # production: all four
               oracle.Codebooks(c["vq_hi"], c["scl_hi"]),                               # nothing below threshold
               oracle.Codebooks(c["vq_hi"][:1], c["scl_hi"], c["vq_lo"], None)):        # 1-stage VQ, no low scalar
        o = P.encode(feat, CB, 0.09, 0.28, True)
        rebuilt = np.zeros_like(o["idx"])
        total = 0
        for b in range(feat.shape[0]):
            data, nbits = bitstream.pack(o["idx"][b], CB.sizes)
            assert len(data) == (nbits + 7) // 8
            rebuilt[b] = bitstream.unpack(data, feat.shape[1], CB.sizes)
            total += nbits
        assert np.array_equal(rebuilt, o["idx"])
        assert abs(total / (feat.shape[0] * feat.shape[1]) - bitstream.bits_per_frame(o["idx"], CB.sizes)) < 1e-9
        # the receiver needs nothing but the symbols and the pitch columns
        dec = P.decode(CB, rebuilt, feat[:, :, 18:20])
        assert np.array_equal(dec, o["c_in"])
    # an all-quiet utterance costs the two flag bits (+ the below-threshold fields when those codebooks exist)
    quiet = np.full((10, 4), -1, np.int32)
    assert bitstream.pack(quiet, [256, 0, 1024, 1024, 0])[1] == 20


# ---- feature-file formats either side of the path ----
def test_feature_windows_and_synthesis_frames(tmp_path):
    from fpcodec_amd import features_io as F
    nfr = 15 * 7 + 9
    a = np.arange(nfr * 36, dtype=np.float32).reshape(nfr, 36)
    a.tofile(tmp_path / "u_features.f32")
    w = F.f32_to_windows(str(tmp_path / "u_features.f32"))
    assert w.shape == (7, 19, 36)           # len // (15*36) windows (write_small_files.py:56), all inside the data
    for k in range(7):                      # hop 15, 19 frames each (strides of write_small_files.py:60-64)
        assert np.array_equal(w[k], a[15 * k:15 * k + 19])
    q = w + 1000.0
    nm, qf = F.synthesis_frames(w, q, chunks=3)
    assert nm.shape == (45, 36) and qf.shape == (45, 36)
    centre = np.concatenate([a[15 * k + 2:15 * k + 17] for k in (4, 5, 6)])  # the LAST 3 windows, centre frames
    expect = centre.copy()
    expect[:, -2:] += 1000.0               # pitch columns come from the quantised file (dataset_syn.py:72)
    assert np.array_equal(nm, expect / np.float32(24.1))
    assert np.array_equal(qf, centre + 1000.0)
    nm2, _ = F.synthesis_frames(w[:2], None, chunks=5)   # short utterance: doubled until long enough
    assert nm2.shape == (75, 36)
    F.frames_to_f32(str(tmp_path / "o.f32"), centre)
    from fpcodec_amd.lpcnet import read_features
    assert np.array_equal(read_features(str(tmp_path / "o.f32")), centre)
    assert F.f32_to_windows(a[:10].ravel()).shape == (0, 19, 36)


def test_wav_writer_normalisation(tmp_path):
    import wave
    from fpcodec_amd import features_io as F
    x = np.random.default_rng(3).normal(0, 300.0, 4000)
    pcm = F.write_wav(str(tmp_path / "t.wav"), x)
    ref = x / np.std(x)
    ref = ref / np.max(np.abs(ref))        # synthesis_qtz.py:46-47
    assert np.array_equal(pcm, np.rint(ref * 32767.0).astype(np.int16)) and np.abs(pcm).max() == 32767
    with wave.open(str(tmp_path / "t.wav"), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 4000)
        assert np.array_equal(np.frombuffer(w.readframes(4000), dtype="<i2"), pcm)


def test_entropy_coded_stream_meets_the_entropy_figure(oracle, synth):
    """arithmetic-coded symbols decode exactly and cost what generate_qtz_features.py:94-101,202 predicts"""
    from fpcodec_amd import bitstream
    c = synth.codebooks()
    CB = oracle.Codebooks(c["vq_hi"], c["scl_hi"], c["vq_lo"], c["scl_lo"])
    P = oracle.Predictor(synth.predictor_state_dict())
    feat = synth.predictor_features(4, 150, utt0=900)
    o = P.encode(feat, CB, 0.09, 0.28, True)
    cb_tot = CB.split_hist(o["hist"])
    keep = (float(o["ind1"].mean()), float(o["ind2"].mean()))
    models = bitstream.Models(CB.sizes, cb_tot, keep)
    total_bits = 0
    for b in range(4):
        data, nbits = bitstream.entropy_pack(o["idx"][b], models)
        assert np.array_equal(bitstream.entropy_unpack(data, 150, models), o["idx"][b])
        total_bits += nbits
    n = 4 * 150
    ent = sum(oracle.cal_entropy(h) * h.sum() for h in cb_tot if h.sum() > 0) / n     # codebook symbols
    ent += sum(-(p * np.log2(p) + (1 - p) * np.log2(1 - p)) for p in keep)          # the two flags, coded too
    coded = total_bits / n
    fixed = bitstream.bits_per_frame(o["idx"], CB.sizes)
    assert coded < fixed and abs(coded - ent) < 0.03 * ent + 4 * 40 / n, (coded, ent, fixed)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`--gpus N` is honoured: under a launcher whose WORLD_SIZE differs the bench exits non-zero instead of silently
    measuring another configuration (and without a launcher it starts its own ranks: covered on the GPU box)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """one process per GPU: under the RCCL backend a world larger than the visible device count must not be folded onto
    fewer GPUs and reported as an N-GPU line (round-2 review item 7); here no GPU is visible at all"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    env.pop("FPC_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "visible GPUs" in (r.stdout + r.stderr)


def test_traffic_record_is_tied_to_the_kernel_sources():
    """bench.py takes roofline.traffic only from a PMC record whose kernel_source_sha256 equals the hash of the decode
    kernel's sources in the tree (a stale record reads null, never an outdated number)"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    h = bench.decode_kernel_hash()
    assert len(h) == 64
    recs = [n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_traffic.json")]
    assert recs
    tagged = [n for n in recs if "kernel_source_sha256" in json.load(open(os.path.join(ROOT, "profiles", n)))]
    # round-1/2 records carry no hash and can never match; a record that carries one names a 64-digit hash
    for n in tagged:
        assert len(json.load(open(os.path.join(ROOT, "profiles", n)))["kernel_source_sha256"]) == 64


def test_wavernn_module_surface_without_gpu(synth):
    """the torch.nn.Module-like surface the reference's scripts touch (synthesis_qtz.py:79-87, train_frame.py:235-250):
    constructor keywords, .to/.cuda/.train/.eval chaining, state_dict round trip, parameters(), strict=False"""
    import torch
    from fpcodec_amd.wavernn import Wavernn
    m = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18, attn_units=128, bidirectional=False,
                packing=False).to("cuda")
    assert m.cuda() is m and m.train() is m and m.training and m.eval() is m and not m.training
    sd = synth.predictor_state_dict()
    with pytest.raises(KeyError):  # a never-loaded model has nothing to keep for missing keys
        m.load_state_dict({k: v for k, v in sd.items() if k != "dual_fc.0.bias"}, strict=False)
    m.load_state_dict(sd)
    ps = m.parameters()
    assert len(ps) == 10 and ps.model is m and all(isinstance(p, torch.Tensor) for p in ps)
    assert [tuple(p.shape) for p in ps] == [tuple(v.shape) for v in m.state_dict().values()]
    # transfer load (train_frame.py:244-248): extra keys ignored, missing keys keep their values
    part = {k: np.zeros_like(v) for k, v in sd.items() if k.startswith("rnn1")}
    part["mask_rnn.weight_ih_l0"] = np.zeros((3, 3), np.float32)
    m.load_state_dict(part, strict=False)
    got = m.state_dict()
    assert float(got["rnn1.weight_hh_l0"].abs().max()) == 0.0
    assert np.array_equal(got["rnn2.weight_ih_l0"].numpy(), sd["rnn2.weight_ih_l0"])
    with pytest.raises(KeyError):
        m.load_state_dict(part, strict=True)


def test_train_cb_file_formats_and_scalar_codebook(tmp_path):
    """codebook files in the reference's formats (train_cb.py:217,219-221): (S, N, 17) float64 / object array of
    stages / (n, 1) float64 KMeans centres; all three load through the repo's codebook reader"""
    from fpcodec_amd import train_cb
    from fpcodec_amd.vq_func import read_vq_file as _read_vq_file
    rng = np.random.default_rng(3)
    st = [rng.normal(size=(8, 17)), rng.normal(size=(8, 17))]
    train_cb.save_codebook(str(tmp_path / "a.npy"), st)
    a = np.load(str(tmp_path / "a.npy"))
    assert a.shape == (2, 8, 17) and a.dtype == np.float64
    train_cb.save_codebook(str(tmp_path / "b.npy"), [st[0], st[1][:5]])
    b = np.load(str(tmp_path / "b.npy"), allow_pickle=True)
    assert b.dtype == object and b[1].shape == (5, 17)
    assert [s.shape for s in _read_vq_file(str(tmp_path / "b.npy"))] == [(8, 17), (5, 17)]
    vals = np.concatenate([rng.normal(-1, .01, 200), rng.normal(0.5, .01, 300), rng.normal(2, .01, 100)])
    c = train_cb.train_scalar_codebook(vals, 3, backend="sklearn")
    assert c.shape == (3, 1) and c.dtype == np.float64
    assert np.allclose(np.sort(c[:, 0]), [-1, .5, 2], atol=0.01)


def test_feature_io_vs_reference_golden(tmp_path, golden):
    """G11: the reference's own statements (compiled from its files alone, tests/golden/make_golden_io.py) of the
    .f32 windows (write_small_files.py:56-64), the encoded-frame windows (generate_qtz_features.py:65-70), the
    synthesis frames (dataset_syn.py:66-97; chunks 3 / all / more than the utterance has) and saveaudio's
    normalisation (synthesis_qtz.py:39-50)"""
    import importlib.util
    import torch
    import wave
    from fpcodec_amd import features_io as F
    spec = importlib.util.spec_from_file_location("mgio", os.path.join(ROOT, "tests", "golden", "make_golden_io.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    g = golden("g11_feature_io")
    assert np.array_equal(F.f32_to_windows(m.f32_inputs()), g["w_f32"])
    enc = m.enc_inputs()
    w = F.frames_to_windows(enc)
    assert w.shape == (10, 19, 36) and np.array_equal(w, g["w_enc"])          # (154 - 4) // 15 = 10
    assert F.frames_to_windows(np.zeros((300, 36), np.float32)).shape == (19, 19, 36)  # 3 s: (300 - 4) // 15
    assert F.frames_to_windows(enc[0, :18]).shape == (0, 19, 36)
    F.save_windows(str(tmp_path / "u_features.pt"), enc)
    assert np.array_equal(torch.load(str(tmp_path / "u_features.pt")).numpy(), g["w_enc"])
    f, q = m.syn_inputs()
    for chunks in (3, 0, 20):
        nm, qf = F.synthesis_frames(f, q, chunks=chunks)
        assert np.array_equal(nm, g[f"nm_{chunks}"]) and np.array_equal(qf, g[f"qf_{chunks}"]), chunks
    x = m.wave_inputs()
    # the reference normalises the float32 array in place (float32 arithmetic); libsndfile then scales by 0x7FFF
    ref = g["wav"]
    assert np.abs(F.normalise_wave(x) - ref).max() < 1e-6 and str(g["wav_name"]).endswith("_truth.wav")
    pcm = F.write_wav(str(tmp_path / "t.wav"), x)
    assert np.abs(pcm.astype(np.int64) - np.rint(ref.astype(np.float64) * 32767.0)).max() <= 1
    with wave.open(str(tmp_path / "t.wav"), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()) == (1, 2, 16000, 4800)


def test_keras_checkpoint_name_mapping(synth, tmp_path):
    """tools/h5_to_npz.py: a dict with Keras weight paths (save_weights layout, a CuDNNGRU-style flat bias, nested
    cell names) maps onto the 19 arrays of fpc_lpcnet_weights; wrong shapes and ambiguous names are refused"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("h5_to_npz", os.path.join(ROOT, "tools", "h5_to_npz.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from fpcodec_amd import _lib
    w = synth.lpcnet_weights()
    named = {}
    for key, layer, weight, shape in m.MAPPING:
        nest = f"{layer}/{layer}/gru_cell" if layer == "gru_b" else f"{layer}/{layer}"
        named[f"{nest}/{weight}:0"] = w[key].reshape(-1) if key == "gru_a_bias" else w[key]
    named["top_level_model_weights/whatever:0"] = np.zeros(3)
    got = m.from_keras_named(named)
    assert sorted(got) == sorted(_lib.LPCNET_KEYS)
    for k in _lib.LPCNET_KEYS:
        assert got[k].dtype == np.float32 and got[k].shape == _lib.LPCNET_SHAPES[k] and np.array_equal(got[k], w[k])
    np.savez(str(tmp_path / "m.npz"), **got)
    with np.load(str(tmp_path / "m.npz")) as z:
        assert all(np.array_equal(z[k], w[k]) for k in _lib.LPCNET_KEYS)
    bad = dict(named)
    bad["gru_a/gru_a/kernel:0"] = np.zeros((640, 1152), np.float32)  # a model generation with other sizes
    with pytest.raises(ValueError):
        m.from_keras_named(bad)
    dup = dict(named)
    dup["other/dual_fc/kernel:0"] = w["md_kernel"]
    with pytest.raises(KeyError):
        m.from_keras_named(dup)


def test_no_hand_placed_vector_loads_in_the_predictor_sources():
    """rounds 2-3 placed the weight loads of some predictor kernels and their waits by hand (inline assembly the compiler does
    not count); a window register copied between its load and its wait gave wrong results once the code shape changed.  Round
    4 moved every chain to plain loads + scheduling group barriers: no kernel source may issue a vector memory LOAD from inline
    assembly any more (the remaining asm statements are register reads, DPP moves, waits without operands and the plain
    granule store of the same-XCD hop)."""
    src = os.path.join(ROOT, "feature-predictor-for-speech-codec_amd", "csrc")
    for f in ("predictor.hip", "predictor_df.h", "predictor_ws.h", "cb_train.hip", "ceps2lpc.hip"):
        text = open(os.path.join(src, f)).read()
        for stmt in re.findall(r'asm\s*(?:volatile)?\s*\((.*?)\);', text, re.S):
            assert "_load_" not in stmt, (f, stmt[:120])


def _scalar_residuals(n, seed):
    """float32-born scalar residuals (what train_cb.py:174-175 collects), Laplace-shaped like prediction residuals"""
    rs = np.random.RandomState(seed)
    return (rs.laplace(size=n) * 0.1).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("n,k", [(5000, 8), (12000, 32), (3000, 64)])
def test_kmeans_oracle_is_pinned_to_sklearn(n, k):
    """oracle/kmeans1d_oracle.py (what csrc/kmeans1d.hip is bit-identical to) against scikit-learn itself, the reference's call
    (train_cb.py:219-226): the seeding picks sklearn's seeds index for index (same RandomState stream, same candidates, same
    winners), Lloyd stops in the same iteration, centres and inertia agree to rounding -- sklearn's long sums have no
    specified order, the oracle's have one"""
    from sklearn.cluster import KMeans, kmeans_plusplus
    from sklearn.utils.extmath import row_norms
    sys.path.insert(0, ROOT)
    from oracle import kmeans1d_oracle as KO
    from fpcodec_amd import train_cb
    v = _scalar_residuals(n, 3 + k)
    c, inertia, n_iter, seeds = KO.fit(v, k, n_init=3)
    km = KMeans(n_clusters=k, random_state=0, n_init=3).fit(v[:, None])
    assert np.abs(c - km.cluster_centers_).max() < 1e-12
    assert abs(inertia - km.inertia_) <= 1e-12 * km.inertia_ and n_iter == km.n_iter_
    X = v[:, None] - v[:, None].mean(axis=0)
    _, idx = kmeans_plusplus(X, k, random_state=np.random.RandomState(0), x_squared_norms=row_norms(X, squared=True))
    assert np.array_equal(idx, seeds[0])
    # the product's host half draws the same stream
    f0, u0, t0 = KO.draws(n, k, 3)
    f1, u1, t1 = train_cb.kmeans_draws(n, k, 3)
    assert t0 == t1 == 2 + int(np.log(k)) and np.array_equal(f0, f1) and np.array_equal(u0, u1)


@pytest.mark.parametrize("n,k", [(5000, 8), (12000, 32), (3000, 64)])
def test_kmeans_oracle_c_twins_equal_the_numpy_definition(n, k):
    """oracle/fpc_oracle.c::orc_km_assign / orc_km_sums (what lets the oracle run the production sizes in the GPU suite) against
    kmeans1d_oracle._assign / _sums: the same labels, the same sums bit for bit, and the same fit"""
    from oracle import kmeans1d_oracle as KO
    rs = np.random.RandomState(7 + k)
    v = (rs.laplace(size=n) * 0.1).astype(np.float32).astype(np.float64)
    x = v - v.mean()
    centers = np.sort(rs.choice(x, k, replace=False))
    centers[1] = centers[0]  # a tie: the first minimum wins in both
    la, lb = KO._assign(x, centers), KO._assign_c(x, centers)
    assert np.array_equal(la, lb)
    (ta, ca), (tb, cb) = KO._sums(x, la, k), KO._sums_c(x, la, k)
    assert np.array_equal(ta, tb) and np.array_equal(ca, cb)
    a, b = KO.fit(v, k, n_init=2), KO.fit(v, k, n_init=2, fast=True)
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


def test_kmeans_oracle_relocates_an_empty_cluster_like_sklearn():
    """a cluster that loses every point (sklearn: _relocate_empty_clusters_dense -- the empty cluster takes the point farthest
    from its centre, the labels stay): forced by handing the Lloyd loop two equal centres; against sklearn's own
    _kmeans_single_lloyd from the same initial centres"""
    from sklearn.cluster import _kmeans as SK
    sys.path.insert(0, ROOT)
    from oracle import kmeans1d_oracle as KO
    v = _scalar_residuals(2000, 9)
    x = v - v.mean()
    init = np.array([x[5], x[5], x[700], x[1500]])  # the second of two equal centres wins no point (first minimum)
    labels = KO._assign(x, init)
    s, cnt = KO._sums(x, labels, 4)
    assert cnt[1] == 0.0
    KO._relocate(x, labels, init, s, cnt)
    far = int(np.argmax((x - init[labels]) ** 2))
    assert cnt[1] == 1.0 and s[1] == x[far]
    sk_labels, sk_inertia, sk_centers, sk_iter = SK._kmeans_single_lloyd(
        x[:, None].copy(), np.ones(len(x)), init[:, None].copy(), max_iter=300, tol=float(np.var(x) * 1e-4), n_threads=1)
    # the oracle's loop from the same start
    centers, labels_old = init.copy(), np.full(len(x), -1, dtype=np.int32)
    for it in range(300):
        labels = KO._assign(x, centers)
        s, cnt = KO._sums(x, labels, 4)
        KO._relocate(x, labels, centers, s, cnt)
        new = KO._average(s, cnt)
        shift = float(np.cumsum((centers - new) ** 2)[-1])
        centers = new
        if np.array_equal(labels, labels_old) or shift <= float(np.var(x) * 1e-4):
            break
        labels_old = labels
    assert it + 1 == sk_iter and np.abs(centers - sk_centers[:, 0]).max() < 1e-12


@pytest.mark.parametrize("case", ["two values", "three values", "all equal", "n = k"])
def test_kmeans_oracle_more_clusters_than_distinct_values_like_sklearn(case):
    """clusters that stay empty (no relocation: every point sits on its centre): sklearn's _average_centers puts them "at the
    location of the biggest cluster" while it averages in place -- the averaged centre if that cluster has a lower index, its raw
    sum if not; restated in the oracle, compared with KMeans itself.  The cases are exact in binary (values, mean, sums): with
    inexact sums sklearn's NEXT iteration relocates the surplus centres to whichever points its own rounding noise (1e-16) leaves
    farthest from their centres -- positions no other summation order reproduces, and not worth reproducing (sklearn warns
    "Number of distinct clusters found smaller than n_clusters")"""
    import warnings
    from sklearn.cluster import KMeans
    sys.path.insert(0, ROOT)
    from oracle import kmeans1d_oracle as KO
    v, k = {"two values": (np.repeat([1.0, -2.0], 700), 3), "three values": (np.repeat([3.0, -1.0, 0.5], [128, 256, 128]), 5),
            "all equal": (np.full(1000, 0.25), 4), "n = k": (np.arange(5.0) ** 2, 5)}[case]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=k, random_state=0, n_init=3).fit(v[:, None])
    c, inertia, n_iter, _ = KO.fit(v, k, n_init=3)
    assert np.abs(c - km.cluster_centers_).max() < 1e-12 and n_iter == km.n_iter_ and abs(inertia - km.inertia_) < 1e-12


def test_scalar_codebook_gpu_backend_needs_the_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fpcodec_amd import train_cb
    with pytest.raises(built.FpcError, match="no CPU fallback"):
        train_cb.train_scalar_codebook(np.arange(100.0), 4)
    with pytest.raises(ValueError):
        train_cb.train_scalar_codebook(np.arange(100.0), 4, backend="numpy")
