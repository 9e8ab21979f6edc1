"""Soak of k_decode2 against the oracle: more weight sets, densities, seeds and voiced mixes than the driver-run stress test
(tests/test_gpu_parity.py::test_vocoder_stress_parity_randomised[1]), odd batch sizes (the last workgroup decodes one utterance
twice), with and without chunking.    python tools/stress_parity_pairs.py   (exit code 1 on any difference)"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd, concurrent.futures as cf
from fpcodec_amd.lpcnet import LPCNet
from oracle import oracle as O
synth = fpcodec_amd.synth
bad = tot = 0
cases = [(1004, (0.05, 0.05, 0.2), 31), (77, (0.03, 0.06, 0.18), 48), (5, (0.02, 0.02, 0.1), 17), (901, (0.06, 0.04, 0.22), 40),
         (33, (0.01, 0.03, 0.15), 25), (2024, (0.05, 0.05, 0.24), 36)]
for wseed, dens, B in cases:
    w = synth.lpcnet_weights(seed=wseed, density=dens)
    voc = LPCNet(w); orc = O.LPCNet(w)
    voc.set_pairing(1)
    T = 30
    f = synth.vocoder_features_raw(B, T, utt0=wseed * 10)
    f[:, :, 19] = np.random.default_rng(wseed).uniform(-0.5, 1.0, (B, T))
    f[:, :, 20:] = O.ceps2lpc(f.reshape(-1, 36)[:, :20])[0].reshape(B, T, 16)
    sd = np.random.default_rng(wseed + 1).integers(0, 2 ** 62, B).astype(np.uint64)
    pcm = voc.synthesize(f, sd).cpu().numpy()
    spw = voc.last_streams_per_workgroup()
    voc.set_chunk_frames(4 + wseed % 7)
    chunked = voc.synthesize(f, sd).cpu().numpy()
    with cf.ThreadPoolExecutor(16) as ex:
        refs = list(ex.map(lambda b: orc.synthesize(f[b], int(sd[b])), range(B)))
    nb = sum(0 if np.array_equal(pcm[b], refs[b]) else 1 for b in range(B))
    nc = 0 if np.array_equal(chunked, pcm) else 1
    print(f"weights {wseed} {dens} variant {voc.kernel_variant()} {spw}/wg B={B}: mismatching utterances {nb}, chunked pass {'same' if nc == 0 else 'DIFFERENT'}", flush=True)
    bad += nb + nc
    tot += B * (T * 160 - 17)
print("samples compared", tot, "TOTAL MISMATCH", bad)
sys.exit(1 if bad else 0)
