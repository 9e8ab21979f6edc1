#!/usr/bin/env python3
"""Benchmark of the hot path: LPCNet synthesis samples/s on synthetic 3-second utterances.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 ...          # starts one rank per GPU itself (torch.distributed.run child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the vocoder (frame-rate conditioning kernels + persistent decode kernel)
over one batch of --streams independent 3 s utterances per GPU (BASELINE config 3: 256
utterances on one MI355X; with N GPUs the global list of N x 256 utterances is split by
`parallel.shard_range`: config 4, weak scaling, no collective in the data path - RCCL only
gathers the report).  Inputs (features, seeds, weights) are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  Outside the timed region the line also carries BASELINE
config 5 (`e2e`: predictor + residual VQ encode -> ceps2lpc -> decode on this rank's share of
128 x N utterances, codebook-usage histograms summed over all ranks before the entropy, as
src/generate_qtz_features.py:184,202 sums them over utterances; `two_batches_one_decode` /
`four_batches_one_paired_decode`: the one-GPU throughput forms, two encode batches per 256-stream decode
launch, four per 512-stream launch of k_decode2), `many_stream` (512 and 1 024
utterances on one GPU: k_decode2, two utterances per workgroup, beside rounds of k_decode), the
single-stream latency view (config 2), a second decode at 50 % voiced frames, and the CPU baselines.
`roofline.occupancy` / `many_stream.occupancy` / `e2e.predictor_roofline.occupancy`: waves per SIMD, CUs
busy and issue shares from the committed SQ-counter passes (profiles/*_counters.json, tied to the sources).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 145600.0      # 2 x 72 797 MAC per output sample per stream
HBM_BYTES_PER_SAMPLE = 2.9      # 2 B PCM out + 144 B features / 160 samples
PEAK_F32_TFLOPS = 157.3         # MI355X_MICROARCH.md: FP32 vector peak (== f32 MFMA dense peak)
PEAK_HBM_GBS = 8000.0
E2E_PER_GPU = 128               # BASELINE config 5: 1024 utterances on 8 GPUs
PRED_FLOP_PER_FRAME = 1328640.0  # 2 x 664 320 MAC per frame and utterance (GRU 20->384, 384->128, FC 128->18)
VQ2_FLOP = 6 * 1024 * 17 * 3.0   # 2-stage search: 1 + 5 scans of 1 024 entries, (sub, mul, add) per coordinate
VQ1_FLOP = 512 * 17 * 3.0        # below the threshold: one scan of 512 entries


def _host_threads():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU share (cgroup v2 quota) rather than the host's thread count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return min(cores, 16)  # one GPU's share of the host on the measurement boxes


def cpu_baseline(frames=64):
    """the CPU oracle (C restatement) on a bounded sample of the same workload: one utterance per host core
    (ctypes releases the GIL, the oracle is re-entrant), and one core alone for the latency view"""
    import concurrent.futures as cf
    import fpcodec_amd
    from oracle import oracle as O
    synth = fpcodec_amd.synth
    w = synth.lpcnet_weights()
    orc = O.LPCNet(w)
    f = synth.vocoder_features_raw(1, frames)[0]
    f[:, 20:] = O.ceps2lpc(f[:, :20])[0]
    per = frames * 160 - 17
    n1, reps1, t0 = 0, 0, time.time()
    while time.time() - t0 < 4.0:
        orc.synthesize(f, 1004 + reps1)
        n1 += per
        reps1 += 1
    single = n1 / (time.time() - t0)
    cores = _host_threads()

    def worker(k):
        n, t1 = 0, time.time()
        while time.time() - t1 < 6.0:
            orc.synthesize(f, 2000 + 100 * k + n)
            n += 1
        return n

    t0 = time.time()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(worker, range(cores)))
    dt = time.time() - t0
    out = {"value": done * per / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "single_thread_value": single,
           "sample": f"{done} x one {frames}-frame utterance ({done * per} samples) through oracle/fpc_oracle.c "
                     f"(orc_lpcnet_synthesize), one utterance at a time on each of {cores} host threads for 6 s; "
                     f"single_thread_value from {reps1} utterances on one thread",
           "note": "kind 'port': the reference's own vocoder (xiph/LPCNet, Keras) is absent from the reference tree "
                   "and cannot be timed anywhere; the reference's Python ENCODER measured in the build container "
                   "(8-core Xeon 2.1 GHz, BASELINE.md section 2) needs 17.0 s per 3 s utterance (0.176x real time)"}
    out["encode"] = cpu_baseline_encode(cores)
    return out


def cpu_baseline_encode(cores, frames=100):
    """encode side of the hot path on the host: the oracle's closed-loop encoder (GRU predictor, thresholds,
    scalar + 2-stage M-best VQ) and ceps2lpc, one utterance at a time per host thread"""
    import concurrent.futures as cf
    import fpcodec_amd
    from oracle import oracle as O
    synth = fpcodec_amd.synth
    pred = O.Predictor(synth.predictor_state_dict())
    cbs = synth.codebooks()
    cb = O.Codebooks(cbs["vq_hi"], cbs["scl_hi"], cbs["vq_lo"], cbs["scl_lo"])
    feat = synth.predictor_features(1, frames, utt0=7000)

    def one():
        enc = pred.encode(feat, cb, 0.09, 0.28, qtz=True)
        O.ceps2lpc(np.ascontiguousarray(enc["c_in"].reshape(-1, 20) * np.float32(24.1)))

    t0, n1 = time.time(), 0
    while time.time() - t0 < 3.0:
        one()
        n1 += 1
    single = n1 * frames / (time.time() - t0)

    def worker(k):
        n, t1 = 0, time.time()
        while time.time() - t1 < 5.0:
            one()
            n += 1
        return n

    t0 = time.time()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(worker, range(cores)))
    dt = time.time() - t0
    return {"value": done * frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_value": single, "rtf": done * frames / dt / 100.0,
            "sample": f"{done} x one {frames}-frame utterance through oracle/fpc_oracle.c (orc_encode + orc_ceps2lpc) "
                      f"on {cores} host threads for 5 s; single_thread_value from {n1} utterances on one thread",
            "reference_python_s_per_3s_utterance": 17.0}


def decode_kernel_hash():
    """sha256 over the sources k_decode is compiled from: ties a PMC record to the kernel it was taken on"""
    import hashlib
    h = hashlib.sha256()
    for rel in ("lpcnet_decode.h", "lpcnet.hip"):
        with open(os.path.join(ROOT, "feature-predictor-for-speech-codec_amd", "csrc", rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def decode2_kernel_hash():
    """the same for k_decode2 (two utterances per workgroup: lpcnet_decode2.h on top of k_decode's helpers)"""
    import hashlib
    h = hashlib.sha256()
    for rel in ("lpcnet_decode.h", "lpcnet_decode2.h", "lpcnet.hip"):
        with open(os.path.join(ROOT, "feature-predictor-for-speech-codec_amd", "csrc", rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def predictor_kernel_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in ("predictor.hip", "predictor_ws.h", "predictor_wsd.h", "predictor_df.h"):
        with open(os.path.join(ROOT, "feature-predictor-for-speech-codec_amd", "csrc", rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def counters_record(kernel, khash):
    """occupancy / issue shares of a kernel from the committed SQ-counter passes (profiles/*_counters.json, written by
    tools/counters_round.py from separate rocprofv3 --pmc passes): only a record taken on THESE sources is quoted"""
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted((n for n in os.listdir(pdir) if n.endswith("_counters.json")), reverse=True):
        try:
            rec = json.load(open(os.path.join(pdir, name))).get(kernel)
        except Exception:
            continue
        if rec and rec.get("kernel_source_sha256") == khash:
            return dict(rec, source=f"profiles/{name}")
    return None


def _pci_bus_id(dev):
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev)) == 0:
            return buf.value.decode()
    except Exception:
        pass
    return "unknown"


def e2e_config5(voc, torch, synth, parallel, rank, world, L=300):
    """BASELINE config 5, this rank's share of the global utterance list: encode (GRU predictor + thresholds +
    scalar/2-stage VQ) -> x24.1 -> ceps2lpc -> LPCNet decode; the codebook-usage histograms and the frame /
    sample counts of all ranks are summed (parallel.gather_report) before the entropies are taken, as
    src/generate_qtz_features.py:184,202 sums cb_tot over utterances."""
    import tempfile
    from fpcodec_amd import bitstream
    from fpcodec_amd.synthesis_qtz import encode_features
    from fpcodec_amd.vq_func import cal_entropy, load_codebooks
    from fpcodec_amd.wavernn import Wavernn
    d = tempfile.mkdtemp()
    paths = {}
    for k, v in synth.codebooks().items():
        paths[k] = os.path.join(d, k + ".npy")
        np.save(paths[k], v)
    cfg = dict(scl_cb_path=paths["scl_hi"], cb_path=paths["vq_hi"], bl_scl_cb_path=paths["scl_lo"],
               bl_cb_path=paths["vq_lo"], l1=0.09, l2=0.28, qtz=True)
    model = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    model.load_state_dict(synth.predictor_state_dict())
    if world > max(1, torch.cuda.device_count()):
        # ranks folded onto fewer devices (the gloo rehearsal): the predictor's multi-workgroup forms need all workgroups
        # of a group resident at once, which two processes on one GPU cannot promise each other -- one workgroup per
        # utterance (include/fpcodec.h "Kernel forms"; unpinned, every group would decide for the fallback after 10 ms)
        model.set_split(1)
    total = E2E_PER_GPU * world
    lo, hi = parallel.shard_range(total, rank, world)
    B = hi - lo
    nm = np.zeros((B, L, 36), np.float32)  # every utterance of the share is its own AR(1) draw: the bitrate and the
    nm[:, :, :20] = synth.predictor_features(B, L, utt0=5000 + lo)  # keep-rates are statistics of B x L distinct frames
    nm_d = torch.from_numpy(nm).cuda()
    seeds = torch.from_numpy(synth.seeds(B, utt0=5000 + lo).astype(np.int64)).cuda()
    pcm = torch.empty(B, L * 160, dtype=torch.int16, device="cuda")

    feats, r, i1, i2, cb_tot = encode_features(model, cfg, nm_d)  # warm-up
    voc.synthesize(feats, seeds, out=pcm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats, r, i1, i2, cb_tot = encode_features(model, cfg, nm_d)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    voc.synthesize(feats, seeds, out=pcm)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # ---- the one-GPU THROUGHPUT form of the same path: a 128-utterance batch fills the chip for the encoder (8 groups of 16
    # utterances x 32 workgroups) but only half of it for the vocoder (128 of 256 CUs), and k_decode costs the same 65 ms for
    # 256 streams as for 128.  So: encode two batches back to back, decode both in ONE 256-stream launch.
    nm2 = np.zeros((B, L, 36), np.float32)
    nm2[:, :, :20] = synth.predictor_features(B, L, utt0=9000 + lo)
    nm2_d = torch.from_numpy(nm2).cuda()
    seeds2 = torch.from_numpy(synth.seeds(B, utt0=9000 + lo).astype(np.int64)).cuda()
    pcm2 = torch.empty(2 * B, L * 160, dtype=torch.int16, device="cuda")
    sd2 = torch.cat([seeds, seeds2])

    def two_batches():
        fa = encode_features(model, cfg, nm_d)[0]
        fb_ = encode_features(model, cfg, nm2_d)[0]
        voc.synthesize(torch.cat([fa, fb_]), sd2, out=pcm2)
        return fa
    two_batches()
    torch.cuda.synchronize()
    tq0 = time.perf_counter()
    fa = two_batches()
    torch.cuda.synchronize()
    tq1 = time.perf_counter()
    pair_ok = bool(torch.equal(fa, feats)) and bool(torch.equal(pcm2[:B], pcm))  # same utterances, same waveforms
    rep2 = parallel.gather_report(tq1 - tq0, 2 * B * (L * 160 - 17))  # MAX elapsed, SUM samples over the ranks
    two = {"utterances_per_launch": 2 * B, "ms_per_two_batches": (tq1 - tq0) * 1e3,
           "rtf_aggregate_this_rank": 2 * B * (L * 160 - 17) / (tq1 - tq0) / 16000.0,
           "rtf_aggregate": rep2["samples"] / rep2["elapsed_s"] / 16000.0, "utterances_all_ranks": 2 * E2E_PER_GPU * world,
           "first_batch_identical_to_the_per_batch_run": pair_ok,
           "note": "encode batch A, encode batch B, ceps2lpc, ONE decode launch over both (2 x 128 = one workgroup per CU); "
                   "encode_ms / decode_ms above are the per-128 figures of BASELINE config 5's share; four batches per launch "
                   "(k_decode2): four_batches_one_paired_decode"}
    # ---- and with k_decode2 (two utterances per workgroup for batches larger than the CU count): FOUR encode batches, one
    # 512-stream decode launch
    nm4 = [nm_d, nm2_d]
    sd4 = [seeds, seeds2]
    for k in (2, 3):
        x = np.zeros((B, L, 36), np.float32)
        x[:, :, :20] = synth.predictor_features(B, L, utt0=9000 + 1000 * k + lo)
        nm4.append(torch.from_numpy(x).cuda())
        sd4.append(torch.from_numpy(synth.seeds(B, utt0=9000 + 1000 * k + lo).astype(np.int64)).cuda())
    sd4c = torch.cat(sd4)
    pcm4 = torch.empty(4 * B, L * 160, dtype=torch.int16, device="cuda")

    def four_batches():
        fs = [encode_features(model, cfg, x)[0] for x in nm4]
        voc.synthesize(torch.cat(fs), sd4c, out=pcm4)
    four_batches()
    torch.cuda.synchronize()
    tr0 = time.perf_counter()
    four_batches()
    torch.cuda.synchronize()
    tr1 = time.perf_counter()
    rep4 = parallel.gather_report(tr1 - tr0, 4 * B * (L * 160 - 17))
    four = {"utterances_per_launch": 4 * B, "ms_per_four_batches": (tr1 - tr0) * 1e3,
            "streams_per_workgroup": voc.last_streams_per_workgroup(),
            "rtf_aggregate_this_rank": 4 * B * (L * 160 - 17) / (tr1 - tr0) / 16000.0,
            "rtf_aggregate": rep4["samples"] / rep4["elapsed_s"] / 16000.0,
            "first_two_batches_identical_to_the_two_batch_run": bool(torch.equal(pcm4[:2 * B], pcm2)),
            "note": "four encode batches, ONE decode launch over 4 x 128 utterances on k_decode2 (two utterances per workgroup: "
                    "lpcnet_decode2.h) -- the one-GPU form with the highest throughput"}
    del pcm4
    # ---- diagnostic: the encoder of batch k + 1 on a side stream while the vocoder decodes batch k.  The decode holds half of
    # every XCD, so some of the encoder's groups cannot become resident and decide for the row-split fallback (fpcodec.h
    # "Kernel forms": a busy GPU costs speed, never a timeout) -- fallback_groups_per_batch of its 8 groups; the launches of
    # those batches take 10 ms instead of 4 (profiles/r05_kernel_stats.csv).  Kept as a check of that guarantee, not as
    # the schedule to use: the two-batch form above is.
    pipe = None
    try:
        s_enc = torch.cuda.Stream()
        keep, cur, nb, evs = [feats], feats, 3, []
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(nb):
            voc.synthesize(cur, seeds, out=pcm)                      # asynchronous, on the current stream
            with torch.cuda.stream(s_enc):
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                nxt = encode_features(model, cfg, nm_d)[0]
                eb.record()
                evs.append((ea, eb))
            torch.cuda.current_stream().wait_stream(s_enc)
            keep.append(nxt)                                          # (allocated on the side stream: alive until the end)
            cur = nxt
        torch.cuda.synchronize()
        t_pipe = (time.perf_counter() - tp0) / nb * 1e3
        enc_beside = [a.elapsed_time(b) for a, b in evs]
        same = bool(torch.equal(keep[-1], feats))
        del keep
        # how many of the encoder's 8 groups fell back: a second pass that reads the count after every batch (the read
        # synchronises, so this pass is not the one that is timed)
        fb = 0
        for _ in range(nb):
            voc.synthesize(feats, seeds, out=pcm)
            with torch.cuda.stream(s_enc):
                encode_features(model, cfg, nm_d)
                fb += model.fallback_groups()
            torch.cuda.current_stream().wait_stream(s_enc)
        torch.cuda.synchronize()
        pipe = {"batches": nb, "ms_per_batch": t_pipe, "encode_ms_beside_the_decode": enc_beside,
                "fallback_groups_per_batch": fb / nb,
                "note": "diagnostic, not the schedule to use: decode of batch k and encode of batch k + 1 concurrently (two streams, "
                        "one process); encode_ms_beside_the_decode: HIP events on the side stream around the encoder of each batch "
                        "(alone: encode_ms); fallback_groups_per_batch of the encoder's 8 groups, from a second, synchronising pass"}
    except RuntimeError as e:  # (launch / availability errors only: this diagnostic leg must not take the line down)
        pipe, same = {"error": repr(e)[:200]}, True
    if not same:  # a wrong result is not a diagnostic: the run fails
        raise AssertionError("the pipelined encoder's features differ from the sequential run's")
    # receiver side: the same utterances rebuilt from the symbols alone
    enc = model.encoder(cfg, nm_d[:, :, :20], None, cfg["l1"], cfg["l2"], qtz=True, return_indices=True)
    idx = enc[7]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rec = model.decode_indices(cfg, idx, nm_d[:, :, 18:20])
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    assert torch.equal(rec, enc[0]), "decoder output differs from the encoder's reconstruction"
    # ---- the predictor kernels against THEIR roofline (f32 MFMA for the batched GRU rows): HIP events on the
    # launch stream around one fpc_encode / one fpc_predictor_forward call of this share (weights-stationary kernels)
    def ev_ms(fn):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b)
    x20 = nm_d[:, :, :20].contiguous()
    enc_call_ms = ev_ms(lambda: model.encoder(cfg, x20, None, cfg["l1"], cfg["l2"], qtz=True, return_indices=True))
    # ... and around the raw C-ABI calls (no Python wrapper, no histogram download, no status synchronisation inside the events)
    from fpcodec_amd import _lib
    cbh = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg["bl_cb_path"], cfg["bl_scl_cb_path"])
    bufs = [torch.empty(B, L, n, device="cuda") for n in (20, 18, 18, 18, 1, 1)]
    sym = torch.empty(B, L, 4, device="cuda", dtype=torch.int32)
    hist = torch.zeros(cbh.hist_size, device="cuda", dtype=torch.int64)
    s1, s2 = torch.zeros(B, 384, device="cuda"), torch.zeros(B, 128, device="cuda")
    yb = torch.empty(B, L, 18, device="cuda")
    hnd = model._handle()

    def raw_encode():
        _lib.check(_lib.lib().fpc_encode(hnd, cbh.handle, x20.data_ptr(), B, L, float(cfg["l1"]), float(cfg["l2"]), 1,
                                         *[t.data_ptr() for t in bufs], sym.data_ptr(), hist.data_ptr(), None,
                                         _lib.stream_ptr()), "fpc_encode")

    def raw_forward():
        _lib.check(_lib.lib().fpc_predictor_forward(hnd, x20.data_ptr(), B, L, s1.data_ptr(), s2.data_ptr(), yb.data_ptr(),
                                                    _lib.stream_ptr()), "fpc_predictor_forward")
    enc_ms = min(ev_ms(raw_encode) for _ in range(3))
    fwd_ms = min(ev_ms(raw_forward) for _ in range(3))
    model.check()
    # ---- the training step at the reference's batch (train_frame.py:198-204: 100 x 150 frames), SURVEY 8(f) row 4 ----
    from fpcodec_amd.train_frame import Trainer
    tm = Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    tm.load_state_dict(synth.predictor_state_dict())
    tfeat = torch.from_numpy(synth.predictor_features(100, 150, utt0=6000)).cuda()
    tr = Trainer(tm)
    tr.step(tfeat)
    torch.cuda.synchronize()
    tt0 = time.perf_counter()
    for _ in range(5):
        tloss = tr.step(tfeat)
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - tt0) / 5 * 1e3
    coded = float(i2.sum())  # frames whose residual took the 2-stage search (the others: one stage of 512)
    enc_flop = B * L * PRED_FLOP_PER_FRAME + coded * VQ2_FLOP + (B * L - coded) * VQ1_FLOP
    sizes = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg["bl_cb_path"], cfg["bl_scl_cb_path"]).sizes
    # ---- the cross-rank record: max elapsed, total samples, summed histograms (+ flag and frame counts) ----
    flat = np.concatenate([np.asarray(h, np.float64).ravel() for h in cb_tot] +
                          [[float(i1.sum()), float(i2.sum()), float(B * L)]]).astype(np.int64)
    rep = parallel.gather_report(t2 - t0, B * (L * 160 - 17), flat)
    g = rep["hist"]
    offs = np.cumsum([0] + [int(np.asarray(h).size) for h in cb_tot])
    cb_glob = [g[offs[k]:offs[k + 1]].astype(np.float64) for k in range(len(cb_tot))]
    n_frames = float(g[-1])
    keep = [float(g[-3]) / n_frames, float(g[-2]) / n_frames]
    ent = [cal_entropy(h.copy()) if np.sum(h) > 0 else 0.0 for h in cb_glob]
    bits_frame = sum(e * float(np.sum(h)) for e, h in zip(ent, cb_glob)) / n_frames + 2.0  # + the two threshold flags
    out = {"utterances": int(n_frames) // L, "utterances_this_rank": B, "distinct_utterances": int(n_frames) // L, "ranks": world,
           "encode_ms": (t1 - t0) * 1e3, "decode_ms": (t2 - t1) * 1e3,
           "rtf_aggregate": rep["samples"] / rep["elapsed_s"] / 16000.0, "keep_rates": keep,
           "entropy_bits_per_symbol": ent, "bits_per_frame": bits_frame, "bitrate_bps": bits_frame * 100.0,
           "decode_features_ms": (t4 - t3) * 1e3, "two_batches_one_decode": two, "four_batches_one_paired_decode": four, "pipelined": pipe,
           "predictor_roofline": {
               "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F32_TFLOPS, "kernel": "k_encode_wsd",
               "kernel_ms": enc_ms, "call_ms": enc_call_ms, "achieved": enc_flop / (enc_ms * 1e-3) / 1e12,
               "frac": enc_flop / (enc_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
               "forward_kernel": "k_forward_ws", "forward_ms": fwd_ms,
               "forward_achieved": B * L * PRED_FLOP_PER_FRAME / (fwd_ms * 1e-3) / 1e12,
               "forward_frac": B * L * PRED_FLOP_PER_FRAME / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
               "algorithmic_flop": enc_flop,
               "note": "SURVEY 8(d): 1 328 640 FLOP per frame and utterance for the GRU rows and the output layer + the float64 "
                       "searches (2 x 51 flop per entry and target: 1 + 5 scans of 1 024 entries above the threshold, one of 512 "
                       "below it), against the dense f32 MFMA peak; kernel_ms / forward_ms: HIP events on the launch stream around "
                       "one raw C-ABI call (granule memset + kernel + the empty fallback launch + histogram kernel), best of 3; "
                       "call_ms: around Wavernn.encoder (adds the histogram download and the status synchronisation)"},
           "train_step": {"batch": 100, "frames": 150, "step_ms": train_ms, "frames_per_s": 100 * 150 / (train_ms * 1e-3),
                          "loss": float(tloss),
                          "note": "fpc_trainer_step, 5 steps back to back (host clock): forward k_forward_ws<true>, loss, "
                                  "backward k_train_bwd_ws, weight gradients on f32 MFMA, Adam (train_frame.py:53-120)"},
           "gpu_time_s": (t2 - t0) + (t4 - t2) + 2 * (tq1 - tq0) + 2 * (tr1 - tr0) + (enc_call_ms + 6 * (enc_ms + fwd_ms)) * 1e-3 + 6 * train_ms * 1e-3}
    if rank == 0:  # framing figures on rank 0's share (the arithmetic coder is plain Python: a sample of it)
        idx_h = idx.cpu().numpy()
        fixed_bits = bitstream.bits_per_frame(idx_h, sizes)
        models = bitstream.Models(sizes, cb_glob, tuple(keep))
        nsub = min(B, 4)
        packed = [bitstream.entropy_pack(idx_h[k], models) for k in range(nsub)]
        assert all(np.array_equal(bitstream.entropy_unpack(packed[k][0], L, models), idx_h[k]) for k in range(2))
        coded_bits = sum(p[1] for p in packed) / (nsub * L)
        out.update({"fixed_length_bits_per_frame": fixed_bits, "fixed_length_bitrate_bps": fixed_bits * 100.0,
                    "arithmetic_coded_bits_per_frame": coded_bits, "arithmetic_coded_bitrate_bps": coded_bits * 100.0})
    return out


def _spawn(args):
    """`python bench.py --gpus N` outside a launcher: start one rank per GPU as a child process group and relay
    its output (nothing in this process has touched the GPU yet)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--secs", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip BASELINE config 5 (extra key 'e2e', outside the timed region)")
    ap.add_argument("--e2e", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--dump-pcm", default=None, help="directory: every rank saves its PCM block and shard range (tests)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                 f"`python bench.py --gpus N` or torch.distributed.run --nproc-per-node N ... --gpus N")

    import torch
    import fpcodec_amd
    from fpcodec_amd import _lib, parallel
    from fpcodec_amd.ceps2lpc import ceps2lpc_v
    from fpcodec_amd.lpcnet import LPCNet

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    backend = os.environ.get("FPC_BENCH_BACKEND", "nccl")  # "nccl" == RCCL on ROCm; "gloo" for rehearsals
    if world > 1 and backend == "nccl" and world > torch.cuda.device_count():
        # one process per GPU: a measured N-GPU line must come from N distinct GPUs (ranks folded onto fewer devices
        # are a rehearsal: FPC_BENCH_BACKEND=gloo)
        sys.exit(f"bench.py: --gpus {world} under RCCL needs {world} visible GPUs, found {torch.cuda.device_count()}")
    dev = local % ndev  # (the modulo only acts in the gloo rehearsal on fewer GPUs than ranks)
    torch.cuda.set_device(dev)
    _lib.require_gpu()
    dist = None
    # FPC_BENCH_FORCE_DIST=1: initialise the process group even for one rank, so that the RCCL code path of the report
    # (barrier, all_reduce, all_gather_object on device tensors) can be exercised on a one-GPU box
    if world > 1 or os.environ.get("FPC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    synth = fpcodec_amd.synth
    T = args.secs * 100
    # the global utterance list (streams x world) is split in contiguous blocks, one per rank
    lo, hi = parallel.shard_range(args.streams * world, rank, world)
    B = hi - lo
    # synthetic features: a few distinct utterances tiled to B keep host-side generation short without
    # changing the device work (every stream has its own seed = its global utterance number)
    nuniq = min(B, 16)
    base = synth.vocoder_features_raw(nuniq, T, utt0=0)
    feats = torch.from_numpy(np.stack([base[(lo + k) % nuniq] for k in range(B)])).cuda()
    lpc = ceps2lpc_v(feats.reshape(-1, 36)[:, :20].contiguous())[1]
    feats[:, :, 20:] = lpc.reshape(B, T, 16)
    seeds = torch.from_numpy(synth.seeds(B, utt0=lo).astype(np.int64)).cuda()
    voiced_fraction = float((1.5 * feats[:, :, 19] - 0.5 > 0).float().mean().item())
    voc = LPCNet(synth.lpcnet_weights())
    pcm = torch.empty(B, T * 160, dtype=torch.int16, device="cuda")

    def step():
        voc.synthesize(feats, seeds, out=pcm)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec_ms = []
    for _ in range(args.steps):
        step()
        dec_ms.append(voc.last_decode_ms())  # HIP events around the decode kernel on its stream
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    samples_step = B * (T * 160 - 17)
    rep = parallel.gather_report(dt_local, samples_step * args.steps)  # MAX elapsed, SUM samples over ranks
    dt, total = rep["elapsed_s"], rep["samples"]
    props = torch.cuda.get_device_properties(dev)
    ranks = parallel.gather_records({  # who ran what where: a SCALE record shows N distinct GPUs by itself
        "rank": rank, "device_index": dev, "device_name": props.name,
        "pci_bus_id": _pci_bus_id(dev), "uuid": str(getattr(props, "uuid", "")),
        "utterances": [int(lo), int(hi)], "samples": int(samples_step * args.steps), "elapsed_s": float(dt_local),
        "decode_ms": float(np.mean(dec_ms))})
    value = total / dt
    dec_s = float(np.mean(dec_ms)) / 1e3
    dec_rate = samples_step / dec_s  # per GPU, decode kernel only

    if args.dump_pcm:
        os.makedirs(args.dump_pcm, exist_ok=True)
        np.savez(os.path.join(args.dump_pcm, f"rank{rank}.npz"), pcm=pcm.cpu().numpy(), lo=lo, hi=hi)

    # HBM bytes per k_decode launch: PMC counters cannot be read inside this process, so the figure comes from the
    # committed PMC passes (tools/traffic_round.sh) -- but only from a record taken on THIS kernel (hash of the decode
    # kernel's sources) and this workload; a stale record reads null, never a silently outdated number
    traffic = None
    traffic_src = None
    khash = decode_kernel_hash()
    for name in sorted((n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_traffic.json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if tj.get("kernel_source_sha256") == khash and B == 256 and T == 300:
            traffic, traffic_src = tj["hbm_bytes_per_launch"], name
            break
    if traffic is None:
        traffic_src = f"none matches kernel sources {khash[:12]}"
    out = {
        "metric": "LPCNet synthesis samples/sec (16 kHz RTF) per GPU; 1/2/4/8-GPU throughput",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 3/4: {args.streams} independent {args.secs} s utterances per GPU "
                               f"(T={T} frames, {T * 160} samples each), GRU_A=384 block-sparse, fixed Philox RNG",
                   "streams_per_gpu": args.streams, "frames": T, "voiced_fraction": voiced_fraction,
                   "voiced_note": "fraction of frames with pdf sharpening (1.5*pitch_corr-0.5 > 0); SURVEY 8(d) "
                                  "specifies pitch-corr U(-.4,.4); voiced frames cost more: see voiced_50"},
        "rtf_aggregate": value / 16000.0,
        "rtf_per_stream": dec_rate / B / 16000.0,
        "ranks": ranks,
        "distinct_gpus": len({(r["pci_bus_id"], r["uuid"]) for r in ranks}),
        "roofline": {
            "bound": "valu_f32",
            "note": "per-stream latency-bound recurrence; ceiling = FP32 vector rate (157.3 TF = f32 MFMA dense peak); "
                    "not HBM, not MFMA (SURVEY 8d)",
            "kernel": "k_decode", "achieved": dec_rate * FLOP_PER_SAMPLE / 1e12, "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s", "frac": dec_rate * FLOP_PER_SAMPLE / 1e12 / PEAK_F32_TFLOPS,
            "traffic": traffic,
            "traffic_unit": f"bytes/launch (rocprofv3 PMC FETCH_SIZE+WRITE_SIZE, profiles/{traffic_src})",
            "launch_ms": dec_s * 1e3, "cycles_per_sample_at_2p4GHz": dec_s * 2.4e9 / (T * 160 - 17),
            "hbm": {"achieved": dec_rate * HBM_BYTES_PER_SAMPLE / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": dec_rate * HBM_BYTES_PER_SAMPLE / 1e9 / PEAK_HBM_GBS},
            # waves per SIMD, CUs busy, issue shares per sample, the issue floor (north_star: "occupancy against gfx950 peak")
            "occupancy": counters_record("k_decode", khash),
        },
    }
    e2e = None
    if not args.no_e2e:
        e2e = e2e_config5(voc, torch, synth, parallel, rank, world)  # every rank: holds the report's collectives
    if rank == 0:
        # single-stream latency view (BASELINE config 2), outside the timed region
        one = torch.empty(1, T * 160, dtype=torch.int16, device="cuda")
        voc.synthesize(feats[:1], seeds[:1], out=one)
        torch.cuda.synchronize()
        voc.synthesize(feats[:1], seeds[:1], out=one)
        ms1 = voc.last_decode_ms()
        out["single_stream"] = {"decode_ms": ms1, "samples_per_s": (T * 160 - 17) / (ms1 / 1e3),
                                "rtf": (T * 160 - 17) / (ms1 / 1e3) / 16000.0}
        # the same batch with every second frame voiced (pdf sharpening + fifth barrier)
        fv = feats.clone()
        fv[:, 0::2, 19] = 0.9
        voc.synthesize(fv, seeds, out=pcm)
        torch.cuda.synchronize()
        voc.synthesize(fv, seeds, out=pcm)
        msv = voc.last_decode_ms()
        out["voiced_50"] = {"voiced_fraction": float((1.5 * fv[:, :, 19] - 0.5 > 0).float().mean().item()),
                            "decode_ms": msv, "samples_per_s": samples_step / (msv / 1e3),
                            "rtf_per_stream": samples_step / (msv / 1e3) / B / 16000.0}
        # ---- more utterances than compute units: k_decode2 walks two utterances through each workgroup (lpcnet_decode2.h);
        # beside it the same batch as rounds of k_decode (fpc_lpcnet_set_pairing(-1)).  Same PCM (checked here by hash).
        many, many_s = [], 0.0
        for Bm in (512, 1024):
            fm = feats.repeat((Bm + B - 1) // B, 1, 1)[:Bm].contiguous()
            sm = torch.from_numpy(synth.seeds(Bm, utt0=20000).astype(np.int64)).cuda()
            pm = torch.empty(Bm, T * 160, dtype=torch.int16, device="cuda")
            rec = {"streams": Bm}
            for mode, key in ((0, "two_per_workgroup"), (-1, "one_per_workgroup")):
                voc.set_pairing(mode)
                voc.synthesize(fm, sm, out=pm)
                torch.cuda.synchronize()
                voc.synthesize(fm, sm, out=pm)
                ms = voc.last_decode_ms()
                many_s += 2 * ms / 1e3
                n = Bm * (T * 160 - 17)
                rec[key] = {"decode_ms": ms, "samples_per_s": n / (ms / 1e3), "streams_per_workgroup": voc.last_streams_per_workgroup(),
                            "frac": n / (ms / 1e3) * FLOP_PER_SAMPLE / 1e12 / PEAK_F32_TFLOPS,
                            "pcm_sha1": __import__("hashlib").sha1(pm.cpu().numpy().tobytes()).hexdigest()[:16]}
            voc.set_pairing(0)
            rec["ratio"] = rec["one_per_workgroup"]["decode_ms"] / rec["two_per_workgroup"]["decode_ms"]
            rec["same_pcm"] = rec["one_per_workgroup"]["pcm_sha1"] == rec["two_per_workgroup"]["pcm_sha1"]
            many.append(rec)
            del fm, pm
        out["many_stream"] = {"kernel": "k_decode2", "cases": many, "occupancy": counters_record("k_decode2", decode2_kernel_hash()),
                              "note": "B > compute units on ONE GPU (configs 4 / 5 on fewer than 8 GPUs): decode launch by HIP "
                                      "events; frac = samples/s x 145 600 FLOP / 157.3 TF as in `roofline`; one_per_workgroup = "
                                      "the grid of B workgroups of k_decode running in rounds"}
        if not all(c["same_pcm"] for c in many):
            raise AssertionError("k_decode2 and k_decode disagree on the PCM of the many_stream leg")
        if e2e is not None:
            out["e2e"] = e2e
            e2e["predictor_roofline"]["occupancy"] = counters_record("k_encode_wsd", predictor_kernel_hash())
        # how long the GPU worked inside this process (host clock around synchronised sections; the rest of the run is input
        # synthesis and, at --gpus 1, the CPU baseline): timed steps + warm-up + the e2e legs + the two latency views
        out["gpu_time_s"] = float(dt_local) + args.warmup * dec_s + (e2e["gpu_time_s"] if e2e is not None else 0.0) + \
            2 * ms1 / 1e3 + 2 * msv / 1e3 + many_s
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
