#!/usr/bin/env python3
"""Benchmark of the hot path: LPCNet synthesis samples/s on synthetic 3-second utterances.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the vocoder (frame-rate conditioning kernels + persistent decode kernel)
over one batch of --streams independent 3 s utterances per GPU (BASELINE config 3: 256
utterances on one MI355X; with N GPUs every rank decodes its own 256: config 4, weak scaling,
no collective in the data path - RCCL only gathers the timings).  Inputs (features, seeds,
weights) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 145600.0      # 2 x 72 797 MAC per output sample per stream
HBM_BYTES_PER_SAMPLE = 2.9      # 2 B PCM out + 144 B features / 160 samples
PEAK_F32_TFLOPS = 157.3         # MI355X_MICROARCH.md: FP32 vector == f32 MFMA dense peak
PEAK_HBM_GBS = 8000.0


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
    while time.time() - t0 < 5.0:
        orc.synthesize(f, 1004 + reps1)
        n1 += per
        reps1 += 1
    single = n1 / (time.time() - t0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU share (cgroup v2 quota) rather than the host's thread count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    cores = min(cores, 16)  # one GPU's share of the host on the measurement boxes

    def worker(k):
        n, t1 = 0, time.time()
        while time.time() - t1 < 8.0:
            orc.synthesize(f, 2000 + 100 * k + n)
            n += 1
        return n

    t0 = time.time()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(worker, range(cores)))
    dt = time.time() - t0
    return {"value": done * per / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "single_thread_value": single,
            "sample": f"{done} x one {frames}-frame utterance ({done * per} samples) through oracle/fpc_oracle.c "
                      f"(orc_lpcnet_synthesize), one utterance at a time on each of {cores} host threads for 8 s; "
                      f"single_thread_value from {reps1} utterances on one thread"}


def e2e_config5(voc, torch, synth, B=128, L=300):
    """BASELINE config 5, one GPU's share: encode (GRU predictor + thresholds + scalar/2-stage VQ) ->
    x24.1 -> ceps2lpc -> LPCNet decode for B utterances; bitrate from codebook-usage entropies
    (src/generate_qtz_features.py:94-101,202)."""
    import tempfile
    from fpcodec_amd.synthesis_qtz import encode_features
    from fpcodec_amd.vq_func import cal_entropy
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
    nu = 8
    nm = np.zeros((B, L, 36), np.float32)
    nm[:, :, :20] = np.tile(synth.predictor_features(nu, L, utt0=5000), (B // nu + 1, 1, 1))[:B]
    nm_d = torch.from_numpy(nm).cuda()
    seeds = torch.from_numpy(synth.seeds(B, utt0=5000).astype(np.int64)).cuda()
    pcm = torch.empty(B, L * 160, dtype=torch.int16, device="cuda")

    def run():
        feats, r, i1, i2, cb_tot = encode_features(model, cfg, nm_d)
        voc.synthesize(feats, seeds, out=pcm)
        return i1, i2, cb_tot

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats, r, i1, i2, cb_tot = encode_features(model, cfg, nm_d)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    voc.synthesize(feats, seeds, out=pcm)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # receiver side: the same utterances rebuilt from the symbols alone
    from fpcodec_amd import bitstream
    from fpcodec_amd.vq_func import load_codebooks
    enc = model.encoder(cfg, nm_d[:, :, :20], None, cfg["l1"], cfg["l2"], qtz=True, return_indices=True)
    idx = enc[7]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rec = model.decode_indices(cfg, idx, nm_d[:, :, 18:20])
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    assert torch.equal(rec, enc[0]), "decoder output differs from the encoder's reconstruction"
    sizes = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg["bl_cb_path"], cfg["bl_scl_cb_path"]).sizes
    fixed_bits = bitstream.bits_per_frame(idx.cpu().numpy(), sizes)
    idx_h = idx.cpu().numpy()
    models = bitstream.Models(sizes, cb_tot, (float(i1.mean()), float(i2.mean())))
    nsub = min(B, 8)  # the arithmetic coder is plain Python: a sample of the utterances
    coded_bits = sum(bitstream.entropy_pack(idx_h[k], models)[1] for k in range(nsub)) / (nsub * L)
    assert all(np.array_equal(bitstream.entropy_unpack(bitstream.entropy_pack(idx_h[k], models)[0], L, models), idx_h[k])
               for k in range(2))
    n = B * L
    ent = [cal_entropy(h) if np.sum(h) > 0 else 0.0 for h in cb_tot]
    bits_frame = sum(e * float(np.sum(h)) for e, h in zip(ent, cb_tot)) / n + 2.0  # + the two threshold flags
    return {"utterances": B, "encode_ms": (t1 - t0) * 1e3, "decode_ms": (t2 - t1) * 1e3,
            "rtf_aggregate": B * 3.0 / (t2 - t0), "keep_rates": [float(i1.mean()), float(i2.mean())],
            "entropy_bits_per_symbol": ent, "bits_per_frame": bits_frame, "bitrate_bps": bits_frame * 100.0,
            "fixed_length_bits_per_frame": fixed_bits, "fixed_length_bitrate_bps": fixed_bits * 100.0,
            "arithmetic_coded_bits_per_frame": coded_bits, "arithmetic_coded_bitrate_bps": coded_bits * 100.0,
            "decode_features_ms": (t4 - t3) * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--secs", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="also time BASELINE config 5 per GPU (predictor + residual VQ encode -> ceps2lpc -> decode, "
                         "128 utterances) and report bitrate; extra key 'e2e', outside the timed region")
    args = ap.parse_args()

    import torch
    import fpcodec_amd
    from fpcodec_amd import _lib
    from fpcodec_amd.ceps2lpc import ceps2lpc_v
    from fpcodec_amd.lpcnet import LPCNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev = local % ndev  # one process per GPU; the modulo only matters for single-GPU rehearsals
    torch.cuda.set_device(dev)
    _lib.require_gpu()
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("FPC_BENCH_BACKEND", "nccl")  # "nccl" == RCCL on ROCm; "gloo" for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    synth = fpcodec_amd.synth
    B, T = args.streams, args.secs * 100
    # synthetic features: distinct utterances per rank; a few distinct ones tiled to B keep
    # host-side generation short without changing the device work (every stream has its own seed)
    nuniq = min(B, 16)
    base = synth.vocoder_features_raw(nuniq, T, utt0=rank * 100000)
    feats = torch.from_numpy(np.tile(base, ((B + nuniq - 1) // nuniq, 1, 1))[:B].copy()).cuda()
    lpc = ceps2lpc_v(feats.reshape(-1, 36)[:, :20].contiguous())[1]
    feats[:, :, 20:] = lpc.reshape(B, T, 16)
    seeds = torch.from_numpy(synth.seeds(B, utt0=rank * 100000).astype(np.int64)).cuda()
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
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    samples_step = B * (T * 160 - 17)
    total = samples_step * args.steps * world
    value = total / dt
    dec_s = float(np.mean(dec_ms)) / 1e3
    dec_rate = samples_step / dec_s  # per GPU, decode kernel only

    traffic = None  # HBM bytes per k_decode launch from the committed PMC passes (same workload only)
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if B == 256 and T == 300:
            traffic = tj["hbm_bytes_per_launch"]
    except Exception:
        pass
    out = {
        "metric": "LPCNet synthesis samples/sec (16 kHz RTF) per GPU; 1/2/4/8-GPU throughput",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 3/4: {B} independent {args.secs} s utterances per GPU "
                               f"(T={T} frames, {T * 160} samples each), GRU_A=384 block-sparse, fixed Philox RNG",
                   "streams_per_gpu": B, "frames": T},
        "rtf_aggregate": value / 16000.0,
        "rtf_per_stream": dec_rate / B / 16000.0,
        "roofline": {
            "bound": "mfma", "note": "f32 VALU sparse mat-vec; f32 vector peak == f32 MFMA dense peak (157.3 TF)",
            "kernel": "k_decode", "achieved": dec_rate * FLOP_PER_SAMPLE / 1e12, "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s", "frac": dec_rate * FLOP_PER_SAMPLE / 1e12 / PEAK_F32_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC FETCH_SIZE+WRITE_SIZE, profiles/r01_traffic.json)",
            "launch_ms": dec_s * 1e3,
            "hbm": {"achieved": dec_rate * HBM_BYTES_PER_SAMPLE / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": dec_rate * HBM_BYTES_PER_SAMPLE / 1e9 / PEAK_HBM_GBS},
        },
    }
    if rank == 0:
        # single-stream latency view (BASELINE config 2), outside the timed region
        one = torch.empty(1, T * 160, dtype=torch.int16, device="cuda")
        voc.synthesize(feats[:1], seeds[:1], out=one)
        torch.cuda.synchronize()
        voc.synthesize(feats[:1], seeds[:1], out=one)
        ms1 = voc.last_decode_ms()
        out["single_stream"] = {"decode_ms": ms1, "samples_per_s": (T * 160 - 17) / (ms1 / 1e3),
                                "rtf": (T * 160 - 17) / (ms1 / 1e3) / 16000.0}
        if args.e2e:
            out["e2e"] = e2e_config5(voc, torch, synth)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
