"""fpc_lpcnet_synthesize at the benchmark's size (256 x 300 frames) in one pass and in chunks: time of the whole call (HIP
events around it: frame-rate launches + sample loops), workspace, PCM hash.   python tools/chunk_bench.py [chunk ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from fpcodec_amd.ceps2lpc import ceps2lpc_v
synth = fpcodec_amd.synth
B, T = 256, 300
f = torch.from_numpy(synth.vocoder_features_raw(B, T)).cuda()
f[:, :, 20:] = ceps2lpc_v(f.reshape(-1, 36)[:, :20].contiguous())[1].reshape(B, T, 16)
voc = LPCNet(synth.lpcnet_weights())
sd = synth.seeds(B)
chunks = [int(a) for a in sys.argv[1:]] or [0, 25, 50, 100, 150, 0]
for c in chunks:
    voc.set_chunk_frames(c)
    ms = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); pcm = voc.synthesize(f, sd); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    h = hashlib.sha1(pcm.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"chunk {c:4d}: call {min(ms[1:]):8.3f} ms  decode span {voc.last_decode_ms():8.3f} ms  workspace "
          f"{voc.workspace_bytes(B, T) / 1e6:7.1f} MB  pcm {h}", flush=True)
