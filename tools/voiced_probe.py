import sys; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from fpcodec_amd.ceps2lpc import ceps2lpc_v
synth=fpcodec_amd.synth
B=int(sys.argv[1]); T=100
voc=LPCNet(synth.lpcnet_weights()); sd=synth.seeds(B)
for frac in (0.0, 0.5, 1.0):
    f=synth.vocoder_features_raw(B,T)
    rng=np.random.default_rng(5); v=rng.random((B,T))<frac
    f[:,:,19]=np.where(v, 0.8, -0.2)   # pitch correlation: voiced frames sharpen the pdf (1.5*0.8-0.5 = 0.7)
    ft=torch.from_numpy(f).cuda(); ft[:,:,20:]=ceps2lpc_v(ft.reshape(-1,36)[:,:20].contiguous())[1].reshape(B,T,16)
    voc.synthesize(ft,sd); torch.cuda.synchronize(); voc.synthesize(ft,sd); torch.cuda.synchronize()
    ms=voc.last_decode_ms(); print(f"voiced fraction {frac}: decode {ms:.2f} ms, {ms*1e-3*2.4e9/(T*160-17):.0f} cycles/sample, {(T*160-17)/(ms*1e-3)/16000:.1f}x RT per stream")
