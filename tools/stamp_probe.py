import sys, os; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from fpcodec_amd.ceps2lpc import ceps2lpc_v
synth=fpcodec_amd.synth
B=int(sys.argv[1]); T=100
f=torch.from_numpy(synth.vocoder_features_raw(B,T)).cuda()
f[:,:,20:]=ceps2lpc_v(f.reshape(-1,36)[:,:20].contiguous())[1].reshape(B,T,16)
voc=LPCNet(synth.lpcnet_weights())
sd=synth.seeds(B)
voc.synthesize(f,sd); torch.cuda.synchronize()
voc.synthesize(f,sd); torch.cuda.synchronize()
print("decode ms", voc.last_decode_ms(), "cycles/sample@2.4GHz", voc.last_decode_ms()*1e-3*2.4e9/(T*160-17))
