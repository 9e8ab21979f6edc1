import sys; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from oracle import oracle as O
synth=fpcodec_amd.synth
w=synth.lpcnet_weights(); voc=LPCNet(w); orc=O.LPCNet(w)
T=1500
f=synth.vocoder_features_raw(2,T,utt0=7)
f[:,:,20:]=O.ceps2lpc(f.reshape(-1,36)[:,:20])[0].reshape(2,T,16)
sd=synth.seeds(2,utt0=7)
pcm=voc.synthesize(f,sd).cpu().numpy()
for b in range(2):
    ref=orc.synthesize(f[b],int(sd[b]))
    nz=np.nonzero(pcm[b]!=ref)[0]
    print("utt",b,"T",T,"mismatches",nz.size, "first", nz[:3], "pcm range", pcm[b].min(), pcm[b].max())
