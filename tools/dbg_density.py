import sys; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from oracle import oracle as O
synth=fpcodec_amd.synth
for dens in [(0.01,0.01,0.01),(0.04,0.04,0.04),(0.05,0.05,0.2)]:
    w=synth.lpcnet_weights(density=dens)
    f=synth.vocoder_features_raw(1,2); f[:,:,20:]=O.ceps2lpc(f.reshape(-1,36)[:,:20])[0].reshape(1,2,16)
    pcm=LPCNet(w).synthesize(f,synth.seeds(1)).cpu().numpy()[0]
    orc=O.LPCNet(w); ref=orc.synthesize(f[0],int(synth.seeds(1)[0]))
    nz=np.nonzero(pcm!=ref)[0]
    print(dens,'blocks',orc.nblocks,'mismatches',nz.size, nz[:4])
