import sys; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd, concurrent.futures as cf
from fpcodec_amd.lpcnet import LPCNet
from oracle import oracle as O
synth=fpcodec_amd.synth
bad=0
for wseed, dens in ((1004,(0.05,0.05,0.2)), (77,(0.03,0.06,0.18)), (5,(0.02,0.02,0.1))):
    w=synth.lpcnet_weights(seed=wseed, density=dens)
    voc=LPCNet(w); orc=O.LPCNet(w)
    B,T=48,40
    f=synth.vocoder_features_raw(B,T,utt0=wseed*10)
    f[:, :, 19] = np.random.default_rng(wseed).uniform(-0.5, 1.0, (B, T))   # a wide mix of voiced / unvoiced frames
    f[:,:,20:]=O.ceps2lpc(f.reshape(-1,36)[:,:20])[0].reshape(B,T,16)
    sd=np.random.default_rng(wseed+1).integers(0,2**62,B).astype(np.uint64)
    pcm=voc.synthesize(f,sd).cpu().numpy()
    with cf.ThreadPoolExecutor(16) as ex:
        refs=list(ex.map(lambda b: orc.synthesize(f[b], int(sd[b])), range(B)))
    nb=sum(0 if np.array_equal(pcm[b],refs[b]) else 1 for b in range(B))
    print("weights", wseed, dens, "variant", voc.kernel_variant(), "mismatching utterances:", nb, "of", B, flush=True)
    bad+=nb
print("TOTAL MISMATCH", bad)
