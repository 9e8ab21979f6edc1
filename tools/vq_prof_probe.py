import sys, os, tempfile; sys.path.insert(0,'.')
import numpy as np, fpcodec_amd
from fpcodec_amd import vq_func
synth=fpcodec_amd.synth
d=tempfile.mkdtemp(); c=synth.codebooks(); p={}
for k,v in c.items():
    p[k]=os.path.join(d,k+'.npy'); np.save(p[k],v)
r=synth.cb_training_vectors(256*8, seed_offset=3)*np.float32(0.3)
q,_=vq_func.vq_quantize(r,p['vq_hi'])
q,_=vq_func.vq_quantize(r,p['vq_hi'])
print("2-stage phase stamps (cycles from start; median over WGs): dist1 | select1 | targets | - | dist5 | select5 | merge | end")
print(np.median(q[:, :8],0))
q,_=vq_func.vq_quantize(r,p['vq_lo'])
print("1-stage 512:", np.median(q[:, :8],0))
