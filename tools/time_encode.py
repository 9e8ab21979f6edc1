import sys, os, time, tempfile; sys.path.insert(0,'.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
from fpcodec_amd.ceps2lpc import ceps2lpc_v
synth=fpcodec_amd.synth
d=tempfile.mkdtemp(); c=synth.codebooks(); p={}
for k,v in c.items():
    p[k]=os.path.join(d,k+'.npy'); np.save(p[k],v)
cfg=dict(scl_cb_path=p['scl_hi'],cb_path=p['vq_hi'],bl_scl_cb_path=p['scl_lo'],bl_cb_path=p['vq_lo'])
m=Wavernn(20,384,128,18); m.load_state_dict(synth.predictor_state_dict())
for B in (1,16,128,256):
    f=torch.from_numpy(np.tile(synth.predictor_features(min(B,8),300),( (B+7)//8,1,1))[:B].copy()).cuda()
    m.encoder(cfg,f,None,0.09,0.28); torch.cuda.synchronize()
    t=time.perf_counter(); out=m.encoder(cfg,f,None,0.09,0.28); torch.cuda.synchronize(); dt=time.perf_counter()-t
    t=time.perf_counter(); e,lpc,rc=ceps2lpc_v((out[0]*24.1).reshape(-1,20)); torch.cuda.synchronize(); dt2=time.perf_counter()-t
    t=time.perf_counter(); y=m.forward(f); torch.cuda.synchronize(); dt3=time.perf_counter()-t
    print(f"B={B}: encode {dt*1e3:.2f} ms ({B*300/dt:.0f} frames/s, {B*3/dt:.1f}x RT aggregate)  ceps2lpc {dt2*1e3:.2f} ms  forward {dt3*1e3:.2f} ms")
