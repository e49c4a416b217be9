import time
t=time.time()
from ctrlhair_amd import procedural as P
t=time.time(); sd=P.sean_state_dict(0,64); print('sean_state_dict ngf64 %.1f s'%(time.time()-t))
t=time.time(); sd=P.sean_state_dict(0,16); print('ngf16 %.1f s'%(time.time()-t))
from ctrlhair_amd.hair_editor import procedural_weights
t=time.time(); w=procedural_weights(0,16); print('procedural_weights 16 %.1f s'%(time.time()-t))
import torch
from ctrlhair_amd.sean.generator import SeanGenerator
sd=P.sean_state_dict(0,64)
for f in (0,1):
    t=time.time(); g=SeanGenerator(0,f16x3=f).load_state_dict(sd,max_batch=2,max_size=256); torch.cuda.synchronize(); print('build f16x3=%d %.1f s'%(f,time.time()-t)); g.handle.close()
