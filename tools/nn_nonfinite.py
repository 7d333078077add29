import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from genpc_amd import chamfer_3D, _lib
from oracle import oracle
rng = np.random.default_rng(8)
a = rng.random((1,700,3),dtype=np.float32)-np.float32(0.5); b = rng.random((1,900,3),dtype=np.float32)-np.float32(0.5)
a[0,13]=np.nan; b[0,5,1]=np.nan; b[0,77]=np.inf
exp = oracle.chamfer_forward(a,b,1)
for path in range(4):
    _lib.lib.genpc_nn_tune(path,0)
    A=torch.from_numpy(a).cuda(); B=torch.from_numpy(b).cuda()
    d1=torch.zeros(1,700,device="cuda"); d2=torch.zeros(1,900,device="cuda"); i1=torch.zeros(1,700,dtype=torch.int32,device="cuda"); i2=torch.zeros(1,900,dtype=torch.int32,device="cuda")
    chamfer_3D.forward(A,B,d1,d2,i1,i2); torch.cuda.synchronize()
    got=[d1.cpu().numpy(),d2.cpu().numpy(),i1.cpu().numpy(),i2.cpu().numpy()]
    for nm,g,e in zip(("d1","d2","i1","i2"),got,exp):
        bad=np.argwhere(~((g==e)|(np.isnan(g)&np.isnan(e))))
        if len(bad): print("path",path,nm,"bad at",bad[:5].tolist(),"got",g[tuple(bad[0])],"exp",e[tuple(bad[0])])
print("done")
