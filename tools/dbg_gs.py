import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bayesian_inference_for_nn_amd import engine as eng
os.environ["PYZ_SVGD_GS_FUSED"]="1"; os.environ["PYZ_SVGD_GS_ZIGZAG"]="0"
def run(res, dims, acts, M, parts, x, y, steps):
    os.environ["PYZ_SVGD_GS_RESIDENT"]="1" if res else "0"
    spec=eng.MLPSpec(dims, acts, "scce"); D=spec.n_params
    plan=eng.MLPPlan(spec, max_batch=len(x), max_particles=M)
    p=torch.as_tensor(parts).cuda(); am=torch.zeros((M,D),device="cuda"); av=torch.zeros((M,D),device="cuda"); loss=torch.zeros(1,device="cuda")
    xd=torch.as_tensor(x).cuda(); yd=torch.as_tensor(y).cuda()
    for t in range(1,steps+1):
        plan.svgd_step(p,p,0,am,av,xd,yd,1e-3,1.0,t,loss,sweep="gauss_seidel")
    torch.cuda.synchronize()
    return p.cpu().numpy(), am.cpu().numpy(), av.cpu().numpy()
for dims, acts, M, sc in (((5,7,3),("tanh","softmax"),5,0.1), ((64,40,24,10),("relu","relu","softmax"),7,0.015)):
    spec=eng.MLPSpec(dims, acts, "scce"); D=spec.n_params
    rng=np.random.default_rng(7); n=90
    x=rng.normal(size=(n,dims[0])).astype(np.float32); y=rng.integers(0,dims[-1],size=n).astype(np.int32)
    parts=(rng.normal(size=(M,D))*sc).astype(np.float32)
    for steps in (1,2):
        a=run(True,dims,acts,M,parts,x,y,steps); b=run(False,dims,acts,M,parts,x,y,steps)
        for nm,u,v in zip(("p","m","v"),a,b):
            d=np.argwhere(u!=v)
            print(dims, "steps",steps,nm,"ndiff",len(d), "first",d[:6].tolist(), "rows", sorted(set(d[:,0].tolist()))[:10])
