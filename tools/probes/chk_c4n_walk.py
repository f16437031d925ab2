import sys, os, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"memc-net_amd"))
import my_package._ext.my_lib as L
from tools import measure as M  # noqa: E402
M.use()                             # the measurement build: ablation / A-B arms live only there
from tools import synth
dev=torch.device("cuda:0")
for (B,C,H,W) in ((2,64,720,1280),(1,8,100,200),(3,12,64,72)):
    t=synth.torch_inputs(dev,B,C,H,W,flow_kind="smooth")
    outs=[]
    for v in (-1,30,31):
        M.set_variant("fi_fwd",v)
        o=torch.full_like(t["x"],float("nan")); assert L.FilterInterpolationLayer_gpu_forward(t["x"],t["flow"],t["filt"],o)==0
        outs.append(o)
    M.set_variant("fi_fwd",-1)
    print((B,C,H,W),"equal:",torch.equal(outs[0],outs[1]),torch.equal(outs[0],outs[2]), "nan:", bool(torch.isnan(outs[1]).any()))
