import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, time, torch
from conftest import load_golden
from oracle import oracle
from tricolour_amd import flagging
d,_=load_golden("G11_uvcontsub.npz")
for name in ("a","b"):
    kw={k[len("kw_%s_"%name):]: d[k].tolist() for k in d.files if k.startswith("kw_%s_"%name)}
    got=flagging.uvcontsub_flagger(d["vis"], d["flags"], **kw)
    exp=d["out_"+name]
    print(name, "differ", int((got!=exp).sum()), "of", exp.size)
g=torch.Generator(device="cuda"); g.manual_seed(1)
shape=(16,4,1024,4096)
vis=torch.complex(torch.randn(shape,generator=g,device="cuda")+3, torch.randn(shape,generator=g,device="cuda"))
fl=torch.zeros(shape,dtype=torch.bool,device="cuda")
kw=dict(major_cycles=7, or_original_from_cycle=1, taylor_degrees=20, sigma=15.0)
out=flagging.uvcontsub_flagger(vis,fl,**kw); torch.cuda.synchronize()
t=time.time(); out=flagging.uvcontsub_flagger(vis,fl,**kw); torch.cuda.synchronize(); dt=time.time()-t
print("uvcontsub 64 windows 7 cycles: %.1f ms -> %.0f Mvis/s" % (dt*1e3, vis.numel()/dt/1e6))
