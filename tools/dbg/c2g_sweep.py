import sys, torch
sys.path.insert(0, '.')
from tools.bench_paths import WINDOWS, gpu_time
from nnmnkwii_amd import _hip
B,T,sd=256,1000,60
for dt in (torch.float64, torch.float32):
    m=torch.randn(B,T,3*sd,dtype=dt,device='cuda'); vg=torch.rand(3*sd,dtype=dt,device='cuda')+0.1
    pw=_hip.prepack_windows(WINDOWS)
    for name,v in (("global",vg),("unit",None)):
        r=[]
        for algo in (2,3):
            r.append(gpu_time(lambda: _hip.forward(m,v,pw,algo=algo,want_status=False),steps=20,warmup=3))
        print(str(dt)[6:], name, "wave %.4f strip %.4f"%tuple(r))
