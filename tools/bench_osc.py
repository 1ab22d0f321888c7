import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "alive-vc_amd"))
from module import ops
dev="cuda"; N,H,Lf=128,64,450
g=torch.Generator(device=dev).manual_seed(3)
amps=torch.rand(N,H,Lf,device=dev,generator=g); f0=100+100*torch.rand(N,1,Lf,device=dev,generator=g)
def run(): return ops.oscillator(amps, f0)
for _ in range(3): o=run()
a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): o=run()
e.record(); torch.cuda.synchronize()
print("oscillator: %.3f ms per 128 windows; checksum %.6f" % (a.elapsed_time(e)/10, float(o[0].double().abs().sum())))
