"""Developer tool: latency kernel vs throughput kernel over batch sizes (sets WINO_3X3_ALGO per
subprocess because the library reads it once)."""
import os, subprocess, sys
code = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
for C in (128, 256):
    w=(torch.rand(C,C,3,3)-0.5).to(dev); s=(torch.rand(C)-0.5).to(dev); b=(torch.rand(C)-0.5).to(dev)
    U=pkg.filter_transform_f2(w)
    for N in (1,2,3,4,6,8,12,16,24,32,48,64,96):
        x=(torch.rand(N,16,16,C)-0.5).to(dev); out=torch.empty(N,16,16,C,device=dev)
        for _ in range(10): pkg.conv3x3_bn_relu(x,U,b,s,out=out)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): pkg.conv3x3_bn_relu(x,U,b,s,out=out)
        e1.record(); torch.cuda.synchronize()
        print(os.environ["WINO_3X3_ALGO"], C, N, "%.1f us" % (e0.elapsed_time(e1)*10))
'''
for algo in ("big", "small"):
    env = dict(os.environ, WINO_3X3_ALGO=algo)
    subprocess.run([sys.executable, "-c", code], env=env)
