"""Developer tool: per-launch HIP-event timings (min/median/max) next to wall clock."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
def run(name, step, n=60):
    for _ in range(10): step()
    torch.cuda.synchronize()
    for rep in range(2):
        evs=[torch.cuda.Event(enable_timing=True) for _ in range(n+1)]
        t0=time.perf_counter(); evs[0].record()
        for i in range(n):
            step(); evs[i+1].record()
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        d=sorted(evs[i].elapsed_time(evs[i+1])*1e3 for i in range(n))
        print(f"{name:18s} host-issue {((t1-t0)/n*1e6):6.1f} wall {((t2-t0)/n*1e6):7.1f} ev-total/n {(evs[0].elapsed_time(evs[n])*1e3/n):7.1f} per-launch min/med/max {d[0]:7.1f} {d[n//2]:7.1f} {d[-1]:7.1f}")
N=128
for (Cin,Kout,relu) in ((512,128,True),(128,512,False),(1024,256,True),(256,1024,False)):
    A=((torch.rand(N*196,Cin)-0.5)*40).to(dev); B=((torch.rand(Cin,Kout)-0.5)*40).to(dev)
    s=(torch.rand(Kout)-0.5).to(dev); b=(torch.rand(Kout)-0.5).to(dev); out=torch.empty(N*196,Kout,device=dev)
    run(f"1x1 {Cin}->{Kout}", lambda: pkg.conv1x1_bn(A,B,b,s,relu,out=out))
for C in (128,256):
    x=(torch.rand(N,16,16,C)-0.5).to(dev); w=(torch.rand(C,C,3,3)-0.5).to(dev)
    s=(torch.rand(C)-0.5).to(dev); b=(torch.rand(C)-0.5).to(dev)
    U=pkg.filter_transform_f2(w); out=torch.empty(N,16,16,C,device=dev)
    run(f"3x3 {C}", lambda: pkg.conv3x3_bn_relu(x,U,b,s,out=out))
