import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
N,C,K=128,256,256
x=(torch.rand(N,16,16,C)-0.5).to(dev); w=(torch.rand(K,C,3,3)-0.5).to(dev)
s=(torch.rand(K)-0.5).to(dev); b=(torch.rand(K)-0.5).to(dev)
U=pkg.filter_transform_f2(w); out=torch.empty(N,16,16,K,device=dev)
step=lambda: pkg.conv3x3_bn_relu(x,U,b,s,out=out)
for _ in range(20): step()
torch.cuda.synchronize()
for rep in range(3):
    evs=[torch.cuda.Event(enable_timing=True) for _ in range(41)]
    t0=time.perf_counter()
    evs[0].record()
    for i in range(40):
        step(); evs[i+1].record()
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    d=[evs[i].elapsed_time(evs[i+1])*1e3 for i in range(40)]
    print("rep",rep,"host issue us/step %.1f"%((t1-t0)/40*1e6),"wall us/step %.1f"%((t2-t0)/40*1e6),"ev total/40 %.1f"%(evs[0].elapsed_time(evs[40])*1e3/40), "per-launch min/med/max %.1f %.1f %.1f"%(min(d),sorted(d)[20],max(d)))
# no events
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(200): step()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("plain: host issue us/step %.1f wall us/step %.1f"%((t1-t0)/200*1e6,(t2-t0)/200*1e6))
