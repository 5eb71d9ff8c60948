import sys, json
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import cudabrot_amd as cb
dev=torch.device("cuda",0); w=h=4096; T=262144
dims=cb.FractalDimensions.make(w,h); it=cb.IterationControl(20000,20)
hist=torch.zeros(w*h,dtype=torch.int64,device=dev); states=torch.empty(cb.rng_state_bytes(T),dtype=torch.uint8,device=dev)
counters=torch.zeros(17,dtype=torch.int64,device=dev); carry=torch.zeros(cb.carry_bytes(T),dtype=torch.uint8,device=dev)
s=torch.cuda.current_stream().cuda_stream
cb.initialize_rng(1337,0,T,states.data_ptr(),s)
spt=3200; wsb=cb.scatter_workspace_bytes(dims,T,spt); ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
def draw(n):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(); cb.draw_buddhabrot(dims,hist.data_ptr(),it,states.data_ptr(),T,n,counters.data_ptr(),cb.CB_KERNEL_DEFAULT,s,ws.data_ptr(),wsb,carry.data_ptr()); b.record()
    cb.flush_scatter(dims,hist.data_ptr(),T,ws.data_ptr(),wsb,s); torch.cuda.synchronize(); return round(a.elapsed_time(b),2)
print("cold start:", [draw(spt) for _ in range(8)])
print("drain:", draw(0))
print("after drain:", [draw(spt) for _ in range(8)])
print("drain:", draw(0), draw(0))
