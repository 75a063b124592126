import torch, time
dev=torch.device("cuda",0)
for mb in (2.6, 10.5, 84):
    n=int(mb*1e6/4)
    h=torch.empty(n,dtype=torch.float32).pin_memory(); d=torch.empty(n,dtype=torch.float32,device=dev)
    for _ in range(3): d.copy_(h,non_blocking=True)
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(10): d.copy_(h,non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    t=time.perf_counter()
    for _ in range(10): h.copy_(d,non_blocking=True)
    torch.cuda.synchronize(); dt2=(time.perf_counter()-t)/10
    print("%.1f MB: H2D %.1f GB/s (%.0f us)  D2H %.1f GB/s (%.0f us)"%(mb, mb/1e3/dt, dt*1e6, mb/1e3/dt2, dt2*1e6))
