import time, torch, numpy as np
a = np.random.randint(0,255,(4096,4096),dtype=np.uint8)
t = torch.from_numpy(a)
d = torch.empty((4096,4096),dtype=torch.uint8,device='cuda')
p = torch.empty((4096,4096),dtype=torch.uint8).pin_memory()
for name, src in (("pageable",t),("pinned",p)):
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter(); d.copy_(src); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"H2D 16.8MB {name}: {dt*1e3:.2f} ms")
out = np.empty(7_300_000,dtype=np.uint8); to = torch.from_numpy(out)
ds = torch.zeros(7_300_000,dtype=torch.uint8,device='cuda')
for name, dst in (("pageable",to),("pinned",p.view(-1)[:7_300_000])):
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter(); dst.copy_(ds); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"D2H 7.3MB {name}: {dt*1e3:.2f} ms")
t0=time.perf_counter(); p.copy_(t); print(f"host memcpy 16.8MB: {(time.perf_counter()-t0)*1e3:.2f} ms")
t0=time.perf_counter(); x=np.empty(17_800_000,dtype=np.uint8); x[:7_300_000]=1; print(f"fresh 17.8MB buffer, touch 7.3MB: {(time.perf_counter()-t0)*1e3:.2f} ms")
