import torch, time
for mb in (8, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"H2D pinned {mb} MiB: {dt*1e3:.3f} ms = {(mb<<20)/dt/1e9:.1f} GB/s")
    # two streams at once
    s2 = torch.cuda.Stream()
    h2 = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d2 = torch.empty_like(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"  two streams: {2*(mb<<20)/dt/1e9:.1f} GB/s aggregate")
