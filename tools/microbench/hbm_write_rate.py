import torch, time
x = torch.empty(1280*1024*1024//8, dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("add_scalar (r+w)", lambda: x.add_(1.0))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)/10
    gb = x.numel()*8/1e9
    print(name, f"{ms:.3f} ms", f"write {gb/ms:.2f} TB/s" if name=="fill" else f"{2*gb/ms:.2f} TB/s total")
