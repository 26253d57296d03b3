import torch
dev = torch.device("cuda:0")
for mb in (268, 537):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("sum", lambda: x.sum())):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(mb, "MB", name, f"{us:.1f} us", f"{mb * 1.048576 / us:.2f} TB/s (one-sided bytes)")
