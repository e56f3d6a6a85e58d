import torch, statistics
dev = torch.device("cuda:0")
x = torch.empty(2 * 4800 * 4800, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)
us = t(lambda: x.fill_(1.0)); print(f"fill 184MB: {us:.1f} us -> {x.numel()*4/us/1e3:.0f} GB/s write")
us = t(lambda: x.zero_()); print(f"zero 184MB: {us:.1f} us -> {x.numel()*4/us/1e3:.0f} GB/s write")
us = t(lambda: y.copy_(x)); print(f"copy 184MB: {us:.1f} us -> {2*x.numel()*4/us/1e3:.0f} GB/s r+w")
big = torch.empty(1 << 30, dtype=torch.float32, device=dev)
us = t(lambda: big.fill_(1.0), 10); print(f"fill 4GB: {us:.1f} us -> {big.numel()*4/us/1e3:.0f} GB/s write")
