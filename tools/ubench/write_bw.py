"""HBM write / copy bandwidth reachable by plain kernels on this box (torch fill_ / copy_), for comparison with k_thin_rt."""
import torch, time
for gb in (0.25, 1.0, 4.0):
    n = int(gb * 2**30 / 8)
    x = torch.empty(n, dtype=torch.float64, device='cuda')
    y = torch.empty(n, dtype=torch.float64, device='cuda')
    for name, fn, bytes_ in (('fill_', lambda: x.fill_(1.5), 8 * n), ('copy_', lambda: y.copy_(x), 16 * n), ('sum', lambda: x.sum(), 8 * n)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%4.2f GiB %-6s %7.3f ms  %6.2f TB/s' % (gb, name, ms, bytes_ / ms / 1e9))
