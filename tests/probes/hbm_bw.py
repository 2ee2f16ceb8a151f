"""achievable HBM write / read / copy bandwidth with framework kernels (reference points for DESIGN.md)"""
import time, torch
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")   # 4 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
gb = x.numel() * 4 / 1e9
print("fill  (write) %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("sum   (read)  %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("copy  (r+w)   %.2f TB/s" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
