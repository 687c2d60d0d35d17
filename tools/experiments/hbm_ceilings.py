"""Write-only / copy / read-only HBM rates of this GPU with stock torch kernels (fill_, copy_, sum): the ceilings the HBM-side analysis of
DESIGN section 5 is priced against (profiles/r02_hbm_ceilings.txt)."""
import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
print("fill  1 GiB: %.2f TB/s (write only)" % (x.numel() * 4 / t(lambda: x.fill_(1.0)) / 1e12))
print("copy  1 GiB: %.2f TB/s (read + write bytes)" % (2 * x.numel() * 4 / t(lambda: y.copy_(x)) / 1e12))
print("sum   1 GiB: %.2f TB/s (read only)" % (x.numel() * 4 / t(lambda: x.sum()) / 1e12))
s = x[: 58 * 1024 * 1024 // 4]
print("fill 58 MiB: %.2f TB/s" % (s.numel() * 4 / t(lambda: s.fill_(1.0), 50) / 1e12))
