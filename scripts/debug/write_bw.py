"""HBM write / read / copy ceilings of this part as plain streaming kernels see them (torch ops on 16 GB)."""
import time, torch
n = 2_000_000_000
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
w = t(lambda: x.zero_())
r = t(lambda: x.sum())
c = t(lambda: y.copy_(x))
print("write %.2f TB/s  read %.2f TB/s  copy %.2f TB/s (read + write counted)" % (8 * n / w / 1e12, 8 * n / r / 1e12, 16 * n / c / 1e12))
