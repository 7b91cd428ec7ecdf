"""Calibration: what a plain streaming copy / read / write reaches on this GPU (torch kernels),
to put the solver kernels' GB/s in perspective (the roofline peak stays the 8 TB/s spec)."""
import torch
n = 1 << 27  # 1 GiB of fp64
x = torch.empty(n, dtype=torch.float64, device='cuda').normal_()
y = torch.empty_like(x)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
t = timeit(lambda: y.copy_(x)); print(f"copy   1 GiB -> 1 GiB: {2 * n * 8 / t / 1e12:.2f} TB/s (read+write)")
t = timeit(lambda: x.sum());    print(f"read   1 GiB (sum)   : {n * 8 / t / 1e12:.2f} TB/s")
t = timeit(lambda: y.fill_(1.0)); print(f"write  1 GiB (fill)  : {n * 8 / t / 1e12:.2f} TB/s")
t = timeit(lambda: torch.add(x, y, out=y)); print(f"axpy   2 reads + 1 write: {3 * n * 8 / t / 1e12:.2f} TB/s")
m = 1 << 24  # 128 MiB, the size of one N=4096 fp64 array
xs, ys = x[:m], y[:m]
t = timeit(lambda: ys.copy_(xs), 100); print(f"copy 128 MiB -> 128 MiB: {2 * m * 8 / t / 1e12:.2f} TB/s (fits the 256 MB Infinity Cache)")
