#!/usr/bin/env python
"""Reference points for the Gram kernel's write stream: a pure fill (torch's vectorised fill kernel, hipMemsetAsync) and a copy of the
same size as the 1024 x 50 000 fp64 Gram matrix (409.6 MB), timed with HIP events."""
import torch

n = 1024 * 50000
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.randn(n, dtype=torch.float64, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


mb = n * 8 / 1e6
t = timed(lambda: x.fill_(1.5))
print(f"fill_  {mb:.1f} MB: {t:.1f} us  {mb / t:.2f} TB/s written")
t = timed(lambda: x.zero_())
print(f"zero_  {mb:.1f} MB: {t:.1f} us  {mb / t:.2f} TB/s written")
t = timed(lambda: x.copy_(y))
print(f"copy_  {mb:.1f} MB: {t:.1f} us  {mb / t:.2f} TB/s written (+ as much read)")
t = timed(lambda: torch.exp(y, out=x))
print(f"exp    {mb:.1f} MB: {t:.1f} us  {mb / t:.2f} TB/s written (+ as much read, one exp per element)")
