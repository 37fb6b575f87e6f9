"""Achievable HBM streaming rates on this box (context for the roofline fractions: the node kernels are
write-dominated): device-wide fill (write only), copy (read + write) and read-reduce, 2 GiB operands."""
import json
import torch

n = 1 << 28  # 2 GiB of float64
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


t_fill = timed(lambda: a.fill_(1.0))
t_copy = timed(lambda: b.copy_(a))
t_read = timed(lambda: a.sum())
print(json.dumps({"fill_write_only_TBs": n * 8 / t_fill / 1e12, "copy_read_plus_write_TBs": 2 * n * 8 / t_copy / 1e12, "sum_read_only_TBs": n * 8 / t_read / 1e12}))
