"""Throughput of ungar_gn_hessian (J^T diag(d) J on the FP64 matrix cores) for the ANYmal block size."""
import sys, time, json
sys.path.insert(0, ".")
import torch, ungar_amd
rows, cols, count = 37, 49, 81920
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
J = torch.rand((count, rows, cols), generator=gen, device="cuda", dtype=torch.float64)
d = torch.rand((count, rows), generator=gen, device="cuda", dtype=torch.float64)
G = torch.empty((count, cols, cols), dtype=torch.float64, device="cuda")
for _ in range(3): ungar_amd.gn_hessian(J, d, G, rows, cols, count)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in evs:
    s.record(); ungar_amd.gn_hessian(J, d, G, rows, cols, count); e.record()
torch.cuda.synchronize()
ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
bytes_ = count * 8 * (rows * cols + rows + cols * cols)
tiles = ((cols + 15) // 16) ** 2
flops_issued = count * tiles * ((rows + 3) // 4) * 2048
if "--unit-fastest" in sys.argv:
    Jt = J.reshape(count, rows * cols).t().contiguous()
    dt = d.t().contiguous()
    for _ in range(3): ungar_amd.gn_hessian_unit_fastest(Jt, dt, G, rows, cols, count)
    torch.cuda.synchronize()
    for s, e in evs:
        s.record(); ungar_amd.gn_hessian_unit_fastest(Jt, dt, G, rows, cols, count); e.record()
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    bytes_ = count * 8 * (rows * cols + rows + cols * (cols + 1) // 2)
    flops_issued = count * 10 * ((rows + 3) // 4) * 2048
    print(json.dumps({"kernel": "GnHessianUpperSoaKernel<4>", "nodes": count, "ms": ms, "nodes_per_s": count / ms * 1e3, "hbm_GBs": bytes_ / ms / 1e6,
                      "hbm_frac_of_8TBs": bytes_ / ms / 1e6 / 8000, "mfma_TFs_issued": flops_issued / ms / 1e9, "mfma_frac_of_78.6TF": flops_issued / ms / 1e9 / 78.6}))
    sys.exit(0)
upper = "--upper" in sys.argv
if upper:
    for _ in range(3): ungar_amd.gn_hessian(J, d, G, rows, cols, count, upper_only=True)
    torch.cuda.synchronize()
    for s, e in evs:
        s.record(); ungar_amd.gn_hessian(J, d, G, rows, cols, count, upper_only=True); e.record()
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
    bytes_ = count * 8 * (rows * cols + rows + cols * (cols + 1) // 2)
    tiles = ((cols + 15) // 16) * ((cols + 15) // 16 + 1) // 2
    flops_issued = count * tiles * ((rows + 3) // 4) * 2048
print(json.dumps({"kernel": "GnHessianKernel<4, upper>" if upper else "GnHessianKernel<4>", "nodes": count, "ms": ms, "nodes_per_s": count / ms * 1e3, "hbm_GBs": bytes_ / ms / 1e6,
                  "hbm_frac_of_8TBs": bytes_ / ms / 1e6 / 8000, "mfma_TFs_issued": flops_issued / ms / 1e9,
                  "mfma_frac_of_78.6TF": flops_issued / ms / 1e9 / 78.6}))
