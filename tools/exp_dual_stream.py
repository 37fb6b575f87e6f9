"""Experiment: co-schedule the LDS-resident 'anymal' kernel (1 wavefront/CU) with the register-resident
'anymal_reg' kernel on a second stream so that the other three SIMDs of each CU are not idle."""
import sys, time
sys.path.insert(0, ".")
import torch, ungar_amd, bench
N, batch = 20, 4096
count = N * batch
x, u, w, p = bench.synth_device_inputs("anymal", count, 0, torch)
f = torch.empty((37, count), dtype=torch.float64, device="cuda"); J = torch.empty((37 * 49, count), dtype=torch.float64, device="cuda")
Op = ungar_amd.Operand
ma, mb = ungar_amd.NodeModel("anymal"), ungar_amd.NodeModel("anymal_reg")
s2 = torch.cuda.Stream()
def ops(lo, hi):
    n = hi - lo
    return (n, Op(x[:, lo:], 1, 1, count), Op(u[:, lo:], 1, 1, count), None, Op.per_instance(p, 1, shared=True), Op(f[:, lo:], 1, 1, count), Op(J[:, lo:], 1, 1, count))
def run(frac, iters=20):
    c1 = int(count * frac) // 64 * 64
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        ev = torch.cuda.Event(); ev.record()
        if c1 > 0: ma.dense_jacobian(*ops(0, c1))
        if c1 < count:
            s2.wait_event(ev)
            mb.dense_jacobian(*ops(c1, count), stream=s2.cuda_stream)
            e2 = torch.cuda.Event(); e2.record(s2); torch.cuda.current_stream().wait_event(e2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    return dt
for frac in (1.0, 0.0, 0.8, 0.7, 0.65, 0.6, 0.5):
    run(frac, 3); dt = run(frac)
    print(f"frac_lds={frac:.2f}  ms={dt*1e3:.3f}  evals/s={count/dt:.3e}  roofline={count*15192/dt/8e12:.4f}")
