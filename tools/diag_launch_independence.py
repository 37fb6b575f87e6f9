import torch, numpy as np, sys
sys.path.insert(0, "/root/repo")
import ungar_amd as ua
from ungar_amd import workloads as W
N=20; instances=8192; count=instances*N
x,u,_,p = W.synth_device_inputs("anymal", count, 5, torch)
m = ua.NodeModel("anymal")
def ev(x,u,count,knots):
    f = torch.full((37, count), float("nan"), dtype=torch.float64, device="cuda")
    J = torch.full((1813, count), float("nan"), dtype=torch.float64, device="cuda")
    Op=ua.Operand
    m.dense_jacobian(count, Op.soa(x,count,knots), Op.soa(u,count,knots), None, Op.per_instance(p,m.np,shared=True), Op.soa(f,count,knots), Op.soa(J,count,knots), knots=knots)
    torch.cuda.synchronize(); return f,J
f,J = ev(x,u,count,N)
lo=(count//2//N)*N
for n in (4096, 4095, 40000, 81920):
    sub=slice(lo,lo+n)
    f2,J2 = ev(x[:,sub].contiguous(), u[:,sub].contiguous(), n, 1)
    d=(J2-J[:,sub]).abs(); nz=(d>0)
    print(n, "f equal", torch.equal(f2,f[:,sub]), "J differ entries", int(nz.sum()), "max", float(d.max()), "rows with diffs", torch.unique(nz.nonzero()[:,0])[:20].tolist() if nz.any() else [])
