import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import *
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip_amd.projections import create_projection_map
z = load("g1_syn2000.npz"); p = problem(z)
key = sys.argv[1] if len(sys.argv) > 1 else "simplex1|0.001|zero|f32"
mk, g, ln, dn = key.split("|")
pt, pp = SINGLE_MAPS[mk]
f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, create_projection_map(pt, dict(pp), p["n"]), "cuda:0"), float(g))
lam = torch.from_numpy(z[f"lam_{ln}"]).to(torch.float32 if dn=="f32" else torch.float64).cuda()
res = f.calculate(lam, gamma=float(g), save_primal=True)
x = res.primal_var.cpu().numpy(); want = z[f"{key}|x"]
bad = np.nonzero(np.abs(x - want) > 1e-4)[0]
print("nnz", len(x), "bad", len(bad), f.info())
cp = p["colptr"]
cols = np.unique(np.searchsorted(cp, bad, side="right") - 1)
print("bad cols", len(cols), cols[:20])
for j in cols[:6]:
    k0, k1 = cp[j], cp[j+1]
    print(j, k0, k1, "got", x[k0:k1], "want", want[k0:k1])
print("grad err", np.abs(res.dual_gradient.cpu().numpy() - z[f"{key}|grad"]).max())
lens = np.diff(cp)
tiles=[]; cur=0; start=None
for j,l in enumerate(lens):
    if l==0: continue
    if cur>0 and cur+l>64:
        tiles.append((start,cur)); cur=0
    if cur==0: start=cp[j]
    cur+=l
tiles.append((start,cur))
badset=set(bad.tolist())
for t,(s0,c0) in enumerate(tiles):
    bl=[k-s0 for k in range(s0,s0+c0) if k in badset]
    if bl: print("tile",t,"start",s0,"count",c0,"bad lanes",bl[0],"..",bl[-1],"n",len(bl))
