import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_render_op import *
import gflow_amd.render as R
sa = to_dev(random_scene(1200, 96, 80, seed=1, sigma_px=2.0))
sb = to_dev(random_scene(900, 96, 80, seed=2, sigma_px=3.0))
def run(s, together_with=None):
    leaves = {k: s[k].clone().requires_grad_(True) for k in NAMES}
    out = R.render(leaves, dict(intr=s["intr"], extr=s["extr"], W=s["W"], H=s["H"]), 0.0)
    other = together_with() if together_with else None
    (out["rgb"].sum() + out["depth_map"].sum()).backward()
    return {k: leaves[k].grad.clone() for k in NAMES}, other
a1, _ = run(sa); a2, _ = run(sa); b1, _ = run(sb)
n_a, n_b = run(sa, together_with=lambda: run(sb)[0])
for k in NAMES:
    d12 = (a1[k] - a2[k]).abs(); dn = (a1[k] - n_a[k]).abs(); db = (b1[k] - n_b[k]).abs()
    rel = lambda d, ref: (d / (ref.abs() * 1e-4 + 1e-7)).max().item()
    print(k, "alone-alone %.3g (x tol %.2f)" % (d12.max().item(), rel(d12, a2[k])), "alone-nested %.3g (x tol %.2f)" % (dn.max().item(), rel(dn, n_a[k])),
          "b: %.3g (x tol %.2f)" % (db.max().item(), rel(db, n_b[k])), "max |g| %.3g" % a1[k].abs().max().item())
bad = ((a1["rotate"] - n_a["rotate"]).abs() > 1e-4 * n_a["rotate"].abs() + 1e-7).nonzero()
print("bad rows", bad[:10].tolist(), len(bad))
for r, c in bad[:5].tolist():
    print(r, c, a1["rotate"][r].tolist(), n_a["rotate"][r].tolist(), sa["scale"][r].tolist())
