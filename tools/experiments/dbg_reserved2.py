import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from gflow_amd import synthetic as S
from gflow_amd.trainer import SimpleGaussian
DEV = "cuda"
f = S.make_clip(1, 96, 128, seed=6)[0]
kw = dict(lr=4e-3, lambda_rgb=1.0, lambda_depth=1e-2, lambda_var=1.0, densify_interval=0, move_mask=f["move_mask"])
tr = SimpleGaussian(f["image"], f["depth"], num_points=1500, device=DEV, seed=0)
tr.load_camera(focal=f["focal"], pp=f["pp"])
tr.init_gaussians_from_image(f["image"], f["depth"], num_points=1500)
st = tr.make_stepper(iterations=17, snapshot_interval=8, **kw)
for i in range(17):
    st.run(1)
    torch.cuda.synchronize()
    e = tr.engine
    print(i, e.overflow.tolist(), int(e.step.item()), e.K, e._reserved_N, e.N, float(e.sums[0]) if hasattr(e, "sums") else None)
