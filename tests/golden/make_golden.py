"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Only input/output DATA is written (small .npz files); no reference source is
copied.  The GPU box has no /root/reference, so nothing at test time runs this.

What the reference lets us import: utils/pytorch_ssim.py,
utils/geometry.py, utils/color.py, utils/trainer_functions.py.  The rasteriser (msplat) cannot be imported.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/gflow/utils"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "gflow_amd", "data")


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are already committed")
    torch.manual_seed(0)
    ssim = load("pytorch_ssim")
    geometry = load("geometry")
    color = load("color")

    # ---- SSIM (gflow/utils/pytorch_ssim.py:7-63), value + full gradient, small
    a = torch.rand(1, 3, 32, 48, requires_grad=True)
    b = torch.rand(1, 3, 32, 48)
    val = ssim.SSIM()(a, b)
    val.backward()
    small = dict(img1=a.detach().numpy(), img2=b.numpy(), value=val.item(), grad1=a.grad.numpy())
    # correlated pair (more representative of render vs gt)
    c = torch.rand(1, 3, 40, 56)
    d = (c + 0.05 * torch.randn_like(c)).clamp(0, 1).requires_grad_(True)
    val2 = ssim.SSIM()(d, c)
    val2.backward()
    small.update(img3=d.detach().numpy(), img4=c.numpy(), value34=val2.item(), grad3=d.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "ssim_small.npz"), **small)

    # ---- SSIM at 480p: value + gradient probes; inputs regenerated from the seed
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(1, 3, 480, 854, generator=g)
    y = (x + 0.1 * torch.rand(1, 3, 480, 854, generator=g)).clamp(0, 1)
    x.requires_grad_(True)
    v = ssim.SSIM()(x, y)
    v.backward()
    probes = np.array([[0, 0, 0], [1, 5, 5], [2, 239, 427], [0, 479, 853], [1, 100, 700], [2, 7, 850]])
    gp = np.array([x.grad[0, c_, i, j].item() for c_, i, j in probes])
    np.savez_compressed(os.path.join(HERE, "ssim_480p.npz"), seed=1234, value=v.item(), probes=probes, grad_probes=gp)

    # ---- pix2world / depth2pts3d (gflow/utils/geometry.py:95-120)
    uv = torch.tensor([[10., 20.], [0., 0.], [853., 479.], [427., 240.], [100.5, 33.25]])
    dep = torch.tensor([[1.5], [2.0], [0.7], [3.3], [1.0]])
    intr = torch.tensor([427., 427., 427., 240.])
    extr_id = torch.eye(4)[:3]
    ang = 0.3
    Rm = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    extr_rt = torch.cat([Rm, torch.tensor([[0.1], [-0.2], [0.3]])], dim=1)
    out_id = geometry.pix2world(uv, dep, intr, extr_id)
    out_rt = geometry.pix2world(uv, dep, intr, extr_rt)
    np.savez_compressed(os.path.join(HERE, "pix2world.npz"), uv=uv.numpy(), depth=dep.numpy(), intr=intr.numpy(),
                        extr_id=extr_id.numpy(), extr_rt=extr_rt.numpy(), xyz_id=out_id.numpy(), xyz_rt=out_rt.numpy())

    # ---- colour maps (gflow/utils/color.py:24-44): LUTs + one non_zero vector
    ramp = (torch.arange(256).float() / 255.0).unsqueeze(1)
    # exact LUT rows: feed values whose (x*255).long() hits every index after the
    # function's own normalisation (min 0, max 1+1e-5 scaling): use the raw table
    from matplotlib import cm
    turbo = cm.get_cmap("turbo")(np.arange(256))[:, :3].astype(np.float32)
    rainbow = cm.get_cmap("gist_rainbow")(np.arange(256))[:, :3].astype(np.float32)
    depth_vec = torch.tensor([[0.0], [1.2], [3.4], [0.0], [2.2], [5.0], [1.2001], [4.9]])
    nz = color.apply_float_colormap(depth_vec, colormap="turbo", non_zero=True)
    full = color.apply_float_colormap(ramp, colormap="gist_rainbow")
    np.savez_compressed(os.path.join(HERE, "colormap.npz"), turbo=turbo, gist_rainbow=rainbow,
                        depth_vec=depth_vec.numpy(), turbo_non_zero=nz.numpy(),
                        ramp=ramp.numpy(), rainbow_ramp=full.numpy())
    os.makedirs(DATA, exist_ok=True)
    np.savez_compressed(os.path.join(DATA, "colormaps.npz"), turbo=turbo, gist_rainbow=rainbow)

    # ---- trajectory poly-lines (gflow/utils/trainer_functions.py:5-40)
    tf = load("trainer_functions")
    g2 = torch.Generator().manual_seed(77)
    x1 = torch.rand(9, 3, generator=g2)
    x2 = x1 + torch.tensor([[0.0, 0.0, 0.0], [0.005, 0.0, 0.0], [0.02, 0.01, 0.0], [0.1, -0.05, 0.02], [0.0, 0.3, 0.0],
                            [0.019999, 0.0, 0.0], [0.03, 0.0, 0.0], [-0.25, 0.1, 0.4], [1e-4, 1e-4, 1e-4]])
    col = torch.rand(9, 3, generator=g2)
    lx, lc = tf.gen_line_set(x1, x2, col, device="cpu")
    np.savez_compressed(os.path.join(HERE, "line_set.npz"), xyz1=x1.numpy(), xyz2=x2.numpy(), rgb=col.numpy(),
                        line_xyz=lx.numpy(), line_rgb=lc.numpy())
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
