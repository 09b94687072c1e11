"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run once in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Only input/output DATA is written (small .npz files); no reference source is
copied.  The GPU box has no /root/reference, so nothing at test time runs this.

What the reference lets us import: utils/pytorch_ssim.py,
utils/geometry.py, utils/color.py, utils/trainer_functions.py.  The rasteriser (msplat) cannot be imported.

utils/read.py does not import here (its first lines import imageio and torchvision, both absent).  Three of its
functions -- read_flow, read_depth, read_camera (read.py:7-38, 60-89) -- are numpy / json / torch only on the paths
GFlow takes them without --resize / --blur, so ``load_functions`` compiles exactly those three ``def`` blocks out
of the reference file (nothing is copied or written: the AST is executed in memory) in a namespace with the real
numpy, torch and json.  The name ``transforms`` they mention is bound to ``_NoTorchvision``: its ``Compose`` accepts
ONLY the empty list (torchvision documents the empty composition as the identity) and ``Resize`` / ``GaussianBlur``
raise -- no torchvision arithmetic is imitated, the resize / blur paths are simply not captured
(gflow_amd/io.py's Resize is pinned by the hand-worked fixtures of tests/test_host_logic.py instead).
read_mask / image_path_to_tensor / complex_texture_sampling need imageio / torchvision / cv2 for their arithmetic
and cannot be captured at all.
"""
import ast
import importlib.util
import json
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


class _NoTorchvision:
    """see the module docstring: the identity for an EMPTY composition, nothing else"""

    @staticmethod
    def Compose(lst):
        assert list(lst) == [], "only the no-resize / no-blur path of read.py is captured"
        return lambda x: x

    @staticmethod
    def Resize(*a, **k):
        raise RuntimeError("torchvision is absent: the resize path is not captured")

    GaussianBlur = Resize


def load_functions(name, wanted):
    """the ``def`` blocks ``wanted`` of a reference module whose import fails on absent packages, executed in memory"""
    path = os.path.join(REF, name + ".py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert sorted(n.name for n in body) == sorted(wanted)
    ns = {"np": np, "torch": torch, "json": json, "transforms": _NoTorchvision}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return {k: ns[k] for k in wanted}


def readers_fixture():
    """read_flow / read_depth / read_camera of the reference on small hand-made files; the fixture keeps the files'
    BYTES (so that the test hands gflow_amd/io.py the very same files) and what the reference returned."""
    import tempfile
    ref = load_functions("read", ["read_flow", "read_depth", "read_camera"])
    g = np.random.default_rng(5)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        # Middlebury .flo: magic, w, h, then h*w*2 floats; 5 rows x 7 columns, signed fractional values
        flow = (g.standard_normal((5, 7, 2)) * 3.0).astype(np.float32)
        flow[0, 0] = (-0.0, 1e-7)
        flo = os.path.join(d, "a_pred.flo")
        with open(flo, "wb") as f:
            np.array([202021.25], np.float32).tofile(f)
            np.array([7, 5], np.int32).tofile(f)
            flow.tofile(f)
        bad = os.path.join(d, "bad.flo")
        with open(bad, "wb") as f:
            np.array([202021.0], np.float32).tofile(f)
            np.array([7, 5], np.int32).tofile(f)
            flow.tofile(f)
        out["flo_bytes"] = np.fromfile(flo, np.uint8)
        out["flo_bad_bytes"] = np.fromfile(bad, np.uint8)
        out["flow"] = ref["read_flow"](flo).numpy()
        assert ref["read_flow"](bad) is None
        # depth: float64 on disk (MASt3R writes float32; the reader casts either), with scale / offset
        depth = g.uniform(0.5, 6.0, (4, 6))
        npy = os.path.join(d, "00000.npy")
        np.save(npy, depth)
        out["depth_npy_bytes"] = np.fromfile(npy, np.uint8)
        out["depth"] = ref["read_depth"](npy).numpy()
        out["depth_scaled"] = ref["read_depth"](npy, depth_scale=0.5, depth_offset=0.25).numpy()
        # cameras: focal averaged over the files, pp of the LAST file rounded (python round: halves to even),
        # pose[:3] per file
        cams = []
        for i, (focal, pp) in enumerate(((498.25, (426.5, 239.5)), (503.5, (427.5, 240.49)), (500.125, (428.5, 240.5)))):
            pose = np.eye(4)
            pose[:3, :3] = np.linalg.qr(g.standard_normal((3, 3)))[0]
            pose[:3, 3] = g.standard_normal(3) * 0.1
            cams.append({"focal": focal, "pp": list(pp), "pose": pose.tolist()})
        paths = []
        for i, c in enumerate(cams):
            paths.append(os.path.join(d, f"{i:05d}.json"))
            with open(paths[-1], "w") as f:
                json.dump(c, f)
        out["camera_json"] = np.array([json.dumps(c) for c in cams])
        focal, pp, poses = ref["read_camera"](paths)
        out["focal"], out["pp"], out["poses"] = np.float64(focal), np.array(pp, dtype=np.int64), poses
        f1, pp1, _ = ref["read_camera"](paths[:2])
        out["focal_first_two"], out["pp_first_two"] = np.float64(f1), np.array(pp1, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "readers.npz"), **out)


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
    readers_fixture()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
