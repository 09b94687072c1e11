"""Frame-boundary extras: the trajectory render ``SimpleGaussian.eval`` / ``render_traj`` against an
oracle restatement of gflow/utils/render.py:110-156 and gflow/trainer.py:713-811, and the checkpoint round trip through a
restatement of the reference's loader (gflow/viewer.py:50-64) (``-m gpu``)."""
import os

import numpy as np
import pytest
import torch

from oracle import fit_oracle as FO
from oracle import loss_oracle as LO
from oracle import msplat_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("xyz", "scale", "rotate", "opacity", "rgb")


def _fitted(tmp_path=None, iters=12):
    from gflow_amd import synthetic as S
    from gflow_amd.trainer import SimpleGaussian
    frame = S.make_frame(96, 128, seed=5)
    tr = SimpleGaussian(frame["image"], frame["depth"], num_points=1500, device=DEV, seed=0,
                        log_dir=None if tmp_path is None else str(tmp_path))
    tr.load_camera(focal=frame["focal"], pp=frame["pp"])
    tr.init_gaussians_from_image(frame["image"], frame["depth"], num_points=1500)
    tr.train(iterations=iters, lr=4e-3, lr_camera=1e-3, lambda_rgb=1.0, lambda_depth=0.1, lambda_var=10.0,
             move_mask=frame["move_mask"], densify_interval=6, densify_times=1, snapshot_interval=0)
    return tr, frame


def _u8(img_chw):
    return (torch.clamp(img_chw.detach().permute(1, 2, 0), 0.0, 1.0).numpy() * 255).astype(np.uint8)


def _img_close(a, b, what, frac=2e-3):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert (d > 1).mean() <= frac, f"{what}: {(d > 1).mean():.2e} of the values differ by more than one level"


def test_eval_renders_match_the_oracle_over_two_frames():
    tr, frame = _fitted()
    g = torch.Generator().manual_seed(2)
    traj_index = torch.randperm(tr.current_pts_num(), generator=g)[:40]
    traj_state = {}

    def oracle_eval(first):
        """trainer.py:713-811 on the CPU with the oracle operators."""
        raw = {k: v.detach().cpu() for k, v in tr._attributes.items()}
        act = FO.activate(raw)
        intr, extr = tr.intr.cpu(), tr.get_extr().detach().cpu()
        n = traj_index.shape[0]
        xyz_now = raw["xyz"][traj_index].float()
        if first:
            col = torch.arange(0, 1, 1 / n).float().unsqueeze(1)
            lut = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "colormap.npz"))["gist_rainbow"])
            col = col - col.min()
            col = torch.nan_to_num(torch.clip(col / (col.max() + 1e-5), 0, 1), 0)
            traj_state.update(xyz=xyz_now, scale=torch.ones(n, 3), op=torch.logit(0.99 * torch.ones(n, 1)) / 10.0,
                              rgb=torch.nan_to_num(torch.logit(lut[(col * 255).long()[..., 0]]), posinf=1e6, neginf=-1e6),
                              last_xyz=xyz_now)          # (+-inf of logit(0 / 1) kept finite: see SimpleGaussian.eval)
            traj_state["last_rgb"] = traj_state["rgb"]
        else:
            from gflow_amd.trajectory import gen_line_set          # pinned to the reference by test_host_logic
            lx, lc = gen_line_set(traj_state["last_xyz"], xyz_now, traj_state["last_rgb"])
            traj_state["xyz"] = torch.cat([traj_state["xyz"], lx])
            traj_state["scale"] = torch.ones(traj_state["xyz"].shape[0], 3) * 1e-6
            traj_state["op"] = torch.cat([traj_state["op"] * 0.5, torch.logit(0.99 * torch.ones(lx.shape[0], 1)) / 10.0])
            traj_state["rgb"] = torch.cat([traj_state["rgb"], lc])
            traj_state["last_xyz"] = xyz_now
        full = MO.render_multiple([*act, intr, extr, tr.bg, tr.W, tr.H], ["rgb", "center", "depth_map_color"])
        rot = torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(traj_state["xyz"].shape[0], 1)
        traj = MO.render_traj([traj_state["xyz"], traj_state["scale"], rot, traj_state["op"], traj_state["rgb"], intr, extr,
                               tr.bg, tr.W, tr.H], n, 0.1, 0.3)
        return _u8(full["rgb"]), _u8(full["center"]), _u8(full["depth_map_color"]), _u8(traj)

    for step in range(2):
        got = tr.eval(traj_index=traj_index, line_scale=0.1, point_scale=0.3, alpha=0.5)
        want = oracle_eval(first=(step == 0))
        for a, b, name in zip(got[:4], want, ("rgb", "center", "depth colour", "trajectories")):
            assert a.shape == (tr.H, tr.W, 3) and a.dtype == np.uint8
            _img_close(a, b, f"frame {step} {name}")
        screen = (1 - (1 - got[0] / 255.0) * (1 - got[3] / 255.0)) * 255
        assert np.array_equal(got[4], screen.astype(np.uint8))
        assert got[3].max() > 0                                       # something was drawn
        # move the tracked splats before the next frame
        with torch.no_grad():
            tr._attributes["xyz"][traj_index.to(DEV)] += 0.03 * torch.randn(40, 3, generator=g).to(DEV)


def test_checkpoint_round_trip_through_the_reference_loader(tmp_path):
    """save_checkpoint writes the reference's keys (trainer.py:252-272) as compact, contiguous tensors; a loader
    restated from gflow/viewer.py:50-64 (activations :20-36) re-renders the frame bit-identically; load_checkpoint
    (trainer.py:274-288) restores a trainer that renders the same image."""
    import gflow_amd.render as R
    from gflow_amd.trainer import SimpleGaussian
    tr, frame = _fitted(tmp_path)
    tr.save_checkpoint(ckpt_name="0000")
    path = os.path.join(str(tmp_path), "ckpt", "0000.tar")
    assert tr.checkpoint_path == path and os.path.exists(path)
    n = tr.current_pts_num()
    assert os.path.getsize(path) < 1.5 * (n * 14 * 4 + n * (1 + 8)) + 200_000      # not the 8x-capacity storage
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"attributes", "intr", "extr", "still_mask", "move_seg", "last_uv", "width", "height"}
    assert set(ck["attributes"].keys()) == set(NAMES)
    for k in NAMES:
        assert ck["attributes"][k].is_contiguous() and ck["attributes"][k].shape[0] == n
        assert ck["attributes"][k].untyped_storage().nbytes() == ck["attributes"][k].numel() * 4
    assert tuple(ck["extr"].shape) == (3, 4) and int(ck["width"]) == tr.W and int(ck["height"]) == tr.H
    assert ck["still_mask"].shape[0] == n and ck["last_uv"].shape == (n, 2)
    # --- the viewer's loader, restated: activations of viewer.py:20-36, fields of :50-64
    att = {k: v.to(DEV) for k, v in ck["attributes"].items()}         # torch.load(..., map_location=device), :52
    xyz, scale, rotate = att["xyz"], torch.abs(att["scale"]), torch.nn.functional.normalize(att["rotate"])
    opacity, rgb = torch.sigmoid(10.0 * att["opacity"]), torch.sigmoid(att["rgb"])
    group = [xyz, scale, rotate, opacity, rgb, ck["intr"].to(DEV), ck["extr"].to(DEV), tr.bg, tr.W, tr.H]
    with torch.no_grad():
        again = R.render_multiple(group, ["rgb"])["rgb"]
        now = R.render_multiple(tr._input_group(detach=True), ["rgb"])["rgb"]
    assert torch.equal(again, now)
    # --- load_checkpoint into a fresh trainer
    tr2 = SimpleGaussian(frame["image"], frame["depth"], num_points=1500, device=DEV, seed=1)
    tr2.load_checkpoint(path)
    with torch.no_grad():
        other = R.render_multiple(tr2._input_group(detach=True), ["rgb"])["rgb"]
    assert (other - now).abs().max().item() < 2e-4                 # (the pose goes through a matrix -> quaternion round trip)
    assert torch.equal(tr2.still_mask.cpu(), tr.still_mask.cpu())
    with pytest.raises(RuntimeError):
        SimpleGaussian(frame["image"], frame["depth"], num_points=10, device=DEV).save_checkpoint()
