"""Readers for a prepared GFlow sequence and a writer that lays a clip out the same way.

Mirrors ``gflow/utils/read.py`` and ``gflow/utils/conversion.py`` and the directory convention of
``gflow/fit_video.py:79-96``: next to the image folder ``<seq>/`` sit

    <seq>_depth_mast3r_s2/*.npy      per-frame depth                     (read.py:60-70)
    <seq>_flow_unimatch/*pred.flo    forward flow i -> i+1, Middlebury    (read.py:7-38)
    <seq>_flow_unimatch/*occ_bwd.png occlusion mask of frame i+1          (fit_video.py:88-89,248)
    <seq>_epipolar/*_open.png        move mask                           (fit_video.py:96-97)
    <seq>_camera_mast3r_s2/*.json    {"focal", "pp", "pose"}              (read.py:72-89)

``Resize(n, antialias=True)`` of torchvision (shorter side to n, bilinear with antialiasing) is
``torch.nn.functional.interpolate(..., mode="bilinear", antialias=True)``; imageio / torchvision are
not needed, PIL reads and writes the PNGs.
"""
import json
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

FLO_MAGIC = 202021.25


def _resize_chw(t, resize):
    """torchvision.transforms.Resize(int, antialias=True) on a (C,H,W) float tensor."""
    if resize is None:
        return t
    _, h, w = t.shape
    if h <= w:
        nh, nw = resize, int(resize * w / h)
    else:
        nh, nw = int(resize * h / w), resize
    if (nh, nw) == (h, w):
        return t
    return F.interpolate(t.unsqueeze(0), size=(nh, nw), mode="bilinear", antialias=True, align_corners=False).squeeze(0)


def image_path_to_tensor(image_path, resize=None):
    """conversion.py:6-19 -> (H,W,3) float in [0,1]."""
    from PIL import Image
    img = np.asarray(Image.open(image_path))
    t = torch.from_numpy(img.copy())
    if t.dim() == 2:
        t = t.unsqueeze(-1)
    t = t.permute(2, 0, 1).float() / (255.0 if img.dtype == np.uint8 else 65535.0 if img.dtype == np.uint16 else 1.0)
    return _resize_chw(t, resize).permute(1, 2, 0)[..., :3]


def read_flow(fn, resize=None):
    """read.py:7-38: Middlebury .flo (little endian) -> (H,W,2); None on a bad magic number."""
    with open(fn, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != np.float32(FLO_MAGIC):
            print("Magic number incorrect. Invalid .flo file")
            return None
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    flow = torch.from_numpy(np.resize(data, (h, w, 2)).copy()).permute(2, 0, 1)
    return _resize_chw(flow, resize).permute(1, 2, 0)


def write_flow(fn, flow):
    flow = np.asarray(flow, dtype=np.float32)
    h, w, _ = flow.shape
    with open(fn, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow.tofile(f)


def read_mask(mask_path, resize=None):
    """read.py:41-57 -> (H,W) bool (any non-zero channel)."""
    from PIL import Image
    mask = np.asarray(Image.open(mask_path))
    if mask.ndim == 3:
        t = torch.tensor(mask.copy(), dtype=torch.float32).permute(2, 0, 1)
    elif mask.ndim == 2:
        t = torch.tensor(mask.copy(), dtype=torch.float32).unsqueeze(0)
    else:
        raise ValueError("The mask should be 2D or 3D")
    t = _resize_chw(t, resize)
    if t.shape[-1] > 1:
        t = t.sum(dim=0)
    return t.squeeze() > 0


def read_depth(depth_path, resize=None, depth_scale=1.0, depth_offset=0.0):
    """read.py:60-70 -> (H,W) float."""
    t = torch.tensor(np.load(depth_path), dtype=torch.float32).unsqueeze(0)
    return _resize_chw(t, resize).squeeze(0) * depth_scale + depth_offset


def read_camera(camera_paths):
    """read.py:72-89 -> (mean focal, rounded principal point of the last file, poses[:, :3])."""
    focal_list, pose_list, pp = [], [], None
    for camera_path in camera_paths:
        with open(camera_path, "r") as f:
            d = json.load(f)
        focal_list.append(d["focal"])
        pose_list.append(d["pose"][:3])
        pp = [round(d["pp"][0]), round(d["pp"][1])]
    return float(np.array(focal_list).mean()), pp, np.array(pose_list)


# ----------------------------------------------------------------- sequence level
def sequence_paths(sequence_path, frame_start=0, frame_range=-1, skip_interval=1):
    """The file lists of fit_video.py:79-99, same globbing, slicing and ordering."""
    sp = str(sequence_path).rstrip("/")
    img = sorted(Path(sp).glob("*.png")) + sorted(Path(sp).glob("*.jpg"))
    if frame_range == -1:
        frame_range = len(img) - 1
    cut = lambda lst, n=frame_range: lst[frame_start:frame_start + n][::skip_interval]
    flow_dir = Path(sp + "_flow_unimatch")
    return dict(
        img=cut(img),
        depth=cut(sorted(Path(sp + "_depth_mast3r_s2").glob("*.npy"))),
        occ=cut(sorted(flow_dir.glob("*occ_bwd.png")) + sorted(flow_dir.glob("*occ_bwd.jpg")), frame_range - 1),
        flow=cut(sorted(flow_dir.glob("*pred.flo"))),
        move=cut(sorted(Path(sp + "_epipolar").glob("*_open.png"))),
        camera=cut(sorted(Path(sp + "_camera_mast3r_s2").glob("*.json"))),
    )


def load_sequence(sequence_path, resize=None, frame_start=0, frame_range=-1, skip_interval=1, depth_offset=0.0):
    """Frames in the dict form ``gflow_amd.fit_video.fit_clip`` takes.  Frame i carries the flow
    i -> i+1 (fit_video.py:249: frame i+1 is fitted with flow_paths[i]) and the occlusion mask that
    fit_video.py:248 reads for it (img_occ_paths[i-1])."""
    p = sequence_paths(sequence_path, frame_start, frame_range, skip_interval)
    focal, pp, poses = read_camera(p["camera"])
    frames = []
    for i, ip in enumerate(p["img"]):
        fr = dict(image=image_path_to_tensor(ip, resize), focal=focal, pp=pp, name=os.path.basename(ip).split(".")[0])
        fr["depth"] = read_depth(p["depth"][i], resize, depth_offset=depth_offset).unsqueeze(-1)
        H, W = fr["image"].shape[:2]
        fr["move_mask"] = read_mask(p["move"][i], resize) if i < len(p["move"]) else torch.zeros(H, W, dtype=torch.bool)
        fr["flow"] = read_flow(p["flow"][i], resize) if i < len(p["flow"]) else torch.zeros(H, W, 2)
        if i >= 1 and i - 1 < len(p["occ"]):
            fr["occ_mask"] = image_path_to_tensor(p["occ"][i - 1], resize)
        if i < len(poses):
            fr["extr"] = torch.tensor(poses[i], dtype=torch.float32)
        frames.append(fr)
    return frames


def write_sequence(frames, sequence_path):
    """Lay a clip (e.g. gflow_amd.synthetic.make_clip) out on disk in the reference's convention, so
    that the same folder can be read back here or handed to the reference's fit_video.py."""
    from PIL import Image
    sp = str(sequence_path).rstrip("/")
    dirs = {k: sp + s for k, s in (("img", ""), ("depth", "_depth_mast3r_s2"), ("flow", "_flow_unimatch"),
                                   ("move", "_epipolar"), ("camera", "_camera_mast3r_s2"))}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    u8 = lambda t: (torch.as_tensor(t).float().clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
    for i, fr in enumerate(frames):
        name = f"{i:05d}"
        Image.fromarray(u8(fr["image"])).save(os.path.join(dirs["img"], name + ".png"))
        np.save(os.path.join(dirs["depth"], name + ".npy"), torch.as_tensor(fr["depth"]).squeeze(-1).cpu().numpy())
        Image.fromarray(u8(torch.as_tensor(fr["move_mask"]).float())).save(os.path.join(dirs["move"], name + "_open.png"))
        if i + 1 < len(frames):
            write_flow(os.path.join(dirs["flow"], name + "_pred.flo"), torch.as_tensor(fr["flow"]).cpu().numpy())
        if i >= 1:
            occ = fr.get("occ_mask")
            occ = torch.zeros(fr["image"].shape[:2]) if occ is None else torch.as_tensor(occ).float()
            if occ.dim() == 3:
                occ = occ[..., 0]
            Image.fromarray(u8(occ)).save(os.path.join(dirs["flow"], f"{i - 1:05d}_occ_bwd.png"))
        pose = torch.eye(4)
        if fr.get("extr") is not None:
            pose[:3] = torch.as_tensor(fr["extr"]).float()
        with open(os.path.join(dirs["camera"], name + ".json"), "w") as f:
            json.dump({"focal": float(fr["focal"]), "pp": [float(fr["pp"][0]), float(fr["pp"][1])], "pose": pose.tolist()}, f)
    return sp
