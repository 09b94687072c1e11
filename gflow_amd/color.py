"""Colour maps on the device (reference: gflow/utils/color.py:24-44).

The reference moves the depth vector to the host, indexes a matplotlib table and
copies the result back -- every iteration (render.py:77-81).  Here the 256x3 tables
ship as data (gflow_amd/data/colormaps.npz, captured from matplotlib through the
reference's own function by tests/golden/make_golden.py) and the lookup runs in a
HIP kernel."""
import os

import numpy as np
import torch

from . import _lib as L

_LUTS = {}


def lut(name, device):
    key = (name, str(device))
    if key not in _LUTS:
        data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "colormaps.npz"))
        if name not in data:
            raise ValueError(f"unknown colormap {name!r}")
        _LUTS[key] = torch.from_numpy(data[name]).float().contiguous().to(device)
    return _LUTS[key]


def apply_float_colormap(image, colormap="turbo", non_zero=False):
    """image (...,1) float -> (...,3).  Same normalisation as color.py:24-44."""
    if colormap == "grey" or not non_zero:
        if non_zero:
            image = image - torch.min(image[image != 0])
        else:
            image = image - torch.min(image)
        image = torch.nan_to_num(torch.clip(image / (torch.max(image) + 1e-5), 0, 1), 0)
        if colormap == "grey":
            return image.expand(*image.shape[:-1], 3).contiguous()
        return lut(colormap, image.device)[(image * 255).long()[..., 0]]
    lib = L.load()
    L.need_device(image)
    v = image.detach().float().contiguous().reshape(-1)
    n = v.numel()
    out = torch.empty((n, 3), dtype=torch.float32, device=v.device)
    ws = L.scratch(16, v.device)
    L.check(lib.gfl_colormap_nonzero(L.ptr(v), n, L.ptr(lut(colormap, v.device)), L.ptr(out), L.ptr(ws), ws.numel(),
                                     L.stream()), "colormap")
    return out.reshape(*image.shape[:-1], 3)


def print_color(msg, color="green"):
    codes = {"red": 91, "green": 92, "yellow": 93, "blue": 94, "purple": 95, "cyan": 96, "white": 97}
    print(f"\033[{codes[color]}m {msg}\033[00m" if color in codes else msg)
