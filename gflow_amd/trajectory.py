"""Trajectory poly-lines for ``SimpleGaussian.eval`` (reference: gflow/utils/trainer_functions.py:5-40).

``gen_line_set`` samples every segment xyz1[i] -> xyz2[i] at L_i = max(2, int(100 |xyz2[i] - xyz1[i]|)) points; the
first L_i - 1 of each segment form the line set (segment by segment), the N end points follow at the end.  The
reference builds the list in a Python double loop; here it is one vectorised pass on the device, point for point the
same values (pinned by tests/golden/line_set.npz, captured from the reference's function)."""
import torch


def gen_line_set(xyz1, xyz2, rgb, device=None):
    """xyz1, xyz2 (N,3), rgb (N,C) -> (line_set_xyz (sum(L_i - 1) + N, 3), line_set_rgb (same rows, C))."""
    device = xyz1.device if device is None else device
    xyz1, xyz2, rgb = xyz1.to(device), xyz2.to(device), rgb.to(device)
    n = xyz1.shape[0]
    diff = xyz2 - xyz1
    length = torch.norm(diff, dim=1)
    L = torch.clamp((length * 100).to(torch.int64), min=2)         # int() truncates
    cnt = L - 1                                                    # line points of segment i: j = 0 .. L_i - 2
    seg = torch.repeat_interleave(torch.arange(n, device=device), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    j = torch.arange(seg.shape[0], device=device) - first[seg]
    # t = j / (L - 1) is a Python float in the reference: formed in double, then rounded to the tensors' float32
    t = (j.double() / (L[seg] - 1).double()).to(xyz1.dtype).unsqueeze(1)
    line_xyz = xyz1[seg] + t * diff[seg]
    end_xyz = xyz1 + 1.0 * diff                                    # j = L - 1: t = 1.0
    return torch.cat([line_xyz, end_xyz]), torch.cat([rgb[seg], rgb])
