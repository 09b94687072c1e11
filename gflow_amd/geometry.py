"""Back-projection helpers on the hot path (reference: gflow/utils/geometry.py:95-120)."""
import torch


def inv(mat):
    """torch.linalg.inv without its error check (the check reads a status word back from the device: a full stop of
    the host in the middle of a fit; a singular extrinsic does not occur -- it is a rotation and a translation)."""
    return torch.linalg.inv_ex(mat, check_errors=False).inverse


def depth2pts3d(depth, xys, focal, pp):
    """depth (N,1), xys (N,2) pixels -> camera-space points (N,3) (geometry.py:118-119)."""
    return torch.cat((depth * (xys - pp) / focal, depth), dim=-1)


def geotrf(Trf, pts):
    """Apply a 4x4 rigid transform to (N,3) points (the only case pix2world uses)."""
    return pts @ Trf[:3, :3].T + Trf[:3, 3]


def pix2world(uv, depth, intr, extr):
    """uv (N,2), depth (N,1), intr (4,) [fx,fy,cx,cy], extr (3,4) world->camera.
    Like the reference (geometry.py:105-106) the single focal intr[0] is used for both
    axes."""
    rel = depth2pts3d(depth, uv, intr[0], intr[2:])
    # camera -> world of the rigid world -> camera transform [R | t]: [R^T | -R^T t].  The reference inverts the 4x4
    # numerically (geometry.py:107); for a rotation + translation that is the same matrix up to rounding, and the
    # closed form is three small kernels instead of an LU factorisation in the middle of a fit
    Rt = extr[:3, :3].T
    return rel @ Rt.T + (-(Rt @ extr[:3, 3]))
