"""CPU restatement of the frame-boundary state of gflow/trainer.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ alone.

Two functions, each following the reference line by line with the reference's own formulation --
boolean gathers and nested scatter-backs -- which is deliberately NOT how gflow_amd/trainer.py does it
(``torch.where`` over the full rows), so that a wrong mask, clamp or truncation shows up as a difference:

  warp_moving   trainer.py:347-376  "pre-update": before a joint-stage ``train`` of a later frame the splats that
                were labelled moving are carried along the ground-truth flow and re-lifted at the ground-truth depth;
  relabel       trainer.py:588-602,621-625  "post-update": still / moving labels from the epipolar move mask at the
                splats' projected positions, the labels of earlier frames kept, and the ``last_*`` stash.

  mask_prompt_points / propagated_points   trainer.py:290-330, 611-615  the splats under the first frame's mask prompt and
                where they project after a later frame's fit (the points the reference hands to its concave hull).

``pix2world`` is the golden-pinned restatement of loss_oracle.py (tests/golden/pix2world.npz was captured from the
reference's geometry.py).  Everything is index arithmetic on float32 inputs: the comparison is exact for the masks
and to float32 rounding for the lifted positions."""
import torch

from .loss_oracle import pix2world


def _inside(uv, W, H):
    return (uv[:, 0] > 0) & (uv[:, 0] < W - 1) & (uv[:, 1] > 0) & (uv[:, 1] < H - 1)


def warp_moving(xyz, last_uv, last_still_mask, gt_flow, gt_depth, intr, extr, W, H):
    """trainer.py:348-376.  xyz (N,3) raw positions (N >= M); last_uv (>=M,2) projections after the previous frame's
    fit; last_still_mask (M,) bool; gt_flow (H,W,2) flow from the previous frame to this one; gt_depth (H,W,1) of
    this frame; intr (4,), extr (3,4).  Returns the new (N,3) positions."""
    M = last_still_mask.shape[0]
    uv_move = last_uv[:M][~last_still_mask]                                # :349
    within = _inside(uv_move, W, H)                                        # :351
    uv_move = uv_move[within]                                              # :352
    y = uv_move[:, 1].long()                                               # :353  (truncation, no clamp: inside by :351)
    x = uv_move[:, 0].long()
    uv_move = uv_move + gt_flow[y, x]                                      # :355-356
    y2 = torch.clamp(uv_move[:, 1].long(), 0, H - 1)                       # :357-361
    x2 = torch.clamp(uv_move[:, 0].long(), 0, W - 1)
    depth_move = gt_depth[y2, x2].reshape(-1, 1)                           # :362
    xyz_move = pix2world(uv_move, depth_move, intr, extr)                  # :364
    out = xyz.clone()                                                      # :366
    temp_1 = out[:M][~last_still_mask].clone()                             # :368
    temp_1[within] = xyz_move                                              # :369
    temp_2 = out[:M].clone()                                               # :371
    temp_2[~last_still_mask] = temp_1                                      # :372
    out[:M] = temp_2                                                       # :374
    return out


def relabel(uv, move_mask, n_now, last_still_mask=None):
    """trainer.py:588-602.  uv (n_rendered,2) of the LAST iteration's render; move_mask (H,W) bool; n_now = current
    number of splats; last_still_mask (M,) bool or None.  Returns (still_mask, still_mask_tentative), both (n_now,)."""
    H, W = move_mask.shape
    within = _inside(uv, W, H)                                             # :590
    y = uv[within][:, 1].long()                                            # :591-592
    x = uv[within][:, 0].long()
    labels = ~move_mask[y, x]                                              # :593
    still = torch.ones(n_now, dtype=torch.bool)                            # :595
    head = still[:uv.shape[0]].clone()
    head[within] = labels                                                  # :596 (n_rendered == n_now in the reference)
    still[:uv.shape[0]] = head
    tentative = still.clone()                                              # :597
    if last_still_mask is not None:
        still[:last_still_mask.shape[0]] = last_still_mask                 # :598-599
    return still, tentative


def mask_prompt_points(uv, mask_prompt, W, H):
    """trainer.py:309-330.  uv (N,2) projections after the first frame's fit, mask_prompt (H,W) bool / 0-1.
    Returns mask_prompt_pts (N,) bool."""
    uv_within = _inside(uv, W, H)                                          # :310
    uvw = uv[uv_within]                                                    # :311
    y = uvw[:, 1].long()                                                   # :313-314
    x = uvw[:, 0].long()
    pts = mask_prompt[y, x].bool()                                         # :328
    out = uv_within.clone()                                                # :329
    out[uv_within] = pts                                                   # :330
    return out


def propagated_points(uv, mask_prompt_pts, W, H):
    """trainer.py:612-614: the projections (of the last render) of the prompt's splats that are inside the image -- what
    FastConcaveHull2D is built from when there are more than four."""
    p = uv[:mask_prompt_pts.shape[0]][mask_prompt_pts]                     # :612
    return p[_inside(p, W, H)]                                             # :613-614
