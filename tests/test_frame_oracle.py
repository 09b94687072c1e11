"""Hand-worked known answers for oracle/frame_oracle.py (CPU): the restatement of the frame-boundary state
(gflow/trainer.py:347-376, 588-602) that tests/test_gpu_frame_state.py holds the trainer against."""
import torch

from oracle import frame_oracle as FR


def test_warp_moves_only_moving_splats_inside_the_image():
    W, H = 8, 6
    intr = torch.tensor([2.0, 2.0, 4.0, 3.0])                    # f = 2, principal point (4, 3)
    extr = torch.eye(4)[:3]
    #                 moving+inside   still+inside   moving+outside(u = W-1)  moving, flow pushes it off the image
    last_uv = torch.tensor([[2.5, 1.5], [3.0, 3.0], [7.0, 2.0], [6.2, 4.9], [1.0, 1.0]])
    last_still = torch.tensor([False, True, False, False])       # M = 4 < N = 5: row 4 was appended later
    xyz = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    flow = torch.zeros(H, W, 2)
    flow[1, 2] = torch.tensor([1.0, 2.0])                         # sampled at trunc(2.5, 1.5) = pixel (x 2, y 1)
    flow[4, 6] = torch.tensor([3.0, 2.5])                         # (6.2, 4.9) -> (9.2, 7.4): clamped to pixel (7, 5)
    depth = torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1) * 0.1 + 1.0
    out = FR.warp_moving(xyz, last_uv, last_still, flow, depth, intr, extr, W, H)
    # row 0: uv -> (3.5, 3.5), depth at pixel (x 3, y 3) = 1 + 0.1 * 27 = 3.7; xyz = d * ((u - cx) / f, (v - cy) / f, 1)
    d0 = 3.7
    assert torch.allclose(out[0], torch.tensor([d0 * (3.5 - 4.0) / 2.0, d0 * (3.5 - 3.0) / 2.0, d0]), atol=1e-6)
    # row 3: uv -> (9.2, 7.4), depth at the clamped pixel (x 7, y 5) = 1 + 0.1 * 47 = 5.7, lifted at the UNclamped uv
    d3 = 5.7
    assert torch.allclose(out[3], torch.tensor([d3 * (9.2 - 4.0) / 2.0, d3 * (7.4 - 3.0) / 2.0, d3]), atol=1e-5)
    for r in (1, 2, 4):                                           # still / on the image border / appended: untouched
        assert torch.equal(out[r], xyz[r])


def test_relabel_keeps_old_labels_and_defaults_to_still():
    W, H = 8, 6
    move = torch.zeros(H, W, dtype=torch.bool)
    move[2, 3] = True
    move[4, 5] = True
    uv = torch.tensor([[3.9, 2.2],      # inside, on a moving pixel (trunc -> x 3, y 2)        -> moving
                       [3.9, 3.0],      # inside, still pixel                                   -> still
                       [0.0, 0.0],      # culled splat (uv 0, 0): not inside                    -> still
                       [7.0, 2.0],      # u = W - 1: not inside                                 -> still
                       [5.5, 4.5]])     # inside, moving pixel                                  -> moving
    still, tent = FR.relabel(uv, move, n_now=6, last_still_mask=None)           # row 5: appended after the render
    assert still.tolist() == [False, True, True, True, False, True]
    assert torch.equal(still, tent)
    still, tent = FR.relabel(uv, move, n_now=6, last_still_mask=torch.tensor([True, False, False]))
    assert tent.tolist() == [False, True, True, True, False, True]              # this frame's own labels
    assert still.tolist() == [True, False, False, True, False, True]            # rows 0..2 keep last frame's labels


def test_mask_prompt_points_on_a_hand_worked_case():
    """Five splats on an 8 x 6 image with a 2 x 2 prompt at pixels x 3..4, y 2..3: inside + under the prompt, inside +
    beside it, on the image's border (not 'within': uv > 0 and < W - 1 are strict), outside, and truncation (3.9 -> 3)."""
    import torch
    from oracle import frame_oracle as FR
    W, H = 8, 6
    prompt = torch.zeros(H, W, dtype=torch.bool)
    prompt[2:4, 3:5] = True
    uv = torch.tensor([[3.2, 2.7], [5.5, 2.5], [0.0, 2.0], [9.0, 3.0], [4.9, 3.9]])
    pts = FR.mask_prompt_points(uv, prompt, W, H)
    assert pts.tolist() == [True, False, False, False, True]
    # later: splat 0 has left the image, splat 4 has moved
    uv2 = torch.tensor([[7.5, 2.0], [5.0, 2.0], [1.0, 1.0], [2.0, 2.0], [2.25, 4.5], [1.0, 1.0]])
    out = FR.propagated_points(uv2, pts, W, H)
    assert out.tolist() == [[2.25, 4.5]]
