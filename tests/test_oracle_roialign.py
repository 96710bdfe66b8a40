"""Known-answer cases for the RoIAlign oracle (torchvision algorithm restatement; the reference
holds no golden vectors at this boundary -- SURVEY.md §8c O3 -- so these are hand-derived)."""
import torch

from oracle.roi_align_ref import roi_align_1x1, roi_align_1x1_loops


def _roi(b, x0, y0, x1, y1):
    return torch.tensor([[b, x0, y0, x1, y1]], dtype=torch.float64)


def test_constant_map_whole_image():
    feat = torch.full((1, 14, 14, 8), 3.25)
    out = roi_align_1x1_loops(feat, _roi(0, 0, 0, 14, 14))
    assert torch.allclose(out, torch.full((1, 8), 3.25))


def test_box_on_one_cell_centre():
    # aligned=True: pixel centre of cell (r=3,c=5) is at (x=5.5,y=3.5); a unit box [5,6]x[3,4]
    # has one sample exactly at the centre -> that cell's value.
    feat = torch.arange(14 * 14 * 2, dtype=torch.float32).reshape(1, 14, 14, 2)
    out = roi_align_1x1_loops(feat, _roi(0, 5, 3, 6, 4))
    assert torch.allclose(out[0], feat[0, 3, 5])


def test_half_cell_shift_is_bilinear_mean():
    # box [5.5,6.5]x[3,4]: single sample at x=6.0-0.5=5.5 -> halfway between cells 5 and 6
    feat = torch.arange(14 * 14, dtype=torch.float32).reshape(1, 14, 14, 1)
    out = roi_align_1x1_loops(feat, _roi(0, 5.5, 3, 6.5, 4))
    assert torch.allclose(out[0, 0], 0.5 * (feat[0, 3, 5, 0] + feat[0, 3, 6, 0]))


def test_two_by_two_grid_average():
    # box [2,4]x[2,4] -> 2x2 samples at centres of cells (2,2),(2,3),(3,2),(3,3)
    feat = torch.randn(1, 6, 6, 3)
    out = roi_align_1x1_loops(feat, _roi(0, 2, 2, 4, 4))
    assert torch.allclose(out[0], feat[0, 2:4, 2:4].reshape(4, 3).mean(0), atol=1e-6)


def test_degenerate_box_gives_zero():
    feat = torch.randn(1, 6, 6, 3)
    assert torch.all(roi_align_1x1_loops(feat, _roi(0, 3, 3, 3, 3)) == 0)      # zero area: no samples, count clamps to 1
    assert torch.all(roi_align_1x1_loops(feat, _roi(0, 4, 4, 2, 2)) == 0)      # negative extent


def test_border_crossing_box_zero_pads():
    # box [-2,2]x[0,1]: 4 x-samples at -1.5-0.5+... = x in {-2.0,-1.0,0.0,1.0} (after -0.5 shift: start -2.5,
    # samples -2.0,-1.0,0.0,1.0); x<-1 -> 0 contribution, x=-1 -> clamped to 0 -> cell 0
    feat = torch.arange(1, 6 * 6 + 1, dtype=torch.float32).reshape(1, 6, 6, 1)
    out = roi_align_1x1_loops(feat, _roi(0, -2, 0, 2, 1))
    row0 = feat[0, 0, :, 0]
    expect = (0.0 + row0[0] + row0[0] + row0[1]) / 4.0
    assert torch.allclose(out[0, 0], expect)


def test_far_right_edge_snaps():
    # samples beyond W-1 but <= W snap to the last cell
    feat = torch.arange(1, 6 * 6 + 1, dtype=torch.float32).reshape(1, 6, 6, 1)
    out = roi_align_1x1_loops(feat, _roi(0, 5.5, 0, 6.5, 1))   # sample x = 5.5 -> low=5>=W-1 -> cell 5
    assert torch.allclose(out[0, 0], feat[0, 0, 5, 0])


def test_vectorised_equals_loops_and_gradcheck():
    torch.manual_seed(0)
    feat = torch.randn(2, 7, 7, 4, dtype=torch.float64)
    xy0 = torch.rand(9, 2, dtype=torch.float64) * 5 - 0.5
    wh = torch.rand(9, 2, dtype=torch.float64) * 4
    rois = torch.cat([torch.randint(0, 2, (9, 1)).double(), xy0, xy0 + wh], dim=1)
    a = roi_align_1x1(feat, rois)
    b = roi_align_1x1_loops(feat, rois)
    assert torch.allclose(a, b, atol=1e-12)
    feat.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda f: roi_align_1x1(f, rois), (feat,), atol=1e-8)
