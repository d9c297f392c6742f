"""The checker the GPU frame tests use (tests/frame_check.py), on stand-in scenes: on the MI355X every frame of the suite
equals the oracle's in every pixel, so the branch that EXPLAINS a differing pixel never runs there — it is exercised here."""
import numpy as np
import pytest
import torch

import frame_check
from frame_check import frame_close

W, H, SPP, S0 = 6, 5, 8, 3


class FakeScene:
    """per-sample radiance table [H, W, SPP, 3] behind both interfaces (device: torch tensor, oracle: numpy)"""

    def __init__(self, table, as_torch):
        self.table, self.as_torch = table, as_torch

    def sample_pixel(self, w, h, x, y, s0, n, seed=None, max_depth=50, precision="f64"):
        assert (w, h, s0, n) == (W, H, S0, SPP)
        v = self.table[y, x].copy()
        return torch.from_numpy(v) if self.as_torch else v


@pytest.fixture()
def tables(tmp_path, monkeypatch):
    monkeypatch.setattr(frame_check, "LOG", str(tmp_path / "allowance.jsonl"))
    rs = np.random.RandomState(3)
    return rs.rand(H, W, SPP, 3) * 4.0


def frames(dev_tab, orc_tab):
    return dev_tab.sum(axis=2).astype(np.float32), orc_tab.sum(axis=2), FakeScene(dev_tab, True), FakeScene(orc_tab, False)


def test_equal_frames_pass(tables):
    got, ref, ds, oc = frames(tables, tables)
    assert frame_close("eq", got, ref, ds, oc, S0, SPP) == 1.0


def test_one_flipped_sample_is_explained(tables):
    dev = tables.copy()
    dev[2, 4, 5] = [0.0, 7.0, 0.5]  # one path took another branch
    got, ref, ds, oc = frames(dev, tables)
    assert frame_close("one", got, ref, ds, oc, S0, SPP) == 1.0 - 1.0 / (W * H)


def test_two_samples_of_a_pixel_differing_fail(tables):
    dev = tables.copy()
    dev[2, 4, 5] += 0.01
    dev[2, 4, 6] += 0.01
    got, ref, ds, oc = frames(dev, tables)
    with pytest.raises(AssertionError, match="samples that differ"):
        frame_close("two", got, ref, ds, oc, S0, SPP)


def test_a_frame_difference_the_samples_do_not_account_for_fails(tables):
    got, ref, ds, oc = frames(tables, tables)
    got[1, 1, 0] += 0.5  # e.g. a sample added twice: every per-sample value still equals the oracle's
    with pytest.raises(AssertionError, match="its samples by"):
        frame_close("lost", got, ref, ds, oc, S0, SPP)


def test_more_differing_pixels_than_the_allowance_fail(tables):
    dev = tables.copy()
    for x in range(3):
        dev[0, x, 1] = 9.0  # three pixels, each with ONE flipped sample: explained, but more than a frame may have
    got, ref, ds, oc = frames(dev, tables)
    with pytest.raises(AssertionError, match="pixels differ"):
        frame_close("many", got, ref, ds, oc, S0, SPP)
