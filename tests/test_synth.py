"""The torch generators of the synthetic inputs (used by bench.py to build frames on the device)
are bit-identical to the numpy generators the parity tests and the oracle use."""
import numpy as np
import pytest

import synth


@pytest.mark.parametrize("kind", ["bench", "noise", "blobs", "soft"])
@pytest.mark.parametrize("W,H,t", [(64, 48, 0), (161, 97, 5), (320, 240, 37)])
def test_torch_generators_match_numpy(kind, W, H, t):
    import torch
    want = synth.FRAME_FNS[kind](W, H, t)
    got = synth.frame_torch(kind, W, H, t, torch.device("cpu")).numpy()
    assert got.dtype == np.uint8 and got.shape == (H, W, 3)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("W,H,t", [(64, 48, 0), (161, 97, 5), (320, 240, 37), (1920, 1080, 19)])
def test_var_flow_torch_matches_numpy_and_varies(W, H, t):
    import torch
    want = synth.var_flow(W, H, t)
    got = synth.flow_torch("var", W, H, t, torch.device("cpu")).numpy()
    assert got.dtype == np.float32 and got.shape == (H, W, 2)
    assert np.array_equal(got, want)
    assert np.array_equal(synth.flow_torch("const", W, H, t, torch.device("cpu")).numpy(), synth.const_flow(W, H))
    # multiples of 1/64, smooth part within +-10 px, bands far out of range, many distinct vectors
    assert np.array_equal(want * 64, np.round(want * 64))
    rows = np.arange(H)
    band_r = (rows >= H // 3) & (rows < H // 3 + 8)
    cols = np.arange(W)
    band_c = (cols >= (2 * W) // 3) & (cols < (2 * W) // 3 + 8)
    inner = want[~band_r][:, ~band_c]
    assert np.abs(inner).max() <= 12.0
    assert (want[band_r][..., 0] == 3 * W).all()
    assert (want[~band_r][:, band_c][..., 1] == -3 * H).all()
    assert len(np.unique(inner.reshape(-1, 2), axis=0)) > min(W, H) // 2
    if t > 0:
        assert not np.array_equal(want, synth.var_flow(W, H, t - 1))


def test_workload_inputs_differ_in_kind():
    """noise: nearly every pixel its own colour; blobs: cells of one colour."""
    n = synth.noise_frame(128, 96, 0)
    b = synth.blobs_frame(128, 96, 0)
    assert np.abs(np.diff(n[..., 2].astype(int), axis=1)).mean() > 15
    assert np.abs(np.diff(b[..., 2].astype(int), axis=1)).mean() < 8
