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


def test_workload_inputs_differ_in_kind():
    """noise: nearly every pixel its own colour; blobs: cells of one colour."""
    n = synth.noise_frame(128, 96, 0)
    b = synth.blobs_frame(128, 96, 0)
    assert np.abs(np.diff(n[..., 2].astype(int), axis=1)).mean() > 15
    assert np.abs(np.diff(b[..., 2].astype(int), axis=1)).mean() < 8
