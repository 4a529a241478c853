"""PRESMOOTH_GAUSSIAN (dense_segmentation.cpp:186-188: cv::GaussianBlur(f32 BGR, Size(3, 3), 1.5)).

OpenCV is un-vendored third-party arithmetic and not in this image: the oracle restates the published
algorithm of the 2.4 line (getGaussianKernel in double -> float, symmetric 3-tap row then column filter,
every operation rounded to f32, BORDER_REFLECT_101) -- PARITY UNPINNED, like the bilateral filter's
convertTo step.  CPU: the oracle against an independent numpy evaluation of the same formulas; where an
OpenCV is importable, against cv2.GaussianBlur itself.  GPU: k_gaussian3 against the oracle bit for bit,
and streams with the option byte for byte."""
import numpy as np
import pytest

import oracle_lib as ol

f32 = np.float32


def numpy_gaussian3(img):
    s = (img.astype(f32) * f32(1.0 / 255.0)).astype(f32)
    H, W = s.shape[:2]
    cf = np.exp(-0.5 / (1.5 * 1.5) * np.array([1.0, 0.0, 1.0])).astype(f32)
    cf = (cf.astype(np.float64) * (1.0 / float(cf.astype(np.float64).sum()))).astype(f32)
    k0, k1 = cf[1], cf[0]

    def reflect(n):
        idx = np.arange(-1, n + 1)
        idx[0] = 1 if n > 1 else 0
        idx[-1] = n - 2 if n > 1 else 0
        return idx
    sp = s[:, reflect(W)]
    r = (sp[:, 1:-1] * k0 + ((sp[:, :-2] + sp[:, 2:]).astype(f32) * k1).astype(f32)).astype(f32)
    rp = r[reflect(H)]
    return (rp[1:-1] * k0 + ((rp[:-2] + rp[2:]).astype(f32) * k1).astype(f32)).astype(f32)


@pytest.mark.parametrize("W,H", [(17, 9), (1, 5), (6, 1), (2, 2), (64, 48)])
def test_oracle_gaussian_equals_numpy_evaluation(W, H):
    rng = np.random.default_rng(W * 100 + H)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    got = ol.preprocess(img, 1)
    assert np.array_equal(got.view(np.uint32), numpy_gaussian3(img).view(np.uint32))
    flat = np.full((H, W, 3), 200, np.uint8)       # a constant image stays constant (to an ulp)
    assert np.abs(ol.preprocess(flat, 1) - f32(200) * f32(1.0 / 255.0)).max() < 1e-6


def test_oracle_gaussian_against_opencv_when_present():
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    tmp = img.astype(np.float32) * np.float32(1.0 / 255.0)
    want = cv2.GaussianBlur(tmp, (3, 3), 1.5)
    got = ol.preprocess(img, 1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "differs from cv2 %s by up to %g" % (cv2.__version__, float(np.abs(got - want).max()))


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    return v


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,kind,pad", [(64, 48, "noise", 0), (161, 97, "smooth", 5), (70, 20, "const", 2),
                                          (8, 8, "noise", 0), (320, 240, "noise", 0)])
def test_gaussian_bit_exact(vsg, W, H, kind, pad):
    from test_gpu_parity import rand_frame, bits
    rng = np.random.default_rng(3)
    buf = np.zeros((H, W * 3 + pad), np.uint8)
    buf[:, :W * 3] = rand_frame(rng, W, H, kind).reshape(H, W * 3)
    view = np.lib.stride_tricks.as_strided(buf, (H, W, 3), (buf.strides[0], 3, 1))
    g = vsg.DenseSegGraph(W, H, 2)
    g.add_frame_bgr(view, presmoothing=1)
    assert np.array_equal(bits(g.smoothed(0)), bits(ol.preprocess(view, 1)))


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,N,kind,flow,chunk", [(64, 48, 25, "probe", True, 10), (96, 64, 22, "smooth", True, 10),
                                                   (80, 60, 12, "noise", False, 20)])
def test_streams_with_gaussian_presmoothing(vsg, W, H, N, kind, flow, chunk):
    from test_gpu_parity import run_streams
    run_streams(vsg, W, H, N, kind, flow, chunk, presmoothing=1)
