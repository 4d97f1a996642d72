"""Plate rectification: the host homography solver (CPU: no GPU needed) and the warp kernel (GPU)
against the oracle's restatement of OpenCV's published algorithm.  Parity with OpenCV itself is unpinned (not installed)."""
import numpy as np
import pytest

from oracle import densebox_oracle as O


def _pts(rs):
    c = np.array([[60, 50], [180, 58], [176, 110], [56, 100]], dtype=np.float64)
    return (c + rs.uniform(-6, 6, size=(4, 2))).tolist()


def test_perspective_matrix_host_solver_matches_oracle_and_maps_corners():
    from densebox_amd import rectify
    rs = np.random.RandomState(0)
    for _ in range(20):
        src = _pts(rs)
        dst = rectify.dst_rectangle(src)
        assert dst == O.perspective_dst_rectangle(src)
        M = rectify.get_perspective_matrix(src, dst)
        assert np.array_equal(M, O.get_perspective_matrix(src, dst))          # same elimination order: bit for bit
        p = np.concatenate([np.float32(src).astype(np.float64), np.ones((4, 1))], axis=1) @ M.T
        assert np.allclose(p[:, :2] / p[:, 2:], np.float32(dst), atol=1e-8)
    ident = rectify.get_perspective_matrix([[0, 0], [10, 0], [10, 5], [0, 5]], [[0, 0], [10, 0], [10, 5], [0, 5]])
    assert np.allclose(ident, np.eye(3), atol=1e-14)
    with pytest.raises(RuntimeError):
        rectify.get_perspective_matrix([[0, 0], [1, 1], [2, 2], [3, 3]], [[0, 0], [1, 0], [1, 1], [0, 1]])


def test_oracle_warp_properties():
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    out = O.warp_perspective_u8(img, np.eye(3), (80, 56))                      # identity: copy + zero border
    assert np.array_equal(out[:37, :53], img) and out[37:].max() == 0 and out[:, 53:].max() == 0
    shift = np.array([[1, 0, 4], [0, 1, 2], [0, 0, 1]], dtype=np.float64)      # integer translation
    out = O.warp_perspective_u8(img, shift, (60, 40))
    assert np.array_equal(out[2:39, 4:57], img) and out[:2].max() == 0 and out[:, :4].max() == 0
    half = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], dtype=np.float64)     # half-pixel shift: mean of neighbours, rounded
    out = O.warp_perspective_u8(img, half, (53, 37))
    exp = (img[:, :-1].astype(np.int64) + img[:, 1:].astype(np.int64) + 1) >> 1
    assert np.array_equal(out[:, 1:], exp)


@pytest.mark.gpu
def test_warp_kernel_bit_exact_vs_oracle():
    from densebox_amd import rectify
    rs = np.random.RandomState(2)
    for (h, w, c) in [(120, 200, 3), (61, 47, 1)]:
        img = rs.randint(0, 256, size=(h, w, c)).astype(np.uint8)
        for _ in range(4):
            src = (np.array([[0.3 * w, 0.4 * h], [0.8 * w, 0.45 * h], [0.78 * w, 0.8 * h], [0.28 * w, 0.75 * h]]) +
                   rs.uniform(-4, 4, size=(4, 2))).tolist()
            got = rectify.perspective_transform(img, src)
            ref = O.perspective_transform(img, src)
            assert got.shape == ref.shape == (int(h * 1.5 + 0.5), int(w * 1.5 + 0.5), c)
            assert np.array_equal(got, ref)
    import torch
    t = torch.as_tensor(img).cuda()
    out = rectify.perspective_transform(t, src)
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), ref)


def test_oracle_against_an_independent_float64_warp():
    """No OpenCV here (parity with it stays unpinned), but the restatement can be checked against INDEPENDENT arithmetic: the homography
    against numpy.linalg.solve of the same eight equations, the warp against a float64 bilinear resampling (scipy.ndimage.map_coordinates,
    order 1, constant 0 border) through the exactly inverted matrix.  OpenCV's scheme differs from exact bilinear interpolation by its
    documented quantisations only -- source coordinates rounded to 1/32 pixel, weights to 2^-15 -- so on an image whose neighbouring pixels
    differ by at most G levels the two agree to G / 32 + 1 levels everywhere (and to +-1 on a smooth image)."""
    from scipy import ndimage
    rs = np.random.RandomState(5)
    for trial in range(4):
        src = _pts(rs)
        dst = O.perspective_dst_rectangle(src)
        M = O.get_perspective_matrix(src, dst)
        s, d = np.float32(src).astype(np.float64), np.float32(dst).astype(np.float64)
        A, b = np.zeros((8, 8)), np.zeros(8)
        for i in range(4):
            A[i] = [s[i, 0], s[i, 1], 1, 0, 0, 0, -s[i, 0] * d[i, 0], -s[i, 1] * d[i, 0]]; b[i] = d[i, 0]
            A[i + 4] = [0, 0, 0, s[i, 0], s[i, 1], 1, -s[i, 0] * d[i, 1], -s[i, 1] * d[i, 1]]; b[i + 4] = d[i, 1]
        Mref = np.concatenate([np.linalg.solve(A, b), [1.0]]).reshape(3, 3)
        assert np.allclose(M, Mref, rtol=1e-9, atol=1e-9)
        h, w = 120, 240
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        smooth = np.stack([96 + 80 * np.sin(xx / 23.0 + c) * np.cos(yy / 17.0) for c in range(3)], axis=-1)
        G = 6                                                            # neighbouring pixels of `smooth` differ by < 6 levels
        img = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)
        assert int(np.abs(np.diff(img.astype(np.int64), axis=0)).max()) < G and int(np.abs(np.diff(img.astype(np.int64), axis=1)).max()) < G
        dw, dh = 300, 160
        got = O.warp_perspective_u8(img, M, (dw, dh)).astype(np.float64)
        Minv = np.linalg.inv(Mref)
        oy, ox = np.mgrid[0:dh, 0:dw].astype(np.float64)
        den = Minv[2, 0] * ox + Minv[2, 1] * oy + Minv[2, 2]
        sx, sy = (Minv[0, 0] * ox + Minv[0, 1] * oy + Minv[0, 2]) / den, (Minv[1, 0] * ox + Minv[1, 1] * oy + Minv[1, 2]) / den
        ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [sy, sx], order=1, mode='constant', cval=0.0)
                        for c in range(3)], axis=-1)
        inside = (sx >= 1) & (sx <= w - 2) & (sy >= 1) & (sy <= h - 2)     # (the border rows blend with the constant 0: compared separately)
        assert inside.mean() > 0.3
        diff = np.abs(got - ref)
        assert diff[inside].max() <= G / 32.0 + 1.0, diff[inside].max()
        assert (diff[inside] <= 1.0).mean() > 0.999
        outside = (sx < -1) | (sx > w) | (sy < -1) | (sy > h)
        assert got[outside].max() == 0
