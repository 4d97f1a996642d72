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
