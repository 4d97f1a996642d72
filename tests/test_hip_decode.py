"""GPU parity of top-K decode + NMS (bit-exact float64 rows and keep lists, DenseBox.py:3114-3443)."""
import numpy as np
import pytest
import torch

from densebox_amd import synth
from densebox_amd import decode as DC
import densebox_amd as D

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_decode_bit_exact(golden):
    g = golden('decode')
    s, l, hm, ll = T(g['a_s']), T(g['a_l']), T(g['a_hm']), T(g['a_ll'])
    assert np.array_equal(DC.parse_output(s, l, K=10), g['a_parse_output'])
    assert np.array_equal(DC.parse_out_MN(s, l, 240, 240, K=10), g['a_parse_out_MN'])
    assert np.array_equal(DC.parse_out_MN(s.cuda(), l.cuda(), 240, 240, K=50), g['a_parse_out_MN_K50'])
    assert np.array_equal(DC.parse_DetLM(s, l, hm, 240, 240, K=10), g['a_parse_DetLM'])
    assert np.array_equal(DC.parse_DetLMLOC(s, l, hm, ll, 240, 240, K=10), g['a_parse_DetLMLOC'])
    s, l, hm, ll = T(g['b_s']), T(g['b_l']), T(g['b_hm']), T(g['b_ll'])
    assert np.array_equal(DC.parse_out_MN(s, l, 101, 134, K=10), g['b_parse_out_MN'])
    assert np.array_equal(DC.parse_DetLM(s, l, hm, 101, 134, K=7), g['b_parse_DetLM'])
    assert np.array_equal(DC.parse_DetLMLOC(s, l, hm, ll, 101, 134, K=7), g['b_parse_DetLMLOC'])
    with pytest.raises(AssertionError):       # shape assert of the reference (DenseBox.py:3311)
        DC.parse_out_MN(s, l, 240, 240, K=10)


def test_nms_bit_exact(golden):
    g = golden('decode')
    for th, key in ((0.4, 'nms_keep_04'), (0.0, 'nms_keep_00'), (0.7, 'nms_keep_07')):
        assert DC.NMS(g['nms_in'], th) == list(g[key])
    assert DC.NMS(g['nms_big_in'], 0.4) == list(g['nms_big_keep'])
    for k_ in ('a_parse_output', 'a_parse_DetLM', 'a_parse_DetLMLOC', 'a_parse_out_MN_K50'):
        assert DC.NMS(g[k_], 0.4) == list(g[k_ + '_keep'])


def test_detect_1080p(golden):
    """Config 5: whole-image 1920x1080 forward + on-GPU top-K/NMS.  fp32 compute so the top-K set is the reference's."""
    g = golden('net_DenseBox_1080p')
    net = D.DenseBox(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.cuda().eval()
    net.compute_dtype = 'f32'
    dets, keep = net.detect(synth.synth_images(1, 1080, 1920, seed=5).cuda(), K=10, nms_thresh=0.4)
    assert dets.shape == (10, 5) and dets.dtype == np.float64
    assert np.allclose(dets, g['dets'], rtol=1e-4, atol=2e-3)
    assert keep == list(g['keep'])
