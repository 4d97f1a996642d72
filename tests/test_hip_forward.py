"""GPU parity: HIP forward (through the C ABI) vs the CPU oracle and the reference-captured fixtures.

Tolerances (outputs are O(1); loc-type outputs O(10)); error = max |hip - ref| / max(1, max|ref|) per output map:
  f32  : 1e-4                       exact-fp32 MFMA, only summation order differs (achieved <= 8e-6)
  f16  : PER MAP (MAP_TOL below, from profiles/r04_lowprec_errors.json = tools/gpu_lowprec_err.py on these fixtures): north_star's 1e-3 where
         the map meets it (the bbox maps: 0.74e-3 achieved), achieved + 20 % elsewhere (score / heat / offset maps 1.2-1.8e-3, the refined
         score of DenseBoxLMLOC 3.0e-3: fp16 activation rounding accumulated over 13 layers; its INPUTS, the f16 head outputs, carry
         1.5-1.9e-3 and its three convs amplify that) -- a regression of a map that meets 1e-3 fails at 1e-3, not at the worst map's bar.
         RMS error bar 1e-3 on every map (north_star's figure; achieved 2-9e-4)
  bf16 : 8x coarser mantissa: per map, achieved + 20 % (0.57-1.8e-2); RMS bar 8e-3 (achieved <= 5.5e-3)
"""
import os

import numpy as np
import pytest
import torch

from conftest import unpack
from densebox_amd import synth
import densebox_amd as D

pytestmark = pytest.mark.gpu

KINDS = ['DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC']
TOL = {'f32': 1e-4, 'f16': 3.8e-3, 'bf16': 3e-2}            # flat bars: intermediate taps, odd sizes, whole images
# per output map of the 240 x 240 fixtures: north_star's 1e-3 where achieved, else achieved (r04_lowprec_errors.json) + 20 %
MAP_TOL = {
    ('DenseBox', 'f16'): [2.19e-3, 1.0e-3], ('DenseBox', 'bf16'): [1.25e-2, 6.8e-3],
    ('DenseBoxLM', 'f16'): [2.19e-3, 1.0e-3, 1.61e-3, 1.46e-3], ('DenseBoxLM', 'bf16'): [1.25e-2, 6.8e-3, 1.32e-2, 7.6e-3],
    ('DenseBoxLMLOC', 'f16'): [2.19e-3, 3.59e-3, 1.0e-3, 1.71e-3, 2.13e-3], ('DenseBoxLMLOC', 'bf16'): [1.25e-2, 2.12e-2, 6.8e-3, 1.15e-2, 1.33e-2],
}
RMS_TOL = {'f32': 2e-5, 'f16': 1e-3, 'bf16': 8e-3}        # per-map RMS error / max(1, max|ref|)


def _net(kind, dtype, seed=11):
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, seed)
    net = net.cuda().eval()
    net.compute_dtype = dtype
    return net


def _close(a, ref, tol, what, rms_tol=None):
    a = a.detach().float().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(a - ref).max())
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    assert err <= tol * scale, '%s: max err %.3e > %.1e * %.2f' % (what, err, tol, scale)
    if rms_tol is not None:
        rms = float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2)))
        assert rms <= rms_tol * scale, '%s: rms err %.3e > %.1e * %.2f' % (what, rms, rms_tol, scale)
    return err


@pytest.mark.parametrize('dtype', ['f32', 'f16', 'bf16'])
@pytest.mark.parametrize('kind', KINDS)
def test_forward_vs_reference_fixture(golden, kind, dtype):
    g = golden('net_' + kind)
    net = _net(kind, dtype, int(g['param_seed']))
    with torch.no_grad():
        outs = net(synth.synth_images(2, 240, 240, seed=3).cuda())
    assert len(outs) == sum(1 for k in g.files if k.startswith('out240_'))
    for i, o in enumerate(outs):
        assert o.dtype == torch.float32 and o.is_contiguous()
        tol = TOL[dtype] if dtype == 'f32' else MAP_TOL[(kind, dtype)][i]
        _close(o, g['out240_%d' % i], tol, '%s/%s out %d' % (kind, dtype, i), RMS_TOL[dtype])
    # intermediate taps (first conv + first pool) pin the layer kernels individually
    eng = net.engine()
    a11 = eng.read_activation('a11')[0, ::8, ::6, ::6]
    _close(a11, g['tap_conv1_1_sub'], TOL[dtype], 'conv1_1 tap')
    p1 = eng.read_activation('p1')[0, ::8, ::4, ::4]
    _close(p1, g['tap_pool1_sub'], TOL[dtype], 'pool1 tap')


@pytest.mark.parametrize('kind', KINDS)
def test_forward_odd_size(golden, kind):
    """100x132 input: floor pooling (25x33 maps), upsample target = conv3_4's size (DenseBox.py:213-216)."""
    g = golden('net_' + kind)
    net = _net(kind, 'f32', int(g['param_seed']))
    with torch.no_grad():
        outs = net(synth.synth_images(1, 100, 132, seed=4).cuda())
    for i, o in enumerate(outs):
        _close(o, g['outodd_%d' % i], TOL['f32'], '%s odd out %d' % (kind, i))


def test_forward_dropout_injected(golden):
    """Train-mode forward with the reference's recorded Dropout masks (fixture train_DenseBox_dropout)."""
    g = golden('train_DenseBox_dropout')
    net = _net('DenseBox', 'f32', int(g['param_seed']))
    net.train()
    n = int(g['batch'])
    net.dropout_masks = {h: torch.from_numpy(unpack(g['dropmask_%d' % i], (n, 512, 60, 60)))
                         for i, h in enumerate(['det', 'loc'])}
    x, _, _, _ = synth.synth_batch(int(g['n_patch']), seed=int(g['seed']))
    with torch.no_grad():
        outs = net(x[:n].cuda())
    for i, o in enumerate(outs):
        _close(o, g['s0_out_%d' % i], TOL['f32'], 'dropout out %d' % i)


def test_forward_cpu_tensor_raises():
    net = getattr(D, 'DenseBox')(synth.vgg19_standin(seed=0))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 240, 240))


def test_whole_image_1080p(golden):
    """Config 5: whole-image FCN forward at 1080x1920 (maps 270x480), f16."""
    g = golden('net_DenseBox_1080p')
    net = _net('DenseBox', 'f16', 11)
    with torch.no_grad():
        s, l = net(synth.synth_images(1, 1080, 1920, seed=5).cuda())
    assert tuple(s.shape) == tuple(g['score_shape'])
    _close(s[0, 0, ::9, ::8], g['score_sub'], TOL['f16'], '1080p score')
    _close(l[0, :, ::9, ::8], g['loc_sub'], TOL['f16'], '1080p loc')


@pytest.mark.parametrize('size', [(512, 512), (384, 640), (256, 1024)])
def test_single_image_tiles_agree_with_the_fp32_path(size):
    """Single-image maps run their own band tiles (144 x 64 / 144 x 128 on six waves, 288 x 128: one workgroup per CU with a deep LDS ring) and
    the register-tournament top-K: the f16 forward of a 512 x 512-class image against the fp32 path of the same network (no reference
    fixture at these sizes: the fp32 path is pinned by the 240 x 240 / 100 x 132 / 1080p fixtures), and detect() graph replay ==
    eager == the plain forward + parse + NMS calls."""
    from densebox_amd import decode as DC
    h, w = size
    x = synth.synth_images(1, h, w, seed=7).cuda()
    net = _net('DenseBoxLMLOC', 'f32', 11)
    with torch.no_grad():
        ref = [o.clone() for o in net(x)]
    net.compute_dtype = 'f16'
    with torch.no_grad():
        outs = [o.clone() for o in net(x)]
    for i, (o, r) in enumerate(zip(outs, ref)):
        _close(o, r.cpu().numpy(), TOL['f16'], 'single image %dx%d out %d' % (h, w, i), RMS_TOL['f16'])
    from densebox_amd import _lib
    plans = {net.engine().conv_plan(_lib.F16, net.engine().last_plan.B[a].view(), net.engine().last_plan.B[b].view(), 3, 3, 1, ci, co, 3)[1]
             for a, b, ci, co in (('a41', 'a42', 512, 512), ('a31', 'a32', 256, 256), ('a21', 'a22', 128, 128))}
    if size == (512, 512):
        want_plans = {'conv3x3_band_kernel<f16,144,64>', 'conv3x3_band_kernel<f16,144,128>', 'conv3x3_band_kernel<f16,288,128>'}
        if os.environ.get('DBX_BAND144') == '0':
            want_plans = {'conv3x3_band_kernel<f16,192,64>', 'conv3x3_band_kernel<f16,192,128>', 'conv3x3_band_kernel<f16,288,128>'}
        assert plans == want_plans, plans
    dets, keep = net.detect(x, K=10, nms_thresh=0.4)
    dets2, keep2 = net.detect(x, K=10, nms_thresh=0.4)                      # replay
    os.environ['DBX_GRAPH'] = '0'
    try:
        dets3, keep3 = net.detect(x, K=10, nms_thresh=0.4)                  # eager
    finally:
        del os.environ['DBX_GRAPH']
    assert np.array_equal(dets, dets2) and keep == keep2 and np.array_equal(dets, dets3) and keep == keep3
    want = DC.parse_DetLMLOC(outs[1], outs[2], outs[3], outs[4], h, w, K=10)
    assert np.array_equal(dets, want)
