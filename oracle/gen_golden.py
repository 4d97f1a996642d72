#!/usr/bin/env python3
"""Generate tests/golden/*.npz by EXECUTING the reference (/root/reference/DenseBox.py).

TEST INFRASTRUCTURE -- runs only in the build container (the reference never
travels; the GPU box sees only the committed .npz vectors).  Nothing of the
reference is copied: the module is imported unmodified, with

  * stub ``torchvision`` / ``cv2`` modules pre-seeded in ``sys.modules`` (neither
    is installed; DenseBox.py:18-21 imports them at module level),
  * the NumPy>=2 shim: the loc-type label generators get ``.double()`` inputs
    (SURVEY.md 0.9 -- assigning a numpy.float32 into a torch tensor raises under
    NEP-50; values are quarter-integers, exact in either precision),
  * for the training-loop captures: the Dataset classes, ``DataLoader``,
    ``torchvision.models.vgg19``, ``torch.save`` and ``nn.Dropout`` replaced by
    synthetic / recording stand-ins, and recording wrappers around
    ``mask_by_sel``, ``mask_gray_zone_*``, ``Tensor.backward`` and ``SGD.step``.
    The loop bodies themselves (DenseBox.py:2016-2187, :2568-2730, :2836-2926)
    run unmodified.

Usage:  python oracle/gen_golden.py   (writes tests/golden/)
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference/DenseBox.py'

from densebox_amd import synth  # noqa: E402

warnings.filterwarnings('ignore')


# ----------------------------------------------------------------------------- import the reference
def _stub_modules():
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvm = types.ModuleType('torchvision.models')

    class _NoOp:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    for n in ('Compose', 'Resize', 'CenterCrop', 'ToTensor', 'Normalize'):
        setattr(tvt, n, _NoOp)
    tvm.vgg19 = lambda pretrained=True: synth.vgg19_standin(seed=0)
    tv.transforms = tvt
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    sys.modules['torchvision.models'] = tvm
    sys.modules['cv2'] = types.ModuleType('cv2')


def load_reference():
    _stub_modules()
    spec = importlib.util.spec_from_file_location('ref_densebox', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert str(mod.device) == 'cpu'
    return mod


R = load_reference()


def pack(a):
    """{0,1}-valued float map -> packed bits + shape."""
    a = np.asarray(a)
    assert np.all((a == 0) | (a == 1))
    return np.packbits(a.astype(np.uint8).reshape(-1))


def t2n(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- A. label maps and masks
def crafted_labels():
    """240-space integer labels; rows chosen to pin the rounding/clamping cases
    of SURVEY.md 0.10 / 8(c)."""
    rows = [
        # x0, y0, x1, y1,  LU        RU         RD         LD        label
        (80, 88, 160, 120, 78, 86, 162, 87, 161, 122, 79, 121, 1),     # plain (20,22,40,30)/60
        (0, 40, 200, 100, 8, 40, 200, 42, 198, 100, 9, 99, 1),         # float32/float64 boundary pair (0,200)
        (3, 20, 103, 60, 8, 20, 103, 22, 101, 60, 9, 59, 1),           # boundary pair (3,103)
        (4, 16, 204, 64, 8, 16, 204, 18, 202, 64, 10, 63, 1),          # boundary pair (4,204)
        (120, 150, 240, 240, 120, 150, 236, 151, 235, 236, 121, 235, 1),  # touches right/bottom edge
        (100, 100, 112, 104, 100, 100, 112, 100, 112, 104, 100, 104, 1),  # tiny box
        (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0),                       # negative patch
        (33, 57, 131, 95, 31, 55, 133, 58, 130, 97, 35, 96, 1),        # odd coords (quarter-integers)
        (150, 30, 230, 70, 150, 30, 239, 31, 238, 70, 151, 69, 1),     # landmark x=239 -> clamps to 59 in _pn
        (12, 180, 92, 228, 12, 180, 92, 181, 91, 228, 13, 227, 1),
        (8, 8, 48, 24, 8, 8, 48, 8, 48, 24, 8, 24, 1),                 # landmark at the 8-px border -> gray zone at 0
        (61, 99, 178, 141, 60, 98, 180, 100, 177, 143, 62, 140, 1),
        # flat box at the top edge: gray-zone org_y = int(-1.15) = -1 -> the zeroing slice [-1:3]
        # wraps to [59:3] = empty (python slice semantics); landmark at y=1 -> 5x5 slice [-1:4] empty
        (0, 0, 40, 4, 8, 4, 40, 4, 40, 8, 4, 8, 1),
        (200, 232, 240, 240, 200, 232, 236, 232, 236, 236, 200, 236, 1),   # bottom-right corner
    ]
    a = np.asarray(rows, np.float64)
    bbox = torch.from_numpy((a[:, 0:4] / 4.0).astype(np.float32))
    vert = torch.from_numpy((a[:, 4:12] / 4.0).astype(np.float32))
    lab = torch.from_numpy(a[:, 12:13].astype(np.float32))
    return bbox, vert, lab


def gen_labels():
    bbox, vert, lab = crafted_labels()
    B = bbox.size(0)
    # rows usable by the non-_pn landmark functions (no clamp in DenseBox.py:1814-1823)
    ok = np.array([i for i in range(B) if float(lab[i]) == 1.0 and
                   all(int(float(v) + 0.5) < 60 for v in vert[i])])
    d = {'bbox': t2n(bbox), 'vert': t2n(vert), 'lab': t2n(lab), 'ok_rows': ok}

    d['score_map'] = pack(t2n(R.init_score_map(bbox, B, ratio=0.3)))
    d['score_pn'] = pack(t2n(R.init_score(bbox, lab, ratio=0.3)))
    d['loc_map'] = t2n(R.init_loc_map(bbox.double(), B))
    d['loc_pn'] = t2n(R.init_loc(bbox.double(), lab))
    vok = vert[ok]
    d['lm_heat'] = pack(t2n(R.init_lm_heatmap(vok, len(ok))))
    d['lm_heat_pn'] = pack(t2n(R.init_lm_heatmap_pn(vert, lab)))
    d['lm_loc'] = t2n(R.init_lm_locmap(vok.double(), len(ok)))
    d['lm_loc_pn'] = t2n(R.init_lm_locmap_pn(vert.double(), lab))

    # masks: start from gt.clone() (DenseBox.py:2848), select, then gray zone
    rs = np.random.RandomState(7)
    K = 14
    neg = np.stack([rs.choice(3600, K, replace=False) for _ in range(B)]).astype(np.int64)
    neg[0, 0] = -3          # out-of-range ids are skipped (DenseBox.py:1390)
    neg[1, 1] = 3600
    neg[2, 2] = neg[2, 3]   # duplicates are harmless
    d['neg_idx'] = neg
    for tag, gt, grayfn, args in (
            ('', R.init_score_map(bbox, B), R.mask_gray_zone_cls, ()),
            ('_pn', R.init_score(bbox, lab), R.mask_gray_zone_cls_pn, (lab,))):
        m = gt.clone()
        pos = torch.nonzero(gt)
        R.mask_by_sel(m, pos, torch.from_numpy(neg))
        d['mask_sel' + tag] = pack(t2n(m))
        d['pos_idx' + tag] = t2n(pos)
        grayfn(m, bbox, *args, ratio=0.3, gray_border=2.0)
        d['mask_gray' + tag] = pack(t2n(m))

    # landmark masks: per channel select + 5x5 gray zone (DenseBox.py:2115-2156)
    heat = R.init_lm_heatmap_pn(vert, lab)
    lm_mask = heat.clone()
    lm_neg = np.stack([rs.choice(3600, 2, replace=False) for _ in range(4 * B)]
                      ).reshape(4, B, 2).astype(np.int64)
    # force negatives right next to a landmark so the gray zone has something to erase
    lm_neg[0, 0, 0] = int(86 / 4 + 0.5) * 60 + int(78 / 4 + 0.5) + 1
    d['lm_neg_idx'] = lm_neg
    for i in range(4):
        view = lm_mask[:, i, :, :].unsqueeze(1)
        gti = heat[:, i, :, :].unsqueeze(1)
        pos = torch.nonzero(gti)
        R.mask_by_sel(view, pos, torch.from_numpy(lm_neg[i]))
        R.mask_gray_zone_lm(view, pos, i, gray_border=2.0)
    d['lm_mask'] = pack(t2n(lm_mask))

    lo = torch.from_numpy(rs.rand(B, 1, 60, 60).astype(np.float32))
    d['neg_loss_in'] = t2n(lo)
    d['neg_loss_out'] = t2n(R.gen_neg_loss(lo, R.init_score(bbox, lab)))
    np.savez_compressed(os.path.join(OUT, 'labels.npz'), **d)
    print('labels.npz', {k: getattr(v, 'shape', None) for k, v in d.items()})


# ----------------------------------------------------------------------------- B. networks (eval forward)
NETS = {'DenseBox': 'DenseBox', 'DenseBoxLM': 'DenseBoxLM', 'DenseBoxLMLOC': 'DenseBoxLMLOC'}
PARAM_SEED = 11


def build_ref_net(kind):
    net = getattr(R, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, PARAM_SEED)
    return net


def gen_nets():
    x240 = synth.synth_images(2, 240, 240, seed=3)
    xodd = synth.synth_images(1, 100, 132, seed=4)
    for kind in NETS:
        net = build_ref_net(kind).eval()
        d = {'keys': np.array(list(net.state_dict().keys())),
             'param_names': np.array([n for n, _ in net.named_parameters()]),
             'param_seed': PARAM_SEED,
             'param_sums': np.array([float(p.double().sum()) for _, p in net.named_parameters()])}
        with torch.no_grad():
            outs = net.forward(x240)
            for i, o in enumerate(outs):
                d['out240_%d' % i] = t2n(o)
            outs = net.forward(xodd)
            for i, o in enumerate(outs):
                d['outodd_%d' % i] = t2n(o)
            # intermediate taps of the backbone for layer-wise kernel checks (patch 0 only)
            X = x240[:1]
            X = net.conv1_1(X)
            d['tap_conv1_1_sub'] = t2n(X[0, ::8, ::6, ::6])
            X = net.conv1_2(X)
            X = net.pool1(X)
            d['tap_pool1_sub'] = t2n(X[0, ::8, ::4, ::4])
        np.savez_compressed(os.path.join(OUT, 'net_%s.npz' % kind), **d)
        print('net_%s.npz' % kind, len(d['keys']), 'keys', [d['out240_%d' % i].shape for i in range(len(outs))])

    # whole-image 1080x1920 forward (DenseBox only): strided sample + top-K indices
    net = build_ref_net('DenseBox').eval()
    ximg = synth.synth_images(1, 1080, 1920, seed=5)
    with torch.no_grad():
        s, l = net.forward(ximg)
    d = {'score_sub': t2n(s[0, 0, ::9, ::8]), 'loc_sub': t2n(l[0, :, ::9, ::8]),
         'score_shape': np.array(s.shape)}
    top = torch.topk(s.view(1, -1), k=10, dim=1)
    d['top_idx'] = t2n(top.indices[0])
    d['top_val'] = t2n(top.values[0])
    d['dets'] = R.parse_out_MN(s, l, 1080, 1920, K=10)
    d['keep'] = np.asarray(R.NMS(d['dets'], 0.4), np.int64)
    np.savez_compressed(os.path.join(OUT, 'net_DenseBox_1080p.npz'), **d)
    print('net_DenseBox_1080p.npz', d['score_sub'].shape, d['keep'])


# ----------------------------------------------------------------------------- B2. checkpoints (state_dict interop)
CKPT_SEED = 23


def gen_checkpoints():
    """What the reference writes with torch.save(net.state_dict()) (DenseBox.py:2206) and reads back with load_state_dict
    (:1989-1994): per net the key order, every entry's shape / dtype / float64 sum / L1 norm, the groups of keys that share
    storage (the aliased conv wrappers), the small tensors in full, and the eval forward of a net carrying these weights."""
    x = synth.synth_images(1, 240, 240, seed=6)
    d = {'ckpt_seed': CKPT_SEED}
    for kind in NETS:
        net = getattr(R, kind)(synth.vgg19_standin(seed=0))
        synth.fill_params_(net, CKPT_SEED)
        sd = net.state_dict()
        keys = list(sd.keys())
        d[kind + '_keys'] = np.array(keys)
        d[kind + '_shapes'] = np.array([','.join(str(v) for v in sd[k].shape) for k in keys])
        d[kind + '_dtypes'] = np.array([str(sd[k].dtype) for k in keys])
        d[kind + '_sums'] = np.array([float(sd[k].double().sum()) for k in keys])
        d[kind + '_l1'] = np.array([float(sd[k].double().abs().sum()) for k in keys])
        ptr = {}
        for i, k in enumerate(keys):
            ptr.setdefault(sd[k].data_ptr(), []).append(i)
        d[kind + '_alias'] = np.array([';'.join(str(i) for i in g) for g in ptr.values() if len(g) > 1])
        for k in keys:
            if sd[k].numel() <= 4096:
                d['%s_val_%s' % (kind, k)] = t2n(sd[k])
        net.eval()
        with torch.no_grad():
            for i, o in enumerate(net.forward(x)):
                d['%s_out_%d' % (kind, i)] = t2n(o)
    np.savez_compressed(os.path.join(OUT, 'checkpoints.npz'), **d)
    print('checkpoints.npz', {k: len(d[k + '_keys']) for k in NETS}, 'alias groups', {k: len(d[k + '_alias']) for k in NETS})


# ----------------------------------------------------------------------------- C. captured training steps
class _SynthSet(torch.utils.data.Dataset):
    """Replaces the JPEG-reading Dataset classes: same tuple layout as
    LPPatch_Online (img, bbox), LPPatchLM_Online (img, bbox, vertices) and
    DenseBoxDataset (img, bbox, vertices, label) (DenseBox.py:877, :994, :1070)."""

    def __init__(self, kind, n, seed):
        self.kind = kind
        self.x, self.bbox, self.vert, self.lab = synth.synth_batch(
            n, seed=seed, neg_frac=0.34 if kind == 'DenseBoxLMLOC' else 0.0)
        if kind == 'DenseBoxLMLOC':
            self.lab[0, 0] = 1.0 if float(self.bbox[0].abs().sum()) > 0 else 0.0

    def __len__(self):
        return self.x.size(0)

    def __getitem__(self, i):
        if self.kind == 'DenseBox':
            return self.x[i], self.bbox[i]
        if self.kind == 'DenseBoxLM':
            return self.x[i], self.bbox[i], self.vert[i]
        return self.x[i], self.bbox[i], self.vert[i], self.lab[i]


class _RecDropout(torch.nn.Module):
    """Mask-injecting Dropout: 'off' = identity, 'mask' = seeded Bernoulli(0.5)*2,
    recorded so the build can be fed the same masks."""
    mode = 'off'
    gen = None
    rec = []

    def __init__(self, p=0.5, inplace=False):
        super().__init__()

    def forward(self, x):
        if _RecDropout.mode == 'off' or not self.training:
            return x
        m = (torch.rand(x.shape, generator=_RecDropout.gen) >= 0.5).to(x.dtype)
        _RecDropout.rec.append(np.packbits(t2n(m).astype(np.uint8).reshape(-1)))
        return x * m * 2.0


def capture_train(kind, fn_name, n_patch, batch, seed, dropout, lr, fname, kw):
    cap = {'steps': []}
    ds = _SynthSet(kind, n_patch, seed)

    # ---- patches (restored below)
    saved = {}

    def patch(obj, name, val):
        saved[(id(obj), name)] = (obj, name, getattr(obj, name))
        setattr(obj, name, val)

    for cls in ('LPPatch_Online', 'LPPatchLM_Online', 'DenseBoxDataset'):
        patch(R, cls, lambda root=None, transform=None, size=None, _d=ds: _d)
    real_loader = torch.utils.data.DataLoader
    patch(torch.utils.data, 'DataLoader',
          lambda dataset, batch_size, shuffle, num_workers: real_loader(dataset, batch_size=batch_size,
                                                                      shuffle=False, num_workers=0))
    patch(torch, 'save', lambda *a, **k: None)
    patch(torch.nn, 'Dropout', _RecDropout)
    _RecDropout.mode = dropout
    _RecDropout.gen = torch.Generator().manual_seed(99)
    _RecDropout.rec = []

    nets = []
    real_cls = getattr(R, kind)

    # the reference calls super(DenseBox, self) through the module global, so the
    # class object itself must stay in place: wrap its methods instead of subclassing
    real_init, real_forward = real_cls.__init__, real_cls.forward

    def init_wrap(self, vgg19):
        real_init(self, vgg19)
        synth.fill_params_(self, PARAM_SEED)
        nets.append(self)

    def forward_wrap(self, X):
        outs = real_forward(self, X)
        for o in outs:
            o.retain_grad()
        cap['cur'] = {'outs': outs}
        return outs
    patch(real_cls, '__init__', init_wrap)
    patch(real_cls, 'forward', forward_wrap)

    # loc-type generators need the NumPy>=2 shim
    for nm in ('init_loc_map', 'init_loc', 'init_lm_locmap', 'init_lm_locmap_pn'):
        f = getattr(R, nm)

        def shim(*a, _f=f, **k):
            # first argument (positional or whichever of bboxes/vertices is named) -> double
            if a:
                a = (a[0].double(),) + a[1:]
            for key in ('bboxes', 'vertices'):
                if key in k:
                    k[key] = k[key].double()
            return _f(*a, **k)
        patch(R, nm, shim)

    def rec_sel(loss_mask, pos_indices, neg_indices, _f=R.mask_by_sel):
        _f(loss_mask, pos_indices, neg_indices)
        cap['cur'].setdefault('neg_idx', []).append(t2n(neg_indices).copy())
        cap['cur'].setdefault('mask_sel', []).append(pack(t2n(loss_mask)))
    patch(R, 'mask_by_sel', rec_sel)

    def mk_gray(name):
        f = getattr(R, name)

        def g(loss_mask, *a, **k):
            f(loss_mask, *a, **k)
            cap['cur'].setdefault('mask_' + name, []).append(pack(t2n(loss_mask)))
        return g
    for nm in ('mask_gray_zone_cls', 'mask_gray_zone_cls_pn', 'mask_gray_zone_lm'):
        patch(R, nm, mk_gray(nm))

    real_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        cap['cur']['loss'] = float(self.item())
        cap['cur']['loss32'] = t2n(self).copy()
        real_backward(self, *a, **k)
    patch(torch.Tensor, 'backward', rec_backward)

    real_step = torch.optim.SGD.step

    def rec_step(self, *a, **k):
        net = nets[0]
        cur = cap['cur']
        cur['grads'] = {n: (t2n(p.grad).copy() if p.grad is not None else None)
                        for n, p in net.named_parameters()}
        cur['p_before'] = {n: t2n(p).copy() for n, p in net.named_parameters() if n in SMALL}
        real_step(self, *a, **k)
        cur['p_after'] = {n: t2n(p).copy() for n, p in net.named_parameters() if n in SMALL}
        cur['out_grads'] = [t2n(o.grad).copy() for o in cur['outs']]
        cur['outs'] = [t2n(o).copy() for o in cur['outs']]
        cap['steps'].append(cur)
    patch(torch.optim.SGD, 'step', rec_step)

    np.random.seed(1234)
    try:
        getattr(R, fn_name)(src_root='.', dst_root='.', num_epoch=1, base_lr=lr,
                            batch_size=batch, resume=None, is_test=False, **kw)
    finally:
        for obj, name, val in saved.values():
            setattr(obj, name, val)

    # ---- serialise
    d = {'kind': kind, 'seed': seed, 'n_patch': n_patch, 'batch': batch, 'lr': lr,
         'param_seed': PARAM_SEED, 'dropout': dropout,
         'bbox': t2n(ds.bbox), 'vert': t2n(ds.vert), 'lab': t2n(ds.lab),
         'n_steps': len(cap['steps'])}
    for k_, v_ in kw.items():
        d['kw_' + k_] = v_
    if dropout == 'mask':
        for i, m in enumerate(_RecDropout.rec):
            d['dropmask_%d' % i] = m
        d['n_dropmask'] = len(_RecDropout.rec)
    for si, st in enumerate(cap['steps']):
        p = 's%d_' % si
        d[p + 'loss'] = st['loss']
        d[p + 'loss32'] = st['loss32']
        for i, o in enumerate(st['outs']):
            d[p + 'out_%d' % i] = o
            d[p + 'dout_%d' % i] = st['out_grads'][i]
        for i, a in enumerate(st['neg_idx']):
            d[p + 'neg_idx_%d' % i] = a
            d[p + 'mask_sel_%d' % i] = st['mask_sel'][i]
        for nm in ('mask_gray_zone_cls', 'mask_gray_zone_cls_pn', 'mask_gray_zone_lm'):
            for i, a in enumerate(st.get('mask_' + nm, [])):
                d[p + nm + '_%d' % i] = a
        for n, g in st['grads'].items():
            if g is None:
                d[p + 'gnone_' + n] = 1
                continue
            flat = g.reshape(-1).astype(np.float64)
            d[p + 'gstat_' + n] = np.array([flat.sum(), np.abs(flat).sum(), (flat * flat).sum()])
            if n in SMALL:
                d[p + 'g_' + n] = g
            else:
                d[p + 'gsub_' + n] = g.reshape(-1)[::997].copy()
        for n in st['p_before']:
            d[p + 'pb_' + n] = st['p_before'][n]
            d[p + 'pa_' + n] = st['p_after'][n]
    np.savez_compressed(os.path.join(OUT, fname), **d)
    print(fname, 'steps', len(cap['steps']), 'loss', [s['loss'] for s in cap['steps']],
          'half', cap['steps'][0]['neg_idx'][0].shape)


SMALL = {'conv1_1_1.weight', 'conv1_1_1.bias', 'conv1_2_1.bias', 'conv4_4_1.bias',
         'conv5_2_det.weight', 'conv5_2_det.bias', 'conv5_2_loc.weight', 'conv5_2_loc.bias',
         'conv5_1_det.bias', 'conv5_2_landmark.weight', 'conv5_2_lmloc.weight',
         'conv6_1_det.weight', 'conv6_1_det.bias', 'conv6_3_det.weight', 'conv6_3_det.bias',
         'conv6_2_det.bias'}


def gen_train():
    capture_train('DenseBox', 'train_online', 4, 2, seed=21, dropout='off', lr=1e-8,
                  fname='train_DenseBox.npz', kw={'lambda_loc': 3.0})
    capture_train('DenseBox', 'train_online', 2, 2, seed=22, dropout='mask', lr=1e-8,
                  fname='train_DenseBox_dropout.npz', kw={'lambda_loc': 3.0})
    capture_train('DenseBoxLM', 'train_LM_online', 2, 2, seed=23, dropout='off', lr=1e-8,
                  fname='train_DenseBoxLM.npz',
                  kw={'lambda_loc': 3.0, 'lambda_det': 1.0, 'lambda_lm': 0.5})
    capture_train('DenseBoxLMLOC', 'train_densebox_online', 3, 3, seed=24, dropout='off', lr=1e-8,
                  fname='train_DenseBoxLMLOC.npz',
                  kw={'lambda_loc': 3.0, 'lambda_det': 1.0, 'lambda_lm': 0.5})


# ----------------------------------------------------------------------------- D. decode + NMS
def gen_decode():
    rs = np.random.RandomState(5)
    d = {}

    def maps(h, w):
        s = torch.from_numpy(rs.randn(1, 1, h, w).astype(np.float32))
        l = torch.from_numpy((rs.randn(1, 4, h, w) * 6).astype(np.float32))
        hm = torch.from_numpy(rs.randn(1, 4, h, w).astype(np.float32))
        ll = torch.from_numpy((rs.randn(1, 8, h, w) * 6).astype(np.float32))
        return s, l, hm, ll
    s, l, hm, ll = maps(60, 60)
    d['a_s'], d['a_l'], d['a_hm'], d['a_ll'] = t2n(s), t2n(l), t2n(hm), t2n(ll)
    d['a_parse_output'] = R.parse_output(s, l, K=10)
    d['a_parse_out_MN'] = R.parse_out_MN(s, l, 240, 240, K=10)
    d['a_parse_DetLM'] = R.parse_DetLM(s, l, hm, 240, 240, K=10)
    d['a_parse_DetLMLOC'] = R.parse_DetLMLOC(s, l, hm, ll, 240, 240, K=10)
    d['a_parse_out_MN_K50'] = R.parse_out_MN(s, l, 240, 240, K=50)
    for k_ in ('a_parse_output', 'a_parse_DetLM', 'a_parse_DetLMLOC', 'a_parse_out_MN_K50'):
        d[k_ + '_keep'] = np.asarray(R.NMS(d[k_], 0.4), np.int64)
    # non-square, M,N not multiples of 4 -> M//4, N//4
    s, l, hm, ll = maps(101 // 4, 134 // 4)
    d['b_s'], d['b_l'], d['b_hm'], d['b_ll'] = t2n(s), t2n(l), t2n(hm), t2n(ll)
    d['b_parse_out_MN'] = R.parse_out_MN(s, l, 101, 134, K=10)
    d['b_parse_DetLMLOC'] = R.parse_DetLMLOC(s, l, hm, ll, 101, 134, K=7)
    d['b_parse_DetLM'] = R.parse_DetLM(s, l, hm, 101, 134, K=7)

    # NMS on crafted boxes: IoU just above / below 0.4, identical boxes, nested, disjoint
    def box(x, y, w, h, sc):
        return [x, y, x + w, y + h, sc]
    crafted = np.array([
        box(10, 10, 99, 99, 0.90),
        box(10, 10, 99, 99, 0.85),            # identical box
        box(10, 53, 99, 99, 0.80),            # IoU ~0.4 region
        box(10, 52, 99, 99, 0.79),
        box(10, 54, 99, 99, 0.78),
        box(300, 300, 50, 20, 0.70),          # disjoint
        box(30, 30, 20, 20, 0.65),            # nested small
        box(305, 302, 50, 20, 0.95),          # overlaps the disjoint one, higher score
        box(500.5, 20.25, 33.5, 80.75, 0.10),
        box(-20, -5, 60, 60, 0.50),           # negative coords
        box(100, 100, -30, -30, 0.45),        # degenerate (x2<x1): negative 'area'
    ], np.float64)
    d['nms_in'] = crafted
    for th in (0.4, 0.0, 0.7):
        d['nms_keep_%02d' % int(th * 10)] = np.asarray(R.NMS(crafted, th), np.int64)
    big = np.zeros((200, 5))
    big[:, 0] = rs.rand(200) * 400
    big[:, 1] = rs.rand(200) * 300
    big[:, 2] = big[:, 0] + 20 + rs.rand(200) * 120
    big[:, 3] = big[:, 1] + 10 + rs.rand(200) * 60
    big[:, 4] = rs.rand(200)
    d['nms_big_in'] = big
    d['nms_big_keep'] = np.asarray(R.NMS(big, 0.4), np.int64)
    np.savez_compressed(os.path.join(OUT, 'decode.npz'), **d)
    print('decode.npz', d['nms_keep_04'], len(d['nms_big_keep']))


# ----------------------------------------------------------------------------- E. dataset label parsing
def gen_datasets():
    """File-name labels as the reference's Dataset constructors parse them (DenseBox.py:784-860, :927-970, :1038-1052).
    Only __init__ runs (it lists the directory and parses names); the files are empty."""
    import tempfile
    names12 = [
        'img001_label_80_88_160_120_78_86_162_87_161_122_79_121.jpg',
        'neg_7_label_0_0_0_0_0_0_0_0_0_0_0_0.jpg',
        'x_label_33_57_131_95_31_55_133_58_130_97_35_96_extra.png',
        'plate_00012_label_0_40_200_100_8_40_200_42_198_100_9_99.jpg',
        'a_b_label_150_30_230_70_150_30_239_31_238_70_151_69.jpeg',
        'p_label_1_2_3_4_5_6_7_8_9_10_11_12.jpg',
    ]
    names4 = ['q_label_12_34_56_78.jpg', 'r2_label_0_16_239_240.jpg'] + names12[:2]
    d = {}
    with tempfile.TemporaryDirectory() as root:
        for n in names12:
            open(os.path.join(root, n), 'w').close()
        ds = R.DenseBoxDataset(root=root, transform=None, size=(240, 240))
        order = [os.path.split(p)[1] for p in ds.imgs_path]
        d['db_names'] = np.array(order)
        d['db_bbox'] = np.stack([t2n(t) for t in ds.bboxes])
        d['db_vert'] = np.stack([t2n(t) for t in ds.vertices])
        d['db_lab'] = np.stack([t2n(t) for t in ds.labels])
        lm = R.LPPatchLM_Online(root=root, transform=None, size=(240, 240))
        d['lm_names'] = np.array([os.path.split(p)[1] for p in lm.imgs_path])
        d['lm_bbox'] = np.stack([t2n(t) for t in lm.bboxes])
        d['lm_vert'] = np.stack([t2n(t) for t in lm.vertices])
    with tempfile.TemporaryDirectory() as root:
        for n in names4:
            open(os.path.join(root, n), 'w').close()
        lp = R.LPPatch_Online(root=root, transform=None, size=(240, 240))
        d['lp_names'] = np.array([os.path.split(p)[1] for p in lp.imgs_path])
        d['lp_bbox'] = np.stack([t2n(t) for t in lp.labels])
    np.savez_compressed(os.path.join(OUT, 'datasets.npz'), **d)
    print('datasets.npz', d['db_bbox'].shape, d['lm_vert'].shape, d['lp_bbox'].shape)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['labels', 'decode', 'nets', 'checkpoints', 'train', 'datasets']
    for w in which:
        {'labels': gen_labels, 'decode': gen_decode, 'nets': gen_nets, 'checkpoints': gen_checkpoints, 'train': gen_train,
         'datasets': gen_datasets}[w]()
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('total fixture bytes', sz)
