"""Drop-in ``nn.Module`` front end for the three reference networks.

Same class names, constructor signature, attribute names, aliased ``state_dict``
key set/order and ``forward()`` tuple order as the reference:

  DenseBox       DenseBox.py:31-228    -> (scores[N,1,h,w], locs[N,4,h,w])
  DenseBoxLM     DenseBox.py:232-473   -> (scores, locs, landmarks[N,4], refine_scores[N,1])
  DenseBoxLMLOC  DenseBox.py:477-738   -> (score, rf_score, bbox_loc[N,4], lm_heatmap[N,4], lm_loc[N,8])

The modules only hold the fp32 master parameters (ordinary ``nn.Parameter`` s,
so ``.parameters()``, ``.state_dict()``, ``load_state_dict(strict=True)`` of a
reference checkpoint and ``torch.optim.SGD`` all work).  All compute happens in
the HIP engine (``densebox_amd.engine``) through the C ABI of
``libdensebox_hip.so``: there is NO CPU/PyTorch fallback -- ``forward`` on a
non-GPU tensor, or without the built library, raises.
"""
import copy

import torch
import torch.nn as nn

# (attribute stem, index of the conv in vgg19.features) -- DenseBox.py:49-140.
# conv3_3 is constructed and lives in the state_dict but forward() never runs it
# (DenseBox.py:193-195), so it never receives a gradient.
_VGG_LAYOUT = [
    ('conv1_1', 0), ('conv1_2', 2), ('pool1', 4),
    ('conv2_1', 5), ('conv2_2', 7), ('pool2', 9),
    ('conv3_1', 10), ('conv3_2', 12), ('conv3_3', 14), ('conv3_4', 16), ('pool3', 18),
    ('conv4_1', 19), ('conv4_2', 21), ('conv4_3', 23), ('conv4_4', 25),
]

# (head stem, out channels, name of the nn.Sequential wrapper) per network, in
# registration order (DenseBox.py:149-178, :350-396, :595-658).
_HEADS = {
    'DenseBox': [('det', 1, 'output_score'), ('loc', 4, 'output_loc')],
    'DenseBoxLM': [('det', 1, 'output_score'), ('loc', 4, 'output_loc'),
                   ('landmark', 4, 'output_landmark')],
    'DenseBoxLMLOC': [('det', 1, 'output_score'), ('loc', 4, 'output_bbox_loc'),
                      ('lmloc', 8, 'output_lmloc'), ('landmark', 4, 'output_lm_heatmap')],
}

# forward() tuple order, as names of engine outputs
_OUT_ORDER = {
    'DenseBox': ('det', 'loc'),
    'DenseBoxLM': ('det', 'loc', 'landmark', 'refine'),
    'DenseBoxLMLOC': ('det', 'refine', 'loc', 'landmark', 'lmloc'),
}


class _DenseBoxBase(nn.Module):
    KIND = None

    def __init__(self, vgg19):
        super().__init__()
        feats = vgg19.features._modules
        for stem, idx in _VGG_LAYOUT:
            if stem.startswith('pool'):
                setattr(self, stem, copy.deepcopy(feats[str(idx)]))
                continue
            conv = copy.deepcopy(feats[str(idx)])
            act = copy.deepcopy(feats[str(idx + 1)])
            setattr(self, stem + '_1', conv)
            setattr(self, stem + '_2', act)
            setattr(self, stem, nn.Sequential(conv, act))        # second (aliased) registration
        if self.KIND != 'DenseBox':
            self.pool4 = nn.MaxPool2d(kernel_size=2, stride=2, padding=0, dilation=1, ceil_mode=False)
        for stem, k, wrapper in _HEADS[self.KIND]:
            c1 = nn.Conv2d(768, 512, kernel_size=(1, 1))
            c2 = nn.Conv2d(512, k, kernel_size=(1, 1))
            nn.init.xavier_normal_(c1.weight.data)               # weights only; biases keep Conv2d's default
            nn.init.xavier_normal_(c2.weight.data)
            setattr(self, 'conv5_1_' + stem, c1)
            setattr(self, 'conv5_2_' + stem, c2)
            setattr(self, wrapper, nn.Sequential(c1, nn.Dropout(), c2))
        if self.KIND != 'DenseBox':
            self.conv6_1_det = nn.Conv2d(5, 64, kernel_size=(3, 3))    # no padding (DenseBox.py:399-407)
            self.conv6_2_det = nn.Conv2d(64, 64, kernel_size=(5, 5))
            self.conv6_3_det = nn.Conv2d(64, 1, kernel_size=(1, 1))
            for m in (self.conv6_1_det, self.conv6_2_det, self.conv6_3_det):
                nn.init.xavier_normal_(m.weight.data)
        self._engine = None
        # arithmetic type of the HIP path: 'f16' | 'bf16' | 'f32', or None = 'bf16' for training steps (fp32's range: the
        # reference loss is an unscaled sum, a diverging step could overflow f16's 65504 in the gradient maps) and 'f16'
        # for inference (three more mantissa bits: maps within 2.8e-3 of the reference instead of 2.7e-2)
        self.compute_dtype = None
        self.dropout_masks = None         # optional injected {head: uint8 [N,512,h,w]} (parity tests)

    def resolved_dtype(self, train=None):
        if self.compute_dtype is not None:
            return self.compute_dtype
        train = self.training if train is None else train
        return 'bf16' if train else 'f16'

    # ------------------------------------------------------------------ engine plumbing
    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self)
        return self._engine

    def forward(self, X):
        """Reference forward (DenseBox.py:180-228 / :412-473 / :674-738) on the HIP engine.

        Input NCHW fp32 (or fp16/bf16) on a ROCm device; outputs NCHW-contiguous fp32,
        autograd-connected to the parameters so ``loss.backward()`` + ``optimizer.step()``
        work exactly as in the reference training loops (DenseBox.py:2186-2187).
        """
        if not X.is_cuda:
            raise RuntimeError('densebox_amd has no CPU path: forward() needs a tensor on an MI355X '
                               '(got device %s)' % X.device)
        outs = self.engine().forward(X)
        return tuple(outs[k] for k in _OUT_ORDER[self.KIND])

    # additive API (no reference counterpart; semantics = the inline loss section of the train loops)
    def loss(self, outputs, bbox, vertices=None, labels=None, rand_neg_indices=None,
             lm_rand_neg_indices=None, lambda_loc=3.0, lambda_det=1.0, lambda_lm=0.5, **kw):
        from .loss import densebox_loss
        return densebox_loss(self.KIND, outputs, bbox, vertices, labels, rand_neg_indices,
                             lm_rand_neg_indices, lambda_loc, lambda_det, lambda_lm, **kw)

    def detect(self, image, K=10, nms_thresh=0.4):
        """forward -> top-K decode -> NMS (test drivers, DenseBox.py:3788-3799 etc.)."""
        from .decode import detect
        return detect(self, image, K, nms_thresh)


class DenseBox(_DenseBoxBase):
    KIND = 'DenseBox'


class DenseBoxLM(_DenseBoxBase):
    KIND = 'DenseBoxLM'


class DenseBoxLMLOC(_DenseBoxBase):
    KIND = 'DenseBoxLMLOC'
