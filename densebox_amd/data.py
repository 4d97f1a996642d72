"""The input side of the hot path: the reference datasets' file-name labels and image transform.

The reference stores the label of a 240x240 patch in its file name, ``..._label_a_b_..`` (DenseBox.py:784-860, :927-970,
:1038-1052) and divides by 4 into the 60x60 output space; images go through torchvision's ToTensor + ImageNet
Normalize (DenseBox.py:766-772).  Here: the three label parsers (host string work, same results as the dataset
constructors) and the device-side transform, which is fused into the layout kernel -- pass uint8 ``[N,H,W,3]`` tensors to
``net.forward`` (``dbx_u8hwc_to_framed``).  JPEG decoding / directory walking stay out of scope.
"""
import re

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # DenseBox.py:770
IMAGENET_STD = (0.229, 0.224, 0.225)    # DenseBox.py:771

_PAT12 = re.compile('.*_label_' + '_'.join(['([0-9]+)'] * 12) + '.*')     # DenseBox.py:787-789, :928-930
_PAT4 = re.compile('.*_label_([0-9]+)_([0-9]+)_([0-9]+)_([0-9]+)')         # DenseBox.py:1038


def parse_densebox_label(img_name):
    """DenseBoxDataset (DenseBox.py:784-860): 12 integers -> (bbox[4], vertices[8], label[1]) float32 in 60-space;
    an all-zero label marks a negative patch (label 0, zero bbox / vertices)."""
    m = _PAT12.match(img_name)
    if m is None:
        raise ValueError('no 12-integer _label_ field in %r' % img_name)
    d = [float(m.group(i)) for i in range(1, 13)]
    if all(v == 0.0 for v in d):
        return np.zeros(4, np.float32), np.zeros(8, np.float32), np.zeros(1, np.float32)
    q = np.array([v / 4.0 for v in d])          # python float division, then FloatTensor(np.array(...)) -> float32
    return q[:4].astype(np.float32), q[4:].astype(np.float32), np.ones(1, np.float32)


def parse_lm_label(img_name):
    """LPPatchLM_Online (DenseBox.py:927-970): 12 integers -> (bbox[4], vertices[8]); no negative-patch handling."""
    m = _PAT12.match(img_name)
    if m is None:
        raise ValueError('no 12-integer _label_ field in %r' % img_name)
    q = np.array([float(m.group(i)) / 4.0 for i in range(1, 13)])
    return q[:4].astype(np.float32), q[4:].astype(np.float32)


def parse_bbox_label(img_name):
    """LPPatch_Online (DenseBox.py:1038-1052): 4 integers -> bbox[4]."""
    m = _PAT4.match(img_name)
    if m is None:
        raise ValueError('no 4-integer _label_ field in %r' % img_name)
    return np.array([float(m.group(i)) / 4.0 for i in range(1, 5)]).astype(np.float32)


def collate_labels(names):
    """File names -> (bbox[N,4], vertices[N,8], labels[N,1]) float32 tensors, DenseBoxDataset semantics + default collate."""
    b, v, l = zip(*[parse_densebox_label(n) for n in names])
    return torch.from_numpy(np.stack(b)), torch.from_numpy(np.stack(v)), torch.from_numpy(np.stack(l))
