"""Deterministic synthetic weights / patches / labels for tests, fixtures and bench.

Everything is drawn from ``numpy.random.RandomState`` (the frozen legacy
generator), so the same seed gives bit-identical data in the fixture-generating
container and on the GPU box, independent of the torch version.

The reference trains on 240x240 JPEG patches whose 12-integer label lives in the
file name (DenseBox.py:784-860: bbox corners + 4 vertices in 240-space, divided
by 4 into the 60x60 output space; all-zero label = negative patch) and starts
from torchvision's pretrained VGG19 (DenseBox.py:1984).  Neither a dataset nor
torchvision nor the network exist here, so this module synthesises both:
``vgg19_standin`` builds an object with the ``.features._modules['0'..'36']``
layout the reference constructors index (DenseBox.py:44-137), and
``synth_batch`` draws patches + labels with the geometry SURVEY.md 8(d) fixes.
"""
import numpy as np
import torch
import torch.nn as nn

# torchvision VGG19 "E" feature config (conv3x3 pad1 + ReLU(inplace), 'M' = MaxPool2d(2,2))
_VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
              512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # DenseBox.py:770
IMAGENET_STD = (0.229, 0.224, 0.225)    # DenseBox.py:771


class _VGGStandIn(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features


def vgg19_standin(seed=0):
    """VGG19-layout ``features`` stack with seeded He-normal weights.

    Index layout matches torchvision (conv at 0,2,5,7,10,12,14,16,19,21,23,25,...;
    ReLU(inplace) after each conv; MaxPool2d(2,2) at 4,9,18,27,36), which is what
    DenseBox.__init__ deep-copies (DenseBox.py:49-137).
    """
    layers = []
    cin = 3
    for v in _VGG19_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers.append(nn.Conv2d(cin, v, kernel_size=3, padding=1))
            layers.append(nn.ReLU(inplace=True))
            cin = v
    net = _VGGStandIn(nn.Sequential(*layers))
    fill_params_(net, seed)
    return net


def unique_named_params(module):
    """(name, param) for each distinct Parameter, first registered name wins.

    The reference registers every conv twice (attribute + Sequential member,
    DenseBox.py:49-54), so ``state_dict`` has aliased keys; ``named_parameters``
    already de-duplicates in registration order, which is what SGD sees
    (DenseBox.py:2001).
    """
    return list(module.named_parameters())


def fill_params_(module, seed, bias_std=0.05):
    """Overwrite every distinct parameter in registration order, deterministically.

    Weights ~ N(0, 2/fan_in) for convs followed by ReLU (backbone), N(0, 1/fan_in)
    for the linear heads / refine branch (no ReLU after them, DenseBox.py:158-178)
    so that the output maps stay O(1) for unit-variance inputs.  Biases ~ N(0, bias_std).
    """
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in unique_named_params(module):
            shape = tuple(p.shape)
            if p.dim() == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                linear = ('conv5_' in name) or ('conv6_' in name)
                std = np.sqrt((1.0 if linear else 2.0) / fan_in)
                w = rs.standard_normal(size=shape) * std
            else:
                w = rs.standard_normal(size=shape) * bias_std
            p.copy_(torch.from_numpy(w.astype(np.float32)))
    return module


def synth_images(n, h=240, w=240, seed=0):
    """uint8 U[0,255] -> /255 -> ImageNet normalise (DenseBox.py:766-772). NCHW fp32."""
    rs = np.random.RandomState(seed + 1000003)
    u8 = rs.randint(0, 256, size=(n, 3, h, w)).astype(np.float32)
    x = u8 / np.float32(255.0)
    mean = np.asarray(IMAGENET_MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(IMAGENET_STD, np.float32).reshape(1, 3, 1, 1)
    return torch.from_numpy(((x - mean) / std).astype(np.float32))


def synth_labels(n, seed=0, neg_frac=0.1, size=240):
    """Integer labels in 240-space, then /4 into 60-space like DenseBox.py:825-850.

    Per positive patch: w in U{40..120}, h in U{16..48}; box and its 4 landmarks
    (box corners jittered by U{-4..4}, order LU,RU,RD,LD) stay >= 8 px inside the
    patch on all sides.  ``neg_frac`` of the patches are negatives: all-zero
    bbox/vertices, label 0 (DenseBox.py:811-818).
    Returns fp32 tensors bbox[n,4], vertices[n,8], labels[n,1].
    """
    rs = np.random.RandomState(seed + 2000003)
    bbox = np.zeros((n, 4), np.float64)
    vert = np.zeros((n, 8), np.float64)
    lab = np.ones((n, 1), np.float32)
    for i in range(n):
        if rs.random_sample() < neg_frac:
            lab[i, 0] = 0.0
            continue
        bw = int(rs.randint(40, 121))
        bh = int(rs.randint(16, 49))
        margin = 12  # 8 px border + 4 px landmark jitter
        x0 = int(rs.randint(margin, size - margin - bw + 1))
        y0 = int(rs.randint(margin, size - margin - bh + 1))
        x1, y1 = x0 + bw, y0 + bh
        bbox[i] = (x0, y0, x1, y1)
        corners = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
        for j, (cx, cy) in enumerate(corners):
            vert[i, 2 * j] = cx + int(rs.randint(-4, 5))
            vert[i, 2 * j + 1] = cy + int(rs.randint(-4, 5))
    bbox /= 4.0
    vert /= 4.0
    return (torch.from_numpy(bbox.astype(np.float32)),
            torch.from_numpy(vert.astype(np.float32)),
            torch.from_numpy(lab))


def synth_rand_neg_indices(n, k, seed=0, hw=3600):
    """Host-side random negatives, one ``choice(hw, k, replace=False)`` per sample
    from a seeded RandomState -- the injectable stand-in for DenseBox.py:2089-2094."""
    rs = np.random.RandomState(seed + 3000003)
    out = np.zeros((n, k), np.int64)
    for i in range(n):
        out[i] = rs.choice(hw, k, replace=False)
    return torch.from_numpy(out)


def synth_batch(n, seed=0, neg_frac=0.1):
    x = synth_images(n, seed=seed)
    bbox, vert, lab = synth_labels(n, seed=seed, neg_frac=neg_frac)
    return x, bbox, vert, lab
