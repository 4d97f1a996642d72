import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from conftest import unpack
import densebox_amd.labels as LB
g = np.load('tests/golden/labels.npz'); T = torch.from_numpy
bbox, lab = T(g['bbox']), T(g['lab']); B = bbox.size(0)
for tag, labels in (('', None), ('_pn', lab)):
    gt = LB.init_score_map(bbox, labels=labels); m = gt.clone(); pos = torch.nonzero(gt)
    LB.mask_by_sel(m, pos, T(g['neg_idx']))
    if labels is None: LB.mask_gray_zone_cls(m, bbox)
    else: LB.mask_gray_zone_cls_pn(m, bbox, lab)
    ref = unpack(g['mask_gray' + tag], (B, 1, 60, 60)); d = np.argwhere(m.cpu().numpy() != ref)
    print('tag', repr(tag), 'ndiff', len(d), d[:12].tolist())
    for r in d[:6]: print('  hip', m[r[0], 0, r[2], r[3]].item(), 'ref', ref[r[0], 0, r[2], r[3]], 'bbox', bbox[r[0]].tolist())
