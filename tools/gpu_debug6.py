import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from conftest import unpack
import densebox_amd.labels as LB
from densebox_amd.loss import densebox_loss
from densebox_amd import _lib
T = torch.from_numpy
g = np.load('tests/golden/labels.npz'); bbox, lab = T(g['bbox']), T(g['lab'])
for labels in (None, lab):
    host = LB.positive_count(bbox, labels)
    dev = torch.empty(bbox.size(0), dtype=torch.int32, device='cuda')
    bb = bbox.cuda(); lb = labels.cuda() if labels is not None else None
    _lib.check(_lib.lib().dbx_count_positives(_lib.ptr(bb), _lib.ptr(lb), bbox.size(0), _lib.ptr(dev), _lib.stream_ptr()))
    maps = LB.init_score_map(bbox, labels=labels).sum(dim=(1, 2, 3)).cpu().numpy()
    print('host', host.tolist()); print('dev ', dev.cpu().tolist()); print('maps', maps.astype(int).tolist())
for name in ['train_DenseBoxLM', 'train_DenseBoxLMLOC']:
    g = np.load('tests/golden/%s.npz' % name); kind = str(g['kind']); n = int(g['batch'])
    outs = []; i = 0
    while 's0_out_%d' % i in g.files: outs.append(T(g['s0_out_%d' % i]).cuda().requires_grad_(True)); i += 1
    neg0 = g['s0_neg_idx_0']; half = neg0.shape[1] // 2
    lm_rand = np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
    kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
    loss, dbg = densebox_loss(kind, tuple(outs), g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg_indices=neg0[:, half:], lm_rand_neg_indices=lm_rand, return_debug=True, **kw)
    loss.backward()
    print(name, 'loss', float(loss), float(g['s0_loss']))
    for i, o in enumerate(outs):
        ref = g['s0_dout_%d' % i]; d = np.abs(o.grad.cpu().numpy() - ref)
        w = np.unravel_index(d.argmax(), d.shape)
        print('  out', i, ref.shape, 'max|ref| %.3e maxdiff %.3e at %s hip %.5f ref %.5f nbad %d' % (np.abs(ref).max(), d.max(), w, o.grad.cpu().numpy()[w], ref[w], (d > 1e-4 * np.abs(ref).max()).sum()))
