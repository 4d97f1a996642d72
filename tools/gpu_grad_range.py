"""Dynamic range of the 16-bit training step's activation-gradient maps (the d_* frames hold dL/d(pre-activation) in the compute dtype).

The reference loss is an un-normalised SUM (DenseBox.py:2917), so nothing scales the gradients: f16 risks overflow (65504) on large
residuals and loses precision below 2^-14 = 6.1e-5 (subnormals: absolute spacing 6e-8).  Per gradient map of one step: max |v|, the
fraction of non-zero elements that are subnormal in f16, the share of the map's energy they carry -- on the reference fixture (3 patches)
and on bench.py's workload (batch 64 synthetic).  usage: python tools/gpu_grad_range.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import densebox_amd as D
from densebox_amd import synth, labels as LB

doc = {}
for tag, n, seed in (('fixture-like 3 patches', 3, 24), ('bench workload, batch 64', 64, 100)):
    for dt in ('f16', 'bf16'):
        net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0))
        synth.fill_params_(net, 11)
        net = net.cuda().train()
        net.compute_dtype = dt
        x, bbox, vert, lab = synth.synth_batch(n, seed=seed, neg_frac=0.1 if n > 8 else 0.0)
        _, half = LB.neg_counts(int(LB.positive_count(bbox, lab).sum()), n)
        rn = synth.synth_rand_neg_indices(n, half, seed=1)
        lrn = synth.synth_rand_neg_indices(4 * n, 1, seed=2).reshape(4, n, 1)
        outs = net(x.cuda())
        loss = net.loss(outs, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn)
        loss.backward()
        eng = net.engine()
        P = eng.last_plan
        rows = {}
        for name in sorted(k for k in P.B if k.startswith('d_')):
            v = eng.read_activation(name).abs()
            nz = v > 0
            sub = nz & (v < 2.0 ** -14)
            e = float((v.double() ** 2).sum())
            rows[name] = {'absmax': float(v.max()), 'nonzero_frac': float(nz.float().mean()),
                          'sub_f16_frac_of_nonzero': float(sub.sum()) / max(1, int(nz.sum())),
                          'sub_f16_energy_share': float((v[sub].double() ** 2).sum()) / max(e, 1e-300),
                          'median_nonzero': float(v[nz].median()) if bool(nz.any()) else 0.0}
        doc['%s / %s' % (tag, dt)] = {'loss': float(loss.detach()), 'maps': rows}
        print('== %s / %s: loss %.1f' % (tag, dt, float(loss.detach())))
        for k, r in rows.items():
            print('%-8s absmax %.3e  median|nz| %.3e  nonzero %.3f  f16-subnormal: %.4f of nonzero, %.2e of energy'
                  % (k, r['absmax'], r['median_nonzero'], r['nonzero_frac'], r['sub_f16_frac_of_nonzero'], r['sub_f16_energy_share']))
        del net, eng, P
        torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(doc, open(sys.argv[1], 'w'), indent=1)
