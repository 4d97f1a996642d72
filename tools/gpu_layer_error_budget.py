"""Where the 16-bit forward's distance to the fp32 reference comes from, rounding point by rounding point.

north_star asks for score / bbox / landmark maps within 1e-3 of the reference; the f16 path reaches that on the bbox maps and in RMS on
every map, not in max norm on the score / landmark maps (tests/test_hip_forward.py MAP_TOL).  This tool answers "which activation would have
to stay in fp32 to get there": it runs the EXACT-fp32 engine path on the reference fixture (tests/golden/net_*.npz: the reference's own
outputs for 2 patches of 240 x 240) and rounds, in place and behind the producing launch (Engine.act_hook), exactly the buffers the 16-bit
path stores in 16 bits -- one at a time, all of them, all but the k largest contributors -- and the weights the 16-bit path packs in 16
bits.  A 16-bit MFMA multiplies its rounded operands exactly and accumulates in fp32, so "fp32 path + rounded operands" IS the 16-bit path up
to fp32 summation order; the first rows of the table check that against the real 16-bit kernels.

usage: python tools/gpu_layer_error_budget.py [out.json] [--dtype f16|bf16]
error = max |out - ref| / max(1, max|ref|) per output map (the tests' measure) and the same with RMS instead of max.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import densebox_amd as D
from densebox_amd import synth, _lib

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
OUTN = {'DenseBox': ['score', 'bbox'], 'DenseBoxLMLOC': ['score', 'refine', 'bbox', 'lm_heat', 'lm_loc']}
ACTS = ['x0', 'a11', 'p1', 'a21', 'p2', 'a31', 'a32', 'c34', 'a41', 'a42', 'a43', 'a44']       # p3 = max over c34: no rounding of its own
LAYERS = ['conv1_1_1', 'conv1_2_1', 'conv2_1_1', 'conv2_2_1', 'conv3_1_1', 'conv3_2_1', 'conv3_4_1', 'conv4_1_1', 'conv4_2_1', 'conv4_3_1',
          'conv4_4_1']
dtype = 'f16'
args = [a for a in sys.argv[1:]]
if '--dtype' in args:
    dtype = args[args.index('--dtype') + 1]
    del args[args.index('--dtype'):args.index('--dtype') + 2]
TDT = {'f16': torch.float16, 'bf16': torch.bfloat16}[dtype]


def rnd_(t):
    t.copy_(t.to(TDT).float())


def buf_view(P, name):
    if name in ('c34', 'ups'):
        b = P.B['fusion']
        t = P.ws[b.off:b.off + b.bytes].view(torch.float32).view(b.n, b.hp, b.wp, b.c)
        return t[..., 512:] if name == 'c34' else t[..., :512]
    b = P.B[name]
    return P.ws[b.off:b.off + b.bytes].view(torch.float32)


def errors(outs, g):
    res = []
    for i, o in enumerate(outs):
        ref = g['out240_%d' % i]
        a = o.float().cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        res.append((float(np.abs(a - ref).max()) / scale, float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2))) / scale))
    return res


doc = {'what': __doc__.split('\n')[0], 'dtype': dtype, 'kinds': {}}
for kind in ('DenseBox', 'DenseBoxLMLOC'):
    g = np.load(os.path.join(G, 'net_%s.npz' % kind))
    x = synth.synth_images(2, 240, 240, seed=3).cuda()

    def make(dt):
        net = getattr(D, kind)(synth.vgg19_standin(seed=0))
        synth.fill_params_(net, int(g['param_seed']))
        net = net.cuda().eval()
        net.compute_dtype = dt
        return net

    def run(acts=(), wlayers=(), wheads=False, real=None):
        net = make(real or 'f32')
        eng = net.engine()
        if real is None:
            with torch.no_grad():
                for st in wlayers:
                    rnd_(getattr(net, st).weight.data)
            active = set(acts)
            eng.act_hook = lambda name, P: rnd_(buf_view(P, name)) if name in active else None
            if wheads:
                with torch.no_grad():
                    net(x)                                      # fills the folded-heads cache; its packed fp32 images are then rounded in place
                ent = eng.wcache[('folded', _lib.F32)][1]
                for t in (ent[0], ent[3], ent[4]):
                    rnd_(t.view(torch.float32))
        with torch.no_grad():
            outs = net(x)
        torch.cuda.synchronize()
        return errors(outs, g)

    rows = {}
    rows['real %s kernels' % dtype] = run(real=dtype)
    rows['fp32 kernels, nothing rounded'] = run()
    rows['simulated: all weights + all activations'] = run(ACTS, LAYERS, True)
    rows['all weights (backbone + folded heads)'] = run((), LAYERS, True)
    rows['all activations'] = run(ACTS)
    for a in ACTS:
        rows['act ' + a] = run([a])
    for st in LAYERS:
        rows['wgt ' + st] = run((), [st])
    rows['wgt folded heads'] = run((), (), True)
    # which roundings would have to go for the worst map to reach 1e-3: drop the largest single contributors one after the other
    names = OUTN[kind]
    singles = {k: v for k, v in rows.items() if k.startswith(('act ', 'wgt '))}
    worst_map = max(range(len(names)), key=lambda i: rows['simulated: all weights + all activations'][i][0])
    order = sorted(singles, key=lambda k: -singles[k][worst_map][1])
    acts_on, wl_on, wh_on = list(ACTS), list(LAYERS), True
    removal = []
    for k in order[:12]:
        if k.startswith('act '):
            acts_on.remove(k[4:])
        elif k == 'wgt folded heads':
            wh_on = False
        else:
            wl_on.remove(k[4:])
        e = run(acts_on, wl_on, wh_on)
        removal.append({'kept_in_fp32': k, 'errors': e})
    doc['kinds'][kind] = {'maps': names, 'rows': rows, 'worst_map': names[worst_map], 'removal_order': removal}
    print('=== %s (%s) -- max-rel / rms-rel per map: %s' % (kind, dtype, ', '.join(names)))
    for k, v in rows.items():
        print('%-44s %s' % (k, '  '.join('%.2e/%.2e' % e for e in v)))
    q = [float(np.sqrt(sum(singles[k][i][1] ** 2 for k in singles))) for i in range(len(names))]
    print('%-44s %s' % ('quadrature sum of the single rows (rms)', '  '.join('         %.2e' % v for v in q)))
    print('--- keeping the largest contributors to "%s" in fp32, one more per row (cumulative):' % names[worst_map])
    for r in removal:
        print('%-44s %s' % ('+ ' + r['kept_in_fp32'], '  '.join('%.2e/%.2e' % e for e in r['errors'])))
if args:
    json.dump(doc, open(args[0], 'w'), indent=1)
