"""Achieved low-precision forward error per output map against the reference-captured fixtures (tests/golden/net_*.npz):
max |hip - ref| / max(1, max|ref|) for f32 / f16 / bf16 at 240x240 (all three nets), 100x132 and 1080p (DenseBox).
usage: python tools/gpu_lowprec_err.py [out.json]   -- north_star asks for 1e-3 of the reference; this is what the path achieves."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import densebox_amd as D
from densebox_amd import synth

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
OUTN = {'DenseBox': ['score', 'loc'], 'DenseBoxLM': ['score', 'loc', 'landmark', 'refine'],
        'DenseBoxLMLOC': ['cls', 'score(refine)', 'bbox', 'lm_heat', 'lm_loc']}
res = {}
for kind in ('DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC'):
    g = np.load(os.path.join(G, 'net_%s.npz' % kind))
    for dt in ('f32', 'f16', 'bf16'):
        net = getattr(D, kind)(synth.vgg19_standin(seed=0)); synth.fill_params_(net, int(g['param_seed']))
        net = net.cuda().eval(); net.compute_dtype = dt
        with torch.no_grad():
            outs = net(synth.synth_images(2, 240, 240, seed=3).cuda())
            odd = net(synth.synth_images(1, 100, 132, seed=4).cuda())
        for tag, os_, key in (('240x240', outs, 'out240_%d'), ('100x132', odd, 'outodd_%d')):
            for i, o in enumerate(os_):
                ref = g[key % i]
                a = o.float().cpu().numpy()
                scale = max(1.0, float(np.abs(ref).max()))
                res['%s/%s/%s/out%d' % (kind, dt, tag, i)] = {'max_abs_err': float(np.abs(a - ref).max()), 'scale': scale,
                                                             'rel': float(np.abs(a - ref).max()) / scale,
                                                             'rms_rel': float(np.sqrt(np.mean((a - ref) ** 2))) / scale}
g = np.load(os.path.join(G, 'net_DenseBox_1080p.npz'))
for dt in ('f32', 'f16', 'bf16'):
    net = D.DenseBox(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dt
    with torch.no_grad():
        s, l = net(synth.synth_images(1, 1080, 1920, seed=5).cuda())
    for name, a, ref in (('score', s[0, 0, ::9, ::8], g['score_sub']), ('loc', l[0, :, ::9, ::8], g['loc_sub'])):
        a = a.float().cpu().numpy(); scale = max(1.0, float(np.abs(ref).max()))
        res['DenseBox/%s/1920x1080/%s' % (dt, name)] = {'max_abs_err': float(np.abs(a - ref).max()), 'scale': scale,
                                                       'rel': float(np.abs(a - ref).max()) / scale,
                                                       'rms_rel': float(np.sqrt(np.mean((a - ref) ** 2))) / scale}
worst = {}
for k, v in res.items():
    dt = k.split('/')[1]
    worst[dt] = max(worst.get(dt, 0.0), v['rel'])
doc = {'what': 'max |hip - reference| / max(1, max|reference|) per output map, reference = tests/golden fixtures (fp32 PyTorch CPU)',
       'worst_rel_by_dtype': worst, 'maps': res}
print(json.dumps(worst))
for k in sorted(res):
    print('%-48s rel %.3e  rms %.3e  (scale %.2f)' % (k, res[k]['rel'], res[k]['rms_rel'], res[k]['scale']))
if len(sys.argv) > 1:
    json.dump(doc, open(sys.argv[1], 'w'), indent=1)
