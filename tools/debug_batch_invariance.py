import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import densebox_amd as D
from densebox_amd import synth
kind, dtype = sys.argv[1], sys.argv[2]
net = getattr(D, kind)(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(64, 240, 240, seed=5).cuda()
with torch.no_grad():
    full = [o.clone() for o in net(x)]
    perm = torch.roll(torch.arange(64), 19).cuda()
    rolled = net(x[perm])
    for i, (a, b) in enumerate(zip(rolled, full)):
        d = (a - b[perm]).abs()
        bad = (d.flatten(1).max(dim=1).values > 0).nonzero().flatten().tolist()
        print(kind, dtype, 'out', i, tuple(a.shape), 'max diff %.3e' % float(d.max()), 'images differing:', bad[:10], len(bad))
