import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import densebox_amd as D
from densebox_amd import synth
from densebox_amd.optim import SGD
mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = 'f16'
xa = synth.synth_images(1, 240, 240, seed=1).cuda()
d1, k1 = net.detect(xa, K=10); d1b, _ = net.detect(xa, K=10)
print('first ok', d1[0, :3]); sys.stdout.flush()
if mode == 'train':
    net.train(); opt = SGD(net.parameters(), lr=2e-9, momentum=0.9, weight_decay=5e-8)
    x, bbox, vert, lab = synth.synth_batch(2, seed=7, neg_frac=0.0)
    outs = net(x.cuda()); loss = net.loss(outs, bbox, vert, lab); loss.backward(); opt.step(); net.eval()
elif mode == 'bump':
    with torch.no_grad():
        for p in net.parameters(): p.mul_(1.01)
elif mode == 'trainfwd':
    net.train()
    x, bbox, vert, lab = synth.synth_batch(2, seed=7, neg_frac=0.0)
    with torch.no_grad(): outs = net(x.cuda())
    net.eval()
    with torch.no_grad():
        for p in net.parameters(): p.mul_(1.01)
torch.cuda.synchronize(); print('modified', mode); sys.stdout.flush()
d2, k2 = net.detect(xa, K=10)
print('second ok', d2[0, :3]); sys.stdout.flush()
d3, k3 = net.detect(xa, K=10)
print('third ok', np.array_equal(d2, d3))
