import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import densebox_amd as D
from densebox_amd import synth
kind, dtype, n, h, w = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
net = getattr(D, kind)(synth.vgg19_standin(0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(n, h, w, seed=1).cuda()
with torch.no_grad():
    for _ in range(3): net(x)
    eng = net.engine(); eng.profile = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net(x); e1.record(); torch.cuda.synchronize()
print('total ms', e0.elapsed_time(e1))
for c in eng.profile:
    us = c['start'].elapsed_time(c['end']) * 1e3
    print('%-50s %9.1f us %8.1f TF' % (c['kernel'], us, c['flops'] / us / 1e6))
