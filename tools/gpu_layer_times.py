"""Per-launch times of one training step (HIP events around every conv / wgrad call of the engine), in call order.
usage: python tools/gpu_layer_times.py [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import densebox_amd as D
from densebox_amd import synth, labels as LB
from densebox_amd.dist import DataParallel
from densebox_amd.optim import SGD
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
kind, n = 'DenseBoxLMLOC', 64
net = getattr(D, kind)(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().train(); net.compute_dtype = dtype
dp = DataParallel(net, SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8))
x, bbox, vert, lab = synth.synth_batch(n, seed=100, neg_frac=0.1); x = x.cuda()
rs = np.random.RandomState(1234)
p = dp.global_positive_num(bbox, lab); _, half = LB.neg_counts(p, n)
def step():
    rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)])
    lrn = rs.randint(0, 3600, size=(4, n, 1))
    return dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p)
for _ in range(4): step()
eng = net.engine(); eng.profile = []
R = 4
for _ in range(R): step()
torch.cuda.synchronize()
calls = eng.profile; eng.profile = None
per = len(calls) // R
tot = 0
for i in range(per):
    us = np.mean([calls[i + r * per]['start'].elapsed_time(calls[i + r * per]['end']) * 1e3 for r in range(R)])
    c = calls[i]; tot += us
    print('%2d %-44s %8.1f GFLOP %8.1f us %7.0f TFLOP/s' % (i, c['kernel'], c['flops'] / 1e9, us, c['flops'] / us / 1e6))
print('sum of profiled launches: %.1f us' % tot)
