"""Wall time of every training step of a fresh process (the bench workload), to see how long the ramp to steady state is.
usage: python tools/gpu_step_times.py [dtype] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import densebox_amd as D
from densebox_amd import synth, labels as LB
from densebox_amd.dist import DataParallel
from densebox_amd.optim import SGD
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 64
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().train(); net.compute_dtype = dtype
dp = DataParallel(net, SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8))
x, bbox, vert, lab = synth.synth_batch(n, seed=100, neg_frac=0.1); x = x.cuda()
rs = np.random.RandomState(1234)
ts = []
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = dp.global_positive_num(bbox, lab); _, half = LB.neg_counts(p, n)
    rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)])
    lrn = rs.randint(0, 3600, size=(4, n, 1))
    dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
for a in range(0, steps, 20):
    print('steps %3d-%3d: mean %.3f ms  min %.3f  max %.3f' % (a, min(a + 20, steps) - 1, ts[a:a + 20].mean(), ts[a:a + 20].min(), ts[a:a + 20].max()))
