"""Host time to ENQUEUE one training step (no device synchronisation inside) vs the synchronised step time.
usage: python tools/gpu_host_time.py [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import densebox_amd as D
from densebox_amd import synth, labels as LB
from densebox_amd.dist import DataParallel
from densebox_amd.optim import SGD
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
n = 64
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().train(); net.compute_dtype = dtype
dp = DataParallel(net, SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8))
x, bbox, vert, lab = synth.synth_batch(n, seed=100, neg_frac=0.1); x = x.cuda()
rs = np.random.RandomState(1234)
def step():
    p = dp.global_positive_num(bbox, lab); _, half = LB.neg_counts(p, n)
    rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)])
    lrn = rs.randint(0, 3600, size=(4, n, 1))
    dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p)
for _ in range(10): step()
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(30):
    t0 = time.perf_counter(); step(); host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize(); t_all = (time.perf_counter() - t_all) * 1e3 / 30
sync = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); sync.append((time.perf_counter() - t0) * 1e3)
print('host enqueue per step: first 5 %s ms, median %.2f ms;  un-synchronised loop %.2f ms/step;  synchronised every step %.2f ms (median)' %
      (['%.2f' % v for v in host[:5]], float(np.median(host)), t_all, float(np.median(sync))))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
