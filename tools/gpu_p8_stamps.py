"""Launch boundaries of the real training step WITHOUT a profiler: a library built with -DDBX_P8_STAMP (tools/build_variant.sh stamp conv_igemm.hip
-DDBX_P8_STAMP; run with DBX_LIB=.../libdensebox_hip_stamp.so) stamps s_memrealtime (100 MHz) when the first workgroup of every 8-phase-kernel
launch starts and when its last one ends.  The forward pass launches conv3_1, conv3_2, conv3_4 and conv4_1 .. conv4_4 on that kernel with nothing
in between (except pool3): the idle time between those launches is the boundary the un-instrumented step pays.
usage: DBX_LIB=... python tools/gpu_p8_stamps.py [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import densebox_amd as D
from densebox_amd import synth, _lib, labels as LB
from densebox_amd.dist import DataParallel
from densebox_amd.optim import SGD
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = _lib.lib()
dev = torch.device('cuda', 0)
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.to(dev).train(); net.compute_dtype = 'f16'
dp = DataParallel(net, SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8))
x, bbox, vert, lab = synth.synth_batch(64, seed=100, neg_frac=0.1); x = x.to(dev)
rs = np.random.RandomState(1234)
for _ in range(steps):
    p = dp.global_positive_num(bbox, lab); _, half = LB.neg_counts(p, 64)
    rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(64)]); lrn = rs.randint(0, 3600, size=(4, 64, 1))
    dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 4096))()
L.dbx_lab_p8_stamps.restype = C.c_int
n = L.dbx_lab_p8_stamps(buf, 4096)
st = np.array(buf[:2 * n], dtype=np.uint64).reshape(n, 2).astype(np.int64)
per = n // steps
print('%d stamped launches, %d per step; last step (us from its first stamped launch; gap = idle time since the previous STAMPED launch ended):' % (n, per))
s = st[(steps - 1) * per:]
t0 = s[0, 0]
small = []
for i in range(per):
    gap = (s[i, 0] - s[i - 1, 1]) * 0.01 if i else 0.0
    print('  launch %2d  start %9.1f  duration %8.1f  gap %8.2f%s' % (i, (s[i, 0] - t0) * 0.01, (s[i, 1] - s[i, 0]) * 0.01, gap, '   <- back to back' if 0 < gap < 15 else ''))
    if 0 < gap < 15: small.append(gap)
# all steps: the back-to-back boundaries
allg = []
for k in range(1, steps):
    s = st[k * per:(k + 1) * per]
    allg += [float((s[i, 0] - s[i - 1, 1]) * 0.01) for i in range(1, per) if 0 < (s[i, 0] - s[i - 1, 1]) * 0.01 < 15]
if allg:
    allg.sort()
    print('back-to-back boundaries between 8-phase launches over %d steps: n %d  median %.2f us  min %.2f  max %.2f' % (steps - 1, len(allg), allg[len(allg) // 2], allg[0], allg[-1]))
