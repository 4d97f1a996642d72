"""What CUs taken by a communication kernel cost the training step: ONE side stream runs tools/comm_hog.hip -- G workgroups of one
wave, each spinning for `us` microseconds -- `bursts` times per step while the step runs: a stand-in for RCCL's channel workgroups during
the overlapped gradient all-reduce (they cannot share a CU with the 512-register MFMA kernels, and the persistent kernels launch one
workgroup per CU).  Build the stand-in first: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/comm_hog.hip -o tools/libcomm_hog.so
usage: python tools/gpu_comm_interference.py [dtype] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import densebox_amd as D
from densebox_amd import synth, labels as LB
from densebox_amd.dist import DataParallel
from densebox_amd.optim import SGD
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 64
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11); net = net.cuda().train(); net.compute_dtype = dtype
dp = DataParallel(net, SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8))
x, bbox, vert, lab = synth.synth_batch(n, seed=100, neg_frac=0.1); x = x.cuda()
rs = np.random.RandomState(1234)
import ctypes
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcomm_hog.so'))
side = torch.cuda.Stream()


def run(G, us, bursts):
    ts = []
    for i in range(steps + 10):
        p = dp.global_positive_num(bbox, lab); _, half = LB.neg_counts(p, n)
        rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)])
        lrn = rs.randint(0, 3600, size=(4, n, 1))
        if i == 10:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in range(bursts):
            assert hog.comm_hog(G, 64, us, ctypes.c_void_p(side.cuda_stream)) == 0
        dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for G, us, bursts in [(0, 0, 0), (8, 1000, 1), (16, 1000, 1), (32, 1000, 1), (64, 1000, 1), (16, 3000, 1), (32, 3000, 1), (16, 300, 4), (0, 0, 0)]:
    print('G=%2d spinning workgroups x %4d us x %d per step: %.3f ms/step' % (G, us, bursts, run(G, us, bursts)), flush=True)
