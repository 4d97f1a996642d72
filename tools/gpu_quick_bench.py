"""Quick forward timing (not the contract bench): python tools/gpu_quick_bench.py [kind] [dtype] [N]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import densebox_amd as D
from densebox_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else 'DenseBox'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f16'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
GF = {'DenseBox': 41.98, 'DenseBoxLM': 44.95, 'DenseBoxLMLOC': 47.81}[kind]
net = getattr(D, kind)(synth.vgg19_standin(0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(N, 240, 240, seed=1).cuda()
with torch.no_grad():
    for _ in range(3): net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 10
    for _ in range(it): net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
print('%s %s N=%d fwd %.3f ms  %.1f patches/s  %.1f TFLOP/s' % (kind, dtype, N, dt * 1e3, N / dt, N * GF / dt / 1e3))
