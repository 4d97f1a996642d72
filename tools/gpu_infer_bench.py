"""Inference throughput: whole-image forward + on-GPU top-K decode + NMS (BASELINE.json: fps at 512x512; config 5: 1920x1080).
usage: python tools/gpu_infer_bench.py [kind] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import densebox_amd as D
from densebox_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else 'DenseBox'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f16'
net = getattr(D, kind)(synth.vgg19_standin(0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dtype
GF240 = {'DenseBox': 41.98, 'DenseBoxLM': 44.95, 'DenseBoxLMLOC': 47.81}[kind]
for (h, w, n, K) in [(512, 512, 1, 10), (512, 512, 16, 10), (1080, 1920, 1, 10), (1080, 1920, 1, 1000)]:
    x = synth.synth_images(n, h, w, seed=1).cuda()
    def run():
        if n == 1:
            return net.detect(x, K=K, nms_thresh=0.4)
        with torch.no_grad():
            return net(x)
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 20
    for _ in range(it): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    gf = GF240 * (h * w) / (240 * 240) * n
    what = 'forward+topK(%d)+NMS (results on host)' % K if n == 1 else 'forward only, batch %d' % n
    print('%s %s %dx%d  %s: %.3f ms  %.1f img/s  %.1f TFLOP/s' % (kind, dtype, w, h, what, dt * 1e3, n / dt, gf / dt / 1e3))
