"""One-image inference (detect(): hipGraph replay) repeated a few times, for a rocprofv3 kernel trace.
usage: python tools/gpu_infer_trace.py [H] [W] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import densebox_amd as D
from densebox_amd import synth
h = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16'
net = D.DenseBoxLMLOC(synth.vgg19_standin(0)); synth.fill_params_(net, 11); net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(1, h, w, seed=1).cuda()
for _ in range(12): net.detect(x, K=10, nms_thresh=0.4)
torch.cuda.synchronize()
