"""Layer-by-layer max-error report of the HIP forward vs torch CPU (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import densebox_amd as D
from densebox_amd import synth
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
kind = sys.argv[2] if len(sys.argv) > 2 else 'DenseBox'
net = getattr(D, kind)(synth.vgg19_standin(0)); synth.fill_params_(net, 11)
P = {n: p.detach().clone() for n, p in net.named_parameters()}
net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(2, 240, 240, seed=3)
with torch.no_grad():
    outs = net(x.cuda())
eng = net.engine()
def conv(n, t, pad=1): return F.conv2d(t, P[n + '.weight'], P[n + '.bias'], padding=pad)
ref = {}
t = x
ref['x0'] = t
t = F.relu(conv('conv1_1_1', t)); ref['a11'] = t
t = F.relu(conv('conv1_2_1', t)); ref['a12'] = t
t = F.max_pool2d(t, 2, 2); ref['p1'] = t
t = F.relu(conv('conv2_1_1', t)); ref['a21'] = t
t = F.relu(conv('conv2_2_1', t)); ref['a22'] = t
t = F.max_pool2d(t, 2, 2); ref['p2'] = t
t = F.relu(conv('conv3_1_1', t)); ref['a31'] = t
t = F.relu(conv('conv3_2_1', t)); ref['a32'] = t
t = F.relu(conv('conv3_4_1', t)); c34 = t
t = F.max_pool2d(t, 2, 2); ref['p3'] = t
t = F.relu(conv('conv4_1_1', t)); ref['a41'] = t
t = F.relu(conv('conv4_2_1', t)); ref['a42'] = t
t = F.relu(conv('conv4_3_1', t)); ref['a43'] = t
t = F.relu(conv('conv4_4_1', t)); ref['a44'] = t
ups = F.interpolate(t, size=c34.shape[2:], mode='bilinear', align_corners=True)
ref['fusion'] = torch.cat((ups, c34), 1)
for name, r in ref.items():
    c = 3 if name == 'x0' else None
    a = eng.read_activation(name, 0, c).cpu()
    err = (a - r).abs()
    print('%-7s shape %-22s max|ref| %8.3f  max err %.3e  mean err %.3e  first bad idx %s' % (
        name, tuple(r.shape), r.abs().max(), err.max(), err.mean(),
        tuple(int(v) for v in (err > 1e-2 * max(1, float(r.abs().max()))).nonzero()[0]) if (err > 1e-2 * max(1, float(r.abs().max()))).any() else None))
