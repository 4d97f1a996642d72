import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import densebox_amd as D
from densebox_amd import synth
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
net = D.DenseBox(synth.vgg19_standin(0)); synth.fill_params_(net, 11)
mode = sys.argv[2] if len(sys.argv) > 2 else 'real'
with torch.no_grad():
    if mode == 'tap':   # weight = delta at tap (ky,kx), cout o <- cin (o%3), bias 0
        w = torch.zeros_like(net.conv1_1_1.weight); ky, kx = int(sys.argv[3]), int(sys.argv[4])
        for o in range(64): w[o, o % 3, ky, kx] = 1.0 + o
        net.conv1_1_1.weight.copy_(w); net.conv1_1_1.bias.zero_()
P = {n: p.detach().clone() for n, p in net.named_parameters()}
net = net.cuda().eval(); net.compute_dtype = dtype
x = synth.synth_images(1, 240, 240, seed=3)
with torch.no_grad(): net(x.cuda())
a = net.engine().read_activation('a11').cpu()
r = F.relu(F.conv2d(x, P['conv1_1_1.weight'], P['conv1_1_1.bias'], padding=1))
err = (a - r).abs()
print('max err', err.max().item())
print('err by channel (max):', [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()])
print('err by x%16:', [round(v, 3) for v in torch.stack([err[..., i::16].max() for i in range(16)]).tolist()])
print('err by y%4:', [round(v, 3) for v in torch.stack([err[:, :, i::4].max() for i in range(4)]).tolist()])
print('sample hip', a[0, :8, 5, 5].tolist()); print('sample ref', r[0, :8, 5, 5].tolist())
print('ratio', (a[0, :8, 5, 5] / r[0, :8, 5, 5].clamp(min=1e-6)).tolist())
