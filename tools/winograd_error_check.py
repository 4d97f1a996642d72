"""error of Winograd F(2x2, 3x3) with 16-bit transformed operands on conv4_2's real operands,
against the direct convolution with 16-bit operands -- both with fp32 accumulation, both against the fp32 reference.  CPU only.
usage: python tools/winograd_error_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import densebox_amd as D
from densebox_amd import synth
from oracle import densebox_oracle as O

torch.manual_seed(0)
net = D.DenseBoxLMLOC(synth.vgg19_standin(seed=0)); synth.fill_params_(net, 11)
P = {k: v.detach() for k, v in net.named_parameters()}
x = synth.synth_images(2, 240, 240, seed=3)
# activations up to conv4_1 (the oracle's forward is one function: redo the prefix with torch)
def cr(t, s): return F.relu(F.conv2d(t, P[s + '.weight'], P[s + '.bias'], padding=1))
a = cr(cr(x, 'conv1_1_1'), 'conv1_2_1'); a = F.max_pool2d(a, 2)
a = cr(cr(a, 'conv2_1_1'), 'conv2_2_1'); a = F.max_pool2d(a, 2)
a = cr(cr(cr(a, 'conv3_1_1'), 'conv3_2_1'), 'conv3_4_1'); a = F.max_pool2d(a, 2)
a41 = cr(a, 'conv4_1_1')                                   # [2, 512, 30, 30]
w, b = P['conv4_2_1.weight'], P['conv4_2_1.bias']
ref = F.conv2d(a41, w, b, padding=1)

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

def wino(xin, wt, bias, dt):
    n, c, h, wd = xin.shape
    xp = F.pad(xin, (1, 1, 1, 1))
    # tiles of 4x4 with stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                 # [n, c, th, tw, 4, 4]
    V = torch.einsum('ij,nchwjk,lk->nchwil', BT, t, BT).to(dt).float()
    U = torch.einsum('ij,ocjk,lk->ocil', G, wt, G).to(dt).float()
    M = torch.einsum('ocil,nchwil->nohwil', U, V)          # fp32 accumulation over c
    Y = torch.einsum('ij,nohwjk,lk->nohwil', AT, M, AT)    # [n, o, th, tw, 2, 2]
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, wt.shape[0], 2 * th, 2 * tw) + bias.view(1, -1, 1, 1)

scale = float(ref.abs().max())
for name, dt in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    direct = F.conv2d(a41.to(dt).float(), w.to(dt).float(), b, padding=1)
    wg = wino(a41, w, b, dt)
    for tag, y in (('direct', direct), ('winograd F(2,3)', wg)):
        e = (y - ref).abs()
        print('%-5s %-16s max err / max|ref| %.3e   rms / max|ref| %.3e' % (name, tag, float(e.max()) / scale, float((e ** 2).mean().sqrt()) / scale))
chk = (wino(a41, w, b, torch.float32) - ref).abs().max() / scale
print('fp32 winograd vs fp32 direct (algebra check): %.2e' % float(chk))
