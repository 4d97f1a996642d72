import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, torch.nn.functional as F
import test_hip_backward as TB
from oracle import densebox_oracle as O
class G:
    def __call__(self, name): return np.load('tests/golden/%s.npz' % name)
name = sys.argv[1] if len(sys.argv) > 1 else 'train_DenseBoxLM'
g, kind, net, n, x = TB._setup(G(), name, 'f32')
outs, loss = TB._step(g, kind, net, n, x, 0)
loss.backward(); torch.cuda.synchronize()
eng = net.engine()
# oracle with intermediate grads
P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.named_parameters()}
taps = {}
def conv(nm, t, pad): return F.conv2d(t, P[nm + '.weight'], P[nm + '.bias'], padding=pad)
X = x[:n]
for nm in O.BACKBONE[:7]:
    X = F.relu(conv(nm, X, 1)); X.retain_grad(); taps[nm] = X
    if nm in O.POOL_AFTER: X = F.max_pool2d(X, 2, 2)
c34 = X
X = F.max_pool2d(X, 2, 2)
for nm in O.BACKBONE[7:]:
    X = F.relu(conv(nm, X, 1)); X.retain_grad(); taps[nm] = X
ups = F.interpolate(X, size=c34.shape[2:], mode='bilinear', align_corners=True)
fusion = torch.cat((ups, c34), 1); fusion.retain_grad()
o = {}
for hn, _ in O.HEADS[kind]:
    o[hn] = conv('conv5_2_' + hn, conv('conv5_1_' + hn, fusion, 0), 0)
if kind == 'DenseBox': outs_o = (o['det'], o['loc'])
else:
    f2 = torch.cat((o['landmark'], o['det']), 1); r = F.max_pool2d(f2, 2, 2); r = conv('conv6_1_det', r, 0); r = conv('conv6_2_det', r, 0)
    r = F.interpolate(r, size=o['det'].shape[2:], mode='bilinear', align_corners=True); rf = conv('conv6_3_det', r, 0)
    outs_o = (o['det'], o['loc'], o['landmark'], rf) if kind == 'DenseBoxLM' else (o['det'], rf, o['loc'], o['landmark'], o['lmloc'])
neg0 = g['s0_neg_idx_0']; half = neg0.shape[1] // 2
lm_rand = None if kind == 'DenseBox' else np.stack([g['s0_neg_idx_%d' % (1 + j)][:, 1:] for j in range(4)])
kw = {k[3:]: float(g[k]) for k in g.files if k.startswith('kw_')}
res = O.loss_step(kind, outs_o, g['bbox'][:n], g['vert'][:n], g['lab'][:n], rand_neg=neg0[:, half:], lm_rand_neg=lm_rand, **kw)
res['loss'].backward()
def cmp(label, hip, ref):
    err = (hip - ref).abs(); sc = ref.abs().max().item()
    bad = (err > 1e-4 * sc).nonzero()
    print('%-10s max|ref| %.3e  max err %.3e (%.1e rel)  #bad %d  first %s' % (label, sc, err.max().item(), err.max().item() / sc, bad.shape[0], bad[:5].tolist()))
    return bad
cmp('d_fus_ups', eng.read_activation('d_ups').cpu(), fusion.grad[:, :512])
# relu-gated grads: dZ = dY * (Y>0)
def gated(t): return t.grad * (t > 0).float()
bad = cmp('d_c34', eng.read_activation('d_c34').cpu(), gated(taps['conv3_4_1']))
cmp('d_a32', eng.read_activation('d_a32').cpu(), gated(taps['conv3_2_1']))
cmp('d_a31', eng.read_activation('d_a31').cpu(), gated(taps['conv3_1_1']))
cmp('d_a44', eng.read_activation('d_a44').cpu(), gated(taps['conv4_4_1']))
cmp('d_a41', eng.read_activation('d_a41').cpu(), gated(taps['conv4_1_1']))
if bad.shape[0]:
    b = bad[0]; n_, c_, y_, x_ = [int(v) for v in b]
    print('at', b.tolist(), 'hip', eng.read_activation('d_c34')[n_, c_, y_, x_].item(), 'ref', gated(taps['conv3_4_1'])[n_, c_, y_, x_].item(),
          'fusion-part ref', (fusion.grad[:, 512:] * (c34 > 0).float())[n_, c_, y_, x_].item())
    win = c34[n_, c_, (y_//2)*2:(y_//2)*2+2, (x_//2)*2:(x_//2)*2+2]
    print('pool window (cpu act):', win.tolist())
    hw = eng.read_activation('fusion', 512, 256)[n_, c_, (y_//2)*2:(y_//2)*2+2, (x_//2)*2:(x_//2)*2+2]
    print('pool window (hip act):', hw.tolist())
