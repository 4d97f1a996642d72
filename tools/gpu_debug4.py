import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_hip_backward as TB
class G:
    def __call__(self, name): return np.load('tests/golden/%s.npz' % name)
g, kind, net, n, x = TB._setup(G(), 'train_DenseBoxLM', 'f32')
outs, loss = TB._step(g, kind, net, n, x, 0)
P = net.engine().last_plan
torch.cuda.synchronize()
before = P.ws.clone()
loss.backward(); torch.cuda.synchronize()
after = P.ws
fwd = ['x0','a11','a12','p1','a21','a22','p2','a31','a32','fusion','p3','a41','a42','a43','a44','hid','rf_in','rf_p','rf_1','rf_2','rf_u']
for name in fwd:
    b = P.B[name]
    lo, hi = b.off - b.guard, b.off + ((b.bytes + 255)//256*256) + b.guard
    ch = (before[lo:hi] != after[lo:hi]).nonzero().flatten()
    print('%-7s changed bytes %d' % (name, ch.numel()), (ch[:4] + lo - b.off).tolist() if ch.numel() else '')
# also: are the frames (halo) of every buffer zero after everything?
for name, b in P.B.items():
    t = after[b.off:b.off + b.bytes].view(torch.float32).view(b.n, b.hp, b.wp, b.c)
    if b.pad:
        inner = t[:, b.pad:-b.pad, b.pad:-b.pad]
        tot = t.abs().sum().item(); inn = inner.abs().sum().item()
        if abs(tot - inn) > 0: print('  frame of', name, 'is NOT zero: ', tot - inn)
    gz = after[b.off - b.guard:b.off].abs().sum().item() + after[b.off + b.bytes:b.off + b.bytes + b.guard].abs().sum().item()
    if gz: print('  guard of', name, 'not zero', gz)
