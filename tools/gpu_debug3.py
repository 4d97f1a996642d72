import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_hip_backward as TB
class G:
    def __call__(self, name): return np.load('tests/golden/%s.npz' % name)
for name in sys.argv[1:]:
    g, kind, net, n, x = TB._setup(G(), name, 'f32')
    outs, loss = TB._step(g, kind, net, n, x, 0)
    print(name, 'loss', float(loss), float(g['s0_loss']))
    loss.backward()
    for pname, prm in net.named_parameters():
        if 's0_gnone_' + pname in g.files: continue
        gr = prm.grad.detach().float().cpu().numpy()
        if 's0_g_' + pname in g.files: ref, got = g['s0_g_' + pname], gr
        else: ref, got = g['s0_gsub_' + pname], gr.reshape(-1)[::997]
        print('  %-26s relerr %.2e   l1 %.4e ref %.4e' % (pname, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30), np.abs(gr).sum(), g['s0_gstat_' + pname][1]))
