"""torch.autograd glue: one Function spans the whole network so that the reference's
``loss.backward(); optimizer.step()`` (DenseBox.py:2186-2187) drives the HIP backward."""
import torch


class NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, X, *params):
        outs = engine.forward_raw(X, train=True)
        names = engine.output_names()
        ctx.engine = engine
        ctx.names = names
        ctx.plan = engine.last_plan
        ctx.pnames = [n for n, _ in engine.net.named_parameters()]
        return tuple(outs[n] for n in names)

    @staticmethod
    def backward(ctx, *grads):
        eng = ctx.engine
        if eng.last_plan is not ctx.plan:
            raise RuntimeError('densebox_amd: backward() after another forward() on the same module -- the '
                               'activations of this graph were overwritten (one graph in flight per module)')
        G = eng.backward_raw(dict(zip(ctx.names, grads)))
        sink = eng.grad_sink
        if sink is not None:
            if getattr(sink, 'in_step', False):
                # data-parallel: gradients live in the reducer's flat buffer and are still being all-reduced;
                # dist.DataParallel.step() attaches them as .grad once the collective is enqueued behind them
                return (None, None) + (None,) * len(ctx.pnames)
            # the reference loop body (net(x); loss.backward(); opt.step()) on a module wrapped by DataParallel
            if sink.world > 1:
                raise RuntimeError('densebox_amd: this module is wrapped by dist.DataParallel -- its gradients are produced and '
                                   'all-reduced inside DataParallel.step(); call that (or DataParallel.close() first) instead '
                                   'of loss.backward()')
            return (None, None) + tuple(sink.views.get(n) for n in ctx.pnames)
        return (None, None) + tuple(G.get(n) for n in ctx.pnames)
