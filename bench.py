#!/usr/bin/env python3
"""bench.py -- DenseBox training throughput on MI355X (contract: one JSON line from rank 0).

A "step" = one full training step of the reference loop body (DenseBox.py:2016-2187) on a batch of synthetic
240x240 patches already resident in HBM: forward, fused dense loss with hard-negative mining, backward,
gradient all-reduce over RCCL (N>1), fused SGD.  Workload = BASELINE.json configs[2]/[3]: DenseBoxLMLOC (the net
the reference's __main__ trains: score + bbox + landmark heat-maps + landmark offsets + refine), batch 64 per GPU,
f16 compute (--dtype bf16 for configs[3]'s type; same MFMA rate) / fp32 accumulate / fp32 master weights.  Weak scaling:
per-GPU batch fixed.  The line also carries the bf16 step rate (`bf16`) and the inference half of the metric (`inference`).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import densebox_amd as D                      # noqa: E402
from densebox_amd import synth, labels as LB  # noqa: E402
from densebox_amd.dist import init_from_env, DataParallel, preflight, report_failure  # noqa: E402
from densebox_amd.optim import SGD            # noqa: E402

# algorithmic FLOP per 240x240 patch (SURVEY.md 8d / BASELINE.md 2): forward, and forward+dgrad+wgrad
FWD_GFLOP = {'DenseBox': 41.98, 'DenseBoxLM': 44.95, 'DenseBoxLMLOC': 47.81}
STEP_GFLOP = {'DenseBox': 125.7, 'DenseBoxLM': 134.6, 'DenseBoxLMLOC': 143.2}
MFMA_PEAK_TF = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3}     # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
PMC_FILE = 'r06_pmc_traffic.json'


def conv_flops(eng_calls):
    return sum(c['flops'] for c in eng_calls)


def host_cpu():
    """Model string, physical core count and logical CPU count of the host (/proc/cpuinfo; BASELINE.md section 4 asks for both)."""
    model, cores, logical = None, set(), 0
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'processor':
                logical += 1
            elif k == 'model name' and model is None:
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return {'model': model, 'physical_cores': len(cores) or None, 'logical_cpus': logical or os.cpu_count(),
            'sockets': len({p for p, _ in cores}) or None}


def cpu_baseline(kind, seconds=20.0):
    """Reference algorithm on the host cores: the CPU oracle (torch fp32 nn ops + numpy bookkeeping) running the same
    training step on a bounded sample (batch 8, >= 5 timed steps), torch threads = the host's physical cores.  Reported baseline,
    not a target."""
    from oracle import densebox_oracle as O
    cpu = host_cpu()
    threads = cpu['physical_cores'] or torch.get_num_threads()
    torch.set_num_threads(int(threads))
    n = 8
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.named_parameters()}
    x, bbox, vert, lab = synth.synth_batch(n, seed=3, neg_frac=0.0)
    _, half = LB.neg_counts(int(LB.positive_count(bbox, lab).sum()), n)
    rn = synth.synth_rand_neg_indices(n, half, seed=1).numpy()
    lrn = synth.synth_rand_neg_indices(4 * n, 1, seed=2).reshape(4, n, 1).numpy()
    bufs = {}

    def step():
        for p in P.values():
            p.grad = None
        outs = O.forward(kind, P, x)
        res = O.loss_step(kind, outs, bbox.numpy(), vert.numpy(), lab.numpy(), rand_neg=rn, lm_rand_neg=lrn)
        res['loss'].backward()
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is None:
                    continue
                newp, bufs[k] = O.sgd_step(p.detach(), p.grad, bufs.get(k), 1e-9)
                p.copy_(newp)
    step()                                    # warm-up
    t0 = time.perf_counter()
    it = 0
    while True:
        step()
        it += 1
        el = time.perf_counter() - t0
        if (el >= seconds and it >= 5) or it >= 50:
            break
    out = {'value': round(n * it / el, 3), 'unit': 'patches/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'host': cpu,
           'sample': '%d training steps of %d synthetic 240x240 patches (%s, fp32, oracle/densebox_oracle.py), %.1f s, %d torch threads'
                     % (it, n, kind, el, torch.get_num_threads())}
    # BASELINE.md section 4, cases (i) and (iii): eval forward at N = 1 / 8 and the whole-image inference chain (forward + top-K
    # decode + NMS) at 512x512 and 1920x1080, same oracle, bounded to a few seconds each
    Pd = {k: v.detach() for k, v in P.items()}

    def timed(fn, budget, max_it):
        fn()
        t0 = time.perf_counter()
        k = 0
        while True:
            fn()
            k += 1
            e = time.perf_counter() - t0
            if e >= budget or k >= max_it:
                return e / k, k
    cases = {}
    # one `DenseBox` (score + bbox heads) training step at N = 8, same oracle
    n8 = 8
    net8 = D.DenseBox(synth.vgg19_standin(seed=0))
    synth.fill_params_(net8, 11)
    P8 = {k: v.detach().clone().requires_grad_(True) for k, v in net8.named_parameters()}
    x8, bbox8, vert8, lab8 = synth.synth_batch(n8, seed=7, neg_frac=0.0)
    _, half8 = LB.neg_counts(int(LB.positive_count(bbox8, None).sum()), n8)
    rn8 = synth.synth_rand_neg_indices(n8, half8, seed=1).numpy()

    def step8():
        for p in P8.values():
            p.grad = None
        res = O.loss_step('DenseBox', O.forward('DenseBox', P8, x8), bbox8.numpy(), vert8.numpy(), None, rand_neg=rn8, lm_rand_neg=None)
        res['loss'].backward()
    sec, k = timed(step8, 6.0, 6)
    cases['train_step_DenseBox_N8_240x240'] = {'patches_per_s': round(n8 / sec, 3), 'iterations': k}
    with torch.no_grad():
        for nn_ in (1, 8):
            xf = synth.synth_batch(nn_, seed=5, neg_frac=0.0)[0]
            sec, k = timed(lambda: O.forward('DenseBox', Pd, xf), 2.0, 10)
            cases['forward_DenseBox_N%d_240x240' % nn_] = {'patches_per_s': round(nn_ / sec, 2), 'iterations': k}
        for name, (h, w) in (('512x512', (512, 512)), ('1920x1080', (1080, 1920))):
            img = synth.synth_images(1, h, w, seed=1)

            def chain():
                o = O.forward('DenseBox', Pd, img)
                sc, lc = o[0], o[1]
                dets = O.parse_det(sc[0, 0].numpy(), lc[0].numpy(), h, w, K=10)
                O.nms(dets, 0.4)
            sec, k = timed(chain, 4.0, 5)
            cases['inference_DenseBox_%s' % name] = {'img_per_s': round(1.0 / sec, 3), 'iterations': k}
    out['cases'] = cases
    return out


def achieved_tolerance(kind, dtype, dev):
    """Forward error of this compute dtype against the reference's own outputs, MEASURED in this run: the eval-mode forward of `kind` on the
    two 240x240 patches of tests/golden/net_<kind>.npz (outputs captured by running the reference, oracle/gen_golden.py) -- per output map
    max |hip - ref| / max(1, max|ref|) and the same in RMS.  north_star asks for 1e-3: fp32 meets it in max norm, f16 on the bbox maps and in
    RMS on every map (13 stacked layers rounding to 11 bits; profiles/r06_layer_error_budget_f16.txt), bf16 in neither -- the line carries the
    numbers so that `dtype` is never read as "1e-3 parity at f16"."""
    path = os.path.join(ROOT, 'tests', 'golden', 'net_%s.npz' % kind)
    if not os.path.exists(path):
        return None
    g = np.load(path)
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, int(g['param_seed']))
    net = net.to(dev).eval()
    net.compute_dtype = dtype
    with torch.no_grad():
        outs = net(synth.synth_images(2, 240, 240, seed=3).to(dev))
    names = {'DenseBox': ['score', 'bbox'], 'DenseBoxLM': ['score', 'bbox', 'landmark', 'refine'],
             'DenseBoxLMLOC': ['score', 'refine', 'bbox', 'lm_heat', 'lm_loc']}[kind]
    per = {}
    for i, o in enumerate(outs):
        ref = g['out240_%d' % i]
        a = o.float().cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        per[names[i]] = {'max': float('%.3g' % (float(np.abs(a - ref).max()) / scale)),
                         'rms': float('%.3g' % (float(np.sqrt(np.mean((a.astype(np.float64) - ref) ** 2))) / scale))}
    return {'max': max(v['max'] for v in per.values()), 'rms': max(v['rms'] for v in per.values()), 'per_map': per,
            'of': 'max(1, max|reference map|) per output map of %s, reference-captured fixture (fp32 PyTorch CPU), 2 patches' % kind,
            'north_star': 1e-3, 'source': 'measured in this run (tests/golden/net_%s.npz)' % kind}


def csrc_hash():
    """Hash of the kernel sources: a PMC summary is only quoted for the kernels it was measured on."""
    import hashlib
    d = os.path.join(ROOT, 'densebox_amd', 'csrc')
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp')):
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:12]


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed PMC summary (profiles/r03_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, FETCH_SIZE doubled per the gfx950 correction
    in MI355X_MICROARCH.md; tools/pmc_traffic.py stamps it with csrc_hash()).  None when the summary is absent, was
    collected on other kernel sources, or does not list the family."""
    path = os.path.join(ROOT, 'profiles', PMC_FILE)
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    if doc.get('csrc_hash') != csrc_hash():
        return None
    # the library names a family by its kernel template and the leading template arguments (dbx_conv_plan / dbx_conv_wgrad_plan);
    # a mangled symbol carries them as one contiguous run: name I <type code> Li<int>E ... -- matched as that exact prefix
    m = re.match(r'(\w+)<(\w+)((?:,\d+)*)>', family)
    if not m:
        return None
    code = {'bf16': 'DF16b', 'f16': 'DF16_', 'f32': 'f'}[m.group(2)]
    targs = [v for v in m.group(3).split(',') if v]
    if m.group(1) == 'conv3x3_p8_kernel':
        # the 8-phase template is <T, KS, FLAGS, EPIK> while its plan names are <T,KS> (every 3x3 / 1x1 epilogue but the fused heads) and
        # <T,1,1> (KS 1, EPIK 1: the fused heads forward): match KS and EPIK by POSITION, the FLAGS argument in between is not part of the name
        ks = targs[0]
        heads = len(targs) > 1
        pat = re.compile(r'conv3x3_p8_kernelI%sLi%sELi\d+ELi(\d+)E' % (re.escape(code), ks))
        hits = []
        for k, v in doc.get('kernels', {}).items():
            mm = pat.search(k)
            if mm and (mm.group(1) == '1') == (heads and ks == '1') and (heads or ks != '1' or mm.group(1) != '1'):
                hits.append(v)
    else:
        want = '%sI%s%s' % (m.group(1), code, ''.join('Li%sE' % v for v in targs))
        hits = [v for k, v in doc.get('kernels', {}).items() if want in k]
    if not hits:
        return None
    # (a family may be several instantiations of one template behind the named arguments -- the 8-phase kernel's forward, gated
    #  data-gradient and pooling epilogues all run under one plan name: launch-weighted mean over exactly those instantiations)
    nl = sum(v['launches'] for v in hits)
    return round(sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hits) / nl) if nl else None


def inference(kind, dev):
    """Second half of BASELINE.json's metric: whole-image forward + on-GPU top-10 decode + NMS, one image at a time (the
    reference test drivers, DenseBox.py:3772-3799), results on the host.  f16, eval mode (folded heads, hipGraph replay)."""
    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.to(dev).eval()
    net.compute_dtype = 'f16'
    res = {}
    for name, (h, w, K) in (('512x512', (512, 512, 10)), ('1920x1080', (1080, 1920, 10)), ('1920x1080_top100', (1080, 1920, 100)),
                            ('1920x1080_top1000', (1080, 1920, 1000))):
        x = synth.synth_images(1, h, w, seed=1).to(dev)
        for _ in range(3):
            net.detect(x, K=K, nms_thresh=0.4)
        torch.cuda.synchronize()
        it = 30 if h <= 512 else 15
        t0 = time.perf_counter()
        for _ in range(it):
            dets, keep = net.detect(x, K=K, nms_thresh=0.4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        res[name] = {'img_per_s': round(1.0 / dt, 1), 'ms': round(dt * 1e3, 3), 'topk': K,
                     'tflops': round(FWD_GFLOP[kind] * h * w / (240.0 * 240.0) / dt / 1e3, 1)}
    return {'metric': 'inference fps (forward + top-K + NMS, one image, results on host)', 'net': kind, 'dtype': 'f16',
            'value': res['512x512']['img_per_s'], 'unit': 'img/s at 512x512', 'sizes': res}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # rank 0's stdout passes through; a job that dies without a JSON line (a rank killed before it could report) still ends in one
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    seen = False
    for line in proc.stdout:
        seen = seen or line.startswith('{')
        sys.stdout.write(line); sys.stdout.flush()
    rc = proc.wait()
    if rc != 0 and not seen:
        from densebox_amd.dist import rccl_info
        print(json.dumps({'error': 'launcher exited with code %d before rank 0 printed a line' % rc, 'rank': None, 'rccl': rccl_info()}), flush=True)
    return rc


def measured_peaks(dev, dtype):
    """Achievable peaks measured on this box: a large library GEMM (hipBLASLt through torch.mm) in the
    compute dtype and a streaming device copy.  Reference points for the roofline fractions, never part of the product path."""
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[dtype]
    m = 8192
    a = torch.randn(m, m, device=dev).to(tdt); b = torch.randn(m, m, device=dev).to(tdt)
    for _ in range(3):
        torch.mm(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(a, b)
    e1.record(); torch.cuda.synchronize()
    gemm = 10 * 2.0 * m ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
    dst.copy_(src)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    copy = 10 * 2.0 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return {'library_gemm_8192_tflops': round(gemm, 1), 'stream_copy_GBps': round(copy, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--kind', default='DenseBoxLMLOC', choices=list(FWD_GFLOP))
    ap.add_argument('--dtype', default='f16', choices=['bf16', 'f16', 'f32'],
                    help='compute dtype; f16 = BASELINE configs[1]/[2] and 10x closer to the reference than bf16 (profiles/r02_lowprec_errors.json)')
    ap.add_argument('--batch', type=int, default=64, help='patches per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--no-inference', action='store_true')
    ap.add_argument('--dry-run', action='store_true', help='launcher / rendezvous only: no GPU work (CPU test of the N>1 launch path)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))            # one rank per GPU; rank 0 of the child job prints the JSON line
    try:
        run(args)
    except SystemExit:
        raise
    except BaseException as e:                 # a failed rendezvous / collective / launch: ONE parsable line, then the traceback
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            report_failure('bench.py --gpus %d failed' % args.gpus, e)
        raise


def run(args):
    rank, world, local = init_from_env()
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    if args.dry_run:
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({'metric': 'training patches/sec (240x240)', 'dry_run': True, 'n_gpus': world,
                              'rank_sum': int(t.item()), 'backend': dist.get_backend() if world > 1 else None}))
        if world > 1:
            dist.destroy_process_group()
        return
    if world > torch.cuda.device_count():
        # several ranks share a GPU (DBX_DIST_BACKEND=gloo, functional runs of the N > 1 path on a 1-GPU box): create the device
        # contexts one after the other -- eight processes initialising on one (virtual) device at the same instant faulted once in
        # the first allocation's fill kernel
        time.sleep(0.3 * rank)
    local = local % torch.cuda.device_count()          # (several ranks may share a GPU under DBX_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    preflight(dev)                                     # first collective under a watchdog: a broken RCCL set-up reports instead of hanging
    kind, n = args.kind, args.batch

    net = getattr(D, kind)(synth.vgg19_standin(seed=0))
    synth.fill_params_(net, 11)
    net = net.to(dev).train()
    net.compute_dtype = args.dtype
    opt = SGD(net.parameters(), lr=1e-9, momentum=0.9, weight_decay=5e-8)       # DenseBox.py:2001-2004, :3841-3850
    dp = DataParallel(net, opt)

    # synthetic data, resident in HBM before the timed region; every rank draws its own shard of the global batch
    x, bbox, vert, lab = synth.synth_batch(n, seed=100 + rank, neg_frac=0.1)
    x = x.to(dev)
    if os.environ.get('DBX_BENCH_ZERO') == '1':
        # power experiment, never a reported number: all-zero images and parameters -- same instruction streams,
        # (almost) no operand toggling; how much faster the step runs says how far the part is power-managed on real data
        x.zero_()
        with torch.no_grad():
            for p_ in net.parameters():
                p_.zero_()
    rs = np.random.RandomState(1234 + rank)
    use_lab = lab if kind == 'DenseBoxLMLOC' else None
    half = 0

    def one_step():
        # everything the reference loop body does per iteration stays inside the step: the global positive count
        # (DenseBox.py:2070-2074; host loop over the boxes + one int64 all-reduce on the gloo side group) and the host RNG
        # draws (DenseBox.py:2089-2094, :2133-2138)
        nonlocal half
        p_global = dp.global_positive_num(bbox, use_lab)       # (consumes the count prefetched behind the previous step's launches)
        _, half = LB.neg_counts(p_global, n * world)
        rn = np.stack([rs.choice(3600, half, replace=False) for _ in range(n)]) if half else np.zeros((n, 0), np.int64)
        lrn = rs.randint(0, 3600, size=(4, n, 1)) if kind != 'DenseBox' else None
        out = dp.step(x, bbox, vert, lab, rand_neg_indices=rn, lm_rand_neg_indices=lrn, positive_num_global=p_global)
        # the NEXT step's positive count depends on its labels only: its int64 all-reduce (gloo side group) flies while this step's
        # kernels drain instead of standing in front of the next forward's first launch
        dp.prefetch_positive_num(bbox, use_lab)
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.detach())
    assert np.isfinite(loss_val), 'training step produced a non-finite loss'
    # 16-bit training runs without loss scaling (the reference loss is an un-normalised sum: the risk is overflow of the f16
    # activation gradients, not underflow): every parameter gradient of the last timed step must be finite (outside the timed region)
    gflat = dp.reducer.flat
    assert bool(torch.isfinite(gflat).all()), 'non-finite parameter gradient in %s training' % args.dtype
    grad_absmax = float(gflat.abs().max())

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream: 3 extra, untimed steps.
    # EVERY rank runs them (they contain collectives); only rank 0 records events.
    eng = net.engine()
    if rank == 0:
        eng.profile = []
    for _ in range(3):
        one_step()
    barrier()

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = n * world * args.steps / dt
        calls = eng.profile
        eng.profile = None
        fam = {}
        for c in calls:
            f = fam.setdefault(c['kernel'], {'us': 0.0, 'flops': 0.0, 'launches': 0})
            f['us'] += c['start'].elapsed_time(c['end']) * 1e3
            f['flops'] += c['flops']
            f['launches'] += 1
        dom = max(fam, key=lambda k: fam[k]['us'])
        fd = fam[dom]
        ach = fd['flops'] / (fd['us'] * 1e-6) / 1e12
        peak = MFMA_PEAK_TF[args.dtype]
        roof = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': pmc_traffic(dom), 'traffic_unit': 'HBM bytes per launch (PMC)',
                'launches_per_step': fd['launches'] // 3, 'avg_launch_us': round(fd['us'] / fd['launches'], 2),
                'families': {k: {'tflops': round(v['flops'] / (v['us'] * 1e-6) / 1e12, 1), 'us_per_step': round(v['us'] / 3, 1),
                                 'launches_per_step': v['launches'] // 3} for k, v in fam.items()}}
        # north_star's own figure: the whole 3x3 conv stack (every 3x3 layer's forward, data gradient and weight gradient incl.
        # its split reduction) as algorithmic FLOP over the HIP-event time of those launches, against the same peak
        def agg(pred):
            us = sum(v['us'] for k, v in fam.items() if pred(k))
            fl = sum(v['flops'] for k, v in fam.items() if pred(k))
            return {'tflops': round(fl / (us * 1e-6) / 1e12, 1), 'frac': round(fl / (us * 1e-6) / 1e12 / peak, 4),
                    'us_per_step': round(us / 3, 1)} if us > 0 else None
        def is3(k):          # a 3x3 forward / data-gradient family (the ws / p8 templates also serve 1x1 GEMMs: <T,WM,1,..> / <T,1>)
            return k.startswith('conv3x3_') and not re.match(r'conv3x3_ws_kernel<\w+,\d+,1,', k) and not re.match(r'conv3x3_p8_kernel<\w+,1[,>]', k)
        # the dominant forward / data-gradient conv family on its own (round 1's dominant kernel was one: continuity of the series)
        convs = {k: v for k, v in fam.items() if is3(k)}
        if convs:
            dc = max(convs, key=lambda k: convs[k]['us'])
            ac = convs[dc]['flops'] / (convs[dc]['us'] * 1e-6) / 1e12
            roof['dominant_conv'] = {'kernel': dc, 'achieved': round(ac, 1), 'frac': round(ac / peak, 4),
                                     'launches_per_step': convs[dc]['launches'] // 3, 'us_per_step': round(convs[dc]['us'] / 3, 1),
                                     'traffic': pmc_traffic(dc)}
        # (the ws kernel template serves 3x3 layers <T,WM,3,EPIK> and the 1x1 head GEMMs <T,1,1,EPIK>: only the former belong here)
        roof['stack_3x3'] = {'fwd_dgrad': agg(is3),
                             'with_wgrad': agg(lambda k: is3(k) or k.startswith(('wgrad_all9', 'wgrad_row3', 'wgrad3x3')))}
        out = {
            'metric': 'training patches/sec (240x240)', 'value': round(value, 1), 'unit': 'patches/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'tolerance': achieved_tolerance(kind, args.dtype, dev), 'data': 'synthetic',
            'config': {'workload': 'full training step (fwd + fused dense loss w/ hard-negative mining + bwd + grad '
                                   'all-reduce + SGD) of %s on 240x240 patches' % kind,
                       'net': kind, 'batch_per_gpu': n, 'global_batch': n * world, 'patch': '240x240',
                       'parallelism': 'dp%d' % world, 'half_neg': half, 'loss': round(loss_val, 2),
                       'grad_finite': True, 'grad_absmax': round(grad_absmax, 3),
                       'f16_overflow_guard': {'on': bool(dp.guard_f16 and args.dtype == 'f16'), 'skipped_steps': dp.skipped_steps()}},
            'step_tflops_per_gpu': round(n * STEP_GFLOP[kind] / (ms * 1e-3) / 1e3, 1),
            'roofline': roof,
            # the live process group the gradient all-reduce ran on ("nccl" is RCCL on ROCm): lets a scaling run be checked for
            # "RCCL saw N ranks"
            'rccl': {'backend': dist.get_backend() if dist.is_initialized() else None,
                     'world': dist.get_world_size() if dist.is_initialized() else 1,
                     'collective': bool(dp.reducer.collective), 'bucket_bytes': int(dp.reducer.bucket_elems * 4)},
        }
        mp = measured_peaks(dev, args.dtype)
        roof['measured_peak'] = mp
        roof['frac_of_measured_gemm'] = round(ach / mp['library_gemm_8192_tflops'], 4)
        out['roofline']['traffic_source'] = 'profiles/%s (null unless collected on these kernel sources)' % PMC_FILE
    # the other 16-bit type on the same workload (configs[3] names bf16): a short secondary measurement, every rank takes part
    other = {'f16': 'bf16', 'bf16': 'f16'}.get(args.dtype)
    if other and not args.no_inference:
        net.compute_dtype = other
        for _ in range(3):
            one_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            one_step()
        barrier()
        dt2 = time.perf_counter() - t1
        net.compute_dtype = args.dtype
        if rank == 0:
            out[other] = {'value': round(n * world * 10 / dt2, 1), 'unit': 'patches/s', 'ms_per_step': round(dt2 / 10 * 1e3, 3), 'steps': 10}
    if rank == 0:
        if not args.no_inference:
            out['inference'] = inference(kind, dev)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(kind, args.cpu_seconds)
    dp.drain_prefetch()
    barrier()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
