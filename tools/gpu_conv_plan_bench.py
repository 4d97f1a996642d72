"""Per-layer timing of the 3x3 backbone layers the way the engine runs them: dbx_conv_plan picks the kernel, the weights are packed in the
layout that kernel wants (plain or fragment order), forward (bias + ReLU) and data gradient (ReLU gate, transposed / flipped weights).
A/B across kernel selections = run it under different environments in one call:
    DBX_P8=0 python tools/gpu_conv_plan_bench.py f16; DBX_P8=1 python tools/gpu_conv_plan_bench.py f16
usage: python tools/gpu_conv_plan_bench.py [dtype] [N] [reps]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densebox_amd import _lib
from densebox_amd._lib import View, ConvDesc, check, ptr, stream_ptr
dtn = sys.argv[1] if len(sys.argv) > 1 else 'f16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dt = _lib.DTYPE_ID[dtn]
L = _lib.lib()
tdt = {'bf16': torch.bfloat16, 'f16': torch.float16}[dtn]
# (name, H, cin, cout)
LAYERS = [('conv2_2', 120, 128, 128), ('conv3_1', 60, 128, 256), ('conv3_2', 60, 256, 256), ('conv4_1', 30, 256, 512), ('conv4_2', 30, 512, 512)]
KEEP = []


def framed(n, h, c, relu=False):
    hp = h + 2
    guard = max(8 * hp, 576 + hp) * c
    flat = torch.zeros(guard * 2 + n * hp * hp * c, dtype=tdt, device='cuda')
    KEEP.append(flat)
    t = flat[guard:guard + n * hp * hp * c].view(n, hp, hp, c)
    v = torch.randn((n, h, h, c), device='cuda')
    t[:, 1:h + 1, 1:h + 1] = (torch.relu(v) if relu else v).to(tdt)
    return t


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def run(name, H, ci, co, epi, mode_plain, mode_frag):
    x = framed(N, H, ci, relu=True); y = framed(N, H, co); g = framed(N, H, co, relu=True)
    xv = View(C.c_void_p(x.data_ptr()), N, H, H, 1, ci, 0, ci); yv = View(C.c_void_p(y.data_ptr()), N, H, H, 1, co, 0, co)
    gv = View(C.c_void_p(g.data_ptr()), N, H, H, 1, co, 0, co)
    d = ConvDesc(dt, 3, 3, 1, ci, co, epi)
    plan = _lib.ConvPlan()
    check(L.dbx_conv_plan(C.byref(d), C.byref(xv), C.byref(yv), C.byref(plan)))
    w = (torch.randn(co, ci, 3, 3, device='cuda') * (2.0 / (9 * ci)) ** 0.5)
    out = torch.zeros(L.dbx_conv_packed_elems(C.byref(d)) * 2, dtype=torch.uint8, device='cuda')
    check(L.dbx_pack_weight(dt, mode_frag if plan.w_frag else mode_plain, ptr(w), co, ci, 3, 3, ptr(out), co, ci, 0, 0, stream_ptr()))
    if plan.w_frag:
        d = ConvDesc(dt, 3, 3, 1, ci, co, epi | _lib.CONV_WFRAG)
    b = torch.zeros(co, device='cuda')
    us = sorted(timeit(lambda: check(L.dbx_conv_forward(C.byref(d), C.byref(xv), ptr(out), ptr(b), C.byref(yv), C.byref(gv) if epi & _lib.EPI_GATE else None,
                                                        None, 0, stream_ptr()))) for _ in range(REPS))
    fl = 2.0 * N * H * H * 9 * ci * co
    print('%-8s %-5s %4d->%-4d %-40s %8.1f us [%.1f..%.1f] %7.1f TFLOP/s' % (name, 'fwd' if epi & _lib.EPI_RELU else 'dgrad', ci, co, plan.name.decode(),
                                                                          us[len(us) // 2], us[0], us[-1], fl / us[len(us) // 2] / 1e6), flush=True)
    KEEP.clear()
    return us[len(us) // 2]


tot = 0.0
print('DBX_P8=%s DBX_WS=%s dtype %s N %d' % (os.environ.get('DBX_P8'), os.environ.get('DBX_WS'), dtn, N))
for name, H, ci, co in LAYERS:
    tot += run(name, H, ci, co, _lib.EPI_BIAS | _lib.EPI_RELU, 0, 4)
    # data gradient of the same layer: a forward conv co -> ci with the transposed / flipped weights (dbx_pack_weight modes 1 / 5 take the
    # OIHW tensor of the forward layer; here a random tensor of the transposed shape in the forward modes times the same)
    tot += run(name, H, co, ci, _lib.EPI_GATE, 0, 4)
print('total %.1f us' % tot)
