"""Host-side execution engine: plans framed-NHWC buffers in one HBM workspace and drives the
HIP kernels of libdensebox_hip.so for forward / backward of the three DenseBox networks.

torch is used for device memory, streams and autograd bookkeeping only; every FLOP of the
network runs in the library.  Layer graph = DenseBox.py:180-228 / :412-473 / :674-738.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import View, ConvDesc, check, ptr, stream_ptr

# (param stem, cin, cout) of the executed backbone, in order; pools after conv1_2, conv2_2, (conv3_4 -> pool3)
_BACKBONE = [('conv1_1_1', 3, 64), ('conv1_2_1', 64, 64), ('conv2_1_1', 64, 128), ('conv2_2_1', 128, 128),
             ('conv3_1_1', 128, 256), ('conv3_2_1', 256, 256), ('conv3_4_1', 256, 256),
             ('conv4_1_1', 256, 512), ('conv4_2_1', 512, 512), ('conv4_3_1', 512, 512), ('conv4_4_1', 512, 512)]

# heads in hidden-buffer order: (stem, k)
_HEADS = {
    'DenseBox': [('det', 1), ('loc', 4)],
    'DenseBoxLM': [('det', 1), ('loc', 4), ('landmark', 4)],
    'DenseBoxLMLOC': [('det', 1), ('loc', 4), ('landmark', 4), ('lmloc', 8)],
}


# 3x3 backbone layer -> (input buffer, output buffer) of its forward conv and (dz buffer, dx buffer) of its data gradient
_FWD_IO = {'conv1_1_1': ('x0', 'a11'), 'conv1_2_1': ('a11', 'a12'), 'conv2_1_1': ('p1', 'a21'), 'conv2_2_1': ('a21', 'a22'),
           'conv3_1_1': ('p2', 'a31'), 'conv3_2_1': ('a31', 'a32'), 'conv3_4_1': ('a32', 'c34'), 'conv4_1_1': ('p3', 'a41'),
           'conv4_2_1': ('a41', 'a42'), 'conv4_3_1': ('a42', 'a43'), 'conv4_4_1': ('a43', 'a44')}
_BWD_IO = {'conv4_4_1': ('d_a44', 'd_a43'), 'conv4_3_1': ('d_a43', 'd_a42'), 'conv4_2_1': ('d_a42', 'd_a41'),
           'conv4_1_1': ('d_a41', 'd_p3'), 'conv3_4_1': ('d_c34', 'd_a32'), 'conv3_2_1': ('d_a32', 'd_a31'),
           'conv3_1_1': ('d_a31', 'd_p2'), 'conv2_2_1': ('d_a22', 'd_a21'), 'conv2_1_1': ('d_a21', 'd_p1'),
           'conv1_2_1': ('d_a12', 'd_a11')}
_CH = {s: (ci, co) for s, ci, co in [('conv1_1_1', 3, 64), ('conv1_2_1', 64, 64), ('conv2_1_1', 64, 128), ('conv2_2_1', 128, 128),
                                     ('conv3_1_1', 128, 256), ('conv3_2_1', 256, 256), ('conv3_4_1', 256, 256),
                                     ('conv4_1_1', 256, 512), ('conv4_2_1', 512, 512), ('conv4_3_1', 512, 512),
                                     ('conv4_4_1', 512, 512)]}


def _align(x, a=256):
    return (x + a - 1) // a * a


class Buf:
    """A framed NHWC buffer inside the workspace."""

    def __init__(self, name, n, h, w, c, pad, dtype_id):
        self.name, self.n, self.h, self.w, self.c, self.pad = name, n, h, w, c, pad
        self.dtype_id = dtype_id
        self.es = _lib.ESIZE[dtype_id]
        self.hp, self.wp = h + 2 * pad, w + 2 * pad
        self.bytes = n * self.hp * self.wp * c * self.es
        # guard band: the weight-gradient and 3x3 band kernels walk the frame linearly and read up to a tile (512+16
        # pixels) plus a frame row before/after it; keep those reads inside the (zeroed) allocation
        # (the register-streamed-weights conv reads a 520-pixel band + two skipped halo rows + a frame row past its last tile)
        self.guard = _align(max(4 * (self.wp + 2), 576 + 4 * self.wp) * c * self.es)
        self.base = None     # device address of element (0,-pad,-pad,0)

    def view(self, c_off=0, c=None):
        return View(C.c_void_p(self.base), self.n, self.h, self.w, self.pad, self.c, c_off,
                    self.c - c_off if c is None else c)


class Plan:
    """Workspace layout for one (N, H, W, dtype, train) configuration."""

    @staticmethod
    def env_flags():
        """The environment switches that decide which buffers a plan holds (part of the plan cache key)."""
        return tuple(os.environ.get(k, '1') != '0' for k in ('DBX_LIN_BWD', 'DBX_REFINE_LINEAR', 'DBX_POOL_IDX', 'DBX_HEADS_GEN')) + \
            (os.environ.get('DBX_F32_LIN', '0') == '1',)

    def __init__(self, kind, n, h, w, dtype_id, device, train):
        self.kind, self.n, self.h, self.w, self.dtype_id, self.train = kind, n, h, w, dtype_id, train
        self.flags = Plan.env_flags()
        lin_bwd, lin_refine, use_idx, heads_gen, f32_lin = self.flags
        # fp32 keeps the full-resolution heads backward unless DBX_F32_LIN=1 (parity suite: the 16-bit step's structure -- backward by linearity
        # of the up-sampling, hidden gradient generated in its consumers -- on the exact-fp32 kernels, against the reference-captured gradients)
        lin_bwd = lin_bwd and (dtype_id != _lib.F32 or f32_lin)
        es = _lib.ESIZE[dtype_id]
        self.cin0 = 16 // es                      # conv1_1 input channels padded to one 16-byte chunk
        # refine branch (cat(landmarks, score) -> pool -> 3x3 -> 5x5 -> bilinear -> 1x1; 61 MMAC per patch) runs in the compute type.
        # Round 3 tried it on the exact-fp32 path in eval mode (its inputs ARE the fp32 head outputs): the refined score of
        # DenseBoxLMLOC stayed the worst 16-bit map (3.1e-3 of max|ref| in f16 -- the error is in its inputs, the f16 head outputs, and
        # its convs amplify it) while the three fp32 GEMMs added 0.1 ms to a 0.53 ms 512x512 inference: reverted (rdt == dtype).
        self.rdt = dtype_id
        self.crf = 8 if _lib.ESIZE[self.rdt] == 2 else 32    # refine input channels (5) padded: 16 B chunk, or a 128 B K step in f32
        nh = len(_HEADS[kind])
        h2, w2, h4, w4, h8, w8 = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
        self.h4, self.w4 = h4, w4
        B = {}

        def add(name, hh, ww, c, pad=1, dt=None):
            B[name] = Buf(name, n, hh, ww, c, pad, dtype_id if dt is None else dt)
        add('x0', h, w, self.cin0)
        # conv1_2's full-resolution output is only written when its pooling is NOT fused (fp32 / odd sizes) or when a backward pass
        # will re-read it (training without the arg-max nibbles); otherwise the name aliases a11 (never written: the fused call skips
        # the full map) instead of holding 480 MB at batch 64
        self.a12_alias = es == 2 and h % 2 == 0 and w % 2 == 0 and (not train or use_idx)
        add('a11', h, w, 64)
        if not self.a12_alias:
            add('a12', h, w, 64)
        add('p1', h2, w2, 64)
        add('a21', h2, w2, 128); add('a22', h2, w2, 128); add('p2', h4, w4, 128)
        add('a31', h4, w4, 256); add('a32', h4, w4, 256)
        add('fusion', h4, w4, 768)                # [0:512) = upsampled conv4_4, [512:768) = conv3_4 (concat for free)
        add('p3', h8, w8, 256)
        add('a41', h8, w8, 512); add('a42', h8, w8, 512); add('a43', h8, w8, 512); add('a44', h8, w8, 512)
        add('hid', h4, w4, 512 * nh, pad=0)
        rf_convs = kind != 'DenseBox' and train and not lin_refine      # the three refine convs on the MFMA kernels (A/B, tests)
        if rf_convs:
            # frames chosen so that each conv's (dz, x) pair is congruent for the weight gradient AND dz has the
            # k-1 pixel frame its data gradient needs: rf_p(+1) ~ d_rf_1(+2), rf_1(+2) ~ d_rf_2(+4)
            add('rf_in', h4, w4, self.crf, pad=0, dt=self.rdt)
            add('rf_p', h8, w8, self.crf, pad=1, dt=self.rdt)
            add('rf_1', h8 - 2, w8 - 2, 64, pad=2, dt=self.rdt)
            add('rf_2', h8 - 6, w8 - 6, 64, pad=0, dt=self.rdt)
            add('rf_u', h4, w4, 64, pad=0, dt=self.rdt)
        if train:
            # gradients (dZ = dL/d pre-activation) live in frames congruent to the matching activation
            for nm in ('a11', 'a21', 'a22', 'a31', 'a32', 'a41', 'a42', 'a43', 'a44'):
                add('d_' + nm, B[nm].h, B[nm].w, B[nm].c)
            add('d_a12', h, w, 64)
            add('d_c34', h4, w4, 256)
            if not lin_bwd:
                add('d_ups', h4, w4, 512, pad=0)          # (the heads' backward by linearity never forms it)
            add('d_p1', h2, w2, 64, pad=0); add('d_p2', h4, w4, 128, pad=0); add('d_p3', h8, w8, 256, pad=0)
            add('d_g44', h8, w8, 512 * nh, pad=1)         # up^T(d_hid): the hidden gradient on conv4_4's grid, congruent with 'a44'
            add('d_out', h4, w4, self.crf * nh, pad=0)    # dL/d(head outputs), one crf-channel slot per head
            # The hidden gradient itself (944 MB at batch 64) is not held when its consumers GENERATE it (dbx_heads1_wgrad_gen /
            # dbx_heads1_dgrad_gen / dbx_head2_backward_up without a d_hid: round 4; DBX_HEADS_GEN=0 keeps it in memory).  A backward that
            # cannot use them (an injected dropout mask, a side stream) gets the buffer on demand (Engine._d_hid).
            L = _lib.lib()
            # (the generating kernels' own limits, checked by DBX_REQUIRE at backward time: < 2^24 pixels, d_out below 2 GiB, k <= 8 per head --
            #  a plan outside them keeps the hidden gradient in memory instead of failing in backward_raw)
            gen_fits = n * h4 * w4 < (1 << 24) and n * h4 * w4 * self.crf * nh * es < (1 << 31) and self.crf <= 8
            if dtype_id == _lib.F32:
                # (DBX_F32_LIN=1: the two consumers run their fp32 reference instantiations; dbx_head2_backward_up has no fp32 form that leaves
                #  d_hid out, so the buffer exists -- and is poisoned behind that call, see backward_raw)
                self.heads_gen = bool(heads_gen and lin_bwd)
            else:
                self.heads_gen = bool(heads_gen and lin_bwd and gen_fits and
                                      L.dbx_head2_backward_up_fused(dtype_id, C.byref(B['hid'].view()), C.byref(B['d_g44'].view())) and
                                      L.dbx_heads1_wgrad_gen_ok(dtype_id, C.byref(B['fusion'].view(512, 256)), nh))
            if not self.heads_gen or dtype_id == _lib.F32:
                add('d_hid', h4, w4, 512 * nh, pad=1)     # congruent with 'fusion'
            if rf_convs:
                add('d_rfo', h4, w4, self.crf, pad=0)
                add('d_rf_u', h4, w4, 64, pad=0)
                add('d_rf_2', h8 - 6, w8 - 6, 64, pad=4)
                add('d_rf_1', h8 - 2, w8 - 2, 64, pad=2)
                add('d_rf_p', h8, w8, self.crf, pad=0)
                add('d_rf_in', h4, w4, self.crf, pad=0)
        off = 0
        for b in B.values():
            off += b.guard
            b.off = off
            off += _align(b.bytes) + b.guard
        self.total = off
        if not train:
            self.heads_gen = False
        self.drop_active = self.drop_hash = False
        self.frag = {}             # (stem, 'f' | 'b') -> fragment-order weights? (Engine._frag)
        self.drop_seed = 0
        self.mask_buf = None
        self.ws = torch.zeros(off, dtype=torch.uint8, device=device)
        # training: arg-max nibbles of the three pooling layers (dbx_maxpool2x2_idx layout, half a byte per pooled element): the
        # pooling backward reads them instead of the un-pooled activations, and conv1_2's full-resolution output is never written
        # (its only other reader was pool1's backward).  DBX_POOL_IDX=0 keeps the activation-reading backward (A/B, tests).
        self.pool_idx = None
        if train and os.environ.get('DBX_POOL_IDX', '1') != '0':
            self.pool_idx = {k: torch.empty(n * (hh // 2) * (ww // 2) * (c // 2) + 16, dtype=torch.uint8, device=device)
                             for k, (hh, ww, c) in {'a12': (h, w, 64), 'a22': (h2, w2, 128), 'fusion': (h4, w4, 256)}.items()}
        base = self.ws.data_ptr()
        assert base % 256 == 0
        for b in B.values():
            b.base = base + b.off
        if self.a12_alias:
            B['a12'] = B['a11']
        self.B = B


# parameter -> engine, for optim.SGD.step() (which is handed bare parameters, like torch.optim.SGD): a registry on the side instead of an
# attribute on the Parameter (attributes travel with deepcopy / pickle of the module; a weak reference does not pickle)
_PARAM_ENGINE = {}


def register_params(engine, params):
    if len(_PARAM_ENGINE) > 4096:
        for k in [k for k, (rp, re_) in _PARAM_ENGINE.items() if rp() is None or re_() is None]:
            del _PARAM_ENGINE[k]
    re_ = weakref.ref(engine)
    for p in params:
        ent = _PARAM_ENGINE.get(id(p))
        if ent is None or ent[0]() is not p or ent[1]() is not engine:
            _PARAM_ENGINE[id(p)] = (weakref.ref(p), re_)


def engine_of(params):
    """The one engine all of `params` were registered by (a training-mode forward), or None."""
    eng = None
    for p in params:
        ent = _PARAM_ENGINE.get(id(p))
        if ent is None or ent[0]() is not p:
            return None
        e = ent[1]()
        if e is None or (eng is not None and e is not eng):
            return None
        eng = e
    return eng


class Engine:
    def __init__(self, net):
        self.net = net
        self.kind = net.KIND
        self.L = _lib.lib()
        self.plans = {}
        self.wcache = {}       # (name, mode, dtype) -> (version key, packed tensor)
        self.bias_cache = {}
        self.last_plan = None
        self.grad_sink = None      # optional dist.GradReducer: flat gradient views + readiness callbacks
        self.profile = None        # list -> every conv / wgrad launch appends {kernel, flops, start, end} (HIP events)
        self._defer = None         # not None while the multi-tensor pack table is being recorded
        self._defer_keep = []
        self._table_keys = set()   # cache keys whose buffers the pack table refreshes
        self._tables = {}          # (dtype, train) -> (device job table, count, max_elems)
        self._wsig = None
        self._plans_conv = {}      # problem signature -> (kernel id, name, fragment-order weights?) from dbx_conv_plan
        # lab hook (tools/gpu_layer_error_budget.py): called as act_hook(buffer name, plan) right behind the launch that produced the buffer,
        # on the launch stream -- e.g. to round ONE fp32 activation to 16 bits in place.  None in every product / test path.
        self.act_hook = None

    def grad_order(self):
        """Parameter names in the order backward_raw() finishes their gradients (deepest first)."""
        heads = _HEADS[self.kind]
        names = []
        if self.kind != 'DenseBox':
            for s_ in ('conv6_3_det', 'conv6_2_det', 'conv6_1_det'):
                names += [s_ + '.weight', s_ + '.bias']
        for s_, _ in heads:
            names += ['conv5_2_%s.weight' % s_, 'conv5_2_%s.bias' % s_]
        names += ['conv5_1_%s.weight' % s_ for s_, _ in heads] + ['conv5_1_%s.bias' % s_ for s_, _ in heads]
        for s_, _, _ in reversed(_BACKBONE):
            names += [s_ + '.weight', s_ + '.bias']
        return names

    # ------------------------------------------------------------------ parameters
    def _param(self, name):
        mod, attr = name.rsplit('.', 1)
        return getattr(getattr(self.net, mod), attr)

    def _packed(self, key, builder, versions):
        ent = self.wcache.get(key)
        if ent is not None and (ent[0] == versions or key in self._table_keys):
            return ent[1]       # table-covered buffers were refreshed by _prepare_weights() at the top of forward
        # re-pack into the existing buffer when there is one: its zero padding never changes
        t = builder(ent[1] if ent is not None else None)
        self.wcache[key] = (versions, t)
        return t

    def _pack(self, dt, mode, w, rows_pad, cin_pad, kh, kw, out=None, row_off=0, k_off=0):
        """fp32 OIHW parameter -> packed compute-dtype matrix [rows_pad][ktot].
        (row_off / k_off may be negative: elements that land outside [0, rows_pad) x [0, cin_pad) are skipped -- channel slices)"""
        d = ConvDesc(dt, kh, kw, 0, cin_pad, rows_pad, 0, 0)
        elems = self.L.dbx_conv_packed_elems(C.byref(d))
        if out is None:
            out = torch.zeros(elems * _lib.ESIZE[dt], dtype=torch.uint8, device=w.device)
        co, ci = w.shape[0], w.shape[1]
        if self._defer is not None:       # table build: record the job, the multi-tensor launch does the work
            es = _lib.ESIZE[dt]
            ktot = (kh * kw * cin_pad * es + 127) // 128 * 128 // es
            self._defer.append((w.data_ptr(), out.data_ptr(), co, ci, kh * kw, mode, rows_pad if mode >= 4 else ktot, cin_pad,
                                row_off, k_off, rows_pad))
            self._defer_keep.append(out)
            return out
        check(self.L.dbx_pack_weight(dt, mode, ptr(w.detach()), co, ci, kh, kw, ptr(out), rows_pad, cin_pad,
                                     row_off, k_off, stream_ptr()))
        return out

    def _bias(self, names, pad_to):
        ps = [self._param(n + '.bias') for n in names]
        key = tuple(names)
        ver = tuple((p._version, p.data_ptr()) for p in ps)
        ent = self.bias_cache.get(key)
        if ent is not None and (ent[0] == ver or key in self._table_keys):
            return ent[1]
        b = ent[1] if ent is not None else torch.zeros(pad_to, dtype=torch.float32, device=ps[0].device)
        o = 0
        for p in ps:
            if self._defer is not None:
                self._defer.append((p.data_ptr(), b.data_ptr(), p.numel(), 1, 1, 2, 0, 0, o, 0, 0))
            else:
                b[o:o + p.numel()].copy_(p.detach())
            o += p.numel()
        if self._defer is not None:
            self._defer_keep.append(b)
        self.bias_cache[key] = (ver, b)
        return b

    def _w_fwd(self, dt, stem, cin_pad, cout_pad, frag=False):
        """Packed forward weights; frag: MFMA-fragment order (pack mode 4) for the register-streamed-weights kernel."""
        w = self._param(stem + '.weight')
        mode = 4 if frag else 0
        return self._packed((stem, mode, dt),
                            lambda old: self._pack(dt, mode, w, cout_pad, cin_pad, w.shape[2], w.shape[3], out=old),
                            (w._version, w.data_ptr()))

    def conv_plan(self, dt, x, y, kh, kw, cpad, cin_pad, cout_pad, epi):
        """The library's kernel choice for this problem (dbx_conv_plan): (ConvPlan.kernel, name, wants fragment-order weights)."""
        key = (dt, x.n, x.h, x.w, x.pad, x.ld, x.c_off, x.c, y.pad, y.ld, y.c_off, y.c, kh, kw, cpad, cin_pad, cout_pad,
               epi & ~_lib.CONV_WFRAG)
        ent = self._plans_conv.get(key)
        if ent is None:
            d = ConvDesc(dt, kh, kw, cpad, cin_pad, cout_pad, epi & ~_lib.CONV_WFRAG, 0)
            out = _lib.ConvPlan()
            check(self.L.dbx_conv_plan(C.byref(d), C.byref(x), C.byref(y), C.byref(out)))
            ent = (out.kernel, out.name.decode(), bool(out.w_frag))
            self._plans_conv[key] = ent
        return ent

    def _w_heads1(self, dt, frag=False):
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_1_%s.weight' % s) for s, _ in heads]
        mode = 4 if frag else 0

        def build(out):
            for i, w in enumerate(ws):
                out = self._pack(dt, mode, w, 512 * len(ws), 768, 1, 1, out=out, row_off=512 * i)
            return out
        return self._packed(('heads1', mode, dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _w_heads2(self, dt):
        """Stage-2 head weights as one block-diagonal matrix [64 rows][512 nh]: head i's k_i rows at row sum(k_<i), columns 512 i .."""
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_2_%s.weight' % s) for s, _ in heads]

        def build(out):
            r = 0
            for i, (w, (_, k)) in enumerate(zip(ws, heads)):
                out = self._pack(dt, 0, w, 64, 512 * len(ws), 1, 1, out=out, row_off=r, k_off=512 * i)
                r += k
            return out
        return self._packed(('heads2', 0, dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _w_heads2_frag(self, dt):
        """Stage-2 head weights in MFMA-fragment order for the fused heads forward (dbx_heads_forward_fused): one mode-4 image of 256 rows
        x 512 nh columns, head i's k_i rows at rows 0.. and columns 512 i.. (the kernel only reads the first 32-row block of each 16-channel
        column block)."""
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_2_%s.weight' % s) for s, _ in heads]

        def build(out):
            for i, w in enumerate(ws):
                out = self._pack(dt, 4, w, 256, 512 * len(ws), 1, 1, out=out, row_off=0, k_off=512 * i)
            return out
        return self._packed(('heads2', 4, dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _w_heads2_rows0(self, dt):
        """Stage-2 head weights for the fused heads forward on the 8-phase kernel: one plain (mode 0) image of 64 rows x 512 nh columns, head
        i's k_i rows at rows 0.. and columns 512 i.. -- a lane's MFMA fragment (row = output, eight consecutive hidden channels) is one
        16-byte load."""
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_2_%s.weight' % s) for s, _ in heads]

        def build(out):
            for i, w in enumerate(ws):
                out = self._pack(dt, 0, w, 64, 512 * len(ws), 1, 1, out=out, row_off=0, k_off=512 * i)
            return out
        return self._packed(('heads2', 'rows0', dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _heads_fused(self, P, dt):
        """Both 1x1 convs of the heads in one kernel (dbx_heads_forward_fused)?  16-bit training plans whose 768 -> 512 nh GEMM runs on the
        register-streamed-weights kernel with hash dropout; DBX_HEADS_FUSED=0 keeps the two GEMMs (A/B, tests)."""
        key = ('heads2', 'fused')
        r = P.frag.get(key)
        if r is None:
            heads = _HEADS[self.kind]
            nh = len(heads)
            r = 0           # 0: two GEMMs; 1: fused on the ws kernel (fragment-order weights); 2: fused on the 8-phase kernel (plain weights)
            if P.train and dt != _lib.F32 and os.environ.get('DBX_HEADS_FUSED', '1') != '0':
                d = ConvDesc(dt, 1, 1, 0, 768, 512 * nh, _lib.EPI_BIAS | _lib.EPI_DROPHASH, 0)
                ks = (C.c_int32 * nh)(*[k for _, k in heads])
                r = int(self.L.dbx_heads_forward_fusable(C.byref(d), C.byref(P.B['fusion'].view()), C.byref(P.B['hid'].view()), ks, nh))
            P.frag[key] = r
        return r

    def _frag_heads(self, P, dt, which):
        """Fragment-order weights for the heads' 768 -> 512 nh GEMM ('f') / its split-destination data gradient ('b')?"""
        key = ('heads1', which)
        r = P.frag.get(key)
        if r is None:
            B, nh = P.B, len(_HEADS[self.kind])
            if which == 'f':
                r = self.conv_plan(dt, B['fusion'].view(), B['hid'].view(), 1, 1, 0, 768, 512 * nh, _lib.EPI_BIAS | _lib.EPI_DROPHASH)[2]
            elif which == 'ba' and 'd_g44' in B and dt != _lib.F32:
                r = self.conv_plan(dt, B['d_g44'].view(), B['d_a44'].view(), 1, 1, 0, 512 * nh, 512, _lib.EPI_GATE)[2]
            elif which == 'bc' and getattr(P, 'heads_gen', False):
                r = dt != _lib.F32                         # dbx_heads1_dgrad_gen takes the fragment-order image (its fp32 reference form the plain one)
            elif which == 'bc' and 'd_hid' in B and dt != _lib.F32:
                r = self.conv_plan(dt, B['d_hid'].view(), B['d_c34'].view(), 1, 1, 0, 512 * nh, 256, _lib.EPI_GATE)[2]
            elif which in ('ba', 'bc'):
                r = False
            elif 'd_hid' in B and 'd_ups' in B and dt != _lib.F32:
                dv = B['d_ups'].view()
                both = View(dv.ptr, dv.n, dv.h, dv.w, dv.pad, 768, 0, 768)      # both destinations' couts, for planning only
                r = self.conv_plan(dt, B['d_hid'].view(), both, 1, 1, 0, 512 * nh, 768, 0)[2]
            else:
                r = False
            P.frag[key] = r
        return r

    def _weight_getters(self, dt, train, P):
        """Touch every packed weight / bias the step uses (same calls as forward_raw / backward_raw make)."""
        kind = self.kind
        heads = _HEADS[kind]
        nh = len(heads)
        for stem, cin, cout in _BACKBONE:
            self._w_fwd(dt, stem, P.cin0 if cin == 3 else cin, max(64, cout), frag=self._frag(P, dt, stem, 'f'))
            self._bias([stem], max(64, cout))
        if train:             # (eval runs the folded heads: _w_heads_folded, packed on its own)
            self._w_heads1(dt, frag=self._frag_heads(P, dt, 'f'))
            self._bias(['conv5_1_' + s_ for s_, _ in heads], 512 * nh)
            self._w_heads2(dt)
            if self._heads_fused(P, dt) == 1:
                self._w_heads2_frag(dt)
            elif self._heads_fused(P, dt) == 2:
                self._w_heads2_rows0(dt)
            self._bias(['conv5_2_' + s_ for s_, _ in heads], 64)
        # the refine branch runs from its fp32 parameters (folded 7x7 conv, dbx_refine_backward) unless DBX_REFINE_LINEAR=0 in training
        rf_convs = kind != 'DenseBox' and train and os.environ.get('DBX_REFINE_LINEAR', '1') == '0'
        if rf_convs:
            if P.rdt == dt:       # (the table is one dtype)
                self._w_fwd(dt, 'conv6_1_det', P.crf, 64)
                self._w_fwd(dt, 'conv6_2_det', 64, 64)
                self._w_fwd(dt, 'conv6_3_det', 64, 64)
            self._bias(['conv6_1_det'], 64); self._bias(['conv6_2_det'], 64); self._bias(['conv6_3_det'], 64)
        if train:
            for stem, cin, cout in _BACKBONE[1:]:
                self._w_bwd(dt, stem, max(64, cin), cout, frag=self._frag(P, dt, stem, 'b'))
            if self._lin_bwd(dt):
                self._w_heads1_bwd_part(dt, 'a', frag=self._frag_heads(P, dt, 'ba'))
                self._w_heads1_bwd_part(dt, 'c', frag=self._frag_heads(P, dt, 'bc'))
            else:
                self._w_heads1_bwd(dt, frag=self._frag_heads(P, dt, 'b'))
            if rf_convs:
                self._w_bwd(dt, 'conv6_3_det', 64, P.crf)
                self._w_bwd(dt, 'conv6_2_det', 64, 64)
                self._w_bwd(dt, 'conv6_1_det', 64, 64)

    def _prepare_weights(self, dt, train, P):
        """Re-pack all parameters with ONE kernel launch when any of them changed (e.g. after an optimizer step)."""
        params = [p for _, p in self.net.named_parameters()]
        lay = tuple(self._frag(P, dt, st, wh) for st, _, _ in _BACKBONE for wh in ('f', 'b'))   # layouts the kernels of this plan want
        if train:
            lay += (self._heads_fused(P, dt),)
            lay += (self._frag_heads(P, dt, 'f'),) + ((self._frag_heads(P, dt, 'ba'), self._frag_heads(P, dt, 'bc')) if self._lin_bwd(dt)
                                                      else (self._frag_heads(P, dt, 'b'),))
        sig = (dt, train, tuple((p._version, p.data_ptr()) for p in params), lay)
        if sig == self._wsig:
            return
        tkey = (dt, train, P.cin0, P.crf, lay, tuple(dp for _, dp in sig[2]))   # the table stores EVERY parameter's pointer
        tab = self._tables.get(tkey)
        if tab is None:
            import numpy as np
            self._defer, keys_before = [], set(self.wcache) | set(self.bias_cache)
            self.wcache.clear(); self.bias_cache.clear(); self._table_keys = set()
            self._weight_getters(dt, train, P)
            jobs, self._defer = self._defer, None
            self._table_keys = set(self.wcache) | set(self.bias_cache)
            rec = np.zeros(len(jobs), dtype=[('src', '<u8'), ('dst', '<u8'), ('co', '<i4'), ('ci', '<i4'), ('taps', '<i4'),
                                             ('mode', '<i4'), ('ktot', '<i8'), ('cin_pad', '<i4'), ('row_off', '<i4'),
                                             ('k_off', '<i4'), ('rows_lim', '<i4')])
            for i, j in enumerate(jobs):
                rec[i] = j
            dev = params[0].device
            tab = (torch.from_numpy(rec.view(np.uint8).copy()).to(dev), len(jobs), max(j[2] * j[3] * j[4] for j in jobs), jobs)
            self._tables = {tkey: tab}
            self._sgd_tables = {}
        check(self.L.dbx_pack_multi(dt, ptr(tab[0]), tab[1], tab[2], stream_ptr()))
        self._wsig = sig
        if train:
            # optim.SGD.step() finds the engine through its parameters and folds the NEXT re-packing into the update (sgd_pack_step)
            self._last_prep = (dt, tkey, lay)
            register_params(self, params)

    def sgd_pack_step(self, live, ptrs, lr, momentum, weight_decay, first, guard=None, step_id=0):
        """optim.SGD.step() with this engine's re-packing folded in (dbx_sgd_pack_step): one job per parameter updates it and emits the packed
        images the last training plan uses, so the next forward finds its weights packed and launches nothing.  `live`: the parameters with
        gradients, in the order of `ptrs` ([param, grad, momentum] pointers on the device).  Returns False when the plain two-launch path
        has to run (no training plan yet, parameters moved or changed behind the engine's back, DBX_SGD_PACK=0)."""
        prep = getattr(self, '_last_prep', None)
        if prep is None or self._wsig is None or os.environ.get('DBX_SGD_PACK', '1') == '0':
            return False
        dt, tkey, lay = prep
        tab = self._tables.get(tkey)
        params = [p for _, p in self.net.named_parameters()]
        if tab is None or self._wsig[0] != dt or not self._wsig[1] or self._wsig[3] != lay:
            return False
        if tuple((p._version, p.data_ptr()) for p in params) != self._wsig[2]:
            return False                                  # a parameter changed since the packed copies were made: the regular path re-packs
        mine = {id(p) for p in params}
        if any(id(p) not in mine for p in live):
            return False                                  # the optimizer also holds parameters of something else
        pidx = {p.data_ptr(): i for i, p in enumerate(live)}
        skey = (tkey, tuple(pidx))
        st = self._sgd_tables.get(skey)
        if st is False:
            return False                                  # (this configuration does not fit the fused kernel: decided once, not per step)
        if st is None:
            import numpy as np
            by_src, order = {}, []
            for j in tab[3]:                              # (src, dst, co, ci, taps, mode, ktot, cin_pad, row_off, k_off, rows_lim)
                if j[0] not in by_src:
                    by_src[j[0]] = []
                    order.append(j[0])
                by_src[j[0]].append(j)
            es = _lib.ESIZE[dt]
            recs = []
            for src in order:
                js = by_src[src]
                if src not in pidx:
                    continue                              # no gradient: unchanged, its packed copies stay valid
                if len(js) > 4 or len({(j[2], j[3], j[4]) for j in js}) != 1:
                    self._sgd_tables[skey] = False
                    return False
                co, ci, taps = js[0][2], js[0][3], js[0][4]
                tiled = es == 2 and taps <= 25 and all(
                    j[5] != 2 and (ci if j[5] in (0, 4) else co) % 8 == 0 and j[9] % 8 == 0 and j[7] % 8 == 0 for j in js)
                recs.append((src, pidx[src], co, ci, taps, len(js), int(tiled), js))
            packed = set(order)
            for p in live:                                # parameters nobody packs (the folded refine convs): plain update
                if p.data_ptr() not in packed:
                    recs.append((p.data_ptr(), pidx[p.data_ptr()], p.numel(), 1, 1, 0, 0, []))
            dst_t = [('dst', '<u8'), ('ktot', '<i8'), ('mode', '<i4'), ('cin_pad', '<i4'), ('row_off', '<i4'), ('k_off', '<i4'),
                     ('rows_lim', '<i4'), ('pad', '<i4')]
            rec = np.zeros(len(recs), dtype=[('p', '<u8'), ('pidx', '<i4'), ('co', '<i4'), ('ci', '<i4'), ('taps', '<i4'), ('ndst', '<i4'),
                                             ('tiled', '<i4'), ('d', dst_t, (4,))])
            assert rec.dtype.itemsize == 192
            for i, (src, pi, co, ci, taps, nd, tiled, js) in enumerate(recs):
                rec[i]['p'], rec[i]['pidx'], rec[i]['co'], rec[i]['ci'], rec[i]['taps'] = src, pi, co, ci, taps
                rec[i]['ndst'], rec[i]['tiled'] = nd, tiled
                for k, j in enumerate(js):
                    rec[i]['d'][k] = (j[1], j[6], j[5], j[7], j[8], j[9], j[10], 0)
            st = (torch.from_numpy(rec.view(np.uint8).copy()).to(params[0].device), len(recs), max(r[2] * r[3] * r[4] for r in recs))
            if len(self._sgd_tables) >= 8:                  # (a caller cycling through many live-parameter sets: keep the cache small)
                self._sgd_tables.pop(next(iter(self._sgd_tables)))
            self._sgd_tables[skey] = st
        # (guard: optim.SGD's overflow guard -- a step whose gradients were not finite changes nothing, see dbx_grad_guard; the version counters
        #  move either way: the host does not know, and the packed images are current in both cases)
        check(self.L.dbx_sgd_pack_step_guarded(dt, ptr(st[0]), st[1], st[2], ptr(ptrs), lr, momentum, weight_decay, 1 if first else 0,
                                               ptr(guard), step_id, stream_ptr()))
        torch.autograd.graph.increment_version(live)
        self._wsig = (dt, True, tuple((p._version, p.data_ptr()) for p in params), lay)
        return True

    def _w_heads_folded(self, dt):
        """Eval mode: each head's two 1x1 convs folded into one 768->k map; all heads stacked into ONE [ktot x 768] GEMM."""
        heads = _HEADS[self.kind]
        names = []
        for s_, _ in heads:
            names += ['conv5_1_%s.weight' % s_, 'conv5_1_%s.bias' % s_, 'conv5_2_%s.weight' % s_, 'conv5_2_%s.bias' % s_]
        ps = [self._param(n) for n in names]
        ver = tuple((p._version, p.data_ptr()) for p in ps)
        ent = self.wcache.get(('folded', dt))
        if ent is not None and ent[0] == ver:
            return ent[1]
        dev = ps[0].device
        ktot = sum(k for _, k in heads)
        wf = torch.empty((ktot, 768, 1, 1), dtype=torch.float32, device=dev)
        bf = torch.zeros(64, dtype=torch.float32, device=dev)
        o = 0
        for i, (s_, k) in enumerate(heads):
            w1, b1, w2, b2 = ps[4 * i:4 * i + 4]
            check(self.L.dbx_fold_heads(ptr(w2.detach()), ptr(b2.detach()), ptr(w1.detach()), ptr(b1.detach()), k,
                                        C.c_void_p(wf.data_ptr() + o * 768 * 4), C.c_void_p(bf.data_ptr() + o * 4),
                                        stream_ptr()))
            o += k
        saved, self._defer = self._defer, None          # pack immediately (not part of the multi-tensor table)
        wp = self._pack(dt, 0, wf, 64, 768, 1, 1)
        # the two halves of the concat separately (eval by linearity: the conv4_4 part runs on conv4_4's own grid)
        wpa = self._pack(dt, 0, wf[:, :512].contiguous(), 64, 512, 1, 1)
        wpc = self._pack(dt, 0, wf[:, 512:].contiguous(), 64, 256, 1, 1)
        self._defer = saved
        self.wcache[('folded', dt)] = (ver, (wp, bf, ktot, wpa, wpc))
        return wp, bf, ktot, wpa, wpc

    def _w_refine_folded(self):
        """Eval mode: conv6_1 (3x3) -> conv6_2 (5x5) -> up-sampling -> conv6_3 (1x1) as one un-padded 7x7 conv 5 -> 1 (dbx_fold_refine):
        (w [1][5][7][7], b [1]) in fp32, cached on the parameter versions."""
        names = ['conv6_1_det.weight', 'conv6_1_det.bias', 'conv6_2_det.weight', 'conv6_2_det.bias', 'conv6_3_det.weight', 'conv6_3_det.bias']
        ps = [self._param(nm) for nm in names]
        ver = tuple((p._version, p.data_ptr()) for p in ps)
        ent = self.wcache.get(('refold',))
        if ent is not None and ent[0] == ver:
            return ent[1]
        dev = ps[0].device
        ci, cm = ps[0].shape[1], ps[0].shape[0]
        assert ci == 5 and cm <= 64 and tuple(ps[0].shape[2:]) == (3, 3) and tuple(ps[2].shape[2:]) == (5, 5)
        if ent is not None and ent[1][0].device == dev:          # training re-folds after every optimizer step: same three buffers
            wf, bf, vf = ent[1]
        else:
            wf = torch.empty((1, ci, 7, 7), dtype=torch.float32, device=dev)
            bf = torch.empty(1, dtype=torch.float32, device=dev)
            vf = torch.empty((cm, 5, 5), dtype=torch.float32, device=dev)
        check(self.L.dbx_fold_refine(*[ptr(p.detach().float().contiguous()) for p in ps], ci, cm, ptr(wf), ptr(bf), ptr(vf), stream_ptr()))
        self.wcache[('refold',)] = (ver, (wf, bf, vf))
        return wf, bf, vf

    # ------------------------------------------------------------------ plumbing
    def _frag(self, P, dt, stem, which):
        """Does the library run this backbone layer's forward ('f') / data-gradient ('b') conv with fragment-order weights
        (the register-streamed-weights kernel) under plan P?  Decided by dbx_conv_plan on the very views the call uses."""
        key = (stem, which)
        r = P.frag.get(key)
        if r is None:
            B = P.B
            cin, cout = _CH[stem]

            def vw(name):
                return B['fusion'].view(512, 256) if name == 'c34' else B[name].view()
            if which == 'f':
                src, dst = _FWD_IO[stem]
                cin_pad = P.cin0 if cin == 3 else cin
                r = self.conv_plan(dt, vw(src), vw(dst), 3, 3, 1, cin_pad, max(64, cout), _lib.EPI_BIAS | _lib.EPI_RELU)[2]
            else:
                dz, dx = _BWD_IO.get(stem, (None, None))           # conv1_1 has no data gradient
                # the epilogue of the real call: ReLU-gated, except the data gradients that feed a pooling layer (conv2_1, conv3_1, conv4_1)
                gated = stem not in ('conv2_1_1', 'conv3_1_1', 'conv4_1_1')
                r = self.conv_plan(dt, vw(dz), vw(dx), 3, 3, 1, cout, max(64, cin), _lib.EPI_GATE if gated else 0)[2] if dz in B else False
            P.frag[key] = r
        return r

    def _next_drop_seed(self):
        """Per-step seed of the counter-based dropout hash.  The stream starts from torch.initial_seed() (torch.manual_seed
        reproduces a run; reference: nn.Dropout draws from torch's generator, DenseBox.py:160) mixed with `sample_offset`
        (the index of this rank's first patch in the global batch, set by dist.DataParallel), so ranks draw different masks."""
        st = getattr(self, '_seed_state', None)
        if st is None or st[0] != torch.initial_seed():
            base = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x5eed) & 0xffffffffffffffff
            st = [torch.initial_seed(), (base ^ (base >> 32)) & 0xffffffff]
            self._seed_state = st
        st[1] = (st[1] * 1664525 + 1013904223) & 0xffffffff
        x = (st[1] ^ (int(getattr(self, 'sample_offset', 0)) * 0x85EBCA6B)) & 0xffffffff
        return x

    def captured_refs(self):
        """Strong references to every device buffer a just-captured hipGraph of this engine's launches points into (workspace
        plan, packed / folded weights, biases, job tables).  decode.detect() stores them next to the graph: the engine keeps
        ONE plan and rebuilds its weight caches on dtype / mode changes, which would otherwise free memory a replay reads."""
        return (list(self.plans.values()), getattr(self, 'last_plan', None), dict(self.wcache), dict(self.bias_cache),
                dict(self._tables))

    def plan(self, n, h, w, dt, device, train):
        key = (n, h, w, dt, train, Plan.env_flags())
        p = self.plans.get(key)
        if p is None:
            p = Plan(self.kind, n, h, w, dt, device, train)
            self.plans = {key: p}      # keep one plan alive (shape changes re-plan)
        return p

    def _conv(self, dt, x, y, wpk, bias, kh, kw, cpad, cin_pad, cout_pad, epi, gate=None, dropmask=None, dm_ld=0,
              alg_ci=None, drop_seed=0):
        d = ConvDesc(dt, kh, kw, cpad, cin_pad, cout_pad, epi, drop_seed)
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        check(self.L.dbx_conv_forward(C.byref(d), C.byref(x), ptr(wpk), ptr(bias), C.byref(y),
                                      C.byref(gate) if gate is not None else None,
                                      C.c_void_p(dropmask) if dropmask else None, dm_ld, stream_ptr()))
        if prof is not None:
            ev1.record()
            # the kernel the library selected (dbx_conv_plan), algorithmic FLOP = 2 * pixels * taps * Cin * Cout (real)
            name = self.conv_plan(dt, x, y, kh, kw, cpad, cin_pad, cout_pad, epi)[1]
            ci = alg_ci if alg_ci is not None else cin_pad
            prof.append({'kernel': name, 'flops': 2.0 * y.n * y.h * y.w * kh * kw * ci * y.c, 'start': ev0, 'end': ev1})

    # ------------------------------------------------------------------ forward
    def forward(self, X):
        net = self.net
        train = net.training and torch.is_grad_enabled()
        params = [p for _, p in net.named_parameters()]
        if train and any(p.requires_grad for p in params):
            from .autograd import NetFunction
            outs = NetFunction.apply(self, X, *params)
            names = self.output_names()
            return dict(zip(names, outs))
        return self.forward_raw(X, train=net.training)

    def output_names(self):
        return [s for s, _ in _HEADS[self.kind]] + ([] if self.kind == 'DenseBox' else ['refine'])

    def forward_raw(self, X, train):
        """Runs the network; returns dict name -> fp32 NCHW tensor.  Activations stay in the plan's workspace."""
        u8 = X.dtype == torch.uint8
        if u8:
            assert X.dim() == 4 and X.size(3) == 3, 'uint8 input must be [N,H,W,3] (HWC, RGB)'
            n, h, w, _ = X.shape
        else:
            assert X.dim() == 4 and X.size(1) == 3, 'input must be [N,3,H,W]'
            n, _, h, w = X.shape
        L, kind = self.L, self.kind
        dt = _lib.DTYPE_ID[self.net.resolved_dtype(train)]
        assert h >= 8 and w >= 8, 'input smaller than the /8 stride'
        if kind != 'DenseBox':
            assert h // 8 >= 7 and w // 8 >= 7, 'refine branch (3x3 + 5x5 un-padded convs) needs H/8, W/8 >= 7'
        dev = X.device
        P = self.plan(n, h, w, dt, dev, train)
        self.last_plan = P
        B = P.B
        s = stream_ptr()
        self._prepare_weights(dt, train, P)
        if u8:
            # raw patches: ToTensor + ImageNet Normalize (DenseBox.py:766-772) fused into the layout kernel
            from .data import IMAGENET_MEAN, IMAGENET_STD
            Xc = X.detach().contiguous()
            check(L.dbx_u8hwc_to_framed(dt, ptr(Xc), C.byref(B['x0'].view()), (C.c_float * 3)(*IMAGENET_MEAN),
                                        (C.c_float * 3)(*IMAGENET_STD), s))
        else:
            Xf = X.detach().to(torch.float32).contiguous()
            check(L.dbx_nchw_to_framed(dt, ptr(Xf), 3, C.byref(B['x0'].view()), s))
        RELU = _lib.EPI_BIAS | _lib.EPI_RELU

        def hook(*names):
            if self.act_hook is not None:
                for nm in names:
                    self.act_hook(nm, P)
        hook('x0')

        def conv3(stem, src, dst, cin, cout, dst_view=None):
            cin_pad = P.cin0 if cin == 3 else cin
            frag = self._frag(P, dt, stem, 'f')
            wp = self._w_fwd(dt, stem, cin_pad, max(64, cout), frag=frag)
            self._conv(dt, B[src].view(), dst_view if dst_view is not None else B[dst].view(), wp,
                       self._bias([stem], max(64, cout)), 3, 3, 1, cin_pad, max(64, cout),
                       RELU | (_lib.CONV_WFRAG if frag else 0), alg_ci=cin)

        PI = P.pool_idx

        def pool(xv, yv, key):
            if PI is not None:
                check(L.dbx_maxpool2x2_idx(dt, C.byref(xv), C.byref(yv), ptr(PI[key]), s))
            else:
                check(L.dbx_maxpool2x2(dt, C.byref(xv), C.byref(yv), s))

        conv3('conv1_1_1', 'x0', 'a11', 3, 64)
        hook('a11')
        d12 = ConvDesc(dt, 3, 3, 1, 64, 64, RELU, 0)
        a11v, a12v, p1v = B['a11'].view(), B['a12'].view(), B['p1'].view()
        if L.dbx_conv_pool_fusable(C.byref(d12), C.byref(a11v), C.byref(a12v)):
            # conv1_2 + pool1 in one kernel; the full-resolution map is only kept when a backward pass will read it
            prof = self.profile
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            # training without the nibbles (DBX_POOL_IDX=0): pool1's backward re-reads the full map, so it is written
            check(L.dbx_conv_forward_pool_idx(C.byref(d12), C.byref(a11v), ptr(self._w_fwd(dt, 'conv1_2_1', 64, 64, frag=False)),
                                              ptr(self._bias(['conv1_2_1'], 64)), C.byref(a12v), C.byref(p1v),
                                              1 if (train and PI is None) else 0, ptr(PI['a12']) if PI else None, s))
            if prof is not None:
                ev1.record()
                prof.append({'kernel': self.conv_plan(dt, a11v, a12v, 3, 3, 1, 64, 64, RELU)[1],
                             'flops': 2.0 * a12v.n * a12v.h * a12v.w * 9 * 64 * 64, 'start': ev0, 'end': ev1})
        else:
            if P.a12_alias:                                  # (the plan's alias rule and the library's fusability rule must agree: never an assert)
                raise RuntimeError('conv1_2 + pool1 not fusable on a plan without a full-resolution conv1_2 map')
            conv3('conv1_2_1', 'a11', 'a12', 64, 64)
            pool(a12v, p1v, 'a12')
        hook('p1')              # (rounding commutes with the max: the pooled map stands for conv1_2's output)
        def conv3_pool(stem, src, yv, pv, c, key, keep_full):
            """3x3 conv + bias + ReLU with the following MaxPool2d(2, 2) in the 8-phase kernels' epilogue when the library takes it (16-bit,
            even H, W): the pooling kernel's re-read of the map goes away, and so does the map itself when nothing else reads it (a22:
            its other reader was pool2's backward, which reads the nibbles)."""
            d = ConvDesc(dt, 3, 3, 1, c, c, RELU, 0)
            xv = B[src].view()
            if not L.dbx_conv_pool_fusable(C.byref(d), C.byref(xv), C.byref(yv)):
                conv3(stem, src, None, c, c, dst_view=yv)
                pool(yv, pv, key)
                return
            prof = self.profile
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            check(L.dbx_conv_forward_pool_idx(C.byref(d), C.byref(xv), ptr(self._w_fwd(dt, stem, c, c, frag=False)), ptr(self._bias([stem], c)),
                                              C.byref(yv), C.byref(pv), 1 if keep_full else 0, ptr(PI[key]) if PI else None, s))
            if prof is not None:
                ev1.record()
                prof.append({'kernel': self.conv_plan(dt, xv, yv, 3, 3, 1, c, c, RELU)[1],
                             'flops': 2.0 * yv.n * yv.h * yv.w * 9 * c * c, 'start': ev0, 'end': ev1})

        conv3('conv2_1_1', 'p1', 'a21', 64, 128)
        hook('a21')
        # the full conv2_2 map is read by pool2's backward only when the nibbles are off (DBX_POOL_IDX=0)
        conv3_pool('conv2_2_1', 'a21', B['a22'].view(), B['p2'].view(), 128, 'a22', train and PI is None)
        hook('p2')
        conv3('conv3_1_1', 'p2', 'a31', 128, 256)
        hook('a31')
        conv3('conv3_2_1', 'a31', 'a32', 256, 256)
        hook('a32')
        c34 = B['fusion'].view(512, 256)
        conv3_pool('conv3_4_1', 'a32', c34, B['p3'].view(), 256, 'fusion', True)       # writes fusion[:, 512:768] and pool3
        hook('c34', 'p3')
        conv3('conv4_1_1', 'p3', 'a41', 256, 512)
        hook('a41')
        conv3('conv4_2_1', 'a41', 'a42', 512, 512)
        hook('a42')
        conv3('conv4_3_1', 'a42', 'a43', 512, 512)
        hook('a43')
        conv3('conv4_4_1', 'a43', 'a44', 512, 512)
        hook('a44')
        # eval: the folded heads are linear in fusion = [up(a44); c34] and have only sum(k) outputs, so W [up(a44); c34] = up(W_a a44) + W_c c34:
        # the conv4_4 part runs on conv4_4's own grid and its 17 fp32 planes are up-sampled instead of 512 activation channels
        # (DBX_EVAL_LINEAR=0: up-sample the activations and run one 768-channel GEMM, as training must)
        lin_eval = not train and os.environ.get('DBX_EVAL_LINEAR', '1') != '0'
        if not lin_eval:
            check(L.dbx_upsample_bilinear(dt, C.byref(B['a44'].view()), C.byref(B['fusion'].view(0, 512)), s))
            hook('ups')

        heads = _HEADS[kind]
        nh = len(heads)
        outs = {}
        h4, w4 = P.h4, P.w4
        if not train:
            # eval: Dropout is the identity and the heads have no non-linearity -> one folded 768 -> sum(k) GEMM
            wp, bf, ktot, wpa, wpc = self._w_heads_folded(dt)
            big = torch.empty((n, ktot, h4, w4), dtype=torch.float32, device=dev)
            yv = View(C.c_void_p(big.data_ptr()), n, h4, w4, 0, ktot, 0, ktot)
            if lin_eval:
                h8, w8 = B['a44'].h, B['a44'].w
                ga = torch.empty((n, ktot, h8, w8), dtype=torch.float32, device=dev)
                gv = View(C.c_void_p(ga.data_ptr()), n, h8, w8, 0, ktot, 0, ktot)
                self._conv(dt, B['a44'].view(), gv, wpa, None, 1, 1, 0, 512, 64, _lib.EPI_F32_NCHW)
                check(L.dbx_upsample_bilinear_nchw_f32(ptr(ga), n * ktot, h8, w8, ptr(big), h4, w4, s))
                self._conv(dt, B['fusion'].view(512, 256), yv, wpc, bf, 1, 1, 0, 256, 64, _lib.EPI_BIAS | _lib.EPI_F32_NCHW | _lib.EPI_ACCUM)
            else:
                self._conv(dt, B['fusion'].view(), yv, wp, bf, 1, 1, 0, 768, 64, _lib.EPI_BIAS | _lib.EPI_F32_NCHW)
            o = 0
            for stem, k in heads:
                outs[stem] = big[:, o:o + k].contiguous()
                o += k
        else:
            # train: one GEMM 768 -> 512*nh over the shared fusion tensor (dropout in its epilogue), then 512 -> k per head
            epi = _lib.EPI_BIAS
            dm = None
            P.drop_active = bool(self._dropout_p() > 0.0)
            P.drop_hash = P.drop_active and self.net.dropout_masks is None
            if P.drop_hash:
                # keep bits come from a counter-based hash of (seed, pixel, channel): nothing to store or re-read
                P.drop_seed = self._next_drop_seed()
                epi |= _lib.EPI_DROPHASH
            elif P.drop_active:
                dm = self._fill_dropout(P, heads)       # injected masks (parity tests)
                epi |= _lib.EPI_DROPMASK
            hfrag = (epi & _lib.EPI_DROPMASK) == 0 and self._frag_heads(P, dt, 'f')
            ktot = sum(k for _, k in heads)
            big = None
            b1 = self._bias(['conv5_1_' + s_ for s_, _ in heads], 512 * nh)
            b2 = self._bias(['conv5_2_' + s_ for s_, _ in heads], 64)
            fk = self._heads_fused(P, dt) if (P.drop_hash and not (epi & _lib.EPI_DROPMASK)) else 0
            if fk == 1 and not hfrag:
                fk = 0
            if fk:
                # both 1x1 convs of every head in one pass: the second (512 -> k) runs on the hidden tile while it is in registers;
                # the 944 MB hidden map is written for the backward pass but not read back (dbx_heads_forward_fused)
                need = L.dbx_heads_forward_fused_scratch_bytes(nh, n * h4 * w4)
                if getattr(self, '_hf_scratch', None) is None or self._hf_scratch.numel() < need:
                    self._hf_scratch = torch.empty(int(need), dtype=torch.uint8, device=dev)
                d = ConvDesc(dt, 1, 1, 0, 768, 512 * nh, epi | (_lib.CONV_WFRAG if fk == 1 else 0), P.drop_seed)
                prof = self.profile
                if prof is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                # every head's own contiguous [N][k][H][W] tensor, as the reference returns them (no slice copies)
                for stem, k in heads:
                    outs[stem] = torch.empty((n, k, h4, w4), dtype=torch.float32, device=dev)
                check(L.dbx_heads_forward_fused_heads(C.byref(d), C.byref(B['fusion'].view()), ptr(self._w_heads1(dt, frag=fk == 1)), ptr(b1),
                                                      C.byref(B['hid'].view()), ptr(self._w_heads2_frag(dt) if fk == 1 else self._w_heads2_rows0(dt)), ptr(b2),
                                                      (C.c_int32 * nh)(*[k for _, k in heads]), nh,
                                                      (C.c_void_p * nh)(*[outs[stem].data_ptr() for stem, _ in heads]), ptr(self._hf_scratch), s))
                if prof is not None:
                    ev1.record()
                    prof.append({'kernel': ('conv3x3_ws_kernel<%s,1,1,2>' if fk == 1 else 'conv3x3_p8_kernel<%s,1,1>') % ('f16', 'bf16', 'f32')[dt],
                                 'flops': 2.0 * n * h4 * w4 * (768 * 512 * nh + 512 * ktot), 'start': ev0, 'end': ev1})
            else:
                self._conv(dt, B['fusion'].view(), B['hid'].view(), self._w_heads1(dt, frag=hfrag), b1, 1, 1, 0, 768, 512 * nh,
                           epi | (_lib.CONV_WFRAG if hfrag else 0),
                           dropmask=dm, dm_ld=512 * nh, drop_seed=P.drop_seed if P.drop_hash else 0)
                # stage 2: the nh Conv1x1(512 -> k) as ONE block-diagonal GEMM 512 nh -> sum(k) over the full hidden rows (4 KiB contiguous
                # per pixel instead of four strided 1-KiB slices in four launches); the heads' outputs are channel ranges of one tensor
                big = torch.empty((n, ktot, h4, w4), dtype=torch.float32, device=dev)
                yv = View(C.c_void_p(big.data_ptr()), n, h4, w4, 0, ktot, 0, ktot)
                self._conv(dt, B['hid'].view(), yv, self._w_heads2(dt), b2, 1, 1, 0,
                           512 * nh, 64, _lib.EPI_BIAS | _lib.EPI_F32_NCHW, alg_ci=512)
            o = 0
            for stem, k in heads:
                if big is not None:
                    outs[stem] = big[:, o:o + k].contiguous()
                o += k
        lin_rf = os.environ.get('DBX_REFINE_LINEAR', '1') != '0'     # training: the branch by its linear structure (0: the three convs; A/B, tests)
        P.refine_fwd = None
        if kind != 'DenseBox' and (not train or lin_rf):
            # refine branch (DenseBox.py:464-471): nothing after the pooling is non-linear, the 1x1 conv commutes with the up-sampling
            # -> cat + pool + ONE un-padded 7x7 conv 5 -> 1 in a single fp32 kernel on the heads' fp32 outputs, then the up-sampling of
            # that one map (was: two layout kernels, pool, three convs, a 64-channel up-sampling: 60 us of a 0.46-ms image).  Training
            # too: its backward needs none of the 64-channel intermediates either (csrc/refine_ops.hip)
            wf7, bf7, vf7 = self._w_refine_folded()
            lmk, det = outs['landmark'], outs['det']
            if not lmk.is_contiguous():
                lmk = lmk.contiguous()
            if not det.is_contiguous():
                det = det.contiguous()
            small = torch.empty((n, 1, h4 // 2 - 6, w4 // 2 - 6), dtype=torch.float32, device=dev)
            check(L.dbx_refine_eval(ptr(lmk), ptr(det), n, h4, w4, ptr(wf7), ptr(bf7), ptr(small), s))
            o = torch.empty((n, 1, h4, w4), dtype=torch.float32, device=dev)
            check(L.dbx_upsample_bilinear_nchw_f32(ptr(small), n, h4 // 2 - 6, w4 // 2 - 6, ptr(o), h4, w4, s))
            outs['refine'] = o
            if train:
                P.refine_fwd = (lmk, det, wf7, vf7)  # the backward pass re-derives everything else from these (dbx_refine_backward)
        elif kind != 'DenseBox':
            # refine branch: cat(landmarks, score) -> pool4 -> 3x3 -> 5x5 -> bilinear -> 1x1   (DenseBox.py:464-471)
            rdt = P.rdt
            saved, self._defer = self._defer, None            # (fp32 refine weights: packed on their own, cached on the parameter versions)
            w61, w62, w63 = (self._w_fwd(rdt, 'conv6_1_det', P.crf, 64), self._w_fwd(rdt, 'conv6_2_det', 64, 64),
                             self._w_fwd(rdt, 'conv6_3_det', 64, 64))
            self._defer = saved
            rin = B['rf_in'].view()
            check(L.dbx_nchw_to_framed_ch(rdt, ptr(outs['landmark']), 4, C.byref(rin), 0, s))
            check(L.dbx_nchw_to_framed_ch(rdt, ptr(outs['det']), 1, C.byref(rin), 4, s))
            check(L.dbx_maxpool2x2(rdt, C.byref(rin), C.byref(B['rf_p'].view()), s))
            self._conv(rdt, B['rf_p'].view(), B['rf_1'].view(), w61,
                       self._bias(['conv6_1_det'], 64), 3, 3, 0, P.crf, 64, _lib.EPI_BIAS, alg_ci=5)
            self._conv(rdt, B['rf_1'].view(), B['rf_2'].view(), w62,
                       self._bias(['conv6_2_det'], 64), 5, 5, 0, 64, 64, _lib.EPI_BIAS)
            check(L.dbx_upsample_bilinear(rdt, C.byref(B['rf_2'].view()), C.byref(B['rf_u'].view()), s))
            o = torch.empty((n, 1, h4, w4), dtype=torch.float32, device=dev)
            yv = View(C.c_void_p(o.data_ptr()), n, h4, w4, 0, 1, 0, 1)
            self._conv(rdt, B['rf_u'].view(), yv, w63,
                       self._bias(['conv6_3_det'], 64), 1, 1, 0, 64, 64, _lib.EPI_BIAS | _lib.EPI_F32_NCHW)
            outs['refine'] = o
        return outs

    def _dropout_p(self):
        """nn.Dropout() of the heads (DenseBox.py:160): the reference default p=0.5 or 0 (disabled)."""
        drops = [m for m in self.net.modules() if isinstance(m, torch.nn.Dropout)]
        p = drops[0].p if drops else 0.0
        if p not in (0.0, 0.5):
            raise RuntimeError('densebox_amd: the head Dropout supports p=0.5 (reference) or p=0, got %r' % p)
        return p

    def _fill_dropout(self, P, heads):
        """Injected keep-masks {head: [N,512,h,w]} (NCHW like the reference's Dropout input) -> uint8 [M][512*nh] buffer."""
        nh = len(heads)
        inj = self.net.dropout_masks
        dev = P.ws.device
        m = torch.empty((P.n, P.h4, P.w4, nh, 512), dtype=torch.uint8, device=dev)
        for i, (stem, _) in enumerate(heads):
            m[:, :, :, i, :] = inj[stem].to(dev).permute(0, 2, 3, 1).to(torch.uint8)
        P.mask_buf = m                              # kept alive until the backward pass has used it
        return m.data_ptr()

    # debugging / tests: read an activation back as fp32 NCHW
    def read_activation(self, name, c_off=0, c=None):
        P = self.last_plan
        b = P.B[name]
        v = b.view(c_off, c)
        out = torch.empty((b.n, v.c, b.h, b.w), dtype=torch.float32, device=P.ws.device)
        check(self.L.dbx_framed_to_nchw_f32(P.dtype_id, C.byref(v), ptr(out), stream_ptr()))
        return out

    # ------------------------------------------------------------------ backward
    def _w_bwd(self, dt, stem, rows_pad, cin_pad, frag=False):
        """dgrad weights: rows = input channels, K = [flipped tap][output channel]."""
        w = self._param(stem + '.weight')
        mode = 5 if frag else 1
        return self._packed((stem, mode, dt),
                            lambda old: self._pack(dt, mode, w, rows_pad, cin_pad, w.shape[2], w.shape[3], out=old),
                            (w._version, w.data_ptr()))

    def _w_heads1_bwd(self, dt, frag=False):
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_1_%s.weight' % s) for s, _ in heads]
        mode = 5 if frag else 1

        def build(out):
            for i, w in enumerate(ws):
                out = self._pack(dt, mode, w, 768, 512 * len(ws), 1, 1, out=out, k_off=512 * i)
            return out
        return self._packed(('heads1', mode, dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _w_heads1_bwd_part(self, dt, part, frag=False):
        """dgrad weights of one part of the fusion concat: rows = that part's input channels (part 'a': 0..511, the up-sampled
        conv4_4; 'c': 512..767, conv3_4), K = the 512 nh hidden channels of all heads."""
        heads = _HEADS[self.kind]
        ws = [self._param('conv5_1_%s.weight' % s) for s, _ in heads]
        mode = 5 if frag else 1
        rows, roff = (512, 0) if part == 'a' else (256, -512)

        def build(out):
            for i, w in enumerate(ws):
                out = self._pack(dt, mode, w, rows, 512 * len(ws), 1, 1, out=out, row_off=roff, k_off=512 * i)
            return out
        return self._packed(('heads1_' + part, mode, dt), build, tuple((w._version, w.data_ptr()) for w in ws))

    def _lin_bwd(self, dt):
        """Heads backward by linearity of the bilinear up-sampling (16-bit types): W [up(a44); c34] = up(W_a a44) + W_c c34, so the
        conv4_4 part of the 768 -> 512 nh GEMM's data and weight gradients runs on conv4_4's 30x30 grid (a quarter of the pixels)
        after ONE transposed up-sampling of the hidden gradient.  DBX_LIN_BWD=0 keeps the full-resolution GEMMs."""
        return (dt != _lib.F32 or os.environ.get('DBX_F32_LIN', '0') == '1') and os.environ.get('DBX_LIN_BWD', '1') != '0'

    def _d_hid(self, P):
        """The hidden-gradient buffer; plans whose heads backward generates it hold none: made on first use (injected masks, side streams)."""
        b = P.B.get('d_hid')
        if b is None:
            hb = P.B['hid']
            b = Buf('d_hid', hb.n, hb.h, hb.w, hb.c, 1, hb.dtype_id)
            P._d_hid_store = torch.zeros(2 * b.guard + _align(b.bytes) + 256, dtype=torch.uint8, device=P.ws.device)
            b.base = _align(P._d_hid_store.data_ptr() + b.guard)
            P.B['d_hid'] = b
        return b

    def _wgrad(self, dt, dz, x, kh, kw, cpad, co, ci, dw, db, accumulate=0, ci_total=None, ci_off=0, gen=None, pool=None):
        """gen = (d_out view, w2 pointers, ks, nh, use_hash, seed): dz is the heads' hidden gradient, generated in the kernel (dz: shape only).
        pool = (dy view, nibbles tensor, channels of the pooled layer): dz is that pooling's backward, taken from its sources (dz: shape only)."""
        need = self.L.dbx_conv_wgrad_scratch_bytes(dt, C.byref(dz), C.byref(x), kh, kw)
        if getattr(self, '_wg_scratch', None) is None or self._wg_scratch.numel() < need:
            self._wg_scratch = torch.empty(int(need * 1.25) + 1024, dtype=torch.uint8, device=dw.device)
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if pool is not None:
            dyv, nib, nch, write_dz = pool
            check(self.L.dbx_conv_wgrad_pool_dz(dt, C.byref(dyv), ptr(nib), nch, C.byref(dz), C.byref(x), kh, kw, cpad, co, ci, ptr(dw), ptr(db),
                                                ptr(self._wg_scratch), accumulate, write_dz, stream_ptr()))
        elif gen is not None:
            dov, w2p, ks, nh, use_hash, seed = gen
            check(self.L.dbx_heads1_wgrad_gen(dt, C.byref(dov), C.byref(x), w2p, ks, nh, use_hash, seed, ci, ptr(dw), ci_total, ci_off, ptr(db),
                                              ptr(self._wg_scratch), stream_ptr()))
        elif ci_total is None:
            check(self.L.dbx_conv_wgrad(dt, C.byref(dz), C.byref(x), kh, kw, cpad, co, ci, ptr(dw), ptr(db),
                                        ptr(self._wg_scratch), accumulate, stream_ptr()))
        else:
            check(self.L.dbx_conv_wgrad_slice(dt, C.byref(dz), C.byref(x), kh, kw, cpad, co, ci, ptr(dw), ci_total, ci_off, ptr(db),
                                              ptr(self._wg_scratch), accumulate, stream_ptr()))
        if prof is not None:
            ev1.record()
            buf = C.create_string_buffer(64)                      # the library's own choice (dbx_conv_wgrad_plan)
            check(self.L.dbx_conv_wgrad_plan(dt, C.byref(dz), C.byref(x), kh, kw, buf, 64, None))
            name = buf.value.decode()
            if gen is not None:
                name = name.replace('>', ',gen>')
            if pool is not None:
                name = name.replace('>', ',pool+dz>' if pool[3] else ',pool>')
            prof.append({'kernel': name, 'flops': 2.0 * dz.n * dz.h * dz.w * kh * kw * ci * co, 'start': ev0, 'end': ev1})

    def backward_raw(self, grad_outs):
        """grad_outs: dict output-name -> fp32 NCHW tensor (or None).  Returns dict param-name -> fp32 gradient.
        Must follow a forward_raw(train=True) on the same plan (activations are read from the workspace)."""
        L, kind = self.L, self.kind
        P = self.last_plan
        assert P is not None and P.train, 'backward needs a preceding training-mode forward'
        B, dt, n = P.B, P.dtype_id, P.n
        s = stream_ptr()
        dev = P.ws.device
        heads = _HEADS[kind]
        nh = len(heads)
        h4, w4 = P.h4, P.w4
        G = {}

        sink = self.grad_sink

        def new_grad(name):
            p = self._param(name)
            g = sink.grad_view(name) if sink is not None else torch.empty_like(p, dtype=torch.float32)
            G[name] = g
            return g

        def gout(name, k):
            g = grad_outs.get(name)
            if g is None:
                return torch.zeros((n, k, h4, w4), dtype=torch.float32, device=dev)
            return g.to(torch.float32).contiguous()

        # Weight gradients are off the critical path (only dz -> dgrad -> next dz is a chain): they go to a side stream so
        # their workgroups fill the CUs that the data-gradient kernels' last rounds leave idle.  Every gradient producer and
        # its sink.ready() (which may launch an all-reduce ordered after the CURRENT stream) run on that stream; the main
        # stream joins it before backward_raw returns.  Was worth +0.6 % in round 2; since the fused single-stream paths of rounds 3-4 (which it
        # switches off) it is 4.6 % SLOWER than the default (profiles/r04_sgd_pack_ab.txt).  OPT-IN (DBX_SIDE_STREAM=1); overlapping
        # kernels make per-kernel durations in a rocprofv3 trace of the step meaningless, and never on while profiling.
        main = torch.cuda.current_stream()
        side = None
        multi = sink is not None and getattr(sink, 'world', 1) > 1      # (collectives keep the single-stream ordering)
        if self.profile is None and not multi and os.environ.get('DBX_SIDE_STREAM', '0') == '1':
            if getattr(self, '_side', None) is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side

        def on_side(fn):
            if side is None:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                return fn()

        def conv_bwd(stem, dz, x, kh, kw, cpad, co, ci):
            dw, db = new_grad(stem + '.weight'), new_grad(stem + '.bias')
            # conv1_2's weight gradient takes pool1's backward from its sources -- d_p1 and the arg-max nibbles, 148 MB at batch 64 -- instead of
            # re-reading the 472-MB d_a12 (dbx_conv_wgrad_pool_dz; DBX_POOL_WGRAD=0: from the map) and WRITES that map as a by-product for
            # conv1_2's data gradient, which runs behind it: pool1's backward has no launch of its own (DBX_POOL_WGRAD=2: keeps the launch)
            pool = None
            if stem == 'conv1_2_1' and pool12:
                pool = (B['d_p1'].view(), P.pool_idx['a12'], 64, 1 if pool12 == 1 else 0)

            def run():
                self._wgrad(dt, dz, x, kh, kw, cpad, co, ci, dw, db, pool=pool)
                if sink is not None:
                    sink.ready([stem + '.weight', stem + '.bias'])
            on_side(run)

        def dgrad(stem, src, dst, kh, kw, cpad, rows_pad, cin_pad, gate=None, epi=0, dropmask=None):
            frag = stem in _BWD_IO and kh == 3 and self._frag(P, dt, stem, 'b')
            wp = self._w_bwd(dt, stem, rows_pad, cin_pad, frag=frag)
            e = epi | (_lib.EPI_GATE if gate is not None else 0) | (_lib.CONV_WFRAG if frag else 0)
            self._conv(dt, src, dst, wp, None, kh, kw, cpad, cin_pad, rows_pad, e, gate=gate, dropmask=dropmask,
                       dm_ld=512 * nh)

        # ---- refine branch (DenseBox.py:464-471) backwards
        override = {}
        if kind != 'DenseBox' and P.refine_fwd is not None:
            # by its linear structure: g = up^T(d refine), the folded conv's weight gradient (245 numbers), then every parameter
            # gradient as a small contraction and the heads' incoming gradients + the branch's contribution (csrc/refine_ops.hip)
            lmk, det, wf7, vf7 = P.refine_fwd
            pn = ['conv6_1_det.weight', 'conv6_1_det.bias', 'conv6_2_det.weight', 'conv6_2_det.bias', 'conv6_3_det.weight', 'conv6_3_det.bias']
            ps = [self._param(nm).detach() for nm in pn]
            gs = [new_grad(nm) for nm in pn]

            def f32c(t):
                return None if t is None else t.to(torch.float32).contiguous()
            g_lm, g_det = f32c(grad_outs.get('landmark')), f32c(grad_outs.get('det'))
            o_lm = torch.empty((n, 4, h4, w4), dtype=torch.float32, device=dev)
            o_det = torch.empty((n, 1, h4, w4), dtype=torch.float32, device=dev)
            need = L.dbx_refine_backward_scratch_bytes(n, h4, w4)
            if getattr(self, '_rf_scratch', None) is None or self._rf_scratch.numel() < need:
                self._rf_scratch = torch.empty(int(need), dtype=torch.uint8, device=dev)
            d_rf = gout('refine', 1)
            check(L.dbx_refine_backward(ptr(d_rf), ptr(lmk), ptr(det), n, h4, w4, *[ptr(p) for p in ps], ps[0].shape[0], ptr(wf7), ptr(vf7),
                                        ptr(g_lm), ptr(g_det), ptr(o_lm), ptr(o_det), *[ptr(g_) for g_ in gs], ptr(self._rf_scratch), s))
            if sink is not None:
                sink.ready(pn)
            override = {'landmark': o_lm, 'det': o_det}
            # (P.refine_fwd stays: the forward pass resets it, and a second backward over the same forward -- retain_graph -- re-runs this path)
        elif kind != 'DenseBox' and os.environ.get('DBX_REFINE_LINEAR', '1') != '0':
            raise RuntimeError('densebox_amd: backward of the refine branch without its forward state (no training-mode forward on this plan)')
        elif kind != 'DenseBox':
            d_rfo = B['d_rfo'].view()
            check(L.dbx_nchw_to_framed(dt, ptr(gout('refine', 1)), 1, C.byref(d_rfo), s))
            conv_bwd('conv6_3_det', d_rfo, B['rf_u'].view(), 1, 1, 0, 1, 64)
            dgrad('conv6_3_det', d_rfo, B['d_rf_u'].view(), 1, 1, 0, 64, P.crf)
            check(L.dbx_upsample_bilinear_bwd(dt, C.byref(B['d_rf_u'].view()), C.byref(B['d_rf_2'].view()), None, s))
            conv_bwd('conv6_2_det', B['d_rf_2'].view(), B['rf_1'].view(), 5, 5, 0, 64, 64)
            dgrad('conv6_2_det', B['d_rf_2'].view(), B['d_rf_1'].view(), 5, 5, 4, 64, 64)
            conv_bwd('conv6_1_det', B['d_rf_1'].view(), B['rf_p'].view(), 3, 3, 0, 64, 5)
            dgrad('conv6_1_det', B['d_rf_1'].view(), B['d_rf_p'].view(), 3, 3, 2, 64, 64)
            check(L.dbx_maxpool2x2_bwd(dt, C.byref(B['rf_in'].view()), C.byref(B['d_rf_p'].view()),
                                       C.byref(B['d_rf_in'].view()), 0, 0, s))

        # ---- head outputs: dL/dout into the per-head slots (+ the refine input gradient)
        slot = {}
        srcs = []
        for i, (stem, k) in enumerate(heads):
            slot[stem] = B['d_out'].view(P.crf * i, P.crf)
            src = override.get(stem)
            srcs.append(src if src is not None else gout(stem, k))
        if override or kind == 'DenseBox':         # every head's gradient is final: all slots in one launch
            check(L.dbx_nchw_to_framed_slots(dt, (C.c_void_p * nh)(*[t.data_ptr() for t in srcs]), (C.c_int32 * nh)(*[k for _, k in heads]), nh, P.crf,
                                             C.byref(B['d_out'].view()), s))
        else:
            for (stem, k), t in zip(heads, srcs):
                check(L.dbx_nchw_to_framed(dt, ptr(t), k, C.byref(slot[stem]), s))
        if kind != 'DenseBox' and not override:
            check(L.dbx_framed_add_ch(dt, C.byref(B['d_rf_in'].view()), 0, 4, C.byref(slot['landmark']), 0, s))
            check(L.dbx_framed_add_ch(dt, C.byref(B['d_rf_in'].view()), 4, 1, C.byref(slot['det']), 0, s))

        # ---- heads: 512 -> k (per head), dropout, then the shared 768 -> 512*nh GEMM
        # stage-2 weight/bias gradients of all heads: one streaming pass over the hidden map
        hv = B['hid'].view()
        need = L.dbx_head2_wgrad_scratch_bytes(nh, hv.n * hv.h)
        if getattr(self, '_h2_scratch', None) is None or self._h2_scratch.numel() < need:
            self._h2_scratch = torch.empty(int(need), dtype=torch.uint8, device=dev)
        dw2 = [new_grad('conv5_2_%s.weight' % st) for st, _ in heads]
        db2 = [new_grad('conv5_2_%s.bias' % st) for st, _ in heads]
        prof = self.profile
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        w2s = [self._param('conv5_2_%s.weight' % st).detach() for st, _ in heads]      # fp32 [k,512,1,1]
        ks = (C.c_int32 * nh)(*[k for _, k in heads])
        h2_names = ['conv5_2_%s.weight' % st for st, _ in heads] + ['conv5_2_%s.bias' % st for st, _ in heads]
        mask_p = C.c_void_p(P.mask_buf.data_ptr()) if (P.drop_active and not P.drop_hash) else None
        hash_on, hash_seed = (1 if P.drop_hash else 0), (P.drop_seed if P.drop_hash else 0)
        lin = self._lin_bwd(dt)
        up_done = False
        # the hidden gradient d_hid = keep * (d_out W2): generated inside its two 60x60 consumers (no 944 MB map written and read twice) when
        # the plan holds no buffer for it and nothing needs it in memory (an injected dropout mask, the side-stream schedule do)
        gen = bool(getattr(P, 'heads_gen', False) and lin and side is None and mask_p is None)
        if gen and dt != _lib.F32:
            dhid_v = View(None, hv.n, hv.h, hv.w, 1, 512 * nh, 0, 512 * nh)     # "not in memory" for dbx_head2_backward_up
        else:
            dhid_v = self._d_hid(P).view()
        w2p = (C.c_void_p * nh)(*[w.data_ptr() for w in w2s])
        if side is not None:
            def run_h2():
                check(L.dbx_head2_wgrad(dt, C.byref(B['d_out'].view()), C.byref(hv), ks, nh,
                                        (C.c_void_p * nh)(*[t.data_ptr() for t in dw2]), (C.c_void_p * nh)(*[t.data_ptr() for t in db2]),
                                        ptr(self._h2_scratch), stream_ptr()))
                if sink is not None:
                    sink.ready(h2_names)
            on_side(run_h2)
            check(L.dbx_head2_dgrad(dt, C.byref(B['d_out'].view()), (C.c_void_p * nh)(*[w.data_ptr() for w in w2s]), ks, nh,
                                    C.byref(dhid_v), mask_p, 512 * nh, hash_on, hash_seed, s))
        elif lin:   # one pass over the pixels, and the hidden gradient goes to conv4_4's grid (d_g44 = up^T(d_hid)) while it is in registers
            check(L.dbx_head2_backward_up(dt, C.byref(B['d_out'].view()), C.byref(hv), (C.c_void_p * nh)(*[w.data_ptr() for w in w2s]), ks, nh,
                                          C.byref(dhid_v), mask_p, 512 * nh, hash_on, hash_seed,
                                          (C.c_void_p * nh)(*[t.data_ptr() for t in dw2]), (C.c_void_p * nh)(*[t.data_ptr() for t in db2]),
                                          ptr(self._h2_scratch), C.byref(B['d_g44'].view()), s))
            up_done = True
            if gen and dt == _lib.F32:
                # fp32 parity runs of the generating structure: d_hid had to be written (no fp32 form of the call leaves it out); nothing behind
                # this point may read it -- a consumer that did would return NaN gradients
                bh = P.B['d_hid']
                P.ws[bh.off:bh.off + bh.bytes].view(torch.float32).fill_(float('nan'))
        else:       # one pass over the pixels: the d_hid write overlaps the hid read
            check(L.dbx_head2_backward(dt, C.byref(B['d_out'].view()), C.byref(hv), (C.c_void_p * nh)(*[w.data_ptr() for w in w2s]), ks, nh,
                                       C.byref(dhid_v), mask_p, 512 * nh, hash_on, hash_seed,
                                       (C.c_void_p * nh)(*[t.data_ptr() for t in dw2]), (C.c_void_p * nh)(*[t.data_ptr() for t in db2]),
                                       ptr(self._h2_scratch), s))
            if sink is not None:
                sink.ready(h2_names)
        if prof is not None:
            ev1.record()
            prof.append({'kernel': ('head2_backward_up_kernel<%s>' if up_done else 'head2_wgrad_kernel<%s>') % ('f16', 'bf16', 'f32')[dt],
                         'flops': 2.0 * hv.n * hv.h * hv.w * 512 * sum(k for _, k in heads) * (1 if side is not None else 2), 'start': ev0, 'end': ev1})
        w1n = ['conv5_1_%s.weight' % st for st, _ in heads]
        b1n = ['conv5_1_%s.bias' % st for st, _ in heads]
        if sink is not None:           # the heads' conv5_1 gradients are adjacent in the flat buffer (grad_order)
            dw1, db1 = sink.region(w1n).view(512 * nh, 768, 1, 1), sink.region(b1n)
        else:
            dw1 = torch.empty((512 * nh, 768, 1, 1), dtype=torch.float32, device=dev)
            db1 = torch.empty((512 * nh,), dtype=torch.float32, device=dev)
        c34 = B['fusion'].view(512, 256)
        if lin and not up_done:
            # the hidden gradient on conv4_4's grid: d_g44 = up^T(d_hid) (one HBM-bound pass over the 2048-channel map)
            check(L.dbx_upsample_bilinear_bwd(dt, C.byref(dhid_v), C.byref(B['d_g44'].view()), None, s))

        def run_h1():
            if lin:       # columns 0..511 of dW1 from (d_g44, a44) at 30x30, columns 512..767 (+ the bias) from (d_hid, c34) at 60x60
                self._wgrad(dt, B['d_g44'].view(), B['a44'].view(), 1, 1, 0, 512 * nh, 512, dw1, None, ci_total=768, ci_off=0)
                if gen:
                    self._wgrad(dt, dhid_v, c34, 1, 1, 0, 512 * nh, 256, dw1, db1, ci_total=768, ci_off=512,
                                gen=(B['d_out'].view(), w2p, ks, nh, hash_on, hash_seed))
                else:
                    self._wgrad(dt, dhid_v, c34, 1, 1, 0, 512 * nh, 256, dw1, db1, ci_total=768, ci_off=512)
            else:
                self._wgrad(dt, dhid_v, B['fusion'].view(), 1, 1, 0, 512 * nh, 768, dw1, db1)
            if sink is not None:
                sink.ready(w1n + b1n)
        on_side(run_h1)
        for i, (stem, _) in enumerate(heads):
            G['conv5_1_%s.weight' % stem] = dw1[512 * i:512 * (i + 1)]
            G['conv5_1_%s.bias' % stem] = db1[512 * i:512 * (i + 1)]
        row_bytes = 512 * nh * _lib.ESIZE[dt]
        if lin:
            # data gradients of the two concat parts: d_a44 = gate(W_a^T d_g44) at 30x30, d_c34 = gate(W_c^T d_hid) at 60x60
            fa, fc = self._frag_heads(P, dt, 'ba'), self._frag_heads(P, dt, 'bc')
            self._conv(dt, B['d_g44'].view(), B['d_a44'].view(), self._w_heads1_bwd_part(dt, 'a', frag=fa), None, 1, 1, 0, 512 * nh, 512,
                       _lib.EPI_GATE | (_lib.CONV_WFRAG if fa else 0), gate=B['a44'].view())
            if gen:
                if prof is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                check(L.dbx_heads1_dgrad_gen(dt, C.byref(B['d_out'].view()), w2p, ks, nh, hash_on, hash_seed,
                                             ptr(self._w_heads1_bwd_part(dt, 'c', frag=dt != _lib.F32)), C.byref(B['d_c34'].view()), C.byref(c34), s))
                if prof is not None:
                    ev1.record()
                    prof.append({'kernel': 'heads1_dgrad_gen_kernel<%s>' % ('f16', 'bf16', 'f32')[dt],
                                 'flops': 2.0 * hv.n * hv.h * hv.w * 512 * nh * (256 + 8), 'start': ev0, 'end': ev1})
            else:
                if fc and getattr(P, 'heads_gen', False):      # (the plan chose the fragment image for the generating kernel: does this GEMM take it?)
                    fc = self.conv_plan(dt, dhid_v, B['d_c34'].view(), 1, 1, 0, 512 * nh, 256, _lib.EPI_GATE)[2]
                self._conv(dt, dhid_v, B['d_c34'].view(), self._w_heads1_bwd_part(dt, 'c', frag=fc), None, 1, 1, 0, 512 * nh, 256,
                           _lib.EPI_GATE | (_lib.CONV_WFRAG if fc else 0), gate=c34)
        elif dt != _lib.F32:
            bfrag = self._frag_heads(P, dt, 'b')
            w1t = self._w_heads1_bwd(dt, frag=bfrag)              # [768 rows][512*nh] (or its fragment-order image)
            # one pass over d_hid for both branches of the concat: couts 0..511 -> d_ups, 512..767 -> d_c34 (ReLU-gated)
            d = ConvDesc(dt, 1, 1, 0, 512 * nh, 768, _lib.CONV_WFRAG if bfrag else 0, 0)
            prof = self.profile
            if prof is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            check(L.dbx_conv_forward_split(C.byref(d), C.byref(dhid_v), ptr(w1t), None, C.byref(B['d_ups'].view()), None,
                                           C.byref(B['d_c34'].view()), C.byref(c34), 512, _lib.EPI_GATE, s))
            if prof is not None:
                ev1.record()
                hv = dhid_v
                dv = B['d_ups'].view()
                both = View(dv.ptr, dv.n, dv.h, dv.w, dv.pad, 768, 0, 768)
                prof.append({'kernel': self.conv_plan(dt, hv, both, 1, 1, 0, 512 * nh, 768, 0)[1] if bfrag else
                             'conv_igemm_dma_kernel<%s,256,256>' % ('f16', 'bf16', 'f32')[dt],
                             'flops': 2.0 * hv.n * hv.h * hv.w * 512 * nh * 768, 'start': ev0, 'end': ev1})
        else:
            w1t = self._w_heads1_bwd(dt, frag=False)
            self._conv(dt, dhid_v, B['d_ups'].view(), w1t, None, 1, 1, 0, 512 * nh, 512, 0)
            self._conv(dt, dhid_v, B['d_c34'].view(), w1t[512 * row_bytes:], None, 1, 1, 0, 512 * nh, 256,
                       _lib.EPI_GATE, gate=c34)
        if not lin:
            check(L.dbx_upsample_bilinear_bwd(dt, C.byref(B['d_ups'].view()), C.byref(B['d_a44'].view()),
                                              C.byref(B['a44'].view()), s))

        # ---- backbone, deepest first.  (stem, dz, x, cin, cout, where the data gradient goes, its ReLU gate)
        chain = [
            ('conv4_4_1', 'd_a44', 'a43', 512, 512, 'd_a43', 'a43'),
            ('conv4_3_1', 'd_a43', 'a42', 512, 512, 'd_a42', 'a42'),
            ('conv4_2_1', 'd_a42', 'a41', 512, 512, 'd_a41', 'a41'),
            ('conv4_1_1', 'd_a41', 'p3', 256, 512, 'd_p3', None),
            ('pool', 'fusion', 'd_p3', 'd_c34', 1),
            ('conv3_4_1', 'd_c34', 'a32', 256, 256, 'd_a32', 'a32'),
            ('conv3_2_1', 'd_a32', 'a31', 256, 256, 'd_a31', 'a31'),
            ('conv3_1_1', 'd_a31', 'p2', 128, 256, 'd_p2', None),
            ('pool', 'a22', 'd_p2', 'd_a22', 0),
            ('conv2_2_1', 'd_a22', 'a21', 128, 128, 'd_a21', 'a21'),
            ('conv2_1_1', 'd_a21', 'p1', 64, 128, 'd_p1', None),
            ('pool', 'a12', 'd_p1', 'd_a12', 0),
            ('conv1_2_1', 'd_a12', 'a11', 64, 64, 'd_a11', 'a11'),
            ('conv1_1_1', 'd_a11', 'x0', 3, 64, None, None),
        ]
        fused11 = False
        # pool1's backward inside conv1_2's weight gradient?  0: no; 1: yes, and that kernel writes d_a12 for the data gradient; 2: yes, the map
        # still comes from dbx_maxpool2x2_bwd_idx (A/B)
        pool12 = 0
        if P.pool_idx is not None and os.environ.get('DBX_POOL_WGRAD', '1') != '0' and \
                L.dbx_conv_wgrad_pool_dz_ok(dt, C.byref(B['d_a12'].view()), C.byref(B['a11'].view()), 3, 3):
            # (the by-product needs the weight gradient IN FRONT of the data gradient on one stream: not under the side-stream schedule)
            pool12 = 2 if (os.environ.get('DBX_POOL_WGRAD') == '2' or side is not None) else 1
        for item in chain:
            if item[0] == 'conv1_1_1' and fused11:
                continue                                # its weight gradient came out of conv1_2's data-gradient kernel
            if item[0] == 'conv1_2_1':
                # conv1_2's data gradient feeds conv1_1's weight gradient and nothing else: one kernel, no 472 MB map in between
                stem, dzn, xn, cin, cout, dxn, gaten = item
                d12 = ConvDesc(dt, 3, 3, 1, 64, 64, _lib.EPI_GATE, 0)
                dzv, gv, x0v = B[dzn].view(), B[gaten].view(), B['x0'].view()
                if side is None and os.environ.get('DBX_FUSE_WG1', '1') != '0' and L.dbx_conv_dgrad_wgrad1_fusable(C.byref(d12), C.byref(dzv), C.byref(gv), C.byref(x0v)):
                    conv_bwd(stem, dzv, B[xn].view(), 3, 3, 1, cout, cin)
                    need = L.dbx_conv_dgrad_wgrad1_scratch_bytes()
                    if getattr(self, '_wg1_scratch', None) is None or self._wg1_scratch.numel() < need:
                        self._wg1_scratch = torch.empty(int(need), dtype=torch.uint8, device=dev)
                    dw1, db1 = new_grad('conv1_1_1.weight'), new_grad('conv1_1_1.bias')
                    if prof is not None:
                        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ev0.record()
                    check(L.dbx_conv_dgrad_wgrad1(C.byref(d12), C.byref(dzv), ptr(self._w_bwd(dt, stem, 64, 64, frag=False)),
                                                  C.byref(gv), C.byref(x0v), 3, ptr(dw1), ptr(db1), ptr(self._wg1_scratch), 0, s))
                    if prof is not None:
                        ev1.record()
                        prof.append({'kernel': self.conv_plan(dt, dzv, B[dxn].view(), 3, 3, 1, 64, 64, _lib.EPI_GATE)[1],
                                     'flops': 2.0 * dzv.n * dzv.h * dzv.w * 9 * 64 * (64 + 3), 'start': ev0, 'end': ev1})
                    if sink is not None:
                        sink.ready(['conv1_1_1.weight', 'conv1_1_1.bias'])
                    fused11 = True
                    continue
            if item[0] == 'pool':
                _, xname, dyname, dxname, acc = item
                if xname == 'a12' and pool12 == 1:
                    continue                            # (conv1_2's weight gradient produces d_a12 on its way)
                if P.pool_idx is not None:
                    check(L.dbx_maxpool2x2_bwd_idx(dt, ptr(P.pool_idx[xname]), C.byref(B[dyname].view()), C.byref(B[dxname].view()),
                                                   acc, 1, s))
                    continue
                xv = c34 if xname == 'fusion' else B[xname].view()
                check(L.dbx_maxpool2x2_bwd(dt, C.byref(xv), C.byref(B[dyname].view()), C.byref(B[dxname].view()),
                                           acc, 1, s))
                continue
            stem, dzn, xn, cin, cout, dxn, gaten = item
            conv_bwd(stem, B[dzn].view(), B[xn].view(), 3, 3, 1, cout, cin)
            if dxn is not None:
                dgrad(stem, B[dzn].view(), B[dxn].view(), 3, 3, 1, max(64, cin), cout,
                      gate=B[gaten].view() if gaten else None)
        if side is not None:
            main.wait_stream(side)
        return G
