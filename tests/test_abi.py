"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports exactly the symbols
include/densebox_hip.h declares; the ctypes table matches the header; product code never touches the oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'densebox_hip.h')


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dbx_[a-z0-9_]+)\s*\(', src)))


def _header_abi_version():
    return int(re.search(r'#define\s+DBX_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))


def test_library_builds_and_exports_every_declared_symbol():
    from densebox_amd import _build, _lib
    lib = _build.build(verbose=False)
    assert os.path.exists(lib)
    L = _lib.lib()
    declared = _header_functions()
    assert len(declared) >= 25
    assert _lib.MISSING == [], 'declared but not exported: %s' % _lib.MISSING
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    out = subprocess.run(['nm', '-D', '--defined-only', lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (dbx_[a-z0-9_]+)', out))
    assert set(declared) <= exported, set(declared) - exported
    assert L.dbx_version() == _lib.ABI_VERSION == _header_abi_version()
    assert isinstance(L.dbx_last_error(), bytes)


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument validation happens on the host before any launch: error code + message, never an exception/abort."""
    import ctypes as C
    from densebox_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc(_lib.F16, 3, 3, 1, 64, 64, 0)
    x = _lib.View(C.c_void_p(0x1000), 1, 8, 8, 0, 64, 0, 64)      # frame 0 < conv padding 1
    y = _lib.View(C.c_void_p(0x2000), 1, 8, 8, 1, 64, 0, 64)
    rc = L.dbx_conv_forward(C.byref(d), C.byref(x), C.c_void_p(0x3000), None, C.byref(y), None, None, 0, None)
    assert rc == -1 and b'frame' in L.dbx_last_error()
    d2 = _lib.ConvDesc(7, 3, 3, 1, 64, 64, 0)
    assert L.dbx_conv_forward(C.byref(d2), C.byref(y), C.c_void_p(0x3000), None, C.byref(y), None, None, 0, None) == -3
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_packed_weight_size():
    import ctypes as C
    from densebox_amd import _lib
    L = _lib.lib()
    # 3x3, cin 64, f16: K = 576 elements (1152 B = 9 x 128 B); conv1_1 f16 (cin_pad 8): 144 B -> 256 B = 128 elements
    assert L.dbx_conv_packed_elems(C.byref(_lib.ConvDesc(_lib.F16, 3, 3, 1, 64, 64, 0))) == 64 * 576
    assert L.dbx_conv_packed_elems(C.byref(_lib.ConvDesc(_lib.F16, 3, 3, 1, 8, 64, 0))) == 64 * 128
    assert L.dbx_conv_packed_elems(C.byref(_lib.ConvDesc(_lib.F32, 3, 3, 1, 4, 64, 0))) == 64 * 64


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, 'densebox_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
                assert '/root/reference' not in src, f


def test_no_cpu_fallback():
    import torch
    import densebox_amd as D
    from densebox_amd import synth
    net = D.DenseBoxLM(synth.vgg19_standin(seed=0))
    with pytest.raises(RuntimeError, match='no CPU path'):
        net(torch.zeros(1, 3, 240, 240))


def test_integration_doc_structs_match_the_binding():
    """The ctypes structs INTEGRATION.md shows to third-party callers have the fields of densebox_amd/_lib.py (round-2 verdict: the
    doc's ConvDesc had lost drop_seed, a binding copied from it passed a short struct)."""
    import re
    from densebox_amd import _lib
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    for cls in (_lib.View, _lib.ConvDesc):
        m = re.search(r'class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n' % cls.__name__, doc, re.S)
        assert m, 'INTEGRATION.md does not show ' + cls.__name__
        names = re.findall(r"\('(\w+)', C\.(\w+)\)", m.group(1))
        assert [n for n, _ in names] == [f[0] for f in cls._fields_], (cls.__name__, names)
        import ctypes as C
        assert [getattr(C, t) for _, t in names] == [f[1] for f in cls._fields_], (cls.__name__, names)     # (c_int32 is c_int)
