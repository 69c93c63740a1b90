"""ctypes binding of libhvgan.so (the C ABI declared in include/hvgan.h).

The header is the single source of truth: structs and prototypes are parsed from it, so the Python
side cannot drift from the C side.  ``get()`` loads the library or raises -- there is NO fallback:
every operator of this package fails loudly when the HIP library is missing.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'hvgan.h')
LIB_PATH = os.environ.get('HVGAN_LIB') or os.path.join(_HERE, 'libhvgan.so')   # HVGAN_LIB: A/B a kernel variant build

F32, F16 = 0, 1
ACT = {'none': 0, 'elu': 1, 'relu': 2, 'lrelu': 3, 'sigmoid': 4, 'clamp': 5}
NORM = {'none': 0, 'batch': 1, 'instance': 2}
ERRORS = {-1: 'HV_ERR_ARG (invalid argument)', -2: 'HV_ERR_UNSUPPORTED (shape not supported by the kernels)',
          -3: 'HV_ERR_WORKSPACE (workspace too small)'}

_BASE = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double, 'long long': ctypes.c_longlong,
    'size_t': ctypes.c_size_t, 'void': None, 'char': ctypes.c_char,
}


def _ctype(tname, structs):
    t = tname.replace('const', '').strip()
    nptr = t.count('*')
    t = t.replace('*', '').strip()
    if nptr:
        if t in structs:
            return ctypes.POINTER(structs[t])
        if t == 'char':
            return ctypes.c_char_p
        return ctypes.c_void_p
    if t in structs:
        return structs[t]
    return _BASE[t]


def _strip_comments(src):
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    return re.sub(r'//[^\n]*', ' ', src)


def parse_header(path=HEADER):
    """-> (structs: name -> ctypes.Structure subclass, protos: name -> (restype, [argtypes]))."""
    src = _strip_comments(open(path).read())
    structs = {}
    for body, name in re.findall(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r'((?:const\s+)?(?:long long|size_t|int|float|double|\w+)\s*\**)\s*(.*)', decl)
            tname, names = m.group(1), m.group(2)
            for nm in names.split(','):
                nm = nm.strip()
                extra = nm.count('*')
                nm = nm.replace('*', '').strip()
                fields.append((nm, _ctype(tname + '*' * extra, structs)))
        structs[name] = type(name, (ctypes.Structure,), {'_fields_': fields})
    protos = {}
    body = re.sub(r'typedef\s+struct\s*\{.*?\}\s*\w+\s*;', ' ', src, flags=re.S)
    for ret, name, args in re.findall(r'((?:const\s+)?(?:int|size_t|char)\s*\**)\s*(hv_\w+)\s*\(([^;{]*?)\)\s*;', body, flags=re.S):
        argtypes = []
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                m = re.match(r'((?:const\s+)?(?:long long|size_t|\w+)\s*\**)\s*\w*$', a)
                argtypes.append(_ctype(m.group(1), structs))
        protos[name] = (_ctype(ret, structs), argtypes)
    return structs, protos


_CTYPES_NAME = {ctypes.c_int: 'ctypes.c_int', ctypes.c_float: 'ctypes.c_float', ctypes.c_double: 'ctypes.c_double',
                ctypes.c_longlong: 'ctypes.c_longlong', ctypes.c_size_t: 'ctypes.c_size_t', ctypes.c_void_p: 'ctypes.c_void_p',
                ctypes.c_char_p: 'ctypes.c_char_p', ctypes.c_char: 'ctypes.c_char'}


def ctypes_source(struct_name, path=HEADER, width=118):
    """Python source of the ctypes mirror of one struct of include/hvgan.h, generated from the header: the snippet INTEGRATION.md shows
    a maintainer of the reference (tools/gen_integration_stub.py writes it there; tests/test_host_cpu.py checks the two agree)."""
    structs, _ = parse_header(path)
    fields = ['("%s", %s)' % (n, _CTYPES_NAME[t]) for n, t in structs[struct_name]._fields_]
    lines, cur = [], '    _fields_ = ['
    for i, f in enumerate(fields):
        piece = f + (', ' if i + 1 < len(fields) else ']')
        if len(cur) + len(piece) > width:
            lines.append(cur.rstrip())
            cur = ' ' * 16
        cur += piece
    lines.append(cur)
    return 'class %s(ctypes.Structure):          # generated from include/hvgan.h (healthivert-gan_amd/lib.py: ctypes_source)\n%s\n' % (
        struct_name, '\n'.join(lines))


class HipLibraryMissing(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "libhvgan.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.structs, self.protos = parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self.cdll, name)   # AttributeError if the header declares a symbol the .so lacks
            fn.restype = res
            fn.argtypes = args
        for k, v in self.structs.items():
            setattr(self, k, v)

    def call(self, name, *args):
        if _DIAG_SKIP_C and name in _DIAG_SKIP_C:      # timing-only diagnostics (tools/marginal_step.sh): the entry point is not called, results are wrong
            return
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            if rc <= -1000:
                raise RuntimeError("%s: HIP error %d" % (name, -rc - 1000))
            raise RuntimeError("%s: %s" % (name, ERRORS.get(rc, 'error %d' % rc)))

    def size(self, name, *args):
        return int(getattr(self.cdll, name)(*args))


_lib = None
_DIAG_SKIP_C = frozenset(x for x in os.environ.get('HV_DIAG_SKIP_C', '').split(',') if x)


def get():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def available():
    return os.path.exists(LIB_PATH)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("healthivert-gan_amd operators run only on an MI355X device tensor "
                               "(got a %s tensor); there is no CPU path." % t.device)
