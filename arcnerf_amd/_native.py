"""ctypes binding of libarcnerf_hip.so (C ABI declared in include/arcnerf_hip.h).

argtypes are derived from the header itself, so the header is the single source of truth.  There is NO fallback:
if the library is missing or a kernel launch fails the call raises RuntimeError (the product path must fail loudly).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, 'include', 'arcnerf_hip.h')
LIB_PATH = os.path.join(_HERE, 'lib', 'libarcnerf_hip.so')

MAX_LEVELS = 32
ACT = {None: 0, 'none': 0, 'relu': 1, 'sigmoid': 2, 'truncexp': 3, 'exponential': 3, 'softplus': 4, 'squareplus': 5, 'sine': 6}


class HashGridDesc(C.Structure):
    _fields_ = [('n_levels', C.c_int32), ('n_feat', C.c_int32), ('resolutions', C.c_int32 * MAX_LEVELS),
                ('offsets', C.c_int64 * (MAX_LEVELS + 1)), ('min_xyz', C.c_float * 3), ('max_xyz', C.c_float * 3)]


class MlpDesc(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('dims', C.c_int32 * 10), ('act_hidden', C.c_int32), ('act_out', C.c_int32),
                ('has_bias', C.c_int32), ('softplus_beta', C.c_float)]


_SCALARS = {'int': C.c_int, 'int32_t': C.c_int32, 'uint32_t': C.c_uint32, 'int64_t': C.c_int64, 'uint64_t': C.c_uint64, 'float': C.c_float,
            'double': C.c_double}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    protos = {}
    for m in re.finditer(r'\b(int|int64_t|void|const char \*)\s*(arcn_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = ' '.join(args.split())
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(C.c_void_p)
                else:
                    ty = a.replace('const ', '').split()[0]
                    argtypes.append(_SCALARS[ty])
        restype = {'int': C.c_int, 'int64_t': C.c_int64, 'void': None, 'const char *': C.c_char_p}[ret]
        protos[name] = (restype, argtypes)
    return protos


_lib = None


def lib():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libarcnerf_hip.so is not built ({}); run `python -c "import __graft_entry__ as g; '
                               'g.build()"` or `make -C arcnerf_amd/csrc`'.format(LIB_PATH))
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().arcn_last_error()
        raise RuntimeError('arcnerf_hip {} failed (rc={}): {}'.format(what, rc, msg.decode() if msg else ''))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), 'arcnerf_hip ops need contiguous tensors'
    return t.data_ptr()


def stream():
    """Raw handle of torch's current stream on the current device.  Through the C binding the compiled-kernel launchers of torch use
    (0.3 us); `torch.cuda.current_stream().cuda_stream` builds a Python Stream object first (7 us per call - every launch asks)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def make_hashgrid_desc(resolutions, offsets, n_feat, min_xyz, max_xyz):
    d = HashGridDesc()
    d.n_levels = len(resolutions)
    d.n_feat = int(n_feat)
    for i, r in enumerate(resolutions):
        d.resolutions[i] = int(r)
    for i, o in enumerate(offsets):
        d.offsets[i] = int(o)
    for k in range(3):
        d.min_xyz[k] = float(min_xyz[k])
        d.max_xyz[k] = float(max_xyz[k])
    return d


def make_mlp_desc(dims, act_hidden='relu', act_out=None, has_bias=False, beta=1.0):
    d = MlpDesc()
    d.n_layers = len(dims) - 1
    for i, v in enumerate(dims):
        d.dims[i] = int(v)
    d.act_hidden = ACT[act_hidden.lower() if isinstance(act_hidden, str) else act_hidden]
    d.act_out = ACT[act_out.lower() if isinstance(act_out, str) else act_out]
    d.has_bias = int(bool(has_bias))
    d.softplus_beta = float(beta)
    return d
