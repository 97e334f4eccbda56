# coding: utf-8
"""ctypes binding of ``libzero_hip.so`` -- the only door between the Python host
code and the HIP kernels.

The prototypes are parsed from ``include/zero_hip.h`` so that the header *is*
the binding (a symbol declared there but missing from the library is an import
error, which tests/test_abi.py relies on).  There is no CPU fallback: if the
library is absent or a call fails, this raises.
"""

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# ZERO_HIP_LIB: an alternative build of the same library (the AddressSanitizer build of scripts/asan_host.sh)
LIB_PATH = os.environ.get("ZERO_HIP_LIB") or os.path.join(_HERE, "csrc", "libzero_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "zero_hip.h")


class ZeroHipError(RuntimeError):
    pass


_CTYPE = [
    ("*", ctypes.c_void_p),
    ("zk_stream_t", ctypes.c_void_p),
    ("size_t", ctypes.c_size_t),
    ("uint32_t", ctypes.c_uint32),
    ("uint64_t", ctypes.c_uint64),
    ("long", ctypes.c_long),
    ("float", ctypes.c_float),
    ("int", ctypes.c_int),
]


def _ctype_of(decl):
    for key, ct in _CTYPE:
        if key in decl:
            return ct
    raise ValueError("unknown C type in %r" % decl)


def parse_header(path=HEADER_PATH, experiments=True):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype.  experiments=False leaves out the
    ``#ifdef ZK_EXPERIMENTS`` sections (entry points only a ``make EXPERIMENTS=1`` library exports)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    if not experiments:
        text = re.sub(r"#ifdef ZK_EXPERIMENTS.*?#endif", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|size_t|uint32_t|int)\s+(zk_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        if ret == "int":
            restype = ctypes.c_int
        elif ret == "size_t":
            restype = ctypes.c_size_t
        elif ret == "uint32_t":
            restype = ctypes.c_uint32
        else:
            restype = ctypes.c_char_p
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argtypes.append(_ctype_of(a))
                argnames.append(re.split(r"[\s\*]+", a)[-1])
        protos[name] = (restype, argtypes, argnames)
    return protos


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ZeroHipError(
                "HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C zero_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        self._dll = ctypes.CDLL(LIB_PATH)
        self.ncalls = 0
        self.recording = False        # a layer program is being recorded (func.Engine.run_program)
        core = parse_header(experiments=False)
        every = parse_header(experiments=True)
        # the experiment entry points (negative results kept as evidence, `make EXPERIMENTS=1`) come as a set
        self.experiments = all(hasattr(self._dll, n) for n in every if n not in core)
        self.protos = every if self.experiments else core
        for name, (restype, argtypes, _) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise ZeroHipError("symbol %s declared in include/zero_hip.h is not exported by %s"
                                   % (name, LIB_PATH))
            fn.restype = restype
            fn.argtypes = argtypes
        self._dll.zk_last_error_string.restype = ctypes.c_char_p

    def raw(self, name):
        return getattr(self._dll, name)

    # entry points that append an op while a layer program is being recorded (include/zero_hip.h, zk_prog_*)
    RECORDABLE = frozenset(["zk_gemm", "zk_attn_fwd", "zk_attn_bwd", "zk_add_ln_fwd", "zk_prog_begin", "zk_prog_end"])

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        if self.recording and name not in self.RECORDABLE:
            # it would launch immediately, ahead of the ops recorded before it
            raise ZeroHipError("%s cannot be part of a layer program" % name)
        self.ncalls += 1          # lets callers tell whether anything was enqueued between two points
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            msg = self._dll.zk_last_error_string()
            raise ZeroHipError("%s failed (rc=%d): %s" % (name, rc, msg.decode() if msg else ""))
        return 0

    def query(self, name, *args):
        """Call a size_t-returning *_workspace function."""
        return int(getattr(self._dll, name)(*args))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def ptr(t, offset_elems=0):
    """Device address of a torch tensor (+ element offset); None -> NULL."""
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()
