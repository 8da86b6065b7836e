"""ctypes binding of liblmkd_hip.so.  The prototypes are parsed from include/lmkd.h (the single
source of truth for the C ABI), so every declared symbol must be exported or the import fails.

There is NO fallback: if the shared library is missing, or a call returns non-zero, a
RuntimeError is raised (the reference raises Python exceptions at the same places)."""
import ctypes
import os
import re

from . import _audit

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "lmkd.h")
# LMKD_LIB: another build of the same sources (tools/ab_build.sh: kernel A/B comparisons on one GPU box)
LIB_PATH = os.environ.get("LMKD_LIB") or os.path.join(_HERE, "liblmkd_hip.so")

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "char": ctypes.c_char,
    "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long|const char\s*\*)\s+(lmkd_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _CT[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = a.replace("const ", "").split()
                    ty = " ".join(toks[:-1])
                    argtypes.append(_CT[ty])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liblmkd_hip.so not found at %s: build it with `python lite-mkd_amd/build.py` "
                "(there is no CPU or PyTorch fallback for the hot path)" % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64, and a process must not end up with two HIP runtimes - loaded after torch,
        # liblmkd_hip.so binds to the runtime torch brought (loaded before it, the library initialises /opt/rocm's copy and the devices
        # are then invisible to one of the two)
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)       # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        if os.environ.get("LMKD_PATCH16"):      # A/B measurements: 2 = every two-plane patch launch on the 16x16x32 kernel (layer 1 too), 0 = all on 32x32x16
            self.cdll.lmkd_conv_set_patch16(int(os.environ["LMKD_PATCH16"]))
        if os.environ.get("LMKD_WGRAD_WINDOW"):      # A/B measurements: the per-segment workgroup target of the window weight gradient's slab plan (512 = round 4's)
            self.cdll.lmkd_conv_set_wgrad_window(int(os.environ["LMKD_WGRAD_WINDOW"]))
        if os.environ.get("LMKD_CONV_PERSISTENT", "1") == "0":      # A/B measurements: one workgroup per tile in the LDS-patch kernels (same results)
            self.cdll.lmkd_conv_set_persistent(0)

    def last_error(self):
        return self.cdll.lmkd_last_error().decode()

    def call(self, name, *args):
        """Call an int-returning entry point; non-zero -> RuntimeError with the library's message."""
        if _audit.ON:      # LMKD_STREAM_AUDIT: the tensors named to this call (ops._p) against the stream it launches on
            _audit.check(name)
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, self.last_error()))

    def value(self, name, *args):
        return getattr(self.cdll, name)(*args)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
