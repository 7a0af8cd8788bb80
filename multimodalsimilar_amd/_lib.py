"""ctypes binding of libmmsim_hip.so, generated from include/mmsim_hip.h.

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "mmsim_hip.h")
LIBPATH = os.environ.get("MMSIM_LIB") or os.path.join(HERE, "libmmsim_hip.so")      # MMSIM_LIB: A/B a second build of the library

_SCALARS = {"int": ctypes.c_int, "float": ctypes.c_float, "unsigned long long": ctypes.c_uint64,
            "unsigned int": ctypes.c_uint32, "long long": ctypes.c_int64}


MIN_VERSION = 300      # include/mmsim_hip.h: the header this binding is generated from


class MmsimError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    decls = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(mmsim_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argl.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    toks = a.split(" ")
                    typ, nm = " ".join(toks[:-1]), toks[-1]
                    argl.append((_SCALARS[typ], nm))
        decls[name] = (restype, argl)
    return decls


class _Lib:
    def __init__(self):
        self._dll = None
        self._decls = None

    def load(self):
        if self._dll is not None:
            return self
        # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (loaded first), otherwise
        # two runtimes coexist and launches fail with "no ROCm-capable device is detected"
        import torch  # noqa: F401
        if not os.path.exists(LIBPATH):
            raise MmsimError(
                f"{LIBPATH} not found: the HIP library has not been built "
                "(run `python -m multimodalsimilar_amd.build`); there is no CPU fallback")
        self._dll = ctypes.CDLL(LIBPATH)
        self._dll.mmsim_version.restype = ctypes.c_int
        if self._dll.mmsim_version() < MIN_VERSION:
            raise MmsimError(f"{LIBPATH} is version {self._dll.mmsim_version()}, this binding needs >= {MIN_VERSION} "
                             "(fp16 image-tower tensors, mmsim_embed_ln_bwd2): rebuild with `python -m multimodalsimilar_amd.build`")
        self._decls = parse_header()
        for name, (restype, argl) in self._decls.items():
            fn = getattr(self._dll, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = [t for t, _ in argl]
        return self

    def last_error(self):
        return (self._dll.mmsim_last_error() or b"").decode()

    def __getattr__(self, name):
        # lib.gemm_bf16(...) -> checked call of mmsim_gemm_bf16
        self.load()
        full = "mmsim_" + name
        if full not in self._decls:
            raise AttributeError(name)
        fn = getattr(self._dll, full)
        restype = self._decls[full][0]

        def call(*args):
            rc = fn(*args)
            if restype is ctypes.c_int and name not in _VALUE_RETURNING and not name.endswith("_eligible") and rc != 0:
                raise MmsimError(f"{full}: {self.last_error()} (code {rc})")
            return rc
        object.__setattr__(self, name, call)
        return call


# entry points whose int return is a value, not a status (plus every *_eligible predicate)
_VALUE_RETURNING = ("version", "device_count", "get_deterministic", "embed_ln_bwd_scratch_floats")

lib = _Lib()
