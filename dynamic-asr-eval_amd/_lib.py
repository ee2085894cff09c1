"""ctypes binding of libdyneval_hip.so (the C-ABI declared in include/dyneval.h).

The product path has NO CPU fallback: if the shared library is missing or an entry point returns an error the
caller gets an exception, never a silently different implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYN_LIB_PATH") or os.path.join(_HERE, "libdyneval_hip.so")  # override: A/B builds only
_lib = None


class DynError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """Mirror of dyn_gemm_desc (include/dyneval.h)."""
    _fields_ = [
        ("trans_a", ctypes.c_int32), ("trans_b", ctypes.c_int32),
        ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
        ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("sa1", ctypes.c_int64), ("sa2", ctypes.c_int64),
        ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64), ("sb1", ctypes.c_int64), ("sb2", ctypes.c_int64),
        ("C", ctypes.c_void_p), ("ldc", ctypes.c_int64), ("sc1", ctypes.c_int64), ("sc2", ctypes.c_int64),
        ("bias", ctypes.c_void_p),
        ("nb1", ctypes.c_int64), ("nb2", ctypes.c_int64),
        ("split_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
        ("tile_m", ctypes.c_int32), ("tile_n", ctypes.c_int32), ("tail_slices", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("C_in", ctypes.c_void_p),
        ("epilogue", ctypes.c_int32), ("reserved2_", ctypes.c_int32),
        ("aux", ctypes.c_void_p),
        ("a_colsum", ctypes.c_void_p),
        ("a_colsum_beta", ctypes.c_float), ("reserved3_", ctypes.c_float),
        ("counters", ctypes.c_void_p), ("n_counters", ctypes.c_int64),
        ("bias_s1", ctypes.c_int64), ("bias_s2", ctypes.c_int64),
    ]


class DecoderDesc(ctypes.Structure):
    """Mirror of dyn_decoder_desc (include/dyneval.h)."""
    _fields_ = [
        ("d_model", ctypes.c_int32), ("heads", ctypes.c_int32), ("d_ff", ctypes.c_int32), ("vocab", ctypes.c_int32),
        ("layers", ctypes.c_int32), ("n_enc", ctypes.c_int32), ("max_positions", ctypes.c_int32), ("reserved_", ctypes.c_int32),
        ("eps", ctypes.c_float), ("reserved2_", ctypes.c_float),
        ("embed", ctypes.c_void_p), ("pos_table", ctypes.c_void_p), ("norm_out_w", ctypes.c_void_p), ("norm_out_b", ctypes.c_void_p),
        ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p),
        ("layer_ptrs", ctypes.POINTER(ctypes.c_void_p)),
        ("tokens", ctypes.c_void_p), ("logits", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("scratch_floats", ctypes.c_int64),
    ]


DEC_PTRS_PER_LAYER = 18


def load():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DynError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C dynamic-asr-eval_amd/csrc`). "
            "There is no CPU fallback for the product path.")
    # Load order matters: PyTorch-ROCm ships its OWN HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and this library
    # is linked against the same SONAME.  With torch imported first the loader binds libdyneval_hip.so to torch's runtime — one HIP
    # runtime in the process, torch's streams and allocations are valid in our launches.  Loaded first, the system runtime under
    # /opt/rocm comes in as well and every launch from here fails with "no ROCm-capable device is detected" (seen with build()
    # followed by smoke() in one process).  The host side is PyTorch-based anyway (memory, streams), so import it here.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in prototypes().items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().dyn_last_error().decode("utf-8", "replace")
        raise DynError(f"{what} failed with code {rc}: {msg}")


_CTYPES = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "uint64_t": ctypes.c_uint64,
}
_PROTOS = None


def prototypes():
    """Parse include/dyneval.h into {name: (restype, argtypes)} so the header is the single source of truth for
    the binding (an int64_t passed as a bare Python int would otherwise be truncated to a C int)."""
    global _PROTOS
    if _PROTOS is not None:
        return _PROTOS
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "dyneval.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef struct \{.*?\} \w+;", "", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", "", text, flags=re.M)
    protos = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(dyn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret = ret.strip()
        if ret == "const char*":
            restype = ctypes.c_char_p
        else:
            restype = _CTYPES[ret]
        argtypes = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                argtypes.append(ctypes.POINTER(GemmDesc) if "dyn_gemm_desc" in a else
                                ctypes.POINTER(DecoderDesc) if "dyn_decoder_desc" in a else ctypes.c_void_p)
            else:
                argtypes.append(_CTYPES[a.replace("const ", "").split()[0]])
        protos[name] = (restype, argtypes)
    _PROTOS = protos
    return protos


def exported_symbols():
    """Names declared in include/dyneval.h (used by the CPU-side ABI test)."""
    return sorted(prototypes().keys())
