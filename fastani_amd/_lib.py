"""Loader of the HIP product library.  Fails loudly: there is no CPU implementation behind fastani_amd."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libfastani_amd.so")
# ANI_LIB_VARIANT=<tag>: an A/B build of the same sources for profiling, csrc/libfastani_amd.<tag>.so — a file name inside the
# package only, never an arbitrary path
_variant = os.environ.get("ANI_LIB_VARIANT", "")
if _variant:
    if not _variant.replace("_", "").replace("-", "").isalnum():
        raise ImportError("fastani_amd: ANI_LIB_VARIANT must be a plain tag")
    LIB_PATH = os.path.join(HERE, "csrc", "libfastani_amd.%s.so" % _variant)

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("fastani_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        try:
            # PyTorch-ROCm bundles its own libamdhip64; when both live in one process torch must be imported first so that
            # a single HIP runtime is shared (plumbing only — nothing here needs torch)
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib
