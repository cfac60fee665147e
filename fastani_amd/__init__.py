"""fastani_amd — MI355X-native ANI engine (FastANI hot path: Sketch -> Map -> computeCGI) behind a C-ABI.

    from fastani_amd import engine
    e = engine()                         # loads csrc/libfastani_amd.so (HIP, gfx950); raises if missing / no GPU
"""
from .api import (AniError, CGI_DT, DeviceGenomes, Engine, FragmentSet, HostGenomes, MAPPING_DT, MINIMIZER_DT, Params, Sketch, SketchWriter, UploadedGenomes)


def engine(device=0):
    from . import _lib
    return Engine(_lib.load(), device)
