"""MI355X (gfx950) image-filter hot path: Python host side over the C-ABI.

The directory name carries hyphens (it mirrors the reference repository's name), so it is loaded by
path: see `load_package()` in the repo-root `__graft_entry__.py`.  The product is
lib/libmi355_imgfilter.so (csrc/); this package is only the ctypes door to it.
"""
from .imgfilter import (  # noqa: F401
    FILTER_GAUSS,
    FILTER_GRAY,
    FILTER_GRAY1,
    FILTER_PIPELINE,
    FILTER_SOBEL,
    GAUSS_EXACT,
    GAUSS_FAST,
    IMPL_AUTO,
    IMPL_MFMA,
    INPUT_BGR,
    INPUT_RGBA,
    IMPL_TILE,
    IMPL_VALU,
    Context,
    Group,
    Mi355Error,
    build_library,
    declared_symbols,
    gauss_weights,
    gauss_weights_image2d,
    group_shard,
    library_path,
    load_library,
)
