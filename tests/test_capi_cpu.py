"""CPU suite: the C-ABI library loads without a GPU, exports every symbol include/*.h declares, and
its pure-host logic (coefficient generator, argument validation, error strings) is right.
No kernel is launched here."""
import ctypes

import numpy as np


def test_library_builds_and_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = pkg.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "missing export: %s" % n
    assert b"gfx950" in lib.mi355_build_info()


def test_host_weight_generator_matches_reference_vectors(pkg, golden_weights):
    """mi355_gauss_weights (product code, capi.hip) against the reference's own output."""
    n = 0
    for key, bits in golden_weights.items():
        k, s, layout = key.split(",")
        if layout != "buffers":
            continue
        got = pkg.gauss_weights(int(k[2:]), float(s[6:])).reshape(-1).view(np.uint32)
        assert got.tolist() == bits, key
        n += 1
    assert n >= 15


def test_argument_validation_needs_no_gpu(pkg):
    lib = pkg.load_library()
    buf = (ctypes.c_float * 9)()
    assert lib.mi355_gauss_weights(4, 1.0, buf) == -1          # even kernel_size
    assert lib.mi355_gauss_weights(65, 1.0, buf) == -1         # > MI355_MAX_GAUSS_K
    assert lib.mi355_gauss_weights(3, 0.0, buf) == -1
    assert lib.mi355_gauss_weights(3, float("nan"), buf) == -1
    assert lib.mi355_gauss_weights(3, 1.0, None) == -1
    assert lib.mi355_gauss_weights(3, 1.0, buf) == 0
    assert lib.mi355_filter_out_bpp(pkg.FILTER_GAUSS) == 4
    assert lib.mi355_filter_out_bpp(pkg.FILTER_SOBEL) == 1
    assert lib.mi355_filter_out_bpp(99) == -1
    # a null context is rejected before anything touches HIP
    assert lib.mi355_sync(None) == -1
    assert lib.mi355_gray_rgba8(None, None, None, 4, 4, None) == -1
    assert lib.mi355_filter_dev(None, 0, None, None, 4, 4, 1, 0, 0.0) == -1
    assert lib.mi355_ctx_destroy(None) == -1
    assert lib.mi355_strerror(-1) == b"bad argument"
    assert lib.mi355_strerror(0) == b"ok"
    n = ctypes.c_int(-5)
    assert lib.mi355_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_no_product_file_references_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    import os
    import __graft_entry__ as entry
    bad = []
    for root, _, files in os.walk(entry.PKG_DIR):
        if os.sep + "lib" in root:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                for needle in ("liboracle", "from oracle", "import oracle", "oracle/_ref", "oracle_gauss"):
                    if needle in text:
                        bad.append((f, needle))
    assert not bad, bad
