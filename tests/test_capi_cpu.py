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


def test_image2d_weight_generator_matches_reference_vectors(pkg, golden_weights):
    """mi355_gauss_weights_image2d (product code) against the reference's own image-mode generator's output."""
    n = 0
    for key, bits in golden_weights.items():
        k, s, layout = key.split(",")
        if layout != "image2d":
            continue
        got = pkg.gauss_weights_image2d(int(k[2:]), float(s[6:])).reshape(-1).view(np.uint32)
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
    assert lib.mi355_pool_alloc(None, 2, 4, 4, 1, 5, 1.5, 3, None, None, None) == -1
    assert lib.mi355_pool_free(None, None, None) == -1
    assert lib.mi355_strerror(-1) == b"bad argument"
    assert lib.mi355_strerror(0) == b"ok"
    n = ctypes.c_int(-5)
    assert lib.mi355_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_group_shard_is_bench_shard_range_and_group_validation(pkg):
    """mi355_group_shard (pure host function of the product library) makes the split bench.py's shard_range makes over
    ranks; group creation without a GPU fails with an error code, nothing crashes."""
    import bench
    for n in (1, 2, 3, 4, 8):
        for total in (0, 1, 7, 8, 9, 512):
            spans = [pkg.group_shard(m, n, total) for m in range(n)]
            assert sum(c for _, c in spans) == total and spans[0][0] == 0
            if total:
                assert spans == [bench.shard_range(m, n, 256, total) for m in range(n)]
    assert pkg.group_shard(3, 8, 512) == (192, 64)      # BASELINE config 5
    lib = pkg.load_library()
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.mi355_group_shard(2, 2, 8, ctypes.byref(a), ctypes.byref(b)) == -1
    assert lib.mi355_group_shard(0, 0, 8, ctypes.byref(a), ctypes.byref(b)) == -1
    assert lib.mi355_group_shard(0, 2, 8, None, ctypes.byref(b)) == -1
    g = ctypes.c_void_p()
    assert lib.mi355_group_create(0, None, ctypes.byref(g)) == -1
    assert lib.mi355_group_create(2, None, None) == -1
    assert lib.mi355_group_destroy(None) == -1
    assert lib.mi355_group_filter_batched(None, 0, None, None, 4, 4, 1, 0, 0.0, None) == -1
    n = ctypes.c_int(0)
    lib.mi355_device_count(ctypes.byref(n))
    if n.value == 0:
        assert lib.mi355_group_create(2, None, ctypes.byref(g)) == -3 and not g.value   # MI355_ERR_NO_DEVICE


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


def test_header_is_plain_c_and_links_from_c(pkg, tmp_path):
    """The drop-in boundary is a C ABI: include/mi355_imgfilter.h must compile as C99 (what a cgo / JNI / ctypes-free
    C host includes) and a C program must link against the library and get sane answers from the entry points that
    need no GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(pkg.imgfilter.library_path())
    src = tmp_path / "c_host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "mi355_imgfilter.h"
int main(void) {
    float w[25];
    mi355_ctx* ctx = (mi355_ctx*)0;
    int n = -1;
    if (mi355_gauss_weights(5, 1.5f, w) != MI355_OK) return 1;
    if (mi355_gauss_weights(4, 1.5f, w) != MI355_ERR_BAD_ARG) return 2;
    if (mi355_filter_out_bpp(MI355_FILTER_SOBEL) != 1 || mi355_filter_out_bpp(MI355_FILTER_GAUSS) != 4) return 3;
    if (mi355_device_count(&n) != MI355_OK || n < 0) return 4;
    if (n == 0 && mi355_ctx_create(0, &ctx) != MI355_ERR_NO_DEVICE) return 5;
    if (ctx) mi355_ctx_destroy(ctx);
    if (!mi355_strerror(MI355_ERR_BAD_ARG) || !strlen(mi355_build_info())) return 6;
    printf("%.9g %s\n", (double)w[12], mi355_build_info());
    return 0;
}
''')
    exe = tmp_path / "c_host"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src),
           "-L", lib_dir, "-lmi355_imgfilter", "-Wl,-rpath," + lib_dir, "-o", str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    centre = float(run.stdout.split()[0])
    assert abs(centre - float(pkg.gauss_weights(5, 1.5)[2, 2])) < 1e-9
