import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
rng = np.random.default_rng(5)
with pkg.Context(0) as ctx:
    for (h, w, alpha) in [(1, 2, None), (1, 2, 255), (3, 6, None), (30, 100, None), (30, 100, 255)]:
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if alpha is not None:
            img[..., 3] = alpha
        ctx.set_impl(pkg.IMPL_TILE); t = ctx.gauss(img, 11, 3.0)
        ctx.set_impl(pkg.IMPL_AUTO); g = ctx.gauss(img, 11, 3.0)
        d = (t.astype(int) - g.astype(int))
        print(h, w, alpha, "mismatch", int((d != 0).sum()), "of", d.size)
        if (d != 0).any():
            idx = np.argwhere(d != 0)[:6]
            for i in idx:
                print("   ", tuple(i), "tile", t[tuple(i)], "wide", g[tuple(i)])
