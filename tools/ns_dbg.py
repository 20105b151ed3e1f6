import numpy as np, sys
sys.path.insert(0, '.')
from swiftwatcher_amd import _lib
for name in ["ialm_128x160x7", "ialm_64x96x21", "ialm_64x96x64"]:
    g = np.load("tests/golden/%s.npz" % name)
    fr = g["frames"]; n, H, W = fr.shape
    out = {}
    for m in (0, 1):
        ctx = _lib.Context(0); ctx.set_eig_method(m)
        A, E, it = ctx.ialm(fr.reshape(n, H * W))
        out[m] = (A, E, it); ctx.close()
    r = g["rows"]
    print(name, "iters", out[0][2], out[1][2], int(g["iters"]),
          "NS vs golden dA %.2e" % np.abs(out[0][0][r] - g["A_rows"]).max(),
          "Jac vs golden dA %.2e" % np.abs(out[1][0][r] - g["A_rows"]).max(),
          "NS vs Jac %.2e" % np.abs(out[0][0] - out[1][0]).max())
