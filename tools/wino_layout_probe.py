"""Times the Winograd 3x3 kernel alone: dense destination against the Fire layout (behind the expand1x1 channels of a wider tile)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
n = 4096
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for cin, cout, t in ((64, 256, 18), (48, 192, 16), (48, 192, 14), (32, 128, 16), (16, 64, 14), (16, 64, 12)):
    x = torch.randn((n, t, t, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3)) * 0.05
    ww = torch.empty(16 * cin * cout)
    assert lib.swk_winograd_f2x2_3x3_weights(w.data_ptr(), cout, cin, ww.data_ptr()) == 0
    ww = ww.to(dev)
    bias = torch.zeros(cout, device=dev)
    o = t - 2
    for name, dH, dC, off, c_off in (("dense", o, cout, 0, 0), ("fire", o, 2 * cout, 0, cout), ("fire+ring", o + 3, 2 * cout, 1, cout)):
        dst = torch.zeros((n, dH, dH, dC), device=dev)
        ts = []
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.swk_nhwc_conv3x3_winograd_bias_relu_place(stream, x.data_ptr(), n, t, cin, ww.data_ptr(), bias.data_ptr(), cout,
                                                               dst.data_ptr(), dH, dH, dC, off, off, c_off)
            e1.record()
            assert rc == 0
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%3d->%3d tile %2d  %-10s %8.1f us" % (cin, cout, t, name, 1e3 * min(ts[1:])))
