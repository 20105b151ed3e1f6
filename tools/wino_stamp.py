"""Where a phase of the Winograd 3x3 kernel spends its cycles: diagnostic build of csrc/cnn_wino3x3.hip with s_memtime brackets
(-DSWK_WINO_STAMP -> tools/libwino_stamp.so, built by tools/wino_stamp.sh).  Shares, not lengths: the stamps fence the schedule."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libwino_stamp.so"))
dev = torch.device("cuda", 0)
names = ["issue copies + patch loads", "operand reads + MFMAs", "patch transform/store + output update", "wait for copies", "barrier",
         "between phases (loop, setup, epilogue)"]
for cin, cout, t in ((64, 256, 18), (64, 256, 13), (48, 192, 16), (32, 128, 16)):
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    x = torch.randn((n, t, t, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3)) * 0.05
    ww = torch.empty(16 * cin * cout)
    assert lib.swk_winograd_f2x2_3x3_weights(ctypes.c_void_p(w.data_ptr()), cout, cin, ctypes.c_void_p(ww.data_ptr())) == 0
    ww = ww.to(dev)
    bias = torch.zeros(cout, device=dev)
    o = t - 2
    dst = torch.empty((n, o, o, cout), device=dev)
    buf = torch.zeros(512 * 8 * 6, dtype=torch.int64, device=dev)
    assert lib.swk_wino_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(2):
        buf.zero_()
        rc = lib.swk_nhwc_conv3x3_winograd_bias_relu_place(stream, ctypes.c_void_p(x.data_ptr()), n, t, cin, ctypes.c_void_p(ww.data_ptr()),
                                                           ctypes.c_void_p(bias.data_ptr()), cout, ctypes.c_void_p(dst.data_ptr()), o, o, cout, 0, 0, 0)
        assert rc == 0, rc
        torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(-1, 6)
    b = b[b.sum(axis=1) > 0]
    tot = b.sum(axis=1).mean()
    print("%d -> %d, tile %d: %d waves, %.0f stamp ticks per wave" % (cin, cout, t, len(b), tot))
    for i, nm in enumerate(names):
        print("   %-45s %5.1f %%" % (nm, 100.0 * b[:, i].mean() / tot))
