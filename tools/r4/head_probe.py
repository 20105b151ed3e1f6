"""Time swk_nhwc_head2_relu_mean alone on a (rows, 512, 9, 9) tensor -- a pure streaming read of rows x 166 KB -- behind different
predecessors on the same stream: nothing, a kernel that has just written its input, one that has written another buffer of that size,
a matrix-core-heavy product.  (Inside a forward the same kernel takes 360 us for 8,192 rows; alone 220.)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import _lib    # noqa: E402
lib = _lib.load()
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.randn((rows, 512, 9, 9), device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randn((rows, 512, 9, 9), device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn((2, 512), device=dev) * 0.05
b = torch.zeros(2, device=dev); ring = torch.zeros(2, device=dev)
out = torch.empty((rows, 2), device=dev)
m1 = torch.randn((8192, 8192), device=dev); m2 = torch.randn((8192, 8192), device=dev); m3 = torch.empty((8192, 8192), device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def head():
    assert lib.swk_nhwc_head2_relu_mean(st, x.data_ptr(), rows, 81, 512, w.data_ptr(), b.data_ptr(), ring.data_ptr(), 169.0, out.data_ptr()) == 0
pre = {"nothing": lambda: None, "wrote_its_input": lambda: x.mul_(1.0), "wrote_another_buffer": lambda: y.mul_(1.0),
       "read_another_buffer": lambda: y.sum(), "matrix_product": lambda: torch.mm(m1, m2, out=m3)}
for name, fn in pre.items():
    for _ in range(2):
        fn(); head()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(8):
        fn()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); head(); e.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(e)
    ms = tot / 8
    print("var %s rows %d after %-22s %.1f us, %.2f TB/s" % (os.environ.get("SWK_HEAD_VAR", "0"), rows, name, ms * 1e3, rows * 81 * 512 * 4 / ms / 1e9))
