"""Per-layer steady-state time of the receptive-field cropped classifier (diagnostic)."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd.segment_classification import SegmentClassifier, SqueezeNet10

torch.manual_seed(0)
clf = SegmentClassifier.from_state_dict(SqueezeNet10(2).state_dict(), batch_size=2048)
c = clf.cropped
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.randn(B, 3, 40, 40, device="cuda")
m = c.model
F = torch.nn.functional

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

rows = []
with torch.no_grad():
    t, y = timed(lambda: torch.relu(m.features[0](x))); rows.append(("conv1+relu", t, tuple(y.shape)))
    a, b = c.pool1_slice
    t, y = timed(lambda: m.features[2](y[:, :, a:b, a:b])); rows.append(("pool1", t, tuple(y.shape)))
    cur = y
    for kind, layer, tile, off, n, pad, crop in c.plan:
        def paste():
            tt = tile.expand(cur.shape[0], -1, -1, -1).clone()
            tt[:, :, off:off + n, off:off + n] = cur
            return tt
        t, tt = timed(paste); rows.append((kind + " paste", t, tuple(tt.shape)))
        if kind == "pool":
            t, cur = timed(lambda: layer(tt)); rows.append(("pool", t, tuple(cur.shape)))
            continue
        t, sq = timed(lambda: layer.squeeze_activation(layer.squeeze(tt))); rows.append(("squeeze", t, tuple(sq.shape)))
        t, e3 = timed(lambda: torch.relu(F.conv2d(sq, layer.expand3x3.weight, layer.expand3x3.bias))); rows.append(("expand3x3", t, tuple(e3.shape)))
        cc, cn = crop
        t, e1 = timed(lambda: torch.relu(layer.expand1x1(sq[:, :, cc:cc + cn, cc:cc + cn]))); rows.append(("expand1x1", t, tuple(e1.shape)))
        t, cur = timed(lambda: torch.cat([e1, e3], 1)); rows.append(("cat", t, tuple(cur.shape)))
    t, s = timed(lambda: torch.relu(m.classifier[1](cur)).sum(dim=(2, 3))); rows.append(("head", t, tuple(s.shape)))
    t, _ = timed(lambda: c(x)); rows.append(("TOTAL forward", t, ()))
for r in rows: print("%-16s %8.3f ms  %s" % r)
print("sum of parts %.3f ms" % sum(r[1] for r in rows[:-1]))
