"""Forward of k rows at row0 against the same rows of one big forward: which (k, row0) differ?"""
import os, sys, tempfile, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftwatcher_amd.segment_classification import SegmentClassifier, setup_model    # noqa: E402

rows = 8192
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "w.pt")
    torch.manual_seed(0)
    torch.save(setup_model(2).state_dict(), path)
    clf = SegmentClassifier(path, batch_size=rows)
net = clf.cropped
x = torch.randn((rows, 3, 40, 40), generator=torch.Generator().manual_seed(1)).to(clf.device).contiguous(memory_format=torch.channels_last)
ref = net(x).clone()
again = net(x).clone()
print("repeat equal", torch.equal(ref, again))
for k, row0 in ((2752, 0), (2688, 0), (2752, 2752), (2688, 5504), (1000, 0), (1000, 1000), (96, 0), (33, 0), (1, 0), (4096, 4096), (2048, 6144),
                (2720, 0), (2784, 0), (3072, 0), (2560, 0)):
    got = net(x[row0:row0 + k], row0=row0).clone()
    want = ref[row0:row0 + k]
    bad = (got != want).any(dim=1)
    print(k, row0, "equal" if not bad.any() else "DIFFER rows %s max %.3g" % (bad.nonzero().flatten()[:8].tolist(), (got - want).abs().max().item()))
