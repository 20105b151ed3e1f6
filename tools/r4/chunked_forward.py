"""Would a large forward gain from running depth-first over chunks of rows small enough for the Infinity Cache (256 MB)?
8,192 rows as forwards of R rows each, round-robin over `chains` streams, every chain on its own rows of the persistent tiles (working set =
chains x R x ~0.3 MB per layer pair).  Prints milliseconds per 8,192 rows."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftwatcher_amd.segment_classification import SegmentClassifier, setup_model    # noqa: E402


def main():
    rows = 8192
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.pt")
        torch.manual_seed(0)
        torch.save(setup_model(2).state_dict(), path)
        clf = SegmentClassifier(path, batch_size=rows)
    net = clf.cropped
    dev = clf.device
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((rows, 3, 40, 40), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ref = net(x).clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    out = {}

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    out["whole"] = round(timed(lambda: net(x)), 3)
    for R in (512, 1024, 2048, 4096):
        for chains in (1, 2, 4, 8):
            if R * chains > rows:
                continue
            res = [None] * (rows // R)

            def run():
                cur = torch.cuda.current_stream(dev)
                for s in streams[:chains]:
                    s.wait_stream(cur)
                for i in range(rows // R):
                    c = i % chains
                    with torch.cuda.stream(streams[c]):
                        res[i] = net(x[i * R:(i + 1) * R], row0=c * R)
                for s in streams[:chains]:
                    cur.wait_stream(s)

            ms = timed(run)
            ok = bool(torch.equal(torch.cat(res), ref))
            out["R%d_x%d" % (R, chains)] = [round(ms, 3), ok]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
