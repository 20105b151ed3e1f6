"""Does a large classifier forward gain from running as several forwards on disjoint rows, one stream each?
The 1x1 / pool+squeeze kernels sit at half the HBM rate, the Winograd kernels at 70 % of the matrix pipe with 1 TB/s:
side by side they might fill each other's gaps.  Prints milliseconds per 8,192 rows for 1, 2, 3 and 4 streams (equal
shares, and a staggered start for two)."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swiftwatcher_amd.segment_classification import SegmentClassifier, setup_model    # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
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
    out = {"rows": rows}

    def timed(fn, reps=6):
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

    out["one_stream_ms"] = round(timed(lambda: net(x)), 3)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    for parts, stagger in ((2, False), (2, True), (3, False), (4, False)):
        share = -(-rows // parts // 32) * 32
        cuts = [(i * share, min(rows, (i + 1) * share)) for i in range(parts)]
        res = [None] * parts

        def run():
            cur = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event()
            ev.record(cur)
            for i, (lo, hi) in enumerate(cuts):
                s = streams[i]
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    if stagger and i:
                        # half a share first on this stream: the two streams then run different layers at any time
                        mid = (lo + hi) // 2 // 32 * 32
                        ra = net(x[lo:mid], row0=lo)
                        rb = net(x[mid:hi], row0=mid)
                        res[i] = torch.cat([ra, rb])
                    else:
                        res[i] = net(x[lo:hi], row0=lo)
            for s in streams[:parts]:
                cur.wait_stream(s)

        ms = timed(run)
        got = torch.cat(res)
        out["%d_streams%s_ms" % (parts, "_staggered" if stagger else "")] = round(ms, 3)
        out["%d_streams%s_equal" % (parts, "_staggered" if stagger else "")] = bool(torch.equal(got, ref))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
