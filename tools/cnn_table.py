"""Markdown table "where the classifier's time goes" from a rocprofv3 kernel-stats CSV of the default bench and its JSON line.
    python3 tools/cnn_table.py profiles/r2_kernel_stats_default_bench.csv profiles/r2_bench_lines/default.json"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
line = json.load(open(sys.argv[2]))
cnn = line["roofline_cnn"]
rows_per_forward = cnn["rows_per_step"] / cnn["forwards_per_step"]          # the last forward of a step is a partial one


def total(pred):
    return sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / 1e6, sum(int(r["Calls"]) for r in rows if pred(r["Name"]))


def per_forward(pred, fwd):
    """average duration x launches per forward, kernel by kernel (MIOpen's find phase launches a few extra)"""
    return sum(float(r["AverageNs"]) / 1e6 * round(int(r["Calls"]) / fwd) for r in rows if pred(r["Name"]) and int(r["Calls"]) >= fwd)


fwd = total(lambda n: "k_segment_inputs" in n)[1]
groups = [
    ("`k_wino3x3_relu_place<8,…>` (fire8, fire9 expand3x3: 64 → 256, Winograd)", lambda n: "k_wino3x3_relu_place<8" in n, 55.5e6),
    ("`k_wino3x3_relu_place<6,…>` (fire6, fire7: 48 → 192)", lambda n: "k_wino3x3_relu_place<6" in n, 28.2e6),
    ("`k_wino3x3_relu_place<4,…>` (fire4, fire5: 32 → 128)", lambda n: "k_wino3x3_relu_place<4" in n, 10.9e6),
    ("`k_wino3x3_relu_place<2,…>` (fire2, fire3: 16 → 64)", lambda n: "k_wino3x3_relu_place<2" in n, 2.25e6),
    ("`k_conv1x1_relu_place<…>` (8 squeeze + 8 expand1x1, the latter on the segment-dependent pixels)", lambda n: "k_conv1x1_relu_place" in n, 20.5e6),
    ("`k_maxpool3s2` × 3", lambda n: "k_maxpool3s2" in n, 0),
    ("`k_conv7x7s2_relu` (conv1: 7×7, stride 2, 3 → 96, + bias + ReLU)", lambda n: "k_conv7x7s2_relu" in n, 4.08e6),
    ("head (512 → 2, CK kernel)", lambda n: "kernel_grouped_conv_fwd_xdl_cshuffle" in n, 0.12e6),
]
print("| kernel(s) | per forward (%d rows on average) | per step (×%d) | direct-convolution MACs per segment | rate (direct-equivalent) |" % (round(rows_per_forward), cnn["forwards_per_step"]))
print("|---|---|---|---|---|")
s = 0.0
for name, pred, macs in groups:
    per = per_forward(pred, fwd)
    s += per
    rate = "%.0f TFLOP/s" % (2 * macs * rows_per_forward / (per * 1e-3) / 1e12) if macs and per > 0 else ""
    print("| %s | %.2f ms | %.1f ms | %s | %s |" % (name, per, per * cnn["forwards_per_step"], ("%.1f M" % (macs / 1e6)) if macs else "—", rate))
ev = cnn["net_ms_per_step"] / cnn["forwards_per_step"]
print("| remainder: ReLU / sum / decision of the head (PyTorch element-wise kernels), gaps | ≈%.2f ms | ≈%.1f ms | | |" % (max(ev - s, 0.0), max(ev - s, 0.0) * cnn["forwards_per_step"]))
print("| **sum** (HIP events around the forwards) | **%.2f ms** | **%.1f ms** | %.1f M useful, %.1f M executed | %.1f TFLOP/s executed = **%.2f of 157.3**; %.1f direct-equivalent |"
      % (ev, cnn["net_ms_per_step"], cnn["macs_per_segment_useful"] / 1e6, cnn["macs_per_segment_executed"] / 1e6, cnn["achieved"], cnn["frac"],
         cnn["direct_equivalent"]["achieved"]))
