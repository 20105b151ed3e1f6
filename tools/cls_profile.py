"""Steady-state kernel breakdown of the cropped classifier forward (diagnostic; torch.profiler)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd.segment_classification import SegmentClassifier, SqueezeNet10
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
clf = SegmentClassifier.from_state_dict(SqueezeNet10(2).state_dict(), batch_size=2048)
x = torch.randn(2048, 3, 40, 40, device="cuda")
for _ in range(5):
    clf.cropped(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        clf.cropped(x)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=90))
