import cProfile, pstats, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import pipeline, synthetic
crop_region = [(748, 452), (1172, 664)]
queue, n_windows = 21, 24
total = n_windows * queue
clip = synthetic.full_frames(5, total, crop_region, birds=12)[::-1]
frames = [clip[i] for i in range(total)]
roi_mask = np.zeros((212, 424), np.uint8); roi_mask[100:, :] = 255
pipeline.count_swifts(frames[:42], crop_region, roi_mask, windows_per_call=24)
pr = cProfile.Profile(); pr.enable()
pipeline.count_swifts(frames, crop_region, roi_mask, windows_per_call=24)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
