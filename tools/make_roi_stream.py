"""Write a ROI stream file (swiftwatcher_amd/io_roi_stream.py) from decoded frames: a .npy array (frames, H, W, 3) uint8 or a headerless
raw file of consecutive H x W x 3 BGR frames, memory-mapped -- only the crop region and its margin are read.

    python tools/make_roi_stream.py frames.npy out.swkroi --corners 790,620,1130,622 [--fps 30]
    python tools/make_roi_stream.py frames.raw out.swkroi --shape 1080,1920,3 --corners 790,620,1130,622

--corners x1,y1,x2,y2 = the chimney's two top corners (the crop region follows from them like in the reference,
image_filtering.py:31-53); --crop x0,y0,x1,y1 gives the crop region directly."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import image_filtering as img                       # noqa: E402
from swiftwatcher_amd.io_frames import RawFileReader                      # noqa: E402
from swiftwatcher_amd.io_roi_stream import RoiStreamWriter                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("source")
ap.add_argument("out")
ap.add_argument("--shape", help="H,W,3 of a raw file's frames")
ap.add_argument("--corners", help="x1,y1,x2,y2")
ap.add_argument("--crop", help="x0,y0,x1,y1")
ap.add_argument("--fps", type=float, default=30.0)
ap.add_argument("--min-seg-size", default="24,24")
a = ap.parse_args()
shape = tuple(int(v) for v in a.shape.split(",")) if a.shape else None
reader = RawFileReader(a.source, frame_shape=shape, fps=a.fps)
if a.crop:
    x0, y0, x1, y1 = (int(v) for v in a.crop.split(","))
    crop_region = [(x0, y0), (x1, y1)]
elif a.corners:
    x1, y1, x2, y2 = (int(v) for v in a.corners.split(","))
    crop_region = img.generate_crop_region([(x1, y1), (x2, y2)])
else:
    raise SystemExit("--corners or --crop is needed")
frames = reader.frames
with RoiStreamWriter(a.out, frames.shape[1:3], crop_region, fps=a.fps, min_seg_size=tuple(int(v) for v in a.min_seg_size.split(",")),
                     channels=1 if frames.ndim == 3 else frames.shape[3]) as w:
    for i in range(frames.shape[0]):
        w.append(frames[i])
print("%d frames, crop region %r, %.1f MB -> %.1f MB" % (frames.shape[0], crop_region, frames.nbytes / 1e6, os.path.getsize(a.out) / 1e6))
