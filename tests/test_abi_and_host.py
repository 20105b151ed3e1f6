"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol
include/swk.h declares (no compute calls without a GPU), struct layouts match, host logic
(crop geometry, segment boxes, FrameQueue bookkeeping) behaves like the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from swiftwatcher_amd.csrc import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    declared = set()
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):          # swk.h (the boundary) and swk_debug.h (switches, counters)
        assert name.endswith(".h")
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        here = set(re.findall(r"\b(swk_[a-z0-9_]+)\s*\(", hdr))
        assert len(here) >= 15 and not (here & declared)
        if name == "swk.h":          # the boundary carries no A/B switch, profiling hook or diagnostic
            assert not [n for n in here if n.startswith(("swk_set_", "swk_prof_")) or (n.startswith("swk_last_") and n != "swk_last_error")]
        declared |= here
    lib = ctypes.CDLL(built_lib)
    for name in sorted(declared):
        assert hasattr(lib, name), "libswk.so does not export %s" % name
    from swiftwatcher_amd import _lib
    assert set(_lib.EXPORTS) == declared
    assert _lib.load().swk_abi_version() == _lib.ABI_VERSION == 2


def test_every_entry_point_is_mapped_to_the_reference_in_the_integration_notes():
    """INTEGRATION.md section 4 is the maintainer's table: each C symbol next to the reference function it replaces."""
    from swiftwatcher_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _lib.EXPORTS if n not in doc and not (n.startswith("swk_prof_") and "swk_prof_*" in doc)]
    assert not missing, missing


def test_struct_layouts_and_defaults(built_lib):
    import subprocess
    import tempfile
    from swiftwatcher_amd import _lib
    # sizes as the C compiler lays out include/swk.h
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include <stdio.h>\n#include "swk.h"\nint main(){printf("%zu %zu %zu %zu", '
                             'sizeof(swk_params), sizeof(swk_input), sizeof(swk_output), sizeof(swk_segment));}')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "sz")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "sz")]).split()]
    assert sizes == [ctypes.sizeof(_lib.Params), ctypes.sizeof(_lib.Input), ctypes.sizeof(_lib.Output),
                     ctypes.sizeof(_lib.Segment)]
    assert ctypes.sizeof(_lib.Segment) == 48
    p = _lib.default_params()
    # the reference's hard-coded literals
    assert (p.lmbda, p.tol, p.maxiter) == (0.01, 0.001, 100)
    assert (p.bil_d, p.bil_sigma_color, p.bil_sigma_space) == (7, 15.0, 1.0)
    assert (p.thresh, p.open_kh, p.open_kw) == (15, 3, 3)
    assert (p.connectivity, p.label_order, p.gray_mode) == (8, _lib.ORDER_BLOCK2X2, _lib.GRAY_Q14)
    with pytest.raises(TypeError):
        _lib.default_params(nonsense=1)


def test_no_gpu_means_loud_failure(built_lib):
    """There is no CPU fallback: without a gfx950 device a context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from swiftwatcher_amd import _lib
    with pytest.raises(_lib.SwkError):
        _lib.Context(0)
    from swiftwatcher_amd import image_filtering as img
    with pytest.raises(_lib.SwkError):
        img.thresh_to_zero(np.zeros((4, 4), np.uint8), 15)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "swiftwatcher_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the CPU oracle", ""), "%s mentions the oracle" % f


def test_crop_geometry_golden(golden_dir):
    from swiftwatcher_amd import image_filtering as img
    g = np.load(os.path.join(golden_dir, "crop_regions.npz"))
    frame = np.random.default_rng(int(g["frame_seed"])).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    for corners, region, ext, s in zip(g["corners"], g["regions"], g["extents"], g["crop_sums"]):
        c = [tuple(int(v) for v in corners[0]), tuple(int(v) for v in corners[1])]
        assert img.determine_chimney_extents(c) == tuple(int(v) for v in ext)
        got = img.generate_crop_region(c)
        assert [tuple(p) for p in got] == [tuple(int(v) for v in p) for p in region]
        assert int(img.crop_frame(frame, got).astype(np.int64).sum()) == int(s)


def test_extract_segment_images_golden(golden_dir):
    from swiftwatcher_amd import image_filtering as img
    g = np.load(os.path.join(golden_dir, "segment_crops.npz"))
    frame = np.random.default_rng(int(g["frame_seed"])).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    cr = [tuple(int(v) for v in g["crop_region"][0]), tuple(int(v) for v in g["crop_region"][1])]
    segs = [img.RegionProps(1, tuple(int(v) for v in b), (0.0, 0.0), 1) for b in g["bboxes"]]
    ims = img.extract_segment_images(segs, frame, (24, 24), cr)
    for im, shp, s, fp, lp in zip(ims, g["shapes"], g["sums"], g["first_px"], g["last_px"]):
        assert im.shape == tuple(shp) and int(im.astype(np.int64).sum()) == int(s)
        np.testing.assert_array_equal(im[0, 0], fp)
        np.testing.assert_array_equal(im[-1, -1], lp)


def test_framequeue_bookkeeping():
    from swiftwatcher_amd.data_structures import FrameQueue, Frame
    q = FrameQueue(queue_size=5)
    assert q.maxlen == 5 and q.is_empty()
    frames = [np.full((8, 8, 3), i, np.uint8) for i in range(5)]
    q.push_list_of_frames(frames, [0, 1, 2, -1, -1], ["a", "b", "c", "00:00:00.000", "00:00:00.000"])
    assert q.frames_read == 5 and q[0].null and not q[4].null      # newest (left) = last pushed
    assert [f.frame_number for f in q] == [-1, -1, 2, 1, 0]
    q.preprocess_queue([(2, 1), (6, 5)], None)
    assert q[0].processed_frames["crop"].shape == (4, 4, 3)
    assert list(q[0].processed_frames.keys()) == ["crop", "grayscale"]
    first = q.pop_frame()
    assert first.frame_number == 0 and q.frames_processed == 1
    q.pop_frame(); q.pop_frame()
    assert q.frames_processed == 3
    q.pop_frame(); q.pop_frame()
    assert q.frames_processed == 3 and q.is_empty()               # null frames are not counted
    assert Frame().null


def test_array_reader_mirrors_framereader_bookkeeping():
    from swiftwatcher_amd.io_frames import ArrayReader
    frames = [np.full((4, 5, 3), i, np.uint8) for i in range(5)]
    r = ArrayReader(frames, fps=10.0)
    assert (r.start_frame, r.end_frame, r.total_frames) == (0, 5, 5)
    got, nums, stamps = r.get_n_frames(8)
    assert nums == [0, 1, 2, 3, 4, 5, -1, -1]            # frame "5" is inside the inclusive range test ...
    assert int(got[5][0, 0, 0]) == 4 and r.read_errors == 1   # ... and is served by the last good frame
    assert not got[6].any() and got[6].shape == (4, 5, 3) and stamps[6] == "00:00:00.000"
    assert (stamps[3] - stamps[0]).total_seconds() == pytest.approx(0.3)
    assert r.frames_read == 5


def test_raw_file_reader_has_the_frame_reader_semantics(tmp_path):
    """Memory-mapped frame files behind the reference FrameReader's bookkeeping (io_video.py:13-82): frames, numbers,
    the re-delivered last frame, null frames past the end."""
    from swiftwatcher_amd.io_frames import ArrayReader, RawFileReader
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(5, 6, 8, 3), dtype=np.uint8)
    raw = tmp_path / "clip.bgr"
    frames.tofile(raw)
    npy = tmp_path / "clip.npy"
    np.save(npy, frames)
    ref = ArrayReader(list(frames))
    want = ref.get_n_frames(8)
    for reader in (RawFileReader(str(raw), frame_shape=(6, 8, 3)), RawFileReader(str(npy))):
        assert reader.total_frames == 5
        got = reader.get_n_frames(8)
        assert got[1] == want[1] == [0, 1, 2, 3, 4, 5, -1, -1]
        for a, b in zip(got[0], want[0]):
            np.testing.assert_array_equal(a, b)
        assert reader.read_errors == ref.read_errors == 1
    with pytest.raises(ValueError):
        RawFileReader(str(raw))


@pytest.mark.gpu
def test_library_before_torch_shares_one_hip_runtime():
    """A process that creates a library context BEFORE importing torch must still get a working PyTorch-ROCm (one HIP
    runtime in the process, not the system copy plus torch's bundled one), in a fresh interpreter."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from swiftwatcher_amd import _lib\n"
        "assert 'torch' not in sys.modules\n"
        "ctx = _lib.Context(0)\n"
        "out = ctx.thresh_tozero_u8(np.arange(32, dtype=np.uint8), 15)\n"
        "import torch\n"
        "assert torch.cuda.is_available(), 'torch lost the GPU: two HIP runtimes in one process'\n"
        "x = torch.arange(8, device='cuda').float().sum().item()\n"
        "out2 = ctx.thresh_tozero_u8(np.arange(32, dtype=np.uint8), 15)\n"
        "assert x == 28.0 and out[16] == 16 and (out == out2).all()\n"
        "print('one runtime ok')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "one runtime ok" in p.stdout, p.stdout + p.stderr


def test_host_staging_helpers(built_lib):
    """swk_stage_frames / swk_cut_boxes (host code of the library, no GPU): the window's crops and a window's segment images are
    plain copies -- equal to numpy slicing for every layout they accept, numpy itself for the layouts they do not."""
    import ctypes
    from swiftwatcher_amd import _lib
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, size=(90, 130, 3), dtype=np.uint8) for _ in range(21)]
    dst = np.empty((21, 40, 50, 3), np.uint8)
    _lib.stage_frames(frames, 11, 51, 60, 110, dst)
    np.testing.assert_array_equal(dst, np.stack([f[11:51, 60:110] for f in frames]))
    big = [rng.integers(0, 256, size=(400, 700, 3), dtype=np.uint8) for _ in range(9)]          # > 1 MB: the thread pool's path
    dst = np.empty((9, 300, 500, 3), np.uint8)
    _lib.stage_frames(big, 50, 350, 100, 600, dst)
    np.testing.assert_array_equal(dst, np.stack([f[50:350, 100:600] for f in big]))
    gray = [np.ascontiguousarray(f[:, :, 1]) for f in frames]
    dst = np.empty((21, 40, 50), np.uint8)
    _lib.stage_frames(gray, 11, 51, 60, 110, dst)
    np.testing.assert_array_equal(dst, np.stack([f[11:51, 60:110] for f in gray]))
    strided = [f[:, ::2] for f in frames]                                                          # not row-contiguous: numpy's copy
    dst = np.empty((21, 40, 20, 3), np.uint8)
    _lib.stage_frames(strided, 11, 51, 5, 25, dst)
    np.testing.assert_array_equal(dst, np.stack([f[11:51, 5:25] for f in strided]))
    lib = _lib.load()
    ptrs = (ctypes.c_void_p * 2)(frames[0].ctypes.data, None)
    assert lib.swk_stage_frames(ptrs, 2, 390, 0, 4, 0, 30, dst.ctypes.data, 1) != 0               # a null frame pointer is refused
    # segment boxes of several frames into one buffer
    boxes = np.array([[0, 24, 0, 24], [10, 50, 20, 44], [80, 90, 100, 130], [5, 5, 7, 9], [30, 31, 0, 130]], np.int32)
    frame_of = np.array([0, 3, 3, 7, 20], np.int32)
    buf, offs = _lib.cut_boxes(frames, frame_of, boxes)
    at = 0
    for (r0, r1, c0, c1), f, o in zip(boxes.tolist(), frame_of.tolist(), offs.tolist()):
        want = frames[f][r0:r1, c0:c1]
        assert o == at
        np.testing.assert_array_equal(buf[o:o + want.size].reshape(want.shape), want)
        at += want.size
    assert at == buf.size
    bad = boxes.copy()
    bad[1, 0] = -3
    with pytest.raises(_lib.SwkError):
        _lib.cut_boxes(frames, frame_of, bad)
    empty_buf, empty_offs = _lib.cut_boxes(frames, np.zeros(0, np.int32), np.zeros((0, 4), np.int32))
    assert empty_buf.size == 0 and empty_offs.size == 0


def test_a_batch_learns_its_own_generation_under_the_context_lock():
    """Two threads run batches on one Context (a reader that segments ahead beside the loop's own segment_queue): each must get the
    generation of ITS batch, or swk_segment_inputs_last is later asked about the other thread's frames (seen once on the GPU box as
    '*total does not match the batch').  The library call is replaced by a slow stub; no GPU."""
    import threading
    import time
    from swiftwatcher_amd import _lib

    class SlowLib:
        def __init__(self):
            self.running = 0
            self.overlap = False

        def swk_batch_run(self, *args):
            self.running += 1
            self.overlap |= self.running > 1
            time.sleep(0.05)
            self.running -= 1
            return 0

    ctx = object.__new__(_lib.Context)
    ctx._lock = threading.RLock()
    ctx._lib = SlowLib()
    ctx._h = None
    ctx.generation = 0
    got = []

    def run():
        g = ctx.batch_run_raw(_lib.Input(), _lib.Params(), _lib.Output())
        got.append((g, ctx.generation))          # (the attribute may already belong to another thread's batch)

    threads = [threading.Thread(target=run) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(g for g, _ in got) == [1, 2, 3, 4] and not ctx._lib.overlap
    ctx._h = None


def test_export_segments_writes_the_reference_files(tmp_path):
    """Frame.export_segments (data_structures.py:65-113, `--export`): file names, the tinted overlay (box corners inclusive, 0.6 / 0.4
    blend) and the >= 24 x 24 cut from the FULL frame, read back from the PNGs (Pillow; cv2 is installed nowhere: PARITY UNPINNED)."""
    from PIL import Image
    from swiftwatcher_amd.data_structures import Frame, Segment
    from swiftwatcher_amd.image_filtering import RegionProps
    rng = np.random.default_rng(5)
    full = rng.integers(0, 256, size=(120, 200, 3), dtype=np.uint8)
    crop_region = [(40, 30), (160, 100)]
    fr = Frame(full, 17, "00:00:00.567")
    Frame.src_video = "clip 1"
    try:
        fr.processed_frames["crop"] = full[30:100, 40:160]
        fr.segments = [Segment(RegionProps(1, (10, 20, 16, 30), (12.5, 24.0), 40), 17, "t", None),
                       Segment(RegionProps(2, (50, 5, 69, 47), (60.0, 25.0), 500), 17, "t", None)]
        fr.export_segments((24, 24), crop_region, tmp_path / "segments")
    finally:
        Frame.src_video = None
    names = sorted(p.name for p in (tmp_path / "segments").glob("*.png"))
    assert names == ['"clip 1"_17_1_2.png', '"clip 1"_17_2_2.png']
    assert sorted(p.name for p in (tmp_path / "segments" / "overlay").glob("*.png")) == names
    crop = full[30:100, 40:160].astype(np.float32)
    ov = np.asarray(Image.open(tmp_path / "segments" / "overlay" / names[0]))[..., ::-1]          # back to BGR
    exp = crop.copy()
    box = np.zeros(crop.shape[:2], bool)
    box[10:17, 20:31] = True
    exp[box] = np.rint(0.6 * np.array([0, 0, 255], np.float32) + 0.4 * crop[box])
    np.testing.assert_array_equal(ov, exp.astype(np.uint8))
    seg = np.asarray(Image.open(tmp_path / "segments" / names[0]))[..., ::-1]
    np.testing.assert_array_equal(seg, full[30 + 1:30 + 25, 40 + 13:40 + 37])          # 6 x 10 box grown to 24 x 24, cut from the full frame
    seg2 = np.asarray(Image.open(tmp_path / "segments" / names[1]))[..., ::-1]
    np.testing.assert_array_equal(seg2, full[30 + 48:30 + 72, 40 + 5:40 + 47])          # 19 rows grown by 2 + 3, 42 columns kept
