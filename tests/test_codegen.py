"""Code-generation guard for the hot kernel (no GPU needed: hipcc cross-compiles).  The IALM pass k_ialm_pass_m<NK, MODE>
(csrc/ialm_mstate.hip) is bound by the f64 execution unit and lives on two waves per SIMD at 64 frames and four at the CLI's
queue of 21: a small source change can cost a register tier (or spill), duplicate matrix instructions, or flip a tile's loads to
load-wait pairs (+42 % per launch in round 1).  Checked on the instantiations the benchmark configurations
run: NK = 16 k-steps (64 frames) and NK = 6 (21 frames), steady passes (MODE 2) and first-iteration passes (MODE 1)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def pass_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    from swiftwatcher_amd.csrc import build
    out = tmp_path_factory.mktemp("asm") / "ialm_mstate.s"
    src = os.path.join(ROOT, "swiftwatcher_amd", "csrc", "ialm_mstate.hip")
    flags = [f for f in build.FLAGS if f not in ("-Wall",)]          # the library's own flags
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", src, "-o", str(out)])
    return open(out).read().splitlines()


def _kernel(lines, nk, mode):
    sym = "_ZN3swk13k_ialm_pass_mILi%dELi%dEEEvNS_11IalmBuffersEii:" % (nk, mode)
    start = next(i for i, l in enumerate(lines) if l.startswith(sym))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta = "\n".join(lines[end:end + 120])
    return lines[start:end], meta


# copies: the per-window switches of a pass (sparse-image stores, U read, U written) select one of eight specialised tile loops in a
# steady pass; the first-iteration pass has two (sparse-image stores on / off)
@pytest.mark.parametrize("nk,mode,copies,waves_per_simd", [(16, 2, 8, 2), (6, 2, 8, 4), (16, 1, 2, 2), (6, 1, 2, 4)])
def test_pass_keeps_its_registers_and_its_loads_in_flight(pass_asm, nk, mode, copies, waves_per_simd):
    body, meta = _kernel(pass_asm, nk, mode)
    count = lambda pat: sum(1 for l in body if re.search(pat, l))
    nb = (nk + 3) // 4
    assert count(r"scratch_") == 0, "register spills in the streaming pass"
    assert int(re.search(r"NumVgprs: (\d+)", meta).group(1)) + int(re.search(r"NumAgprs: (\d+)", meta).group(1)) <= 512 // waves_per_simd
    assert int(re.search(r"Occupancy: (\d+)", meta).group(1)) >= waves_per_simd
    # per 16-pixel tile: the A update (NK k-steps x NB out-frame blocks) + the symmetric Gram (NB (NB + 1) / 2 block pairs x 4 k-steps)
    assert count(r"v_mfma_f64_16x16x4") == copies * (nk * nb + 2 * nb * (nb + 1))
    # a tile's loads are issued together: at most two full waits per copy of the loop (tile end, prologue / epilogue)
    # (the first-iteration pass, 1 launch in 16, also waits for its table of X / dual_norm)
    # (+ 4 since round 4: the joins behind the loop copies that carry the largest |U| read -- the stopping norm's error bound -- out of
    #  the loop wait once each, after the loop)
    assert count(r"vmcnt\(0\)") <= (2 * copies + 4 if mode == 2 else 8), "the compiler serialised the tile's loads"
    assert count(r"buffer_load") >= copies * nk            # M of every k-step at least, per copy
