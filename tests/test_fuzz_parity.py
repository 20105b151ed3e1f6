"""A fixed dozen of the random windows of tests/fuzz_parity.py (frames per window, ROI shape, noise kind, duplicated / null frames drawn
from a seed) against the oracle: iteration count, the six stage images, the region records.  The open-ended run is
`python3 tests/fuzz_parity.py <seconds>`; its result for this round is kept in profiles/."""
import pytest


@pytest.mark.gpu
def test_random_windows_match_the_oracle():
    import fuzz_parity
    from swiftwatcher_amd import _lib
    ctx = _lib.Context(0)
    kinds = set()
    for seed in range(100000, 100012):
        cfg, problems = fuzz_parity.check(ctx, seed)
        kinds.add((cfg["kind"], cfg["tail"]))
        assert not problems, "seed %d %r: %s" % (seed, cfg, "; ".join(problems))
    assert len(kinds) >= 5
    ctx.close()
    # the same random windows' second half on a context without the integer start and with the accurate first iteration forced:
    # the float64 start pass and the double-double Gram matrix from the pixels (every staging-tile width occurs among the seeds)
    alt = _lib.Context(0)
    alt.set_integer_start(0)
    alt.set_start_refine(1e-12)
    frames_seen = set()
    for seed in range(100012, 100024):
        cfg, problems = fuzz_parity.check(alt, seed)
        frames_seen.add((cfg["n"] + 15) // 16)
        assert not problems, "seed %d %r (f64 start, refinement forced): %s" % (seed, cfg, "; ".join(problems))
    assert alt.refined_windows[0] >= 6 and len(frames_seen) >= 3
    alt.close()
