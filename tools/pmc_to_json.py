"""profiles/pmc_ialm_pass.json from a measurement set:
HBM traffic of one launch of the steady-state IALM pass = FETCH_SIZE x calibration + WRITE_SIZE.

MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies a 128-B request as 64 B; "other access widths are uncalibrated:
calibrate on a known byte count in your own access pattern".  The pass mixes 128-B (f64), 64-B (f32) and 16-B (u8)
row segments, so the read factor is taken from tools/pass_probe (shape 0: exactly the pass's loads and stores, no
arithmetic, known 11 B read per element): factor = 11 / counted.  WRITE_SIZE is used as counted (it reports the
partial-line writes of the u8 plane at their burst size: that is traffic)."""
import json, re, sys

meas = sys.argv[1]
summ = json.load(open(meas + "/pmc_summary.json"))
key = [k for k in summ if "k_ialm_pass_m<16, 2>" in k][0]
c = summ[key]
probe = open(meas + "/pass_probe.txt").read()
counted = float(re.search(r"k_probe<0>.*FETCH_SIZE counted B/elt ([0-9.]+)", probe).group(1))
counted_w = float(re.search(r"k_probe<0>.*WRITE_SIZE counted B/elt ([0-9.]+)", probe).group(1))
factor = 11.0 / counted
W, n, P = 128, 64, 89888
elems = W * n * P
# the per-dispatch means include the lag launch that finds every window finished (0 bytes): 15 dispatches, 14 live
live = 14.0 / 15.0
bench = json.loads([l for l in open(meas + "/bench_segment.log") if l.startswith("{")][-1])
K = bench["config"]["ialm_iters_mean"]
alg = (bench["roofline"]["bytes_per_element_iteration"] * K - 10.125) / (K - 1.0)
fetch_kb, write_kb = c["FETCH_SIZE"] / live, c["WRITE_SIZE"] / live
read_b, write_b = fetch_kb * 1024 * factor, write_kb * 1024
out = {
    "kernel": key.replace("void swk::", ""), "variant": 3, "windows_per_dispatch": W, "n": n, "P": P,
    "FETCH_SIZE_KB_per_live_dispatch": fetch_kb, "WRITE_SIZE_KB_per_live_dispatch": write_kb,
    "fetch_calibration": {"probe": "tools/pass_probe shape 0 (same loads/stores, known 11 B read + 11 B written per element)",
                          "counted_read_B_per_element": counted, "factor": factor,
                          "counted_write_B_per_element": counted_w},
    "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": read_b + write_b,
    "hbm_bytes_per_window_pass": (read_b + write_b) / W,
    "algorithmic_bytes_per_element_steady_pass": alg,
    "algorithmic_bytes_per_window_pass": alg * n * P,
    "note": "algorithmic bytes: what the library books per window-iteration (swk_prof_pass_bytes_per_element: X 1 + M 8+8 + U 2 or 1/8 each way + 1 when the sparse image is stored), averaged over the steady-state passes of the same bench run (first pass = 10.125 B taken out)",
    "traffic_over_algorithmic": (read_b + write_b) / (alg * elems),
    "counters_mean_per_dispatch": c,
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "traffic_over_algorithmic", "fetch_calibration")}, indent=1))
