"""profiles/pmc_ialm_pass.json from round 4's counters (tools/r4/measure_r4.sh a: FETCH_SIZE and WRITE_SIZE of k_ialm_pass_m<16, 2> in two
separate rocprofv3 --pmc runs of `bench.py --no-classify --steps 1 --warmup 0`).  The read calibration (FETCH_SIZE tallies a 128-B request
as 64 B and is uncalibrated for other widths, MI355X_MICROARCH.md) is round 2's: tools/pass_probe.hip, the pass's exact load / store mix
without arithmetic, known 11 B read per element -> factor 11 / 6.489; the access pattern of the pass has not changed since.
    python tools/r4/pmc_json.py gpurun_out/r4m <commit>"""
import json
import sys

meas, commit = sys.argv[1], sys.argv[2]
summ = json.load(open(meas + "/pmc_summary.json"))
key = [k for k in summ if "k_ialm_pass_m<16, 2>" in k][0]
c = summ[key]
old = json.load(open("profiles/pmc_ialm_pass.json"))
factor = old["fetch_calibration"]["factor"]
W, n, P = 128, 64, 89888
elems = W * n * P
live = 14.0 / 15.0          # 15 dispatches per step, the last finds every window finished (0 bytes)
bench = json.loads([ln for ln in open(meas + "/pmc_FETCH_SIZE.json") if ln.startswith("{")][-1])
K = bench["config"]["ialm_iters_mean"]
alg = (bench["roofline"]["bytes_per_element_iteration"] * K - 10.125) / (K - 1.0)
fetch_kb, write_kb = c["FETCH_SIZE"] / live, c["WRITE_SIZE"] / live
read_b, write_b = fetch_kb * 1024 * factor, write_kb * 1024
out = dict(old)
out.update({
    "measured_at": "round 4, commit %s (tools/r4/measure_r4.sh a); read calibration factor from round 2's probe" % commit,
    "FETCH_SIZE_KB_per_live_dispatch": fetch_kb, "WRITE_SIZE_KB_per_live_dispatch": write_kb,
    "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
    "hbm_bytes_per_window_pass": (read_b + write_b) / W,
    "algorithmic_bytes_per_element_steady_pass": alg, "algorithmic_bytes_per_window_pass": alg * n * P,
    "traffic_over_algorithmic": (read_b + write_b) / (alg * elems),
    "counters_mean_per_dispatch_round2": old.get("counters_mean_per_dispatch"),
    "counters_mean_per_dispatch": {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"]}})
json.dump(out, open("profiles/pmc_ialm_pass.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("measured_at", "hbm_bytes_per_launch", "traffic_over_algorithmic")}, indent=1))
