#!/usr/bin/env python3
"""gpurun_out/prof_round/ (tools/profile_round.sh) -> profiles/r02_*: kernel stats CSVs, traffic_C4.json (HBM bytes per frame
per kernel, VALU / TA busy of k_pool), a text summary of the issue counters."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_round")
dst = os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"

def short(name):
    return name.replace("void prt::", "").replace("prt::", "").split("(")[0]

def counters(d):
    fs = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return agg, n

for tag, name in (("trace", "%s_pool_C4_kernel_stats.csv" % rnd), ("trace_other", "%s_all_workloads_kernel_stats.csv" % rnd)):
    fs = glob.glob(os.path.join(src, tag, "*", "*kernel_stats.csv"))
    if fs:
        shutil.copy(fs[0], os.path.join(dst, name))

fetch, nf = counters("fetch"); write, _ = counters("write")
sq1, n1 = counters("sq1"); sq2, _ = counters("sq2"); ta, nta = counters("ta")
FAST = "k_pool<256, 5, false, true, false, false, false, 0, false, true>"  # the timed frames' kernel (COUNT = false, EXACT = false, SHARED = true: block-shared pools)
import hashlib, subprocess
lib = os.path.join(ROOT, "par_raytracer_amd", "libprt_hip.so")
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE).stdout.decode().strip()
except Exception:
    commit = None
sys.path.insert(0, ROOT)
from par_raytracer_amd import capi
out = {"command": "rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-workloads --steps 5 --warmup 1",
       "round": rnd, "workload": "C4",
       # bench.py prints these figures only while the library it runs is the one they were measured on
       "library": {"sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "abi_version": int(capi.hip_lib().prt_abi_version()),
                   "build_flags": int(capi.hip_lib().prt_build_flags()), "kernel": FAST, "commit": commit},
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB; raw = FETCH + WRITE (lower bound), corrected = 2*FETCH + WRITE (gfx950 FETCH_SIZE halving, upper bound)",
       "kernels": {}}
for k in sorted(fetch):
    if not k.startswith("k_"): continue
    fr = max(1, nf[k]["FETCH_SIZE"])
    f, w = fetch[k]["FETCH_SIZE"] / fr, write.get(k, {}).get("WRITE_SIZE", 0.0) / fr
    out["kernels"][k] = {"launches_in_profile": fr, "fetch_KiB_per_launch": f, "write_KiB_per_launch": w,
                         "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
if FAST in out["kernels"]:
    out["hbm_bytes_per_frame_k_pool"] = out["kernels"][FAST]["hbm_bytes_per_launch_corrected"]
    out["hbm_bytes_per_frame_k_pool_raw"] = out["kernels"][FAST]["hbm_bytes_per_launch_raw"]
if FAST in sq1:
    c = sq1[FAST]
    # SQ_ACTIVE_INST_VALU counts, per SIMD-resident wave, the quad-cycles a vector instruction was in flight; summed over the
    # waves of a SIMD it is that SIMD's vector-pipe busy time.  SQ_BUSY_CYCLES is per SE-level SQ; the ratio below follows
    # (ACTIVE_INST_VALU / WAVE_CYCLES) x resident waves per SIMD.
    out["valu_issue_share_x5_waves_raw_k_pool"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * 5.0, 4)
    out["valu_busy_k_pool"] = min(1.0, out["valu_issue_share_x5_waves_raw_k_pool"])      # the raw ratio can pass 1: fewer than 5 waves per SIMD in the drain tail
    out["wait_frac_k_pool"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    out["sq_k_pool"] = {k2: c[k2] / max(1, n1[FAST][k2]) for k2 in c}
    out["sq_k_pool"].update({k2: sq2[FAST][k2] / max(1, n1[FAST]["SQ_WAVES"]) for k2 in sq2.get(FAST, {})})
if FAST in ta:
    c = ta[FAST]
    per_xcd_active = c["GRBM_GUI_ACTIVE"] / 8.0
    out["ta_busy_k_pool"] = round(c["TA_TA_BUSY_sum"] / 256.0 / per_xcd_active, 4)       # 256 texture addressers (one per compute unit)
    out["ta_busy_avr_k_pool"] = round(c["TA_BUSY_avr"] / per_xcd_active, 4)
tcp, _ = counters("tcp")
if FAST in tcp and tcp[FAST].get("TCP_TA_TCP_STATE_READ_sum"):
    out["l1_latency_clk_k_pool"] = round(tcp[FAST]["TCP_TCP_LATENCY_sum"] / tcp[FAST]["TCP_TA_TCP_STATE_READ_sum"], 1)
json.dump(out, open(os.path.join(dst, "traffic_C4.json"), "w"), indent=1)
with open(os.path.join(dst, "%s_pool_C4_issue_stalls.txt" % rnd), "w") as f:
    f.write("k_pool (fast kernel, one launch per C4 frame), rocprofv3 --pmc passes of bench.py (tools/profile_round.sh)\n")
    for k in (FAST,):
        for name, d in (("sq1", sq1), ("sq2", sq2), ("ta", ta)):
            if k in d:
                f.write("%s: %s\n" % (name, json.dumps({a: "%.4g" % b for a, b in d[k].items()})))
    if "valu_busy_k_pool" in out:
        c = sq1[FAST]
        f.write("wave cycles: %.1f%% waiting (s_waitcnt), %.1f%% issue stalls, %.1f%% issuing, of which vector %.1f%% of the wave's cycles\n" % (
            100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
        f.write("vector pipe busy (x5 waves per SIMD): %.1f%%;  texture addresser busy: %s (TA_TA_BUSY_sum / 256 / per-XCD GRBM_GUI_ACTIVE), %s (TA_BUSY_avr)\n" % (
            100 * out["valu_busy_k_pool"], out.get("ta_busy_k_pool"), out.get("ta_busy_avr_k_pool")))
print(json.dumps({k: out[k] for k in out if k not in ("kernels", "sq_k_pool")}, indent=1))
for k, v in out["kernels"].items():
    print("%-90s x%d  raw %.2f GB  corrected %.2f GB" % (k[:90], v["launches_in_profile"], v["hbm_bytes_per_launch_raw"] / 1e9, v["hbm_bytes_per_launch_corrected"] / 1e9))
