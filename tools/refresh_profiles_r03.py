#!/usr/bin/env python3
"""Copies the summaries of the last tools/gpu_profile_r03.sh run (gpurun_out/r03/prof/), the last default bench line and the
kernel timelines from gpurun_out/ into profiles/r03/ and recomputes pmc_traffic.json's figures from the new counters."""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
new = json.load(open(R + "gpurun_out/r03/prof/summary_pmc.json"))


def per(run, kern, ctr):
    for k, v in new[run].items():
        if k.startswith(kern) and ctr in v:
            return v[ctr]["per_launch"]
    return None


old = json.load(open(R + "profiles/r03/pmc_traffic.json"))
KiB = 1024.0
c = old["commit_n4096_c22"]
c["FETCH_SIZE_bytes"] = per("pmc_commit_fetch", "k_msm_comb28", "FETCH_SIZE") * KiB
c["WRITE_SIZE_bytes"] = per("pmc_commit_write", "k_msm_comb28", "WRITE_SIZE") * KiB
c["hbm_bytes_per_launch"] = c["FETCH_SIZE_bytes"] + c["WRITE_SIZE_bytes"]
c["over_algorithmic"] = c["hbm_bytes_per_launch"] / c["algorithmic_bytes_per_launch"]
v = old["verify_n65536_c22"]
v["FETCH_SIZE_bytes_raw"] = per("pmc_verify_fetch", "k_challenge", "FETCH_SIZE") * KiB
v["FETCH_SIZE_bytes_corrected_x2"] = 2 * v["FETCH_SIZE_bytes_raw"]
v["WRITE_SIZE_bytes"] = per("pmc_verify_write", "k_challenge", "WRITE_SIZE") * KiB
v["hbm_bytes_per_launch"] = v["FETCH_SIZE_bytes_corrected_x2"] + v["WRITE_SIZE_bytes"]
ev = 2 * per("pmc_verify_fetch", "k_eval_frac", "FETCH_SIZE") * KiB + per("pmc_verify_write", "k_eval_frac", "WRITE_SIZE") * KiB
dc = per("pmc_verify_fetch", "k_g1_decompress_range", "FETCH_SIZE") * KiB + per("pmc_verify_write", "k_g1_decompress_range", "WRITE_SIZE") * KiB
v["other_kernels_of_the_call"] = {"k_eval_frac<16>": {"hbm_bytes": ev}, "k_g1_decompress_range": {"hbm_bytes": dc}}
v["call_total_hbm_bytes"] = v["hbm_bytes_per_launch"] + ev + dc
v["call_over_algorithmic"] = v["call_total_hbm_bytes"] / v["algorithmic_bytes_per_launch"]
json.dump(old, open(R + "profiles/r03/pmc_traffic.json", "w"), indent=1)
shutil.copy(R + "gpurun_out/r03/prof/summary_pmc.json", R + "profiles/r03/pmc_counters_per_kernel.json")
for a, b in (("default", "bench_default_kernel_stats.csv"), ("commit", "trace_commit_kernel_stats.csv"), ("proof", "trace_proof4096_kernel_stats.csv"),
             ("verify", "trace_verify65536_kernel_stats.csv")):
    shutil.copy(R + "gpurun_out/r03/prof/summary_trace_%s_kernel_stats.csv" % a, R + "profiles/r03/" + b)
for src, dst in (("bench_default.json", "bench_default.json"), ("single_commit_timeline.txt", "single_commit_kernel_timeline.txt"),
                 ("verify_timeline.txt", "verify65536_kernel_timeline.txt")):
    if os.path.exists(R + "gpurun_out/" + src):
        shutil.copy(R + "gpurun_out/" + src, R + "profiles/r03/" + dst)
for w in ("default", "commit", "proof", "verify"):
    line = [l for l in open(R + "gpurun_out/r03/prof/trace_%s.log" % w) if l.startswith('{"metric"')][0]
    open(R + "profiles/r03/bench_%s_under_rocprofv3.json" % w, "w").write(line)
    d = json.loads(line)
    print(w, "under rocprofv3:", round(d["value"]), "blobs/s", round(d["ms_per_step"], 3), "ms")
d = json.load(open(R + "profiles/r03/bench_default.json"))
print("default:", round(d["value"]), round(d["ms_per_step"], 3), "frac", d["roofline"]["frac"], [(round(m["value"]), m.get("roofline_frac")) for m in d["secondary_metrics"]],
      d["extra"]["single_blob_latency_ms"])
print("eval VALU per launch:", per("pmc_verify_sq", "k_eval_frac", "SQ_INSTS_VALU"), "-> per element", per("pmc_verify_sq", "k_eval_frac", "SQ_INSTS_VALU") / (65536 * 4096 / 64))
print("verify call HBM bytes:", v["call_total_hbm_bytes"], "x algorithmic", v["call_over_algorithmic"])
