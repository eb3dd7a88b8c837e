#!/usr/bin/env python3
"""usage: refresh_profiles.py <tag>   (e.g. r05)
Copies the summaries of the last `tools/gpu_profile.sh <tag>` run (gpurun_out/<tag>/prof/), the last default bench line
(gpurun_out/<tag>/bench_default.json) and the kernel timelines into profiles/<tag>/ and computes pmc_traffic.json from the
counters.  One script for every round (rounds 3 and 4 each had a copy)."""
import json
import os
import shutil
import sys

TAG = sys.argv[1]
MSM = "k_msm_comb30"  # the fixed-base MSM kernel (k_msm_comb28 until round 4)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
P = R + "gpurun_out/%s/prof/" % TAG
D = R + "profiles/%s/" % TAG
os.makedirs(D, exist_ok=True)
new = json.load(open(P + "summary_pmc.json"))
KiB = 1024.0


def per(run, kern, ctr):
    for k, v in new.get(run, {}).items():
        if (k == kern or k.startswith(kern + "<")) and ctr in v:
            return v[ctr]["per_launch"]
    return None


def hbm(run, kern, corr=1.0):
    f, w = per(run + "_fetch", kern, "FETCH_SIZE"), per(run + "_write", kern, "WRITE_SIZE")
    if f is None or w is None:
        return None
    return {"FETCH_SIZE_bytes": f * KiB * corr, "fetch_correction": corr, "WRITE_SIZE_bytes": w * KiB, "hbm_bytes": f * KiB * corr + w * KiB}


out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB per dispatch), separate passes, per launch; correction x2 on FETCH_SIZE for the wide coalesced streaming "
               "readers (k_challenge, k_eval_frac) as MI355X_MICROARCH.md prescribes, x1 for gathers (the fixed-base MSM kernel: matches the known gather bytes) and for the point decoder"}
c = hbm("pmc_commit", MSM)
alg_c = 131120 * 4096
out["commit_n4096_c22"] = dict(c, kernel=MSM, algorithmic_bytes_per_launch=alg_c, hbm_bytes_per_launch=c["hbm_bytes"], over_algorithmic=c["hbm_bytes"] / alg_c,
                               known_gather_bytes=49152 * 4096 * 96)
alg_v = 131168 * 65536
ch, ev, dc = hbm("pmc_verify", "k_challenge", 2.0), hbm("pmc_verify", "k_eval_frac", 2.0), hbm("pmc_verify", "k_g1_decompress_range")
rest = {}
for k in ("k_var_buckets_flat", "k_var_bitsums", "k_var_count", "k_var_scatter", "k_var_scan_lean", "k_transcript_leaves", "k_transcript_nodes", "k_batch_scalars"):
    h = hbm("pmc_verify", k)
    if h:
        launches = new["pmc_verify_fetch"][[x for x in new["pmc_verify_fetch"] if x == k or x.startswith(k + "<")][0]]["FETCH_SIZE"]["launches"]
        rest[k] = dict(h, launches_in_run=launches)
out["verify_n65536_c22"] = {"kernel": "k_challenge", "algorithmic_bytes_per_launch": alg_v, "hbm_bytes_per_launch": ch["hbm_bytes"], "k_challenge": ch, "k_eval_frac<16>": ev,
                            "k_g1_decompress_range": dc, "small_kernels_per_launch": rest,
                            "call_total_hbm_bytes_three_large_kernels": ch["hbm_bytes"] + ev["hbm_bytes"] + dc["hbm_bytes"],
                            "call_over_algorithmic": (ch["hbm_bytes"] + ev["hbm_bytes"] + dc["hbm_bytes"]) / alg_v,
                            "decoder_scratch_note": "k_g1_decompress_range: private_segment_fixed_size 0 (round 3: 720 B per lane = 2.76 GB of WRITE_SIZE per call)"}
json.dump(out, open(D + "pmc_traffic.json", "w"), indent=1)
shutil.copy(P + "summary_pmc.json", D + "pmc_counters_per_kernel.json")
for a, b in (("default", "bench_default_kernel_stats.csv"), ("commit", "trace_commit_kernel_stats.csv"), ("proof", "trace_proof4096_kernel_stats.csv"),
             ("proof2", "trace_proof4096_two_calls_in_flight_kernel_stats.csv"), ("proof3", "trace_proof4096_three_calls_in_flight_kernel_stats.csv"),
             ("verify", "trace_verify65536_kernel_stats.csv")):
    if os.path.exists(P + "summary_trace_%s_kernel_stats.csv" % a):
        shutil.copy(P + "summary_trace_%s_kernel_stats.csv" % a, D + b)
for src, dst in ((P + "timeline_verify.txt", "verify65536_kernel_timeline.txt"), (P + "timeline_proof2.txt", "proof4096_two_calls_in_flight_kernel_timeline.txt"),
                 (P + "timeline_proof3.txt", "proof4096_three_calls_in_flight_kernel_timeline.txt"), (P + "timeline_proof.txt", "proof4096_kernel_timeline.txt"),
                 (R + "gpurun_out/%s/bench_default.json" % TAG, "bench_default.json")):
    if os.path.exists(src):
        shutil.copy(src, D + dst)
for w in ("default", "commit", "proof", "proof2", "proof3", "verify"):
    try:
        line = [l for l in open(P + "trace_%s.log" % w) if l.startswith('{"metric"')][0]
    except (OSError, IndexError):
        continue
    open(D + "bench_%s_under_rocprofv3.json" % w, "w").write(line)
    d = json.loads(line)
    print(w, "under rocprofv3:", round(d["value"]), "blobs/s", round(d["ms_per_step"], 3), "ms")
if os.path.exists(D + "bench_default.json"):
    d = json.loads([l for l in open(D + "bench_default.json") if l.startswith("{")][-1])
    print("default:", round(d["value"]), round(d["ms_per_step"], 3), "frac", d["roofline"]["frac"], "valu_issue", (d["roofline"].get("valu_issue") or {}).get("frac"),
          [(round(m["value"]), m.get("valu_issue_frac"), m.get("value_two_calls_in_flight") or m.get("value_three_calls_in_flight")) for m in d["secondary_metrics"]], d["extra"]["single_blob_latency_ms"])
print("decoder WRITE_SIZE per launch (bytes):", dc["WRITE_SIZE_bytes"], " verify call HBM / algorithmic:", out["verify_n65536_c22"]["call_over_algorithmic"])
print("commit traffic / algorithmic:", out["commit_n4096_c22"]["over_algorithmic"])
for run, kern in (("pmc_commit_sq", MSM), ("pmc_verify_sq", "k_challenge"), ("pmc_verify_sq", "k_eval_frac"), ("pmc_verify_sq", "k_g1_decompress_range")):
    print(run, kern, "SQ_INSTS_VALU per launch:", per(run, kern, "SQ_INSTS_VALU"))
