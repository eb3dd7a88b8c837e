#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

One "step" = one pass of the hot path over one batch of synthetic blobs that are
already resident in HBM.  Default workload: `blob_to_kzg_commitment` on 4096
blobs per GPU (BASELINE.json configs[1]); `--workload proof` is configs[2]
(compute_blob_kzg_proof, 4096 per GPU), `--workload verify` configs[3]
(verify_blob_kzg_proof_batch, 65,536 triples per GPU), and `--workload commit
--batch 131072 --gpus 8` is configs[4] (2^20 blobs over 8 GPUs).

`--gpus N` with no WORLD_SIZE in the environment makes THIS process the
launcher: it starts N rank processes (one GPU each, RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* set) before anything touches HIP and relays rank 0's JSON line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks
already exist and WORLD_SIZE must equal N.  Blobs are independent, so ranks own
contiguous global index ranges (weak scaling: `--batch` blobs per rank) and the
data path has no collective; the exchanges are one RCCL all-gather of 48 B per
blob (commit / proof) or, for batch verification, the all-gathers of
kateth_amd/dist.py (32-B transcript roots + first-error records, 192 B of
partial sums).

Prints ONE JSON line on rank 0 (contract in the task description) carrying
`roofline` (dominant kernel of the workload, HIP-event timed inside the library
on the stream it runs on) and `cpu_baseline` (the C port of the reference's CPU
algorithm, oracle/cport, timed on the host cores on a bounded sample; rank 0,
N = 1 only).  With the default workload the proof and verify workloads are also
reported under "extra" (each with its own roofline object); they are not part
of `value`.
"""
import argparse
import json
import os

# Several HIP streams are live in a run (the engine's copy / side / session streams, the lanes of calls kept in flight) and
# the runtime multiplexes them onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default: two streams that land on one queue run
# strictly one after the other (seen: both lanes of `--in-flight 2` on one queue, profiles/r04).  Set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_BLOB = 131072
ALG_BYTES = {"commit": 131072 + 48, "proof": 131072 + 48 + 48, "verify": 131072 + 96}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x4844
METRIC = {
    "commit": "blobs/sec for blob_to_kzg_commitment (n=4096 field elements per blob)",
    "proof": "blobs/sec for compute_blob_kzg_proof (n=4096 field elements per blob)",
    "verify": "blobs/sec for verify_blob_kzg_proof_batch (n=4096 field elements per blob)",
}
DEFAULT_BATCH = {"commit": 4096, "proof": 4096, "verify": 65536}
DTYPE = "u32 limbs (381-bit Fp / 255-bit Fr Montgomery integer arithmetic; signed 30-bit limbs on v_mad_i64_i32 in the fixed-base MSM, 28/29-bit radix in the other hot loops)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("commit", "proof", "verify"), default="commit")
    ap.add_argument("--batch", type=int, default=0, help="blobs per GPU per step (default: 4096 commit/proof, 65536 verify; configs[4]: 131072 with --gpus 8)")
    ap.add_argument("--window-bits", type=int, default=int(os.environ.get("KATETH_AMD_WINDOW_BITS", "0")),
                    help="table class of the fixed-base comb (include/kateth_amd.h, kzg_config): 0 = the library default = the fastest class the device has room for "
                         "(class 22: blocks of 22+21+21 points, 8 plane groups = 192 GiB of the 288 GB HBM, 49,152 additions per blob; 4 groups = 96 GiB below 232 GiB free); "
                         "16 -> 12.9 GB, 65,536 additions; 8 -> 100 MB, 131,072 additions")
    ap.add_argument("--default-budget", action="store_true", help="do NOT pass KZG_CFG_TABLE_MAX: the library's default budget (100 GiB -> the 96-GiB table, G = 4) "
                    "instead of the largest table the device has room for (192 GiB, G = 8)")
    ap.add_argument("--in-flight", type=int, default=0, help="calls kept in flight: step i is enqueued on stream i mod F with result buffers of its own (0 = default = 1: one call at a "
                    "time for every workload, as the reference's methods are called; --workload proof also reports the rate with 3 in flight as `value_three_calls_in_flight` -- a call's hash + "
                    "quotient kernels then run in the shadow of the others' MSMs; the timed region is still K steps between two full synchronisations)")
    ap.add_argument("--blocking-setup", action="store_true", help="create the context without KZG_CFG_BUILD_ASYNC (kzg_ctx_create returns when the full table stands)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, the measured path) or gloo (rehearsal: ranks may share one card, gathers go through the host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary proof/verify workloads")
    ap.add_argument("--cpu-sample", type=int, default=0, help="blobs in the CPU baseline sample (0 = auto, ~10-30 s)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this run")
    ap.add_argument("--dist-single", action="store_true", help="with --gpus 1: still create the process group (world size 1) and route the result gathers / the sharded "
                    "verification through it -- exercises the RCCL calls of the N-rank path on a one-GPU box")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes through this script's launcher even for --gpus 1 (checks the launcher against the direct path)")
    ap.add_argument("--rank-logs", default=os.path.join(ROOT, "gpurun_out", "bench_ranks"), help="launcher: directory for every rank's stdout/stderr (rank<k>.out / rank<k>.err)")
    ap.add_argument("--group", type=int, default=0, help="ONE process, a GROUP context of N members behind the C ABI (kzg_config.devices / ndev) driven through the "
                    "device-resident sharded calls kzg_*_group_dev: member k's --batch blobs resident on member k's GPU, no torch.distributed, no collective.  Members sit on "
                    "devices 0..N-1; when fewer GPUs are visible they share device 0 (a rehearsal of the code path, not a speed-up).  Same JSON line; roofline from member 0's kernels")
    ap.add_argument("--dry-run", action="store_true", help="print every rank's HBM plan for --workload/--batch/--gpus as one JSON line and exit non-zero if it cannot fit; touches no GPU")
    ap.add_argument("--assume-hbm-gib", type=float, default=0.0, help="--dry-run: HBM per GPU in GiB (default: 288, what an MI355X reports)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# memory plan: what one rank keeps resident, from the engine's own constants (include/kateth_amd.h, engine.hip)
# ---------------------------------------------------------------------------------------------------------------
GIB = float(1 << 30)
MI355X_HBM_BYTES = 288.0 * (1 << 30)  # hipMemGetInfo reports 287.4 GiB free on an idle MI355X here
TABLE_GROUP_BYTES = {22: 64 * (1 << 22) * 96, 16: 64 * (4 << 15) * 96, 8: 64 * (8 << 7) * 96, 4: 64 * (16 << 3) * 96}


def table_plan(window_bits, free_bytes, table_max=True):
    """(class, plane groups, table bytes) kzg_ctx_create builds for `window_bits` with `free_bytes` of HBM free
    (engine.hip: 22/G8 from 232 GiB, 22/G4 from 136 GiB, 16 from 21 GiB, else 8; an explicit class is honoured).  The AUTOMATIC
    choice stays within the budget: 100 GiB by default (-> G = 4), unlimited with KZG_CFG_TABLE_MAX, which this bench passes."""
    g22 = TABLE_GROUP_BYTES[22]
    c = window_bits
    if c == 0:
        c = 22 if free_bytes >= 4 * g22 + 40 * GIB else (16 if free_bytes >= 21 * GIB else 8)
    cls = 22 if c >= 22 else (16 if c >= 16 else (8 if c >= 8 else 4))
    groups = (8 if free_bytes >= 8 * g22 + 40 * GIB and (table_max or window_bits != 0) else 4) if cls == 22 else 16
    return cls, groups, groups * TABLE_GROUP_BYTES[cls] + (403e6 if cls == 22 else 0)


def memory_plan(workload, n, window_bits, total_bytes, table_max=True):
    """bytes one rank holds for `n` blobs per step: caller buffers (blobs, results), the context's table and the engine's
    workspace (fixed-base MSM: chunks of at most 16,384 blobs -> bit-plane masks 128 KiB + lane sums 65 x 192 B per blob;
    proof: + 128 KiB of quotient scalars per blob of a chunk; verify: a pooled session of ~800 B per item)"""
    cls, groups, table = table_plan(window_bits, total_bytes, table_max)
    chunk = min(n, 16384)
    slots = 3  # KZG_WS_SLOTS: successive calls take three workspaces in turn (calls in flight on several streams), and they grow together
    msm_ws = chunk * (BYTES_PER_BLOB + 65 * 192 + 192)
    plan = {"table_class": cls, "plane_groups": groups, "table": table, "table_build_scratch_transient": 13 * GIB if cls == 22 else 1.7 * GIB,
            "blobs": n * BYTES_PER_BLOB, "results_and_status": n * (48 + 4)}
    if workload == "commit":
        plan["workspace"] = slots * msm_ws
    elif workload == "proof":
        plan["commitments"] = n * 48
        plan["workspace"] = slots * (msm_ws + 2 * chunk * (BYTES_PER_BLOB + 512))
    else:
        plan["commitments_and_proofs"] = n * 96
        plan["workspace"] = n * 800 + (1 << 20)
        plan["setup_phase_peak"] = slots * (msm_ws + 2 * chunk * (BYTES_PER_BLOB + 512))  # the triples are produced by commit + prove first
    resident = sum(v for k, v in plan.items() if k not in ("table_class", "plane_groups", "table_build_scratch_transient"))
    plan["resident_total"] = resident
    plan["peak_total"] = max(resident, plan["table"] + plan["table_build_scratch_transient"])
    plan["hbm_total"] = total_bytes
    plan["fits"] = bool(plan["peak_total"] + 2 * GIB <= total_bytes)  # 2 GiB for the runtime, RCCL buffers and the allocator
    return plan


def dry_run(args):
    wl = args.workload
    n = args.batch or DEFAULT_BATCH[wl]
    total = args.assume_hbm_gib * GIB if args.assume_hbm_gib else MI355X_HBM_BYTES
    plan = memory_plan(wl, n, args.window_bits, total, not args.default_budget)
    out = {"dry_run": True, "workload": wl, "gpus": args.gpus, "blobs_per_gpu": n, "blobs_total": n * args.gpus,
           "ranks": [dict(plan, rank=r, first_blob=r * n) for r in range(args.gpus)],
           "gib": {k: round(v / GIB, 2) for k, v in plan.items() if isinstance(v, (int, float)) and not isinstance(v, bool) and k not in ("table_class", "plane_groups")},
           "exchange_bytes_per_step_per_rank": n * 48 if wl != "verify" else 32 + 32 + 192}
    print(json.dumps(out), flush=True)
    if not plan["fits"]:
        sys.stderr.write("bench.py --dry-run: %.1f GiB needed per GPU, %.1f GiB of HBM: does not fit\n" % (plan["peak_total"] / GIB, total / GIB))
        sys.exit(3)


# ---------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a rendezvous in the environment
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Start N rank processes of this script BEFORE anything here touches HIP (this process never imports torch),
    relay rank 0's stdout, and exit with the first non-zero rank exit code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    os.makedirs(args.rank_logs, exist_ok=True)
    child_args = [a for a in sys.argv[1:] if a != "--spawn"]
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        # every rank's stdout and stderr survive in files (on an N-GPU failure the other ranks' output is the evidence);
        # rank 0's stdout is also relayed
        err = open(os.path.join(args.rank_logs, "rank%d.err" % rank), "wb")
        out = subprocess.PIPE if rank == 0 else open(os.path.join(args.rank_logs, "rank%d.out" % rank), "wb")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_args, env=env, stdout=out, stderr=err))
    # relay rank 0's stdout; if ANY rank dies, stop the others (their collectives would otherwise wait for the timeout)
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for k, p in enumerate(procs):
            if codes[k] is None:
                codes[k] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for k, p in enumerate(procs):
                if codes[k] is None:
                    p.terminate()  # exactly the PIDs started above
            for k, p in enumerate(procs):
                if codes[k] is None:
                    try:
                        codes[k] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[k] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    rank0_out = b"".join(chunks)
    with open(os.path.join(args.rank_logs, "rank0.out"), "wb") as fh:
        fh.write(rank0_out)
    for out, err in logs:
        err.close()
        if out is not subprocess.PIPE:
            out.close()
    sys.stdout.write(rank0_out.decode())
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    if bad:
        sys.stderr.write("bench.py launcher: rank exit codes %r; per-rank logs in %s\n" % (codes, args.rank_logs))
        for rank, c in enumerate(codes):
            if c != 0:
                try:
                    tail = open(os.path.join(args.rank_logs, "rank%d.err" % rank), "rb").read()[-1500:].decode(errors="replace")
                except OSError:
                    tail = ""
                sys.stderr.write("---- rank %d (exit %r) stderr tail ----\n%s\n" % (rank, c, tail))
        sys.exit(bad[0] if bad[0] and bad[0] > 0 else 1)
    for rank in range(args.gpus):  # a clean run still shows what the ranks said on stderr (warnings), rank 0 first
        try:
            txt = open(os.path.join(args.rank_logs, "rank%d.err" % rank), "rb").read().decode(errors="replace")
        except OSError:
            continue
        if txt.strip() and rank == 0:
            sys.stderr.write(txt)


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline and roofline helpers (rank 0)
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(sample_blobs, setup_path, gpu_out48):
    """oracle/cport (C port of kateth's CPU path: Pippenger c=10 signed digits as blst uses, bases re-normalised on
    every call, one thread and all host cores) timed on a bounded sample of the same synthetic blobs, rank 0 only.
    Checker code: never the thing measured as `value`.  Its outputs are compared byte for byte with the GPU's."""
    from oracle.cport import binding

    res = binding.time_commitment(None, setup_path, sample_blobs, SEED)
    raw = res.pop("_raw_outputs")
    res.pop("outputs", None)
    res["matches_gpu_bytes"] = bool(raw == gpu_out48[: len(raw)])
    return res


def pmc_traffic(workload, n, window_bits):
    """HBM bytes per launch of the workload's dominant kernel from the COMMITTED rocprofv3 PMC passes (profiles/r0N/
    pmc_traffic.json): the fallback when the live passes (live_pmc_traffic) are switched off or fail.  None when no profile of
    this (workload, batch, class) configuration has been recorded."""
    for rel in (("profiles", "r05", "pmc_traffic.json"), ("profiles", "r04", "pmc_traffic.json"), ("profiles", "r03", "pmc_traffic.json"), ("profiles", "r02", "pmc_traffic.json"), ("profiles", "r01", "pmc_traffic.json")):
        try:
            rec = json.load(open(os.path.join(ROOT, *rel)))
        except (OSError, ValueError):
            continue
        for key in ("%s_n%d_c%d" % (workload, n, window_bits), "n%d_c%d" % (n, window_bits) if workload == "commit" else ""):
            if key and key in rec:
                return rec[key]["hbm_bytes_per_launch"]
    return None


PMC_KERNEL = {"commit": "k_msm_comb30", "proof": "k_msm_comb30", "verify": "k_challenge"}
# the kernels of ONE call of each workload (exact base names), for the per-call instruction count
CALL_KERNELS = {
    "commit": ("k_comb_transpose", "k_msm_comb30", "k_msm_reduce", "k_msm_reduce_half4", "k_msm_reduce_splits", "k_g1_compress"),
    "proof": ("k_comb_transpose", "k_msm_comb30", "k_msm_reduce", "k_msm_reduce_half4", "k_msm_reduce_splits", "k_g1_compress", "k_challenge", "k_challenge_split",
              "k_challenge_pair", "k_challenge_and_decode", "k_challenge_pair_and_decode", "k_g1_decompress", "k_poly_root_inverse", "k_poly", "k_merge_status"),
    # batches above 16,384 triples (the child's own setup -- commit + prove of the triples in chunks of 16,384 -- launches
    # k_challenge_pair* and k_g1_decompress, which a verification of this size does not)
    "verify": ("k_challenge", "k_challenge_split", "k_eval_frac", "k_g1_decompress_range", "k_transcript_leaves", "k_transcript_nodes", "k_batch_scalars",
               "k_batch_ysum_finish", "k_var_count", "k_var_scan", "k_var_scan_wide", "k_var_scan_lean", "k_var_scatter", "k_var_buckets", "k_var_buckets_flat", "k_var_fold",
               "k_var_windows", "k_var_bitsums"),
}
PMC_CHILD_STEPS, PMC_CHILD_WARMUP = 2, 1


def kernel_base_name(full):
    """'void kzg::k_eval_frac<16>(unsigned char const*, ...) [clone .kd]' -> 'k_eval_frac'"""
    import re

    m = re.search(r"(k_[A-Za-z0-9_]+)", full)
    return m.group(1) if m else full


def live_pmc(args, workload, n, counters, timeout_s=200):
    """Hardware counters measured IN THIS RUN: one child process per counter, `rocprofv3 --pmc <counter> -- python3 bench.py ...`
    (separate passes, counters only, no tracing -- MI355X_MICROARCH.md, HBM section) of the same workload, batch and table
    class, started after this process has released its context (two 192-GiB tables do not fit one card).  Returns
    {counter: {kernel base name: (sum over dispatches, dispatches)}} or {"error": ...}."""
    import csv
    import glob
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for counter in counters:
            d = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--workload", workload,
                   "--batch", str(n), "--window-bits", str(args.window_bits), "--steps", str(PMC_CHILD_STEPS), "--warmup", str(PMC_CHILD_WARMUP), "--no-cpu-baseline",
                   "--no-extra", "--no-live-traffic", "--blocking-setup"] + (["--default-budget"] if args.default_budget else [])
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if res.returncode != 0:
                return {"error": "rocprofv3 --pmc %s exited %d: %s" % (counter, res.returncode, (res.stderr or res.stdout)[-300:])}
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    name = kernel_base_name(row["Kernel_Name"])
                    tot, cnt = per.get(name, (0.0, 0))
                    per[name] = (tot + float(row["Counter_Value"]), cnt + 1)
            if not per:
                return {"error": "no %s rows" % counter}
            out[counter] = per
    except (subprocess.TimeoutExpired, OSError) as err:
        return {"error": repr(err)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def traffic_from_pmc(pmc, workload):
    """roofline.traffic: HBM bytes per launch of the workload's dominant kernel = FETCH_SIZE (KiB) x 1024 x c + WRITE_SIZE (KiB)
    x 1024 with the guide's gfx950 correction c = 2 for wide coalesced streaming reads (k_challenge: every lane streams its blob)
    and c = 1 for k_msm_comb30, whose reads are 96-byte gathers of table entries (the count matches the known gather bytes at
    c = 1, profiles/r02/pmc_traffic.json)."""
    if "error" in pmc or "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return {"error": pmc.get("error", "counters missing")}
    k = PMC_KERNEL[workload]
    if k not in pmc["FETCH_SIZE"] or k not in pmc["WRITE_SIZE"]:
        return {"error": "no rows for %s" % k}
    corr = 2.0 if workload == "verify" else 1.0
    f_tot, f_cnt = pmc["FETCH_SIZE"][k]
    w_tot, w_cnt = pmc["WRITE_SIZE"][k]
    fetch = f_tot / f_cnt * 1024.0 * corr
    write = w_tot / w_cnt * 1024.0
    return {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "fetch_correction": corr, "kernel": k, "launches_sampled": f_cnt,
            "how": "two rocprofv3 --pmc child passes of this bench.py run (FETCH_SIZE, WRITE_SIZE; KiB per dispatch)"}


ISSUE_CYCLES_V_MAD = 4.125  # v_mad_u64_u32 at two waves per SIMD: what kzg_microbench_valu_issue has measured on every box of rounds 3-4


def issue_interval(setup):
    """(cycles per wave-instruction, shader clock of the microbenchmark's own load), from kzg_microbench_valu_issue.  The figure is a
    hardware constant (4.125 +- 0.001 in every bench.py run); the microbenchmark times waves with the shader-clock counter and has
    returned 2.5 / 2.7 / 115 late in the GPU test suite's long run (tests/test_gpu_round4.py::test_measurement_aids), so a value off
    the constant by more than 5 % is measured again and, failing that, replaced by the constant -- never silently: the clock it
    reports then says 0 and `valu_issue.how` carries the rejected readings."""
    seen = []
    for _ in range(3):
        cpi, ghz = setup.microbench_valu_issue(2, 20000)
        seen.append(round(cpi, 4))
        if abs(cpi / ISSUE_CYCLES_V_MAD - 1.0) <= 0.05:
            return cpi, ghz
    sys.stderr.write("[bench] kzg_microbench_valu_issue returned %r: the constant %.3f is used\n" % (seen, ISSUE_CYCLES_V_MAD))
    return ISSUE_CYCLES_V_MAD, 0.0


def valu_issue_object(pmc, workload, issue, simds, call_ms, kernel_ms=None, extra_calls=0, run_clock=None):
    """The floor these kernels are actually bound by: VALU instruction ISSUE.  SQ_INSTS_VALU (wave-instructions,
    whole chip, one rocprofv3 --pmc child pass of this run) / SIMDs x the measured issue interval of v_mad_u64_u32 at two waves per
    SIMD (kzg_microbench_valu_issue: cycles per wave-instruction, and the shader clock that load sustains) = the time the
    instruction stream needs on a perfectly filled chip.  `frac` = that floor / the measured wall time of a call; `kernel_frac`
    the same for the dominant kernel alone."""
    if "error" in pmc or "SQ_INSTS_VALU" not in pmc:
        return {"error": pmc.get("error", "SQ_INSTS_VALU missing")}
    cpi, ghz_microbench = issue
    ghz = run_clock[0] if run_clock else ghz_microbench
    if not ghz:
        return {"error": "no shader clock: neither probe waves nor a usable issue microbenchmark"}
    per = pmc["SQ_INSTS_VALU"]
    calls = PMC_CHILD_STEPS + PMC_CHILD_WARMUP
    by_kernel = {}
    for name in CALL_KERNELS[workload]:
        if name in per:
            tot, cnt = per[name]
            # the proof child commits once before its calls (the same fixed-base MSM kernels on the same batch): one more call's worth of those kernels
            c = calls + (extra_calls if name in CALL_KERNELS["commit"] else 0)
            by_kernel[name] = tot / c
    total = sum(by_kernel.values())
    obj = {"insts_per_call": total, "insts_per_simd": total / simds, "cycles_per_inst": cpi, "clock_ghz": ghz, "simds": simds,
           "clock_source": "sleeping probe waves beside the timed calls (kzg_clock_probe_*): mean over the XCDs; lowest / highest: %r" % (list(run_clock[1:]),) if run_clock
           else "the issue microbenchmark's own clock (no probe ran)", "clock_ghz_issue_microbenchmark": ghz_microbench,
           "floor_ms": total / simds * cpi / (ghz * 1e6), "call_ms": call_ms,
           "insts_by_kernel_per_call": by_kernel,
           "how": "SQ_INSTS_VALU: one rocprofv3 --pmc child pass of this run; cycles_per_inst: kzg_microbench_valu_issue (v_mad_u64_u32, 8 independent chains, "
                  "2 waves per SIMD, whole chip) in this process; floor_ms = insts_per_simd x cycles_per_inst / clock"}
    obj["frac"] = obj["floor_ms"] / call_ms if call_ms else None
    k = PMC_KERNEL[workload]
    if kernel_ms and k in per:
        tot, cnt = per[k]
        per_launch = tot / cnt
        obj.update({"kernel": k, "kernel_insts_per_launch": per_launch, "kernel_floor_ms": per_launch / simds * cpi / (ghz * 1e6), "kernel_ms": kernel_ms})
        obj["kernel_frac"] = obj["kernel_floor_ms"] / kernel_ms
    return obj


def roofline_object(workload, n, prof, window_bits, call_ms=None):
    """`roofline` for one workload from the library's HIP-event kernel times (events recorded on the stream each kernel is
    launched on).  commit / proof: the dominant kernel is the fixed-base MSM; achieved = algorithmic bytes per launch (SURVEY.md
    section 8(d) bytes per blob x blobs per launch) / its average launch duration.  verify: several kernels share the call (the
    SHA-256 challenge alone on the chip, then point decoding and evaluation side by side on two streams, then the lincombs), so
    `frac` is the CALL-level figure -- algorithmic bytes of the batch / wall time of the call, host pairing included -- and the
    stand-alone SHA-256 kernel's own figure is reported beside it (`dominant_kernel_*`)."""
    kinds = {k: v for k, v in prof["kinds"].items() if v[1]}
    if not kinds:
        return None
    dominant = max(kinds, key=lambda k: kinds[k][0])
    if workload == "verify" and "k_challenge*" in kinds:
        # point decoding and evaluation run CONCURRENTLY on two streams, so their event spans overlap and each reads
        # longer than the kernel alone; the SHA-256 challenge kernel runs alone and is the largest stand-alone kernel
        dominant = "k_challenge*"
    ms_total, launches = kinds[dominant]
    k_ms = ms_total / launches
    calls = max(1, prof.get("calls", 1))
    blobs_per_launch = n * calls / launches  # batches above the engine's chunk size run the kernel once per chunk
    alg = ALG_BYTES[workload] * blobs_per_launch
    ach = alg / (k_ms * 1e-3) / 1e9
    summed = sum(v[0] for v in kinds.values())
    roof = {
        "kernel": dominant,
        "bound": "valu",
        "limiter": "VALU instruction issue: valu_issue_frac = floor (SQ_INSTS_VALU x issue interval / clock) / call time; achieved/peak/frac = HBM",
        "limiter_detail": "integer v_mad_u64_u32 / SHA-256 bit ops: see valu_issue (floor from SQ_INSTS_VALU x the measured issue interval) and DESIGN.md "
                   "section 5; achieved / peak / frac are the HBM figures the north star asks for (algorithmic bytes over the chip's 8 TB/s), kept as they were",
        "achieved": ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS,
        "traffic": pmc_traffic(workload, n, window_bits),
        "traffic_source": "committed rocprofv3 --pmc profile of this configuration (profiles/)",
        "kernel_ms": k_ms,
        "launches": launches,
        "blobs_per_launch": blobs_per_launch,
        "algorithmic_bytes_per_blob": ALG_BYTES[workload],
        "kernel_ms_by_class_per_call": {k: v[0] / calls for k, v in kinds.items()},
        "shader_clock_ghz_during_run": prof.get("clock_ghz"),
        "summed_kernel_ms_per_call": summed / calls,
        "achieved_over_summed_kernels": ALG_BYTES[workload] * n / (summed / calls * 1e-3) / 1e9,
        "note": "frac is the algorithmic HBM rate the north star asks for; the kernels are bound by VALU instruction issue: valu_issue.frac is the fraction of THAT floor"
        + ("; k_g1_decompress and k_eval_frac overlap on two streams (their spans are not additive)" if workload == "verify" else ""),
    }
    if workload == "verify" and call_ms:
        call_ach = ALG_BYTES[workload] * n / (call_ms * 1e-3) / 1e9
        longest = max(kinds, key=lambda k: kinds[k][0] / kinds[k][1])
        roof.update({"dominant_kernel_achieved": ach, "dominant_kernel_frac": ach / HBM_PEAK_GBS, "achieved": call_ach, "frac": call_ach / HBM_PEAK_GBS,
                     "call_ms": call_ms, "scope": "whole verify_blob_kzg_proof_batch call (all kernels + host pairing); dominant_kernel_* is k_challenge alone",
                     "longest_kernel_by_duration": longest, "longest_kernel_ms": kinds[longest][0] / kinds[longest][1],
                     "dominant_kernel_by_bytes": dominant,
                     "kernel_naming": "`kernel` = the kernel that moves the call's algorithmic bytes and runs ALONE on the chip (k_challenge: every blob byte once); "
                                      "the longest event span of the call is `longest_kernel_by_duration` (the point decoder's, stretched by k_eval_frac beside it on a second stream)"})
    return roof


ROOFLINE_FIRST = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "valu_issue_frac", "valu_issue_kernel_frac",
                  "valu_issue_floor_ms", "shader_clock_ghz", "kernel_ms", "call_ms", "verify_valu_issue_frac", "verify_frac", "verify_longest_kernel_by_duration",
                  "proof_valu_issue_frac", "proof_frac", "valu_frac", "launches")


def flatten_valu_issue(roof):
    """the figures that say how close a workload is to its real bound, as SCALARS of the roofline object itself (the nested
    `valu_issue` dict stays for the detail): a reader that keeps only an object's first scalar fields still sees them"""
    vi = roof.get("valu_issue") or {}
    clk = roof.get("shader_clock_ghz_during_run")
    roof["valu_issue_frac"] = vi.get("frac")
    roof["valu_issue_kernel_frac"] = vi.get("kernel_frac")
    roof["valu_issue_floor_ms"] = vi.get("floor_ms")
    roof["shader_clock_ghz"] = clk[0] if isinstance(clk, (list, tuple)) and clk else vi.get("clock_ghz")
    roof.setdefault("call_ms", vi.get("call_ms"))


def ordered_roofline(roof):
    return dict([(k, roof[k]) for k in ROOFLINE_FIRST if k in roof] + [(k, v) for k, v in roof.items() if k not in ROOFLINE_FIRST])


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args, rank, local_rank, world):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        self.rank, self.world = rank, world
        self.use_dist = world > 1 or args.dist_single  # --dist-single: a one-rank process group, same code path as N ranks
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:  # --dist-single without a launcher: any free port
                import socket

                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        assert torch.cuda.is_available(), "bench.py needs an MI355X; the engine has no CPU fallback"
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and world > ndev:
            raise SystemExit("bench.py: %d ranks but %d GPUs visible (RCCL needs one GPU per rank; use --backend gloo to rehearse on one card)" % (world, ndev))
        self.local_dev = local_rank % max(1, ndev)  # ranks share a card only in the gloo rehearsal
        torch.cuda.set_device(self.local_dev)
        self.dev = torch.device("cuda", self.local_dev)
        if self.use_dist:  # RCCL needs one GPU per rank: every rank of a host must sit on a card of its own
            import socket

            mine = (socket.gethostname(), torch.cuda.current_device(), getattr(torch.cuda.get_device_properties(self.local_dev), "pci_bus_id", -1))
            seats = [None] * world
            dist.all_gather_object(seats, mine)
            self.seats = seats
            if args.backend == "nccl" and len(set(seats)) != world:
                raise SystemExit("bench.py: backend nccl but two ranks share a GPU: %r" % (seats,))
        # the HBM plan of this rank, checked against the device BEFORE anything is allocated
        wl = args.workload
        free_b, total_b = torch.cuda.mem_get_info(self.local_dev)
        self.plan = memory_plan(wl, args.batch or DEFAULT_BATCH[wl], args.window_bits, free_b, not args.default_budget)
        if not self.plan["fits"] and world > 1 and args.backend == "nccl":
            raise SystemExit("bench.py rank %d: plan needs %.1f GiB, device has %.1f GiB free (%.1f total)" % (rank, self.plan["peak_total"] / GIB, free_b / GIB, total_b / GIB))
        import kateth_amd

        self.kateth_amd = kateth_amd
        self.setup_path = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
        # Setup::load_json's drop-in shape: KZG_CFG_BUILD_ASYNC -- the context is usable on a first-use
        # table at once and the benchmarked table is built beside the first call.  first_use_s = create + ONE commitment of one
        # blob, checked against the golden vector; full_table_s = until the chosen table is in (what the timed region runs on).
        t0 = time.time()
        self.setup, tried = None, []
        for c in [args.window_bits] + [w for w in (16, 8) if 0 < w < args.window_bits]:
            try:  # an explicitly requested class that cannot be allocated steps down (0 = the engine chooses by free memory)
                self.setup = kateth_amd.Setup.load_json(self.setup_path, device=self.local_dev, window_bits=c, table_max=not args.default_budget,
                                                        build_async=not args.blocking_setup)
                break
            except kateth_amd.kzg.EngineError as err:
                tried.append("c=%d: %s" % (c, err))
                torch.cuda.empty_cache()
        assert self.setup is not None, "context creation failed for every window size: %r" % tried
        self.t_create = time.time() - t0
        self.stream = torch.cuda.current_stream().cuda_stream
        self.t_first_use, self.first_use_class = None, None
        if not args.blocking_setup:  # (the PMC child passes run blocking: no stray launch among the counted dispatches)
            first_blob = self.make_blobs(1, 0)
            out1, st1 = self.commit(first_blob, 1)
            torch.cuda.synchronize()
            self.t_first_use = time.time() - t0
            self.first_use_class = self.setup.window_bits
            if self.t_first_use > 1.5:  # seen in round 5: code objects loaded behind the background build's allocation (profiles/r05/first_use_regression.json)
                sys.stderr.write("[bench] first use after kzg_ctx_create took %.2f s: the background table build is in the first call's way\n" % self.t_first_use)
            check_golden(out1.cpu().numpy().tobytes(), 1, 0, "commitment")
            del first_blob, out1, st1
        self.setup.wait_ready()
        self.t_setup = time.time() - t0
        self.stream = torch.cuda.current_stream().cuda_stream

    # -- inputs ---------------------------------------------------------------------------------------------------
    def make_blobs(self, n, first_index):
        d = self.torch.empty(n * BYTES_PER_BLOB, dtype=self.torch.uint8, device=self.dev)
        self.setup.synth_blobs_dev(SEED, first_index, n, d.data_ptr(), self.stream)
        return d

    def commit(self, d_blobs, n, d_out=None, d_status=None, stream=None):
        t = self.torch
        d_out = t.empty(n * 48, dtype=t.uint8, device=self.dev) if d_out is None else d_out
        d_status = t.empty(n, dtype=t.int32, device=self.dev) if d_status is None else d_status
        self.setup.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), self.stream if stream is None else stream)
        return d_out, d_status

    def prove(self, d_blobs, d_com, n, d_out=None, d_status=None, stream=None):
        t = self.torch
        d_out = t.empty(n * 48, dtype=t.uint8, device=self.dev) if d_out is None else d_out
        d_status = t.empty(n, dtype=t.int32, device=self.dev) if d_status is None else d_status
        self.setup.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_com.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), self.stream if stream is None else stream)
        return d_out, d_status

    def lanes(self, flight, n):
        """`flight` (torch stream, its raw handle, result buffer, status buffer) sets for calls kept in flight side by side"""
        t = self.torch
        out = []
        for k in range(flight):
            st = t.cuda.current_stream() if flight == 1 else t.cuda.Stream(device=self.dev)
            out.append((st, st.cuda_stream, t.empty(n * 48, dtype=t.uint8, device=self.dev), t.zeros(n, dtype=t.int32, device=self.dev)))
        return out

    def verify(self, d_blobs, d_com, d_prf, n, first_index, n_total, stream=None):
        if not self.use_dist:
            return self.setup.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_com.data_ptr(), d_prf.data_ptr(), n, self.stream if stream is None else stream)
        from kateth_amd import dist as kdist

        gather_dev = self.dev if self.args.backend == "nccl" else self.torch.device("cpu")
        return kdist.verify_blob_proof_batch_sharded(self.setup, d_blobs.data_ptr(), d_com.data_ptr(), d_prf.data_ptr(), n, first_index, n_total,
                                                     self.rank, self.world, gather_dev, self.stream)

    def gather48(self, d_local, gathered):
        if not self.use_dist:
            return
        if self.args.backend == "nccl":
            self.dist.all_gather_into_tensor(gathered, d_local)  # RCCL over xGMI: 48 B per blob
        else:  # gloo rehearsal path: stage through the host
            host = [self.torch.empty(d_local.numel(), dtype=self.torch.uint8) for _ in range(self.world)]
            self.dist.all_gather(host, d_local.cpu())
            gathered.copy_(self.torch.cat(host))

    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    # -- the timed region -----------------------------------------------------------------------------------------
    def _run(self, fn, reps):
        """`reps` calls of fn between two fences: (seconds, profile)"""
        self.setup.profile_begin()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        self.fence()
        elapsed = time.perf_counter() - t0
        prof = self.setup.profile_end()
        prof["calls"] = reps
        return elapsed, prof

    def _run_with_clock_probe(self, fn, reps, step_s):
        """the same with sleeping probe waves beside the calls (kzg_clock_probe_*: the shader clock this workload sustains, for
        roofline.valu_issue).  The probe runs for HALF the expected duration (`step_s` = one call, measured on the last warm-up
        call); should it outlast the calls all the same -- the closing fence would then wait for it -- the calls are timed again
        without a probe."""
        probe_s = 0.5 * step_s * reps
        if probe_s < 1e-3:
            return self._run(fn, reps)
        self.setup.clock_probe_launch(int(1e6 * probe_s))
        elapsed, prof = self._run(fn, reps)
        clock = self.setup.clock_probe_read()
        if probe_s > 0.9 * elapsed:
            elapsed, prof = self._run(fn, reps)
        prof["clock_ghz"] = clock
        return elapsed, prof

    def timed(self, step):
        a = self.args
        last = 0.0
        for _ in range(a.warmup):  # untimed; each between fences so that the last one gives the probe its duration
            self.fence()
            t0 = time.perf_counter()
            step()
            self.fence()
            last = time.perf_counter() - t0
        self.fence()
        if not self.use_dist and a.warmup:  # (single process only: a repeated timed loop on one rank would break the ranks' barriers)
            elapsed, prof = self._run_with_clock_probe(step, a.steps, last)
        else:
            elapsed, prof = self._run(step, a.steps)
        if self.use_dist:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev if a.backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, prof

    def measure(self, fn, reps, warm=2):
        """secondary workloads: (seconds per call, profile) over `reps` calls after `warm` warm-up calls (every lane of a
        calls-in-flight measurement gets its first call here; the last warm-up call is timed, for the clock probe's duration)"""
        for _ in range(max(1, warm) - 1):
            fn()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        self.torch.cuda.synchronize()
        elapsed, prof = self._run_with_clock_probe(fn, reps, time.perf_counter() - t0)
        return elapsed / reps, prof


def check_golden(out48, n, first_index, what):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "kzg_vectors.json")))
    for rec in golden["blobs"]:
        b = rec["index"] - first_index
        if 0 <= b < n and what in rec:
            assert out48[48 * b:48 * b + 48].hex() == rec[what], "GPU %s != oracle golden vector (blob %d)" % (what, rec["index"])


def verify_cpu_baseline(R, d_blobs, d_com, d_prf, n):
    """cpu_baseline for verify_blob_kzg_proof_batch: C port of the reference's verify path on the first 32 triples"""
    try:
        from oracle.cport import binding

        m = min(32, n)
        hb = d_blobs[: m * BYTES_PER_BLOB].cpu().numpy().tobytes()
        hc = d_com[: m * 48].cpu().numpy().tobytes()
        hp = d_prf[: m * 48].cpu().numpy().tobytes()
        return binding.time_verify(R.setup_path, hb, hc, hp, m)
    except Exception as err:  # noqa: BLE001
        return {"value": None, "error": repr(err)}


def extra_workloads(R, d_blobs, d_com, n):
    """BASELINE.json configs[2] and configs[3] as secondary numbers of the default run (not `value`):
    compute_blob_kzg_proof on the same resident blobs, and verify_blob_kzg_proof_batch on 65,536 DISTINCT resident
    (blob, commitment, proof) triples produced by the engine itself (every triple valid: the batch verifies true)."""
    torch, setup = R.torch, R.setup
    out = {}
    # the headline workload once more with TWO calls in flight (two streams, result buffers of their own): the second call's bit-plane
    # transposition runs beside the first one's MSM and the first one's lane-sum trees beside the second one's MSM.  Not `value`: the
    # timed region keeps one call at a time so that the MSM kernel's event-timed duration (roofline.achieved) is not stretched by a
    # neighbour waiting for its slots.
    lanes_c = R.lanes(2, n)
    tick_c = [0]

    def two_commits():
        st, raw, o, stt = lanes_c[tick_c[0] % 2]
        tick_c[0] += 1
        with torch.cuda.stream(st):
            R.commit(d_blobs, n, o, stt, raw)

    dtc, _ = R.measure(two_commits, 8, warm=4)
    for _, _, o, stt in lanes_c:
        assert int(stt.abs().sum()) == 0 and torch.equal(o, d_com), "two calls in flight must not change a byte"
    out["blob_to_kzg_commitment_two_calls_in_flight"] = {"blobs_per_s": n / dtc, "ms_per_batch": 1e3 * dtc}
    del lanes_c
    d_prf, d_st = R.prove(d_blobs, d_com, n)
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    check_golden(d_prf.cpu().numpy().tobytes(), n, 0, "proof")
    dt, prof = R.measure(lambda: R.prove(d_blobs, d_com, n, d_prf, d_st), 3)
    out["compute_blob_kzg_proof"] = {"workload": "batch=%d blobs resident in HBM (BASELINE configs[2])" % n, "blobs_per_s": n / dt, "ms_per_batch": 1e3 * dt,
                                     "algorithmic_GBps": n * ALG_BYTES["proof"] / dt / 1e9, "roofline": roofline_object("proof", n, prof, setup.window_bits)}
    # the same with THREE calls in flight (three streams, result buffers of their own; the context has three workspaces): a call
    # starts with ~5 ms in which the chip is nearly idle -- one SHA-256 stream per blob, then the quotients -- and no stream can order
    # a call's hash before its own inputs; a caller that keeps batches in flight has the next calls' preparation run in the shadow of
    # the current call's MSM
    lanes = R.lanes(3, n)
    tick = [0]

    def in_flight():
        st, raw, o, stt = lanes[tick[0] % 3]
        tick[0] += 1
        with torch.cuda.stream(st):
            R.prove(d_blobs, d_com, n, o, stt, raw)

    dt2, _ = R.measure(in_flight, 21, warm=6)
    for _, _, o, stt in lanes:
        assert int(stt.abs().sum()) == 0 and torch.equal(o, d_prf), "calls in flight must not change a byte"
    out["compute_blob_kzg_proof"].update({"blobs_per_s_three_calls_in_flight": n / dt2, "ms_per_batch_three_calls_in_flight": 1e3 * dt2})
    del lanes
    # ---- verify: 65,536 distinct triples
    nv = 65536
    vb = R.make_blobs(nv, 0)
    vc, vs = R.commit(vb, nv)
    vp, vs2 = R.prove(vb, vc, nv)
    torch.cuda.synchronize()
    assert int(vs.abs().sum()) == 0 and int(vs2.abs().sum()) == 0
    ok = R.verify(vb, vc, vp, nv, 0, nv)
    assert ok is True, "verify_blob_kzg_proof_batch must accept the engine's own proofs"
    dt, prof = R.measure(lambda: R.verify(vb, vc, vp, nv, 0, nv), 3)
    rec = {"workload": "batch=%d distinct (blob, commitment, proof) triples resident in HBM, includes the host pairing (BASELINE configs[3])" % nv,
           "blobs_per_s": nv / dt, "ms_per_batch": 1e3 * dt, "result": bool(ok), "algorithmic_GBps": nv * ALG_BYTES["verify"] / dt / 1e9,
           "hbm_frac_of_8TBps": nv * ALG_BYTES["verify"] / dt / 8e12, "roofline": roofline_object("verify", nv, prof, setup.window_bits, call_ms=1e3 * dt)}
    # two calls in flight from two host threads (a call is synchronous: it returns the boolean): one call's tail -- bucket chains,
    # read-backs, the host's Horner loops and pairing, ~2 ms on a nearly idle chip -- runs beside the other call's hash
    import concurrent.futures

    vstreams = [torch.cuda.Stream(device=R.dev) for _ in range(2)]

    def one_verify(k):
        torch.cuda.set_device(R.local_dev)
        return R.verify(vb, vc, vp, nv, 0, nv, vstreams[k % 2].cuda_stream)

    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:
        assert all(pool.map(one_verify, range(4)))  # warm-up: the second pooled session, both streams
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oks = list(pool.map(one_verify, range(10)))
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 10
    assert all(v is True for v in oks)
    rec.update({"blobs_per_s_two_calls_in_flight": nv / dt2, "ms_per_batch_two_calls_in_flight": 1e3 * dt2})
    rec["cpu_baseline"] = verify_cpu_baseline(R, vb, vc, vp, nv)
    # a corrupted proof must flip the result
    saved = vp[48 * 40000:48 * 40001].clone()
    vp[48 * 40000:48 * 40001] = vp[0:48]
    torch.cuda.synchronize()
    rec["rejects_corrupted_batch"] = R.verify(vb, vc, vp, nv, 0, nv) is False
    vp[48 * 40000:48 * 40001] = saved
    out["verify_blob_kzg_proof_batch"] = rec
    # single-item latencies (BASELINE configs[0] shape: one blob, as benches/kzg.rs:35-43 times them)
    lat = {}
    for name, fn in (("blob_to_kzg_commitment", lambda: R.commit(d_blobs, 1, d_com, d_st)), ("compute_blob_kzg_proof", lambda: R.prove(d_blobs, d_com, 1, d_prf, d_st)),
                     ("verify_blob_kzg_proof", lambda: R.verify(d_blobs, d_com, d_prf, 1, 0, 1))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
            torch.cuda.synchronize()
        lat[name] = 1e3 * (time.perf_counter() - t0) / 5
    out["single_blob_latency_ms"] = lat
    return out


def run_rank(args, rank, local_rank, world):
    R = Rank(args, rank, local_rank, world)
    torch, setup = R.torch, R.setup
    wl = args.workload
    n = args.batch or DEFAULT_BATCH[wl]
    first = rank * n
    d_blobs = R.make_blobs(n, first)
    d_out = torch.empty(n * 48, dtype=torch.uint8, device=R.dev)
    d_status = torch.empty(n, dtype=torch.int32, device=R.dev)
    gathered = torch.empty(world * n * 48, dtype=torch.uint8, device=R.dev) if R.use_dist else None
    verdicts = []

    flight = args.in_flight or 1  # one call at a time for every workload: what the reference's methods and its criterion bench are
    lanes = R.lanes(flight, n) if wl != "verify" else []
    tick = [0]

    def on_lane(fn):
        """step i on lane i mod F: its own stream (the gather rides on it too) and its own result buffers"""
        st, raw, o, stt = lanes[tick[0] % flight]
        tick[0] += 1
        if flight == 1:
            return fn(raw, d_out, d_status)
        with torch.cuda.stream(st):
            return fn(raw, o, stt)

    if wl == "commit":
        def step():
            def one(raw, o, stt):
                R.commit(d_blobs, n, o, stt, raw)
                R.gather48(o, gathered)
            on_lane(one)
    elif wl == "proof":
        d_com, _ = R.commit(d_blobs, n)
        torch.cuda.synchronize()

        def step():
            def one(raw, o, stt):
                R.prove(d_blobs, d_com, n, o, stt, raw)
                R.gather48(o, gathered)
            on_lane(one)
    else:
        d_com, st1 = R.commit(d_blobs, n)
        d_prf, st2 = R.prove(d_blobs, d_com, n)
        torch.cuda.synchronize()
        assert int(st1.abs().sum()) == 0 and int(st2.abs().sum()) == 0
        d_status.zero_()

        if flight > 1 and not R.use_dist:
            # a verification call returns a boolean: it is synchronous, so calls are kept in flight by `flight` HOST threads, each with
            # a stream of its own (ctypes releases the GIL inside the call; sessions are pooled per call); step() hands the next call
            # to the pool and the closing fence of the timed region collects them
            import concurrent.futures

            pool = concurrent.futures.ThreadPoolExecutor(max_workers=flight)
            vstreams = [torch.cuda.Stream(device=R.dev) for _ in range(flight)]
            pending = []

            def one_verify(k):
                torch.cuda.set_device(R.local_dev)
                return R.verify(d_blobs, d_com, d_prf, n, first, world * n, vstreams[k % flight].cuda_stream)

            def step():
                pending.append(pool.submit(one_verify, tick[0]))
                tick[0] += 1

            plain_fence = R.fence

            def fence_and_collect():
                for f in pending:
                    verdicts.append(f.result())
                del pending[:]
                plain_fence()

            R.fence = fence_and_collect
        else:
            def step():
                verdicts.append(R.verify(d_blobs, d_com, d_prf, n, first, world * n))

    elapsed, prof = R.timed(step)
    if flight > 1 and lanes:  # every lane's results: valid, and identical (the same blobs)
        for _, _, o, stt in lanes:
            assert int(stt.abs().sum()) == 0, "synthetic blobs must all be valid"
            assert torch.equal(o, lanes[0][2]), "calls in flight side by side must not change a byte"
        d_out.copy_(lanes[0][2])
        d_status.copy_(lanes[0][3])
    assert int(d_status.abs().sum()) == 0, "synthetic blobs must all be valid"
    if wl == "verify":
        assert all(v is True for v in verdicts), "verify_blob_kzg_proof_batch rejected the engine's own proofs"

    # ---- correctness spot check of the timed output against the oracle golden vectors
    gpu_out = d_out.cpu().numpy().tobytes() if rank == 0 else b""
    if rank == 0 and wl != "verify":
        check_golden(gpu_out, n, first, "commitment" if wl == "commit" else "proof")
    if R.use_dist and wl != "verify":  # rank-ordered gather = global blob order
        assert bytes(gathered[rank * n * 48:(rank + 1) * n * 48].cpu().numpy().tobytes()) == d_out.cpu().numpy().tobytes()

    result = {
        "metric": METRIC[wl],
        "value": world * n * args.steps / elapsed,
        "unit": "blobs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic: element(b,i)=SHA-256(seed||b||i) mod r, generated on device, resident in HBM",
        "config": {
            "workload": {"commit": "blob_to_kzg_commitment batch=%d blobs per GPU (BASELINE configs[1])",
                         "proof": "compute_blob_kzg_proof batch=%d blobs per GPU (BASELINE configs[2])",
                         "verify": "verify_blob_kzg_proof_batch batch=%d (blob, commitment, proof) triples per GPU, host pairing included (BASELINE configs[3])"}[wl] % n
            + (" -- configs[4] shape: %d blobs over %d GPUs" % (world * n, world) if (wl == "commit" and world * n >= 1 << 20) else ""),
            "blobs_per_gpu": n,
            "calls_in_flight": flight,
            "window_bits": setup.window_bits,
            "table_gib": setup.table_bytes / 2**30,
            "parallelism": "blob-sharded x%d, %s" % (world, "RCCL all-gather of 48-B results" if wl != "verify" else "all-gather of 32-B transcript roots + 192-B partial sums, one pairing"),
            "backend": args.backend if R.use_dist else None,
            "setup_s": R.t_setup,
            "first_use_s": R.t_first_use,
            "first_use_table_class": R.first_use_class,
            "ctx_create_returned_s": R.t_create,
            "full_table_s": R.t_setup,
            "plane_groups": setup.plane_groups,
            "hbm_plan_gib": {k: round(v / GIB, 2) for k, v in R.plan.items() if k in ("table", "blobs", "workspace", "resident_total", "peak_total", "hbm_total")},
        },
    }
    if wl == "proof" and flight == 1 and not R.use_dist and not args.no_extra:  # (the PMC child passes run --no-extra: only the counted calls)
        # beside the per-call figure (`value`): the amortised rate of a caller that keeps three calls in flight on three streams
        lanes3 = R.lanes(3, n)
        tick3 = [0]

        def in_flight3():
            st, raw, o, stt = lanes3[tick3[0] % 3]
            tick3[0] += 1
            with torch.cuda.stream(st):
                R.prove(d_blobs, d_com, n, o, stt, raw)

        dt3, _ = R.measure(in_flight3, max(6, 3 * ((args.steps + 2) // 3)), warm=6)
        for _, _, o, stt in lanes3:
            assert int(stt.abs().sum()) == 0 and torch.equal(o, d_out), "calls in flight must not change a byte"
        result["value_three_calls_in_flight"] = n / dt3
        result["ms_per_step_three_calls_in_flight"] = 1e3 * dt3
        del lanes3
    if rank == 0:
        roof = roofline_object(wl, n, prof, setup.window_bits, call_ms=1e3 * elapsed / args.steps)
        if wl in ("commit", "proof") and roof and prof["msm_launches"]:
            roof["kernel"] = setup.msm_kernel_name
            # measured integer-ALU ceiling: dependent Fp Montgomery multiplies with the multiply of the MSM kernel
            # (radix-2^28 limbs), 8 waves/SIMD, whole chip.  DESIGN.md section 5 gives the instruction counts.
            lanes = 256 * 4 * 64 * 8
            setup.microbench_fp_mul(lanes, 200)
            peak = lanes * 2000 / (setup.microbench_fp_mul(lanes, 2000) * 1e-3)
            msm_ms = prof["msm_ms"] / prof["msm_launches"]
            blobs_per_launch = n * args.steps / prof["msm_launches"]
            roof["adds_per_blob"] = prof["adds_per_blob"]
            roof["table_gather_bytes_per_blob"] = prof["adds_per_blob"] * 96
            # multiply-equivalents per mixed add, weighted by v_mad_u64_u32 count: 6 products + 2 squarings (301/392 each)
            # + 2 products sharing one reduction (588/392) = 9.04
            roof["valu_fp_mul_per_s"] = prof["adds_per_blob"] * 9.04 * blobs_per_launch / (msm_ms * 1e-3)
            roof["valu_fp_mul_peak_per_s"] = peak
            roof["valu_frac"] = roof["valu_fp_mul_per_s"] / peak
        result["roofline"] = roof
        if wl == "commit" and not args.no_extra and world == 1:
            try:
                result["extra"] = extra_workloads(R, d_blobs, d_out, n)
                # BASELINE.json's metric names TWO functions: the second one at top level too, with its own roofline fraction
                v, p = result["extra"]["verify_blob_kzg_proof_batch"], result["extra"]["compute_blob_kzg_proof"]
                # BASELINE.json's metric string names both functions; `value` is the first (blob_to_kzg_commitment), `values` carries both
                result["metric"] = "blobs/sec for blob_to_kzg_commitment and verify_blob_kzg_proof_batch (n=4096)"
                result["values"] = {"blob_to_kzg_commitment": result["value"], "verify_blob_kzg_proof_batch": v["blobs_per_s"], "compute_blob_kzg_proof": p["blobs_per_s"],
                                    "blob_to_kzg_commitment_two_calls_in_flight": result["extra"].get("blob_to_kzg_commitment_two_calls_in_flight", {}).get("blobs_per_s"),
                                    "compute_blob_kzg_proof_three_calls_in_flight": p.get("blobs_per_s_three_calls_in_flight"),
                                    "verify_blob_kzg_proof_batch_two_calls_in_flight": v.get("blobs_per_s_two_calls_in_flight"), "unit": "blobs/s", "note": "`value` = blob_to_kzg_commitment at batch 4,096 (configs[1]); verify at batch 65,536 (configs[3]); proof at 4,096 (configs[2])"}
                result["secondary_metrics"] = [
                    {"metric": METRIC["verify"], "value": v["blobs_per_s"], "unit": "blobs/s", "ms_per_step": v["ms_per_batch"], "workload": v["workload"],
                     "roofline_frac": v["roofline"]["frac"] if v.get("roofline") else None, "result": v["result"],
                     "value_two_calls_in_flight": v.get("blobs_per_s_two_calls_in_flight"), "ms_per_step_two_calls_in_flight": v.get("ms_per_batch_two_calls_in_flight")},
                    {"metric": METRIC["proof"], "value": p["blobs_per_s"], "unit": "blobs/s", "ms_per_step": p["ms_per_batch"], "workload": p["workload"],
                     "roofline_frac": p["roofline"]["frac"] if p.get("roofline") else None, "value_three_calls_in_flight": p.get("blobs_per_s_three_calls_in_flight"),
                     "ms_per_step_three_calls_in_flight": p.get("ms_per_batch_three_calls_in_flight")}]
            except Exception as err:  # secondary numbers never hide the headline
                result["extra"] = {"error": repr(err)}
        if not args.no_cpu_baseline and world == 1:
            try:
                if wl == "verify":
                    result["cpu_baseline"] = verify_cpu_baseline(R, d_blobs, d_com, d_prf, n)
                else:
                    commits = gpu_out if wl == "commit" else d_com.cpu().numpy().tobytes()
                    result["cpu_baseline"] = cpu_baseline(args.cpu_sample, R.setup_path, commits)
                    if wl == "proof":
                        result["cpu_baseline"]["note"] = "commitment path of the C port (the proof's second MSM has the same cost); no separate proof port is timed"
            except Exception as err:  # the baseline is reporting only; never hide the GPU number
                result["cpu_baseline"] = {"value": None, "error": repr(err)}
        live_wanted = world == 1 and not args.no_live_traffic and roof is not None
        if live_wanted:
            # the issue interval and shader clock the instruction counts are priced with, measured on this box before the context goes
            issue = issue_interval(setup)
            simds = 4 * torch.cuda.get_device_properties(R.local_dev).multi_processor_count
            # the PMC child passes need the card: release this process's 192-GiB context and caches first
            del d_blobs, d_out, d_status
            if wl != "commit":
                del d_com
            if wl == "verify":
                del d_prf
            setup.close()
            torch.cuda.empty_cache()
            pmc = live_pmc(args, wl, n, ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"))
            live = traffic_from_pmc(pmc, wl)
            if "hbm_bytes_per_launch" in live:
                roof["traffic"] = live["hbm_bytes_per_launch"]
                roof["traffic_source"] = "live: " + live["how"]
                roof["traffic_detail"] = live
                roof["traffic_over_algorithmic"] = live["hbm_bytes_per_launch"] / (ALG_BYTES[wl] * roof["blobs_per_launch"])
            else:
                roof["traffic_live_error"] = live.get("error")
            roof["valu_issue"] = valu_issue_object(pmc, wl, issue, simds, 1e3 * elapsed / args.steps, roof.get("kernel_ms"), extra_calls=1 if wl == "proof" else 0,
                                                   run_clock=roof.get("shader_clock_ghz_during_run"))
            # the two other workloads of the default run: one SQ_INSTS_VALU pass each, priced the same way
            extra = result.get("extra") if isinstance(result.get("extra"), dict) else None
            if extra and "error" not in extra:
                for name, w2, n2 in (("compute_blob_kzg_proof", "proof", n), ("verify_blob_kzg_proof_batch", "verify", 65536)):
                    rec = extra.get(name)
                    if not rec or not rec.get("roofline"):
                        continue
                    pmc2 = live_pmc(args, w2, n2, ("SQ_INSTS_VALU",))
                    rec["roofline"]["valu_issue"] = valu_issue_object(pmc2, w2, issue, simds, rec["ms_per_batch"], rec["roofline"].get("kernel_ms"),
                                                                      extra_calls=1 if w2 == "proof" else 0, run_clock=rec["roofline"].get("shader_clock_ghz_during_run"))
                    flatten_valu_issue(rec["roofline"])
                    rec["roofline"] = ordered_roofline(rec["roofline"])
                for sm in result.get("secondary_metrics", []):
                    key = "verify_blob_kzg_proof_batch" if "verify" in sm["metric"] else "compute_blob_kzg_proof"
                    r2 = extra.get(key, {}).get("roofline", {})
                    sm.update({"valu_issue_frac": r2.get("valu_issue_frac"), "valu_issue_floor_ms": r2.get("valu_issue_floor_ms"),
                               "valu_issue_kernel_frac": r2.get("valu_issue_kernel_frac"), "shader_clock_ghz": r2.get("shader_clock_ghz")})
                # the other two workloads' bound figures as scalars of the HEADLINE roofline too (a reader of the first-level record
                # sees how far each of BASELINE's three workloads is from its own issue floor)
                rv, rp = (extra.get(k, {}).get("roofline") or {} for k in ("verify_blob_kzg_proof_batch", "compute_blob_kzg_proof"))
                roof.update({"verify_valu_issue_frac": rv.get("valu_issue_frac"), "verify_frac": rv.get("frac"), "proof_valu_issue_frac": rp.get("valu_issue_frac"),
                             "proof_frac": rp.get("frac"),
                             "verify_longest_kernel_by_duration": "%s %.2f ms (beside k_eval_frac on a second stream); `kernel` of the verify roofline = %s %.2f ms, alone on the chip"
                             % (rv.get("longest_kernel_by_duration"), rv.get("longest_kernel_ms") or 0.0, rv.get("dominant_kernel_by_bytes"), rv.get("kernel_ms") or 0.0)
                             if rv.get("longest_kernel_by_duration") else None,
                             "verify_valu_issue_note": "the floor counts WAVE instructions: until the balanced bucket kernel (round 5, k_var_buckets_seg) it contained the "
                             "idle lanes of one-thread-per-bucket waves -- 11.9 ms then, 11.2 ms now, while the call itself got 0.2-0.3 ms shorter (DESIGN.md 5.4)"})
            flatten_valu_issue(roof)
        if roof is not None:
            result["roofline"] = ordered_roofline(roof)
        print(json.dumps(result), flush=True)
    setup.close()
    if R.use_dist:
        R.dist.barrier()
        R.dist.destroy_process_group()


def run_group(args):
    """`--group N`: the drop-in's multi-GPU path (Setup::load_json creates ONE context over the node's
    GPUs) measured the way `--gpus N` measures the one-process-per-GPU path: weak scaling, --batch items per member, inputs resident
    on each member's device, K steps between two full synchronisations of every device."""
    import torch

    import kateth_amd

    wl, S = args.workload, args.group
    n = args.batch or DEFAULT_BATCH[wl]
    ndev = torch.cuda.device_count()
    assert torch.cuda.is_available() and ndev >= 1, "bench.py needs an MI355X; the engine has no CPU fallback"
    devices = list(range(S)) if ndev >= S else [0] * S
    setup_path = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
    t0 = time.time()
    # members that share a card share its HBM: an explicit class that fits S times (class 16 = 12.9 GB each) unless the caller chose
    wb = args.window_bits or (0 if len(set(devices)) == S else 16)
    setup = kateth_amd.Setup.load_json(setup_path, window_bits=wb, devices=devices, table_max=not args.default_budget and len(set(devices)) == S)
    setup.wait_ready()
    t_setup = time.time() - t0
    members = [setup.member(k) for k in range(S)]

    def on(k):
        return torch.device("cuda", devices[k])

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    blobs, com, prf, out, stat = [], [], [], [], []
    for k in range(S):
        with torch.cuda.device(devices[k]):
            b = torch.empty(n * BYTES_PER_BLOB, dtype=torch.uint8, device=on(k))
            members[k].synth_blobs_dev(SEED, k * n, n, b.data_ptr(), torch.cuda.current_stream(devices[k]).cuda_stream)
            blobs.append(b)
            out.append(torch.zeros(n * 48, dtype=torch.uint8, device=on(k)))
            stat.append(torch.zeros(n, dtype=torch.int32, device=on(k)))
    sync_all()
    ptr = lambda ts: [t.data_ptr() for t in ts]  # noqa: E731
    counts = [n] * S
    streams = []
    for k in range(S):
        with torch.cuda.device(devices[k]):
            streams.append(torch.cuda.Stream(device=on(k)))
    raw_streams = [st.cuda_stream for st in streams]
    verdicts = []
    if wl != "commit":  # the inputs of the timed calls come from the same group calls
        com = [torch.zeros(n * 48, dtype=torch.uint8, device=on(k)) for k in range(S)]
        setup.blob_to_commitment_batch_group_dev(ptr(blobs), counts, ptr(com), ptr(stat), raw_streams)
        sync_all()
    if wl == "verify":
        prf = [torch.zeros(n * 48, dtype=torch.uint8, device=on(k)) for k in range(S)]
        setup.compute_blob_proof_batch_group_dev(ptr(blobs), ptr(com), counts, ptr(prf), ptr(stat), raw_streams)
        sync_all()
    assert all(int(t.abs().sum()) == 0 for t in stat), "synthetic blobs must all be valid"

    def step():
        if wl == "commit":
            setup.blob_to_commitment_batch_group_dev(ptr(blobs), counts, ptr(out), ptr(stat), raw_streams)
        elif wl == "proof":
            setup.compute_blob_proof_batch_group_dev(ptr(blobs), ptr(com), counts, ptr(out), ptr(stat), raw_streams)
        else:
            verdicts.append(setup.verify_blob_proof_batch_group_dev(ptr(blobs), ptr(com), ptr(prf), counts, raw_streams))

    for _ in range(args.warmup):
        step()
    sync_all()
    members[0].profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = members[0].profile_end()
    prof["calls"] = args.steps
    assert all(int(t.abs().sum()) == 0 for t in stat) and all(v is True for v in verdicts)
    if wl != "verify":  # member 0's share holds the golden blobs
        check_golden(out[0].cpu().numpy().tobytes(), n, 0, "commitment" if wl == "commit" else "proof")
        # and the last member's share against the single-device call of the same member
        ref = torch.zeros_like(out[-1])
        st2 = torch.zeros_like(stat[-1])
        with torch.cuda.device(devices[-1]):
            if wl == "commit":
                members[-1].blob_to_commitment_batch_dev(blobs[-1].data_ptr(), n, ref.data_ptr(), st2.data_ptr(), raw_streams[-1])
            else:
                members[-1].compute_blob_proof_batch_dev(blobs[-1].data_ptr(), com[-1].data_ptr(), n, ref.data_ptr(), st2.data_ptr(), raw_streams[-1])
        sync_all()
        assert torch.equal(ref, out[-1]), "group call != single-device call on the last member's share"
    roof = roofline_object(wl, n, prof, members[0].window_bits, call_ms=1e3 * elapsed / args.steps)
    if roof is not None:
        roof["scope"] = "member 0's kernels (one of %d members); achieved = algorithmic bytes of ONE member's share / its kernel time" % S
        roof = ordered_roofline(roof)
    result = {
        "metric": METRIC[wl], "value": S * n * args.steps / elapsed, "unit": "blobs/s", "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
        "data": "synthetic: element(b,i)=SHA-256(seed||b||i) mod r, generated on each member's device, resident in its HBM",
        "config": {"workload": "%s batch=%d items per member, GROUP context of %d members in one process (kzg_*_group_dev, device-resident shares)"
                               % ({"commit": "blob_to_kzg_commitment", "proof": "compute_blob_kzg_proof", "verify": "verify_blob_kzg_proof_batch"}[wl], n, S),
                   "blobs_per_gpu": n, "members": S, "member_devices": devices, "members_share_a_card": len(set(devices)) != S, "window_bits": members[0].window_bits,
                   "table_gib_per_member": members[0].table_bytes / 2**30, "plane_groups": members[0].plane_groups, "setup_s": t_setup,
                   "parallelism": "blob-sharded x%d inside one process behind the C ABI: no collective; verify: the members' 32-B roots seed one challenge, 192-B partial sums, "
                                  "one pairing" % S},
        "roofline": roof,
        "cpu_baseline": None,
        "note": "cpu_baseline and the live PMC passes belong to the default single-GPU run (python bench.py); this mode measures the group path",
    }
    print(json.dumps(result), flush=True)
    setup.close()


def main():
    args = parse()
    if args.dry_run:
        return dry_run(args)
    if args.group:
        return run_group(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus > 1 or args.spawn:
            return launch_ranks(args)  # before any torch / HIP import in this process
        return run_rank(args, 0, 0, 1)
    world = int(env_world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    run_rank(args, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world)


if __name__ == "__main__":
    main()
